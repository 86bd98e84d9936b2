import torch
for M,N,K in ((8192,8192,8192),(30000,512,5120)):
    A=torch.randn(M,K,device="cuda").bfloat16(); B=torch.randn(N,K,device="cuda").bfloat16()
    for _ in range(3): C=torch.matmul(A,B.t())
    torch.cuda.synchronize()
