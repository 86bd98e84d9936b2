"""isolated timing of the step's GEMM shapes by layout (idle chip): finds kernels that are slow by themselves vs slowed by overlap"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_ab import bench, L
V = [0]
bench("enc in  NT 5120x2048x500", L.GEMM_NT, 5120, 2048, 500, V, out_f32=1)
bench("dcat    NN 5120x1024x512", L.GEMM_NN, 5120, 1024, 512, V, out_f32=0)
bench("dR      NN 5120x512x512", L.GEMM_NN, 5120, 512, 512, V, out_f32=0)
bench("emb     NN 5120x500x2048", L.GEMM_NN, 5120, 500, 2048, V, out_f32=1)
bench("dh1v    NN 256x2048x2048 s8", L.GEMM_NN, 256, 2048, 2048, V, out_f32=1, split=8)
bench("dWih d  TN 2048x500x5120 s4", L.GEMM_TN, 2048, 500, 5120, V, out_f32=1, split=4)
bench("dWhh d  TN 2048x512x4864 s4", L.GEMM_TN, 2048, 512, 4864, V, out_f32=1, split=4)
bench("dWo     TN 512x1024x5120 s8", L.GEMM_TN, 512, 1024, 5120, V, out_f32=1, split=8)
bench("dWih e  TN 1024x500x5120 s8", L.GEMM_TN, 1024, 500, 5120, V, out_f32=1, split=8)
bench("dWhh e  TN 1024x256x4864 s16", L.GEMM_TN, 1024, 256, 4864, V, out_f32=1, split=16)
bench("dW2 img TN 2048x2048x256", L.GEMM_TN, 2048, 2048, 256, V, out_f32=1)
bench("dWg     TN 30000x512x5120", L.GEMM_NN, 30000, 512, 5120, V, out_f32=1)
bench("dO      TN 5120x512x30000 s6", L.GEMM_TN, 5120, 512, 30000, V, out_f32=1, split=6)
