import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_ab import bench, L
V = [3564, 2567, 2564, 2566]
bench("dWg NN 30000x512x5120", L.GEMM_NN, 30000, 512, 5120, V, out_f32=1)
bench("dO  TN 5120x512x30000 s6", L.GEMM_TN, 5120, 512, 30000, V, out_f32=1, split=6)
bench("sq  NT 8192x8192x8192", L.GEMM_NT, 8192, 8192, 8192, V, out_f32=0, rounds=3)
bench("sq  NT 4096x4096x4096", L.GEMM_NT, 4096, 4096, 4096, V, out_f32=0, rounds=3)
