"""A/B of the NT (both operands K-contiguous) 128 x 128 GEMM main loops on the step's shapes: 1284 = two-stage 64-deep LDS-DMA loop
(64 KiB, two workgroups per CU), 4284 = three-stage 32-deep loop (48 KiB, three per CU), 64 = 64 x 64 register-staged.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm_ab as G
from variational_mmt_amd import _lib as L
V = [1284, 4284, 64]
G.bench("enc/dec gx NT 5120x2048x500", L.GEMM_NT, 5120, 2048, 500, V, out_f32=1)
G.bench("zx NT 256x2048x256", L.GEMM_NT, 256, 2048, 256, V, out_f32=1)
G.bench("Q NT 5120x512x512", L.GEMM_NT, 5120, 512, 512, V, out_f32=0)
G.bench("AH NT 5120x512x1024", L.GEMM_NT, 5120, 512, 1024, V, out_f32=0)
G.bench("mu_v NT 256x2048x2048", L.GEMM_NT, 256, 2048, 2048, V, out_f32=1)
G.bench("h1v NT 256x2048x256", L.GEMM_NT, 256, 2048, 256, V, out_f32=0)
G.bench("big NT 5120x30000x512", L.GEMM_NT, 5120, 30000, 512, [3564, 4284], out_f32=0, rounds=3)
