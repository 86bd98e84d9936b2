"""Achieved HBM bandwidth of the path's memory-bound kernels, alone on the chip (HIP events around 20 launches each, after warm-up;
bytes = the kernel's ALGORITHMIC traffic), against the 8 TB/s HBM3E peak of MI355X_MICROARCH.md:

  * vmmt_gather_rows        the image-row gather from the HBM-resident feature table (TrainerMultimodal.py:632-639): 8 KB contiguous per
                            triplet -- at the benchmark's 256 rows (2 MB: latency) and at 65 536 rows (0.5 GB read + 0.5 GB written)
  * vmmt_standardise_rows   (x - mean) / std over a resident table, in place (train_mm_vi_model1.py:499-501): 8 B per element
  * vmmt_sumsq              the clip norm: 4 B per parameter
  * vmmt_adam_step          clip + Adam: 28 B per parameter (p, m, v read + written, g read)
  * vmmt_zero_multi         gradient clearing: 4 B per element written
    python tools/hbm_kernels.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd import _lib as L

lib = L.lib()
dev = torch.device("cuda", 0)
PEAK = 8000.0


def timed(name, nbytes, fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    gbs = nbytes / (us * 1e-6) / 1e9
    print("%-62s %9.1f us  %8.1f MB  %7.0f GB/s  %5.2f of peak" % (name, us, nbytes / 1e6, gbs, gbs / PEAK), flush=True)


D = 2048
N = 290000                                           # BASELINE config 4's table: 2.4 GB
g = torch.Generator(device=dev).manual_seed(1)
table = torch.empty(N, D, device=dev)
for lo in range(0, N, 65536):
    table[lo:lo + 65536].uniform_(0.0, 1.0, generator=g)
for rows in (256, 65536):
    idx = torch.randint(0, N, (rows,), device=dev, generator=g)
    out = torch.zeros(rows, D, device=dev)
    timed("vmmt_gather_rows f32 -> f32, %d rows of 8 KB" % rows, 2 * rows * D * 4,
          lambda: L.check(lib.vmmt_gather_rows(L.F32, table.data_ptr(), D, idx.data_ptr(), out.data_ptr(), D, rows, D, None), "gather"))
mean, std = torch.rand(D, device=dev), torch.rand(D, device=dev) + 0.5
timed("vmmt_standardise_rows, %d x %d table in place" % (N, D), 2 * N * D * 4,
      lambda: L.check(lib.vmmt_standardise_rows(table.data_ptr(), D, mean.data_ptr(), std.data_ptr(), N, D, None), "standardise"), reps=5)
del table
torch.cuda.empty_cache()
n = 55_500_000 // 4 * 4                              # the optimised arena of BASELINE config 2
p, gr, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
v.abs_()
ss = torch.zeros(L.SUMSQ_SCRATCH, device=dev)
timed("vmmt_sumsq over the %.1f M-parameter arena" % (n / 1e6), 4 * n,
      lambda: L.check(lib.vmmt_sumsq(gr.data_ptr(), n, ss.data_ptr(), 0, None), "sumsq"))
timed("vmmt_adam_step over the arena (clip + Adam)", 28 * n,
      lambda: L.check(lib.vmmt_adam_step(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), n, 0.002, 0.9, 0.999, 1e-9, 7, 5.0, ss.data_ptr(), 1.0, 0,
                                         None, None, None), "adam"))
timed("vmmt_adam_step, 256 workgroups (the background half's grid cap)", 28 * n,
      lambda: L.check(lib.vmmt_adam_step(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), n, 0.002, 0.9, 0.999, 1e-9, 7, 5.0, ss.data_ptr(), 1.0, 256,
                                         None, None, None), "adam"))
arr = (L.ZeroDesc * 1)(L.ZeroDesc(gr.data_ptr(), 4 * n, 0))
tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
timed("vmmt_zero_multi over the gradient arena", 4 * n,
      lambda: L.check(lib.vmmt_zero_multi(tab.data_ptr(), 1, (4 * n + 16383) // 16384, None), "zero"))
print("-- vmmt_adam_step by grid cap (0 = uncapped: 4096 workgroups), 21.8 M parameters (the foreground half) and the whole arena")
for nn in (21_800_000 // 4 * 4, n):
    for cap in (0, 2048, 1024, 768, 512, 384, 256):
        timed("vmmt_adam_step %.1f M parameters, grid cap %d" % (nn / 1e6, cap), 28 * nn,
              lambda: L.check(lib.vmmt_adam_step(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), nn, 0.002, 0.9, 0.999, 1e-9, 7, 5.0, ss.data_ptr(),
                                                 1.0, cap, None, None, None), "adam"))
