"""microbenchmark of the lazy embedding-table kernels alone on the chip (csrc/optim.hip): the catch-up of a batch's rows by the number of
replayed steps, the update with / without rolling rows, against the dense Adam kernel over the same table.  python tools/lazy_rows_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd import _lib as L  # noqa: E402

lib = L.lib()
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
R, C, NTOK = int(os.environ.get("R", 30000)), int(os.environ.get("C", 500)), int(os.environ.get("NTOK", 5120))
g = torch.Generator().manual_seed(1)
p = torch.rand(R, C, device=dev) - 0.5
m, v, gr = torch.rand(R, C, device=dev) * 1e-3, torch.rand(R, C, device=dev) * 1e-6, torch.rand(R, C, device=dev) * 1e-3
flags = torch.zeros(2 * R, dtype=torch.int32, device=dev)
last = torch.zeros(R, dtype=torch.int32, device=dev)
hist = torch.zeros(L.LAZY_HIST_WORDS, dtype=torch.int32, device=dev)
sq = torch.zeros(L.SUMSQ_SCRATCH, dtype=torch.float32, device=dev)
rowsq = torch.zeros(R, dtype=torch.float32, device=dev)
ids = torch.randint(0, R, (NTOK,), generator=g).to(dev)
junk = torch.empty(64 << 20, device=dev)          # 256 MB: pushes the table out of the caches between timed launches


def timeit(fn, prep, n=12):
    ts = []
    for _ in range(n):
        prep()
        junk.zero_()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


STEP = 40
# a ring with STEP recorded steps
hv = hist[4:].view(torch.float32).view(L.LAZY_HIST, 4)
for s in range(1, STEP + 1):
    hv[s % L.LAZY_HIST, 0], hv[s % L.LAZY_HIST, 1], hv[s % L.LAZY_HIST, 2] = 0.002, 1.0, 1.0
    hist[4 + 4 * (s % L.LAZY_HIST) + 3] = s


def prep_catchup(behind):
    def f():
        hist[0] = STEP
        last.fill_(STEP - behind)
        flags.zero_()
        L.check(lib.vmmt_rows_mark(ids.data_ptr(), NTOK, flags.data_ptr(), R, hist.data_ptr(), st), "mark")
    return f


def catchup(mode=0):
    L.check(lib.vmmt_rows_catchup(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), R, C, flags.data_ptr(), last.data_ptr(), hist.data_ptr(),
                                  0.9, 0.999, 1e-9, mode, st), "catchup")


print("table %d x %d, %d ids -> %d distinct rows" % (R, C, NTOK, int(torch.unique(ids).numel())))
print("mark %.1f us" % timeit(lambda: lib.vmmt_rows_mark(ids.data_ptr(), NTOK, flags.data_ptr(), R, hist.data_ptr(), st), lambda: None))
for behind in (0, 1, 4, 8, 16):
    print("catch-up of the flagged rows, %2d steps behind: %.1f us" % (behind, timeit(catchup, prep_catchup(behind))))
print("flush (every row, 8 behind): %.1f us" % timeit(lambda: catchup(1), prep_catchup(8)))
for roll, behind in ((0, 0), (16, 15), (16, 4), (8, 7)):
    def prep():
        prep_catchup(0)()
        last.fill_(STEP - behind)
        last[torch.unique(ids)] = STEP
    print("update: flagged rows + rolling 1/%d rows %d behind: %.1f us" % (roll, behind, timeit(
        lambda: L.check(lib.vmmt_adam_rows_step(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), R, C, flags.data_ptr(), last.data_ptr(),
                                                hist.data_ptr(), 0.002, 0.9, 0.999, 1e-9, STEP + 1, roll, 5.0, sq.data_ptr(), 1.0, None, st), "rows"), prep)))
print("norm of the flagged rows: %.1f us" % timeit(lambda: lib.vmmt_sumsq_rows(gr.data_ptr(), R, C, flags.data_ptr(), hist.data_ptr(), rowsq.data_ptr(),
                                                                                sq.data_ptr(), 3, st), prep_catchup(0)))
print("dense Adam over the table: %.1f us" % timeit(lambda: lib.vmmt_adam_step(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), R * C, 0.002, 0.9, 0.999,
                                                                              1e-9, STEP + 1, 5.0, sq.data_ptr(), 1.0, 0, None, None, st), lambda: None))
print("dense norm over the table: %.1f us" % timeit(lambda: lib.vmmt_sumsq(gr.data_ptr(), R * C, sq.data_ptr(), 0, st), lambda: None))
