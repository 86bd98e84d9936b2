"""RCCL rehearsal on a one-GPU box: a one-rank "nccl" process group (that IS RCCL on ROCm) driven through the very call forms dp.GradSync
uses under data parallelism -- in-place reduce_scatter_tensor / all_gather_into_tensor on slices of the flat arena, asynchronous
all-reduce of a slice, work.wait() from a side stream, the list form of all_gather, broadcast -- so that an API misuse shows up before
the driver's first multi-GPU run.  Values are those of a world of one (every collective is the identity); what is checked is that RCCL
accepts the calls, leaves the data alone and orders them with the streams.  Prints per-call host + device times.
    python tools/rccl_rehearsal.py"""
import os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd.dp import GradSync

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n = 61 * 1024 * 1024 // 4 * 4
flat = torch.randn(n, device=dev)
ref = flat.clone()
s = GradSync(flat=flat, sharded=True)
s.dist, s.world, s.rank, s.backend = dist, 1, 0, dist.get_backend()      # (a world of one never attaches by itself; or VMMT_DP_FORCE=1)
side = torch.cuda.Stream(device=dev)
segs = [(0, 1 << 20), (1 << 20, 15 * (1 << 20)), (15 * (1 << 20), n)]


def timed(name, fn, reps=5):
    torch.cuda.synchronize()
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    host = (time.perf_counter() - t0) / reps
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / reps
    print("%-44s host %7.1f us   complete %8.1f us" % (name, host * 1e6, tot * 1e6), flush=True)


def rs():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        works = [s.reduce_scatter(flat, lo, hi) for lo, hi in segs]
    for w in works:
        w.wait()


def ag():
    with torch.cuda.stream(side):
        works = [s.all_gather(flat, lo, hi) for lo, hi in segs]
        for w in works:
            w.wait()
    torch.cuda.current_stream().wait_stream(side)


def ar():
    works = [dist.all_reduce(flat[lo:hi], async_op=True) for lo, hi in segs]
    for w in works:
        w.wait()


def seg_on_comm():
    # the form the backward plan uses since round 4: a synchronous collective on the COMM stream behind its producer stream
    for lo, hi in segs:
        s.reduce_segment(flat, lo, hi, torch.cuda.current_stream())
    torch.cuda.current_stream().wait_stream(s.comm_stream())


assert s._probe("reduce_scatter", flat) and s._probe("all_gather", flat), "RCCL lacks the tensor collectives?"
timed("reduce_segment x3 (COMM stream, synchronous ops)", seg_on_comm)
timed("reduce_scatter_tensor x3 segments (in place)", rs)
timed("all_gather_into_tensor x3 segments (in place)", ag)
timed("all_reduce x3 segments (async)", ar)
rows = s.all_gather_rows(torch.arange(8, device=dev, dtype=torch.float32))
assert rows.shape == (1, 8) and torch.equal(rows[0], torch.arange(8, device=dev, dtype=torch.float32))
k = torch.ones(1, device=dev)
dist.all_reduce(k, async_op=True).wait()
dist.broadcast(flat[:4096], 0, async_op=True).wait()
torch.cuda.synchronize()
assert torch.equal(flat, ref) and float(k) == 1.0
print("rccl rehearsal ok: world 1, %d MB arena, values untouched" % (n * 4 >> 20))
dist.destroy_process_group()
