"""What an event RECORD between two dependent kernels costs the stream, by event flavour: torch.cuda.Event (hipEventDisableTiming: a
system-scope release at the record) against hipEventDisableTiming | hipEventReleaseToDevice / hipEventDisableSystemFence created through
the HIP runtime itself.  python tools/probe/event_cost.py"""
import ctypes, os
import torch

hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
dev = torch.device("cuda:0")
x = torch.zeros(1024, device=dev)
y = torch.zeros(1 << 26, device=dev)
big = torch.zeros(1 << 24, device=dev)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
N = 200


def mk(flags):
    e = ctypes.c_void_p()
    assert hip.hipEventCreateWithFlags(ctypes.byref(e), flags) == 0
    return e


def run(kind, flags=0, dirty=False):
    evs = [mk(flags) for _ in range(N)] if kind in ("raw", "rawwait") else [torch.cuda.Event() for _ in range(N)]
    torch.cuda.synchronize()
    with torch.cuda.stream(s0):
        y.add_(1.0); y.add_(1.0); y.add_(1.0)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s0):
        t0.record(s0)
        for i in range(N):
            (big if dirty else x).mul_(1.0)
            if kind == "torch":
                evs[i].record(s0)
            elif kind == "raw":
                assert hip.hipEventRecord(evs[i], ctypes.c_void_p(s0.cuda_stream)) == 0
            elif kind == "torchwait":
                evs[i].record(s0); s1.wait_event(evs[i])
            elif kind == "rawwait":
                hip.hipEventRecord(evs[i], ctypes.c_void_p(s0.cuda_stream)); hip.hipStreamWaitEvent(ctypes.c_void_p(s1.cuda_stream), evs[i], 0)
            x.mul_(1.0)
        t1.record(s0)
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e3 / N


for dirty in (False, True):
    for _ in range(2):
        r = [run("none", dirty=dirty), run("torch", dirty=dirty), run("raw", 0x2, dirty), run("raw", 0x2 | 0x40000000, dirty), run("raw", 0x2 | 0x20000000, dirty),
             run("torchwait", dirty=dirty), run("rawwait", 0x2 | 0x40000000, dirty)]
    print(("64-MiB kernel + small kernel" if dirty else "two small kernels") + ": no record %.2f | torch event %.2f | raw DisableTiming %.2f | + ReleaseToDevice %.2f | "
          "+ DisableSystemFence %.2f | torch record + other stream waits %.2f | raw ReleaseToDevice ditto %.2f  (us per iteration)" % tuple(r))
