"""What a cross-stream wait costs the stream that waits, when the event it waits for is long complete on the device but was NOT complete
when the host enqueued the wait (the host runs ahead): N x [small kernel ; wait(event of the other stream) ; small kernel] against
N x [small kernel ; small kernel].  python tools/probe/wait_cost.py"""
import time
import torch

dev = torch.device("cuda:0")
x = torch.zeros(1024, device=dev)
y = torch.zeros(1 << 26, device=dev)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
N = 200


def run(with_wait, fresh):
    evs = []
    torch.cuda.synchronize()
    with torch.cuda.stream(s0):
        y.add_(1.0); y.add_(1.0); y.add_(1.0)             # ~0.3 ms of device work in front: the host enqueues everything below meanwhile
    with torch.cuda.stream(s1):
        for i in range(N if fresh else 1):
            x.add_(1.0)
            e = torch.cuda.Event(); e.record(s1); evs.append(e)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s0):
        t0.record(s0)
        for i in range(N):
            x.mul_(1.0)
            if with_wait:
                s0.wait_event(evs[i if fresh else 0])
            x.mul_(1.0)
        t1.record(s0)
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e3 / N


for _ in range(2):
    a = run(False, False); b = run(True, False); c = run(True, True)
print("per iteration of two small kernels: no wait %.2f us | wait for ONE old event %.2f us | wait for a different (complete) event each time %.2f us" % (a, b, c))
