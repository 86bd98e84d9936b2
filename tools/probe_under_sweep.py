"""What a small kernel on another stream costs while the vocabulary sweep (gen2p_kernel: 240 workgroups, one per CU, every register and
the whole LDS of its CU) is running: the aux stream's chain of the step sits 200 us in its first `act_bwd8_kernel` until the sweep ends
(profiles/r6_step_timeline.txt) where the kernels in front of it take their usual time.  Each candidate is launched `n` times back to back
on a second stream, 20 us behind the sweep's start, and timed with HIP events on that stream.   python tools/probe_under_sweep.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd import _lib as L
lib = L.lib()
H, M, V = 512, 20 * 256, 30000
T = torch.bfloat16
g = torch.Generator().manual_seed(0)
Vp = (V + 255) // 256 * 256
W = (torch.randn(Vp, H, generator=g) * 0.05).to(T).cuda(); W[V:] = 0
O = torch.randn(M, H, generator=g).to(T).cuda()
bias = (torch.randn(Vp, generator=g) * 0.1).cuda()
y = torch.randint(2, V, (M,), generator=g).cuda()
tl = torch.zeros(M, device="cuda")
ws = torch.zeros(lib.vmmt_gen_fused_ws_floats(M, V, H), device="cuda")
ldp = (V + 31) // 32 * 32
Pw = torch.zeros(M, ldp, device="cuda", dtype=T)
P = lambda t: C.c_void_p(t.data_ptr())
sA, sB = torch.cuda.Stream(), torch.cuda.Stream(priority=0)
lo = torch.cuda.Stream(priority=max(torch.cuda.Stream.priority_range()))


def sweep(st):
    L.check(lib.vmmt_gen_fwd_dO(L.BF16, P(W), H, Vp, P(bias), P(O), H, P(y), M, V, H, P(ws), P(tl), P(Pw), ldp, None, C.c_void_p(st.cuda_stream)), "sweep")


B, Z, D = 256, 256, 2048
a = torch.randn(B, D, device="cuda").to(T); b = torch.randn(B, D, device="cuda").to(T); o = torch.zeros(B, D, device="cuda", dtype=T)
a32 = torch.randn(B, D, device="cuda")
big = torch.randn(M, H, device="cuda").to(T); big2 = torch.randn(M, H, device="cuda").to(T); bigo = torch.zeros(M, H, device="cuda", dtype=T)


def act(R, Cc, st, inplace=False, f32=False):
    src = a32 if f32 else a
    L.check(lib.vmmt_act_bwd(L.BF16, L.ACT_RELU, P(src), D, 1 if f32 else 0, P(b), D, None, 0, P(src if inplace and not f32 else o), D, R, Cc, C.c_void_p(st.cuda_stream)), "act")


def mul(R, Cc, st):
    L.check(lib.vmmt_mul(L.BF16, P(a), D, P(b), D, P(o), D, R, Cc, C.c_void_p(st.cuda_stream)), "mul")


def mulbig(st):
    L.check(lib.vmmt_mul(L.BF16, P(big), H, P(big2), H, P(bigo), H, M, H, C.c_void_p(st.cuda_stream)), "mul")


cands = [("act_bwd8 256x256 (32 wg)", lambda st: act(B, Z, st)),
         ("act_bwd8 256x256 in place", lambda st: act(B, Z, st, inplace=True)),
         ("act_bwd8<float> 256x256", lambda st: act(B, Z, st, f32=True)),
         ("act_bwd8 256x128 (16 wg)", lambda st: act(B, 128, st)),
         ("act_bwd8 256x64 (8 wg)", lambda st: act(B, 64, st)),
         ("act_bwd8 256x2048 (256 wg)", lambda st: act(B, D, st)),
         ("mul8 256x256 (32 wg)", lambda st: mul(B, Z, st)),
         ("mul8 256x2048 (256 wg)", lambda st: mul(B, D, st)),
         ("mul8 5120x512 (1280 wg)", mulbig)]


def run(fn, n, under, st):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if under:
        s0.record(sA); sweep(sA); e0.record(sA)
        torch.cuda._sleep(40000)             # (on the null stream: nothing; the second stream starts a little behind by the launches' own latency)
    s.record(st)
    for _ in range(n):
        fn(st)
    e.record(st)
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n, (s0.elapsed_time(e0) * 1e3 if under else 0.0)


for _ in range(3):
    sweep(sA)
for name, fn in cands:
    for _ in range(2):
        fn(sB)
    for st_name, st in (("default-priority stream", sB), ("lowest-priority stream", lo)):
        alone, _ = run(fn, 4, False, st)
        und, sw = run(fn, 4, True, st)
        print("%-30s %-24s alone %7.1f us/launch   under the sweep %7.1f us/launch   (sweep %6.1f us)" % (name, st_name, alone, und, sw))
