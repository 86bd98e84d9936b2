"""The generator's two passes alone at the bench shape (cfg-2: M = 20 x 256 tokens, V = 30 000, H = 512; `--config 5`: H = 1024,
M = 50 x 256): timing with HIP events, and the process tools/pmc_gen.sh profiles.  `python tools/gen_one.py [idx]` keeps the
arg-max index (the decoding flavour of the forward pass)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd import _lib as L
lib = L.lib()
H, M = (1024, 50 * 256) if "--config5" in sys.argv else (512, 20 * 256)
if "--m" in sys.argv:            # e.g. --config5 --m 16384: config 5 as bench.py runs it (batch 512 x 32 target positions)
    M = int(sys.argv[sys.argv.index("--m") + 1])
V = 30000
keep_idx = "idx" in sys.argv
T = torch.bfloat16
g = torch.Generator().manual_seed(0)
Vp = (V + 255) // 256 * 256
W = (torch.randn(Vp, H, generator=g) * 0.05).to(T).cuda(); W[V:] = 0
O = torch.randn(M, H, generator=g).to(T).cuda()
bias = (torch.randn(Vp, generator=g) * 0.1).cuda()
y = torch.randint(2, V, (M,), generator=g).cuda()
if "zeros" in sys.argv:          # power check: the same instruction stream on all-zero operands
    W.zero_(); O.zero_()
npart = lib.vmmt_gen_npart(V)
pm = torch.zeros(npart * M, device="cuda"); ps = torch.zeros_like(pm)
pi = torch.zeros(npart * M, device="cuda", dtype=torch.int32)
tl = torch.zeros(M, device="cuda"); lse = torch.zeros(M, device="cuda"); nll = torch.zeros(M, device="cuda")
st = torch.zeros(8, device="cuda"); db = torch.zeros(Vp, device="cuda")
GT = torch.zeros(Vp, M, device="cuda", dtype=T)
P = lambda t: C.c_void_p(t.data_ptr())
def fwd():
    L.check(lib.vmmt_gen_loss_fwd(L.BF16, P(W), H, P(bias), P(O), H, P(y), M, V, H, 1, P(pm), P(ps), P(pi) if keep_idx else None, P(tl),
                                  P(lse), P(nll), P(st), None), "fwd")
def bwd():
    L.check(lib.vmmt_gen_loss_bwd_db(L.BF16, P(W), H, P(bias), P(O), H, P(y), M, V, H, 1, P(lse), 1.0 / 256, P(GT), M, P(db), 0, None), "bwd")
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): fn()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print("%s %8.1f us  %6.0f TF/s" % (name, us, 2.0 * M * V * H / us / 1e6))

# ---- the fused pass (csrc/generator_fused.hip): statistics + dO in one sweep of Wg, softmax weights P stored for the dWg GEMM
if lib.vmmt_gen_fused_applies(L.BF16, H, H, M, V, H):
    ws = torch.zeros(lib.vmmt_gen_fused_ws_floats(M, V, H), device="cuda")
    Mp = (M + 31) // 32 * 32
    y32 = torch.zeros(Mp, device="cuda", dtype=torch.int32)
    dO = torch.zeros(M, H, device="cuda")
    with_p = "nop" not in sys.argv
    ldp = (V + 31) // 32 * 32; Mk = (M + 63) // 64 * 64
    Pw = torch.zeros(M, ldp, device="cuda", dtype=T); cs_ = torch.zeros(16, (M + 127) // 128 * 128, device="cuda")
    Os_ = torch.zeros(16, Mk, H, device="cuda", dtype=T)
    def f1():
        L.check(lib.vmmt_gen_fwd_dO(L.BF16, P(W), H, Vp, P(bias), P(O), H, P(y), M, V, H, P(ws), P(tl), P(Pw) if with_p else None, ldp, None, None), "fwd dO")
        if "nocombine" not in sys.argv:
            L.check(lib.vmmt_gen_fwd_combine(L.BF16, P(W), H, P(O), H, P(y), M, V, H, 1, 1.0 / 256, P(ws), P(tl), P(lse), P(nll), P(y32),
                                             P(dO), H, P(st), P(cs_), P(Os_), H, Mk * H, None, None), "combine")
    for _ in range(3): f1()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f1()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print("fused fwd+dO (+combine) %8.1f us  %6.0f TF/s (2 products)" % (us, 4.0 * M * V * H / us / 1e6))
    import ctypes
    h = ctypes.CDLL(os.environ["VMMT_LIB_PATH"]) if "VMMT_LIB_PATH" in os.environ else None
    if h is not None and hasattr(h, "vmmt_g2_probe_read"):       # diagnostic build: tools/exp_build.sh generator_fused.hip PROBE
        buf = (ctypes.c_ulonglong * 16)()
        f1(); torch.cuda.synchronize()
        h.vmmt_g2_probe_read(buf)
        v = list(buf[:8]); n = max(1, v[6])
        print("second product in quarters of 8 MFMAs: %.0f %.0f %.0f %.0f" % (buf[8] / n, buf[9] / n, buf[10] / n, (v[4] - 0) / n))
        print("tiles %d, in-kernel clock %d MHz; s_memtime ticks per tile: vmcnt-wait %.0f  barrier %.0f  S-phase %.0f  elementwise %.0f  PV-phase %.0f  loop-tail %.0f"
              % (n, v[7], v[0] / n, v[1] / n, v[2] / n, v[3] / n, v[4] / n, v[5] / n))
