"""GPU tests (MI355X): individual libvmmt kernels through the C-ABI vs plain fp32/fp64 torch-CPU math."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from variational_mmt_amd import _lib as L
    return L, L.lib()


def _dev(t, dt):
    return t.to(device="cuda", dtype=dt).contiguous()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("layout", ["NT", "TN", "NN"])
@pytest.mark.parametrize("shape", [(70, 45, 33, 0), (130, 257, 100, 128), (64, 64, 64, 64), (5, 500, 2048, 0), (300, 1, 77, 64)])
def test_gemm_layouts(dtype, layout, shape):
    L, lib = _lib()
    M, N, K, tile = shape
    torch.manual_seed(M * 7 + N * 3 + K)
    T = torch.float32 if dtype == "f32" else torch.bfloat16
    code = L.F32 if dtype == "f32" else L.BF16
    a = (torch.randn(M, K) * 0.5).to(T)
    b = (torch.randn(N, K) * 0.5).to(T)
    ref = a.double() @ b.double().t()
    ld_pad = 8
    if layout == "NT":
        A = torch.zeros(M, K + ld_pad, dtype=T); A[:, :K] = a
        Bm = torch.zeros(N, K + ld_pad, dtype=T); Bm[:, :K] = b
        lay = L.GEMM_NT
    elif layout == "TN":
        A = torch.zeros(K, M + ld_pad, dtype=T); A[:, :M] = a.t()
        Bm = torch.zeros(K, N + ld_pad, dtype=T); Bm[:, :N] = b.t()
        lay = L.GEMM_TN
    else:
        A = torch.zeros(M, K + ld_pad, dtype=T); A[:, :K] = a
        Bm = torch.zeros(K, N + ld_pad, dtype=T); Bm[:, :N] = b.t()
        lay = L.GEMM_NN
    Ad, Bd = A.cuda(), Bm.cuda()
    Cd = torch.full((M, N + 3), 7.0, dtype=torch.float32, device="cuda")
    bias = torch.randn(N)
    biasd = bias.cuda()
    args = L.GemmArgs(code, lay, Ad.data_ptr(), Ad.stride(0), Bd.data_ptr(), Bd.stride(0), Cd.data_ptr(), Cd.stride(0), M, N, K,
                      0, 0, biasd.data_ptr(), N, 1, 0, L.ACT_NONE, 1, 0, 1.0, None, 1, tile)
    L.check(lib.vmmt_gemm(C.byref(args), None), "gemm")
    torch.cuda.synchronize()
    out = Cd.cpu()
    assert torch.all(out[:, N:] == 7.0), "wrote outside the valid columns"
    want = ref + bias.double()
    tol = 1e-4 if dtype == "f32" else 2e-2
    err = (out[:, :N].double() - want).abs().max().item()
    assert err <= tol * max(1.0, want.abs().max().item()), err


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_gemm_epilogues(dtype):
    L, lib = _lib()
    T = torch.float32 if dtype == "f32" else torch.bfloat16
    code = L.F32 if dtype == "f32" else L.BF16
    torch.manual_seed(3)
    M, N, K, R = 96, 80, 48, 12
    a, b = (torch.randn(M, K) * 0.3).to(T), (torch.randn(N, K) * 0.3).to(T)
    add = torch.randn(R, N)
    ref = a.double() @ b.double().t() + add.double()[torch.arange(M) % R]
    tol = 1e-4 if dtype == "f32" else 2e-2
    for act, fn in ((L.ACT_RELU, torch.relu), (L.ACT_TANH, torch.tanh), (L.ACT_SOFTPLUS, torch.nn.functional.softplus),
                    (L.ACT_SIGMOID, torch.sigmoid)):
        Ad, Bd, addd = a.cuda(), b.cuda(), add.cuda()
        Cd = torch.zeros(M, N, dtype=T, device="cuda")
        args = L.GemmArgs(code, L.GEMM_NT, Ad.data_ptr(), K, Bd.data_ptr(), K, Cd.data_ptr(), N, M, N, K, 0, 0, addd.data_ptr(), N, R,
                          0, act, 0, 0, 1.0, None, 1, 0)
        L.check(lib.vmmt_gemm(C.byref(args), None), "gemm")
        err = (Cd.cpu().double() - fn(ref)).abs().max().item()
        assert err <= tol * 3, (act, err)
    # scatter-add epilogue (embedding gradient) with padding rows dropped
    ids = torch.randint(0, 10, (M,))
    ids[::7] = 1
    table = torch.zeros(10, N, dtype=torch.float32, device="cuda")
    Ad, Bd, idd = a.cuda(), b.cuda(), ids.cuda()
    args = L.GemmArgs(code, L.GEMM_NT, Ad.data_ptr(), K, Bd.data_ptr(), K, table.data_ptr(), N, M, N, K, 0, 0, None, 0, 0, 0, 0, 1, 0, 1.0,
                      idd.data_ptr(), 1, 0)
    L.check(lib.vmmt_gemm(C.byref(args), None), "gemm")
    want = torch.zeros(10, N, dtype=torch.float64)
    prod = a.double() @ b.double().t()
    for m in range(M):
        if ids[m] != 1:
            want[ids[m]] += prod[m]
    assert (table.cpu().double() - want).abs().max().item() <= tol * 5


@pytest.mark.parametrize("hole, shadow", [(h, sh) for h in [(0, 0), (4096, 30000), (0, 8192), (60000, 40000)] for sh in [None, (100000, 20480), (1024, 2048)]
                                          if not (sh and h[1] and sh[0] < h[0] + h[1] and h[0] < sh[0] + sh[1])])      # (the shadowed piece lies on one side of the hole)
def test_adam_ranges_equal_one_launch_per_piece(hole, shadow):
    """vmmt_adam_step_ranges: a range with a hole (a lazily updated table) and a shadow over a part of it, in ONE launch -- the bits of one
    vmmt_adam_step per piece; the hole's elements are not touched"""
    L, lib = _lib()
    torch.manual_seed(3)
    n = 131072 + 8 + 3                       # (an odd tail behind the last whole group of four)
    base = [torch.randn(n).cuda(), (torch.randn(n) * 3).cuda(), (torch.rand(n) * 0.1).cuda(), (torch.rand(n) * 0.01).cuda()]
    ss = torch.zeros(L.SUMSQ_SCRATCH, device="cuda")
    L.check(lib.vmmt_sumsq(base[1].data_ptr(), n, ss.data_ptr(), 0, None), "sumsq")
    args = (0.002, 0.9, 0.999, 1e-9, 7, 5.0)
    # reference: one launch per piece
    a = [t.clone() for t in base]
    sh_a = torch.full((shadow[1] + 8,) if shadow else (8,), 9.0, device="cuda", dtype=torch.bfloat16)
    cuts = sorted({0, n, hole[0], hole[0] + hole[1]} | ({shadow[0], shadow[0] + shadow[1]} if shadow else set()))
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        if hi <= lo or (hole[1] and hole[0] <= lo < hole[0] + hole[1]):
            continue
        shp = sh_a.data_ptr() if (shadow and lo == shadow[0]) else None
        L.check(lib.vmmt_adam_step(a[0].data_ptr() + 4 * lo, a[1].data_ptr() + 4 * lo, a[2].data_ptr() + 4 * lo, a[3].data_ptr() + 4 * lo, hi - lo,
                                   *args, ss.data_ptr(), 1.0, 0, shp, None, None), "adam")
    b = [t.clone() for t in base]
    sh_b = torch.full_like(sh_a, 9.0)
    L.check(lib.vmmt_adam_step_ranges(b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr(), b[3].data_ptr(), n, hole[0], hole[1], *args,
                                      ss.data_ptr(), 1.0, 0, sh_b.data_ptr() if shadow else None, shadow[0] if shadow else 0,
                                      shadow[1] if shadow else 0, None, None), "adam ranges")
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert torch.equal(sh_a, sh_b)
    if hole[1]:
        for x, y in zip(b, base):
            assert torch.equal(x[hole[0]:hole[0] + hole[1]], y[hole[0]:hole[0] + hole[1]])
    if shadow:
        assert torch.equal(sh_b[:shadow[1]], b[0][shadow[0]:shadow[0] + shadow[1]].to(torch.bfloat16)) and (sh_b[shadow[1]:] == 9.0).all()
    # arguments the kernel's vector groups cannot take
    assert lib.vmmt_adam_step_ranges(b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr(), b[3].data_ptr(), n, 6, 8, *args, ss.data_ptr(), 1.0, 0,
                                     None, 0, 0, None, None) == 1          # VMMT_EINVAL


def test_adam_and_sumsq():
    L, lib = _lib()
    torch.manual_seed(0)
    n = 100003
    p, g = torch.randn(n), torch.randn(n) * 3
    m, v = torch.rand(n) * 0.1, torch.rand(n) * 0.01
    pd, gd, md, vd = [x.clone().cuda() for x in (p, g, m, v)]
    # arena pointers must be 16-byte aligned: torch allocations are
    ss = torch.zeros(L.SUMSQ_SCRATCH, device="cuda")
    # two arena segments -> two slots; Adam adds the slot totals in index order
    n0 = 40000
    L.check(lib.vmmt_sumsq(gd.data_ptr(), n0, ss.data_ptr(), 0, None), "sumsq")
    L.check(lib.vmmt_sumsq(gd.data_ptr() + 4 * n0, n - n0, ss.data_ptr(), 3, None), "sumsq")
    tot = float((g.double() ** 2).sum())
    assert abs(ss[:L.SUMSQ_SLOTS].sum().item() - tot) <= 1e-4 * tot
    # bit-reproducible: the partials are added in index order, whatever order the workgroups arrive in
    first = ss[:L.SUMSQ_SLOTS].clone()
    for _ in range(5):
        ss[:L.SUMSQ_SLOTS].zero_()
        L.check(lib.vmmt_sumsq(gd.data_ptr(), n0, ss.data_ptr(), 0, None), "sumsq")
        L.check(lib.vmmt_sumsq(gd.data_ptr() + 4 * n0, n - n0, ss.data_ptr(), 3, None), "sumsq")
        assert torch.equal(ss[:L.SUMSQ_SLOTS], first)
    sh = torch.full((n + 8,), 9.0, device="cuda", dtype=torch.bfloat16)        # the optional bf16 shadow of the updated parameters
    L.check(lib.vmmt_adam_step(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), n, 0.002, 0.9, 0.999, 1e-9, 3, 5.0,
                               ss.data_ptr(), 1.0, 0, sh.data_ptr(), None, None), "adam")
    torch.cuda.synchronize()
    assert torch.equal(sh[:n], pd.to(torch.bfloat16)) and (sh[n:] == 9.0).all()
    # the guard word (vmmt.h: `skip`): set -> the launch changes nothing and counts itself; clear -> the same update as without it
    keep = [t.clone() for t in (pd, md, vd, sh)]
    guard = torch.tensor([0x300, 0], device="cuda", dtype=torch.int32)
    L.check(lib.vmmt_adam_step(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), n, 0.002, 0.9, 0.999, 1e-9, 4, 5.0,
                               ss.data_ptr(), 1.0, 0, sh.data_ptr(), guard.data_ptr(), None), "adam")
    torch.cuda.synchronize()
    assert all(torch.equal(a_, b_) for a_, b_ in zip(keep, (pd, md, vd, sh))) and guard.tolist() == [0x300, 1]
    coef = min(1.0, 5.0 / (tot ** 0.5 + 1e-6))
    gg = g.double() * coef
    m2 = 0.9 * m.double() + 0.1 * gg
    v2 = 0.999 * v.double() + 0.001 * gg * gg
    p2 = p.double() - 0.002 / (1 - 0.9 ** 3) * m2 / (v2.sqrt() / (1 - 0.999 ** 3) ** 0.5 + 1e-9)
    assert (pd.cpu().double() - p2).abs().max().item() < 1e-5
    assert (md.cpu().double() - m2).abs().max().item() < 1e-6
    assert (vd.cpu().double() - v2).abs().max().item() < 1e-6


def test_gather_rows_and_image_loss():
    L, lib = _lib()
    torch.manual_seed(1)
    N, D, B = 50, 2048, 9
    table = torch.rand(N, D)
    idx = torch.randint(0, N, (B,))
    td, idd = table.cuda(), idx.cuda()
    out = torch.zeros(B, D, device="cuda")
    L.check(lib.vmmt_gather_rows(L.F32, td.data_ptr(), D, idd.data_ptr(), out.data_ptr(), D, B, D, None), "gather")
    assert torch.equal(out.cpu(), table[idx])
    mu = torch.randn(B, D)
    stats = torch.zeros(8, device="cuda")
    dmu = torch.zeros(B, D, device="cuda")
    mud = mu.cuda()
    L.check(lib.vmmt_image_loss(L.F32, mud.data_ptr(), D, out.data_ptr(), D, B, D, 1.0 / B, dmu.data_ptr(), D, stats.data_ptr(), None), "img")
    muq = mu.double().requires_grad_(True)
    v = table[idx].double()
    a = muq / muq.pow(2).sum(1, keepdim=True).sqrt()
    vh = v / v.pow(2).sum(1, keepdim=True).sqrt()
    import math
    logp = (-0.5 * (vh - a) ** 2 - 0.5 * math.log(2 * math.pi)).sum(0).mean(0)
    (-logp / B).backward()
    s = stats.cpu()
    assert abs(s[L.STAT_IMG_LOGPROB].item() - logp.item()) < 1e-4
    assert abs(s[L.STAT_IMG_COS].item() - (a * vh).sum(1).sum().item()) < 1e-4
    assert (dmu.cpu().double() - muq.grad).abs().max().item() < 1e-9 + 1e-4 * muq.grad.abs().max().item()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("layout", ["TN", "NN"])
def test_gemm_split_k_atomic(dtype, layout):
    """split-K: partial sums are atomically added into C (which already holds a partial gradient); addend added once."""
    L, lib = _lib()
    T = torch.float32 if dtype == "f32" else torch.bfloat16
    code = L.F32 if dtype == "f32" else L.BF16
    torch.manual_seed(5)
    M, N, K = 200, 72, 1000
    a, b = (torch.randn(M, K) * 0.2).to(T), (torch.randn(N, K) * 0.2).to(T)
    if layout == "TN":
        Ad, Bd, lay = a.t().contiguous().cuda(), b.t().contiguous().cuda(), L.GEMM_TN
        lda, ldb = M, N
    else:
        Ad, Bd, lay = a.cuda(), b.t().contiguous().cuda(), L.GEMM_NN
        lda, ldb = K, N
    init = torch.randn(M, N)
    bias = torch.randn(N)
    for split in (3, 7, 64):
        Cd = init.clone().cuda()
        biasd = bias.cuda()
        args = L.GemmArgs(code, lay, Ad.data_ptr(), lda, Bd.data_ptr(), ldb, Cd.data_ptr(), N, M, N, K, 0, 0, biasd.data_ptr(), N, 1, 0,
                          L.ACT_NONE, 1, 0, 1.0, None, 1, 0, split)
        L.check(lib.vmmt_gemm(C.byref(args), None), "gemm")
        want = init.double() + a.double() @ b.double().t() + bias.double()
        err = (Cd.cpu().double() - want).abs().max().item()
        assert err <= (1e-4 if dtype == "f32" else 3e-2), (split, err)


@pytest.mark.parametrize("M,V,H", [(5120, 3000, 512), (200, 1000, 128), (37, 515, 64), (264, 130, 192), (8, 67, 64)])
def test_generator_kernel_bf16(M, V, H):
    """fused projection + log-softmax + NLL (vmmt_gen_loss_fwd / _bwd, bf16) against fp64 math on the same bf16-rounded operands,
    with and without the arg-max index partials (decoding / training).  Ragged M / V exercise the clamped edge tiles, the -inf
    rows of the last vocabulary tile and the generic kernel behind the backward pass (M % 8 != 0)."""
    from variational_mmt_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(M + V)
    T = torch.bfloat16
    ld = (H + 63) // 64 * 64
    W = torch.zeros(V + 256, ld, dtype=T)
    O = torch.zeros(M + 256, ld, dtype=T)
    W[:V, :H] = (torch.randn(V, H, generator=g) * 0.3).to(T)
    O[:M, :H] = torch.randn(M, H, generator=g).to(T)
    bias = torch.randn(V, generator=g) * 0.2
    y = torch.randint(0, V, (M,), generator=g)
    y[::5] = 1                                                     # pad targets carry no loss
    y[1], y[2], y[3] = V - 1, 0, min(V - 1, 127)                   # first / last row of the vocabulary, a tile edge
    logits = O[:M, :H].double() @ W[:V, :H].double().t() + bias.double()
    lse_ref = torch.logsumexp(logits, 1)
    w = (y != 1).double()
    nll_ref = ((lse_ref - logits.gather(1, y.view(-1, 1)).view(-1)) * w)
    inv_norm = 1.0 / 7.0
    G_ref = (torch.softmax(logits, 1) - torch.nn.functional.one_hot(y, V).double()) * (w * inv_norm).view(-1, 1)
    Wd, Od, bd, yd = W.cuda(), O.cuda(), bias.cuda(), y.cuda()
    npart = lib.vmmt_gen_npart(V)
    ldgt = (M + 63) // 64 * 64
    outs = {}
    for keep_idx in (True, False):
        v = keep_idx
        pm = torch.zeros(npart * M, device="cuda"); ps = torch.zeros_like(pm)
        pi = torch.full((npart * M,), -7, device="cuda", dtype=torch.int32)
        tl = torch.zeros(M, device="cuda"); lse = torch.zeros(M, device="cuda"); nll = torch.zeros(M, device="cuda")
        st = torch.zeros(8, device="cuda")
        GT = torch.zeros(V + 256, ldgt, device="cuda", dtype=T)
        P = lambda t: C.c_void_p(t.data_ptr())
        L.check(lib.vmmt_gen_loss_fwd(L.BF16, P(Wd), ld, P(bd), P(Od), ld, P(yd), M, V, ld, 1, P(pm), P(ps), P(pi) if keep_idx else None,
                                      P(tl), P(lse), P(nll), P(st), None), "gen fwd")
        L.check(lib.vmmt_gen_loss_bwd(L.BF16, P(Wd), ld, P(bd), P(Od), ld, P(yd), M, V, ld, 1, P(lse), inv_norm, P(GT), ldgt, None), "gen bwd")
        torch.cuda.synchronize()
        assert (lse.cpu().double() - lse_ref).abs().max().item() <= 2e-4 * max(1.0, lse_ref.abs().max().item()), v
        assert (nll.cpu().double() - nll_ref).abs().max().item() <= 5e-4 * max(1.0, nll_ref.abs().max().item()), v
        s = st.cpu()
        assert abs(s[L.STAT_NLL].item() - nll_ref.sum().item()) <= 1e-4 * abs(nll_ref.sum().item()) + 1e-3
        assert int(round(s[L.STAT_NWORDS].item())) == int(w.sum().item())
        correct = ((logits.argmax(1) == y) & (y != 1)).sum().item()
        assert abs(int(round(s[L.STAT_NCORRECT].item())) - correct) <= 1          # a near-tie may flip one arg-max
        if keep_idx:                                                              # the decoding flavour: arg-max index + its logit
            oi = torch.zeros(M, device="cuda", dtype=torch.int64); om = torch.zeros(M, device="cuda")
            L.check(lib.vmmt_gen_argmax(P(pm), P(pi), M, npart, P(oi), P(om), None), "argmax")
            torch.cuda.synchronize()
            top = logits.max(1)
            assert (om.cpu().double() - top.values).abs().max().item() <= 2e-4 * max(1.0, top.values.abs().max().item())
            picked = logits.gather(1, oi.cpu().view(-1, 1)).view(-1)
            assert (picked - top.values).abs().max().item() <= 1e-3              # the index of (a near-tie of) the maximum
        else:
            assert (pi == -7).all()                                               # training: the index partials are not touched
        Gt = GT[:V, :M].float().cpu().t().double()
        assert (Gt - G_ref).abs().max().item() <= 4e-3 * inv_norm + 1e-6, v       # bf16 storage of G^T
        assert (GT[:V, M:] == 0).all() and (GT[V:] == 0).all(), v                 # nothing outside [V][M] is written
        outs[v] = (lse.clone(), nll.clone(), GT.clone())
        # the same pass with the fused bias gradient (vmmt_gen_loss_bwd_db): identical G^T, dbias += row sums of the stored G^T
        GT2 = torch.zeros_like(GT)
        db = torch.full((V,), 0.5, device="cuda")
        L.check(lib.vmmt_gen_loss_bwd_db(L.BF16, P(Wd), ld, P(bd), P(Od), ld, P(yd), M, V, ld, 1, P(lse), inv_norm, P(GT2), ldgt, P(db),
                                         0, None), "gen bwd db")
        torch.cuda.synchronize()
        assert torch.equal(GT2, GT), v
        ref_db = 0.5 + GT[:V, :M].float().sum(1)
        assert (db - ref_db).abs().max().item() <= 1e-5 + 1e-5 * ref_db.abs().max().item(), v
        # ... and walked in vocabulary chunks (whole 128-row tiles each, as the engine does): the same bits
        if V > 256:
            GT3 = torch.zeros_like(GT)
            db3 = torch.full((V,), 0.5, device="cuda")
            cuts = [0, 128, 384, V] if V > 384 else [0, 128, V]
            for v0, v1 in zip(cuts[:-1], cuts[1:]):
                L.check(lib.vmmt_gen_loss_bwd_db(L.BF16, C.c_void_p(Wd.data_ptr() + 2 * v0 * ld), ld, C.c_void_p(bd.data_ptr() + 4 * v0), P(Od), ld,
                                                 P(yd), M, v1 - v0, ld, 1, P(lse), inv_norm, C.c_void_p(GT3.data_ptr() + 2 * v0 * ldgt), ldgt,
                                                 C.c_void_p(db3.data_ptr() + 4 * v0), v0, None), "gen bwd chunk")
            torch.cuda.synchronize()
            assert torch.equal(GT3, GT), v
            assert (db3 - db).abs().max().item() <= 1e-5 + 1e-5 * ref_db.abs().max().item(), v       # f32 atomics: order of the token tiles
    for k in range(3):
        assert torch.equal(outs[True][k], outs[False][k]), k                      # with / without the index: same arithmetic, same bits


@pytest.mark.parametrize("M,V,H,ramp", [(5120, 3000, 512, 0.0), (200, 1000, 256, 0.0), (37, 515, 512, 0.0), (264, 130, 256, 0.0),
                                        (300, 2100, 512, 0.09), (8, 67, 256, 0.0),
                                        # H = 1024: gen2w_kernel (two waves per 32 tokens, 64-token blocks)
                                        (5200, 3000, 1024, 0.0), (37, 515, 1024, 0.0), (300, 2100, 1024, 0.09), (81, 67, 1024, 0.0)])
def test_generator_fused_pass_bf16(M, V, H, ramp):
    """csrc/generator_fused.hip: vmmt_gen_fwd_dO (softmax statistics + dL/dO in one sweep of Wg, softmax weights P stored on the way),
    dL/dWg as the per-slice product of P with the scaled decoder outputs O'_s, and vmmt_gen_dW_finish (dL/db, one-hot term) against fp64
    math on the same bf16-rounded operands.  `ramp` adds a bias that grows by `ramp` per vocabulary row (190 over 2100 rows): the lazy
    softmax reference of the sweep has to move -- rescale its accumulators and rewrite the P it has stored -- several times inside one
    vocabulary slice."""
    from variational_mmt_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(M * 7 + V)
    T = torch.bfloat16
    assert lib.vmmt_gen_fused_applies(L.BF16, H, H, M, V, H) == 1
    W = torch.zeros(V + 256, H, dtype=T)
    W[:V] = (torch.randn(V, H, generator=g) * 0.3).to(T)
    O = (torch.randn(M, H, generator=g) * 0.5).to(T)
    bias = torch.randn(V, generator=g) * 0.2 + ramp * torch.arange(V).float()
    y = torch.randint(0, V, (M,), generator=g)
    y[::5] = 1                                                     # pad targets carry no loss
    y[1], y[2], y[3] = V - 1, 0, min(V - 1, 127)
    logits = O.double() @ W[:V].double().t() + bias.double()
    lse_ref = torch.logsumexp(logits, 1)
    w = (y != 1).double()
    nll_ref = (lse_ref - logits.gather(1, y.view(-1, 1)).view(-1)) * w
    inv_norm = 1.0 / 7.0
    G_ref = (torch.softmax(logits, 1) - torch.nn.functional.one_hot(y, V).double()) * (w * inv_norm).view(-1, 1)
    dO_ref = G_ref @ W[:V].double()
    dW_ref = G_ref.t() @ O.double()
    db_ref = G_ref.sum(0)
    Wd, Od, bd, yd = W.cuda(), O.cuda(), bias.cuda(), y.cuda()
    P = lambda t: C.c_void_p(t.data_ptr())
    ws = torch.zeros(lib.vmmt_gen_fused_ws_floats(M, V, H), device="cuda")
    Mp = (M + 31) // 32 * 32
    tl = torch.zeros(M, device="cuda"); lse = torch.zeros(M, device="cuda"); nll = torch.zeros(M, device="cuda")
    y32 = torch.zeros(Mp, device="cuda", dtype=torch.int32)
    dO = torch.full((M, H + 4), 7.0, device="cuda"); st = torch.zeros(8, device="cuda")
    sc = inv_norm
    ns, vps, mpad = C.c_int(), C.c_int(), C.c_int64()
    L.check(lib.vmmt_gen_fused_geometry(M, V, H, C.byref(ns), C.byref(vps), C.byref(mpad)), "geometry")
    ns, vps, mpad = ns.value, vps.value, mpad.value
    ldp = (V + 31) // 32 * 32 + 32
    Pw = torch.full((M + 2, ldp), 3.0, device="cuda", dtype=T)
    cs = torch.zeros(ns, mpad, device="cuda")
    Mk = (M + 63) // 64 * 64
    Os = torch.zeros(ns, Mk, H, device="cuda", dtype=T)
    L.check(lib.vmmt_gen_fwd_dO(L.BF16, P(Wd), H, V + 256, P(bd), P(Od), H, P(yd), M, V, H, P(ws), P(tl), P(Pw), ldp, None, None), "gen fwd dO")
    assert lib.vmmt_gen_fwd_dO(L.BF16, P(Wd), H, V, P(bd), P(Od), H, P(yd), M, V, H, P(ws), P(tl), P(Pw), ldp, None, None) != 0     # W too short for the prefetch
    L.check(lib.vmmt_gen_fwd_combine(L.BF16, P(Wd), H, P(Od), H, P(yd), M, V, H, 1, inv_norm, P(ws), P(tl), P(lse), P(nll), P(y32),
                                     P(dO), H + 4, P(st), P(cs), P(Os), H, Mk * H, None, None), "gen fwd combine")
    torch.cuda.synchronize()
    assert (lse.cpu().double() - lse_ref).abs().max().item() <= 2e-4 * max(1.0, lse_ref.abs().max().item())
    assert (nll.cpu().double() - nll_ref).abs().max().item() <= 5e-4 * max(1.0, nll_ref.abs().max().item())
    s = st.cpu()
    assert abs(s[L.STAT_NLL].item() - nll_ref.sum().item()) <= 1e-4 * abs(nll_ref.sum().item()) + 1e-3
    assert int(round(s[L.STAT_NWORDS].item())) == int(w.sum().item())
    correct = ((logits.argmax(1) == y) & (y != 1)).sum().item()
    assert abs(int(round(s[L.STAT_NCORRECT].item())) - correct) <= 1
    assert (dO[:, H:] == 7.0).all()                                                # nothing outside the outputs is written
    # the softmax weights enter the second MFMA product as bf16 (2^-9 relative each, as the stored G^T of the unfused path does)
    e1 = (dO[:, :H].cpu().double() - dO_ref).abs().max().item()
    assert e1 <= 6e-3 * sc * max(1.0, W[:V].float().abs().max().item()), e1
    assert (y32[M:] == -1).all() and (y32[:M].cpu() == torch.where(y == 1, -1, y).int()).all()
    assert (Pw[M:] == 3.0).all() and (Pw[:, (V + 31) // 32 * 32:] == 3.0).all() and (Os[:, M:] == 0).all()
    # the same fold in two launches (the training step: the dWg product starts between them): _stats leaves dO alone, _dO writes dO alone
    dO2 = torch.full((M, H + 4), 7.0, device="cuda"); st2 = torch.zeros(8, device="cuda")
    lse2, nll2, cs2, Os2, y322 = torch.zeros_like(lse), torch.zeros_like(nll), torch.zeros_like(cs), torch.zeros_like(Os), torch.zeros_like(y32)
    L.check(lib.vmmt_gen_fwd_combine_stats(L.BF16, P(Wd), H, P(Od), H, P(yd), M, V, H, 1, inv_norm, P(ws), P(tl), P(lse2), P(nll2), P(y322),
                                           P(dO2), H + 4, P(st2), P(cs2), P(Os2), H, Mk * H, None, None), "combine stats")
    torch.cuda.synchronize()
    assert (dO2 == 7.0).all() and torch.equal(lse2, lse) and torch.equal(nll2, nll) and torch.equal(cs2, cs) and torch.equal(Os2, Os)
    assert torch.equal(st2[:7], st[:7]) and int(st2[L.STAT_TICKET].view(torch.int32)) == 0 and torch.equal(y322, y32)      # (the ticket resets itself)
    keep = [t.clone() for t in (lse2, nll2, cs2, Os2, st2, y322)]
    L.check(lib.vmmt_gen_fwd_combine_dO(L.BF16, P(Wd), H, P(Od), H, P(yd), M, V, H, 1, inv_norm, P(ws), P(tl), P(lse2), P(nll2), P(y322),
                                        P(dO2), H + 4, P(st2), P(cs2), P(Os2), H, Mk * H, None, None), "combine dO")
    torch.cuda.synchronize()
    assert torch.equal(dO2, dO) and all(torch.equal(a_, b_) for a_, b_ in zip(keep, (lse2, nll2, cs2, Os2, st2, y322)))
    dW2 = torch.empty(V, H, device="cuda")
    for s_ in range(ns):
        v0, v1 = s_ * vps, min(V, (s_ + 1) * vps)
        if v0 < v1:
            dW2[v0:v1] = Pw[:M, v0:v1].float().t() @ Os[s_, :M].float()
    db2 = torch.full((V,), 0.5, device="cuda")
    L.check(lib.vmmt_gen_dW_finish(L.BF16, P(Pw), ldp, P(cs), P(Od), H, P(y32), M, V, H, inv_norm, P(dW2), H, P(db2), 0, None, None), "dW finish")
    torch.cuda.synchronize()
    e4 = (dW2.cpu().double() - dW_ref).norm().item() / max(1e-30, dW_ref.norm().item())
    assert e4 <= 6e-3, e4
    e4m = (dW2.cpu().double() - dW_ref).abs().max().item()
    assert e4m <= 2e-2 * sc * max(1.0, O.float().abs().max().item()) * max(1.0, (M / 256.0) ** 0.5), e4m
    e5 = (db2.cpu().double() - 0.5 - db_ref).abs().max().item()
    assert e5 <= 4e-3 * max(sc * max(1.0, (M / 256.0) ** 0.5), db_ref.abs().max().item()), e5     # P is rounded to bf16 before the token sum


def test_gemm_per_row_block_b_operand():
    """vmmt_gemm_args.b_batch_rows: output rows [i r, (i + 1) r) multiply with the B operand at B + i * b_batch_stride (the generator's
    weight gradient: one scaled copy of the decoder outputs per vocabulary slice) -- against one torch product per block."""
    from variational_mmt_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(3)
    T = torch.bfloat16
    Mr, N, K, r, nb = 1100, 512, 640, 512, 3                      # rows, columns, reduction, rows per block, blocks
    At = (torch.randn(K, Mr + 4, generator=g) * 0.5).to(T).cuda()          # A^T as stored (GEMM_TN): [K][lda]
    Bs = (torch.randn(nb, K, N, generator=g) * 0.5).to(T).cuda()
    Cd = torch.full((Mr, N), 9.0, device="cuda")
    a = L.GemmArgs(L.BF16, L.GEMM_TN, At.data_ptr(), Mr + 4, Bs.data_ptr(), N, Cd.data_ptr(), N, Mr, N, K, 0, 0, None, 0, 0, 0, L.ACT_NONE,
                   1, 0, 1.0, None, 1, 0, 0, r, K * N)
    L.check(lib.vmmt_gemm(C.byref(a), None), "gemm")
    torch.cuda.synchronize()
    for i in range(nb):
        r0, r1 = i * r, min(Mr, (i + 1) * r)
        want = At[:, r0:r1].float().t() @ Bs[i].float()
        assert (Cd[r0:r1] - want).abs().max().item() <= 2e-2 * max(1.0, want.abs().max().item()), i
    bad = L.GemmArgs(L.BF16, L.GEMM_TN, At.data_ptr(), Mr + 4, Bs.data_ptr(), N, Cd.data_ptr(), N, Mr, N, K, 0, 0, None, 0, 0, 0, L.ACT_NONE,
                     1, 0, 1.0, None, 1, 0, 0, 100, K * N)
    assert lib.vmmt_gemm(C.byref(bad), None) != 0                   # blocks must be whole tiles (multiples of 256 rows)


def test_gemm_gathered_a_operand_equals_gather_then_gemm():
    """vmmt_gemm_args.a_row_ids: row m of the A operand is row ids[m] of a table with UNPADDED rows (E = 500 elements: every other row
    starts 8 bytes off a 16-byte boundary) -- the embedding lookup as the operand fetch of the LSTM's input projection
    (modules/Embeddings.py:169-188 feeding Models.py:124-129).  Same bits as vmmt_gather_rows into a zero-padded buffer + the plain product."""
    from variational_mmt_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(11)
    T = torch.bfloat16
    R, E, Kp, N = 3000, 500, 512, 640
    for M in (1000, 5120, 37):                                     # partial last tile, the benchmark's token count, less than one tile
        table32 = torch.randn(R, E, generator=g) * 0.5
        table = torch.zeros(R + 64, E, dtype=T, device="cuda")    # (+ rows of slack: the last row's slab reads 12 elements beyond its end)
        table[:R] = table32.to(T).cuda()
        ids = torch.randint(0, R, (M,), generator=g)
        ids[0], ids[-1] = R - 1, 0
        idsd = ids.cuda()
        W = torch.zeros(N, Kp, dtype=T, device="cuda")            # zero columns E .. Kp - 1: what the operand reads beyond a row meets zeros
        W[:, :E] = (torch.randn(N, E, generator=g) * 0.5).to(T).cuda()
        bias = torch.randn(N, generator=g).cuda()
        X = torch.zeros(M, Kp, dtype=T, device="cuda")
        X[:, :E] = table[idsd]
        outs = []
        for gathered in (False, True):
            Cd = torch.full((M, N), 7.0, device="cuda")
            if gathered:
                a = L.GemmArgs(L.BF16, L.GEMM_NT, table.data_ptr(), E, W.data_ptr(), Kp, Cd.data_ptr(), N, M, N, Kp, 0, 0, bias.data_ptr(), N, 1, 0,
                               L.ACT_NONE, 1, 0, 1.0, None, 1, 0, 0, 0, 0)
                a.a_row_ids = idsd.data_ptr()
            else:
                a = L.GemmArgs(L.BF16, L.GEMM_NT, X.data_ptr(), Kp, W.data_ptr(), Kp, Cd.data_ptr(), N, M, N, Kp, 0, 0, bias.data_ptr(), N, 1, 0,
                               L.ACT_NONE, 1, 0, 1.0, None, 1, 128, 0, 0, 0)
            L.check(lib.vmmt_gemm(C.byref(a), None), "gemm")
            torch.cuda.synchronize()
            outs.append(Cd)
        assert torch.equal(outs[0], outs[1]), (M, (outs[0] - outs[1]).abs().max().item())
        want = X[:, :E].float() @ W[:, :E].float().t() + bias
        assert (outs[1] - want).abs().max().item() <= 2e-2 * max(1.0, want.abs().max().item())
    # refused where the gathered loop does not apply: fp32, another layout, a reduction length that is not whole slabs
    bad = L.GemmArgs(L.BF16, L.GEMM_NN, table.data_ptr(), E, W.data_ptr(), Kp, Cd.data_ptr(), N, M, N, Kp, 0, 0, None, 0, 0, 0, L.ACT_NONE, 1, 0, 1.0, None, 1, 0, 0, 0, 0)
    bad.a_row_ids = idsd.data_ptr()
    assert lib.vmmt_gemm(C.byref(bad), None) != 0
    bad = L.GemmArgs(L.BF16, L.GEMM_NT, table.data_ptr(), E, W.data_ptr(), Kp, Cd.data_ptr(), N, M, N, E, 0, 0, None, 0, 0, 0, L.ACT_NONE, 1, 0, 1.0, None, 1, 0, 0, 0, 0)
    bad.a_row_ids = idsd.data_ptr()
    assert lib.vmmt_gemm(C.byref(bad), None) != 0


def test_gemm_weighted_column_sums_of_the_k_strided_operand():
    """vmmt_gemm_args.colsum_w / colsum_out: colsum_out[m] += sum_k A[k][m] w_block(m)[k] out of the same pass as C = A^T B (the
    generator's bias gradient next to its weight gradient), where vmmt_gemm_colsum_applies() says so; the product itself unchanged"""
    from variational_mmt_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(5)
    T = torch.bfloat16
    Mr, N, K, r = 24980, 512, 704, 8192                            # ragged last tile (24980 = 97 * 256 + 148), 4 row blocks
    nb = (Mr + r - 1) // r
    lda = Mr + 4
    At = (torch.randn(K, lda, generator=g) * 0.5).to(T).cuda()
    Bs = (torch.randn(nb, K, N, generator=g) * 0.5).to(T).cuda()
    w = torch.randn(nb, K + 64, generator=g).cuda()                # stride K + 64 between the blocks' weight vectors
    out = torch.full((Mr + 8,), 0.25, device="cuda")
    outs = []
    for with_sums in (True, False):
        Cd = torch.full((Mr, N), 9.0, device="cuda")
        a = L.GemmArgs(L.BF16, L.GEMM_TN, At.data_ptr(), lda, Bs.data_ptr(), N, Cd.data_ptr(), N, Mr, N, K, 0, 0, None, 0, 0, 0, L.ACT_NONE,
                       1, 0, 1.0, None, 1, 0, 0, r, K * N)
        if with_sums:
            a.colsum_w, a.colsum_w_stride, a.colsum_out = w.data_ptr(), K + 64, out.data_ptr()
            assert lib.vmmt_gemm_colsum_applies(C.byref(a)) == 1
        L.check(lib.vmmt_gemm(C.byref(a), None), "gemm")
        torch.cuda.synchronize()
        outs.append(Cd)
    assert torch.equal(outs[0], outs[1])                           # the product does not notice
    for i in range(nb):
        r0, r1 = i * r, min(Mr, (i + 1) * r)
        want = 0.25 + (At[:, r0:r1].double() * w[i, :K].double()[:, None]).sum(0)
        got = out[r0:r1].double()
        assert (got - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item()), (i, (got - want).abs().max().item())
    assert (out[Mr:] == 0.25).all()
    small = L.GemmArgs(L.BF16, L.GEMM_TN, At.data_ptr(), lda, Bs.data_ptr(), N, outs[0].data_ptr(), N, 1024, N, K, 0, 0, None, 0, 0, 0, L.ACT_NONE,
                       1, 0, 1.0, None, 1, 0, 0, 0, 0)
    small.colsum_w, small.colsum_out = w.data_ptr(), out.data_ptr()
    assert lib.vmmt_gemm_colsum_applies(C.byref(small)) == 0 and lib.vmmt_gemm(C.byref(small), None) != 0     # small products: not offered


def test_gemm_plain_column_sums_with_split_k():
    """colsum_w == NULL: colsum_out[m] (and colsum_out2[m]) += sum_k A[k][m] next to a split-K weight-gradient product (an LSTM bias
    gradient, both nn.LSTM bias vectors, out of dW_ih = dgates^T x)"""
    from variational_mmt_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(6)
    T = torch.bfloat16
    Mr, N, K = 1000, 500, 5120                                     # ragged rows and columns
    At = (torch.randn(K, 1024, generator=g) * 0.5).to(T).cuda()
    Bm = (torch.randn(K, 512, generator=g) * 0.5).to(T).cuda()
    for split in (1, 4):
        Cd = torch.zeros(Mr, N, device="cuda")
        o1, o2 = torch.full((Mr + 8,), 0.5, device="cuda"), torch.full((Mr + 8,), -1.0, device="cuda")
        a = L.GemmArgs(L.BF16, L.GEMM_TN, At.data_ptr(), 1024, Bm.data_ptr(), 512, Cd.data_ptr(), N, Mr, N, K, 0, 0, None, 0, 0, 0, L.ACT_NONE,
                       1, 1 if split == 1 else 0, 1.0, None, 1, 128 if split == 1 else 0, split, 0, 0)
        a.colsum_out, a.colsum_out2 = o1.data_ptr(), o2.data_ptr()
        assert lib.vmmt_gemm_colsum_applies(C.byref(a)) == 1
        L.check(lib.vmmt_gemm(C.byref(a), None), "gemm")
        torch.cuda.synchronize()
        want_c = At[:, :Mr].float().t() @ Bm[:, :N].float()
        assert (Cd - want_c).abs().max().item() <= 2e-2 * want_c.abs().max().item()
        want = At[:, :Mr].double().sum(0)
        assert (o1[:Mr].double() - 0.5 - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item()), split
        assert (o2[:Mr].double() + 1.0 - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item()), split
        assert (o1[Mr:] == 0.5).all() and (o2[Mr:] == -1.0).all()


def test_scatter_add_rows_with_padding_row():
    """vmmt_scatter_add_rows: out[ids[r]] += X[r] with the padding id dropped (modules/Embeddings.py:118) against index_add_"""
    from variational_mmt_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(11)
    R, D, V = 777, 500, 300
    X = torch.randn(R, 512, generator=g)
    ids = torch.randint(0, V, (R,), generator=g)
    ids[::7] = 1
    out0 = torch.randn(V, D, generator=g)
    Xd, idd, od = X.cuda(), ids.cuda(), out0.clone().cuda()
    L.check(lib.vmmt_scatter_add_rows(C.c_void_p(Xd.data_ptr()), 512, C.c_void_p(idd.data_ptr()), 1, C.c_void_p(od.data_ptr()), D, R, D, None),
            "scatter")
    torch.cuda.synchronize()
    keep = ids != 1
    want = out0.double().index_add_(0, ids[keep], X[keep, :D].double())
    assert (od.cpu().double() - want).abs().max().item() <= 1e-4
    assert torch.equal(od[1].cpu(), out0[1])                          # the padding row receives nothing


def test_compact_nonpad_lists_the_target_rows_in_order():
    import ctypes as C
    from variational_mmt_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(9)
    for M, frac in ((5120, 0.26), (1, 0.0), (1000, 1.0), (16640, 0.5), (777, 0.03)):
        y = torch.randint(2, 50, (M,), generator=g)
        y[torch.rand(M, generator=g) < frac] = 1
        want = torch.nonzero(y != 1).flatten().int()
        n = want.numel()
        Mc = max(128, (n + 127) // 128 * 128)
        rows = torch.full((Mc + 5,), 77, dtype=torch.int32, device="cuda")
        cnt = torch.zeros(2, dtype=torch.int32, device="cuda")
        yd = y.cuda()
        L.check(lib.vmmt_compact_nonpad(C.c_void_p(yd.data_ptr()), M, 1, Mc, C.c_void_p(rows.data_ptr()), C.c_void_p(cnt.data_ptr()), None), "compact")
        torch.cuda.synchronize()
        assert cnt.tolist() == [n, 0]
        if n > 128:         # a caller that promises too few tokens: the list is cut, the flag is raised (and stays)
            short = torch.zeros(128, dtype=torch.int32, device="cuda")
            L.check(lib.vmmt_compact_nonpad(C.c_void_p(yd.data_ptr()), M, 1, 128, C.c_void_p(short.data_ptr()), C.c_void_p(cnt.data_ptr()), None), "compact")
            torch.cuda.synchronize()
            assert cnt.tolist() == [n, 1] and torch.equal(short.cpu(), want[:128])
        assert torch.equal(rows[:n].cpu(), want) and (rows[n:Mc] == -1).all() and (rows[Mc:] == 77).all()


@pytest.mark.parametrize("M,V,H", [(1500, 3000, 512), (2600, 4100, 256), (1300, 2500, 1024)])
def test_generator_over_compacted_tokens_equals_the_dense_calls(M, V, H):
    """the fused generator calls over the rows that carry a target only (vmmt_compact_nonpad + `rows`) against the same calls over all
    rows: statistics, per-token lse / NLL (zero where no token stands), dO (zero rows at pads), dWg and db -- the same sums up to the order
    of the f32 additions (the vocabulary slices follow the token count)"""
    import ctypes as C
    from variational_mmt_amd import _lib as L
    lib = L.lib()
    T = torch.bfloat16
    g = torch.Generator().manual_seed(M + V)
    Vp = (V + 31) // 32 * 32 + 256
    W = (torch.randn(Vp, H, generator=g) * 0.05).to(T).cuda(); W[V:] = 0
    O = (torch.randn(M, H, generator=g) * 0.5).to(T).cuda()
    bias = (torch.randn(Vp, generator=g) * 0.1).cuda()
    y = torch.randint(2, V, (M,), generator=g)
    y[torch.rand(M, generator=g) < 0.3] = 1                     # 30 % pads
    yd = y.cuda()
    n = int((y != 1).sum())
    Mc = (n + 127) // 128 * 128
    assert Mc < M
    inv_norm = 1.0 / 40.0
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None

    def run(Mx, rows):
        ns, vps, mpad = C.c_int(), C.c_int(), C.c_int64()
        L.check(lib.vmmt_gen_fused_geometry(Mx, V, H, C.byref(ns), C.byref(vps), C.byref(mpad)), "geometry")
        ns, vps, mpad = ns.value, vps.value, mpad.value
        ws = torch.zeros(lib.vmmt_gen_fused_ws_floats(Mx, V, H), device="cuda")
        ldp = (V + 31) // 32 * 32
        Mk = Mx + 64
        Pw = torch.zeros(Mk, ldp, device="cuda", dtype=T)
        tl = torch.zeros(M, device="cuda"); lse = torch.zeros(M, device="cuda"); nll = torch.zeros(M, device="cuda")
        y32 = torch.zeros((Mx + 31) // 32 * 32, device="cuda", dtype=torch.int32)
        dO = torch.zeros(M, H, device="cuda"); st = torch.zeros(8, device="cuda")
        cs = torch.zeros(ns, mpad, device="cuda"); Os = torch.zeros(ns, Mk, H, device="cuda", dtype=T)
        L.check(lib.vmmt_gen_fwd_dO(L.BF16, P(W), H, Vp, P(bias), P(O), H, P(yd), Mx, V, H, P(ws), P(tl), P(Pw), ldp, P(rows), None), "sweep")
        L.check(lib.vmmt_gen_fwd_combine(L.BF16, P(W), H, P(O), H, P(yd), Mx, V, H, 1, inv_norm, P(ws), P(tl), P(lse), P(nll), P(y32),
                                         P(dO), H, P(st), P(cs), P(Os), H, Mk * H, P(rows), None), "combine")
        dW = torch.empty(V, H, device="cuda")
        for s_ in range(ns):
            v0, v1 = s_ * vps, min(V, (s_ + 1) * vps)
            if v0 < v1:
                dW[v0:v1] = Pw[:Mx, v0:v1].float().t() @ Os[s_, :Mx].float()
        db = torch.zeros(V, device="cuda")
        L.check(lib.vmmt_gen_dW_finish(L.BF16, P(Pw), ldp, P(cs), P(O), H, P(y32), Mx, V, H, inv_norm, P(dW), H, P(db), 0, P(rows), None), "finish")
        torch.cuda.synchronize()
        return dict(lse=lse.cpu(), nll=nll.cpu(), dO=dO.cpu(), st=st.cpu(), dW=dW.cpu(), db=db.cpu())

    dense = run(M, None)
    rows = torch.full((Mc,), -1, dtype=torch.int32, device="cuda")
    L.check(lib.vmmt_compact_nonpad(P(yd), M, 1, Mc, P(rows), None, None), "compact")
    comp = run(Mc, rows)
    tok = y != 1
    assert (comp["lse"][~tok] == 0).all() and (comp["nll"][~tok] == 0).all() and (comp["dO"][~tok] == 0).all()
    assert (dense["dO"][~tok] == 0).all()
    assert (comp["lse"][tok] - dense["lse"][tok]).abs().max().item() <= 1e-4 * dense["lse"].abs().max().item()
    assert (comp["nll"][tok] - dense["nll"][tok]).abs().max().item() <= 2e-4 * dense["nll"].abs().max().item()
    assert abs(comp["st"][L.STAT_NLL].item() - dense["st"][L.STAT_NLL].item()) <= 1e-5 * abs(dense["st"][L.STAT_NLL].item())
    assert comp["st"][L.STAT_NWORDS].item() == dense["st"][L.STAT_NWORDS].item() == n
    assert abs(comp["st"][L.STAT_NCORRECT].item() - dense["st"][L.STAT_NCORRECT].item()) <= 1
    assert (comp["dO"] - dense["dO"]).abs().max().item() <= 2e-3 * dense["dO"].abs().max().item()
    assert (comp["dW"] - dense["dW"]).norm().item() <= 2e-3 * dense["dW"].norm().item()
    assert (comp["db"] - dense["db"]).abs().max().item() <= 2e-3 * max(1e-6, dense["db"].abs().max().item())


def test_gemm_group_equals_separate_launches():
    """vmmt_gemm_group: the weight-gradient products of a layer as ONE grid -- five products of different shapes (ragged rows / columns,
    a short reduction, two members accumulating into the SAME C, plain column sums riding in one of them, a padded row-block map)
    against fp64 math and against the same products issued one by one; a group with an ineligible member falls back to one launch
    per member with the same results"""
    from variational_mmt_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(8)
    T = torch.bfloat16
    K1, K2 = 4864, 256
    A1 = (torch.randn(K1, 1024, generator=g) * 0.5).to(T).cuda()          # "dgates" of steps 1.. (K-strided)
    B1 = (torch.randn(K1, 256, generator=g) * 0.5).to(T).cuda()           # h_prev
    A2 = (torch.randn(K2, 1024, generator=g) * 0.5).to(T).cuda()          # step 0
    B2 = (torch.randn(K2, 256, generator=g) * 0.5).to(T).cuda()
    A3 = (torch.randn(5120, 1024, generator=g) * 0.5).to(T).cuda()        # all steps
    B3 = (torch.randn(5120, 512, generator=g) * 0.5).to(T).cuda()         # x (500 valid columns)
    A4 = (torch.randn(5120, 512, generator=g) * 0.5).to(T).cuda()
    B4 = (torch.randn(5120, 1024, generator=g) * 0.5).to(T).cuda()

    def problems(Chh, Cih, Cw, b1, b2, Cpad):
        def mk(A, lda, Bm, ldb, Cd, ldc, M, N, K, split):
            return L.GemmArgs(L.BF16, L.GEMM_TN, A.data_ptr(), lda, Bm.data_ptr(), ldb, Cd.data_ptr(), ldc, M, N, K, 0, 0, None, 0, 0, 0, L.ACT_NONE,
                              1, 0, 1.0, None, 1, 0, split, 0, 0)
        ps = [mk(A1, 1024, B1, 256, Chh, 256, 1000, 256, K1, 4),          # dW_hh, steps 1..     (ragged rows)
              mk(A2, 1024, B2, 256, Chh, 256, 1000, 256, K2, 2),          # dW_hh, step 0: the SAME C
              mk(A3, 1024, B3, 512, Cih, 500, 1000, 500, 5120, 4),        # dW_ih + both bias gradients
              mk(A4, 512, B4, 1024, Cw, 1000, 512, 1000, 5120, 4),        # an attention-shaped product
              mk(A3, 1024, B3, 512, Cpad, 500, 1024, 500, 5120, 3)]       # padded gate blocks: 4 x 256 rows computed, 4 x 250 stored
        ps[2].colsum_out, ps[2].colsum_out2 = b1.data_ptr(), b2.data_ptr()
        ps[4].c_row_blk, ps[4].c_row_valid = 256, 250
        return ps

    def buffers():
        return (torch.zeros(1000, 256, device="cuda"), torch.zeros(1000, 500, device="cuda"), torch.zeros(512, 1000, device="cuda"),
                torch.zeros(1008, device="cuda"), torch.zeros(1008, device="cuda"), torch.zeros(1000, 500, device="cuda"))
    one, grp = buffers(), buffers()
    ps1, psg = problems(*one), problems(*grp)
    for a in ps1:
        L.check(lib.vmmt_gemm(C.byref(a), None), "gemm")
    arr = (L.GemmArgs * len(psg))(*psg)
    assert lib.vmmt_gemm_group_applies(arr, len(psg)) == 1
    L.check(lib.vmmt_gemm_group(arr, len(psg), None), "gemm_group")
    torch.cuda.synchronize()
    want_hh = A1[:, :1000].double().t() @ B1.double() + A2[:, :1000].double().t() @ B2.double()
    want_ih = A3[:, :1000].double().t() @ B3[:, :500].double()
    want_w = A4.double().t() @ B4[:, :1000].double()
    full = A3.double().t() @ B3[:, :500].double()
    want_pad = torch.cat([full[q * 256:q * 256 + 250] for q in range(4)])
    want_b = A3[:, :1000].double().sum(0)
    for name, got, want in (("hh", grp[0], want_hh), ("ih", grp[1], want_ih), ("w", grp[2], want_w), ("pad", grp[5], want_pad)):
        err = (got.double().cpu() - want.cpu()).abs().max().item()
        assert err <= 2e-3 * want.abs().max().item(), (name, err)                      # bf16 operands, f32 accumulation
        # against the one-by-one launches: the same tiles and splits, another order of the atomic adds
        one_t = one[{"hh": 0, "ih": 1, "w": 2, "pad": 5}[name]]
        assert (got - one_t).abs().max().item() <= 1e-5 * want.abs().max().item(), name
    for b in (grp[3], grp[4]):
        assert (b[:1000].double().cpu() - want_b.cpu()).abs().max().item() <= 1e-4 * want_b.abs().max().item() and (b[1000:] == 0).all()
    # a member that is not a grouped kind (no split: plain store) -> one launch per member, same numbers
    Cs = torch.zeros(512, 1000, device="cuda")
    mixed = [psg[3], psg[3]]
    mixed[1] = L.GemmArgs(L.BF16, L.GEMM_TN, A4.data_ptr(), 512, B4.data_ptr(), 1024, Cs.data_ptr(), 1000, 512, 1000, 5120, 0, 0, None, 0, 0, 0,
                          L.ACT_NONE, 1, 0, 1.0, None, 1, 0, 1, 0, 0)
    grp[2].zero_()
    arr2 = (L.GemmArgs * 2)(*mixed)
    assert lib.vmmt_gemm_group_applies(arr2, 2) == 0
    L.check(lib.vmmt_gemm_group(arr2, 2, None), "gemm_group fallback")
    torch.cuda.synchronize()
    for got in (grp[2], Cs):
        assert (got.double().cpu() - want_w.cpu()).abs().max().item() <= 2e-3 * want_w.abs().max().item()


def test_mul_and_act_bwd_vector_paths_equal_the_scalar_kernels():
    """vmmt_mul / vmmt_act_bwd take eight bf16 elements per thread where rows are 16-byte aligned; the same data in buffers whose leading
    dimension breaks the alignment goes through the scalar kernels: identical bits, and the padding columns stay untouched"""
    from variational_mmt_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(12)
    T = torch.bfloat16
    R, Cc = 777, 504
    a = (torch.randn(R, Cc, generator=g)).to(T)
    mk = ((torch.rand(R, Cc, generator=g) > 0.5).float() * 2.0).to(T)
    yv = torch.tanh(torch.randn(R, Cc, generator=g)).to(T)
    dy32 = torch.randn(R, Cc, generator=g)

    def buf(src, ld, dt):
        t = torch.full((R, ld), 7.0, dtype=dt, device="cuda")
        t[:, :Cc] = src.to(dt).cuda()
        return t
    outs = {}
    for ld in (512, 513):                 # 512: vector path; 513: rows not 16-byte aligned -> scalar kernels
        A, M, Y, D32, Db = buf(a, ld, T), buf(mk, ld, T), buf(yv, ld, T), buf(dy32, ld, torch.float32), buf(dy32, ld, T)
        o_mul = torch.full((R, ld), 3.0, dtype=T, device="cuda")
        L.check(lib.vmmt_mul(L.BF16, A.data_ptr(), ld, M.data_ptr(), ld, o_mul.data_ptr(), ld, R, Cc, None), "mul")
        res = [o_mul]
        for act in (L.ACT_TANH, L.ACT_RELU, L.ACT_SIGMOID, L.ACT_SOFTPLUS):
            for dyb, f32 in ((D32, 1), (Db, 0)):
                for m in (M, None):
                    o = torch.full((R, ld), 3.0, dtype=T, device="cuda")
                    L.check(lib.vmmt_act_bwd(L.BF16, act, dyb.data_ptr(), ld, f32, Y.data_ptr(), ld, m.data_ptr() if m is not None else None, ld if m is not None else 0,
                                             o.data_ptr(), ld, R, Cc, None), "act_bwd")
                    res.append(o)
        # in place (dropout backward between layers: x = x * mask)
        L.check(lib.vmmt_mul(L.BF16, A.data_ptr(), ld, M.data_ptr(), ld, A.data_ptr(), ld, R, Cc, None), "mul in place")
        res.append(A)
        torch.cuda.synchronize()
        assert all((o[:, Cc:] == (7.0 if o is A else 3.0)).all() for o in res)
        outs[ld] = [o[:, :Cc].clone() for o in res]
    assert all(torch.equal(x, y_) for x, y_ in zip(outs[512], outs[513]))
    want = (a.float() * mk.float()).to(T)
    assert torch.equal(outs[512][0].cpu(), want)
    want_t = (dy32 * mk.float() * (1 - yv.float() ** 2)).to(T)
    assert (outs[512][1].cpu().float() - want_t.float()).abs().max().item() <= 2e-2 * want_t.float().abs().max().item()


def test_probe_where_reports_xcd_and_cu_of_every_workgroup():
    """vmmt_probe_where (diagnostic behind tools/probe_cu_mask.py): every workgroup of a resident grid reports the XCD and the CU it runs on --
    8 XCDs, at most 256 distinct CUs, and a grid that outlasts its own dispatch spreads over most of them"""
    L, lib = _lib()
    n = 1024
    out = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    L.check(lib.vmmt_probe_where(out.data_ptr(), n, 256, 100, None), "probe")
    torch.cuda.synchronize()
    v = out.cpu().numpy().astype("uint32")
    assert (v != 0xFFFFFFFF).all()
    xcc, hw = v & 15, v >> 8
    assert set(xcc.tolist()) <= set(range(8)) and len(set(xcc.tolist())) == 8
    cus = set(zip(xcc.tolist(), ((hw >> 13) & 7).tolist(), ((hw >> 12) & 1).tolist(), ((hw >> 8) & 15).tolist()))
    assert 128 <= len(cus) <= 256
    assert lib.vmmt_probe_where(out.data_ptr(), 0, 256, 100, None) == 1          # VMMT_EINVAL
