"""GPU: vmmt_attn_fwd / vmmt_attn_bwd against fp64 math on the same (bf16-rounded) operands, for every dispatch path:
   * T', S <= 32: one 32 x 32 tile, Q and the memory both staged in LDS (BASELINE configs 1-4)
   * T', S <= 64, H <= 1024: attn_{fwd,bwd}_big -- 2 x 2 score tiles, the source memory staged in LDS (129 KB at H = 1024), queries /
     gradients straight from global memory into MFMA fragments (BASELINE config 5: S = T' = 64, H = 1024)
   * fp32 parity mode: the generic kernels.
   * S > 64 (up to 256) or T' > 64: the one-wave-per-query kernels (long sources at translation time).
Reference arithmetic: onmt/modules/GlobalAttention.py:113 (bmm), :171-176 (mask), :179-180 (softmax), :184 (bmm)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ru(x, m):
    return (x + m - 1) // m * m


def _buf(rows, cols, dtype, fill=None):
    t = torch.zeros(_ru(rows, 64) + 64, _ru(cols, 64), dtype=dtype, device="cuda")
    if fill is not None:
        t[:rows, :cols] = fill.to(device="cuda", dtype=dtype)
    return t


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("Tp,S,H,B", [(20, 20, 512, 9), (64, 64, 1024, 5), (33, 17, 1024, 4), (7, 64, 96, 6), (64, 9, 512, 3),
                                     (50, 41, 256, 7), (1, 64, 1024, 8), (24, 30, 1024, 3),
                                     # beyond the per-sentence kernels (S > 64 or T' > 64): attn_fwd_long / vmmt_attn_bwd_long
                                     (5, 65, 96, 4), (70, 100, 512, 3), (1, 256, 1024, 2), (130, 20, 128, 5), (66, 200, 500, 2)])
def test_attention_forward_backward(Tp, S, H, B, dtype):
    from variational_mmt_amd import _lib as L
    lib = L.lib()
    T = torch.bfloat16 if dtype == "bf16" else torch.float32
    code = L.BF16 if dtype == "bf16" else L.F32
    g = torch.Generator().manual_seed(Tp * 1000 + S * 10 + B)
    lens = torch.randint(1, S + 1, (B,), generator=g)
    lens[0] = S
    lens, _ = torch.sort(lens, descending=True)
    q = (torch.randn(Tp, B, H, generator=g) * 0.3).to(T)
    ctx = (torch.randn(S, B, H, generator=g) * 0.3).to(T)
    for b in range(B):
        ctx[lens[b]:, b] = 0                       # encoder memory is zero at pads (packed sequence)
    dcat_l = (torch.randn(Tp, B, H, generator=g) * 0.1).to(T)
    # ---- fp64 reference on the rounded operands
    qd, cd, dd = q.double(), ctx.double(), dcat_l.double()
    sc = torch.einsum("tbh,sbh->tbs", qd, cd)
    mask = torch.arange(S).view(1, 1, S) >= lens.view(1, B, 1)
    sc = sc.masked_fill(mask, float("-inf"))
    pr = torch.softmax(sc, dim=2)
    cvec = torch.einsum("tbs,sbh->tbh", pr, cd)
    dp = torch.einsum("tbh,sbh->tbs", dd, cd)
    ds = pr * (dp - (pr * dp).sum(2, keepdim=True))
    dq_ref = torch.einsum("tbs,sbh->tbh", ds, cd)
    dctx_ref = torch.einsum("tbs,tbh->sbh", pr, dd) + torch.einsum("tbs,tbh->sbh", ds, qd)
    dctx_ref = dctx_ref.masked_fill((torch.arange(S).view(S, 1, 1) >= lens.view(1, B, 1)), 0.0)
    # ---- device buffers (time-major rows t*B + b, as the engine lays them out)
    M, MS = Tp * B, S * B
    Q = _buf(M, H, T, q.reshape(M, H))
    CTX = _buf(MS, H, T, ctx.reshape(MS, H))
    CAT = _buf(M, 2 * H, T)
    DCAT = _buf(M, 2 * H, T)
    DCAT[:M, :H] = dcat_l.reshape(M, H).cuda()
    DQ = _buf(M, H, T)
    DCTX = _buf(MS, H, T)
    DCTX.fill_(7.0)                                # every valid row must be overwritten
    probs = torch.zeros(M * S, dtype=torch.float32, device="cuda")
    ld = lens.cuda()
    P = lambda t: C.c_void_p(t.data_ptr())
    L.check(lib.vmmt_attn_fwd(code, P(Q), Q.shape[1], P(CTX), CTX.shape[1], P(ld), P(CAT), CAT.shape[1], P(probs), Tp, B, S, H, None), "attn fwd")
    torch.cuda.synchronize()
    got_p = probs.view(Tp, B, S).cpu().double()
    tol_p, tol_c, tol_g = (2e-2, 2e-2, 3e-2) if dtype == "bf16" else (1e-5, 1e-5, 2e-5)
    assert (got_p - pr).abs().max().item() <= tol_p * (1.0 if dtype == "bf16" else 1.0)
    assert (got_p.sum(2) - 1).abs().max().item() <= 1e-5
    assert (got_p[mask.expand(Tp, B, S)] == 0).all()
    got_c = CAT[:M, :H].float().cpu().double().view(Tp, B, H)
    assert (got_c - cvec).abs().max().item() <= tol_c * max(1.0, cvec.abs().max().item())
    assert (CAT[:M, H:] == 0).all() and (CAT[M:] == 0).all()              # only the left half of [c ; r], only valid rows
    # backward consumes the probabilities the forward stored
    if S > 64 or Tp > 64:
        assert lib.vmmt_attn_bwd(code, P(DCAT), DCAT.shape[1], P(probs), P(Q), Q.shape[1], P(CTX), CTX.shape[1], P(ld), P(DQ), DQ.shape[1],
                                 P(DCTX), DCTX.shape[1], Tp, B, S, H, None) != 0 or (S <= 64 and dtype == "f32")
        dots = torch.zeros(M, device="cuda")
        L.check(lib.vmmt_attn_bwd_long(code, P(DCAT), DCAT.shape[1], P(probs), P(Q), Q.shape[1], P(CTX), CTX.shape[1], P(ld), P(DQ),
                                       DQ.shape[1], P(DCTX), DCTX.shape[1], Tp, B, S, H, P(dots), None), "attn bwd long")
    else:
        L.check(lib.vmmt_attn_bwd(code, P(DCAT), DCAT.shape[1], P(probs), P(Q), Q.shape[1], P(CTX), CTX.shape[1], P(ld), P(DQ), DQ.shape[1],
                                  P(DCTX), DCTX.shape[1], Tp, B, S, H, None), "attn bwd")
    torch.cuda.synchronize()
    got_dq = DQ[:M, :H].float().cpu().double().view(Tp, B, H)
    got_dc = DCTX[:MS, :H].float().cpu().double().view(S, B, H)
    assert (got_dq - dq_ref).abs().max().item() <= tol_g * max(1e-3, dq_ref.abs().max().item())
    assert (got_dc - dctx_ref).abs().max().item() <= tol_g * max(1e-3, dctx_ref.abs().max().item())
    assert (DQ[M:] == 0).all()


@pytest.mark.parametrize("Tp,S,H,B", [(20, 20, 512, 256), (32, 32, 96, 5), (7, 31, 160, 6), (1, 1, 64, 3), (20, 17, 1024, 4)])
def test_attention_backward_small_sentences_low_lds_kernel(Tp, S, H, B):
    """T', S <= 32 in bf16 (BASELINE configs 1-4): `attn_bwd_lite` -- MFMA fragments of dC / Hs straight from global memory, the K-strided
    tiles through one 2.5 KiB buffer per wave, 15 KiB of LDS in all so that it shares a CU with the dWg product's workgroups -- against fp64
    math, and against `attn_bwd_fast` (three staged images), which serves the same call when the rows are not 16-byte aligned."""
    from variational_mmt_amd import _lib as L
    lib = L.lib()
    T = torch.bfloat16
    g = torch.Generator().manual_seed(Tp * 100 + S + H)
    lens = torch.randint(1, S + 1, (B,), generator=g)
    lens[0] = S
    lens, _ = torch.sort(lens, descending=True)
    q = (torch.randn(Tp, B, H, generator=g) * 0.3).to(T)
    ctx = (torch.randn(S, B, H, generator=g) * 0.3).to(T)
    for b in range(B):
        ctx[lens[b]:, b] = 0
    dcat_l = (torch.randn(Tp, B, H, generator=g) * 0.1).to(T)
    qd, cd, dd = q.double(), ctx.double(), dcat_l.double()
    sc = torch.einsum("tbh,sbh->tbs", qd, cd).masked_fill(torch.arange(S).view(1, 1, S) >= lens.view(1, B, 1), float("-inf"))
    pr = torch.softmax(sc, dim=2)
    dp = torch.einsum("tbh,sbh->tbs", dd, cd)
    ds = pr * (dp - (pr * dp).sum(2, keepdim=True))
    dq_ref = torch.einsum("tbs,sbh->tbh", ds, cd)
    dctx_ref = (torch.einsum("tbs,tbh->sbh", pr, dd) + torch.einsum("tbs,tbh->sbh", ds, qd)).masked_fill(
        torch.arange(S).view(S, 1, 1) >= lens.view(1, B, 1), 0.0)
    M, MS = Tp * B, S * B
    probs = pr.float().reshape(M * S).cuda()
    ld = lens.cuda()
    P = lambda t: C.c_void_p(t.data_ptr())
    outs = []
    for pad in (0, 4):                                  # leading dimensions H + 64 (aligned rows) / H + 68 (8-byte rows: the staged kernel)
        def buf(rows, fill=None, val=0.0):
            t = torch.full((rows + 8, H + 64 + pad), val, dtype=T, device="cuda")
            if fill is not None:
                t[:rows, :H] = fill.cuda()
            return t
        Q, CTX, DCAT = buf(M, q.reshape(M, H)), buf(MS, ctx.reshape(MS, H)), buf(M, dcat_l.reshape(M, H))
        DQ, DCTX = buf(M, val=7.0), buf(MS, val=7.0)
        L.check(lib.vmmt_attn_bwd(L.BF16, P(DCAT), DCAT.shape[1], P(probs), P(Q), Q.shape[1], P(CTX), CTX.shape[1], P(ld), P(DQ), DQ.shape[1],
                                  P(DCTX), DCTX.shape[1], Tp, B, S, H, None), "attn bwd")
        torch.cuda.synchronize()
        assert (DQ[:, H:] == 7.0).all() and (DQ[M:] == 7.0).all() and (DCTX[:, H:] == 7.0).all() and (DCTX[MS:] == 7.0).all()
        got_dq = DQ[:M, :H].float().cpu().double().view(Tp, B, H)
        got_dc = DCTX[:MS, :H].float().cpu().double().view(S, B, H)
        assert (got_dq - dq_ref).abs().max().item() <= 3e-2 * max(1e-3, dq_ref.abs().max().item())
        assert (got_dc - dctx_ref).abs().max().item() <= 3e-2 * max(1e-3, dctx_ref.abs().max().item())
        outs.append((got_dq, got_dc))
    # the two kernels differ in the order of the f32 partial sums only
    assert (outs[0][0] - outs[1][0]).abs().max().item() <= 8e-3 * max(1e-3, dq_ref.abs().max().item())
    assert (outs[0][1] - outs[1][1]).abs().max().item() <= 8e-3 * max(1e-3, dctx_ref.abs().max().item())
