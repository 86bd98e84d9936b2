"""GPU: beam search (variational_mmt_amd.decode.beam_decode + the host Beam mirror) against fixtures
produced by the reference's own TranslatorMultimodalVI.translate_batch + Beam + GNMTGlobalScorer (one sentence per call, as
translate_mm_vi.py:80-82 forces), and the beam kernel against a torch restatement of Beam.advance."""
import ctypes as C
import types

import pytest
import torch

from tests.golden_util import BEAM_CASES, load

pytestmark = pytest.mark.gpu


def _model(c, p, dtype):
    from variational_mmt_amd.engine import Dims
    from variational_mmt_amd.onmt.Models import NMTVIModel
    m = NMTVIModel(Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, 0.0, conditional=c.conditional), dtype=dtype,
                   device="cuda", param_init=0.0, conditional=c.conditional)
    m.engine.load_state_dict(p)
    return m


@pytest.mark.parametrize("batched", [False, True])
@pytest.mark.parametrize("name", BEAM_CASES)
def test_beam_search_matches_reference(name, batched):
    from variational_mmt_amd.onmt.translate import GNMTGlobalScorer, TranslatorMultimodalVI
    c, p, bt, z, (B, S, max_len) = load(name)
    K, n_best, min_length = [int(x) for x in z["beam"]]
    alpha, beta = [float(x) for x in z["scorer"]]
    model = _model(c, p, "f32")
    fields = {"tgt": types.SimpleNamespace(vocab=types.SimpleNamespace(stoi={"<blank>": 1, "<s>": 2, "</s>": 3}))}
    tr = TranslatorMultimodalVI(model, fields, beam_size=K, n_best=n_best, max_length=max_len, min_length=min_length,
                                global_scorer=GNMTGlobalScorer(alpha, beta))
    if batched:      # all sentences in one decoding batch of K*B rows (lengths are sorted descending in the fixture)
        groups = [list(range(B))]
    else:            # the reference's way: one sentence per call
        groups = [[b] for b in range(B)]
    for g in groups:
        n0 = int(bt["src_len"][g[0]])
        ret = tr.translate_batch(types.SimpleNamespace(src=(bt["src"][:n0, g], bt["src_len"][g])))
        for j, b in enumerate(g):
            bm = tr.last_beams[j]
            steps = int(z["steps"][b])
            n = int(bt["src_len"][b])
            assert len(bm.prev_ks) == steps
            assert torch.equal(torch.stack(bm.next_ys), torch.from_numpy(z["hist_next"][b, :steps + 1]))
            assert torch.equal(torch.stack(bm.prev_ks), torch.from_numpy(z["hist_prev"][b, :steps]))
            hs = torch.stack(bm.all_scores[1:] + [bm.scores])
            assert (hs - torch.from_numpy(z["hist_score"][b, :steps])).abs().max().item() <= 2e-4
            for i in range(n_best):
                m = int(z["pred_len"][b, i])
                assert ret["predictions"][j][i] == z["pred"][b, i, :m].tolist()
                assert abs(ret["scores"][j][i] - float(z["score"][b, i])) <= 5e-4
                a = ret["attention"][j][i]
                assert tuple(a.shape) == (m, n)
                assert (a - torch.from_numpy(z["attention"][b, i, :m, :n])).abs().max().item() <= 2e-4


def test_beam_search_bf16_runs_and_scores_are_consistent():
    """bf16 storage: beams may diverge from the fp32 reference at near-ties; what must hold is internal consistency --
    every recorded score is its parent's score plus a log-probability <= 0, tokens are in range, parents < K."""
    from variational_mmt_amd.decode import beam_decode
    c, p, bt, z, (B, S, max_len) = load("beam_bi_l1")
    K = int(z["beam"][0])
    model = _model(c, p, "bf16")
    rec = beam_decode(model.engine, bt["src"], bt["src_len"], K, max_len=max_len)
    sc, pv, nx = rec["scores"], rec["prev"].long(), rec["next"]
    assert sc.shape == (max_len, B, K) and (nx >= 0).all() and (nx < c.vt).all() and (pv >= 0).all() and (pv < K).all()
    assert (sc[:, :, :-1] >= sc[:, :, 1:]).all()                      # sorted best first
    alive = sc[1:] > -1e19
    parent = torch.gather(sc[:-1], 2, pv[1:])
    assert ((sc[1:] <= parent + 1e-5) | ~alive).all()
    # the first position agrees with the fp32 reference up to bf16 rounding of the log-probabilities
    assert (sc[0] - torch.from_numpy(z["hist_score"][:, 0])).abs().max().item() <= 0.1


def _advance_ref(logits, scores, cur, first, mask_eos, eos, K):
    """Beam.advance's arithmetic for one sentence (Beam.py:77-103) in torch"""
    V = logits.shape[1]
    wp = torch.log_softmax(logits, 1)
    if mask_eos:
        wp[:, eos] = -1e20
    if first:
        cand = wp[0]
    else:
        cand = wp + scores.unsqueeze(1)
        cand[cur == eos] = -1e20
    best, ids = cand.reshape(-1).topk(K, 0, True, True)
    pk = torch.div(ids, V, rounding_mode="floor")
    return best, pk, ids - pk * V


@pytest.mark.parametrize("B,K,V", [(1, 5, 30000), (7, 3, 1000), (3, 12, 501), (2, 1, 64), (4, 16, 257), (30, 5, 4100), (2, 16, 2048)])
def test_beam_advance_kernel(B, K, V):
    from variational_mmt_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(B * 1000 + K)
    R = K * B
    ld = (V + 63) // 64 * 64
    eos = 3
    for first, mask_eos in [(1, 0), (0, 0), (0, 1), (1, 1)]:
        logits = torch.randn(R, ld, generator=g) * 3
        scores = -torch.rand(B, K, generator=g).cumsum(1)
        cur = torch.randint(4, V, (K, B), generator=g)
        if not first and K > 1:
            cur[K // 2, :] = eos                                         # a finished beam in every sentence
        d = lambda t: t.cuda()
        lg, scd, curd = d(logits), d(scores.clone()), d(cur.reshape(-1))
        nxt = torch.zeros(R, dtype=torch.int64, device="cuda")
        sel = torch.zeros(R, dtype=torch.int64, device="cuda")
        hs = torch.zeros(B, K, device="cuda")
        hp = torch.zeros(B, K, dtype=torch.int32, device="cuda")
        hn = torch.zeros(B, K, dtype=torch.int64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        P = lambda t: C.c_void_p(t.data_ptr())
        wsb = torch.zeros(lib.vmmt_beam_advance_ws_bytes(B, K, V) // 4, device="cuda")
        L.check(lib.vmmt_beam_advance(P(lg), ld, B, K, V, P(curd), P(scd), first, mask_eos, eos, P(nxt), P(sel), P(hs), P(hp), P(hn),
                                      P(wsb), wsb.numel() * 4, C.c_void_p(st)), "vmmt_beam_advance")
        torch.cuda.synchronize()
        for b in range(B):
            rows = torch.arange(K) * B + b
            best, pk, tok = _advance_ref(logits[rows, :V].clone(), scores[b], cur[:, b], first, mask_eos, eos, K)
            live = best > -1e19                                           # among -1e20 candidates the choice is arbitrary
            assert (hs[b].cpu() - best).abs()[live].max().item() <= 1e-4
            assert torch.equal(hp[b].cpu().long()[live], pk[live]) and torch.equal(hn[b].cpu()[live], tok[live])
            assert torch.equal(scd[b].cpu(), hs[b].cpu())
            assert torch.equal(nxt.cpu()[rows], hn[b].cpu()) and torch.equal(sel.cpu()[rows], hp[b].cpu().long() * B + b)
    with pytest.raises(RuntimeError):
        L.check(lib.vmmt_beam_advance(None, ld, B, K, V, None, None, 0, 0, eos, None, None, None, None, None, None, 0, None), "beam")
    with pytest.raises(RuntimeError):     # K > 16
        L.check(lib.vmmt_beam_advance(P(lg), ld, B, 17, V, P(curd), P(scd), 0, 0, eos, P(nxt), P(sel), P(hs), P(hp), P(hn), P(wsb),
                                      wsb.numel() * 4, None), "beam")
    with pytest.raises(RuntimeError):     # scratch too small
        L.check(lib.vmmt_beam_advance(P(lg), ld, B, K, V, P(curd), P(scd), 0, 0, eos, P(nxt), P(sel), P(hs), P(hp), P(hn), P(wsb),
                                      wsb.numel() * 4 - 4, None), "beam")


def test_rows_select_kernel():
    from variational_mmt_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(5)
    for dt, cols, ld in [(torch.float32, 33, 40), (torch.bfloat16, 64, 128), (torch.bfloat16, 31, 33), (torch.float32, 512, 512)]:
        src = torch.randn(20, ld, generator=g).to(dt).cuda()
        dst = torch.zeros(9, ld + 2 * (ld % 2) + 4, dtype=dt, device="cuda")
        rows = torch.tensor([3, 3, 19, 0, 7, 1, 1, 12, 5], dtype=torch.int64, device="cuda")
        esz = src.element_size()
        if (cols * esz) % 2:
            continue
        L.check(lib.vmmt_rows_select(C.c_void_p(src.data_ptr()), ld * esz, C.c_void_p(rows.data_ptr()), C.c_void_p(dst.data_ptr()),
                                     dst.shape[1] * esz, 9, cols * esz, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rows_select")
        assert torch.equal(dst[:, :cols].cpu(), src[rows][:, :cols].cpu())
        assert (dst[:, cols:] == 0).all()
