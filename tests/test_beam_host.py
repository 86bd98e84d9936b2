"""Host mirror of onmt/translate/Beam.py (variational_mmt_amd.onmt.translate.Beam: the bookkeeping that consumes the device
beam search's per-position records) on the CPU: fed with the records of the oracle's beam search it must reproduce the
n-best lists, scores and attention matrices of the reference's own translate_batch (tests/golden/beam_*.npz)."""
import pytest
import torch

from oracle import vi1_oracle as O
from tests.golden_util import BEAM_CASES, load
from variational_mmt_amd.onmt.translate.Beam import Beam, GNMTGlobalScorer


@pytest.mark.parametrize("name", BEAM_CASES)
def test_replay_reproduces_reference_results(name):
    c, p, bt, z, (B, S, max_len) = load(name)
    K, n_best, min_length = [int(x) for x in z["beam"]]
    alpha, beta = [float(x) for x in z["scorer"]]
    for b in range(B):
        n = int(bt["src_len"][b])
        r = O.beam_search(p, c, bt["src"][:n, b], K, n_best=n_best, max_len=max_len, alpha=alpha, beta=beta, min_length=min_length)
        bm = Beam(K, 1, 2, 3, n_best=n_best, global_scorer=GNMTGlobalScorer(alpha, beta), min_length=min_length)
        # more positions than the reference ran are offered: the mirror must stop consuming at done()
        for t in range(r["steps"]):
            assert not bm.done()
            bm.advance_from_device(r["hist_score"][t], r["hist_prev"][t].to(torch.int32), r["hist_next"][t + 1], r["attn_rows"][t])
        assert r["steps"] == max_len or bm.done()
        scores, ks = bm.sort_finished(minimum=n_best)
        for i, (times, k) in enumerate(ks[:n_best]):
            hyp, att = bm.get_hyp(times, k)
            m = int(z["pred_len"][b, i])
            assert [int(x) for x in hyp] == z["pred"][b, i, :m].tolist()
            assert abs(float(scores[i]) - float(z["score"][b, i])) <= 1e-4
            assert (att - torch.from_numpy(z["attention"][b, i, :m, :n])).abs().max().item() <= 1e-5


def test_sort_finished_pads_with_beam_zero_as_executed():
    bm = Beam(3, 1, 2, 3, n_best=2)
    bm.advance_from_device(torch.tensor([-1.0, -2.0, -3.0]), torch.tensor([0, 0, 0]), torch.tensor([7, 8, 9]))
    bm.advance_from_device(torch.tensor([-1.5, -2.5, -3.5]), torch.tensor([0, 1, 0]), torch.tensor([5, 3, 6]))
    assert not bm.done() and len(bm.finished) == 1                     # </s> on beam 1, not on top
    scores, ks = bm.sort_finished(minimum=2)
    assert ks == [(2, 0), (2, 1)] and [float(s) for s in scores] == [-1.5, -2.5]
    hyp, _ = bm.get_hyp(2, 1)
    assert [int(x) for x in hyp] == [8, 3]


@pytest.mark.parametrize("name", BEAM_CASES)
@pytest.mark.parametrize("with_attn,scorer", [(True, False), (False, False), (True, True)])
def test_load_records_equals_position_by_position(name, with_attn, scorer):
    """Beam.load_records (whole-array bookkeeping, what TranslatorMultimodalVI._replay uses without a global scorer) leaves the beam in
    the state T calls of advance_from_device leave it in: same finished list in the same order, same back-pointers, tokens,
    attention, and therefore the same n-best extraction"""
    c, p, bt, z, (B, S, max_len) = load(name)
    K, n_best, min_length = [int(x) for x in z["beam"]]
    for b in range(B):
        n = int(bt["src_len"][b])
        alpha, beta = ([float(x) for x in z["scorer"]] if scorer else (0.0, 0.0))
        if scorer and alpha == 0.0 and beta == 0.0:
            alpha, beta = 0.6, 0.2                           # the fixture ran without re-scoring: exercise it all the same
        r = O.beam_search(p, c, bt["src"][:n, b], K, n_best=n_best, max_len=max_len, alpha=alpha, beta=beta, min_length=min_length)
        T = r["steps"]
        sc = torch.stack([r["hist_score"][t] for t in range(T)])
        pv = torch.stack([r["hist_prev"][t].to(torch.int32) for t in range(T)])
        nx = torch.stack([r["hist_next"][t + 1] for t in range(T)])
        at = torch.stack([r["attn_rows"][t] for t in range(T)]) if with_attn else None
        mk = lambda: Beam(K, 1, 2, 3, n_best=n_best, min_length=min_length, global_scorer=GNMTGlobalScorer(alpha, beta) if scorer else None)
        one, two = mk(), mk()
        for t in range(T):
            one.advance_from_device(sc[t], pv[t], nx[t], None if at is None else at[t])
        two.load_records(sc, pv, nx, at)
        assert one.eos_top == two.eos_top and one.done() == two.done() and len(one.finished) == len(two.finished)
        for (s1, t1, k1), (s2, t2, k2) in zip(one.finished, two.finished):
            assert (t1, k1) == (t2, k2) and float(s1) == float(s2)
        assert torch.equal(one.scores, two.scores) and len(one.all_scores) == len(two.all_scores)
        for x, y in zip(one.prev_ks + one.next_ys + one.all_scores + one.attn, two.prev_ks + two.next_ys + two.all_scores + two.attn):
            assert torch.equal(x, y)
        assert len(one.attn) == len(two.attn) == (T if with_attn else 0)
        if scorer:
            assert torch.equal(one.global_state["coverage"], two.global_state["coverage"])
        s1, k1 = one.sort_finished(minimum=n_best)
        s2, k2 = two.sort_finished(minimum=n_best)
        assert k1 == k2 and [float(v) for v in s1] == [float(v) for v in s2]
        for times, k in k1[:n_best]:
            h1, a1 = one.get_hyp(times, k)
            h2, a2 = two.get_hyp(times, k)
            assert [int(v) for v in h1] == [int(v) for v in h2] and ((a1 is None and a2 is None) or torch.equal(a1, a2))
