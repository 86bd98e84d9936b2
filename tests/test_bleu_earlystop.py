"""BLEU restatement (onmt/bleu.py) against what tools/multi-bleu.perl itself printed (tests/golden/bleu_cases.json, made by
oracle/make_bleu_golden.py), and the EarlyStop mirror's decisions against the reference's own EarlyStop on score sequences
(tests/golden/earlystop_cases.json)."""
import json
import os
import pickle
import types

import pytest

from variational_mmt_amd.onmt import bleu
from variational_mmt_amd.onmt.EarlyStop import EarlyStop

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BLEU = json.load(open(os.path.join(G, "bleu_cases.json"), encoding="utf-8"))
STOP = json.load(open(os.path.join(G, "earlystop_cases.json")))


def _lines(text):
    out = text.split("\n")
    if out and out[-1] == "":
        out.pop()
        return [x + "\n" for x in out]
    return [x + "\n" for x in out[:-1]] + [out[-1]]


@pytest.mark.parametrize("i", range(len(BLEU)))
def test_bleu_line_identical_to_the_script(i):
    c = BLEU[i]
    r = bleu.multi_bleu(_lines(c["hyp"]), [_lines(c["ref"])], lowercase=c["lc"])
    assert (r["line"] or "") == c["line"]


@pytest.mark.parametrize("i", range(len(BLEU)))
def test_piped_score_with_bpe_joined(i, tmp_path):
    c = BLEU[i]
    hp, rp = tmp_path / "hyp", tmp_path / "ref"
    hp.write_text(c["hyp"], encoding="utf-8")
    rp.write_text(c["ref"], encoding="utf-8")
    assert bleu.score_files(str(hp), str(rp), bpe=True) == c["piped"]


def test_debpe():
    assert bleu.debpe("ein gro@@ ßer ro@@ ter hund@@") == "ein großer roter hund"
    assert bleu.debpe("kind@@ ern @@") == "kindern "


@pytest.mark.parametrize("i", range(len(STOP)))
def test_early_stop_decisions_match_reference(i):
    c = STOP[i]
    es = EarlyStop("src", "tgt", "bleu", 0, 500, c["patience"], multimodal_model_type="vi-model1", img_fname="x")
    it = iter(c["scores"])
    es.translate_ = lambda *a: None
    es.compute_bleus = lambda *a: ([""], [str(next(it))], [""])
    for n, (b, s) in enumerate(zip(c["is_best"], c["stop"])):
        assert bool(es.add_run("snapshot", (n + 1) * 500)) == b
        assert es.signal_early_stopping == s


def test_constructor_contract():
    with pytest.raises(AssertionError):
        EarlyStop("s", "t", "rouge", 0, 500, 10)
    with pytest.raises(AssertionError):
        EarlyStop("s", "t", "bleu", 0, 500, 10, multimodal_model_type="vi-model1")      # image features file missing
    es = EarlyStop("s", "t", "perplexity", 0, 500, 10)
    assert es.add_run("x", 500) is False and es.batch_size == 1 and es.beam_size == 1
    with pytest.raises(RuntimeError):
        EarlyStop("s", "t", "bleu", 0, 500, 10).translate_("a", "b", "c")               # no live model attached


def test_drop_metric_scores_files(tmp_path):
    """TrainerMultimodal.drop_metric_scores (reference :491-551): file names and contents"""
    from variational_mmt_amd.onmt.TrainerMultimodal import TrainerMultimodal
    tr = TrainerMultimodal.__new__(TrainerMultimodal)
    tr.n_model_updates = 1500
    tr.early_stop = types.SimpleNamespace(early_stop_criteria="bleu", results_bleu={500: 20.5, 1000: 31.25, 1500: 31.25},
                                          results_meteor={500: 40.0, 1000: 50.0, 1500: 51.0})
    opt = types.SimpleNamespace(save_model=str(tmp_path / "m"))
    f = tr.drop_metric_scores(opt, 1, None, None, overwrite=True, checkpoint_type="best")
    assert f.endswith("m_BestModelBleu.pkl")
    assert pickle.load(open(f, "rb")) == {"n_updates": 1500, "bleu": 31.25, "meteor": 51.0}     # ties: the later evaluation
    f = tr.drop_metric_scores(opt, 1, None, None, overwrite=True, checkpoint_type="last")
    assert f.endswith("m_MostCurrentModel.pkl")
    assert pickle.load(open(f, "rb")) == {"n_updates": 1500, "bleu": [20.5, 31.25, 31.25], "meteor": [40.0, 50.0, 51.0]}
