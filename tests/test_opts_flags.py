"""The build's flag module (variational_mmt_amd/opts.py) against the reference's parsers: every flag of model_opts / train_opts /
train_mm_vi_model1_opts / translate_opts / translate_mm_vi_opts with the same spelling, destination, default, type, choices, nargs,
action kind and required-ness -- from the committed listing (tests/golden/opts_flags.json, written by oracle/make_opts_golden.py from
the real reference), and the run scripts' command lines parse to the same values."""
import argparse
import json
import os

import pytest

from oracle.make_opts_golden import FUNCS, listing
from variational_mmt_amd import opts

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _norm(rows):
    out = {}
    for r in rows:
        d = dict(r)
        if isinstance(d["default"], float) and d["default"] == int(d["default"]) and d["type"] == "float":
            d["default"] = float(d["default"])
        out[tuple(d.pop("flags"))] = d
    return out


@pytest.mark.parametrize("fn", FUNCS)
def test_flags_match_the_reference_listing(fn):
    ref = _norm(json.load(open(os.path.join(G, "opts_flags.json")))[fn])
    mine = _norm(json.loads(json.dumps(listing(opts)))[fn])
    assert list(mine) == list(ref), "flags or their order differ"
    for k in ref:
        a, b = dict(mine[k]), dict(ref[k])
        for d in (a, b):
            if isinstance(d["default"], (int, float)) and not isinstance(d["default"], bool):
                d["default"] = float(d["default"])
        assert a == b, (k, a, b)


def _parser():
    p = argparse.ArgumentParser()
    opts.model_opts(p)
    opts.train_opts(p)
    opts.train_mm_vi_model1_opts(p)
    return p


def test_run_script_command_line():
    """the command of run_translated_m30k_only.sh:46-57 (fixed prior) and :59-71 (--conditional)"""
    argv = ("-data D -save_model M -gpuid 0 -epochs 30 -batch_size 40 -path_to_train_img_feats tr.hdf5 -path_to_valid_img_feats va.hdf5 "
            "-optim adam -learning_rate 0.002 -use_global_image_features --multimodal_model_type vi-model1 --z_latent_dim 500 "
            "-dropout 0.5 -start_decay_at 8 -overwrite_model_file -evaluate_every_n_model_updates 500 -early_stopping_criteria bleu "
            "-src v.en -tgt v.de -patience 10").split()
    # (the run scripts spell it `--use_global_image_features`; argparse's prefix matching is not relied upon)
    argv[argv.index("-use_global_image_features")] = "--use_global_image_features"
    o = opts.finalise(_parser().parse_args(argv))
    assert (o.rnn_size, o.src_word_vec_size, o.enc_layers, o.dec_layers, o.encoder_type, o.brnn) == (500, 500, 2, 2, "rnn", False)
    assert o.gpuid == [0] and o.z_latent_dim == 500 and o.optim == "adam" and o.learning_rate == 0.002 and o.batch_size == 40
    assert o.max_grad_norm == 5 and o.param_init == 0.1 and o.dropout == 0.5 and not o.conditional and o.image_loss == "logprob"
    assert o.early_stopping_criteria == "bleu" and o.patience == 10 and o.overwrite_model_file
    o2 = opts.finalise(_parser().parse_args(argv + ["--conditional", "-layers", "1", "-word_vec_size", "620", "-encoder_type", "brnn"]))
    assert o2.conditional and (o2.enc_layers, o2.dec_layers, o2.src_word_vec_size, o2.tgt_word_vec_size, o2.brnn) == (1, 1, 620, 620, True)


def test_required_deprecated_and_sru():
    p = _parser()
    base = ["-data", "D", "-path_to_train_img_feats", "a", "-path_to_valid_img_feats", "b", "--multimodal_model_type", "vi-model1"]
    with pytest.raises(SystemExit):
        p.parse_args(base)                                           # --z_latent_dim is required
    ok = base + ["--z_latent_dim", "8"]
    assert p.parse_args(ok).rnn_type == "LSTM"
    assert p.parse_args(ok + ["-rnn_type", "GRU"]).rnn_type == "GRU"      # parses; the model constructor refuses it
    with pytest.raises(AssertionError):
        p.parse_args(ok + ["-rnn_type", "SRU"])                          # CheckSRU: not available on this build
    with pytest.raises((SystemExit, argparse.ArgumentTypeError)):
        p.parse_args(ok + ["-brnn"])                                     # deprecated flag


def test_md_help(capsys):
    p = argparse.ArgumentParser(description="x")
    opts.add_md_help_argument(p)
    opts.translate_opts(p)
    opts.translate_mm_vi_opts(p)
    with pytest.raises(SystemExit):
        p.parse_args(["-md"])
    out = capsys.readouterr().out
    assert "**-beam_size**" in out and "**-path_to_test_img_feats**" in out
