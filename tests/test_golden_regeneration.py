"""CPU, build container only: the committed golden fixtures are what the committed generator makes of the REAL reference today.

`oracle/make_golden.py` imports the reference from /root/reference (oracle/ref_harness.py) -- which exists in the build container and
nowhere else: on the GPU box this test skips.  One tiny case is regenerated into a temporary directory and every array is compared
with the committed fixture: same key set (incl. the per-token NLL `f_tok_nll` and the 12-entry `cfg`), same shapes, same BITS.
(hygiene: six of eight step fixtures predated two keys of the generator; all were regenerated in round 6.)"""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "onmt")), reason="the reference tree is only mounted in the build container")
@pytest.mark.parametrize("name", ["tiny_bi_l1", "cond_uni_l2"])
def test_committed_fixture_is_what_the_generator_writes(tmp_path, name, monkeypatch):
    from oracle import make_golden as MG
    monkeypatch.setattr(MG, "OUT", str(tmp_path))
    cwd = os.getcwd()
    try:
        MG.run_case(name, *MG.CASES[name])
    finally:
        os.chdir(cwd)                      # (the harness imports the reference with its directory as the working directory)
    new = np.load(os.path.join(str(tmp_path), name + ".npz"), allow_pickle=True)
    old = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=True)
    assert sorted(new.files) == sorted(old.files), (sorted(set(new.files) ^ set(old.files)))
    assert "f_tok_nll" in old.files and old["cfg"].shape == (12,)
    for k in old.files:
        assert new[k].shape == old[k].shape and new[k].dtype == old[k].dtype, k
        assert np.array_equal(new[k], old[k], equal_nan=True), (k, float(np.abs(new[k].astype(np.float64) - old[k].astype(np.float64)).max()))


def test_every_step_fixture_carries_the_generators_current_keys():
    """(runs everywhere) no fixture predates a key of oracle/make_golden.py"""
    from tests.golden_util import CASES, COND_CASES
    for name in list(CASES) + list(COND_CASES) + ["cfg1_shape", "script_shape"]:
        z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=True)
        assert "f_tok_nll" in z.files and z["cfg"].shape == (12,), name
