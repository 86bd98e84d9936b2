"""Register budgets the step's schedule depends on (CPU: read from variational_mmt_amd/kernel_resources.json, which the build writes from
the compiler's own -Rpass-analysis=kernel-resource-usage remarks).  A gfx950 SIMD has 512 registers per lane for all of its waves
(arch + accumulation registers, allocated in blocks of 8).  The persistent recurrences hold one wave per SIMD for a whole sequence, and the
plan runs GEMMs of other streams on the same CUs next to them: that only happens while the two kernels' registers add up to <= 512.
Measured when an epilogue change took the 128 x 128 NT product from 220 to 236 registers: the step went from 1.775 to 1.88 ms (same box)
-- the products no longer fitted beside lstm_seq_fwd_kernel<512> and waited for it."""
import json
import os

import pytest

from variational_mmt_amd import build as B


@pytest.fixture(scope="module")
def res():
    if not os.path.exists(B.RESOURCES):
        B.build(verbose=False)
    with open(B.RESOURCES) as f:
        return json.load(f)


def _total(v):
    return ((v["vgprs"] + 3) // 4 * 4 + v["agprs"] + 7) // 8 * 8          # unified file: accumulation registers start at a multiple of 4


def _find(res, *parts):
    hits = [k for k in res if all(p in k for p in parts)]
    assert len(hits) == 1, (parts, hits)
    return _total(res[hits[0]])


def test_no_kernel_spills(res):
    assert len(res) > 100
    bad = {k: v for k, v in res.items() if v.get("vgpr_spill", 0) or v.get("scratch", 0)}
    assert not bad, bad


def test_gemms_fit_beside_the_persistent_recurrences(res):
    g = "gemm_kernelItLi128ELi128ELi64ELi64E"        # bf16, 128 x 128 tiles, LDS-DMA loop (the trailing ...Li1E)
    nt, tn, nn = (_find(res, g + lay + "ELi64ELb1ELi1ELi0E") for lay in ("Lb1ELb1", "Lb0ELb0", "Lb1ELb0"))        # (...Li0E: the plain epilogue)
    # (16 sentences x 32 units per workgroup, the W_hh fragments in registers)
    fwd512, bwd512 = _find(res, "lstm_seq_fwd_kernelILi512E"), _find(res, "lstm_seq_bwd_kernelILi512E")
    fwd256, bwd256 = _find(res, "lstm_seq_fwd_kernelILi256E"), _find(res, "lstm_seq_bwd_kernelILi256E")
    # forward: the decoder's recurrence (H = 512) runs beside NT products (image network, target-side input projection)
    assert fwd512 + nt <= 512, (fwd512, nt)
    # backward at H = 512: the recurrence holds its SIMDs ALONE, by measurement (LABNOTES round 5: with a quarter / three eighths / half of
    # its fragments in LDS so that the products fit beside it again -- 280 / 264 / 248 registers -- the step was equal / 20 us / 25 us
    # slower, in five configurations): it only has to fit the file
    assert bwd512 <= 512, bwd512
    # the encoder's directions (H = 256) DO host every product (made exclusive by an LDS pad, the step lost 65 us)
    assert max(fwd256, bwd256) + max(tn, nn, nt) <= 512, (fwd256, bwd256, tn, nn, nt)
    # --conditional: encoder_tgt's recurrences (2 x 256, a stream of their own, ~1 ms each way) are still running when the decoder's start.  Two
    # persistent launches that cannot share a SIMD place their workgroups around each other -- the later one spins until the earlier one is
    # through (+60 us on the decoder's backward, profiles timeline) and, launched at the same moment, two partially placed grids could wait
    # for each other until the hand-off bound: they must fit one register file together
    assert bwd512 + bwd256 <= 512 and fwd512 + fwd256 <= 512 and 2 * max(fwd256, bwd256) <= 512, (fwd512, fwd256, bwd512, bwd256)
    # the grouped weight-gradient launch (vmmt_gemm_group) is a guest of the backward recurrences like the products it replaces
    grp = _find(res, "gemm_group_kernelItLi128ELi128ELi64ELi64ELb0ELb0")
    assert bwd256 + grp <= 512 and grp <= tn, (bwd256, grp, tn)


def test_one_wave_per_simd_kernels_stay_within_the_file(res):
    for name in ("gen2_kernelILi512ELb1E", "gen2w_kernelILb1E", "lstm_seq_fwd_kernelILi1024E", "lstm_seq_bwd_kernelILi1024E"):
        assert _find(res, name) <= 512, name


def test_attention_backward_fits_beside_the_dwg_product(res):
    """attn_bwd_lite exists to run WHILE the generator's dWg product (256 x 128 tiles, 8 waves = 2 per SIMD, 144 KiB of LDS) holds every
    CU: one of its waves per SIMD next to two of the product's, and its 15 KiB of LDS in what the product leaves of a CU's 160"""
    dwg = _find(res, "gemm_kernelItLi256ELi128ELi64ELi64ELb0ELb0ELi64ELb1ELi3E")
    lite = _find(res, "attn_bwd_lite")
    assert 2 * dwg + lite <= 512, (dwg, lite)
    assert 3 * (256 + 128) * 64 * 2 + 6 * 32 * 40 * 2 <= 160 * 1024          # the product's three operand stages + P, dS and four tile buffers


def test_small_elementwise_kernels_fit_beside_the_vocabulary_sweep(res):
    """gen2p_kernel holds one wave per SIMD on 240 CUs; what it leaves of a SIMD's 512 registers per lane is what a kernel of another stream
    may take to run there.  A kernel without LDS that needs more sat in the dispatcher until the sweep had ended (tools/probe_under_sweep.py:
    80 us per launch against 5): the aux stream's small activation backward (4 elements per thread) stays inside that budget"""
    free = 512 - _find(res, "gen2p_kernel", "Lb1")
    assert free >= 24
    for k, v in res.items():
        if "act_bwd8_kernel" in k and "Li4E" in k:
            assert _total(v) <= free, (k, v, free)
    assert sum(1 for k in res if "act_bwd8_kernel" in k and "Li4E" in k) == 2
