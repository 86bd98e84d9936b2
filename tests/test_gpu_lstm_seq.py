"""GPU: the persistent recurrence kernel (csrc/lstm_seq.hip: one launch per sequence, W_hh resident in LDS, in-launch hand-off of
h_t between the workgroups of a row group) against the per-step kernels it replaces -- BIT-identical outputs (same MFMA chains,
same fold order, same cell arithmetic), at the benchmark shape, with ragged lengths, both directions, several row groups, and
repeated launches on the same buffers (stale-line hazards show up as mismatches on a re-run)."""
import pytest
import torch

from oracle import vi1_oracle as O

pytestmark = pytest.mark.gpu


def _engine(c, p, persistent, dropout=0.0):
    from variational_mmt_amd.engine import Dims, Engine
    e = Engine(Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, dropout, conditional=c.conditional), dtype="bf16", device="cuda:0")
    e.persistent_lstm = persistent
    e.load_state_dict(p)
    return e


@pytest.mark.parametrize("hid,layers,brnn,B,S,T,cond", [(512, 1, True, 256, 20, 21, False), (256, 2, True, 70, 9, 12, False),
                                                        (128, 1, False, 33, 7, 8, False), (512, 2, False, 40, 6, 7, False),
                                                        (128, 1, True, 24, 6, 9, True)])
def test_persistent_recurrence_is_bit_identical_to_the_step_kernels(hid, layers, brnn, B, S, T, cond):
    c = O.Cfg(vs=97, vt=101, emb=64, hid=hid, z=32, layers=layers, brnn=brnn, conditional=cond)
    p = O.init_params(c, seed=3)
    # three DIFFERENT batches through the same buffers, then the first one again: a consumer that read a stale line (its L1 / L2
    # still holding the previous run's h_t at that address) would reproduce the previous batch's values, not this one's
    bts = [O.synth_batch(c, B=B, S=S, T=T, n_img=64, seed=5 + i, fixed_len=False) for i in range(3)]
    bts.append(bts[0])
    outs = []
    for persistent in (False, True):
        e = _engine(c, p, persistent)
        e.set_image_table(bts[0]["table"])
        snaps = []
        for bt in bts:
            ws = e.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=True, eps=bt["eps"], tgt_len=bt["tgt_len"])
            torch.cuda.synchronize()
            snap = dict(enc_out=[b.t.clone() for b in ws.enc_out], enc_gates=[b.t.clone() for b in ws.enc_gates],
                        enc_c=[b.t.clone() for b in ws.enc_c], hn=[b.t.clone() for b in ws.hn], cn=[b.t.clone() for b in ws.cn],
                        cat=ws.cat.t.clone(), dec_gates=[b.t.clone() for b in ws.dec_gates], dec_c=[b.t.clone() for b in ws.dec_c],
                        AH=ws.AH.t.clone())
            if cond:
                snap.update(enct_out=[b.t.clone() for b in ws.enct_out], enct_c=[b.t.clone() for b in ws.enct_c])
            snaps.append(snap)
        if persistent:
            assert any(name == "vmmt_lstm_seq_fwd" for _f, _a, name, _k, _s in ws.plan_fwd_train)
            assert e.lstm_seq_errors() and all(x == 0 for x in e.lstm_seq_errors())       # every in-launch wait completed
        outs.append(snaps)
    for i, (a, b) in enumerate(zip(*outs)):
        for k in a:
            for x, y in zip(a[k] if isinstance(a[k], list) else [a[k]], b[k] if isinstance(b[k], list) else [b[k]]):
                assert torch.equal(x, y), ("persistent differs from per-step", i, k, (x.float() - y.float()).abs().max().item())
    assert not torch.equal(outs[1][0]["AH"], outs[1][1]["AH"])                              # the batches do differ
    for k in outs[1][0]:                                                                    # batch 0 again: same bits as the first time
        for x, y in zip(*(v[k] if isinstance(v[k], list) else [v[k]] for v in (outs[1][0], outs[1][3]))):
            assert torch.equal(x, y), ("re-run differs", k)
