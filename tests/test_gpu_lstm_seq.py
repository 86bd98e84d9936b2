"""GPU: the persistent recurrence kernel (csrc/lstm_seq.hip: one launch per sequence, W_hh resident in registers, in-launch hand-off of
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
                                                        (128, 1, True, 24, 6, 9, True),
                                                        # H = 64: two k-steps for the four waves' K quarters (two of them stay empty)
                                                        (64, 1, False, 35, 6, 7, False),
                                                        # H = 1024 (BASELINE config 5's decoder; its encoder: 2 x 512): 64 unit slices per row
                                                        # group, so 200 sentences run as two persistent launches of <= 128 rows
                                                        (1024, 2, True, 200, 7, 6, False), (1024, 1, False, 40, 5, 6, False)])
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
                if hid <= 512:
                    assert torch.equal(x, y), ("persistent differs from per-step", i, k, (x.float() - y.float()).abs().max().item())
                else:       # H = 1024: another (equally valid) order of the f32 additions, see test_persistent_backward_recurrence_is_bit_identical
                    d_ = (x.float() - y.float()).abs()
                    assert d_.max().item() <= 2e-2 and d_.mean().item() <= 2e-4, ("persistent differs from per-step", i, k, d_.max().item(), d_.mean().item())
    assert not torch.equal(outs[1][0]["AH"], outs[1][1]["AH"])                              # the batches do differ
    for k in outs[1][0]:                                                                    # batch 0 again: same bits as the first time
        for x, y in zip(*(v[k] if isinstance(v[k], list) else [v[k]] for v in (outs[1][0], outs[1][3]))):
            assert torch.equal(x, y), ("re-run differs", k)


def _bwd_case(B, H, ndir, T, seed, lens_on):
    """synthetic backward recurrence: descriptors exactly as the engine chains them (dgates_next[t] = dgates_out[t-1]; both
    directions walk the time axis in opposite order), random saved activations, ragged lengths, injected final-state gradients"""
    import ctypes as C
    from variational_mmt_amd import _lib as L
    g = torch.Generator().manual_seed(seed)
    bf = torch.bfloat16
    dev = "cuda"
    M = T * B
    ldg = 4 * H * ndir
    t_ = dict(
        whhT=[(torch.randn(H, 4 * H, generator=g) * 0.05).to(bf).to(dev) for _ in range(ndir)],
        gates=torch.rand(M, ldg, generator=g).to(bf).to(dev),
        c=(torch.randn(M, H * ndir, generator=g) * 0.5).to(dev),
        cn=(torch.randn(B, H * ndir, generator=g) * 0.5).to(dev),
        dha=(torch.randn(M, H * ndir, generator=g) * 0.1).to(bf).to(dev),
        dhn=(torch.randn(B, H * ndir, generator=g) * 0.1).to(dev),
        dcn=(torch.randn(B, H * ndir, generator=g) * 0.1).to(dev),
        dcc0=(torch.randn(B, H * ndir, generator=g) * 0.1).to(dev),
    )
    lens = torch.randint(max(1, T // 2), T + 1, (B,), generator=g)
    lens[0] = T
    lens, _ = torch.sort(lens, descending=True)
    t_["lens"] = lens.to(dev) if lens_on else None

    def build(dg, dcc, dh0=None):
        arr = (L.LstmDirBwd * ((T + 1) * ndir))()
        if dh0 is not None:          # trailing mode-1 step (decoder: gradient of the initial hidden state = dgates of t = 0 times W_hh)
            a = arr[T * ndir]
            a.dgates_next, a.ld_dgn = dg.data_ptr(), ldg
            a.w_hh_t, a.ld_wt = t_["whhT"][0].data_ptr(), 4 * H
            a.dh0_out, a.ld_dh0 = dh0.data_ptr(), H
        for step in range(T):
            for k in range(ndir):
                t = (T - 1 - step) if k == 0 else step
                tn = (t + 1) if k == 0 else (t - 1)
                tp = (t - 1) if k == 0 else (t + 1)
                a = arr[step * ndir + k]
                esz = 2
                if step > 0:
                    a.dgates_next, a.ld_dgn = dg.data_ptr() + (tn * B * ldg + k * 4 * H) * esz, ldg
                a.w_hh_t, a.ld_wt = t_["whhT"][k].data_ptr(), 4 * H
                a.dh_above, a.ld_dha = t_["dha"].data_ptr() + (t * B * H * ndir + k * H) * 2, H * ndir
                a.gates, a.ld_gates = t_["gates"].data_ptr() + (t * B * ldg + k * 4 * H) * 2, ldg
                a.c_t, a.ld_ct = t_["c"].data_ptr() + (t * B * H * ndir + k * H) * 4, H * ndir
                if 0 <= tp < T:
                    a.c_prev, a.ld_cp = t_["c"].data_ptr() + (tp * B * H * ndir + k * H) * 4, H * ndir
                elif not lens_on:
                    a.c_prev, a.ld_cp = t_["cn"].data_ptr() + k * H * 4, H * ndir
                a.dc_carry, a.ld_dcc = dcc.data_ptr() + k * H * 4, H * ndir
                a.dgates_out, a.ld_dgo = dg.data_ptr() + (t * B * ldg + k * 4 * H) * 2, ldg
                a.dh_n, a.ld_dhn = t_["dhn"].data_ptr() + k * H * 4, H * ndir
                a.dc_n, a.ld_dcn = t_["dcn"].data_ptr() + k * H * 4, H * ndir
                a.t = t
                a.inject = (1 if k == 0 else 2) if lens_on else 0
        return arr
    return t_, build, M, ldg


@pytest.mark.parametrize("B,H,ndir,T,lens_on", [(256, 512, 1, 20, False), (256, 256, 2, 20, True), (70, 128, 2, 9, True), (33, 64, 1, 7, False),
                                                (40, 512, 1, 6, False), (128, 1024, 1, 8, False), (100, 1024, 1, 5, True)])
def test_persistent_backward_recurrence_is_bit_identical(B, H, ndir, T, lens_on):
    """vmmt_lstm_seq_bwd against vmmt_lstm_chain_bwd on the same inputs (the training step's own backward inputs carry
    float-atomic noise, so the comparison is made at the kernel level): dgates of every step and the final dL/dc, bit for bit,
    over three different input sets through the same exchange buffers and the first one again"""
    import ctypes as C
    from variational_mmt_amd import _lib as L
    lib = L.lib()
    sync = torch.zeros(lib.vmmt_lstm_seq_sync_words(), dtype=torch.int32, device="cuda")
    xchg = torch.zeros(max(16, lib.vmmt_lstm_seq_xchg_bytes_bwd(ndir, B, H)), dtype=torch.uint8, device="cuda")
    first = None
    for rep, seed in enumerate((11, 12, 13, 11)):
        t_, build, M, ldg = _bwd_case(B, H, ndir, T, seed, lens_on)
        outs = []
        for mode in ("chain", "seq"):
            dg = torch.full((M + 64, ldg), 3.0, dtype=torch.bfloat16, device="cuda")
            dcc = t_["dcc0"].clone()
            with_dh0 = ndir == 1 and not lens_on          # the decoder's shape of the call
            dh0 = torch.full((B, H), -7.0, device="cuda") if with_dh0 else None
            arr = build(dg, dcc, dh0)
            lp = t_["lens"].data_ptr() if t_["lens"] is not None else None
            if mode == "chain":
                L.check(lib.vmmt_lstm_chain_bwd(L.BF16, ndir, T, arr, lp, B, H, 0, None), "chain bwd")
                if with_dh0:
                    last = C.cast(C.byref(arr, T * ndir * C.sizeof(L.LstmDirBwd)), C.POINTER(L.LstmDirBwd))
                    L.check(lib.vmmt_lstm_step_bwd(L.BF16, ndir, last, lp, B, H, 1, None), "dh0 step")
            else:
                dev_arr = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
                L.check(lib.vmmt_lstm_seq_bwd(L.BF16, ndir, T, arr, dev_arr.data_ptr(), lp, B, H, 1 if with_dh0 else 0, sync.data_ptr(),
                                              xchg.data_ptr(), None), "seq bwd")
            torch.cuda.synchronize()
            outs.append((dg.clone(), dcc.clone(), dh0.clone() if with_dh0 else torch.zeros(1)))
        assert int(sync[2].item()) == 0                              # every in-launch wait completed
        if H <= 512:
            assert torch.equal(outs[0][0], outs[1][0]), ("dgates differ", rep, (outs[0][0].float() - outs[1][0].float()).abs().max().item())
            assert torch.equal(outs[0][1], outs[1][1]), ("dc carry differs", rep)
            assert torch.equal(outs[0][2], outs[1][2]), ("dh0 differs", rep)
        else:
            # H = 1024: the step kernels walk the reduction in chunks of 512 / 2048 with four partial sums each, the persistent kernel
            # in four quarters of the whole length: the same numbers up to the order of the f32 additions (bf16 results within an ulp
            # or two, a few steps deep)
            for i, what in enumerate(("dgates", "dc carry", "dh0")):
                a_, b_ = outs[0][i].float(), outs[1][i].float()
                assert (a_ - b_).abs().max().item() <= 2e-2 * max(1.0, b_.abs().max().item()), (what, rep, (a_ - b_).abs().max().item())
                assert (a_ - b_).abs().mean().item() <= 2e-4 * max(1.0, b_.abs().mean().item()) + 1e-5, (what, rep, (a_ - b_).abs().mean().item())
        assert not (outs[1][2] == -7.0).any()
        assert (outs[1][0][M:] == 3.0).all()                         # nothing written beyond the rows of the sequence
        if rep == 0:
            first = outs[1]
        if rep == 3:
            assert torch.equal(first[0], outs[1][0]) and torch.equal(first[1], outs[1][1])
    assert int(sync[0].item()) == 4                                  # four launches, four epochs


@pytest.mark.parametrize("B,H,ndir,T,cuts", [(21, 256, 2, 24, (0, 6, 12, 24)), (256, 512, 1, 20, (0, 9, 20)), (70, 128, 2, 9, (0, 2, 4, 9))])
def test_persistent_backward_recurrence_cut_into_pieces(B, H, ndir, T, cuts):
    """a recurrence issued as several vmmt_lstm_seq_bwd calls (step 0 of a later piece carries dgates_next = the plain dgates
    buffer of the piece before; dc_carry goes through memory) gives the bits of the single call -- the conditional model cuts
    encoder_tgt's 2 x B-step backward this way to run the weight gradients of the finished part next to the rest"""
    import ctypes as C
    from variational_mmt_amd import _lib as L
    lib = L.lib()
    t_, build, M, ldg = _bwd_case(B, H, ndir, T, 21, False)
    outs = []
    for pieces in ((0, T), cuts):
        dg = torch.full((M + 64, ldg), 3.0, dtype=torch.bfloat16, device="cuda")
        dcc = t_["dcc0"].clone()
        arr = build(dg, dcc, None)
        keep = []
        for s0, s1 in zip(pieces[:-1], pieces[1:]):
            n = s1 - s0
            part = (L.LstmDirBwd * (n * ndir)).from_buffer_copy(bytes(arr)[s0 * ndir * C.sizeof(L.LstmDirBwd):s1 * ndir * C.sizeof(L.LstmDirBwd)])
            dev_arr = torch.frombuffer(bytearray(bytes(part)), dtype=torch.uint8).cuda()
            sync = torch.zeros(lib.vmmt_lstm_seq_sync_words(), dtype=torch.int32, device="cuda")
            xchg = torch.zeros(max(16, lib.vmmt_lstm_seq_xchg_bytes_bwd(ndir, B, H)), dtype=torch.uint8, device="cuda")
            L.check(lib.vmmt_lstm_seq_bwd(L.BF16, ndir, n, part, dev_arr.data_ptr(), None, B, H, 0, sync.data_ptr(), xchg.data_ptr(), None), "seq bwd")
            keep.append((part, dev_arr, sync, xchg))
        torch.cuda.synchronize()
        assert all(int(k[2][2].item()) == 0 for k in keep)
        assert all(int(k[2][0].item()) == 1 for k in keep)          # every piece ran as the persistent kernel (one epoch each)
        outs.append((dg.clone(), dcc.clone()))
    assert torch.equal(outs[0][0], outs[1][0]), (outs[0][0].float() - outs[1][0].float()).abs().max().item()
    assert torch.equal(outs[0][1], outs[1][1])
