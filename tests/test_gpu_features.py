"""HDF5 image-feature file -> HBM table -> device row gather, against the host arithmetic of the reference driver
(train_mm_vi_model1.py:460-501: read whole node, optional (x - mean) / std in fp32 numpy; TrainerMultimodal.py:632-639:
fancy-index rows by batch.indices).  Bit-exact: this is data movement plus one IEEE subtract/divide."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H5 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h5")
EXPECTED = np.load(os.path.join(H5, "expected.npz"))


def _p(name):
    return os.path.join(H5, name + ".h5")


@pytest.mark.parametrize("fn,node", [("pt_array", "global_feats"), ("pt_earray", "global_feats"),
                                     ("pt_carray_zlib_shuffle", "global_feats"), ("h5py_default", "local_feats"),
                                     ("h5py_latest", "global_feats"), ("pt_array", "logits")])
@pytest.mark.parametrize("slab_bytes", [64 << 20, 1500])
def test_table_reaches_hbm_intact(fn, node, slab_bytes):
    from variational_mmt_amd.features import load_image_table
    t = load_image_table(_p(fn), node, device="cuda:0", slab_bytes=slab_bytes)      # 1500 B: several slabs, both staging buffers
    e = EXPECTED[fn + "::" + node].astype(np.float32)
    assert t.dtype == torch.float32 and t.is_cuda and tuple(t.shape) == e.shape
    assert np.array_equal(t.cpu().numpy(), e)


def test_standardised_table_and_gather_bit_exact():
    from variational_mmt_amd import _lib as L
    from variational_mmt_amd.features import load_image_table
    t = load_image_table(_p("pt_array"), "global_feats", device="cuda:0", mean_path=_p("pt_mean"), std_path=_p("pt_std"))
    x = EXPECTED["pt_array::global_feats"]
    m, s = EXPECTED["pt_mean::global_feats_mean"], EXPECTED["pt_std::global_feats_stds"]
    e = (x - m[None, :]) / s[None, :]                                                  # the driver's host arithmetic
    assert e.dtype == np.float32
    assert np.array_equal(t.cpu().numpy(), e)
    # rows by batch.indices (repeats allowed: the 290 k set upsamples images)
    idx = torch.tensor([36, 0, 5, 5, 17, 36, 1], dtype=torch.int64, device="cuda:0")
    out = torch.empty((idx.numel(), e.shape[1]), dtype=torch.float32, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    L.check(L.lib().vmmt_gather_rows(L.F32, C.c_void_p(t.data_ptr()), e.shape[1], C.c_void_p(idx.data_ptr()),
                                     C.c_void_p(out.data_ptr()), e.shape[1], idx.numel(), e.shape[1], C.c_void_p(st)), "gather")
    assert np.array_equal(out.cpu().numpy(), e[idx.cpu().numpy()])


def test_standardise_kernel_ragged_shapes():
    from variational_mmt_amd import _lib as L
    g = torch.Generator().manual_seed(3)
    for R, D, ld in [(1, 1, 1), (7, 5, 5), (33, 2048, 2048), (19, 70, 72), (1000, 130, 130)]:
        x = torch.randn(R, ld, generator=g)
        m = torch.randn(D, generator=g)
        s = torch.rand(D, generator=g) + 0.1
        e = x.clone()
        e[:, :D] = torch.from_numpy((x[:, :D].numpy() - m.numpy()[None, :]) / s.numpy()[None, :])
        xd, md, sd = x.cuda(), m.cuda(), s.cuda()
        st = torch.cuda.current_stream().cuda_stream
        L.check(L.lib().vmmt_standardise_rows(C.c_void_p(xd.data_ptr()), ld, C.c_void_p(md.data_ptr()), C.c_void_p(sd.data_ptr()),
                                              R, D, C.c_void_p(st)), "standardise")
        assert torch.equal(xd.cpu(), e), (R, D, ld)                                   # padding columns untouched
    with pytest.raises(RuntimeError):
        L.check(L.lib().vmmt_standardise_rows(None, 4, None, None, 1, 4, None), "standardise")


def test_model_takes_a_feature_file_path():
    """`set_image_tables` given the HDF5 path trains on the same rows as given the numpy array read from it."""
    from oracle import vi1_oracle as O
    from variational_mmt_amd.engine import Dims, Engine
    from variational_mmt_amd.features import load_image_table
    c = O.Cfg(vs=31, vt=37, emb=16, hid=32, z=8, img=64, layers=1, brnn=True)
    p = O.init_params(c, seed=0)
    bt = O.synth_batch(c, B=5, S=6, T=7, n_img=37, seed=2, fixed_len=False)
    table = EXPECTED["pt_array::global_feats"]
    stats = []
    for tab in (torch.from_numpy(table), load_image_table(_p("pt_array"), "global_feats", device="cuda:0")):
        e = Engine(Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, 0.0), dtype="f32", device="cuda:0")
        e.load_state_dict(p)
        e.set_image_table(tab)
        ws = e.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=True, eps=bt["eps"])
        e.loss_backward(ws, normalization=5)
        torch.cuda.synchronize()
        stats.append(e.read_stats(ws))
    # same rows either way; the statistics are accumulated with float atomics, so two runs agree to rounding, not bit for bit
    assert abs(stats[0]["elbo"] - stats[1]["elbo"]) <= 1e-6 * abs(stats[0]["elbo"])
    assert abs(stats[0]["img_feats_loss"] - stats[1]["img_feats_loss"]) <= 1e-6 * abs(stats[0]["img_feats_loss"])
    r, Lo, _g = O.step_grads(p, c, bt["src"], bt["src_len"], bt["tgt"], torch.from_numpy(table)[bt["indices"]], bt["eps"])
    assert abs(stats[1]["elbo"] - float(Lo["elbo"])) / abs(float(Lo["elbo"])) < 2e-5


def test_prefetching_iterator_delivers_the_same_batches():
    """onmt.io.OrderedIterator with its background prefetch thread (host-to-device copies on a stream of their own) against the same
    iterator building every batch in the consumer's thread: same order, same tensors, on the device"""
    import random

    from variational_mmt_amd.onmt import io as oio
    D = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "textdata")
    ds = oio.load_dataset(os.path.join(D, "demo.train.1.pt"))
    fields = oio.load_fields_from_vocab(oio.load_vocab(os.path.join(D, "demo.vocab.pt")))
    ds.fields = dict((k, f) for k, f in fields.items() if k in ds.examples[0].__dict__)
    got = {}
    for pf in (0, 3):
        random.seed(11)
        it = oio.OrderedIterator(dataset=ds, batch_size=5, device="cuda", sort=False, train=True, sort_within_batch=True, repeat=False,
                                 prefetch=pf)
        assert it.prefetch == pf
        got[pf] = [(b.src[0].clone(), b.src[1].clone(), b.tgt[0].clone(), b.tgt[1].clone(), b.indices.clone()) for b in it]
        torch.cuda.synchronize()
    assert len(got[0]) == len(got[3]) == (len(ds) + 4) // 5
    for a, b in zip(got[0], got[3]):
        for x, y in zip(a, b):
            assert x.is_cuda and torch.equal(x, y)
    # an abandoned epoch does not leave the producer thread blocked
    it = oio.OrderedIterator(dataset=ds, batch_size=5, device="cuda", sort=False, train=True, sort_within_batch=True, repeat=False, prefetch=2)
    g = iter(it)
    next(g)
    g.close()
