"""CPU: the background checkpoint writer of the trainer mirror (onmt/TrainerMultimodal.py: _CheckpointWriter, _PickledAs) and the
driver's dataset residency (train_mm_vi_model1._Shards).  Reference behaviour they must not change: a checkpoint is a plain
torch.save pickle whose 'optim' entry is an `onmt.Optim.Optim` (TrainerMultimodal.py:580-587), a split is re-opened every epoch
(train_mm_vi_model1.py:347-385)."""
import os
import pickle
import types

import pytest
import torch

import variational_mmt_amd


def test_writer_writes_in_order_and_reports_errors(tmp_path):
    from variational_mmt_amd.onmt.TrainerMultimodal import _CheckpointWriter
    w = _CheckpointWriter()
    a, b = str(tmp_path / "a.pt"), str(tmp_path / "b.pt")
    for i in range(4):                                   # the same file several times: the last one wins, no partial file is left
        w.put({"epoch": i, "t": torch.full((1000,), float(i))}, a)
    w.put({"epoch": 9}, b)
    w.wait()
    assert torch.load(a, weights_only=False)["epoch"] == 3 and torch.load(b, weights_only=False)["epoch"] == 9
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".partial")]
    w.put({"x": 1}, str(tmp_path / "no_such_dir" / "c.pt"))
    with pytest.raises(Exception):
        w.wait()
    w.put({"epoch": 10}, b)                              # the writer keeps working after an error was reported
    w.wait()
    assert torch.load(b, weights_only=False)["epoch"] == 10


def test_frozen_optimiser_state_pickles_as_the_optim_class():
    onmt = variational_mmt_amd.install_as_onmt()
    from variational_mmt_amd.onmt.TrainerMultimodal import _PickledAs
    opt = onmt.Optim("adam", 0.002, 5.0, lr_decay=0.5, start_decay_at=8)
    opt.lr = 0.0005
    state = opt.__getstate__()
    back = pickle.loads(pickle.dumps(_PickledAs(type(opt), state), protocol=2))
    assert type(back) is type(opt) and back.lr == 0.0005 and back.start_decay_at == 8 and back.method == "adam"
    assert pickle.dumps(_PickledAs(type(opt), state), protocol=2).find(b"_PickledAs") < 0      # nothing of the wrapper reaches the file


def test_single_file_split_stays_resident_between_epochs(tmp_path, monkeypatch):
    from variational_mmt_amd import train_mm_vi_model1 as drv
    from variational_mmt_amd.onmt.io import textdata as td
    onmt = variational_mmt_amd.install_as_onmt()
    ex = td.Example()
    ex.src, ex.tgt, ex.indices = ("a", "b"), ("c",), 0
    path = str(tmp_path / "d.train.1.pt")
    torch.save(td.TextDataset([ex], []), path)
    loads = []
    real = onmt.io.load_dataset
    monkeypatch.setattr(onmt.io, "load_dataset", lambda p: (loads.append(p), real(p))[1])
    opt = types.SimpleNamespace(data=str(tmp_path / "d"), batch_size=1, valid_batch_size=1, batch_type="sents", gpuid=[0])
    drv._Shards._resident.clear()
    s1 = drv._Shards(onmt, opt, "train", {}, 0, 1)
    d1 = s1._load(path)
    d2 = drv._Shards(onmt, opt, "train", {}, 0, 1)._load(path)
    assert d1 is d2 and len(loads) == 1                  # second epoch: the same object, with its id cache
    os.utime(path, (1, 1))                               # the file changed: read again
    d3 = drv._Shards(onmt, opt, "train", {}, 0, 1)._load(path)
    assert d3 is not d1 and len(loads) == 2
    # a split in several shards is streamed shard by shard as before
    torch.save(td.TextDataset([ex], []), str(tmp_path / "d.train.2.pt"))
    s2 = drv._Shards(onmt, opt, "train", {}, 0, 1)
    assert len(s2.files) == 2 and s2._load(path) is not s2._load(path)
