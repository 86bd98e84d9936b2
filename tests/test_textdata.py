"""onmt/io/textdata.py: reading the reference's dataset / vocabulary pickles without torchtext (fixture pickled through the
reference's own classes, oracle/make_textdata_golden.py) and the batching rules of onmt.io.OrderedIterator / torchtext 0.2.3
(restated; torchtext is not available to produce golden batches, so these tests check the rules themselves)."""
import json
import os
import random

import pytest
import torch

from variational_mmt_amd.onmt import io as oio
from variational_mmt_amd.onmt.io import textdata as T

D = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "textdata")
PLAIN = json.load(open(os.path.join(D, "demo.json"), encoding="utf-8"))


def _load(split):
    ds = oio.load_dataset(os.path.join(D, "demo.%s.1.pt" % split))
    fields = oio.load_fields_from_vocab(oio.load_vocab(os.path.join(D, "demo.vocab.pt")))
    ds.fields = dict((k, f) for k, f in fields.items() if k in ds.examples[0].__dict__)      # train_mm_vi_model1.py:388-403
    return ds, fields


@pytest.mark.parametrize("split", ["train", "valid"])
def test_dataset_pickle_contents(split):
    ds, _ = _load(split)
    assert isinstance(ds, oio.TextDataset) and ds.data_type == "text" and len(ds) == len(PLAIN[split])
    for ex, row in zip(ds.examples, PLAIN[split]):
        assert list(ex.src) == row["src"] and list(ex.tgt) == row["tgt"] and ex.indices == row["indices"]
    assert ds.sort_key(ds.examples[0]) == len(PLAIN[split][0]["src"])


def test_vocab_pickle_and_fields():
    vocab = dict(oio.load_vocab(os.path.join(D, "demo.vocab.pt")))
    assert vocab["src"].itos == PLAIN["vocab"]["src"] and vocab["tgt"].itos == PLAIN["vocab"]["tgt"]
    assert vocab["tgt"].itos[:4] == ["<unk>", "<blank>", "<s>", "</s>"]               # DatasetBase.py:7-11, IO.py:221-226
    fields = oio.load_fields_from_vocab(vocab.items())
    assert fields["src"].vocab.stoi["never-seen"] == 0 and len(fields["tgt"].vocab) == len(PLAIN["vocab"]["tgt"])
    assert fields["src"].include_lengths and fields["tgt"].init_token == "<s>" and fields["tgt"].eos_token == "</s>"
    # round trip through save_fields_to_vocab (what drop_checkpoint stores under 'vocab')
    again = dict(oio.save_fields_to_vocab(fields))
    assert again["src"].itos == vocab["src"].itos and isinstance(again["src"].stoi, dict)


def test_foreign_classes_are_refused(tmp_path):
    import pickle
    p = tmp_path / "x.pt"
    torch.save({"a": 1}, str(p))
    with pytest.raises(TypeError):
        oio.load_dataset(str(p))
    with pytest.raises(pickle.UnpicklingError):
        T._PickleModule.loads(b"ctorchtext.data.iterator\nIterator\n.")


def test_field_pad_and_numericalise():
    v = T.Vocab(["<unk>", "<blank>", "<s>", "</s>", "a", "b", "c"])
    f = T.Field(init_token="<s>", eos_token="</s>", pad_token="<blank>", include_lengths=True, vocab=v)
    t, lens = f.process([("a", "b", "c"), ("b",), ("zzz", "a")])
    assert t.shape == (5, 3) and lens.tolist() == [5, 3, 4]
    assert t[:, 0].tolist() == [2, 4, 5, 6, 3] and t[:, 1].tolist() == [2, 5, 3, 1, 1] and t[:, 2].tolist() == [2, 0, 4, 3, 1]
    g = T.Field(pad_token="<blank>", include_lengths=True, vocab=v)
    t, lens = g.process([("a",), ("a", "b")])
    assert t.t().tolist() == [[4, 1], [4, 5]] and lens.tolist() == [1, 2]
    idx = T.Field(use_vocab=False, sequential=False).process([7, 3, 9])
    assert idx.tolist() == [7, 3, 9] and idx.dtype == torch.int64


def test_batch_generator_rules():
    data = list(range(10))
    assert list(T.batch(data, 4)) == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]
    # token budget (train_mm_vi_model1.py:199-201): an example that overshoots the budget opens the next batch
    lens = [3, 3, 3, 5, 2, 9, 1]
    fn = lambda new, count, sofar: sofar + new + 1                                       # noqa: E731
    assert list(T.batch(lens, 12, fn)) == [[3, 3, 3], [5, 2], [9, 1]]


def test_random_shuffler_keeps_global_state_and_is_reproducible():
    random.seed(123)
    before = random.getstate()
    a = T.RandomShuffler()
    b = T.RandomShuffler()                      # both start from the same global state
    x, y = a(range(20)), b(range(20))
    assert x == y and sorted(x) == list(range(20)) and x != list(range(20))
    assert random.getstate() == before          # the module-level generator is untouched
    assert a(range(20)) != x                    # the private state advanced


def test_training_iterator_pools_sorted_batches():
    ds, _ = _load("train")
    random.seed(7)
    it = oio.OrderedIterator(dataset=ds, batch_size=8, device=None, sort=False, train=True, sort_within_batch=True, repeat=False)
    assert len(it) == 8                                                                  # ceil(57 / 8)
    batches = list(it)
    assert sum(b.batch_size for b in batches) == 57 and len(batches) == 8
    seen = sorted(int(i) for b in batches for i in b.indices)
    assert seen == list(range(57))                                                       # every example exactly once
    for b in batches:
        src, lens = b.src
        tgt, tl = b.tgt
        assert lens.tolist() == sorted(lens.tolist(), reverse=True)                      # decreasing source length
        assert src.shape == (int(lens.max()), b.batch_size) and tgt.shape[0] == int(tl.max())
        assert (tgt[0] == 2).all() and all(int(tgt[int(tl[j]) - 1, j]) == 3 for j in range(b.batch_size))
        for j in range(b.batch_size):
            row = PLAIN["train"][int(b.indices[j])]
            assert int(lens[j]) == len(row["src"]) and int(tl[j]) == len(row["tgt"]) + 2
            assert (src[int(lens[j]):, j] == 1).all() and (tgt[int(tl[j]):, j] == 1).all()
    # one pool (57 < 100 * 8): the pool is sorted by length before batching, so batches hold neighbouring lengths
    spans = [int(b.src[1].max() - b.src[1].min()) for b in batches]
    assert max(spans) <= 3
    # a second epoch reshuffles
    assert [b.indices.tolist() for b in it] != [b.indices.tolist() for b in batches]


def test_validation_iterator_keeps_order_of_batches():
    ds, _ = _load("valid")
    it = oio.OrderedIterator(dataset=ds, batch_size=4, device=None, sort=False, train=False, sort_within_batch=True, repeat=False)
    batches = list(it)
    assert [b.batch_size for b in batches] == [4, 4, 3]
    for k, b in enumerate(batches):
        assert sorted(b.indices.tolist()) == list(range(4 * k, min(11, 4 * k + 4)))     # consecutive examples
        assert b.src[1].tolist() == sorted(b.src[1].tolist(), reverse=True)
    assert list(it)[0].indices.tolist() == batches[0].indices.tolist()                  # deterministic


def _replay_contract(ds, batch_size, batch_size_fn, seed):
    """the contract of onmt.io.OrderedIterator.create_batches (onmt/io/IO.py:382-393) for training, replayed independently of
    textdata.pool(): shuffled data -> pools of 100 batches' worth -> each pool sorted by the dataset's sort key and cut into batches
    -> the batches of ONE pool in shuffled order; a pool is exhausted before the next one starts"""
    it = oio.OrderedIterator(dataset=ds, batch_size=batch_size, batch_size_fn=batch_size_fn, device=None, sort=False, train=True,
                             sort_within_batch=True, repeat=False)
    it.random_shuffler = T.RandomShuffler(random.Random(seed).getstate())
    twin = T.RandomShuffler(random.Random(seed).getstate())
    order = [ds[i] for i in twin(range(len(ds)))]                     # Iterator.data() with shuffle=True
    got = [b.indices.tolist() for b in it]
    pools = list(T.batch(order, batch_size * 100, batch_size_fn))
    pos = 0
    for p in pools:
        want = [sorted(ex.indices for ex in b) for b in T.batch(sorted(p, key=ds.sort_key), batch_size, batch_size_fn)]
        mine = got[pos:pos + len(want)]
        pos += len(want)
        assert sorted(sorted(b) for b in mine) == sorted(want)        # exactly this pool's batches, in some (shuffled) order
        for b in mine:                                                # sort_within_batch: decreasing source length
            lens = [len(next(e for e in p if e.indices == i).src) for i in b]
            assert lens == sorted(lens, reverse=True)
    assert pos == len(got)
    return got, pools


def test_training_iterator_replays_the_reference_contract_over_several_pools():
    ds, _ = _load("train")
    big = oio.TextDataset(examples=[], fields=ds.fields)
    for rep in range(6):                                              # 342 examples: four pools at batch size 1
        for ex in ds.examples:
            e2 = T.Example()
            e2.src, e2.tgt, e2.indices = ex.src, ex.tgt, ex.indices + 57 * rep
            big.examples.append(e2)
    got, pools = _replay_contract(big, 1, None, seed=11)
    assert len(pools) == 4 and len(got) == 342
    got3, pools3 = _replay_contract(big, 3, None, seed=12)
    assert len(pools3) == 2 and sum(len(b) for b in got3) == 342
    # token budget (-batch_type tokens, train_mm_vi_model1.py:197-201)
    fn = lambda new, count, sofar: sofar + max(len(new.tgt), len(new.src)) + 1      # noqa: E731
    got_t, _ = _replay_contract(big, 40, fn, seed=13)
    assert sum(len(b) for b in got_t) == 342
    for b in got_t:
        exs = [big.examples[i] for i in b]
        assert sum(max(len(e.tgt), len(e.src)) + 1 for e in exs) <= 40 or len(exs) == 1


def test_evaluation_iterator_replays_the_reference_contract():
    """IO.py:389-393: not training -> consecutive batches of the data IN ORDER, each sorted by the sort key (then reversed by
    sort_within_batch to decreasing length: torchtext Iterator.__iter__)"""
    ds, _ = _load("valid")
    it = oio.OrderedIterator(dataset=ds, batch_size=3, device=None, sort=False, train=False, sort_within_batch=True, repeat=False)
    got = [b.indices.tolist() for b in it]
    want = []
    for b in T.batch(ds.examples, 3):
        s = sorted(b, key=ds.sort_key)
        s.sort(key=ds.sort_key, reverse=True)
        want.append([e.indices for e in s])
    assert got == want


def test_fast_batch_equals_field_by_field():
    """Batch builds its tensors from cached word ids with vectorised padding (one staging buffer, one copy); the result must be what
    Field.pad + Field.numericalize give field by field, for every batch of a training and of a validation epoch"""
    from variational_mmt_amd.onmt.io import textdata as td
    for split, train in (("train", True), ("valid", False)):
        ds, _ = _load(split)
        random.seed(5)
        it = oio.OrderedIterator(dataset=ds, batch_size=7, device=None, sort=False, train=train, sort_within_batch=True, repeat=False)
        it.create_batches()
        n = 0
        for mb in it.batches:
            mb = sorted(mb, key=ds.sort_key, reverse=True)
            fast = td.Batch(mb, ds, None, train)
            slow = td.Batch()
            slow.dataset, slow.train = ds, train
            slow.slow(mb, None)
            for name in ("src", "tgt"):
                assert torch.equal(getattr(fast, name)[0], getattr(slow, name)[0]) and torch.equal(getattr(fast, name)[1], getattr(slow, name)[1])
                assert getattr(fast, name)[0].is_contiguous() and getattr(fast, name)[0].dtype == torch.int64
            assert torch.equal(fast.indices, slow.indices)
            n += 1
        assert n == (len(ds) + 6) // 7
