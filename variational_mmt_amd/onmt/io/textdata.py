"""The batching side of the training driver without torchtext: what `onmt.io.OrderedIterator` (onmt/io/IO.py:382-393) and
the torchtext 0.2.3 classes underneath it (requirements.txt: torchtext==0.2.3 -- NOT vendored in the reference, absent from
this image) do for a text dataset, restated from that release's published source:

  Example      attribute bag (src / tgt token tuples, indices)                      torchtext/data/example.py
  Vocab        itos / stoi (defaultdict -> <unk> = 0) / freqs                       torchtext/vocab.py + onmt/io/IO.py:17-27
  Field        pad (+ <s> / </s>), numericalise, lengths                            torchtext/data/field.py  (pad, numericalize)
  batch, pool  fixed-size or token-budget batches; pools of 100 batches sorted by
               length, batched, batches shuffled                                    torchtext/data/iterator.py
  RandomShuffler   a private `random` state, `random.sample(data, len(data))`       torchtext/data/utils.py
  Iterator / OrderedIterator / Batch                                                torchtext/data/iterator.py, batch.py

Parity is UNPINNED for the torchtext part (no torchtext here to produce golden batches); tests check the restated rules on
their own terms (tests/test_textdata.py).  The pickle layout of `.train.N.pt` / `.vocab.pt` files IS pinned: a fixture is
written by the reference's own `onmt.io.TextDataset` / `save_fields_to_vocab` (oracle/make_textdata_golden.py).
`load_dataset` / `load_vocab` read those files by mapping the pickled class names onto the classes below, so neither
torchtext nor the reference has to be importable."""
import math
import pickle
import random
from collections import Counter, defaultdict  # noqa: F401
from contextlib import contextmanager

import torch

PAD_WORD, UNK_WORD, BOS_WORD, EOS_WORD = "<blank>", "<unk>", "<s>", "</s>"


class Example(object):
    pass


def _zero():
    return 0


class Vocab(object):
    """`stoi` answers 0 (<unk>) for unknown words (onmt/io/IO.py:21-23)."""

    def __init__(self, itos=None, freqs=None):
        self.itos = list(itos or [])
        self.freqs = freqs if freqs is not None else Counter()
        self.stoi = defaultdict(_zero, {w: i for i, w in enumerate(self.itos)})
        self.vectors = None

    def __getstate__(self):
        return dict(self.__dict__, stoi=dict(self.stoi))

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.stoi = defaultdict(_zero, self.stoi)

    def __len__(self):
        return len(self.itos)


class Field(object):
    """The slice of torchtext.data.Field this path uses (sequential text with a vocabulary, or a plain integer per example)."""

    def __init__(self, sequential=True, use_vocab=True, init_token=None, eos_token=None, pad_token="<pad>", include_lengths=False,
                 vocab=None):
        self.sequential, self.use_vocab = sequential, use_vocab
        self.init_token, self.eos_token, self.pad_token = init_token, eos_token, pad_token
        self.include_lengths = include_lengths
        if vocab is not None:                     # like torchtext: the attribute exists only once a vocabulary is attached
            self.vocab = vocab

    def pad(self, minibatch):
        minibatch = list(minibatch)
        if not self.sequential:
            return minibatch
        max_len = max(len(x) for x in minibatch)
        padded, lengths = [], []
        for x in minibatch:
            row = ([] if self.init_token is None else [self.init_token]) + list(x[:max_len]) + \
                  ([] if self.eos_token is None else [self.eos_token]) + [self.pad_token] * max(0, max_len - len(x))
            padded.append(row)
            lengths.append(len(row) - max(0, max_len - len(x)))
        return (padded, lengths) if self.include_lengths else padded

    def numericalize(self, arr, device=None):
        lengths = None
        if self.include_lengths:
            arr, lengths = arr
            lengths = torch.tensor(lengths, dtype=torch.int64, device="cpu")
        if self.use_vocab:
            arr = [[self.vocab.stoi[w] for w in ex] for ex in arr] if self.sequential else [self.vocab.stoi[x] for x in arr]
        t = torch.tensor(arr, dtype=torch.int64, device="cpu")
        if self.sequential:
            t = t.t().contiguous()                      # [len, batch]
        if device is not None and device != -1:
            t = t.to(device)
            lengths = lengths.to(device) if lengths is not None else None
        return (t, lengths) if self.include_lengths else t

    def process(self, batch, device=None, train=True):
        return self.numericalize(self.pad(batch), device=device)


def get_fields(n_src_features=0, n_tgt_features=0):
    """TextDataset.get_fields (onmt/io/TextDataset.py:180-240) without the copy-attention fields (src_map, alignment)."""
    if n_src_features or n_tgt_features:
        raise NotImplementedError("word features are outside the VI_Model1 path")
    return {"src": Field(pad_token=PAD_WORD, include_lengths=True),
            "tgt": Field(init_token=BOS_WORD, eos_token=EOS_WORD, pad_token=PAD_WORD, include_lengths=True),
            "indices": Field(use_vocab=False, sequential=False)}


class TextDataset(object):
    """What a `.train.N.pt` / `.valid.N.pt` file holds (onmt/io/TextDataset.py:16-86, preprocess.py:97-110: `fields` emptied
    before saving)."""
    data_type = "text"

    def __init__(self, examples=None, fields=None):
        self.examples = list(examples or [])
        self.fields = fields if fields is not None else []
        self.data_type = "text"
        self.src_vocabs = []
        self.n_src_feats = self.n_tgt_feats = 0

    def __setstate__(self, d):
        self.__dict__.update(d)

    @staticmethod
    def sort_key(ex):
        return len(ex.src)

    def __len__(self):
        return len(self.examples)

    def __getitem__(self, i):
        return self.examples[i]

    def __iter__(self):
        return iter(self.examples)

    def load_fields(self, vocab_dict):
        """ONMTDatasetBase.load_fields (onmt/io/DatasetBase.py:35-46)"""
        from . import load_fields_from_vocab
        fields = load_fields_from_vocab(dict(vocab_dict).items(), self.data_type)
        self.fields = dict((k, f) for k, f in fields.items() if k in self.examples[0].__dict__)


# ---- reading the reference's pickles -----------------------------------------------------------------------------------
_CLASS_MAP = {
    ("onmt.io.TextDataset", "TextDataset"): TextDataset,
    ("onmt.io.DatasetBase", "ONMTDatasetBase"): TextDataset,
    ("torchtext.data.example", "Example"): Example,
    ("torchtext.data", "Example"): Example,
    ("torchtext.vocab", "Vocab"): Vocab,
    ("torchtext.data.dataset", "Dataset"): TextDataset,
    ("torchtext.data", "Dataset"): TextDataset,
}


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        hit = _CLASS_MAP.get((module, name))
        if hit is not None:
            return hit
        if module.startswith("torchtext") or module.startswith("onmt."):
            raise pickle.UnpicklingError("%s.%s is not part of the text-dataset pickles this loader reads" % (module, name))
        return super(_Unpickler, self).find_class(module, name)


class _PickleModule(object):
    """the `pickle_module` argument of torch.load (both the legacy and the zip container call .Unpickler / .load)"""
    __name__ = "pickle"
    Unpickler = _Unpickler
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL

    @staticmethod
    def load(f, **kw):
        return _Unpickler(f, **kw).load()

    @staticmethod
    def loads(b, **kw):
        import io
        return _Unpickler(io.BytesIO(b), **kw).load()


def load_dataset(path):
    """`torch.load(pt_file)` of lazily_load_dataset (train_mm_vi_model1.py:372-376) -> TextDataset"""
    ds = torch.load(path, map_location="cpu", pickle_module=_PickleModule, weights_only=False)
    if not isinstance(ds, TextDataset):
        raise TypeError("%s holds a %s, not a text dataset" % (path, type(ds).__name__))
    return ds


def load_vocab(path):
    """`torch.load(opt.data + '.vocab.pt')` (train_mm_vi_model1.py:393) -> list of (name, Vocab)"""
    return torch.load(path, map_location="cpu", pickle_module=_PickleModule, weights_only=False)


# ---- batching (torchtext/data/iterator.py, utils.py) -------------------------------------------------------------------
def batch(data, batch_size, batch_size_fn=None):
    """consecutive examples until `batch_size_fn` reaches the budget; an example that overshoots opens the next batch"""
    if batch_size_fn is None:
        batch_size_fn = lambda new, count, sofar: count      # noqa: E731
    minibatch, size_so_far = [], 0
    for ex in data:
        minibatch.append(ex)
        size_so_far = batch_size_fn(ex, len(minibatch), size_so_far)
        if size_so_far == batch_size:
            yield minibatch
            minibatch, size_so_far = [], 0
        elif size_so_far > batch_size:
            yield minibatch[:-1]
            minibatch, size_so_far = minibatch[-1:], batch_size_fn(ex, 1, 0)
    if minibatch:
        yield minibatch


class RandomShuffler(object):
    """shuffles with a private copy of the `random` module's state (the global state is put back afterwards)"""

    def __init__(self, random_state=None):
        self._random_state = random_state if random_state is not None else random.getstate()

    @contextmanager
    def use_internal_state(self):
        old = random.getstate()
        random.setstate(self._random_state)
        yield
        self._random_state = random.getstate()
        random.setstate(old)

    @property
    def random_state(self):
        return self._random_state

    @random_state.setter
    def random_state(self, s):
        self._random_state = s

    def __call__(self, data):
        with self.use_internal_state():
            data = list(data)
            return random.sample(data, len(data))


def pool(data, batch_size, key, batch_size_fn=None, random_shuffler=None):
    """pools of 100 batches: sort each pool by `key`, cut it into batches, shuffle the batches of the pool"""
    if random_shuffler is None:
        random_shuffler = lambda x: random.sample(list(x), len(list(x)))      # noqa: E731
    if batch_size_fn is None:
        # fixed-size batches: the same pools and batches as the general path below, cut by slicing
        data = list(data)
        for i in range(0, len(data), batch_size * 100):
            p = sorted(data[i:i + batch_size * 100], key=key)
            for b in random_shuffler([p[j:j + batch_size] for j in range(0, len(p), batch_size)]):
                yield b
        return
    for p in batch(data, batch_size * 100, batch_size_fn):
        for b in random_shuffler(list(batch(sorted(p, key=key), batch_size, batch_size_fn))):
            yield b


class _Staging(object):
    """pinned host buffers for the ONE host-to-device copy a batch costs, used round-robin; a buffer is reused only when the copy that
    last read it has finished (an event per buffer)"""

    def __init__(self, n=4):
        self.bufs, self.events, self.k = [None] * n, [None] * n, 0
        self.waited = 0.0                                  # seconds spent waiting for the device to catch up (bench.py --through-trainer)

    def get(self, numel):
        k = self.k = (self.k + 1) % len(self.bufs)
        if self.events[k] is not None:
            import time
            t = time.perf_counter()
            self.events[k].synchronize()                  # the host is len(bufs) batches ahead of the device: it waits here
            self.waited += time.perf_counter() - t
        b = self.bufs[k]
        if b is None or b.numel() < numel:
            b = self.bufs[k] = torch.empty(max(numel, 16384), dtype=torch.int64, device="cpu").pin_memory()
        return k, b


_staging = {}


class Batch(object):
    """src = (ids [S,B], lengths [B]), tgt = (ids [T,B] with <s> / </s>, lengths [B]), indices [B].

    Same tensors as `field.process(...)` field by field (Field.pad + Field.numericalize above -- what torchtext does and what
    `Batch.slow()` still does; tests/test_textdata.py compares the two), built the cheap way: every example's word ids are looked
    up ONCE (cached on the example, per vocabulary), a batch is padded with vectorised numpy index arithmetic, and all of its
    tensors travel to the device in one pinned, non-blocking copy.  Field by field the host needed 2.6 ms per 256-sentence batch
    (8 k dictionary look-ups, 1 k list appends, five tensor constructions and five H2D copies: tools/trainer_host_profile.py) --
    more than the GPU needs for the training step itself."""

    def __init__(self, data=None, dataset=None, device=None, train=True, copy_stream=None):
        """copy_stream: the HIP stream the host-to-device copy is issued on (an iterator's prefetch thread); the batch then carries
        `_ready`, the event a consumer's stream has to wait for (OrderedIterator does, before it hands the batch out)"""
        if data is None:
            return
        self.batch_size = len(data)
        self.dataset = dataset
        self.train = train
        fields = [(n, f) for n, f in dataset.fields.items() if f is not None]
        if not all(isinstance(f, Field) and (f.use_vocab == f.sequential) for _n, f in fields):
            return self.slow(data, device)
        import numpy as np
        B = len(data)
        parts, total = [], 0          # (name, array [L, B] or [B], lengths or None)
        for name, f in fields:
            vals = [x.__dict__[name] for x in data]
            if not f.sequential:
                arr = np.asarray(vals, dtype=np.int64)
                parts.append((name, arr, None))
                total += arr.size
                continue
            # the dataset's word ids as one flat array + offsets, looked up once per (field, vocabulary)
            cache = dataset.__dict__.setdefault("_id_cache", {})
            ent = cache.get(name)
            if ent is None or ent[0] != id(f.vocab) or ent[3] != len(dataset.examples):
                stoi = f.vocab.stoi
                lens_all = np.fromiter((len(x.__dict__[name]) for x in dataset.examples), dtype=np.int64, count=len(dataset.examples))
                flat_all = np.fromiter((stoi[w] for x in dataset.examples for w in x.__dict__[name]), dtype=np.int64, count=int(lens_all.sum()))
                ent = cache[name] = (id(f.vocab), flat_all, np.concatenate(([0], np.cumsum(lens_all))), len(dataset.examples))
                for i, x in enumerate(dataset.examples):
                    x.__dict__["_row"] = (id(dataset), i)
            _vid, flat_all, offs, _n = ent
            did = id(dataset)
            try:
                rows = np.fromiter((x._row[1] if x._row[0] == did else -1 for x in data), dtype=np.int64, count=B)
            except AttributeError:
                rows = np.full(B, -1, dtype=np.int64)
            if B and rows.min() < 0:        # examples that are not this dataset's: field by field
                return self.slow(data, device)
            raw = offs[rows + 1] - offs[rows]
            head = 0 if f.init_token is None else 1
            tail = 0 if f.eos_token is None else 1
            L = int(raw.max()) + head + tail
            stoi = f.vocab.stoi
            arr = np.full((L, B), stoi[f.pad_token], dtype=np.int64)
            col = np.repeat(np.arange(B), raw)
            within = np.arange(int(raw.sum())) - np.repeat(np.cumsum(raw) - raw, raw)
            flat = flat_all[np.repeat(offs[rows], raw) + within]
            row = within + head
            arr[row, col] = flat
            if head:
                arr[0, :] = stoi[f.init_token]
            if tail:
                arr[raw + head, np.arange(B)] = stoi[f.eos_token]
            lens = raw + head + tail
            if name == "tgt":       # decoder rows that carry a target (every position but <s>): the engine's generator runs over these only
                self.n_tgt_tokens = int(lens.sum()) - B
            parts.append((name, arr, lens if f.include_lengths else None))
            total += arr.size + (B if f.include_lengths else 0)
        on_gpu = device is not None and device != -1 and torch.device(device if not isinstance(device, int) else "cuda:%d" % device).type == "cuda"
        if on_gpu:
            dev = torch.device(device if not isinstance(device, int) else "cuda:%d" % device)
            st = _staging.setdefault(str(dev), _Staging())
            k, host = st.get(total)
        else:
            host = torch.empty(total, dtype=torch.int64, device="cpu")
        hv = host.numpy()
        o, views = 0, []
        for name, arr, lens in parts:
            n = arr.size
            hv[o:o + n] = arr.reshape(-1)
            views.append((name, o, arr.shape, None if lens is None else o + n))
            o += n
            if lens is not None:
                hv[o:o + B] = lens
                o += B
        if on_gpu:
            cs = copy_stream if copy_stream is not None else torch.cuda.current_stream(dev)
            with torch.cuda.stream(cs):
                flat = host[:total].to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(cs)
            st.events[k] = ev
            if copy_stream is not None:
                self._ready, self._flat = ev, flat
        else:
            flat = host
        for name, o, shape, lo in views:
            t = flat[o:o + int(np.prod(shape))].view(*shape)
            setattr(self, name, t if lo is None else (t, flat[lo:lo + B]))

    def slow(self, data, device):
        """field by field, the way torchtext builds a batch (Field.process = pad + numericalize)"""
        for name, field in self.dataset.fields.items():
            if field is not None:
                setattr(self, name, field.process([x.__dict__[name] for x in data], device=device, train=self.train))
        return self


class OrderedIterator(object):
    """onmt.io.OrderedIterator (onmt/io/IO.py:382-393, the part the reference itself holds: `create_batches` = pool() when training,
    else consecutive batch()es each `sorted(b, key=self.sort_key)`) over torchtext.data.Iterator (its constructor defaults, data(),
    init_epoch, __len__, __iter__ with sort_within_batch: restated from torchtext 0.2.3, torchtext/data/iterator.py -- PARITY
    UNPINNED, see the module docstring).  The reference constructs it as OrderedIterator(dataset, batch_size, batch_size_fn,
    device, sort=False, train=is_train, sort_within_batch=True, repeat=False) (train_mm_vi_model1.py:181-187).
    tests/test_textdata.py replays that contract independently of pool() on the committed dataset pickles."""

    def __init__(self, dataset, batch_size, sort_key=None, device=None, batch_size_fn=None, train=True, repeat=None, shuffle=None,
                 sort=None, sort_within_batch=None, dp_rank=0, dp_world=1, dp_seed=1234, prefetch=None):
        """`dp_world` > 1 (an extension; the reference is single-GPU): every rank walks the SAME sequence of global minibatches
        of `batch_size * dp_world` examples (a shuffler seeded with `dp_seed` on every rank) and numericalises only its share
        -- the examples `dp_rank, dp_rank + dp_world, ...` of the length-sorted minibatch, so the ranks' shares are disjoint,
        sorted, equally long (+-1) and of near-equal token count.  A batch carries `global_batch_size` and `global_ntokens`
        (what the trainer needs to normalise like the single-process run on the whole minibatch) -- no collective, no host
        sync.  A training minibatch with fewer examples than ranks (the tail of an epoch: < dp_world examples) is skipped so
        that every rank sees the same number of batches."""
        self.dp_rank, self.dp_world = int(dp_rank), int(dp_world)
        assert 0 <= self.dp_rank < self.dp_world
        self.batch_size, self.train, self.dataset = batch_size * self.dp_world, train, dataset
        self.batch_size_fn = batch_size_fn
        self.iterations = 0
        self.repeat = train if repeat is None else repeat
        self.shuffle = train if shuffle is None else shuffle
        self.sort = (not train) if sort is None else sort
        self.sort_within_batch = self.sort if sort_within_batch is None else sort_within_batch
        self.sort_key = dataset.sort_key if sort_key is None else sort_key
        self.device = device
        self.random_shuffler = RandomShuffler(random.Random(dp_seed).getstate()) if self.dp_world > 1 else RandomShuffler()
        self._iterations_this_epoch = 0
        # `prefetch` batches are prepared ahead by a background thread (an extension: padding, the pinned staging copy and the
        # host-to-device copy on a stream of its own overlap the main thread's kernel launches; the epoch's sort / pool work no
        # longer sits in front of the first step).  Default 0 = build in __iter__: with the batches assembled from cached ids the
        # main thread needs ~0.2 ms per batch, and a second Python thread costs it more in interpreter-lock hand-offs than it
        # saves (bench.py --through-trainer: 2.12 against 2.05 ms per step).  For loaders with heavier per-batch host work.
        # The ORDER and CONTENT of the batches do not depend on it (tests/test_gpu_features.py).
        self.prefetch = 0 if prefetch is None else int(prefetch)

    def data(self):
        if self.sort:
            return sorted(self.dataset, key=self.sort_key)
        if self.shuffle:
            return [self.dataset[i] for i in self.random_shuffler(range(len(self.dataset)))]
        return self.dataset

    def create_batches(self):
        if self.train:
            self.batches = pool(self.data(), self.batch_size, self.sort_key, self.batch_size_fn,
                                random_shuffler=self.random_shuffler)
        else:
            self.batches = [sorted(b, key=self.sort_key) for b in batch(self.data(), self.batch_size, self.batch_size_fn)]

    def init_epoch(self):
        self.create_batches()
        self._iterations_this_epoch = 0
        if not self.repeat:
            self.iterations = 0

    def __len__(self):
        return int(math.ceil(len(self.dataset) / self.batch_size))

    def _epoch_batches(self, copy_stream=None):
        """one epoch's Batch objects, in order"""
        self.init_epoch()
        for minibatch in self.batches:
            self.iterations += 1
            self._iterations_this_epoch += 1
            if self.sort_within_batch:
                if self.sort:
                    minibatch.reverse()
                else:
                    minibatch.sort(key=self.sort_key, reverse=True)
            if self.dp_world == 1:
                yield Batch(minibatch, self.dataset, self.device, self.train, copy_stream)
                continue
            if len(minibatch) < self.dp_world:
                continue
            b = Batch(minibatch[self.dp_rank::self.dp_world], self.dataset, self.device, self.train, copy_stream)
            b.global_batch_size = len(minibatch)
            # non-pad positions of tgt[1:]: the sentence's tokens + </s>  (normalization == "tokens", TrainerMultimodal.py:336-341)
            b.global_ntokens = sum(len(getattr(ex, "tgt", ())) + 1 for ex in minibatch)
            yield b

    def _prefetched(self):
        import queue
        import threading
        dev = torch.device(self.device if not isinstance(self.device, int) else "cuda:%d" % self.device)
        q, stop = queue.Queue(maxsize=self.prefetch), threading.Event()
        cur_dev = dev.index if dev.index is not None else torch.cuda.current_device()

        def produce():
            try:
                torch.cuda.set_device(cur_dev)
                cs = torch.cuda.Stream(device=dev)
                for b in self._epoch_batches(cs):
                    while not stop.is_set():
                        try:
                            q.put(b, timeout=0.1)
                            break
                        except queue.Full:
                            pass
                    if stop.is_set():
                        return
                q.put(None)
            except BaseException as e:        # hand the failure to the consumer instead of dying silently
                q.put(e)
        th = threading.Thread(target=produce, name="vmmt-batch-prefetch", daemon=True)
        th.start()
        try:
            while True:
                b = q.get()
                if b is None:
                    return
                if isinstance(b, BaseException):
                    raise b
                main = torch.cuda.current_stream(dev)
                main.wait_event(b._ready)               # the ids are on the device before this stream's next kernel reads them
                b._flat.record_stream(main)
                del b._ready, b._flat
                yield b
        finally:
            stop.set()

    def __iter__(self):
        while True:
            if self.prefetch > 0:
                for b in self._prefetched():
                    yield b
            else:
                for b in self._epoch_batches():
                    yield b
            if not self.repeat:
                return
