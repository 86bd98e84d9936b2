from collections import defaultdict, Counter


class Vocab(object):
    """Just enough of torchtext.vocab.Vocab: stoi / itos / __len__ / freqs."""

    def __init__(self, counter=None, specials=("<unk>", "<blank>"), max_size=None,
                 min_freq=1, **kw):
        counter = counter if counter is not None else Counter()
        self.freqs = counter
        self.itos = list(specials)
        for w, _ in sorted(counter.items(), key=lambda kv: (-kv[1], kv[0])):
            if w not in self.itos:
                self.itos.append(w)
        self.stoi = defaultdict(int)
        self.stoi.update({w: i for i, w in enumerate(self.itos)})

    def __len__(self):
        return len(self.itos)
