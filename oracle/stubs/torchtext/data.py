class Field(object):
    def __init__(self, **kw):
        self.kw = kw
        self.vocab_cls = None
        self.vocab = None


class Dataset(object):
    def __init__(self, examples=None, fields=None, filter_pred=None):
        self.examples = examples
        self.fields = fields


class Example(object):
    pass


class Iterator(object):
    def __init__(self, *a, **kw):
        pass


def pool(*a, **kw):
    raise NotImplementedError("torchtext stub")


def batch(*a, **kw):
    raise NotImplementedError("torchtext stub")
