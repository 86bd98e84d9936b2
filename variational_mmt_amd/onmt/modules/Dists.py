"""Gaussian parameter holder with the interface the loss consumes (reference: onmt/modules/Dists.py:11-60)."""


class Normal(object):
    def __init__(self, loc, scale):
        self.loc, self.scale = loc, scale

    def params(self):
        return [self.loc, self.scale]

    def mean(self):
        return self.loc
