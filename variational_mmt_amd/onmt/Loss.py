"""LossComputeBase surface (reference: onmt/Loss.py:15-132) for the kernel-backed loss."""
import torch.nn as nn

from . import io


class LossComputeBase(nn.Module):
    def __init__(self, generator, tgt_vocab):
        super(LossComputeBase, self).__init__()
        self._generator = [generator]          # not registered: parameters belong to the model
        self.tgt_vocab = tgt_vocab
        self.padding_idx = tgt_vocab.stoi[io.PAD_WORD]
        self.cur_dataset = None

    @property
    def generator(self):
        return self._generator[0]
