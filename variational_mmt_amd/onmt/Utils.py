"""Mirror of onmt/Utils.py (reference) without its import-time asserts on tools/multi-bleu.perl and a hard-coded
METEOR jar (SURVEY.md hazard H10)."""
import torch

MODEL_TYPES = ["vi-model1"]


def aeq(*args):
    first = args[0]
    assert all(a == first for a in args[1:]), "Not all arguments have the same value: " + str(args)


def sequence_mask(lengths, max_len=None):
    """boolean [B, max_len]: position < length (onmt/Utils.py:23-32)"""
    max_len = int(max_len or lengths.max())
    return torch.arange(0, max_len, device=lengths.device).unsqueeze(0) < lengths.unsqueeze(1)


def use_gpu(opt):
    return (hasattr(opt, "gpuid") and len(opt.gpuid) > 0) or (hasattr(opt, "gpu") and opt.gpu > -1)
