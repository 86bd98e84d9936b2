"""`onmt.io.DatasetBase` as a module path (onmt/io/DatasetBase.py:7-11,14-46): the special tokens and the dataset base class the text
dataset pickles may name."""
from .textdata import BOS_WORD, EOS_WORD, PAD_WORD, TextDataset  # noqa: F401

UNK = 0
ONMTDatasetBase = TextDataset
