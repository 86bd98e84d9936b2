"""`onmt.io.TextDataset` as a module path: the `.train.N.pt` / `.valid.N.pt` files pickle their dataset object as
`onmt.io.TextDataset.TextDataset` (onmt/io/TextDataset.py:16, preprocess.py:97-110), and the driver reads them with a plain
`torch.load` (train_mm_vi_model1.py:372-376) -- so the class must be importable under that path."""
from .textdata import TextDataset  # noqa: F401
