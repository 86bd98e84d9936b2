"""Mirror of onmt.translate for the VI_Model1 hot path: TranslatorMultimodalVI with beam size 1."""
from .TranslatorMultimodalVI import TranslatorMultimodalVI  # noqa: F401
