"""Mirror of onmt.translate for the VI_Model1 path: TranslatorMultimodalVI (arg-max and beam search), Beam, GNMTGlobalScorer
(SURVEY.md 8f-2).  The text-only Translator and TranslationBuilder (vocabulary look-up of the predictions) are not mirrored."""
from .Beam import Beam, GNMTGlobalScorer  # noqa: F401
from .TranslatorMultimodalVI import TranslatorMultimodalVI  # noqa: F401
