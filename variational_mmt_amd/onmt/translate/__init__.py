"""Mirror of onmt.translate for the VI_Model1 path: TranslatorMultimodalVI (arg-max and beam search), Beam, GNMTGlobalScorer
(SURVEY.md 8f-2), TranslationBuilder / Translation (ids -> words for the driver's output file).  The text-only Translator is not
mirrored."""
from .Beam import Beam, GNMTGlobalScorer  # noqa: F401
from .TranslatorMultimodalVI import TranslatorMultimodalVI  # noqa: F401
from .Translation import Translation, TranslationBuilder  # noqa: F401
