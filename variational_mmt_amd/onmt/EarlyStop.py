"""Mirror of onmt/EarlyStop.py: early stopping / model selection on validation BLEU every N model updates
(SURVEY.md 8f-4).  Same constructor, `add_run`, `_do_early_stop`, `compute_bleus`, `results_bleu` / `results_meteor` /
`signal_early_stopping` as the reference.  Differences in mechanism, not in policy:

  * `translate_` does not spawn `python translate_mm_vi.py -batch_size 1 -beam_size 1` on a checkpoint written to disk
    (EarlyStop.py:246-271): the validation source file is translated IN PROCESS by the live model with arg-max decoding
    (beam size 1, as the reference uses here) on the GPU, many sentences per batch; the snapshot file name is accepted and
    ignored when a model is attached (attach_model).
  * BLEU is computed by onmt/bleu.py (a restatement of tools/multi-bleu.perl + the sed de-BPE step), not by a perl pipe.
  * METEOR needs java + meteor-1.5.jar (EarlyStop.py:172): when neither is present the score is recorded as nan and
    criterion 'meteor' raises at construction."""
import math
import os
import shutil
import tempfile
from glob import glob

import numpy

from . import bleu as _bleu
from .Utils import MODEL_TYPES


class EarlyStop(object):
    def __init__(self, src, tgt, early_stop_criteria, start_early_stop_at, evaluate_every_nupdates, patience,
                 multimodal_model_type=None, img_fname=None, gpuid=0):
        criteria = ["perplexity", "bleu", "meteor"]
        assert early_stop_criteria in criteria, \
            "ERROR: Invalid parameter value: '%s'. Accepted values: %s." % (early_stop_criteria, str(criteria))
        assert multimodal_model_type in [None] + MODEL_TYPES, \
            "ERROR: Invalid parameter value: '%s'. Accepted values: %s." % (multimodal_model_type, str([None] + MODEL_TYPES))
        if multimodal_model_type is not None:
            assert img_fname is not None, "Must provide image features file name for multimodal_model_type: %s" % multimodal_model_type
        self.src, self.tgt, self.img_fname = src, tgt, img_fname
        self.early_stop_criteria = early_stop_criteria
        self.start_early_stop_at = start_early_stop_at
        self.evaluate_every_nupdates = evaluate_every_nupdates
        self.patience = patience
        self.multimodal_model_type = multimodal_model_type
        self.signal_early_stopping = False
        self.results_bleu, self.results_meteor = {}, {}
        self.batch_size, self.beam_size = 1, 1
        try:
            self.gpuid = gpuid[0]
        except Exception:
            self.gpuid = gpuid
        self.meteor_jar = os.environ.get("METEOR_JAR")
        self.have_meteor = bool(self.meteor_jar and os.path.isfile(self.meteor_jar) and shutil.which("java"))
        if early_stop_criteria == "meteor" and not self.have_meteor:
            raise RuntimeError("early stopping on METEOR needs java and meteor-1.5.jar (set METEOR_JAR)")
        self._model = self._fields = None
        self.decode_batch_size, self.max_length = 64, 100

    def attach_model(self, model, fields, decode_batch_size=64, max_length=100):
        """the live model translates the validation set (TrainerMultimodal does this at construction)"""
        self._model, self._fields = model, fields
        self.decode_batch_size, self.max_length = decode_batch_size, max_length

    def _do_early_stop(self):
        """EarlyStop.py:60-88: stop when the best score so far lies more than `patience` evaluations back."""
        results = self.results_bleu if self.early_stop_criteria == "bleu" else self.results_meteor
        if len(results) + 1 < self.patience:
            return False
        vals = [v for _k, v in sorted(results.items(), key=lambda kv: kv[0])]
        last, before = float(vals[-1]), vals[:-1]
        if before and max(before) >= last:
            max_position = numpy.array(before).argmax()
            if len(before) + 1 - max_position > self.patience:
                self.signal_early_stopping = True
                return True
        return False

    def add_run(self, curr_model_snapshot, n_model_updates):
        """EarlyStop.py:91-146: translate the validation source with the current model, score, update the stopping signal;
        returns whether the current score is the best so far (ties count as best)."""
        if self.early_stop_criteria == "perplexity" or self.early_stop_criteria is None:
            return False
        tf = tempfile.NamedTemporaryFile(delete=False)
        tf.close()
        try:
            self.translate_(self.src, curr_model_snapshot, tf.name)
            _names, bleus, _files = self.compute_bleus(tf.name, self.tgt, "valid")
            self.results_bleu[n_model_updates] = float(bleus[0])
            _names, meteors, _files = self.compute_meteors(tf.name, self.tgt, "valid")
            self.results_meteor[n_model_updates] = float(meteors[0])
        finally:
            os.unlink(tf.name)
        self._do_early_stop()
        results = self.results_bleu if self.early_stop_criteria == "bleu" else self.results_meteor
        vals = [v for _k, v in sorted(results.items(), key=lambda kv: kv[0])]
        last, before = float(vals[-1]), vals[:-1]
        return True if not before else not (max(before) > last)

    def compute_bleus(self, hypotheses_fname, references_fname, split="valid"):
        """EarlyStop.py:205-243: one score STRING per hypothesis file matching the glob."""
        assert split in ["valid", "test2016"], "Must compute BLEU for either valid or test set test2016!"
        names, scores, files = [], [], []
        for hypfile in glob(hypotheses_fname):
            scores.append(_bleu.score_files(hypfile, references_fname, bpe=True))
            names.append(hypfile.replace(".pt.translations-%s" % split, ".pt"))
            files.append(hypfile)
        return names, scores, files

    def compute_meteors(self, hypotheses_fname, references_fname, split="valid"):
        assert split == "valid"
        files = glob(hypotheses_fname)
        if not self.have_meteor:
            return [f.replace(".pt.translations-%s" % split, ".pt") for f in files], [float("nan")] * len(files), files
        import subprocess
        names, scores = [], []
        for hypfile in files:
            with tempfile.NamedTemporaryFile("w", encoding="utf-8", suffix=".hyp") as th, \
                    tempfile.NamedTemporaryFile("w", encoding="utf-8", suffix=".ref") as tr:
                for path, out in ((hypfile, th), (references_fname, tr)):
                    for line in open(path, encoding="utf-8"):
                        out.write(_bleu.debpe(line.rstrip("\n")) + "\n")
                    out.flush()
                res = subprocess.run(["java", "-Xmx2G", "-jar", self.meteor_jar, th.name, tr.name, "-l", "de", "-norm"],
                                     capture_output=True, text=True).stdout
            final = [ln for ln in res.splitlines() if ln.startswith("Final score:")]
            scores.append(float(final[-1].split()[2]) * 100 if final else float("nan"))
            names.append(hypfile.replace(".pt.translations-%s" % split, ".pt"))
        return names, scores, files

    def translate_(self, source_fname, model_fname, hypfname_out):
        if self._model is None:
            raise RuntimeError("EarlyStop.translate_: no live model attached (attach_model); this build does not spawn "
                               "translate_mm_vi.py on a checkpoint file")
        from .translate.translate_file import translate_file
        translate_file(self._model, self._fields, source_fname, hypfname_out, batch_size=self.decode_batch_size,
                       beam_size=self.beam_size, max_length=self.max_length)


def is_nan(x):
    return isinstance(x, float) and math.isnan(x)
