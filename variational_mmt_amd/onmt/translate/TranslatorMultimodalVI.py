"""Host-side mirror of onmt/translate/TranslatorMultimodalVI.py: `translate_batch` returns the reference's result dictionary
(predictions / scores / attention / gold_score / batch, TranslatorMultimodalVI.py:218-243).

beam_size == 1 and no global scorer: arg-max decoding (variational_mmt_amd.decode.greedy_decode); a hypothesis ends at the first
</s> and its score is the sum of its tokens' log-probabilities.  Otherwise: beam search -- the device runs every position of
Beam.advance for all sentences of the batch (decode.beam_decode), the host `Beam` mirror replays the records into the
reference's bookkeeping.  The reference translates ONE sentence per call (translate_mm_vi.py:80-82); here a batch holds any
number of sentences, each with its own beam and its own stopping position, i.e. the result for sentence b equals the
reference's result for a batch holding only sentence b (sources sorted by decreasing length, as for training).
Copy attention is outside the path (the VI models are built without it) and raises."""
import torch

from ...decode import beam_decode, greedy_decode
from .Beam import Beam, GNMTGlobalScorer  # noqa: F401


class TranslatorMultimodalVI(object):
    def __init__(self, model, fields, beam_size=1, n_best=1, max_length=100, global_scorer=None, copy_attn=False, cuda=True,
                 beam_trace=False, min_length=0, test_img_feats=None, multimodal_model_type="vi-model1"):
        if copy_attn:
            raise NotImplementedError("copy attention is outside the VI_Model1 path")
        if n_best > beam_size:
            raise ValueError("n_best %d > beam_size %d" % (n_best, beam_size))
        self.model, self.fields = model, fields
        self.max_length, self.min_length = max_length, min_length
        self.beam_size, self.n_best = beam_size, n_best
        self.global_scorer = global_scorer
        self.multimodal_model_type = multimodal_model_type
        self.test_img_feats = test_img_feats          # accepted for signature parity; decoding does not read the image

    def _specials(self):
        stoi = getattr(self.fields["tgt"].vocab, "stoi", None) or {}
        return stoi.get("<blank>", 1), stoi.get("<s>", 2), stoi.get("</s>", 3)

    def translate_batch(self, batch, data=None, sent_idx=None):
        src, src_lengths = batch.src
        if src.dim() == 3:
            src = src[:, :, 0]
        if self.beam_size == 1 and self.global_scorer is None and not self.min_length:
            return self._greedy(batch, src, src_lengths)
        return self._beam(batch, src, src_lengths)

    def _greedy(self, batch, src, src_lengths):
        _pad, bos, eos = self._specials()
        toks, logp = greedy_decode(self.model.engine, src, src_lengths, max_len=self.max_length, bos=bos, eos=eos)
        toks, logp = toks.cpu(), logp.cpu()            # ONE device-to-host copy per batch
        B = toks.shape[1]
        ret = {"predictions": [], "scores": [], "attention": [], "gold_score": [0] * B, "batch": batch}
        for b in range(B):
            col = toks[:, b].tolist()
            n = col.index(eos) + 1 if eos in col else len(col)
            ret["predictions"].append([col[:n]])
            ret["scores"].append([float(logp[:n, b].sum())])
            ret["attention"].append([None])
        return ret

    def _replay(self, rec, B, with_attn):
        pad, bos, eos = self._specials()
        beams = [Beam(self.beam_size, pad, bos, eos, n_best=self.n_best, global_scorer=self.global_scorer,
                      min_length=self.min_length) for _ in range(B)]
        n = rec["scores"].shape[0]
        if self.global_scorer is None or (type(self.global_scorer) is GNMTGlobalScorer and with_attn):
            # a beam advances until it is done (Beam.py:117-124): the number of positions each one takes, for the whole batch at once
            # (see `stop` in _beam), then every beam's bookkeeping in whole-array operations
            fin = rec["next"] == eos
            done = (fin[:, :, 0].cumsum(0) > 0) & (fin.sum(2).cumsum(0) >= self.n_best)           # [n, B]
            steps = torch.where(done.any(0), done.int().argmax(0) + 1, torch.full((B,), n, dtype=torch.int64)).tolist()
            lens = rec["src_len"].tolist()
            for b, bm in enumerate(beams):
                T = steps[b]
                bm.load_records(rec["scores"][:T, b], rec["prev"][:T, b], rec["next"][:T, b],
                                rec["attn"][:T, :, b, :int(lens[b])] if with_attn else None)
            return beams
        for b, bm in enumerate(beams):
            for t in range(n):
                if bm.done():
                    break
                at = rec["attn"][t, :, b, :int(rec["src_len"][b])] if with_attn else None
                bm.advance_from_device(rec["scores"][t, b], rec["prev"][t, b], rec["next"][t, b], at)
        return beams

    def _beam(self, batch, src, src_lengths):
        pad, bos, eos = self._specials()
        B = int(src.shape[1])
        n_best = self.n_best

        def stop(rec):
            # Beam.done for every sentence, from the records alone: eos_top = </s> has headed the beam at some position; finished counts
            # one entry per </s> on the beam (Beam.py:108-124).  Both only grow, and a beam that is done stops advancing, so "done at
            # some position <= n" is what the replay below will find.  (Replaying the Beam mirror here cost 10x the device time.)
            fin = rec["next"] == eos                                  # [n, B, K]
            top = fin[:, :, 0].cumsum(0) > 0
            cnt = fin.sum(2).cumsum(0)
            return bool(((top & (cnt >= n_best)).any(0)).all())
        rec = beam_decode(self.model.engine, src, src_lengths, self.beam_size, max_len=self.max_length, min_length=self.min_length,
                          bos=bos, eos=eos, pad=pad, stop=stop)
        beams = self._replay(rec, B, True)
        ret = {"predictions": [], "scores": [], "attention": [], "gold_score": [0] * B, "batch": batch}
        for bm in beams:                               # _from_beam, TranslatorMultimodalVI.py:226-243
            scores, ks = bm.sort_finished(minimum=self.n_best)
            hyps, attn = [], []
            for times, k in ks[:self.n_best]:
                hyp, att = bm.get_hyp(times, k)
                hyps.append([int(x) for x in hyp])
                attn.append(att)
            ret["predictions"].append(hyps)
            ret["scores"].append([float(s) for s in scores])
            ret["attention"].append(attn)
        self.last_beams = beams
        return ret
