"""Host-side mirror of onmt/translate/TranslatorMultimodalVI.py for beam size 1: `translate_batch` returns the reference's
result dictionary (predictions / scores / attention / gold_score / batch, TranslatorMultimodalVI.py:218-243), produced by
variational_mmt_amd.decode.greedy_decode.  A hypothesis ends at the first </s> (Beam.py: a finished beam stops growing) and
its score is the sum of the log-probabilities of its tokens (GNMTGlobalScorer with alpha = beta = 0).  Beam sizes > 1, copy
attention and n-best lists are outside what is built and raise NotImplementedError."""
import torch

from ...decode import greedy_decode


class TranslatorMultimodalVI(object):
    def __init__(self, model, fields, beam_size=1, n_best=1, max_length=100, global_scorer=None, copy_attn=False, cuda=True,
                 beam_trace=False, min_length=0, test_img_feats=None, multimodal_model_type="vi-model1"):
        if beam_size != 1 or n_best != 1:
            raise NotImplementedError("beam search proper (Beam.py) is not built: beam_size = n_best = 1 only (SURVEY.md 8f-2)")
        if copy_attn or min_length:
            raise NotImplementedError("copy attention / min_length are outside the hot path")
        self.model, self.fields = model, fields
        self.max_length = max_length
        self.beam_size, self.n_best = 1, 1
        self.multimodal_model_type = multimodal_model_type

    def translate_batch(self, batch, data=None, sent_idx=None):
        src, src_lengths = batch.src
        if src.dim() == 3:
            src = src[:, :, 0]
        eos = self.fields["tgt"].vocab.stoi["</s>"] if hasattr(self.fields["tgt"].vocab, "stoi") else 3
        bos = self.fields["tgt"].vocab.stoi["<s>"] if hasattr(self.fields["tgt"].vocab, "stoi") else 2
        toks, logp = greedy_decode(self.model.engine, src, src_lengths, max_len=self.max_length, bos=bos)
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
