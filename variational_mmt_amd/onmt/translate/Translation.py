"""Predictions of `TranslatorMultimodalVI.translate_batch` -> words (mirror of onmt/translate/Translation.py:7-151 as the driver
translate_mm_vi.py:133-160 uses it): `TranslationBuilder(data, fields, n_best, replace_unk, has_tgt).from_batch(result)` returns one
`Translation` per sentence, in corpus order, with `pred_sents` (n-best word lists, cut before </s>), `pred_scores`, `attns`,
`gold_sent` / `gold_score` and `log(n)`.  Copy-attention vocabularies (ids beyond the target vocabulary) are outside the VI_Model1
path; `replace_unk` substitutes the most attended source word for <unk> (Translation.py:41-45)."""
import torch

from .. import io


class Translation(object):
    def __init__(self, src, src_raw, pred_sents, attn, pred_scores, tgt_sent, gold_score):
        self.src, self.src_raw, self.pred_sents, self.attns = src, src_raw, pred_sents, attn
        self.pred_scores, self.gold_sent, self.gold_score = pred_scores, tgt_sent, gold_score

    def log(self, sent_number):
        lines = ["", "SENT %d: %s" % (sent_number, self.src_raw), "PRED %d: %s" % (sent_number, " ".join(self.pred_sents[0]))]
        print("PRED SCORE: %.4f" % self.pred_scores[0])
        out = "\n".join(lines) + "\n"
        if self.gold_sent is not None:
            out += "GOLD %d: %s\n" % (sent_number, " ".join(self.gold_sent)) + "GOLD SCORE: %.4f" % self.gold_score
        if len(self.pred_sents) > 1:
            print("\nBEST HYP:")
            for score, sent in zip(self.pred_scores, self.pred_sents):
                out += "[%.4f] %s\n" % (score, sent)
        return out


class TranslationBuilder(object):
    def __init__(self, data, fields, n_best=1, replace_unk=False, has_tgt=False):
        self.data, self.fields, self.n_best, self.replace_unk, self.has_tgt = data, fields, n_best, replace_unk, has_tgt

    def _words(self, ids, src_raw, attn):
        vocab = self.fields["tgt"].vocab
        words = []
        for t in ids:
            t = int(t)
            if t >= len(vocab):
                raise NotImplementedError("token id %d beyond the target vocabulary: copy attention is outside the VI_Model1 path" % t)
            if vocab.itos[t] == io.EOS_WORD:
                break
            words.append(vocab.itos[t])
        if self.replace_unk and attn is not None and src_raw is not None:
            unk = vocab.itos[io.UNK]
            for i, w in enumerate(words):
                if w == unk:
                    words[i] = src_raw[int(torch.as_tensor(attn[i]).argmax())]
        return words

    def from_batch(self, translation_batch):
        batch = translation_batch["batch"]
        n = batch.batch_size
        assert len(translation_batch["gold_score"]) == len(translation_batch["predictions"]) == n
        idx = torch.as_tensor(batch.indices).cpu().tolist()
        order = sorted(range(n), key=lambda j: idx[j])                       # corpus order
        src = batch.src[0].cpu() if isinstance(batch.src, tuple) else batch.src.cpu()
        tgt = None
        if self.has_tgt:
            tgt = (batch.tgt[0] if isinstance(batch.tgt, tuple) else batch.tgt).cpu()
        out = []
        for j in order:
            raw = self.data.examples[idx[j]].src
            attn = translation_batch["attention"][j]
            preds = [self._words(translation_batch["predictions"][j][k], raw, attn[k] if attn is not None else None)
                     for k in range(self.n_best)]
            gold = self._words(tgt[1:, j].tolist(), raw, None) if tgt is not None else None
            s = src[:, j] if src.dim() == 2 else src[:, j, 0]
            out.append(Translation(s, raw, preds, attn, translation_batch["scores"][j], gold, translation_batch["gold_score"][j]))
        return out
