"""Host mirror of onmt/translate/Beam.py for the device beam search (variational_mmt_amd.decode.beam_decode).

The arithmetic of `Beam.advance` (log-probabilities + running scores, </s> masking, top-K over K x V, Beam.py:63-103) runs on
the GPU (`vmmt_beam_advance`); this class receives each position's result -- scores, parent beams, tokens, attention rows --
through `advance_from_device` and keeps the reference's bookkeeping as executed: `finished` entries with their globally
re-scored value (Beam.py:108-115), the stopping rule (`eos_top` and `done`, :117-124), `sort_finished` including its habit
of padding with beam 0 over and over (its loop index is never advanced, :134-142) and `get_hyp` (:150-160).
`GNMTGlobalScorer` is the reference's length / coverage re-scoring (:163-183)."""
import torch


class GNMTGlobalScorer(object):
    def __init__(self, alpha, beta):
        self.alpha, self.beta = alpha, beta

    def score(self, beam, logprobs):
        cov = beam.global_state["coverage"]
        pen = self.beta * torch.min(cov, torch.ones_like(cov)).log().sum(1)
        l_term = ((5 + len(beam.next_ys)) ** self.alpha) / ((5 + 1) ** self.alpha)
        return logprobs / l_term + pen

    def update_global_state(self, beam):
        if len(beam.prev_ks) == 1:
            beam.global_state["coverage"] = beam.attn[-1]
        else:
            beam.global_state["coverage"] = beam.global_state["coverage"].index_select(0, beam.prev_ks[-1]).add(beam.attn[-1])


class Beam(object):
    def __init__(self, size, pad, bos, eos, n_best=1, cuda=False, global_scorer=None, min_length=0):
        self.size = size
        self.scores = torch.zeros(size, device="cpu")
        self.all_scores = []
        self.prev_ks = []
        self.next_ys = [torch.full((size,), pad, dtype=torch.int64, device="cpu")]
        self.next_ys[0][0] = bos
        self._eos = eos
        self.eos_top = False
        self.attn = []
        self.finished = []
        self.n_best = n_best
        self.global_scorer = global_scorer
        self.global_state = {}
        self.min_length = min_length

    def get_current_state(self):
        return self.next_ys[-1]

    def get_current_origin(self):
        return self.prev_ks[-1]

    def advance_from_device(self, best_scores, prev_k, next_y, attn_rows=None):
        """one position computed by vmmt_beam_advance: `best_scores` [K] f32, `prev_k` [K] parents, `next_y` [K] tokens,
        `attn_rows` [K, S_b] = this position's attention of the K decoder rows BEFORE re-ordering (None: not recorded)."""
        prev_k = prev_k.to(torch.int64)
        self.all_scores.append(self.scores)
        self.scores = best_scores
        self.prev_ks.append(prev_k)
        self.next_ys.append(next_y.to(torch.int64))
        if attn_rows is not None:
            self.attn.append(attn_rows.index_select(0, prev_k))
            if self.global_scorer is not None:
                self.global_scorer.update_global_state(self)
        toks = self.next_ys[-1].tolist()                  # (one host conversion instead of K tensor comparisons)
        for i in range(self.size):
            if toks[i] == self._eos:
                self.finished.append((self._final_score(i), len(self.next_ys) - 1, i))
        if toks[0] == self._eos:
            self.eos_top = True

    def load_records(self, scores, prev, nxt, attn=None):
        """`advance_from_device` for positions 0 .. T-1 at once: scores / prev / nxt [T, K] (host tensors), attn [T, K, S_b] or None.
        Same state as T calls in a row (a fresh beam; the caller cuts T at the position where the beam is done); without a global
        scorer nothing here depends on the position before it, so the bookkeeping is a handful of whole-array operations instead of
        ~10 small ones per position (which made the host replay 4x the cost of the device search)."""
        T = int(scores.shape[0])
        gs = self.global_scorer
        gnmt = type(gs) is GNMTGlobalScorer and attn is not None        # the reference's re-scoring (translate_mm_vi.py always builds one)
        if (gs is not None and not gnmt) or self.prev_ks or T == 0:
            for t in range(T):
                self.advance_from_device(scores[t], prev[t], nxt[t], None if attn is None else attn[t])
            return
        prev = prev.to(torch.int64)
        nxt = nxt.to(torch.int64)
        self.all_scores = [self.scores] + list(scores[:T - 1].unbind(0))
        self.scores = scores[T - 1]
        self.prev_ks = list(prev.unbind(0))
        self.next_ys = self.next_ys + list(nxt.unbind(0))
        cov = None
        if attn is not None:
            sel = attn.gather(1, prev.unsqueeze(2).expand(T, self.size, attn.shape[2]))       # attn[t].index_select(0, prev[t]) for every t
            self.attn = list(sel.unbind(0))
            if gnmt:                                        # update_global_state position by position: T small operations
                cov = [self.attn[0]]
                for t in range(1, T):
                    cov.append(cov[-1].index_select(0, self.prev_ks[t]).add(self.attn[t]))
                self.global_state["coverage"] = cov[-1]
        fin = (nxt == self._eos)
        final = {}
        for t, i in fin.nonzero().tolist():                 # row-major: position by position, beam by beam, as the loop appends them
            if gnmt:
                if t not in final:                          # GNMTGlobalScorer.score on the state the beam had at that position
                    pen = gs.beta * torch.min(cov[t], torch.ones_like(cov[t])).log().sum(1)
                    l_term = ((5 + (t + 2)) ** gs.alpha) / ((5 + 1) ** gs.alpha)
                    final[t] = scores[t] / l_term + pen
                self.finished.append((final[t][i], t + 1, i))
            else:
                self.finished.append((scores[t, i], t + 1, i))
        self.eos_top = bool(fin[:, 0].any())

    def _final_score(self, i):
        if self.global_scorer is not None and self.attn:
            return self.global_scorer.score(self, self.scores)[i]
        return self.scores[i]

    def done(self):
        return self.eos_top and len(self.finished) >= self.n_best

    def sort_finished(self, minimum=None):
        if minimum is not None:
            while len(self.finished) < minimum:                      # as executed: beam 0 every time
                self.finished.append((self._final_score(0), len(self.next_ys) - 1, 0))
        self.finished.sort(key=lambda a: -a[0])
        return [sc for sc, _, _ in self.finished], [(t, k) for _, t, k in self.finished]

    def get_hyp(self, timestep, k):
        hyp, attn = [], []
        for j in range(len(self.prev_ks[:timestep]) - 1, -1, -1):
            hyp.append(self.next_ys[j + 1][k])
            if self.attn:
                attn.append(self.attn[j][k])
            k = self.prev_ks[j][k]
        return hyp[::-1], (torch.stack(attn[::-1]) if attn else None)
