"""Corpus BLEU exactly as `tools/multi-bleu.perl` prints it (the Moses script the reference pipes its validation
translations through, onmt/EarlyStop.py:205-243 via onmt/Utils.py:4), restated in Python so that BLEU-driven early stopping
needs no perl subprocess.  Kept quirks: every line loses its LAST character (`chop`, so a final line without a newline loses
a real one); tokens are whitespace-separated; the reference length of a sentence is the closest (ties: shorter) among its
references, 9999 when the hypothesis has more lines than the references; an n-gram order that never matches contributes
log = -9999999999 (BLEU 0.00); the printed score has two decimals and EarlyStop reads it back from the text.
`debpe` is the `sed -r 's/(@@ )|(@@ ?$)//g'` step applied to both sides first (EarlyStop.py:212-221)."""
import math
import re
from collections import Counter

_BPE = re.compile(r"(@@ )|(@@ ?$)")


def debpe(line):
    """one line WITHOUT its newline -> sub-word units joined back into words"""
    return _BPE.sub("", line)


def _chop(line):
    return line[:-1]


def _ngrams(words, n):
    return Counter(tuple(words[i:i + n]) for i in range(len(words) - n + 1))


def multi_bleu(hyp_lines, ref_lines_list, lowercase=False):
    """hyp_lines: the hypothesis file's lines INCLUDING their newlines (as read from the file); ref_lines_list: one such list per
    reference file.  Returns dict(bleu, precisions[4], bp, ratio, hyp_len, ref_len, line) with `line` the script's output line
    (None where the script dies: empty hypothesis against a non-empty reference)."""
    refs = [[_chop(x) for x in r] for r in ref_lines_list]
    correct, total = [0] * 5, [0] * 5
    seen = [False] * 5
    len_t = len_r = 0
    for s, raw in enumerate(hyp_lines):
        h = _chop(raw)
        if lowercase:
            h = h.lower()
        words = h.split()
        ref_max = [Counter() for _ in range(5)]
        closest_diff, closest_len = 9999, 9999
        for r in refs:
            if s >= len(r):
                continue
            rw = (r[s].lower() if lowercase else r[s]).split()
            diff = abs(len(words) - len(rw))
            if diff < closest_diff:
                closest_diff, closest_len = diff, len(rw)
            elif diff == closest_diff and len(rw) < closest_len:
                closest_len = len(rw)
            for n in range(1, 5):
                for g, c in _ngrams(rw, n).items():
                    if ref_max[n][g] < c:
                        ref_max[n][g] = c
        len_t += len(words)
        len_r += closest_len
        for n in range(1, 5):
            for g, c in _ngrams(words, n).items():
                seen[n] = True
                total[n] += c
                correct[n] += min(c, ref_max[n].get(g, 0))
    prec = [0.0] * 5
    for n in range(1, 5):
        prec[n] = (correct[n] / total[n]) if (seen[n] and total[n]) else 0.0
    if len_r == 0:
        return dict(bleu=0.0, precisions=[0.0] * 4, bp=0.0, ratio=0.0, hyp_len=0, ref_len=0,
                    line="BLEU = 0, 0/0/0/0 (BP=0, ratio=0, hyp_len=0, ref_len=0)")
    if len_t == 0:
        return dict(bleu=0.0, precisions=[0.0] * 4, bp=0.0, ratio=0.0, hyp_len=0, ref_len=len_r, line=None)   # the script dies here
    bp = math.exp(1 - len_r / len_t) if len_t < len_r else 1.0
    lg = lambda x: math.log(x) if x else -9999999999.0
    bleu = bp * math.exp((lg(prec[1]) + lg(prec[2]) + lg(prec[3]) + lg(prec[4])) / 4)
    line = "BLEU = %.2f, %.1f/%.1f/%.1f/%.1f (BP=%.3f, ratio=%.3f, hyp_len=%d, ref_len=%d)" % (
        100 * bleu, 100 * prec[1], 100 * prec[2], 100 * prec[3], 100 * prec[4], bp, len_t / len_r, len_t, len_r)
    return dict(bleu=100 * bleu, precisions=[100 * p for p in prec[1:]], bp=bp, ratio=len_t / len_r, hyp_len=len_t, ref_len=len_r,
                line=line)


def score_files(hyp_path, ref_path, bpe=True):
    """the pipeline of EarlyStop.compute_bleus (cat | sed | multi-bleu.perl ref | cut -d, -f1 | cut -d' ' -f3) -> the score as
    the STRING the reference stores ('23.45')."""
    def read(p):
        with open(p, "r", encoding="utf-8", newline="") as f:
            lines = f.read().split("\n")
        if lines and lines[-1] == "":
            lines.pop()
            return [(debpe(x) if bpe else x) + "\n" for x in lines]
        # no newline at the end of the file: GNU sed keeps it that way, so the script's `chop` eats a real character
        return [(debpe(x) if bpe else x) + "\n" for x in lines[:-1]] + [debpe(lines[-1]) if bpe else lines[-1]]
    r = multi_bleu(read(hyp_path), [read(ref_path)])
    if r["line"] is None:
        return ""
    return r["line"].split(",")[0].split(" ")[2]
