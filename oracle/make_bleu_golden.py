"""TEST INFRASTRUCTURE -- generates tests/golden/bleu_cases.json: hypothesis / reference texts and what the reference's
BLEU pipeline prints for them, obtained by RUNNING the script where it lies (/root/reference/tools/multi-bleu.perl, through the
same `sed` de-BPE step and `cut`s as onmt/EarlyStop.py:205-243).  Build container only (needs perl + the mounted reference)."""
import json
import os
import random
import subprocess
import tempfile

SCRIPT = "/root/reference/tools/multi-bleu.perl"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "bleu_cases.json")
VOCAB = ["der", "die", "das", "ein", "eine", "mann", "frau", "hund", "läuft", "sitzt", "auf", "dem", "straße", "park", "ro@@", "ter",
         "gro@@", "ßen", "bank", ".", ",", "und", "mit", "kind@@", "ern", "Ein", "Mann"]


def run(hyp_text, ref_text, lc=False):
    with tempfile.TemporaryDirectory() as d:
        hp, rp = os.path.join(d, "hyp"), os.path.join(d, "ref")
        open(hp, "w", encoding="utf-8").write(hyp_text)
        open(rp, "w", encoding="utf-8").write(ref_text)
        raw = subprocess.run("perl %s %s %s < %s" % (SCRIPT, "-lc" if lc else "", rp, hp), shell=True, capture_output=True,
                             text=True).stdout.rstrip("\n")
        # EarlyStop.compute_bleus: de-BPE both sides with sed, score, cut out the number
        piped = subprocess.run("sed -r 's/(@@ )|(@@ ?$)//g' %s > %s.w; cat %s | sed -r 's/(@@ )|(@@ ?$)//g' | perl %s %s.w | cut -d, -f1 | "
                               "cut -d' ' -f3" % (rp, rp, hp, SCRIPT, rp), shell=True, capture_output=True, text=True).stdout.strip()
    return raw, piped


def main():
    rnd = random.Random(7)
    sent = lambda n: " ".join(rnd.choice(VOCAB) for _ in range(n))
    cases = []
    for i in range(24):
        n = rnd.randint(1, 25)
        refs = [sent(rnd.randint(1, 14)) for _ in range(n)]
        hyps = []
        for r in refs:
            w = r.split()
            if rnd.random() < 0.6 and len(w) > 2:
                w[rnd.randrange(len(w))] = rnd.choice(VOCAB)
            if rnd.random() < 0.3:
                w = w[:max(1, len(w) - 2)]
            if rnd.random() < 0.2:
                w = w + [rnd.choice(VOCAB)]
            if rnd.random() < 0.08:
                w = []
            hyps.append(" ".join(w))
        if i % 6 == 3:
            hyps = hyps[:max(1, n - 2)]              # fewer hypotheses than references
        if i % 6 == 4:
            hyps = hyps + [sent(3)]                  # more hypotheses than references (the 9999 quirk)
        hyp_text, ref_text = "\n".join(hyps) + "\n", "\n".join(refs) + "\n"
        if i % 8 == 5:
            hyp_text = hyp_text[:-1]                 # no newline at the end: `chop` eats a real character
        cases.append(dict(hyp=hyp_text, ref=ref_text, lc=bool(i % 5 == 0)))
    cases.append(dict(hyp="a b c d e\n", ref="a b c d e\n", lc=False))
    cases.append(dict(hyp="a b\n", ref="a b\n", lc=False))                      # no 3-/4-grams at all
    cases.append(dict(hyp="x y z w\n", ref="a b c d\n", lc=False))
    cases.append(dict(hyp="a  b   c d\n", ref="a b c d\n", lc=False))            # runs of spaces
    for c in cases:
        c["line"], c["piped"] = run(c["hyp"], c["ref"], c["lc"])
    json.dump(cases, open(OUT, "w", encoding="utf-8"), ensure_ascii=False, indent=0)
    for c in cases:
        print(repr(c["line"]), repr(c["piped"]))


if __name__ == "__main__":
    main()


def earlystop_cases():
    """decisions of the reference's OWN EarlyStop (onmt/EarlyStop.py) on sequences of validation scores: its translate / BLEU /
    METEOR steps are replaced by the next score of the sequence, everything else (`add_run`, `_do_early_stop`) runs as is."""
    import contextlib
    import io
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import ref_harness as RH
    onmt, _ = RH.import_reference()
    import onmt.EarlyStop
    rnd = random.Random(11)
    out = []
    for patience in (3, 4, 6, 10):
        for _rep in range(4):
            seq = [round(rnd.uniform(10, 40) + (i if rnd.random() < 0.5 else -i) * rnd.random(), 2) for i in range(14)]
            if _rep == 3:
                seq = sorted(seq)[:7] + [seq[0]] * 7          # plateau with ties
            es = onmt.EarlyStop.EarlyStop("src", "tgt", "bleu", 0, 500, patience, multimodal_model_type="vi-model1", img_fname="x")
            it = iter(seq)
            es.translate_ = lambda *a: None
            es.compute_bleus = lambda *a: ([""], [str(next(it))], [""])
            es.compute_meteors = lambda *a: ([""], [0.0], [""])
            best, stop = [], []
            with contextlib.redirect_stdout(io.StringIO()):
                for n in range(len(seq)):
                    best.append(bool(es.add_run("snapshot", (n + 1) * 500)))
                    stop.append(bool(es.signal_early_stopping))
            out.append(dict(patience=patience, scores=seq, is_best=best, stop=stop))
    return out


if __name__ == "__main__":
    p = os.path.join(os.path.dirname(OUT), "earlystop_cases.json")
    cases = earlystop_cases()
    json.dump(cases, open(p, "w"), indent=0)
    print("early-stop cases:", len(cases), [c["stop"].index(True) if True in c["stop"] else -1 for c in cases])
