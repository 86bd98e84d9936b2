"""TEST INFRASTRUCTURE -- lists the command-line flags the REAL reference's parsers define (opts.py imported from /root/reference
through oracle/ref_harness.py) into tests/golden/opts_flags.json: per parser function, per flag: option strings, destination,
default, type name, choices, nargs, action class, required.  Data only (no source text).  Run in the build container:
    python oracle/make_opts_golden.py
"""
import argparse
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_harness as RH          # noqa: E402

FUNCS = ("model_opts", "train_opts", "train_mm_vi_model1_opts", "translate_opts", "translate_mm_vi_opts", "add_md_help_argument")


def listing(opts_module):
    out = {}
    for fn in FUNCS:
        p = argparse.ArgumentParser()
        getattr(opts_module, fn)(p)
        rows = []
        for a in p._actions:
            if isinstance(a, argparse._HelpAction):
                continue
            rows.append(dict(flags=list(a.option_strings), dest=a.dest, default=a.default,
                             type=getattr(a.type, "__name__", None) if a.type else None,
                             choices=list(a.choices) if a.choices else None, nargs=a.nargs, action=type(a).__name__,
                             required=bool(a.required)))
        out[fn] = rows
    return out


if __name__ == "__main__":
    _, opts = RH.import_reference()
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "opts_flags.json")
    json.dump(listing(opts), open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)
