"""profiles/traffic.json holds the HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (bench.py prints them as
`roofline.traffic`).  Every entry names the SOURCE TEXT it was measured on: the kernel's region of its .hip file (between
`// >>> traffic-key NAME` and `// <<< traffic-key NAME`) plus the `common` region of helpers it uses, as a sha256.  Editing another
kernel of the same file leaves the entry valid; editing this kernel makes tests/test_traffic_json.py fail until the PMC pass is re-run.

    python tools/traffic_key.py                       the current hashes
    python tools/traffic_key.py merge FRAGMENT.json   take `configN` entries measured on the GPU box (tools/pmc_summary.py --fragment) into
                                                      profiles/traffic.json, stamped with the current hash of their kernel's region
"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCE = "variational_mmt_amd/csrc/generator_fused.hip"
KERNEL_OF_CONFIG = {"config2": "gen2p", "config5": "gen2w"}


def region(text, name):
    m = re.search(r"// >>> traffic-key %s\b[^\n]*\n(.*?)// <<< traffic-key %s\b" % (name, name), text, re.S)
    if m is None:
        raise KeyError("no traffic-key region %r in %s" % (name, SOURCE))
    return m.group(1)


def key_hash(name, root=ROOT):
    text = open(os.path.join(root, SOURCE)).read()
    h = hashlib.sha256()
    h.update(region(text, "common").encode())
    h.update(region(text, name).encode())
    return h.hexdigest()[:16]


def main():
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if len(sys.argv) >= 3 and sys.argv[1] == "merge":
        cur = json.load(open(path))
        frag = json.load(open(sys.argv[2]))
        for cfg, ent in frag.items():
            k = KERNEL_OF_CONFIG[cfg]
            ent = dict(ent, kernel_source=SOURCE, kernel_region=k, kernel_region_sha16=key_hash(k))
            ent.pop("kernel_source_sha16", None)
            cur[cfg] = dict(cur.get(cfg, {}), **ent)
            cur[cfg].pop("kernel_source_sha16", None)
        json.dump(cur, open(path, "w"), indent=1)
        print("merged", sorted(frag), "into", path)
        return
    for k in ("gen2", "gen2p", "gen2w"):
        print(k, key_hash(k))


if __name__ == "__main__":
    main()
