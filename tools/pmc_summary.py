import csv, glob, sys, collections
def load(d):
    f = glob.glob(d + '/*/*counter_collection.csv')
    if not f: return {}
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return agg
fe, wr = load(sys.argv[1]), load(sys.argv[2])
print("per-launch HBM traffic from PMC (FETCH_SIZE x2 gfx950 correction for wide coalesced reads; units KiB -> bytes)")
print("%-70s %8s %12s %12s" % ("kernel", "launches", "read MB", "write MB"))
names = sorted(set(fe) | set(wr), key=lambda n: -(sum(fe.get(n, [0])) + sum(wr.get(n, [0]))))
for n in names[:16]:
    f, w = fe.get(n, []), wr.get(n, [])
    rd = (sum(f) / max(1, len(f))) * 1024 * 2 / 1e6
    wt = (sum(w) / max(1, len(w))) * 1024 / 1e6
    print("%-70s %8d %12.1f %12.1f" % (n.replace('void vmmt::', '').replace('unsigned short', 'bf16')[:70], max(len(f), len(w)), rd, wt))

# --fragment OUT.json --config N: the dominant sweep kernel's bytes per launch as a traffic.json fragment (tools/traffic_key.py merge)
if "--fragment" in sys.argv:
    import json
    out = sys.argv[sys.argv.index("--fragment") + 1]
    cfg = "config" + (sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "2")
    cand = [n for n in names if "gen2p_kernel" in n or "gen2w_kernel" in n or "gen2_kernel" in n]
    if cand:
        n = cand[0]
        f, w = fe.get(n, []), wr.get(n, [])
        ent = {"kernel": n.replace("void vmmt::", "").split("(")[0], "read_bytes": round(sum(f) / max(1, len(f)) * 1024 * 2), "write_bytes": round(sum(w) / max(1, len(w)) * 1024),
               "launches_averaged": max(len(f), len(w)),
               "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `python bench.py --no-cpu-baseline --steps 3 --warmup 2` (tools/pmc.sh), "
                      "KiB -> bytes, FETCH_SIZE doubled (gfx950 wide-read correction, MI355X_MICROARCH.md section HBM); mean over the launches of the run"}
        json.dump({cfg: ent}, open(out, "w"), indent=1)
        print("wrote", out, ent["kernel"], ent["read_bytes"], ent["write_bytes"])
