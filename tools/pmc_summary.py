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
