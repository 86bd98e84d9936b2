import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# step boundaries: prepare_batch_kernel is the first kernel of every step's forward plan (the background half of the previous
# step's Adam runs a little later, underneath the encoder: it belongs to the window it executes in)
first = [i for i, r in enumerate(rows) if 'prepare_batch_kernel' in r['Kernel_Name']]
i0, i1 = first[-3], first[-2]                  # one full steady-state step
step = rows[i0:i1]
t0 = int(step[0]['Start_Timestamp'])
def short(n):
    n = n.replace('void vmmt::', '').replace('vmmt::', '').replace('unsigned short', 'bf16')
    return n[:46]
streams = sorted(set(r['Stream_Id'] for r in step))
print("streams:", streams, "kernels:", len(step), "span %.1f us" % ((int(step[-1]['End_Timestamp']) - t0) / 1e3))
# phase summary on main stream: merge consecutive same-name kernels
main = streams[0]
cur = None
out = []
for r in step:
    name = short(r['Kernel_Name']); s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
    key = (r['Stream_Id'], name)
    if cur and cur[0] == key:
        cur[2] = e; cur[3] += 1; cur[4] += e - s
    else:
        if cur: out.append(cur)
        cur = [key, s, e, 1, e - s]
out.append(cur)
for (sid, name), s, e, n, busy in out:
    print("%8.1f -> %8.1f  [%s] x%-3d busy %7.1f  %s" % (s, e, sid, n, busy, name))
