import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find step boundaries: adam_kernel occurrences
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
i0, i1 = adam[-3] + 1, adam[-2] + 1           # one full step
step = rows[i0:i1]
# include pack kernels after adam up to next gather? keep simple
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
