"""step timeline from a rocprofv3 results database (--kernel-trace; the .db is what `gpurun` brings back when no csv was asked for):
python tools/timeline_db.py <results.db> [min busy us]   -- one steady-state step, consecutive launches of one kernel merged"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
floor = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = db.execute("select s.kernel_name, d.start, d.end, d.queue_id from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
first = [i for i, r in enumerate(rows) if 'prepare_batch' in r[0]]
step = rows[first[-3]:first[-2]]
t0 = step[0][1]
queues = {q: i for i, q in enumerate(sorted(set(r[3] for r in step)))}


def short(n):
    n = re.sub(r'^_ZN4vmmt\d+', '', n)
    n = re.sub(r'^_Z\d+', '', n)
    return n[:56]


print("kernels: %d  span %.1f us" % (len(step), (max(r[2] for r in step) - t0) / 1e3))
cur, out = None, []
for n, s, e, q in step:
    key = (queues[q], short(n))
    s, e = (s - t0) / 1e3, (e - t0) / 1e3
    if cur and cur[0] == key:
        cur[2] = e; cur[3] += 1; cur[4] += e - s
    else:
        if cur:
            out.append(cur)
        cur = [key, s, e, 1, e - s]
out.append(cur)
for (q, name), s, e, n, busy in out:
    if busy >= floor:
        print("%9.1f -> %9.1f  [%d] x%-3d busy %8.1f  %s" % (s, e, q, n, busy, name))
