#!/usr/bin/env python3
"""Where does a chain of small dependent kernels spend its time?  From a rocprofv3 --kernel-trace rocpd database: per hardware queue,
split the dispatches into busy segments (idle > 60 us ends a segment) and report per segment the span, the kernel time, the gaps, and
which (previous kernel -> next kernel) transitions carry the gap time.  usage: tools/queue_gaps.py results.db [kernel-name-substring]"""
import collections
import sqlite3
import sys

import numpy as np

db = sqlite3.connect(sys.argv[1])
rows = list(db.cursor().execute("select queue_id, start, end, name from kernels order by start"))
byq = collections.defaultdict(list)
for q, s, e, n in rows:
    byq[q].append((s, e, n.split("(")[0].replace("void ", "")))
want = sys.argv[2] if len(sys.argv) > 2 else None
for q, v in sorted(byq.items()):
    if len(v) < 50 or (want and not any(want in x[2] for x in v)):
        continue
    segs, cur = [], [v[0]]
    for a in v[1:]:
        if a[0] - cur[-1][1] > 60000:
            segs.append(cur); cur = [a]
        else:
            cur.append(a)
    segs.append(cur)
    segs = [s for s in segs[len(segs) // 4:] if len(s) > 3]
    if not segs:
        continue
    span = np.mean([s[-1][1] - s[0][0] for s in segs]) / 1e3
    busy = np.mean([sum(e - b for b, e, _ in s) for s in segs]) / 1e3
    top = collections.Counter(x[2] for x in v).most_common(3)
    print(f"queue {q}: {len(v)} dispatches ({', '.join(n for n, _ in top)} ...); {len(segs)} busy segments: span {span:.1f} us, "
          f"kernel time {busy:.1f} us, gaps {span - busy:.1f} us, {np.mean([len(s) for s in segs]):.1f} dispatches")
    g = collections.defaultdict(list)
    for s in segs:
        for (b0, e0, n0), (b1, e1, n1) in zip(s[:-1], s[1:]):
            g[(n0, n1)].append(b1 - e0)
    for tot, k in sorted(((sum(x) / len(segs) / 1e3, k) for k, x in g.items()), reverse=True)[:8]:
        x = g[k]
        print(f"    {tot:6.1f} us/segment  x{len(x) / len(segs):.2f}  avg {np.mean(x) / 1e3:5.1f} us   {k[0]} -> {k[1]}")
