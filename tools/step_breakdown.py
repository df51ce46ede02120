#!/usr/bin/env python
"""Per-kernel time of ONE step out of a rocprofv3 --kernel-trace database (rocpd .db): the dispatches are cut into steps at
idle gaps > 0.3 ms (the host's synchronize between steps), the last long segment is listed -- span, busy time and the
kernels by total time.  usage: python tools/step_breakdown.py <results.db> [name width]"""
import collections
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    rows = list(db.cursor().execute("select name,start,end from kernels order by start"))
    seg, s = [], 0
    for i in range(len(rows) - 1):
        if rows[i + 1][1] - rows[i][2] > 3e5:
            seg.append((s, i))
            s = i + 1
    seg.append((s, len(rows) - 1))
    seg = [(a, b) for a, b in seg if b - a > 900]
    print("# segments (first dispatch, last dispatch, span ms):", [(a, b, round((rows[b][2] - rows[a][1]) / 1e6, 2)) for a, b in seg])
    a, b = seg[-1]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for n, s, e in rows[a:b + 1]:
        k = re.sub(r"^void ", "", n)[:width]
        agg[k][0] += 1
        agg[k][1] += (e - s) / 1e6
    print("# last segment: %d dispatches, busy %.3f ms of a %.3f ms span" % (b - a + 1, sum(v[1] for v in agg.values()),
                                                                            (rows[b][2] - rows[a][1]) / 1e6))
    print("calls  total_ms   avg_us  kernel")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
        print("%5d %9.3f %8.1f  %s" % (v[0], v[1], v[1] / v[0] * 1e3, k))


if __name__ == "__main__":
    main()
