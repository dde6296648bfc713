"""Per-kernel breakdown of the second half (the timed step) of a rocprofv3 --kernel-trace results.db.
usage: python tools/trace_breakdown.py <results.db> [units_per_step]"""
import collections
import sqlite3
import statistics
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
t0, t1 = rows[0][1], rows[-1][2]
half = [r for r in rows if r[1] > (t0 + t1) / 2]
units = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: [0, 0.0])
for r in half:
    agg[r[0][:78]][0] += 1
    agg[r[0][:78]][1] += r[2] - r[1]
span = half[-1][2] - half[0][1]
busy = sum(r[2] - r[1] for r in half)
print("span ms %.2f  busy ms %.2f  per unit ms %.3f" % (span / 1e6, busy / 1e6, span / 1e6 / units))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-80s %6d %9.3f ms %9.1f us avg  %5.1f%%" % (k, v[0], v[1] / 1e6, v[1] / 1e3 / v[0], 100 * v[1] / span))
gaps = [half[i + 1][1] - half[i][2] for i in range(len(half) - 1)]
print("gaps: mean us %.2f  sum ms %.2f" % (statistics.mean(gaps) / 1e3, sum(g for g in gaps if g > 0) / 1e6))
