"""Timeline of the last `count` kernel dispatches of a rocprofv3 --kernel-trace results.db (start offset, duration, queue).
usage: python tools/trace_timeline.py <results.db> [skip_from_end] [count]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.{qcol} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 200
sel = rows[len(rows) - skip - count: len(rows) - skip]
t0 = sel[0][1]
prev_end = {}
for name, st, en, gx, gy, q in sel:
    short = name.split("(")[0].replace("(anonymous namespace)::", "")[:44]
    gap = (st - prev_end[q]) / 1e3 if q in prev_end else 0.0
    print("%9.1f us  +%7.1f us  gapq %6.1f  q%-3s %-44s grid %dx%d" % ((st - t0) / 1e3, (en - st) / 1e3, gap, q, short, gx, gy))
    prev_end[q] = en
