"""Per-queue summary of the LAST EP sweep in a rocprofv3 kernel trace: python tools/sweep_summary.py <results.db>
(start = the sweep's ep_winit launch; prints the block-kernel start times, per-queue busy time and the tail after the last block kernel)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.{qcol} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
starts = [i for i, r in enumerate(rows) if "ep_winit" in r[0]]
blocks = [i for i, r in enumerate(rows) if "ep_block" in r[0]]
if len(starts) < 2:
    sys.exit("no streamed sweeps in this trace")
first_block = max(b for b in blocks if b < starts[-1])     # block kernel 0 of the last sweep starts just before its winit
sel = rows[first_block:]
t0 = sel[0][1]
end = max(r[2] for r in sel)
bk = [(r[1] - t0) / 1e3 for r in sel if "ep_block" in r[0]]
print("sweep span %.1f us, %d block kernels, mean period %.1f us, last block kernel ends %.1f us, tail %.1f us" % (
    (end - t0) / 1e3, len(bk), (bk[-1] - bk[0]) / max(1, len(bk) - 1), max((r[2] - t0) / 1e3 for r in sel if "ep_block" in r[0]),
    (end - max(r[2] for r in sel if "ep_block" in r[0])) / 1e3))
print("block periods:", " ".join("%.0f" % (b - a) for a, b in zip(bk, bk[1:])))
busy = {}
for name, st, en, q in sel:
    short = name.split("(")[0].replace("(anonymous namespace)::", "")
    busy.setdefault(q, {}).setdefault(short[:40], [0, 0.0])
    busy[q][short[:40]][0] += 1
    busy[q][short[:40]][1] += (en - st) / 1e3
for q in sorted(busy):
    tot = sum(v[1] for v in busy[q].values())
    print("queue %s: busy %.1f us" % (q, tot))
    for k, v in sorted(busy[q].items(), key=lambda kv: -kv[1][1]):
        print("    %-40s %4d  %9.1f us  avg %7.1f" % (k, v[0], v[1], v[1] / v[0]))
