"""Timeline of one training step from a rocprofv3 rocpd sqlite database (kernels view).  usage: timeline_db.py results.db [min_us]"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
minus = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
rows = c.execute("select name, start, end, queue_id, grid_x, workgroup_x from kernels order by start").fetchall()
adam = [i for i, r in enumerate(rows) if "adam_flat_dev_kernel" in r[0]]
a, b = adam[-3], adam[-2]
seg = rows[a + 1:b + 1]
base = seg[0][1]
qs = sorted(set(r[3] for r in seg))
def short(n):
    return n.replace("void mser::", "").replace("mser::", "").split("(")[0][:40]
print("queues", len(qs), " step span %.1f us" % ((seg[-1][2] - base) / 1e3))
for r in seg:
    st, du = (r[1] - base) / 1e3, (r[2] - r[1]) / 1e3
    if du >= minus or "persist" in r[0] or "fused" in r[0]:
        print(f"q{qs.index(r[3])} {st:8.1f} +{du:7.1f} grid {r[4]//max(r[5],1):5d}  {short(r[0])}")
for q in qs:
    rs = [r for r in seg if r[3] == q]
    print(f"queue {qs.index(q)}: n={len(rs):4d} first {(min(r[1] for r in rs)-base)/1e3:8.1f} last {(max(r[2] for r in rs)-base)/1e3:8.1f} busy {sum(r[2]-r[1] for r in rs)/1e3:8.1f}")
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    k = short(r[0]); agg[k][0] += 1; agg[k][1] += (r[2] - r[1]) / 1e3
print("--- per-kernel totals in this step")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{t:9.1f} us  n={n:4d}  avg {t/n:7.1f}  {k}")
