"""Report for tools/ubench/graph_order.py from the rocprofv3 kernel-trace database:  python graph_order_report.py <dir> [K=10]"""
import glob, os, sqlite3, sys
d = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
db = sqlite3.connect(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
qcol = "queue_id" if "queue_id" in cols else "0"
rows = db.execute(f"select name, start, end, {qcol} from kernels order by start").fetchall()
def kind(n):
    for k in ("tanh", "cos", "sin", "add", "mul"):
        if k in n.lower():
            return k
    return "?"
rows = [(kind(n), s, e, q) for n, s, e, q in rows]
tags = [i for i, r in enumerate(rows) if r[0] == "tanh"]
names = ["early/back-to-back", "early/synced", "late/back-to-back", "late/synced"]
for t, name in zip(range(len(tags)), names):
    seg = rows[tags[t] + 1: tags[t + 1] if t + 1 < len(tags) else len(rows)]
    starts = [i for i, r in enumerate(seg) if r[0] == "cos"]
    lat, chain, total, qs = [], [], [], set()
    for a, b in zip(starts, starts[1:] + [len(seg)]):
        rep = seg[a:b]
        adds = [r for r in rep if r[0] == "add"]
        sins = [r for r in rep if r[0] == "sin"]
        if len(adds) <= K or not sins:
            continue
        lat.append((sins[0][1] - adds[K][2]) / 1e3)
        chain.append((adds[-1][2] - adds[0][1]) / 1e3 / len(adds))
        total.append((max(r[2] for r in rep) - rep[0][1]) / 1e3)
        qs |= {(r[0], r[3]) for r in rep}
    lat.sort(); total.sort()
    if lat:
        print(f"{name:20s} replays {len(lat):3d}  side branch starts {lat[len(lat)//2]:7.1f} us (min {lat[0]:.1f}, max {lat[-1]:.1f}) after node K;"
              f" chain cadence {sum(chain)/len(chain):.2f} us/kernel; replay {total[len(total)//2]:.1f} us; queues {sorted(qs)}")
