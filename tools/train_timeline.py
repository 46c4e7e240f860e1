"""Timeline of ONE training step (bench.py --mode train-step, augmented inputs) from a rocprofv3 kernel trace (.db): the kernels between
two launches of the frustum-geometry kernel (the first kernel of a step), start offset / duration / queue.
    python tools/train_timeline.py <kernel-trace dir> [steps from the end, default 2]"""
import glob
import os
import sqlite3
import sys


def short(name):
    n = name.split("(")[0]
    for p in ("void ", "bevamd::", "slab::", "tile::", "at::native::", "(anonymous namespace)::", "wgslab::", "bn::"):
        n = n.replace(p, "")
    return n.strip()[:90]


def main():
    d = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    db = sqlite3.connect(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    qcol = "queue_id" if "queue_id" in cols else "0"
    rows = db.execute(f"select name, start, end, {qcol} from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "lss_geometry_kernel" in r[0]]
    if len(marks) < back + 1:
        print("not enough steps in the trace", len(marks))
        return
    lo, hi = marks[-back - 1], marks[-back]
    t0 = rows[lo][1]
    qs = sorted({r[3] for r in rows[lo:hi]})
    print(f"# {hi - lo} kernels, {(rows[hi][1] - t0) / 1e3:.1f} us wall, queues {qs}")
    busy = sum(r[2] - r[1] for r in rows[lo:hi]) / 1e3
    print(f"# sum of kernel durations {busy:.1f} us")
    print("#  start us   dur us     gap   q  kernel")
    last_end = {}
    for name, s, e, q in rows[lo:hi]:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {gap:7.1f}  {qs.index(q)}  {short(name)}")


if __name__ == "__main__":
    main()
