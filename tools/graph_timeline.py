"""Timeline of ONE replay of the benchmark step from a rocprofv3 kernel trace (.db): every kernel with its start offset, duration
and queue, in start order — where the conv stream waits for the geometry stream, which kernels overlap, where the GPU idles.

    python tools/graph_timeline.py <kernel-trace dir> [replay index from the end, default 2] > timeline.txt

A replay is cut at the first kernel of the step (the depth raster fill); the trace must come from `bench.py` (graph mode)."""
import glob
import os
import sqlite3
import sys


def short(name):
    n = name.split("(")[0]
    for p in ("void ", "bevamd::", "slab::", "tile::", "at::native::", "(anonymous namespace)::"):
        n = n.replace(p, "")
    return n.strip()[:72]


def main():
    d = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    db = sqlite3.connect(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
    scol = "stream_id" if "stream_id" in cols else None
    sel = "name, start, end" + (f", {qcol}" if qcol else ", 0") + (f", {scol}" if scol else ", 0")
    rows = db.execute(f"select {sel} from kernels order by start").fetchall()
    print("# columns of `kernels`:", ", ".join(cols))
    # step boundaries: the packed raster kernel runs once per step
    marks = [i for i, r in enumerate(rows) if "depth_raster_batch_packed" in r[0] or "depth_raster_batch_points" in r[0]
             or "depth_raster_batch_winner" in r[0]]
    if len(marks) < back + 1:
        print("not enough steps in the trace")
        return
    # the kernel in front of the raster (round 4: the inverse-rotation kernel; before: the map fill) belongs to the step too: start
    # one kernel earlier — and, under the default schedule, at the voxelizer's key kernel if that started first on its own queue
    def first_of_step(m):
        lo = m - 1
        for j in range(max(0, m - 4), m):
            if "vox_key_batch" in rows[j][0] or "mat3_inverse" in rows[j][0]:
                lo = min(lo, j)
        return lo
    lo, hi = first_of_step(marks[-back - 1]), first_of_step(marks[-back])
    seg = rows[lo:hi]
    t0 = seg[0][1]
    qs = sorted({r[3] for r in seg})
    ss = sorted({r[4] for r in seg})
    print(f"# {len(seg)} kernels, {(max(r[2] for r in seg) - t0) / 1e3:.1f} us wall, queues {qs}, streams {ss}")
    print(f"# {'start us':>9s} {'dur us':>8s} {'gap':>7s}  q/s  kernel")
    last_end = {}
    busy = 0.0
    for n, s, e, q, st in seg:
        key = (q, st)
        gap = (s - last_end[key]) / 1e3 if key in last_end else 0.0
        last_end[key] = e
        busy += (e - s) / 1e3
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {gap:7.1f}  {qs.index(q)}/{ss.index(st)}  {short(n)}")
    print(f"# sum of kernel durations {busy:.1f} us")


if __name__ == "__main__":
    main()
