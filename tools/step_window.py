"""Kernels of one whole benchmark step from a rocprofv3 kernel trace: the LAST window between two depth-raster launches that is
between lo and hi milliseconds long (default 3-7).  python tools/step_window.py <trace dir> [lo_ms hi_ms]"""
import glob, os, sqlite3, sys
from graph_timeline import short
d = sys.argv[1]
lo_ms, hi_ms = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (3.0, 7.0)
db = sqlite3.connect(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
q = "queue_id" if "queue_id" in cols else "0"
rows = db.execute(f"select name, start, end, {q} from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "depth_raster_batch_points" in r[0]]
wins = [(a, b) for a, b in zip(marks, marks[1:]) if lo_ms * 1e6 <= rows[b][1] - rows[a][1] <= hi_ms * 1e6]
print(f"# {len(marks)} raster launches, {len(wins)} windows of {lo_ms}-{hi_ms} ms")
print("# window lengths ms:", [round((rows[b][1] - rows[a][1]) / 1e6, 2) for a, b in zip(marks, marks[1:])])
a, b = wins[-3] if len(wins) >= 3 else wins[-1]
t0 = rows[a][1]
qs = sorted({r[3] for r in rows[a:b]})
for n, s, e, qq in rows[a:b]:
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  q{qs.index(qq)}  {short(n)}")
print(f"# window {(rows[b][1] - t0) / 1e3:.1f} us")
