"""The kernels around the loss of the LAST traced training step, in start order, all queues: from `before_us` before lsce_fwd_k to `after_us` after it
(rocprofv3 --kernel-trace database).  Durations are reliable under the profiler, gaps are not (per-launch host cost).
usage: python tools/rocpd_tail.py <db> [before_us] [after_us]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
before, after = (float(sys.argv[2]) if len(sys.argv) > 2 else 700.0), (float(sys.argv[3]) if len(sys.argv) > 3 else 1500.0)
cols = [r[1] for r in db.execute("pragma table_info(rocpd_kernel_dispatch)")]
qcol = "stream_id" if "stream_id" in cols else "queue_id"
rows = db.execute(f"select s.kernel_name, d.start, d.end, d.{qcol} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
anchors = [st for name, st, en, q in rows if "lsce_fwd_k" in name]
a = anchors[-1]
sel = [(n, s, e, q) for n, s, e, q in rows if s >= a - before * 1e3 and s <= a + after * 1e3]
short = lambda n: (n.split("N_1")[-1] if "at6native" not in n else "ATen:" + n.split("at6native")[-1])[:64]
tot = {}
print(f"t = 0 at lsce_fwd_k; {len(sel)} kernels")
for n, s, e, q in sel:
    print(f"  q{q} {(s - a) / 1e3:9.1f} us +{(e - s) / 1e3:7.1f}  {short(n)}")
    tot[q] = tot.get(q, 0) + (e - s)
print("kernel time per queue in the window [us]:", {q: round(v / 1e3, 1) for q, v in tot.items()})
