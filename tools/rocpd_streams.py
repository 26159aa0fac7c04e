"""Per-queue view of a rocprofv3 kernel trace of bench.py (shipped multi-stream schedule): for the last two training steps, the busy
time of every HIP queue and the idle gaps of the busiest one (the main stream: the critical path) with the kernels around them."""
import sqlite3, sys, collections

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(rocpd_kernel_dispatch)")]
qcol = "stream_id" if "stream_id" in cols else "queue_id"
rows = db.execute(f"select s.kernel_name, d.start, d.end, d.{qcol} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
marks = [st for name, st, en, q in rows if "transpose_table_k" in name]
lo, hi = marks[-3], marks[-1]
sel = [(n, s, e, q) for n, s, e, q in rows if s >= lo and e <= hi]
print(f"columns: {qcol}; window {(hi - lo) / 1e6:.2f} ms (two steps), {len(sel)} kernels")
byq = collections.defaultdict(list)
for n, s, e, q in sel:
    byq[q].append((s, e, n))
for q, v in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    print(f"queue {q}: {len(v)} kernels, busy {sum(e - s for s, e, _ in v) / 1e6:.2f} ms")
main = max(byq.items(), key=lambda kv: sum(e - s for s, e, _ in kv[1]))[1]
gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(main, main[1:]):
    if s1 > e0:
        gaps.append(((s1 - e0) / 1e3, n0, n1, (e0 - lo) / 1e6))
print(f"main queue: {len(gaps)} gaps totalling {sum(g[0] for g in gaps) / 1e3:.2f} ms")
short = lambda n: n.split("N_1")[-1][:48]
for g in sorted(gaps, key=lambda g: -g[0])[:40]:
    print(f"  {g[0]:8.1f} us at t={g[3]:7.2f} ms  after {short(g[1])}  before {short(g[2])}")
hist = collections.Counter()
for g in gaps:
    hist[min(int(g[0] // 10) * 10, 100)] += g[0]
print("gap time by gap length (us bucket -> total us):", dict(sorted(hist.items())))
# what the other queues run while the main queue sits in its two largest gaps of the first step
big = sorted(gaps, key=lambda g: -g[0])[:4]
for g in sorted(big, key=lambda g: g[3])[:4]:
    t0 = lo + int(g[3] * 1e6); t1 = t0 + int(g[0] * 1e3)
    print(f"--- during the {g[0]:.0f} us gap at t={g[3]:.2f} ms:")
    for n, s, e, q in sel:
        if e > t0 and s < t1 and (s, e, n) not in main:
            print(f"    q{q} {(s - lo) / 1e6:8.3f} +{(e - s) / 1e3:7.1f} us  {short(n)}")
# small-copy / fill kernels of the first of the two steps: when, on which queue, between which kernels
step_end = lo + (hi - lo) // 2
print("--- copy / fill / elementwise kernels in the first step:")
prev = {}
for n, s, e, q in sel:
    if s < step_end and ("copyBuffer" in n or "fillBuffer" in n or "at6native" in n or "at_native" in n):
        print(f"    q{q} t={(s - lo) / 1e6:8.3f} +{(e - s) / 1e3:6.1f} us {n[:70]}   after {short(prev.get(q, ''))}")
    prev[q] = n
# the GPU as a whole: union of the kernel intervals of all queues over the two-step window (time in which NO kernel runs = launch / dependency bubbles)
iv = sorted((s, e) for n, s, e, q in sel)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
idle = []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        idle.append(((s - cur_e) / 1e3, (cur_e - lo) / 1e6))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"--- all queues: some kernel running {busy / 1e6:.2f} ms of the {(hi - lo) / 1e6:.2f} ms window; {len(idle)} idle intervals, {sum(i[0] for i in idle) / 1e3:.2f} ms in total; "
      f"longest: " + ", ".join(f"{i[0]:.0f} us at t={i[1]:.2f}" for i in sorted(idle, key=lambda i: -i[0])[:8]))
