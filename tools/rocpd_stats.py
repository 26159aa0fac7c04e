"""Per-kernel statistics (count, total/avg/min/max ms, % of GPU time) from a rocprofv3 rocpd SQLite database."""
import sqlite3
import sys


def stats(path, skip_first_fraction=0.0):
    db = sqlite3.connect(path)
    rows = db.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    agg = {}
    for name, st, en in rows:
        a = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
        dur = (en - st) / 1e6
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values())
    out = ["name,calls,total_ms,avg_ms,min_ms,max_ms,percent"]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) < 110 else name[:107] + "..."
        out.append(f"\"{short}\",{a[0]},{a[1]:.3f},{a[1]/a[0]:.4f},{a[2]:.4f},{a[3]:.4f},{100*a[1]/tot:.2f}")
    # family totals: template instantiations of one kernel (e.g. the epilogue variants of gemm_nt_kernel<256,256,2,4,...>)
    import re
    fam = {}
    for name, a in agg.items():
        m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
        if not m:
            continue
        n = int(m.group(1))
        base = name[m.end():m.end() + n]
        t = re.match(r"ILi(\d+)ELi(\d+)E", name[m.end() + n:])
        key = base + (f"<{t.group(1)},{t.group(2)},...>" if t else "")
        f = fam.setdefault(key, [0, 0.0, 0])
        f[0] += a[0]; f[1] += a[1]; f[2] += 1
    out.append("# family totals (all template instantiations): name,calls,total_ms,avg_ms,percent,instantiations")
    for key, f in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        if f[2] > 1:
            out.append(f"# \"{key}\",{f[0]},{f[1]:.3f},{f[1]/f[0]:.4f},{100*f[1]/tot:.2f},{f[2]}")
    # GPU occupancy of the training steps: between consecutive full-table AdamW / transpose_table launches (one per step) --
    # union of the kernel intervals vs wall time, and the largest idle gaps with the kernels around them
    marks = [st for name, st, en in rows if "transpose_table_k" in name]
    if len(marks) >= 3:
        lo, hi = marks[-3], marks[-1]                                 # the last two steps of the trace
        iv = [(st, en, name) for name, st, en in rows if st >= lo and en <= hi]
        busy, cur_s, cur_e, gaps, last_name = 0, None, None, [], ""
        for st, en, name in iv:
            if cur_s is None:
                cur_s, cur_e = st, en
            elif st <= cur_e:
                cur_e = max(cur_e, en)
            else:
                busy += cur_e - cur_s
                gaps.append(((st - cur_e) / 1e3, last_name, name))
                cur_s, cur_e = st, en
            last_name = name
        busy += cur_e - cur_s
        out.append(f"# last two steps: wall {(hi - lo) / 1e6:.2f} ms, GPU busy (union of kernel intervals) {busy / 1e6:.2f} ms, "
                   f"{len(gaps)} idle gaps totalling {sum(g[0] for g in gaps) / 1e3:.2f} ms")
        for g in sorted(gaps, key=lambda g: -g[0])[:8]:
            out.append(f"#   gap {g[0]:.1f} us between {g[1][:60]} and {g[2][:60]}")
    return "\n".join(out), tot, (rows[-1][2] - rows[0][1]) / 1e6


if __name__ == "__main__":
    txt, tot, span = stats(sys.argv[1])
    print(txt)
    print(f"# total kernel time {tot:.1f} ms over a {span:.1f} ms span")
