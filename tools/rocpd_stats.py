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
    return "\n".join(out), tot, (rows[-1][2] - rows[0][1]) / 1e6


if __name__ == "__main__":
    txt, tot, span = stats(sys.argv[1])
    print(txt)
    print(f"# total kernel time {tot:.1f} ms over a {span:.1f} ms span")
