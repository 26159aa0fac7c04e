"""tools/pmc.sh output (one 'kernel counter value n=..' line per counter) -> profiles/rNN_pmc_gemm_nt.json, the file bench.py reads
roofline.traffic from.  Records the sha of the kernel sources it was collected on: bench.py withholds the number once they change."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

src, dst, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
c, n = {}, 0
for line in open(src):
    parts = line.split()
    if len(parts) >= 3 and parts[-1].startswith("n="):
        c[parts[-3]] = float(parts[-2]); n = int(parts[-1][2:])
fetch, write = c["FETCH_SIZE"], c["WRITE_SIZE"]
out = {
    "kernel": "gemm_nt_kernel<256,256,2,4,...> (all epilogue instantiations, launches of the bench step)",
    "command": cmd, "launches_averaged": n, "source_sha16": bench.source_sha16("gemm_nt.hip", "common.h"),
    "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
    "gfx950_correction": "FETCH_SIZE counts 128-B requests at 64 B for wide coalesced streams -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE taken as is",
    "hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
    "TCC_HIT_per_launch": c.get("TCC_HIT_sum"), "TCC_MISS_per_launch": c.get("TCC_MISS_sum"),
    "SQ_VALU_MFMA_BUSY_CYCLES_per_launch": c.get("SQ_VALU_MFMA_BUSY_CYCLES"), "GRBM_GUI_ACTIVE_per_launch": c.get("GRBM_GUI_ACTIVE"),
    "mfma_util": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 128.0), 3) if c.get("GRBM_GUI_ACTIVE") else None,
    "SQ_LDS_BANK_CONFLICT_per_launch": c.get("SQ_LDS_BANK_CONFLICT"), "SQ_LDS_IDX_ACTIVE_per_launch": c.get("SQ_LDS_IDX_ACTIVE"),
    "SQ_WAIT_INST_ANY_frac_of_wave_cycles": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3) if c.get("SQ_WAVE_CYCLES") else None,
    "insts_per_mfma": {k: round(c[k2] / c["SQ_INSTS_MFMA"], 2) for k, k2 in (("valu", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU"), ("lds", "SQ_INSTS_LDS"), ("vmem", "SQ_INSTS_VMEM")) if c.get("SQ_INSTS_MFMA")},
}
json.dump(out, open(dst, "w"), indent=2)
print(json.dumps(out))
