# usage (gpurun): bash tools/clock_probe.sh   -- shader clock under the GEMM K loop, shipped operands vs aliased (NT_HOT) operands:
# GRBM_GUI_ACTIVE cycles / kernel duration per gemm_nt launch of tools/nt_harness (build it first: tools/nt_hot.sh check 1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for h in 0 2; do
  rm -rf gpurun_out/_clk
  NT_HOT=$h rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/_clk -o p -- tools/nt_harness time 2 0 0 > /dev/null 2>&1
  python - $h <<'PY'
import csv, glob, sys, collections
cc = glob.glob("gpurun_out/_clk/**/*counter_collection.csv", recursive=True)
kt = glob.glob("gpurun_out/_clk/**/*kernel_trace.csv", recursive=True)
if not cc or not kt:
    print("missing csv", cc, kt); sys.exit(0)
dur = {}
for r in csv.DictReader(open(kt[0])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
rows = collections.defaultdict(list)
for r in csv.DictReader(open(cc[0])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    d = dur.get(r["Dispatch_Id"])
    if not d or "gemm_nt_kernel<256" not in d[1] and "gemm_nt_kernelILi256" not in d[1]: continue
    if d[0] < 100000: continue                      # launches >= 0.1 ms
    rows[d[1][:70]].append(float(r["Counter_Value"]) / d[0])     # cycles per ns = GHz
allv = [v for vs in rows.values() for v in vs]
print(f"NT_HOT={sys.argv[1]}: {len(allv)} launches, GRBM_GUI_ACTIVE / duration: mean {sum(allv)/max(len(allv),1):.3f} GHz, min {min(allv):.3f}, max {max(allv):.3f}")
PY
done
rm -rf gpurun_out/_clk
