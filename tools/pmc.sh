#!/bin/bash
# usage: tools/pmc.sh <kernel-name-substring> <out.txt> -- <command...>     (one rocprofv3 --pmc pass per counter group)
pat=$1; out=$2; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > $out
for pm in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf gpurun_out/_pmc; rocprofv3 --pmc $pm --output-format csv -d gpurun_out/_pmc -o p -- "$@" > /dev/null 2>&1
  python - "$pat" >> $out <<PY
import csv, glob, collections, sys
f = glob.glob("gpurun_out/_pmc/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f[0])):
    if sys.argv[1] in row["Kernel_Name"]:
        agg[sys.argv[1] + " (all matching launches)"][row["Counter_Name"]].append(float(row["Counter_Value"]))
for kn, d in agg.items():
    for k, v in d.items():
        print(f"{kn} {k} {sum(v)/len(v):.0f} n={len(v)}")
PY
done
rm -rf gpurun_out/_pmc
cat $out
