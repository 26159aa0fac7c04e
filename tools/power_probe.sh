# usage (gpurun): bash tools/power_probe.sh   -- package power and shader clock (rocm-smi, 5 Hz) while the bench step runs, idle before / after
cd $GRAFT_REPO_ROOT
rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk" | head -12
( for i in $(seq 1 150); do rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Package Power\|sclk" | tr '\n' ' ' | sed 's/  */ /g'; echo; sleep 0.2; done ) > gpurun_out/power_samples.txt &
SM=$!
sleep 2
timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-kernel-timing --train-only 2>/dev/null | tail -1 | cut -c1-120
wait $SM
python - <<'PY'
import re
pw, ck = [], []
for l in open("gpurun_out/power_samples.txt"):
    m = re.search(r"Package Power \(W\): ([0-9.]+)", l); c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", l)
    if m: pw.append(float(m.group(1)))
    if c: ck.append(int(c.group(1)))
print("samples", len(pw), "power W: min", min(pw) if pw else None, "max", max(pw) if pw else None, "p50", sorted(pw)[len(pw)//2] if pw else None)
print("sclk MHz: min", min(ck) if ck else None, "max", max(ck) if ck else None, "p50", sorted(ck)[len(ck)//2] if ck else None)
print(open("gpurun_out/power_samples.txt").read()[:1500])
PY
