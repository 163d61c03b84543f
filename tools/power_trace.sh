# Board power and clocks while the driver bench runs (evidence for the "cycles saved come back as lower clock" observations of DESIGN
# §3.1): rocm-smi polled every 0.5 s in the background, the bench in the foreground.  usage (GPU box): bash tools/power_trace.sh <tag>
tag=${1:-r4}
R=$GRAFT_REPO_ROOT
cd $R
( for i in $(seq 1 200); do date +%s.%N | tr '\n' ' '; /opt/rocm/bin/rocm-smi --showpower --showclocks --showuse --csv 2>/dev/null | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/${tag}_power_raw.txt 2>&1 &
poll=$!
python3 bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --alt-steps 0 > gpurun_out/${tag}_power_bench.json 2> gpurun_out/${tag}_power_bench.err
kill $poll 2>/dev/null
python3 - "$tag" <<'PY'
import re, sys, json, statistics
tag = sys.argv[1]
rows = []
for l in open(f"gpurun_out/{tag}_power_raw.txt"):
    nums = re.findall(r"[-+]?\d+\.?\d*", l)
    rows.append(l.strip()[:400])
print("\n".join(rows[:3]))
print("...")
print("\n".join(rows[len(rows)//2:len(rows)//2+3]))
PY
