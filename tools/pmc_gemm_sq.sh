# SQ counters of the four big GEMM launches of a transformer block (tools/gemm_launch.py), separate --pmc passes, no trace domains.
# usage (GPU box): bash tools/pmc_gemm_sq.sh <tag>  -> gpurun_out/<tag>_gemm_sq.json
tag=${1:-r4}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 250 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/${tag}_pmc_gemm_$i -- python3 $R/tools/gemm_launch.py 3 > $R/gpurun_out/${tag}_pmc_gemm_$i.log 2>&1 || { tail -5 $R/gpurun_out/${tag}_pmc_gemm_$i.log; exit 1; }
done
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_pmc_gemm_trace -- python3 $R/tools/gemm_launch.py 3 > $R/gpurun_out/${tag}_pmc_gemm_trace.log 2>&1 || exit 1
cd $R
python3 - "$tag" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
tag = sys.argv[1]
# dispatch order within an iteration: qkv (epi 0), out+gate (epi 2, K 3072), ff1+gelu (epi 1), ff2+gate (epi 2, K 12288)
names = ["qkv 3072->9216 bias", "out 3072->3072 gated residual", "ff1 3072->12288 bias+GELU", "ff2 12288->3072 gated residual"]
per = defaultdict(lambda: defaultdict(list))
for f in glob.glob(f"gpurun_out/{tag}_pmc_gemm_[0-9]/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "gemm_kernel<" in r["Kernel_Name"]]
    by_ctr = defaultdict(list)
    for r in rows:
        by_ctr[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, v in by_ctr.items():
        v.sort()
        for j, (_, val) in enumerate(v):
            per[names[j % 4]][c].append(val)
dur = defaultdict(list)
for f in glob.glob(f"gpurun_out/{tag}_pmc_gemm_trace/**/*kernel_trace.csv", recursive=True):
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)) if "gemm_kernel<" in r["Kernel_Name"]))
    for j, (a, b) in enumerate(rows):
        dur[names[j % 4]].append((b - a) / 1e3)
out = {"collection": "rocprofv3 --pmc, two SQ groups in separate passes, no trace domains; tools/gemm_launch.py 3 (M = 35552 rows, product epilogues, random data); "
                     "per-dispatch means; durations from a separate --kernel-trace pass; units as in r4_attn_sq.json", "kernels": {}}
flops = {names[0]: 2.0 * 35552 * 9216 * 3072, names[1]: 2.0 * 35552 * 3072 * 3072, names[2]: 2.0 * 35552 * 12288 * 3072, names[3]: 2.0 * 35552 * 3072 * 12288}
for key in names:
    c = per[key]
    m = {k: sum(v) / len(v) for k, v in c.items()}
    d = sorted(dur.get(key, [0.0]))[len(dur.get(key, [0.0])) // 2]
    rec = {"median_duration_us": d, "TFLOPs": round(flops[key] / (d * 1e-6) / 1e12, 1) if d else None, **{k: round(v) for k, v in sorted(m.items())}}
    if d and m.get("GRBM_GUI_ACTIVE"):
        rec["effective_clock_GHz"] = round(m["GRBM_GUI_ACTIVE"] / 8 / (d * 1e-6) / 1e9, 3)
        simd = 256 * 4 * (m["GRBM_GUI_ACTIVE"] / 8)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            rec["mfma_pipe_busy_frac_of_all_simd_cycles"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd, 4)
        if m.get("SQ_INSTS_MFMA"):
            rec["mfma_pipe_busy_frac_from_inst_count_x16"] = round(m["SQ_INSTS_MFMA"] * 16 / simd, 4)
    if m.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in m:
                rec[k + "_frac_of_wave_cycles"] = round(m[k] / m["SQ_WAVE_CYCLES"], 4)
    out["kernels"][key] = rec
json.dump(out, open(f"gpurun_out/{tag}_gemm_sq.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
