# SQ counter groups for the SHIPPED attention launches at HEAD: separate rocprofv3 --pmc passes (8 SQ slots per
# pass, no trace domains), tools/attn_launch.py = the product's self-attention <64> (bound proven, tail split, fused-QKV layout) and
# cross-attention <128> launches.  usage (GPU box): bash tools/pmc_attn_sq.sh <tag>  -> gpurun_out/<tag>_attn_sq.json
tag=${1:-r4}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -k 10 250 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/${tag}_pmc_sq_$i -- python3 $R/tools/attn_launch.py 4 > $R/gpurun_out/${tag}_pmc_sq_$i.log 2>&1 || { tail -5 $R/gpurun_out/${tag}_pmc_sq_$i.log; exit 1; }
done
# wall time of the same launches (kernel trace, its own pass: never mixed with --pmc)
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_pmc_sq_trace -- python3 $R/tools/attn_launch.py 4 > $R/gpurun_out/${tag}_pmc_sq_trace.log 2>&1 || exit 1
cd $R
python3 - "$tag" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
tag = sys.argv[1]
def kname(n):
    """attn_fwd_kernel<D, OUT_F32, FAST, NW, BOUND>: the bound-centred launch and (cross-attention: bound only TESTED per workgroup) the
    launch of the exact kernel on the complement — empty on this data — are different kernels and are kept apart"""
    a = n[n.index("attn_fwd_kernel<") + 16:n.index(">")].replace(" ", "").split(",")
    return ("self" if a[0] == "64" else "cross") + f"<{a[0]}> " + ("bound-centred loop" if a[4] == "true" else "exact kernel on the (empty) complement")
per = defaultdict(lambda: defaultdict(list))
for f in glob.glob(f"gpurun_out/{tag}_pmc_sq_[0-9]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "attn_fwd_kernel" in n:
            per[kname(n)][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = defaultdict(list)
for f in glob.glob(f"gpurun_out/{tag}_pmc_sq_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "attn_fwd_kernel" in n:
            dur[kname(n)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {"collection": "rocprofv3 --pmc, three SQ groups in separate passes, no trace domains; tools/attn_launch.py 4 (product launches, random data); "
                     "per-dispatch means.  Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; "
                     "SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per 32x32x16 bf16 MFMA) summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs; "
                     "durations from a separate --kernel-trace pass", "kernels": {}}
for key, c in per.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    d = sorted(dur.get(key, [0.0]))[len(dur.get(key, [0.0])) // 2]
    rec = {"dispatches_per_counter": len(next(iter(c.values()))), "median_duration_us": d, **{k: round(v) for k, v in sorted(m.items())}}
    if d and m.get("GRBM_GUI_ACTIVE"):
        clk = m["GRBM_GUI_ACTIVE"] / 8 / (d * 1e-6)                       # cycles per second
        rec["effective_clock_GHz"] = round(clk / 1e9, 3)
        simd_cycles = 256 * 4 * (m["GRBM_GUI_ACTIVE"] / 8)               # SIMD-cycles available during the dispatch
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            rec["mfma_pipe_busy_frac_of_all_simd_cycles"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, 4)
        if m.get("SQ_INSTS_MFMA"):
            rec["mfma_busy_cycles_per_mfma_inst"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / m["SQ_INSTS_MFMA"], 2)
            rec["mfma_pipe_busy_frac_from_inst_count_x32"] = round(m["SQ_INSTS_MFMA"] * 32 / simd_cycles, 4)
    if m.get("SQ_WAVE_CYCLES"):
        w = m["SQ_WAVE_CYCLES"]
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if k in m:
                rec[k + "_frac_of_wave_cycles"] = round(m[k] / w, 4)
    if m.get("SQ_INSTS_MFMA"):
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD"):
            if k in m:
                rec[k + "_per_mfma"] = round(m[k] / m["SQ_INSTS_MFMA"], 3)
    out["kernels"][key] = rec
json.dump(out, open(f"gpurun_out/{tag}_attn_sq.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
