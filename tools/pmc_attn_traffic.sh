# HBM-side traffic of the attention launches as the product issues them (tools/attn_launch.py: self-attention <64> bound proven + tail
# split on the fused-QKV layout, cross-attention <128>): FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes, no trace domains
# (MI355X_MICROARCH.md, HBM).
# usage (GPU box): bash tools/pmc_attn_traffic.sh <tag>   -> gpurun_out/<tag>_attn_pmc.json
tag=${1:-r4}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 250 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/${tag}_pmc_attn_$ctr -- python3 $R/tools/attn_launch.py 3 > $R/gpurun_out/${tag}_pmc_attn_$ctr.log 2>&1 || exit 1
done
cd $R
python3 - "$tag" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
tag = sys.argv[1]
per = defaultdict(lambda: defaultdict(list))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/{tag}_pmc_attn_{ctr}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr and ("attn_fwd" in r["Kernel_Name"] or "attn_combine" in r["Kernel_Name"]):
                n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
                per[n[:n.index("(")] if "(" in n else n][ctr].append(float(r["Counter_Value"]))
out = {"collection": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, separate passes, no trace domains, tools/attn_launch.py 3 (self [2,17776,48,64] fused-QKV layout + cross [2,17550|4050,16,128], product flags)",
       "gfx950_correction": "read bytes = 2 x FETCH_SIZE x 1024 (128-B requests tallied at 64 B); WRITE_SIZE exact for 16-B-per-lane stores",
       "algorithmic_bytes_per_launch": 4 * 2 * 17776 * 48 * 64 * 2, "kernels": {}}
for k, v in per.items():
    fk, wk = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]), sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
    out["kernels"][k] = {"dispatches": len(v["FETCH_SIZE"]), "FETCH_SIZE_raw_KB": round(fk), "WRITE_SIZE_raw_KB": round(wk),
                         "read_bytes_corrected": int(2 * fk * 1024), "write_bytes": int(wk * 1024), "traffic_bytes_per_launch": int(2 * fk * 1024 + wk * 1024)}
main = [v for k, v in out["kernels"].items() if k.startswith("attn_fwd_kernel<64")]
comb = [v for k, v in out["kernels"].items() if k.startswith("attn_combine")]
if main:
    out["traffic_bytes_per_launch"] = main[0]["traffic_bytes_per_launch"] + (comb[0]["traffic_bytes_per_launch"] if comb else 0)
json.dump(out, open(f"gpurun_out/{tag}_attn_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
