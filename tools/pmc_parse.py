"""Summarise tools/pmc_traffic.sh output: per kernel name, mean FETCH_SIZE / WRITE_SIZE per dispatch and the corrected
HBM-side bytes (gfx950: FETCH_SIZE counts 128-B requests at 64 B -> read bytes = 2 * FETCH_SIZE KB; MI355X_MICROARCH.md).
usage: python tools/pmc_parse.py gpurun_out [tag ...]"""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
tags = sys.argv[2:] or ["attn", "gemm"]
out = {}
for tag in tags:
    per = defaultdict(lambda: defaultdict(list))
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(root, f"pmc_{tag}_{ctr}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == ctr:
                    per[r["Kernel_Name"]][ctr].append(float(r["Counter_Value"]))
    for k, v in per.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v and ("attn_fwd" in k or "gemm_kernel" in k or "Cijk" in k):
            fk, wk = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]), sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
            out[f"{tag}: {k[:110]}"] = {"dispatches": len(v["FETCH_SIZE"]), "FETCH_SIZE_raw_KB": round(fk), "WRITE_SIZE_raw_KB": round(wk),
                                        "read_bytes_corrected": int(2 * fk * 1024), "write_bytes": int(wk * 1024),
                                        "traffic_bytes_per_launch": int(2 * fk * 1024 + wk * 1024)}
print(json.dumps(out, indent=1))
