"""gpurun_out/parity.jsonl (appended by tests/test_models_gpu.record_parity during `pytest -m gpu` on the GPU box) -> profiles/<tag>_parity.json:
the last record of every comparison, in first-seen order.   usage: python tools/collect_parity.py [tag]"""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if a != "--force"]
tag = args[0] if args else "r4"
recs, order = {}, []
with open(os.path.join(root, "gpurun_out", "parity.jsonl")) as f:
    for line in f:
        if line.strip():
            r = json.loads(line)
            if r["test"] not in recs:
                order.append(r["test"])
            recs[r["test"]] = r
if len(order) < 40 and "--force" not in sys.argv:          # gpurun replaces parity.jsonl on every merge: a partial test run leaves a partial file
    raise SystemExit(f"only {len(order)} records in gpurun_out/parity.jsonl (a full `pytest -m gpu` writes > 60): not overwriting "
                     f"profiles/{tag}_parity.json; pass --force to do it anyway")
out = {"source": "tests -m gpu on MI355X (gpurun), tests/test_models_gpu.record_parity: every _check_deep comparison and the north-star "
                 "tolerance record of tests/test_pipeline_gpu.py (last run of each)", "records": [recs[t] for t in order]}
with open(os.path.join(root, "profiles", f"{tag}_parity.json"), "w") as f:
    json.dump(out, f, indent=1)
print(len(order), "records ->", f"profiles/{tag}_parity.json")
