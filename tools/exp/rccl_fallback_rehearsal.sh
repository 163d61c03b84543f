# One-card rehearsal of bench.py's LOUD RCCL -> gloo fallback: two ranks on the same GPU make RCCL's communicator creation fail
# (duplicate device), which is the failure the fallback is for.  Expect: error text on stderr, a JSON line with
# collective_fallback / collective_backend_warning / config.collective_backend == "gloo".
tag=${1:-r4}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TCX_BENCH_SINGLE_DEVICE=1 timeout -k 10 600 python3 bench.py --gpus 2 --steps 2 --warmup 1 --layers 2 --alt-steps 0 > gpurun_out/${tag}_rccl_fallback.json 2> gpurun_out/${tag}_rccl_fallback.err
echo "rc=$?"
grep -c "ALL RANKS FALL BACK" gpurun_out/${tag}_rccl_fallback.err
python3 - "$tag" <<'PY'
import json
import sys
for line in open(f"gpurun_out/{sys.argv[1]}_rccl_fallback.json"):
    if line.startswith("{"):
        r = json.loads(line)
        print({k: r.get(k) for k in ("n_gpus", "value", "collective_fallback", "collective_backend_warning")}, r["config"]["collective_backend"], r["config"].get("debug_env"), r["config"].get("ranks"))
PY
