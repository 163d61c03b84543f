"""CPU tests of bench.py's N-rank plumbing (no GPU): `python bench.py --gpus N` must start N ranks itself, as fresh child
processes, and rank 0 must print ONE JSON line with n_gpus = N.  `--selftest-dist` swaps the GPU work for a gloo rehearsal of
the same control flow (rendezvous on 127.0.0.1, barrier, max-reduce of the elapsed time, the single all-gather)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*argv, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH, *argv], capture_output=True, text=True, timeout=300, env=env)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_launch_command_is_the_torchrun_line():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    args = bench.parse(["--gpus", "4"])
    assert (args.gpus, args.steps > 0, args.warmup >= 0) == (4, True, True)


def test_gpus_2_starts_two_ranks_and_prints_one_line():
    r = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--selftest-dist")
    assert r.returncode == 0, r.stderr[-2000:]
    recs = _json_lines(r.stdout)
    assert len(recs) == 1, r.stdout
    assert recs[0]["n_gpus"] == 2 and recs[0]["steps"] == 3 and recs[0]["warmup"] == 1
    assert recs[0]["gathered"] == [2, 3, 2, 4, 4]
    assert "torch.distributed.run" in r.stderr            # the parent launched the ranks, it did not run as a rank itself


def test_under_torchrun_env_the_process_is_a_rank_not_a_launcher():
    # the driver's own N > 1 launch: RANK / WORLD_SIZE already set -> no second launcher level
    r = _run("--gpus", "1", "--selftest-dist", env_extra={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode == 0, r.stderr[-2000:]
    recs = _json_lines(r.stdout)
    assert len(recs) == 1 and recs[0]["n_gpus"] == 1
    assert "launching" not in r.stderr


def test_single_gpu_default_does_not_launch():
    r = _run("--selftest-dist")
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_lines(r.stdout)[0]["n_gpus"] == 1 and "launching" not in r.stderr
