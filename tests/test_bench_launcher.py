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


def test_gpus_8_starts_eight_ranks():
    # the driver's largest launch (one rank per GPU of the node), rehearsed on gloo
    r = _run("--gpus", "8", "--steps", "2", "--warmup", "1", "--selftest-dist")
    assert r.returncode == 0, r.stderr[-2000:]
    recs = _json_lines(r.stdout)
    assert len(recs) == 1 and recs[0]["n_gpus"] == 8 and recs[0]["gathered"] == [8, 3, 2, 4, 4], r.stdout
    assert recs[0]["collective_backend"] == "nccl" and "collective_fallback" not in recs[0]


def test_plain_torchrun_rank_gets_the_dmabuf_ipc_variable():
    """HSA_ENABLE_IPC_MODE_LEGACY=0 is required for RCCL on this pool.  A rank started by a PLAIN torchrun line (the driver's own
    N > 1 launch, not bench.py's launcher) and an environment that lacks the variable must still end up with it set — before
    torch / HSA come up (bench.main sets it first thing)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2", "--selftest-dist"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    recs = _json_lines(r.stdout)
    assert len(recs) == 1 and recs[0]["n_gpus"] == 2 and recs[0]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0", r.stdout
    # the product runner's package import sets it too (python -m trajectorycrafter_amd.run ... under torchrun)
    r = subprocess.run([sys.executable, "-c", "import trajectorycrafter_amd, os; print(os.environ['HSA_ENABLE_IPC_MODE_LEGACY'])"],
                       capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode == 0 and r.stdout.strip() == "0", r.stderr[-1000:]


def test_unreachable_peer_ends_with_an_error_line_not_a_hang():
    """One rank of a 2-rank job whose peer never starts: the rendezvous is bounded (dp.DEFAULT_TIMEOUT_S = 120 s, here 6 s via
    TCX_DIST_TIMEOUT_S) and the rank exits 3 with an error line instead of waiting for the driver's 600 s kill."""
    import socket
    import time
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    t0 = time.time()
    r = _run("--gpus", "2", "--selftest-dist", env_extra={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                                           "MASTER_PORT": str(port), "TCX_DIST_TIMEOUT_S": "6"})
    assert r.returncode == 3, (r.returncode, r.stderr[-1500:])
    assert "init_process_group('gloo') FAILED" in r.stderr and not _json_lines(r.stdout)
    assert time.time() - t0 < 90
    sys.path.insert(0, ROOT)
    from trajectorycrafter_amd import dp
    assert dp.dist_timeout().total_seconds() == 120.0 and dp.dist_timeout(7).total_seconds() == 7.0


def test_rccl_error_and_hang_fall_back_loudly_on_all_ranks():
    """The data-plane rules (bench.DataPlane) with a stand-in for the RCCL call: an exception, a call that never returns on every
    rank, and one that never returns on ONE rank only — each ends with all ranks on the gloo gather, the reason in the JSON line
    (`collective_fallback`), exit code 0, within the guard time; TCX_BENCH_RCCL_FATAL=1 turns it into exit code 3."""
    for sim, needle in (("error", "simulated RCCL error"), ("hang", "CollectiveHang"), ("hang:1", "failed on another rank")):
        r = _run("--gpus", "2", "--selftest-dist", env_extra={"TCX_SELFTEST_RCCL": sim, "TCX_RCCL_GUARD_S": "3"})
        assert r.returncode == 0, (sim, r.stderr[-2000:])
        recs = _json_lines(r.stdout)
        assert len(recs) == 1 and recs[0]["collective_backend"] == "gloo" and recs[0]["gathered"] == [2, 3, 2, 4, 4], (sim, r.stdout)
        assert needle in recs[0]["collective_fallback"], (sim, recs[0])
        assert "ALL RANKS FALL BACK to gloo" in r.stderr
    r = _run("--gpus", "2", "--selftest-dist", env_extra={"TCX_SELFTEST_RCCL": "error", "TCX_BENCH_RCCL_FATAL": "1"})
    assert r.returncode != 0 and not _json_lines(r.stdout)
