"""BASELINE configs[3] as a PRODUCT path on one card: `driver.run_orbits` = the trajectory variants of one clip
(reference inference_orbits.py:248-300), data-parallel over the ranks of a torch.distributed group with ONE all-gather.

Two fresh rank processes share the test box's single GPU (gloo rendezvous; RCCL needs one GPU per rank and is exercised by the
driver's 8-GPU run) and run the whole per-variant chain — poses, point-cloud render with `mask=True`, both VAE encodes, the
CFG/DDIM loop, decode — then the gather.  Checked: every rank ends with ALL variants in variant order, bit-identical to the
sequential single-process loop (the reference's form) run in this process; each rank computed only its own variants
(`gather=False` output = variants r, r + W, ...); variants differ from each other."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "orbit_rank_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(world, out_dir, n_var):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world),
               TCX_BENCH_SINGLE_DEVICE="1", TCX_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, WORKER, str(out_dir), str(n_var)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(out)
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"


def test_two_ranks_of_the_pipeline_on_one_card(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from safetensors.torch import load_file
    n_var, world = 4, 2
    # the sequential loop (reference form), in a fresh single process as well: same code path as a rank, no process group
    seq_dir = tmp_path / "seq"
    seq_dir.mkdir()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, WORKER, str(seq_dir), str(n_var)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    seq = load_file(str(seq_dir / "rank0.safetensors"))
    want = seq["gathered"]
    assert want.shape == (n_var, 3, 9, 32, 48) and want.dtype == torch.float32
    assert torch.equal(seq["mine"], want)                                            # world 1: gather is the identity
    assert float(want.min()) >= 0 and float(want.max()) <= 1 and torch.isfinite(want).all()
    for i in range(n_var):
        for j in range(i + 1, n_var):
            assert not torch.equal(want[i], want[j]), f"variants {i} and {j} rendered the same clip"

    dp_dir = tmp_path / "dp"
    dp_dir.mkdir()
    _run_ranks(world, dp_dir, n_var)
    for rank in range(world):
        got = load_file(str(dp_dir / f"rank{rank}.safetensors"))
        assert torch.equal(got["gathered"], want), f"rank {rank}: gathered clips differ from the sequential loop"
        assert torch.equal(got["mine"], want[rank::world]), f"rank {rank} did not compute exactly variants {list(range(rank, n_var, world))}"
