"""Data-parallel runner: independent trajectories / clips sharded over the GPUs of one node.

The reference runs the 8 orbit variants of one clip sequentially on one GPU
(inference_orbits.py:274-300) or as separate single-GPU SLURM array tasks
(slurm_run_orbits_auto.sh:23-30): the units are mutually independent, so the path shards with NO
data-path collective during denoising.  One process per GPU (torchrun / torch.distributed, backend
"nccl" = RCCL over xGMI); rank r owns trajectories r, r+W, ...; a single all-gather at the end
reassembles the decoded frames (or the latents) on every rank.
"""
from __future__ import annotations

import datetime
import os
import threading
from typing import Callable, List, Optional, Sequence, Tuple

# dmabuf IPC: without it RCCL fails on this pool with `hipIpcGetMemHandle: invalid argument`.  Set before the HSA runtime starts
# (first HIP call), also for ranks started by a plain torchrun line that did not export it.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch                                  # noqa: E402
import torch.distributed as dist              # noqa: E402

# An unreachable peer or a stuck RCCL bootstrap must end the job in about two minutes with an error, not sit out torch's
# 10 / 30-minute defaults (the driver's limit for a bench run is 600 s).  TCX_DIST_TIMEOUT_S overrides.
DEFAULT_TIMEOUT_S = 120.0


def dist_timeout(seconds: Optional[float] = None) -> datetime.timedelta:
    if seconds is None:
        seconds = float(os.environ.get("TCX_DIST_TIMEOUT_S", DEFAULT_TIMEOUT_S))
    return datetime.timedelta(seconds=float(seconds))


def env_rank() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init_distributed(backend: str | None = None, timeout_s: Optional[float] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from the torchrun environment (no-op for a single process).  `timeout_s` (default
    `DEFAULT_TIMEOUT_S` / TCX_DIST_TIMEOUT_S) bounds the rendezvous and every collective of the default group."""
    rank, world, local = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            if os.environ.get("TCX_BENCH_SINGLE_DEVICE") != "1":
                torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())   # binds the RCCL communicator to this rank's GPU
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=dist_timeout(timeout_s), **kw)
    return rank, world, local


class CollectiveHang(RuntimeError):
    pass


def guarded(fn: Callable, timeout_s: float, what: str, device: Optional[int] = None):
    """Run `fn()` (an RCCL call that may block for ever: communicator bootstrap, a collective whose peer never arrives) on a
    daemon thread and wait at most `timeout_s`.  Returns fn's result, re-raises its exception, or raises `CollectiveHang` and
    ABANDONS the thread (it sits in native code with the GIL released; the caller must leave through `os._exit` in the end)."""
    box = {}

    def run():
        try:
            if device is not None:
                torch.cuda.set_device(device)
            box["out"] = fn()
        except BaseException as e:                              # noqa: BLE001 — handed to the waiting thread
            box["err"] = e

    th = threading.Thread(target=run, name=f"tcx-guarded:{what}", daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        raise CollectiveHang(f"{what} did not return within {timeout_s:.0f} s")
    if "err" in box:
        raise box["err"]
    return box.get("out")


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin ownership: trajectory i belongs to rank i % world."""
    return list(range(rank, n_items, world))


def all_gather_cat(local: torch.Tensor, group=None) -> torch.Tensor:
    """Single all-gather of equally shaped per-rank tensors -> concatenation in rank order (`group`: default group if None)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    local = local.contiguous()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local, group=group)
    return out


def run_trajectories(run_one: Callable[[int], torch.Tensor], n_items: int, gather: bool = True) -> torch.Tensor:
    """Run `run_one(i)` (-> tensor [1, ...]) for the trajectories this rank owns and all-gather.

    The result on every rank is `[n_items, ...]` ordered by trajectory index (rank-major gather re-ordered to trajectory order).
    n_items need not be a multiple of the world size: ranks with one trajectory fewer pad their block with zeros for the single
    all-gather and the padding is dropped afterwards (every rank must own at least one trajectory: n_items >= world size)."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    if n_items < world:
        raise ValueError(f"n_items ({n_items}) must be at least the world size ({world}): every rank runs at least one trajectory")
    mine = shard_indices(n_items, rank, world)
    local = torch.cat([run_one(i) for i in mine], dim=0)
    if not gather or world == 1:
        return local
    per = (n_items + world - 1) // world
    if len(mine) < per:                                 # ragged: pad to the common block size
        local = torch.cat([local, local.new_zeros((per - len(mine),) + tuple(local.shape[1:]))], dim=0)
    allr = all_gather_cat(local)                        # [world * per, ...] rank-major
    order = [(i % world) * per + i // world for i in range(n_items)]    # trajectory i = j * world + r sits at row r * per + j
    return allr[torch.tensor(order, device=allr.device)]
