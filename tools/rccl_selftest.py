"""RCCL smoke test on whatever GPUs the launcher gives it (world 1 on the one-GPU test box; `torchrun --nproc-per-node N` on a node):
the SAME group structure bench.py uses for N > 1 — a gloo control group (barrier, MIN / MAX reductions of host tensors) plus an RCCL
("nccl") data group probed with a one-element all-reduce and then used for the clip-sized all-gather of bf16 channels-last frames."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
dist.init_process_group("gloo", rank=rank, world_size=world)
g = dist.new_group(backend="nccl")
one = torch.ones(1, device="cuda")
dist.all_reduce(one, group=g)
torch.cuda.synchronize()
assert float(one.item()) == world
ok = torch.tensor([1], dtype=torch.int32)
dist.all_reduce(ok, op=dist.ReduceOp.MIN)
x = torch.full((1, 49, 480, 720, 3), 1.5 + rank, device="cuda", dtype=torch.bfloat16)
out = torch.empty((world,) + tuple(x.shape[1:]), device="cuda", dtype=torch.bfloat16)
dist.all_gather_into_tensor(out, x, group=g)
t = torch.tensor([1.25 + rank], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert int(ok.item()) == 1 and float(t) == 1.25 + world - 1 and all(float(out[r].float().mean()) == 1.5 + r for r in range(world))
if rank == 0:
    print(f"rccl world={world} ok: gloo control group + nccl data group, gathered {tuple(out.shape)}",
          torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else None)
dist.destroy_process_group()
