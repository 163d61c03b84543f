#!/usr/bin/env python
"""Headline benchmark: denoised video-latents/sec at 49 frames, 480x720 (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload = BASELINE configs[2] ("Full 50-step DDIM, 49-frame 480x720, random-init weights, 1 MI355X + MAGViT VAE
decode") on every GPU.  One **step** = one classifier-free-guidance denoising step of that clip at full size — the
reference's loop body, models/pipeline_trajectorycrafter.py:1093-1178: a B=2 forward of the 42-block / 6.1 B-parameter
CrossTransformer3DModel (S = 17 776 tokens) + the fused CFG / DDIM update — executed by the product entry point
`TrajCrafter_Pipeline.denoise_step`; nothing is skipped or cached.  After W untimed warm-up steps exactly K steps are
timed between barrier + synchronize pairs (max over ranks).  The steps are consecutive steps of ONE real trajectory
(timesteps 999, 979, ...), so the data the kernels see is what a clip produces.  Then ONE full VAE decode of the
latents (13 latent frames -> 49 frames 480x720) is timed the same way, and for N > 1 the single RCCL all-gather of the path
(the decoder's bf16 output, 102 MB per rank).  A clip is 50 steps + 1 decode (+ 1 gather), hence

    value = n_gpus / (50 * step_s + decode_s + gather_s)          [video-latents/s, whole job]

with every term measured in this run and printed in `config` (a whole 50-step clip takes ~31 s, which is why a step is
not a clip: the driver runs --steps 20 --warmup 5 under a 600 s limit).  Inputs (latents, prompt embeddings, inpaint /
reference latents) are synthetic and resident in HBM before the timed region.

N > 1: one process per GPU.  Started by torchrun (RANK / WORLD_SIZE in the environment) each process is one rank; started
plainly as `python bench.py --gpus N`, this process launches `python -m torch.distributed.run --nproc-per-node N bench.py
...` itself BEFORE touching the GPU and relays its output.  Every rank denoises its own independent trajectory (weak
scaling, seeds 43 + rank as in inference_orbits.py:274-300); there is no collective inside the denoising.

Rank 0 prints ONE JSON line (driver contract) with
  "roofline"     - the dominant kernel (self-attention, tcx_attn_fwd D=64): algorithmic FLOP per launch / average launch
                   duration measured live with HIP events on the launch stream over the timed steps;
  "cpu_baseline" - the oracle (CPU port of the reference) timed on the host cores on a bounded sample (<= 30 s).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 MFMA (MI355X_MICROARCH.md; 2:1-sparsity figures never used)
CPU_BASELINE_BUDGET_S = 90.0       # hard limit for the child (its sample is sized for ~20 s on 16 cores)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="timed CFG denoising steps per rank")
    ap.add_argument("--warmup", type=int, default=2, help="untimed warm-up steps per rank")
    ap.add_argument("--denoise-steps", type=int, default=50, help="steps of a clip (the clip-time formula and the DDIM schedule)")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=720)
    ap.add_argument("--frames", type=int, default=49)
    ap.add_argument("--layers", type=int, default=42, help="(debug) fewer layers => NOT the benchmark config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="(debug) skip the VAE decode => NOT the benchmark config")
    ap.add_argument("--alt-steps", type=int, default=5,
                    help="after the timed steps: this many steps on each of the two other softmax paths the attention can take with "
                         "other weights (bound unproven / exact running max everywhere); 0 = skip.  Never part of `value`.")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="(child of the bench) time the oracle sample on the CPU, print its JSON")
    ap.add_argument("--selftest-dist", action="store_true",
                    help="(CPU test hook) run only the launcher + rank plumbing on gloo, no GPU, no model")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------- launcher (N > 1, no GPU use)
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_command(n: int, argv, port: int):
    """The torchrun line this process starts for `python bench.py --gpus N ...` (the one the driver itself uses for N > 1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__), *argv]


def launch_ranks(args, argv) -> int:
    """Start N fresh rank processes (children; this parent never initialises the GPU) and exit with their code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: required for RCCL on this pool
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = launch_command(args.gpus, argv, _free_port())
    print("[bench] launching", " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


# ------------------------------------------------------------------------------------------- model + inputs
def build_models(args, device):
    import torch
    from trajectorycrafter_amd import init_weights as iw
    from trajectorycrafter_amd.models.autoencoder_magvit import AutoencoderKLCogVideoX
    from trajectorycrafter_amd.models.crosstransformer3d import CrossTransformer3DModel
    from trajectorycrafter_amd.models.pipeline_trajectorycrafter import TrajCrafter_Pipeline
    from trajectorycrafter_amd.scheduler import DDIMScheduler

    BF16 = torch.bfloat16
    tcfg = dict(iw.TRANSFORMER_5B, num_layers=args.layers)
    with torch.device("meta"):
        tr = CrossTransformer3DModel(**tcfg)
        vae = AutoencoderKLCogVideoX()
    tr.load_state_dict(iw.random_state_dict(iw.transformer_param_shapes(dict(tr.config)), seed=0, dtype=BF16, device=device),
                       strict=True, assign=True)
    vsd = iw.random_state_dict(iw.vae_param_shapes(dict(vae.config), decoder=True, encoder=True), seed=1, dtype=BF16,
                               device=device)
    vae.load_state_dict(vsd, strict=True, assign=True)
    tr.eval(), vae.eval()
    return TrajCrafter_Pipeline(None, None, vae, tr, DDIMScheduler())


def make_inputs(args, device, seed):
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    T = (args.frames - 1) // 4 + 1
    h, w = args.height // 8, args.width // 8
    rn = lambda *s: torch.randn(*s, device=device, dtype=torch.bfloat16, generator=g)
    return dict(latents=rn(1, T, 16, h, w), prompt_embeds=rn(1, 226, 4096), negative_prompt_embeds=rn(1, 226, 4096),
                inpaint_latents=rn(2, T, 17, h, w), ref_latents=rn(2, 3, 16, h, w))


# ------------------------------------------------------------------------------------------- CPU baseline (oracle)
def _median_time(fn, reps: int):
    ts = []
    out = None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2], out


def cpu_baseline(args, threads: int = 16, reps: int = 3, row_tokens: int = 2048, vae_crop=(15, 23)):
    """Oracle (CPU restatement of the reference, fp32) on the host cores, bounded to ~20 s, every piece `reps` times (median):
      (a) the row-wise / GEMM part of ONE CogVideoXBlock (2 x LayerNormZero, qkv + out projections, q/k LayerNorm + RoPE, FFN,
          gated residuals) on `row_tokens` of the S = 17 776 tokens — every one of these ops is linear in the token count;
      (b) its joint attention on ONE of the 48 heads at the FULL token count (quadratic in S, so never subsampled; heads are
          independent and scale linearly);
      (c) / (d) the same split for ONE PerceiverCrossAttention (1 of 16 heads);
      (e) one 2-latent-frame chunk of the VAE decoder at default widths on a `vae_crop` latent window of the 60 x 90 grid (every
          decoder op is linear in the pixel count), once.
    A clip is extrapolated as 2 * steps forwards x (42 blocks + 21 cross layers) + 6.5 decode chunks.  Reported baseline, not
    the target; the reference's own CPU path cannot run here (no `diffusers`, no weights)."""
    import torch
    from oracle import diffusers_restated as dr
    from oracle import transformer as otr
    from oracle import vae as ovae
    from oracle.pipeline import prepare_rotary
    from oracle.prec import Prec
    from trajectorycrafter_amd import init_weights as iw

    threads = max(1, min(threads, os.cpu_count() or 1))        # the box's CPU share for one GPU is 16 cores
    torch.set_num_threads(threads)
    cfg = dict(otr.DEFAULT_CONFIG, **dict(iw.TRANSFORMER_5B, num_layers=2))
    shapes = {k: v for k, v in iw.transformer_param_shapes(cfg).items()
              if k.startswith(("transformer_blocks.0.", "perceiver_cross_attention.0."))}
    sd = iw.random_state_dict(shapes, seed=0)
    T = (args.frames - 1) // 4 + 1
    gh, gw = args.height // 16, args.width // 16
    Sv, S, Sr, D = T * gh * gw, T * gh * gw + 226, 3 * gh * gw, 3072
    Ms = min(row_tokens, Sv)                                   # video tokens in the row sample (+ the 226 text tokens)
    g = torch.Generator().manual_seed(0)
    p = Prec("fp32")
    x = torch.randn(1, 226 + Ms, D, generator=g)
    temb, ref = torch.randn(1, 512, generator=g), torch.randn(1, min(Sr, Ms), D, generator=g)
    cos, sin = prepare_rotary(args.height, args.width, T, 2, 64)
    pre, cp = "transformer_blocks.0.", "perceiver_cross_attention.0."

    def rows():                                                # (a)
        n1, e1, gate, egate = dr.layer_norm_zero(p, sd, pre + "norm1.", x[:, 226:], x[:, :226], temb, 1e-5)
        h = torch.cat([e1, n1], 1)
        q, k, v = (p.linear(h, sd[pre + f"attn1.{n}.weight"], sd[pre + f"attn1.{n}.bias"]).view(1, 226 + Ms, 48, 64).transpose(1, 2)
                   for n in ("to_q", "to_k", "to_v"))
        q = p.layer_norm(q, sd[pre + "attn1.norm_q.weight"], sd[pre + "attn1.norm_q.bias"], 1e-6)
        k = p.layer_norm(k, sd[pre + "attn1.norm_k.weight"], sd[pre + "attn1.norm_k.bias"], 1e-6)
        q = torch.cat([q[:, :, :226], dr.apply_rotary_emb(q[:, :, 226:], cos[:Ms], sin[:Ms])], 2)
        k = torch.cat([k[:, :, :226], dr.apply_rotary_emb(k[:, :, 226:], cos[:Ms], sin[:Ms])], 2)
        o = p.linear(v.transpose(1, 2).reshape(1, 226 + Ms, D), sd[pre + "attn1.to_out.0.weight"], sd[pre + "attn1.to_out.0.bias"])
        x2 = x + torch.cat([egate.expand(1, 226, D), gate.expand(1, Ms, D)], 1) * o
        n2, e2, gate, egate = dr.layer_norm_zero(p, sd, pre + "norm2.", x2[:, 226:], x2[:, :226], temb, 1e-5)
        ff = dr.feed_forward(p, sd, pre + "ff.", torch.cat([e2, n2], 1))
        return x2 + torch.cat([egate.expand(1, 226, D), gate.expand(1, Ms, D)], 1) * ff

    t_rows, x3 = _median_time(rows, reps)
    t_rows *= S / (226 + Ms)
    qh, kh, vh = (torch.randn(1, 1, S, 64, generator=g) for _ in range(3))
    t_attn = _median_time(lambda: dr.sdpa(p, qh, kh, vh, 0.125), reps)[0] * 48                     # (b)

    def crows():                                               # (c)
        xn = p.layer_norm(ref, sd[cp + "norm1.weight"], sd[cp + "norm1.bias"], 1e-5)
        ln = p.layer_norm(x3[:, 226:], sd[cp + "norm2.weight"], sd[cp + "norm2.bias"], 1e-5)
        cq = p.linear(ln, sd[cp + "to_q.weight"]) * 128 ** -0.25
        ckv = p.linear(xn, sd[cp + "to_kv.weight"])
        return p.linear(cq, sd[cp + "to_out.weight"]), ckv

    # the q / out projections scale with Sv, the kv projection with Sr: both sampled at Ms rows
    t_crows = _median_time(crows, reps)[0] * (Sv / Ms)
    cq, ck, cv = torch.randn(1, 1, Sv, 128, generator=g), torch.randn(1, 1, Sr, 128, generator=g), torch.randn(1, 1, Sr, 128, generator=g)
    t_cattn = _median_time(lambda: dr.sdpa(p, cq, ck, cv, 1.0), reps)[0] * 16                      # (d)

    # (e) VAE decoder, default widths, one 2-latent-frame chunk on a crop of the latent grid
    vcfg = dict(ovae.DEFAULT_CONFIG)
    vsd = iw.random_state_dict(iw.vae_param_shapes(vcfg, decoder=True, encoder=False), seed=1)
    ch, cw = min(vae_crop[0], args.height // 8), min(vae_crop[1], args.width // 8)
    z = torch.randn(1, 16, 2, ch, cw, generator=g)
    t0 = time.perf_counter()
    ovae.decoder_forward(p, vsd, vcfg, z, {})
    t_chunk = (time.perf_counter() - t0) * ((args.height // 8) * (args.width // 8)) / (ch * cw)
    n_chunks = T / 2.0
    t_block, t_cross = t_rows + t_attn, t_crows + t_cattn
    denoise_s = 2 * args.denoise_steps * (42 * t_block + 21 * t_cross)
    clip_s = denoise_s + n_chunks * t_chunk
    return {"value": 1.0 / clip_s, "unit": "video-latents/s", "cores": threads, "kind": "port",
            "sample": f"oracle fp32 on {threads} threads, {args.frames}f {args.height}x{args.width} (S={S}), B=1, median of {reps}: CogVideoXBlock = "
                      f"rows/GEMMs {t_rows:.2f} s ({226 + Ms} of {S} tokens timed, linear) + attention {t_attn:.2f} s (1/48 heads timed at full S); "
                      f"PerceiverCrossAttention = rows/GEMMs {t_crows:.2f} s + attention {t_cattn:.2f} s (1/16 heads); VAE decode chunk "
                      f"{t_chunk:.1f} s ({ch}x{cw} of the {args.height // 8}x{args.width // 8} latent grid timed once, linear); clip extrapolated as "
                      f"{2 * args.denoise_steps} forwards x (42 blocks + 21 cross layers) + {n_chunks:.1f} decode chunks; patch / time embeds excluded",
            "extrapolated_clip_seconds": clip_s, "extrapolated_denoise_seconds": denoise_s, "extrapolated_decode_seconds": n_chunks * t_chunk}


def cpu_baseline_bounded(args):
    """`cpu_baseline` in a CHILD process (`bench.py --cpu-baseline-only`, CPU only, never touches the GPU) with a hard time
    limit: a slow host cannot hold back the JSON line, nothing is left running at exit, and this process leaves through the
    normal interpreter shutdown (so a profiler attached to it still writes its output)."""
    skipped = lambda why: {"value": None, "unit": "video-latents/s", "cores": None, "kind": "port", "sample": f"skipped: {why}"}
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--frames", str(args.frames), "--height", str(args.height),
           "--width", str(args.width), "--denoise-steps", str(args.denoise_steps)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")     # the child is a CPU program
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=CPU_BASELINE_BUDGET_S)
    except subprocess.TimeoutExpired:
        return skipped(f"the oracle sample did not finish within {CPU_BASELINE_BUDGET_S:.0f} s on this host")
    if r.returncode != 0:
        return skipped(f"child exited {r.returncode}: {r.stderr.strip()[-300:]}")
    try:
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:                                       # the baseline is a reported extra: never lose the bench line
        return skipped(f"{type(e).__name__}: {e}")


# ------------------------------------------------------------------------------------------- rank plumbing
class DataPlane:
    """State of the data-plane collective (the one all-gather per clip) and the rules for leaving RCCL.

    Every RCCL call runs as a *stage*: on a guard thread (`dp.guarded`, bounded by TCX_RCCL_GUARD_S, default 90 s), followed by a
    MIN vote over the gloo control group, so all ranks keep RCCL or all fall back to the host-staged gloo gather together — a rank
    that fails or hangs early never leaves the others inside an RCCL call.  The fallback is LOUD (stderr + `collective_fallback`
    in the JSON line); TCX_BENCH_RCCL_FATAL=1 turns it into exit code 3."""

    def __init__(self, rank: int, local: int, backend: str):
        self.rank, self.local, self.backend = rank, local, backend
        self.group, self.error, self.abandoned = None, None, 0
        self.guard_s = float(os.environ.get("TCX_RCCL_GUARD_S", 90.0))

    def vote(self, err) -> bool:
        import torch
        import torch.distributed as dist
        ok = torch.tensor([0 if err else 1], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)                  # gloo, host tensor
        return int(ok.item()) == 1

    def step(self, fn, what: str):
        """-> (result, error text or None).  A call that neither returns nor raises within guard_s is reported as hung and its
        thread abandoned in native code (the process must then leave through os._exit, see `finish`)."""
        from trajectorycrafter_amd import dp
        try:
            return dp.guarded(fn, self.guard_s, what, device=self.local), None
        except dp.CollectiveHang as e:
            self.abandoned += 1
            return None, f"CollectiveHang: {e}"
        except Exception as e:
            return None, f"{type(e).__name__}: {str(e)[:400]}"

    def stage(self, fn, what: str):
        """step + vote; on a lost vote every rank falls back.  -> result (None after a fallback)."""
        out, err = self.step(fn, what)
        if not self.vote(err):
            self.fall_back(err, what)
            return None
        return out

    def fall_back(self, err, stage: str) -> None:
        self.error = err or f"RCCL {stage} failed on another rank (see its stderr)"
        if err:
            print(f"[bench] rank {self.rank}: RCCL {stage} FAILED: {err}", file=sys.stderr, flush=True)
        if os.environ.get("TCX_BENCH_RCCL_FATAL") == "1":
            sys.stdout.flush(), sys.stderr.flush()
            os._exit(3)
        print(f"[bench] rank {self.rank}: ALL RANKS FALL BACK to gloo (host-mediated all-gather) — flagged in the JSON line as "
              "`collective_fallback`; this is NOT an RCCL measurement", file=sys.stderr, flush=True)
        self.backend, self.group = "gloo", None

    def finish(self) -> None:
        import torch.distributed as dist
        dist.barrier()
        if self.abandoned:                                          # a thread is still inside a hung RCCL call: no orderly teardown
            sys.stdout.flush(), sys.stderr.flush()
            os._exit(0)
        dist.destroy_process_group()


def _init_control_plane(rank):
    """The gloo control group (rendezvous bounded by dp.DEFAULT_TIMEOUT_S / TCX_DIST_TIMEOUT_S): an unreachable peer ends this
    rank with an error line and exit code 3 in about two minutes instead of the 600 s kill of the driver."""
    from trajectorycrafter_amd import dp
    try:
        dp.init_distributed("gloo")
    except Exception as e:
        print(f"[bench] rank {rank}: init_process_group('gloo') FAILED after <= {dp.dist_timeout().total_seconds():.0f} s: "
              f"{type(e).__name__}: {str(e)[:500]}", file=sys.stderr, flush=True)
        sys.stdout.flush()
        os._exit(3)                                             # a half-built process group can block the normal shutdown


def _selftest_rank(args, rank, world):
    """CPU-only rehearsal of the N-rank control flow on gloo (tests/test_bench_launcher.py): rendezvous, barrier, max-reduce
    of the elapsed time, the single all-gather, one JSON line from rank 0."""
    import torch
    import torch.distributed as dist
    from trajectorycrafter_amd import dp
    if world > 1:
        _init_control_plane(rank)
        dist.barrier()
    t0 = time.perf_counter()
    local = torch.full((1, 3, 2, 4, 4), float(43 + rank))
    elapsed = torch.tensor([time.perf_counter() - t0 + 1e-3 * (rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    # the data-plane rules (DataPlane: guard thread + vote + loud fallback) with a stand-in for the RCCL call: TCX_SELFTEST_RCCL =
    # "ok" | "error" | "hang" | "hang:<rank>" (only that rank's call blocks — the others must still fall back with it)
    plane = DataPlane(rank, None, "nccl" if world > 1 else "gloo")
    sim = os.environ.get("TCX_SELFTEST_RCCL", "ok")

    def fake_rccl():
        if sim == "error":
            raise RuntimeError("simulated RCCL error")
        if sim == "hang" or sim == f"hang:{rank}":
            time.sleep(3600)
        return "communicator"

    if world > 1:
        plane.group = plane.stage(fake_rccl, "new_group(simulated)")
    frames = dp.all_gather_cat(local)
    assert frames.shape[0] == world and frames[:, 0, 0, 0, 0].tolist() == [float(43 + r) for r in range(world)]
    if rank == 0:
        rec = {"metric": "selftest", "value": world / float(elapsed), "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "gathered": list(frames.shape), "collective_backend": plane.backend if world > 1 else None,
               "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
        if plane.error is not None:
            rec["collective_fallback"] = f"gloo, because RCCL failed: {plane.error[:400]}"
        print(json.dumps(rec), flush=True)
    if world > 1:
        plane.finish()


def run_rank(args):
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        print(f"[bench] WORLD_SIZE={world} overrides --gpus {args.gpus}", file=sys.stderr, flush=True)
    if args.selftest_dist:
        return _selftest_rank(args, rank, world)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # rehearsal knob (never set by the driver): several ranks on ONE GPU with gloo to exercise the N > 1 control flow
    if os.environ.get("TCX_BENCH_SINGLE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist
    from trajectorycrafter_amd import dp, ops

    # Two groups.  Control plane (barriers, the max-over-ranks of the timings, the agreement below): a gloo group, always.  Data
    # plane = THE collective of the path, the one all-gather per clip: an RCCL ("nccl") group over xGMI.  Each rank probes it with a
    # one-element all-reduce and the MIN of the outcomes over the gloo group decides for ALL ranks together — a failure on some
    # ranks only can therefore not leave the others waiting inside an RCCL call.  If RCCL cannot run, the gather goes over gloo
    # (host-mediated) LOUDLY: error text on stderr and in the JSON line's top level (`collective_fallback`), so a host-staged gather
    # can never pass for an RCCL scaling number; TCX_BENCH_RCCL_FATAL=1 makes it exit code 3 instead.  TCX_DIST_BACKEND=gloo asks
    # for gloo outright (the one-card rehearsal).
    plane = DataPlane(rank, local, os.environ.get("TCX_DIST_BACKEND", "nccl"))
    devices = None
    if world > 1:
        _init_control_plane(rank)
        mine = {"rank": rank, "local_rank": local, "device": torch.cuda.get_device_name(local), "device_count": torch.cuda.device_count(),
                "host": socket.gethostname()}
        devices = [None] * world
        dist.all_gather_object(devices, mine)                            # gloo: who runs where, printed in config.ranks
        if plane.backend == "nccl":
            # separately voted stages: (1) communicator group creation, (2) a one-element all-reduce
            # The RCCL group's own timeout is LONG on purpose: the bound on a stuck call is the guard (90 s).  torch's NCCL watchdog
            # tears the whole process down when a collective exceeds the group timeout — with 120 s it would kill this rank half a
            # minute after the guard had already declared the call hung and the ranks had fallen back, before the flagged JSON line
            # is printed.  (The product runner has no fallback and keeps the 120 s: there an abort IS the wanted outcome.)
            plane.group = plane.stage(lambda: dist.new_group(backend="nccl", timeout=dp.dist_timeout(1800.0)), "new_group(nccl)")
        if plane.backend == "nccl":
            def probe():
                one = torch.ones(1, device=device)
                dist.all_reduce(one, group=plane.group)
                torch.cuda.synchronize()
                if float(one.item()) != world:
                    raise RuntimeError(f"RCCL probe all-reduce returned {float(one.item())}, expected {world}")
            plane.stage(probe, "probe all_reduce")
        if plane.backend != "nccl":
            print(f"[bench] WARNING: collective backend is {plane.backend!r}, not RCCL: the all-gather is host-mediated; this is a rehearsal, "
                  "not a scaling measurement", file=sys.stderr, flush=True)
    def device_sync():
        # barrier + torch.cuda.synchronize() (driver contract).  Only after an RCCL call has been declared hung — its kernel may
        # still spin on the card, and a device-wide synchronize would wait for it for ever — the launch stream alone is waited for.
        if plane.abandoned:
            torch.cuda.current_stream().synchronize()
        else:
            torch.cuda.synchronize()

    t_init = time.perf_counter()
    pipe = build_models(args, device)
    inp = make_inputs(args, device, seed=43 + rank)            # one independent trajectory per rank (seeds 43..50)
    st = pipe.prepare_denoise(prompt=None, height=args.height, width=args.width, num_frames=args.frames,
                              num_inference_steps=args.denoise_steps, guidance_scale=6.0, **inp)
    device_sync()
    t_init = time.perf_counter() - t_init

    def barrier():
        device_sync()
        if world > 1:
            dist.barrier()
        device_sync()

    def timed(fn, reps=1):
        """fn() `reps` times between barrier + synchronize pairs -> seconds, max over ranks."""
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64)                    # control plane: gloo, host tensor
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    it = {"i": 0}

    def one_step():                                             # consecutive steps of one trajectory; wraps after a clip
        t = st.timesteps[it["i"] % len(st.timesteps)]
        it["i"] += 1
        return pipe.denoise_step(st, t)

    for _ in range(args.warmup):
        one_step()
    ops.attn_timing_start()                                     # HIP events around every tcx_attn_fwd launch
    elapsed, _ = timed(one_step, args.steps)
    attn = ops.attn_timing_stop()
    step_s = elapsed / args.steps
    lat = st.latents
    assert torch.isfinite(lat.float()).all(), "non-finite latents after the timed steps"

    # ---- what the headline depends on.  The self-attention's fastest loop is taken when the q / k LayerNorm
    # parameters PROVE the score bound (random-init: yes).  A real checkpoint may not: time the same steps on the two other paths,
    # after the timed region, never part of `value`: (i) bound unproven = per-workgroup test + complement launch, (ii) the exact
    # running-max kernel for every workgroup of self- and cross-attention = the cost if no row passed the test.
    softmax_path = pipe.transformer.softmax_path_in_use()
    alt = {}
    if args.alt_steps > 0:
        for name, path in (("unproven", "unproven"), ("exact_softmax", "exact")):
            pipe.transformer.set_softmax_path(path)
            one_step()                                          # warm-up of the other kernels
            ops.attn_timing_start()
            el, _ = timed(one_step, args.alt_steps)
            at = ops.attn_timing_stop().get(64, {"n": 0, "ms": 0.0})
            alt[name] = {"ms_per_step": 1e3 * el / args.alt_steps, "attn_avg_launch_ms": at["ms"] / max(at["n"], 1), "steps": args.alt_steps,
                         "paths": pipe.transformer.softmax_path_in_use()}
        pipe.transformer.set_softmax_path("auto")
        assert torch.isfinite(st.latents.float()).all(), "non-finite latents after the alternative-path steps"

    decode_s = gather_s = 0.0
    frames_shape = None
    if not args.no_decode:
        if world == 1:
            pipe.decode_latents(lat)                            # warm-up decode (allocator, weight permutes)
            decode_s, frames = timed(lambda: pipe.decode_latents(lat))       # product call: fp32 frames [1,3,49,H,W] in [0,1]
        else:
            # each rank decodes its own clip to the decoder's bf16 output (channels-last, 102 MB), ONE all-gather moves
            # that, and the fp32 frame conversion (pipeline decode_latents :514-517) runs on the gathered tensor: bit-identical
            # to decode_latents per clip at half the bytes of gathering fp32 frames
            dec = lambda: pipe.vae.decode_cl_bf16(lat.permute(0, 2, 1, 3, 4), scale=1.0 / pipe.vae.config.scaling_factor)
            dec()
            decode_s, out_cl = timed(dec)
            # THE collective of the path, once per clip.  RCCL calls run behind the guard (dp.guarded) and every stage is voted over
            # gloo: warm-up (channel set-up is where a hang would sit), then the timed call.  Host-staged gloo gather otherwise.
            if plane.backend == "nccl":
                plane.stage(lambda: dp.all_gather_cat(out_cl, group=plane.group), "all_gather warm-up")
            if plane.backend == "nccl":
                last = {}

                def gat():
                    out, last["err"] = plane.step(lambda: dp.all_gather_cat(out_cl, group=plane.group), "all_gather_into_tensor(nccl)")
                    return None if out is None else pipe.vae.cl_to_frames(out)

                gather_s, frames = timed(gat)
                if not plane.vote(last["err"]):
                    plane.fall_back(last["err"], "all_gather")
            if plane.backend != "nccl":
                gat = lambda: pipe.vae.cl_to_frames(dp.all_gather_cat(out_cl.cpu()).to(device))    # default (gloo) group, host tensors
                gat()
                gather_s, frames = timed(gat)
        assert frames.shape[0] == world and torch.isfinite(frames).all() and 0.0 <= float(frames.min()) and float(frames.max()) <= 1.0
        frames_shape = list(frames.shape)

    if rank == 0:
        S = ((args.frames - 1) // 4 + 1) * (args.height // 16) * (args.width // 16) + 226
        flop_per_launch = 4.0 * S * S * 3072 * 2                # 4*S^2*D per sample x B=2 (CFG)
        sa = attn.get(64, {"n": 0, "ms": 0.0})
        avg_ms = sa["ms"] / max(sa["n"], 1)
        achieved = flop_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        fwd_flop = {(49, 480, 720): 3.5585e14, (49, 384, 672): 2.3387e14}.get((args.frames, args.height, args.width))
        traffic, traffic_src = None, None
        for name in ("r4_attn_pmc.json", "r3_attn_pmc.json", "r2_attn_pmc.json", "r1_attn_pmc.json"):   # PMC counters cannot be read inside the timed run: committed
            pmc = os.path.join(ROOT, "profiles", name)          # rocprofv3 --pmc result of the same kernel and shape
            if os.path.exists(pmc) and (args.frames, args.height, args.width) == (49, 480, 720):
                with open(pmc) as f:
                    traffic = json.load(f).get("traffic_bytes_per_launch")
                traffic_src = f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)"
                break
        benchmark_config = (args.denoise_steps, args.height, args.width, args.frames, args.layers, args.no_decode) == (50, 480, 720, 49, 42, False)
        clip_s = args.denoise_steps * step_s + decode_s + gather_s
        rec = {
            "metric": "denoised video-latents/sec (49f, 480x720)", "value": world / clip_s,
            "unit": "video-latents/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * step_s, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("configs[2]: full 50-step DDIM (CFG 6, B=2 per step) + VAE decode, 49f 480x720, random-init 42-layer "
                                    "CrossTransformer3D (the 6.1 B-param 5B model); one independent trajectory per GPU") if benchmark_config else
                                   (f"DEBUG (not the benchmark config): {args.denoise_steps}-step DDIM, {args.frames}f {args.height}x{args.width}, "
                                    f"{args.layers} layers, decode={not args.no_decode}"),
                       "step": "one CFG denoising step at full size: B=2 forward of the 42-block model + fused CFG/DDIM update "
                               "(TrajCrafter_Pipeline.denoise_step = reference pipeline_trajectorycrafter.py:1093-1178)",
                       "value_formula": "n_gpus / (denoise_steps * ms_per_step + decode_ms + allgather_ms) * 1000",
                       "frames": args.frames, "height": args.height, "width": args.width, "denoise_steps": args.denoise_steps,
                       "layers": args.layers, "vae_decode": not args.no_decode, "global_batch_clips": world,
                       "parallelism": f"dp{world}", "collective_backend": (plane.backend if world > 1 else None), "control_backend": ("gloo" if world > 1 else None),
                       "ranks": devices if world > 1 else [{"rank": 0, "device": torch.cuda.get_device_name(local), "device_count": torch.cuda.device_count()}],
                       "decode_ms": 1e3 * decode_s, "allgather_ms": 1e3 * gather_s,
                       "clip_seconds": clip_s, "timed_steps_seconds": elapsed, "init_seconds": t_init,
                       "attn_bound_proven": softmax_path["self"] == "proven", "attn_softmax_path": softmax_path,
                       "ms_per_step_unproven": alt.get("unproven", {}).get("ms_per_step"),
                       "ms_per_step_exact_softmax": alt.get("exact_softmax", {}).get("ms_per_step"),
                       "attn_tflops_unproven": (flop_per_launch / (alt["unproven"]["attn_avg_launch_ms"] * 1e-3) / 1e12
                                                if alt.get("unproven", {}).get("attn_avg_launch_ms") else None),
                       "attn_tflops_exact_softmax": (flop_per_launch / (alt["exact_softmax"]["attn_avg_launch_ms"] * 1e-3) / 1e12
                                                     if alt.get("exact_softmax", {}).get("attn_avg_launch_ms") else None),
                       "value_if_unproven": (world / (args.denoise_steps * alt["unproven"]["ms_per_step"] * 1e-3 + decode_s + gather_s)
                                             if "unproven" in alt else None),
                       "value_if_exact_softmax": (world / (args.denoise_steps * alt["exact_softmax"]["ms_per_step"] * 1e-3 + decode_s + gather_s)
                                                  if "exact_softmax" in alt else None),
                       "softmax_path_note": "value / ms_per_step use the path the loaded weights select (attn_softmax_path); the *_unproven and "
                                            "*_exact_softmax figures bracket what other q/k-LayerNorm weights can cost, timed after the timed region",
                       "frames_out": frames_shape,
                       "transformer_mfma_frac": (None if fwd_flop is None else
                                                 2 * fwd_flop * (args.layers / 42) / step_s / 1e12 / PEAK_BF16_TFLOPS)},
            "roofline": {"kernel": "tcx_attn_fwd<64> (joint self-attention, 42 launches per step)", "bound": "mfma",
                         "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "launches": sa["n"], "avg_launch_ms": avg_ms,
                         "algorithmic_flop_per_launch": flop_per_launch},
        }
        dbg = {k: os.environ[k] for k in ("TCX_CONV_GENERIC", "TCX_LIB", "TCX_DIST_BACKEND", "TCX_BENCH_SINGLE_DEVICE") if os.environ.get(k)}
        if dbg:                                                 # switches that change what runs: never unrecorded
            rec["config"]["debug_env"] = dbg
        if world > 1 and plane.backend != "nccl":
            rec["collective_backend_warning"] = f"{plane.backend}: host-mediated all-gather, NOT RCCL — rehearsal only"
        if plane.error is not None:
            rec["collective_fallback"] = f"gloo, because RCCL failed: {plane.error[:400]}"
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline_bounded(args)
        print(json.dumps(rec), flush=True)
    if world > 1:
        plane.finish()


def main(argv=None):
    # dmabuf IPC, required for RCCL on this pool: set before torch / the HSA runtime come up, also when a plain torchrun line
    # (the driver's N > 1 launch) started this process and did not export it
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.cpu_baseline_only:                                  # child process of cpu_baseline_bounded: CPU only
        print(json.dumps(cpu_baseline(args)), flush=True)
        return
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args, argv))                     # no torch.cuda / HIP call has happened in this process
    run_rank(args)


if __name__ == "__main__":
    main()
