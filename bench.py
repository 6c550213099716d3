#!/usr/bin/env python3
"""Headline benchmark: batched bf16 inference of the hybrid CNN-stem + ViT radar classifier
(`cnn_vit_radar.pt` layout: resnet18 stem + E256/H4/L4, BASELINE.json configs[1]) on synthetic
micro-Doppler-shaped inputs, frames/s aggregated over N GPUs (batch sharded, no data-path
collective).

  python bench.py --gpus N --steps K --warmup W [--batch B]
N>1 runs one rank per GPU over RCCL: either the caller launches the ranks (`python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...`, RANK / LOCAL_RANK / WORLD_SIZE in the environment), or, when WORLD_SIZE
is not set, this script starts that launcher itself as a CHILD process before anything touches the GPU and relays
rank 0's JSON line and the child's exit status.  Rank 0 prints ONE JSON line.

A "step" is one forward pass of the hot path (image pack -> stem -> patch embed -> encoder ->
head) over one batch already resident in HBM.  `roofline` is measured live with HIP events around
every launch of the dominant kernel (the LDS-resident 3x3 conv of layers 2-4, csrc/conv3x3.hip); `cpu_baseline` times the torch fp32
oracle (the reference's own torch.nn arithmetic) on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the pool's host driver only supports dmabuf IPC: RCCL needs this before the HIP runtime starts (normally exported
# by the environment already)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

R_CFG = dict(num_classes=11, embed_dim=256, num_heads=4, num_layers=4, dropout=0.1, backbone="resnet18")
# BASELINE.json configs[3]: ViT-B/16-scale radar ViT (pure-ViT `patch16` layout), large-batch bf16 training
VITB_CFG = dict(num_classes=11, embed_dim=768, num_heads=12, num_layers=12, dropout=0.1, backbone="patch16")
VITB_FWD_FLOPS_PER_SAMPLE = 35.13e9
# BASELINE.json configs[4]: ViT-L/16-scale inference, MX-fp8 weights + activations for the encoder GEMMs
VITL_CFG = dict(num_classes=11, embed_dim=1024, num_heads=16, num_layers=24, dropout=0.1, backbone="patch16")
VITL_FWD_FLOPS_PER_SAMPLE = 123.11e9
FP8_MFMA_PEAK_TFLOPS = 5000.0        # dense, block-scaled f8f6f4 MFMA, MI355X_MICROARCH.md:444
BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense, /opt/skills/guides/MI355X_MICROARCH.md:43
HBM_PEAK_GBS = 8000.0            # spec, MI355X_MICROARCH.md:36
FWD_FLOPS_PER_SAMPLE = 3.965e9   # BASELINE.md §2 (R hybrid)


def host_info():
    """CPU model and PHYSICAL core count of this box (SURVEY.md §8d asks for both beside the CPU baseline); `cores` in
    a cpu_baseline object is the number of threads the oracle actually used."""
    model, phys, logical = "unknown", set(), 0
    try:
        pid = cid = None
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                k, _, v = line.partition(":")
                k, v = k.strip(), v.strip()
                if k == "model name":
                    model = v
                elif k == "processor":
                    logical += 1
                elif k == "physical id":
                    pid = v
                elif k == "core id":
                    cid = v
                    phys.add((pid, cid))
    except OSError:
        pass
    return {"cpu_model": model, "host_physical_cores": len(phys) or None, "host_logical_cpus": logical or os.cpu_count()}


def cpu_baseline(sd, seconds=20.0, batch=8):
    """Oracle (kind='port': same torch.nn calls as the reference, train_vit.py:93-127 + ResNet-18
    restatement) on the host cores; mirrors the reference's batch-8 loop (infer.py:459-469)."""
    from oracle.ref_model import CNNViTHybridRef
    from vit_amd.synth import micro_doppler_batch
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(cores)
    ref = CNNViTHybridRef(**R_CFG)
    ref.load_state_dict(sd)
    ref.eval()
    x = micro_doppler_batch(batch, seed=1234)
    with torch.no_grad():
        for _ in range(2):
            ref(x)
        n, t0 = 0, time.perf_counter()
        while n < 25 and time.perf_counter() - t0 < seconds:
            ref(x)
            n += 1
        dt = time.perf_counter() - t0
    return {"value": round(n * batch / dt, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} forwards of batch {batch}, torch {torch.__version__} fp32 eval fast path, "
                      f"same weights/inputs generator", **host_info()}


def cpu_train_baseline(sd, cfg, steps=4, batch=32, seconds=25.0):
    """The oracle's training step on the host cores beside the `train` figure: forward (train mode) + label-smoothed
    CE + backward + clip + AdamW at the reference's batch size 32 (train_vit.py:407-437,505-515); SURVEY.md §8d."""
    from oracle.ref_model import CNNViTHybridRef
    from vit_amd.synth import micro_doppler_batch
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(cores)
    ref = CNNViTHybridRef(**cfg)
    if sd is not None:
        ref.load_state_dict(sd)
    ref.train()
    x = micro_doppler_batch(batch, seed=99)
    y = torch.randint(0, cfg["num_classes"], (batch,), generator=torch.Generator().manual_seed(0))
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-4, weight_decay=0.05)
    crit = torch.nn.CrossEntropyLoss(label_smoothing=0.05)

    def step():
        opt.zero_grad(set_to_none=True)
        crit(ref(x), y).backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        opt.step()

    step()
    n, t0 = 0, time.perf_counter()
    while n < steps and time.perf_counter() - t0 < seconds:
        step()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(n * batch / dt, 2), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{n} optimiser steps of batch {batch} (fp32, torch {torch.__version__}), same model / loss / optimiser",
            **host_info()}


def max_over_ranks(dist, dt: float, dev) -> float:
    """MAX of the timed interval over all ranks (RCCL: a device tensor; the gloo rehearsal: a host tensor, because
    gloo's device staging must not be used on this platform, vit_amd/dp.py)."""
    if dist is None:
        return dt
    on_dev = "nccl" in str(dist.get_backend()).lower()
    t = torch.tensor([dt], dtype=torch.float64, device=dev if on_dev else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _say(rank, msg):
    if os.environ.get("RVIT_BENCH_VERBOSE"):
        print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)


def kernel_roofline(kernels, peak, steps):
    """Roofline object of the dominant MFMA kernel of a leg from its HIP-event records {name: {ms, flops, launches}}
    (the spans of vit_amd/ops.py carry the algorithmic FLOPs of each launch), and the same figure per kernel family."""
    mf = {n: k for n, k in kernels.items() if k["flops"] > 0}
    if not mf:
        return None, {}
    dom = max(mf, key=lambda n: mf[n]["ms"])
    k = mf[dom]
    ach = k["flops"] / (k["ms"] * 1e-3) / 1e12
    roof = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(ach / peak, 4), "traffic": None, "avg_launch_us": round(k["ms"] / k["launches"] * 1e3, 2),
            "launches_per_step": k["launches"] // steps, "ms_per_step": round(k["ms"] / steps, 3)}
    fam = {}
    for n, k in mf.items():
        f = "attn" if n.startswith("attn") else "wgrad" if "wgrad" in n else "gemm" if n.startswith("gemm") else n
        d = fam.setdefault(f, {"ms": 0.0, "flops": 0.0})
        d["ms"] += k["ms"]
        d["flops"] += k["flops"]
    # every family is priced against the peak of the MFMA it runs on: only the block-scaled MX GEMMs reach the fp8 peak
    # (the MX-fp8 mode's attention - bf16 MFMA, or un-scaled fp8 at the same rate - and its patch-embedding conv are not)
    def fam_peak(f):
        mx = f == "gemm" and any(n.startswith("gemm_mx") for n in mf)   # un-scaled fp8 MFMAs run at the bf16 rate
        return peak if (mx or peak == BF16_MFMA_PEAK_TFLOPS) else BF16_MFMA_PEAK_TFLOPS
    split = {f: {"ms_per_step": round(d["ms"] / steps, 3), "tflops": round(d["flops"] / (d["ms"] * 1e-3) / 1e12, 1),
                 "peak": fam_peak(f), "frac_of_peak": round(d["flops"] / (d["ms"] * 1e-3) / 1e12 / fam_peak(f), 4)}
             for f, d in sorted(fam.items())}
    return roof, split


def train_leg(args, dev, dist, world, rank, sd, cfg=None, fwd_flops=FWD_FLOPS_PER_SAMPLE, batch=None, steps=None,
              profile=True, cpu=True):
    """Secondary figure (BASELINE.json configs[3] shape on the R model): one optimiser step =
    forward in train() mode (BatchNorm batch statistics) + label-smoothed CE + backward on librvit +
    bucketed gradient all-reduce (N>1, overlapped with backward) + grad-norm clipping + AdamW
    (train_vit.py:407-437).  Same timing discipline as the headline number."""
    from vit_amd import ops
    from vit_amd.loss import CrossEntropyLoss
    from vit_amd.model import CNNViTHybrid
    from vit_amd.synth import micro_doppler_batch
    cfg = dict(R_CFG if cfg is None else cfg, dropout=args.train_dropout)
    model = CNNViTHybrid(**cfg)
    if sd is None:
        from vit_amd.synth import synth_state_dict
        sd = synth_state_dict({k: v.shape for k, v in model.state_dict().items()}, seed=0)
    model.load_state_dict(sd)
    model = model.to(dev).train()
    if world > 1:
        model.set_data_parallel()
    B = args.train_batch if batch is None else batch
    n_steps = args.train_steps if steps is None else steps
    x = torch.cat([micro_doppler_batch(min(64, B - i), seed=99 + 1000 * rank + i) for i in range(0, B, 64)]).to(dev)
    y = torch.randint(0, cfg["num_classes"], (B,), generator=torch.Generator().manual_seed(rank)).to(dev)
    fused = not args.stock_optimizer
    if fused:   # clip + AdamW in two launches over the flat gradient buffer (vit_amd/optim.py)
        from vit_amd.optim import FusedAdamW
        opt = FusedAdamW(model, lr=1e-4, weight_decay=0.05, max_grad_norm=1.0)
    else:
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.05, fused=True)
    crit = CrossEntropyLoss(label_smoothing=0.05)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = crit(model(x), y)
        loss.backward()
        if not fused:
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        return loss

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    _say(rank, "train: warm-up")
    for _ in range(3):
        step()
    sync()
    _say(rank, "train: timed steps")
    t0 = time.perf_counter()
    for _ in range(n_steps):
        loss = step()
    t_host = time.perf_counter() - t0      # host time to ENQUEUE the steps (before the sync): launch-bound if ~ dt
    sync()
    _say(rank, "train: timed steps done")
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dist, dt, dev)
    kernels = {}
    if profile and (rank == 0 or dist is not None):  # every rank must run the same collectives
        rec = []
        ops.set_profiler(rec)
        for _ in range(2):
            step()
        torch.cuda.synchronize(dev)
        ops.set_profiler(None)
        for name, e0, e1, fl, nb in rec:
            k = kernels.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0})
            k["launches"] += 1
            k["ms"] += e0.elapsed_time(e1)
            k["flops"] += fl
    if rank != 0:
        return None
    sps = world * B * n_steps / dt
    cpu_res = None
    if cpu and world == 1 and not args.no_cpu_baseline and cfg["embed_dim"] <= 256:   # the R model only: ViT-B steps take minutes on a CPU
        cpu_res = cpu_train_baseline(sd, cfg)
    roof, split = kernel_roofline(kernels, BF16_MFMA_PEAK_TFLOPS, 2)
    return {"metric": "train_samples_per_sec", "cpu_baseline": cpu_res, "value": round(sps, 1), "unit": "samples/s",
            "ms_per_step": round(dt / n_steps * 1e3, 3),
            "host_enqueue_ms_per_step": round(t_host / n_steps * 1e3, 3), "steps": n_steps, "warmup": 3,
            "launch_plan_replays": dict(getattr(model.train_engine, "plan_replays", {})),   # vit_amd/plan.py
            "dtype": "bf16", "roofline": roof, "kernel_families": split,
            "batch_per_gpu": B,
            "global_batch": world * B, "dropout": cfg["dropout"], "final_loss": round(float(loss.detach()), 4),
            "includes": "forward(train-mode BN) + CE(label_smoothing=0.05) + backward + "
                        + ("bucketed RCCL all-reduce + " if world > 1 else "") + ("librvit fused clip+AdamW" if fused else "clip_grad_norm_ + torch AdamW(fused=True)"),
            "model_tflops": round(sps * 3 * fwd_flops / 1e12, 2),
            "mfma_util_model": round(sps * 3 * fwd_flops / 1e12 / (BF16_MFMA_PEAK_TFLOPS * world), 4),
            "kernel_ms_per_step": {n: round(v["ms"] / 2, 4) for n, v in sorted(kernels.items())}}


def vitl_infer(args, dev, dist, world, rank, batch=None, steps=None, warmup=None):
    """BASELINE.json configs[4]: ViT-L/16-scale forward, encoder Linears on the MX-fp8 MFMA; the same model in
    bf16 is timed beside it.  Weights: random normal (std 0.02, torch generator) - throughput only.  Returns the
    result object on rank 0 (None elsewhere)."""
    from vit_amd import ops
    from vit_amd.model import CNNViTHybrid
    from vit_amd.synth import micro_doppler_batch
    torch.manual_seed(0)
    model = CNNViTHybrid(**VITL_CFG)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() >= 2:
                p.normal_(0.0, 0.02)
    model = model.to(dev).eval()
    B = batch if batch is not None else (args.batch if args.batch != 512 else 256)   # configs[4] sweeps 8..1024; 256 by default
    n_steps = args.steps if steps is None else steps
    n_warm = args.warmup if warmup is None else warmup
    x = torch.cat([micro_doppler_batch(min(64, B - i), seed=1234 + 1000 * rank + i) for i in range(0, B, 64)]).to(dev)

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    res = {}
    kernels = {}
    for prec in ("bf16", "mxfp8"):
        model.set_inference_precision(prec)
        with torch.no_grad():
            for _ in range(n_warm):
                model(x)
            sync()
            t0 = time.perf_counter()
            for _ in range(n_steps):
                model(x)
            sync()
            dt = time.perf_counter() - t0
        dt = max_over_ranks(dist, dt, dev)
        res[prec] = dt
        if rank == 0 and prec == "mxfp8":
            rec = []
            ops.set_profiler(rec)
            with torch.no_grad():
                for _ in range(2):
                    model(x)
            torch.cuda.synchronize(dev)
            ops.set_profiler(None)
            for name, e0, e1, fl, nb in rec:
                k = kernels.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0})
                k["launches"] += 1
                k["ms"] += e0.elapsed_time(e1)
                k["flops"] += fl
    if rank != 0:
        return None
    dt = res["mxfp8"]
    value = world * B * n_steps / dt
    roof, split = kernel_roofline(kernels, FP8_MFMA_PEAK_TFLOPS, 2)
    out = {"metric": "inference_frames_per_sec", "value": round(value, 1), "unit": "frames/s", "n_gpus": world,
           "steps": n_steps, "warmup": n_warm, "ms_per_step": round(dt / n_steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp8-e4m3 (MX)",
           "data": "synthetic",
           "config": {"workload": "ViT-L/16-scale radar ViT (patch16, E1024/H16/L24, 11 classes) batched inference, "
                                  "MX-fp8 encoder Linears (weights + activations), " + ("fp8-MFMA attention" if os.environ.get("RVIT_FP8_ATTENTION") == "1" else "bf16-MFMA attention") + " with MX-fp8 output, fp32 LayerNorm statistics",
                      "batch_per_gpu": B, "global_batch": world * B, "parallelism": f"batch-shard x{world}"},
           "model_tflops": round(value * VITL_FWD_FLOPS_PER_SAMPLE / 1e12, 2),
           "mfma_util_model": round(value * VITL_FWD_FLOPS_PER_SAMPLE / 1e12 / (FP8_MFMA_PEAK_TFLOPS * world), 4),
           "roofline": roof, "kernel_families": split,
           "bf16_same_model": {"value": round(world * B * n_steps / res["bf16"], 1), "unit": "frames/s",
                               "ms_per_step": round(res["bf16"] / n_steps * 1e3, 3)},
           "kernel_ms_per_step": {n: round(v["ms"] / 2, 4) for n, v in sorted(kernels.items())}}
    return out


def spawn_ranks(n: int) -> int:
    """`bench.py --gpus N` without a launcher: start `torch.distributed.run` with N ranks of this same command line
    as a child process (never exec: nothing here may have touched the GPU yet, and nothing has - `import torch` and
    `device_count()` do not initialise HIP), stream its output through, return its exit status."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("RVIT_BENCH_BACKEND", "nccl") == "nccl":
        print(f"[bench] --gpus {n} requested but {have} visible: RCCL needs one GPU per rank", file=sys.stderr, flush=True)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:          # rank 0's JSON line (and anything else the ranks print) passes through
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-batch", type=int, default=256, help="samples per GPU per training step")
    ap.add_argument("--train-steps", type=int, default=8)
    ap.add_argument("--train-dropout", type=float, default=0.1)
    ap.add_argument("--stock-optimizer", action="store_true",
                    help="training legs: clip_grad_norm_ + torch.optim.AdamW(fused=True) instead of vit_amd.optim.FusedAdamW")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary training-throughput leg")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the ViT-B/16-scale training and ViT-L/16-scale MX-fp8 inference legs (BASELINE configs[3], [4])")
    ap.add_argument("--train-timeout", type=float, default=240.0,
                    help="seconds after which a stalled training leg is abandoned (the inference line is still printed)")
    ap.add_argument("--workload", choices=["hybrid", "vitb-train", "vitl-infer-fp8"], default="hybrid",
                    help="hybrid = the headline (BASELINE configs[1] inference + configs[2] training leg); "
                         "vitb-train = BASELINE configs[3], ViT-B/16-scale training samples/s; "
                         "vitl-infer-fp8 = BASELINE configs[4], ViT-L/16-scale MX-fp8 inference frames/s")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: the launcher's world size is used",
              file=sys.stderr, flush=True)

    if os.environ.get("RVIT_BENCH_VERBOSE"):  # where is every thread after 45 s?  (hang diagnosis)
        import faulthandler
        faulthandler.dump_traceback_later(45, exit=False)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # one process per GPU over RCCL; RVIT_BENCH_BACKEND=gloo + several ranks on one card is only a rehearsal of
    # the multi-process code path on a 1-GPU box (scripts/rehearse_multigpu.sh), never a measurement
    backend = os.environ.get("RVIT_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    if args.workload == "vitb-train":
        tr = train_leg(args, dev, dist, world, rank, None, cfg=VITB_CFG, fwd_flops=VITB_FWD_FLOPS_PER_SAMPLE)
        if rank == 0:
            out = {"metric": "train_samples_per_sec", "value": tr["value"], "unit": "samples/s", "n_gpus": world,
                   "steps": tr["steps"], "warmup": 3, "ms_per_step": tr["ms_per_step"], "higher_is_better": True,
                   "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                   "config": {"workload": "ViT-B/16-scale radar ViT (patch16, E768/H12/L12, 11 classes) training step: "
                                          + tr["includes"], "batch_per_gpu": tr["batch_per_gpu"],
                              "global_batch": tr["global_batch"], "parallelism": f"dp{world}"},
                   "model_tflops": tr["model_tflops"], "mfma_util_model": tr["mfma_util_model"],
                   "roofline": tr["roofline"], "kernel_families": tr["kernel_families"],
                   "kernel_ms_per_step": tr["kernel_ms_per_step"]}
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.workload == "vitl-infer-fp8":
        res = vitl_infer(args, dev, dist, world, rank)
        if rank == 0:
            print(json.dumps(res), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    from vit_amd import ops
    from vit_amd.model import CNNViTHybrid
    from vit_amd.synth import micro_doppler_batch, synth_state_dict

    model = CNNViTHybrid(**R_CFG)
    sd = synth_state_dict({k: v.shape for k, v in model.state_dict().items()}, seed=0)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    B = args.batch
    # per-rank synthetic batch (seed differs per rank), generated in chunks, resident in HBM
    x = torch.cat([micro_doppler_batch(min(64, B - i), seed=1234 + 1000 * rank + i) for i in range(0, B, 64)]).to(dev)

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    _say(rank, "inference: warm-up")
    with torch.no_grad():
        for _ in range(args.warmup):
            model(x)
        sync()
        _say(rank, "inference: timed steps")
        t0 = time.perf_counter()
        for _ in range(args.steps):
            model(x)
        sync()
        dt = time.perf_counter() - t0
    dt = max_over_ranks(dist, dt, dev)

    # ---- per-kernel timing with HIP events on the launch stream (same steps, instrumented) ------
    roofline = None
    kernels = {}
    if rank == 0:
        rec = []
        ops.set_profiler(rec)
        with torch.no_grad():
            for _ in range(min(args.steps, 5)):
                model(x)
        torch.cuda.synchronize(dev)
        ops.set_profiler(None)
        for name, e0, e1, fl, nb in rec:
            k = kernels.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            k["launches"] += 1
            k["ms"] += e0.elapsed_time(e1)
            k["flops"] += fl
            k["bytes"] += nb
        dom = max(kernels, key=lambda n: kernels[n]["ms"])
        k = kernels[dom]
        avg_ms = k["ms"] / k["launches"]
        if k["flops"] > 0:
            ach = k["flops"] / k["launches"] / (avg_ms * 1e-3) / 1e12
            roofline = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": BF16_MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                        "avg_launch_us": round(avg_ms * 1e3, 2), "launches_per_step": k["launches"] // min(args.steps, 5)}
        else:
            ach = k["bytes"] / k["launches"] / (avg_ms * 1e-3) / 1e9
            roofline = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                        "avg_launch_us": round(avg_ms * 1e3, 2), "launches_per_step": k["launches"] // min(args.steps, 5)}

    _say(rank, "inference: done")
    out = None
    if rank == 0:
        if roofline is not None:
            roofline["algorithmic_bytes_per_launch"] = round(kernels[roofline["kernel"]]["bytes"] / kernels[roofline["kernel"]]["launches"])
            # HBM traffic of the same kernel from the committed PMC passes (profiles/, same command and batch).  The
            # file carries the hash of the kernel's source: a traffic figure measured on another version of the kernel
            # is not replayed (traffic stays null until scripts/profile_r03.sh + make_pmc_json_r03.py are run again).
            here = os.path.dirname(os.path.abspath(__file__))
            pmc = os.path.join(here, "bench_pmc_r03.json")
            if os.path.exists(pmc):
                import hashlib
                with open(pmc) as fh:
                    t = json.load(fh)
                src = os.path.join(here, t.get("kernel_source", ""))
                same_src = os.path.isfile(src) and hashlib.sha256(open(src, "rb").read()).hexdigest() == t.get("kernel_source_sha256")
                if t.get("kernel") == roofline["kernel"] and t.get("batch_per_gpu") == B and same_src:
                    roofline["traffic"] = t["traffic_bytes_per_launch"]
                    roofline["traffic_source"] = ("bench_pmc_r03.json = profiles/r03_pmc_traffic.json (rocprofv3 --pmc "
                                                  "FETCH_SIZE x2 + WRITE_SIZE, per launch, same command, same kernel source)")
        frames = world * B * args.steps
        value = frames / dt
        out = {
            "metric": "inference_frames_per_sec", "value": round(value, 1), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "cnn_vit_radar hybrid (resnet18 stem + ViT E256/H4/L4, 11 classes) batched "
                                   "inference, 224x224 synthetic micro-Doppler maps, synthetic weights",
                       "batch_per_gpu": B, "global_batch": world * B, "parallelism": f"batch-shard x{world}"},
            "model_tflops": round(value * FWD_FLOPS_PER_SAMPLE / 1e12, 2),
            "mfma_util_model": round(value * FWD_FLOPS_PER_SAMPLE / 1e12 / (BF16_MFMA_PEAK_TFLOPS * world), 4),
            "roofline": roofline,
            "kernel_ms_per_step": {n: round(v["ms"] / min(args.steps, 5), 4) for n, v in sorted(kernels.items())},
        }

    # ---- secondary legs: BASELINE configs[2] (hybrid training), configs[3] (ViT-B/16-scale training), configs[4]
    # (ViT-L/16-scale MX-fp8 inference).  The headline above is already measured: a leg that stalls (a wedged
    # collective, a lost peer) must not lose it.  After --train-timeout seconds in ONE leg every rank gives up: rank 0
    # still prints the line, with that leg's `error` set, and EVERY rank exits with status 3 - a process that touched
    # the GPU and stalled has failed.
    import gc
    import threading
    if not args.no_train:
        del x

    def run_leg(name, fn):
        def give_up():
            import faulthandler
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
            if out is not None:
                out[name] = {"error": f"leg did not finish within {args.train_timeout:.0f} s; abandoned, exit status 3"}
                print(json.dumps(out), flush=True)
            sys.stderr.flush()
            os._exit(3)

        dog = threading.Timer(args.train_timeout, give_up)
        dog.daemon = True
        dog.start()
        _say(rank, f"leg {name}")
        res = fn()
        dog.cancel()
        gc.collect()
        torch.cuda.empty_cache()
        if out is not None and res is not None:
            out[name] = res
        return res

    if not args.no_train:
        train = run_leg("train", lambda: train_leg(args, dev, dist, world, rank, sd))
        # SURVEY.md §8d config 3 names batch 32 (the reference's, train_vit.py:505-515) and 512 per GPU
        sweep = []
        for b in (32, 512):
            if b != args.train_batch and not args.no_extra_legs:
                r = run_leg(f"train_b{b}", lambda b=b: train_leg(args, dev, dist, world, rank, sd, batch=b, steps=8,
                                                                  profile=False, cpu=False))
                if out is not None:
                    out.pop(f"train_b{b}", None)
                    sweep.append({k: r[k] for k in ("batch_per_gpu", "global_batch", "value", "unit", "ms_per_step",
                                                    "host_enqueue_ms_per_step", "steps", "mfma_util_model")})
        if out is not None and train is not None:
            train["other_batches"] = sweep
    if not args.no_extra_legs:
        def vitb():
            tr = train_leg(args, dev, dist, world, rank, None, cfg=VITB_CFG, fwd_flops=VITB_FWD_FLOPS_PER_SAMPLE,
                           batch=256, steps=6, cpu=False)
            if tr is not None:
                tr["config"] = {"workload": "ViT-B/16-scale radar ViT (patch16, E768/H12/L12, 11 classes) training step "
                                            "(BASELINE configs[3])", "parallelism": f"dp{world}"}
            return tr
        run_leg("vitb_train", vitb)
        run_leg("vitl_fp8", lambda: vitl_infer(args, dev, dist, world, rank, batch=256, steps=10, warmup=3))

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
