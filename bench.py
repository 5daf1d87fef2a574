#!/usr/bin/env python3
"""bench.py -- Wide&Deep training throughput on the MI355X embedding path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one synthetic Criteo-shaped batch that is already resident
in HBM: id dedup + inverted index, deep/wide lookups, the 5-layer MLP forward/backward, the fused
segment-sum + LazyAdam apply on the deep table, the FTRL apply on the wide table and the dense
optimizer (mindrec_amd/wide_deep.py; reference models/wide_deep/src/wide_and_deep.py:472-492).
Workload = BASELINE.json configs[1]: vocab 200 M, dim 80, batch 16384 per GPU, 26 categorical slots,
fp32 tables.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--vocab", type=int, default=200_000_000)
    ap.add_argument("--emb-dim", type=int, default=80)
    ap.add_argument("--batch", type=int, default=16384, help="per-GPU batch (reference passes batch_size per worker)")
    ap.add_argument("--fields", type=int, default=26, help="26 = north-star categorical slots; 39 = reference field_size")
    ap.add_argument("--dist", default="uniform", choices=["uniform", "zipf"])
    ap.add_argument("--mlp-dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--n-batches", type=int, default=4, help="distinct resident batches cycled through")
    ap.add_argument("--split-state", action="store_true", help="p, m, v as three separate arrays (default: fused rows)")
    ap.add_argument("--no-overlap-plan", action="store_true", help="run dedup + inverted index on the main stream")
    ap.add_argument("--no-overlap-wide-apply", action="store_true", help="wide FTRL after the deep apply on the main stream")
    ap.add_argument("--late-wide", choices=["auto", "on", "off"], default="auto", help="wide branch on the side stream under the hidden-layer GEMMs (auto: when sharded)")
    ap.add_argument("--no-early-route", action="store_true", help="shards: request exchange on the main stream (waits for the previous step)")
    ap.add_argument("--parallel-dw-from", type=int, default=0)
    ap.add_argument("--parallel-dw", action="store_true", help="weight-gradient GEMMs on a parallel branch of the backward (measured slower)")
    ap.add_argument("--overlap-dw0", action="store_true", help="first-layer weight-gradient GEMM beside the sparse apply (side stream)")
    ap.add_argument("--dynamic-embedding", action="store_true", help="hash tables keyed by the raw ids (reference --dynamic_embedding=True); "
                    "use a --vocab small enough for --hash-capacity, e.g. --vocab 3000000")
    ap.add_argument("--hash-capacity", type=int, default=1 << 22)
    ap.add_argument("--host-cache-rows", type=int, default=0, help="tables in pinned host DRAM behind a device cache of this many rows "
                    "(the reference's vocab_cache_size); keep --vocab x 976 B within the host's RAM")
    ap.add_argument("--no-relu-epilogue", action="store_true", help="hidden layers as addmm + a separate ReLU pass")
    ap.add_argument("--no-plan-first", action="store_true", help="plan queued behind the gathers")
    ap.add_argument("--no-graph-front", action="store_true", help="one GPU: only the MLP as HIP graphs, lookups / plan issued kernel by kernel")
    ap.add_argument("--no-graph-mlp", action="store_true", help="issue the fused MLP step kernel by kernel instead of replaying its HIP graph")
    ap.add_argument("--overlap-wide", action="store_true", help="also run wide_sum on the side stream (measured slower)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the product path); gloo stages collectives through the host and lets "
                         "several ranks share one GPU (debugging only)")
    return ap.parse_args()


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if xs else float("nan")


def cpu_baseline(args, seconds):
    """The oracle (CPU restatement, scalar C, 1 thread) on a bounded sample of the same workload:
    the embedding path only (lookup + wide sum + sparse LazyAdam + sparse FTRL) at the same batch
    shape, with the table scaled down to fit host RAM (random-row bandwidth is insensitive to V once
    V*D*4 >> LLC).  MindSpore's own CPU path cannot be timed: it is not installable here."""
    import numpy as np
    from oracle import oracle as O
    V = min(args.vocab, 2_000_000)
    D, B, Fd = args.emb_dim, args.batch, args.fields
    rng = np.random.default_rng(1000)
    p = (rng.standard_normal((V, D)) * 0.01).astype(np.float32)
    m = np.zeros_like(p); v = np.zeros_like(p)
    w = (rng.standard_normal((V, 1)) * 0.01).astype(np.float32); wa = np.ones_like(w); wl = np.zeros_like(w)
    g = rng.standard_normal((B * Fd, D)).astype(np.float32)
    gw = rng.standard_normal((B * Fd, 1)).astype(np.float32)
    wts = np.ones((B, Fd), np.float32)
    steps, t0 = 0, time.perf_counter()
    while True:
        ids = rng.integers(0, V, size=(B, Fd)).astype(np.int32)
        ts = time.perf_counter()
        O.gather_rows(p, ids, wts)
        O.wide_sum(w, ids, wts, 0.0)
        O.sparse_lazy_adam(p, m, v, ids, g, wts, grad_scale=1 / 1024)
        O.sparse_ftrl(w, wa, wl, ids, gw, None, grad_scale=1 / 1024)
        steps += 1
        if time.perf_counter() - t0 > seconds or steps >= 200:
            break
        _ = ts
    dt = time.perf_counter() - t0
    t_embed = dt / steps
    # the dense net of the same step (fwd + bwd, fp32) on the host's cores through torch's CPU GEMMs
    import torch
    import torch.nn.functional as F
    threads = max(1, min(os.cpu_count() or 1, 16))
    torch.set_num_threads(threads)
    dims = [Fd * D, 1024, 512, 256, 128, 1]
    Ws = [(torch.randn(dims[i], dims[i + 1]) * 0.01).requires_grad_(True) for i in range(5)]
    bs = [torch.zeros(dims[i + 1], requires_grad=True) for i in range(5)]
    x = torch.randn(B, Fd * D, requires_grad=True)
    y = (torch.rand(B, 1) < 0.25).float()
    t_mlp, reps = 0.0, 0
    for it in range(4):
        t1 = time.perf_counter()
        h = x
        for i in range(5):
            h = torch.addmm(bs[i], h, Ws[i])
            if i < 4:
                h = torch.relu(h)
        F.binary_cross_entropy_with_logits(h, y).backward()
        if it:                      # first pass warms the thread pool
            t_mlp += time.perf_counter() - t1
            reps += 1
        if time.perf_counter() - t0 > 2.5 * seconds:
            break
    t_mlp = t_mlp / max(reps, 1)
    whole = B / (t_embed + t_mlp) if reps else None
    return {"value": round(whole, 1) if whole else round(B / t_embed, 1), "unit": "samples/s", "cores": threads if reps else 1,
            "kind": "port", "embedding_path_only": round(B / t_embed, 1), "embedding_ms": round(t_embed * 1e3, 1),
            "mlp_ms": round(t_mlp * 1e3, 1) if reps else None,
            "sample": f"whole step = embedding path (lookup+wide_sum+sparse LazyAdam+sparse FTRL: oracle/mrec_oracle.c, 1 thread, "
                      f"{steps} steps) + MLP {dims[0]}-1024-512-256-128-1 fwd+bwd (torch CPU fp32, {threads} threads, {reps} steps); "
                      f"batch {B}x{Fd}, dim {D}, table scaled to V={V}, uniform ids; MindSpore CPU not installable here"}


def main():
    # RCCL / cross-process tensor sharing on this pool needs dmabuf IPC (the driver image exports this already)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    args = parse()
    import torch
    import torch.distributed as dist
    from mindrec_amd import _lib
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, embedding_bytes, synthetic_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (mindrec_amd has no CPU fallback)")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rc = _lib.lib().mrec_device_ok()
    if rc != 0:
        raise SystemExit(f"libmrec_hip.so cannot run on this device: {_lib.lib().mrec_strerror(rc).decode()}")
    group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    cfg = WideDeepConfig(vocab_size=args.vocab, emb_dim=args.emb_dim, field_size=args.fields, batch_size=args.batch,
                         mlp_dtype=args.mlp_dtype, fused_state=not args.split_state,
                         overlap_plan=not args.no_overlap_plan, overlap_wide=args.overlap_wide,
                         graph_mlp=not args.no_graph_mlp, graph_front=not args.no_graph_front, plan_first=not args.no_plan_first, relu_epilogue=not args.no_relu_epilogue,
                         dynamic_embedding=args.dynamic_embedding, hash_capacity=args.hash_capacity,
                         host_cache_rows=args.host_cache_rows, overlap_dw0=args.overlap_dw0, parallel_dw=args.parallel_dw, parallel_dw_from=args.parallel_dw_from, early_route=not args.no_early_route, late_wide={'auto': None, 'on': True, 'off': False}[args.late_wide],
                         overlap_wide_apply=not args.no_overlap_wide_apply)
    eng = WideDeepEngine(cfg, dev, rank=rank, world=world, group=group)
    batches = [synthetic_batch(cfg, dev, args.dist, seed=1000 + i, rank=rank) for i in range(args.n_batches)]
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Priming (setup, whatever --warmup says): workspaces, GEMM library handles and the HIP graph of the MLP
    # step come into being in the engine's first three steps.
    # (2 x n_batches more steps let a one-GPU engine bind a front graph to each of the rotating input batches.)
    for i in range(3 + 2 * len(batches)):
        eng.train_step(*batches[i % len(batches)])
    barrier()
    for i in range(args.warmup):
        eng.train_step(*batches[i % len(batches)])
    barrier()
    from mindrec_amd import ops
    ktimers = [ops.KernelTimer() for _ in range(args.steps)]     # HIP events around k_apply_main only
    t0 = time.perf_counter()
    for i in range(args.steps):
        eng.deep_apply_timer = ktimers[i]      # armed by the engine right before the deep-table LazyAdam launch
        eng.train_step(*batches[i % len(batches)])
    barrier()
    dt = time.perf_counter() - t0
    graphs_used = {"front": eng._front_graph is not None or bool(eng._front_bound), "mlp": eng._mlp_graph is not None}
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # Per-phase device times (informational "kernels_ms"): HIP events around every phase, recorded in a few
    # EXTRA steps after the timed region -- two dozen timing events per step serialise the queue and cost
    # 7-10 % of the step, so they must not sit inside the measurement.  Only the two events around the
    # dominant kernel (roofline.avg_ms) are recorded in the timed steps.
    eng.timers = {}
    for i in range(min(args.steps, 8) + 2):
        eng.train_step(*batches[i % len(batches)])
    barrier()
    # the first two of these steps are dropped: with phase timers on, a one-GPU engine leaves its whole-front graph
    # and captures the MLP graphs instead, which happens here
    kern_ms = {k: [a.elapsed_time(b) for a, b in evs][2:] for k, evs in eng.timers.items()}
    eng.timers = None
    N = args.batch * args.fields
    plan = eng.last_plan
    U = plan.U                                  # unique ids of the last step's (local) apply
    n_apply = plan.n
    bf16_io = eng._fused_bf16()                       # gather writes / apply reads bf16 rows (also on the wire)
    by = embedding_bytes(n_apply, U, args.emb_dim, act_bytes=2 if bf16_io else 4)
    kmain = [t.ms() for t in ktimers]
    apply_ms = sum(kmain) / len(kmain)
    achieved = by["apply_deep"] / (apply_ms * 1e-3) / 1e9
    peak = 8000.0
    # HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/, collected
    # with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of this same command and corrected
    # per MI355X_MICROARCH.md); only quoted when this run is the workload that was profiled.
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")      # collected on the bf16-gradient variant
    default_cfg = (args.vocab == 200_000_000 and args.emb_dim == 80 and args.batch == 16384 and args.fields == 26
                   and args.dist == "uniform" and not args.split_state and world == 1 and args.mlp_dtype == "bf16")
    if default_cfg and os.path.exists(pmc_path):
        traffic = json.load(open(pmc_path)).get("apply_main_adam", {}).get("total_bytes")
    # streaming-copy ceiling of this box, measured in this run (outside the timed region): 1 GiB device-to-device
    copy_gbps = None
    if rank == 0:
        src = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbps = round(5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        del src, dst
    out = {
        "metric": "samples/sec Wide&Deep Criteo batch16384",
        "value": round(args.batch * world * args.steps / dt, 1),
        "unit": "samples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"Wide&Deep Criteo (BASELINE configs[1]): vocab {args.vocab}, dim {args.emb_dim}, "
                               f"batch {args.batch}/GPU, {args.fields} fields, fp32 tables ({'split' if args.split_state else 'fused-row'} "
                               f"state layout), {args.dist} ids{', hash tables keyed by id (dynamic_embedding)' if args.dynamic_embedding else ''}"
                               f"{f', tables in host DRAM behind a {args.host_cache_rows}-row device cache' if args.host_cache_rows else ''}, "
                               f"MLP {cfg.field_size * cfg.emb_dim}-1024-512-256-128-1 in {args.mlp_dtype}",
                   "global_batch": args.batch * world, "id_dist": args.dist, "hip_graphs": graphs_used, "unique_frac": round(U / max(n_apply, 1), 4),
                   "parallelism": "1 GPU" if world == 1 else f"tables row-sharded x{world} (RCCL all-to-all), MLP dp{world}"},
        "roofline": {"bound": "hbm", "kernel": "k_apply_main<4,int,UpdAdam,%s> (fused segment-sum + LazyAdam row update)" % ("bf16_t" if bf16_io else "float"),
                     "row_gradient_dtype": "bf16" if bf16_io else "f32",
                     "achieved": round(achieved, 1), "peak": peak, "unit": "GB/s", "frac": round(achieved / peak, 4),
                     "traffic": traffic, "algorithmic_bytes": by["apply_deep"], "avg_ms": round(apply_ms, 5),
                     "measured_copy_gbps": copy_gbps,
                     "timing": "HIP events around k_apply_main on its launch stream (mrec_profile_next_apply), "
                               "averaged over the timed steps"},
        "kernels_ms": {k: round(sum(v) / len(v), 5) for k, v in sorted(kern_ms.items())},
        "embed_gbps": {
            "lookup": round(by["lookup"] / (median(kern_ms["gather_deep"]) * 1e-3) / 1e9, 1) if (world == 1 and "gather_deep" in kern_ms) else None,
            "apply_deep": round(achieved, 1),
        },
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
