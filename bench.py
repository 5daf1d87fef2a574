#!/usr/bin/env python3
"""bench.py -- Wide&Deep training throughput on the MI355X embedding path.

    python bench.py --gpus N --steps K --warmup W          (N > 1: spawns N ranks itself, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one synthetic Criteo-shaped batch that is already resident
in HBM: id dedup + inverted index, deep/wide lookups, the 5-layer MLP forward/backward (hand-written MFMA
kernels), the fused segment-sum + LazyAdam apply on the deep table, the FTRL apply on the wide table and the
dense optimizer (mindrec_amd/wide_deep.py; reference models/wide_deep/src/wide_and_deep.py:472-492).
Workload = BASELINE.json configs[1]: vocab 200 M, dim 80, batch 16384 per GPU, 26 categorical slots,
fp32 tables.  Rank 0 prints ONE JSON line.

Timing: W untimed warmup steps, then R (--repeats, default 5) blocks of EXACTLY K steps, each bracketed by a
barrier + synchronize on both sides and reduced with MAX over the ranks; `ms_per_step` / `value` are the MEDIAN
block, min / max are reported beside it.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps steps; the median block is reported")
    ap.add_argument("--prime-steps", type=int, default=300, help="untimed steps of setup in front of --warmup (graphs captured, clocks settled)")
    ap.add_argument("--vocab", type=int, default=200_000_000)
    ap.add_argument("--emb-dim", type=int, default=80)
    ap.add_argument("--batch", type=int, default=16384, help="per-GPU batch (reference passes batch_size per worker)")
    ap.add_argument("--fields", type=int, default=26, help="26 = north-star categorical slots; 39 = reference field_size")
    ap.add_argument("--dist", default="uniform", choices=["uniform", "zipf"])
    ap.add_argument("--mlp-dtype", default="fp16", choices=["fp16", "bf16", "fp32"],
                    help="fp16 = the reference's own mixed precision (use_mixed_precision, wide_and_deep.py:119-128); bf16 runs the same kernels")
    ap.add_argument("--dropout", action="store_true", help="dropout_flag: True as in benchmarks/wide_deep/default_config.yaml:15 (Dropout(0.5) on every "
                    "DenseLayer input; models/wide_deep/default_config.yaml:27, the configuration of configs[1], has it off)")
    ap.add_argument("--sink-size", type=int, default=5, help="training steps per host call (the reference's dataset_sink_mode / sink_size: "
                    "train_and_eval.py:98-101, train_and_eval_distribute.py:115-116); with the whole-step graph a sink is ONE graph launch "
                    "(sinks of 1 / 2 / 3 / 5 / 6 / 10 / 15 / 30 steps measured on one box, round 5: 0.636 / 0.636 / 0.631 / 0.631-0.636 / 0.633 / "
                    "0.639 / 0.642 / 0.640 ms per step -- the host enqueues the next sink while this one runs; longer graphs are no faster)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="budget of each cpu_baseline leg")
    ap.add_argument("--n-batches", type=int, default=4, help="distinct resident batches cycled through")
    ap.add_argument("--split-state", action="store_true", help="p, m, v as three separate arrays (default: fused rows)")
    ap.add_argument("--dynamic-embedding", action="store_true", help="hash tables keyed by the raw ids (reference --dynamic_embedding=True); "
                    "use a --vocab small enough for --hash-capacity, e.g. --vocab 3000000")
    ap.add_argument("--hash-capacity", type=int, default=1 << 22)
    ap.add_argument("--host-cache-rows", type=int, default=0, help="tables in pinned host DRAM behind a device cache of this many rows "
                    "(the reference's vocab_cache_size); keep --vocab x 976 B within the host's RAM")
    ap.add_argument("--graphs", default="step", choices=["step", "front", "mlp", "none"],
                    help="what replays as HIP graphs: the whole step (default; sinks of --sink-size steps as one graph), everything in front "
                    "of the optimizers, the dense net only, or nothing (kernel by kernel)")
    ap.add_argument("--shard-protocol", action="store_true", help="one GPU: run the row-shard protocol (routing kernels + RCCL collectives "
                    "that talk to themselves) -- what a rank of an N-GPU job does besides moving bytes over xGMI")
    ap.add_argument("--shard-uniques", type=float, default=0.0, help="row shards: exchange the batch's UNIQUE ids (one fp32 row / one summed "
                    "gradient row per unique id) with room for this many unique ids per position (Criteo-like ids: 0.3); 0: one 16-bit "
                    "row per position")
    ap.add_argument("--capacity-factor", type=float, default=None, help="row shards: request slots per owner = ceil(factor * ids / ranks); "
                    "default 1.05 for uniform ids (an owner's share of 425 984 uniform ids is within 0.6 %% of the mean at 12 sigma for 8 ranks), "
                    "1.25 otherwise (the engine's default); a run that drops a position is refused")
    ap.add_argument("--comm-timeout", type=int, default=180, help="seconds after which a rendezvous or a collective that a peer never joined "
                    "raises instead of hanging")
    ap.add_argument("--stamps", default="roofline", choices=["roofline", "always", "never"],
                    help="the kernels' own wall-clock stamps (they cost the step ~6 us, profiles/r05_stamps_ab.txt): 'roofline' (default) = off in "
                    "the timed blocks `value` comes from, on in --stamp-blocks extra blocks of the same steps right behind them, which the "
                    "roofline figures come from; 'always' = on throughout (what the runs under rocprofv3 that calibrate the clock offsets use)")
    ap.add_argument("--stamp-blocks", type=int, default=2)
    ap.add_argument("--no-zipf39", action="store_true", help="skip the secondary Criteo-like measurement (Zipf ids, 39 fields) behind the timed region")
    args = ap.parse_args()
    if args.capacity_factor is None:
        args.capacity_factor = 1.05 if args.dist == "uniform" else 1.25
    return args


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if xs else float("nan")


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start N ranks (one per GPU) with
    torch.distributed.run as a CHILD process -- this process has made no GPU call yet and makes none -- relay the
    ranks' output (rank 0 prints the JSON line) and exit with the child's code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env)
    raise SystemExit(proc.returncode)


def _usable_cores():
    """Cores this process may actually use: the scheduler affinity, cut by the container's CPU quota (a GPU box gives a job
    16 cores of a 256-thread host through cgroup cpu.max: threads beyond the quota only take turns)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline(args, seconds):
    """CPU legs on the GPU box's host cores, each on a bounded sample of the same workload (same batch shape, table
    scaled down to fit host RAM: random-row bandwidth is insensitive to V once V*D*4 >> LLC):
      port_1t   the oracle (oracle/mrec_oracle.c, scalar C) on ONE thread: lookup + wide sum + sparse LazyAdam + sparse FTRL
      port_mt   the same restatement on all host threads (the *_mt entries; bit-identical results)
      torch_cpu an independent second opinion: torch.unique + index_select + index_add_ + vectorised Adam/FTRL row updates
      mlp       the dense net of the same step (fwd + bwd, fp32) through torch's CPU GEMMs
    `value` = whole step = port_mt embedding path + mlp.  MindSpore's own CPU path cannot be timed: not installable here."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    from oracle import oracle as O
    ncpu = _usable_cores()
    threads = min(ncpu, 64)
    V = min(args.vocab, 2_000_000)
    D, B, Fd = args.emb_dim, args.batch, args.fields
    rng = np.random.default_rng(1000)
    p = (rng.standard_normal((V, D)) * 0.01).astype(np.float32)
    m = np.zeros_like(p); v = np.zeros_like(p)
    w = (rng.standard_normal((V, 1)) * 0.01).astype(np.float32); wa = np.ones_like(w); wl = np.zeros_like(w)
    g = rng.standard_normal((B * Fd, D)).astype(np.float32)
    gw = rng.standard_normal((B * Fd, 1)).astype(np.float32)
    wts = np.ones((B, Fd), np.float32)

    def oracle_leg(th, budget):
        steps, t0 = 0, time.perf_counter()
        while True:
            ids = rng.integers(0, V, size=(B, Fd)).astype(np.int32)
            O.gather_rows(p, ids, wts, threads=th)
            O.wide_sum(w, ids, wts, 0.0, threads=th)
            O.sparse_lazy_adam(p, m, v, ids, g, wts, grad_scale=1 / 1024, threads=th)
            O.sparse_ftrl(w, wa, wl, ids, gw, None, grad_scale=1 / 1024, threads=th)
            steps += 1
            if time.perf_counter() - t0 > budget or steps >= 200:
                break
        return (time.perf_counter() - t0) / steps, steps

    t_1t, n_1t = oracle_leg(0, seconds)
    t_mt, n_mt = oracle_leg(threads, seconds)

    # second opinion: PyTorch-CPU restatement of the same embedding path
    torch.set_num_threads(threads)
    tp, tm, tv = torch.from_numpy(p), torch.from_numpy(m), torch.from_numpy(v)
    tw, twa, twl = torch.from_numpy(w), torch.from_numpy(wa), torch.from_numpy(wl)
    tg, tgw = torch.from_numpy(g), torch.from_numpy(gw)
    steps, t0 = 0, time.perf_counter()
    while True:
        ids = torch.from_numpy(rng.integers(0, V, size=(B * Fd,)).astype(np.int64))
        emb = tp.index_select(0, ids)                                           # lookup
        _ = tw.index_select(0, ids).view(B, Fd).sum(dim=1)                      # wide sum
        uq, inv = torch.unique(ids, return_inverse=True)
        gs = torch.zeros((uq.numel(), D)).index_add_(0, inv, tg).mul_(1 / 1024)  # RowTensor dedup
        mm = tm.index_select(0, uq).mul_(0.9).add_(gs, alpha=0.1)
        vv = tv.index_select(0, uq).mul_(0.999).addcmul_(gs, gs, value=0.001)
        tp.index_copy_(0, uq, tp.index_select(0, uq) - 3.5e-4 * mm / (vv.sqrt() + 1e-8))
        tm.index_copy_(0, uq, mm); tv.index_copy_(0, uq, vv)
        gws = torch.zeros((uq.numel(), 1)).index_add_(0, inv, tgw).mul_(1 / 1024)
        a0 = twa.index_select(0, uq); a1 = a0 + gws * gws
        ln = twl.index_select(0, uq) + gws - (a1.sqrt() - a0.sqrt()) / 5e-2 * tw.index_select(0, uq)
        tw.index_copy_(0, uq, (ln.clamp(-1e-8, 1e-8) - ln) / (a1.sqrt() / 5e-2 + 2e-8))
        twa.index_copy_(0, uq, a1); twl.index_copy_(0, uq, ln)
        steps += 1
        del emb
        if time.perf_counter() - t0 > seconds or steps >= 200:
            break
    t_torch, n_torch = (time.perf_counter() - t0) / steps, steps

    # the dense net of the same step (fwd + bwd, fp32) on the host's cores through torch's CPU GEMMs
    dims = [Fd * D, 1024, 512, 256, 128, 1]
    Ws = [(torch.randn(dims[i], dims[i + 1]) * 0.01).requires_grad_(True) for i in range(5)]
    bs = [torch.zeros(dims[i + 1], requires_grad=True) for i in range(5)]
    x = torch.randn(B, Fd * D, requires_grad=True)
    y = (torch.rand(B, 1) < 0.25).float()
    t_mlp, reps, t0 = 0.0, 0, time.perf_counter()
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
        if time.perf_counter() - t0 > 1.5 * seconds:
            break
    t_mlp = t_mlp / max(reps, 1)
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    by = (B * Fd) * 4 + 2 * (B * Fd) * D * 4 + (B * Fd) * 4 + (B * Fd) * D * 4 + 6 * (B * Fd) * D * 4      # lookup + apply, U = N
    return {"value": round(B / (t_mt + t_mlp), 1) if reps else round(B / t_mt, 1), "unit": "samples/s", "cores": threads,
            "kind": "port", "host_cpu": model, "host_cores_usable": ncpu, "host_threads": os.cpu_count(),
            "legs": {
                "port_1t": {"samples_per_s": round(B / t_1t, 1), "ms": round(t_1t * 1e3, 1), "cores": 1, "steps": n_1t,
                            "embed_gbps": round(by / t_1t / 1e9, 2)},
                "port_mt": {"samples_per_s": round(B / t_mt, 1), "ms": round(t_mt * 1e3, 1), "cores": threads, "steps": n_mt,
                            "embed_gbps": round(by / t_mt / 1e9, 2)},
                "torch_cpu": {"samples_per_s": round(B / t_torch, 1), "ms": round(t_torch * 1e3, 1), "cores": threads, "steps": n_torch,
                              "embed_gbps": round(by / t_torch / 1e9, 2)},
                "mlp_torch_cpu_fp32": {"ms": round(t_mlp * 1e3, 1) if reps else None, "cores": threads, "steps": reps}},
            "sample": f"embedding path (lookup + wide_sum + sparse LazyAdam + sparse FTRL) at batch {B}x{Fd}, dim {D}, table scaled to "
                      f"V={V}, uniform ids, a fresh batch per step; value = all-core oracle leg + torch-CPU MLP {dims[0]}-1024-512-256-128-1 "
                      f"fwd+bwd; MindSpore CPU not installable here"}


def zipf39_line(args, eng, dev, peak, clock_off=None):
    """The embedding path under Criteo-like ids (Zipf, 39 fields) through a second engine sharing the first one's tables."""
    import torch
    from mindrec_amd import ops
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, embedding_bytes, synthetic_batch
    cfg = WideDeepConfig(vocab_size=args.vocab, emb_dim=args.emb_dim, field_size=39, batch_size=args.batch, mlp_dtype=args.mlp_dtype,
                         graphs=args.graphs, dropout_flag=args.dropout)
    e2 = WideDeepEngine(cfg, dev, tables_from=eng)
    batches = [synthetic_batch(cfg, dev, "zipf", seed=2000 + i) for i in range(4)]
    S = max(1, args.sink_size)
    for i in range(5):
        e2.train_step(*batches[i % 4])
    e2.train_steps([batches[j % 4] for j in range(S)])
    torch.cuda.synchronize()
    steps = 4 * S
    t0 = time.perf_counter()
    for i in range(0, steps, S):
        e2.train_steps([batches[(i + j) % 4] for j in range(S)])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    st = e2._step_state.embed_ms(range(e2.step_count - steps + 1, e2.step_count + 1))
    main_ms = e2._step_state.apply_ms(range(e2.step_count - steps + 1, e2.step_count + 1))
    plan = e2.last_plan
    U, n = plan.U, plan.n
    by = embedding_bytes(n, U, args.emb_dim, act_bytes=2)
    apply_b = by["apply_deep"] + U * 24 + args.batch * 4
    lookup_b = by["lookup"] + U * 4 + n * 4
    off = clock_off or {"apply_main_us": 0.0, "lookup_us": 0.0}      # dispatch overhead by rocprofv3's clock (see _measure)
    a_st, l_st, all_st = sum(main_ms) / len(main_ms), sum(a for a, _ in st) / len(st), sum(b for _, b in st) / len(st)
    a_ms, l_ms, all_ms = a_st + off["apply_main_us"] * 1e-3, l_st + off["lookup_us"] * 1e-3, all_st + off["apply_main_us"] * 1e-3
    del e2
    return {"workload": f"same tables, batch {args.batch} x 39 fields, Zipf(1.05) ids per slot + the 13 constant dense-field ids", "unique_frac": round(U / n, 4),
            "ms_per_step": round(ms, 4), "samples_per_s": round(args.batch / ms * 1e3, 1),
            "kernel": "k_apply_main (dominant)", "algorithmic_bytes": apply_b, "avg_ms": round(a_ms, 5),
            "achieved": round(apply_b / (a_ms * 1e-3) / 1e9, 1), "peak": peak, "unit": "GB/s", "frac": round(apply_b / (a_ms * 1e-3) / 1e9 / peak, 4),
            "embedding_path": {"algorithmic_bytes": lookup_b + apply_b, "lookup_ms": round(l_ms, 5), "apply_ms_incl_finishing_kernel": round(all_ms, 5),
                               "frac": round((lookup_b + apply_b) / ((l_ms + all_ms) * 1e-3) / 1e9 / peak, 4),
                               "frac_stamps": round((lookup_b + apply_b) / ((l_st + all_st) * 1e-3) / 1e9 / peak, 4),
                               # begin of k_apply_main -> the LAST workgroup of the finishing pass that had a run to finish (round 5:
                               # until round 4 only the last-dispatched 256 workgroups raised the end stamp, which missed the ones that
                               # finish the 13 constant ids' 16384-entry runs -- first in the grid, last to end: ~29 us of this)
                               "finishing_pass_ms": round(all_st - a_st, 5)},
            "timing": f"in-graph kernel stamps over {len(st)} steps behind the timed region" + (" + the dispatch overhead by rocprofv3's clock" if clock_off else "")}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                   # never returns
    # RCCL / cross-process tensor sharing on this pool needs dmabuf IPC (the driver image exports this already)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1 or args.shard_protocol:
        # A shard's step lives on several streams (main, the side stream of the request exchange, RCCL's own); with the HIP
        # runtime's default of 4 hardware queues they collided on one queue and ran back to back: 1.52 -> 1.21 ms per step on
        # one GPU talking to itself.  (Must be set before the runtime starts; the one-GPU graph path is 0.6 % slower with 8.)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist
    from mindrec_amd import _lib, ops
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, embedding_bytes, synthetic_batch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (mindrec_amd has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rc = _lib.lib().mrec_device_ok()
    if rc != 0:
        raise SystemExit(f"libmrec_hip.so cannot run on this device: {_lib.lib().mrec_strerror(rc).decode()}")
    if world > 1 or args.shard_protocol:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        # "nccl" IS RCCL on ROCm.  The timeout bounds the rendezvous AND every later collective: a rank that died leaves its
        # peers with an error after --comm-timeout seconds instead of a hang (the launcher then takes the job down, exit code != 0)
        import datetime
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=args.comm_timeout))

    eng = None
    try:
        out = _measure(args, world, rank, dev)
    except BaseException as e:      # noqa: BLE001  (SystemExit included: every exit path releases the graphs that hold RCCL kernels)
        if world > 1 or args.shard_protocol:
            print(f"[bench rank {rank}] {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        raise
    finally:
        if world > 1 or args.shard_protocol:
            try:
                for e_ in list(_ENGINES):
                    e_.release_graphs()           # graphs that hold RCCL kernels must go before the process group does
                _ENGINES.clear()
                dist.destroy_process_group()
            except Exception as e:      # noqa: BLE001
                print(f"[bench rank {rank}] teardown: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
    if rank == 0:
        # the LAST line of stdout: whatever the communication library has printf'd into the C runtime's buffer (RCCL announces
        # the path it was loaded from) comes out first
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


_ENGINES = []


def _measure(args, world, rank, dev):
    import torch
    import torch.distributed as dist
    from mindrec_amd import ops
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, embedding_bytes, synthetic_batch
    cfg = WideDeepConfig(vocab_size=args.vocab, emb_dim=args.emb_dim, field_size=args.fields, batch_size=args.batch,
                         mlp_dtype=args.mlp_dtype, fused_state=not args.split_state, graphs=args.graphs,
                         dynamic_embedding=args.dynamic_embedding, hash_capacity=args.hash_capacity,
                         host_cache_rows=args.host_cache_rows, shard_capacity_factor=args.capacity_factor, dropout_flag=args.dropout,
                         shard_unique_factor=args.shard_uniques)
    eng = WideDeepEngine(cfg, dev, rank=rank, world=world, shard_protocol=args.shard_protocol)
    _ENGINES.append(eng)
    batches = [synthetic_batch(cfg, dev, args.dist, seed=1000 + i, rank=rank) for i in range(args.n_batches)]
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    S = max(1, args.sink_size)            # (a sink is ONE graph of whole steps wherever the step has a graph: one GPU, or a shard over RCCL)

    def run_steps(n, timers=None):
        """n training steps over the resident batches, S per host call where S > 1 (a remainder step by step)."""
        i = 0
        while i < n:
            if S > 1 and n - i >= S and eng._step_graph is not None:       # (a sink is a graph of whole steps: where the step has one)
                eng.train_steps([batches[(i + j) % len(batches)] for j in range(S)])
                i += S
            else:
                if timers is not None:
                    eng.deep_apply_timer = timers[i]      # armed by the engine right before the deep-table LazyAdam launch
                eng.train_step(*batches[i % len(batches)])
                i += 1

    # Priming (setup, whatever --warmup says): workspaces and the HIP graphs come into being in the engine's first steps.
    for i in range(5):
        eng.train_step(*batches[i % len(batches)])
    if S > 1:
        run_steps(S)
    barrier()
    # ... and the chip's clocks and the host's code paths settle: on a fresh box the first timed block of a process ran up to
    # 45 % slower than its fourth (min / max of the blocks: 0.633 / 0.932 ms).  A fixed count, the same on every rank.
    run_steps(args.prime_steps)
    barrier()
    ss_ = getattr(eng, "_step_state", None)
    # (one GPU only: with ranks, a rank whose capture failed would run a different number of steps -- and collectives -- than its peers)
    stamps_split = bool(args.stamps == "roofline" and ss_ is not None and eng._step_graph is not None and world == 1)
    if ss_ is not None and (stamps_split or args.stamps == "never"):
        ss_.set_stamps(False)
    run_steps(args.warmup)
    barrier()
    block_s, kmain = [], []
    for r in range(max(1, args.repeats)):
        ktimers = [ops.KernelTimer() for _ in range(args.steps)]     # HIP events around k_apply_main only
        barrier()
        t0 = time.perf_counter()
        run_steps(args.steps, ktimers)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        block_s.append(dt)
        if eng._step_graph is None:
            kmain += [t.ms() for t in ktimers]
    embed_stamps = []
    apply_timing = ("HIP events around k_apply_main on its launch stream (mrec_profile_next_apply), "
                    "averaged over the {n} timed steps")
    n_stamped_blocks = len(block_s)
    stamped_ms_per_step = None
    if stamps_split:
        # the same steps again, stamped: the blocks the roofline figures are read from (not `value`)
        ss_.set_stamps(True)
        run_steps(args.steps)                      # (untimed: the first stamped sink)
        n_stamped_blocks = max(1, args.stamp_blocks)
        sb = []
        for r in range(n_stamped_blocks):
            barrier()
            t0 = time.perf_counter()
            run_steps(args.steps)
            barrier()
            sb.append(time.perf_counter() - t0)
        stamped_ms_per_step = round(median(sb) / args.steps * 1e3, 4)
    if eng._step_graph is not None and args.stamps != "never":
        # The whole step is one HIP graph: events recorded inside a captured graph cannot be timed on this stack (hipError 400,
        # tools/probes/graph_event_probe.py), so the kernel stamps the device wall clock itself -- first workgroup in, last wave
        # out -- into a ring in the device-side step state, read here after the timed region.
        last = eng.step_count
        n_timed = min(n_stamped_blocks * args.steps, ops.StepState.RING - 1)
        kmain = eng._step_state.apply_ms(range(last - n_timed + 1, last + 1))
        embed_stamps = eng._step_state.embed_ms(range(last - n_timed + 1, last + 1))      # (lookup, apply incl. k_apply_long) per step
        apply_timing = ("device wall-clock stamps written by k_apply_main itself (first workgroup begin -> last wave end; the step "
                        "is one HIP graph, whose event nodes cannot be timed), averaged over the last {n} timed steps"
                        + (f" -- of {n_stamped_blocks} extra blocks of the same steps run right behind the blocks `value` comes from, with the "
                           f"stamps switched on ({stamped_ms_per_step} ms per step there); `value`'s blocks run without them" if stamps_split else ""))
    dt = median(block_s)
    graphs_used = {"step": eng._step_graph is not None, "front": eng._front_graph is not None, "mlp": eng._mlp_graph is not None,
                   "sink_size": S if (any(k[0] == S and v for k, v in eng._sink_graphs.items()) and args.steps >= S) else 1}

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
    # the same kernel between two HIP events, in four more eager steps (a cross-check of the stamps; phase timers still on, so
    # the step runs kernel by kernel)
    ev_ms = []
    for i in range(4):
        t = ops.KernelTimer()
        eng.deep_apply_timer = t
        eng.train_step(*batches[i % len(batches)])
        torch.cuda.synchronize()
        try:
            ev_ms.append(t.ms())
        except RuntimeError:
            pass
    eng.timers = None
    plan = eng.last_plan
    U = plan.U                                  # unique ids of the last step's (local) apply
    n_apply = plan.n
    io16 = eng._mfma                            # gather writes / apply reads 16-bit rows (also on the wire)
    by = embedding_bytes(n_apply, U, args.emb_dim, act_bytes=2 if io16 else 4)
    # Folded wide branch (one-GPU fused rows [p | w accum linear | m | v]): the wide FTRL runs inside the deep apply and the
    # wide lookup inside the deep gather.  Their algorithmic bytes join those kernels' figures: the apply reads and writes the
    # 12-byte record of each touched row (U * 24) and reads one logit gradient per sample (B * 4); the gather reads the
    # wide word of each row once (U * 4) and writes one product per position (N * 4; ids and weights are already counted).
    fold = bool(eng._fold_wide and world == 1)
    apply_bytes = by["apply_deep"] + (U * 24 + args.batch * 4 if fold else 0)
    lookup_bytes = by["lookup"] + (U * 4 + n_apply * 4 if fold else 0)
    if not kmain:                 # (--stamps never on a whole-step graph: only the eager cross-check is left)
        kmain = ev_ms or [float("nan")]
        apply_timing = "HIP events around k_apply_main in {n} eager extra steps (--stamps never)"
    apply_ms_stamps = sum(kmain) / len(kmain)
    # The profiler's clock.  rocprofv3 times a dispatch from the command processor's begin to its end signal; the kernels' own
    # stamps (first workgroup in -> last wave out) leave out the few microseconds in front of the first workgroup and behind the
    # last wave.  That difference is a property of the launch, measured where BOTH clocks saw the same launches -- this command
    # under `rocprofv3 --kernel-trace --stats` (tools/clock_offsets.py: rocprof's average duration - the stamps' average, per
    # kernel, committed as profiles/rNN_clock_offsets.json) -- and added here, so that `roofline.frac` is what the profiler's
    # clock would read on THIS box; the stamps stay as secondary fields.
    clock_off, clock_src = {"apply_main_us": 0.0, "lookup_us": 0.0}, None
    if eng._step_graph is not None:
        for rnd in ("r05", "r04"):
            cp = os.path.join(ROOT, "profiles", f"{rnd}_clock_offsets.json")
            if os.path.exists(cp):
                c_ = json.load(open(cp))
                clock_off = {k: float(c_[k]) for k in clock_off}
                clock_src = f"profiles/{rnd}_clock_offsets.json"
                break
    apply_ms = apply_ms_stamps + clock_off["apply_main_us"] * 1e-3
    achieved = apply_bytes / (apply_ms * 1e-3) / 1e9
    peak = 8000.0
    # HBM bytes per launch of the dominant kernel come from the committed PMC passes (profiles/: rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE in separate runs of this same command, corrected per MI355X_MICROARCH.md) -- NOT measured in this run;
    # quoted only when this run is the workload that was profiled, and labelled with their source.
    traffic, traffic_source = None, None
    default_cfg = (args.vocab == 200_000_000 and args.emb_dim == 80 and args.batch == 16384 and args.fields == 26
                   and args.dist == "uniform" and not args.split_state and world == 1 and args.mlp_dtype in ("bf16", "fp16")
                   and not args.shard_protocol)
    for name in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        pmc_path = os.path.join(ROOT, "profiles", name)
        if default_cfg and os.path.exists(pmc_path):
            traffic = json.load(open(pmc_path)).get("apply_main_adam", {}).get("total_bytes")
            traffic_source = f"profiles/{name} (separate rocprofv3 --pmc passes of this command; not measured in this run)"
            break
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

    def med_ms(name):
        return median(kern_ms[name]) if name in kern_ms and kern_ms[name] else None

    lookup_ms, wide_ms, wapply_ms = med_ms("gather_deep"), med_ms("wide_sum"), med_ms("apply_wide")
    dt_name = {"bf16": "bf16", "fp16": "f16", "fp32": "f32"}[args.mlp_dtype]
    out = {
        "metric": "samples/sec Wide&Deep Criteo batch16384",
        "value": round(args.batch * world * args.steps / dt, 1),
        "unit": "samples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "repeats": len(block_s),
        "ms_per_step_min": round(min(block_s) / args.steps * 1e3, 4),
        "ms_per_step_max": round(max(block_s) / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": dt_name,
        "data": "synthetic",
        "config": {"workload": f"Wide&Deep Criteo (BASELINE configs[1]): vocab {args.vocab}, dim {args.emb_dim}, "
                               f"batch {args.batch}/GPU, {args.fields} fields, fp32 tables and fp32 optimizer state ({'split' if args.split_state else 'fused-row'} "
                               f"layout), {args.dist} ids{', hash tables keyed by id (dynamic_embedding)' if args.dynamic_embedding else ''}"
                               f"{f', tables in host DRAM behind a {args.host_cache_rows}-row device cache' if args.host_cache_rows else ''}, "
                               f"MLP {cfg.field_size * cfg.emb_dim}-1024-512-256-128-1 in {args.mlp_dtype} "
                               f"({'hand-written MFMA kernels' if eng._mfma else (f'hand-written fp32 DenseLayers, MatMuls: {cfg.fp32_matmul}' if getattr(eng, '_f32net', False) else 'torch GEMMs')}; looked-up rows and row gradients in {dt_name})"
                               f"{f', {S} steps per host call (sink_size)' if graphs_used.get('sink_size', 1) > 1 else ''}"
                               f"{', Dropout(0.5) on every DenseLayer input' if args.dropout else ''}",
                   "kernel_stamps": ("off in the timed blocks; on in %d extra blocks of the same steps behind them (%s ms per step), which the roofline "
                                     "figures are read from" % (n_stamped_blocks, stamped_ms_per_step)) if stamps_split else args.stamps,
                   "global_batch": args.batch * world, "id_dist": args.dist, "hip_graphs": graphs_used, "unique_frac": round(U / max(n_apply, 1), 4),
                   "parallelism": ("1 GPU" + (", row-shard protocol over RCCL with itself" if args.shard_protocol else "")) if world == 1 else f"tables row-sharded x{world} (RCCL all-to-all), MLP dp{world}"},
        "roofline": {"bound": "hbm",
                     "kernel": ("k_apply_main<4,int,UpdAdam,%s,WIDE> (segment-sum + LazyAdam row update + FTRL on the row's wide record)" if fold else
                                "k_apply_main<4,int,UpdAdam,%s> (fused segment-sum + LazyAdam row update)") % (dt_name + "_t" if io16 else "float"),
                     "row_gradient_dtype": dt_name if io16 else "f32", "wide_folded": fold,
                     "achieved": round(achieved, 1), "peak": peak, "unit": "GB/s", "frac": round(achieved / peak, 4),
                     "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes": apply_bytes, "avg_ms": round(apply_ms, 5),
                     "clock": ("rocprofv3's: the kernel's own stamps + the dispatch overhead measured under rocprofv3 (" + clock_src + ")") if clock_src
                              else "the kernel's own stamps / HIP events (no committed clock offsets)",
                     "avg_ms_stamps": round(apply_ms_stamps, 5), "frac_stamps": round(apply_bytes / (apply_ms_stamps * 1e-3) / 1e9 / peak, 4),
                     "avg_ms_hip_events_eager": round(sum(ev_ms) / len(ev_ms), 5) if ev_ms else None,
                     "measured_copy_gbps": copy_gbps,
                     "timing": apply_timing.format(n=len(kmain))},
        # medians over n eager extra steps behind the timed region (HIP events around each phase on the main stream: a phase that only
        # ENQUEUES work on the side stream -- the plan -- is not listed; one cold outlier does not move a median)
        "kernels_ms": {k: {"median": round(median(v), 5), "n": len(v)} for k, v in sorted(kern_ms.items()) if v and k != "plan"},
    }
    lookup_timing = "torch events around the gather in eager extra steps after the timed region"
    apply_all_ms = None
    lookup_ms_stamps = apply_all_stamps = None
    if fold and embed_stamps:
        # in-graph truth: both kernels stamp the device wall clock themselves in every timed step (+ the dispatch overhead, above)
        lookup_ms_stamps = sum(a for a, _ in embed_stamps) / len(embed_stamps)
        apply_all_stamps = sum(b for _, b in embed_stamps) / len(embed_stamps)
        lookup_ms = lookup_ms_stamps + clock_off["lookup_us"] * 1e-3
        apply_all_ms = apply_all_stamps + clock_off["apply_main_us"] * 1e-3
        lookup_timing = (f"device wall-clock stamps written by the lookup kernel itself inside the timed steps' graph (first workgroup begin -> last "
                         f"wave end of the last-dispatched workgroups), averaged over {len(embed_stamps)} timed steps")
    if world == 1 and lookup_ms:
        out["roofline_lookup"] = {"bound": "hbm", "kernel": "k_gather_rows (EmbeddingLookup, mask fused%s)" % (" + the row's wide word" if fold else ""),
                                  "achieved": round(lookup_bytes / (lookup_ms * 1e-3) / 1e9, 1),
                                  "peak": peak, "unit": "GB/s", "frac": round(lookup_bytes / (lookup_ms * 1e-3) / 1e9 / peak, 4),
                                  "algorithmic_bytes": lookup_bytes, "avg_ms": round(lookup_ms, 5), "timing": lookup_timing,
                                  "avg_ms_stamps": round(lookup_ms_stamps, 5) if lookup_ms_stamps else None}
        tot_b = tot_ms = None
        if fold:
            # both tables' lookup and apply are these kernels (the per-sample sum of the wide products is in the head kernel); the
            # apply's finishing kernel (k_apply_long: the runs that cross windows) and the launch gap in front of it are counted
            tot_b = lookup_bytes + apply_bytes
            tot_ms = lookup_ms + (apply_all_ms if apply_all_ms is not None else apply_ms)
        elif wide_ms and wapply_ms:
            tot_b = by["lookup"] + by["apply_deep"] + by["wide_lookup"] + by["apply_wide"]
            tot_ms = lookup_ms + apply_ms + wide_ms + wapply_ms
        if tot_b:
            out["roofline_embedding_path"] = {"what": "EmbeddingLookup + sparse apply, deep AND wide tables (north-star quantity)",
                                              "achieved": round(tot_b / (tot_ms * 1e-3) / 1e9, 1), "peak": peak, "unit": "GB/s",
                                              "frac": round(tot_b / (tot_ms * 1e-3) / 1e9 / peak, 4), "algorithmic_bytes": tot_b,
                                              "sum_ms": round(tot_ms, 5), "lookup_ms": round(lookup_ms, 5),
                                              "apply_ms_incl_finishing_kernel": round(apply_all_ms, 5) if apply_all_ms is not None else None,
                                              "timing": ("in-graph kernel stamps + the dispatch overhead of both kernels by rocprofv3's clock (" + str(clock_src) + ")")
                                                        if apply_all_ms is not None else "events in eager extra steps; k_apply_long not counted",
                                              "frac_stamps": round(tot_b / ((lookup_ms_stamps + apply_all_stamps) * 1e-3) / 1e9 / peak, 4)
                                                             if apply_all_stamps is not None else None,
                                              "finishing_pass_ms": round(apply_all_stamps - apply_ms_stamps, 5) if apply_all_stamps is not None else None}
    if default_cfg and "roofline_embedding_path" in out:
        # The same quantities from rocprofv3's clock: average kernel durations of the committed `rocprofv3 --kernel-trace --stats`
        # run of this command (profiles/rNN_bench_kernel_summary.txt; NOT measured in this run).  The apply's finishing pass has no
        # kernel of its own since round 4 (it is the first workgroups of the dense Adam launch, k_finish_dense_adam).
        for rnd in ("r05", "r04"):
            sp = os.path.join(ROOT, "profiles", f"{rnd}_bench_kernel_summary.txt")
            if os.path.exists(sp):
                avg = {}
                for ln in open(sp):
                    for key in ("k_apply_main<", "k_gather_rows_w16<", "k_gather_rows<", "k_apply_long<", "k_finish_dense_adam<"):
                        if ln.startswith(key) and key not in avg:
                            try:
                                avg[key] = float(ln.split()[-3])
                            except (ValueError, IndexError):
                                pass
                ga = avg.get("k_gather_rows_w16<", avg.get("k_gather_rows<"))
                if "k_apply_main<" in avg and ga:
                    t_us = avg["k_apply_main<"] + ga + avg.get("k_apply_long<", 0.0)
                    # the finishing pass has no rocprof duration of its own: its length by this run's stamps (end of the finishing
                    # work - end of k_apply_main) is added for the figure that counts everything
                    fin_us = max(0.0, (apply_all_stamps - apply_ms_stamps) * 1e3) if (apply_all_stamps is not None and "k_apply_long<" not in avg) else 0.0
                    out["roofline_embedding_path"]["rocprof"] = {
                        "source": f"profiles/{rnd}_bench_kernel_summary.txt (separate rocprofv3 --kernel-trace --stats run of this command)",
                        "apply_main_us": avg["k_apply_main<"], "lookup_us": ga, "apply_long_us": avg.get("k_apply_long<"),
                        "finishing_pass": "inside k_finish_dense_adam (%.1f us with the dense Adam)" % avg["k_finish_dense_adam<"] if "k_finish_dense_adam<" in avg else "own kernel",
                        "frac_without_finishing_pass": round(out["roofline_embedding_path"]["algorithmic_bytes"] / (t_us * 1e-6) / 1e9 / peak, 4),
                        "finishing_pass_us_by_stamps": round(fin_us, 2),
                        "frac": round(out["roofline_embedding_path"]["algorithmic_bytes"] / ((t_us + fin_us) * 1e-6) / 1e9 / peak, 4),
                        "apply_main_frac": round(apply_bytes / (avg["k_apply_main<"] * 1e-6) / 1e9 / peak, 4),
                        "lookup_frac": round(lookup_bytes / (ga * 1e-6) / 1e9 / peak, 4)}
                break
    if world == 1 and eng._mfma:
        # exact HBM bytes of one k_dense_adam4_slabs launch (tools/pmc_summary.py checks its counter correction on these)
        n_el = eng.dense_flat.numel()
        slab_el = sum(t.numel() for t in list(eng._dw.values()) + list(eng._db.values()))
        covered = sum(t[0].numel() for t in list(eng._dw.values()) + list(eng._db.values()))
        out["dense_adam_bytes"] = {"read": 4 * (3 * n_el + slab_el + (n_el - covered)), "write": 4 * 3 * n_el + 2 * n_el}
    if (world == 1 and not args.shard_protocol and not args.no_zipf39 and fold and default_cfg and eng._step_graph is not None):
        # SURVEY 8(d) distribution (C): Criteo-like ids -- Zipf(1.05) per slot + the 13 constant dense-field ids of field_size 39
        # (process_data.py:138-147) -- on a second engine that trains on the SAME tables: the duplicate-heavy case of the same
        # kernels (U / N ~ 0.2), a secondary line, not `value`
        try:
            out["roofline_zipf39"] = zipf39_line(args, eng, dev, peak, clock_off if clock_src else None)
        except Exception as e:       # noqa: BLE001  (a secondary measurement must not take the line down)
            out["roofline_zipf39"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    if world > 1 or args.shard_protocol:
        # what a first multi-GPU run needs in order to be diagnosable from its one line
        uq = float(args.shard_uniques)
        N_ = args.batch * args.fields
        ns = eng.k.shard_capacity(max(int(N_ * uq), 1) if uq > 0 else N_, world, args.capacity_factor) * world
        W = eng.k.shard_msg_words(args.emb_dim, torch.float32 if uq > 0 else eng._act)[1]
        off = (world - 1) / world                  # a rank's own chunk never moves
        id_b = 8 if batches[0][0].dtype == torch.int32 else 16

        def model(nw, uqf, act):                   # bytes a rank puts on xGMI per step and direction at nw ranks (its own chunk stays)
            slots = eng.k.shard_capacity(max(int(N_ * uqf), 1) if uqf > 0 else N_, nw, args.capacity_factor) * nw
            Wm = eng.k.shard_msg_words(args.emb_dim, torch.float32 if uqf > 0 else act)[1]
            o_ = (nw - 1) / nw
            return {"slots_per_rank": slots, "request": int(slots * id_b * o_), "answer": int(slots * Wm * 4 * o_), "gradient": int(slots * Wm * 4 * o_)}
        out["rccl"] = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                       "own_chunk_bypass": bool(eng._bypass),          # did _exchange_selftest keep the per-peer-list exchange?
                       "whole_step_graph": eng._step_graph is not None,
                       "slots_per_rank": ns, "capacity_factor": args.capacity_factor,
                       "a2a_bytes_per_rank_per_step": {"request": int(ns * id_b * off), "answer": int(ns * W * 4 * off),
                                                       "gradient": int(ns * W * 4 * off)},
                       "exchange": "unique ids (fp32 rows / summed gradient rows)" if uq > 0 else "positions (16-bit rows)", "unique_factor": uq,
                       # the byte model of an 8-rank node, whatever this run's world size (at world 1 nothing moves): today's
                       # per-position exchange against the unique-level one at this run's unique ids per position
                       "a2a_bytes_model_n8": {"positions": model(8, 0.0, eng._act),
                                              "uniques": model(8, uq if uq > 0 else min(1.0, round(U / max(n_apply, 1), 4)), eng._act),
                                              "unique_ids_per_position": uq if uq > 0 else round(U / max(n_apply, 1), 4)},
                       "allreduce_bytes_per_step": int(eng.dense_grad_full.numel() * 4),
                       "kernels_ms_keys": ["route", "a2a_rows", "unroute", "a2a_grads", "allreduce_dense"]}
        # dropped positions: the count every rank holds is the SUM over all ranks (it rides the dense all-reduce), so all ranks
        # take the same branch here -- nobody is left inside a barrier
        eng.check_shard_overflow()
        out["config"]["shard_capacity"] = {"factor": args.capacity_factor, "dropped_positions": 0}
    eng.check_cache()           # host_cache_rows: a batch that did not fit the device cache is latched on the device
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
    if world > 1 or args.shard_protocol:
        dist.barrier()
    return out


if __name__ == "__main__":
    main()
