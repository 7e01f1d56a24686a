#!/usr/bin/env python3
"""Headline benchmark: CSA forward + backward query-points/sec (BASELINE.json metric) on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5                      # BASELINE.json configs[2] (the metric's config)
    python bench.py --same-work                                         # N = 1 with the per-GPU work of the N > 1 path (K+2 evaluations)
    python bench.py --config 2 | --config 5                             # configs[1] / configs[4] as workloads of their own
    python bench.py --gpus N --steps K --warmup W                      # N > 1 without a launcher: starts its own N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --heads 8 --K 4 --shapes 8                          # the reference's published geometry (get_csa_pred.py:35-36)

A "step" = one pass of the hot path over one batch of synthetic shapes, exactly what
MID-FC/csa_training.py:202-211 does per batch: model(x, neighbours) -> masked cross-entropy -> backward
(no optimizer step, no data loading; features resident in HBM).

Workloads (`--config`, numbering of SURVEY.md §8d = BASELINE.json configs[i-1]):
  3 (default)  32 query shapes x 10000 points x 256 channels, K = 3, 20 blocks of 500 — the configuration the metric is
               quoted on.  For N > 1 (config 4): every rank owns 32 shapes of a 32*N-shape collection (weak scaling),
               neighbours are drawn from the whole collection, the ranks exchange the point features over RCCL/xGMI inside the
               timed region, and the weight gradients are all-reduced at the end of the step.
  2            4 query shapes x 10000 x 256, K = 2.
  5            8 query shapes x 50000 points x 96 channels (d_k = d_v = 96), K = 4, 100 blocks of 500.
n_heads = 1 (csa_training.py:37 default), 39 classes (PartNet Chair), train mode (dropout 0.1 live, csa_training.py:192).

Prints ONE JSON line (rank 0).  EVERY launching entry point of the C ABI is bracketed by HIP events on the launch stream in
every timed step (csn_amd._lib.set_call_hook); `launches` lists them (mean ms per step), `roofline` describes the longest one
WHATEVER it is — today a fused block-attention launch (forward: S = Q K^T, softmax, P V; backward: dP = dO V^T, dS, dQ = dS K)
— and `roofline_other` the next attention launch.
`cpu_baseline` is the oracle's faithful op-for-op port of the reference timed on the host cores over a bounded sample.
"""
import argparse
import datetime
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"fp32": 157.3,       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4, 64 FLOP/clk/SIMD
               "bf16x3": 2500.0,    # dense bf16 MFMA; the mode issues 3 bf16 matrix FLOPs per algorithmic FLOP
               "bf16": 2500.0, "fp16": 2500.0}
PEAK_HBM_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E ~8 TB/s
MATH_MODES = {"fp32": 0, "bf16x3": 1, "bf16": 2, "fp16": 3}
DTYPE_TEXT = {"fp32": "f32", "bf16x3": "bf16x3 (fp32 operands split into 2 bf16, 3 bf16 MFMA per product, fp32 accumulate)",
              "bf16": "bf16 (one bf16 MFMA per product, fp32 accumulate; outside the 1e-4 contract)",
              "fp16": "fp16 forward / bf16 backward (one MFMA per product, fp32 accumulate; outside the 1e-4 contract)"}
N_CLS = 39
INIT_TIMEOUT_S = 180                 # process-group timeout: rendezvous and every collective of the run (a step is milliseconds)
LAUNCH_GRACE_S = 5.0                 # self-launch: seconds a rank gets between SIGTERM and SIGKILL when another rank has failed
RENDEZVOUS_RETRY_S = 20.0            # self-launch: rank 0 failing this soon after the launch = rendezvous (port taken): one retry
CONFIGS = {
    2: dict(B=4, K=2, N=10000, C=256, d=256, T=500, nb=20, name="BASELINE configs[1]"),
    3: dict(B=32, K=3, N=10000, C=256, d=256, T=500, nb=20, name="BASELINE configs[2]"),
    5: dict(B=8, K=4, N=50000, C=96, d=96, T=500, nb=100, name="BASELINE configs[4]"),
}
# HBM bytes per launch of the attention kernels, measured by separate rocprofv3 --pmc passes and summarised into this
# committed file by scripts/pmc_summary.py (never measured by the bench run itself: counters cannot be collected beside
# the timing); entries are keyed "<config>/<math>/<fwd|bwd>"
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "attn_hbm_traffic.json")


def algorithmic_flops_fwd(B, K, N, C, D, T):
    """unique matmul FLOPs of one forward (projections once per shape, 2K+1 evaluations)."""
    return B * ((K + 1) * 6 * N * C * D + (2 * K + 1) * (4 * N * T * D + 2 * N * D * C))


def masked_ce(logits, label):
    """csa_training.py:94-108 restated: mean cross-entropy over the points with label > 0 (label 0 = unlabelled is
    ignored; class-major logits are consumed in place instead of being transposed and gathered)."""
    from csn_amd.functional import masked_cross_entropy
    return masked_cross_entropy(logits, label, 0)[0]


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, sample_shapes, threads, dropout=True, H=1):
    """The oracle's port of the reference on the host, fwd + bwd, in the same mode as the headline run (train mode: both
    dropouts live, csa_training.py:192).  Reference geometry (20 x 500, d = 256): the faithful op sequence (chunk loop, index
    gather, growing cat, 2K+2 MHA calls); other geometries: the closed form of the same arithmetic."""
    from oracle import csa_oracle as orc
    torch.set_num_threads(threads)
    K, N, C, d, T, nb = (cfg[k] for k in ("K", "N", "C", "d", "T", "nb"))
    rng = np.random.default_rng(99)
    p = {k: v.requires_grad_(True) for k, v in orc.make_params(rng, H, d_model=C, d_k=d, d_v=d, n_cls=N_CLS, csa=True).items()}
    x = orc.synth_points(rng, (sample_shapes, C, N, 1))
    nb_ = orc.synth_points(rng, (sample_shapes, K + 1, C, N, 1))
    nb_[:, 0] = x
    lab = orc.synth_labels(rng, sample_shapes, N, N_CLS)
    pd = 0.1 if dropout else 0.0
    faithful = (N, C, d, T, nb) == (10000, 256, 256, 500, 20)
    if faithful:
        mha = lambda a, b, c, pp, h, **kw: orc.mha_faithful(a, b, c, pp, h, p_attn_drop=pd, p_out_drop=pd)
        what = "oracle/csa_oracle.py mha_faithful"
    else:
        mha = lambda a, b, c, pp, h, **kw: orc.mha_blockdiag(a, b, c, pp, h, d_k=d, d_v=d, block=T, n_blocks=nb)
        what = "oracle/csa_oracle.py mha_blockdiag (closed form, eval-mode arithmetic)"

    def step():
        for v in p.values():
            v.grad = None
        orc.masked_ce_loss(orc.forward_csa(x, nb_, p, H, mha=mha), lab).backward()

    step()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    t = sorted(times)[1]
    return {"value": sample_shapes * N / t, "unit": "points/s", "cores": threads, "cpu_model": cpu_model_name(), "kind": "port",
            "sample": f"{sample_shapes} of the {cfg['B']} query shapes (K={K}, {N} pts x {C} ch, "
                      f"{'train mode, dropout 0.1' if dropout and faithful else 'eval-mode arithmetic'}, fwd+bwd), median of 3 "
                      f"steps after 1 warm-up, {t:.2f} s/step; {what}"}


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: this process — BEFORE anything of it touches a GPU —
    starts N fresh copies of itself, one rank per GPU, with the environment `torch.distributed.run` would give them
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT on 127.0.0.1), relays rank 0's JSON line (every rank's stderr
    passes through) and exits non-zero if any rank does.  The launcher form of the docstring keeps working: with WORLD_SIZE
    already in the environment nothing is spawned.

    Failure path (a rank that dies while the others sit inside an RCCL collective would otherwise hold the run until the
    collective's own timeout): the parent POLLS all ranks; on the first non-zero exit — or when the parent itself is told to
    stop (SIGTERM / SIGINT, e.g. from `timeout`), or raises — the remaining ranks get SIGTERM, then SIGKILL after
    `LAUNCH_GRACE_S`, each in its own session so that the signal reaches whatever the rank started; the parent then exits
    non-zero within seconds.  Ranks are only ever fresh children: nothing that touched a GPU is re-executed.  The rendezvous port is
    picked by bind-and-close: another process can take it in between; rank 0 then fails within seconds — that one case gets ONE
    more attempt with another port."""
    import signal
    import socket
    import subprocess
    import threading
    def free_port():
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            return str(sk.getsockname()[1])
    given_port = os.environ.get("MASTER_PORT")
    procs, out_lines = [], []

    def stop_ranks(why):
        alive = [(r, p) for r, p in enumerate(procs) if p.poll() is None]
        if not alive:
            return
        print(f"bench: {why}: stopping ranks {[r for r, _ in alive]} (pids {[p.pid for _, p in alive]})", file=sys.stderr, flush=True)
        for sig, grace in ((signal.SIGTERM, LAUNCH_GRACE_S), (signal.SIGKILL, LAUNCH_GRACE_S)):
            for _, p in alive:
                if p.poll() is None:
                    try:
                        os.killpg(p.pid, sig)                  # start_new_session: the rank leads its own process group
                    except (ProcessLookupError, PermissionError):
                        pass
            t_end = time.monotonic() + grace
            while time.monotonic() < t_end and any(p.poll() is None for _, p in alive):
                time.sleep(0.05)

    def on_signal(signum, _frame):
        stop_ranks(f"parent received signal {signum}")
        sys.exit(128 + signum)

    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    deadline = float(os.environ.get("CSN_BENCH_LAUNCH_TIMEOUT_S", "0")) or None
    bad = []
    try:
        # A self-picked port can be taken between the probe and rank 0's bind: the ranks then fail at rendezvous, within seconds and
        # before any of them has done work — ONE more attempt with another port (fresh processes again).
        for attempt in range(1 if given_port else 2):
            port = given_port or (os.environ.get("CSN_BENCH_FIRST_PORT") if attempt == 0 else None) or free_port()   # (FIRST_PORT: the test's taken port)
            del procs[:], out_lines[:]
            t0 = time.monotonic()
            for r in range(args.gpus):
                env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                           MASTER_ADDR="127.0.0.1", MASTER_PORT=port, CSN_BENCH_CHILD="1")
                env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
                procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                              stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True,
                                              start_new_session=True))
            reader = threading.Thread(target=lambda: out_lines.extend(procs[0].stdout), daemon=True)   # rank 0's pipe never fills
            reader.start()
            while True:
                codes = [p.poll() for p in procs]
                bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
                if bad or all(c == 0 for c in codes):
                    break
                if deadline is not None and time.monotonic() - t0 > deadline:
                    bad = [("launcher", f"no result after CSN_BENCH_LAUNCH_TIMEOUT_S={deadline:g} s")]
                    break
                time.sleep(0.1)
            if bad:
                stop_ranks(f"ranks failed (rank, exit code): {bad}")
            reader.join(timeout=5)
            rendezvous_failure = bad and bad[0][0] == 0 and time.monotonic() - t0 < RENDEZVOUS_RETRY_S and not any(l.startswith("{") for l in out_lines)
            if not rendezvous_failure:
                break
            if attempt == 0 and not given_port:
                print(f"bench: rank 0 failed {time.monotonic() - t0:.1f} s after the launch (port {port} taken?): one more attempt "
                      "with another port", file=sys.stderr, flush=True)
    finally:
        stop_ranks("launcher leaving")
        for sg, h in old.items():
            signal.signal(sg, h)
    for line in out_lines:                            # stdout carries the JSON line only; anything else a library printed goes to stderr
        line = line.rstrip("\n")
        print(line, file=sys.stdout if line.startswith("{") and not bad else sys.stderr, flush=True)
    if bad:
        print(f"bench: ranks failed (rank, exit code): {bad}", file=sys.stderr, flush=True)
        sys.exit(1)
    sys.exit(0)


def launch_check():
    """CSN_BENCH_LAUNCH_CHECK=1 (tests/test_bench_launch.py, no GPU): the ranks of a self-launched run only prove that they
    exist — process group over gloo on the CPU, one all-reduce, rank 0 prints what it saw — and leave.  Failure modes the test
    drives: `fail` (the last rank exits 3 AFTER the collective), `die_before_collective` (the last rank exits 3 BEFORE it while
    the others behave like ranks stuck inside an RCCL collective: whatever gloo tells them, they do not return), `sleep` (every
    rank stays in the step until it is stopped: the parent-is-signalled case)."""
    import datetime
    import torch.distributed as dist
    mode = os.environ.get("CSN_BENCH_LAUNCH_CHECK")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    print(f"bench-rank-pid {rank} {os.getpid()}", file=sys.stderr, flush=True)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=INIT_TIMEOUT_S))
    t = torch.tensor([float(rank + 1)])
    if mode == "die_before_collective":
        if rank == world - 1:
            os._exit(3)
        try:
            dist.all_reduce(t)
        except Exception:
            pass
        time.sleep(600)                                                  # (an RCCL rank would still be inside the collective)
    dist.all_reduce(t)
    if mode == "sleep":
        print(f"bench-rank-ready {rank}", file=sys.stderr, flush=True)
        time.sleep(600)
    if mode == "fail" and rank == world - 1:
        sys.exit(3)                                                      # (the parent must notice a rank that dies)
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_ranks_seen": dist.get_world_size(), "rank_sum": t.item(),
                          "local_rank": int(os.environ["LOCAL_RANK"]), "argv": sys.argv[1:]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)       # median of >= 20 timed steps after >= 5 warm-ups
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=3,
                    help="workload: 3 = the metric's configuration (default), 2 / 5 = BASELINE configs[1] / configs[4]")
    ap.add_argument("--shapes", type=int, default=None, help="query shapes per GPU (default: the configuration's)")
    ap.add_argument("--K", type=int, default=None)
    ap.add_argument("--heads", type=int, default=1,
                    help="n_heads (d_k = d_v = d per head): 1 = csa_training.py:37's default and the metric's configuration; 8 with "
                         "--K 4 = the reference's published checkpoint (get_csa_pred.py:35-36)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--math", choices=sorted(MATH_MODES), default="bf16x3",
                    help="arithmetic of the contractions: exact fp32 matrix cores; three bf16 products per fp32 product "
                         "(default: inside the 1e-4 contract); one bf16 / fp16 product (outside it, reported with its error)")
    ap.add_argument("--same-work", action="store_true",
                    help="N = 1 only: run the per-GPU work of the N > 1 path (K+2 evaluations per shape: every pooled descriptor "
                         "computed once, by its owner; the 32 shapes are their own collection, no exchange) instead of the "
                         "reference's 2K+2 — the like-for-like N = 1 point of the scaling curve")
    ap.add_argument("--no-named-modes", action="store_true",
                    help="skip config.named_modes (BASELINE configs[1] in bf16 and configs[4] in fp16 after the default headline)")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the secondary runs (other math mode, eval-mode arithmetic): what the profiles are taken with")
    args = ap.parse_args()
    H = args.heads

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                                                # (never returns; nothing has touched a GPU yet)
    if os.environ.get("CSN_BENCH_LAUNCH_CHECK"):
        return launch_check()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_ranks_seen = 1
    # CSN_BENCH_GROUP_OF_ONE=1 (development aid, one GPU): the multi-GPU code path — process group, sharded step, exchange,
    # gradient all-reduce, max-over-ranks timing — in a world of ONE rank over RCCL: what a one-GPU box can rehearse of it
    grouped = world > 1 or os.environ.get("CSN_BENCH_GROUP_OF_ONE") == "1"
    if grouped:
        # the process group comes up BEFORE anything touches the GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            for k, v in (("MASTER_PORT", "29581"), ("RANK", "0"), ("WORLD_SIZE", "1")):
                os.environ.setdefault(k, v)
        # CSN_BENCH_BACKEND=gloo + CSN_BENCH_ONE_GPU=1 rehearse the N > 1 code path with all ranks on one GPU (no RCCL)
        backend = os.environ.get("CSN_BENCH_BACKEND", "nccl")
        if os.environ.get("CSN_BENCH_ONE_GPU") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(seconds=INIT_TIMEOUT_S))
        else:
            dist.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=INIT_TIMEOUT_S))
        n_ranks_seen = dist.get_world_size()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import types
    ctx = types.SimpleNamespace(world=world, rank=rank, local_rank=local_rank, n_ranks_seen=n_ranks_seen, grouped=grouped, dev=dev)
    out = run(args, ctx)
    if out is not None:
        if named_modes_wanted(args, grouped):
            out["config"]["named_modes"] = named_modes(args, ctx)
        print(json.dumps(out), flush=True)
    if grouped:
        dist.destroy_process_group()


# BASELINE.json names a precision for configs[1] ("bf16") and configs[4] ("fp16 MFMA"): after the headline the default run times
# those two workloads in those modes (a few steps each), so that the driver's line carries them too (config.named_modes)
NAMED_MODES = [(2, "bf16"), (5, "fp16")]
NAMED_STEPS, NAMED_WARMUP = 5, 2


def named_modes_wanted(args, grouped):
    return (not grouped and not args.no_named_modes and not args.headline_only and not args.same_work and args.config == 3
            and args.math == "bf16x3" and args.shapes is None and args.K is None and args.heads == 1
            and not any(os.environ.get(k) for k in ("CSN_BENCH_HOST_NB", "CSN_BENCH_SPLIT")))


def named_modes(args, ctx):
    import copy
    res = []
    for config, math in NAMED_MODES:
        a = copy.copy(args)
        a.config, a.math, a.steps, a.warmup, a.headline_only, a.no_cpu_baseline = config, math, NAMED_STEPS, NAMED_WARMUP, True, True
        o = run(a, ctx)
        r = o["roofline"]
        res.append({"baseline_config": CONFIGS[config]["name"], "workload": o["config"]["workload"], "math": math, "dtype": o["dtype"],
                    "steps": NAMED_STEPS, "warmup": NAMED_WARMUP, "ms_per_step": o["ms_per_step"], "points_per_s": o["value"],
                    "loss": o["config"]["loss"], "step_tflops_algorithmic": o["config"]["step_tflops_algorithmic"],
                    "dominant_launch": {k: r.get(k) for k in ("kernel", "launch_ms", "achieved", "peak", "unit", "frac", "traffic",
                                                              "traffic_source", "hbm", "profile")},
                    "launches": o["launches"]})
    return res


def run(args, ctx):
    """One workload (args.config / args.math / args.steps ...) timed on this rank; rank 0 returns the JSON object of the line."""
    world, rank, local_rank, n_ranks_seen, grouped, dev = ctx.world, ctx.rank, ctx.local_rank, ctx.n_ranks_seen, ctx.grouped, ctx.dev
    H = args.heads
    if grouped:
        import torch.distributed as dist
    import csn_amd
    from csn_amd import functional as CF, tuning
    from csn_amd.csa_models import get_model
    csn_amd.build()
    set_math = lambda name: csn_amd._lib.check(csn_amd.lib().csn_set_math_mode(MATH_MODES[name]))
    set_math(args.math)

    cfg = dict(CONFIGS[args.config])
    if args.shapes is not None:
        cfg["B"] = args.shapes
    if args.K is not None:
        cfg["K"] = args.K
    B, K, N, C, d, T, nb = (cfg[k] for k in ("B", "K", "N", "C", "d", "T", "nb"))
    D = H * d
    S = B * world
    torch.manual_seed(0)
    model = get_model("csa", N_CLS, H, K, d_model=C, d_k=d, d_v=d, block=T, n_blocks=nb).to(dev)
    model.trust_neighbor_slot0 = True      # the stack built below has the shape itself in slot 0, like CSADatasetK
    params = [p for n, p in model.named_parameters() if not n.startswith("fc_1")]

    rng = np.random.default_rng(1234 + (args.config - 1) + 1000 * rank)
    feats = torch.from_numpy(rng.standard_normal(size=(B, C, N)).astype(np.float32)).to(dev)
    label = torch.from_numpy(np.where(rng.random(size=(B, N)) < 0.1, 0,
                                      rng.integers(0, N_CLS, size=(B, N))).astype(np.int64)).to(dev)
    shard = None
    # K independent synthetic neighbour maps per query shape.  The neighbour stack (B, K+1, C, N, 1) is an input of the
    # step — what CSADatasetK hands the model, slot 0 = the shape itself (features_data_loader.py:66-82) — so it is
    # resident in HBM before the timed region.  (N > 1: only the same-work reference run below uses it.)
    x_nb_resident = torch.empty((B, K + 1, C, N, 1), device=dev, dtype=torch.float32)
    x_nb_resident[:, 0, :, :, 0] = feats
    if not grouped:
        for k in range(K):
            x_nb_resident[:, k + 1, :, :, 0] = torch.from_numpy(rng.standard_normal(size=(B, C, N)).astype(np.float32)).to(dev)
    else:
        # config 4: K-regular shape graph over the whole collection (never self), same on every rank
        from csn_amd.sharding import ShapeGraphShard, regular_graph
        shard = ShapeGraphShard(regular_graph(S, K), B, rank, world, dev)

    # CSN_BENCH_HOST_NB=1 (N = 1, development aid): the neighbour stack stays in pinned HOST memory and crosses PCIe inside
    # every step, the way csa_training.py:198-202 hands it over — the PCIe-inclusive rate DESIGN.md quotes; never `value`
    x_nb_host = None
    if not grouped and os.environ.get("CSN_BENCH_HOST_NB") in ("1", "2"):
        x_nb_host = x_nb_resident.cpu()                                   # "2": pageable, as the reference's DataLoader hands it over
        if os.environ["CSN_BENCH_HOST_NB"] == "1":
            x_nb_host = x_nb_host.pin_memory()

    # HIP events around the launching C-ABI calls: entry point -> [(start, end), ...].  An event record is a barrier packet in the
    # queue — around every one of a step's ~45 calls they cost the config-3 step 0.5 ms — so EVERY call is bracketed only in the
    # warm-up steps (which is where `launches` and the choice of the longest launch come from), and the timed steps bracket the
    # attention launches and that longest launch alone (`roofline.launch_ms` is live, from the timed region)
    call_events = {}
    watch = [None]                                                       # None: every call; a set: those entry points
    ATTN_CALLS = {"csn_block_attn_fwd_f32": "fwd", "csn_block_attn_fwd_grouped_f32": "fwd", "csn_block_attn_bwd_dq_f32": "bwd", "csn_block_attn_bwd_dq_recompute_f32": "bwd",
                  "csn_block_attn_bwd_dkv_flash_f32": "dkv"}  # dkv: the key-stationary dK / dV launch of the score-recomputing flow
    _open = {}

    def call_hook(name, phase):
        if watch[0] is not None and name not in watch[0]:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        if phase == "begin":
            _open[name] = ev
        else:
            call_events.setdefault(name, []).append((_open.pop(name), ev))
    exchange_mode = os.environ.get("CSN_EXCHANGE", "alltoall")           # "allgather": the whole collection to every rank
    # --math bf16 / fp16: the neighbour features cross xGMI as bf16 (the consumers round them to 16 bits anyway): config 4 as
    # SURVEY.md §8(e) sizes it; the parity modes exchange the fp32 maps
    payload_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(args.math)      # (fp16 consumers keep 11 bits: an fp16 payload)
    # an fp16 payload has fp16's RANGE too: checked once, outside the timed region, on what is sent — features beyond half of
    # fp16's largest number travel as bf16 instead (no silent inf on real feature files); `config.exchange` says what was done
    payload_range = None
    if payload_dtype is torch.float16 and grouped:
        amax = float(feats.abs().max())
        tiny = float((feats.abs() < 6.0e-8).logical_and(feats != 0).float().mean())
        if amax > 32752.0:
            payload_dtype = torch.bfloat16
        payload_range = {"send_amax": amax, "fraction_below_fp16_subnormal": tiny, "fp16_max": 65504.0,
                         "payload_kept_fp16": payload_dtype is torch.float16}
    overlap = os.environ.get("CSN_OVERLAP", "1") != "0"                  # exchange in flight under the self-attention evaluations

    # CSN_BENCH_SPLIT=1 (N = 1, development aid): run the two-phase evaluation order of the multi-GPU path with the stack
    # already complete, to price its extra work against the single call
    split_probe = not grouped and os.environ.get("CSN_BENCH_SPLIT") == "1"

    class _ReadyStack:
        """stands in for csn_amd.sharding.PendingStack with the exchange already complete (development aids below)"""
        reuse_descriptors = False

        def __init__(self, stack, graph=None):
            self.stack, self.graph = stack, graph
            self.reuse_descriptors = graph is not None

        def wait(self):
            return self.stack

        def gather_pooled(self, own_pooled):
            return own_pooled[self.graph]            # every neighbour is one of this process's own shapes

    # The per-GPU work of the N > 1 path without any exchange ("same work"): the B shapes are their own collection (K-regular
    # graph among them), neighbour descriptors are taken from their "owners" — K+2 evaluations per shape instead of the 2K+2
    # the reference runs.  N = 1: selected by --same-work.  N > 1: every rank times it before the group step, so that the
    # scaling line carries its own like-for-like single-GPU reference (config.n1_same_work_ms_per_step).
    from csn_amd.sharding import regular_graph as _regular_graph
    local_graph = torch.from_numpy(_regular_graph(B, K)).to(dev) if (args.same_work or grouped) else None
    if local_graph is not None:
        x_nb_resident[:, 1:, :, :, 0] = feats[local_graph]
    same_work_n1 = not grouped and args.same_work

    def step(record=False, local=False):
        """local: the same-work step on this rank's own shapes (no exchange, no gradient all-reduce)"""
        for p in params:
            p.grad = None
        if shard is not None and not local:
            if overlap:
                x_nb = shard.exchange_async(feats, mode=exchange_mode,   # the model overlaps its self-attention with it
                                            reuse_descriptors=os.environ.get("CSN_REUSE", "1") != "0",
                                            payload_dtype=payload_dtype if exchange_mode == "alltoall" else None)
            elif exchange_mode == "alltoall":
                x_nb = shard.exchange_neighbours(feats)                  # neighbour-only all-to-all
            else:
                x_nb = shard.neighbour_stack(feats, shard.exchange(feats))   # all-gather of point features over xGMI
        else:
            x_nb = x_nb_host if x_nb_host is not None else x_nb_resident   # (B, K+1, C, N, 1), slot 0 = self
            if split_probe:
                x_nb = _ReadyStack(x_nb)
            elif local or same_work_n1:
                x_nb = _ReadyStack(x_nb_resident, local_graph)
        csn_amd._lib.set_call_hook(call_hook if record else None)       # HIP events around every launch of the layer
        try:
            logits = model(feats.unsqueeze(-1), "train", x_nb)
            loss = masked_ce(logits, label)
            loss.backward()
        finally:
            csn_amd._lib.set_call_hook(None)
        if shard is not None and not local:
            shard.allreduce_grads(params)                                # one 1.6 MB bucket
        return loss

    def timed(train_mode, local=False):
        """W warm-up + K timed steps, barrier + synchronize on both sides, max over ranks.  Every run starts from the same
        generator state, so the dropout masks (pure functions of seed and position) are the same in every math mode and
        the losses of two modes can be compared.  Returns the bracketed wall time of the K steps (max over ranks) and the
        per-step times (HIP events at the step boundaries on the launch stream, per step the max over ranks): the metric's
        value is quoted on their MEDIAN, the mean of the bracketed time beside it."""
        model.train(train_mode)                                          # train: dropout p = 0.1 live (csa_training.py:192)
        torch.manual_seed(1)
        call_events.clear()
        watch[0] = None
        group = grouped and not local
        for w in range(args.warmup):
            if w == 1:
                call_events.clear()                                      # (the first warm-up step pays for allocations)
            step(record=True, local=local)
        torch.cuda.synchronize()
        n_warm = max(1, args.warmup - 1) if args.warmup else 0
        warm_calls = {k: float(np.sum([a.elapsed_time(b) for a, b in v])) / n_warm for k, v in call_events.items()} if n_warm else {}
        if warm_calls:
            watch[0] = set(ATTN_CALLS) | {max(warm_calls, key=warm_calls.get)}
        call_events.clear()
        if group:
            dist.barrier()
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        for i in range(args.steps):
            marks[i].record()
            loss = step(record=True, local=local)
        marks[args.steps].record()
        torch.cuda.synchronize()
        if group:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        step_ms = torch.tensor([marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)], device=dev, dtype=torch.float64)
        if group:
            tmax = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(step_ms, op=dist.ReduceOp.MAX)
            el = tmax.item()
        step_ms = step_ms.cpu().numpy()
        # per STEP: the attention launches of one step summed (N = 1: one forward and one backward launch; the overlapped
        # multi-GPU path: two of each — own shapes first, the evaluations that need neighbour data after the exchange)
        per_call = {k: float(np.sum([a.elapsed_time(b) for a, b in v])) / args.steps for k, v in call_events.items()}
        watch[0] = None
        # every call of the step from the warm-up steps, the watched ones overwritten with their timed-region means
        ms = {"fwd": float("nan"), "bwd": float("nan"), "dkv": float("nan"), "calls": {**warm_calls, **per_call},
              "calls_live": sorted(per_call)}
        for name, which in ATTN_CALLS.items():
            if name in per_call:
                ms[which] = per_call[name] if np.isnan(ms[which]) else ms[which] + per_call[name]
        gnorm = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params if p.grad is not None)).item())
        return el, float(loss.item()), ms, gnorm, step_ms

    # headline first (the training step as the reference runs it), then the secondary runs: eval-mode arithmetic in the
    # headline mode, and the same train-mode step in the other arithmetic mode
    other = "fp32" if args.math != "fp32" else "bf16x3"
    same_work_ms = None
    if grouped:
        # the like-for-like single-GPU reference of this line: the same K+2 evaluations per shape on this rank's own shapes
        _, _, _, _, sw = timed(True, local=True)
        same_work_ms = float(np.median(sw))
    elapsed, loss_val, attn_ms, gnorm, step_ms = timed(True)
    if not args.headline_only:
        if not grouped and x_nb_host is None and not same_work_n1:
            # the module's DEFAULT path beside the headline: without `trust_neighbor_slot0` CrossShapeAt re-assembles the neighbour
            # stack with the query features in slot 0 (csa_models.py:344-356 of this package; 1.3 GB at config 3) — what a caller
            # gets whose stack does not come from CSADatasetK
            model.trust_neighbor_slot0 = False
            _, loss_default, _, _, step_ms_default = timed(True)
            model.trust_neighbor_slot0 = True
        else:
            step_ms_default = None
        elapsed_eval, loss_eval, _, _, step_ms_eval = timed(False)      # eval-mode arithmetic (dropout off), gradients on
        set_math(other)
        elapsed_other, loss_other, attn_ms_other, gnorm_other, step_ms_other = timed(True)
        set_math(args.math)
    reuse = (grouped and overlap and os.environ.get("CSN_REUSE", "1") != "0") or same_work_n1
    # train mode: the pooled and the mixed self evaluation differ (2K+2 per shape); with descriptor reuse (N > 1) the K
    # neighbour self-attention evaluations per shape are their owners' work: K+2 per shape
    n_evals = B * ((K + 2) if reuse else (2 * K + 2))

    if rank == 0:
        launch_flops = n_evals * 4 * N * T * D         # forward: QK^T + PV; backward: dO V^T + dS K — the same count
        traffic_tab = {}
        if os.path.exists(TRAFFIC_FILE):
            with open(TRAFFIC_FILE) as fh:
                traffic_tab = json.load(fh)

        def roof(math, which, ms_all):
            ms = ms_all[which]
            flash = not np.isnan(ms_all.get("dkv", float("nan")))       # the step ran the score-recomputing flow
            fast = math != "fp32"
            ach = launch_flops / (ms * 1e-3) / 1e12
            peak = PEAK_TFLOPS[math]
            tmpl = f"<{d // 32}>" if which == "dkv" else f"<{d // 32},{'true' if which == 'bwd' else 'false'}{',true' if fast else ''}>"
            # the committed counter passes are of the single-call step at the configuration's own size: other launch shapes
            # (descriptor reuse / the two-phase order of the N > 1 path, --shapes, --K) have no measured traffic
            same_launch = (B, K) == (CONFIGS[args.config]["B"], CONFIGS[args.config]["K"]) and not reuse and not split_probe
            entry = traffic_tab.get(f"{args.config}/{math}/{which}") if same_launch else None
            base = "csn_attn_dkv_kernel" if which == "dkv" else "csn_attn_bf16x3_kernel" if fast else "csn_attn_f32_kernel"
            what = {"bwd": " (fused block attention backward: dP, dS, dQ" + ("; scores recomputed)" if flash else ")"),
                    "fwd": " (fused block attention forward)",
                    "dkv": " (key-stationary attention backward: dV, dK from recomputed P, dS)"}[which]
            # the kernel's NAME comes from the committed kernel-trace statistics of this command where there are any (the name
            # rocprofv3 prints); the template arguments this run derives from its geometry must be among that name's, in order
            prof = None
            if entry and entry.get("kernel"):
                pn = entry["kernel"]
                p_args = [t.strip() for t in pn[pn.index("<") + 1:pn.index(">")].split(",")] if "<" in pn else []
                want, it_ = [t for t in tmpl.strip("<>").split(",")], iter(p_args)
                prof = {"kernel_stats_avg_ms": entry.get("kernel_stats_avg_ms"), "source": entry.get("kernel_stats_source"),
                        "name_matches_this_run": pn.startswith(base + "<") and all(t in it_ for t in want)}
            r = {"bound": "mfma",
                 "kernel": (entry["kernel"] if prof and prof["name_matches_this_run"] else base + tmpl) + what,
                 "profile": prof,
                 "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                 "traffic": entry["bytes_per_launch"] if entry else None,
                 "traffic_source": (entry["source"] + " (rocprofv3 --pmc passes of this command, not measured in this run)") if entry else None,
                 "launch_ms": ms, "launch_ms_is": f"mean over {args.steps} timed steps of the step's launches of this kernel (HIP events on the launch stream)",
                 "flops_per_launch": launch_flops,
                 "note": f"algorithmic FLOPs: 4*T*D = {4 * T * D // 1000} kFLOP per query point per evaluation ({H} head(s) of d = {d}) x {N} points x "
                         f"{n_evals} evaluations in the launch"
                         + (" (dV = P^T dO and dK = dS^T Qs; the S and dP products this kernel recomputes are not credited)" if which == "dkv" else "")
                         + (" (dP and dQ; the recomputed S product is not credited)" if which == "bwd" and flash else "")}
            if math == "bf16x3":
                r["note"] += "; this mode issues 3 bf16 matrix FLOPs per algorithmic FLOP, so the matrix pipe sees 3x `achieved`"
            if entry:
                # the matrix roofline binds on algorithmic bytes; the bytes the kernel really moves (it keeps the
                # score blocks in HBM between forward and backward) against the HBM roof, so that both distances are on the line
                gbs = entry["bytes_per_launch"] / (ms * 1e-3) / 1e9
                r["hbm"] = {"achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                            "note": "measured traffic / this run's launch time"}
            return r

        # algorithmic matmul FLOPs per step of the other launching entry points (what a non-attention launch would be priced on)
        E_all, S_all = n_evals, B * (K + 1)
        OTHER_FLOPS = {
            "csn_block_attn_bwd_dkv_f32": (launch_flops, "dV = P^T dO and dK = dS^T Qs on the P / dS planes: 4*T*d per query point and evaluation"),
            "csn_project_f32": (2.0 * 3 * D * C * N * S_all, "Q / K / V projections of every slot (+ the logit layer)"),
            "csn_outproj_ln_fwd_f32": (2.0 * C * D * N * E_all, "out-projection of every evaluation (+ residual + LayerNorm)"),
            "csn_outproj_ln_bwd_f32": (2.0 * 2 * C * D * N * E_all, "LayerNorm backward, dCtx = W_fc^T dZ and the W_fc gradient"),
            "csn_project_wgrad_f32": (2.0 * 3 * D * C * N * S_all, "projection weight gradients (+ the logit layer's)"),
        }

        def roof_other_call(math, name, ms):
            fl, what = OTHER_FLOPS.get(name, (None, "no FLOP model for this entry point"))
            r = {"bound": "mfma", "kernel": f"{name} (C-ABI entry point: {what})", "launch_ms": ms,
                 "launch_ms_is": f"mean over {args.steps} timed steps of the step's calls of this entry point (HIP events on the launch stream)",
                 "achieved": None, "peak": PEAK_TFLOPS[math], "unit": "TFLOP/s", "frac": None, "traffic": None, "flops_per_launch": fl,
                 "note": "NOT an attention launch: the longest launch of the step has changed — re-profile (scripts/run_profile.sh)"}
            if fl:
                r["achieved"] = fl / (ms * 1e-3) / 1e12
                r["frac"] = r["achieved"] / PEAK_TFLOPS[math]
            return r

        def roofs(math, ms):
            # the longest launching entry point of the step, whatever it is; beside it the longest (other) attention launch
            ran = sorted((k for k in ("fwd", "bwd", "dkv") if not np.isnan(ms[k])), key=lambda k: -ms[k]) or ["bwd", "fwd"]
            calls = ms.get("calls", {})
            top = max(calls, key=calls.get) if calls else None
            if top is not None and top not in ATTN_CALLS:
                return roof_other_call(math, top, calls[top]), roof(math, ran[0], ms)
            dom, oth = ran[0], (ran[1] if len(ran) > 1 else ran[0])
            return roof(math, dom, ms), roof(math, oth, ms)

        dominant, second = roofs(args.math, attn_ms)
        med_ms = float(np.median(step_ms))
        out = {
            "metric": "CSA fwd+bwd points/sec (10k pts x 256 ch, K=3)" if args.config == 3 else
                      f"CSA fwd+bwd points/sec ({N // 1000}k pts x {C} ch, K={K})",
            "value": S * N / (med_ms * 1e-3),
            "unit": "points/s", "n_gpus": world, "n_ranks_seen": n_ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": med_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "timing": {"value_is": "shapes x points / MEDIAN step time (SURVEY.md 8(d)); step times = HIP events at the step "
                                   "boundaries on the launch stream, per step the max over ranks",
                       "ms_per_step_mean": elapsed / args.steps * 1e3,
                       "value_mean": S * N * args.steps / elapsed,
                       "mean_is": "the K timed steps bracketed by barrier + synchronize on both sides (wall clock, max over ranks) / K",
                       "ms_per_step_min": float(step_ms.min()), "ms_per_step_max": float(step_ms.max())},
            "dtype": DTYPE_TEXT[args.math],
            "data": "synthetic",
            "config": {"workload": f"{cfg['name']}: CSA K={K}, {B} query shapes/GPU x {N} pts x {C} ch, n_heads={H}, d_k=d_v={d}, "
                                   f"{nb} blocks of {T}, {N_CLS} classes, fwd + masked CE + bwd, train mode (dropout 0.1 live, "
                                   + ("K+2 evaluations/shape: every shape's pooled SSA descriptor is computed once, by its owner"
                                      if reuse else "2K+2 evaluations/shape") + f"), math mode {args.math}",
                       "evaluations_per_shape": n_evals // B,
                       "shapes_total": S, "K": K,
                       "parallelism": (f"shape-graph sharded x{world}, exchange {exchange_mode}" + (" overlapped" if overlap else "")
                                       + (", descriptor reuse" if reuse else "")) if grouped else "single GPU",
                       "loss": loss_val, "grad_norm": gnorm,
                       "step_tflops_algorithmic": 3 * algorithmic_flops_fwd(S, K, N, C, D, T) / (med_ms * 1e-3) / 1e12},
            "roofline": dominant, "roofline_other": second,
            # every launching entry point of the step, mean ms per step (HIP events around the C-ABI calls; a call may be several
            # kernels: csn_outproj_ln_bwd_f32 = LayerNorm backward + dCtx + W_fc gradient, csn_block_attn_bwd_dkv_f32 = dV + dK).
            # launches_live: the entry points bracketed inside the timed steps; the others were bracketed in the warm-up steps
            "launches": {k: round(v, 4) for k, v in sorted(attn_ms.get("calls", {}).items(), key=lambda kv: -kv[1])},
            "launches_live": attn_ms.get("calls_live", []),
        }
        if grouped:
            sent, recvd = getattr(shard, "payload_bytes", (None, None))
            out["config"]["exchange"] = {"mode": exchange_mode, "payload_dtype": str(payload_dtype or torch.float32).replace("torch.", ""),
                                         "payload_range_check": payload_range,
                                         "bytes_sent_per_rank": sent, "bytes_received_per_rank": recvd,
                                         "note": "rank 0's neighbour-only all-to-all of one step (None: the all-gather fallback moves the whole collection)"}
            out["config"]["n1_same_work_ms_per_step"] = same_work_ms
            out["config"]["scaling_note"] = (
                f"this line runs K+2 = {K + 2} evaluations per shape (descriptor reuse: a neighbour's pooled SSA descriptor is its "
                f"owner's), the default N = 1 line runs the reference's 2K+2 = {2 * K + 2}: compare N > 1 lines with "
                "`bench.py --same-work` at N = 1, or with n1_same_work_ms_per_step — that step (same evaluations, own shapes as "
                "the collection, no exchange, no gradient all-reduce) timed on rank 0 of THIS run before the group steps"
                if reuse else "same evaluations per shape as the N = 1 line (descriptor reuse off)")
        if same_work_n1:
            out["config"]["scaling_note"] = ("--same-work: the per-GPU work of the N > 1 path (K+2 evaluations per shape, own shapes "
                                             "as the collection, no exchange) — the like-for-like N = 1 point of the scaling curve; "
                                             "NOT the metric's headline (that is the default run: the reference's 2K+2)")
        if not args.headline_only:
            if step_ms_default is not None:
                out["config"]["default_path"] = {"ms_per_step": float(np.median(step_ms_default)),
                                                 "points_per_s": S * N / (float(np.median(step_ms_default)) * 1e-3), "loss": loss_default,
                                                 "note": "the same step with model.trust_neighbor_slot0 = False (the module's default): the "
                                                         "neighbour stack is re-assembled with the query features in slot 0 first; the "
                                                         "headline sets the flag because its stack has the CSADatasetK form (slot 0 = self)"}
            out["config"]["dropout_off"] = {"points_per_s": S * N / (float(np.median(step_ms_eval)) * 1e-3),
                                            "ms_per_step": float(np.median(step_ms_eval)), "loss": loss_eval,
                                            "note": "same step with eval-mode arithmetic (2K+1 evaluations/shape), gradients on"}
            d_o, s_o = roofs(other, attn_ms_other)
            out["config"][f"math_{other}"] = {"points_per_s": S * N / (float(np.median(step_ms_other)) * 1e-3),
                                              "ms_per_step": float(np.median(step_ms_other)), "loss": loss_other,
                                              "grad_norm": gnorm_other, "roofline": d_o, "roofline_other": s_o,
                                              "note": "the same train-mode step (same dropout masks) in the other arithmetic mode"}
            # the timed step is also a checked step: same masks, two arithmetic modes — loss and gradient norm must agree
            tol = 1e-4 if {args.math, other} <= {"fp32", "bf16x3"} else 5e-2
            ok = abs(loss_val - loss_other) <= tol * max(1.0, abs(loss_other)) and abs(gnorm - gnorm_other) <= max(tol, 1e-3) * gnorm_other
            out["config"]["cross_mode_check"] = {"loss_abs_diff": abs(loss_val - loss_other),
                                                 "grad_norm_rel_diff": abs(gnorm - gnorm_other) / gnorm_other, "tolerance": tol,
                                                 "ok": bool(ok)}
            if not ok:
                print(f"bench: WARNING the {args.math} step disagrees with the {other} step beyond {tol}: "
                      f"loss {loss_val} vs {loss_other}, |grad| {gnorm} vs {gnorm_other}", file=sys.stderr, flush=True)
        if not grouped and not args.no_cpu_baseline:
            cores = min(len(os.sched_getaffinity(0)), 16)          # the GPU box gives one GPU a 16-core share
            sample = 1 if H > 1 else (4 if args.config == 3 else (2 if args.config == 2 else 1))
            out["cpu_baseline"] = cpu_baseline(cfg, min(sample, B), cores, H=H)
        return out
    return None


if __name__ == "__main__":
    main()
