#!/usr/bin/env python3
"""Headline benchmark: CSA forward + backward query-points/sec (BASELINE.json metric) on N MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic shapes, exactly what
MID-FC/csa_training.py:202-211 does per batch: model(x, neighbours) -> masked cross-entropy -> backward
(no optimizer step, no data loading; features resident in HBM).

Workload at N = 1: BASELINE.json configs[2] — 32 query shapes x 10000 points x 256 channels, K = 3
neighbour shapes each (independent synthetic maps), n_heads = 1 (csa_training.py:37 default), 39 classes
(PartNet Chair).  For N > 1 (configs[3]): every rank owns 32 shapes of a 32*N-shape collection (weak
scaling), neighbours are drawn from the whole collection, the ranks all-gather the point features over
RCCL/xGMI inside the timed region, and the 11 weight gradients are all-reduced at the end of the step.

Prints ONE JSON line (rank 0).  `roofline` is for the fused block-attention forward kernel
(csn_attn_f32_kernel<8,false>), timed live with HIP events on the launch stream; `cpu_baseline` is the
oracle's faithful op-for-op port of the reference timed on the host cores over a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4, 64 FLOP/clk/SIMD
# HBM bytes of one fused-attention-forward launch per evaluation, from the PMC passes in profiles/r1s_pmc_hbm_traffic.txt:
# (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / 256 evaluations = (2 x 3.911e6 + 7.631e6) KB / 256 (gfx950: FETCH_SIZE counts half of
# a 16-byte-per-lane read stream).  Algorithmic: 3 x 10.24 MB of Q/K/V in, 20 MB of scores + 10.24 MB of context out = 61 MB.
ATTN_FWD_HBM_BYTES_PER_EVAL = (2 * 3.911e6 + 7.631e6) * 1024 / 256
PEAK_BF16_MATRIX_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA (the bf16x3 mode issues 3 bf16 FLOPs per algorithmic FLOP)
N_POINTS, C, T, H, D_HEAD, N_CLS = 10000, 256, 500, 1, 256, 39


def algorithmic_flops_fwd(B, K, N=N_POINTS, Cc=C, Dd=H * D_HEAD, Tt=T):
    """unique matmul FLOPs of one forward (projections once per shape, 2K+1 evaluations)."""
    return B * ((K + 1) * 6 * N * Cc * Dd + (2 * K + 1) * (4 * N * Tt * Dd + 2 * N * Dd * Cc))


def attn_fwd_flops(B, K, N=N_POINTS, Dd=H * D_HEAD, Tt=T):
    """QK^T + PV of the fused attention forward launch: (2K+1) evaluations x 4*N*T*D."""
    return B * (2 * K + 1) * 4 * N * Tt * Dd


def masked_ce(logits, label):
    """csa_training.py:94-108 restated: mean cross-entropy over the points with label > 0 (label 0 = unlabelled is
    ignored; class-major logits are consumed in place instead of being transposed and gathered)."""
    return torch.nn.functional.cross_entropy(logits.squeeze(-1), label, ignore_index=0)


def cpu_baseline(K, sample_shapes, threads, dropout=True):
    """Oracle's faithful port (20 x 500 chunk loop, growing cat, 2K+2 MHA calls) on the host, fwd + bwd,
    in the same mode as the headline run (train mode: both dropouts live, csa_training.py:192)."""
    from oracle import csa_oracle as orc
    torch.set_num_threads(threads)
    rng = np.random.default_rng(99)
    p = {k: v.requires_grad_(True) for k, v in orc.make_params(rng, H, n_cls=N_CLS, csa=True).items()}
    x = orc.synth_points(rng, (sample_shapes, C, N_POINTS, 1))
    nb = orc.synth_points(rng, (sample_shapes, K + 1, C, N_POINTS, 1))
    nb[:, 0] = x
    lab = orc.synth_labels(rng, sample_shapes, N_POINTS, N_CLS)

    def step():
        for v in p.values():
            v.grad = None
        pd = 0.1 if dropout else 0.0
        logits = orc.forward_csa(x, nb, p, H, mha=lambda a, b, c, pp, h, **kw: orc.mha_faithful(a, b, c, pp, h, p_attn_drop=pd,
                                                                                                   p_out_drop=pd))
        orc.masked_ce_loss(logits, lab).backward()

    step()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    t = sorted(times)[1]
    return {"value": sample_shapes * N_POINTS / t, "unit": "points/s", "cores": threads, "kind": "port",
            "sample": f"{sample_shapes} of the 32 query shapes (K={K}, {'train mode, dropout 0.1' if dropout else 'eval-mode arithmetic'}, fwd+bwd), median of 3 steps after "
                      f"1 warm-up, {t:.2f} s/step; oracle/csa_oracle.py mha_faithful"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--shapes", type=int, default=32, help="query shapes per GPU")
    ap.add_argument("--K", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--math", choices=["fp32", "bf16x3"], default="bf16x3",
                    help="arithmetic of the contractions: exact fp32 matrix cores, or three bf16 products per fp32 product")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the secondary runs (other math mode, eval-mode arithmetic): what the profiles are taken with")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # CSN_BENCH_BACKEND=gloo + CSN_BENCH_ONE_GPU=1 rehearse the N > 1 code path with all ranks on one GPU (no RCCL)
        backend = os.environ.get("CSN_BENCH_BACKEND", "nccl")
        if os.environ.get("CSN_BENCH_ONE_GPU") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import csn_amd
    from csn_amd import functional as CF
    from csn_amd.csa_models import get_model
    csn_amd.build()
    csn_amd._lib.check(csn_amd.lib().csn_set_math_mode(1 if args.math == "bf16x3" else 0))

    B, K = args.shapes, args.K
    S = B * world
    torch.manual_seed(0)
    model = get_model("csa", N_CLS, H, K).to(dev)
    model.trust_neighbor_slot0 = True      # the stack built below has the shape itself in slot 0, like CSADatasetK
    params = [p for n, p in model.named_parameters() if not n.startswith("fc_1")]

    rng = np.random.default_rng(1234 + 2 + 1000 * rank)
    feats = torch.from_numpy(rng.standard_normal(size=(B, C, N_POINTS)).astype(np.float32)).to(dev)
    label = torch.from_numpy(np.where(rng.random(size=(B, N_POINTS)) < 0.1, 0,
                                      rng.integers(0, N_CLS, size=(B, N_POINTS))).astype(np.int64)).to(dev)
    shard = None
    if world == 1:
        # configs[2]: K independent synthetic neighbour maps per query shape, resident in HBM
        nbr_maps = torch.from_numpy(rng.standard_normal(size=(B, K, C, N_POINTS)).astype(np.float32)).to(dev)
        # the neighbour stack (B, K+1, C, N, 1) is an input of the step — what CSADatasetK hands the model, slot 0 = the
        # shape itself (features_data_loader.py:66-82) — so it is resident in HBM before the timed region
        x_nb_resident = torch.cat((feats[:, None], nbr_maps), dim=1).unsqueeze(-1).contiguous()
        del nbr_maps
    else:
        # configs[3]: K-regular shape graph over the whole collection (never self), same on every rank
        from csn_amd.sharding import ShapeGraphShard, regular_graph
        shard = ShapeGraphShard(regular_graph(S, K), B, rank, world, dev)

    attn_events = []
    exchange_mode = os.environ.get("CSN_EXCHANGE", "allgather")          # "alltoall": neighbour-only exchange (sharding.py)
    overlap = os.environ.get("CSN_OVERLAP", "1") != "0"                  # all-gather in flight under the self-attention evaluations

    # CSN_BENCH_SPLIT=1 (N = 1, development aid): run the two-phase evaluation order of the multi-GPU path with the stack
    # already complete, to price its extra work against the single call
    split_probe = world == 1 and os.environ.get("CSN_BENCH_SPLIT") == "1"

    class _ReadyStack:
        def __init__(self, stack):
            self.stack = stack

        def wait(self):
            return self.stack

    def step(record=False):
        for p in params:
            p.grad = None
        if shard is not None:
            if exchange_mode == "allgather" and overlap:
                x_nb = shard.exchange_async(feats)                       # the model overlaps its self-attention with it
            elif exchange_mode == "alltoall":
                x_nb = shard.exchange_neighbours(feats)                  # neighbour-only all-to-all (opt-in)
            else:
                x_nb = shard.neighbour_stack(feats, shard.exchange(feats))   # all-gather of point features over xGMI
        else:
            x_nb = x_nb_resident                                         # (B, K+1, C, N, 1), slot 0 = self
            if split_probe:
                x_nb = _ReadyStack(x_nb)
        if record:
            CF.EVENT_SINK = attn_events
        logits = model(feats.unsqueeze(-1), "train", x_nb)
        CF.EVENT_SINK = None
        loss = masked_ce(logits, label)
        loss.backward()
        if shard is not None:
            shard.allreduce_grads(params)                                # one 1.6 MB bucket
        return loss

    def timed(train_mode):
        """W warm-up + K timed steps, barrier + synchronize on both sides, max over ranks."""
        model.train(train_mode)                                          # train: dropout p = 0.1 live (csa_training.py:192)
        attn_events.clear()
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step(record=True)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = tmax.item()
        ms = float(np.mean([a.elapsed_time(b) for a, b in attn_events])) if attn_events else float("nan")
        return el, float(loss.item()), ms

    # headline first (the training step as the reference runs it), then the secondary runs: eval-mode arithmetic in the
    # headline mode, and the same train-mode step in the other arithmetic mode
    other = "fp32" if args.math == "bf16x3" else "bf16x3"
    elapsed, loss_val, attn_ms = timed(True)
    if not args.headline_only:
        elapsed_eval, loss_eval, _ = timed(False)      # eval-mode arithmetic (dropout off), gradients on
        csn_amd._lib.check(csn_amd.lib().csn_set_math_mode(1 if other == "bf16x3" else 0))
        elapsed_other, loss_other, attn_ms_other = timed(True)
        csn_amd._lib.check(csn_amd.lib().csn_set_math_mode(1 if args.math == "bf16x3" else 0))
    n_evals = B * (2 * K + 2)                          # train mode: the pooled and the mixed self evaluation differ

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = S * N_POINTS * args.steps / elapsed
        launch_flops = n_evals * 4 * N_POINTS * T * H * D_HEAD       # QK^T + PV of every evaluation in the launch

        def roof(math, ms):
            fast = math == "bf16x3"
            ach = launch_flops / (ms * 1e-3) / 1e12
            peak = PEAK_BF16_MATRIX_TFLOPS if fast else PEAK_F32_MATRIX_TFLOPS
            r = {"bound": "mfma", "kernel": ("csn_attn_bf16x3_kernel<8,false,true>" if fast else "csn_attn_f32_kernel<8,false>")
                                            + " (fused block attention forward)",
                 "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                 "traffic": ATTN_FWD_HBM_BYTES_PER_EVAL * n_evals, "launch_ms": ms,
                 "flops_per_launch": launch_flops,
                 "note": "algorithmic FLOPs: 512 kFLOP per query point per evaluation x 10000 points x evaluations in the launch"}
            if fast:
                r["note"] += "; this mode issues 3 bf16 matrix FLOPs per algorithmic FLOP, so the matrix pipe sees 3x `achieved`"
            return r

        out = {
            "metric": "CSA fwd+bwd points/sec (10k pts x 256 ch, K=3)", "value": S * N_POINTS * args.steps / elapsed,
            "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16x3 (fp32 operands split into 2 bf16, 3 bf16 MFMA per product, fp32 accumulate)" if args.math == "bf16x3"
                     else "f32",
            "data": "synthetic",
            "config": {"workload": f"CSA K={K}, {B} query shapes/GPU x {N_POINTS} pts x {C} ch, n_heads={H}, d_k=d_v={D_HEAD}, "
                                   f"20 blocks of {T}, {N_CLS} classes, fwd + masked CE + bwd, train mode (dropout 0.1 live, "
                                   f"2K+2 evaluations/shape), math mode {args.math}",
                       "shapes_total": S, "K": K, "parallelism": f"shape-graph sharded x{world}" if world > 1 else "single GPU",
                       "loss": loss_val,
                       "step_tflops_algorithmic": 3 * algorithmic_flops_fwd(S, K) / (elapsed / args.steps) / 1e12},
            "roofline": roof(args.math, attn_ms),
        }
        if not args.headline_only:
            out["config"]["dropout_off"] = {"points_per_s": S * N_POINTS * args.steps / elapsed_eval,
                                            "ms_per_step": elapsed_eval / args.steps * 1e3, "loss": loss_eval,
                                            "note": "same step with eval-mode arithmetic (2K+1 evaluations/shape), gradients on"}
            out["config"][f"math_{other}"] = {"points_per_s": S * N_POINTS * args.steps / elapsed_other,
                                              "ms_per_step": elapsed_other / args.steps * 1e3, "loss": loss_other,
                                              "roofline": roof(other, attn_ms_other),
                                              "note": "the same train-mode step in the other arithmetic mode"}
        if world == 1 and not args.no_cpu_baseline:
            cores = min(len(os.sched_getaffinity(0)), 16)          # the GPU box gives one GPU a 16-core share
            out["cpu_baseline"] = cpu_baseline(K, 4, cores)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
