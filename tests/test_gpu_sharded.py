"""Multi-process pieces on the MI355X: two ranks share the one GPU of the test box (gloo carries the collectives; RCCL
needs one GPU per rank: here it runs as a world of ONE rank — the same calls on the device, no peer — and at world > 1 in
the driver's multi-GPU bench only)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import csa_oracle as orc

pytestmark = pytest.mark.gpu

SHARDS = (4, 2)                  # query shapes per rank (uneven on purpose)
N_PTS, K = 300, 2


def _feats():
    rng = np.random.default_rng(17)
    return orc.synth_clustered_feats(rng, sum(SHARDS), N_PTS)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from csn_amd import functional as CF
    from csn_amd.sharding import knn_graph_sharded
    f = _feats()
    lo = sum(SHARDS[:rank])
    mine = f[lo:lo + SHARDS[rank]].cuda()
    g = knn_graph_sharded(mine, K, CF.retrieval_measure, pair_budget=2 * sum(SHARDS) * N_PTS)    # two query rows per chunk
    torch.save(g.cpu(), os.path.join(out_dir, f"g{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_knn_graph_sharded_two_ranks_bit_exact(tmp_path):
    """Rows of the retrieval matrix sharded over two ranks (HIP retrieval kernel on each): the gathered int64 table is
    bit-identical to the single-process graph and to the CPU oracle's (csa_models.py:270-280)."""
    from csn_amd import functional as CF
    world = len(SHARDS)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    f = _feats()
    single = CF.retrieval_measure(f.cuda(), f.cuda()).topk(K + 1, dim=-1)[1].cpu()
    ref = orc.knn_graph(f, f, K)
    assert torch.equal(single, ref)
    for r in range(world):
        got = torch.load(os.path.join(tmp_path, f"g{r}.pt"))
        assert got.dtype == torch.int64 and got.shape == (sum(SHARDS), K + 1)
        assert torch.equal(got, single)


# ---- the CSA step itself, sharded by query shape: neighbour-only exchange, overlapped evaluation, descriptor reuse -------
CSA_B, CSA_K, CSA_N, CSA_CLS = 2, 2, 400, 6
CSA_GEO = dict(block=100, n_blocks=4)


def _csa_collection(world, B=None):
    """A small collection with per-shape channel offsets (well-conditioned compatibility gradients, see
    oracle.conditioned_csa_case) + one set of weights."""
    rng = np.random.default_rng(23)
    S = (B or CSA_B) * world
    p = orc.make_params(rng, 1, n_cls=CSA_CLS, csa=True)
    p["attention.fc.weight"] = p["attention.fc.weight"] * 4.0
    feats = orc.synth_points(rng, (S, 256, CSA_N)) + orc.synth_points(rng, (S, 256, 1))
    labels = orc.synth_labels(rng, S, CSA_N, CSA_CLS)
    return p, feats, labels


def _csa_worker(rank, world, port, out_dir, mode, backend="gloo", B=None):
    B = B or CSA_B
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from csn_amd import _lib
    from csn_amd.csa_models import get_model
    from csn_amd.sharding import ShapeGraphShard, regular_graph
    _lib.check(_lib.lib().csn_set_math_mode(mode))
    p, feats, labels = _csa_collection(world, B)
    shard = ShapeGraphShard(regular_graph(B * world, CSA_K), B, rank, world, torch.device("cuda"))
    lo, hi = shard.first, shard.first + B
    mine, lab = feats[lo:hi].cuda(), labels[lo:hi].cuda()
    out = {}
    for tag, kw in (("reuse", dict(mode="alltoall", reuse_descriptors=True)),
                    ("a2a", dict(mode="alltoall", reuse_descriptors=False)),
                    ("gather", dict(mode="allgather", reuse_descriptors=False))):
        model = get_model("csa", CSA_CLS, 1, CSA_K, **CSA_GEO)
        model.load_state_dict(p, strict=False)
        model = model.cuda().eval()
        logits = model(mine.unsqueeze(-1), "test", shard.exchange_async(mine, **kw))
        loss = orc.masked_ce_loss(logits, lab)
        loss.backward()
        params = [q for n, q in model.named_parameters() if q.grad is not None]
        shard.allreduce_grads(params)
        out[tag] = {"logits": logits.detach().cpu(), "loss": loss.item(),
                    "grads": {n: q.grad.cpu() for n, q in model.named_parameters() if q.grad is not None}}
    torch.save(out, os.path.join(out_dir, f"csa{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", [0, 1], ids=["fp32", "bf16x3"])
def test_sharded_csa_step_with_descriptor_reuse_equals_single_process(tmp_path, mode):
    """Two ranks (sharing the test box's GPU) run the CSA step on their halves of a 4-shape collection three ways —
    neighbour-only all-to-all with the neighbours' pooled descriptors taken from their owners, the same without reuse, and
    the all-gather fallback — and each must reproduce the single-process module on the whole collection: logits, loss and,
    after the gradient all-reduce, all 11 weight gradients (eval mode: the arithmetic is deterministic)."""
    from csn_amd import _lib
    from csn_amd.csa_models import get_model
    from csn_amd.sharding import ShapeGraphShard, regular_graph
    world = 2
    mp.spawn(_csa_worker, args=(world, _free_port(), str(tmp_path), mode), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"csa{r}.pt")) for r in range(world)]
    _lib.check(_lib.lib().csn_set_math_mode(mode))
    try:
        p, feats, labels = _csa_collection(world)
        graph = regular_graph(CSA_B * world, CSA_K)
        model = get_model("csa", CSA_CLS, 1, CSA_K, **CSA_GEO)
        model.load_state_dict(p, strict=False)
        model = model.cuda().eval()
        total, ref_logits = 0.0, []
        for r in range(world):
            sh = ShapeGraphShard(graph, CSA_B, r, world, torch.device("cpu"))
            lo, hi = sh.first, sh.first + CSA_B
            stack = sh.neighbour_stack(feats[lo:hi], feats)
            logits = model(feats[lo:hi].cuda().unsqueeze(-1), "test", stack.cuda())
            loss = orc.masked_ce_loss(logits, labels[lo:hi].cuda())
            ref_logits.append((logits.detach().cpu(), loss.item()))
            total = total + loss / world
        total.backward()
        ref_grads = {n: q.grad.cpu() for n, q in model.named_parameters() if q.grad is not None}
        assert len(ref_grads) == 11
        for tag in ("reuse", "a2a", "gather"):
            for r in range(world):
                got = res[r][tag]
                assert (got["logits"] - ref_logits[r][0]).abs().max().item() < 2e-5, tag
                assert abs(got["loss"] - ref_logits[r][1]) < 1e-5, tag
                assert set(got["grads"]) == set(ref_grads)
                for n, g in ref_grads.items():
                    assert (got["grads"][n] - g).abs().max().item() <= 2e-4 * g.abs().max().item(), (tag, n)
    finally:
        _lib.lib().csn_set_math_mode(1)


def test_sharded_step_over_rccl_world_of_one(tmp_path):
    """The sharded step with backend "nccl" (= RCCL) in a world of one rank: every collective of the path — the neighbour
    all_to_all_single with explicit splits, the all-gather fallback, the descriptor gather and its gradient, the one-bucket
    gradient all-reduce — is issued on the device through RCCL as the multi-GPU job issues it, and the step equals the plain
    module.  (What a one-GPU box can check of the RCCL path: call forms, dtypes, contiguity, stream hand-over.)"""
    from csn_amd import _lib
    from csn_amd.csa_models import get_model
    from csn_amd.sharding import ShapeGraphShard, regular_graph
    world, mode, B = 1, 1, 4
    mp.spawn(_csa_worker, args=(world, _free_port(), str(tmp_path), mode, "nccl", B), nprocs=world, join=True)
    res = torch.load(os.path.join(tmp_path, "csa0.pt"))
    _lib.check(_lib.lib().csn_set_math_mode(mode))
    p, feats, labels = _csa_collection(world, B)
    model = get_model("csa", CSA_CLS, 1, CSA_K, **CSA_GEO)
    model.load_state_dict(p, strict=False)
    model = model.cuda().eval()
    sh = ShapeGraphShard(regular_graph(B, CSA_K), B, 0, 1, torch.device("cpu"))
    stack = sh.neighbour_stack(feats, feats)
    logits = model(feats.cuda().unsqueeze(-1), "test", stack.cuda())
    loss = orc.masked_ce_loss(logits, labels.cuda())
    loss.backward()
    ref = {n: q.grad.cpu() for n, q in model.named_parameters() if q.grad is not None}
    for tag in ("reuse", "a2a", "gather"):
        got = res[tag]
        assert (got["logits"] - logits.detach().cpu()).abs().max().item() < 2e-5, tag
        assert abs(got["loss"] - loss.item()) < 1e-5, tag
        for n, g in ref.items():
            assert (got["grads"][n] - g).abs().max().item() <= 2e-4 * g.abs().max().item(), (tag, n)
