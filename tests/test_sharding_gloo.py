"""N > 1 path on CPU: world_size-2 gloo run of the shape-graph sharding (csn_amd/sharding.py) must reproduce the
single-process result.  The compute function is the CPU oracle at a tiny size (the HIP kernels cannot run here);
what is under test is the host logic: ownership ranges, the feature all-gather, neighbour-stack assembly in graph
order with slot 0 = self, and the one-bucket gradient all-reduce."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import csa_oracle as orc

B_PER_RANK, K, C, N, N_CLS = 2, 2, 256, 32, 5
KW = dict(d_k=32, d_v=32, block=16, n_blocks=2)


def _collection(world):
    rng = np.random.default_rng(77)
    S = B_PER_RANK * world
    p = orc.make_params(rng, 1, d_model=C, d_k=32, d_v=32, n_cls=N_CLS, csa=True)
    feats = orc.synth_points(rng, (S, C, N))
    labels = orc.synth_labels(rng, S, N, N_CLS)
    return p, feats, labels


def _loss(p, x_stack, feats, labels, neighbour_pooled=None):
    logits = orc.forward_csa(feats.unsqueeze(-1), x_stack, p, 1, neighbour_pooled=neighbour_pooled, **KW)
    return orc.masked_ce_loss(logits, labels)


def _worker(rank, world, port, out_dir):
    from csn_amd.sharding import ShapeGraphShard, regular_graph
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    p, feats, labels = _collection(world)
    graph = regular_graph(B_PER_RANK * world, K)
    shard = ShapeGraphShard(graph, B_PER_RANK, rank, world, torch.device("cpu"))
    lo, hi = shard.first, shard.first + B_PER_RANK
    mine = feats[lo:hi].clone()
    params = {k: torch.nn.Parameter(v.clone()) for k, v in p.items()}
    gathered = shard.exchange(mine)
    assert torch.equal(gathered, feats)                                   # every rank sees the whole collection
    stack = shard.neighbour_stack(mine, gathered)
    assert stack.shape == (B_PER_RANK, K + 1, C, N, 1)
    assert torch.equal(stack[:, 0, :, :, 0], mine)                        # slot 0 = the shape itself
    for b in range(B_PER_RANK):
        for k in range(K):
            assert torch.equal(stack[b, k + 1, :, :, 0], feats[graph[lo + b, k]])
    assert torch.equal(shard.exchange_neighbours(mine), stack)             # neighbour-only all-to-all builds the same stack
    for mode in ("alltoall", "allgather"):                                 # the overlapped forms: the model calls wait() late
        pending = shard.exchange_async(mine, mode=mode)
        assert torch.equal(pending.wait(), stack) and pending.wait() is pending.wait()
    loss = _loss(params, stack, mine, labels[lo:hi])
    loss.backward()
    shard.allreduce_grads(params.values(), average=True)
    out = {"loss": loss.item(), "grads": {k: v.grad.clone() for k, v in params.items()}}
    # descriptor reuse: every shape's pooled SSA descriptor comes from its owner (differentiable all-gather) — same loss,
    # and after the gradient all-reduce the same weight gradients, with K fewer evaluations per shape on every rank
    for v in params.values():
        v.grad = None
    pending = shard.exchange_async(mine, reuse_descriptors=True)
    loss_r = _loss(params, pending.wait(), mine, labels[lo:hi], neighbour_pooled=pending.gather_pooled)
    loss_r.backward()
    shard.allreduce_grads(params.values(), average=True)
    out["loss_reuse"] = loss_r.item()
    out["grads_reuse"] = {k: v.grad.clone() for k, v in params.items()}
    torch.save(out, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_sharded_step_equals_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]
    # single process: the same two per-rank losses, averaged
    from csn_amd.sharding import ShapeGraphShard, regular_graph
    p, feats, labels = _collection(world)
    graph = regular_graph(B_PER_RANK * world, K)
    params = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    total = 0.0
    for r in range(world):
        sh = ShapeGraphShard(graph, B_PER_RANK, r, world, torch.device("cpu"))
        lo, hi = sh.first, sh.first + B_PER_RANK
        stack = sh.neighbour_stack(feats[lo:hi], feats)
        loss = _loss(params, stack, feats[lo:hi], labels[lo:hi])
        assert abs(loss.item() - res[r]["loss"]) < 1e-6
        total = total + loss / world
    total.backward()
    for k, v in params.items():
        for r in range(world):
            got = res[r]["grads"][k]
            assert torch.allclose(got, v.grad, rtol=1e-4, atol=1e-7), k
        assert torch.equal(res[0]["grads"][k], res[1]["grads"][k])         # ranks agree bit-for-bit after the all-reduce
        for r in range(world):                                             # descriptor reuse: same averaged gradients
            assert torch.allclose(res[r]["grads_reuse"][k], v.grad, rtol=2e-4, atol=2e-7), k
    for r in range(world):
        assert abs(res[r]["loss_reuse"] - res[r]["loss"]) < 1e-6


def _a2a_worker(rank, world, port):
    from csn_amd.sharding import ShapeGraphShard, regular_graph
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    B, Kn = 3, 3
    S = B * world
    feats = torch.arange(S * 4 * 5, dtype=torch.float32).reshape(S, 4, 5)
    graph = regular_graph(S, Kn, seed=5)
    shard = ShapeGraphShard(graph, B, rank, world, torch.device("cpu"))
    mine = feats[shard.first:shard.first + B].clone()
    for _ in range(2):                                                      # buffers are reused between steps
        stack = shard.exchange_neighbours(mine).clone()                     # (both calls fill the shard's one stack buffer)
        ref = shard.neighbour_stack(mine, feats)
        assert torch.equal(stack, ref)
    assert sum(shard._recv_splits) <= min(B * Kn, S - B)
    # the same exchange with a bf16 payload (bench.py --math bf16): half the bytes on the wire, remote slots arrive rounded to
    # bf16 and widened again, own slots (never sent) stay exact; a gradient-free parameter joins the bucket as zeros
    feats2 = feats * 1.001 + 0.37                                              # (values that bf16 does not hold exactly)
    mine2 = feats2[shard.first:shard.first + B].clone()
    pend = shard.exchange_async(mine2, payload_dtype=torch.bfloat16, reuse_descriptors=False)
    stack = pend.wait().clone()
    ref = shard.neighbour_stack(mine2, feats2)
    own = torch.from_numpy((graph[shard.first:shard.first + B] >= shard.first) & (graph[shard.first:shard.first + B] < shard.first + B))
    assert stack.dtype == torch.float32 and torch.equal(stack[:, 0], ref[:, 0])
    for b in range(B):
        for k in range(Kn):
            want = ref[b, k + 1] if own[b, k] else ref[b, k + 1].bfloat16().float()
            assert torch.equal(stack[b, k + 1], want)
    sent, recvd = shard.payload_bytes
    assert sent == sum(shard._send_splits) * 4 * 5 * 2 and recvd == sum(shard._recv_splits) * 4 * 5 * 2
    pa, pb = torch.nn.Parameter(torch.zeros(2)), torch.nn.Parameter(torch.zeros(3))
    pa.grad = torch.full((2,), float(rank + 1))
    if rank == 0:
        pb.grad = torch.ones(3)                                                # only rank 0 has a gradient for pb
    pc = torch.nn.Parameter(torch.zeros(4))                                    # no gradient on ANY rank: stays None, as at N = 1
    shard.allreduce_grads([pa, pb, pc], average=False)
    assert torch.equal(pa.grad, torch.full((2,), float(sum(range(1, world + 1))))) and torch.equal(pb.grad, torch.ones(3))
    assert pc.grad is None
    dist.barrier()
    dist.destroy_process_group()


def _a2a8_worker(rank, world, port):
    """config 4's world size: the asynchronous neighbour-only exchange, the descriptor table and the gradient bucket."""
    from csn_amd.sharding import ShapeGraphShard, regular_graph
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    B, Kn, C = 4, 3, 6
    S = B * world
    feats = torch.arange(S * C * 8, dtype=torch.float32).reshape(S, C, 8)
    graph = regular_graph(S, Kn)
    shard = ShapeGraphShard(graph, B, rank, world, torch.device("cpu"))
    mine = feats[shard.first:shard.first + B].clone()
    for mode in ("alltoall", "allgather"):
        pending = shard.exchange_async(mine, mode=mode, reuse_descriptors=True)
        # descriptors from their owners: row b, k = the pooled map of shape graph[first + b][k]; gradients sum back to the owners
        own = mine.mean(dim=2).clone().requires_grad_(True)
        table = pending.gather_pooled(own)
        want = feats.mean(dim=2)[torch.from_numpy(graph[shard.first:shard.first + B])]
        assert torch.equal(table.detach(), want)
        table.sum().backward()
        uses = torch.from_numpy((graph.reshape(-1)[None, :] == torch.arange(shard.first, shard.first + B).numpy()[:, None]).sum(axis=1))
        assert torch.equal(own.grad, uses.float()[:, None].expand(B, C))       # one unit of gradient per use, from every rank
        stack = pending.wait().clone()
        assert torch.equal(stack, shard.neighbour_stack(mine, feats))
    assert sum(shard._recv_splits) <= B * Kn
    g = [torch.nn.Parameter(torch.zeros(3))]
    g[0].grad = torch.full((3,), float(rank))
    shard.allreduce_grads(g)
    assert torch.allclose(g[0].grad, torch.full((3,), (world - 1) / 2))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_exchange_descriptor_table_and_gradient_bucket():
    """The collectives of the N = 8 run (BASELINE configs[3]) on 8 gloo ranks with small maps: all_to_all_single with
    uneven splits in flight, the differentiable descriptor all-gather, the one-bucket gradient all-reduce."""
    mp.spawn(_a2a8_worker, args=(8, _free_port()), nprocs=8, join=True)


def test_neighbour_only_exchange_four_ranks():
    """all_to_all_single with uneven splits over 4 gloo ranks: every rank receives only the shapes its graph rows name."""
    mp.spawn(_a2a_worker, args=(4, _free_port()), nprocs=4, join=True)


def test_graph_rejects_self_neighbours_and_bad_sizes():
    from csn_amd.sharding import ShapeGraphShard, regular_graph
    g = regular_graph(8, 3)
    assert g.shape == (8, 3) and g.dtype == np.int64
    assert all(s not in g[s] and len(set(g[s])) == 3 for s in range(8))
    with pytest.raises(ValueError):
        ShapeGraphShard(g, 3, 0, 2, torch.device("cpu"))
    bad = g.copy()
    bad[1, 0] = 1
    with pytest.raises(ValueError):
        ShapeGraphShard(bad, 4, 0, 2, torch.device("cpu"))


# ---- kNN graph build sharded by query shape (SURVEY §8e collective 4) ----------------------------------------------------
_KNN_Q, _KNN_C = (3, 2, 1), (2, 2, 3)            # query / candidate shapes per rank (uneven on purpose)


def _knn_sets():
    rng = np.random.default_rng(91)
    return orc.synth_clustered_feats(rng, sum(_KNN_Q), 40, C=32), orc.synth_clustered_feats(rng, sum(_KNN_C), 52, C=32)


def _knn_worker(rank, world, port, out_dir):
    from csn_amd.sharding import knn_graph_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    q, c = _knn_sets()
    q0, c0 = sum(_KNN_Q[:rank]), sum(_KNN_C[:rank])
    ql, cl = q[q0:q0 + _KNN_Q[rank]].clone(), c[c0:c0 + _KNN_C[rank]].clone()
    g_qc = knn_graph_sharded(ql, 2, orc.retrieval_measure, cand_local=cl, pair_budget=7 * 40)   # one query row per chunk
    g_qq = knn_graph_sharded(ql, 2, orc.retrieval_measure)                                      # queries = candidates
    torch.save({"qc": g_qc, "qq": g_qq}, os.path.join(out_dir, f"knn{rank}.pt"))
    with pytest.raises(ValueError):
        knn_graph_sharded(ql, 7, orc.retrieval_measure, cand_local=cl)                          # 7 candidates, 8 wanted
    dist.barrier()
    dist.destroy_process_group()


def test_knn_graph_sharded_three_ranks_uneven(tmp_path):
    """Every rank scores its own query rows against the all-gathered candidates; the gathered index table equals the
    single-process get_knn_graph bit for bit, on every rank, with uneven shards."""
    world = 3
    mp.spawn(_knn_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    q, c = _knn_sets()
    ref_qc, ref_qq = orc.knn_graph(q, c, 2), orc.knn_graph(q, q, 2)
    assert ref_qc.dtype == torch.int64 and ref_qc.shape == (sum(_KNN_Q), 3)
    assert torch.equal(ref_qq[:, 0], torch.arange(sum(_KNN_Q)))                      # a shape retrieves itself first
    for r in range(world):
        got = torch.load(os.path.join(tmp_path, f"knn{r}.pt"))
        assert got["qc"].dtype == torch.int64 and torch.equal(got["qc"], ref_qc)
        assert torch.equal(got["qq"], ref_qq)
