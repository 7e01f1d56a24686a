"""Multi-process pieces on the MI355X: two ranks share the one GPU of the test box (gloo carries the collectives; RCCL
needs one GPU per rank and is exercised by the driver's multi-GPU bench only)."""
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
