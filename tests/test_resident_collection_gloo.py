"""Mini-batch training over a collection that is resident across ranks (csn_amd.sharding.ResidentCollection +
csn_amd.training.train_layers_sharded) on CPU with gloo, world 2 and 4 with UNEVEN ownership: every step's neighbour stack
must be the tensor a DataLoader over CSADatasetK collates for that batch (features_data_loader.py:107-140: slot 0 = the
shape, slots 1..K its neighbours in graph order), and a whole epoch must leave every rank with the parameters of the
single-process loop over the same batches (csa_training.py:191-222, gradients averaged over the ranks of a step).
The compute function is the CPU oracle at a tiny size; what is under test is the host logic: ownership ranges, the sampler
every rank evaluates for all ranks, the per-step exchange plan and its cache, the all-to-all with per-step splits, the
stack assembly from [own cache | received], and the gradient bucket."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import csa_oracle as orc

S, K, C, N, N_CLS, BATCH = 7, 2, 256, 32, 5, 2
KW = dict(d_k=32, d_v=32, block=16, n_blocks=2)


class _Shapes:
    """in-memory dataset with the FeaturesDataset item contract: (feats (1, C, N, 1), label (N,))"""

    def __init__(self):
        rng = np.random.default_rng(123)
        self.feats = orc.synth_points(rng, (S, C, N))
        self.labels = orc.synth_labels(rng, S, N, N_CLS)

    def __len__(self):
        return S

    def __getitem__(self, i):
        return self.feats[i][None, :, :, None], self.labels[i]


class _OracleCSA(torch.nn.Module):
    """the oracle's CSA forward behind the drop-in call signature model(feats, mode, neighbours-or-pending)"""

    def __init__(self):
        super().__init__()
        p = orc.make_params(np.random.default_rng(7), 1, d_model=C, d_k=32, d_v=32, n_cls=N_CLS, csa=True)
        self.names = list(p)
        self.p = torch.nn.ParameterList([torch.nn.Parameter(v.clone()) for v in p.values()])

    def forward(self, feats, mode, nbrs):
        stack = nbrs.wait() if callable(getattr(nbrs, "wait", None)) else nbrs
        return orc.forward_csa(feats, stack, dict(zip(self.names, self.p)), 1, **KW)


def _table():
    from csn_amd.sharding import regular_graph
    return regular_graph(S, K, seed=11)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    from csn_amd.sharding import ResidentCollection
    from csn_amd.training import train_layers_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    src, table = _Shapes(), _table()
    coll = ResidentCollection.from_source(src, table, "cpu", rank, world, n_points=N)
    lo, hi = int(coll.bounds[rank]), int(coll.bounds[rank + 1])
    assert len(coll.cache) == hi - lo and torch.equal(coll.cache.feats, src.feats[lo:hi])      # only the own share is resident
    steps = coll.epoch_batches(BATCH, epoch=0, shuffle=True, seed=3)
    assert len(steps) == max(-(-int(n) // BATCH) for n in np.diff(coll.bounds))
    for batches in steps + steps[:1]:                       # (the repeated step comes out of the plan cache)
        assert all((coll.owner(b) == r).all() and len(b) == BATCH for r, b in enumerate(batches))
        plan = coll.plan(batches)
        assert coll.plan(batches) is plan
        ids = batches[rank]
        f, lab = coll.batch(plan)
        assert torch.equal(f[..., 0], src.feats[ids]) and torch.equal(lab, src.labels[ids])
        stack = coll.exchange_async(plan).wait()
        want = src.feats[np.concatenate((ids[:, None], table[ids]), axis=1)]                    # (B, K+1, C, N)
        assert torch.equal(stack[..., 0], want)
        assert plan.n_recv <= BATCH * K and sum(plan.recv_splits) == plan.n_recv
    torch.manual_seed(0)
    model = _OracleCSA()
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    loss = train_layers_sharded(model, coll, opt, N_CLS, BATCH, epoch=0, shuffle=True, seed=3)
    torch.save({"loss": loss, "params": [p.detach().clone() for p in model.parameters()], "steps": steps},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_minibatch_epoch_equals_the_single_process_loop(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"r{r}.pt"), weights_only=False) for r in range(world)]
    from csn_amd.training import loss_functions_seg
    src, table = _Shapes(), _table()
    steps = res[0]["steps"]
    for r in range(1, world):                               # every rank derived the same batches for all ranks
        assert all(np.array_equal(a, b) for sa, sb in zip(steps, res[r]["steps"]) for a, b in zip(sa, sb))
    # single process: per step, the losses of all ranks' batches, gradients averaged, one optimizer step
    torch.manual_seed(0)
    model = _OracleCSA()
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    losses = np.zeros(world)
    for batches in steps:
        opt.zero_grad()
        for r, ids in enumerate(batches):
            stack = src.feats[np.concatenate((ids[:, None], table[ids]), axis=1)].unsqueeze(-1)
            out = model(src.feats[ids].unsqueeze(-1), "train", stack)
            loss, _ = loss_functions_seg(out, src.labels[ids], N_CLS)
            (loss / world).backward()
            losses[r] += loss.item() / len(steps)
        opt.step()
    for r in range(world):
        assert abs(res[r]["loss"] - losses[r]) < 1e-5
        for got, ref, p0 in zip(res[r]["params"], model.parameters(), res[0]["params"]):
            assert torch.equal(got, p0)                      # replicas stay identical
            assert torch.allclose(got, ref.detach(), rtol=1e-4, atol=1e-6)


def test_ownership_ranges_and_rejections():
    from csn_amd.data import DeviceFeatureCache
    from csn_amd.sharding import ResidentCollection
    assert ResidentCollection.split_bounds(7, 4).tolist() == [0, 2, 4, 6, 7]
    assert ResidentCollection.split_bounds(8, 2).tolist() == [0, 4, 8]
    src, table = _Shapes(), _table()
    with pytest.raises(ValueError):                          # a cache that is not the rank's range
        ResidentCollection(DeviceFeatureCache(src, "cpu", first=0, count=3, n_points=N), table, 0, 2)
    coll = ResidentCollection.from_source(src, table, "cpu", 0, 1, n_points=N)
    with pytest.raises(ValueError):
        coll.plan([np.array([0, 1]), np.array([2, 3])])      # two batches for one rank
    coll2 = ResidentCollection(DeviceFeatureCache(src, "cpu", first=0, count=4, n_points=N), table, 0, 2)
    with pytest.raises(ValueError):
        coll2.plan([np.array([5, 6]), np.array([4, 5])])     # rank 0's batch holds shapes of rank 1
    # world 1: the stack straight out of the cache, no collective
    plan = coll.plan([np.array([3, 6])])
    want = src.feats[np.concatenate((np.array([3, 6])[:, None], table[[3, 6]]), axis=1)]
    assert torch.equal(coll.neighbour_stack(plan)[..., 0], want)
