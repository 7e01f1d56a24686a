"""SURVEY.md §8f row 4 on the device: the feature-file data path feeding the drop-in module.  A DeviceFeatureCache on the
MI355X (every file read once, neighbour stacks by indexed gather out of HBM) against a DataLoader over CSADatasetK (K
np.load's per item, features_data_loader.py:107-140) THROUGH CrossShapeAt; the per-rank caches and their shards; and
mini-batch training out of a ResidentCollection against train_layers over the loader (csa_training.py:191-222)."""
import os

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

from oracle import csa_oracle as orc

pytestmark = pytest.mark.gpu

NP, K, N_CLS = 1000, 2, 6
GEO = dict(block=200, n_blocks=5)           # (blocks are multiples of 4 points)


def _files(root, n_shapes, rng, sizes):
    os.makedirs(os.path.join(root, "fc_1"))
    os.makedirs(os.path.join(root, "point_labels"))
    for i in range(n_shapes):
        n = sizes[i % len(sizes)]
        np.save(os.path.join(root, "fc_1", f"s{i:02d}.npy"), rng.standard_normal((1, 256, n, 1)).astype(np.float32))
        np.save(os.path.join(root, "point_labels", f"s{i:02d}.npy"), rng.integers(0, N_CLS, size=(n,)))


@pytest.fixture()
def dataset(tmp_path, monkeypatch):
    from csn_amd import data as D
    monkeypatch.setattr(os, "listdir", lambda p, _ls=os.listdir: sorted(_ls(p)))
    rng = np.random.default_rng(5)
    root = str(tmp_path / "train")
    _files(root, 6, rng, sizes=(NP, 700, 510))                                   # two kinds get wrap-around padded
    graph = np.array([[0, 3, 5], [1, 0, 2], [4, 2, 1], [3, 1, 0], [4, 5, 0], [5, 2, 3]])
    return D.CSADatasetK(root, root, graph, K, n_points=NP), graph


def _model(train=False):
    from csn_amd.csa_models import get_model
    torch.manual_seed(3)
    m = get_model("csa", N_CLS, 1, K, **GEO).cuda()
    return m.train(train)


def test_device_cache_feeds_the_module_like_the_loader(dataset):
    from csn_amd import data as D
    ds, graph = dataset
    cache = D.DeviceFeatureCache(ds, "cuda", n_points=NP)
    assert cache.feats.is_cuda and cache.feats.shape == (6, 256, NP)
    table = D.neighbour_table(graph, K)
    model = _model()
    model.trust_neighbor_slot0 = True
    with torch.no_grad():
        for (f, lab, nb), (cf, cl, cnb) in zip(DataLoader(ds, 3, shuffle=False), cache.batches(3, table)):
            assert torch.equal(f.cuda(), cf) and torch.equal(lab.cuda(), cl) and torch.equal(nb.cuda(), cnb)
            ref = model(f.cuda(), "test", nb.cuda())                              # the loader's stack, moved to the device
            got = model(cf, "test", cnb)                                          # the cache's stack, gathered on the device
            assert torch.equal(ref, got)
            host = model(f.cuda(), "test", nb)                                    # as csa_training.py:198-202 hands it over: on the CPU
            assert (host - got).abs().max().item() < 2e-5


def test_rank_caches_and_their_shards(dataset):
    from csn_amd import data as D
    ds, graph = dataset
    table = D.neighbour_table(graph, K)
    whole = D.DeviceFeatureCache(ds, "cuda", n_points=NP)
    world = 2
    parts = [D.DeviceFeatureCache.for_rank(ds, "cuda", r, world, n_points=NP) for r in range(world)]
    assert [p.first for p in parts] == [0, 3] and all(len(p) == 3 for p in parts)
    assert torch.equal(torch.cat([p.feats for p in parts]), whole.feats) and torch.equal(torch.cat([p.labels for p in parts]), whole.labels)
    with pytest.raises(IndexError):
        parts[1].batch([0])                                                       # a shape of the other rank
    with pytest.raises(ValueError):
        D.DeviceFeatureCache.for_rank(ds, "cuda", 0, 4, n_points=NP)               # 6 shapes do not split over 4 ranks
    for r, part in enumerate(parts):
        shard = part.shard(table, r, world)                                        # the ShapeGraphShard its exchange feeds
        assert (shard.first, shard.B, shard.K, shard.S) == (part.first, 3, K, 6)
        ids = np.arange(part.first, part.first + 3)
        want = whole.neighbour_stack(ids, table)
        assert torch.equal(shard.neighbour_stack(part.feats, whole.feats), want)   # the all-gathered collection indexed by the shard


class _Ready:
    """a neighbour stack that is already complete, in the form the module takes a pending one (same evaluation order)"""
    reuse_descriptors = False

    def __init__(self, stack):
        self.stack = stack

    def wait(self):
        return self.stack


def test_minibatch_training_out_of_the_resident_collection(dataset):
    from csn_amd import data as D, training as T
    from csn_amd.sharding import ResidentCollection
    ds, graph = dataset
    table = D.neighbour_table(graph, K)
    coll = ResidentCollection.from_source(ds, table, "cuda", 0, 1, n_points=NP)
    steps = coll.epoch_batches(2, epoch=0, shuffle=False)
    assert [b[0].tolist() for b in steps] == [[0, 1], [2, 3], [4, 5]]
    outs = []
    for which in ("collection", "loader"):
        model = _model(train=True)
        opt, _ = T.make_optimizer(model)
        if which == "collection":
            torch.manual_seed(17)                                                  # the dropout seeds of both runs
            loss = T.train_layers_sharded(model, coll, opt, N_CLS, 2, epoch=0, shuffle=False)
        else:
            batches = [(f.cuda(), lab.cuda(), _Ready(nb.cuda().contiguous())) for f, lab, nb in DataLoader(ds, 2, shuffle=False)]
            torch.manual_seed(17)                                                  # (after the loader: it draws from the generator)
            loss = T.train_layers(model, batches, opt, N_CLS, "cuda")
        outs.append((loss, [p.detach().clone() for n, p in model.named_parameters() if not n.startswith("fc_1")]))
    (l0, p0), (l1, p1) = outs
    assert abs(l0 - l1) < 1e-6
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)                                                   # same batches, same masks, same kernels
