"""Feature-file data path of the CSA layer (MID-FC/features_data_loader.py), restated.

On-disk format (produced by the O-CNN extraction, ocnn_extraction/tfsolver.py:197-268): per shape one
``fc_1/<name>.npy`` float32 ``(1, 256, n, 1)`` and one ``point_labels/<name>.npy`` int ``(n,)``.  Shapes with fewer
than 10000 points are wrap-around padded to 10000 (features_data_loader.py:37-43).  ``CSADatasetK`` stacks a shape
with its K nearest shapes from a kNN graph, slot 0 = the shape itself (features_data_loader.py:107-140).
A synthetic in-memory variant with the same item contract is provided for tests and benchmarks.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch
from torch.utils.data import Dataset

N_POINTS = 10000


def pad_points(feats: np.ndarray, label: Optional[np.ndarray] = None, n_points: int = N_POINTS):
    """Wrap-around padding along the point axis (features_data_loader.py:37-43, :101-104)."""
    n = feats.shape[2]
    if n < n_points:
        rem = n_points - n
        feats = np.concatenate((feats, feats[:, :, :rem]), axis=2)
        if label is not None:
            label = np.concatenate((label, label[:rem]), axis=0)
    return feats, label


class FeaturesDataset(Dataset):
    """(feats (1, 256, N, 1), label (N,)) per shape (features_data_loader.py:9-48)."""

    def __init__(self, dataroot: str, attention_type: str = "backbone_fc_ssa_logit", n_points: int = N_POINTS):
        self.features_dir = os.path.join(dataroot, "fc_1")
        self.labels_dir = os.path.join(dataroot, "point_labels")
        self.files = os.listdir(self.features_dir)
        self.n_points = n_points                 # the reference pads to 10000 (features_data_loader.py:37); other geometries differ

    def __len__(self):
        return len(self.files)

    def load(self, name: str):
        feats = np.load(os.path.join(self.features_dir, name))
        label = np.load(os.path.join(self.labels_dir, name)).astype(int)
        return pad_points(feats, label, self.n_points)

    def __getitem__(self, idx):
        feats, label = self.load(self.files[idx])
        return torch.from_numpy(feats), torch.from_numpy(label)


class CSADatasetK(Dataset):
    """(feats (256, N, 1), label (N,), neighbor_feats (K+1, 256, N, 1)); neighbours come from ``dataroot_K`` in kNN-graph
    order, skipping the shape itself, slot 0 = the shape (features_data_loader.py:79-140)."""

    def __init__(self, dataroot: str, dataroot_K: str, knn_graph, K: int, n_points: int = N_POINTS):
        self.own = FeaturesDataset(dataroot, n_points=n_points)
        self.nbr = FeaturesDataset(dataroot_K, n_points=n_points)
        self.K = K
        self.knn_graph = np.copy(knn_graph)

    def __len__(self):
        return len(self.own)

    def __getitem__(self, idx):
        feats, label = self.own.load(self.own.files[idx])
        stack = [feats]
        for kidx in self.knn_graph[idx]:
            if kidx != idx:
                stack.append(pad_points(np.load(os.path.join(self.nbr.features_dir, self.nbr.files[kidx])), None, self.nbr.n_points)[0])
            if len(stack) == self.K + 1:
                break
        nb = torch.from_numpy(np.array(stack))
        return torch.from_numpy(feats).squeeze(0), torch.from_numpy(label), nb.squeeze(1)


class SyntheticShapes(Dataset):
    """In-memory stand-in with the item contract of FeaturesDataset / CSADatasetK: clustered random features whose part
    labels are a (noisy) function of the features, so a few optimisation steps measurably reduce the loss."""

    def __init__(self, n_shapes: int, n_cls: int, K: Optional[int] = None, knn_graph=None, seed: int = 0,
                 n_points: int = N_POINTS, channels: int = 256, neighbor_source: "Optional[SyntheticShapes]" = None):
        rng = np.random.default_rng(seed)
        proto = rng.standard_normal(size=(n_cls, channels)).astype(np.float32)
        self.labels = rng.integers(1, n_cls, size=(n_shapes, n_points))
        noise = rng.standard_normal(size=(n_shapes, n_points, channels)).astype(np.float32)
        self.feats = np.ascontiguousarray((0.6 * proto[self.labels] + noise).transpose(0, 2, 1))[..., None]   # (S, C, N, 1)
        self.labels[rng.random(size=self.labels.shape) < 0.1] = 0
        self.K = K
        self.knn_graph = None if knn_graph is None else np.asarray(knn_graph)
        self.neighbor_source = neighbor_source or self          # test shapes take their neighbours from the train set

    def __len__(self):
        return len(self.feats)

    def __getitem__(self, idx):
        f, l = torch.from_numpy(self.feats[idx]), torch.from_numpy(self.labels[idx].astype(np.int64))
        if self.K is None:
            return f[None], l                                                     # (1, C, N, 1) like FeaturesDataset
        stack = [f]
        for kidx in self.knn_graph[idx]:
            if kidx != idx:                                     # same id test as features_data_loader.py:127
                stack.append(torch.from_numpy(self.neighbor_source.feats[kidx]))
            if len(stack) == self.K + 1:
                break
        return f, l, torch.stack(stack)


# ---------------------------------------------------------------------------------------------------------
# device-resident feature cache: the K np.load's per item of CSADatasetK become index arithmetic
# ---------------------------------------------------------------------------------------------------------
def neighbour_table(knn_graph, K: int) -> np.ndarray:
    """(S, K) int64 neighbour ids per shape out of a kNN graph (S, >= K+1) exactly as CSADatasetK.__getitem__ picks them
    (features_data_loader.py:124-135): walk the graph row in order, skip the shape itself, stop at K neighbours."""
    g = np.asarray(knn_graph)
    out = np.empty((g.shape[0], K), dtype=np.int64)
    for idx, row in enumerate(g):
        picked = [int(k) for k in row if int(k) != idx][:K]
        if len(picked) < K:
            raise ValueError(f"graph row {idx} holds fewer than {K} neighbours besides the shape itself")
        out[idx] = picked
    return out


class DeviceFeatureCache:
    """All shapes of a feature dataset resident in device memory, keyed by shape id: ``feats`` (S_local, C, N) fp32 and
    ``labels`` (S_local, N) int64 for the ids [first, first + S_local) this process owns (the whole dataset by default; one
    contiguous range per rank when the collection is sharded, the same ranges ShapeGraphShard uses).

    The reference's CSADatasetK re-reads K neighbour files from disk for EVERY item of EVERY epoch
    (features_data_loader.py:124-135: ``np.load`` per neighbour) and stacks them on the host; here every file is read once
    (wrap-around padded like features_data_loader.py:37-43), and a batch's (B, K+1, C, N, 1) neighbour stack is one indexed
    gather out of HBM — or, sharded, the input of the neighbour exchange (``shard()``), so the exchange of SURVEY §8e is fed
    straight from the cache.  288 GB of HBM hold ~28000 shapes of 10000 x 256 fp32: every PartNet category fits one GPU."""

    def __init__(self, source, device, first: int = 0, count: Optional[int] = None, n_points: int = N_POINTS):
        n_total = len(source)
        count = n_total - first if count is None else count
        if first < 0 or count < 0 or first + count > n_total:
            raise ValueError(f"owned range [{first}, {first + count}) outside the {n_total} shapes of the dataset")
        self.first, self.n_total, self.device = first, n_total, torch.device(device)
        feats, labels = [], []
        for idx in range(first, first + count):
            f, lab = self._load(source, idx, n_points)
            feats.append(f)
            labels.append(lab)
        self.feats = torch.stack(feats).to(self.device) if feats else torch.empty((0, 0, n_points), device=self.device)
        self.labels = torch.stack(labels).to(self.device) if labels else torch.empty((0, n_points), dtype=torch.int64, device=self.device)

    @staticmethod
    def _load(source, idx: int, n_points: int):
        """One shape as ((C, N) fp32, (N,) int64), from a FeaturesDataset (files), a CSADatasetK (its own shapes) or any
        dataset with the FeaturesDataset item contract."""
        if isinstance(source, CSADatasetK):
            source = source.own
        if isinstance(source, FeaturesDataset):
            f, lab = source.load(source.files[idx])                   # (1, C, N, 1), padded
            return torch.from_numpy(f[0, :, :n_points, 0].astype(np.float32)), torch.from_numpy(lab[:n_points].astype(np.int64))
        item = source[idx]
        f, lab = item[0], item[1]
        f = f.reshape(f.shape[-3], f.shape[-2])                       # (1, C, N, 1) or (C, N, 1) -> (C, N)
        return f[:, :n_points].float().contiguous(), lab[:n_points].long()

    def __len__(self):
        return self.feats.shape[0]

    def _local(self, ids) -> torch.Tensor:
        """Cache-local row numbers of global shape ids, on the device.  The range check runs on the HOST copy of the ids (numpy /
        lists / CPU tensors: every caller in this package) — reading min / max of a device tensor would stall the launch thread
        once per step, under the exchange it is meant to overlap; ids that already live on the device are taken as checked."""
        if isinstance(ids, torch.Tensor) and ids.is_cuda:
            return ids.reshape(-1).to(torch.int64) - self.first
        host = np.asarray(ids.cpu() if isinstance(ids, torch.Tensor) else ids, dtype=np.int64).reshape(-1)
        if host.size and (host.min() < self.first or host.max() >= self.first + len(self)):
            raise IndexError("shape id outside this cache's owned range: fetch it through the exchange of csn_amd.sharding")
        return torch.from_numpy(host - self.first).to(self.device)

    def batch(self, ids):
        """(feats (B, C, N, 1), labels (B, N)) of the given shape ids — what FeaturesDataset / the first two items of
        CSADatasetK hand the model."""
        loc = self._local(ids)
        return self.feats.index_select(0, loc).unsqueeze(-1), self.labels.index_select(0, loc)

    def neighbour_stack(self, ids, nbr_table, neighbour_cache: "Optional[DeviceFeatureCache]" = None) -> torch.Tensor:
        """(B, K+1, C, N, 1): slot 0 = the shape itself, slots 1..K = its neighbours in graph order — bit for bit the tensor
        a DataLoader over CSADatasetK collates (features_data_loader.py:124-140), built by one indexed gather.  ``nbr_table``
        = neighbour_table(knn_graph, K); ``neighbour_cache``: where the neighbours live when they come from another set
        (test shapes take their neighbours from the training set, csa_training.py:288-290)."""
        src = self if neighbour_cache is None else neighbour_cache
        ids_t = torch.as_tensor(ids, dtype=torch.int64).reshape(-1)
        nbr = torch.as_tensor(np.asarray(nbr_table)[ids_t.numpy()], dtype=torch.int64)            # (B, K)
        B, K = nbr.shape
        own = self.feats.index_select(0, self._local(ids_t))                                        # (B, C, N)
        out = torch.empty((B, K + 1) + tuple(own.shape[1:]), device=self.device, dtype=own.dtype)
        out[:, 0] = own
        if K:
            out[:, 1:] = src.feats.index_select(0, src._local(nbr.reshape(-1))).view((B, K) + tuple(own.shape[1:]))
        return out.unsqueeze(-1)

    def batches(self, batch_size: int, nbr_table=None, neighbour_cache: "Optional[DeviceFeatureCache]" = None):
        """Iterate the owned shapes in id order like ``DataLoader(dataset, batch_size, shuffle=False)``: yields
        (feats, label) or, with a neighbour table, (feats (B, C, N, 1), label, neighbour stack) — all on the device."""
        for lo in range(self.first, self.first + len(self), batch_size):
            ids = np.arange(lo, min(lo + batch_size, self.first + len(self)))
            f, lab = self.batch(ids)
            yield (f, lab) if nbr_table is None else (f, lab, self.neighbour_stack(ids, nbr_table, neighbour_cache))

    # -- sharded collections -------------------------------------------------------------------------------------
    @classmethod
    def for_rank(cls, source, device, rank: int, world: int, n_points: int = N_POINTS) -> "DeviceFeatureCache":
        """The cache of rank ``rank``'s share of the collection: shapes [rank * S/world, (rank + 1) * S/world)."""
        S = len(source)
        if S % world:
            raise ValueError(f"{S} shapes do not split evenly over {world} ranks")
        per = S // world
        return cls(source, device, first=rank * per, count=per, n_points=n_points)

    def shard(self, nbr_table, rank: int, world: int):
        """The ShapeGraphShard whose exchange this cache feeds: ``shard.exchange_async(cache.feats)`` (or
        ``exchange_neighbours``) yields the neighbour stack of every owned shape without touching the host."""
        from .sharding import ShapeGraphShard
        return ShapeGraphShard(np.asarray(nbr_table), len(self), rank, world, self.device)
