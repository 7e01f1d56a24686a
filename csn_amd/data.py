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

    def __init__(self, dataroot: str, attention_type: str = "backbone_fc_ssa_logit"):
        self.features_dir = os.path.join(dataroot, "fc_1")
        self.labels_dir = os.path.join(dataroot, "point_labels")
        self.files = os.listdir(self.features_dir)

    def __len__(self):
        return len(self.files)

    def load(self, name: str):
        feats = np.load(os.path.join(self.features_dir, name))
        label = np.load(os.path.join(self.labels_dir, name)).astype(int)
        return pad_points(feats, label)

    def __getitem__(self, idx):
        feats, label = self.load(self.files[idx])
        return torch.from_numpy(feats), torch.from_numpy(label)


class CSADatasetK(Dataset):
    """(feats (256, N, 1), label (N,), neighbor_feats (K+1, 256, N, 1)); neighbours come from ``dataroot_K`` in kNN-graph
    order, skipping the shape itself, slot 0 = the shape (features_data_loader.py:79-140)."""

    def __init__(self, dataroot: str, dataroot_K: str, knn_graph, K: int):
        self.own = FeaturesDataset(dataroot)
        self.nbr = FeaturesDataset(dataroot_K)
        self.K = K
        self.knn_graph = np.copy(knn_graph)

    def __len__(self):
        return len(self.own)

    def __getitem__(self, idx):
        feats, label = self.own.load(self.own.files[idx])
        stack = [feats]
        for kidx in self.knn_graph[idx]:
            if kidx != idx:
                stack.append(pad_points(np.load(os.path.join(self.nbr.features_dir, self.nbr.files[kidx])))[0])
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
