#!/usr/bin/env python3
"""The ``save_knn_graph.py`` that MID-FC/run_save_knn.py:50 launches but the reference repository does not contain.

Builds the train/test kNN shape graphs with a trained SSA model and writes ``<graphs_dir>/train.npy`` and
``test.npy`` (int64 ``(S, K+1)``, candidate ids in descending retrieval score) — the files csa_training.py:286-290
reads.  Flags follow run_save_knn.py:52-60; ``--dataroot`` replaces the reference's hard-coded cluster path
(csa_training.py:269-275: <root>/<Part>_train_feats, <Part>_test_feats).
"""
import argparse
import os

import numpy as np
import torch
from torch.utils.data import DataLoader


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--ssa_logs_dir", type=str, required=True, help="directory holding trained_layers.pth of the SSA run")
    ap.add_argument("--graphs_dir", type=str, required=True)
    ap.add_argument("--partname", type=str, required=True)
    ap.add_argument("--dataroot", type=str, required=True)
    ap.add_argument("--n_heads", type=int, default=1)
    ap.add_argument("--num_workers", type=int, default=0)
    ap.add_argument("--batch_size", type=int, default=1)
    ap.add_argument("--num_classes", type=int, required=True)
    ap.add_argument("--K", type=int, default=10)
    args = ap.parse_args(argv)

    from .csa_models import get_model
    from .data import FeaturesDataset
    from .training import BIG_CLASSES, load_trained_ssa_layers, update_knn_graphs

    device = torch.device("cuda")
    model = get_model("ssa", args.num_classes, args.n_heads).to(device)
    load_trained_ssa_layers(model, os.path.join(args.ssa_logs_dir, "trained_layers.pth"))
    train = DataLoader(FeaturesDataset(os.path.join(args.dataroot, f"{args.partname}_train_feats")), args.batch_size,
                       shuffle=False, num_workers=args.num_workers)
    test = DataLoader(FeaturesDataset(os.path.join(args.dataroot, f"{args.partname}_test_feats")), args.batch_size,
                      shuffle=False, num_workers=args.num_workers)
    train_g, test_g = update_knn_graphs(model, train, test, args.K, device, big_category=args.partname in BIG_CLASSES)
    os.makedirs(args.graphs_dir, exist_ok=True)
    np.save(os.path.join(args.graphs_dir, "train.npy"), train_g)
    np.save(os.path.join(args.graphs_dir, "test.npy"), test_g)
    print(f"saved {train_g.shape} / {test_g.shape} kNN graphs to {args.graphs_dir}")


if __name__ == "__main__":
    main()
