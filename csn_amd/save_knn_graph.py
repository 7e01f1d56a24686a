#!/usr/bin/env python3
"""The ``save_knn_graph.py`` that MID-FC/run_save_knn.py:50 launches but the reference repository does not contain.

Builds the train/test kNN shape graphs with a trained SSA model and writes ``<graphs_dir>/train.npy`` and
``test.npy`` (int64 ``(S, K+1)``, candidate ids in descending retrieval score) — the files csa_training.py:286-290
reads.

Command line = what the launcher builds (run_save_knn.py:60-72), nothing more is required::

    python save_knn_graph.py --ssa_logs_dir=<logs>/<Part> --graphs_dir=<logs>/knn_graphs/<Part> --partname=<Part>
                             --n_heads=H --num_workers=W --batch_size=B --num_classes=C [--testing]

* ``--ssa_logs_dir`` is the DIRECTORY holding ``trained_layers.pth`` (utils.py:29-31 joins the file name).
* The feature files are found under a data root the launcher does not pass — the reference's scripts hard-code it
  (csa_training.py:269-272: ``<root>/{train,test}_data_features/<Part>``).  Here: ``--dataroot`` or the environment
  variable ``CSN_DATAROOT``; under it the reference's pattern ``<root>/<split>_data_features/<Part>`` is looked for first,
  then this repository's earlier ``<root>/<Part>_<split>_feats``.
* ``--K`` (not passed by the launcher) defaults to 10, the launcher's own default (run_save_knn.py:35): the table has K+1
  columns and csa_training.py:286-295 hands it to ``CSADatasetK(…, K)`` with any training K ≤ that, which reads the first
  K+1 usable columns of a row.
* ``--testing`` (run_save_knn.py:69-70) is the reference's smoke switch (csa_training.py:27,218-219: stop after the first
  batch): only the first ``TESTING_SHAPES`` shapes of each split are scored, the tables have that many rows.
"""
import argparse
import os

import numpy as np
import torch
from torch.utils.data import DataLoader, Subset

TESTING_SHAPES = 16
DATAROOT_ENV = "CSN_DATAROOT"


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--ssa_logs_dir", type=str, required=True, help="directory holding trained_layers.pth of the SSA run")
    ap.add_argument("--graphs_dir", type=str, required=True)
    ap.add_argument("--partname", type=str, required=True)
    ap.add_argument("--dataroot", type=str, default=None, help=f"feature root (default: ${DATAROOT_ENV})")
    ap.add_argument("--n_heads", type=int, default=1)
    ap.add_argument("--num_workers", type=int, default=0)
    ap.add_argument("--batch_size", type=int, default=1)
    ap.add_argument("--num_classes", type=int, required=True)
    ap.add_argument("--K", type=int, default=10)
    ap.add_argument("--testing", action="store_true")
    return ap


def parse_args(argv=None) -> argparse.Namespace:
    args = build_parser().parse_args(argv)
    if args.dataroot is None:
        args.dataroot = os.environ.get(DATAROOT_ENV)
    return args


def split_root(dataroot: str, split: str, partname: str) -> str:
    """Directory with ``fc_1/`` and ``point_labels/`` of one split: the reference's layout (csa_training.py:269-272) first."""
    cands = [os.path.join(dataroot, f"{split}_data_features", partname), os.path.join(dataroot, f"{partname}_{split}_feats")]
    for c in cands:
        if os.path.isdir(os.path.join(c, "fc_1")):
            return c
    raise FileNotFoundError(f"no feature directory for split '{split}' of '{partname}': looked for " + " and ".join(cands))


def checkpoint_path(ssa_logs_dir: str) -> str:
    """utils.py:29-31: the logs directory holds trained_layers.pth (a path to the file itself is taken too)."""
    return ssa_logs_dir if os.path.isfile(ssa_logs_dir) else os.path.join(ssa_logs_dir, "trained_layers.pth")


def main(argv=None):
    args = parse_args(argv)
    if not args.dataroot:
        raise SystemExit(f"save_knn_graph: no data root — pass --dataroot or set {DATAROOT_ENV} (the reference hard-codes "
                         "its cluster path, csa_training.py:269)")

    from .csa_models import get_model
    from .data import FeaturesDataset
    from .training import BIG_CLASSES, load_trained_ssa_layers, update_knn_graphs

    device = torch.device("cuda")
    model = get_model("ssa", args.num_classes, args.n_heads).to(device)
    load_trained_ssa_layers(model, checkpoint_path(args.ssa_logs_dir))
    sets = {s: FeaturesDataset(split_root(args.dataroot, s, args.partname)) for s in ("train", "test")}
    if args.testing:
        sets = {s: Subset(d, range(min(len(d), TESTING_SHAPES))) for s, d in sets.items()}
    train = DataLoader(sets["train"], args.batch_size, shuffle=False, num_workers=args.num_workers)
    test = DataLoader(sets["test"], args.batch_size, shuffle=False, num_workers=args.num_workers)
    K = min(args.K, len(sets["train"]) - 1)                 # topk(K+1) over the candidates (csa_models.py:278)
    big = args.partname in BIG_CLASSES and not args.testing  # k-means centres need ≥ 10 shapes per cluster (csa_models.py:321)
    train_g, test_g = update_knn_graphs(model, train, test, K, device, big_category=big)
    os.makedirs(args.graphs_dir, exist_ok=True)
    np.save(os.path.join(args.graphs_dir, "train.npy"), train_g)
    np.save(os.path.join(args.graphs_dir, "test.npy"), test_g)
    print(f"saved {train_g.shape} / {test_g.shape} kNN graphs to {args.graphs_dir}")


if __name__ == "__main__":
    main()
