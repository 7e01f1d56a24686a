"""Inputs of golden set G9 (the feature-file data path, MID-FC/features_data_loader.py:79-140): seeded synthetic shape files
and the two kNN graphs.  Shared by make_golden.py (which runs the REFERENCE's datasets over these files, in the build
container) and tests/test_training_host.py (which rebuilds the same files anywhere and checks csn_amd.data against the
stored outputs).  Nothing here comes from the reference: only file sizes, a seed and two small integer tables."""
import hashlib
import os

import numpy as np

SEED = 909
TRAIN_SIZES = (10000, 7000, 5100, 10000, 7000)          # points per shape file: full, and two that get wrap-around padded
TEST_SIZES = (5100, 10000, 7000)
K = 2
TRAIN_GRAPH = [[0, 3, 4], [1, 0, 2], [4, 2, 1], [3, 1, 0], [2, 4, 0]]      # the shape itself first, third or absent in its row
TEST_GRAPH = [[2, 4, 0], [3, 1, 0], [0, 1, 2]]                            # ids into the TRAIN set (csa_training.py:288-290)


def write_files(root, sizes, rng):
    """<root>/fc_1/sNN.npy float32 (1, 256, n, 1) and <root>/point_labels/sNN.npy int64 (n,), drawn from rng in file order —
    the on-disk contract of features_data_loader.py:18-32."""
    os.makedirs(os.path.join(root, "fc_1"))
    os.makedirs(os.path.join(root, "point_labels"))
    for i, n in enumerate(sizes):
        np.save(os.path.join(root, "fc_1", f"s{i:02d}.npy"), rng.standard_normal((1, 256, n, 1)).astype(np.float32))
        np.save(os.path.join(root, "point_labels", f"s{i:02d}.npy"), rng.integers(0, 5, size=(n,)))


def write_both(tmp):
    """The train and test trees under tmp, from one generator stream; returns their roots."""
    rng = np.random.default_rng(SEED)
    tr, te = os.path.join(tmp, "train"), os.path.join(tmp, "test")
    write_files(tr, TRAIN_SIZES, rng)
    write_files(te, TEST_SIZES, rng)
    return tr, te


def digest(t) -> str:
    """sha256 over dtype, shape and raw bytes of a tensor / array: equal digests = bit-for-bit equal items."""
    a = np.ascontiguousarray(t.numpy() if hasattr(t, "numpy") else t)
    return hashlib.sha256(str(a.dtype).encode() + str(a.shape).encode() + a.tobytes()).hexdigest()
