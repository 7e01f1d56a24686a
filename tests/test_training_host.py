"""Host-side caller logic (csn_amd/training.py, csn_amd/data.py) — CPU tests of the pure functions."""
import os

import numpy as np
import torch

from csn_amd import data as D
from csn_amd import training as T
from oracle import csa_oracle as orc


def test_masked_loss_matches_oracle_restatement():
    rng = np.random.default_rng(1)
    logit = orc.synth_points(rng, (2, 7, 50, 1))
    lab = orc.synth_labels(rng, 2, 50, 7)
    loss, accu = T.loss_functions_seg(logit, lab, 7)
    assert abs(loss.item() - orc.masked_ce_loss(logit, lab).item()) < 1e-7
    flat = logit.squeeze(-1).permute(0, 2, 1).reshape(-1, 7)
    keep = lab.reshape(-1) > 0
    assert abs(accu.item() - (flat[keep].argmax(1) == lab.reshape(-1)[keep]).float().mean().item()) < 1e-7


def test_iou_counts_match_loop_restatement():
    """csa_training.py:110-134 as python loops over classes."""
    rng = np.random.default_rng(2)
    n_cls = 5
    logit = orc.synth_points(rng, (3, n_cls, 40, 1))
    lab = orc.synth_labels(rng, 3, 40, n_cls)
    intsc, union = T.IoU_per_shape(logit, lab, n_cls)
    pred = logit.squeeze(-1).permute(0, 2, 1).reshape(-1, n_cls).argmax(1)
    l = lab.reshape(-1)
    keep = l > 0
    pred, l = pred[keep], l[keep]
    for k in range(n_cls):
        assert intsc[k].item() == ((pred == k) & (l == k)).sum().item()
        assert union[k].item() == ((pred == k) | (l == k)).sum().item()
    iou = sum(intsc[k].item() / (union[k].item() + 1e-10) for k in range(n_cls)) / (n_cls - 1)
    assert abs(T.mean_iou(intsc, union) - iou) < 1e-12


def test_wraparound_padding_and_dataset_contract(tmp_path):
    root = tmp_path / "Bag_train_feats"
    os.makedirs(root / "fc_1")
    os.makedirs(root / "point_labels")
    rng = np.random.default_rng(3)
    sizes = [10000, 7000, 9999]
    for i, n in enumerate(sizes):
        np.save(root / "fc_1" / f"s{i}.npy", rng.standard_normal(size=(1, 256, n, 1)).astype(np.float32))
        np.save(root / "point_labels" / f"s{i}.npy", rng.integers(0, 4, size=n))
    ds = D.FeaturesDataset(str(root))
    for idx in range(3):
        feats, label = ds[idx]
        assert feats.shape == (1, 256, 10000, 1) and label.shape == (10000,)
        raw = np.load(root / "fc_1" / ds.files[idx])
        n = raw.shape[2]
        assert np.array_equal(feats.numpy()[:, :, :n], raw)
        assert np.array_equal(feats.numpy()[:, :, n:], raw[:, :, :10000 - n])      # wrap-around (features_data_loader.py:37-43)
    graph = np.array([[0, 1, 2], [1, 2, 0], [2, 0, 1]])
    csa = D.CSADatasetK(str(root), str(root), graph, K=2)
    f, l, nb = csa[1]
    assert f.shape == (256, 10000, 1) and nb.shape == (3, 256, 10000, 1)
    assert torch.equal(nb[0], f)                                                     # slot 0 = the shape itself
    assert torch.equal(nb[1], ds[2][0][0]) and torch.equal(nb[2], ds[0][0][0])       # graph order, self skipped


def test_synthetic_dataset_item_contract():
    ds = D.SyntheticShapes(4, 5, n_points=64, channels=256)
    f, l = ds[0]
    assert f.shape == (1, 256, 64, 1) and l.shape == (64,) and l.dtype == torch.int64
    ds = D.SyntheticShapes(4, 5, K=2, knn_graph=[[0, 1, 2], [1, 0, 3], [2, 3, 0], [3, 2, 1]], n_points=64)
    f, l, nb = ds[1]
    assert nb.shape == (3, 256, 64, 1) and torch.equal(nb[0], f)


def test_optimizer_and_warm_start_follow_the_reference():
    from csn_amd.csa_models import get_model
    ssa, csa = get_model("ssa", 6, 1), get_model("csa", 6, 1, 2)
    opt, sched = T.make_optimizer(csa, lr=1e-3, weight_decay=5e-4)
    g = opt.param_groups[0]
    assert g["betas"] == (0.5, 0.999) and g["weight_decay"] == 5e-4 and g["lr"] == 1e-3      # csa_training.py:307
    sched.step()
    assert abs(opt.param_groups[0]["lr"] - 1e-4) < 1e-12                                     # StepLR gamma 0.1 (:308)
    T.load_trained_ssa_layers(csa, ssa.state_dict())                                         # utils.py:29-39
    for k, v in ssa.state_dict().items():
        assert torch.equal(csa.state_dict()[k], v)
