"""Host-side caller logic (csn_amd/training.py, csn_amd/data.py) — CPU tests of the pure functions."""
import os

import pytest

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


class _TinySeg(torch.nn.Module):
    """model(feats, mode, neighbours) -> logits (B, n_cls, N, 1): a 1x1 convolution, enough to watch the optimizer."""

    def __init__(self, C, n_cls):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(n_cls, C))

    def forward(self, feats, mode=None, neighbor_feats=None):
        return torch.einsum("kc,bcn->bkn", self.w, feats.squeeze(-1)).unsqueeze(-1)


def _micro_batches(rng, n, C=8, N=32, n_cls=4):
    out = []
    for _ in range(n):
        f = torch.from_numpy(rng.standard_normal((2, C, N, 1)).astype(np.float32))
        lab = torch.from_numpy(rng.integers(1, n_cls, size=(2, N)).astype(np.int64))
        out.append((f, lab, torch.zeros(2, 1, C, N, 1)))
    return out


def test_train_layers_reference_order_versus_repaired_accumulation():
    """csa_training.py:196 zeroes the gradients at the top of EVERY iteration, so with gradient_accumulation_steps = 2 only
    the second micro-batch's half-scaled gradient reaches optimizer.step() (:213-215).  reference_semantics=True (default)
    reproduces exactly that; False accumulates both."""
    rng = np.random.default_rng(3)
    batches = _micro_batches(rng, 2)
    grads = []
    for f, lab, _ in batches:
        m = _TinySeg(8, 4)
        loss, _ = T.loss_functions_seg(m(f), lab, 4)
        (loss / 2).backward()
        grads.append(m.w.grad.clone())
    for ref_mode, want in ((True, grads[1]), (False, grads[0] + grads[1])):
        m = _TinySeg(8, 4)
        opt = torch.optim.SGD(m.parameters(), lr=1.0)
        T.train_layers(m, batches, opt, 4, torch.device("cpu"), accumulation_steps=2, reference_semantics=ref_mode)
        assert torch.allclose(m.w.detach(), -want, atol=1e-7), ref_mode


def test_train_layers_nan_loss_guard_as_written_and_repaired():
    """:206-207 multiplies a NaN loss by 0 — still NaN — so the reference's step is poisoned; the repaired loop skips it."""
    rng = np.random.default_rng(4)
    good, bad = _micro_batches(rng, 2)
    bad = (bad[0] * float("nan"), bad[1], bad[2])
    for ref_mode in (True, False):
        m = _TinySeg(8, 4)
        opt = torch.optim.SGD(m.parameters(), lr=1.0)
        mean_loss = T.train_layers(m, [good, bad], opt, 4, torch.device("cpu"), accumulation_steps=1, reference_semantics=ref_mode)
        assert np.isfinite(mean_loss) and mean_loss > 0              # the running loss skips the NaN batch in both
        assert torch.isnan(m.w).any().item() == ref_mode


def test_validate_layers_skips_nan_batches_entirely():
    """csa_training.py:239-240: a NaN batch is `continue`d — neither its loss nor its IoU counts are accumulated."""
    rng = np.random.default_rng(5)
    good, bad = _micro_batches(rng, 2)
    bad = (bad[0] * float("nan"), bad[1], bad[2])
    m = _TinySeg(8, 4)
    with torch.no_grad():
        m.w.copy_(torch.from_numpy(rng.standard_normal((4, 8)).astype(np.float32)))
    iou_both, loss_both = T.validate_layers(m, [good, bad], 4, torch.device("cpu"))
    iou_good, loss_good = T.validate_layers(m, [good], 4, torch.device("cpu"))
    assert abs(iou_both - iou_good) < 1e-12 and abs(loss_both * 2 - loss_good) < 1e-9


def _write_feature_files(root, n_shapes, rng, sizes):
    os.makedirs(os.path.join(root, "fc_1"))
    os.makedirs(os.path.join(root, "point_labels"))
    for i in range(n_shapes):
        n = sizes[i % len(sizes)]
        np.save(os.path.join(root, "fc_1", f"s{i:02d}.npy"), rng.standard_normal((1, 256, n, 1)).astype(np.float32))
        np.save(os.path.join(root, "point_labels", f"s{i:02d}.npy"), rng.integers(0, 5, size=(n,)))


def test_device_feature_cache_builds_the_same_neighbour_stacks_as_csadatasetk(tmp_path, monkeypatch):
    """the cache (every file read once, stacks by indexed gather) against a DataLoader over the restated
    CSADatasetK (K np.load's per item, features_data_loader.py:124-140): features, labels and neighbour stacks bit for bit,
    including wrap-around padded shapes, self-skipping graph rows, and test shapes whose neighbours live in the train set."""
    from torch.utils.data import DataLoader
    monkeypatch.setattr(os, "listdir", lambda p, _ls=os.listdir: sorted(_ls(p)))       # a fixed file order for both readers
    rng = np.random.default_rng(12)
    tr, te = str(tmp_path / "train"), str(tmp_path / "test")
    _write_feature_files(tr, 6, rng, sizes=(64, 40, 64, 51))
    _write_feature_files(te, 3, rng, sizes=(64, 33))
    K = 2
    g_train = np.array([[0, 3, 5], [1, 0, 2], [4, 2, 1], [3, 1, 0], [4, 5, 0], [5, 2, 3]])     # self first, third or absent
    g_test = np.array([[2, 4, 0], [5, 1, 3], [0, 1, 2]])                                        # ids into the TRAIN set
    ds_train = D.CSADatasetK(tr, tr, g_train, K, n_points=64)
    ds_test = D.CSADatasetK(te, tr, g_test, K, n_points=64)
    cache_tr = D.DeviceFeatureCache(ds_train, "cpu", n_points=64)
    cache_te = D.DeviceFeatureCache(ds_test, "cpu", n_points=64)
    assert len(cache_tr) == 6 and cache_tr.feats.shape == (6, 256, 64) and cache_tr.labels.dtype == torch.int64
    tab_tr = D.neighbour_table(g_train, K)
    assert tab_tr.tolist() == [[3, 5], [0, 2], [4, 1], [1, 0], [5, 0], [2, 3]]                 # self skipped, graph order kept
    for (f, lab, nb), (cf, cl, cnb) in zip(DataLoader(ds_train, 4, shuffle=False), cache_tr.batches(4, tab_tr)):
        assert torch.equal(f, cf) and torch.equal(lab, cl) and torch.equal(nb, cnb)
    # test shapes: the reference compares graph entries with the TEST index (features_data_loader.py:127), so a train id that
    # happens to equal the test shape's own index is skipped — neighbour_table reproduces that rule
    tab_te = D.neighbour_table(g_test, K)
    assert tab_te.tolist() == [[2, 4], [5, 3], [0, 1]]
    for (f, lab, nb), (cf, cl, cnb) in zip(DataLoader(ds_test, 2, shuffle=False), cache_te.batches(2, tab_te, cache_tr)):
        assert torch.equal(f, cf) and torch.equal(lab, cl) and torch.equal(nb, cnb)
    with pytest.raises(IndexError):
        D.DeviceFeatureCache(ds_train, "cpu", first=2, count=2, n_points=64).batch([0])


def test_g9_data_path_against_the_reference_loader(tmp_path, monkeypatch):
    """Golden set G9: the REFERENCE's CSADatasetK / FeaturesDataset (features_data_loader.py:9-48, 79-140) were run over
    seeded shape files (10000, 7000 and 5100 points: two kinds get wrap-around padded) with a train graph and a test graph
    whose ids point into the train set; tests/golden/make_golden.py stored per item the shapes, dtypes, a sha256 of the raw
    bytes and strided samples.  The same files are rebuilt here and csn_amd.data.CSADatasetK, FeaturesDataset and
    DeviceFeatureCache must reproduce every item bit for bit."""
    from tests.golden import g9_spec as g9
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "g9_data_path.npz"))
    monkeypatch.setattr(os, "listdir", lambda p, _ls=os.listdir: sorted(_ls(p)))       # the order the goldens were made with
    tr, te = g9.write_both(str(tmp_path))
    cache_tr = None
    for tag, root, graph in (("train", tr, g9.TRAIN_GRAPH), ("test", te, g9.TEST_GRAPH)):
        ds = D.CSADatasetK(root, tr, np.array(graph), g9.K)
        plain = D.FeaturesDataset(root)
        assert [len(ds), len(plain)] == z[f"g9_{tag}_len"].tolist()
        cache = D.DeviceFeatureCache(ds, "cpu")
        cache_tr = cache if tag == "train" else cache_tr
        table = D.neighbour_table(np.array(graph), g9.K)
        for i in range(len(ds)):
            f, lab, nb = ds[i]
            pf, pl = plain[i]
            assert list(f.shape) + list(lab.shape) + list(nb.shape) + list(pf.shape) == z[f"g9_{tag}_{i}_shapes"].tolist()
            assert [str(t.dtype) for t in (f, lab, nb, pf, pl)] == z[f"g9_{tag}_{i}_dtypes"].tolist()
            assert np.array_equal(f[::16, ::997, 0].numpy(), z[f"g9_{tag}_{i}_feats"])
            assert np.array_equal(lab[::97].numpy(), z[f"g9_{tag}_{i}_label"])
            assert np.array_equal(nb[:, ::16, ::997, 0].numpy(), z[f"g9_{tag}_{i}_nb"])
            assert [g9.digest(t) for t in (f, lab, nb, pf, pl)] == z[f"g9_{tag}_{i}_sha"].tolist()
            # the device-resident cache hands out the same three tensors (as a batch of one) without touching the files again
            cf, cl = cache.batch([i])
            cnb = cache.neighbour_stack([i], table, cache_tr)
            assert [g9.digest(t) for t in (cf[0], cl[0], cnb[0])] == z[f"g9_{tag}_{i}_sha"].tolist()[:3]


def test_head_width_generalisation_host_side():
    """d_k != d_v / odd widths: parameters keep the reference's shapes; the kernels see zero-padded weights at one width."""
    import torch
    from csn_amd import functional as CF
    from csn_amd.csa_models import MultiHeadAttention
    from csn_amd.minkowski_attention import MultiHeadAttention as MinkMHA
    assert [CF.kernel_head_width(d) for d in (1, 32, 33, 96, 100, 129, 256)] == [32, 32, 64, 96, 128, 256, 256]
    with pytest.raises(ValueError):
        CF.kernel_head_width(257)
    for cls in (MultiHeadAttention, MinkMHA):
        m = cls(3, 96, 48, 80)
        assert m.w_qs.weight.shape == (144, 96) and m.w_vs.weight.shape == (240, 96) and m.fc.weight.shape == (96, 240)
        assert m.d_head == 96
        wq, wk, wv, wfc = m.kernel_weights()
        assert wq.shape == wk.shape == wv.shape == (288, 96) and wfc.shape == (96, 288)
        assert torch.equal(wq.view(3, 96, 96)[:, :48], m.w_qs.weight.view(3, 48, 96)) and float(wq.detach().view(3, 96, 96)[:, 48:].abs().max()) == 0.0
        assert torch.equal(wfc.view(96, 3, 96)[:, :, :80], m.fc.weight.view(96, 3, 80)) and float(wfc.detach().view(96, 3, 96)[:, :, 80:].abs().max()) == 0.0
        wfc.sum().backward()
        assert m.fc.weight.grad.shape == (96, 240)
        same = cls(2, 128, 64, 64)
        assert same.kernel_weights()[0] is same.w_qs.weight            # equal, instanced widths: the parameters themselves
    geo = MultiHeadAttention(3, 96, 48, 80, block=100, n_blocks=4).geometry()
    assert geo.d_head == 96 and abs(geo.temperature - 48 ** 0.5) < 1e-12
    assert MultiHeadAttention(1, 256, 256, 256).geometry().temperature == 0.0


# ---------------------------------------------------------------------------------------------------------
# save_knn_graph CLI: the argv of its only caller (MID-FC/run_save_knn.py:60-72)
# ---------------------------------------------------------------------------------------------------------
def launcher_argv(ssa_logs_dir, name, n_heads=1, num_workers=3, batch_size=1, num_classes=9, testing=False):
    """The command words run_save_knn.py:60-72 builds for one category (its defaults: :31-38), after the script name."""
    words = [f"--ssa_logs_dir={ssa_logs_dir}/{name}", f"--graphs_dir={ssa_logs_dir}/knn_graphs/{name}", f"--partname={name}",
             f"--n_heads={n_heads}", f"--num_workers={num_workers}", f"--batch_size={batch_size}", f"--num_classes={num_classes}"]
    return words + (["--testing"] if testing else [])


@pytest.mark.parametrize("testing", [False, True])
def test_save_knn_graph_cli_takes_the_launchers_argv(testing, monkeypatch, tmp_path):
    from csn_amd import save_knn_graph as S
    monkeypatch.delenv(S.DATAROOT_ENV, raising=False)
    args = S.parse_args(launcher_argv("logs/ssa_n_heads_1", "Bottle", testing=testing))
    assert args.ssa_logs_dir == "logs/ssa_n_heads_1/Bottle" and args.graphs_dir == "logs/ssa_n_heads_1/knn_graphs/Bottle"
    assert (args.partname, args.n_heads, args.num_workers, args.batch_size, args.num_classes) == ("Bottle", 1, 3, 1, 9)
    assert args.testing is testing and args.K == 10 and args.dataroot is None
    with pytest.raises(SystemExit, match="CSN_DATAROOT"):               # no root anywhere: a clear message, not a traceback
        S.main(launcher_argv("logs/ssa_n_heads_1", "Bottle", testing=testing))
    monkeypatch.setenv(S.DATAROOT_ENV, str(tmp_path))
    assert S.parse_args(launcher_argv("l", "Bed")).dataroot == str(tmp_path)
    assert S.parse_args(launcher_argv("l", "Bed") + ["--dataroot=/x"]).dataroot == "/x"


def test_save_knn_graph_cli_finds_the_reference_data_layout(tmp_path):
    """csa_training.py:269-272: <root>/{train,test}_data_features/<Part>; the earlier <Part>_<split>_feats as a fallback."""
    from csn_amd import save_knn_graph as S
    ref = tmp_path / "train_data_features" / "Bed" / "fc_1"
    ref.mkdir(parents=True)
    old = tmp_path / "Bed_test_feats" / "fc_1"
    old.mkdir(parents=True)
    assert S.split_root(str(tmp_path), "train", "Bed") == str(ref.parent)
    assert S.split_root(str(tmp_path), "test", "Bed") == str(old.parent)
    with pytest.raises(FileNotFoundError, match="Vase"):
        S.split_root(str(tmp_path), "train", "Vase")
    assert S.checkpoint_path(str(tmp_path)) == str(tmp_path / "trained_layers.pth")      # utils.py:29-31
    (tmp_path / "w.pth").write_bytes(b"")
    assert S.checkpoint_path(str(tmp_path / "w.pth")) == str(tmp_path / "w.pth")
