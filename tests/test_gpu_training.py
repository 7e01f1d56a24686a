"""The caller loop of MID-FC/csa_training.py on the MI355X through csn_amd.training: SSA pre-training, kNN-graph
construction with the trained model (bit-exact indices vs the CPU oracle), CSA training on that graph."""
import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

from csn_amd import data as D
from csn_amd import training as T
from oracle import csa_oracle as orc

pytestmark = pytest.mark.gpu
N_PTS, BLOCK, NB, N_CLS = 400, 100, 4, 5


def _small(model):
    model.attention.block, model.attention.n_blocks = BLOCK, NB        # 4 blocks of 100 instead of 20 of 500
    return model.cuda()


def test_ssa_then_graph_then_csa_training():
    from csn_amd.csa_models import get_model
    dev = torch.device("cuda")
    torch.manual_seed(0)
    train_set = D.SyntheticShapes(8, N_CLS, seed=1, n_points=N_PTS)
    test_set = D.SyntheticShapes(3, N_CLS, seed=2, n_points=N_PTS)
    train_ld, test_ld = DataLoader(train_set, 4, shuffle=False), DataLoader(test_set, 1, shuffle=False)

    # ---- SSA pre-training (ssa_training.py:125-156): the loss goes down -----------------------------------
    ssa = _small(get_model("ssa", N_CLS, 1))
    opt, _ = T.make_optimizer(ssa, lr=2e-3, weight_decay=0.0)
    losses = [T.train_layers(ssa, train_ld, opt, N_CLS, dev, accumulation_steps=1) for _ in range(12)]
    assert losses[-1] < 0.8 * losses[0], losses
    iou, vloss = T.validate_layers(ssa, test_ld, N_CLS, dev)
    assert 0.0 <= iou <= 1.0 and np.isfinite(vloss)

    # ---- kNN graphs with the trained model (csa_training.py:136-163), indices bit-exact vs the oracle -----
    K = 2
    train_g, test_g = T.update_knn_graphs(ssa, train_ld, test_ld, K, dev)
    assert train_g.shape == (8, K + 1) and test_g.shape == (3, K + 1) and train_g.dtype == np.int64
    assert (train_g[:, 0] == np.arange(8)).all()                      # every shape retrieves itself first
    p = {k: v.detach().cpu() for k, v in ssa.state_dict().items()}
    feats = lambda ds: torch.cat([orc.mha_blockdiag(torch.from_numpy(ds.feats[i:i + 1]), torch.from_numpy(ds.feats[i:i + 1]),
                                                    torch.from_numpy(ds.feats[i:i + 1]), p, 1, block=BLOCK, n_blocks=NB)
                                  for i in range(len(ds))])
    with torch.no_grad():
        f_tr, f_te = feats(train_set), feats(test_set)
        assert np.array_equal(train_g, orc.knn_graph(f_tr, f_tr, K).numpy())
        assert np.array_equal(test_g, orc.knn_graph(f_te, f_tr, K).numpy())

    # ---- CSA on that graph, warm-started from the SSA weights (utils.py:29-39, csa_training.py:191-222) ---
    csa = _small(get_model("csa", N_CLS, 1, K))
    T.load_trained_ssa_layers(csa, ssa.state_dict())
    csa_train_set = D.SyntheticShapes(8, N_CLS, K=K, knn_graph=train_g, seed=1, n_points=N_PTS)
    csa_train = DataLoader(csa_train_set, 4, shuffle=False)
    csa_test = DataLoader(D.SyntheticShapes(3, N_CLS, K=K, knn_graph=test_g, seed=2, n_points=N_PTS,
                                            neighbor_source=csa_train_set), 1, shuffle=False)
    opt, sched = T.make_optimizer(csa, lr=1e-3, weight_decay=5e-4)
    l0 = T.train_layers(csa, csa_train, opt, N_CLS, dev, accumulation_steps=2)
    for _ in range(5):
        l1 = T.train_layers(csa, csa_train, opt, N_CLS, dev, accumulation_steps=2)
    assert l1 < l0
    iou2, _ = T.validate_layers(csa, csa_test, N_CLS, dev)
    assert 0.0 <= iou2 <= 1.0


def test_big_category_graph_maps_candidate_ids_back():
    from csn_amd.csa_models import get_model
    dev = torch.device("cuda")
    torch.manual_seed(0)
    train_set = D.SyntheticShapes(20, N_CLS, seed=5, n_points=N_PTS)
    test_set = D.SyntheticShapes(2, N_CLS, seed=6, n_points=N_PTS)
    model = _small(get_model("ssa", N_CLS, 1)).eval()
    train_ld, test_ld = DataLoader(train_set, 1, shuffle=False), DataLoader(test_set, 1, shuffle=False)
    tr, te = T.update_knn_graphs(model, train_ld, test_ld, 1, dev, big_category=True)
    centres = set(np.asarray(model.get_center_shape_indices(train_ld)).tolist())
    assert len(centres) <= 2                                          # 20 // 10 k-means centres (csa_models.py:321)
    assert tr.shape == (20, 2) and te.shape == (2, 2)
    assert set(tr.reshape(-1).tolist()) <= centres and set(te.reshape(-1).tolist()) <= centres


@pytest.mark.parametrize("testing", [False, True])
def test_save_knn_graph_cli_writes_the_files_csa_training_reads(tmp_path, monkeypatch, testing):
    """``python -m csn_amd.save_knn_graph`` — the script MID-FC/run_save_knn.py:50 launches and the reference does not ship — run
    through its main() with EXACTLY the argv the launcher builds (run_save_knn.py:60-72, with and without --testing; the data
    root comes from CSN_DATAROOT in the reference's <root>/<split>_data_features/<Part> layout, csa_training.py:269-272):
    feature files on disk in the O-CNN layout (fc_1/*.npy (1, 256, n, 1), point_labels/*.npy), a saved SSA checkpoint, and out
    come train.npy / test.npy as int64 (S, K+1) tables (csa_training.py:286-290 reads them) that equal what update_knn_graphs
    builds from the same loaders."""
    from csn_amd import save_knn_graph
    from csn_amd.csa_models import get_model
    from tests.test_training_host import launcher_argv
    rng = np.random.default_rng(9)
    n_cls, part = 4, "Bottle"
    root = tmp_path / "data"
    sizes = {"train": [10000, 7000, 10000, 5100, 10000], "test": [10000, 6000]}          # short shapes are wrap-around padded
    for split, ns in sizes.items():
        d = root / f"{split}_data_features" / part
        (d / "fc_1").mkdir(parents=True)
        (d / "point_labels").mkdir()
        for i, n in enumerate(ns):
            np.save(d / "fc_1" / f"shape_{i:02d}.npy", (rng.standard_normal((1, 256, n, 1)) + 0.3 * i).astype(np.float32))
            np.save(d / "point_labels" / f"shape_{i:02d}.npy", rng.integers(0, n_cls, size=n))
    torch.manual_seed(1)
    ssa = get_model("ssa", n_cls, 1).cuda().eval()
    logs = tmp_path / "logs" / "ssa_n_heads_1"
    (logs / part).mkdir(parents=True)
    torch.save(ssa.state_dict(), logs / part / "trained_layers.pth")                     # csa_training.py:324-326
    monkeypatch.setenv(save_knn_graph.DATAROOT_ENV, str(root))
    monkeypatch.setattr(save_knn_graph, "TESTING_SHAPES", 3)
    save_knn_graph.main(launcher_argv(str(logs), part, num_workers=0, num_classes=n_cls, testing=testing))
    graphs = logs / "knn_graphs" / part
    tr, te = np.load(graphs / "train.npy"), np.load(graphs / "test.npy")
    n_tr = 3 if testing else 5
    K = n_tr - 1                                                                          # the launcher's K = 10, cut to the candidates
    assert tr.dtype == np.int64 and te.dtype == np.int64 and tr.shape == (n_tr, K + 1) and te.shape == (2, K + 1)
    assert (tr[:, 0] == np.arange(n_tr)).all() and tr.min() >= 0 and tr.max() < n_tr and te.max() < n_tr
    train_set = D.FeaturesDataset(str(root / "train_data_features" / part))
    if testing:
        train_set = torch.utils.data.Subset(train_set, range(3))
    train_ld = DataLoader(train_set, 1, shuffle=False)
    test_ld = DataLoader(D.FeaturesDataset(str(root / "test_data_features" / part)), 1, shuffle=False)
    tr2, te2 = T.update_knn_graphs(ssa, train_ld, test_ld, K, torch.device("cuda"))
    assert np.array_equal(tr, tr2) and np.array_equal(te, te2)
