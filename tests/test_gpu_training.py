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
