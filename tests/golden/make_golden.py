"""Generate the golden vectors under tests/golden/ by running the REFERENCE module itself.

Run once, in the build container (the only place /root/reference exists):

    python tests/golden/make_golden.py

Inputs and parameters are drawn from ``numpy.random.default_rng(seed)`` (a version-stable stream) through
``oracle.csa_oracle.make_params`` / ``synth_points`` so the tests can regenerate them bit-for-bit on
any box; only the reference's OUTPUTS (sub-sampled where large) are stored.  The reference is imported,
never copied: nothing from it is written anywhere but numbers.
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference/MID-FC")

import csa_models as ref  # noqa: E402  (the reference)
from oracle import csa_oracle as orc  # noqa: E402

torch.set_num_threads(8)

ROW_STRIDE = 97


def sample_rows(t: torch.Tensor, stride: int = ROW_STRIDE) -> np.ndarray:
    """(B, N, C) -> every stride-th row along N."""
    return t.detach()[:, ::stride].contiguous().numpy().astype(np.float32)


def stats(t: torch.Tensor) -> np.ndarray:
    t = t.detach().double()
    return np.array([t.mean().item(), t.norm().item(), t.abs().max().item()], dtype=np.float64)


def load_into(model, p):
    missing, unexpected = model.load_state_dict(p, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("fc_1.") for k in missing), missing
    return model


def g1_sdpa(out):
    for tag, (B, H, T, d), seed in [("a", (1, 1, 500, 256), 101), ("b", (2, 8, 500, 256), 102),
                                    ("c", (1, 2, 64, 32), 103)]:
        rng = np.random.default_rng(seed)
        q, k, v = (orc.synth_points(rng, (B, H, T, d)) for _ in range(3))
        m = ref.ScaledDotProductAttention(temperature=d ** 0.5).eval()
        o, pr = m(q, k, v)
        out[f"g1{tag}_shape"] = np.array([B, H, T, d, seed])
        out[f"g1{tag}_out"] = o.reshape(B * H, T, d)[:, ::31].numpy()
        out[f"g1{tag}_prob"] = pr.reshape(B * H, T, T)[:, ::31].numpy()
        out[f"g1{tag}_out_stats"] = stats(o)


def g2_self_attention(out):
    cases = [(500, 256, 1), (512, 128, 2), (2048, 128, 1), (2048, 96, 2), (512, 256, 2)]
    for i, (N, C, H) in enumerate(cases):
        seed = 200 + i
        rng = np.random.default_rng(seed)
        p = orc.make_params(rng, H, d_model=C, d_k=C, d_v=C, csa=False)
        x = orc.synth_points(rng, (1, C, N, 1))
        m = ref.MultiHeadAttention(H, C, C, C).eval()
        m.load_state_dict({k[len("attention."):]: v for k, v in p.items() if k.startswith("attention.")})
        y, _ = m.self_attention(x)
        out[f"g2_{i}_cfg"] = np.array([N, C, H, seed])
        out[f"g2_{i}_rows"] = sample_rows(y, 29)
        out[f"g2_{i}_stats"] = stats(y)


def g3_mha_forward(out):
    for i, H in enumerate([1, 8]):
        seed = 300 + i
        rng = np.random.default_rng(seed)
        p = orc.make_params(rng, H, csa=False)
        xa = orc.synth_points(rng, (1, 256, 10000, 1))
        xb = orc.synth_points(rng, (1, 256, 10000, 1))
        m = ref.MultiHeadAttention(H, 256, 256, 256).eval()
        m.load_state_dict({k[len("attention."):]: v for k, v in p.items() if k.startswith("attention.")})
        with torch.no_grad():
            ys, _ = m(xa, xa, xa, "test")
            yc, attn_last = m(xa, xb, xb, "test")
        out[f"g3_{i}_cfg"] = np.array([H, seed])
        out[f"g3_{i}_self_rows"] = sample_rows(ys)
        out[f"g3_{i}_cross_rows"] = sample_rows(yc)
        out[f"g3_{i}_self_stats"] = stats(ys)
        out[f"g3_{i}_cross_stats"] = stats(yc)
        out[f"g3_{i}_attn_last_row0"] = attn_last[0, :, 0].numpy()      # (H, 500) last chunk, query 0


def grads_pack(model, out, key, truth=None):
    """Reference gradients (sub-sampled where large) + their norms.  ``truth`` = the same gradients from the
    float64 oracle: the reference's own fp32 deviation from it is stored as the per-tensor noise floor
    (``gnoise``) — the compatibility-head gradients are differences of large numbers and are only accurate
    to ~1e-3..1e-2 relative in the reference itself."""
    for name, prm in model.named_parameters():
        if prm.grad is None:
            out[f"{key}_nograd_{name}"] = np.zeros(1)
            continue
        g = prm.grad.detach()
        out[f"{key}_gstats_{name}"] = stats(g)
        if truth is not None:
            out[f"{key}_gnoise_{name}"] = np.array([(g.double() - truth[name]).abs().max().item()])
        if g.numel() <= 10000:
            out[f"{key}_grad_{name}"] = g.numpy().astype(np.float32)
        else:
            out[f"{key}_grad_{name}"] = g.reshape(g.shape[0], -1)[::17, ::13].contiguous().numpy().astype(np.float32)


def truth64(fwd, p, lab):
    """float64 oracle gradients (the yardstick for the reference's own fp32 rounding noise)."""
    q = {k: v.double().clone().requires_grad_(True) for k, v in p.items()}
    orc.masked_ce_loss(fwd(q), lab).backward()
    return {k: v.grad for k, v in q.items()}


def labels_for(rng, B, N, n_cls):
    return orc.synth_labels(rng, B, N, n_cls)       # ~10 % unlabeled -> exercises the mask (csa_training.py:101)


def g4_csa(out):
    n_cls = 39
    for i, (B, K, H) in enumerate([(1, 2, 1), (2, 3, 1), (1, 2, 8)]):
        seed = 400 + i
        rng = np.random.default_rng(seed)
        p = orc.make_params(rng, H, n_cls=n_cls, csa=True)
        x = orc.synth_points(rng, (B, 256, 10000, 1))
        nb = orc.synth_points(rng, (B, K + 1, 256, 10000, 1))
        nb[:, 0] = x
        lab = labels_for(rng, B, 10000, n_cls)
        model = load_into(ref.get_model("csa", n_cls, H, K), p).eval()
        logits = model(x, "test", nb)
        loss = orc.masked_ce_loss(logits, lab)
        loss.backward()
        with torch.no_grad():
            feats, comp, pooled = orc.csa_feats(x, nb, p, H, return_parts=True)   # oracle's comp, cross-checked below
            ref_feats = model.get_csa_feats(x, nb, "test")
        assert (feats - ref_feats).abs().max().item() < 2e-5
        out[f"g4_{i}_cfg"] = np.array([B, K, H, n_cls, seed])
        out[f"g4_{i}_logit_rows"] = logits.detach().squeeze(-1).permute(0, 2, 1)[:, ::ROW_STRIDE].contiguous().numpy()
        out[f"g4_{i}_feat_rows"] = ref_feats.squeeze(-1).permute(0, 2, 1)[:, ::ROW_STRIDE].contiguous().numpy()
        out[f"g4_{i}_feat_stats"] = stats(ref_feats)
        out[f"g4_{i}_loss"] = np.array([loss.item()], dtype=np.float64)
        out[f"g4_{i}_comp_oracle"] = comp.numpy()
        grads_pack(model, out, f"g4_{i}", truth64(lambda q: orc.forward_csa(x.double(), nb.double(), q, H), p, lab))


def g5_ssa(out):
    n_cls = 39
    for i, (B, H) in enumerate([(2, 1), (1, 8)]):
        seed = 500 + i
        rng = np.random.default_rng(seed)
        p = orc.make_params(rng, H, n_cls=n_cls, csa=False)
        x = orc.synth_points(rng, (B, 256, 10000, 1))
        lab = labels_for(rng, B, 10000, n_cls)
        model = load_into(ref.get_model("ssa", n_cls, H), p).eval()
        logits = model(x, "train")
        loss = orc.masked_ce_loss(logits, lab)
        loss.backward()
        out[f"g5_{i}_cfg"] = np.array([B, H, n_cls, seed])
        out[f"g5_{i}_logit_rows"] = logits.detach().squeeze(-1).permute(0, 2, 1)[:, ::ROW_STRIDE].contiguous().numpy()
        out[f"g5_{i}_loss"] = np.array([loss.item()], dtype=np.float64)
        grads_pack(model, out, f"g5_{i}", truth64(lambda q: orc.forward_ssa(x.double(), q, H), p, lab))


def g6_retrieval(out):
    model = ref.get_model("ssa", 4, 1).eval()
    for i, (S, N, K) in enumerate([(6, 300, 2), (16, 1000, 3)]):
        seed = 600 + i
        rng = np.random.default_rng(seed)
        # clustered features so the ranking is not a coin toss between near-equal scores
        f = orc.synth_clustered_feats(rng, S, N)
        with torch.no_grad():
            r = model.get_retrieval_measure(f, f)
            g = model.get_knn_graph(f, f, K)
        out[f"g6_{i}_cfg"] = np.array([S, N, K, seed])
        out[f"g6_{i}_measure"] = r.numpy()
        out[f"g6_{i}_graph"] = g.numpy().astype(np.int64)


G7_CASES = [(1, 2, 1, 39, 4.0, 1.0, 0.8), (2, 3, 1, 39, 4.0, 3.0, 1.0),     # B, K, H, n_cls, fc scale, w_qs scale, offset
            (1, 4, 8, 39, 4.0, 1.0, 0.8)]                                   # the published checkpoint's geometry (get_csa_pred.py:35-36)


def g7_csa_conditioned(out):
    """CSA forward + all 11 gradients on inputs where the compatibility-head gradients are well-conditioned
    (oracle.conditioned_csa_case): there the reference's own fp32 noise (gnoise, stored) is ~1e-6 relative, so the GPU
    tests hold every tensor to 1e-4 with no noise allowance."""
    for i, (B, K, H, n_cls, fc_s, q_s, off) in enumerate(G7_CASES):
        seed = 700 + i
        rng = np.random.default_rng(seed)
        p, x, nb, lab = orc.conditioned_csa_case(rng, B, K, H, n_cls, fc_s, q_s, off)
        model = load_into(ref.get_model("csa", n_cls, H, K), p).eval()
        logits = model(x, "test", nb)
        loss = orc.masked_ce_loss(logits, lab)
        loss.backward()
        with torch.no_grad():
            feats, comp, pooled = orc.csa_feats(x, nb, p, H, return_parts=True)
            ref_feats = model.get_csa_feats(x, nb, "test")
        assert (feats - ref_feats).abs().max().item() < 5e-5
        out[f"g7_{i}_cfg"] = np.array([B, K, H, n_cls, seed])
        out[f"g7_{i}_scales"] = np.array([fc_s, q_s, off], dtype=np.float64)
        out[f"g7_{i}_logit_rows"] = logits.detach().squeeze(-1).permute(0, 2, 1)[:, ::ROW_STRIDE].contiguous().numpy()
        out[f"g7_{i}_feat_rows"] = ref_feats.squeeze(-1).permute(0, 2, 1)[:, ::ROW_STRIDE].contiguous().numpy()
        out[f"g7_{i}_loss"] = np.array([loss.item()], dtype=np.float64)
        out[f"g7_{i}_comp_oracle"] = comp.numpy()
        grads_pack(model, out, f"g7_{i}", truth64(lambda q: orc.forward_csa(x.double(), nb.double(), q, H), p, lab))
        worst = max(float(out[f"g7_{i}_gnoise_{n}"][0]) / float(out[f"g7_{i}_gstats_{n}"][2])
                    for n, prm in model.named_parameters() if prm.grad is not None)
        print(f"g7 case {i}: comp {comp.numpy().round(4).tolist()}, worst reference-vs-fp64 relative gradient noise {worst:.2e}", flush=True)
        assert worst < 5e-5, "the case is not well-conditioned"


G8_CASES = [(2, 256, 64, 128, 10000, "forward"), (3, 96, 48, 80, 400, "self_attention"), (1, 256, 256, 96, 10000, "forward")]


def g8_mha_unequal_head_widths(out):
    """MultiHeadAttention with d_k != d_v and head widths the kernels have no instance for (csa_models.py:42 allows both; no
    caller uses them): outputs of the chunked forward (cross: K / V from a second shape) or the unchunked self_attention,
    and the gradients of the six weight tensors under loss = sum(y * g)."""
    for i, (H, C, dk, dv, N, how) in enumerate(G8_CASES):
        seed = 800 + i
        rng = np.random.default_rng(seed)
        p = orc.make_params(rng, H, d_model=C, d_k=dk, d_v=dv, csa=False)
        xa = orc.synth_points(rng, (1, C, N, 1))
        xb = orc.synth_points(rng, (1, C, N, 1))
        g = orc.synth_points(rng, (1, N, C))
        m = ref.MultiHeadAttention(H, C, dk, dv).eval()
        m.load_state_dict({k[len("attention."):]: v for k, v in p.items() if k.startswith("attention.")})
        y = m(xa, xb, xb, "test")[0] if how == "forward" else m.self_attention(xa)[0]
        (y * g).sum().backward()
        out[f"g8_{i}_cfg"] = np.array([H, C, dk, dv, N, seed, 1 if how == "forward" else 0])
        out[f"g8_{i}_rows"] = sample_rows(y, 29)
        out[f"g8_{i}_stats"] = stats(y)
        for name, prm in m.named_parameters():
            gr = prm.grad.detach()
            out[f"g8_{i}_gstats_{name}"] = stats(gr)
            out[f"g8_{i}_grad_{name}"] = (gr if gr.numel() <= 10000 else gr.reshape(gr.shape[0], -1)[::5, ::7].contiguous()).numpy().astype(np.float32)


# ---- G9: the feature-file data path (features_data_loader.py:79-140); inputs in g9_spec.py -----------------------------
import g9_spec as g9  # noqa: E402


def g9_data_path(out):
    """Items of the REFERENCE's CSADatasetK / FeaturesDataset over seeded temp files: per item the shapes, dtypes, sha256 of the
    raw bytes (bit-for-bit pin) and a strided sample of feats / label / neighbor_feats, for a train graph and a test graph."""
    import tempfile
    import features_data_loader as fdl          # the reference (MID-FC on sys.path)
    listdir = os.listdir
    os.listdir = lambda p: sorted(listdir(p))    # os.listdir order is filesystem-dependent: both readers see the sorted order
    try:
        with tempfile.TemporaryDirectory() as tmp:
            tr, te = g9.write_both(tmp)
            for tag, root, graph in (("train", tr, g9.TRAIN_GRAPH), ("test", te, g9.TEST_GRAPH)):
                ds = fdl.CSADatasetK(root, tr, np.array(graph), g9.K)
                plain = fdl.FeaturesDataset(root, "backbone_fc_csa_logit")
                out[f"g9_{tag}_len"] = np.array([len(ds), len(plain)])
                for i in range(len(ds)):
                    f, lab, nb = ds[i]
                    pf, pl = plain[i]
                    out[f"g9_{tag}_{i}_shapes"] = np.array(list(f.shape) + list(lab.shape) + list(nb.shape) + list(pf.shape))
                    out[f"g9_{tag}_{i}_dtypes"] = np.array([str(f.dtype), str(lab.dtype), str(nb.dtype), str(pf.dtype), str(pl.dtype)])
                    out[f"g9_{tag}_{i}_sha"] = np.array([g9.digest(f), g9.digest(lab), g9.digest(nb), g9.digest(pf), g9.digest(pl)])
                    out[f"g9_{tag}_{i}_feats"] = f[::16, ::997, 0].numpy()
                    out[f"g9_{tag}_{i}_label"] = lab[::97].numpy()
                    out[f"g9_{tag}_{i}_nb"] = nb[:, ::16, ::997, 0].numpy()
    finally:
        os.listdir = listdir


def g10_after_fc_false(out):
    """`CrossShapeAt(..., after_fc=False)` (csa_models.py:147, 191-202): the attention layer is skipped, the model is the logit
    layer alone — on EVERY point of the input (the 20 x 500 chunking lives in the attention, csa_models.py:83-90)."""
    n_cls = 39
    for i, (kind, B, N, H, K) in enumerate([("ssa", 2, 10000, 1, None), ("csa", 1, 7001, 8, 2), ("ssa", 1, 12000, 1, None)]):
        seed = 1000 + i
        rng = np.random.default_rng(seed)
        p = orc.make_params(rng, H, n_cls=n_cls, csa=kind == "csa")
        x = orc.synth_points(rng, (B, 256, N, 1))
        lab = labels_for(rng, B, N, n_cls)
        model = load_into(ref.CrossShapeAt(n_cls, 256, H, K, attention_type=kind, after_fc=False), p).eval()
        nb = orc.synth_points(rng, (B, K + 1, 256, N, 1)) if kind == "csa" else None
        logits = model(x, "train", nb)
        assert tuple(logits.shape) == (B, n_cls, N, 1)
        loss = orc.masked_ce_loss(logits, lab)
        loss.backward()
        out[f"g10_{i}_cfg"] = np.array([0 if kind == "ssa" else 1, B, N, H, K or 0, n_cls, seed])
        out[f"g10_{i}_logit_rows"] = logits.detach().squeeze(-1).permute(0, 2, 1)[:, ::ROW_STRIDE].contiguous().numpy()
        out[f"g10_{i}_loss"] = np.array([loss.item()], dtype=np.float64)
        grads = {n: prm.grad for n, prm in model.named_parameters() if prm.grad is not None}
        assert sorted(grads) == ["logit.weight"], sorted(grads)           # nothing else is on the path
        out[f"g10_{i}_grad_logit.weight"] = grads["logit.weight"].detach().numpy().astype(np.float32)


def g11_big_category_graph(out):
    """The big-category branch of update_knn_graphs (csa_training.py:138-155) run on the REFERENCE: k-means centre shapes
    (`get_center_shape_indices`, csa_models.py:302-332 — sklearn KMeans, whatever version this image carries: the GPU box is the same
    image) and the candidate-relative kNN tables of `get_knn_graph_big` (:334-404) for the train and the test loader."""
    S_train, S_test, n_centers, K, seed = 30, 3, 3, 2, 1100
    rng = np.random.default_rng(seed)
    p = orc.make_params(rng, 1, n_cls=4, csa=False)
    train = orc.synth_clustered_shapes(rng, S_train, n_centers)
    test = orc.synth_clustered_shapes(rng, S_test, n_centers)
    model = load_into(ref.get_model("ssa", 4, 1), {k: v for k, v in p.items() if not k.startswith("compat")}).eval()
    model.device = torch.device("cpu")                                         # (csa_training.py:283)
    lab = torch.zeros((1, 10000), dtype=torch.int64)
    loader = lambda shapes: [(x[None], lab) for x in shapes]                   # batches of one: (1, 1, 256, N, 1), like DataLoader(FeaturesDataset, 1)
    with torch.no_grad():
        centres = model.get_center_shape_indices(loader(train))
    assert len(centres) == S_train // 10
    tr = model.get_knn_graph_big(loader(train), loader(train), centres.copy(), K)
    te = model.get_knn_graph_big(loader(test), loader(train), centres.copy(), K)
    meas = model.get_retrieval_measure_big(loader(test), loader(train), centres.copy())
    out["g11_cfg"] = np.array([S_train, S_test, n_centers, K, seed])
    out["g11_centres"] = np.sort(np.asarray(centres)).astype(np.int64)
    out["g11_train_graph"] = tr.numpy().astype(np.int64)
    out["g11_test_graph"] = te.numpy().astype(np.int64)
    out["g11_test_measure"] = meas.numpy().astype(np.float32)


def main():
    only = set(sys.argv[1:])
    for name, fn in [("g1_sdpa", g1_sdpa), ("g2_self_attention", g2_self_attention),
                     ("g3_mha_forward", g3_mha_forward), ("g4_csa", g4_csa), ("g5_ssa", g5_ssa),
                     ("g6_retrieval", g6_retrieval),
                     ("g7_csa_conditioned", g7_csa_conditioned), ("g8_mha_unequal_head_widths", g8_mha_unequal_head_widths),
                     ("g9_data_path", g9_data_path), ("g10_after_fc_false", g10_after_fc_false),
                     ("g11_big_category_graph", g11_big_category_graph)]:
        if only and name not in only:
            continue
        out = {}
        fn(out)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, "->", path, f"{os.path.getsize(path) / 1024:.0f} KiB", flush=True)


if __name__ == "__main__":
    main()
