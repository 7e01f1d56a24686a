"""Pins oracle/csa_oracle.py to the golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only; inputs are regenerated from the recorded seeds."""
import os

import numpy as np
import pytest
import torch

from oracle import csa_oracle as orc

ROW_STRIDE = 97
TOL = 2e-5          # reference (MKL, 8 threads) vs oracle (different blocking): fp32 rounding only


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _stats(t):
    t = t.detach().double()
    return np.array([t.mean().item(), t.norm().item(), t.abs().max().item()])


def _close(a, b, tol=TOL):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max()
    assert err <= tol * max(1.0, np.abs(b).max()), err


def test_g1_sdpa(golden_dir):
    g = _load(golden_dir, "g1_sdpa")
    for tag in "abc":
        B, H, T, d, seed = g[f"g1{tag}_shape"]
        rng = np.random.default_rng(int(seed))
        q, k, v = (orc.synth_points(rng, (B, H, T, d)) for _ in range(3))
        o, pr = orc.sdpa(q, k, v, float(d) ** 0.5)
        _close(o.reshape(B * H, T, d)[:, ::31].numpy(), g[f"g1{tag}_out"])
        _close(pr.reshape(B * H, T, T)[:, ::31].numpy(), g[f"g1{tag}_prob"], 1e-6)
        _close(_stats(o), g[f"g1{tag}_out_stats"], 1e-5)


def test_g2_self_attention(golden_dir):
    g = _load(golden_dir, "g2_self_attention")
    i = 0
    while f"g2_{i}_cfg" in g:
        N, C, H, seed = (int(v) for v in g[f"g2_{i}_cfg"])
        rng = np.random.default_rng(seed)
        p = orc.make_params(rng, H, d_model=C, d_k=C, d_v=C, csa=False)
        x = orc.synth_points(rng, (1, C, N, 1))
        y = orc.mha_full_self(x, p, H, C, C)
        _close(y[:, ::29].numpy(), g[f"g2_{i}_rows"])
        _close(_stats(y), g[f"g2_{i}_stats"], 1e-5)
        i += 1
    assert i == 5


@pytest.mark.parametrize("flavour", ["blockdiag", "faithful"])
def test_g3_mha_forward(golden_dir, flavour):
    g = _load(golden_dir, "g3_mha_forward")
    for i in range(2):
        H, seed = (int(v) for v in g[f"g3_{i}_cfg"])
        if flavour == "faithful" and H == 8:
            continue                                  # slow; blockdiag covers H=8
        rng = np.random.default_rng(seed)
        p = orc.make_params(rng, H, csa=False)
        xa = orc.synth_points(rng, (1, 256, 10000, 1))
        xb = orc.synth_points(rng, (1, 256, 10000, 1))
        with torch.no_grad():
            if flavour == "blockdiag":
                ys = orc.mha_blockdiag(xa, xa, xa, p, H)
                yc, attn = orc.mha_blockdiag(xa, xb, xb, p, H, return_attn=True)
                last_row0 = attn[0, -1, :, 0]
            else:
                ys, _ = orc.mha_faithful(xa, xa, xa, p, H)
                yc, attn = orc.mha_faithful(xa, xb, xb, p, H)
                last_row0 = attn[0, :, 0]
        _close(ys[:, ::ROW_STRIDE].numpy(), g[f"g3_{i}_self_rows"])
        _close(yc[:, ::ROW_STRIDE].numpy(), g[f"g3_{i}_cross_rows"])
        _close(_stats(ys), g[f"g3_{i}_self_stats"], 1e-5)
        _close(_stats(yc), g[f"g3_{i}_cross_stats"], 1e-5)
        _close(last_row0.numpy(), g[f"g3_{i}_attn_last_row0"], 1e-6)


def _check_grads(g, key, grads):
    seen = 0
    for name, gr in grads.items():
        if f"{key}_nograd_{name}" in g:
            assert gr is None or float(gr.abs().max()) == 0.0
            continue
        ref = g[f"{key}_grad_{name}"]
        got = gr.numpy() if gr.numel() <= 10000 else gr.reshape(gr.shape[0], -1)[::17, ::13].contiguous().numpy()
        scale = max(np.abs(ref).max(), 1e-30)
        # 1e-4 relative + the reference's own fp32 noise floor on this tensor (vs the float64 oracle, see make_golden.py)
        noise = float(g[f"{key}_gnoise_{name}"][0])
        assert np.abs(got - ref).max() <= 1e-4 * scale + 4.0 * noise, (name, np.abs(got - ref).max(), scale, noise)
        st = g[f"{key}_gstats_{name}"]
        assert abs(gr.double().norm().item() - st[1]) <= 1e-4 * st[1] + 4.0 * noise * np.sqrt(gr.numel()), name
        seen += 1
    return seen


def _run_model(p, fwd, lab):
    p = {k: v.clone().requires_grad_(not k.startswith("fc_1")) for k, v in p.items()}
    logits = fwd(p)
    loss = orc.masked_ce_loss(logits, lab)
    loss.backward()
    return logits.detach(), loss.item(), {k: v.grad for k, v in p.items()}


def test_g4_csa(golden_dir):
    g = _load(golden_dir, "g4_csa")
    for i in range(3):
        B, K, H, n_cls, seed = (int(v) for v in g[f"g4_{i}_cfg"])
        if H == 8 and os.environ.get("CSN_SLOW", "0") != "1":
            continue                                   # ~1 min of CPU; run with CSN_SLOW=1
        rng = np.random.default_rng(seed)
        p = orc.make_params(rng, H, n_cls=n_cls, csa=True)
        x = orc.synth_points(rng, (B, 256, 10000, 1))
        nb = orc.synth_points(rng, (B, K + 1, 256, 10000, 1))
        nb[:, 0] = x
        lab = orc.synth_labels(rng, B, 10000, n_cls)
        logits, loss, grads = _run_model(p, lambda q: orc.forward_csa(x, nb, q, H), lab)
        _close(logits.squeeze(-1).permute(0, 2, 1)[:, ::ROW_STRIDE].numpy(), g[f"g4_{i}_logit_rows"])
        assert abs(loss - g[f"g4_{i}_loss"][0]) < 1e-5
        with torch.no_grad():
            feats, comp, _ = orc.csa_feats(x, nb, p, H, return_parts=True)
        _close(feats.squeeze(-1).permute(0, 2, 1)[:, ::ROW_STRIDE].numpy(), g[f"g4_{i}_feat_rows"])
        _close(comp.numpy(), g[f"g4_{i}_comp_oracle"], 1e-6)
        assert _check_grads(g, f"g4_{i}", grads) == 11


def test_g7_conditioned_csa_all_gradients_at_1e4(golden_dir):
    """G7: the compatibility-head gradients are well-conditioned here (oracle.conditioned_csa_case), so the oracle must
    reproduce all 11 of the reference's gradients to 1e-4 relative with NO noise allowance."""
    g = _load(golden_dir, "g7_csa_conditioned")
    for i in range(1 if os.environ.get("CSN_SLOW", "0") != "1" else 3):       # cases 1 (B = 2, K = 3), 2 (8 heads, K = 4): CSN_SLOW=1
        B, K, H, n_cls, seed = (int(v) for v in g[f"g7_{i}_cfg"])
        fc_s, q_s, off = (float(v) for v in g[f"g7_{i}_scales"])
        p, x, nb, lab = orc.conditioned_csa_case(np.random.default_rng(seed), B, K, H, n_cls, fc_s, q_s, off)
        logits, loss, grads = _run_model(p, lambda q: orc.forward_csa(x, nb, q, H), lab)
        _close(logits.squeeze(-1).permute(0, 2, 1)[:, ::ROW_STRIDE].numpy(), g[f"g7_{i}_logit_rows"])
        assert abs(loss - g[f"g7_{i}_loss"][0]) < 1e-5
        seen = 0
        for name, gr in grads.items():
            if gr is None:
                continue
            ref = g[f"g7_{i}_grad_{name}"]
            got = gr.numpy() if gr.numel() <= 10000 else gr.reshape(gr.shape[0], -1)[::17, ::13].contiguous().numpy()
            assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), name
            assert float(g[f"g7_{i}_gnoise_{name}"][0]) <= 5e-5 * float(g[f"g7_{i}_gstats_{name}"][2]), name   # the case IS conditioned
            seen += 1
        assert seen == 11


def test_g5_ssa(golden_dir):
    g = _load(golden_dir, "g5_ssa")
    for i in range(2):
        B, H, n_cls, seed = (int(v) for v in g[f"g5_{i}_cfg"])
        if H == 8 and os.environ.get("CSN_SLOW", "0") != "1":
            continue
        rng = np.random.default_rng(seed)
        p = orc.make_params(rng, H, n_cls=n_cls, csa=False)
        x = orc.synth_points(rng, (B, 256, 10000, 1))
        lab = orc.synth_labels(rng, B, 10000, n_cls)
        logits, loss, grads = _run_model(p, lambda q: orc.forward_ssa(x, q, H), lab)
        _close(logits.squeeze(-1).permute(0, 2, 1)[:, ::ROW_STRIDE].numpy(), g[f"g5_{i}_logit_rows"])
        assert abs(loss - g[f"g5_{i}_loss"][0]) < 1e-5
        assert _check_grads(g, f"g5_{i}", grads) == 7


def test_g10_after_fc_false_is_the_logit_layer_on_every_point(golden_dir):
    """golden set G10: the reference's CrossShapeAt(..., after_fc=False) (csa_models.py:191-202), 'ssa' and 'csa', point counts
    10000 / 7001 / 12000 — the oracle's restatement (forward_logit_only) against logits, loss and the one gradient."""
    g = _load(golden_dir, "g10_after_fc_false")
    for i in range(3):
        kind, B, N, H, K, n_cls, seed = (int(v) for v in g[f"g10_{i}_cfg"])
        rng = np.random.default_rng(seed)
        p = orc.make_params(rng, H, n_cls=n_cls, csa=kind == 1)
        x = orc.synth_points(rng, (B, 256, N, 1))
        lab = orc.synth_labels(rng, B, N, n_cls)
        w = p["logit.weight"].clone().requires_grad_(True)
        logits = orc.forward_logit_only(x, {"logit.weight": w})
        assert tuple(logits.shape) == (B, n_cls, N, 1)
        loss = orc.masked_ce_loss(logits, lab)
        loss.backward()
        _close(logits.detach().squeeze(-1).permute(0, 2, 1)[:, ::ROW_STRIDE].numpy(), g[f"g10_{i}_logit_rows"])
        assert abs(loss.item() - g[f"g10_{i}_loss"][0]) < 1e-5
        ref = g[f"g10_{i}_grad_logit.weight"]
        assert np.abs(w.grad.numpy() - ref).max() <= 1e-5 * np.abs(ref).max()


def test_g6_retrieval_and_knn_indices_bit_exact(golden_dir):
    g = _load(golden_dir, "g6_retrieval")
    for i in range(2):
        S, N, K, seed = (int(v) for v in g[f"g6_{i}_cfg"])
        rng = np.random.default_rng(seed)
        f = orc.synth_clustered_feats(rng, S, N)
        r = orc.retrieval_measure(f, f)
        _close(r.numpy(), g[f"g6_{i}_measure"], 1e-6)
        graph = orc.knn_graph(f, f, K)
        assert graph.dtype == torch.int64
        assert np.array_equal(graph.numpy(), g[f"g6_{i}_graph"])          # bit-exact index table


def test_g11_big_category_graph(golden_dir):
    """golden set G11: the big-category branch (csa_training.py:138-155) on the reference — k-means centre shapes
    (csa_models.py:302-332) and the candidate-relative kNN tables of get_knn_graph_big (:334-404): the oracle's restatement gives
    the same centres and the same index tables (the test loader's rows; the train table is checked on the GPU)."""
    g = _load(golden_dir, "g11_big_category_graph")
    S_train, S_test, n_centers, K, seed = (int(v) for v in g["g11_cfg"])
    rng = np.random.default_rng(seed)
    p = orc.make_params(rng, 1, n_cls=4, csa=False)
    train = orc.synth_clustered_shapes(rng, S_train, n_centers)
    test = orc.synth_clustered_shapes(rng, S_test, n_centers)
    with torch.no_grad():
        centres = orc.center_shape_indices(train, p, 1)
        assert np.array_equal(np.sort(centres), g["g11_centres"])
        meas, graph = orc.knn_graph_big(test, train, centres, K, p, 1)
    _close(meas.numpy(), g["g11_test_measure"], 1e-6)
    assert graph.dtype == torch.int64 and np.array_equal(graph.numpy(), g["g11_test_graph"])


def test_compat_layout_is_the_reference_one_for_batches():
    """B > 1: the reference scores shape b against rows b*(K+1).. of the neighbour-major stack
    (csa_models.py:220,227).  For B == 1 both layouts agree."""
    rng = np.random.default_rng(7)
    p = orc.make_params(rng, 1, csa=True)
    pooled = orc.synth_points(rng, (3, 4, 256))
    a = orc.compatibility(pooled, p, "reference")
    b = orc.compatibility(pooled, p, "per_shape")
    assert not torch.allclose(a, b)
    assert torch.allclose(orc.compatibility(pooled[:1], p, "reference"),
                          orc.compatibility(pooled[:1], p, "per_shape"))
    # explicit statement of the scramble
    stack = torch.cat([pooled[:, k] for k in range(4)], dim=0)          # rows k*B + b
    keys = stack.view(3, 4, 256)
    uq = torch.nn.functional.normalize(torch.nn.functional.linear(pooled[:, 0], p["compatibility_q.weight"], p["compatibility_q.bias"]), dim=-1)
    uk = torch.nn.functional.normalize(torch.nn.functional.linear(keys, p["compatibility_k.weight"], p["compatibility_k.bias"]), dim=-1)
    ref = torch.softmax(torch.einsum("bc,bkc->bk", uq, uk), dim=-1)
    assert torch.allclose(a, ref, atol=1e-7)


def test_pointmajor_mha_equals_full_self():
    """The MinkowskiNet restatement (oracle.mha_pointmajor) on q = k = v reproduces the MID-FC unchunked self-attention
    restatement, which the G2 goldens pin to the reference."""
    import numpy as np
    rng = np.random.default_rng(3)
    H, C, N = 4, 256, 96
    p = orc.make_params(rng, H, d_model=C, d_k=C // H, d_v=C // H)
    x = orc.synth_points(rng, (2, C, N, 1))
    ref = orc.mha_full_self(x, p, H, C // H, C // H)                       # (B, N, C)
    pts = x.squeeze(-1).permute(0, 2, 1).contiguous()                      # (B, N, C) point-major
    out, attn = orc.mha_pointmajor(pts, pts, pts, p, H, C // H, C // H)
    assert (out - ref).abs().max().item() < 2e-5
    assert attn.shape == (2, H, N, N) and (attn.sum(-1) - 1).abs().max().item() < 1e-5


def _g8_case(g, i):
    H, C, dk, dv, N, seed, chunked = (int(v) for v in g[f"g8_{i}_cfg"])
    rng = np.random.default_rng(seed)
    p = orc.make_params(rng, H, d_model=C, d_k=dk, d_v=dv, csa=False)
    xa = orc.synth_points(rng, (1, C, N, 1))
    xb = orc.synth_points(rng, (1, C, N, 1))
    gy = orc.synth_points(rng, (1, N, C))
    return H, C, dk, dv, N, chunked, p, xa, xb, gy


def test_g8_unequal_head_widths(golden_dir):
    """G8: d_k != d_v and head widths that are no multiple of 32 — the oracle against the reference's MultiHeadAttention
    (chunked cross call / unchunked self_attention), outputs and the gradients of all six weight tensors."""
    g = _load(golden_dir, "g8_mha_unequal_head_widths")
    i = 0
    while f"g8_{i}_cfg" in g:
        H, C, dk, dv, N, chunked, p, xa, xb, gy = _g8_case(g, i)
        q = {k: v.clone().requires_grad_(True) for k, v in p.items() if k.startswith("attention.")}
        y = orc.mha_blockdiag(xa, xb, xb, q, H, dk, dv) if chunked else orc.mha_full_self(xa, q, H, dk, dv)
        (y * gy).sum().backward()
        _close(y.detach()[:, ::29].numpy(), g[f"g8_{i}_rows"])
        _close(_stats(y), g[f"g8_{i}_stats"], 1e-5)
        for name, t in q.items():
            ref = g[f"g8_{i}_grad_{name[len('attention.'):]}"]
            gr = t.grad
            got = (gr if gr.numel() <= 10000 else gr.reshape(gr.shape[0], -1)[::5, ::7]).numpy()
            assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), name
        i += 1
    assert i == 3
