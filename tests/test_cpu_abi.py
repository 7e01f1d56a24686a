"""CPU-side checks (no GPU needed): libcsn_hip.so builds for gfx950, loads, exports every symbol that
include/csn_hip.h declares, rejects bad arguments before touching the device, and the Python surface
mirrors the reference's names.  No compute call is made here."""
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from csn_amd import _lib
    _lib.build()
    return _lib


def _declared():
    src = open(os.path.join(ROOT, "include", "csn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(csn_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(L):
    names = _declared()
    assert len(names) >= 10
    out = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (csn_[a-z0-9_]+)$", out, flags=re.M))
    missing = [n for n in names if n not in exported]
    assert not missing, missing
    assert sorted(L.EXPORTS) == names            # the ctypes binding covers exactly the header
    # ... and the header covers exactly what the library exports (-fvisibility=hidden + CSN_API): no internal launcher, no
    # mangled C++ symbol; the `__hip_*` data words are the HIP toolchain's own registration markers
    others = [l.split()[-1] for l in out.splitlines() if l.strip() and l.split()[-1] not in names and not l.split()[-1].startswith("__hip_")]
    assert not others, others


def test_library_loads_and_reports_version(L):
    lib = L.lib()
    assert lib.csn_version() == 17
    assert lib.csn_status_string(0) == b"ok"
    assert b"workspace" in lib.csn_status_string(-6)


def test_argument_validation_happens_on_the_host(L):
    lib = L.lib()
    # null pointers / bad sizes are rejected before any launch, so this is safe without a GPU
    assert lib.csn_project_f32(None, 0, 4, None, 1, 1, None, 0, 4, 1, 4, 0, 1.0, 0, 0, None) == -1
    assert lib.csn_block_attn_fwd_f32(None, None, None, 0, 0, None, None, 4, None, 0, None, None, 1, 1, 32, 4, 1, 32, 0.0, 0.0, 0, 0, 0, None) == -1
    assert lib.csn_retrieval_measure_f32(None, None, None, 1, 1, 1, 1, 4, None, 0, None) == -1
    assert lib.csn_wgrad_workspace_floats(256, 256, 0, 10) == 0
    n = lib.csn_wgrad_workspace_floats(256, 256, 128, 10000)
    assert n % (256 * 256) == 0 and n // (256 * 256) >= 128
    with pytest.raises(L.CsnError):
        L.check(-5, "demo")


def test_python_surface_mirrors_reference_names():
    import csn_amd.csa_models as m
    for name in ("get_model", "CrossShapeAt", "MultiHeadAttention", "ScaledDotProductAttention",
                 "backbone_fc_ssa_logit", "backbone_fc_csa_logit"):
        assert hasattr(m, name)
    model = m.get_model("csa", 39, 1, 3)
    assert model.attention.block == 500 and model.attention.n_blocks == 20       # csa_models.py:83-84
    assert model.attention.norm.eps == 1e-6                                      # csa_models.py:57
    assert model.attention.w_qs.bias is None and model.attention.fc.bias is None
    assert model.K == 3
    for fn in ("forward_ssa", "forward_csa", "get_ssa_feats", "get_csa_feats", "get_retrieval_measure", "get_knn_graph",
               "get_all_feats", "get_center_shape_indices", "get_knn_graph_big"):
        assert callable(getattr(model, fn))


def test_cpu_tensors_are_refused_not_silently_computed():
    import csn_amd.csa_models as m
    from csn_amd import CsnError
    if torch.cuda.is_available():
        pytest.skip("covered by the gpu suite")
    model = m.get_model("ssa", 4, 1).eval()
    with pytest.raises((CsnError, RuntimeError, AssertionError)):
        model(torch.zeros(1, 256, 10000, 1), "test")


def test_oracle_is_not_imported_by_the_product():
    for root, _, files in os.walk(os.path.join(ROOT, "csn_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(root, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), f


def test_eval_plan_colours_and_groups_cover_the_same_sharing():
    """Host logic of the backward's slot sharing (csn_amd.functional.EvalPlan): colours never repeat a slot inside one
    launch; groups list every evaluation once, adjacent by slot, biggest groups first."""
    import numpy as np
    import torch
    from csn_amd.functional import EvalPlan
    B, K1 = 3, 4
    b, k = np.meshgrid(np.arange(B), np.arange(K1), indexing="ij")
    q = np.concatenate(((b * K1).reshape(-1), (b * K1 + k)[:, 1:].reshape(-1), (b * K1)[:, 0]))       # csa_train shape
    kv = np.concatenate(((b * K1 + k).reshape(-1), (b * K1 + k)[:, 1:].reshape(-1), (b * K1)[:, 0]))
    plan = EvalPlan(q, kv, B * K1, torch.device("cpu"))
    assert plan.E == q.size
    for colours, slots in ((plan.dq_colors, q), (plan.dkv_colors, kv)):
        seen = np.concatenate([c.numpy() for c in colours])
        assert sorted(seen.tolist()) == list(range(plan.E))
        for c in colours:
            assert len(set(slots[c.numpy()].tolist())) == c.numel()
    for items, off, n, slots in ((plan.q_group_items, plan.q_group_off, plan.n_q_groups, q),
                                 (plan.kv_group_items, plan.kv_group_off, plan.n_kv_groups, kv)):
        items, off = items.numpy(), off.numpy()
        assert sorted(items.tolist()) == list(range(plan.E)) and off[0] == 0 and off[-1] == plan.E and n == off.size - 1
        sizes = np.diff(off)
        assert (sizes > 0).all() and (np.diff(sizes) <= 0).all()                    # biggest groups first
        group_slots = [set(slots[items[off[g]:off[g + 1]]].tolist()) for g in range(n)]
        assert all(len(s) == 1 for s in group_slots)                                 # one slot per group ...
        assert len(set.union(*group_slots)) == n == len(set(slots.tolist()))         # ... and one group per slot
    assert plan.n_q_groups == B * K1 and int(np.diff(plan.q_group_off.numpy()).max()) == K1 + 1    # own slots: K+2 evaluations


def test_half_and_double_tensors_are_refused_before_any_pointer_is_taken():
    """The library reads its operands as fp32 words: a half or double tensor must be refused, not reinterpreted (a half buffer
    read as fp32 runs past its allocation).  The gate checks the type before the device, so this runs without a GPU."""
    from csn_amd import CsnError
    from csn_amd import functional as CF
    for bad in (torch.float16, torch.float64, torch.bfloat16, torch.int64):
        with pytest.raises(CsnError, match="fp32"):
            CF._need_cuda(torch.zeros(4, dtype=bad))
    with pytest.raises(CsnError, match="fp32"):
        CF.project(torch.zeros(1, 4, 4, dtype=torch.float64), torch.zeros(4, 4))
    with pytest.raises(CsnError, match="fp32"):
        CF.project(torch.zeros(1, 4, 4), torch.zeros(4, 4, dtype=torch.float16))
    with pytest.raises(CsnError, match="fp32"):
        CF.compat_head(torch.zeros(1, 2, 4, dtype=torch.float64), torch.zeros(4, 4), torch.zeros(4), torch.zeros(4, 4), torch.zeros(4))
    # the call sites that carry 16-bit maps by design name the type they take, and only that one
    CF._need_cuda(None)
    with pytest.raises(CsnError, match="cuda"):
        CF._need_cuda(torch.zeros(4, dtype=torch.bfloat16), also=(torch.bfloat16,))      # type accepted, device refused
    with pytest.raises(CsnError, match="fp32"):
        CF._need_cuda(torch.zeros(4, dtype=torch.float16), also=(torch.bfloat16,))
