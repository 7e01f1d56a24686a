"""The store-data hazard of gfx950 (DESIGN, platform findings; csn_common.h `csn_store_guard`): a 12 / 16-byte buffer store with a
REGISTER soffset needs two wait states before its data registers are rewritten, and hipcc pads them only for an immediate
soffset.  hipcc -S needs no GPU, so every CPU test run scans the assembly of every source that ships in libcsn_hip.so and
fails on any site within two instructions of such a store — a compiler bump or code motion cannot bring the wrong bits back
unnoticed."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scanner():
    spec = importlib.util.spec_from_file_location("scan_store_hazard", os.path.join(ROOT, "scripts", "dev", "scan_store_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_scanner_sees_the_pattern_it_is_looking_for():
    s = _scanner()
    bad1 = ["_Zk:", "\tbuffer_store_dwordx4 v[0:3], v10, s[44:47], s64 offen", "\tv_add_f32_e32 v0, v0, v1"]
    bad2 = ["_Zk:", "\tbuffer_store_dwordx4 v[0:3], v8, s[0:3], s12 offen", "\ts_add_u32 s1, s1, s2", "\tv_add_u32_e32 v0, v18, v19"]
    imm = ["_Zk:", "\tbuffer_store_dwordx4 v[0:3], v8, s[0:3], 0 offen offset:64", "\tv_add_u32_e32 v0, v18, v19"]     # hipcc pads this form itself
    padded = ["_Zk:", "\tbuffer_store_dwordx4 v[0:3], v8, s[0:3], s12 offen", "\ts_nop 1", "\tv_add_u32_e32 v0, v18, v19"]
    far = ["_Zk:", "\tbuffer_store_dwordx4 v[0:3], v8, s[0:3], s12 offen", "\tv_mov_b32_e32 v9, v4", "\tv_mov_b32_e32 v10, v4",
           "\tv_add_u32_e32 v0, v18, v19"]
    assert [x[2] for x in s.sites_in(bad1)] == [1]
    assert [x[2] for x in s.sites_in(bad2)] == [2]
    assert s.sites_in(imm) == [] and s.sites_in(padded) == [] and s.sites_in(far) == []
    # `s_nop 0` is ONE wait state: the rewrite behind it is still inside the window; two of them close it
    nop0 = ["_Zk:", "\tbuffer_store_dwordx4 v[0:3], v8, s[0:3], s12 offen", "\ts_nop 0", "\tv_add_u32_e32 v0, v18, v19"]
    nop00 = ["_Zk:", "\tbuffer_store_dwordx4 v[0:3], v8, s[0:3], s12 offen", "\ts_nop 0", "\ts_nop 0", "\tv_add_u32_e32 v0, v18, v19"]
    assert [x[2] for x in s.sites_in(nop0)] == [2] and s.sites_in(nop00) == []
    # destinations beyond VALU / loads: LDS permutes, typed-buffer loads, returning atomics; a 64-bit vaddr store form; reads are not hits
    for instr in ("ds_bpermute_b32 v2, v9, v10", "ds_swizzle_b32 v1, v9 offset:swizzle(SWAP,1)", "tbuffer_load_format_x v3, v9, s[0:3], 0 offen",
                  "buffer_atomic_add_u32 v0, v9, s[0:3], 0 offen sc0", "v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]"):
        assert [x[2] for x in s.sites_in(["_Zk:", "\tbuffer_store_dwordx4 v[0:3], v[8:9], s[0:3], s12 addr64", "\t" + instr])] == [1], instr
    for instr in ("ds_write_b32 v0, v1", "buffer_store_dword v0, v9, s[0:3], 0 offen", "v_cmp_gt_u32_e32 vcc, v0, v1", "s_mov_b32 s0, s1"):
        assert s.sites_in(["_Zk:", "\tbuffer_store_dwordx4 v[0:3], v8, s[0:3], s12 offen", "\t" + instr, "\ts_nop 0"]) == [], instr
    # a store at a loop tail is checked against the loop head the branch goes back to (and against the fall-through)
    loop = ["_Zk:", ".LBB0_1:", "\tv_add_u32_e32 v0, v18, v19", "\tv_mov_b32_e32 v30, v31",
            "\tbuffer_store_dwordx4 v[0:3], v8, s[0:3], s12 offen", "\ts_cbranch_scc1 .LBB0_1", "\tv_mov_b32_e32 v40, v41", "\ts_endpgm"]
    assert [(x[2], x[3]) for x in s.sites_in(loop)] == [(2, "v_add_u32_e32 v0, v18, v19")]
    fall = ["_Zk:", ".LBB0_1:", "\tv_mov_b32_e32 v30, v31", "\tbuffer_store_dwordx4 v[0:3], v8, s[0:3], s12 offen",
            "\ts_cbranch_scc1 .LBB0_1", "\tv_mov_b32_e32 v1, v41", "\ts_endpgm"]
    assert [(x[2], x[3]) for x in s.sites_in(fall)] == [(2, "v_mov_b32_e32 v1, v41")]


def test_no_shipped_kernel_rewrites_store_data_within_two_wait_states():
    s = _scanner()
    if not s.have_hipcc():
        pytest.skip(f"no hipcc ({s.HIPCC}): the assembly scan needs the ROCm compiler")
    from csn_amd import _lib
    found = s.scan(list(_lib.SOURCES), max_distance=2)
    assert sorted(found) == sorted(_lib.SOURCES)
    bad = [f"{f}: {fn[:50]} line {i}: {st} -> +{j}: {nxt}" for f, sites in found.items() for i, st, j, nxt, fn in sites]
    assert not bad, "\n".join(bad)
