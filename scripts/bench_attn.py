#!/usr/bin/env python3
"""Timing of the fused block-attention entry points alone at config-3 geometry (development aid).

    python scripts/bench_attn.py [--evals 128] [--mode 1] [--drop 0.1] [--check]
"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csn_amd import _lib, functional as CF


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2]


def to_tiles(kv, T, nb, npl=2):
    """fp32 (S, R, nb*T) -> bf16 tile planes (S, R, nb*512*npl): per row and block 16 tiles of [hi 32 | lo 32] (npl = 2) or [32]."""
    S, R, _ = kv.shape
    x = torch.zeros((S, R, nb, 512), device=kv.device, dtype=torch.float32)
    x[..., :T] = kv.view(S, R, nb, T)
    x = x.view(S, R, nb, 16, 32)
    hi = x.bfloat16()
    if npl == 1:
        return hi.reshape(S, R, nb * 512).contiguous()
    lo = (x - hi.float()).bfloat16()
    return torch.stack((hi, lo), dim=4).reshape(S, R, nb * 1024).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", action="store_true", help="K/V as bf16 tile planes (math mode 1)")
    ap.add_argument("--noscores", action="store_true", help="forward without saving the scores (inference form)")
    ap.add_argument("--evals", type=int, default=128)
    ap.add_argument("--slots", type=int, default=32)
    ap.add_argument("--mode", type=int, default=1)
    ap.add_argument("--drop", type=float, default=0.1)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--only", default="fwd,dq,dkv")
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--nb", type=int, default=20)
    ap.add_argument("--dev", default="", help="development switches for the timed runs, key=value,... (csn_dev_set; 4=0: forward on the 8-wave kernel)")
    ap.add_argument("--digest", action="store_true", help="print sha256 digests of every output of one fwd + dq (+ dkv) pass: two builds "
                    "(CSN_LIB_PATH) give the same lines iff they give the same bits")
    ap.add_argument("--recompute", type=int, default=0, help="1: dq rebuilds the scores, P / dS planes still written; 2: nothing written")
    a = ap.parse_args()
    L = _lib.lib()
    H, d, T, nb = 1, a.d, 500, a.nb
    D, NP, Tp = H * d, T * nb, 512
    S, E = a.slots, a.evals
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = torch.randn((S, 3 * D, NP), device="cuda", generator=g)
    qkv[:, :D] *= 0.25
    datt = torch.randn((E, D, NP), device="cuda", generator=g)
    qs = torch.arange(E, device="cuda", dtype=torch.int32) % S
    ks = (torch.arange(E, device="cuda", dtype=torch.int32) * 7 + 3) % S
    att = torch.empty((E, D, NP), device="cuda")
    lse = torch.empty((E, H, NP), device="cuda")
    scores = torch.empty((E, H, nb, T, Tp), device="cuda")
    dscores = torch.empty_like(scores)
    delta = torch.empty((E, H, NP), device="cuda")
    dqkv = torch.zeros((E, 3 * D, NP), device="cuda")
    base = qkv.data_ptr()
    npl = 2 if a.mode == 1 else 1
    if a.tiles:
        kvt = to_tiles(qkv[:, D:], T, nb, npl)
        k_ptr, v_ptr, kv_stride, kvf, kvp = kvt.data_ptr(), kvt.data_ptr() + 2 * D * nb * 512 * npl, 2 * D * nb * 512 * npl, 1, nb * 512 * npl
    else:
        k_ptr, v_ptr, kv_stride, kvf, kvp = base + 4 * D * NP, base + 8 * D * NP, 3 * D * NP, 0, 0
    st = CF._stream()
    seed = 12345678901

    def fwd():
        tl = a.tiles and L.csn_get_math_mode() in (1, 2)
        _lib.check(L.csn_block_attn_fwd_f32(base, k_ptr if tl else base + 4 * D * NP, v_ptr if tl else base + 8 * D * NP,
                                            3 * D * NP, kv_stride if tl else 3 * D * NP, CF._ptr(qs),
                                            CF._ptr(ks), NP, CF._ptr(att), D * NP, None if a.noscores else CF._ptr(scores), CF._ptr(lse), E, H, d, T,
                                            nb, Tp, 8.0, a.drop, seed, kvf if tl else 0, kvp if tl else 0, st), "fwd")

    def dq():
        tl = a.tiles and L.csn_get_math_mode() in (1, 2)
        if a.recompute and tl:
            pt = 1 if a.recompute == 1 else 0
            _lib.check(L.csn_block_attn_bwd_dq_recompute_f32(CF._ptr(datt), CF._ptr(att), D * NP, base, 3 * D * NP, CF._ptr(qs),
                                                             k_ptr, v_ptr, kv_stride, CF._ptr(ks), NP, CF._ptr(scores) if pt else None,
                                                             CF._ptr(dscores) if pt else None, CF._ptr(lse), CF._ptr(delta),
                                                             dqkv.data_ptr(), 3 * D * NP, None, 0, None, E, H, d, T, nb, Tp, a.drop,
                                                             seed, kvp, 0, pt, None, 0, st), "dq recompute")
            return
        _lib.check(L.csn_block_attn_bwd_dq_f32(CF._ptr(datt), CF._ptr(att), D * NP, k_ptr if tl else base + 4 * D * NP,
                                               v_ptr if tl else base + 8 * D * NP,
                                               kv_stride if tl else 3 * D * NP, CF._ptr(ks), NP, CF._ptr(scores), CF._ptr(dscores), CF._ptr(lse),
                                               CF._ptr(delta), dqkv.data_ptr(), 3 * D * NP, None, 0, None, E, H, d, T, nb,
                                               Tp, a.drop, seed, 0, 0, kvf if tl else 0, kvp if tl else 0, 1 if tl else 0, None, 0, st), "dq")

    def dkv():
        gb = dqkv.data_ptr()
        if a.recompute == 2 and a.tiles and (L.csn_attn_bwd_grouping(d, T) & 8):
            _lib.check(L.csn_block_attn_bwd_dkv_flash_f32(CF._ptr(datt), D * NP, base, 3 * D * NP, CF._ptr(qs), k_ptr, v_ptr, kv_stride,
                                                          CF._ptr(ks), kvp, 0, NP, CF._ptr(lse), CF._ptr(delta), gb + 4 * D * NP,
                                                          gb + 8 * D * NP, 3 * D * NP, None, None, 0, None, E, H, d, T, nb, Tp, a.drop,
                                                          seed, None, 0, st), "dkv flash")
            return
        _lib.check(L.csn_block_attn_bwd_dkv_f32(CF._ptr(datt), D * NP, base, 3 * D * NP, CF._ptr(qs), NP, CF._ptr(scores),
                                                CF._ptr(dscores), gb + 4 * D * NP, gb + 8 * D * NP, 3 * D * NP, None, None,
                                                0, None, E, H, d, T, nb, Tp, 0, 0, 0, 0, 1 if (a.tiles and L.csn_get_math_mode() in (1, 2)) else 0, None, 0, st), "dkv")

    flops = 4.0 * T * d * NP * E * H
    L.csn_set_math_mode(a.mode)
    for item in a.dev.split(","):
        if item:
            k, v = item.split("=")
            L.csn_dev_set(int(k), int(v))
    ref = {}
    if a.check:
        L.csn_set_math_mode(0)
        fwd(); ref["att"] = att.clone(); ref["lse"] = lse.clone()
        dq(); ref["dq"] = dqkv[:, :D].clone(); ref["p"] = scores[0].clone()
        dkv(); ref["dk"] = dqkv[:, D:2 * D].clone(); ref["dv"] = dqkv[:, 2 * D:].clone()
        L.csn_set_math_mode(a.mode)
    only = a.only.split(",")
    fwd()
    if "fwd" in only:
        t = timeit(fwd)
        print(f"mode {a.mode} fwd  E={E} drop={a.drop}: {t:7.3f} ms  {flops / t / 1e9:7.1f} TF/s algorithmic", flush=True)
    if a.check:
        print("   att err", (att - ref["att"]).abs().max().item(), "lse err", (lse - ref["lse"]).abs().max().item())
    if "dq" in only or "dkv" in only:
        s0 = scores.clone()
        def dq_fresh():
            scores.copy_(s0); dq()
        t_copy = timeit(lambda: scores.copy_(s0))
        t = timeit(dq_fresh) - t_copy
        print(f"mode {a.mode} dq   E={E} recompute={a.recompute}: {t:7.3f} ms  {flops / t / 1e9:7.1f} TF/s algorithmic", flush=True)
        if a.check:
            print("   dq err", (dqkv[:, :D] - ref["dq"]).abs().max().item(), "scale", ref["dq"].abs().max().item(),
                  "p err", (scores[0] - ref["p"]).abs().max().item())
    if "dkv" in only:
        t = timeit(dkv)
        print(f"mode {a.mode} dkv  E={E} recompute={a.recompute}: {t:7.3f} ms  {flops / t / 1e9:7.1f} TF/s algorithmic", flush=True)
        if a.check:
            print("   dk err", (dqkv[:, D:2 * D] - ref["dk"]).abs().max().item(), "scale", ref["dk"].abs().max().item(),
                  "dv err", (dqkv[:, 2 * D:] - ref["dv"]).abs().max().item(), "scale", ref["dv"].abs().max().item())
    if a.digest:
        import hashlib
        dig = lambda t: hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]
        fwd(); torch.cuda.synchronize()
        print("digest fwd: att", dig(att), "lse", dig(lse), "scores", dig(scores), flush=True)
        dqkv.zero_(); dq(); torch.cuda.synchronize()
        print("digest dq: dq", dig(dqkv[:, :D]), "delta", dig(delta), "P", dig(scores), "dS", dig(dscores), flush=True)
        dkv(); torch.cuda.synchronize()
        print("digest dkv: dk", dig(dqkv[:, D:2 * D]), "dv", dig(dqkv[:, 2 * D:]), flush=True)
    if hasattr(L, "csn_gemm_debug_read") or os.environ.get("CSN_GEMM_STAMPS"):
        import ctypes
        L.csn_gemm_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
        buf = np.zeros(65536 * 8, dtype=np.uint64)
        L.csn_gemm_debug_read(buf.ctypes.data, buf.nbytes)
        full = buf.reshape(65536, 8).astype(np.int64)
        st, inner = full[:, :4], full[:, 4:]
        d_ = np.diff(st, axis=1)
        nslab = 16
        print(f"GEMM stamps (last launch, first 65536 work-groups): prologue={d_[:,0].mean():7.0f} loop={d_[:,1].mean():7.0f} epilogue={d_[:,2].mean():7.0f}"
              f" | per slab: reads+mfma={inner[:,0].mean()/nslab:6.0f} split+lds-write={inner[:,1].mean()/nslab:6.0f} "
              f"issue loads={inner[:,3].mean()/nslab:6.0f} barrier={inner[:,2].mean()/nslab:6.0f}")
    L.csn_set_math_mode(0)


if __name__ == "__main__":
    main()
