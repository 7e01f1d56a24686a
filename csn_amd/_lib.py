"""ctypes binding of libcsn_hip.so (C ABI: include/csn_hip.h) and its in-tree build.

This is the binding a maintainer of the reference would add next to ``MID-FC/csa_models.py``: the
library knows nothing about torch; tensors are handed over as raw device pointers plus sizes and the
current HIP stream.  There is NO fallback: if the library cannot be built/loaded, every op raises.
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import subprocess
from ctypes import c_char_p, c_float, c_int, c_longlong, c_ulonglong, c_void_p
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libcsn_hip.so")
SOURCES = ["gemm_f32.hip", "gemm_bf16x3.hip", "wx_stream.hip", "wx_lnb.hip", "loss.hip", "attn_f32.hip", "attn_bf16x3.hip", "attn_dkv.hip", "outproj_ln.hip", "retrieval.hip", "combine.hip", "compat.hip", "csn_capi.hip"]
HEADERS = ["csn_common.h", "csn_kernels.h", "csn_window.h", "wx_common.h", os.path.join("..", "..", "include", "csn_hip.h")]
ARCH = "gfx950"
BUILD_FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-shared"]

_lib: Optional[ctypes.CDLL] = None


class CsnError(RuntimeError):
    pass


STAMP_PATH = LIB_PATH + ".sha256"


def _source_digest() -> str:
    """sha256 over the names and bytes of every source and header the library is built from (and the build flags)."""
    h = hashlib.sha256(" ".join(BUILD_FLAGS).encode())
    for f in SOURCES + HEADERS:
        h.update(f.encode())
        with open(os.path.join(_CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _stale() -> bool:
    """The library is current iff the digest recorded beside it at build time equals the digest of the sources now
    (content, not modification times: a snapshot copy or a checkout changes mtimes without changing a byte)."""
    if not (os.path.exists(LIB_PATH) and os.path.exists(STAMP_PATH)):
        return True
    with open(STAMP_PATH) as fh:
        return fh.read().strip() != _source_digest()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into csn_amd/libcsn_hip.so (in-tree, so it travels to the GPU box): one object
    per translation unit, compiled side by side, then one link."""
    if not force and not _stale():
        return LIB_PATH
    # several ranks may import the package at once (mp.spawn in the tests, torchrun): one of them builds, the others wait
    # on the lock and find the library current when they get it
    import fcntl
    import tempfile
    lock_path = os.path.join(_HERE, ".build.lock")                 # (git-ignored; opened for append: nothing is truncated)
    if not os.access(_HERE, os.W_OK):
        lock_path = os.path.join(tempfile.gettempdir(), "csn_amd_build_" + hashlib.sha256(_HERE.encode()).hexdigest()[:16] + ".lock")
    with open(lock_path, "a") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():
                return LIB_PATH
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose: bool) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    obj_dir = os.path.join(_HERE, "_obj")
    os.makedirs(obj_dir, exist_ok=True)
    compile_flags = [f for f in BUILD_FLAGS if f != "-shared"]
    objs, errors = [], []
    max_jobs = max(1, min(len(SOURCES), os.cpu_count() or 1))     # translation units side by side, never more than the cores
    pending = list(SOURCES)
    running = []
    while pending or running:
        while pending and len(running) < max_jobs:
            f = pending.pop(0)
            obj = os.path.join(obj_dir, f.replace(".hip", ".o"))
            cmd = [hipcc] + compile_flags + ["-c", os.path.join(_CSRC, f), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            running.append((f, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        f, obj, proc = running.pop(0)
        out, _ = proc.communicate()
        if proc.returncode != 0:
            errors.append(f"{f}:\n{out}")
        objs.append(obj)
    if errors:
        raise CsnError("hipcc failed:\n" + "\n".join(errors))
    link = [hipcc] + BUILD_FLAGS + ["-o", LIB_PATH + ".tmp"] + objs
    if verbose:
        print(" ".join(link), flush=True)
    res = subprocess.run(link, capture_output=True, text=True)
    if res.returncode != 0:
        raise CsnError("hipcc link failed:\n" + res.stdout + res.stderr)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    with open(STAMP_PATH, "w") as fh:
        fh.write(_source_digest() + "\n")
    global _lib
    _lib = None
    return LIB_PATH


_SIGNATURES = {
    "csn_version": (c_int, []),
    "csn_set_math_mode": (c_int, [c_int]),
    "csn_set_thread_math_mode": (c_int, [c_int]),
    "csn_get_math_mode": (c_int, []),
    "csn_get_thread_math_mode": (c_int, []),
    "csn_set_thread_act16": (c_int, [c_int]),
    "csn_get_thread_act16": (c_int, []),
    "csn_set_thread_score_layout": (c_int, [c_int]),
    "csn_get_thread_score_layout": (c_int, []),
    "csn_status_string": (c_char_p, [c_int]),
    "csn_dev_set": (c_int, [c_int, c_int]),
    "csn_dev_get": (c_int, [c_int]),
    "csn_wgrad_workspace_floats": (c_longlong, [c_int, c_int, c_int, c_int]),
    "csn_project_f32": (c_int, [c_void_p, c_longlong, c_int, c_void_p, c_int, c_int, c_void_p, c_longlong, c_int,
                                c_int, c_int, c_int, c_float, c_int, c_longlong, c_void_p]),
    "csn_block_attn_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_longlong, c_longlong, c_void_p, c_void_p, c_int,
                                       c_void_p, c_longlong, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                       c_int, c_float, c_float, c_ulonglong, c_int, c_longlong, c_void_p]),
    "csn_block_attn_fwd_grouped_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_longlong, c_longlong, c_void_p, c_void_p, c_int,
                                               c_void_p, c_longlong, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                               c_int, c_float, c_float, c_ulonglong, c_int, c_longlong, c_void_p, c_void_p, c_int,
                                               c_void_p]),
    "csn_block_attn_bwd_dq_f32": (c_int, [c_void_p, c_void_p, c_longlong, c_void_p, c_void_p, c_longlong, c_void_p, c_int,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_void_p, c_int,
                                          c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_ulonglong,
                                          c_int, c_longlong, c_int, c_longlong, c_int, c_void_p, c_int, c_void_p]),
    "csn_block_attn_bwd_dq_recompute_f32": (c_int, [c_void_p, c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_void_p,
                                                    c_void_p, c_longlong, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                                    c_void_p, c_void_p, c_longlong, c_void_p, c_int, c_void_p, c_int, c_int,
                                                    c_int, c_int, c_int, c_int, c_float, c_ulonglong, c_longlong, c_int, c_int,
                                                    c_void_p, c_int, c_void_p]),
    "csn_block_attn_bwd_dkv_flash_f32": (c_int, [c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_void_p, c_void_p,
                                                 c_longlong, c_void_p, c_longlong, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                                 c_void_p, c_longlong, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                                                 c_int, c_int, c_int, c_float, c_ulonglong, c_void_p, c_int, c_void_p]),
    "csn_block_attn_bwd_dkv_f32": (c_int, [c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_int, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_longlong, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                           c_int, c_int, c_int, c_int, c_int, c_int, c_longlong, c_int, c_longlong,
                                           c_int, c_void_p, c_int, c_void_p]),
    "csn_attn_bwd_grouping": (c_int, [c_int, c_int]),
    "csn_cross_attn_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_longlong, c_longlong, c_int, c_int, c_void_p, c_longlong,
                                       c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float,
                                       c_ulonglong, c_void_p]),
    "csn_cross_attn_bwd_f32": (c_int, [c_void_p, c_void_p, c_longlong, c_void_p, c_void_p, c_void_p, c_longlong, c_longlong,
                                       c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_longlong, c_longlong, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                       c_ulonglong, c_void_p]),
    "csn_varlen_attn_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_longlong, c_longlong, c_int, c_int, c_void_p, c_longlong,
                                        c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_float,
                                        c_float, c_ulonglong, c_void_p]),
    "csn_varlen_attn_bwd_f32": (c_int, [c_void_p, c_void_p, c_longlong, c_void_p, c_void_p, c_void_p, c_longlong, c_longlong,
                                        c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_longlong, c_longlong, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                        c_float, c_ulonglong, c_void_p]),
    "csn_project_qkv_f32": (c_int, [c_void_p, c_longlong, c_int, c_void_p, c_int, c_int, c_void_p, c_longlong, c_int, c_void_p,
                                    c_longlong, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "csn_outproj_ln_workspace_floats": (c_longlong, [c_int, c_int, c_int, c_int]),
    "csn_masked_ce_workspace_bytes": (c_longlong, [c_int, c_int]),
    "csn_masked_ce_fwd_f32": (c_int, [c_void_p, c_longlong, c_int, c_void_p, c_longlong, c_int, c_int, c_int, c_int, c_void_p,
                                      c_void_p, c_longlong, c_void_p, c_void_p]),
    "csn_masked_ce_bwd_f32": (c_int, [c_void_p, c_longlong, c_int, c_void_p, c_longlong, c_int, c_int, c_int, c_int, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_void_p]),
    "csn_outproj_ln_fwd_f32": (c_int, [c_void_p, c_longlong, c_void_p, c_void_p, c_longlong, c_void_p, c_void_p,
                                       c_longlong, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float,
                                       c_ulonglong, c_void_p, c_void_p, c_longlong, c_void_p]),
    "csn_outproj_ln_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_longlong, c_void_p, c_longlong, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_int,
                                       c_int, c_int, c_int, c_float, c_ulonglong, c_int, c_longlong, c_void_p, c_int,
                                       c_void_p, c_int, c_void_p]),
    "csn_project_wgrad_f32": (c_int, [c_void_p, c_longlong, c_int, c_void_p, c_longlong, c_int, c_void_p, c_int, c_int,
                                      c_int, c_int, c_float, c_int, c_void_p, c_longlong, c_void_p]),
    "csn_retrieval_measure_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                          c_longlong, c_void_p]),
    "csn_rowsum_f32": (c_int, [c_void_p, c_void_p, c_longlong, c_int, c_longlong, c_void_p]),
    "csn_mix_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                c_void_p]),
    "csn_mix_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                c_int, c_void_p, c_void_p, c_void_p]),
    "csn_compat_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                   c_int, c_int, c_void_p]),
    "csn_compat_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
}

EXPORTS = tuple(_SIGNATURES)
# keys of csn_dev_set / csn_dev_get (include/csn_hip.h, development section)
DEV_BIG_TILES, DEV_WIDE_GEMM, DEV_WIDE_FORMS, DEV_WX, DEV_LNB_GROUP = 0, 1, 2, 3, 5
DEV_WX_DEFAULT = 9                     # streaming kernel (1) + LayerNorm backward fused into the dCtx stream (8)


def lib() -> ctypes.CDLL:
    """The loaded library; raises CsnError if it is missing (build it with csn_amd.build())."""
    global _lib
    if _lib is None:
        path = os.environ.get("CSN_LIB_PATH", LIB_PATH)      # development aid: compare two builds side by side
        if not os.path.exists(path):
            raise CsnError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU or eager fallback for the CSA kernels)")
        handle = ctypes.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.csn_version() != 17:
            raise CsnError("libcsn_hip.so ABI version mismatch")
        _lib = _Hooked(handle)
    return _lib


# Every C-ABI call can be bracketed by a caller-supplied hook (bench.py: HIP events on the launch stream around EVERY entry
# point of the step, so that the roofline names the longest launch whatever it is).  None (the default) costs one comparison.
_call_hook = None


def set_call_hook(fn) -> None:
    """fn(name, phase) with phase "begin" / "end" around every call that takes a stream (i.e. launches kernels); None removes it."""
    global _call_hook
    _call_hook = fn


class _Hooked:
    """The loaded library with its launching entry points wrapped for the call hook."""

    def __init__(self, handle):
        self._handle = handle
        for name, (_, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            launches = bool(args) and args[-1] is c_void_p and name not in ("csn_status_string",)
            setattr(self, name, self._wrap(name, fn) if launches else fn)

    @staticmethod
    def _wrap(name, fn):
        def call(*a):
            hook = _call_hook
            if hook is None:
                return fn(*a)
            hook(name, "begin")
            try:
                return fn(*a)
            finally:
                hook(name, "end")
        return call


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = lib().csn_status_string(status).decode()
        raise CsnError(f"{what or 'csn call'} failed with status {status}: {msg}")
