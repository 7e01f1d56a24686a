#!/usr/bin/env python3
"""Scan the gfx950 assembly of the shipped kernel sources for the store-data hazard found in wx_stream.hip (DESIGN, platform
findings): a 12 / 16-byte buffer store with a REGISTER soffset whose data registers are written again within the next two
instructions.  The hardware needs two wait states there; hipcc pads them only when the soffset is an immediate.  The failure
measured was at distance 1; distance 2 leaves ONE wait state where the rule for the immediate form asks for two, so both count.

    python3 scripts/dev/scan_store_hazard.py            # every source in csn_amd._lib.SOURCES; exit status 1 on any site

tests/test_cpu_store_hazard.py runs `scan()` over the same list on every CPU test run (hipcc -S needs no GPU)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "csn_amd", "csrc")
HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
STORE = re.compile(r"^\s*buffer_store_dwordx[34]\s+v\[(\d+):(\d+)\],\s*(v\d+|v\[\d+:\d+\]|off),\s*s\[\d+:\d+\],\s*(\S+)")
# an instruction that WRITES vector registers: anything whose first operand is a VGPR (range), except the forms whose first operand
# is read — stores, LDS writes, exports (returning atomics, ds_bpermute / ds_swizzle, tbuffer loads are all caught by the rule)
NOT_A_DEST = ("buffer_store", "tbuffer_store", "global_store", "flat_store", "scratch_store", "ds_write", "ds_store", "ds_gws",
              "exp", "s_", "v_nop", "v_cmp_", "v_cmpx_", "v_readlane", "v_readfirstlane")
DEST = re.compile(r"^\s*(\w+)\s+(v\[(\d+):(\d+)\]|v(\d+)\b)")
BRANCH = re.compile(r"^\s*(s_branch|s_cbranch_\w+)\s+(\S+)")
NOP = re.compile(r"^\s*s_nop\s+(\d+)")


def have_hipcc():
    return os.path.exists(HIPCC) or shutil.which(HIPCC) is not None


def assembly(source, extra_flags=()):
    """gfx950 device assembly of one source, built with the library's own compile flags."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from csn_amd import _lib
    flags = [f for f in _lib.BUILD_FLAGS if f not in ("-shared", "-fPIC")]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        res = subprocess.run([HIPCC] + flags + list(extra_flags) + ["-S", "--cuda-device-only", os.path.join(CSRC, source), "-o", out],
                             capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"{HIPCC} -S {source} failed ({res.returncode}):\n{res.stderr[-4000:]}")
        with open(out) as fh:
            return fh.read().splitlines()


def _writes(instr, lo, hi):
    d = DEST.match(instr)
    if not d or d.group(1).startswith(NOT_A_DEST):
        return False
    a, b = (int(d.group(3)), int(d.group(4))) if d.group(3) else (int(d.group(5)), int(d.group(5)))
    return a <= hi and b >= lo


def sites_in(lines, max_distance=2):
    """[(line number, store, distance, overwriting instruction, kernel)] for every 12 / 16-byte buffer store with a register
    soffset whose data registers are written again before `max_distance` wait states have passed.  Wait states: every
    instruction issued behind the store is one, `s_nop N` is N + 1 (so `s_nop 1` behind the store closes the window, `s_nop 0`
    does not).  The walk follows control flow: behind an `s_branch` it continues at the target, behind an `s_cbranch_*` on both
    sides — a store at a loop tail is checked against the loop head."""
    code = [(i, l) for i, l in enumerate(lines) if l.startswith("\t") and not l.strip().startswith((";", "."))]
    names = {i: l[:-1] for i, l in enumerate(lines) if l.endswith(":") and l.startswith("_Z")}
    line_to_code, k = {}, 0
    for i in range(len(lines)):                       # label line -> index of the first instruction behind it
        while k < len(code) and code[k][0] < i:
            k += 1
        line_to_code[i] = k
    labels = {l[:-1].strip(): line_to_code[i] for i, l in enumerate(lines) if l.endswith(":") and not l.startswith("\t")}
    found = []

    def walk(k, waited, lo, hi, budget):
        """first hazard from instruction index k on with `waited` wait states already behind the store, or None"""
        while k < len(code) and waited < max_distance and budget > 0:
            nxt = code[k][1]
            budget -= 1
            n = NOP.match(nxt)
            if n:
                waited += int(n.group(1)) + 1
                k += 1
                continue
            if _writes(nxt, lo, hi):
                return waited + 1, nxt.strip()
            waited += 1
            br = BRANCH.match(nxt)
            if br and br.group(2) in labels:
                hit = walk(labels[br.group(2)], waited, lo, hi, budget)
                if hit or br.group(1) == "s_branch":
                    return hit
            k += 1
        return None

    for k, (i, l) in enumerate(code):
        m = STORE.match(l)
        if not m or not m.group(4).startswith("s"):
            continue
        hit = walk(k + 1, 0, int(m.group(1)), int(m.group(2)), 8)
        if hit:
            fn = max((x for x in names if x < i), default=None)
            found.append((i, l.strip(), hit[0], hit[1], names.get(fn, "?")))
    return found


def scan(sources=None, max_distance=2, jobs=None):
    """{source: sites} over the shipped sources (csn_amd._lib.SOURCES), compiled side by side."""
    if sources is None:
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        from csn_amd import _lib
        sources = list(_lib.SOURCES)
    jobs = jobs or max(1, min(len(sources), os.cpu_count() or 1))
    with ThreadPoolExecutor(jobs) as pool:
        asm = list(pool.map(assembly, sources))
    return {f: sites_in(a, max_distance) for f, a in zip(sources, asm)}


if __name__ == "__main__":
    total = 0
    for f, sites in scan().items():
        for i, st, j, nxt, fn in sites:
            print(f"{f}: {fn[:60]} line {i}: {st}  ->  +{j}: {nxt}")
        print(f"{f}: {len(sites)} site(s)", flush=True)
        total += len(sites)
    sys.exit(1 if total else 0)
