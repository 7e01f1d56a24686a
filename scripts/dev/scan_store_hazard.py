#!/usr/bin/env python3
"""Scan the gfx950 assembly of the shipped kernel sources for the store-data hazard found in wx_stream.hip (DESIGN, platform
findings): a 12 / 16-byte buffer store with a REGISTER soffset whose data registers are written again within the next two
instructions.  The hardware needs two wait states there; hipcc pads them only when the soffset is an immediate.  The failure
measured was at distance 1; distance 2 leaves ONE wait state where the rule for the immediate form asks for two, so both count.

    python3 scripts/dev/scan_store_hazard.py            # every source in csn_amd._lib.SOURCES; exit status 1 on any site

tests/test_cpu_store_hazard.py runs `scan()` over the same list on every CPU test run (hipcc -S needs no GPU)."""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "csn_amd", "csrc")
STORE = re.compile(r"^\s*buffer_store_dwordx[34]\s+v\[(\d+):(\d+)\],\s*(v\d+|off),\s*s\[\d+:\d+\],\s*(\S+)")
DEST = re.compile(r"^\s*(v_\w+|ds_read\w*|buffer_load\w*|global_load\w*|scratch_load\w*)\s+(v\[(\d+):(\d+)\]|v(\d+))")


def assembly(source, extra_flags=()):
    """gfx950 device assembly of one source, built with the library's own compile flags."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from csn_amd import _lib
    flags = [f for f in _lib.BUILD_FLAGS if f not in ("-shared", "-fPIC")]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + flags + list(extra_flags) + ["-S", "--cuda-device-only", os.path.join(CSRC, source), "-o", out],
                       check=True, stderr=subprocess.DEVNULL)
        with open(out) as fh:
            return fh.read().splitlines()


def sites_in(lines, max_distance=2):
    """[(line number, store, distance, overwriting instruction, kernel)] for every 12 / 16-byte buffer store with a register
    soffset whose data registers are a destination of one of the next `max_distance` instructions (an s_nop ends the window:
    `s_nop 1` behind the store is the two wait states)."""
    code = [(i, l) for i, l in enumerate(lines) if l.startswith("\t") and not l.strip().startswith((";", "."))]
    names = {i: l[:-1] for i, l in enumerate(lines) if l.endswith(":") and l.startswith("_Z")}
    found = []
    for k, (i, l) in enumerate(code):
        m = STORE.match(l)
        if not m or not m.group(4).startswith("s"):
            continue
        lo, hi = int(m.group(1)), int(m.group(2))
        for j in range(1, max_distance + 1):
            if k + j >= len(code):
                break
            nxt = code[k + j][1]
            if nxt.strip().startswith("s_nop"):
                break
            d = DEST.match(nxt)
            if d:
                a, b = (int(d.group(3)), int(d.group(4))) if d.group(3) else (int(d.group(5)), int(d.group(5)))
                if a <= hi and b >= lo:
                    fn = max((x for x in names if x < i), default=None)
                    found.append((i, l.strip(), j, nxt.strip(), names.get(fn, "?")))
                    break
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
