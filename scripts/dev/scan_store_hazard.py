#!/usr/bin/env python3
"""Scan the gfx950 assembly of every kernel source for the store-data hazard found in wx_stream.hip: a 12/16-byte buffer store
with a REGISTER soffset whose data registers are written again by one of the next two instructions (hipcc separates these only
when the soffset is an immediate).  The failure measured was at distance 1 (the register written at distance 2 was never
damaged in ~3000 bad elements): prints every site with its distance; exit status 1 if there is one at distance 1."""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
csrc = os.path.join(root, "csn_amd", "csrc")
store = re.compile(r"^\s*buffer_store_dwordx[34]\s+v\[(\d+):(\d+)\],\s*(v\d+|off),\s*s\[\d+:\d+\],\s*(\S+)")
dest = re.compile(r"^\s*(v_\w+|ds_read\w*|buffer_load\w*|global_load\w*|scratch_load\w*)\s+(v\[(\d+):(\d+)\]|v(\d+))")
sites = 0
for f in sorted(os.listdir(csrc)):
    if not f.endswith(".hip"):
        continue
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", os.path.join(csrc, f), "-o", out],
                       check=True, stderr=subprocess.DEVNULL)
        lines = [l for l in open(out).read().splitlines()]
    code = [(i, l) for i, l in enumerate(lines) if l.startswith("\t") and not l.strip().startswith((";", ".")) ]
    func = "?"
    names = {i: l[:-1] for i, l in enumerate(lines) if l.endswith(":") and l.startswith("_Z")}
    n_f = 0
    for k, (i, l) in enumerate(code):
        m = store.match(l)
        if not m or not m.group(4).startswith("s"):
            continue
        lo, hi = int(m.group(1)), int(m.group(2))
        for j in range(1, 3):
            if k + j >= len(code):
                break
            nxt = code[k + j][1]
            if nxt.strip().startswith("s_nop"):
                break
            d = dest.match(nxt)
            if d:
                a, b = (int(d.group(3)), int(d.group(4))) if d.group(3) else (int(d.group(5)), int(d.group(5)))
                if a <= hi and b >= lo:
                    fn = max((x for x in names if x < i), default=None)
                    print(f"{f}: {names.get(fn, '?')[:60]} line {i}: {l.strip()}  ->  +{j}: {nxt.strip()}")
                    sites += 1 if j == 1 else 0
                    n_f += 1
                    break
    print(f"{f}: {n_f} site(s)", flush=True)
sys.exit(1 if sites else 0)
