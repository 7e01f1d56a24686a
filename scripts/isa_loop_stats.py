#!/usr/bin/env python3
"""development aid: instruction mix of one kernel of a `hipcc -S --cuda-device-only` listing, whole kernel and per loop
(a loop = the span between a label and the last backward branch to it).   isa_loop_stats.py file.s kernel-substring"""
import re, sys
from collections import Counter


def classes(ins):
    c = Counter()
    for i in ins:
        op = i.split()[0]
        for key in ("mfma", "ds_read", "ds_write", "buffer_load", "buffer_store", "scratch", "accvgpr", "s_waitcnt", "s_barrier",
                    "v_exp", "v_cvt", "v_perm", "v_mov"):
            if key in op:
                c[key] += 1
        c["all"] += 1
    return dict(c)


def main():
    src, key = sys.argv[1], sys.argv[2]
    lines = open(src).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start + 1:end]
    labels, ins = {}, []
    for l in body:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", t):
                labels[t.split(":")[0]] = len(ins)
            continue
        ins.append(t.split(";")[0].strip())
    print(lines[start].split(":")[0])
    print("  whole kernel:", classes(ins))
    for n, i in enumerate(ins):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", i) or re.match(r"s_branch\s+(\.LBB\d+_\d+)", i)
        if m and m.group(1) in labels and labels[m.group(1)] < n and n - labels[m.group(1)] > 200:
            print(f"  loop {m.group(1)} [{labels[m.group(1)]}..{n}]:", classes(ins[labels[m.group(1)]:n + 1]))


if __name__ == "__main__":
    main()
