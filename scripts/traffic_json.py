#!/usr/bin/env python3
"""Turn the HBM counter summary of scripts/run_profile.sh (hbm.txt: mean FETCH_SIZE / WRITE_SIZE per launch, KB) into
profiles/attn_hbm_traffic.json, which bench.py reads for `roofline.traffic`:
    python scripts/traffic_json.py <hbm.txt> <config> <math> [<source label> [<kernel stats csv> [<its label>]]]
With the kernel-trace statistics of the same command (stats.csv of run_profile.sh) every entry also carries the kernel's NAME as
rocprofv3 prints it and its average duration there — bench.py takes `roofline.kernel` from it (and says so when the launch it
timed is not that template any more).
gfx950: HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md §HBM: FETCH_SIZE counts half of a 16-byte-per-lane
read stream)."""
import json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    path, config, math = sys.argv[1], sys.argv[2], sys.argv[3]
    label = sys.argv[4] if len(sys.argv) > 4 else os.path.relpath(path, ROOT)
    vals = {}
    for line in open(path):
        m = re.match(r"(csn_attn_\w+)<([^>]*?)(?:>\S*)?\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*\d+\s+mean=\s*([0-9.]+)", line)
        if m:
            args = [a.strip() for a in m.group(2).split(",")]          # attention: <mode, DT, BWD, KVP[, RC]> (names may be cut short)
            which = "dkv" if m.group(1) == "csn_attn_dkv_kernel" else ("bwd" if len(args) > 2 and args[2] == "true" else "fwd")
            vals.setdefault(which, {})[m.group(3)] = float(m.group(4))
    names = {}
    if len(sys.argv) > 5:
        import csv
        stats_label = sys.argv[6] if len(sys.argv) > 6 else os.path.relpath(sys.argv[5], ROOT)
        for row in csv.DictReader(open(sys.argv[5])):
            nm = row["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
            m = re.match(r"(csn_attn_\w+)<([^>]*)>", nm)
            if not m:
                continue
            args = [a.strip() for a in m.group(2).split(",")]
            which = "dkv" if m.group(1) == "csn_attn_dkv_kernel" else ("bwd" if len(args) > 2 and args[2] == "true" else "fwd")
            if which not in names or float(row["TotalDurationNs"]) > names[which][2]:
                names[which] = (nm, float(row["AverageNs"]) * 1e-6, float(row["TotalDurationNs"]), stats_label)
    out_path = os.path.join(ROOT, "profiles", "attn_hbm_traffic.json")
    tab = json.load(open(out_path)) if os.path.exists(out_path) else {}
    for which, v in vals.items():
        tab[f"{config}/{math}/{which}"] = {"bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024,
                                           "fetch_size_kb": v["FETCH_SIZE"], "write_size_kb": v["WRITE_SIZE"], "source": label}
        if which in names:
            tab[f"{config}/{math}/{which}"].update({"kernel": names[which][0], "kernel_stats_avg_ms": names[which][1],
                                                    "kernel_stats_source": names[which][3]})
    json.dump(tab, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(tab, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
