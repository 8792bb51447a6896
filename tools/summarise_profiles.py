#!/usr/bin/env python3
"""summarise_profiles.py <dir> <tag>: condense the rocprofv3 output of tools/profile_round.sh into the two small
CSVs kept under profiles/: <tag>_kernel_stats.csv (copy of the --stats table) and <tag>_pmc_summary.csv
(per kernel and counter: dispatches, mean value per dispatch, mean kernel duration inside that pass).  The first
dispatch of every kernel (warm-up step) is dropped from the PMC means."""
import collections
import csv
import glob
import os
import shutil
import sys


def short(name):
    return name.split("(")[0].strip()


def main(d, tag, bench_args=()):
    ks = glob.glob(os.path.join(d, "kt", "**", "*_kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copyfile(ks[0], os.path.join(d, f"{tag}_kernel_stats.csv"))
    rows = []
    for name in ("fetch", "write", "sq", "grbm"):
        cc = glob.glob(os.path.join(d, name, "**", "*_counter_collection.csv"), recursive=True)
        if not cc:
            continue
        per = collections.defaultdict(lambda: collections.defaultdict(dict))  # kernel -> counter -> dispatch -> (val, dur)
        with open(cc[0]) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                if "sfa::" not in k:
                    continue
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                slot = per[k][r["Counter_Name"]]
                did = int(r["Dispatch_Id"])
                v, _ = slot.get(did, (0.0, dur))
                slot[did] = (v + float(r["Counter_Value"]), dur)  # rows of one dispatch (per XCD / dimension) add up
        for k in sorted(per):
            for c in sorted(per[k]):
                ds = sorted(per[k][c])
                keep = ds[len(ds) // 4:] if len(ds) >= 4 else ds  # drop the warm-up step's dispatches
                vals = [per[k][c][i][0] for i in keep]
                durs = [per[k][c][i][1] for i in keep]
                rows.append((k, c, len(keep), sum(vals) / len(vals), sum(durs) / len(durs)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import sigfish_amd
    # bench.py quotes these counters only for this build, workload, batch size and options
    import argparse
    import json
    import re
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ncov_r9_dna_q250")
    ap.add_argument("--reads", type=int, default=None)
    ap.add_argument("--opt", action="append", default=[])
    a, _ = ap.parse_known_args(list(bench_args))
    qlen = int(re.search(r"_q(\d+)$", a.workload).group(1))
    reads = a.reads if a.reads is not None else 100_000 * 250 // qlen
    with open(os.path.join(d, f"{tag}_meta.json"), "w") as f:
        json.dump({"build_id": sigfish_amd.build_id(), "workload": a.workload, "reads": reads, "opts": a.opt,
                   "command": "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e " + " ".join(bench_args)}, f)
        f.write("\n")
    with open(os.path.join(d, f"{tag}_pmc_summary.csv"), "w") as f:
        f.write("kernel,counter,dispatches,mean_value_per_dispatch,mean_duration_ms_in_that_pass\n")
        for k, c, n, v, t in rows:
            f.write(f'"{k}",{c},{n},{v:.1f},{t:.3f}\n')
    print(open(os.path.join(d, f"{tag}_pmc_summary.csv")).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
