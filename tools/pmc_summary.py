#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile.sh into the two files committed under profiles/:

    python tools/pmc_summary.py gpurun_out/prof_<tag> profiles/<tag>

  <tag>_kernel_stats.csv   the --kernel-trace --stats table (per-kernel calls / total / average duration)
  <tag>_pmc_summary.json   per-kernel means of every collected counter and the HBM traffic per launch:
                           FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (the gfx950 correction for
                           16-byte-per-lane streaming reads, MI355X guide, HBM / rocprofv3 section); WRITE_SIZE is exact.
bench.py reads `traffic` from the newest summary whose n_atoms / frames_per_launch match its own launch shape."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def norm(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    src, dst = sys.argv[1], sys.argv[2]
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        # (bench.py runs the persistent-copy probe as a child process, which gets a stats file of its own: the bench process is the one
        #  with the most kernels)
        stats.sort(key=lambda f: -sum(1 for _ in open(f)))
        shutil.copyfile(stats[0], dst + "_kernel_stats.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, set()]))
    for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = agg[norm(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1].add((f, r["Dispatch_Id"]))
    counters, traffic = {}, {}
    for k, d in sorted(agg.items()):
        counters[k] = {c: {"mean_per_launch": v[0] / len(v[1]), "launches": len(v[1])} for c, v in sorted(d.items())}
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            fetch = 2.0 * 1024.0 * d["FETCH_SIZE"][0] / len(d["FETCH_SIZE"][1])
            write = 1024.0 * d["WRITE_SIZE"][0] / len(d["WRITE_SIZE"][1])
            traffic[k] = {"fetch_bytes_per_launch_corrected_x2": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write}
    bench = {}
    for log in glob.glob(os.path.join(src, "*.log")):
        for line in open(log, errors="replace"):
            if line.startswith("{") and '"metric"' in line:
                bench = json.loads(line)
    out = {
        "command": "rocprofv3 --pmc <counter set> -- python3 bench.py %s  (one pass per counter set; tools/profile.sh)" % os.environ.get("PROFILE_ARGS", "(arguments not recorded)"),
        "n_atoms": bench.get("config", {}).get("n_atoms"),
        "frames_per_launch": bench.get("roofline", {}).get("frames_per_launch"),
        "counters": counters, "traffic": traffic,
        "notes": "FETCH_SIZE/WRITE_SIZE reported in KiB; FETCH_SIZE doubled per the gfx950 correction for 16-B-per-lane streaming reads "
                 "(guide: HBM section); WRITE_SIZE exact for 16-B-per-lane streaming stores.",
    }
    json.dump(out, open(dst + "_pmc_summary.json", "w"), indent=1)
    for k, t in traffic.items():
        print("%-40s HBM bytes/launch %.4g" % (k, t["hbm_bytes_per_launch"]))


if __name__ == "__main__":
    main()
