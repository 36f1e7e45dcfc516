#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (csv) into small files under profiles/.

    python3 tools/summarize_prof.py <tag> <stats_dir> [<pmc_dir> ...]

Writes profiles/<tag>_kernel_stats.csv (copy of rocprofv3's *_kernel_stats.csv)
and profiles/<tag>_pmc.json: per kernel and counter, the mean over dispatches;
when the fused search kernel has both HBM counters, also profiles/traffic.json.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE
counts 64 B per 128-B request for wide coalesced reads, so the derived
`hbm_read_bytes` doubles it (MI355X_MICROARCH.md, HBM section).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def main():
    tag, stats_dir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    os.makedirs(root, exist_ok=True)
    for f in glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(root, f"{tag}_kernel_stats.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in pmc_dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, cs in agg.items():
        o = {c: {"mean": sum(v) / len(v), "dispatches": len(v)} for c, v in cs.items()}
        if "FETCH_SIZE" in o:
            o["hbm_read_bytes"] = o["FETCH_SIZE"]["mean"] * 1024 * 2
        if "WRITE_SIZE" in o:
            o["hbm_write_bytes"] = o["WRITE_SIZE"]["mean"] * 1024
        if "hbm_read_bytes" in o and "hbm_write_bytes" in o:
            o["hbm_bytes_per_launch"] = o["hbm_read_bytes"] + o["hbm_write_bytes"]
        out[k] = o
    with open(os.path.join(root, f"{tag}_pmc.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k, o in out.items():
        if "hbm_bytes_per_launch" in o:
            print(k, "hbm bytes/launch", o["hbm_bytes_per_launch"])
            if "k_simulate_fused" in k:
                # the figure bench.py reports as roofline.traffic: the headline run's at the top
                # level, a secondary workload's (TRAFFIC_KEY = its bench.py name) under its name
                path = os.path.join(root, "traffic.json")
                try:
                    cur = json.load(open(path))
                except Exception:
                    cur = {}
                rec = {"kernel": k, "source": f"{tag}_pmc.json",
                       "hbm_read_bytes": o["hbm_read_bytes"], "hbm_write_bytes": o["hbm_write_bytes"],
                       "hbm_bytes_per_launch": o["hbm_bytes_per_launch"]}
                key = os.environ.get("TRAFFIC_KEY")
                if key:
                    cur[key] = rec
                else:
                    cur.update(rec)
                with open(path, "w") as f:
                    json.dump(cur, f, indent=1)


if __name__ == "__main__":
    main()
