#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean counter value per dispatch.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of 1 kB; MI355X_MICROARCH.md (HBM): on gfx950
FETCH_SIZE reads 1/2 of the bytes of wide coalesced reads -> both the raw and the x2-corrected value are printed."""
import csv
import glob
import os
import sys
from collections import defaultdict

import json

root = sys.argv[1]
summary = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, counter, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: [0, 0.0])
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"].split("(")[0][:60]
                acc[name][0] += 1
                acc[name][1] += float(row["Counter_Value"])
    for name, (n, tot) in sorted(acc.items()):
        mean_kb = tot / max(n, 1)
        corr = 2.0 if counter == "FETCH_SIZE" else 1.0
        key = ("k_tower_h1_f16" if "k_tower_h1<azh1::F16>" in name else
               next((k for k in ("k_expand_select", "k_select_gather", "k_expand_backup", "k_tower_f32w", "k_tower_f32_t16", "k_tower_f32", "k_tower_h1",
                                 "k_tower_h2", "k_tower", "k_heads_f32", "k_heads") if k in name), name))
        summary.setdefault(key, {})[counter] = {"dispatches": n, "raw_bytes_per_launch": mean_kb * 1e3,
                                                "corrected_bytes_per_launch": mean_kb * corr * 1e3, "correction": corr}
        print(f"{counter:10s} {name:60s} dispatches={n:6d} mean={mean_kb:12.1f} kB/launch  "
              f"-> {mean_kb * 1e3 / 1e6:9.2f} MB raw, {mean_kb * corr * 1e3 / 1e6:9.2f} MB corrected(x{corr:g})")

with open(os.path.join(root, "pmc_traffic.json"), "w") as fh:
    json.dump(summary, fh, indent=1)
