#!/usr/bin/env python3
"""Per-kernel average of one rocprofv3 --pmc counter (csv output) -> CSV on stdout.
usage: tools/pmc_summary.py gpurun_out/pmc_x/runc/*_counter_collection.csv FETCH_SIZE > profiles/rNN_pmc_fetch_size.csv
FETCH_SIZE is reported by rocprofv3 in KiB-ish units of 1 KB per count / 64 B requests; on gfx950 it under-reports wide
coalesced reads by 2x (MI355X_MICROARCH.md, HBM section) - the corrected column doubles it.  WRITE_SIZE needs no such factor
(profiles/README.md, calibration): its last column is the plain byte count."""
import collections
import csv
import sys

rows = csv.DictReader(open(sys.argv[1]))
name = sys.argv[2] if len(sys.argv) > 2 else "FETCH_SIZE"
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if r["Counter_Name"] != name:
        continue
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][0] += 1
    agg[k][1] += float(r["Counter_Value"])
fac = 2 if name == "FETCH_SIZE" else 1
print(f"kernel,launches,avg_{name}_kb," + ("avg_bytes_corrected_x2" if fac == 2 else "avg_bytes"))
for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k},{n},{v / n:.2f},{v / n * 1024 * fac:.0f}")
