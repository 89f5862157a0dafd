#!/usr/bin/env python3
"""Average per-launch counter values of the headline kernel from rocprofv3 --pmc CSV output."""
import collections, csv, glob, json, sys
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "mel2048"
frames = float(sys.argv[3]) if len(sys.argv) > 3 else 256 * 431
d = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: sum(v) / len(v) for k, v in sorted(d.items())}
out_pf = {k + "_per_frame": v / frames for k, v in out.items()}
print(json.dumps({"per_launch": out, "per_frame": out_pf}, indent=1))
