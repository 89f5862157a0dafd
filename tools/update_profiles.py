#!/usr/bin/env python3
"""Copy the judged summaries of a tools/collect_evidence.sh run from gpurun_out/evidence into
profiles/ (tracked) and refresh profiles/traffic_latest.json, which bench.py reports as
roofline.traffic.   usage: tools/update_profiles.py r01"""
import csv, json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ev, prof, tag = os.path.join(root, "gpurun_out", "evidence"), os.path.join(root, "profiles"), sys.argv[1]
shutil.copy(os.path.join(ev, "bench.json"), os.path.join(prof, f"{tag}_bench_headline.json"))
shutil.copy(os.path.join(ev, "configs.json"), os.path.join(prof, f"{tag}_configs_1gpu.json"))
with open(os.path.join(ev, "kt", "kt_kernel_stats.csv")) as f, \
        open(os.path.join(prof, f"{tag}_rocprofv3_kernel_stats_bench.csv"), "w", newline="") as g:
    w = csv.writer(g)
    for row in csv.reader(f):
        row[0] = row[0][:100]                       # torch's templated kernel names run to kilobytes
        w.writerow(row)
tot = {}
for name, sub, pre in (("FETCH_SIZE", "pmc_fetch", "f"), ("WRITE_SIZE", "pmc_write", "w")):
    src = os.path.join(ev, sub, f"{pre}_counter_collection.csv")
    rows = [r for r in csv.DictReader(open(src)) if "ap_mel2048" in r["Kernel_Name"]]
    with open(os.path.join(prof, f"{tag}_pmc_{name}.csv"), "w", newline="") as g:
        w = csv.DictWriter(g, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    v = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == name]
    tot[name] = sum(v) / len(v)
fetch = tot["FETCH_SIZE"] * 1024 * 2               # KB, and the gfx950 x2 correction (MI355X_MICROARCH.md)
write = tot["WRITE_SIZE"] * 1024
alg = json.load(open(os.path.join(ev, "bench.json")))["roofline"]["algorithmic_bytes_per_launch"]
json.dump({"headline": fetch + write,
           "_detail": {"FETCH_SIZE_KB_raw": tot["FETCH_SIZE"], "fetch_bytes_x2": fetch,
                       "WRITE_SIZE_KB": tot["WRITE_SIZE"], "write_bytes": write, "algorithmic_bytes": alg}},
          open(os.path.join(prof, "traffic_latest.json"), "w"), indent=1)
print("traffic", fetch + write, "=", (fetch + write) / alg, "x algorithmic")
