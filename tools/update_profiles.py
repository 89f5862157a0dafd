#!/usr/bin/env python3
"""Copy the judged summaries of a tools/collect_evidence.sh run from gpurun_out/evidence into
profiles/ (tracked) and refresh profiles/traffic_latest.json, which bench.py reports as
roofline.traffic.   usage: tools/update_profiles.py r02"""
import csv, glob, json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ev, prof, tag = os.path.join(root, "gpurun_out", "evidence"), os.path.join(root, "profiles"), sys.argv[1]
HEAD = "ap_mel2048"


def find(d, pat):
    m = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return m[0] if m else None


def copy_stats(src_dir, dst):
    src = find(src_dir, "*kernel_stats.csv")
    with open(src) as f, open(dst, "w", newline="") as g:
        w = csv.writer(g)
        for row in csv.reader(f):
            row[0] = row[0][:100]                   # torch's templated kernel names run to kilobytes
            w.writerow(row)


def pmc_rows(src_dir, counter, keep=lambda name: name.startswith("ap_") or "ap_" in name[:40]):
    src = find(src_dir, "*counter_collection.csv")
    rows = [r for r in csv.DictReader(open(src)) if r["Counter_Name"] == counter and keep(r["Kernel_Name"])]
    for r in rows:
        r["Kernel_Name"] = r["Kernel_Name"][:100]
    return rows


def write_rows(rows, dst):
    if not rows:
        return
    with open(dst, "w", newline="") as g:
        w = csv.DictWriter(g, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)


shutil.copy(os.path.join(ev, "bench.json"), os.path.join(prof, f"{tag}_bench_headline.json"))
# every BASELINE config + the reference's published rows: the `configs` object of the same bench line
_b = json.load(open(os.path.join(ev, "bench.json")))
json.dump(_b.get("configs", {}), open(os.path.join(prof, f"{tag}_configs_1gpu.json"), "w"), indent=1)
copy_stats(os.path.join(ev, "kt"), os.path.join(prof, f"{tag}_rocprofv3_kernel_stats_bench.csv"))
tot = {}
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    rows = [r for r in pmc_rows(os.path.join(ev, sub), name) if HEAD in r["Kernel_Name"]]
    write_rows(rows, os.path.join(prof, f"{tag}_pmc_{name}.csv"))
    v = [float(r["Counter_Value"]) for r in rows]
    tot[name] = sum(v) / len(v)
fetch = tot["FETCH_SIZE"] * 1024 * 2               # KB, and the gfx950 x2 correction (MI355X_MICROARCH.md)
write = tot["WRITE_SIZE"] * 1024
bench = json.load(open(os.path.join(ev, "bench.json")))
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
src = (f"profiles/{tag}_pmc_FETCH_SIZE.csv + profiles/{tag}_pmc_WRITE_SIZE.csv: separate rocprofv3 --pmc passes of "
       "`python3 bench.py --steps 5 --warmup 1`, FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE, "
       "per launch of the headline kernel; collected by tools/collect_evidence.sh, not in this run")
json.dump({"headline": fetch + write, "_source": {"headline": src},
           "_detail": {"FETCH_SIZE_KB_raw": tot["FETCH_SIZE"], "fetch_bytes_x2": fetch,
                       "WRITE_SIZE_KB": tot["WRITE_SIZE"], "write_bytes": write, "algorithmic_bytes": alg}},
          open(os.path.join(prof, "traffic_latest.json"), "w"), indent=1)
print("headline traffic", fetch + write, "=", (fetch + write) / alg, "x algorithmic")
# per-operator kernel stats and HBM traffic
summary = {}
for d in sorted(glob.glob(os.path.join(ev, "op_*"))):
    if not os.path.isdir(d):
        continue
    op = os.path.basename(d)[3:]
    copy_stats(os.path.join(d, "kt"), os.path.join(prof, f"{tag}_op_{op}_kernel_stats.csv"))
    allrows = []
    per_kernel = {}
    for name, sub in (("FETCH_SIZE", "f"), ("WRITE_SIZE", "w")):
        rows = pmc_rows(os.path.join(d, sub), name)
        allrows += rows
        for r in rows:
            per_kernel.setdefault(r["Kernel_Name"][:60], {}).setdefault(name, []).append(float(r["Counter_Value"]))
    write_rows(allrows, os.path.join(prof, f"{tag}_op_{op}_pmc_FETCH_WRITE.csv"))
    summary[op] = {k: {"fetch_MB_x2": sum(v.get("FETCH_SIZE", [0])) / max(len(v.get("FETCH_SIZE", [1])), 1) * 2 * 1024 / 1e6,
                       "write_MB": sum(v.get("WRITE_SIZE", [0])) / max(len(v.get("WRITE_SIZE", [1])), 1) * 1024 / 1e6}
                   for k, v in per_kernel.items()}
json.dump(summary, open(os.path.join(prof, f"{tag}_op_traffic_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
