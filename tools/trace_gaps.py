"""Kernel durations and launch-to-launch gaps from a rocprofv3 --kernel-trace CSV.
usage: python3 tools/trace_gaps.py <dir> <kernel-substring> [last N]"""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
pat = sys.argv[2]
lastn = int(sys.argv[3]) if len(sys.argv) > 3 else 200
rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-lastn:]
st = np.array([int(r["Start_Timestamp"]) for r in rows], dtype=np.float64)
en = np.array([int(r["End_Timestamp"]) for r in rows], dtype=np.float64)
dur = (en - st) / 1e3
gap = (st[1:] - en[:-1]) / 1e3
period = (st[1:] - st[:-1]) / 1e3
print(f"{len(rows)} dispatches of {rows[0]['Kernel_Name'][:60]}: duration median {np.median(dur):.1f} us "
      f"(min {dur.min():.1f}, max {dur.max():.1f}); gap to next launch median {np.median(gap):.1f} us; "
      f"period median {np.median(period):.1f} us; VGPR {rows[0].get('VGPR_Count')} LDS {rows[0].get('LDS_Block_Size')}")
