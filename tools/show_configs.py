"""Print a tools/bench_configs.py report as a table.  usage: python tools/show_configs.py report.json"""
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if isinstance(v, dict):
        print(f"{k:28s} {v['ms']:9.4f} ms  {v['units_per_s']:.3e} {v.get('unit', '')}/s  "
              f"alg {v.get('alg_GBps', 0):7.0f} GB/s  frac_hbm {v.get('frac_hbm', 0):.3f}"
              + (f"  round-trip err {v['round_trip_max_err']:.1e}" if 'round_trip_max_err' in v else ""))
