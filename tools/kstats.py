#!/usr/bin/env python3
"""Per-kernel register / spill / static instruction statistics from `hipcc -S` output.
usage: tools/kstats.py [substring ...]   (compiles csrc/audioprims.hip for gfx950)"""
import collections, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "mlx-audio-primitives_amd", "csrc", "audioprims.hip")
out = "/tmp/ap_kstats.s"
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                "-o", out, src], check=True, stderr=subprocess.DEVNULL)
text = open(out).read()
meta = {}
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, re.S):
    body = m.group(2)
    g = lambda k: int(re.search(k + r":\s+(\d+)", body).group(1))
    meta[m.group(1)] = (g(r"\.vgpr_count"), g(r"\.vgpr_spill_count"), g(r"\.sgpr_count"),
                        g(r"\.private_segment_fixed_size"))
pats = sys.argv[1:] or ["wave_kernel"]
for name, (vg, sp, sg, scr) in meta.items():
    if not any(p in name for p in pats):
        continue
    body = text[text.index("\n" + name + ":"):]
    body = body[:body.index("s_endpgm")]
    c = collections.Counter()
    for l in body.splitlines():
        l = l.strip()
        if not l or l.startswith((".", ";")) or l.endswith(":"):
            continue
        c[l.split()[0]] += 1
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    lds = sum(v for k, v in c.items() if k.startswith("ds_"))
    print(f"{name}\n   vgpr {vg} spills {sp} scratch {scr} B sgpr {sg} | static: VALU {valu} LDS {lds} "
          f"s_nop {c['s_nop']} s_waitcnt {c['s_waitcnt']} dpp {c['v_mov_b32_dpp']} v_mov {c['v_mov_b32_e32'] + c['v_mov_b64_e32']}")
