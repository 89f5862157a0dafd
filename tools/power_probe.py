"""Board power and clock while every CU runs ONE instruction kind on random operands
(build/ubench_valu power ...), sampled through rocm-smi: what a wave-instruction of each kind costs
in energy.  usage: python tools/power_probe.py"""
import os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "ubench_valu")
CASES = [("valu", 1, 2), ("valu", 2, 2), ("valu", 0, 2), ("valu", 6, 2), ("valu", 14, 2), ("valu", 4, 2), ("valu", 5, 2),
         ("valu", 1, 1), ("valu", 1, 3), ("lds", 0, 2), ("lds", 1, 2), ("lds", 2, 2), ("lds", 3, 2), ("lds", 4, 2)]

def power():
    out = subprocess.run(["rocm-smi", "--showpower"], capture_output=True, text=True, timeout=10).stdout
    m = re.search(r"Power \(W\):\s*([\d.]+)", out)
    return float(m.group(1)) if m else None

print("idle power (W):", [power() for _ in range(3)])
for what, kind, wps in CASES:
    samples = []
    p = subprocess.Popen([EXE, "power", what, str(kind), str(wps), "3.0"], stdout=subprocess.PIPE, text=True)
    t0 = time.time()
    while p.poll() is None:
        w = power()
        if w is not None and time.time() - t0 > 1.0:
            samples.append(w)
    line = p.stdout.read().strip()
    samples.sort()
    med = samples[len(samples) // 2] if samples else float("nan")
    m = re.search(r"= ([\d.e+]+) per second", line)
    rate = float(m.group(1)) if m else float("nan")
    print(f"{line} | power median {med:.0f} W (n={len(samples)})", flush=True)
