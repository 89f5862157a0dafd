"""VGPR liveness of one kernel from hipcc's assembly: where is the register pressure?

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -gline-tables-only -S --cuda-device-only -o k.s file.hip
    python tools/vgpr_live.py k.s <kernel-name-substring> [top]

CFG liveness over the kernel's basic blocks (defs / uses parsed from the operand lists), then the
`top` instructions with the most live VGPRs, each with the source line of its `.loc`."""
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
lines = open(path).read().split("\n")
files = {}
start = None
for i, l in enumerate(lines):
    m = re.match(r"\s*\.file\s+(\d+)\s+(?:\"([^\"]*)\"\s+)?\"([^\"]*)\"", l)
    if m:
        files[int(m.group(1))] = m.group(3)
    if start is None and re.match(r"^[A-Za-z_][\w$.]*:", l) and pat in l and "Lfunc" not in l:
        start = i
assert start is not None, "kernel not found"
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])

VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
ins = []  # (text, loc, defs, uses, label_before, branch_target, kind)
loc = ""
labels = {}
for i in range(start + 1, end + 1):
    l = lines[i].split(";")[0].strip()
    if not l:
        continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        loc = f"{files.get(int(m.group(1)), '?').split('/')[-1]}:{m.group(2)}"
        continue
    if l.startswith("."):
        if l.endswith(":"):
            labels[l[:-1]] = len(ins)
        continue
    if l.endswith(":"):
        labels[l[:-1]] = len(ins)
        continue
    parts = l.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []

    def regs(o):
        r = set()
        for a, b, c in VREG.findall(o):
            if a:
                r.add(int(a))
            else:
                r.update(range(int(b), int(c) + 1))
        return r

    defs, uses = set(), set()
    is_store = re.match(r"(ds_write|ds_store|global_store|buffer_store|scratch_store|flat_store|ds_add|global_atomic|s_)", op) is not None
    if is_store or not ops:
        for o in ops:
            uses |= regs(o)
    else:
        defs |= regs(ops[0])
        k0 = 1
        if op.startswith("v_mad_u64_u32") or op.startswith("v_mad_i64_i32"):
            k0 = 2
        for o in ops[k0:]:
            uses |= regs(o)
        if re.match(r"v_(fmac|mac|pk_fmac|dot\w*c)", op) or op.startswith("v_swap"):
            uses |= defs
        if op.startswith("v_swap"):
            defs |= regs(ops[1])
        if op.startswith("v_cmp") or op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
            uses |= regs(ops[0]) if op.startswith("v_cmp") and "v" in ops[0] else set()
            defs = set() if not op.startswith("v_cmpx") else defs
            if op.startswith("v_cmp") or op.startswith("v_read"):
                defs = set()
    tgt = None
    kind = "n"
    if op.startswith("s_cbranch"):
        tgt, kind = ops[0], "c"
    elif op == "s_branch":
        tgt, kind = ops[0], "b"
    elif op == "s_endpgm":
        kind = "e"
    ins.append([l, loc, defs, uses, tgt, kind])

n = len(ins)
succ = [[] for _ in range(n)]
for i, (_, _, _, _, tgt, kind) in enumerate(ins):
    if kind in ("n", "c") and i + 1 < n:
        succ[i].append(i + 1)
    if kind in ("c", "b") and tgt in labels and labels[tgt] < n:
        succ[i].append(labels[tgt])
live_in = [set() for _ in range(n)]
changed = True
while changed:
    changed = False
    for i in range(n - 1, -1, -1):
        out = set()
        for s in succ[i]:
            out |= live_in[s]
        new = (out - ins[i][2]) | ins[i][3]
        if new != live_in[i]:
            live_in[i] = new
            changed = True
press = [len(s) for s in live_in]
print(f"{n} instructions, max live VGPRs {max(press)}")
# pressure profile by source line region: print the top points, at least 40 instructions apart
order = sorted(range(n), key=lambda i: -press[i])
shown = []
for i in order:
    if all(abs(i - j) > 40 for j in shown):
        shown.append(i)
        print(f"  #{i:5d} live {press[i]:3d}  {ins[i][1]:28s} {ins[i][0][:70]}")
    if len(shown) >= top:
        break
# where do the registers live at the first peak come from?  (source line of the latest definition before it)
if shown:
    pk = shown[0]
    hist = {}
    for r in sorted(live_in[pk]):
        j = pk - 1
        while j >= 0 and r not in ins[j][2]:
            j -= 1
        key = ins[j][1] if j >= 0 else "(kernel entry)"
        hist[key] = hist.get(key, 0) + 1
    print(f"live at #{pk} by defining source line:")
    for k, v in sorted(hist.items(), key=lambda kv: -kv[1]):
        print(f"   {v:4d}  {k}")
# coarse profile
step = max(1, n // 60)
print("profile (instruction index: live):")
print(" ".join(f"{i}:{max(press[i:i + step])}" for i in range(0, n, step)))
