"""Static instruction mix of the loops of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only).
usage: isa_loops.py k.s <substring of the mangled kernel name> [min_instructions]"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
minins = int(sys.argv[3]) if len(sys.argv) > 3 else 150
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end + 1]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    return "other"
ins = [(i, l.split()[0]) for i, l in enumerate(body) if l.startswith("\t") and not l.strip().startswith((".", ";")) and l.split()]
loops = []
for i, op in ins:
    if op.startswith("s_cbranch") or op == "s_branch":
        tgt = body[i].split()[1]
        if tgt in labels and labels[tgt] < i: loops.append((labels[tgt], i))
print(f"{body[0][:100]}  instructions={len(ins)}")
for a, b in sorted(loops):
    sub = [op for i, op in ins if a <= i <= b]
    if len(sub) < minins: continue
    c = collections.Counter(cls(op) for op in sub)
    top = collections.Counter(op for op in sub if cls(op) == "valu").most_common(14)
    print(f"loop lines {start+a+1}-{start+b+1}: n={len(sub)} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
    print("    " + ", ".join(f"{k}:{v}" for k, v in top))
