"""Memory-side view of one kernel in an llvm-objdump listing (tools/kernel_dis.sh): scratch accesses, vector-memory instructions,
`s_waitcnt vmcnt` sites and backward branches, in program order with body-relative line numbers.
usage: dis_loops.py k.dis <substring of the mangled kernel name> [first last]   (first/last: body-relative line range)"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
L = open(path).read().split("\n")
start = next(i for i, l in enumerate(L) if l.endswith(">:") and key in l)
end = next(i for i in range(start, len(L)) if "s_endpgm" in L[i])
body = L[start:end]
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, len(body))
addr = {}
for i, l in enumerate(body):
    m = re.search(r"//\s*([0-9A-Fa-f]{8,}):", l)
    if m: addr[int(m.group(1), 16)] = i
print(body[0], "instructions:", len(body), " scratch:", sum("scratch_" in l for l in body))
for i in range(lo, hi):
    l = body[i]; t = l.split("//")[0].strip()
    if not t: continue
    op = t.split()[0]
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")) or ("s_waitcnt" in op and "vmcnt" in t) or op.startswith(("s_cbranch", "s_branch")):
        print(i, t[:100])
