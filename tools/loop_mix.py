"""Instruction mix of the loops of a `hipcc -S` listing (tools/dev_isa.sh): for every backward branch spanning at least 150 instructions, the
counts by kind — what a lone wave pays issue time for (in the latency layouts every instruction costs about five cycles)."""
import collections, re, sys
L = [l.rstrip() for l in open(sys.argv[1])]
minlen = int(sys.argv[2]) if len(sys.argv) > 2 else 150
ins = []      # (index in file, text)
labels = {}
for i, l in enumerate(L):
    t = l.split(";")[0].strip()
    if not t or t.startswith("//"):
        continue
    m = re.match(r"^([.\w$]+):$", t)
    if m:
        labels[m.group(1)] = len(ins); continue
    if t.startswith("."):
        continue
    ins.append(t)
def kind(t):
    op = t.split()[0]
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")): return "readlane/writelane"
    if op.startswith(("v_mov", "v_accvgpr")): return "v_mov/accvgpr" + ("(dpp)" if "dpp" in op or "quad_perm" in t or "row_" in t else "")
    if op.startswith(("v_cndmask", "v_bfi")): return "select"
    if op.startswith("v_pk_"): return "v_pk"
    if "quad_perm" in t or "row_ror" in t or "row_shr" in t or "row_bcast" in t: return "dpp-alu"
    if op.startswith(("v_fma", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_fmac", "v_med3", "v_subrev_f32", "v_max_f32", "v_min_f32")): return "v_fp"
    if op.startswith("v_"): return "v_int"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_"): return "salu"
    if op.startswith(("global", "buffer", "flat", "scratch")): return "vmem" + ("(scratch)" if op.startswith("scratch") else "")
    return op
for i, t in enumerate(ins):
    if t.startswith(("s_cbranch", "s_branch")):
        tgt = t.split()[-1]
        if tgt in labels and labels[tgt] < i and i - labels[tgt] >= minlen:
            lo, hi = labels[tgt], i
            c = collections.Counter(kind(x) for x in ins[lo:hi + 1])
            print(f"loop {tgt}: {hi - lo + 1} instructions  " + "  ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
