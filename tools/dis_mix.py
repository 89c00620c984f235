"""Instruction mix of one kernel's loops in an llvm-objdump listing (tools/kernel_dis.sh): for every backward branch spanning at least
`minlen` instructions, the opcode histogram.   usage: dis_mix.py k.dis <substring of the mangled kernel name> [minlen] [top]"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 400
top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
L = open(path).read().split("\n")
start = next(i for i, l in enumerate(L) if l.endswith(">:") and key in l)
end = next(i for i in range(start, len(L)) if "s_endpgm" in L[i])
ins = []; full = []; addr = {}
for l in L[start + 1:end]:
    t = l.split("//")[0].strip()
    m = re.search(r"//\s*([0-9A-Fa-f]{8,}):", l)
    if not t or not m: continue
    addr[int(m.group(1), 16)] = len(ins); ins.append(t); full.append(l)
print(L[start], len(ins), "instructions")
for i, t in enumerate(ins):
    if t.startswith(("s_cbranch", "s_branch")):
        m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", full[i])
        if not m: continue
        base = int(re.match(r"^([0-9a-f]+)", L[start]).group(1), 16)
        tgt = addr.get(base + int(m.group(1), 16))
        if tgt is not None and tgt < i and i - tgt >= minlen:
            c = collections.Counter(x.split()[0] for x in ins[tgt:i + 1])
            n = i - tgt + 1
            valu = sum(v for k, v in c.items() if k.startswith("v_") and "mfma" not in k)
            trans = sum(v for k, v in c.items() if k.startswith(("v_exp", "v_rcp", "v_rsq", "v_log", "v_sqrt", "v_sin", "v_cos")))
            print(f"loop @{tgt}..{i}: {n} instructions, VALU {valu} (transcendental {trans}), MFMA {sum(v for k, v in c.items() if 'mfma' in k)}, "
                  f"LDS {sum(v for k, v in c.items() if k.startswith('ds_'))}, VMEM {sum(v for k, v in c.items() if k.startswith(('global', 'buffer', 'scratch', 'flat')))}, "
                  f"SALU {sum(v for k, v in c.items() if k.startswith('s_') and not k.startswith(('s_waitcnt', 's_nop')))}, s_waitcnt {c['s_waitcnt']}, s_nop {c['s_nop']}")
            print("   " + "  ".join(f"{k} {v}" for k, v in c.most_common(top)))
