#!/bin/bash
# Disassemble the gfx950 code object embedded in a hipcc object file (seconds, instead of a second 4-minute `hipcc -S`):
#   tools/kernel_dis.sh sde4mbrl_px4_amd/csrc/sdempc_kernels_duo2.o /tmp/k.dis      (_duo2: TeamPair / TeamBlock2 solve kernels, _duo4: TeamBlock)
# then e.g. tools/dis_loops.py /tmp/k.dis TeamPairELi4ELb0ELb0ELi3   (VMEM / scratch / s_waitcnt vmcnt sites of one kernel)
set -e
obj=${1:?object file}; out=${2:?output listing}
LLVM=/opt/rocm/lib/llvm/bin
tmp=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$obj" $tmp/fatbin
$LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fatbin --output=$tmp/co --unbundle
$LLVM/llvm-objdump -d $tmp/co > "$out"
rm -rf $tmp
ls -la "$out"
