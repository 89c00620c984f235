#!/bin/bash
# Compile ONE latency kernel of translation unit 0 to ISA (device side only, seconds) and print the instruction mix of its step loops:
#   tools/dev_isa.sh [1|2|3|4] [extra flags]     1: speculative kernel m = 4 (default), 2: the same for P = 1, 3: plain cooperative kernel, 4: P = 1 wave team
# Output: /tmp/dev_isa_<n>.s and the per-loop counts of tools/loop_mix.py.
set -e
n=${1:-1}; shift || true
root=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -Wno-unused-result -DSDEMPC_DEV_KERNEL=$n "$@" \
    -S --cuda-device-only -o /tmp/dev_isa_$n.s $root/sde4mbrl_px4_amd/csrc/sdempc_kernels.hip 2>&1 | grep -v "argument unused" || true
python3 $root/tools/loop_mix.py /tmp/dev_isa_$n.s
