#!/bin/bash
# does the runtime on the box load compressed code objects? (--offload-compress)
set -e
cd $GRAFT_REPO_ROOT/tools
cat > /tmp/cp.hip <<'EOT'
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* p) { p[threadIdx.x] = threadIdx.x * 2; }
int main() { int* d; hipMalloc(&d, 256); k<<<1, 64>>>(d); int h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost); printf("compressed object ran: %d %d\n", h[1], h[63]); return h[63] == 126 ? 0 : 1; }
EOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 --offload-compress -O2 -o /tmp/cp /tmp/cp.hip && /tmp/cp
