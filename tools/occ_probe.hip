// occ_probe — how many workgroups of a 128- / 256-thread kernel with ~168 VGPRs does the runtime place on one CU, as a function of the
// dynamic LDS size? (hipOccupancyMaxActiveBlocksPerMultiprocessor) and what does a timed spin kernel show? Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
extern __shared__ float sm[];
template <int BNT>
__global__ void __launch_bounds__(BNT, 3) k(float* out, int iters, unsigned long long* clk) {
    asm volatile("v_mov_b32 v160, 0" ::: "v160");
    float a = threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) a = __builtin_fmaf(a, 1.0000001f, 0.5f);
    sm[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x == 0) { out[blockIdx.x] = sm[(blockIdx.x + 1) % BNT]; clk[blockIdx.x] = __builtin_readcyclecounter() - t0; }
}
template <int BNT>
void probe(const char* name) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 1 << 20); hipMalloc(&clk, 1 << 20);
    hipFuncSetAttribute((const void*)k<BNT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int kb : {8, 16, 20, 21, 22, 23, 24, 26, 28, 30, 31, 32, 40, 50, 52, 53, 54, 64, 80}) {
        int n = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<BNT>, BNT, (size_t)kb * 1024);
        // timed: 256 CUs x n blocks = one full wave of workgroups vs. twice that
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms[2];
        for (int r = 0; r < 2; ++r) {
            int grid = 256 * (r ? 12 * 128 / BNT : 6 * 128 / BNT) ;   // 6 (12) per CU of the 128-thread kernel's worth of waves: 12 / 24 waves per CU
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<BNT>, dim3(grid), dim3(BNT), (size_t)kb * 1024, 0, out, 2000000, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[r], e0, e1);
        }
        printf("%s lds %2d KB: occupancy API %d blocks/CU; 12 waves/CU worth: %.2f ms, 24 waves/CU worth: %.2f ms\n", name, kb, n, ms[0], ms[1]);
    }
}
int main() { probe<128>("BNT=128"); probe<256>("BNT=256"); return 0; }
