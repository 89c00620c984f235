// Per-wave issue cost of scalar / wait / no-op instructions next to vector ones (one wave per SIMD):
// how much wave time does the address arithmetic of the lane-layout step cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned long long* cyc) {
    float a = threadIdx.x, b = 1.0001f, c = 0.5f;
    int s0 = 1, s1 = 2;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { REP16(asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
        if (MODE == 1) { REP16(asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");) }
        if (MODE == 2) { REP16(asm volatile("s_nop 0");) }
        if (MODE == 3) { REP16(asm volatile("s_waitcnt vmcnt(0)");) }
        if (MODE == 4) { REP16(asm volatile("v_fmac_f32_e32 %0, %2, %3\n s_add_u32 %1, %1, %4" : "+v"(a), "+s"(s0) : "v"(b), "v"(c), "s"(s1) : "scc");) }
        if (MODE == 5) { REP16(asm volatile("v_fmac_f32_e32 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(b), "v"(c));) }
        if (MODE == 6) { REP16(asm volatile("v_readlane_b32 %0, %2, 3\n s_nop 1\n v_fmac_f32_e32 %1, %0, %3" : "+s"(s0), "+v"(a) : "v"(c), "v"(b));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = a + s0;
}
template <int MODE>
void run(const char* name, int per_rep, float* d, unsigned long long* cyc) {
    int iters = 100000, blocks = 256;
    k<MODE><<<blocks, 256>>>(d, 1000, cyc); (void)hipDeviceSynchronize();
    k<MODE><<<blocks, 256>>>(d, iters, cyc); (void)hipDeviceSynchronize();
    static unsigned long long h[1024]; (void)hipMemcpy(h, cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks * 4; ++i) s += (double)h[i];
    printf("%-44s %.2f cycles per repetition (%d instructions)\n", name, s / (blocks * 4) / ((double)iters * 16), per_rep);
}
int main() {
    float* d; unsigned long long* cyc; (void)hipMalloc(&d, 256 * 256 * 4); (void)hipMalloc(&cyc, 1024 * 8);
    run<0>("v_fmac (dependent chain)", 1, d, cyc);
    run<1>("s_add_u32 (dependent chain)", 1, d, cyc);
    run<2>("s_nop 0", 1, d, cyc);
    run<3>("s_waitcnt vmcnt(0) (nothing outstanding)", 1, d, cyc);
    run<4>("v_fmac + s_add_u32", 2, d, cyc);
    run<5>("v_fmac + s_waitcnt", 2, d, cyc);
    run<6>("v_readlane + s_nop 1 + v_fmac", 3, d, cyc);
    return 0;
}
