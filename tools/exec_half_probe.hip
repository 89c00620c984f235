// Does a wave64 VALU instruction whose upper 32 lanes are EXEC-masked cost less than a full one on gfx950?
// mode 0: all 64 lanes run the fma loop; mode 1: only lanes 0..31; mode 2: only lanes 32..63; mode 3: even lanes only.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
__global__ void __launch_bounds__(256) probe(float* out, int iters, int mode, unsigned long long* cyc) {
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = 1.0f + i + threadIdx.x;
    const float c = 0.999f, d = 0.5f;
    const int lane = threadIdx.x & 63;
    const bool on = mode == 0 || (mode == 1 && lane < 32) || (mode == 2 && lane >= 32) || (mode == 3 && (lane & 1) == 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (on) {
        for (int it = 0; it < iters; ++it) {
#define F(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
            REP16(F) REP16(F)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; unsigned long long* cyc; (void)hipMalloc(&d, 256 * 8 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8 * 4 * 8);
    static unsigned long long h[256 * 8 * 4];
    const char* names[4] = {"all 64 lanes", "lanes 0..31", "lanes 32..63", "even lanes"};
    for (int wps : {1, 2, 4})
        for (int mode = 0; mode < 4; ++mode) {
            int iters = 100000, blocks = 256 * wps;
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            probe<<<blocks, 256>>>(d, 20000, mode, cyc); (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0); probe<<<blocks, 256>>>(d, iters, mode, cyc); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(h, cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
            double sum = 0; for (int i = 0; i < blocks * 4; ++i) sum += (double)h[i];
            double pw = sum / (blocks * 4) / ((double)iters * 32);
            printf("waves/SIMD %d  %-14s %8.3f ms  per-wave %.2f cyc/instr  per-SIMD %.2f cyc/instr  clock %.2f GHz\n", wps, names[mode], ms, pw, pw / wps, sum / (blocks * 4) / (ms * 1e6));
        }
    return 0;
}
