// VALU issue-rate probe, inline-asm edition (the compiler cannot re-pack or re-encode anything):
// cycles per wave64 instruction per SIMD at 1..4 waves per SIMD for
//   0 v_fmac_f32_e32 (VOP2, 4-byte encoding)      1 v_fma_f32 (VOP3, 8 bytes)      2 v_fmaak_f32 (VOP2 + 32-bit literal, 8 bytes)
//   3 v_pk_fma_f32 (VOP3P, two FMAs per lane)      4 v_med3_f32 (VOP3)             5 v_lshl_add_u32 (VOP3)
//   6 v_mul_f32_e32 (VOP2)                         7 v_add_f32_e32 (VOP2)          8 v_fma_f32 with an SGPR operand
//   9 v_exp_f32                                    10 v_rcp_f32                    11 the mix of the math_mode fast rollout step: one transcendental per 3.8 plain
// 24 independent accumulators per wave: dependent-issue distance 24 instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP24(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23)
template <int MODE>
__global__ void __launch_bounds__(256) probe(float* out, int iters, float seed, unsigned long long* cyc) {
    float a[24];
    f2 p[12];
    for (int i = 0; i < 24; ++i) a[i] = seed + i + threadIdx.x;
    for (int i = 0; i < 12; ++i) p[i] = f2{a[i], a[i] + 1.f};
    float c = seed * 0.999f, d = 0.5f;
    f2 c2 = f2{c, c}, d2 = f2{d, d};
    float sc = __builtin_amdgcn_readfirstlane(c);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define F0(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
#define F1(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
#define F2(i) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f000000" : "+v"(a[i]) : "v"(c));
#define F3(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(i) % 12]) : "v"(c2), "v"(d2));
#define F4(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
#define F5(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(c));
#define F6(i) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
#define F7(i) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
#define F8(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sc), "v"(d));
#define F9(i) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(a[i]));
#define F10(i) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(a[i]));
#define F11(i) if ((i) % 5 == 0) { if ((i) % 10 == 0) { F9(i) } else { F10(i) } } else { F0(i) }
        if (MODE == 9) { REP24(F9) }
        if (MODE == 10) { REP24(F10) }
        if (MODE == 11) { REP24(F11) }
        if (MODE == 0) { REP24(F0) }
        if (MODE == 1) { REP24(F1) }
        if (MODE == 2) { REP24(F2) }
        if (MODE == 3) { REP24(F3) }
        if (MODE == 4) { REP24(F4) }
        if (MODE == 5) { REP24(F5) }
        if (MODE == 6) { REP24(F6) }
        if (MODE == 7) { REP24(F7) }
        if (MODE == 8) { REP24(F8) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    float s = 0;
    for (int i = 0; i < 24; ++i) s += a[i];
    for (int i = 0; i < 12; ++i) s += p[i][0] + p[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, float* d, int wps) {
    int iters = 200000, blocks = 256 * wps;
    static unsigned long long* cyc = nullptr;
    if (!cyc) (void)hipMalloc(&cyc, 256 * 8 * 4 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<MODE><<<blocks, 256>>>(d, 20000, 1.0f, cyc); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); probe<MODE><<<blocks, 256>>>(d, iters, 1.0f, cyc); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[256 * 8 * 4];
    (void)hipMemcpy(h, cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (int i = 0; i < blocks * 4; ++i) sum += (double)h[i];
    double per_wave = sum / (blocks * 4) / ((double)iters * 24);     // shader cycles per instruction as seen by one wave
    printf("%-16s waves/SIMD %d: %8.3f ms  per-wave %.2f cyc/instr -> per-SIMD %.2f cyc/instr  (implied clock %.2f GHz)\n", name, wps, ms, per_wave, per_wave / wps,
           sum / (blocks * 4) / (ms * 1e6));
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int wps : {1, 2, 3, 4}) {
        run<0>("v_fmac_e32", d, wps); run<1>("v_fma VOP3", d, wps); run<2>("v_fmaak literal", d, wps); run<3>("v_pk_fma", d, wps);
        run<4>("v_med3", d, wps); run<5>("v_lshl_add", d, wps); run<6>("v_mul_e32", d, wps); run<7>("v_add_e32", d, wps); run<8>("v_fma VOP3 sgpr", d, wps);
        run<9>("v_exp_f32", d, wps); run<10>("v_rcp_f32", d, wps); run<11>("5 trans + 19 fmac", d, wps);
    }
    return 0;
}
