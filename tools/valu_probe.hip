// VALU issue-rate probe: cycles per wave64 instruction per SIMD for v_fma_f32 / v_pk_fma_f32 / v_mul / dpp-add ...
// at 1, 2, 4 waves per SIMD (block = 256 threads = 1 wave per SIMD; blocks per CU = waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) probe(float* out, int iters, float seed) {
    float a[16];
    f2 p[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f2{a[i], a[i] + 1.f}; }
    const float c = seed * 0.999f, d = 0.5f;
    const f2 c2 = f2{c, c}, d2 = f2{d, d};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) a[i] = __builtin_fmaf(a[i], c, d);
            if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], c2, d2);
            if (MODE == 2) a[i] = a[i] * c;
            if (MODE == 3) a[i] = a[i] + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0xB1, 0xF, 0xF, false));
            if (MODE == 4) a[i] = __builtin_amdgcn_fmed3f(a[i], c, d);
            if (MODE == 5) a[i] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a[i]) + (__builtin_bit_cast(unsigned, c) << 23));
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i] + p[i][0] + p[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, float* d, int wps) {
    int iters = 20000, blocks = 256 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 256>>>(d, 100, 1.0f); hipDeviceSynchronize();
    hipEventRecord(e0); probe<MODE><<<blocks, 256>>>(d, iters, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double inst_per_simd = (double)iters * 16 * wps;       // wave-instructions per SIMD
    int clk; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("%-12s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles at %.2f GHz nominal)\n", name, wps, ms,
           ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * clk * 1e-6, clk * 1e-6);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int wps : {1, 2, 4}) {
        run<0>("v_fma_f32", d, wps); run<1>("v_pk_fma_f32", d, wps); run<2>("v_mul_f32", d, wps); run<3>("v_add_dpp", d, wps); run<4>("v_med3_f32", d, wps); run<5>("v_lshl_add", d, wps);
    }
    return 0;
}
