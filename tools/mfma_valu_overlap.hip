// Do v_mfma_f32_32x32x2_f32 and f32 VALU instructions overlap on one SIMD?
// Blocks of 256 threads (1 wave per SIMD); grid = 256 CUs * k so k waves share a SIMD.
// mode 0: every wave VALU-only; mode 1: every wave MFMA-only; mode 2: even blocks MFMA-only, odd blocks VALU-only;
// mode 3: every wave interleaves 1 MFMA + 14 VALU (independent).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) k(float* out, int iters, int mode, float seed) {
    float a[14];
    for (int i = 0; i < 14; ++i) a[i] = seed + i + threadIdx.x;
    f32x16 acc0 = {0}, acc1 = {0};
    const float c = seed * 0.999f, d = 0.5f;
    bool do_mfma = mode == 1 || mode == 3 || (mode == 2 && (blockIdx.x & 1) == 0);
    bool do_valu = mode == 0 || mode == 3 || (mode == 2 && (blockIdx.x & 1) == 1);
    if (do_mfma && do_valu) {
        for (int it = 0; it < iters; ++it) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c, a[0], acc0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 14; ++i) a[i] = __builtin_fmaf(a[i], c, d);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(d, c, acc1, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 14; ++i) a[i] = __builtin_fmaf(a[i], c, d);
        }
    } else if (do_mfma) {
        for (int it = 0; it < iters; ++it) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c, d, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(d, c, acc1, 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 14; ++i) a[i] = __builtin_fmaf(a[i], c, d);
        }
    }
    float s = 0;
    for (int i = 0; i < 14; ++i) s += a[i];
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    int iters = 20000;
    const char* names[4] = {"VALU-only (28 fma/iter)", "MFMA-only (2 mfma/iter)", "half waves MFMA, half VALU", "interleaved 2 mfma + 28 fma"};
    for (int wps : {2, 4})
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            k<<<256 * wps, 256>>>(d, 100, mode, 1.f); hipDeviceSynchronize();
            hipEventRecord(e0); k<<<256 * wps, 256>>>(d, iters, mode, 1.f); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("waves/SIMD %d  %-30s %.3f ms  (%.1f ns per iteration per wave-slot)\n", wps, names[mode], ms, ms * 1e6 / iters / wps);
        }
    return 0;
}
