// tanh16 throughput probe: SPEC.md §3.4 tanh on a 16-register tile, scalar f32 form vs hand-packed v_pk_*_f32 form,
// at 1..4 waves per SIMD (block = 256 threads = 1 wave per SIMD; blocks per CU = waves per SIMD).
// Both forms perform the same IEEE operations in the same order per value (bit-identical results; checked below).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define FMA(a, b, c) __builtin_fmaf((a), (b), (c))
#define DI __device__ __forceinline__

DI float rcp_spec(float d) {
    float y = __uint_as_float(0x7EF311C7u - __float_as_uint(d));
#pragma unroll
    for (int i = 0; i < 3; ++i) { float e = FMA(-d, y, 1.0f); y = FMA(y, e, y); }
    return y;
}
DI float exp2_spec(float x, float c) {
    float t2 = FMA(x, c, 12582912.0f);
    float n = t2 - 12582912.0f;
    float f = FMA(x, c, -n);
    float p = 0.001327647129073739f;
    p = FMA(p, f, 0.009675540961325169f);
    p = FMA(p, f, 0.05550713092088699f);
    p = FMA(p, f, 0.24022120237350464f);
    p = FMA(p, f, 0.6931469440460205f);
    p = FMA(p, f, 1.0000001192092896f);
    return __uint_as_float(__float_as_uint(p) + (__float_as_uint(t2) << 23));
}
DI float clampf(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
DI void tanh4(float& a0, float& a1, float& a2, float& a3) {
    float d0 = 1.0f + exp2_spec(clampf(a0, -9.0f, 9.0f), 2.885390043258667f);
    float d1 = 1.0f + exp2_spec(clampf(a1, -9.0f, 9.0f), 2.885390043258667f);
    float d2 = 1.0f + exp2_spec(clampf(a2, -9.0f, 9.0f), 2.885390043258667f);
    float d3 = 1.0f + exp2_spec(clampf(a3, -9.0f, 9.0f), 2.885390043258667f);
    float p2 = d0 * d1, p3 = p2 * d2, p4 = p3 * d3;
    float r = rcp_spec(p4);
    float r3 = r * p3; r = r * d3;
    float r2 = r * p2; r = r * d2;
    float r1 = r * d0;
    float r0 = r * d1;
    a0 = FMA(-2.0f, r0, 1.0f); a1 = FMA(-2.0f, r1, 1.0f); a2 = FMA(-2.0f, r2, 1.0f); a3 = FMA(-2.0f, r3, 1.0f);
}
DI void tanh16_scalar(f32x16& v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float a = v[4 * q], b = v[4 * q + 1], c = v[4 * q + 2], d = v[4 * q + 3];
        tanh4(a, b, c, d);
        v[4 * q] = a; v[4 * q + 1] = b; v[4 * q + 2] = c; v[4 * q + 3] = d;
    }
}

// ---- packed form: two values per instruction ----
DI f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
DI f2 splat(float x) { return f2{x, x}; }
DI f2 exp2d_pk(f2 x) {   // 1 + 2^(x*c), x already clamped
    const f2 c = splat(2.885390043258667f), mg = splat(12582912.0f);
    f2 t2 = pk_fma(x, c, mg);
    f2 n = t2 - mg;
    f2 f = pk_fma(x, c, -n);
    f2 p = splat(0.001327647129073739f);
    p = pk_fma(p, f, splat(0.009675540961325169f));
    p = pk_fma(p, f, splat(0.05550713092088699f));
    p = pk_fma(p, f, splat(0.24022120237350464f));
    p = pk_fma(p, f, splat(0.6931469440460205f));
    p = pk_fma(p, f, splat(1.0000001192092896f));
    f2 e;
    e[0] = __uint_as_float(__float_as_uint(p[0]) + (__float_as_uint(t2[0]) << 23));
    e[1] = __uint_as_float(__float_as_uint(p[1]) + (__float_as_uint(t2[1]) << 23));
    return e + splat(1.0f);
}
// two tanh4 groups (a0..a3), (b0..b3) processed together: the two reciprocals share packed Newton steps
DI void tanh8_pk(float* a, float* b) {
    f2 a01 = f2{clampf(a[0], -9.f, 9.f), clampf(a[1], -9.f, 9.f)}, a23 = f2{clampf(a[2], -9.f, 9.f), clampf(a[3], -9.f, 9.f)};
    f2 b01 = f2{clampf(b[0], -9.f, 9.f), clampf(b[1], -9.f, 9.f)}, b23 = f2{clampf(b[2], -9.f, 9.f), clampf(b[3], -9.f, 9.f)};
    f2 da01 = exp2d_pk(a01), da23 = exp2d_pk(a23), db01 = exp2d_pk(b01), db23 = exp2d_pk(b23);
    // prefix products, group a in lane 0 of the pair, group b in lane 1
    f2 d0 = f2{da01[0], db01[0]}, d1 = f2{da01[1], db01[1]}, d2 = f2{da23[0], db23[0]}, d3 = f2{da23[1], db23[1]};
    f2 p2 = d0 * d1, p3 = p2 * d2, p4 = p3 * d3;
    f2 y;
    y[0] = __uint_as_float(0x7EF311C7u - __float_as_uint(p4[0]));
    y[1] = __uint_as_float(0x7EF311C7u - __float_as_uint(p4[1]));
#pragma unroll
    for (int i = 0; i < 3; ++i) { f2 e = pk_fma(-p4, y, splat(1.0f)); y = pk_fma(y, e, y); }
    f2 r = y;
    f2 r3 = r * p3; r = r * d3;
    f2 r2 = r * p2; r = r * d2;
    f2 r1 = r * d0;
    f2 r0 = r * d1;
    const f2 m2 = splat(-2.0f), one = splat(1.0f);
    f2 t0 = pk_fma(m2, r0, one), t1 = pk_fma(m2, r1, one), t2 = pk_fma(m2, r2, one), t3 = pk_fma(m2, r3, one);
    a[0] = t0[0]; a[1] = t1[0]; a[2] = t2[0]; a[3] = t3[0];
    b[0] = t0[1]; b[1] = t1[1]; b[2] = t2[1]; b[3] = t3[1];
}
DI void tanh16_pk(f32x16& v) {
#pragma unroll
    for (int q = 0; q < 4; q += 2) {
        float a[4] = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        float b[4] = {v[4 * q + 4], v[4 * q + 5], v[4 * q + 6], v[4 * q + 7]};
        tanh8_pk(a, b);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[4 * q + i] = a[i]; v[4 * q + 4 + i] = b[i]; }
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) probe(float* out, int iters, float seed) {
    f32x16 v;
    for (int i = 0; i < 16; ++i) v[i] = seed * (0.01f * (float)(threadIdx.x % 97) - 0.4f) + 0.1f * i;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) tanh16_scalar(v); else tanh16_pk(v);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = FMA(v[i], 1.7f, 0.03f * (i - 8));   // keep the arguments spread out
    }
    for (int i = 0; i < 16; ++i) out[(blockIdx.x * 256 + threadIdx.x) * 16 + i] = v[i];
}
template <int MODE>
float run(const char* name, float* d, int wps) {
    int iters = 4000, blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<MODE><<<blocks, 256>>>(d, 50, 1.0f); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); probe<MODE><<<blocks, 256>>>(d, iters, 1.0f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    int clk; (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    double ns = ms * 1e6 / ((double)iters * wps);
    printf("%-8s waves/SIMD %d: %.3f ms -> %.1f ns per tanh16(+16 fma) per SIMD = %.1f cycles at %.2f GHz nominal\n", name, wps, ms, ns, ns * clk * 1e-6, clk * 1e-6);
    return ms;
}
int main() {
    float *d0, *d1; size_t n = (size_t)256 * 4 * 256 * 16;
    (void)hipMalloc(&d0, n * 4); (void)hipMalloc(&d1, n * 4);
    // bit-equality of the two forms
    probe<0><<<256, 256>>>(d0, 7, 1.0f); probe<1><<<256, 256>>>(d1, 7, 1.0f); (void)hipDeviceSynchronize();
    float* h0 = new float[256 * 256 * 16]; float* h1 = new float[256 * 256 * 16];
    (void)hipMemcpy(h0, d0, 256 * 256 * 16 * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(h1, d1, 256 * 256 * 16 * 4, hipMemcpyDeviceToHost);
    printf("scalar vs packed bitwise equal: %s\n", memcmp(h0, h1, 256 * 256 * 16 * 4) == 0 ? "yes" : "NO");
    for (int wps : {1, 2, 3, 4}) { run<0>("scalar", d0, wps); run<1>("packed", d1, wps); }
    return 0;
}
