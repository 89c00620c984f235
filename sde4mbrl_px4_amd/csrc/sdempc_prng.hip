// sdempc_prng.hip — key-derived noise on the device (SPEC.md §7): threefry2x32 counter-based generator with JAX's key
// conventions, mantissa-trick uniform, sqrt(2)*erfinv normal, written straight into the particle-minor layout
// f32[B][G][H][6][32] that the rollout kernels stream.
//
// Path replaced (reference): the key plumbing of the MPC node — jax.random.PRNGKey(seed) and its 3-way split
// (sde4mbrl_px4/mpc_controller/sde_control.py:338-341), one key into and out of every m_mpc call
// (sde_control.py:349-350,400-416,698,717). With this kernel the noise tensor never exists on the host: the boundary hands
// over 8 bytes of key per instance instead of P*H*6 floats over PCIe.
//
// Mapping: integer ALU + a short f32 polynomial per element, HBM-write-bound. One thread computes one threefry block, i.e. the
// two tensor elements e and e + N/2 (N = P*H*6; legacy JAX layout: first half of the counters in x0, second half in x1).
// Threads are ordered particle-fastest so that both stores of a wave fill whole 128-byte rows of the [..][32] layout.
// All f32 arithmetic is the explicit fma sequence of SPEC.md §7.2 (bit-identical to oracle/prng_oracle.c).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sdempc_kernels.h"

namespace sdempc {

#define FMA(a, b, c) __builtin_fmaf((a), (b), (c))
#define DI __device__ __forceinline__

DI uint32_t rotl32(uint32_t x, int r) { return __builtin_rotateleft32(x, r); }

DI void threefry2x32(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
    const uint32_t k2 = k0 ^ k1 ^ 0x1BD11BDAu;
#define TF_ROUND(r) { x0 += x1; x1 = rotl32(x1, r); x1 ^= x0; }
    x0 += k0; x1 += k1;
    TF_ROUND(13) TF_ROUND(15) TF_ROUND(26) TF_ROUND(6)
    x0 += k1; x1 += k2 + 1u;
    TF_ROUND(17) TF_ROUND(29) TF_ROUND(16) TF_ROUND(24)
    x0 += k2; x1 += k0 + 2u;
    TF_ROUND(13) TF_ROUND(15) TF_ROUND(26) TF_ROUND(6)
    x0 += k0; x1 += k1 + 3u;
    TF_ROUND(17) TF_ROUND(29) TF_ROUND(16) TF_ROUND(24)
    x0 += k1; x1 += k2 + 4u;
    TF_ROUND(13) TF_ROUND(15) TF_ROUND(26) TF_ROUND(6)
    x0 += k2; x1 += k0 + 5u;
#undef TF_ROUND
}

DI float log_spec(float t) {
    const uint32_t b = __float_as_uint(t);
    int e = (int)((b >> 23) & 255u) - 126;
    const float m = __uint_as_float((b & 0x007FFFFFu) | 0x3F000000u);
    float x;
    if (m < 0.707106781186547524f) { e -= 1; x = (m + m) - 1.0f; } else x = m - 1.0f;
    const float z = x * x;
    float y = 7.0376836292E-2f;
    y = FMA(y, x, -1.1514610310E-1f);
    y = FMA(y, x, 1.1676998740E-1f);
    y = FMA(y, x, -1.2420140846E-1f);
    y = FMA(y, x, 1.4249322787E-1f);
    y = FMA(y, x, -1.6668057665E-1f);
    y = FMA(y, x, 2.0000714765E-1f);
    y = FMA(y, x, -2.4999993993E-1f);
    y = FMA(y, x, 3.3333331174E-1f);
    y = (y * x) * z;
    const float fe = (float)e;
    y = FMA(-2.12194440e-4f, fe, y);
    y = FMA(-0.5f, z, y);
    float r = x + y;
    r = FMA(0.693359375f, fe, r);
    return r;
}
DI float sqrt_spec(float a) {
    float y = __uint_as_float(0x5F3759DFu - (__float_as_uint(a) >> 1));
    const float h = 0.5f * a;
#pragma unroll
    for (int i = 0; i < 3; ++i) { float t = y * y; t = FMA(-h, t, 1.5f); y = y * t; }
    float s = a * y;
    const float r = FMA(-s, s, a);
    return FMA(r, 0.5f * y, s);
}
DI float erfinv_spec(float u) {
    float w = -log_spec(FMA(-u, u, 1.0f));
    float p;
    if (w < 5.0f) {
        w = w - 2.5f;
        p = 2.81022636e-08f;
        p = FMA(p, w, 3.43273939e-07f);
        p = FMA(p, w, -3.5233877e-06f);
        p = FMA(p, w, -4.39150654e-06f);
        p = FMA(p, w, 0.00021858087f);
        p = FMA(p, w, -0.00125372503f);
        p = FMA(p, w, -0.00417768164f);
        p = FMA(p, w, 0.246640727f);
        p = FMA(p, w, 1.50140941f);
    } else {
        w = sqrt_spec(w) - 3.0f;
        p = -0.000200214257f;
        p = FMA(p, w, 0.000100950558f);
        p = FMA(p, w, 0.00134934322f);
        p = FMA(p, w, -0.00367342844f);
        p = FMA(p, w, 0.00573950773f);
        p = FMA(p, w, -0.0076224613f);
        p = FMA(p, w, 0.00943887047f);
        p = FMA(p, w, 1.00167406f);
        p = FMA(p, w, 2.83297682f);
    }
    return p * u;
}
DI float bits_to_normal(uint32_t bits) {
    const float lo = -0.99999994f;
    const float f = __uint_as_float((bits >> 9) | 0x3F800000u) - 1.0f;
    float u = FMA(f, 2.0f, lo);
    if (!(u > lo)) u = lo;
    return 1.41421354f * erfinv_spec(u);
}

// grid: x = chunks of 256 threads over [pg][r][lane] (pg < ceil(ceil(P/2)/32), r < H*6), y = instance
__global__ void __launch_bounds__(256) sdempc_noise_kernel(const uint32_t* __restrict__ keys, float* __restrict__ out, int P, int G, int HC, int b0) {
    const int b = b0 + blockIdx.y;
    const uint32_t k0 = keys[2 * b], k1 = keys[2 * b + 1];
    const unsigned N = (unsigned)P * (unsigned)HC, half = N / 2;          // N is even (HC = 6H)
    const unsigned k = blockIdx.x * 256u + threadIdx.x;
    const unsigned lane = k & 31u, r = (k >> 5) % (unsigned)HC, pg = (k >> 5) / (unsigned)HC;
    const unsigned p = pg * 32u + lane;
    const unsigned e0 = p * (unsigned)HC + r;
    if (p >= (unsigned)P || e0 >= half) return;
    uint32_t x0 = e0, x1 = e0 + half;
    threefry2x32(k0, k1, x0, x1);
    float* ob = out + (size_t)b * G * HC * 32;
    ob[((size_t)(p >> 5) * HC + r) * 32 + (p & 31u)] = bits_to_normal(x0);
    const unsigned e1 = e0 + half, p1 = e1 / (unsigned)HC, r1 = e1 - p1 * (unsigned)HC;
    ob[((size_t)(p1 >> 5) * HC + r1) * 32 + (p1 & 31u)] = bits_to_normal(x1);
}

hipError_t launch_noise_from_keys(const uint32_t* keys_dev, float* out, int B, int P, int G, int H, hipStream_t st) {
    if (B < 1 || P < 1 || H < 1 || G != (P + 31) / 32 || (long long)P * H * 6 >= (1ll << 31)) return hipErrorInvalidValue;
    const int HC = H * 6;
    if (P & 31) {   // padded particles of the last group read as zero noise
        hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * G * HC * 32, st);
        if (e != hipSuccess) return e;
    }
    const int np1 = (P + 1) / 2, npg = (np1 + 31) / 32;
    const long long threads = (long long)npg * 32 * HC;
    const unsigned gx = (unsigned)((threads + 255) / 256);
    for (int b0 = 0; b0 < B; b0 += 65535) {
        const int nb = B - b0 < 65535 ? B - b0 : 65535;
        sdempc_noise_kernel<<<dim3(gx, nb), 256, 0, st>>>(keys_dev, out, P, G, HC, b0);
    }
    return hipGetLastError();
}

}  // namespace sdempc
