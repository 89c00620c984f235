// sdempc_math.inc.h — SPEC.md §3 elementary functions (software / packed / hardware forms)
// Fragment of sdempc_kernels.hip: included inside namespace sdempc::{exact|fastm} (it is compiled twice, see there); not a
// stand-alone header.
// ------------------------------------------------------------------------------------------------
// SPEC.md §3: elementary functions (bit-reproducible: only fma / mul / add / integer ops)
// ------------------------------------------------------------------------------------------------
DI float rcp_spec(float d) {
    float y = __uint_as_float(0x7EF311C7u - __float_as_uint(d));
#pragma unroll
    for (int i = 0; i < 3; ++i) { float e = FMA(-d, y, 1.0f); y = FMA(y, e, y); }
    return y;
}
DI float rsqrt_spec(float a) {
    float y = __uint_as_float(0x5F3759DFu - (__float_as_uint(a) >> 1));
    float h = 0.5f * a;
#pragma unroll
    for (int i = 0; i < 3; ++i) { float t = y * y; t = FMA(-h, t, 1.5f); y = y * t; }
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
// tanh of 4 values with one shared reciprocal (SPEC.md §3.4)
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
DI void tanh16(f32x16& v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float a = v[4 * q], b = v[4 * q + 1], c = v[4 * q + 2], d = v[4 * q + 3];
        tanh4(a, b, c, d);
        v[4 * q] = a; v[4 * q + 1] = b; v[4 * q + 2] = c; v[4 * q + 3] = d;
    }
}
// Packed form of the same arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two values per instruction, identical IEEE
// operations per value, so results are bit-identical to tanh4). A v_pk instruction costs two issue slots of the vector
// datapath, so it gains nothing once a SIMD is shared by 2+ waves (tools/tanh_probe.hip: 772 vs 710 cycles per tile at two
// waves per SIMD) but a lone wave per SIMD is issue-bound and gets 1.5x (816 vs 1237 cycles): used by the small-batch
// (latency) instantiation of the solve kernel only.
typedef float f32x2 __attribute__((ext_vector_type(2)));
DI f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
DI f32x2 splat2(float x) { return f32x2{x, x}; }
DI f32x2 exp2d_pk(f32x2 x) {   // 1 + 2^(x*c) of two clamped values
    const f32x2 c = splat2(2.885390043258667f), mg = splat2(12582912.0f);
    f32x2 t2 = pk_fma(x, c, mg);
    f32x2 n = t2 - mg;
    f32x2 f = pk_fma(x, c, -n);
    f32x2 p = splat2(0.001327647129073739f);
    p = pk_fma(p, f, splat2(0.009675540961325169f));
    p = pk_fma(p, f, splat2(0.05550713092088699f));
    p = pk_fma(p, f, splat2(0.24022120237350464f));
    p = pk_fma(p, f, splat2(0.6931469440460205f));
    p = pk_fma(p, f, splat2(1.0000001192092896f));
    f32x2 e;
    e[0] = __uint_as_float(__float_as_uint(p[0]) + (__float_as_uint(t2[0]) << 23));
    e[1] = __uint_as_float(__float_as_uint(p[1]) + (__float_as_uint(t2[1]) << 23));
    return e + splat2(1.0f);
}
// two tanh4 groups at once: group a in element 0 of every pair, group b in element 1 (the two reciprocals share the Newton steps)
DI void tanh8_pk(float* a, float* b) {
    f32x2 a01 = f32x2{clampf(a[0], -9.0f, 9.0f), clampf(a[1], -9.0f, 9.0f)}, a23 = f32x2{clampf(a[2], -9.0f, 9.0f), clampf(a[3], -9.0f, 9.0f)};
    f32x2 b01 = f32x2{clampf(b[0], -9.0f, 9.0f), clampf(b[1], -9.0f, 9.0f)}, b23 = f32x2{clampf(b[2], -9.0f, 9.0f), clampf(b[3], -9.0f, 9.0f)};
    f32x2 da01 = exp2d_pk(a01), da23 = exp2d_pk(a23), db01 = exp2d_pk(b01), db23 = exp2d_pk(b23);
    f32x2 d0 = f32x2{da01[0], db01[0]}, d1 = f32x2{da01[1], db01[1]}, d2 = f32x2{da23[0], db23[0]}, d3 = f32x2{da23[1], db23[1]};
    f32x2 p2 = d0 * d1, p3 = p2 * d2, p4 = p3 * d3;
    f32x2 y;
    y[0] = __uint_as_float(0x7EF311C7u - __float_as_uint(p4[0]));
    y[1] = __uint_as_float(0x7EF311C7u - __float_as_uint(p4[1]));
#pragma unroll
    for (int i = 0; i < 3; ++i) { f32x2 e = pk_fma(-p4, y, splat2(1.0f)); y = pk_fma(y, e, y); }
    f32x2 r = y;
    f32x2 r3 = r * p3; r = r * d3;
    f32x2 r2 = r * p2; r = r * d2;
    f32x2 r1 = r * d0;
    f32x2 r0 = r * d1;
    const f32x2 m2 = splat2(-2.0f), one = splat2(1.0f);
    f32x2 t0 = pk_fma(m2, r0, one), t1 = pk_fma(m2, r1, one), t2 = pk_fma(m2, r2, one), t3 = pk_fma(m2, r3, one);
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
// math_mode fast (SPEC.md §10, §10b): the activation is kept as r = 1 / (1 + 2^a') on the transcendental unit, a' = (2 log2 e) a; tanh(a) = 1 - 2 r.
// The pre-scale and the affine map live in the weights (sdempc_create builds them), the derivative 1 - tanh^2 = 4 (r - r^2) leaves its 4 in the
// transposed weights: three instructions per value instead of five. Saturates through inf / 0 without a clamp.
DI void tanh16_hw(f32x16& v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[r]));
}
// derivative factor of the activation from what the forward pass keeps: h = tanh(a): 1 - h^2; math_mode fast keeps r: r - r^2
DI float dact(float h) {
    if constexpr (FAST) return FMA(-h, h, h);
    else return FMA(-h, h, 1.0f);
}
constexpr float TANH_PRESCALE = 2.885390043258667f;      // 2 log2 e (math_mode fast: folded into the layer-1 / layer-2 weights and biases)
template <bool PK>
DI void tanh_tile(f32x16& v) {
    if constexpr (FAST) tanh16_hw(v);
    else if constexpr (PK) tanh16_pk(v);
    else tanh16(v);
}
DI float sigmoid_spec(float x) {
    if constexpr (FAST) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950216293335f));
    float E = exp2_spec(clampf(x, -30.0f, 30.0f), -1.4426950216293335f);
    return rcp_spec(1.0f + E);
}

