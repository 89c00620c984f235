// transc_probe.hip — what do v_rcp_f32 / v_rsq_f32 / v_exp_f32 / v_log_f32 return? (SPEC.md §10a study)
// A tiny C-ABI library: evaluate one of the hardware transcendental instructions on a range of 32-bit input patterns or on an
// explicit input array, results left in device memory for the analysis scripts (torch tensors as device buffers, ctypes).
// Compiled with the product's flags (csrc/Makefile FLAGS), so the wave's FP mode register is the one the solve kernels run with.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int F>
__device__ __forceinline__ float hw(float x) {
    if constexpr (F == 0) return __builtin_amdgcn_rcpf(x);
    else if constexpr (F == 1) return __builtin_amdgcn_rsqf(x);
    else if constexpr (F == 2) return __builtin_amdgcn_exp2f(x);
    else return __builtin_amdgcn_logf(x);
}

template <int F>
__global__ void range_kernel(uint64_t start, uint64_t stride, uint64_t n, uint32_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t bits = (uint32_t)(start + i * stride);
    out[i] = __float_as_uint(hw<F>(__uint_as_float(bits)));
}

template <int F>
__global__ void array_kernel(const uint32_t* __restrict__ in, uint64_t n, uint32_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = __float_as_uint(hw<F>(__uint_as_float(in[i])));
}

// SPEC.md §10c: the two-limb binary16 split of a pair of values as the kernels write it — v_cvt_pk_f16_f32 (round and pack), two v_fma_mix_f32 (the residuals
// x - (float)limb straight from the packed halves), v_cvt_pk_f16_f32 again; out[2 i] = the packed first limbs of (in[2 i], in[2 i + 1]), out[2 i + 1] the second limbs
__global__ void split2h_kernel(const uint32_t* __restrict__ in, uint64_t npairs, uint32_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npairs) return;
    const float x = __uint_as_float(in[2 * i]), y = __uint_as_float(in[2 * i + 1]);
    unsigned p1, p2; float xr, yr;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p1) : "v"(x), "v"(y));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(xr) : "v"(p1), "v"(x));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(yr) : "v"(p1), "v"(y));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p2) : "v"(xr), "v"(yr));
    out[2 * i] = p1; out[2 * i + 1] = p2;
}

// the wave's MODE register (FP round / denorm fields), as the kernels see it
__global__ void mode_kernel(uint32_t* out) {
    uint32_t m;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_MODE)" : "=s"(m));
    out[0] = m;
}

extern "C" {

// out[i] = f(bits = start + i * stride), i < n   (func: 0 rcp, 1 rsq, 2 exp2, 3 log2)
int transc_eval_range(int func, uint64_t start, uint64_t stride, uint64_t n, uint32_t* out_dev) {
    if (n == 0) return 0;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    switch (func) {
        case 0: range_kernel<0><<<blocks, 256>>>(start, stride, n, out_dev); break;
        case 1: range_kernel<1><<<blocks, 256>>>(start, stride, n, out_dev); break;
        case 2: range_kernel<2><<<blocks, 256>>>(start, stride, n, out_dev); break;
        case 3: range_kernel<3><<<blocks, 256>>>(start, stride, n, out_dev); break;
        default: return -1;
    }
    return (int)hipDeviceSynchronize();
}

int transc_split2h(const uint32_t* in_dev, uint64_t npairs, uint32_t* out_dev) {
    if (npairs == 0) return 0;
    split2h_kernel<<<(unsigned)((npairs + 255) / 256), 256>>>(in_dev, npairs, out_dev);
    return (int)hipDeviceSynchronize();
}

int transc_eval_array(int func, const uint32_t* in_dev, uint64_t n, uint32_t* out_dev) {
    if (n == 0) return 0;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    switch (func) {
        case 0: array_kernel<0><<<blocks, 256>>>(in_dev, n, out_dev); break;
        case 1: array_kernel<1><<<blocks, 256>>>(in_dev, n, out_dev); break;
        case 2: array_kernel<2><<<blocks, 256>>>(in_dev, n, out_dev); break;
        case 3: array_kernel<3><<<blocks, 256>>>(in_dev, n, out_dev); break;
        default: return -1;
    }
    return (int)hipDeviceSynchronize();
}

int transc_mode(uint32_t* out_dev) {
    mode_kernel<<<1, 1>>>(out_dev);
    return (int)hipDeviceSynchronize();
}

}  // extern "C"
