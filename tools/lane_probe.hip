// Probe the lane-exchange primitives used for reductions against __shfl_xor.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__global__ void probe(float* out) {
    int l = threadIdx.x;
    float v = (float)(l * l + 1);
    auto r32 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    auto r16 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    out[0 * 64 + l] = __shfl_xor(v, 32);
    out[1 * 64 + l] = __builtin_bit_cast(float, r32[0]);
    out[2 * 64 + l] = __builtin_bit_cast(float, r32[1]);
    out[3 * 64 + l] = __shfl_xor(v, 16);
    out[4 * 64 + l] = __builtin_bit_cast(float, r16[0]);
    out[5 * 64 + l] = __builtin_bit_cast(float, r16[1]);
    out[6 * 64 + l] = __shfl_xor(v, 8);
    out[7 * 64 + l] = dpp_f<0x128>(v);
    out[8 * 64 + l] = dpp_f<0x124>(v);
    out[9 * 64 + l] = __shfl_xor(v, 2);
    out[10 * 64 + l] = dpp_f<0x4E>(v);
    out[11 * 64 + l] = __shfl_xor(v, 1);
    out[12 * 64 + l] = dpp_f<0xB1>(v);
}
int main() {
    float* d; hipMalloc(&d, 13 * 64 * 4); probe<<<1, 64>>>(d); float h[13 * 64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[13] = {"shfl_xor32", "pl32[0]", "pl32[1]", "shfl_xor16", "pl16[0]", "pl16[1]", "shfl_xor8", "row_ror8", "row_ror4", "shfl_xor2", "qp4E", "shfl_xor1", "qpB1"};
    for (int k = 0; k < 13; ++k) { printf("%-10s:", names[k]); for (int l = 0; l < 64; ++l) { int src = -1; for (int s = 0; s < 64; ++s) if (h[k * 64 + l] == (float)(s * s + 1)) src = s; printf(" %d", src); } printf("\n"); }
    return 0;
}
