#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned* out) {
    unsigned l = threadIdx.x;
    unsigned a = l, b = 100 + l;
    auto r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    auto r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[0 * 64 + l] = r32[0]; out[1 * 64 + l] = r32[1]; out[2 * 64 + l] = r16[0]; out[3 * 64 + l] = r16[1];
    unsigned c = l, d = l; asm volatile("" : "+v"(d));
    auto s32 = __builtin_amdgcn_permlane32_swap(c, d, false, false);
    out[4 * 64 + l] = s32[0]; out[5 * 64 + l] = s32[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 6 * 64 * 4); probe<<<1, 64>>>(d); unsigned h[6 * 64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[6] = {"pl32(a,b)[0]", "pl32(a,b)[1]", "pl16(a,b)[0]", "pl16(a,b)[1]", "pl32(c,c')[0]", "pl32(c,c')[1]"};
    for (int k = 0; k < 6; ++k) { printf("%-14s:", names[k]); for (int l = 0; l < 64; ++l) printf(" %u", h[k * 64 + l]); printf("\n"); }
    return 0;
}
