// Dump D = A(32x16 f16) * B(16x32 f16) + C(32x32 f32) from v_mfma_f32_32x32x16_f16 for host-side analysis.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const _Float16* A, const _Float16* B, const float* C, float* D) {   // A[32][16], B[16][32], C/D[32][32]
    int l = threadIdx.x, j = l & 31, h = l >> 5;
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = A[j * 16 + 8 * h + e]; b[e] = B[(8 * h + e) * 32 + j]; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = acc[r];
}
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); FILE* g = fopen(argv[2], "wb");
    int ncase; fread(&ncase, 4, 1, f);
    _Float16 *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, 512 * 2); hipMalloc(&dB, 512 * 2); hipMalloc(&dC, 4096); hipMalloc(&dD, 4096);
    for (int c = 0; c < ncase; ++c) {
        uint16_t A[512], B[512]; float C[1024], D[1024];
        fread(A, 2, 512, f); fread(B, 2, 512, f); fread(C, 4, 1024, f);
        hipMemcpy(dA, A, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B, 1024, hipMemcpyHostToDevice); hipMemcpy(dC, C, 4096, hipMemcpyHostToDevice);
        k<<<1, 64>>>(dA, dB, dC, dD); hipMemcpy(D, dD, 4096, hipMemcpyDeviceToHost);
        fwrite(D, 4, 1024, g);
    }
    fclose(f); fclose(g); printf("wrote %d cases\n", ncase); return 0;
}
