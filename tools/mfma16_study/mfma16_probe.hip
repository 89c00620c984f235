// mfma16_probe — dump D = A·B + C of the 16-bit-operand matrix instructions of gfx950 for offline analysis
// (tools/mfma16_study/fit_model.py).  One workgroup (one wave) per 32x32 tile; operands as raw bit patterns.
//   in : int32 {magic 0x4D464D41, ntiles, dtype (0 = f16, 1 = bf16), reserved}, then per tile A u16[32][16] (row i, k),
//        B u16[16][32] (k, column j), C f32[32][32] (i, j)
//   out: D f32[32][32] per tile
// build: hipcc --offload-arch=gfx950 -O2 -o tools/mfma16_study/mfma16_probe tools/mfma16_study/mfma16_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short short8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// dtype 2: v_mfma_f32_32x32x8_bf16_1k on the k < 8 part of a bf16 case file (lane half h holds k = 4h .. 4h+3)
__global__ void k8(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, const float* __restrict__ C, float* __restrict__ D) {
    const size_t t = blockIdx.x;
    A += t * 512; B += t * 512; C += t * 1024; D += t * 1024;
    const int l = threadIdx.x, j = l & 31, h = l >> 5;
    s16x4 a, b;
    for (int e = 0; e < 4; ++e) { a[e] = (short)A[j * 16 + 4 * h + e]; b[e] = (short)B[(4 * h + e) * 32 + j]; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j];
    acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = acc[r];
}

template <int DT>
__global__ void k(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, const float* __restrict__ C, float* __restrict__ D) {
    const size_t t = blockIdx.x;
    A += t * 512; B += t * 512; C += t * 1024; D += t * 1024;
    const int l = threadIdx.x, j = l & 31, h = l >> 5;
    short8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (short)A[j * 16 + 8 * h + e]; b[e] = (short)B[(8 * h + e) * 32 + j]; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j];
    if (DT == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), acc, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = acc[r];
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s in.bin out.bin [8]\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb"); if (!f) { perror(argv[1]); return 1; }
    int32_t hdr[4]; if (fread(hdr, 4, 4, f) != 4 || hdr[0] != 0x4D464D41) { fprintf(stderr, "bad header\n"); return 1; }
    const size_t nt = hdr[1]; const int dt = hdr[2];
    std::vector<uint16_t> A(nt * 512), B(nt * 512); std::vector<float> C(nt * 1024), D(nt * 1024);
    for (size_t t = 0; t < nt; ++t) {
        if (fread(&A[t * 512], 2, 512, f) != 512 || fread(&B[t * 512], 2, 512, f) != 512 || fread(&C[t * 1024], 4, 1024, f) != 1024) { fprintf(stderr, "short file\n"); return 1; }
    }
    fclose(f);
    uint16_t *dA, *dB; float *dC, *dD;
    CK(hipMalloc(&dA, nt * 1024)); CK(hipMalloc(&dB, nt * 1024)); CK(hipMalloc(&dC, nt * 4096)); CK(hipMalloc(&dD, nt * 4096));
    CK(hipMemcpy(dA, A.data(), nt * 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), nt * 1024, hipMemcpyHostToDevice));
    CK(hipMemcpy(dC, C.data(), nt * 4096, hipMemcpyHostToDevice));
    const bool k8mode = argc > 3 && atoi(argv[3]) == 8;      // third argument 8: the K = 8 instruction on a bf16 file
    if (k8mode) k8<<<dim3((unsigned)nt), dim3(64)>>>(dA, dB, dC, dD);
    else if (dt == 0) k<0><<<dim3((unsigned)nt), dim3(64)>>>(dA, dB, dC, dD); else k<1><<<dim3((unsigned)nt), dim3(64)>>>(dA, dB, dC, dD);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, nt * 4096, hipMemcpyDeviceToHost));
    FILE* g = fopen(argv[2], "wb"); if (!g) { perror(argv[2]); return 1; }
    fwrite(D.data(), 4, nt * 1024, g); fclose(g);
    printf("%s: %zu tiles, dtype %s\n", argv[1], nt, dt ? "bf16" : "f16");
    return 0;
}
