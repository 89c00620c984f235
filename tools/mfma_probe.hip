// Probe: is v_mfma_f32_32x32x2_f32 bit-for-bit a k-ordered fmaf chain (C first, then k0, k1 ...)?
// Also checks the "accumulator register r is the B operand of k-step r" chaining used by the MLP kernel.
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __host__ inline int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// D1 = W1[32xK1] * Z[K1x32] + C1 ; D2 = W2[32x32] * D1 (k permuted through accumulator layout) + C2
__global__ void probe(const float* W1, const float* Z, const float* C1, int K1,
                      const float* W2, const float* C2, float* D1, float* D2) {
    int l = threadIdx.x, j = l & 31, h = l >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = C1[rowmap(r, h) * 32 + j];
    for (int s = 0; s < K1 / 2; ++s) {
        float a = W1[j * K1 + 2 * s + h];      // A[i=l&31][k=2s+h]
        float b = Z[(2 * s + h) * 32 + j];     // B[k=2s+h][j=l&31]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) D1[rowmap(r, h) * 32 + j] = acc[r];
    f32x16 acc2;
    for (int r = 0; r < 16; ++r) acc2[r] = C2[rowmap(r, h) * 32 + j];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float a = W2[j * 32 + rowmap(r, h)];   // A[i][k = rowmap(r,h)]
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, acc[r], acc2, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) D2[rowmap(r, h) * 32 + j] = acc2[r];
}

int main() {
    const int K1 = 6;
    float W1[32 * K1], Z[K1 * 32], C1[1024], W2[1024], C2[1024], D1[1024], D2[1024], R1[1024], R2[1024];
    srand(7);
    auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto& x : W1) x = rnd(); for (auto& x : Z) x = rnd() * 3.f; for (auto& x : C1) x = rnd();
    for (auto& x : W2) x = rnd(); for (auto& x : C2) x = rnd();
    float *dW1, *dZ, *dC1, *dW2, *dC2, *dD1, *dD2;
    hipMalloc(&dW1, sizeof W1); hipMalloc(&dZ, sizeof Z); hipMalloc(&dC1, sizeof C1); hipMalloc(&dW2, sizeof W2);
    hipMalloc(&dC2, sizeof C2); hipMalloc(&dD1, sizeof D1); hipMalloc(&dD2, sizeof D2);
    hipMemcpy(dW1, W1, sizeof W1, hipMemcpyHostToDevice); hipMemcpy(dZ, Z, sizeof Z, hipMemcpyHostToDevice);
    hipMemcpy(dC1, C1, sizeof C1, hipMemcpyHostToDevice); hipMemcpy(dW2, W2, sizeof W2, hipMemcpyHostToDevice);
    hipMemcpy(dC2, C2, sizeof C2, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dW1, dZ, dC1, K1, dW2, dC2, dD1, dD2);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    hipMemcpy(D1, dD1, sizeof D1, hipMemcpyDeviceToHost); hipMemcpy(D2, dD2, sizeof D2, hipMemcpyDeviceToHost);
    // reference: k-ordered fmaf chains
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        float a = C1[i * 32 + j];
        for (int k = 0; k < K1; ++k) a = fmaf(W1[i * K1 + k], Z[k * 32 + j], a);
        R1[i * 32 + j] = a;
    }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        float a = C2[i * 32 + j];
        for (int r = 0; r < 16; ++r) for (int h = 0; h < 2; ++h) {
            int k = rowmap(r, h);
            a = fmaf(W2[i * 32 + k], R1[k * 32 + j], a);
        }
        R2[i * 32 + j] = a;
    }
    int bad1 = 0, bad2 = 0; double m1 = 0, m2 = 0;
    for (int i = 0; i < 1024; ++i) {
        if (memcmp(&D1[i], &R1[i], 4)) { ++bad1; m1 = fmax(m1, fabs(D1[i] - R1[i])); }
        if (memcmp(&D2[i], &R2[i], 4)) { ++bad2; m2 = fmax(m2, fabs(D2[i] - R2[i])); }
    }
    printf("layer1 bit-mismatches %d/1024 (max abs %g); layer2 bit-mismatches %d/1024 (max abs %g)\n", bad1, m1, bad2, m2);
    // alternate hypothesis: order k1 then k0 within an instruction
    int alt = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        float a = C1[i * 32 + j];
        for (int s = 0; s < K1 / 2; ++s) { a = fmaf(W1[i*K1+2*s+1], Z[(2*s+1)*32+j], a); a = fmaf(W1[i*K1+2*s], Z[(2*s)*32+j], a); }
        if (memcmp(&a, &D1[i * 32 + j], 4)) ++alt;
    }
    printf("alt-order(k1,k0) layer1 mismatches %d/1024\n", alt);
    return (bad1 || bad2) ? 1 : 0;
}
