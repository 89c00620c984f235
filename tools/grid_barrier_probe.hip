// Cost of the bounded grid barrier of the cooperative latency path (one arrival counter, thread 0 of every workgroup polls):
// N workgroups x 256 threads, R barriers back to back; variants: agent-scope fences around the counter, or none (sc1 data accesses).
#include <hip/hip_runtime.h>
#include <cstdio>
template <bool FENCE>
__global__ void __launch_bounds__(256) k(unsigned* bar, int nwg, int rounds, float* sink) {
    unsigned epoch = 0;
    float acc = threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
        acc = acc * 1.0001f + 1.0f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        epoch += 1;
        if (threadIdx.x == 0) {
            if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = epoch * (unsigned)nwg;
            unsigned spins = 0;
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 40000000u) break;
            }
            if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
    unsigned* bar; float* sink; (void)hipMalloc(&bar, 4); (void)hipMalloc(&sink, 256 * 256 * 4);
    const int rounds = 2000;
    for (int nwg : {1, 8, 32, 64, 128, 256})
        for (int fence = 0; fence < 2; ++fence) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipMemset(bar, 0, 4);
                (void)hipEventRecord(e0);
                if (fence) k<true><<<nwg, 256>>>(bar, nwg, rounds, sink); else k<false><<<nwg, 256>>>(bar, nwg, rounds, sink);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            }
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("workgroups %3d  %-22s %.2f us per barrier\n", nwg, fence ? "agent-scope fences" : "no fences (sc1 data)", ms * 1e3 / rounds);
        }
    return 0;
}
