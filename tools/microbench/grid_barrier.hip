// Micro-benchmark for the next round's "one-node encoder" question: what does a device-wide barrier INSIDE a persistent kernel cost on
// MI355X (8 XCDs, one L2 each), compared with the ~4.5 us a dependent kernel node costs in a hipGraph?  Every workgroup writes a
// value, all workgroups meet at a barrier built from one global atomic counter (agent-scope release / acquire fences, which is what
// makes the other XCDs' L2 see the writes), every workgroup reads its neighbour's value of THIS round and checks it.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier tools/microbench/grid_barrier.hip && ./grid_barrier [workgroups=256] [rounds=2000]
// The spin is bounded: a workgroup that waits more than ~0.5 s raises a flag and leaves (no hang on a box whose GPU is shared).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void barrier_kernel(unsigned* counter, unsigned* data, int rounds, unsigned* errors, unsigned* timeout) {
    const unsigned nwg = gridDim.x, wg = blockIdx.x;
    for (int r = 1; r <= rounds; ++r) {
        if (threadIdx.x == 0) data[wg] = (unsigned)r * 1000003u + wg;         // this round's payload
        __syncthreads();
        if (threadIdx.x == 0) {
            __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);                // agent scope by default for global atomics
            const unsigned target = nwg * (unsigned)r;
            unsigned spins = 0;
            while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 20000000u) { *timeout = 1u; break; }
            }
        }
        __syncthreads();
        if (*reinterpret_cast<volatile unsigned*>(timeout)) return;
        if (threadIdx.x == 0) {
            const unsigned nb = (wg + nwg / 2 + 1) % nwg;                     // a workgroup on (very likely) another XCD
            const unsigned v = __atomic_load_n(&data[nb], __ATOMIC_RELAXED);
            if (v != (unsigned)r * 1000003u + nb) atomicAdd(errors, 1u);
        }
        __syncthreads();
        // second barrier of the round (nobody may overwrite data[] before everybody has read it): same counter, next target
        if (threadIdx.x == 0) {
            __atomic_fetch_add(counter + 1, 1u, __ATOMIC_RELEASE);
            const unsigned target = nwg * (unsigned)r;
            unsigned spins = 0;
            while (__atomic_load_n(counter + 1, __ATOMIC_ACQUIRE) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 20000000u) { *timeout = 1u; break; }
            }
        }
        __syncthreads();
    }
}
__global__ void small_kernel(unsigned* data, int r) { data[blockIdx.x] = (unsigned)r + threadIdx.x; }

int main(int argc, char** argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 256, rounds = argc > 2 ? atoi(argv[2]) : 2000;
    unsigned *counter, *data, *errors, *timeout;
    hipMalloc(&counter, 8); hipMalloc(&data, nwg * 4); hipMalloc(&errors, 4); hipMalloc(&timeout, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(counter, 0, 8); hipMemset(errors, 0, 4); hipMemset(timeout, 0, 4);
        hipEventRecord(e0);
        hipLaunchKernelGGL(barrier_kernel, dim3(nwg), dim3(256), 0, 0, counter, data, rounds, errors, timeout);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned he, ht; hipMemcpy(&he, errors, 4, hipMemcpyDeviceToHost); hipMemcpy(&ht, timeout, 4, hipMemcpyDeviceToHost);
        printf("persistent kernel, %d workgroups: %.2f us per round of TWO grid barriers + one cross-workgroup hand-over (errors %u, timeout %u)\n",
               nwg, ms * 1000.f / rounds, he, ht);
    }
    // the alternative: a chain of dependent small kernels in a hipGraph
    hipStream_t s; hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int r = 0; r < 200; ++r) hipLaunchKernelGGL(small_kernel, dim3(nwg), dim3(256), 0, s, data, r);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int i = 0; i < 10; ++i) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("hipGraph chain of %d-workgroup store kernels: %.2f us per node\n", nwg, ms * 1000.f / 2000.f);
    return 0;
}
