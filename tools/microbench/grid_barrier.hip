// What does a dependent step cost inside ONE persistent kernel (device-wide barrier: agent-scope release / acquire on a counter) against
// a dependent kernel node of a hipGraph?  The encoder + heads chain of the step is ~70 kernels of 4.7-10 us on 287 x 256 tensors: if a
// device-wide barrier costs well under the ~4.7 us a trivial node takes, one persistent kernel for the chain is the way to shrink it.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier tools/microbench/grid_barrier.hip && ./grid_barrier
// Each step: every workgroup reads 1 KB per thread-block written by ANOTHER workgroup in the previous step (so the barrier must really
// publish data across XCDs), adds one, writes its own 1 KB, then the barrier.  The result is checked.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);                       // agent scope by default for global atomics in HIP
        while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void persistent(float* buf0, float* buf1, unsigned* counter, int steps) {
    const int g = gridDim.x, b = blockIdx.x;
    float* src = buf0; float* dst = buf1;
    for (int s = 0; s < steps; ++s) {
        const int from = (b + 1 + s) % g;                                       // somebody else's block of the previous step
        const float v = __builtin_nontemporal_load(src + from * 256 + threadIdx.x);
        dst[b * 256 + threadIdx.x] = v + 1.0f;
        __threadfence();
        grid_barrier(counter, (unsigned)(s + 1) * g);
        float* t = src; src = dst; dst = t;
    }
}

__global__ __launch_bounds__(256) void one_step(const float* src, float* dst, int s) {
    const int g = gridDim.x, b = blockIdx.x;
    const int from = (b + 1 + s) % g;
    dst[b * 256 + threadIdx.x] = src[from * 256 + threadIdx.x] + 1.0f;
}

int main() {
    const int steps = 200;
    float *b0, *b1; unsigned* counter;
    hipMalloc(&b0, 1024 * 256 * 4); hipMalloc(&b1, 1024 * 256 * 4); hipMalloc(&counter, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int g : {36, 72, 144, 256, 512}) {
        float best = 1e30f; bool ok = true;
        for (int rep = 0; rep < 4; ++rep) {
            hipMemset(b0, 0, 1024 * 256 * 4); hipMemset(b1, 0, 1024 * 256 * 4); hipMemset(counter, 0, 4);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(persistent, dim3(g), dim3(256), 0, 0, b0, b1, counter, steps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            std::vector<float> h(g * 256);
            hipMemcpy(h.data(), (steps % 2) ? b1 : b0, g * 256 * 4, hipMemcpyDeviceToHost);
            for (float v : h) ok = ok && v == (float)steps;
        }
        printf("persistent kernel, %3d workgroups: %6.2f us per step (device-wide barrier + 1 KB exchange), result %s\n", g, best * 1e3 / steps, ok ? "ok" : "WRONG");
    }
    // the same chain as dependent kernel nodes of one hipGraph
    hipStream_t st; hipStreamCreate(&st);
    for (int g : {72, 256}) {
        hipGraph_t graph; hipGraphExec_t exec;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int s = 0; s < steps; ++s) hipLaunchKernelGGL(one_step, dim3(g), dim3(256), 0, st, (s % 2) ? b1 : b0, (s % 2) ? b0 : b1, s);
        hipStreamEndCapture(st, &graph);
        hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, st); hipGraphLaunch(exec, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("hipGraph of %d dependent kernels, %3d workgroups: %6.2f us per node\n", steps, g, best * 1e3 / steps);
    }
    return 0;
}
