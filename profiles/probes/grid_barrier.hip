// Micro-probe (round 3): what does a GRID BARRIER inside one launch cost on MI355X, against the ~4.5 us a dependent launch
// costs?  (1) K barriers in a kernel of G workgroups x B threads -> us per barrier; (2) a dependent chain of loads through
// memory another workgroup wrote in the previous phase: plain vs agent-scope (sc1) loads, L2-warm vs produced remotely.
// Decides whether "several phases in one cooperative launch" (prep_fused_k) can beat separate launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void gbar(unsigned* bar, unsigned phase, int sleep) {
    __syncthreads();
    if (threadIdx.x == 0) {
        (void)atomicAdd(bar, 1u);
        const unsigned target = phase * gridDim.x;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) { if (sleep) __builtin_amdgcn_s_sleep(1); }
    }
    __syncthreads();
}
__global__ void k_barriers(unsigned* bar, int K, int sleep, float* x) {
    for (int k = 1; k <= K; ++k) gbar(bar, (unsigned)k, sleep);
    if (threadIdx.x == 0 && blockIdx.x == 0) { x[0] += 1.f; }
    // reset by the last to leave
    if (threadIdx.x == 0 && atomicAdd(bar + 1, 1u) == gridDim.x - 1) { atomicExch(bar, 0u); atomicExch(bar + 1, 0u); }
}
// phases: each phase, thread i writes y[i] = f(x[perm[i]]) where x was written by the previous phase (by another workgroup)
template <int AGENT>
__global__ void k_phases(unsigned* bar, int K, const int* __restrict__ perm, int* a, int* b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int* src = a; int* dst = b;
    for (int k = 1; k <= K; ++k) {
        if (i < n) {
            const int j = perm[i];
            const int v = AGENT ? __hip_atomic_load(&src[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : src[j];
            if (AGENT) __hip_atomic_store(&dst[i], v + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else dst[i] = v + 1;
        }
        gbar(bar, (unsigned)k, 1);
        int* t = src; src = dst; dst = t;
    }
    if (threadIdx.x == 0 && atomicAdd(bar + 1, 1u) == gridDim.x - 1) { atomicExch(bar, 0u); atomicExch(bar + 1, 0u); }
}
__global__ void k_phase1(const int* __restrict__ perm, const int* __restrict__ src, int* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]] + 1;
}
template <typename F>
static float time_us(F f, hipStream_t s, int reps = 50) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r) f();
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}
int main() {
    hipStream_t s; hipStreamCreate(&s);
    unsigned* bar; hipMalloc(&bar, 256); hipMemset(bar, 0, 256);
    float* x; hipMalloc(&x, 4096); hipMemset(x, 0, 4096);
    const int NMAX = 256 * 1024;
    int *perm, *a, *b; hipMalloc(&perm, NMAX * 4); hipMalloc(&a, NMAX * 4); hipMalloc(&b, NMAX * 4);
    hipMemset(a, 0, NMAX * 4); hipMemset(b, 0, NMAX * 4);
    printf("grid x block : us per barrier (K=1 launch cost subtracted: (t(K=33)-t(K=1))/32), sleep / no sleep\n");
    for (int block : {256, 1024}) for (int grid : {8, 32, 64, 128, 256}) {
        float r[2];
        for (int sl = 0; sl < 2; ++sl) {
            const float t1 = time_us([&] { hipLaunchKernelGGL(k_barriers, dim3(grid), dim3(block), 0, s, bar, 1, sl, x); }, s);
            const float t33 = time_us([&] { hipLaunchKernelGGL(k_barriers, dim3(grid), dim3(block), 0, s, bar, 33, sl, x); }, s);
            r[sl] = (t33 - t1) / 32.f;
            if (sl == 1) printf("%4d x %4d : %6.2f | %6.2f   (launch with one barrier: %.2f us)\n", grid, block, r[1], r[0], t1);
        }
    }
    printf("\nphases of dependent gathers through data the previous phase wrote (n = grid x block), us per phase incl. its barrier\n");
    printf("grid x block : plain ld/st | agent ld/st | separate launches (eager, per launch)\n");
    for (int block : {256, 1024}) for (int grid : {32, 128, 256}) {
        const int n = grid * block;
        std::vector<int> h(n); for (int i = 0; i < n; ++i) h[i] = (int)((1103515245u * (unsigned)i + 12345u) % (unsigned)n);
        hipMemcpy(perm, h.data(), n * 4, hipMemcpyHostToDevice);
        const float p1 = time_us([&] { hipLaunchKernelGGL(k_phases<0>, dim3(grid), dim3(block), 0, s, bar, 1, perm, a, b, n); }, s);
        const float p17 = time_us([&] { hipLaunchKernelGGL(k_phases<0>, dim3(grid), dim3(block), 0, s, bar, 17, perm, a, b, n); }, s);
        const float q1 = time_us([&] { hipLaunchKernelGGL(k_phases<1>, dim3(grid), dim3(block), 0, s, bar, 1, perm, a, b, n); }, s);
        const float q17 = time_us([&] { hipLaunchKernelGGL(k_phases<1>, dim3(grid), dim3(block), 0, s, bar, 17, perm, a, b, n); }, s);
        // 16 dependent launches in a captured graph
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int k = 0; k < 16; ++k) hipLaunchKernelGGL(k_phase1, dim3(grid), dim3(block), 0, s, perm, (k & 1) ? b : a, (k & 1) ? a : b, n);
        hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        const float l16 = time_us([&] { hipGraphLaunch(ge, s); }, s);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
        printf("%4d x %4d : %6.2f | %6.2f | %6.2f\n", grid, block, (p17 - p1) / 16.f, (q17 - q1) / 16.f, l16 / 16.f);
    }
    return 0;
}
