// Micro-probe: what ONE dependent kernel node of a replayed hipGraph costs as a function of its launch shape — workgroup count,
// workgroup size, kernel-argument bytes, static LDS — when the kernel does next to nothing.  A chain of 256 such nodes is
// captured and replayed; time per node = replay time / 256.  (Round 2: why do ~2 us of work cost 10-25 us in the step?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Args64 { long long v[8]; };
struct Args512 { long long v[64]; };
template <int LDS>
__global__ void k_small(float* x, int n) {
    __shared__ float s[LDS > 0 ? LDS : 1];
    if (LDS > 0) { s[threadIdx.x % LDS] = 1.f; __syncthreads(); }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = x[i] + (LDS > 0 ? s[0] : 1.f);
}
template <typename A>
__global__ void k_args(float* x, int n, A a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    long long t = 0;
    for (unsigned q = 0; q < sizeof(A) / 8; ++q) t += a.v[q];
    if (i < n) x[i] = x[i] + (float)t;
}
__global__ void k_dep(float* x, const int* idx, int n) {       // two dependent global round trips (index -> value)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = x[idx[i]] + 1.f;
}
template <typename F>
static float chain(F launch, hipStream_t s, int nodes = 256, int reps = 20) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int k = 0; k < nodes; ++k) launch();
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1e3f / (nodes * reps);
}
int main() {
    hipStream_t s; hipStreamCreate(&s);
    float* x; int* idx; const int NMAX = 4096 * 1024;
    hipMalloc(&x, NMAX * 4); hipMalloc(&idx, NMAX * 4); hipMemset(x, 0, NMAX * 4);
    std::vector<int> h(NMAX); for (int i = 0; i < NMAX; ++i) h[i] = (int)((1103515245u * (unsigned)i + 12345u) % NMAX);
    hipMemcpy(idx, h.data(), NMAX * 4, hipMemcpyHostToDevice);
    printf("grid x block : empty | lds16k | lds64k | args64 | args512 | dependent-gather   (us per graph node)\n");
    for (int block : {256, 1024}) for (int grid : {1, 38, 160, 512, 2048, 8192}) {
        const int n = grid * block <= NMAX ? grid * block : NMAX;
        Args64 a64{}; Args512 a512{};
        const float t0 = chain([&] { hipLaunchKernelGGL(k_small<0>, dim3(grid), dim3(block), 0, s, x, n); }, s);
        const float t1 = chain([&] { hipLaunchKernelGGL(k_small<4096>, dim3(grid), dim3(block), 0, s, x, n); }, s);
        const float t2 = chain([&] { hipLaunchKernelGGL(k_small<16384>, dim3(grid), dim3(block), 0, s, x, n); }, s);
        const float t3 = chain([&] { hipLaunchKernelGGL(k_args<Args64>, dim3(grid), dim3(block), 0, s, x, n, a64); }, s);
        const float t4 = chain([&] { hipLaunchKernelGGL(k_args<Args512>, dim3(grid), dim3(block), 0, s, x, n, a512); }, s);
        const float t5 = chain([&] { hipLaunchKernelGGL(k_dep, dim3(grid), dim3(block), 0, s, x, idx, n); }, s);
        printf("%5d x %4d : %6.2f | %6.2f | %6.2f | %6.2f | %6.2f | %6.2f\n", grid, block, t0, t1, t2, t3, t4, t5);
    }
    return 0;
}
