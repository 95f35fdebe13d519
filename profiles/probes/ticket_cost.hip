// Micro-probe: cost of the "last workgroup" ticket (device-scope fence + same-address atomic) per workgroup count.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void work(float* x, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] = x[i] * 1.0001f + 1.f; }
__global__ void work_ticket(float* x, int n, unsigned* ticket, float* out, int fence) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] = x[i] * 1.0001f + 1.f;
    __shared__ int last;
    if (threadIdx.x == 0) { if (fence) __threadfence(); last = atomicAdd(ticket, 1u) == gridDim.x - 1; }
    __syncthreads();
    if (last && threadIdx.x == 0) { if (fence) __threadfence(); out[0] = x[0]; *ticket = 0; }
}
int main() {
    float* x; unsigned* t; float* o; hipMalloc(&x, 1 << 24); hipMalloc(&t, 64); hipMalloc(&o, 64); hipMemset(t, 0, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {32, 128, 512, 2048}) {
        const int n = blocks * 256;
        float ms[3];
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                for (int k = 0; k < 50; ++k) {
                    if (mode == 0) hipLaunchKernelGGL(work, dim3(blocks), dim3(256), 0, 0, x, n);
                    else hipLaunchKernelGGL(work_ticket, dim3(blocks), dim3(256), 0, 0, x, n, t, o, mode == 2);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            hipEventElapsedTime(&ms[mode], e0, e1);
        }
        printf("blocks=%4d : plain %.2f us, ticket %.2f us, ticket+fence %.2f us per launch\n", blocks, ms[0] * 20, ms[1] * 20, ms[2] * 20);
    }
    return 0;
}
