// Micro-probe: issue rate of v_mfma_f32_32x32x2_f32 with NACC independent accumulators per wave and W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(1024) void probe(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int waves_per_simd) {
    float* out; hipMalloc(&out, 256 * 1024 * sizeof(float) * 4);
    const int threads = 64 * 4 * waves_per_simd;          // one workgroup per CU
    const int iters = 4096 / NACC;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<NACC>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0f, 2.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * NACC * waves_per_simd;
    const double flops = 256.0 * 4 * mfma_per_simd * 4096;
    printf("NACC=%d waves/SIMD=%d : %.1f us, %.1f ns per MFMA per SIMD, %.1f TFLOP/s\n", NACC, waves_per_simd, ms * 1e3,
           ms * 1e6 / mfma_per_simd, flops / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    run<1>(1); run<2>(1); run<4>(1); run<8>(1); run<1>(2); run<2>(2); run<4>(2); run<2>(4); run<4>(4);
    return 0;
}
