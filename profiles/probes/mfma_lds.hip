// Micro-probe: v_mfma_f32_32x32x2_f32 rate when its operands come from ds_read_b64 (3 reads per 4 MFMAs, as in the
// W-stationary GEMM), 1 or 2 wavefronts per SIMD, with and without the LDS reads / with independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>   // 0: operands in registers; 1: operands re-read from LDS every quad; 2: LDS reads issued but MFMA uses registers
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    __shared__ float lds[32768];
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = (float)(i & 7);
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* Ap = lds + lane * 2;
    const float* Bp = lds + 8192 + (w & 3) * 128 + lane * 2;
    f32x16 c0 = {0}, c1 = {0};
    float2 a = *(const float2*)Ap, b0 = *(const float2*)Bp, b1 = *(const float2*)(Bp + 64);
    float sink = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int o = (it & 15) * 512;
        if (MODE >= 1) {
            float2 na = *(const float2*)(Ap + o), nb0 = *(const float2*)(Bp + o), nb1 = *(const float2*)(Bp + o + 64);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 1) { a = na; b0 = nb0; b1 = nb1; } else { sink += na.x + nb0.x + nb1.x; }
        }
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, c1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[0] + sink;
}
template <int MODE>
void run(int waves_per_simd, const char* name) {
    float* out; hipMalloc(&out, 256 * 1024 * sizeof(float));
    const int threads = 256 * waves_per_simd, iters = 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * 4 * waves_per_simd;
    printf("%-44s waves/SIMD=%d : %.1f ns per MFMA per SIMD, %.1f TFLOP/s\n", name, waves_per_simd, ms * 1e6 / mf,
           256.0 * 4 * mf * 4096 / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    run<0>(1, "operands in registers"); run<1>(1, "operands from ds_read_b64 each quad"); run<2>(1, "ds_read_b64 issued, operands in registers");
    run<0>(2, "operands in registers"); run<1>(2, "operands from ds_read_b64 each quad"); run<2>(2, "ds_read_b64 issued, operands in registers");
    return 0;
}
