// Shared device helpers for the gfx950 kernels (64-wide wavefronts throughout).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/grapes_hip.h"

#define GRAPES_WAVE 64

#define GRAPES_LAUNCH_CHECK()                      \
    do {                                           \
        hipError_t _e = hipGetLastError();         \
        if (_e != hipSuccess) return (int)_e;      \
    } while (0)

static inline int grapes_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Effective element count: device word (clamped to the capacity) or the host value.
__device__ __forceinline__ int eff_count(const int32_t* d_n, int n_host) {
    if (d_n == nullptr) return n_host;
    int v = *d_n;
    return v < n_host ? (v < 0 ? 0 : v) : n_host;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ int wave_incl_scan(int v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}

// Exclusive scan over the block (blockDim.x a multiple of 64, <= 1024).  `lds` needs 17 ints.
// Returns the exclusive prefix of v for this thread; *total = block sum.  Contains barriers:
// every thread of the block must call it.
__device__ __forceinline__ int block_excl_scan(int v, int* lds, int* total) {
    const int lane = lane_id();
    const int wid = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    int incl = wave_incl_scan(v);
    __syncthreads();  // protect lds reuse across consecutive calls
    if (lane == 63) lds[wid] = incl;
    __syncthreads();
    if (wid == 0) {
        int x = (lane < nw) ? lds[lane] : 0;
        int xs = wave_incl_scan(x);
        if (lane < nw) lds[lane] = xs - x;
        if (lane == nw - 1) lds[16] = xs;
    }
    __syncthreads();
    int base = lds[wid];
    *total = lds[16];
    return base + incl - v;
}
