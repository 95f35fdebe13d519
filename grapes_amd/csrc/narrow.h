// Shared by spmm_kernels.hip (gcn_aggregate_narrow_k) and sampler_kernels.hip (the aggregation that produces the inclusion
// logits, fused with the sampler's key computation).  Every product-sum below is an explicit fmaf of separately rounded
// factors, so the result does not depend on the translation unit's -ffp-contract setting.
#pragma once
#include "common.h"

// Narrow rows (F <= 16): one lane per destination row; rows longer than 64 entries are summed by
// the whole wavefront (lanes across entries, butterfly reduction), one such row at a time.
#define NARROW_HUGE 512      // longer rows (hub sources in the backward CSR) are reduced by the whole workgroup
// the 256 rows [bbase, bbase + 256) by the calling workgroup (256 threads; ends with a barrier)
__device__ __forceinline__ void narrow_block(const float* __restrict__ h,
                                                              const int32_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ csr,
                                                              const float* __restrict__ dinv,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int n, int F, int relu,
                                                              int narrow_lane_rows, int bbase) {
    __shared__ int s_huge[256];
    __shared__ int s_nhuge;
    __shared__ float s_red[4];
    const int lane = lane_id(), wid = threadIdx.x >> 6;
    {
        if (threadIdx.x == 0) s_nhuge = 0;
        __syncthreads();
        const int base = bbase + wid * 64;
        const int row = base + lane;
        int beg = 0, end = 0;
        float dc = 0.f;
        if (row < n) { beg = rowptr[row]; end = rowptr[row + 1]; dc = dinv[row]; }
        const int len = end - beg;
        const bool is_long = len > narrow_lane_rows;       // longer rows: 64 lanes side by side instead of one lane's serial walk
        if (row < n && !is_long) {
            for (int f = 0; f < F; ++f) {
                float acc = 0.f;
                for (int j = beg; j < end; ++j) {
                    const int s = csr[j];
                    acc = fmaf(dinv[s] * dc, h[(long long)s * F + f], acc);
                }
                float r = fmaf(dc * dc, h[(long long)row * F + f], acc);
                if (bias) r += bias[f];
                if (relu) r = fmaxf(r, 0.f);
                out[(long long)row * F + f] = r;
            }
        }
        if (len > NARROW_HUGE) s_huge[atomicAdd(&s_nhuge, 1)] = row;
        unsigned long long longs = __ballot(is_long && len <= NARROW_HUGE);
        while (longs) {                       // medium rows: one at a time by the whole wavefront
            const int l = __ffsll((long long)longs) - 1;
            longs &= longs - 1;
            const int lbeg = __shfl(beg, l, 64), lend = __shfl(end, l, 64);
            const float ldc = __shfl(dc, l, 64);
            const int lrow = base + l;
            for (int f = 0; f < F; ++f) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                int j = lbeg + lane;
                for (; j + 192 < lend; j += 256) {     // four independent gathers in flight per lane
                    const int s0 = csr[j], s1 = csr[j + 64], s2 = csr[j + 128], s3 = csr[j + 192];
                    a0 = fmaf(dinv[s0] * ldc, h[(long long)s0 * F + f], a0);
                    a1 = fmaf(dinv[s1] * ldc, h[(long long)s1 * F + f], a1);
                    a2 = fmaf(dinv[s2] * ldc, h[(long long)s2 * F + f], a2);
                    a3 = fmaf(dinv[s3] * ldc, h[(long long)s3 * F + f], a3);
                }
                for (; j < lend; j += 64) {
                    const int s = csr[j];
                    a0 = fmaf(dinv[s] * ldc, h[(long long)s * F + f], a0);
                }
                float acc = wave_sum((a0 + a1) + (a2 + a3));   // butterfly: fixed order
                if (lane == 0) {
                    float r = fmaf(ldc * ldc, h[(long long)lrow * F + f], acc);
                    if (bias) r += bias[f];
                    if (relu) r = fmaxf(r, 0.f);
                    out[(long long)lrow * F + f] = r;
                }
            }
        }
        __syncthreads();
        // huge rows: the whole workgroup per row (thread t takes entries t, t+256, ...; four gathers in flight), partial sums
        // combined in a fixed order (butterfly inside a wavefront, wavefronts in index order)
        const int nh = s_nhuge;
        for (int q = 0; q < nh; ++q) {
            const int hrow = s_huge[q];                    // list order varies run to run; each row's result does not depend on it
            const int hbeg = rowptr[hrow], hend = rowptr[hrow + 1];
            const float hdc = dinv[hrow];
            for (int f = 0; f < F; ++f) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                int j = hbeg + (int)threadIdx.x;
                for (; j + 768 < hend; j += 1024) {
                    const int s0 = csr[j], s1 = csr[j + 256], s2 = csr[j + 512], s3 = csr[j + 768];
                    a0 = fmaf(dinv[s0] * hdc, h[(long long)s0 * F + f], a0);
                    a1 = fmaf(dinv[s1] * hdc, h[(long long)s1 * F + f], a1);
                    a2 = fmaf(dinv[s2] * hdc, h[(long long)s2 * F + f], a2);
                    a3 = fmaf(dinv[s3] * hdc, h[(long long)s3 * F + f], a3);
                }
                for (; j < hend; j += 256) {
                    const int s = csr[j];
                    a0 = fmaf(dinv[s] * hdc, h[(long long)s * F + f], a0);
                }
                const float ws = wave_sum((a0 + a1) + (a2 + a3));
                __syncthreads();
                if (lane == 0) s_red[wid] = ws;
                __syncthreads();
                if (threadIdx.x == 0) {
                    float r = fmaf(hdc * hdc, h[(long long)hrow * F + f], (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
                    if (bias) r += bias[f];
                    if (relu) r = fmaxf(r, 0.f);
                    out[(long long)hrow * F + f] = r;
                }
            }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void narrow_body(const float* __restrict__ h, const int32_t* __restrict__ rowptr,
                                            const int32_t* __restrict__ csr, const float* __restrict__ dinv,
                                            const float* __restrict__ bias, float* __restrict__ out, int n_host,
                                            const int32_t* d_n, int F, int relu, int narrow_lane_rows, int bx, int gx) {
    const int n = eff_count(d_n, n_host);
    for (int bbase = bx * 256; bbase < n; bbase += gx * 256)   // uniform per workgroup
        narrow_block(h, rowptr, csr, dinv, bias, out, n, F, relu, narrow_lane_rows, bbase);
}
