// A7 (sparse part): the GCNConv aggregation  out = Â H (+ bias, ReLU)  and its transpose for the
// backward pass, as a gather-SpMM over the per-hop CSR (HBM-bound: 4·F+4 bytes per aggregated edge).
//
//   gcn_aggregate_k<VEC>      one wavefront per destination row; a lane owns VEC consecutive features
//                             (VEC=4: one dwordx4 per lane = a whole 1 KiB row of 256 fp32 per
//                             wave-instruction); indices/weights are wave-uniform scalar loads; four
//                             neighbour rows in flight per wave.
//   gcn_aggregate_long_k      rows longer than GRAPES_LONG_ROW (hub nodes: 10^3..10^4 entries in the
//                             backward CSR): one 512-thread workgroup per row, the 8 wavefronts split the
//                             entries, partial sums meet in LDS and are combined in a fixed order.
//   gcn_aggregate_narrow_k    F <= 16 (the 1-wide logit heads): lanes run across ROWS, and rows
//                             longer than 64 entries are reduced by the whole wavefront
//                             (segmented wave reduce).
//   colsum_*                  bias gradient / ReLU backward / 1-wide dW: deterministic two-stage sums.
// Weights follow PyG: w_rc = dinv[r]·dinv[c] per edge, the unit self-loop (weight dinv[c]²) is added
// last, then the bias.
#include "common.h"

template <int VEC>
__device__ __forceinline__ void ld_vec(const float* __restrict__ p, float (&v)[VEC]) {
    if (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1 % VEC] = t.y; v[2 % VEC] = t.z; v[3 % VEC] = t.w;
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = p[i];
    }
}

// accumulate entries [beg,end) of one CSR row into acc (features f0..f0+VEC)
template <int VEC>
__device__ __forceinline__ void row_accumulate(const float* __restrict__ h, const int32_t* __restrict__ csr,
                                               const float* __restrict__ dinv, int beg, int end, float dc, int F,
                                               int f0, float (&acc)[VEC]) {
    int j = beg;
    for (; j + 4 <= end; j += 4) {
        int s[4]; float w[4]; float val[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; ++u) { s[u] = csr[j + u]; w[u] = dinv[s[u]] * dc; }
#pragma unroll
        for (int u = 0; u < 4; ++u) ld_vec<VEC>(h + (long long)s[u] * F + f0, val[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w[u], val[u][v], acc[v]);
    }
    for (; j < end; ++j) {
        const int s = csr[j];
        const float w = dinv[s] * dc;
        float val[VEC];
        ld_vec<VEC>(h + (long long)s * F + f0, val);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w, val[v], acc[v]);
    }
}

// self-loop + bias + ReLU + store
template <int VEC>
__device__ __forceinline__ void row_finish(const float* __restrict__ h, const float* __restrict__ bias,
                                           float* __restrict__ out, int row, float dc, int F, int f0, int relu,
                                           float (&acc)[VEC]) {
    const float w = dc * dc;
    float self[VEC];
    ld_vec<VEC>(h + (long long)row * F + f0, self);
    float r[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        r[v] = fmaf(w, self[v], acc[v]);
        if (bias) r[v] += bias[f0 + v];
        if (relu) r[v] = fmaxf(r[v], 0.f);
    }
    float* o = out + (long long)row * F + f0;
    if (VEC == 4) {
        *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1 % VEC], r[2 % VEC], r[3 % VEC]);
    } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) o[v] = r[v];
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void gcn_aggregate_k(const float* __restrict__ h, const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ csr, const float* __restrict__ dinv,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       int n_host, const int32_t* d_n, int F, int relu, int skip_long) {
    const int n = eff_count(d_n, n_host);
    const int lane = lane_id();
    const int wave_global = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int row = wave_global; row < n; row += nwaves) {
        const int beg = rowptr[row], end = rowptr[row + 1];
        if (skip_long && end - beg > GRAPES_LONG_ROW) continue;   // handled by gcn_aggregate_long_k
        const float dc = dinv[row];
        for (int f0 = lane * VEC; f0 < F; f0 += 64 * VEC) {
            float acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
            row_accumulate<VEC>(h, csr, dinv, beg, end, dc, F, f0, acc);
            row_finish<VEC>(h, bias, out, row, dc, F, f0, relu, acc);
        }
    }
}

#define LONG_WAVES 8
template <int VEC>
__global__ __launch_bounds__(512) void gcn_aggregate_long_k(const float* __restrict__ h, const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ csr, const float* __restrict__ dinv,
                                                            const float* __restrict__ bias, float* __restrict__ out, int F,
                                                            int relu, const int32_t* __restrict__ long_rows,
                                                            const int32_t* __restrict__ d_n_long) {
    __shared__ float part[LONG_WAVES][64 * VEC];
    const int n_long = *d_n_long;
    const int lane = lane_id(), wid = threadIdx.x >> 6;
    for (int li = blockIdx.x; li < n_long; li += gridDim.x) {
        const int row = long_rows[li];
        const int beg = rowptr[row], end = rowptr[row + 1];
        const float dc = dinv[row];
        const int per = (end - beg + LONG_WAVES - 1) / LONG_WAVES;
        const int wb = beg + wid * per;
        const int we = wb + per < end ? wb + per : end;
        for (int fbase = 0; fbase < F; fbase += 64 * VEC) {
            const int f0 = fbase + lane * VEC;
            float acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
            if (f0 < F && wb < we) row_accumulate<VEC>(h, csr, dinv, wb, we, dc, F, f0, acc);
#pragma unroll
            for (int v = 0; v < VEC; ++v) part[wid][lane * VEC + v] = acc[v];
            __syncthreads();
            if (wid == 0 && f0 < F) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    float t = part[0][lane * VEC + v];
                    for (int w = 1; w < LONG_WAVES; ++w) t += part[w][lane * VEC + v];   // fixed order
                    acc[v] = t;
                }
                row_finish<VEC>(h, bias, out, row, dc, F, f0, relu, acc);
            }
            __syncthreads();
        }
    }
}

// Narrow rows (F <= 16): one lane per destination row; rows longer than 64 entries are summed by
// the whole wavefront (lanes across entries, wave reduction), one such row at a time.
__global__ __launch_bounds__(256) void gcn_aggregate_narrow_k(const float* __restrict__ h,
                                                              const int32_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ csr,
                                                              const float* __restrict__ dinv,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int n_host, const int32_t* d_n, int F, int relu) {
    const int n = eff_count(d_n, n_host);
    const int lane = lane_id();
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int base = wave_global * 64; base < n; base += nwaves * 64) {
        const int row = base + lane;
        int beg = 0, end = 0;
        float dc = 0.f;
        if (row < n) { beg = rowptr[row]; end = rowptr[row + 1]; dc = dinv[row]; }
        const bool is_long = (end - beg) > 64;
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
        unsigned long long longs = __ballot(is_long);
        while (longs) {
            const int l = __ffsll((long long)longs) - 1;
            longs &= longs - 1;
            const int lbeg = __shfl(beg, l, 64), lend = __shfl(end, l, 64);
            const float ldc = __shfl(dc, l, 64);
            const int lrow = base + l;
            for (int f = 0; f < F; ++f) {
                float acc = 0.f;
                for (int j = lbeg + lane; j < lend; j += 64) {
                    const int s = csr[j];
                    acc = fmaf(dinv[s] * ldc, h[(long long)s * F + f], acc);
                }
                acc = wave_sum(acc);   // xor-butterfly: every lane ends with the same, order-fixed sum
                if (lane == 0) {
                    float r = fmaf(ldc * ldc, h[(long long)lrow * F + f], acc);
                    if (bias) r += bias[f];
                    if (relu) r = fmaxf(r, 0.f);
                    out[(long long)lrow * F + f] = r;
                }
            }
        }
    }
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

static int launch_aggregate(const float* h, const int32_t* rowptr, const int32_t* csr, const float* dinv,
                            const float* bias, float* out, int n, const int32_t* d_n, int f, int relu,
                            const int32_t* long_rows, const int32_t* d_n_long, hipStream_t s) {
    if (f <= 16) {
        int grid = grapes_div_up(n, 256); if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(gcn_aggregate_narrow_k, dim3(grid), dim3(256), 0, s, h, rowptr, csr, dinv, bias, out, n, d_n, f, relu);
        GRAPES_LAUNCH_CHECK();
        return 0;
    }
    int grid = grapes_div_up(n, 4); if (grid > 16384) grid = 16384;
    const bool vec = (f % 4 == 0) && aligned16(h) && aligned16(out) && (!bias || aligned16(bias));
    const int skip = (long_rows && d_n_long) ? 1 : 0;
    if (vec)
        hipLaunchKernelGGL((gcn_aggregate_k<4>), dim3(grid), dim3(256), 0, s, h, rowptr, csr, dinv, bias, out, n, d_n, f, relu, skip);
    else
        hipLaunchKernelGGL((gcn_aggregate_k<1>), dim3(grid), dim3(256), 0, s, h, rowptr, csr, dinv, bias, out, n, d_n, f, relu, skip);
    GRAPES_LAUNCH_CHECK();
    if (skip) {
        if (vec)
            hipLaunchKernelGGL((gcn_aggregate_long_k<4>), dim3(128), dim3(512), 0, s, h, rowptr, csr, dinv, bias, out, f, relu, long_rows, d_n_long);
        else
            hipLaunchKernelGGL((gcn_aggregate_long_k<1>), dim3(128), dim3(512), 0, s, h, rowptr, csr, dinv, bias, out, f, relu, long_rows, d_n_long);
        GRAPES_LAUNCH_CHECK();
    }
    return 0;
}

// ============================================================================ column sums
// out[c] (+)= sum_r wrow[r] * val(r,c);  val = src[r][c], optionally gated by (gate[r][c] > 0) (ReLU
// backward); the gated values are optionally written to dst (dpre).  Stage 1: CS_BLOCKS workgroups,
// each owning a fixed, strided set of 64-row chunks; stage 2: fixed-order combine.  Deterministic.
#define CS_BLOCKS 512
#define CS_ROWS 32
__global__ __launch_bounds__(256) void colsum_partial_k(const float* __restrict__ src, const float* __restrict__ gate,
                                                        const float* __restrict__ wrow, float* __restrict__ dst,
                                                        float* __restrict__ partial, int n_host, const int32_t* d_n,
                                                        int F) {
    const int n = eff_count(d_n, n_host);
    for (int c = threadIdx.x; c < F; c += blockDim.x) {
        float acc = 0.f;
        for (int r0 = blockIdx.x * CS_ROWS; r0 < n; r0 += CS_BLOCKS * CS_ROWS) {
            const int r1 = r0 + CS_ROWS < n ? r0 + CS_ROWS : n;
            for (int r = r0; r < r1; ++r) {
                const long long o = (long long)r * F + c;
                float v = src[o];
                if (gate) v = gate[o] > 0.f ? v : 0.f;
                if (dst) dst[o] = v;
                acc += wrow ? wrow[r] * v : v;
            }
        }
        partial[(long long)blockIdx.x * F + c] = acc;
    }
}

// narrow F (<= 16): threads run across rows, block tree-reduce per column
__global__ __launch_bounds__(256) void colsum_partial_narrow_k(const float* __restrict__ src, const float* __restrict__ gate,
                                                               const float* __restrict__ wrow, float* __restrict__ dst,
                                                               float* __restrict__ partial, int n_host,
                                                               const int32_t* d_n, int F) {
    __shared__ float red[4];
    const int n = eff_count(d_n, n_host);
    for (int c = 0; c < F; ++c) {
        float acc = 0.f;
        for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += CS_BLOCKS * blockDim.x) {
            const long long o = (long long)r * F + c;
            float v = src[o];
            if (gate) v = gate[o] > 0.f ? v : 0.f;
            if (dst) dst[o] = v;
            acc += wrow ? wrow[r] * v : v;
        }
        acc = wave_sum(acc);
        if (lane_id() == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) partial[(long long)blockIdx.x * F + c] = (red[0] + red[1]) + (red[2] + red[3]);
        __syncthreads();
    }
}

// 64 columns x 4 partial ranges per workgroup
__global__ __launch_bounds__(256) void colsum_final_k(const float* __restrict__ partial, float* __restrict__ out, int F,
                                                      int accumulate) {
    __shared__ float part[4][64];
    const int g = threadIdx.x >> 6, cl = threadIdx.x & 63;
    const int c = blockIdx.x * 64 + cl;
    float acc = 0.f;
    if (c < F) {
        const int b0 = g * (CS_BLOCKS / 4);
#pragma unroll 8
        for (int b = b0; b < b0 + CS_BLOCKS / 4; ++b) acc += partial[(long long)b * F + c];
    }
    part[g][cl] = acc;
    __syncthreads();
    if (g == 0 && c < F) {
        const float t = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
        out[c] = accumulate ? out[c] + t : t;
    }
}

size_t grapes_colsum_workspace_bytes(int F) { return (size_t)CS_BLOCKS * (F > 0 ? F : 1) * sizeof(float); }

int grapes_colsum_launch(const float* src, const float* gate, const float* wrow, float* dst, float* out, int n,
                         const int32_t* d_n, int F, int accumulate, float* workspace, hipStream_t s) {
    if (F <= 16)
        hipLaunchKernelGGL(colsum_partial_narrow_k, dim3(CS_BLOCKS), dim3(256), 0, s, src, gate, wrow, dst, workspace, n, d_n, F);
    else
        hipLaunchKernelGGL(colsum_partial_k, dim3(CS_BLOCKS), dim3(256), 0, s, src, gate, wrow, dst, workspace, n, d_n, F);
    GRAPES_LAUNCH_CHECK();
    if (out) {
        hipLaunchKernelGGL(colsum_final_k, dim3(grapes_div_up(F, 64)), dim3(256), 0, s, (const float*)workspace, out, F, accumulate);
        GRAPES_LAUNCH_CHECK();
    }
    return 0;
}

// ============================================================================ C-ABI
extern "C" int grapes_gcn_aggregate_fwd(const float* h, const int32_t* rowptr_t, const int32_t* csr_src,
                                        const float* dinv, const float* bias, float* out, int32_t n,
                                        const int32_t* d_n, int32_t f, int32_t relu, const int32_t* long_rows,
                                        const int32_t* d_n_long, grapes_stream_t stream) {
    if (n < 0 || f <= 0) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!h || !rowptr_t || !dinv || !out) return GRAPES_EINVAL;
    return launch_aggregate(h, rowptr_t, csr_src, dinv, bias, out, n, d_n, f, relu, long_rows, d_n_long, (hipStream_t)stream);
}

extern "C" size_t grapes_gcn_aggregate_bwd_workspace_bytes(int32_t n_cap, int32_t f) {
    (void)n_cap;
    return grapes_colsum_workspace_bytes(f);
}

extern "C" int grapes_gcn_aggregate_bwd(const float* dout, const float* relu_out, const int32_t* rowptr_s,
                                        const int32_t* csr_dst, const float* dinv, float* dpre_buf, float* dh,
                                        float* dbias, int32_t accumulate_bias, int32_t n, const int32_t* d_n,
                                        int32_t f, const int32_t* long_rows, const int32_t* d_n_long, void* workspace,
                                        grapes_stream_t stream) {
    if (n < 0 || f <= 0) return GRAPES_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        if (dbias && !accumulate_bias) { hipError_t e = hipMemsetAsync(dbias, 0, (size_t)f * sizeof(float), s); if (e) return (int)e; }
        return 0;
    }
    if (!dout || !rowptr_s || !dinv || !dh || !dpre_buf) return GRAPES_EINVAL;
    const bool need_pass = (relu_out != nullptr) || (dbias != nullptr) || (dpre_buf != dout);
    if (need_pass) {
        if (!workspace) return GRAPES_EINVAL;
        float* dst = (relu_out != nullptr || dpre_buf != dout) ? dpre_buf : nullptr;
        int rc = grapes_colsum_launch(dout, relu_out, nullptr, dst, dbias, n, d_n, f, accumulate_bias, (float*)workspace, s);
        if (rc) return rc;
    }
    return launch_aggregate(dpre_buf, rowptr_s, csr_dst, dinv, nullptr, dh, n, d_n, f, 0, long_rows, d_n_long, s);
}
