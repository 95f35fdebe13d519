// GCNConv on gfx950 (PyG GCNConv semantics; routing in modules/gcn.py:24-42).
//   gcn_prepare      : gcn_norm + per-hop CSR by target / by source           (HBM/latency bound)
//   linear_*         : dense XWᵀ, dHᵀX, dH W on fp32 MFMA v_mfma_f32_32x32x2_f32 (MFMA bound)
//   gcn_aggregate_*  : gather-SpMM, one wavefront per destination row, 16 B/lane (HBM bound)
#include "common.h"

// ============================================================================ gcn_prepare
__global__ void prep_hist_k(const int32_t* __restrict__ es, const int32_t* __restrict__ ed, int e_host,
                            const int32_t* d_e, int n_host, const int32_t* d_n, int32_t* __restrict__ cnt_t,
                            int32_t* __restrict__ cnt_s, int32_t* status) {
    const int e = eff_count(d_e, e_host);
    const int n = eff_count(d_n, n_host);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < e; t += gridDim.x * blockDim.x) {
        const int s = es[t], d = ed[t];
        if ((unsigned)s >= (unsigned)n || (unsigned)d >= (unsigned)n) {
            if (status) atomicOr(status, GRAPES_STATUS_BAD_INDEX);
            continue;
        }
        if (s == d) continue;   // add_remaining_self_loops: existing loops are replaced by the unit loop
        atomicAdd(&cnt_t[d], 1);
        atomicAdd(&cnt_s[s], 1);
    }
}

// One workgroup: exclusive scans of both degree arrays, dinv, and cursor initialisation.
__global__ __launch_bounds__(1024) void prep_scan_k(int n_host, const int32_t* d_n, int32_t* __restrict__ cnt_t,
                                                    int32_t* __restrict__ cnt_s, int32_t* __restrict__ rowptr_t,
                                                    int32_t* __restrict__ rowptr_s, float* __restrict__ dinv) {
    __shared__ int lds[17];
    const int n = eff_count(d_n, n_host);
    int carry_t = 0, carry_s = 0;
    for (int base = 0; base < n; base += blockDim.x) {
        const int i = base + threadIdx.x;
        const int ct = i < n ? cnt_t[i] : 0;
        const int cs = i < n ? cnt_s[i] : 0;
        int tot_t, tot_s;
        const int ex_t = block_excl_scan(ct, lds, &tot_t);
        const int ex_s = block_excl_scan(cs, lds, &tot_s);
        if (i < n) {
            rowptr_t[i] = carry_t + ex_t;
            rowptr_s[i] = carry_s + ex_s;
            cnt_t[i] = carry_t + ex_t;   // becomes the fill cursor
            cnt_s[i] = carry_s + ex_s;
            dinv[i] = 1.0f / sqrtf((float)(ct + 1));   // deg = in-degree + unit self-loop
        }
        carry_t += tot_t;
        carry_s += tot_s;
    }
    if (threadIdx.x == 0) { rowptr_t[n] = carry_t; rowptr_s[n] = carry_s; }
}

__global__ void prep_fill_k(const int32_t* __restrict__ es, const int32_t* __restrict__ ed, int e_host,
                            const int32_t* d_e, int n_host, const int32_t* d_n, int32_t* __restrict__ cur_t,
                            int32_t* __restrict__ cur_s, int32_t* __restrict__ tmp_src,
                            int32_t* __restrict__ tmp_dst) {
    const int e = eff_count(d_e, e_host);
    const int n = eff_count(d_n, n_host);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < e; t += gridDim.x * blockDim.x) {
        const int s = es[t], d = ed[t];
        if ((unsigned)s >= (unsigned)n || (unsigned)d >= (unsigned)n || s == d) continue;
        tmp_src[atomicAdd(&cur_t[d], 1)] = s;
        tmp_dst[atomicAdd(&cur_s[s], 1)] = d;
    }
}

// Canonical (ascending) order inside every CSR row => the fp32 summation order of the
// aggregation does not depend on the atomic fill order.  Rows [0,n) are the by-target rows,
// rows [n,2n) the by-source rows.  Short rows: one lane each; long rows: whole wavefront.
#define SORT_SHORT 8
__global__ __launch_bounds__(256) void prep_sort_rows_k(int n_host, const int32_t* d_n,
                                                        const int32_t* __restrict__ rowptr_t,
                                                        const int32_t* __restrict__ rowptr_s,
                                                        const int32_t* __restrict__ tmp_src,
                                                        const int32_t* __restrict__ tmp_dst,
                                                        int32_t* __restrict__ csr_src, int32_t* __restrict__ csr_dst) {
    const int n = eff_count(d_n, n_host);
    const int total = 2 * n;
    const int lane = lane_id();
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int base = wave_global * 64; base < total; base += nwaves * 64) {
        const int r = base + lane;
        int beg = 0, len = 0;
        const int32_t* in = tmp_src;
        int32_t* out = csr_src;
        if (r < total) {
            if (r < n) { beg = rowptr_t[r]; len = rowptr_t[r + 1] - beg; }
            else { beg = rowptr_s[r - n]; len = rowptr_s[r - n + 1] - beg; in = tmp_dst; out = csr_dst; }
        }
        if (len > 0 && len <= SORT_SHORT) {
            int v[SORT_SHORT];
#pragma unroll
            for (int i = 0; i < SORT_SHORT; ++i) v[i] = i < len ? in[beg + i] : 0x7fffffff;
#pragma unroll
            for (int i = 1; i < SORT_SHORT; ++i) {
#pragma unroll
                for (int j = i; j > 0; --j) {
                    const int a = v[j - 1], b = v[j];
                    v[j - 1] = a < b ? a : b;
                    v[j] = a < b ? b : a;
                }
            }
#pragma unroll
            for (int i = 0; i < SORT_SHORT; ++i)
                if (i < len) out[beg + i] = v[i];
        }
        unsigned long long longs = __ballot(len > SORT_SHORT);
        while (longs) {
            const int l = __ffsll((long long)longs) - 1;
            longs &= longs - 1;
            const int lbeg = __shfl(beg, l, 64);
            const int llen = __shfl(len, l, 64);
            const int rr = base + l;
            const int32_t* lin = rr < n ? tmp_src : tmp_dst;
            int32_t* lout = rr < n ? csr_src : csr_dst;
            for (int i = lane; i < llen; i += 64) {
                const int v = lin[lbeg + i];
                int rank = 0;
                for (int j = 0; j < llen; ++j) {
                    const int u = lin[lbeg + j];
                    rank += (u < v) || (u == v && j < i);
                }
                lout[lbeg + rank] = v;
            }
        }
    }
}

extern "C" size_t grapes_gcn_prepare_workspace_bytes(int32_t n_cap, int32_t e_cap) {
    size_t n = (size_t)(n_cap > 0 ? n_cap : 0) + 1, e = (size_t)(e_cap > 0 ? e_cap : 0) + 1;
    return (2 * n + 2 * e) * sizeof(int32_t);
}

extern "C" int grapes_gcn_prepare(const int32_t* edge_src, const int32_t* edge_dst, int32_t e, const int32_t* d_e,
                                  int32_t n, const int32_t* d_n, int32_t* rowptr_t, int32_t* csr_src,
                                  int32_t* rowptr_s, int32_t* csr_dst, float* dinv, void* workspace,
                                  int32_t* status, grapes_stream_t stream) {
    if (e < 0 || n < 0 || !rowptr_t || !rowptr_s || !dinv || !workspace) return GRAPES_EINVAL;
    if (e > 0 && (!edge_src || !edge_dst || !csr_src || !csr_dst)) return GRAPES_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int32_t* cnt_t = (int32_t*)workspace;
    int32_t* cnt_s = cnt_t + (size_t)n + 1;
    int32_t* tmp_src = cnt_s + (size_t)n + 1;
    int32_t* tmp_dst = tmp_src + (size_t)e + 1;
    hipError_t err = hipMemsetAsync(cnt_t, 0, (2 * ((size_t)n + 1)) * sizeof(int32_t), s);
    if (err != hipSuccess) return (int)err;
    int ge = grapes_div_up(e > 0 ? e : 1, 256); if (ge > 4096) ge = 4096;
    if (e > 0) {
        hipLaunchKernelGGL(prep_hist_k, dim3(ge), dim3(256), 0, s, edge_src, edge_dst, e, d_e, n, d_n, cnt_t, cnt_s, status);
        GRAPES_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(prep_scan_k, dim3(1), dim3(1024), 0, s, n, d_n, cnt_t, cnt_s, rowptr_t, rowptr_s, dinv);
    GRAPES_LAUNCH_CHECK();
    if (e > 0 && n > 0) {
        hipLaunchKernelGGL(prep_fill_k, dim3(ge), dim3(256), 0, s, edge_src, edge_dst, e, d_e, n, d_n, cnt_t, cnt_s, tmp_src, tmp_dst);
        GRAPES_LAUNCH_CHECK();
        int gr = grapes_div_up(2 * (int64_t)n, 256); if (gr > 4096) gr = 4096;
        hipLaunchKernelGGL(prep_sort_rows_k, dim3(gr), dim3(256), 0, s, n, d_n, rowptr_t, rowptr_s, tmp_src, tmp_dst, csr_src, csr_dst);
        GRAPES_LAUNCH_CHECK();
    }
    return 0;
}

// ============================================================================ dense transforms (MFMA fp32)
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GB_M 128
#define GB_N 128
#define GB_K 8
#define GB_LD 132   // LDS row stride (floats): 16 B aligned rows, 2-way-at-most write conflicts

// Operand tile -> registers.  KMAJOR: memory is [k][r] (r contiguous); otherwise [r][k].
template <bool KMAJOR, bool VEC>
__device__ __forceinline__ float4 gemm_load_tile(const float* __restrict__ P, long long ld, int r0, int R, int k0,
                                                 int kend, int tid) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KMAJOR) {
        const int k = k0 + (tid >> 5);
        const int r = r0 + (tid & 31) * 4;
        if (k < kend) {
            const float* p = P + (long long)k * ld + r;
            if (VEC && r + 3 < R) {
                v = *reinterpret_cast<const float4*>(p);
            } else {
                if (r + 0 < R) v.x = p[0];
                if (r + 1 < R) v.y = p[1];
                if (r + 2 < R) v.z = p[2];
                if (r + 3 < R) v.w = p[3];
            }
        }
    } else {
        const int r = r0 + (tid >> 1);
        const int k = k0 + (tid & 1) * 4;
        if (r < R) {
            const float* p = P + (long long)r * ld + k;
            if (VEC && k + 3 < kend) {
                v = *reinterpret_cast<const float4*>(p);
            } else {
                if (k + 0 < kend) v.x = p[0];
                if (k + 1 < kend) v.y = p[1];
                if (k + 2 < kend) v.z = p[2];
                if (k + 3 < kend) v.w = p[3];
            }
        }
    }
    return v;
}

template <bool KMAJOR>
__device__ __forceinline__ void gemm_store_tile(float (*S)[GB_LD], const float4 v, int tid) {
    if (KMAJOR) {
        *reinterpret_cast<float4*>(&S[tid >> 5][(tid & 31) * 4]) = v;
    } else {
        const int r = tid >> 1, k = (tid & 1) * 4;
        S[k + 0][r] = v.x; S[k + 1][r] = v.y; S[k + 2][r] = v.z; S[k + 3][r] = v.w;
    }
}

// C[m][n] = sum_k Aop[m][k] * Bop[n][k].   128x128 tile / workgroup, 4 waves (2x2), each wave a
// 64x64 sub-tile = 2x2 MFMA 32x32 accumulators.  K is streamed through a double-buffered k-major
// LDS image; next tile's global loads are in flight while the current one feeds the MFMAs.
// blockIdx.x -> (tm, tn) keeps the N-tiles of one row panel on one XCD (ids differ by 8).
// blockIdx.y = split-K slab (dW): slab z covers k in [z*kchunk, (z+1)*kchunk) and writes C + z*slab.
template <bool A_KMAJOR, bool B_KMAJOR, bool VEC>
__global__ __launch_bounds__(256, 2) void gemm_mfma_f32_k(const float* __restrict__ A, const float* __restrict__ B,
                                                           float* __restrict__ C, int M_host, int N, int K_host,
                                                           long long lda, long long ldb, long long ldc,
                                                           const int32_t* d_M, const int32_t* d_K, int kchunk,
                                                           long long slab, int nt) {
    __shared__ __attribute__((aligned(16))) float As[2][GB_K][GB_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GB_K][GB_LD];
    const int M = eff_count(d_M, M_host);
    const int K = eff_count(d_K, K_host);
    const int bid = blockIdx.x;
    const int group = bid / (8 * nt), within = bid - group * 8 * nt;
    const int tn = within >> 3, tm = group * 8 + (within & 7);
    const int m0 = tm * GB_M, n0 = tn * GB_N;
    if (m0 >= M) return;
    const int kb = blockIdx.y * kchunk;
    int ke = kb + kchunk; if (ke > K) ke = K;
    if (kb >= K && gridDim.y > 1) return;
    C += (long long)blockIdx.y * slab;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int li = lane & 31, lk = lane >> 5;

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    const int nk = (ke > kb) ? (ke - kb + GB_K - 1) / GB_K : 0;
    if (nk > 0) {
        float4 ra = gemm_load_tile<A_KMAJOR, VEC>(A, lda, m0, M, kb, ke, tid);
        float4 rb = gemm_load_tile<B_KMAJOR, VEC>(B, ldb, n0, N, kb, ke, tid);
        gemm_store_tile<A_KMAJOR>(As[0], ra, tid);
        gemm_store_tile<B_KMAJOR>(Bs[0], rb, tid);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) {
                ra = gemm_load_tile<A_KMAJOR, VEC>(A, lda, m0, M, kb + (kt + 1) * GB_K, ke, tid);
                rb = gemm_load_tile<B_KMAJOR, VEC>(B, ldb, n0, N, kb + (kt + 1) * GB_K, ke, tid);
            }
#pragma unroll
            for (int kk = 0; kk < GB_K; kk += 2) {
                const float a0 = As[cur][kk + lk][wm * 64 + li];
                const float a1 = As[cur][kk + lk][wm * 64 + 32 + li];
                const float b0 = Bs[cur][kk + lk][wn * 64 + li];
                const float b1 = Bs[cur][kk + lk][wn * 64 + 32 + li];
                acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
            }
            if (kt + 1 < nk) {
                gemm_store_tile<A_KMAJOR>(As[cur ^ 1], ra, tid);
                gemm_store_tile<B_KMAJOR>(Bs[cur ^ 1], rb, tid);
            }
            __syncthreads();
        }
    }
    // D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int gm0 = m0 + wm * 64 + row, gm1 = gm0 + 32;
        const int gn0 = n0 + wn * 64 + li, gn1 = gn0 + 32;
        if (gm0 < M) {
            if (gn0 < N) C[(long long)gm0 * ldc + gn0] = acc00[r];
            if (gn1 < N) C[(long long)gm0 * ldc + gn1] = acc01[r];
        }
        if (gm1 < M) {
            if (gn0 < N) C[(long long)gm1 * ldc + gn0] = acc10[r];
            if (gn1 < N) C[(long long)gm1 * ldc + gn1] = acc11[r];
        }
    }
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <bool AK, bool BK_>
static int launch_gemm(const float* A, const float* B, float* C, int M, int N, int K, long long lda, long long ldb,
                       long long ldc, const int32_t* d_M, const int32_t* d_K, int kchunk, int nslab, long long slab,
                       hipStream_t s) {
    const int mt = grapes_div_up(M, GB_M), nt = grapes_div_up(N, GB_N);
    const int grid_x = grapes_div_up(mt, 8) * 8 * nt;
    const bool vec = aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0);
    dim3 grid(grid_x, nslab);
    if (vec)
        hipLaunchKernelGGL((gemm_mfma_f32_k<AK, BK_, true>), grid, dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, d_M, d_K, kchunk, slab, nt);
    else
        hipLaunchKernelGGL((gemm_mfma_f32_k<AK, BK_, false>), grid, dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, d_M, d_K, kchunk, slab, nt);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// ---- f_out == 1 (the logit heads of gcn_gf / gcn_z): GEMV forms, no MFMA.
__global__ __launch_bounds__(256) void gemv_rows_k(const float* __restrict__ x, const float* __restrict__ w,
                                                   float* __restrict__ h, int n_host, const int32_t* d_n, int F) {
    const int n = eff_count(d_n, n_host);
    const int lane = lane_id();
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const bool vec = (F % 4 == 0) && ((((uintptr_t)x) & 15) == 0) && ((((uintptr_t)w) & 15) == 0);
    for (int r = wave_global; r < n; r += nwaves) {
        const float* xr = x + (long long)r * F;
        float acc = 0.f;
        if (vec) {
            for (int f = lane * 4; f < F; f += 256) {
                const float4 a = *reinterpret_cast<const float4*>(xr + f);
                const float4 b = *reinterpret_cast<const float4*>(w + f);
                acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc);
                acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
            }
        } else {
            for (int f = lane; f < F; f += 64) acc = fmaf(xr[f], w[f], acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) h[r] = acc;
    }
}

__global__ __launch_bounds__(256) void outer_rows_k(const float* __restrict__ dh, const float* __restrict__ w,
                                                    float* __restrict__ dx, int n_host, const int32_t* d_n, int F) {
    const int n = eff_count(d_n, n_host);
    const long long total = (long long)n * F;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / F);
        const int f = (int)(i - (long long)r * F);
        dx[i] = dh[r] * w[f];
    }
}

// ---- column sums:  partial[blk][c] = sum over the block's rows of  wrow[r] * val(r,c)
//      val = src[r][c], optionally gated by (gate[r][c] > 0) (ReLU backward); optionally the gated
//      values are written to `dst` (dpre).  Fixed row->block map and fixed order => deterministic.
#define CS_ROWS 64
__global__ __launch_bounds__(256) void colsum_partial_k(const float* __restrict__ src, const float* __restrict__ gate,
                                                        const float* __restrict__ wrow, float* __restrict__ dst,
                                                        float* __restrict__ partial, int n_host, const int32_t* d_n,
                                                        int F) {
    const int n = eff_count(d_n, n_host);
    const int r0 = blockIdx.x * CS_ROWS;
    if (r0 >= n) return;
    const int r1 = r0 + CS_ROWS < n ? r0 + CS_ROWS : n;
    for (int c = threadIdx.x; c < F; c += blockDim.x) {
        float acc = 0.f;
        for (int r = r0; r < r1; ++r) {
            const long long o = (long long)r * F + c;
            float v = src[o];
            if (gate) v = gate[o] > 0.f ? v : 0.f;
            if (dst) dst[o] = v;
            acc += wrow ? wrow[r] * v : v;
        }
        partial[(long long)blockIdx.x * F + c] = acc;
    }
}

__global__ void colsum_final_k(const float* __restrict__ partial, float* __restrict__ out, int n_host,
                               const int32_t* d_n, int F, int accumulate) {
    const int n = eff_count(d_n, n_host);
    const int nblk = (n + CS_ROWS - 1) / CS_ROWS;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < F; c += gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int b = 0; b < nblk; ++b) acc += partial[(long long)b * F + c];
        out[c] = accumulate ? out[c] + acc : acc;
    }
}

// ---- split-K slab reduction for dW (fixed order => deterministic)
__global__ void slab_reduce_k(const float* __restrict__ slabs, float* __restrict__ out, long long count,
                              int k_host, const int32_t* d_k, int kchunk, int accumulate) {
    const int K = eff_count(d_k, k_host);
    const int ns = (K + kchunk - 1) / kchunk;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (long long)gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int z = 0; z < ns; ++z) acc += slabs[(long long)z * count + i];
        out[i] = accumulate ? out[i] + acc : acc;
    }
}

#define DW_KCHUNK 512

extern "C" int grapes_linear_fwd(const float* x, const float* w, float* h, int32_t n, const int32_t* d_n,
                                 int32_t f_in, int32_t f_out, grapes_stream_t stream) {
    if (n < 0 || f_in <= 0 || f_out <= 0) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!x || !w || !h) return GRAPES_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (f_out == 1) {
        int grid = grapes_div_up(n, 4); if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(gemv_rows_k, dim3(grid), dim3(256), 0, s, x, w, h, n, d_n, f_in);
        GRAPES_LAUNCH_CHECK();
        return 0;
    }
    // A = x [n,f_in] (k contiguous), B = w [f_out,f_in] (k contiguous)
    return launch_gemm<false, false>(x, w, h, n, f_out, f_in, f_in, f_in, f_out, d_n, nullptr, f_in + GB_K, 1, 0, s);
}

extern "C" size_t grapes_linear_bwd_weight_workspace_bytes(int32_t n_cap, int32_t f_in, int32_t f_out) {
    if (n_cap <= 0) n_cap = 1;
    if (f_out == 1) return (size_t)grapes_div_up(n_cap, CS_ROWS) * f_in * sizeof(float);
    return (size_t)grapes_div_up(n_cap, DW_KCHUNK) * f_in * f_out * sizeof(float);
}

extern "C" int grapes_linear_bwd_weight(const float* dh, const float* x, float* dw, int32_t n, const int32_t* d_n,
                                        int32_t f_in, int32_t f_out, int32_t accumulate, void* workspace,
                                        grapes_stream_t stream) {
    if (n < 0 || f_in <= 0 || f_out <= 0 || !dw) return GRAPES_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        if (!accumulate) { hipError_t e = hipMemsetAsync(dw, 0, (size_t)f_in * f_out * sizeof(float), s); if (e) return (int)e; }
        return 0;
    }
    if (!dh || !x || !workspace) return GRAPES_EINVAL;
    if (f_out == 1) {   // dW[f] = sum_r dh[r] x[r,f]
        const int nblk = grapes_div_up(n, CS_ROWS);
        hipLaunchKernelGGL(colsum_partial_k, dim3(nblk), dim3(256), 0, s, x, (const float*)nullptr, dh, (float*)nullptr,
                           (float*)workspace, n, d_n, f_in);
        GRAPES_LAUNCH_CHECK();
        hipLaunchKernelGGL(colsum_final_k, dim3(grapes_div_up(f_in, 256)), dim3(256), 0, s, (const float*)workspace, dw, n,
                           d_n, f_in, accumulate);
        GRAPES_LAUNCH_CHECK();
        return 0;
    }
    // dW[f_out,f_in] = sum_r dh[r,f_out] x[r,f_in] :  A = dh (k-major, M=f_out), B = x (k-major, N=f_in), K = n rows
    const int nslab = grapes_div_up(n, DW_KCHUNK);
    const long long slab = (long long)f_in * f_out;
    int rc = launch_gemm<true, true>(dh, x, (float*)workspace, f_out, f_in, n, f_out, f_in, f_in, nullptr, d_n,
                                     DW_KCHUNK, nslab, slab, s);
    if (rc) return rc;
    int grid = grapes_div_up(slab, 256); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(slab_reduce_k, dim3(grid), dim3(256), 0, s, (const float*)workspace, dw, slab, n, d_n, DW_KCHUNK,
                       accumulate);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

extern "C" int grapes_linear_bwd_input(const float* dh, const float* w, float* dx, int32_t n, const int32_t* d_n,
                                       int32_t f_in, int32_t f_out, grapes_stream_t stream) {
    if (n < 0 || f_in <= 0 || f_out <= 0) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!dh || !w || !dx) return GRAPES_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (f_out == 1) {
        int grid = grapes_div_up((int64_t)n * f_in, 256); if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(outer_rows_k, dim3(grid), dim3(256), 0, s, dh, w, dx, n, d_n, f_in);
        GRAPES_LAUNCH_CHECK();
        return 0;
    }
    // dX[n,f_in] = dh[n,f_out] · W[f_out,f_in] : A = dh (k contiguous, K=f_out), B = W (k-major: rows are k)
    return launch_gemm<false, true>(dh, w, dx, n, f_in, f_out, f_out, f_in, f_in, d_n, nullptr, f_out + GB_K, 1, 0, s);
}

// ============================================================================ gather-SpMM
// out[c] = sum_{j in row c} (dinv[nbr_j]*dinv[c]) * H[nbr_j]  +  (dinv[c]*dinv[c]) * H[c]  (+ bias, ReLU)
// One wavefront per destination row; a lane owns VEC consecutive features (VEC=4: one dwordx4
// per lane = a whole 1 KiB row of 256 fp32 per wave-instruction).  Row/neighbour indices and
// weights are wave-uniform (scalar loads); four neighbour rows are kept in flight per wave.
template <int VEC>
__global__ __launch_bounds__(256) void gcn_aggregate_k(const float* __restrict__ h, const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ csr, const float* __restrict__ dinv,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       int n_host, const int32_t* d_n, int F, int relu) {
    const int n = eff_count(d_n, n_host);
    const int lane = lane_id();
    const int wave_global = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int row = wave_global; row < n; row += nwaves) {
        const int beg = rowptr[row], end = rowptr[row + 1];
        const float dc = dinv[row];
        for (int f0 = lane * VEC; f0 < F; f0 += 64 * VEC) {
            float acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
            int j = beg;
            for (; j + 4 <= end; j += 4) {
                int s[4]; float w[4]; float val[4][VEC];
#pragma unroll
                for (int u = 0; u < 4; ++u) { s[u] = csr[j + u]; w[u] = dinv[s[u]] * dc; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float* p = h + (long long)s[u] * F + f0;
                    if (VEC == 4) {
                        const float4 t = *reinterpret_cast<const float4*>(p);
                        val[u][0] = t.x; val[u][1 % VEC] = t.y; val[u][2 % VEC] = t.z; val[u][3 % VEC] = t.w;
                    } else {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) val[u][v] = p[v];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w[u], val[u][v], acc[v]);
            }
            for (; j < end; ++j) {
                const int s = csr[j];
                const float w = dinv[s] * dc;
                const float* p = h + (long long)s * F + f0;
                if (VEC == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(p);
                    acc[0] = fmaf(w, t.x, acc[0]); acc[1 % VEC] = fmaf(w, t.y, acc[1 % VEC]);
                    acc[2 % VEC] = fmaf(w, t.z, acc[2 % VEC]); acc[3 % VEC] = fmaf(w, t.w, acc[3 % VEC]);
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w, p[v], acc[v]);
                }
            }
            {   // unit self-loop, appended last as in add_remaining_self_loops
                const float w = dc * dc;
                const float* p = h + (long long)row * F + f0;
                float* o = out + (long long)row * F + f0;
                if (VEC == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(p);
                    float4 r;
                    r.x = fmaf(w, t.x, acc[0]); r.y = fmaf(w, t.y, acc[1 % VEC]);
                    r.z = fmaf(w, t.z, acc[2 % VEC]); r.w = fmaf(w, t.w, acc[3 % VEC]);
                    if (bias) {
                        const float4 b = *reinterpret_cast<const float4*>(bias + f0);
                        r.x += b.x; r.y += b.y; r.z += b.z; r.w += b.w;
                    }
                    if (relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
                    *reinterpret_cast<float4*>(o) = r;
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        float r = fmaf(w, p[v], acc[v]);
                        if (bias) r += bias[f0 + v];
                        if (relu) r = fmaxf(r, 0.f);
                        o[v] = r;
                    }
                }
            }
        }
    }
}

// Narrow rows (F <= 16, e.g. the 1-wide logit heads): one lane per destination row.
__global__ __launch_bounds__(256) void gcn_aggregate_narrow_k(const float* __restrict__ h,
                                                              const int32_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ csr,
                                                              const float* __restrict__ dinv,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int n_host, const int32_t* d_n, int F, int relu) {
    const int n = eff_count(d_n, n_host);
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < n; row += gridDim.x * blockDim.x) {
        const int beg = rowptr[row], end = rowptr[row + 1];
        const float dc = dinv[row];
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
}

static int launch_aggregate(const float* h, const int32_t* rowptr, const int32_t* csr, const float* dinv,
                            const float* bias, float* out, int n, const int32_t* d_n, int f, int relu, hipStream_t s) {
    if (f <= 16) {
        int grid = grapes_div_up(n, 256); if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(gcn_aggregate_narrow_k, dim3(grid), dim3(256), 0, s, h, rowptr, csr, dinv, bias, out, n, d_n, f, relu);
    } else {
        int grid = grapes_div_up(n, 4); if (grid > 16384) grid = 16384;
        const bool vec = (f % 4 == 0) && aligned16(h) && aligned16(out) && (!bias || aligned16(bias));
        if (vec)
            hipLaunchKernelGGL((gcn_aggregate_k<4>), dim3(grid), dim3(256), 0, s, h, rowptr, csr, dinv, bias, out, n, d_n, f, relu);
        else
            hipLaunchKernelGGL((gcn_aggregate_k<1>), dim3(grid), dim3(256), 0, s, h, rowptr, csr, dinv, bias, out, n, d_n, f, relu);
    }
    GRAPES_LAUNCH_CHECK();
    return 0;
}

extern "C" int grapes_gcn_aggregate_fwd(const float* h, const int32_t* rowptr_t, const int32_t* csr_src,
                                        const float* dinv, const float* bias, float* out, int32_t n,
                                        const int32_t* d_n, int32_t f, int32_t relu, grapes_stream_t stream) {
    if (n < 0 || f <= 0) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!h || !rowptr_t || !dinv || !out) return GRAPES_EINVAL;
    return launch_aggregate(h, rowptr_t, csr_src, dinv, bias, out, n, d_n, f, relu, (hipStream_t)stream);
}

extern "C" size_t grapes_gcn_aggregate_bwd_workspace_bytes(int32_t n_cap, int32_t f) {
    if (n_cap <= 0) n_cap = 1;
    return (size_t)grapes_div_up(n_cap, CS_ROWS) * f * sizeof(float);
}

extern "C" int grapes_gcn_aggregate_bwd(const float* dout, const float* relu_out, const int32_t* rowptr_s,
                                        const int32_t* csr_dst, const float* dinv, float* dpre_buf, float* dh,
                                        float* dbias, int32_t accumulate_bias, int32_t n, const int32_t* d_n,
                                        int32_t f, void* workspace, grapes_stream_t stream) {
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
        const int nblk = grapes_div_up(n, CS_ROWS);
        float* dst = (relu_out != nullptr || dpre_buf != dout) ? dpre_buf : nullptr;
        float* partial = (float*)workspace;
        hipLaunchKernelGGL(colsum_partial_k, dim3(nblk), dim3(256), 0, s, dout, relu_out, (const float*)nullptr, dst, partial,
                           n, d_n, f);
        GRAPES_LAUNCH_CHECK();
        if (dbias) {
            hipLaunchKernelGGL(colsum_final_k, dim3(grapes_div_up(f, 256)), dim3(256), 0, s, (const float*)partial, dbias, n,
                               d_n, f, accumulate_bias);
            GRAPES_LAUNCH_CHECK();
        }
    }
    return launch_aggregate(dpre_buf, rowptr_s, csr_dst, dinv, nullptr, dh, n, d_n, f, 0, s);
}
