// A6: gcn_norm + per-hop CSR construction (PyG gcn_norm semantics).
//   deg[c] = #non-loop edges into c + 1 (unit self-loop);  dinv = deg^-1/2
//   CSR by TARGET (forward aggregation) and by SOURCE (backward), neighbour ids ascending in each
//   row, so the fp32 summation order of the aggregation is a function of the graph only.
// Two build modes:
//   generic            : atomic histogram + scan + atomic-cursor fill + per-row canonical sort
//   SRC_GROUPED flag   : the edge list is what frontier_expand / slice_filter emit — one contiguous
//                        segment per source, destinations ascending inside it.  The by-source CSR is
//                        then written directly (no atomics, no sort: hub rows of 10^4 entries cost
//                        nothing extra); only the short by-target rows are sorted.
#include "common.h"

__global__ void prep_hist_k(const int32_t* __restrict__ es, const int32_t* __restrict__ ed, int e_host,
                            const int32_t* d_e, int n_host, const int32_t* d_n, int grouped,
                            int32_t* __restrict__ cnt_t, int32_t* __restrict__ cnt_s,
                            int32_t* __restrict__ seg_first, int32_t* __restrict__ loops,
                            int32_t* __restrict__ nseg, int32_t* status) {
    const int e = eff_count(d_e, e_host);
    const int n = eff_count(d_n, n_host);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < e; t += gridDim.x * blockDim.x) {
        const int s = es[t], d = ed[t];
        if ((unsigned)s >= (unsigned)n || (unsigned)d >= (unsigned)n) {
            if (status) atomicOr(status, GRAPES_STATUS_BAD_INDEX);
            continue;
        }
        if (grouped && (t == 0 || es[t - 1] != s)) {
            seg_first[s] = t;
            if (atomicAdd(&nseg[s], 1) > 0 && status) atomicOr(status, GRAPES_STATUS_BAD_INDEX);  // not grouped
        }
        if (s == d) {   // add_remaining_self_loops: existing loops are replaced by the unit loop
            if (grouped) atomicAdd(&loops[s], 1);
            continue;
        }
        atomicAdd(&cnt_t[d], 1);
        atomicAdd(&cnt_s[s], 1);
    }
}

// One workgroup: exclusive scans of both degree arrays, dinv, cursor initialisation, long-row lists.
__global__ __launch_bounds__(1024) void prep_scan_k(int n_host, const int32_t* d_n, int32_t* __restrict__ cnt_t,
                                                    int32_t* __restrict__ cnt_s, int32_t* __restrict__ rowptr_t,
                                                    int32_t* __restrict__ rowptr_s, float* __restrict__ dinv,
                                                    int32_t* __restrict__ long_rows, int32_t* __restrict__ n_long,
                                                    int long_cap) {
    __shared__ int lds[17];
    __shared__ int s_nl[2];
    const int n = eff_count(d_n, n_host);
    if (threadIdx.x < 2) s_nl[threadIdx.x] = 0;
    __syncthreads();
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
            if (long_rows) {
                if (ct > GRAPES_LONG_ROW) { const int p = atomicAdd(&s_nl[0], 1); if (p < long_cap) long_rows[p] = i; }
                if (cs > GRAPES_LONG_ROW) { const int p = atomicAdd(&s_nl[1], 1); if (p < long_cap) long_rows[long_cap + p] = i; }
            }
        }
        carry_t += tot_t;
        carry_s += tot_s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        rowptr_t[n] = carry_t; rowptr_s[n] = carry_s;
        if (n_long) {
            n_long[0] = s_nl[0] < long_cap ? s_nl[0] : long_cap;
            n_long[1] = s_nl[1] < long_cap ? s_nl[1] : long_cap;
        }
    }
}

__global__ void prep_fill_k(const int32_t* __restrict__ es, const int32_t* __restrict__ ed, int e_host,
                            const int32_t* d_e, int n_host, const int32_t* d_n, int grouped,
                            int32_t* __restrict__ cur_t, int32_t* __restrict__ cur_s,
                            const int32_t* __restrict__ rowptr_s, const int32_t* __restrict__ seg_first,
                            const int32_t* __restrict__ loops, int32_t* __restrict__ tmp_src,
                            int32_t* __restrict__ tmp_dst, int32_t* __restrict__ csr_dst) {
    const int e = eff_count(d_e, e_host);
    const int n = eff_count(d_n, n_host);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < e; t += gridDim.x * blockDim.x) {
        const int s = es[t], d = ed[t];
        if ((unsigned)s >= (unsigned)n || (unsigned)d >= (unsigned)n || s == d) continue;
        tmp_src[atomicAdd(&cur_t[d], 1)] = s;
        if (grouped) {
            // destinations ascend inside the segment => the dropped loop entries (d == s) precede t iff d > s
            csr_dst[rowptr_s[s] + (t - seg_first[s]) - (d > s ? loops[s] : 0)] = d;
        } else {
            tmp_dst[atomicAdd(&cur_s[s], 1)] = d;
        }
    }
}

// Canonical (ascending) order inside every CSR row.  Rows [0,n) are the by-target rows, rows
// [n,2n) the by-source rows (skipped in grouped mode).  Short rows: one lane each (register
// insertion network); longer rows: the whole wavefront rank-sorts them.
#define SORT_SHORT 8
__global__ __launch_bounds__(256) void prep_sort_rows_k(int n_host, const int32_t* d_n, int both,
                                                        const int32_t* __restrict__ rowptr_t,
                                                        const int32_t* __restrict__ rowptr_s,
                                                        const int32_t* __restrict__ tmp_src,
                                                        const int32_t* __restrict__ tmp_dst,
                                                        int32_t* __restrict__ csr_src, int32_t* __restrict__ csr_dst) {
    const int n = eff_count(d_n, n_host);
    const int total = both ? 2 * n : n;
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
    return (5 * n + 2 * e) * sizeof(int32_t);
}

extern "C" int32_t grapes_gcn_long_rows_capacity(int32_t e_cap) { return e_cap / GRAPES_LONG_ROW + 2; }

extern "C" int grapes_gcn_prepare(const int32_t* edge_src, const int32_t* edge_dst, int32_t e, const int32_t* d_e,
                                  int32_t n, const int32_t* d_n, int32_t flags, int32_t* rowptr_t, int32_t* csr_src,
                                  int32_t* rowptr_s, int32_t* csr_dst, float* dinv, int32_t* long_rows,
                                  int32_t* n_long, void* workspace, int32_t* status, grapes_stream_t stream) {
    if (e < 0 || n < 0 || !rowptr_t || !rowptr_s || !dinv || !workspace) return GRAPES_EINVAL;
    if (e > 0 && (!edge_src || !edge_dst || !csr_src || !csr_dst)) return GRAPES_EINVAL;
    if ((long_rows == nullptr) != (n_long == nullptr)) return GRAPES_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int grouped = (flags & GRAPES_PREP_SRC_GROUPED) ? 1 : 0;
    const size_t n1 = (size_t)n + 1;
    int32_t* cnt_t = (int32_t*)workspace;
    int32_t* cnt_s = cnt_t + n1;
    int32_t* seg_first = cnt_s + n1;
    int32_t* loops = seg_first + n1;
    int32_t* nseg = loops + n1;
    int32_t* tmp_src = nseg + n1;
    int32_t* tmp_dst = tmp_src + (size_t)e + 1;
    hipError_t err = hipMemsetAsync(cnt_t, 0, 5 * n1 * sizeof(int32_t), s);
    if (err != hipSuccess) return (int)err;
    int ge = grapes_div_up(e > 0 ? e : 1, 256); if (ge > 4096) ge = 4096;
    if (e > 0) {
        hipLaunchKernelGGL(prep_hist_k, dim3(ge), dim3(256), 0, s, edge_src, edge_dst, e, d_e, n, d_n, grouped, cnt_t, cnt_s,
                           seg_first, loops, nseg, status);
        GRAPES_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(prep_scan_k, dim3(1), dim3(1024), 0, s, n, d_n, cnt_t, cnt_s, rowptr_t, rowptr_s, dinv, long_rows,
                       n_long, grapes_gcn_long_rows_capacity(e));
    GRAPES_LAUNCH_CHECK();
    if (e > 0 && n > 0) {
        hipLaunchKernelGGL(prep_fill_k, dim3(ge), dim3(256), 0, s, edge_src, edge_dst, e, d_e, n, d_n, grouped, cnt_t, cnt_s,
                           (const int32_t*)rowptr_s, (const int32_t*)seg_first, (const int32_t*)loops, tmp_src, tmp_dst, csr_dst);
        GRAPES_LAUNCH_CHECK();
        const int both = grouped ? 0 : 1;
        int gr = grapes_div_up((both ? 2 : 1) * (int64_t)n, 256); if (gr > 4096) gr = 4096;
        hipLaunchKernelGGL(prep_sort_rows_k, dim3(gr), dim3(256), 0, s, n, d_n, both, (const int32_t*)rowptr_t,
                           (const int32_t*)rowptr_s, (const int32_t*)tmp_src, (const int32_t*)tmp_dst, csr_src, csr_dst);
        GRAPES_LAUNCH_CHECK();
    }
    return 0;
}
