// A6: gcn_norm + per-hop CSR construction (PyG gcn_norm semantics).
//   deg[c] = #non-loop edges into c + 1 (unit self-loop);  dinv = deg^-1/2
//   CSR by TARGET (forward aggregation) and by SOURCE (backward), neighbour ids ascending in each
//   row, so the fp32 summation order of the aggregation is a function of the graph only.
// Two build modes:
//   generic            : atomic histogram + scan + atomic-cursor fill + per-row canonical sort
//   SRC_GROUPED flag   : the edge list is what frontier_expand / slice_filter emit — one contiguous
//                        segment per source, destinations ascending inside it.  Source degrees come
//                        from the segment bounds (no same-address atomics on hub sources) and the
//                        by-source CSR is written directly (no sort: hub rows of 10^4 entries cost
//                        nothing extra); only the short by-target rows are sorted.
// Rows longer than GRAPES_LONG_ROW are cut into chunks of GRAPES_LONG_ROW entries ("items") that the
// aggregation spreads over many workgroups.
#include "common.h"

__global__ void prep_hist_k(const int32_t* __restrict__ es, const int32_t* __restrict__ ed, int e_host,
                            const int32_t* d_e, int n_host, const int32_t* d_n, int grouped,
                            int32_t* __restrict__ cnt_t, int32_t* __restrict__ cnt_s,
                            int32_t* __restrict__ seg_first, int32_t* __restrict__ seg_last,
                            int32_t* __restrict__ loops, int32_t* __restrict__ nseg, int32_t* __restrict__ bad,
                            int32_t* status) {
    const int e = eff_count(d_e, e_host);
    const int n = eff_count(d_n, n_host);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < e; t += gridDim.x * blockDim.x) {
        const int s = es[t], d = ed[t];
        if ((unsigned)s >= (unsigned)n || (unsigned)d >= (unsigned)n) {
            if (status) atomicOr(status, GRAPES_STATUS_BAD_INDEX);
            continue;
        }
        if (grouped) {
            if (t == 0 || es[t - 1] != s) {
                seg_first[s] = t;
                if (atomicAdd(&nseg[s], 1) > 0) {   // a second segment for s: the list is not grouped
                    *bad = 1;
                    if (status) atomicOr(status, GRAPES_STATUS_BAD_INDEX);
                }
            }
            if (t == e - 1 || es[t + 1] != s) seg_last[s] = t;
        }
        if (s == d) {   // add_remaining_self_loops: existing loops are replaced by the unit loop
            if (grouped) atomicAdd(&loops[s], 1);
            continue;
        }
        atomicAdd(&cnt_t[d], 1);
        if (!grouped) atomicAdd(&cnt_s[s], 1);
    }
}

// nseg[n] doubles as the "list was not grouped" flag: then the by-source CSR is left EMPTY (and the
// status word says why) instead of being built from inconsistent segment bounds.
__device__ __forceinline__ int src_degree(int i, int grouped, const int32_t* cnt_s, const int32_t* nseg,
                                          const int32_t* seg_first, const int32_t* seg_last, const int32_t* loops,
                                          int bad) {
    if (!grouped) return cnt_s[i];
    if (bad || nseg[i] <= 0) return 0;
    const int d = seg_last[i] - seg_first[i] + 1 - loops[i];
    return d > 0 ? d : 0;
}

// stage 1 of the two-level scan: per-workgroup (1024 nodes) degree totals
__global__ __launch_bounds__(1024) void prep_scan_count_k(int n_host, const int32_t* d_n, int grouped,
                                                          const int32_t* __restrict__ cnt_t,
                                                          const int32_t* __restrict__ cnt_s,
                                                          const int32_t* __restrict__ nseg,
                                                          const int32_t* __restrict__ seg_first,
                                                          const int32_t* __restrict__ seg_last,
                                                          const int32_t* __restrict__ loops,
                                                          const int32_t* __restrict__ bad,
                                                          int32_t* __restrict__ bsum_t, int32_t* __restrict__ bsum_s,
                                                          int32_t* __restrict__ n_long) {
    __shared__ int lds[17];
    const int n = eff_count(d_n, n_host);
    if (blockIdx.x == 0 && threadIdx.x == 0 && n_long) { n_long[0] = 0; n_long[1] = 0; }   // item counters for stage 2
    if (blockIdx.x * blockDim.x >= n && blockIdx.x > 0) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int isbad = *bad;
    const int ct = i < n ? cnt_t[i] : 0;
    const int cs = i < n ? src_degree(i, grouped, cnt_s, nseg, seg_first, seg_last, loops, isbad) : 0;
    int tt, ts;
    block_excl_scan(ct, lds, &tt);
    block_excl_scan(cs, lds, &ts);
    if (threadIdx.x == 0) { bsum_t[blockIdx.x] = tt; bsum_s[blockIdx.x] = ts; }
}

__device__ __forceinline__ int prefix_of_sums(const int32_t* __restrict__ bsum, int b, int* lds) {
    int acc = 0;
    for (int i = threadIdx.x; i < b; i += blockDim.x) acc += bsum[i];
    int tot;
    block_excl_scan(acc, lds, &tot);
    return tot;
}

// stage 2: row pointers, fill cursors, dinv, long-row work items
__global__ __launch_bounds__(1024) void prep_scan_emit_k(int n_host, const int32_t* d_n, int grouped,
                                                         int32_t* __restrict__ cnt_t, int32_t* __restrict__ cnt_s,
                                                         const int32_t* __restrict__ nseg,
                                                         const int32_t* __restrict__ seg_first,
                                                         const int32_t* __restrict__ seg_last,
                                                         const int32_t* __restrict__ loops,
                                                         const int32_t* __restrict__ bad,
                                                         const int32_t* __restrict__ bsum_t,
                                                         const int32_t* __restrict__ bsum_s,
                                                         int32_t* __restrict__ rowptr_t, int32_t* __restrict__ rowptr_s,
                                                         float* __restrict__ dinv, int32_t* __restrict__ long_items,
                                                         int32_t* __restrict__ n_long, int item_cap) {
    __shared__ int lds[17];
    const int n = eff_count(d_n, n_host);
    if (blockIdx.x * blockDim.x >= n && blockIdx.x > 0) return;
    const int base_t = prefix_of_sums(bsum_t, blockIdx.x, lds);
    const int base_s = prefix_of_sums(bsum_s, blockIdx.x, lds);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int isbad = *bad;
    const int ct = i < n ? cnt_t[i] : 0;
    const int cs = i < n ? src_degree(i, grouped, cnt_s, nseg, seg_first, seg_last, loops, isbad) : 0;
    int tt, ts;
    const int pt = base_t + block_excl_scan(ct, lds, &tt);
    const int ps = base_s + block_excl_scan(cs, lds, &ts);
    if (i < n) {
        rowptr_t[i] = pt;
        rowptr_s[i] = ps;
        cnt_t[i] = pt;   // becomes the fill cursor
        cnt_s[i] = ps;
        dinv[i] = 1.0f / sqrtf((float)(ct + 1));   // deg = in-degree + unit self-loop
        if (long_items) {
            // work items (row, chunk) for rows longer than GRAPES_LONG_ROW; slot order is irrelevant
            // (each item owns its partial-sum slot, the combine walks a row's chunks in order)
            if (ct > GRAPES_LONG_ROW) {
                const int nc = (ct + GRAPES_LONG_ROW - 1) / GRAPES_LONG_ROW;
                const int b = atomicAdd(&n_long[0], nc);
                for (int c = 0; c < nc; ++c)
                    if (b + c < item_cap) { long_items[2 * (b + c)] = i; long_items[2 * (b + c) + 1] = c; }
            }
            if (cs > GRAPES_LONG_ROW) {
                const int nc = (cs + GRAPES_LONG_ROW - 1) / GRAPES_LONG_ROW;
                const int b = atomicAdd(&n_long[1], nc);
                for (int c = 0; c < nc; ++c)
                    if (b + c < item_cap) { long_items[2 * (item_cap + b + c)] = i; long_items[2 * (item_cap + b + c) + 1] = c; }
            }
        }
    }
    const bool last = (blockIdx.x + 1) * blockDim.x >= n;
    if (last && threadIdx.x == 0) {
        rowptr_t[n] = base_t + tt; rowptr_s[n] = base_s + ts;
        if (n_long) n_long[2] = base_t + tt;      // number of aggregated (non-self-loop) edges, for the caller's metric
    }
}

__global__ void prep_fill_k(const int32_t* __restrict__ es, const int32_t* __restrict__ ed, int e_host,
                            const int32_t* d_e, int n_host, const int32_t* d_n, int grouped,
                            int32_t* __restrict__ cur_t, int32_t* __restrict__ cur_s,
                            const int32_t* __restrict__ rowptr_s, const int32_t* __restrict__ seg_first,
                            const int32_t* __restrict__ loops, const int32_t* __restrict__ bad,
                            int32_t* __restrict__ tmp_src, int32_t* __restrict__ tmp_dst,
                            int32_t* __restrict__ csr_dst, int32_t* status) {
    const int e = eff_count(d_e, e_host);
    const int n = eff_count(d_n, n_host);
    const int isbad = *bad;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < e; t += gridDim.x * blockDim.x) {
        const int s = es[t], d = ed[t];
        if ((unsigned)s >= (unsigned)n || (unsigned)d >= (unsigned)n || s == d) continue;
        tmp_src[atomicAdd(&cur_t[d], 1)] = s;
        if (grouped) {
            // destinations ascend inside the segment => the dropped loop entries (d == s) precede t iff d > s
            if (!isbad) {
                const int p = rowptr_s[s] + (t - seg_first[s]) - (d > s ? loops[s] : 0);
                if ((unsigned)p < (unsigned)e_host) csr_dst[p] = d;          // never outside the array
                else if (status) atomicOr(status, GRAPES_STATUS_BAD_INDEX);  // destinations not ascending
            }
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

static inline int scan_blocks(int n) { return grapes_div_up(n > 0 ? n : 1, 1024); }

// Full-graph path (evaluation, eval.py:47-70): the adjacency already IS a CSR with ascending columns and no
// self-loops (graph.DeviceGraph.gcn_prepared strips them once), so gcn_norm reduces to dinv from the row
// lengths plus the hub-row work items — no histogram, no fill, no sort over 10^8 edges.
__global__ void prep_from_csr_k(const int32_t* __restrict__ rowptr, int n, float* __restrict__ dinv,
                                int32_t* __restrict__ items, int32_t* __restrict__ n_items, int item_cap) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int len = rowptr[i + 1] - rowptr[i];
        dinv[i] = 1.0f / sqrtf((float)(len + 1));
        if (items && len > GRAPES_LONG_ROW) {
            const int nc = (len + GRAPES_LONG_ROW - 1) / GRAPES_LONG_ROW;
            const int b = atomicAdd(n_items, nc);
            for (int c = 0; c < nc; ++c)
                if (b + c < item_cap) { items[2 * (b + c)] = i; items[2 * (b + c) + 1] = c; }
        }
    }
}

extern "C" int grapes_gcn_prepare_from_csr(const int32_t* rowptr, int32_t n, float* dinv, int32_t* items,
                                           int32_t* n_items, int32_t item_cap, grapes_stream_t stream) {
    if (n < 0 || !rowptr || !dinv || ((items == nullptr) != (n_items == nullptr))) return GRAPES_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (n_items) { hipError_t e = grapes_zero_async(n_items, sizeof(int32_t), s); if (e) return (int)e; }
    if (n == 0) return 0;
    int grid = grapes_div_up(n, 256); if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(prep_from_csr_k, dim3(grid), dim3(256), 0, s, rowptr, n, dinv, items, n_items, item_cap);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t grapes_gcn_prepare_workspace_bytes(int32_t n_cap, int32_t e_cap) {
    size_t n = (size_t)(n_cap > 0 ? n_cap : 0) + 1, e = (size_t)(e_cap > 0 ? e_cap : 0) + 1;
    return (6 * n + 2 * e + 2 * (size_t)scan_blocks(n_cap) + 4) * sizeof(int32_t);
}

/* capacity (in items) of each half of long_items; an item is (row, chunk) = 2 x int32 */
extern "C" int32_t grapes_gcn_long_items_capacity(int32_t e_cap) { return 2 * (e_cap / GRAPES_LONG_ROW) + 2; }

extern "C" int grapes_gcn_prepare(const int32_t* edge_src, const int32_t* edge_dst, int32_t e, const int32_t* d_e,
                                  int32_t n, const int32_t* d_n, int32_t flags, int32_t* rowptr_t, int32_t* csr_src,
                                  int32_t* rowptr_s, int32_t* csr_dst, float* dinv, int32_t* long_items,
                                  int32_t* n_long, void* workspace, int32_t* status, grapes_stream_t stream) {
    if (e < 0 || n < 0 || !rowptr_t || !rowptr_s || !dinv || !workspace) return GRAPES_EINVAL;
    if (e > 0 && (!edge_src || !edge_dst || !csr_src || !csr_dst)) return GRAPES_EINVAL;
    if ((long_items == nullptr) != (n_long == nullptr)) return GRAPES_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int grouped = (flags & GRAPES_PREP_SRC_GROUPED) ? 1 : 0;
    const size_t n1 = (size_t)n + 1;
    const int G = scan_blocks(n);
    int32_t* cnt_t = (int32_t*)workspace;
    int32_t* cnt_s = cnt_t + n1;
    int32_t* loops = cnt_s + n1;
    int32_t* nseg = loops + n1;
    int32_t* bad = nseg + n1;           // one word (+3 pad), zeroed with the block above it
    int32_t* seg_first = bad + 4;       // not zeroed: only read where nseg > 0
    int32_t* seg_last = seg_first + n1;
    int32_t* bsum_t = seg_last + n1;
    int32_t* bsum_s = bsum_t + G;
    int32_t* tmp_src = bsum_s + G;
    int32_t* tmp_dst = tmp_src + (size_t)e + 1;
    hipError_t err = grapes_zero_async(cnt_t, (4 * n1 + 4) * sizeof(int32_t), s);
    if (err != hipSuccess) return (int)err;
    if (grouped && e > 0) {   // slots a malformed list leaves unwritten must still hold a valid index
        err = grapes_zero_async(csr_dst, (size_t)e * sizeof(int32_t), s);
        if (err != hipSuccess) return (int)err;
    }
    int ge = grapes_div_up(e > 0 ? e : 1, 256); if (ge > 4096) ge = 4096;
    if (e > 0) {
        hipLaunchKernelGGL(prep_hist_k, dim3(ge), dim3(256), 0, s, edge_src, edge_dst, e, d_e, n, d_n, grouped, cnt_t, cnt_s,
                           seg_first, seg_last, loops, nseg, bad, status);
        GRAPES_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(prep_scan_count_k, dim3(G), dim3(1024), 0, s, n, d_n, grouped, (const int32_t*)cnt_t,
                       (const int32_t*)cnt_s, (const int32_t*)nseg, (const int32_t*)seg_first, (const int32_t*)seg_last,
                       (const int32_t*)loops, (const int32_t*)bad, bsum_t, bsum_s, n_long);
    GRAPES_LAUNCH_CHECK();
    hipLaunchKernelGGL(prep_scan_emit_k, dim3(G), dim3(1024), 0, s, n, d_n, grouped, cnt_t, cnt_s, (const int32_t*)nseg,
                       (const int32_t*)seg_first, (const int32_t*)seg_last, (const int32_t*)loops, (const int32_t*)bad, (const int32_t*)bsum_t,
                       (const int32_t*)bsum_s, rowptr_t, rowptr_s, dinv, long_items, n_long,
                       grapes_gcn_long_items_capacity(e));
    GRAPES_LAUNCH_CHECK();
    if (e > 0 && n > 0) {
        hipLaunchKernelGGL(prep_fill_k, dim3(ge), dim3(256), 0, s, edge_src, edge_dst, e, d_e, n, d_n, grouped, cnt_t, cnt_s,
                           (const int32_t*)rowptr_s, (const int32_t*)seg_first, (const int32_t*)loops, (const int32_t*)bad, tmp_src, tmp_dst,
                           csr_dst, status);
        GRAPES_LAUNCH_CHECK();
        const int both = grouped ? 0 : 1;
        int gr = grapes_div_up((both ? 2 : 1) * (int64_t)n, 256); if (gr > 4096) gr = 4096;
        hipLaunchKernelGGL(prep_sort_rows_k, dim3(gr), dim3(256), 0, s, n, d_n, both, (const int32_t*)rowptr_t,
                           (const int32_t*)rowptr_s, (const int32_t*)tmp_src, (const int32_t*)tmp_dst, csr_src, csr_dst);
        GRAPES_LAUNCH_CHECK();
    }
    return 0;
}
