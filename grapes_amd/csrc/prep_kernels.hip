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

// Rows of the resident feature matrix that the hop's gather-SpMM will read ~20 us from now (the batch rows: head_ids), touched
// by EXTRA workgroups of the build's first launch: the build is bound by dependent round trips and leaves HBM idle, the rows
// are cold (X is 1 GB, a step reads 40k random rows of it), and the 256 MB Infinity Cache is memory-side — a line fetched now
// by any compute unit is a cache hit for whichever unit gathers it later.  One 4-byte load per 64-byte sector of a row; the
// values only feed a never-true store, so that the loads are not optimised away.
struct PrefetchRows { const float* X; long long pitch; int row_floats; const int32_t* ids; int32_t* sink; };
__device__ __forceinline__ void prefetch_rows_body(const PrefetchRows& pf, int n, int block, int nblocks) {
    const int spr = (pf.row_floats * 4 + 63) / 64;                    // sectors per row
    const long long total = (long long)n * spr;
    unsigned acc = 0u;
    for (long long i = (long long)block * blockDim.x + threadIdx.x; i < total; i += (long long)nblocks * blockDim.x) {
        const int r = (int)(i / spr), k = (int)(i - (long long)r * spr);
        int c = 16 * k; if (c > pf.row_floats - 1) c = pf.row_floats - 1;
        acc += __float_as_uint(pf.X[(long long)pf.ids[r] * pf.pitch + c]);
    }
    if (acc == 0x7fc12345u && pf.sink) *pf.sink = 1;                   // (a NaN payload no sum of feature words is expected to hit; harmless if it does)
}


__global__ void prep_hist_k(const int32_t* __restrict__ es, const int32_t* __restrict__ ed, int e_host,
                            const int32_t* d_e, int n_host, const int32_t* d_n, int grouped,
                            int32_t* __restrict__ cnt_t, int32_t* __restrict__ cnt_s,
                            int32_t* __restrict__ seg_first, int32_t* __restrict__ seg_last,
                            int32_t* __restrict__ loops, int32_t* __restrict__ nseg, int32_t* __restrict__ bad,
                            int32_t* status, const int32_t* __restrict__ relabel, int32_t* __restrict__ n_long,
                            int ge, PrefetchRows pf) {
    const int e = eff_count(d_e, e_host);
    const int n = eff_count(d_n, n_host);
    if ((int)blockIdx.x >= ge) {              // helper workgroups: see prefetch_rows_body
        prefetch_rows_body(pf, n, (int)blockIdx.x - ge, (int)gridDim.x - ge);
        return;
    }
    if (n_long && blockIdx.x == 0 && threadIdx.x < 2) n_long[threadIdx.x] = 0;     // (when there is no init launch)
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < e; t += ge * blockDim.x) {
        // relabel != NULL: es / ed hold GLOBAL ids, mapped here (no init launch wrote relabelled copies); the segment
        // tests below compare raw ids, which is the same thing under an injective map
        const int sr = es[t];
        const int s = relabel ? relabel[sr] : sr, d = relabel ? relabel[ed[t]] : ed[t];
        if ((unsigned)s >= (unsigned)n || (unsigned)d >= (unsigned)n) {
            if (status) atomicOr(status, GRAPES_STATUS_BAD_INDEX);
            continue;
        }
        if (grouped) {
            if (t == 0 || es[t - 1] != sr) {
                seg_first[s] = t;
                if (atomicAdd(&nseg[s], 1) > 0) {   // a second segment for s: the list is not grouped
                    *bad = 1;
                    if (status) atomicOr(status, GRAPES_STATUS_BAD_INDEX);
                }
            }
            if (t == e - 1 || es[t + 1] != sr) seg_last[s] = t;
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
    // all four loads issued together (seg_first / seg_last of a row without a segment hold stale words: read, not used)
    const int ns = nseg[i], sl = seg_last[i], sf = seg_first[i], lp = loops[i];
    if (bad || ns <= 0) return 0;
    const int d = sl - sf + 1 - lp;
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
                                                         int32_t* __restrict__ n_long, int item_cap,
                                                         unsigned long long* __restrict__ sync, int32_t* status,
                                                         int gs = 0x7fffffff, PrefetchRows pf = PrefetchRows{nullptr, 0, 0, nullptr, nullptr}) {
    __shared__ int lds[17];
    __shared__ unsigned long long lds64;
    const int n = eff_count(d_n, n_host);
    if ((int)blockIdx.x >= gs) {              // helper workgroups of this launch: see prefetch_rows_body
        prefetch_rows_body(pf, n, (int)blockIdx.x - gs, (int)gridDim.x - gs);
        return;
    }
    if (blockIdx.x * blockDim.x >= n && blockIdx.x > 0) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int isbad = *bad;
    const int ct = i < n ? cnt_t[i] : 0;
    const int cs = i < n ? src_degree(i, grouped, cnt_s, nseg, seg_first, seg_last, loops, isbad) : 0;
    int tt, ts;
    int pt = block_excl_scan(ct, lds, &tt);
    int ps = block_excl_scan(cs, lds, &ts);
    int base_t, base_s;
    if (sync) {     // ONE launch: the workgroup totals (two edge counts < 2^31, packed) travel through the look-back scratch
        const int live = n > 0 ? (n + (int)blockDim.x - 1) / (int)blockDim.x : 1;
        const unsigned long long pre = lookback_exclusive(sync, blockIdx.x, ((unsigned long long)ts << 31) | (unsigned)tt,
                                                          &lds64, status);
        lookback_finish(sync, live);
        base_t = (int)(pre & 0x7fffffffull); base_s = (int)(pre >> 31);
    } else {
        base_t = prefix_of_sums(bsum_t, blockIdx.x, lds);
        base_s = prefix_of_sums(bsum_s, blockIdx.x, lds);
    }
    pt += base_t; ps += base_s;
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
                            int32_t* __restrict__ csr_dst, int32_t* status, const int32_t* __restrict__ relabel) {
    const int e = eff_count(d_e, e_host);
    const int n = eff_count(d_n, n_host);
    const int isbad = *bad;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < e; t += gridDim.x * blockDim.x) {
        const int s = relabel ? relabel[es[t]] : es[t], d = relabel ? relabel[ed[t]] : ed[t];
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

// The fill of a COUNTED build (include/grapes_hip.h: grapes_gcn_prepare_counted): the expansion's in-degree atomic already
// returned every entry's slot in its by-target row and the compaction wrote the row starts, so an edge is two relabel
// loads, two row-start loads and two stores — no atomic, no counters.  Helper workgroups (blockIdx >= ge): prefetch_rows_body.
__device__ __forceinline__ void prep_fill_slots_body(const int32_t* __restrict__ es, const int32_t* __restrict__ ed,
                                  const int32_t* __restrict__ slot, int e_host, const int32_t* d_e, int n_host,
                                  const int32_t* d_n, const int32_t* __restrict__ rowptr_t,
                                  const int32_t* __restrict__ rowptr_s, const int32_t* __restrict__ seg_first,
                                  const int32_t* __restrict__ loops, int32_t* __restrict__ tmp_src,
                                  int32_t* __restrict__ csr_dst, int32_t* status, const int32_t* __restrict__ relabel,
                                  int ge, PrefetchRows pf, int32_t* __restrict__ cursor, const int BID, const int NBLK) {
    const int e = eff_count(d_e, e_host);
    const int n = eff_count(d_n, n_host);
    if (BID >= ge) {
        prefetch_rows_body(pf, n, BID - ge, NBLK - ge);
        return;
    }
    for (int t = BID * blockDim.x + threadIdx.x; t < e; t += ge * blockDim.x) {
        const int sl = slot ? slot[t] : 0;
        const int s = relabel[es[t]], d = relabel[ed[t]];
        if ((unsigned)s >= (unsigned)n || (unsigned)d >= (unsigned)n) {
            if (status) atomicOr(status, GRAPES_STATUS_BAD_INDEX);
            continue;
        }
        if (sl < 0 || s == d) continue;                    // self-loop: replaced by the unit loop
        const int rs = rowptr_s[s], sf = seg_first[s], lp = loops[s];
        const int q = slot ? rowptr_t[d] + sl : atomicAdd(&cursor[d], 1);
        if ((unsigned)q < (unsigned)e_host) tmp_src[q] = s;
        else if (status) atomicOr(status, GRAPES_STATUS_BAD_INDEX);
        // destinations ascend inside the segment => the dropped loop entries (d == s) precede t iff d > s
        const int p = rs + (t - sf) - (d > s ? lp : 0);
        if ((unsigned)p < (unsigned)e_host) csr_dst[p] = d;
        else if (status) atomicOr(status, GRAPES_STATUS_BAD_INDEX);
    }
}

struct FillSlotsArgs {
    const int32_t* es; const int32_t* ed; const int32_t* slot; int e_host; const int32_t* d_e; int n_host; const int32_t* d_n;
    const int32_t* rowptr_t; const int32_t* rowptr_s; const int32_t* seg_first; const int32_t* loops; int32_t* tmp_src;
    int32_t* csr_dst; int32_t* status; const int32_t* relabel; int ge; PrefetchRows pf; int32_t* cursor;
};
#define FILL_SLOTS_CALL(A, bid, nblk)                                                                                           \
    prep_fill_slots_body((A).es, (A).ed, (A).slot, (A).e_host, (A).d_e, (A).n_host, (A).d_n, (A).rowptr_t, (A).rowptr_s,           \
                         (A).seg_first, (A).loops, (A).tmp_src, (A).csr_dst, (A).status, (A).relabel, (A).ge, (A).pf, (A).cursor, bid, nblk)
__global__ void prep_fill_slots_k(FillSlotsArgs a) { FILL_SLOTS_CALL(a, (int)blockIdx.x, (int)gridDim.x); }
// the fills of two graph builds side by side in one launch (riders: common.h)
__global__ void prep_fill_slots_pair_k(FillSlotsArgs a, FillSlotsArgs b, int nA) {
    if ((int)blockIdx.x < nA) FILL_SLOTS_CALL(a, (int)blockIdx.x, nA);
    else FILL_SLOTS_CALL(b, (int)blockIdx.x - nA, (int)gridDim.x - nA);
}

// Row heads for the fused gather-SpMM (spmm_kernels.hip): 12 words per by-target row
//   { len, gid_self, w_self = dinv[r]^2, dinv[r],  (gid_j, w_j = dinv[src_j] * dinv[r]) for the first four entries }
// with gid = head_ids[local id] (the row of the resident feature matrix).  A frontier row (1-3 entries) is then fully
// described by ONE 48-byte record: the aggregation needs two dependent memory round trips (head -> feature rows)
// instead of four (rowptr -> csr -> ids/dinv -> feature rows).  Unused entry slots: (gid_self, 0).
#define HEAD_WORDS 12
#define HEAD_ENTRIES 4
__device__ __forceinline__ void head_write_header(int32_t* __restrict__ row_head, const int32_t* __restrict__ head_ids,
                                                  const float* __restrict__ dinv, int r, int len) {
    int32_t* hd = row_head + (long long)r * HEAD_WORDS;
    const float dc = dinv[r];
    const int gid = head_ids[r];
    hd[0] = len; hd[1] = gid; hd[2] = __float_as_int(dc * dc); hd[3] = __float_as_int(dc);
    for (int j = len < HEAD_ENTRIES ? len : HEAD_ENTRIES; j < HEAD_ENTRIES; ++j) { hd[4 + 2 * j] = gid; hd[5 + 2 * j] = 0; }
}
__device__ __forceinline__ void head_write_entry(int32_t* __restrict__ row_head, const int32_t* __restrict__ head_ids,
                                                 const float* __restrict__ dinv, int r, int slot, int src) {
    int32_t* hd = row_head + (long long)r * HEAD_WORDS;
    hd[4 + 2 * slot] = head_ids[src];
    hd[5 + 2 * slot] = __float_as_int(dinv[src] * dinv[r]);
}

// Canonical (ascending) order inside every CSR row.  Rows [0,n) are the by-target rows, rows
// [n,2n) the by-source rows (skipped in grouped mode).  Short rows: one lane each (register
// insertion network); longer rows: the whole wavefront rank-sorts them.
#define SORT_SHORT 8
__device__ __forceinline__ void prep_sort_rows_body(int n_host, const int32_t* d_n, int both,
                                                        const int32_t* __restrict__ rowptr_t,
                                                        const int32_t* __restrict__ rowptr_s,
                                                        const int32_t* __restrict__ tmp_src,
                                                        const int32_t* __restrict__ tmp_dst,
                                                        int32_t* __restrict__ csr_src, int32_t* __restrict__ csr_dst,
                                                        const int32_t* __restrict__ head_ids,
                                                        const float* __restrict__ dinv, int32_t* __restrict__ row_head, const int BID, const int NBLK) {
    const int n = eff_count(d_n, n_host);
    const int total = both ? 2 * n : n;
    const int lane = lane_id();
    const int wave_global = (BID * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (NBLK * blockDim.x) >> 6;
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
            if (row_head && r < n) {
                // the WHOLE record at once: the four entries' ids and weights are requested together (an unused slot re-reads the
                // row's own) and the twelve words leave as three 16-byte stores — entry by entry (`if (i < len) head_write_entry`)
                // every entry was a dependent round trip of its own, then the header another
                int hg[HEAD_ENTRIES]; float hw[HEAD_ENTRIES];
#pragma unroll
                for (int i = 0; i < HEAD_ENTRIES; ++i) {
                    const int src = i < len ? v[i] : r;
                    hg[i] = head_ids[src]; hw[i] = dinv[src];
                }
                const float dc = dinv[r];
                const int gid = head_ids[r];
                int4* hd = reinterpret_cast<int4*>(row_head + (long long)r * HEAD_WORDS);
#pragma unroll
                for (int i = 0; i < HEAD_ENTRIES; ++i) { if (i >= len) { hg[i] = gid; hw[i] = 0.f; } else hw[i] = hw[i] * dc; }
                hd[0] = make_int4(len, gid, __float_as_int(dc * dc), __float_as_int(dc));
                hd[1] = make_int4(hg[0], __float_as_int(hw[0]), hg[1], __float_as_int(hw[1]));
                hd[2] = make_int4(hg[2], __float_as_int(hw[2]), hg[3], __float_as_int(hw[3]));
            }
        } else if (row_head && r < n) head_write_header(row_head, head_ids, dinv, r, len);
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
                if (row_head && rr < n && rank < HEAD_ENTRIES) head_write_entry(row_head, head_ids, dinv, rr, rank, v);
            }
        }
    }
}

struct SortRowsArgs {
    int n_host; const int32_t* d_n; int both; const int32_t* rowptr_t; const int32_t* rowptr_s; const int32_t* tmp_src;
    const int32_t* tmp_dst; int32_t* csr_src; int32_t* csr_dst; const int32_t* head_ids; const float* dinv; int32_t* row_head;
};
#define SORT_ROWS_CALL(A, bid, nblk)                                                                                            \
    prep_sort_rows_body((A).n_host, (A).d_n, (A).both, (A).rowptr_t, (A).rowptr_s, (A).tmp_src, (A).tmp_dst, (A).csr_src, (A).csr_dst, \
                        (A).head_ids, (A).dinv, (A).row_head, bid, nblk)
__global__ __launch_bounds__(256) void prep_sort_rows_k(SortRowsArgs a) { SORT_ROWS_CALL(a, (int)blockIdx.x, (int)gridDim.x); }
// the row orders of two graph builds side by side in one launch (riders: common.h)
__global__ __launch_bounds__(256) void prep_sort_rows_pair_k(SortRowsArgs a, SortRowsArgs b, int nA) {
    if ((int)blockIdx.x < nA) SORT_ROWS_CALL(a, (int)blockIdx.x, nA);
    else SORT_ROWS_CALL(b, (int)blockIdx.x - nA, (int)gridDim.x - nA);
}

#ifdef GRAPES_DIAG   // (measured slower than the launches it replaces: an A/B form of the diagnostic build only)
// ---- the grouped, pre-zeroed build of a hop graph as ONE cooperative launch (grid <= compute units, 1024 threads): the four
// phases of the general path — per-edge counts, row-pointer scan, fill, canonical row order + head records — separated by grid
// barriers instead of launch boundaries (a dependent launch costs ~4.5 us however little it does; a barrier ~1.5 us).  A
// thread keeps its edges (relabelled endpoints and the SLOT the in-degree atomic returned: the entry's place in its row, so the
// fill needs no second atomic) in registers from phase to phase.  Everything a later phase of ANOTHER workgroup reads goes
// through agent-scope accesses or atomics (common.h, grid_barrier).  Outputs are those of the four-launch path, bit for bit.
#define FUSED_EPT 4
__global__ __launch_bounds__(1024) void prep_fused_k(const int32_t* __restrict__ es, const int32_t* __restrict__ ed, int e_host,
                                                     const int32_t* d_e, int n_host, const int32_t* d_n,
                                                     const int32_t* __restrict__ relabel,
                                                     int32_t* cnt_t, int32_t* loops, int32_t* nseg, int32_t* bad,
                                                     int32_t* seg_first, int32_t* seg_last,
                                                     int32_t* tmp_src, int32_t* sp_s, int32_t* sp_d, int32_t* sp_slot,
                                                     int32_t* rowptr_t, int32_t* rowptr_s, float* dinv,
                                                     int32_t* __restrict__ csr_src, int32_t* __restrict__ csr_dst,
                                                     int32_t* __restrict__ long_items, int32_t* n_long, int item_cap,
                                                     const int32_t* __restrict__ head_ids, int32_t* __restrict__ row_head,
                                                     unsigned long long* sync, unsigned* bar, int32_t* status) {
    __shared__ int lds[17];
    __shared__ unsigned long long lds64;
    const int e = eff_count(d_e, e_host);
    const int n = eff_count(d_n, n_host);
    const int tid = threadIdx.x;
    const int T = gridDim.x * blockDim.x;
    const int gt = blockIdx.x * blockDim.x + tid;
    if (n_long && blockIdx.x == 0 && tid < 2) (void)atomicExch(&n_long[tid], 0);           // item counters of phase 2
    // ---------------- phase 1: in-degrees (the atomic's return value is the entry's slot in its row), source segments, loops
    int c_s[FUSED_EPT], c_d[FUSED_EPT], c_slot[FUSED_EPT];
    {
        int k = 0;
        for (int t = gt; t < e; t += T, ++k) {
            const int sr = es[t], dr = ed[t];
            const int pv = t > 0 ? es[t - 1] : -1, nx = t + 1 < e ? es[t + 1] : -1;
            const int s = relabel ? relabel[sr] : sr, d = relabel ? relabel[dr] : dr;
            int slot = -1;
            int ss = s, dd = d;
            if ((unsigned)s >= (unsigned)n || (unsigned)d >= (unsigned)n) {
                if (status) atomicOr(status, GRAPES_STATUS_BAD_INDEX);
                ss = -1; dd = -1;
            } else {
                if (t == 0 || pv != sr) {
                    st_agent(&seg_first[s], t);
                    if (atomicAdd(&nseg[s], 1) > 0) {               // a second segment for s: the list is not grouped
                        (void)atomicExch(bad, 1);
                        if (status) atomicOr(status, GRAPES_STATUS_BAD_INDEX);
                    }
                }
                if (t == e - 1 || nx != sr) st_agent(&seg_last[s], t);
                if (s == d) (void)atomicAdd(&loops[s], 1);          // add_remaining_self_loops: replaced by the unit loop
                else slot = atomicAdd(&cnt_t[d], 1);
            }
            if (k < FUSED_EPT) {
#pragma unroll
                for (int q = 0; q < FUSED_EPT; ++q) if (q == k) { c_s[q] = ss; c_d[q] = dd; c_slot[q] = slot; }
            } else { sp_s[t] = ss; sp_d[t] = dd; sp_slot[t] = slot; }     // (re-read by this thread only)
        }
    }
    grid_barrier(bar, 1u, status);
    // ---------------- phase 2: row pointers, dinv, long-row work items (prep_scan_emit_k over tiles of 1024 nodes)
    const int isbad = ld_agent(bad);
    const int live = n > 0 ? (n + 1023) / 1024 : 1;
    for (int tile = blockIdx.x; tile < live; tile += gridDim.x) {
        const int i = tile * 1024 + tid;
        int ct = 0, cs = 0;
        if (i < n) {
            ct = ld_agent(&cnt_t[i]);
            const int ns = ld_agent(&nseg[i]), sl = ld_agent(&seg_last[i]), sf = ld_agent(&seg_first[i]), lp = ld_agent(&loops[i]);
            if (!isbad && ns > 0) { const int dg = sl - sf + 1 - lp; cs = dg > 0 ? dg : 0; }
        }
        int tt, ts;
        int pt = block_excl_scan(ct, lds, &tt);
        int ps = block_excl_scan(cs, lds, &ts);
        const unsigned long long pre = lookback_exclusive(sync, tile, ((unsigned long long)ts << 31) | (unsigned)tt, &lds64, status);
        lookback_finish(sync, live);
        const int base_t = (int)(pre & 0x7fffffffull), base_s = (int)(pre >> 31);
        pt += base_t; ps += base_s;
        if (i < n) {
            st_agent(&rowptr_t[i], pt);
            st_agent(&rowptr_s[i], ps);
            st_agent_f(&dinv[i], 1.0f / sqrtf((float)(ct + 1)));       // deg = in-degree + unit self-loop
            if (long_items) {
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
        if (tile == live - 1 && tid == 0) {
            st_agent(&rowptr_t[n], base_t + tt); st_agent(&rowptr_s[n], base_s + ts);
            if (n_long) n_long[2] = base_t + tt;      // number of aggregated (non-self-loop) edges, for the caller's metric
        }
        __syncthreads();                               // (lds / lds64 are reused by the next tile)
    }
    grid_barrier(bar, 2u, status);
    // ---------------- phase 3: fill (by-target rows unsorted into tmp_src at row start + slot, by-source rows directly)
    {
        int k = 0;
        for (int t = gt; t < e; t += T, ++k) {
            int s, d, slot;
            if (k < FUSED_EPT) {
                s = c_s[0]; d = c_d[0]; slot = c_slot[0];
#pragma unroll
                for (int q = 1; q < FUSED_EPT; ++q) if (q == k) { s = c_s[q]; d = c_d[q]; slot = c_slot[q]; }
            } else { s = sp_s[t]; d = sp_d[t]; slot = sp_slot[t]; }
            if (slot < 0) continue;                                    // bad index or loop
            st_agent(&tmp_src[ld_agent(&rowptr_t[d]) + slot], s);
            if (!isbad) {
                // destinations ascend inside the segment => the dropped loop entries (d == s) precede t iff d > s
                const int p = ld_agent(&rowptr_s[s]) + (t - ld_agent(&seg_first[s])) - (d > s ? ld_agent(&loops[s]) : 0);
                if ((unsigned)p < (unsigned)e_host) csr_dst[p] = d;          // never outside the array
                else if (status) atomicOr(status, GRAPES_STATUS_BAD_INDEX);  // destinations not ascending
            }
        }
    }
    grid_barrier(bar, 3u, status);
    grid_barrier_finish(bar);
    // ---------------- phase 4: canonical (ascending) order of the by-target rows + head records (prep_sort_rows_k)
    {
        const int lane = lane_id();
        const int wave_global = gt >> 6, nwaves = T >> 6;
        for (int base = wave_global * 64; base < n; base += nwaves * 64) {
            const int r = base + lane;
            int beg = 0, len = 0;
            if (r < n) { beg = ld_agent(&rowptr_t[r]); len = ld_agent(&rowptr_t[r + 1]) - beg; }
            if (len > 0 && len <= SORT_SHORT) {
                int v[SORT_SHORT];
#pragma unroll
                for (int i = 0; i < SORT_SHORT; ++i) v[i] = i < len ? ld_agent(&tmp_src[beg + i]) : 0x7fffffff;
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
                    if (i < len) csr_src[beg + i] = v[i];
                if (row_head) {
                    int32_t* hd = row_head + (long long)r * HEAD_WORDS;
                    const float dr = ld_agent_f(&dinv[r]);
#pragma unroll
                    for (int i = 0; i < HEAD_ENTRIES; ++i)
                        if (i < len) { hd[4 + 2 * i] = head_ids[v[i]]; hd[5 + 2 * i] = __float_as_int(ld_agent_f(&dinv[v[i]]) * dr); }
                }
            }
            if (row_head && r < n) {
                int32_t* hd = row_head + (long long)r * HEAD_WORDS;
                const float dc = ld_agent_f(&dinv[r]);
                const int gid = head_ids[r];
                hd[0] = len; hd[1] = gid; hd[2] = __float_as_int(dc * dc); hd[3] = __float_as_int(dc);
                for (int j = len < HEAD_ENTRIES ? len : HEAD_ENTRIES; j < HEAD_ENTRIES; ++j) { hd[4 + 2 * j] = gid; hd[5 + 2 * j] = 0; }
            }
            unsigned long long longs = __ballot(len > SORT_SHORT);
            while (longs) {
                const int l = __ffsll((long long)longs) - 1;
                longs &= longs - 1;
                const int lbeg = __shfl(beg, l, 64);
                const int llen = __shfl(len, l, 64);
                const int rr = base + l;
                const float drr = row_head ? ld_agent_f(&dinv[rr]) : 0.f;
                for (int i = lane; i < llen; i += 64) {
                    const int v = ld_agent(&tmp_src[lbeg + i]);
                    int rank = 0;
                    for (int j = 0; j < llen; ++j) {
                        const int u = ld_agent(&tmp_src[lbeg + j]);
                        rank += (u < v) || (u == v && j < i);
                    }
                    csr_src[lbeg + rank] = v;
                    if (row_head && rank < HEAD_ENTRIES) {
                        int32_t* hd = row_head + (long long)rr * HEAD_WORDS;
                        hd[4 + 2 * rank] = head_ids[v]; hd[5 + 2 * rank] = __float_as_int(ld_agent_f(&dinv[v]) * drr);
                    }
                }
            }
        }
    }
}
#endif  // GRAPES_DIAG

// First launch of the general build: clears the counters (and, in grouped mode, csr_dst: slots a malformed list
// leaves unwritten must still hold a valid index) and relabels the edge list through node_map (main.py:195,254 —
// the TensorMap lookup of both endpoint rows) in the same pass.
__global__ void prep_init_k(const int32_t* __restrict__ es, const int32_t* __restrict__ ed, int e_host,
                            const int32_t* d_e, const int32_t* __restrict__ node_map, int32_t* __restrict__ rl_src,
                            int32_t* __restrict__ rl_dst, int32_t* __restrict__ counters, size_t counter_words,
                            int32_t* __restrict__ csr_dst, int clear_dst, int32_t* __restrict__ n_long) {
    const int e = eff_count(d_e, e_host);
    if (n_long && blockIdx.x == 0 && threadIdx.x < 2) n_long[threadIdx.x] = 0;     // item counters of the scan launch
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = i0; i < counter_words; i += stride) counters[i] = 0;
    for (size_t t = i0; t < (size_t)e_host; t += stride) {
        if (clear_dst) csr_dst[t] = 0;
        if (node_map && t < (size_t)e) { rl_src[t] = node_map[es[t]]; rl_dst[t] = node_map[ed[t]]; }
    }
}

// Whole build in ONE workgroup for small graphs in grouped mode (the classifier's sampled subgraphs: <= B + hops*K
// nodes, a few hundred edges): counters live in LDS, the two passes over the edge list, the scan and the row sort are
// separated by workgroup barriers instead of launches.  Same outputs as the general path.
#define SMALL_N 2048
#define SMALL_T 1024
// The slice filter's OUTPUT side inside the classifier's graph build (see frontier_expand_fused_k: slice_stage): the hop's
// expansion left, per 64-edge wavefront-block, its surviving edges in edge order with their multiplicities; this workgroup
// scans the blocks' summed multiplicities and writes the edge list (src, dst) x multiplicity in expansion order — what
// slice_emit_k produced in a launch of its own between two hops.  Returns the number of edges (clamped to out_cap).
__device__ __forceinline__ int slice_assemble(const int32_t* __restrict__ stage, int fe_cap, const int32_t* d_fe, int out_cap,
                                              int32_t* __restrict__ out_src, int32_t* __restrict__ out_dst, int32_t* d_out_count,
                                              int32_t* status, int* lds) {
    const int tid = threadIdx.x;
    const int fe = eff_count(d_fe, fe_cap);
    const int nwb_cap = (fe_cap + 63) >> 6, nwb = (fe + 63) >> 6;
    const int32_t* wcnt = stage; const int32_t* wsum = stage + nwb_cap;
    const int32_t* st_s = stage + 2 * nwb_cap; const int32_t* st_d = st_s + fe_cap; const int32_t* st_c = st_d + fe_cap;
    const int per = (nwb + SMALL_T - 1) / SMALL_T;                      // consecutive wavefront-blocks per thread (2 at 131k edges)
    const int w0 = tid * per;
    int mine = 0;
    for (int k = 0; k < per; ++k) { const int wb = w0 + k; if (wb < nwb) mine += wsum[wb]; }
    int tot;
    int pos = block_excl_scan(mine, lds, &tot);
    bool overflow = false;
    if (mine > 0) {
        for (int k = 0; k < per; ++k) {
            const int wb = w0 + k;
            if (wb >= nwb) break;
            const int cnt = wcnt[wb];
            for (int i = 0; i < cnt; ++i) {
                const int q = wb * 64 + i;
                const int sv = st_s[q], dv = st_d[q], c = st_c[q];
                for (int r = 0; r < c; ++r, ++pos) {
                    if (pos < out_cap) { out_src[pos] = sv; out_dst[pos] = dv; }
                    else overflow = true;
                }
            }
        }
    }
    if (overflow && status) atomicOr(status, GRAPES_STATUS_EDGE_OVERFLOW);
    const int total = tot < out_cap ? tot : out_cap;
    if (tid == 0 && d_out_count) *d_out_count = total;
    __syncthreads();                                   // the list is in place (this workgroup's own stores, waited for) before it is read
    return total;
}

__device__ __forceinline__ void prep_small_body(const int32_t* __restrict__ es, const int32_t* __restrict__ ed,
                                                int e_host, const int32_t* d_e, int n_host, const int32_t* d_n,
                                                const int32_t* __restrict__ node_map,
                                                int32_t* __restrict__ rowptr_t, int32_t* __restrict__ csr_src,
                                                int32_t* __restrict__ rowptr_s, int32_t* __restrict__ csr_dst,
                                                float* __restrict__ dinv, int32_t* __restrict__ long_items,
                                                int32_t* __restrict__ n_long, int item_cap,
                                                int32_t* __restrict__ tmp_src, int32_t* status,
                                                const int32_t* __restrict__ head_ids,
                                                int32_t* __restrict__ row_head, const int32_t* __restrict__ slice_stage = nullptr,
                                                int fe_cap = 0, const int32_t* d_fe = nullptr) {
    __shared__ int cnt_t[SMALL_N], segf[SMALL_N], segl[SMALL_N], loops[SMALL_N], nseg[SMALL_N], rps[SMALL_N];
    __shared__ int lds[17];
    __shared__ int s_bad, s_nlong_rows;
    __shared__ int long_rows[SMALL_N];
    const int tid = threadIdx.x;
    // slice_stage: es / ed / d_e are this build's OUTPUTS first (the filtered edge list, assembled here), then its inputs
    const int e = slice_stage ? slice_assemble(slice_stage, fe_cap, d_fe, e_host, const_cast<int32_t*>(es), const_cast<int32_t*>(ed),
                                               const_cast<int32_t*>(d_e), status, lds)
                              : eff_count(d_e, e_host);
    const int n = eff_count(d_n, n_host);          // <= SMALL_N (checked on the host against the capacity)
    for (int i = tid; i < n; i += SMALL_T) { cnt_t[i] = 0; loops[i] = 0; nseg[i] = 0; }
    if (tid == 0) { s_bad = 0; s_nlong_rows = 0; if (n_long) { n_long[0] = 0; n_long[1] = 0; } }
    // a thread's first two edges stay in registers between the two passes over the edge list (their ids and relabelled
    // endpoints are two dependent memory round trips each time; the classifier's graphs have a few hundred edges)
    constexpr int EPT = 2;
    int c_gs[EPT], c_s[EPT], c_d[EPT], c_pv[EPT], c_nx[EPT];
    // (the eight id loads of a thread's two edges together, then their four relabel loads together: every index is clamped into the
    // list — an empty list reads entry 0 of the capacity, e_host >= 1 — and the "no neighbour" / "no edge" cases are selects on the
    // loaded values.  With `e > 0 ? es[tc] : 0` and its like between the loads each edge's ids were a round trip of their own.)
    int r_gd[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) { c_gs[k] = 0; c_pv[k] = 0; c_nx[k] = 0; r_gd[k] = 0; }
    if (e_host > 0) {          // (uniform; an edge list without capacity may have no storage at all)
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int t = tid + k * SMALL_T;
            const int tc = t < e ? t : 0;
            const int tp = tc > 0 ? tc - 1 : 0, tn = tc + 1 < e ? tc + 1 : tc;
            c_gs[k] = es[tc]; c_pv[k] = es[tp]; c_nx[k] = es[tn]; r_gd[k] = ed[tc];
        }
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int t = tid + k * SMALL_T;
        const int tc = t < e ? t : 0;
        c_gs[k] = e > 0 ? c_gs[k] : 0;
        c_pv[k] = (e > 0 && tc > 0) ? c_pv[k] : -1;                  // (ids are >= 0: -1 = "no neighbour")
        c_nx[k] = (e > 0 && tc + 1 < e) ? c_nx[k] : -1;
        r_gd[k] = e > 0 ? r_gd[k] : 0;
    }
    if (node_map) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) { c_s[k] = node_map[c_gs[k]]; c_d[k] = node_map[r_gd[k]]; }
    } else {
#pragma unroll
        for (int k = 0; k < EPT; ++k) { c_s[k] = c_gs[k]; c_d[k] = r_gd[k]; }
    }
    __syncthreads();
    // ---- pass 1: in-degrees, source segments, loops
    for (int t = tid; t < e; t += SMALL_T) {
        csr_dst[t] = 0;
        const int kk = (t - tid) / SMALL_T;
        int gs, s, d;
        if (kk < EPT) { gs = kk == 0 ? c_gs[0] : c_gs[1]; s = kk == 0 ? c_s[0] : c_s[1]; d = kk == 0 ? c_d[0] : c_d[1]; }
        else { gs = es[t]; s = node_map ? node_map[gs] : gs; d = node_map ? node_map[ed[t]] : ed[t]; }
        if ((unsigned)s >= (unsigned)n || (unsigned)d >= (unsigned)n) {
            if (status) atomicOr(status, GRAPES_STATUS_BAD_INDEX);
            continue;
        }
        const int pv = kk < EPT ? (kk == 0 ? c_pv[0] : c_pv[1]) : (t > 0 ? es[t - 1] : -1);
        const int nx = kk < EPT ? (kk == 0 ? c_nx[0] : c_nx[1]) : (t + 1 < e ? es[t + 1] : -1);
        if (t == 0 || pv != gs) {
            segf[s] = t;
            if (atomicAdd(&nseg[s], 1) > 0) { s_bad = 1; if (status) atomicOr(status, GRAPES_STATUS_BAD_INDEX); }
        }
        if (t == e - 1 || nx != gs) segl[s] = t;
        if (s == d) { atomicAdd(&loops[s], 1); continue; }
        atomicAdd(&cnt_t[d], 1);
    }
    __syncthreads();
    // ---- scan (two nodes per thread)
    const int isbad = s_bad;
    const int i0 = 2 * tid, i1 = 2 * tid + 1;
    int ct0 = 0, ct1 = 0, cs0 = 0, cs1 = 0;
    if (i0 < n) { ct0 = cnt_t[i0]; if (!isbad && nseg[i0] > 0) { const int v = segl[i0] - segf[i0] + 1 - loops[i0]; cs0 = v > 0 ? v : 0; } }
    if (i1 < n) { ct1 = cnt_t[i1]; if (!isbad && nseg[i1] > 0) { const int v = segl[i1] - segf[i1] + 1 - loops[i1]; cs1 = v > 0 ? v : 0; } }
    int tt, ts;
    const int pt = block_excl_scan(ct0 + ct1, lds, &tt);
    const int ps = block_excl_scan(cs0 + cs1, lds, &ts);
    __syncthreads();                               // every thread has read its cnt_t entries: they become cursors
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = k ? i1 : i0;
        if (i >= n) continue;
        const int ct = k ? ct1 : ct0, cs = k ? cs1 : cs0;
        const int bt = k ? pt + ct0 : pt, bs = k ? ps + cs0 : ps;
        rowptr_t[i] = bt; rowptr_s[i] = bs;
        cnt_t[i] = bt; rps[i] = bs;
        dinv[i] = 1.0f / sqrtf((float)(ct + 1));
        if (ct > SORT_SHORT) long_rows[atomicAdd(&s_nlong_rows, 1)] = i;
        if (long_items) {
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
    if (tid == 0) { rowptr_t[n] = tt; rowptr_s[n] = ts; if (n_long) n_long[2] = tt; }
    __syncthreads();
    // ---- pass 2: fill (by-target rows unsorted into tmp_src, by-source rows directly)
    for (int t = tid; t < e; t += SMALL_T) {
        const int kk = (t - tid) / SMALL_T;
        int s, d;
        if (kk < EPT) { s = kk == 0 ? c_s[0] : c_s[1]; d = kk == 0 ? c_d[0] : c_d[1]; }
        else { const int gs = es[t]; s = node_map ? node_map[gs] : gs; d = node_map ? node_map[ed[t]] : ed[t]; }
        if ((unsigned)s >= (unsigned)n || (unsigned)d >= (unsigned)n || s == d) continue;
        tmp_src[atomicAdd(&cnt_t[d], 1)] = s;
        if (!isbad) {
            const int p = rps[s] + (t - segf[s]) - (d > s ? loops[s] : 0);
            if ((unsigned)p < (unsigned)e_host) csr_dst[p] = d;
            else if (status) atomicOr(status, GRAPES_STATUS_BAD_INDEX);
        }
    }
    __threadfence_block();
    __syncthreads();
    // ---- canonical order of the by-target rows: short rows one thread each, longer rows by the whole workgroup
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = k ? i1 : i0;
        const int len = i < n ? (k ? ct1 : ct0) : 0;
        if (len > 0 && len <= SORT_SHORT) {
            const int beg = k ? pt + ct0 : pt;
            int v[SORT_SHORT];
#pragma unroll
            for (int q = 0; q < SORT_SHORT; ++q) v[q] = q < len ? tmp_src[beg + q] : 0x7fffffff;
#pragma unroll
            for (int q = 1; q < SORT_SHORT; ++q) {
#pragma unroll
                for (int j = q; j > 0; --j) {
                    const int a = v[j - 1], b = v[j];
                    v[j - 1] = a < b ? a : b;
                    v[j] = a < b ? b : a;
                }
            }
#pragma unroll
            for (int q = 0; q < SORT_SHORT; ++q)
                if (q < len) csr_src[beg + q] = v[q];
            if (row_head) {        // the whole record at once (prep_sort_rows_body: the entries' ids and weights requested together)
                int hg[HEAD_ENTRIES]; float hw[HEAD_ENTRIES];
#pragma unroll
                for (int q = 0; q < HEAD_ENTRIES; ++q) { const int src = q < len ? v[q] : i; hg[q] = head_ids[src]; hw[q] = dinv[src]; }
                const float dc = dinv[i];
                const int gid = head_ids[i];
                int4* hd = reinterpret_cast<int4*>(row_head + (long long)i * HEAD_WORDS);
#pragma unroll
                for (int q = 0; q < HEAD_ENTRIES; ++q) { if (q >= len) { hg[q] = gid; hw[q] = 0.f; } else hw[q] = hw[q] * dc; }
                hd[0] = make_int4(len, gid, __float_as_int(dc * dc), __float_as_int(dc));
                hd[1] = make_int4(hg[0], __float_as_int(hw[0]), hg[1], __float_as_int(hw[1]));
                hd[2] = make_int4(hg[2], __float_as_int(hw[2]), hg[3], __float_as_int(hw[3]));
            }
        } else if (row_head && i < n) head_write_header(row_head, head_ids, dinv, i, len);
    }
    const int nl = s_nlong_rows;                   // rows with more than SORT_SHORT entries: one wavefront each, round-robin
    const int lane = lane_id(), wid = tid >> 6;
    for (int q = wid; q < nl; q += SMALL_T / 64) {
        const int r = long_rows[q];
        const int beg = rowptr_t[r], len = rowptr_t[r + 1] - beg;               // written by this workgroup above
        for (int i = lane; i < len; i += 64) {
            const int v = tmp_src[beg + i];
            int rank = 0;
            for (int j = 0; j < len; ++j) {
                const int u = tmp_src[beg + j];
                rank += (u < v) || (u == v && j < i);
            }
            csr_src[beg + rank] = v;
            if (row_head && rank < HEAD_ENTRIES) head_write_entry(row_head, head_ids, dinv, r, rank, v);
        }
    }
}

__global__ __launch_bounds__(SMALL_T) void prep_small_k(const int32_t* es, const int32_t* ed, int e_host, const int32_t* d_e,
                                                        int n_host, const int32_t* d_n, const int32_t* node_map,
                                                        int32_t* rowptr_t, int32_t* csr_src, int32_t* rowptr_s,
                                                        int32_t* csr_dst, float* dinv, int32_t* long_items, int32_t* n_long,
                                                        int item_cap, int32_t* tmp_src, int32_t* status,
                                                        const int32_t* head_ids, int32_t* row_head) {
    prep_small_body(es, ed, e_host, d_e, n_host, d_n, node_map, rowptr_t, csr_src, rowptr_s, csr_dst, dinv, long_items, n_long,
                    item_cap, tmp_src, status, head_ids, row_head);
}

// Several small graphs over the SAME node set in one launch, one workgroup each (the classifier's per-layer subgraphs).
#define SMALL_BATCH_MAX 8
struct SmallGraph {
    const int32_t* es; const int32_t* ed; const int32_t* d_e; int e_host;
    int32_t* rowptr_t; int32_t* csr_src; int32_t* rowptr_s; int32_t* csr_dst; float* dinv;
    int32_t* long_items; int32_t* n_long; int32_t* tmp_src; int32_t* row_head; int item_cap;
    const int32_t* slice_stage; const int32_t* d_fe; int fe_cap;      // optional: the edge list is assembled from a slice stage
};
struct SmallBatch { SmallGraph g[SMALL_BATCH_MAX]; };
__global__ __launch_bounds__(SMALL_T) void prep_small_batch_k(SmallBatch b, int n_host, const int32_t* d_n,
                                                              const int32_t* node_map, int32_t* status,
                                                              const int32_t* head_ids) {
    const SmallGraph& q = b.g[blockIdx.x];
    prep_small_body(q.es, q.ed, q.e_host, q.d_e, n_host, d_n, node_map, q.rowptr_t, q.csr_src, q.rowptr_s, q.csr_dst, q.dinv,
                    q.long_items, q.n_long, q.item_cap, q.tmp_src, status, head_ids, q.row_head, q.slice_stage, q.fe_cap, q.d_fe);
}

// the classifier's per-layer graphs (one workgroup each) with a recorded fill riding beside them (riders: common.h)
__global__ __launch_bounds__(SMALL_T) void prep_small_batch_fill_pair_k(SmallBatch b, int n_host, const int32_t* d_n,
                                                                        const int32_t* node_map, int32_t* status,
                                                                        const int32_t* head_ids, FillSlotsArgs f, int nA) {
    if ((int)blockIdx.x < nA) {
        const SmallGraph& q = b.g[blockIdx.x];
        prep_small_body(q.es, q.ed, q.e_host, q.d_e, n_host, d_n, node_map, q.rowptr_t, q.csr_src, q.rowptr_s, q.csr_dst, q.dinv,
                        q.long_items, q.n_long, q.item_cap, q.tmp_src, status, head_ids, q.row_head, q.slice_stage, q.fe_cap, q.d_fe);
    } else {
        FILL_SLOTS_CALL(f, (int)blockIdx.x - nA, (int)gridDim.x - nA);
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
    return (6 * n + 4 * e + 2 * (size_t)scan_blocks(n_cap) + 4) * sizeof(int32_t);
}

/* capacity (in items) of each half of long_items; an item is (row, chunk) = 2 x int32 */
extern "C" int32_t grapes_gcn_long_items_capacity(int32_t e_cap) { return 2 * (e_cap / GRAPES_LONG_ROW) + 2; }

extern "C" size_t grapes_gcn_prepare_zero_words(int32_t n) { return 4 * ((size_t)(n > 0 ? n : 0) + 1) + 4; }

struct PrefetchReq { const float* X; long long pitch; int row_floats; };
static int prefetch_in_scan() {       // which launch of the build carries the helpers: 0 = the first (histogram), 1 = the scan
    static int v = -1;
    if (v < 0) { const char* e = grapes_tune_env("GRAPES_PREFETCH_IN_SCAN"); v = e ? atoi(e) : 0; }      // (measured the same either way: 0.580 / 0.583 ms)
    return v;
}

static int gcn_prepare_impl(const int32_t* edge_src, const int32_t* edge_dst, int32_t e, const int32_t* d_e,
                            const int32_t* node_map, int32_t n, const int32_t* d_n, int32_t flags,
                            int32_t* rowptr_t, int32_t* csr_src, int32_t* rowptr_s, int32_t* csr_dst, float* dinv,
                            int32_t* long_items, int32_t* n_long, const int32_t* head_ids, int32_t* row_head,
                            void* workspace, uint64_t* sync, int32_t* status, grapes_stream_t stream, PrefetchReq pfreq);
extern "C" int grapes_gcn_prepare(const int32_t* edge_src, const int32_t* edge_dst, int32_t e, const int32_t* d_e,
                                  const int32_t* node_map, int32_t n, const int32_t* d_n, int32_t flags,
                                  int32_t* rowptr_t, int32_t* csr_src, int32_t* rowptr_s, int32_t* csr_dst, float* dinv,
                                  int32_t* long_items, int32_t* n_long, const int32_t* head_ids, int32_t* row_head,
                                  void* workspace, uint64_t* sync, int32_t* status, grapes_stream_t stream) {
    return gcn_prepare_impl(edge_src, edge_dst, e, d_e, node_map, n, d_n, flags, rowptr_t, csr_src, rowptr_s, csr_dst, dinv, long_items,
                            n_long, head_ids, row_head, workspace, sync, status, stream, PrefetchReq{nullptr, 0, 0});
}
/* grapes_gcn_prepare that ALSO touches the feature rows  prefetch_X[head_ids[r], 0:prefetch_row_floats]  (row pitch prefetch_pitch
 * floats) from extra workgroups of its first launch (general path, head records requested): the rows the hop's gather-SpMM reads
 * next are then Infinity-Cache hits.  No state is kept between calls. */
extern "C" int grapes_gcn_prepare_prefetching(const int32_t* edge_src, const int32_t* edge_dst, int32_t e, const int32_t* d_e,
                                              const int32_t* node_map, int32_t n, const int32_t* d_n, int32_t flags,
                                              int32_t* rowptr_t, int32_t* csr_src, int32_t* rowptr_s, int32_t* csr_dst, float* dinv,
                                              int32_t* long_items, int32_t* n_long, const int32_t* head_ids, int32_t* row_head,
                                              void* workspace, uint64_t* sync, int32_t* status, const float* prefetch_X,
                                              int64_t prefetch_pitch, int32_t prefetch_row_floats, grapes_stream_t stream) {
    if (prefetch_X && (prefetch_pitch <= 0 || prefetch_row_floats <= 0 || prefetch_row_floats > prefetch_pitch)) return GRAPES_EINVAL;
    return gcn_prepare_impl(edge_src, edge_dst, e, d_e, node_map, n, d_n, flags, rowptr_t, csr_src, rowptr_s, csr_dst, dinv, long_items,
                            n_long, head_ids, row_head, workspace, sync, status, stream,
                            PrefetchReq{prefetch_X, (long long)prefetch_pitch, prefetch_row_floats});
}

static int gcn_prepare_impl(const int32_t* edge_src, const int32_t* edge_dst, int32_t e, const int32_t* d_e,
                                  const int32_t* node_map, int32_t n, const int32_t* d_n, int32_t flags,
                                  int32_t* rowptr_t, int32_t* csr_src, int32_t* rowptr_s, int32_t* csr_dst, float* dinv,
                                  int32_t* long_items, int32_t* n_long, const int32_t* head_ids, int32_t* row_head,
                                  void* workspace, uint64_t* sync, int32_t* status, grapes_stream_t stream, PrefetchReq pfreq) {
    if (e < 0 || n < 0 || !rowptr_t || !rowptr_s || !dinv || !workspace) return GRAPES_EINVAL;
    if ((head_ids == nullptr) != (row_head == nullptr)) return GRAPES_EINVAL;
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
    int32_t* rl_src = tmp_dst + (size_t)e + 1;      // relabelled edge list (node_map given)
    int32_t* rl_dst = rl_src + (size_t)e + 1;
    if (grouped && n <= SMALL_N && n > 0) {         // small graph: the whole build in one workgroup
        hipLaunchKernelGGL(prep_small_k, dim3(1), dim3(SMALL_T), 0, s, edge_src, edge_dst, e, d_e, n, d_n, node_map,
                           rowptr_t, csr_src, rowptr_s, csr_dst, dinv, long_items, n_long,
                           grapes_gcn_long_items_capacity(e), tmp_src, status, head_ids, row_head);
        GRAPES_LAUNCH_CHECK();
        return 0;
    }
    // GRAPES_PREP_PREZEROED: the caller has zeroed the first grapes_gcn_prepare_zero_words(n) words of the workspace and (in
    // grouped mode) csr_dst[0..e) in an earlier launch (grapes_frontier_compact can do both): no init launch, and the
    // TensorMap relabel happens inside the two per-edge kernels
    const bool prezeroed = (flags & GRAPES_PREP_PREZEROED) != 0;
    if (!prezeroed) {
        const size_t work = (4 * n1 + 4) > (size_t)e ? (4 * n1 + 4) : (size_t)e;
        int gi = grapes_div_up((int64_t)work, 256 * 4); if (gi < 1) gi = 1; if (gi > 2048) gi = 2048;
        hipLaunchKernelGGL(prep_init_k, dim3(gi), dim3(256), 0, s, edge_src, edge_dst, e, d_e, node_map, rl_src, rl_dst,
                           cnt_t, 4 * n1 + 4, csr_dst, (grouped && e > 0) ? 1 : 0, n_long);
        GRAPES_LAUNCH_CHECK();
    }
    const int32_t* relabel = (prezeroed && node_map) ? node_map : nullptr;
#ifdef GRAPES_DIAG
    static int fused = -1, ncu = 0;
    if (fused < 0) {
        // OFF by default: measured SLOWER than the four launches (0.611-0.627 vs 0.585 ms/step at 16 / 32 / 64 / 129 workgroups,
        // profiles/r03_prep_fused_ab.txt): inside a replayed hipGraph a dependent launch of a small kernel costs 2.4-3.0 us, a
        // phase + grid barrier 1.2-2.1 us (profiles/r03_grid_barrier.txt), and the phases run on fewer lanes with agent-scope
        // (uncached) hand-offs.  Kept as an A/B switch; tests/test_hip_parity.py holds it bit-identical to the four launches.
        const char* ev = grapes_tune_env("GRAPES_PREP_FUSED"); fused = ev ? atoi(ev) : 0;
        int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 0;
    }
    // (the top two words of the look-back scratch are the grid barrier's: tiles use the words below them)
    if (fused && grouped && prezeroed && sync != nullptr && e > 0 && n > 0 && G <= GRAPES_SYNC_SLOTS - 3 && ncu >= 8) {
        // FEW workgroups: a grid barrier costs ~0.6 us + 12 ns per workgroup (profiles/r03_grid_barrier.txt: 0.9 us at 32, 2.0 at
        // 128, 3.7 at 256 — the arrivals are atomics on one address), and these phases are bound by their dependent round
        // trips, not by the number of lanes: 32 workgroups = 32k threads hold a products-sized hop (42k edges, 40k rows)
        static int wg_min = 0;
        if (!wg_min) { const char* ev = grapes_tune_env("GRAPES_PREP_FUSED_WGS"); wg_min = ev ? atoi(ev) : 32; if (wg_min < 1) wg_min = 1; }
        int grid = grapes_div_up(e, 1024 * FUSED_EPT);
        if (grid < wg_min) grid = wg_min;
        if (grid > G && grid > grapes_div_up(e, 1024)) grid = G > grapes_div_up(e, 1024) ? G : grapes_div_up(e, 1024);
        if (grid > ncu) grid = ncu;                    // every workgroup resident: one per compute unit at most
        hipLaunchKernelGGL(prep_fused_k, dim3(grid), dim3(1024), 0, s, edge_src, edge_dst, e, d_e, n, d_n, relabel,
                           cnt_t, loops, nseg, bad, seg_first, seg_last, tmp_src, rl_src, rl_dst, tmp_dst,
                           rowptr_t, rowptr_s, dinv, csr_src, csr_dst, long_items, n_long, grapes_gcn_long_items_capacity(e),
                           head_ids, row_head, (unsigned long long*)sync, (unsigned*)(sync + (GRAPES_SYNC_WORDS - 2)), status);
        GRAPES_LAUNCH_CHECK();
        return 0;
    }
#endif  // GRAPES_DIAG
    const int32_t* es = (node_map && !prezeroed) ? rl_src : edge_src;
    const int32_t* ed = (node_map && !prezeroed) ? rl_dst : edge_dst;
    int ge = grapes_div_up(e > 0 ? e : 1, 256); if (ge > 4096) ge = 4096;
    if (e > 0) {
        // one-shot prefetch request (grapes_gcn_prepare_prefetch): the head_ids rows of X, by extra workgroups of this launch
        PrefetchRows pf{nullptr, 0, 0, nullptr, nullptr};
        int gp = 0;
        if (pfreq.X && head_ids && n > 0 && prefetch_in_scan() == 0) {
            pf = PrefetchRows{pfreq.X, pfreq.pitch, pfreq.row_floats, head_ids, bad + 1};
            const long long sectors = (long long)n * ((pfreq.row_floats * 4 + 63) / 64);
            // ONE sector per thread (id -> word: two dependent loads and out), all of them resident at once: the helpers are gone
            // after ~one HBM round trip, before the histogram workgroups finish their three
            gp = (int)((sectors + 255) / 256); if (gp > 1536) gp = 1536;
        }
        hipLaunchKernelGGL(prep_hist_k, dim3(ge + gp), dim3(256), 0, s, es, ed, e, d_e, n, d_n, grouped, cnt_t, cnt_s,
                           seg_first, seg_last, loops, nseg, bad, status, relabel, prezeroed ? n_long : nullptr, ge, pf);
        GRAPES_LAUNCH_CHECK();
    }
    const bool one_scan = sync != nullptr && G <= GRAPES_SYNC_SLOTS;
    if (!one_scan) {
        hipLaunchKernelGGL(prep_scan_count_k, dim3(G), dim3(1024), 0, s, n, d_n, grouped, (const int32_t*)cnt_t,
                           (const int32_t*)cnt_s, (const int32_t*)nseg, (const int32_t*)seg_first, (const int32_t*)seg_last,
                           (const int32_t*)loops, (const int32_t*)bad, bsum_t, bsum_s, n_long);
        GRAPES_LAUNCH_CHECK();
    }
    {
        PrefetchRows pfs{nullptr, 0, 0, nullptr, nullptr};
        int gps = 0;
        if (pfreq.X && head_ids && n > 0 && e > 0 && prefetch_in_scan() == 1) {       // helpers ride in the scan launch (the longest of the four)
            pfs = PrefetchRows{pfreq.X, pfreq.pitch, pfreq.row_floats, head_ids, bad + 1};
            const long long sectors = (long long)n * ((pfreq.row_floats * 4 + 63) / 64);
            gps = (int)((sectors + 1023) / 1024); if (gps > 512) gps = 512;             // 1024-thread workgroups, one sector per thread
        }
        hipLaunchKernelGGL(prep_scan_emit_k, dim3(G + gps), dim3(1024), 0, s, n, d_n, grouped, cnt_t, cnt_s, (const int32_t*)nseg,
                           (const int32_t*)seg_first, (const int32_t*)seg_last, (const int32_t*)loops, (const int32_t*)bad, (const int32_t*)bsum_t,
                           (const int32_t*)bsum_s, rowptr_t, rowptr_s, dinv, long_items, n_long,
                           grapes_gcn_long_items_capacity(e), (unsigned long long*)(one_scan ? sync : nullptr), status, G, pfs);
        GRAPES_LAUNCH_CHECK();
    }
    if (e > 0 && n > 0) {
        hipLaunchKernelGGL(prep_fill_k, dim3(ge), dim3(256), 0, s, es, ed, e, d_e, n, d_n, grouped, cnt_t, cnt_s,
                           (const int32_t*)rowptr_s, (const int32_t*)seg_first, (const int32_t*)loops, (const int32_t*)bad, tmp_src, tmp_dst,
                           csr_dst, status, relabel);
        GRAPES_LAUNCH_CHECK();
        const int both = grouped ? 0 : 1;
        int gr = grapes_div_up((both ? 2 : 1) * (int64_t)n, 256); if (gr > 4096) gr = 4096;
        hipLaunchKernelGGL(prep_sort_rows_k, dim3(gr), dim3(256), 0, s,
                           SortRowsArgs{n, d_n, both, (const int32_t*)rowptr_t, (const int32_t*)rowptr_s, (const int32_t*)tmp_src,
                                        (const int32_t*)tmp_dst, csr_src, csr_dst, head_ids, (const float*)dinv, row_head});
        GRAPES_LAUNCH_CHECK();
    } else if (row_head && n > 0) {
        return GRAPES_EINVAL;           // heads are written by the row-sort launch (needs e > 0 capacity)
    }
    return 0;
}

/* count (<= 8) graphs over the same n <= 2048 nodes, grouped edge lists, one launch.  Pointer arrays are HOST arrays of
 * device pointers (their values are baked into the launch).  workspaces[i]: grapes_gcn_prepare_workspace_bytes(n, e[i]). */
extern "C" int grapes_gcn_prepare_small_batch(int32_t count, const int32_t* const* edge_src, const int32_t* const* edge_dst,
                                              const int32_t* e, const int32_t* const* d_e, const int32_t* node_map,
                                              int32_t n, const int32_t* d_n, int32_t* const* rowptr_t,
                                              int32_t* const* csr_src, int32_t* const* rowptr_s, int32_t* const* csr_dst,
                                              float* const* dinv, int32_t* const* long_items, int32_t* const* n_long,
                                              const int32_t* head_ids, int32_t* const* row_head, void* const* workspaces,
                                              const int32_t* const* slice_stage, const int32_t* const* d_fe, const int32_t* fe_cap,
                                              int32_t* status, grapes_stream_t stream) {
    if (count < 1 || count > SMALL_BATCH_MAX || n <= 0 || n > SMALL_N) return GRAPES_EINVAL;
    if (slice_stage && (!d_fe || !fe_cap)) return GRAPES_EINVAL;
    if (!edge_src || !edge_dst || !e || !d_e || !rowptr_t || !csr_src || !rowptr_s || !csr_dst || !dinv || !workspaces)
        return GRAPES_EINVAL;
    SmallBatch b;
    for (int i = 0; i < count; ++i) {
        if (e[i] <= 0 || !edge_src[i] || !edge_dst[i] || !rowptr_t[i] || !csr_src[i] || !rowptr_s[i] || !csr_dst[i] || !dinv[i] ||
            !workspaces[i])
            return GRAPES_EINVAL;
        const bool li = long_items && long_items[i], nl = n_long && n_long[i];
        if (li != nl) return GRAPES_EINVAL;
        if ((head_ids == nullptr) != !(row_head && row_head[i])) return GRAPES_EINVAL;
        const size_t n1 = (size_t)n + 1;
        const int G = scan_blocks(n);
        int32_t* tmp_src = (int32_t*)workspaces[i] + 4 * n1 + 4 + 2 * n1 + 2 * (size_t)G;      // same slot as grapes_gcn_prepare
        const bool staged = slice_stage && slice_stage[i];
        if (staged && (!d_e[i] || fe_cap[i] <= 0)) return GRAPES_EINVAL;       // (d_e[i] receives the assembled list's length)
        b.g[i] = SmallGraph{edge_src[i], edge_dst[i], d_e[i], e[i], rowptr_t[i], csr_src[i], rowptr_s[i], csr_dst[i], dinv[i],
                            li ? long_items[i] : nullptr, nl ? n_long[i] : nullptr, tmp_src, row_head ? row_head[i] : nullptr,
                            grapes_gcn_long_items_capacity(e[i]), staged ? slice_stage[i] : nullptr, staged ? d_fe[i] : nullptr,
                            staged ? fe_cap[i] : 0};
    }
    for (int i = count; i < SMALL_BATCH_MAX; ++i) b.g[i] = b.g[0];
    // (a recorded fill — the next step's hop-0 graph build — may ride beside these `count` workgroups: riders, common.h; the fill
    // body does not care about the workgroup size)
    if (const GrapesRiderRecord* r = grapes_rider_match(GRAPES_RK_FILL, 0, 0, (hipStream_t)stream)) {
        FillSlotsArgs Bq; memcpy(&Bq, r->args, sizeof Bq);
        const int gb = grapes_div_up(r->grid, SMALL_T / 256);         // the rider's 256-thread workgroups as 1024-thread ones
        if (Bq.ge > gb) Bq.ge = gb;                                    // (its loops are grid-stride: fewer, larger workgroups cover the same edges)
        hipLaunchKernelGGL(prep_small_batch_fill_pair_k, dim3(count + gb), dim3(SMALL_T), 0, (hipStream_t)stream, b, n, d_n, node_map,
                           status, head_ids, Bq, count);
    } else {
        hipLaunchKernelGGL(prep_small_batch_k, dim3(count), dim3(SMALL_T), 0, (hipStream_t)stream, b, n, d_n, node_map, status, head_ids);
    }
    GRAPES_LAUNCH_CHECK();
    return 0;
}

extern "C" int grapes_gcn_prepare_counted(const int32_t* edge_src, const int32_t* edge_dst, const int32_t* slot, int32_t e,
                                          const int32_t* d_e, const int32_t* node_map, int32_t n, const int32_t* d_n,
                                          const int32_t* rowptr_t, const int32_t* rowptr_s, const int32_t* seg_first,
                                          const int32_t* row_loops, const float* dinv, int32_t* csr_src, int32_t* csr_dst,
                                          int32_t* tmp_src, const int32_t* head_ids, int32_t* row_head, int32_t* status,
                                          const float* prefetch_X, int64_t prefetch_pitch, int32_t prefetch_row_floats,
                                          int32_t* cursor, grapes_stream_t stream) {
    if (e <= 0 || n <= 0 || !edge_src || !edge_dst || (!slot && !cursor) || !node_map || !rowptr_t || !rowptr_s || !seg_first || !row_loops ||
        !dinv || !csr_src || !csr_dst || !tmp_src)
        return GRAPES_EINVAL;
    if ((head_ids == nullptr) != (row_head == nullptr)) return GRAPES_EINVAL;
    if (prefetch_X && (prefetch_pitch <= 0 || prefetch_row_floats <= 0 || prefetch_row_floats > prefetch_pitch)) return GRAPES_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int ge = grapes_div_up(e, 256); if (ge > 4096) ge = 4096;
    ge = grapes_rider_grid(ge);
    PrefetchRows pf{nullptr, 0, 0, nullptr, nullptr};
    int gp = 0;
    if (prefetch_X && head_ids) {
        pf = PrefetchRows{prefetch_X, (long long)prefetch_pitch, prefetch_row_floats, head_ids, nullptr};
        const long long sectors = (long long)n * ((prefetch_row_floats * 4 + 63) / 64);
        gp = (int)((sectors + 255) / 256); if (gp > 1536) gp = 1536;
        gp = grapes_rider_grid(gp);
    }
    // (both launches may be recorded as riders / carry the riders of another build: common.h)
    const FillSlotsArgs FA{edge_src, edge_dst, slot, e, d_e, n, d_n, rowptr_t, rowptr_s, seg_first, row_loops, tmp_src, csr_dst, status,
                           node_map, ge, pf, cursor};
    const int gf = ge + gp;
    auto fill1 = [=](hipStream_t s_) { hipLaunchKernelGGL(prep_fill_slots_k, dim3(gf), dim3(256), 0, s_, FA); };
    int gr = grapes_div_up((int64_t)n, 256); if (gr > 4096) gr = 4096;
    gr = grapes_rider_grid(gr);
    const SortRowsArgs SA{n, d_n, 0, rowptr_t, rowptr_s, (const int32_t*)tmp_src, (const int32_t*)nullptr, csr_src, csr_dst, head_ids,
                          dinv, row_head};
    auto sort1 = [=](hipStream_t s_) { hipLaunchKernelGGL(prep_sort_rows_k, dim3(gr), dim3(256), 0, s_, SA); };
    if (grapes_rider_recording()) {
        grapes_rider_record(grapes_rider_make(GRAPES_RK_FILL, 0, gf, 256, FA, fill1));
        grapes_rider_record(grapes_rider_make(GRAPES_RK_SORT, 0, gr, 256, SA, sort1));
        return 0;
    }
    if (const GrapesRiderRecord* r = grapes_rider_match(GRAPES_RK_FILL, 0, 256, s)) {
        FillSlotsArgs Bq; memcpy(&Bq, r->args, sizeof Bq);
        hipLaunchKernelGGL(prep_fill_slots_pair_k, dim3(gf + r->grid), dim3(256), 0, s, FA, Bq, gf);
    } else {
        fill1(s);
    }
    GRAPES_LAUNCH_CHECK();
    if (const GrapesRiderRecord* r = grapes_rider_match(GRAPES_RK_SORT, 0, 256, s)) {
        SortRowsArgs Bq; memcpy(&Bq, r->args, sizeof Bq);
        hipLaunchKernelGGL(prep_sort_rows_pair_k, dim3(gr + r->grid), dim3(256), 0, s, SA, Bq, gr);
    } else {
        sort1(s);
    }
    GRAPES_LAUNCH_CHECK();
    return 0;
}
