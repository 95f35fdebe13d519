// Integer / index kernels of the hop pipeline (HBM- and latency-bound; no MFMA here).
//   A1 get_neighborhoods      modules/utils.py:74-82
//   A3 slice_adjacency        modules/utils.py:85-95
//   A4 TensorMap              modules/utils.py:98-120
//   A8 frontier compaction    main.py:183-195   feature gather main.py:168,191,199-204
#include "common.h"

// ---------------------------------------------------------------------------- A4 TensorMap
__global__ void tensormap_update_k(int32_t* __restrict__ map, const int32_t* __restrict__ keys,
                                   int n_host, const int32_t* d_n) {
    const int n = eff_count(d_n, n_host);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        map[keys[i]] = i;
}

__global__ void tensormap_map_k(const int32_t* __restrict__ map, const int32_t* __restrict__ keys,
                                int32_t* __restrict__ out, int64_t n_host, const int32_t* d_n) {
    int64_t n = n_host;
    if (d_n) { int64_t v = *d_n; n = v < n_host ? v : n_host; }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = map[keys[i]];
}

extern "C" int grapes_tensormap_update(int32_t* map, const int32_t* keys, int32_t n,
                                       const int32_t* d_n, grapes_stream_t stream) {
    if (!map || (!keys && n > 0) || n < 0) return GRAPES_EINVAL;
    if (n == 0) return 0;
    int grid = grapes_div_up(n, 256); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(tensormap_update_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, map, keys, n, d_n);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

extern "C" int grapes_tensormap_map(const int32_t* map, const int32_t* keys, int32_t* out, int64_t n,
                                    const int32_t* d_n, grapes_stream_t stream) {
    if (!map || ((!keys || !out) && n > 0) || n < 0) return GRAPES_EINVAL;
    if (n == 0) return 0;
    int grid = grapes_div_up(n, 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(tensormap_map_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, map, keys, out, n, d_n);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------- A1 frontier
// One workgroup: row lengths of the queried nodes + exclusive scan (chunks of 1024 with carry).
__global__ __launch_bounds__(1024) void frontier_offsets_k(const int64_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ nodes, int m_host,
                                                           const int32_t* d_m, int32_t* __restrict__ eoff,
                                                           int32_t* d_e_out) {
    __shared__ int lds[17];
    const int m = eff_count(d_m, m_host);
    long long carry = 0;
    for (int base = 0; base < m; base += blockDim.x) {
        const int i = base + threadIdx.x;
        int len = 0;
        if (i < m) {
            const int v = nodes[i];
            len = (int)(rowptr[v + 1] - rowptr[v]);
        }
        int tot;
        const int ex = block_excl_scan(len, lds, &tot);
        if (i < m) {
            long long o = carry + ex;
            eoff[i] = o > 0x7fffffffLL ? 0x7fffffff : (int)o;
        }
        carry += tot;
    }
    if (threadIdx.x == 0) {
        const int e = carry > 0x7fffffffLL ? 0x7fffffff : (int)carry;
        eoff[m] = e;
        if (d_e_out) *d_e_out = e;
    }
}

#define EXPAND_LDS_OFFS 2048
__global__ __launch_bounds__(256) void frontier_expand_k(const int64_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ col,
                                                         const int32_t* __restrict__ nodes, int m_host,
                                                         const int32_t* d_m, const int32_t* __restrict__ eoff,
                                                         int e_cap, int32_t* __restrict__ src,
                                                         int32_t* __restrict__ dst, int32_t* __restrict__ src_pos,
                                                         int32_t* status) {
    __shared__ int s_off[EXPAND_LDS_OFFS + 1];
    const int m = eff_count(d_m, m_host);
    const int e_true = eoff[m];
    const int e = e_true < e_cap ? e_true : e_cap;
    if (blockIdx.x == 0 && threadIdx.x == 0 && e_true > e_cap && status)
        atomicOr(status, GRAPES_STATUS_EDGE_OVERFLOW);
    if ((long long)blockIdx.x * blockDim.x >= e) return;   // uniform per block
    const bool in_lds = m <= EXPAND_LDS_OFFS;
    if (in_lds) {
        for (int i = threadIdx.x; i <= m; i += blockDim.x) s_off[i] = eoff[i];
        __syncthreads();
    }
    const int32_t* offs = in_lds ? s_off : eoff;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < e; t += gridDim.x * blockDim.x) {
        int lo = 0, hi = m;   // invariant: offs[lo] <= t < offs[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (offs[mid] <= t) lo = mid; else hi = mid;
        }
        const int v = nodes[lo];
        src[t] = v;
        dst[t] = col[rowptr[v] + (t - offs[lo])];
        if (src_pos) src_pos[t] = lo;
    }
}

extern "C" int grapes_frontier_offsets(const int64_t* rowptr, const int32_t* nodes, int32_t m,
                                       const int32_t* d_m, int32_t* eoff, int32_t* d_e_out,
                                       grapes_stream_t stream) {
    if (!rowptr || !eoff || (!nodes && m > 0) || m < 0) return GRAPES_EINVAL;
    hipLaunchKernelGGL(frontier_offsets_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, rowptr, nodes, m, d_m,
                       eoff, d_e_out);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

extern "C" int grapes_frontier_expand(const int64_t* rowptr, const int32_t* col, const int32_t* nodes,
                                      int32_t m, const int32_t* d_m, const int32_t* eoff, int32_t e_cap,
                                      int32_t* src, int32_t* dst, int32_t* src_pos, int32_t* status,
                                      grapes_stream_t stream) {
    if (!rowptr || !col || !eoff || m < 0 || e_cap < 0) return GRAPES_EINVAL;
    if (m == 0 || e_cap == 0) return 0;
    if (!nodes || !src || !dst) return GRAPES_EINVAL;
    int grid = grapes_div_up(e_cap, 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(frontier_expand_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, rowptr, col, nodes, m,
                       d_m, eoff, e_cap, src, dst, src_pos, status);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// get_neighborhoods in ONE launch for query lists of up to EXPAND_LDS_OFFS nodes (the step's <= B + K previous nodes):
// every workgroup rebuilds the (short) row-length scan in LDS itself instead of waiting for a separate offsets launch;
// workgroup 0 also publishes eoff / the edge count for the later consumers.
__device__ __forceinline__ void mark_bit(unsigned long long* __restrict__ bits, unsigned long long* __restrict__ bits1,
                                         int id, int num_nodes, int32_t* status);

__device__ __forceinline__ void frontier_expand_fused_body(const int64_t* __restrict__ rowptr,
                                                               const int32_t* __restrict__ col,
                                                               const int32_t* __restrict__ nodes, int m_host,
                                                               const int32_t* d_m, int e_cap, int32_t* __restrict__ eoff,
                                                               int32_t* d_e_out, int32_t* __restrict__ src,
                                                               int32_t* __restrict__ dst, int32_t* status,
                                                               unsigned long long* __restrict__ mark_prev,
                                                               unsigned long long* __restrict__ mark_bits, int num_nodes,
                                                               grapes_slice_remark_args rm, const int32_t* __restrict__ count_mult,
                                                               int32_t* __restrict__ count_bsum, int32_t* __restrict__ slice_stage,
                                                               grapes_hop_count_args hc, const long long* __restrict__ node_ext,
                                                               long long* __restrict__ node_ext_out, const int BID, const int NBLK) {
    __shared__ int s_off[EXPAND_LDS_OFFS + 1];
    __shared__ int s_node[EXPAND_LDS_OFFS];
    __shared__ long long s_beg[EXPAND_LDS_OFFS];
    __shared__ int lds[17];
    // every queried row's (id, begin, length) first — all loads of a thread in flight together, the ids requested inside the
    // CAPACITY before the live count is known (one round trip less; entries past m are dropped) — and kept in LDS, so that an
    // edge later costs ONE global load (its column), not three dependent ones.  A thread owns RPT CONSECUTIVE rows: their
    // offsets are a thread-local prefix plus ONE workgroup scan (a scan per 256-row step cost three barriers per step).
    constexpr int RPT = EXPAND_LDS_OFFS / 256;
    int vv[RPT]; long long b0[RPT], b1[RPT];
    GRAPES_STAMP_NW(0);
    // node_ext (round 5): the queried rows' extents (rowptr[id], rowptr[id + 1]) handed over by whoever wrote the id list (the
    // draw: grapes_gumbel_topk_deferred_ext) — ids, extents and the live count arrive in ONE round trip instead of two dependent ones
    // (uniform conditions OUTSIDE the row loops: with `if (node_ext …)` inside, the eight 16-byte loads of a thread went through one
    // register quadruple, each waited for before the next was issued — eight L2 round trips where one was meant)
    if (m_host > 0) {                                                         // (m_host == 0: nodes may be NULL)
#pragma unroll
        for (int k = 0; k < RPT; ++k) { const int i = RPT * (int)threadIdx.x + k; vv[k] = nodes[i < m_host ? i : m_host - 1]; }
        if (node_ext) {
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int i = RPT * (int)threadIdx.x + k;
                const longlong2 x = *reinterpret_cast<const longlong2*>(node_ext + 2 * (long long)(i < m_host ? i : m_host - 1));
                b0[k] = x.x; b1[k] = x.y;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < RPT; ++k) vv[k] = 0;
    }
    const int m = eff_count(d_m, m_host);
#pragma unroll
    for (int k = 0; k < RPT; ++k) vv[k] = RPT * (int)threadIdx.x + k < m ? vv[k] : 0;      // a stale id past the live count: row 0 stands in (valid, unused)
    if (!node_ext) {               // (uniform; the loads alone in their loop: all RPT row extents in flight together)
#pragma unroll
        for (int k = 0; k < RPT; ++k) { const longlong2 x = *reinterpret_cast<const longlong2*>(rowptr + vv[k]); b0[k] = x.x; b1[k] = x.y; }
    } else {
#pragma unroll
        for (int k = 0; k < RPT; ++k) { const bool in = RPT * (int)threadIdx.x + k < m; b0[k] = in ? b0[k] : 0; b1[k] = in ? b1[k] : 0; }
    }
    // (a loop of its own: with the store inside the loop above every row's extent load was waited for before the next row's was
    // issued — RPT dependent round trips in the launches that get no extents handed over: hop 0's, the evaluation's)
    if (node_ext_out && BID == 0) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int i = RPT * (int)threadIdx.x + k;
            if (i < m) *reinterpret_cast<longlong2*>(node_ext_out + 2 * (long long)i) = make_longlong2(b0[k], b1[k]);
        }
    }
    GRAPES_STAMP(1);                                            // ids + live count + row extents have arrived (two dependent trips)
    long long loc[RPT + 1];
    loc[0] = 0;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int i = RPT * (int)threadIdx.x + k;
        loc[k + 1] = loc[k] + (i < m ? b1[k] - b0[k] : 0);
    }
    // the workgroup scan runs on ints: a thread's total is clamped to (2^31 - 1) / 256.  A clamped thread's rows alone exceed
    // the edge capacity (the launcher keeps e_cap below the clamp), so every edge index below e_cap still resolves exactly.
    const int clampv = 0x7fffffff / 256;
    int tot;
    const int ex = block_excl_scan(loc[RPT] > clampv ? clampv : (int)loc[RPT], lds, &tot);
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int i = RPT * (int)threadIdx.x + k;
        if (i < m) {
            const long long o = (long long)ex + loc[k];
            s_off[i] = o > 0x7fffffffLL ? 0x7fffffff : (int)o;
            s_node[i] = vv[k]; s_beg[i] = b0[k];
        }
    }
    const long long carry = tot;
    const int e_true = carry > 0x7fffffffLL ? 0x7fffffff : (int)carry;
    if (threadIdx.x == 0) s_off[m] = e_true;
    __syncthreads();
    GRAPES_STAMP_NW(2);                                         // the row-length scan is in LDS
    if (BID == 0) {
        for (int i = threadIdx.x; i <= m; i += blockDim.x) eoff[i] = s_off[i];
        if (threadIdx.x == 0) {
            if (d_e_out) *d_e_out = e_true;
            if (e_true > e_cap && status) atomicOr(status, GRAPES_STATUS_EDGE_OVERFLOW);
        }
        if (hc.indeg && hc.n_long && threadIdx.x < 2) hc.n_long[threadIdx.x] = 0;
    }
    // The queried nodes' side jobs, SPREAD over the workgroups (wavefront 0 of workgroup b takes rows 64 b .. 64 b + 63: every
    // workgroup holds the whole scan) and without a look at the bitmap word first — the first version left all m <= 2048 rows to
    // workgroup 0, three per thread, each mark a load THEN an atomic: six dependent round trips (8 us) before that workgroup
    // began its share of the edges, the launch's tail (profiles/r05_index_phase_stamps.txt).
    //  - the hop's marks (grapes_bitmap_mark_hop): queried nodes -> mark_prev, queried nodes with at least one edge and (below)
    //    every neighbour -> mark_bits
    //  - the hop graph's by-source side (include/grapes_hip.h: grapes_hop_count_args): a queried node's edge segment and its
    //    out-degree on the sum of its bitmap word (self-loops come off below, edge by edge)
    if ((mark_bits || hc.indeg) && threadIdx.x < 64) {
        const int W = (num_nodes + 63) >> 6;
        for (int i = BID * 64 + (int)threadIdx.x; i < m; i += NBLK * 64) {
            const int g = s_node[i], len = s_off[i + 1] - s_off[i];
            if ((unsigned)g < (unsigned)num_nodes) {
                const unsigned long long bit = 1ull << (g & 63);
                if (mark_bits) {
                    if (mark_prev) atomicOr(&mark_prev[g >> 6], bit);
                    if (len > 0) atomicOr(&mark_bits[g >> 6], bit);
                }
                if (hc.indeg) {
                    *reinterpret_cast<int2*>(hc.seginfo + 2 * (long long)g) = make_int2(s_off[i], len);
                    if (len > 0) atomicAdd(&hc.wsum[W + (g >> 6)], len);
                }
            } else if (mark_bits && status) {
                atomicOr(status, GRAPES_STATUS_BAD_INDEX);
            }
        }
    }
    if (rm.mult || rm.clear_ids) {   // grapes_slice_remark in the same launch (lists disjoint; clear_bits is not mark_prev)
        const int stride = NBLK * blockDim.x, i0 = BID * blockDim.x + threadIdx.x;
        if (rm.unmark_ids) { const int c = eff_count(rm.d_n_unmark, rm.n_unmark); for (int i = i0; i < c; i += stride) rm.mult[rm.unmark_ids[i]] = 0; }
        if (rm.mark_ids) { const int c = eff_count(rm.d_n_mark, rm.n_mark); for (int i = i0; i < c; i += stride) atomicAdd(&rm.mult[rm.mark_ids[i]], 1); }
        if (rm.clear_ids) {
            const int c = eff_count(rm.d_n_clear, rm.n_clear);
            for (int i = i0; i < c; i += stride) reinterpret_cast<unsigned long long*>(rm.clear_bits)[rm.clear_ids[i] >> 6] = 0ull;
        }
    }
    const int e = e_true < e_cap ? e_true : e_cap;
    GRAPES_STAMP_NW(3);                                         // side jobs issued (workgroup 0: eoff, marks, segments)
    for (int t = BID * blockDim.x + threadIdx.x; t < e; t += NBLK * blockDim.x) {
        int lo = 0, hi = m;   // invariant: s_off[lo] <= t < s_off[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_off[mid] <= t) lo = mid; else hi = mid;
        }
        const int d = col[s_beg[lo] + (t - s_off[lo])];
        const int sn = s_node[lo];
        src[t] = sn;
        dst[t] = d;
        // Everything that needs only `d` is REQUESTED together — the bitmap word of its mark (looked at first: hub neighbours
        // repeat, an atomic per repeat is wasted), the returning in-degree atomic whose result is the entry's slot in its row, its
        // slice multiplicity — and consumed afterwards.  One after the other (mark_bit's load, then the atomic, then the
        // multiplicity: each closed by its own wait) they were three dependent cold round trips behind the column's.
        const bool dv = (unsigned)d < (unsigned)num_nodes;
        const int dc = dv ? d : 0;                                // (a bad index reads / adds 0 to entry 0 and raises the status below)
        const bool counted = hc.indeg && dv && d != sn;
        unsigned long long wold = 0ull;
        int sl = 0, cm = 0;
        if (mark_bits) wold = mark_bits[dc >> 6];
        if (hc.indeg && hc.slot) sl = atomicAdd(&hc.indeg[dc], counted ? 1 : 0);
        if (count_bsum || slice_stage) cm = count_mult[d];        // (as before: num_nodes may be 0 when nothing is marked or counted)
        if (hc.indeg && !hc.slot && counted) atomicAdd(&hc.indeg[d], 1);           // (nobody waits for it: the fill takes its places from row cursors)
        // The consumers, without a branch between them: a store or atomic that has nothing to do adds 0 / ORs 0 / stores the value that
        // is there (hipcc closes every branch that redefines an address register of an outstanding store with vmcnt(0) — four
        // acknowledgements waited for one after the other in this tail).  Only the rare cases keep a branch: a bad index, a self-loop.
        // Addresses and operands are all formed FIRST and pinned in registers of their own (the empty asm): a register that is still the
        // address or operand of an outstanding store may not be redefined before the store is acknowledged, and the allocator reused
        // two pairs for all of them.
        const unsigned long long dbit = 1ull << (d & 63);
        unsigned long long* p_or = mark_bits + (dc >> 6);
        unsigned long long v_or = (dv && !(wold & dbit)) ? dbit : 0ull;
        int32_t* p_sl = hc.slot + t;
        int v_sl = (dv && d != sn) ? sl : -1;
        int32_t* p_ws = hc.wsum + (dc >> 6);
        int v_ws = counted ? 1 : 0;
        int32_t* p_bs = count_bsum + (t >> 10);
        int v_bs = cm > 0 ? cm : 0;
        asm volatile("" : "+v"(p_or), "+v"(v_or), "+v"(p_sl), "+v"(v_sl), "+v"(p_ws), "+v"(v_ws), "+v"(p_bs), "+v"(v_bs));
        // (the pinned pointers come back without their address space: named again, or the stores would be FLAT instructions)
        typedef __attribute__((address_space(1))) unsigned long long* g_u64p;
        typedef __attribute__((address_space(1))) int32_t* g_i32p;
        if (mark_bits) (void)__hip_atomic_fetch_or((g_u64p)p_or, v_or, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (hc.indeg) {            // in-degree of the target; the returned count is this entry's slot in its row
            if (hc.slot) *(g_i32p)p_sl = v_sl;
            (void)__hip_atomic_fetch_add((g_i32p)p_ws, v_ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // first half of grapes_slice_filter: survivors per 1024-edge block (integer atomics: order-free); count_mult must not
        // be re-marked by THIS launch (rm.mult of a remark that touches it belongs in an earlier launch)
        if (count_bsum) (void)__hip_atomic_fetch_add((g_i32p)p_bs, v_bs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mark_bits && !dv && status) atomicOr(status, GRAPES_STATUS_BAD_INDEX);
        if (hc.indeg && dv && d == sn) {          // add_remaining_self_loops: an existing loop is replaced by the unit loop
            atomicAdd(&hc.loops[sn], 1);
            atomicSub(&hc.wsum[((num_nodes + 63) >> 6) + (sn >> 6)], 1);
        }
        // ... or the WHOLE filter's edge-side work (slice_stage): a wavefront owns the 64 consecutive edges t >> 6 == wb and
        // leaves their survivors (edge, multiplicity), in edge order, at stage slots 64 wb .. and their number / summed
        // multiplicity in the two count tables — every wavefront-block below ceil(e / 64) is written, so nothing needs
        // clearing; grapes_gcn_prepare_small_batch (the classifier's graph build, the only consumer of the slices) assembles
        // the edge list from them: no slice_filter launch between two hops (layout: include/grapes_hip.h)
        if (slice_stage) {
            const int nwb = (e_cap + 63) >> 6;
            const int c = cm;
            const unsigned long long mm = __ballot(c > 0);
            const int wb = t >> 6;
            if (mm != 0ull) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
                if (c > 0) {
                    int32_t* st = slice_stage + 2 * nwb;
                    const int q = wb * 64 + rank;
                    st[q] = s_node[lo]; st[e_cap + q] = d; st[2 * e_cap + q] = c;
                }
            }
            int csum = __popcll(mm);
            if (__ballot(c > 1) != 0ull) {             // duplicate columns (multiplicity > 1): rare, summed bit by bit with ballots
                csum = 0;                              // (ballots see the active lanes only: the last block may be ragged)
                for (int b = 0; b < 31; ++b) {
                    csum += __popcll(__ballot(c > 0 && ((c >> b) & 1))) << b;
                    if (__ballot(c > 0 && (c >> (b + 1)) != 0) == 0ull) break;
                }
            }
            if ((t & 63) == 0) { slice_stage[wb] = __popcll(mm); slice_stage[nwb + wb] = csum; }
        }
    }
    GRAPES_STAMP_NW(4);                                         // edges issued
    GRAPES_STAMP(5);                                            // ... and every store / returning atomic back
}

// First kernel of a captured training step that feeds ITSELF (one workgroup): reads the step cursor, copies the next
// batch of target ids out of a device-resident id array (unshuffled sequential chunks, main.py:126: chunk
// (cursor * stride + offset), wrapped), advances the cursor and the indicator epoch, marks the targets' indicator
// (main.py:167-168) and adds the PREVIOUS step's per-graph edge counters (overwritten later in this step) to running
// 64-bit totals — so that a replayed step needs no host-side copy, cast or accumulation launch around it.
__device__ __forceinline__ void step_begin_body(uint32_t* __restrict__ ind_code, uint32_t* d_epoch, int bit,
                                                     const int32_t* __restrict__ ids, int n_ids, int32_t* d_cursor, int stride,
                                                     int offset, int B, int32_t* __restrict__ targets,
                                                     const int32_t* __restrict__ ctr, int ctr_stride, int n_ctr,
                                                     long long* __restrict__ totals) {
    const int cur = *d_cursor;
    uint32_t epoch = d_epoch ? ((*d_epoch & 0xffffffu) + 1u) & 0xffffffu : 0u;
    long long chunk = (long long)cur * stride + offset;
    const int span = n_ids - B > 1 ? n_ids - B : 1;
    const int start = (int)((chunk * B) % span);
    if (totals && cur > 0)
        for (int j = threadIdx.x; j < n_ctr; j += blockDim.x) totals[j] += (long long)ctr[(size_t)j * ctr_stride];
    __syncthreads();                                   // every wavefront has read the cursor and the epoch
    if (threadIdx.x == 0) { *d_cursor = cur + 1; if (d_epoch) *d_epoch = epoch; }
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const int id = ids[start + i];
        targets[i] = id;
        if (ind_code) {
            uint32_t c = ind_code[id];
            if ((c >> 8) != epoch) c = epoch << 8;
            ind_code[id] = c | (1u << bit);
        }
    }
}

__global__ __launch_bounds__(1024) void step_begin_k(uint32_t* __restrict__ ind_code, uint32_t* d_epoch, int bit,
                                                     const int32_t* __restrict__ ids, int n_ids, int32_t* d_cursor, int stride,
                                                     int offset, int B, int32_t* __restrict__ targets,
                                                     const int32_t* __restrict__ ctr, int ctr_stride, int n_ctr,
                                                     long long* __restrict__ totals) {
    step_begin_body(ind_code, d_epoch, bit, ids, n_ids, d_cursor, stride, offset, B, targets, ctr, ctr_stride, n_ctr, totals);
}
struct StepBeginArgs {
    uint32_t* ind_code; uint32_t* d_epoch; int bit; const int32_t* ids; int n_ids; int32_t* d_cursor; int stride; int offset; int B;
    int32_t* targets; const int32_t* ctr; int ctr_stride; int n_ctr; long long* totals;
};

struct ExpandFusedArgs {
    const int64_t* rowptr; const int32_t* col; const int32_t* nodes; int m_host; const int32_t* d_m; int e_cap; int32_t* eoff;
    int32_t* d_e_out; int32_t* src; int32_t* dst; int32_t* status; unsigned long long* mark_prev; unsigned long long* mark_bits;
    int num_nodes; grapes_slice_remark_args rm; const int32_t* count_mult; int32_t* count_bsum; int32_t* slice_stage;
    grapes_hop_count_args hc;
    grapes_draw_finish_args fin;      // fin.sel != NULL: the LAST workgroup of this problem's range ends the draw that produced `nodes`
    const long long* node_ext; long long* node_ext_out;
};
#define EXPAND_FUSED_CALL_(A, bid, nblk)                                                                                           \
    frontier_expand_fused_body((A).rowptr, (A).col, (A).nodes, (A).m_host, (A).d_m, (A).e_cap, (A).eoff, (A).d_e_out, (A).src, (A).dst, \
                               (A).status, (A).mark_prev, (A).mark_bits, (A).num_nodes, (A).rm, (A).count_mult, (A).count_bsum,    \
                               (A).slice_stage, (A).hc, (A).node_ext, (A).node_ext_out, bid, nblk)
#define EXPAND_FUSED_CALL(A, bid, nblk)                                                                                            \
    do {                                                                                                                           \
        const int nf_ = (A).fin.sel ? 1 : 0;                                                                                       \
        if (nf_ && (bid) == (nblk) - 1) draw_finish_body((A).fin);                                                                 \
        else EXPAND_FUSED_CALL_(A, bid, (nblk) - nf_);                                                                             \
    } while (0)
__global__ __launch_bounds__(256) void frontier_expand_fused_k(ExpandFusedArgs a) { EXPAND_FUSED_CALL(a, (int)blockIdx.x, (int)gridDim.x); }
// two expansions side by side in one launch (riders: common.h): workgroups [0, nA) work on `a`, the rest on `b`
__global__ __launch_bounds__(256) void frontier_expand_fused_pair_k(ExpandFusedArgs a, ExpandFusedArgs b, int nA) {
    if ((int)blockIdx.x < nA) { EXPAND_FUSED_CALL(a, (int)blockIdx.x, nA); }
    else { EXPAND_FUSED_CALL(b, (int)blockIdx.x - nA, (int)gridDim.x - nA); }
}

// an expansion with the NEXT step's step_begin riding as one more workgroup (riders: common.h, GRAPES_RK_BEGIN)
__global__ __launch_bounds__(256) void frontier_expand_fused_begin_k(ExpandFusedArgs a, StepBeginArgs c) {
    if (blockIdx.x + 1 < gridDim.x) { EXPAND_FUSED_CALL(a, (int)blockIdx.x, (int)gridDim.x - 1); }
    else step_begin_body(c.ind_code, c.d_epoch, c.bit, c.ids, c.n_ids, c.d_cursor, c.stride, c.offset, c.B, c.targets, c.ctr,
                         c.ctr_stride, c.n_ctr, c.totals);
}

extern "C" int grapes_frontier_expand_fused_ext(const int64_t* rowptr, const int32_t* col, const int32_t* nodes, int32_t m,
                                            const int32_t* d_m, int32_t e_cap, int32_t* eoff, int32_t* d_e_out,
                                            int32_t* src, int32_t* dst, int32_t* status, uint64_t* mark_prev_bits,
                                            uint64_t* mark_bits, int32_t num_nodes, const grapes_slice_remark_args* remark,
                                            const int32_t* count_mult, int32_t* count_bsum, int32_t* slice_stage,
                                            const grapes_hop_count_args* count, const grapes_draw_finish_args* finish,
                                            const int64_t* node_ext, int64_t* node_ext_out, grapes_stream_t stream) {
    if (!rowptr || !col || !eoff || m < 0 || m > EXPAND_LDS_OFFS || e_cap < 0) return GRAPES_EINVAL;
    if ((node_ext && (((uintptr_t)node_ext) & 15) != 0) || (node_ext_out && (((uintptr_t)node_ext_out) & 15) != 0)) return GRAPES_EALIGN;
    grapes_draw_finish_args fin{};
    if (finish) {
        fin = *finish;
        if (!fin.sel || !fin.parts_keys || !fin.parts_emit || fin.emit_block <= 0 || fin.keys_blocks < 0 || fin.n_host < 0 ||
            (fin.hist_words > 0 && !fin.hist) || grapes_rider_recording())
            return GRAPES_EINVAL;
    }
    grapes_hop_count_args hc{};
    if (count) {
        hc = *count;
        if (!hc.indeg || !hc.loops || !hc.seginfo || !hc.wsum || num_nodes <= 0 || !mark_bits) return GRAPES_EINVAL;
    }
    if (e_cap >= 0x7fffffff / 256) return GRAPES_EINVAL;     // (the one-launch form's offset scan: see the kernel; larger: grapes_frontier_offsets + _expand)
    if ((mark_prev_bits || mark_bits) && (!mark_bits || num_nodes <= 0)) return GRAPES_EINVAL;
    if ((count_bsum || slice_stage) && !count_mult) return GRAPES_EINVAL;
    if (count_mult && !count_bsum && !slice_stage) return GRAPES_EINVAL;
    grapes_slice_remark_args rm{};
    if (remark) {
        rm = *remark;
        if (rm.n_unmark < 0 || rm.n_mark < 0 || rm.n_clear < 0) return GRAPES_EINVAL;
        if (!rm.mult && (rm.n_unmark > 0 || rm.n_mark > 0)) return GRAPES_EINVAL;
        if (rm.mult && rm.mult == count_mult && (rm.n_unmark > 0 || rm.n_mark > 0)) return GRAPES_EINVAL;
        if ((rm.n_unmark > 0 && !rm.unmark_ids) || (rm.n_mark > 0 && !rm.mark_ids) || (rm.n_clear > 0 && (!rm.clear_ids || !rm.clear_bits)))
            return GRAPES_EINVAL;
        if (rm.clear_bits && rm.clear_bits == mark_prev_bits) return GRAPES_EINVAL;
        if (rm.n_unmark == 0) rm.unmark_ids = nullptr;
        if (rm.n_mark == 0) rm.mark_ids = nullptr;
        if (rm.n_clear == 0) rm.clear_ids = nullptr;
    }
    if (m > 0 && (!nodes || (e_cap > 0 && (!src || !dst)))) return GRAPES_EINVAL;
    int grid = grapes_div_up(e_cap > 0 ? e_cap : 1, 256);
    {   // at most 512 workgroups: every workgroup rebuilds the row-length scan, and with a capacity of 2^19 edges (Reddit) most of
        // 2048 found no edge — 256 -> 1.073, 512 -> 1.070 / 1.069, 1024 -> 1.076, 2048 -> 1.085 / 1.084 ms/step (GRAPES_EXPAND_GRID, A/B)
        static int gcap = -1;
        if (gcap < 0) { const char* e = grapes_tune_env("GRAPES_EXPAND_GRID"); gcap = e ? atoi(e) : 512; if (gcap < 1) gcap = 512; }
        if (grid > gcap) grid = gcap;
    }
    grid = grapes_rider_grid(grid);
    if (fin.sel) grid += 1;                  // (the workgroup that ends the draw: the last of this problem's range)
    const ExpandFusedArgs A{rowptr, col, nodes, m, d_m, e_cap, eoff, d_e_out, src, dst, status, (unsigned long long*)mark_prev_bits,
                            (unsigned long long*)mark_bits, num_nodes, rm, count_mult, count_bsum, slice_stage, hc, fin,
                            (const long long*)node_ext, (long long*)node_ext_out};
    auto single = [=](hipStream_t s_) { hipLaunchKernelGGL(frontier_expand_fused_k, dim3(grid), dim3(256), 0, s_, A); };
    if (grapes_rider_recording()) { grapes_rider_record(grapes_rider_make(GRAPES_RK_EXPAND, 0, grid, 256, A, single)); return 0; }
    if (const GrapesRiderRecord* rb = grapes_rider_match(GRAPES_RK_BEGIN, 0, 0, (hipStream_t)stream)) {
        StepBeginArgs Cq; memcpy(&Cq, rb->args, sizeof Cq);
        hipLaunchKernelGGL(frontier_expand_fused_begin_k, dim3(grid + 1), dim3(256), 0, (hipStream_t)stream, A, Cq);
    } else if (const GrapesRiderRecord* r = grapes_rider_match(GRAPES_RK_EXPAND, 0, 256, (hipStream_t)stream)) {
        ExpandFusedArgs Bq; memcpy(&Bq, r->args, sizeof Bq);
        hipLaunchKernelGGL(frontier_expand_fused_pair_k, dim3(grid + r->grid), dim3(256), 0, (hipStream_t)stream, A, Bq, grid);
    } else {
        single((hipStream_t)stream);
    }
    GRAPES_LAUNCH_CHECK();
    return 0;
}
extern "C" int grapes_frontier_expand_fused(const int64_t* rowptr, const int32_t* col, const int32_t* nodes, int32_t m,
                                            const int32_t* d_m, int32_t e_cap, int32_t* eoff, int32_t* d_e_out,
                                            int32_t* src, int32_t* dst, int32_t* status, uint64_t* mark_prev_bits,
                                            uint64_t* mark_bits, int32_t num_nodes, const grapes_slice_remark_args* remark,
                                            const int32_t* count_mult, int32_t* count_bsum, int32_t* slice_stage,
                                            grapes_stream_t stream) {
    return grapes_frontier_expand_fused_ext(rowptr, col, nodes, m, d_m, e_cap, eoff, d_e_out, src, dst, status, mark_prev_bits, mark_bits,
                                            num_nodes, remark, count_mult, count_bsum, slice_stage, nullptr, nullptr, nullptr, nullptr, stream);
}
extern "C" size_t grapes_slice_stage_words(int32_t e_cap) { return 2 * (size_t)((e_cap + 63) / 64) + 3 * (size_t)(e_cap > 0 ? e_cap : 0); }

// ---------------------------------------------------------------------------- bitmaps
__global__ void bitmap_mark_k(unsigned long long* __restrict__ bits, unsigned long long* __restrict__ bits1,
                              const int32_t* __restrict__ ids, int64_t n_host, const int32_t* d_n,
                              int num_nodes, int32_t* status) {
    int64_t n = n_host;
    if (d_n) { int64_t v = *d_n; n = v < n_host ? v : n_host; }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int id = ids[i];
        if (id < 0 || id >= num_nodes) {
            if (status) atomicOr(status, GRAPES_STATUS_BAD_INDEX);
            continue;
        }
        const int w = id >> 6;
        const unsigned long long b = 1ull << (id & 63);
        // skip the atomic when the bit is already visible (hub neighbours repeat a lot)
        if (bits[w] & b) continue;
        const unsigned long long old = atomicOr(&bits[w], b);
        if (bits1 && old == 0ull) atomicOr(&bits1[w >> 6], 1ull << (w & 63));
    }
}

__global__ void bitmap_clear_k(unsigned long long* __restrict__ bits, const int32_t* __restrict__ ids,
                               int64_t n_host, const int32_t* d_n) {
    int64_t n = n_host;
    if (d_n) { int64_t v = *d_n; n = v < n_host ? v : n_host; }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        bits[ids[i] >> 6] = 0ull;
}

extern "C" int grapes_bitmap_mark(uint64_t* bits, uint64_t* bits1, const int32_t* ids, int64_t n,
                                  const int32_t* d_n, int32_t num_nodes, int32_t* status,
                                  grapes_stream_t stream) {
    if (!bits || (!ids && n > 0) || n < 0) return GRAPES_EINVAL;
    if (n == 0) return 0;
    int grid = grapes_div_up(n, 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(bitmap_mark_k, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (unsigned long long*)bits, (unsigned long long*)bits1, ids, n, d_n, num_nodes, status);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// Marks the queried nodes that have at least one out-edge (eoff from frontier_offsets) — the
// "source endpoints" of main.py:186 — once per node instead of once per edge (a hub row would
// otherwise issue thousands of atomics on one word).
__global__ void bitmap_mark_rows_k(unsigned long long* __restrict__ bits, unsigned long long* __restrict__ bits1,
                                   const int32_t* __restrict__ nodes, int m_host, const int32_t* d_m,
                                   const int32_t* __restrict__ eoff, int num_nodes, int32_t* status) {
    const int m = eff_count(d_m, m_host);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        if (eoff[i + 1] <= eoff[i]) continue;
        const int id = nodes[i];
        if (id < 0 || id >= num_nodes) {
            if (status) atomicOr(status, GRAPES_STATUS_BAD_INDEX);
            continue;
        }
        const int w = id >> 6;
        const unsigned long long old = atomicOr(&bits[w], 1ull << (id & 63));
        if (bits1 && old == 0ull) atomicOr(&bits1[w >> 6], 1ull << (w & 63));
    }
}

extern "C" int grapes_bitmap_mark_rows(uint64_t* bits, uint64_t* bits1, const int32_t* nodes, int32_t m,
                                       const int32_t* d_m, const int32_t* eoff, int32_t num_nodes, int32_t* status,
                                       grapes_stream_t stream) {
    if (!bits || m < 0 || (m > 0 && (!nodes || !eoff))) return GRAPES_EINVAL;
    if (m == 0) return 0;
    int grid = grapes_div_up(m, 256); if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(bitmap_mark_rows_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)bits,
                       (unsigned long long*)bits1, nodes, m, d_m, eoff, num_nodes, status);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// One launch for the three marks of a hop (main.py:183-187): `previous` -> prev_bits; queried nodes with >= 1 edge and
// every neighbour -> bits (+ summary).  Same effect as bitmap_mark(prev_bits) + bitmap_mark_rows + bitmap_mark(dst).
__device__ __forceinline__ void mark_bit(unsigned long long* __restrict__ bits, unsigned long long* __restrict__ bits1,
                                         int id, int num_nodes, int32_t* status) {
    if (id < 0 || id >= num_nodes) {
        if (status) atomicOr(status, GRAPES_STATUS_BAD_INDEX);
        return;
    }
    const int w = id >> 6;
    const unsigned long long b = 1ull << (id & 63);
    if (bits[w] & b) return;               // already visible: skip the atomic (hub neighbours repeat a lot)
    const unsigned long long old = atomicOr(&bits[w], b);
    if (bits1 && old == 0ull) atomicOr(&bits1[w >> 6], 1ull << (w & 63));
}

__global__ void bitmap_mark_hop_k(unsigned long long* __restrict__ prev_bits, unsigned long long* __restrict__ bits,
                                  unsigned long long* __restrict__ bits1, const int32_t* __restrict__ previous,
                                  int m_host, const int32_t* d_m, const int32_t* __restrict__ eoff,
                                  const int32_t* __restrict__ dst, int e_host, const int32_t* d_e, int num_nodes,
                                  int32_t* status) {
    const int m = eff_count(d_m, m_host);
    const int e = eff_count(d_e, e_host);
    const int stride = gridDim.x * blockDim.x;
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = i0; i < m; i += stride) {
        const int id = previous[i];
        mark_bit(prev_bits, nullptr, id, num_nodes, status);
        if (eoff[i + 1] > eoff[i]) mark_bit(bits, bits1, id, num_nodes, status);
    }
    for (int t = i0; t < e; t += stride) mark_bit(bits, bits1, dst[t], num_nodes, status);
}

extern "C" int grapes_bitmap_mark_hop(uint64_t* prev_bits, uint64_t* bits, uint64_t* bits1, const int32_t* previous,
                                      int32_t m, const int32_t* d_m, const int32_t* eoff, const int32_t* dst, int32_t e,
                                      const int32_t* d_e, int32_t num_nodes, int32_t* status, grapes_stream_t stream) {
    if (!prev_bits || !bits || m < 0 || e < 0 || (m > 0 && (!previous || !eoff)) || (e > 0 && !dst))
        return GRAPES_EINVAL;
    if (m == 0 && e == 0) return 0;
    int grid = grapes_div_up(m > e ? m : e, 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(bitmap_mark_hop_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)prev_bits,
                       (unsigned long long*)bits, (unsigned long long*)bits1, previous, m, d_m, eoff, dst, e, d_e, num_nodes,
                       status);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// Up to four id lists into one bitmap in one launch (main.py:221,252: all_nodes = targets + every hop's samples).
struct MarkLists { const int32_t* ids[4]; int n[4]; const int32_t* d_n[4]; };
__global__ void bitmap_mark_lists_k(unsigned long long* __restrict__ bits, unsigned long long* __restrict__ bits1,
                                    MarkLists L, int num_nodes, int32_t* status, int32_t* __restrict__ unmark_mult) {
    const int stride = gridDim.x * blockDim.x;
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!L.ids[k]) continue;
        const int n = eff_count(L.d_n[k], L.n[k]);
        for (int i = i0; i < n; i += stride) {
            const int id = L.ids[k][i];
            mark_bit(bits, bits1, id, num_nodes, status);
            if (unmark_mult && id >= 0 && id < num_nodes) unmark_mult[id] = 0;   // the slice marks of the step end here
        }
    }
}

extern "C" int grapes_bitmap_mark_lists(uint64_t* bits, uint64_t* bits1, const int32_t* ids0, int32_t n0,
                                        const int32_t* d_n0, const int32_t* ids1, int32_t n1, const int32_t* d_n1,
                                        const int32_t* ids2, int32_t n2, const int32_t* d_n2, const int32_t* ids3,
                                        int32_t n3, const int32_t* d_n3, int32_t num_nodes, int32_t* status,
                                        int32_t* unmark_mult, grapes_stream_t stream) {
    if (!bits || n0 < 0 || n1 < 0 || n2 < 0 || n3 < 0) return GRAPES_EINVAL;
    MarkLists L{{n0 > 0 ? ids0 : nullptr, n1 > 0 ? ids1 : nullptr, n2 > 0 ? ids2 : nullptr, n3 > 0 ? ids3 : nullptr},
                {n0, n1, n2, n3}, {d_n0, d_n1, d_n2, d_n3}};
    int nmax = n0; if (n1 > nmax) nmax = n1; if (n2 > nmax) nmax = n2; if (n3 > nmax) nmax = n3;
    if (nmax == 0) return 0;
    int grid = grapes_div_up(nmax, 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(bitmap_mark_lists_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)bits,
                       (unsigned long long*)bits1, L, num_nodes, status, unmark_mult);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// all_nodes of a step (main.py:221,252) when it is SMALL: the ascending, duplicate-free union of up to four id lists
// (<= UNION_MAX ids in all) by ONE workgroup — bitonic sort in LDS, first occurrences flagged, block scan, emit — instead
// of marking an N-bit map and compacting it (two launches that stream N/8 bytes for a thousand ids).  Also writes the
// TensorMap (node_map[id] = rank) and, like grapes_bitmap_mark_lists, can zero the slice multiplicities at those ids.
#define UNION_MAX 4096
// bitonic network over P = 1024 KPT keys held KPT per thread (1024 threads): see union_sorted_k.  Sorted keys end up in key[].
template <int KPT>
__device__ __forceinline__ void union_sort_regs(int* key, int tid, int P) {
    int v[KPT];
#pragma unroll
    for (int q = 0; q < KPT; ++q) v[q] = key[tid + 1024 * q];
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 1024) {                         // both keys of a pair are this thread's
                const int dq = j >> 10;
#pragma unroll
                for (int q = 0; q < KPT; ++q) {
                    if ((q & dq) == 0 && (q | dq) < KPT) {
                        const int q2 = q | dq;
                        const bool up = ((tid + 1024 * q) & k) == 0;
                        const int a = v[q], b = v[q2];
                        const int mn = a < b ? a : b, mx = a < b ? b : a;
                        v[q] = up ? mn : mx; v[q2] = up ? mx : mn;
                    }
                }
                continue;
            }
            int partner[KPT];
            if (j >= 64) {
                __syncthreads();
#pragma unroll
                for (int q = 0; q < KPT; ++q) key[tid + 1024 * q] = v[q];
                __syncthreads();
#pragma unroll
                for (int q = 0; q < KPT; ++q) partner[q] = key[(tid + 1024 * q) ^ j];
            } else {
#pragma unroll
                for (int q = 0; q < KPT; ++q) partner[q] = __shfl_xor(v[q], j, 64);
            }
#pragma unroll
            for (int q = 0; q < KPT; ++q) {
                const int i = tid + 1024 * q;
                const bool up = (i & k) == 0, lower = (i & j) == 0;
                const int mn = v[q] < partner[q] ? v[q] : partner[q], mx = v[q] < partner[q] ? partner[q] : v[q];
                v[q] = (lower == up) ? mn : mx;
            }
        }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < KPT; ++q) key[tid + 1024 * q] = v[q];
    __syncthreads();
}
__device__ __forceinline__ void union_sorted_body(MarkLists L, int num_nodes, int n_cap, int32_t* __restrict__ out_ids,
                                                       int32_t* __restrict__ node_map, int32_t* __restrict__ counts,
                                                       int32_t* __restrict__ unmark_mult, int32_t* status) {
    __shared__ int key[UNION_MAX];
    __shared__ int lds[17];
    __shared__ int s_total;
    const int tid = threadIdx.x;
    // gather: list k occupies [off_k, off_k + n_k); padding = INT_MAX sorts to the end.  The four device counts and every
    // list's first blockDim ids (inside the host capacities) are requested TOGETHER — one memory round trip, where a loop
    // over the lists paid two dependent ones (count, then ids) per list.
    int nh[4], cv[4], id0[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        nh[k] = L.ids[k] ? L.n[k] : 0;
        const int32_t* pc = (L.ids[k] && L.d_n[k]) ? L.d_n[k] : counts;      // (counts: any valid word; unused then)
        cv[k] = *pc;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int32_t* pi = L.ids[k] ? L.ids[k] : counts;
        id0[k] = pi[(L.ids[k] && tid < nh[k]) ? tid : 0];
    }
    int off = 0;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!L.ids[k]) continue;
        int n = nh[k];
        if (L.d_n[k]) { n = cv[k] < nh[k] ? (cv[k] < 0 ? 0 : cv[k]) : nh[k]; }
        for (int i = tid; i < n; i += blockDim.x) {
            const int id = i == tid ? id0[k] : L.ids[k][i];
            const bool ok = id >= 0 && id < num_nodes;
            bad = bad || !ok;
            key[off + i] = ok ? id : 0x7fffffff;
        }
        off += n;                                    // uniform (device counts are block-uniform)
    }
    int P = 1; while (P < off) P <<= 1;              // power of two >= live ids (<= UNION_MAX by the host check)
    for (int i = off + tid; i < P; i += blockDim.x) key[i] = 0x7fffffff;
    if (bad && status) atomicOr(status, GRAPES_STATUS_BAD_INDEX);
    __syncthreads();
    if (P <= (int)blockDim.x) {
        // one key per thread in a register: compare-exchange partners within 64 lanes come by shuffle, only the 10 stages
        // with a partner distance >= 64 go through LDS (the all-LDS network below costs 55 barriers for 1024 keys)
        int v = tid < P ? key[tid] : 0x7fffffff;
        for (int k = 2; k <= P; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                int partner;
                if (j >= 64) {
                    __syncthreads();
                    key[tid] = v;
                    __syncthreads();
                    partner = key[tid ^ j];
                } else {
                    partner = __shfl_xor(v, j, 64);
                }
                const bool up = (tid & k) == 0, lower = (tid & j) == 0;
                const int mn = v < partner ? v : partner, mx = v < partner ? partner : v;
                v = (lower == up) ? mn : mx;
            }
        __syncthreads();
        key[tid] = v;
        __syncthreads();
    } else if (blockDim.x == 1024 && (P == 2048 || P == 4096)) {
        // 2 / 4 keys per thread in registers (key tid + 1024 q): partners 1024 or 2048 away are the thread's own registers, partners
        // closer than 64 come by shuffle; only the stages with 64 <= distance <= 512 go through LDS — 14 of the 66 stages of a
        // 2048-key network (Reddit: 256 targets + 2 x 512 samples; the all-LDS network below took 28 us of that step)
        if (P == 2048) union_sort_regs<2>(key, tid, P); else union_sort_regs<4>(key, tid, P);
    } else
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const int a = key[i], b = key[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { key[i] = b; key[ixj] = a; }
                }
            }
            __syncthreads();
        }
    // first occurrences, in order: thread t owns the contiguous slice [t * per, (t + 1) * per)
    const int per = (P + (int)blockDim.x - 1) / (int)blockDim.x;
    const int lo = tid * per, hi = lo + per < P ? lo + per : P;
    int mine = 0;
    for (int i = lo; i < hi; ++i) { const int v = key[i]; if (v != 0x7fffffff && (i == 0 || key[i - 1] != v)) ++mine; }
    int tot;
    int pos = block_excl_scan(mine, lds, &tot);
    bool overflow = false;
    for (int i = lo; i < hi; ++i) {
        const int v = key[i];
        if (v != 0x7fffffff && (i == 0 || key[i - 1] != v)) {
            if (pos < n_cap) { out_ids[pos] = v; if (node_map) node_map[v] = pos; }
            else overflow = true;
            if (unmark_mult) unmark_mult[v] = 0;
            ++pos;
        }
    }
    if (tid == 0) { counts[0] = tot < n_cap ? tot : n_cap; counts[1] = counts[0]; }
    if (overflow && status) atomicOr(status, GRAPES_STATUS_NODE_OVERFLOW);
    (void)s_total;
}

__global__ __launch_bounds__(1024) void union_sorted_k(MarkLists L, int num_nodes, int n_cap, int32_t* __restrict__ out_ids,
                                                       int32_t* __restrict__ node_map, int32_t* __restrict__ counts,
                                                       int32_t* __restrict__ unmark_mult, int32_t* status) {
    union_sorted_body(L, num_nodes, n_cap, out_ids, node_map, counts, unmark_mult, status);
}

extern "C" int grapes_bitmap_clear(uint64_t* bits, const int32_t* ids, int64_t n, const int32_t* d_n,
                                   grapes_stream_t stream) {
    if (!bits || (!ids && n > 0) || n < 0) return GRAPES_EINVAL;
    if (n == 0) return 0;
    int grid = grapes_div_up(n, 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(bitmap_clear_k, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (unsigned long long*)bits, ids, n, d_n);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------- frontier compaction
// Sum of bsum[0..b) by the whole workgroup (integer => order-free, deterministic).
__device__ __forceinline__ int block_prefix_of_sums(const int32_t* __restrict__ bsum, int b, int* lds) {
    int acc = 0;
    for (int i = threadIdx.x; i < b; i += blockDim.x) acc += bsum[i];
    int tot;
    block_excl_scan(acc, lds, &tot);
    return tot;
}

// Two launches over ALL level-0 words of the bitmap (N/64 words: 38 K for ogbn-products, 1.7 M = 14 MB for
// papers100M — a few microseconds of streaming either way, cheaper than first listing the non-empty words through a
// summary level): (A) per-workgroup totals, (B) base offset from the totals + workgroup scan, ids emitted in ascending
// order, the words consumed (cleared).
__global__ __launch_bounds__(1024) void compact_count_k(const unsigned long long* __restrict__ bits,
                                                        const unsigned long long* __restrict__ prev_bits, int W,
                                                        int32_t* __restrict__ bsum_b, int32_t* __restrict__ bsum_n) {
    __shared__ int lds[17];
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    int cb = 0, cn = 0;
    if (q < W) {
        const unsigned long long bb = bits[q];
        if (bb) {
            const unsigned long long pp = prev_bits ? prev_bits[q] : 0ull;
            cb = __popcll(bb); cn = __popcll(bb & ~pp);
        }
    }
    int tb, tn;
    block_excl_scan(cb, lds, &tb);
    block_excl_scan(cn, lds, &tn);
    if (threadIdx.x == 0) { bsum_b[blockIdx.x] = tb; bsum_n[blockIdx.x] = tn; }
}

// (rare) work items (row, chunk) of a row longer than GRAPES_LONG_ROW, as prep_scan_emit_k writes them
__device__ __forceinline__ void long_row_items(const grapes_hop_degree_args& hd, int row, int ct, int cs) {
    if (ct > GRAPES_LONG_ROW) {
        const int nc = (ct + GRAPES_LONG_ROW - 1) / GRAPES_LONG_ROW;
        const int b0 = atomicAdd(&hd.n_long[0], nc);
        for (int c = 0; c < nc; ++c)
            if (b0 + c < hd.item_cap) { hd.long_items[2 * (b0 + c)] = row; hd.long_items[2 * (b0 + c) + 1] = c; }
    }
    if (cs > GRAPES_LONG_ROW) {
        const int nc = (cs + GRAPES_LONG_ROW - 1) / GRAPES_LONG_ROW;
        const int b0 = atomicAdd(&hd.n_long[1], nc);
        for (int c = 0; c < nc; ++c)
            if (b0 + c < hd.item_cap) { hd.long_items[2 * (hd.item_cap + b0 + c)] = row; hd.long_items[2 * (hd.item_cap + b0 + c) + 1] = c; }
    }
}

// sync != NULL: the ONE-launch form (<= GRAPES_SYNC_SLOTS workgroups) — the workgroup totals travel through `sync`
// (common.h: lookback_exclusive) instead of a counting launch + bsum arrays.
__device__ __forceinline__ void compact_emit_body(unsigned long long* __restrict__ bits,
                                                       const unsigned long long* __restrict__ prev_bits, int W,
                                                       const int32_t* __restrict__ bsum_b, const int32_t* __restrict__ bsum_n,
                                                       int n_cap, int32_t* __restrict__ batch_nodes,
                                                       int32_t* __restrict__ neighbor_nodes, int32_t* __restrict__ nb_local,
                                                       int32_t* __restrict__ node_map, int32_t* __restrict__ counts,
                                                       int32_t* status, uint32_t* __restrict__ ind_code,
                                                       uint32_t epoch_host, const uint32_t* d_epoch, int ind_bit,
                                                       unsigned long long* __restrict__ sync, int32_t* __restrict__ cand_pos,
                                                       uint32_t* __restrict__ zero_a, size_t words_a,
                                                       uint32_t* __restrict__ zero_b, size_t words_b,
                                                       uint32_t* __restrict__ zero_c, size_t words_c,
                                                       grapes_slice_remark_args rm, int gc,
                                                       grapes_hop_degree_args hd, const int BID, const int NBLK) {
    // gc = number of workgroups that compact (the first ones); workgroups beyond them only help with the side jobs of the launch
    // — the scratch clears and the slice marks: with a small bitmap (Reddit: 15 workgroups) and a large edge capacity (10 MB of
    // scratch to clear) the clears set the launch time (36 us)
    // The launch's side jobs are left to wavefronts 1 .. of EVERY workgroup (one partition of the work for compacting workgroups and
    // helpers alike): a compacting workgroup's wavefront 0 reads the predecessors' totals, and its loads would queue behind its share
    // of the 3 MB of clears in the wavefront's in-order memory pipeline (round 5: the look-back's stragglers).
    const int sj_t = blockDim.x > 64 ? (int)blockDim.x - 64 : (int)blockDim.x, sj_0 = blockDim.x > 64 ? (int)threadIdx.x - 64 : (int)threadIdx.x;
    if (BID >= gc) {
        if (sj_0 < 0) return;
        const size_t stride = (size_t)NBLK * sj_t, i0 = (size_t)BID * sj_t + sj_0;
        if (rm.mult) {
            if (rm.unmark_ids) { const int c = eff_count(rm.d_n_unmark, rm.n_unmark); for (size_t i = i0; i < (size_t)c; i += stride) rm.mult[rm.unmark_ids[i]] = 0; }
            if (rm.mark_ids) { const int c = eff_count(rm.d_n_mark, rm.n_mark); for (size_t i = i0; i < (size_t)c; i += stride) atomicAdd(&rm.mult[rm.mark_ids[i]], 1); }
        }
        for (size_t i = i0; i < words_a; i += stride) zero_a[i] = 0u;
        for (size_t i = i0; i < words_b; i += stride) zero_b[i] = 0u;
        for (size_t i = i0; i < words_c; i += stride) zero_c[i] = 0u;
        return;
    }
    __shared__ int lds[51];
    __shared__ unsigned long long lds64[2];
    // Order matters for latency: the words and the workgroup scan come FIRST and the workgroup's totals are published at once
    // (every later workgroup waits for them); the side jobs of this launch — the slice marks, the scratch of the launches that
    // follow — are plain stores issued while the predecessors' totals travel.  (They used to run first, and the barriers of
    // the scan then waited for their acknowledgement.)
    GRAPES_STAMP_NW(0);
    const uint32_t epoch = d_epoch ? (*d_epoch & 0xffffffu) : epoch_host;
    const int w = BID * blockDim.x + threadIdx.x;
    unsigned long long bb = 0ull, pp = 0ull;
    int wt = 0, wsv = 0;                        // hd: in-degree / out-degree sums of this word's nodes (the expansion's counts)
    if (w < W) {
        bb = bits[w];
        pp = prev_bits ? prev_bits[w] : 0ull;   // unconditional: in flight together with bits[w], not a round trip behind it
        if (hd.indeg) { wt = hd.wsum[w]; wsv = hd.wsum[W + w]; }
    }
    GRAPES_STAMP(1);                            // the words (+ word sums) have arrived
    int tb, tn;
    int posb, posn;
    int post = 0, poss = 0, tt = 0, ts = 0;
    if (blockDim.x <= 512 && hd.indeg) {    // the node counts (packed) and the two edge counts on ONE set of barriers (round 5)
        int pk = __popcll(bb) | (__popcll(bb & ~pp) << 16), tot;
        post = wt; poss = wsv;
        block_excl_scan3(pk, post, poss, lds, &tot, &tt, &ts);
        posb = pk & 0xffff; posn = (int)((unsigned)pk >> 16); tb = tot & 0xffff; tn = (int)((unsigned)tot >> 16);
    } else if (blockDim.x <= 512) {    // both counts in one scan (a workgroup's totals are <= 512 * 64 = 2^15 each)
        int tot;
        const int pk = block_excl_scan(__popcll(bb) | (__popcll(bb & ~pp) << 16), lds, &tot);
        // (unsigned shifts: 512 threads x 64 new neighbours each is 2^15, which the signed form would read back as -2^15)
        posb = pk & 0xffff; posn = (int)((unsigned)pk >> 16); tb = tot & 0xffff; tn = (int)((unsigned)tot >> 16);
    } else {
        posb = block_excl_scan(__popcll(bb), lds, &tb);
        posn = block_excl_scan(__popcll(bb & ~pp), lds, &tn);
    }
    GRAPES_STAMP_NW(7);                         // workgroup scan done
    if (sync && threadIdx.x == 0)            // publish (totals packed 31 + 31 bits: both grid-wide sums are node counts < 2^31)
        __hip_atomic_store(&sync[1 + BID], (1ull << 63) | ((unsigned long long)tn << 31) | (unsigned)tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- hd: the hop graph's degrees ride along (include/grapes_hip.h: grapes_hop_degree_args).  Two more scanned quantities —
    // edges into / out of the nodes before this one, in local = ascending global order — published in a second look-back word;
    // the per-node counts of this thread's first four nodes are requested NOW, so that they travel while the predecessors'
    // totals do (a thread rarely has more: a frontier fills ~1 bit per word)
    constexpr int PRE = 4;
    // (the RAW loaded words; which of them a node uses is decided where it is emitted, after the look-back — picking them here put
    // the wait for all sixteen loads in front of the side jobs and the look-back)
    int l_ct[PRE] = {0, 0, 0, 0}, l_lp[PRE] = {0, 0, 0, 0};
    uint32_t l_cd[PRE] = {0u, 0u, 0u, 0u};
    int2 l_sg[PRE] = {make_int2(0, 0), make_int2(0, 0), make_int2(0, 0), make_int2(0, 0)};
    if (hd.indeg) {
        if (blockDim.x > 512) {
            post = block_excl_scan(wt, lds, &tt);
            poss = block_excl_scan(wsv, lds, &ts);
        }
        if (sync && threadIdx.x == 0)
            __hip_atomic_store(&((unsigned long long*)hd.sync2)[1 + BID], (1ull << 63) | ((unsigned long long)(unsigned)ts << 31) | (unsigned)tt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (straight-line: every load is issued — from a clamped, valid address when the thread has no k-th node or the node is of
        // the other kind — and the values are picked afterwards.  With the loads inside `if (bit) … if (previous) … else …` the
        // compiler closed every branch with vmcnt(0): four dependent round trips, 3.4 us between the publish and the look-back of
        // every workgroup — profiles/r05_index_phase_stamps.txt, phases 1 -> 2 — instead of sixteen loads in flight.)
        GRAPES_STAMP_NW(8);                     // both look-back words published
        // The first wavefront reads the predecessors' totals NOW, in front of its own share of the counter requests below: issuing
        // those sixteen scattered loads takes the workgroup's wavefronts ~2 us of the CU's address path (stamps: published 2.7 us,
        // requested 4.7), and the look-back used to start behind them (done 7.2, emit 9.5).  The other wavefronts request their
        // counters meanwhile; everybody meets at the barrier where the sums are read.
        if (sync) { unsigned long long dummy2; (void)lookback_exclusive2(sync, (unsigned long long*)hd.sync2, BID, lds64, status, &dummy2, 1); }
        // A load that has no node of its kind reads entry 0 — the same address in every such lane, one request per wavefront: with
        // a per-lane dummy (the word's first node) the launch moved 40 MB of lines nobody needed and took 25 us.
        const uint32_t* icp = ind_code ? ind_code : reinterpret_cast<const uint32_t*>(hd.indeg);
        unsigned long long b2 = bb;
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            const bool has = b2 != 0ull;
            const int b = has ? __ffsll((long long)b2) - 1 : 0;
            b2 &= b2 - 1;                                         // (0 stays 0)
            const bool isprev = has && ((pp >> b) & 1ull) != 0ull;
            const int id = has ? w * 64 + b : 0, idp = isprev ? id : 0, idn = (has && !isprev) ? id : 0;
            l_ct[k] = hd.indeg[id];
            l_sg[k] = *reinterpret_cast<const int2*>(hd.seginfo + 2 * (long long)idp);
            l_lp[k] = hd.loops[idp];
            l_cd[k] = icp[idn];                                   // (its indicator word travels now, not after the look-back)
        }
        if (w < W) { if (wt) hd.wsum[w] = 0; if (wsv) hd.wsum[W + w] = 0; }      // consumed: zero at rest again
    }
    GRAPES_STAMP_NW(2);                         // scans done, totals published, counters requested
    if (w < W && bb) bits[w] = 0ull;         // consume
    if (rm.mult && sj_0 >= 0) {   // the slice marks that are due before this hop's expansion (grapes_slice_remark, its two id lists)
        const int stride = NBLK * sj_t, i0 = BID * sj_t + sj_0;
        if (rm.unmark_ids) { const int c = eff_count(rm.d_n_unmark, rm.n_unmark); for (int i = i0; i < c; i += stride) rm.mult[rm.unmark_ids[i]] = 0; }
        if (rm.mark_ids) { const int c = eff_count(rm.d_n_mark, rm.n_mark); for (int i = i0; i < c; i += stride) atomicAdd(&rm.mult[rm.mark_ids[i]], 1); }
    }
    if (sj_0 >= 0) {   // scratch of the launches that follow (grapes_gcn_prepare's counters, its csr_dst), cleared on the way
        const size_t stride = (size_t)NBLK * sj_t, i0 = (size_t)BID * sj_t + sj_0;
        for (size_t i = i0; i < words_a; i += stride) zero_a[i] = 0u;
        for (size_t i = i0; i < words_b; i += stride) zero_b[i] = 0u;
        for (size_t i = i0; i < words_c; i += stride) zero_c[i] = 0u;
    }
    GRAPES_STAMP_NW(3);                         // side jobs issued (marks, clears)
    int base_b, base_n, base_t = 0, base_s = 0;
    if (sync && hd.indeg) {
        unsigned long long pre2;
        const unsigned long long pre = lookback_exclusive2(sync, (unsigned long long*)hd.sync2, BID, lds64, status, &pre2, 2);
        lookback_finish(sync, gc, (unsigned long long*)hd.sync2);
        base_b = (int)(pre & 0x7fffffffull); base_n = (int)(pre >> 31);
        base_t = (int)(pre2 & 0x7fffffffull); base_s = (int)(pre2 >> 31);
    } else if (sync) {
        const unsigned long long pre = lookback_exclusive(sync, BID, 0ull, lds64, status, /*published=*/true);
        lookback_finish(sync, gc);
        base_b = (int)(pre & 0x7fffffffull); base_n = (int)(pre >> 31);
    } else {
        base_b = block_prefix_of_sums(bsum_b, BID, lds);
        base_n = block_prefix_of_sums(bsum_n, BID, lds);
    }
    GRAPES_STAMP_NW(4);                         // the predecessors' totals are here
    posb += base_b; posn += base_n;
    int pt = post + base_t, ps = poss + base_s;          // hd: edges into / out of the nodes before the next one emitted
    bool overflow = false;
    // one emitted node: the lists, the relabel table, the indicator bit; with hd also its rows' starts, its dinv, its segment
    auto emit = [&](int b, int ct, int2 sg, int lp, bool have_code = false, uint32_t code = 0u) {
        const int id = w * 64 + b;
        const bool isprev = ((pp >> b) & 1ull) != 0ull;
        if (hd.indeg) {                           // the counters go back to zero whatever happens to the node
            if (ct) hd.indeg[id] = 0;
            if (lp) hd.loops[id] = 0;
        }
        if (posb < n_cap) {
            batch_nodes[posb] = id;
            if (node_map) node_map[id] = posb;
            if (cand_pos) cand_pos[posb] = isprev ? -1 : posn;    // inverse of nb_local (-1: not a candidate)
            if (!isprev) {
                neighbor_nodes[posn] = id;
                nb_local[posn] = posb;
                ++posn;
                if (ind_code) {                   // main.py:191: indicator column `hop` of the new neighbours
                    uint32_t c = have_code ? code : ind_code[id];
                    if ((c >> 8) != epoch) c = epoch << 8;
                    ind_code[id] = c | (1u << ind_bit);
                }
            }
            if (hd.indeg) {
                int cs = isprev ? sg.y - lp : 0;
                cs = cs > 0 ? cs : 0;
                hd.rowptr_t[posb] = pt; hd.rowptr_s[posb] = ps;
                if (hd.cursor) hd.cursor[posb] = pt;
                hd.dinv[posb] = 1.0f / sqrtf((float)(ct + 1));       // deg = in-degree + unit self-loop (as prep_scan_emit_k)
                if (isprev) { hd.seg_first[posb] = sg.x; hd.row_loops[posb] = lp; }
                if (hd.long_items && (ct > GRAPES_LONG_ROW || cs > GRAPES_LONG_ROW)) long_row_items(hd, posb, ct, cs);
                pt += ct; ps += cs;
            }
        } else {
            overflow = true;
        }
        ++posb;
    };
    if (hd.indeg) {
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            if (bb) {
                const int b = __ffsll((long long)bb) - 1;
                bb &= bb - 1;
                const bool isprev = ((pp >> b) & 1ull) != 0ull;
                emit(b, l_ct[k], isprev ? l_sg[k] : make_int2(0, 0), isprev ? l_lp[k] : 0, true, (!isprev && ind_code) ? l_cd[k] : 0u);
            }
        }
    }
    // the rest of a DENSE word (a Reddit frontier sets ~20 of a word's 64 bits) in batches of CH nodes: everything a node's
    // emit reads — its degree counters, its indicator word — is requested for the whole batch first; walking bit by bit made
    // every node a dependent round trip of its own (lane-per-bit walks of a wavefront's 64 words measured slower still:
    // the chain is then 64 words long)
    constexpr int CH = 8;
    while (bb) {
        int bs[CH], cts[CH], lps[CH]; int2 sgs[CH]; uint32_t cds[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            bs[k] = -1; cts[k] = 0; lps[k] = 0; sgs[k] = make_int2(0, 0); cds[k] = 0u;
            if (bb) {
                const int b = __ffsll((long long)bb) - 1;
                bb &= bb - 1;
                bs[k] = b;
                const int id = w * 64 + b;
                const bool isprev = ((pp >> b) & 1ull) != 0ull;
                if (hd.indeg) {
                    cts[k] = hd.indeg[id];
                    if (isprev) { sgs[k] = *reinterpret_cast<const int2*>(hd.seginfo + 2 * (long long)id); lps[k] = hd.loops[id]; }
                }
                if (ind_code && !isprev) cds[k] = ind_code[id];
            }
        }
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (bs[k] >= 0) emit(bs[k], cts[k], sgs[k], lps[k], true, cds[k]);
    }
    GRAPES_STAMP_NW(5);                         // emit issued
    GRAPES_STAMP(6);                            // ... and landed
    if (BID == gc - 1 && threadIdx.x == 0) {
        const int nb = base_b + tb, nn = base_n + tn;
        counts[0] = nb < n_cap ? nb : n_cap;
        counts[1] = nn < n_cap ? nn : n_cap;
        if (hd.indeg) {
            const int nl = nb < n_cap ? nb : n_cap;
            hd.rowptr_t[nl] = base_t + tt; hd.rowptr_s[nl] = base_s + ts;
            if (hd.n_long) hd.n_long[2] = base_t + tt;       // aggregated (non-self-loop) edges, for the caller's metric
        }
    }
    if (overflow && status) atomicOr(status, GRAPES_STATUS_NODE_OVERFLOW);
}

struct CompactEmitArgs {
    unsigned long long* bits; const unsigned long long* prev_bits; int W; const int32_t* bsum_b; const int32_t* bsum_n; int n_cap;
    int32_t* batch_nodes; int32_t* neighbor_nodes; int32_t* nb_local; int32_t* node_map; int32_t* counts; int32_t* status;
    uint32_t* ind_code; uint32_t epoch_host; const uint32_t* d_epoch; int ind_bit; unsigned long long* sync; int32_t* cand_pos;
    uint32_t* zero_a; size_t words_a; uint32_t* zero_b; size_t words_b; uint32_t* zero_c; size_t words_c;
    grapes_slice_remark_args rm; int gc; grapes_hop_degree_args hd;
};
#define COMPACT_EMIT_CALL(A, bid, nblk)                                                                                              \
    compact_emit_body((A).bits, (A).prev_bits, (A).W, (A).bsum_b, (A).bsum_n, (A).n_cap, (A).batch_nodes, (A).neighbor_nodes,          \
                      (A).nb_local, (A).node_map, (A).counts, (A).status, (A).ind_code, (A).epoch_host, (A).d_epoch, (A).ind_bit,      \
                      (A).sync, (A).cand_pos, (A).zero_a, (A).words_a, (A).zero_b, (A).words_b, (A).zero_c, (A).words_c, (A).rm,       \
                      (A).gc, (A).hd, bid, nblk)
__global__ __launch_bounds__(1024) void compact_emit_k(CompactEmitArgs a) { COMPACT_EMIT_CALL(a, (int)blockIdx.x, (int)gridDim.x); }
// two compactions side by side in one launch (riders: common.h) — each with its own look-back scratch
__global__ __launch_bounds__(1024) void compact_emit_pair_k(CompactEmitArgs a, CompactEmitArgs b, int nA) {
    if ((int)blockIdx.x < nA) COMPACT_EMIT_CALL(a, (int)blockIdx.x, nA);
    else COMPACT_EMIT_CALL(b, (int)blockIdx.x - nA, (int)gridDim.x - nA);
}

// the step's final all_nodes (ONE workgroup) with a recorded compaction riding beside it (riders: common.h): workgroup 0 sorts
// the union, the others compact the rider's bitmap
__global__ __launch_bounds__(1024) void union_sorted_compact_pair_k(MarkLists L, int num_nodes, int n_cap, int32_t* __restrict__ out_ids,
                                                                   int32_t* __restrict__ node_map, int32_t* __restrict__ counts,
                                                                   int32_t* __restrict__ unmark_mult, int32_t* status,
                                                                   CompactEmitArgs b) {
    if (blockIdx.x == 0) union_sorted_body(L, num_nodes, n_cap, out_ids, node_map, counts, unmark_mult, status);
    else COMPACT_EMIT_CALL(b, (int)blockIdx.x - 1, (int)gridDim.x - 1);
}

extern "C" int grapes_union_sorted(const int32_t* ids0, int32_t n0, const int32_t* d_n0, const int32_t* ids1, int32_t n1,
                                   const int32_t* d_n1, const int32_t* ids2, int32_t n2, const int32_t* d_n2,
                                   const int32_t* ids3, int32_t n3, const int32_t* d_n3, int32_t num_nodes, int32_t n_cap,
                                   int32_t* out_ids, int32_t* node_map, int32_t* counts, int32_t* unmark_mult,
                                   int32_t* status, grapes_stream_t stream) {
    if (n0 < 0 || n1 < 0 || n2 < 0 || n3 < 0 || num_nodes <= 0 || n_cap <= 0 || !out_ids || !counts) return GRAPES_EINVAL;
    if ((long long)n0 + n1 + n2 + n3 > UNION_MAX) return GRAPES_EINVAL;
    MarkLists L{{n0 > 0 ? ids0 : nullptr, n1 > 0 ? ids1 : nullptr, n2 > 0 ? ids2 : nullptr, n3 > 0 ? ids3 : nullptr},
                {n0, n1, n2, n3}, {d_n0, d_n1, d_n2, d_n3}};
    if (const GrapesRiderRecord* r = grapes_rider_match(GRAPES_RK_COMPACT, 0, 1024, (hipStream_t)stream)) {
        CompactEmitArgs Bq; memcpy(&Bq, r->args, sizeof Bq);
        hipLaunchKernelGGL(union_sorted_compact_pair_k, dim3(1 + r->grid), dim3(1024), 0, (hipStream_t)stream, L, num_nodes, n_cap, out_ids,
                           node_map, counts, unmark_mult, status, Bq);
    } else {
        hipLaunchKernelGGL(union_sorted_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, L, num_nodes, n_cap, out_ids, node_map, counts,
                           unmark_mult, status);
    }
    GRAPES_LAUNCH_CHECK();
    return 0;
}


// ---- The one-launch compaction for HUGE bitmaps (papers100M: 1.7 M words): a thread owns WPT CONSECUTIVE words, so that the
// grid stays within the look-back scratch (1,695 workgroups of one word per thread -> 212 of eight) — the two-launch form it
// replaces streamed the bitmap twice on 1.7 M threads (35 us per hop) and could not carry the hop graph's degrees.  Same
// outputs, same side jobs; a frontier sets ~2 % of such a bitmap's words, so a thread's words are mostly empty.
template <int WPT>
__global__ __launch_bounds__(1024) void compact_emit_wide_k(unsigned long long* __restrict__ bits,
                                                            const unsigned long long* __restrict__ prev_bits, int W,
                                                            int n_cap, int32_t* __restrict__ batch_nodes,
                                                            int32_t* __restrict__ neighbor_nodes, int32_t* __restrict__ nb_local,
                                                            int32_t* __restrict__ node_map, int32_t* __restrict__ counts,
                                                            int32_t* status, uint32_t* __restrict__ ind_code,
                                                            uint32_t epoch_host, const uint32_t* d_epoch, int ind_bit,
                                                            unsigned long long* __restrict__ sync, int32_t* __restrict__ cand_pos,
                                                            uint32_t* __restrict__ zero_a, size_t words_a,
                                                            uint32_t* __restrict__ zero_b, size_t words_b,
                                                            uint32_t* __restrict__ zero_c, size_t words_c,
                                                            grapes_slice_remark_args rm, int gc, grapes_hop_degree_args hd) {
    const size_t gstride = (size_t)gridDim.x * blockDim.x, gi0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto side_jobs = [&]() {
        if (rm.mult) {
            if (rm.unmark_ids) { const int c = eff_count(rm.d_n_unmark, rm.n_unmark); for (size_t i = gi0; i < (size_t)c; i += gstride) rm.mult[rm.unmark_ids[i]] = 0; }
            if (rm.mark_ids) { const int c = eff_count(rm.d_n_mark, rm.n_mark); for (size_t i = gi0; i < (size_t)c; i += gstride) atomicAdd(&rm.mult[rm.mark_ids[i]], 1); }
        }
        for (size_t i = gi0; i < words_a; i += gstride) zero_a[i] = 0u;
        for (size_t i = gi0; i < words_b; i += gstride) zero_b[i] = 0u;
        for (size_t i = gi0; i < words_c; i += gstride) zero_c[i] = 0u;
    };
    if ((int)blockIdx.x >= gc) { side_jobs(); return; }
    __shared__ int lds[17];
    __shared__ unsigned long long lds64[2];
    const uint32_t epoch = d_epoch ? (*d_epoch & 0xffffffu) : epoch_host;
    const long long w0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * WPT;
    unsigned long long bb[WPT], pp[WPT];
    int wt = 0, wsv = 0, cb = 0, cn = 0;
    if (w0 + WPT <= W && (WPT & 1) == 0) {       // a thread's words as 16-byte loads (its 8 x WPT bytes are contiguous and aligned)
        const ulonglong2* b2 = reinterpret_cast<const ulonglong2*>(bits + w0);
        const ulonglong2* p2 = reinterpret_cast<const ulonglong2*>(prev_bits ? prev_bits + w0 : bits + w0);
#pragma unroll
        for (int k = 0; k < WPT / 2; ++k) { const ulonglong2 v = b2[k]; bb[2 * k] = v.x; bb[2 * k + 1] = v.y; }
#pragma unroll
        for (int k = 0; k < WPT / 2; ++k) {
            const ulonglong2 v = p2[k];
            pp[2 * k] = prev_bits ? v.x : 0ull; pp[2 * k + 1] = prev_bits ? v.y : 0ull;
        }
        if (hd.indeg) {
#pragma unroll
            for (int k = 0; k < WPT; ++k) { wt += hd.wsum[w0 + k]; wsv += hd.wsum[W + w0 + k]; }
        }
    } else {
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            const long long w = w0 + k;
            bb[k] = 0ull; pp[k] = 0ull;
            if (w < W) {
                bb[k] = bits[w];
                pp[k] = prev_bits ? prev_bits[w] : 0ull;
                if (hd.indeg) { wt += hd.wsum[w]; wsv += hd.wsum[W + w]; }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < WPT; ++k) { cb += __popcll(bb[k]); cn += __popcll(bb[k] & ~pp[k]); }
    int tb, tn, tt = 0, ts = 0;
    int posb = block_excl_scan(cb, lds, &tb);
    int posn = block_excl_scan(cn, lds, &tn);
    if (threadIdx.x == 0)
        (void)atomicExch(&sync[1 + blockIdx.x], (1ull << 63) | ((unsigned long long)tn << 31) | (unsigned)tb);
    int post = 0, poss = 0;
    if (hd.indeg) {
        post = block_excl_scan(wt, lds, &tt);
        poss = block_excl_scan(wsv, lds, &ts);
        if (threadIdx.x == 0)
            (void)atomicExch(&((unsigned long long*)hd.sync2)[1 + blockIdx.x], (1ull << 63) | ((unsigned long long)(unsigned)ts << 31) | (unsigned)tt);
    }
#pragma unroll
    for (int k = 0; k < WPT; ++k) {              // consume: the bitmap and the word sums are zero at rest
        const long long w = w0 + k;
        if (w < W) {
            if (bb[k]) bits[w] = 0ull;
            if (hd.indeg && (wt | wsv)) { hd.wsum[w] = 0; hd.wsum[W + w] = 0; }
        }
    }
    side_jobs();
    int base_b, base_n, base_t = 0, base_s = 0;
    if (hd.indeg) {
        unsigned long long pre2;
        const unsigned long long pre = lookback_exclusive2(sync, (unsigned long long*)hd.sync2, blockIdx.x, lds64, status, &pre2);
        lookback_finish(sync, gc, (unsigned long long*)hd.sync2);
        base_b = (int)(pre & 0x7fffffffull); base_n = (int)(pre >> 31);
        base_t = (int)(pre2 & 0x7fffffffull); base_s = (int)(pre2 >> 31);
    } else {
        const unsigned long long pre = lookback_exclusive(sync, blockIdx.x, 0ull, lds64, status, /*published=*/true);
        lookback_finish(sync, gc);
        base_b = (int)(pre & 0x7fffffffull); base_n = (int)(pre >> 31);
    }
    posb += base_b; posn += base_n;
    int pt = post + base_t, ps = poss + base_s;
    bool overflow = false;
    constexpr int CH = 8;
    unsigned nonempty = 0u;
#pragma unroll
    for (int k = 0; k < WPT; ++k) nonempty |= bb[k] ? (1u << k) : 0u;
#pragma unroll 1
    while (nonempty) {                           // the thread's non-empty words, in order (most threads: none)
        const int k = __ffs((int)nonempty) - 1;
        nonempty &= nonempty - 1;
        unsigned long long bk = 0ull, pk = 0ull;
#pragma unroll
        for (int q = 0; q < WPT; ++q) { bk = k == q ? bb[q] : bk; pk = k == q ? pp[q] : pk; }     // (selects: the arrays stay in registers)
        const long long w = w0 + k;
        while (bk) {                             // batches of CH nodes: their counter / indicator loads are requested first
            int bs[CH], cts[CH], lps[CH]; int2 sgs[CH]; uint32_t cds[CH];
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                bs[q] = -1; cts[q] = 0; lps[q] = 0; sgs[q] = make_int2(0, 0); cds[q] = 0u;
                if (bk) {
                    const int b = __ffsll((long long)bk) - 1;
                    bk &= bk - 1;
                    bs[q] = b;
                    const int id = (int)(w * 64 + b);
                    const bool isprev = ((pk >> b) & 1ull) != 0ull;
                    if (hd.indeg) {
                        cts[q] = hd.indeg[id];
                        if (isprev) { sgs[q] = *reinterpret_cast<const int2*>(hd.seginfo + 2 * (long long)id); lps[q] = hd.loops[id]; }
                    }
                    if (ind_code && !isprev) cds[q] = ind_code[id];
                }
            }
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                if (bs[q] < 0) continue;
                const int b = bs[q], ct = cts[q], lp = lps[q];
                const int2 sg = sgs[q];
                const int id = (int)(w * 64 + b);
                const bool isprev = ((pk >> b) & 1ull) != 0ull;
                if (hd.indeg) { if (ct) hd.indeg[id] = 0; if (lp) hd.loops[id] = 0; }
                if (posb < n_cap) {
                    batch_nodes[posb] = id;
                    if (node_map) node_map[id] = posb;
                    if (cand_pos) cand_pos[posb] = isprev ? -1 : posn;
                    if (!isprev) {
                        neighbor_nodes[posn] = id;
                        nb_local[posn] = posb;
                        ++posn;
                        if (ind_code) {
                            uint32_t c = cds[q];
                            if ((c >> 8) != epoch) c = epoch << 8;
                            ind_code[id] = c | (1u << ind_bit);
                        }
                    }
                    if (hd.indeg) {
                        int cs = isprev ? sg.y - lp : 0;
                        cs = cs > 0 ? cs : 0;
                        hd.rowptr_t[posb] = pt; hd.rowptr_s[posb] = ps;
                        if (hd.cursor) hd.cursor[posb] = pt;
                        hd.dinv[posb] = 1.0f / sqrtf((float)(ct + 1));
                        if (isprev) { hd.seg_first[posb] = sg.x; hd.row_loops[posb] = lp; }
                        if (hd.long_items && (ct > GRAPES_LONG_ROW || cs > GRAPES_LONG_ROW)) long_row_items(hd, posb, ct, cs);
                        pt += ct; ps += cs;
                    }
                } else {
                    overflow = true;
                }
                ++posb;
            }
        }
    }
    if ((int)blockIdx.x == gc - 1 && threadIdx.x == 0) {
        const int nb = base_b + tb, nn = base_n + tn;
        counts[0] = nb < n_cap ? nb : n_cap;
        counts[1] = nn < n_cap ? nn : n_cap;
        if (hd.indeg) {
            const int nl = nb < n_cap ? nb : n_cap;
            hd.rowptr_t[nl] = base_t + tt; hd.rowptr_s[nl] = base_s + ts;
            if (hd.n_long) hd.n_long[2] = base_t + tt;
        }
    }
    if (overflow && status) atomicOr(status, GRAPES_STATUS_NODE_OVERFLOW);
}

// ---- SMALL bitmaps (<= COMPACT_SMALL_W = 4096 words: Reddit 3,640, arxiv 2,646, Cora 43), whose frontiers are DENSE — a Reddit hop sets
// ~20 of a word's 64 bits, Cora most of them.  The one-launch kernel gives a word to a thread, which then walks its bits one
// after the other (a chain of dependent emits on 1/64 of the lanes: 41 us on Reddit's 77k-node hop for 3,640 threads of
// work).  Two short launches instead: (1) ONE workgroup scans the words' counts (eight words per thread) and leaves every
// word's exclusive prefixes; (2) one WAVEFRONT per word, lane = bit: a word's nodes are emitted by one store instruction per
// list (their ids are consecutive), their counters come by one load, the row starts of the counted build by a wavefront scan.
// Same outputs, same side jobs, the bitmap consumed.
#define COMPACT_SMALL_W 4096
// exclusive scan of a 64-bit value over the workgroup (two counts packed 32 + 32: both grid-wide sums stay below 2^31)
__device__ __forceinline__ unsigned long long block_excl_scan_u64(unsigned long long v, unsigned long long* lds /* 17 words */,
                                                                  unsigned long long* total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    unsigned long long incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    __syncthreads();
    if (lane == 63) lds[wid] = incl;
    __syncthreads();
    if (wid == 0) {
        unsigned long long x = lane < nw ? lds[lane] : 0ull, xs = x;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const unsigned long long t = __shfl_up(xs, d, 64);
            if (lane >= d) xs += t;
        }
        if (lane < nw) lds[lane] = xs - x;
        if (lane == nw - 1) lds[16] = xs;
    }
    __syncthreads();
    *total = lds[16];
    return lds[wid] + incl - v;
}
__global__ __launch_bounds__(1024) void compact_small_scan_k(const unsigned long long* __restrict__ bits,
                                                             const unsigned long long* __restrict__ prev_bits, int W, int n_cap,
                                                             int32_t* __restrict__ pre /* [4][W] */, int32_t* __restrict__ counts,
                                                             grapes_hop_degree_args hd) {
    // The words' counts go through LDS: read from memory COALESCED (thread t takes words t, t + T, ...), scanned by the thread that
    // owns eight CONSECUTIVE words, and the prefixes leave coalesced again.  (A thread loading its own eight consecutive words
    // made every load instruction of the one workgroup 64 separate requests: 16k requests through one compute unit, 14 us.)
    __shared__ unsigned long long lds[17];
    __shared__ int s_bn[COMPACT_SMALL_W];          // nodes | new neighbours << 16 of a word; then the two prefixes' low / high
    __shared__ int s_t[COMPACT_SMALL_W], s_s[COMPACT_SMALL_W];
    constexpr int WPT = COMPACT_SMALL_W / 1024;
    const int T = blockDim.x;
    for (int w = threadIdx.x; w < W; w += T) {
        const unsigned long long bb = bits[w], pp = prev_bits ? prev_bits[w] : 0ull;
        s_bn[w] = __popcll(bb) | (__popcll(bb & ~pp) << 16);
        if (hd.indeg) { s_t[w] = hd.wsum[w]; s_s[w] = hd.wsum[W + w]; }
    }
    __syncthreads();
    const int w0 = threadIdx.x * WPT;
    int cb[WPT], cn[WPT], ct[WPT], cs[WPT];
    int sb = 0, sn = 0, st = 0, ss = 0;
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
        const int w = w0 + k;
        cb[k] = cn[k] = ct[k] = cs[k] = 0;
        if (w < W) {
            const int v = s_bn[w];
            cb[k] = v & 0xffff; cn[k] = v >> 16;
            if (hd.indeg) { ct[k] = s_t[w]; cs[k] = s_s[w]; }
        }
        sb += cb[k]; sn += cn[k]; st += ct[k]; ss += cs[k];
    }
    // two packed scans (nodes | new neighbours, edges in | edges out) instead of four
    unsigned long long tot1, tot2 = 0ull;
    const unsigned long long p1 = block_excl_scan_u64(((unsigned long long)(unsigned)sn << 32) | (unsigned)sb, lds, &tot1);
    int pb = (int)(p1 & 0xffffffffull), pn = (int)(p1 >> 32), pt = 0, ps = 0;
    const int tb = (int)(tot1 & 0xffffffffull), tn = (int)(tot1 >> 32);
    if (hd.indeg) {
        const unsigned long long p2 = block_excl_scan_u64(((unsigned long long)(unsigned)ss << 32) | (unsigned)st, lds, &tot2);
        pt = (int)(p2 & 0xffffffffull); ps = (int)(p2 >> 32);
    }
    const int tt = (int)(tot2 & 0xffffffffull), ts = (int)(tot2 >> 32);
    // (the scans' last barrier is behind every thread's reads of its words: the arrays are free to hold the prefixes)
    __shared__ int s_pn[COMPACT_SMALL_W];
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
        const int w = w0 + k;
        if (w < W) {
            s_bn[w] = pb; s_pn[w] = pn;
            if (hd.indeg) { s_t[w] = pt; s_s[w] = ps; }
        }
        pb += cb[k]; pn += cn[k]; pt += ct[k]; ps += cs[k];
    }
    __syncthreads();
    for (int w = threadIdx.x; w < W; w += T) {
        pre[w] = s_bn[w]; pre[W + w] = s_pn[w];
        if (hd.indeg) { pre[2 * W + w] = s_t[w]; pre[3 * W + w] = s_s[w]; }
    }
    if (threadIdx.x == 0) {
        counts[0] = tb < n_cap ? tb : n_cap;
        counts[1] = tn < n_cap ? tn : n_cap;
        if (hd.indeg) {
            const int nl = tb < n_cap ? tb : n_cap;
            hd.rowptr_t[nl] = tt; hd.rowptr_s[nl] = ts;
            if (hd.n_long) hd.n_long[2] = tt;
        }
    }
}
__global__ __launch_bounds__(256) void compact_small_emit_k(unsigned long long* __restrict__ bits,
                                                            const unsigned long long* __restrict__ prev_bits, int W, int n_cap,
                                                            const int32_t* __restrict__ pre, int32_t* __restrict__ batch_nodes,
                                                            int32_t* __restrict__ neighbor_nodes, int32_t* __restrict__ nb_local,
                                                            int32_t* __restrict__ node_map, int32_t* status,
                                                            uint32_t* __restrict__ ind_code, uint32_t epoch_host,
                                                            const uint32_t* d_epoch, int ind_bit, int32_t* __restrict__ cand_pos,
                                                            uint32_t* __restrict__ zero_a, size_t words_a,
                                                            uint32_t* __restrict__ zero_b, size_t words_b,
                                                            uint32_t* __restrict__ zero_c, size_t words_c,
                                                            grapes_slice_remark_args rm, int gc, grapes_hop_degree_args hd) {
    {   // the launch's side jobs, shared by every workgroup (helpers beyond gc do nothing else)
        const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (rm.mult) {
            if (rm.unmark_ids) { const int c = eff_count(rm.d_n_unmark, rm.n_unmark); for (size_t i = i0; i < (size_t)c; i += stride) rm.mult[rm.unmark_ids[i]] = 0; }
            if (rm.mark_ids) { const int c = eff_count(rm.d_n_mark, rm.n_mark); for (size_t i = i0; i < (size_t)c; i += stride) atomicAdd(&rm.mult[rm.mark_ids[i]], 1); }
        }
        for (size_t i = i0; i < words_a; i += stride) zero_a[i] = 0u;
        for (size_t i = i0; i < words_b; i += stride) zero_b[i] = 0u;
        for (size_t i = i0; i < words_c; i += stride) zero_c[i] = 0u;
    }
    if ((int)blockIdx.x >= gc) return;
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);                 // one wavefront per word
    if (w >= W) return;
    const unsigned long long bb = bits[w];
    if (bb == 0ull) return;                                            // (uniform over the wavefront)
    const unsigned long long pp = prev_bits ? prev_bits[w] : 0ull;
    const uint32_t epoch = d_epoch ? (*d_epoch & 0xffffffu) : epoch_host;
    const int pb0 = pre[w], pn0 = pre[W + w];
    const int pt0 = hd.indeg ? pre[2 * W + w] : 0, ps0 = hd.indeg ? pre[3 * W + w] : 0;
    const bool bit = ((bb >> lane) & 1ull) != 0ull, isprev = ((pp >> lane) & 1ull) != 0ull;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int id = w * 64 + lane;
    int ct = 0, lp = 0; int2 sg = make_int2(0, 0); uint32_t code = 0u;
    if (bit) {
        if (hd.indeg) {
            ct = hd.indeg[id];
            if (isprev) { sg = *reinterpret_cast<const int2*>(hd.seginfo + 2 * (long long)id); lp = hd.loops[id]; }
        }
        if (ind_code && !isprev) code = ind_code[id];
    }
    int cs = (bit && isprev) ? sg.y - lp : 0;
    cs = cs > 0 ? cs : 0;
    const int ctx = hd.indeg ? wave_incl_scan(ct) - ct : 0, csx = hd.indeg ? wave_incl_scan(cs) - cs : 0;
    if (lane == 0) {                                                   // consumed: the bitmap and the word sums are zero at rest
        bits[w] = 0ull;
        if (hd.indeg) { hd.wsum[w] = 0; hd.wsum[W + w] = 0; }
    }
    if (!bit) return;
    const int pb = pb0 + __popcll(bb & below), pn = pn0 + __popcll(bb & ~pp & below);
    if (hd.indeg) { if (ct) hd.indeg[id] = 0; if (lp) hd.loops[id] = 0; }
    if (pb >= n_cap) { if (status) atomicOr(status, GRAPES_STATUS_NODE_OVERFLOW); return; }
    batch_nodes[pb] = id;
    if (node_map) node_map[id] = pb;
    if (cand_pos) cand_pos[pb] = isprev ? -1 : pn;
    if (!isprev) {
        neighbor_nodes[pn] = id;
        nb_local[pn] = pb;
        if (ind_code) {
            uint32_t c = code;
            if ((c >> 8) != epoch) c = epoch << 8;
            ind_code[id] = c | (1u << ind_bit);
        }
    }
    if (hd.indeg) {
        hd.rowptr_t[pb] = pt0 + ctx; hd.rowptr_s[pb] = ps0 + csx;
        if (hd.cursor) hd.cursor[pb] = pt0 + ctx;
        hd.dinv[pb] = 1.0f / sqrtf((float)(ct + 1));
        if (isprev) { hd.seg_first[pb] = sg.x; hd.row_loops[pb] = lp; }
        if (hd.long_items && (ct > GRAPES_LONG_ROW || cs > GRAPES_LONG_ROW)) long_row_items(hd, pb, ct, cs);
    }
}

#define COMPACT_WIDE_WPT 8

static inline int compact_blocks(int num_nodes) { return grapes_div_up(((int64_t)num_nodes + 63) / 64, 1024); }

extern "C" size_t grapes_frontier_compact_workspace_bytes(int32_t n_cap, int32_t num_nodes) {
    (void)n_cap;
    const size_t W = ((size_t)(num_nodes > 0 ? num_nodes : 1) + 63) / 64;
    const size_t small = W <= COMPACT_SMALL_W ? 4 * W : 0;           // per-word prefixes of the small-bitmap form
    return (4 + 2 * (size_t)compact_blocks(num_nodes > 0 ? num_nodes : 1) + small) * sizeof(int32_t);
}

extern "C" int grapes_frontier_compact_counted(uint64_t* bits, uint64_t* bits1, const uint64_t* prev_bits,
                                       int32_t num_nodes, int32_t n_cap, int32_t* batch_nodes,
                                       int32_t* neighbor_nodes, int32_t* nb_local, int32_t* node_map,
                                       int32_t* counts, uint32_t* ind_code, uint32_t epoch, const uint32_t* d_epoch,
                                       int32_t ind_bit, int32_t* cand_pos, void* zero_a, size_t zero_a_words, void* zero_b,
                                       size_t zero_b_words, void* zero_c, size_t zero_c_words,
                                       const grapes_slice_remark_args* remark, void* workspace,
                                       uint64_t* sync, int32_t* status, const grapes_hop_degree_args* degrees,
                                       grapes_stream_t stream) {
    (void)bits1;     // the summary level of earlier versions is no longer used (may be NULL)
    if (!bits || !batch_nodes || !neighbor_nodes || !nb_local || !counts || !workspace || num_nodes <= 0 || n_cap <= 0)
        return GRAPES_EINVAL;
    grapes_hop_degree_args hd{};
    if (degrees) {
        hd = *degrees;
        if (!hd.indeg || !hd.loops || !hd.seginfo || !hd.wsum || !hd.rowptr_t || !hd.rowptr_s || !hd.dinv || !hd.seg_first ||
            !hd.row_loops || !hd.sync2 || !sync || !prev_bits)
            return GRAPES_EINVAL;
        if (hd.long_items && (!hd.n_long || hd.item_cap <= 0)) return GRAPES_EINVAL;
    }
    if (ind_code && (ind_bit < 0 || ind_bit > 7 || epoch >= (1u << 24))) return GRAPES_EINVAL;
    grapes_slice_remark_args crm{};
    if (remark) {
        crm = *remark;
        if (!crm.mult || crm.n_unmark < 0 || crm.n_mark < 0 || (crm.n_unmark > 0 && !crm.unmark_ids) || (crm.n_mark > 0 && !crm.mark_ids))
            return GRAPES_EINVAL;
        if (crm.n_unmark == 0) crm.unmark_ids = nullptr;
        if (crm.n_mark == 0) crm.mark_ids = nullptr;
    }
    hipStream_t s = (hipStream_t)stream;
    const int W = (int)(((int64_t)num_nodes + 63) / 64);
    const int G = compact_blocks(num_nodes);
    int32_t* bsum_b = (int32_t*)workspace + 4;
    int32_t* bsum_n = bsum_b + G;
    static int one_t = -1;      // threads per workgroup of the one-launch form (GRAPES_COMPACT_THREADS; default below)
    if (one_t < 0) { const char* e = grapes_tune_env("GRAPES_COMPACT_THREADS"); one_t = e ? atoi(e) : 256; if (one_t != 64 && one_t != 128 && one_t != 256 && one_t != 512) one_t = 1024; }
    // (the counted form does more per word — two more scans, the degree loads: 512-thread workgroups, half as many predecessors
    // to look back over, measured 10 us/step faster than 256 there; without the degrees 256 was the faster one)
    static int one_t_env = -1;
    if (one_t_env < 0) one_t_env = grapes_tune_env("GRAPES_COMPACT_THREADS") ? 1 : 0;
    int T1 = (degrees && !one_t_env && W >= 16384) ? 512 : one_t;
    // a small bitmap (Reddit: 3,640 words, arxiv 2,646) as ~64 workgroups of 64 / 128 threads rather than 15 of 256: the dense
    // words' bit-by-bit emit is the launch there, and it runs on as many compute units as there are workgroups (-19 us on Reddit)
    if (!one_t_env && W >= 1024) { while (T1 > 64 && grapes_div_up(W, T1) < 48) T1 >>= 1; }      // (a graph of one workgroup — Cora — keeps 256 threads: they share the launch's clears)
    if (grapes_rider_recording()) T1 = 1024;      // (a recorded compaction rides beside union_sorted_k: 1024-thread workgroups)
    while (T1 < 1024 && grapes_div_up(W, T1) > GRAPES_SYNC_SLOTS) T1 *= 2;
    const int G1 = grapes_div_up(W, T1);
    static int small_on = -1;              // GRAPES_COMPACT_SMALL=0 (A/B): small bitmaps through the one-launch kernel as well
    if (small_on < 0) { const char* e = grapes_tune_env("GRAPES_COMPACT_SMALL"); small_on = (e && atoi(e) == 0) ? 0 : 1; }
    if (small_on && W <= COMPACT_SMALL_W) {
        int32_t* pre = bsum_n + G;                                     // [4][W] behind the two-launch form's block sums
        int st_threads = grapes_div_up(grapes_div_up(W, COMPACT_SMALL_W / 1024), 64) * 64;      // four words per thread
        if (st_threads < 64) st_threads = 64;
        GRAPES_RIDER_OTHER(s, hipLaunchKernelGGL(compact_small_scan_k, dim3(1), dim3(st_threads), 0, s, (const unsigned long long*)bits,
                           (const unsigned long long*)prev_bits, W, n_cap, pre, counts, hd));
        GRAPES_LAUNCH_CHECK();
        const int gcs = grapes_div_up(W, 4);
        const size_t zw = (zero_a ? zero_a_words : 0) + (zero_b ? zero_b_words : 0) + (zero_c ? zero_c_words : 0);
        int GZ = (int)(zw / 4096 > 960 ? 960 : zw / 4096) - gcs;
        if (GZ < 0) GZ = 0;
        GRAPES_RIDER_OTHER(s, hipLaunchKernelGGL(compact_small_emit_k, dim3(gcs + GZ), dim3(256), 0, s, (unsigned long long*)bits,
                           (const unsigned long long*)prev_bits, W, n_cap, (const int32_t*)pre, batch_nodes, neighbor_nodes, nb_local,
                           node_map, status, ind_code, epoch, d_epoch, ind_bit, cand_pos, (uint32_t*)zero_a,
                           zero_a ? zero_a_words : 0, (uint32_t*)zero_b, zero_b ? zero_b_words : 0, (uint32_t*)zero_c,
                           zero_c ? zero_c_words : 0, crm, gcs, hd));
        GRAPES_LAUNCH_CHECK();
        return 0;
    }
    static int wide_force = -1;            // GRAPES_COMPACT_WIDE=2: the eight-words-per-thread kernel for every bitmap (tests)
    if (wide_force < 0) { const char* e = grapes_tune_env("GRAPES_COMPACT_WIDE"); wide_force = (e && atoi(e) == 2) ? 1 : 0; }
    if (sync && G1 <= GRAPES_SYNC_SLOTS && !wide_force) {
        // helper workgroups for the side jobs when the compaction itself is small: ~16k words of clearing per workgroup
        const size_t zw = (zero_a ? zero_a_words : 0) + (zero_b ? zero_b_words : 0) + (zero_c ? zero_c_words : 0);
        int GZ = (int)(zw / 16384 > 240 ? 240 : zw / 16384) - G1;
        if (GZ < 0) GZ = 0;
        const CompactEmitArgs A{(unsigned long long*)bits, (const unsigned long long*)prev_bits, W, nullptr, nullptr, n_cap, batch_nodes,
                                neighbor_nodes, nb_local, node_map, counts, status, ind_code, epoch, d_epoch, ind_bit,
                                (unsigned long long*)sync, cand_pos, (uint32_t*)zero_a, zero_a ? zero_a_words : 0, (uint32_t*)zero_b,
                                zero_b ? zero_b_words : 0, (uint32_t*)zero_c, zero_c ? zero_c_words : 0, crm, G1, hd};
        const int grid = G1 + GZ;
        auto single = [=](hipStream_t s_) { hipLaunchKernelGGL(compact_emit_k, dim3(grid), dim3(T1), 0, s_, A); };
        if (grapes_rider_recording()) { grapes_rider_record(grapes_rider_make(GRAPES_RK_COMPACT, 0, grid, T1, A, single)); return 0; }
        const GrapesRiderRecord* r = grapes_rider_match(GRAPES_RK_COMPACT, 0, T1, s);
        if (r) {
            CompactEmitArgs Bq; memcpy(&Bq, r->args, sizeof Bq);
            if (Bq.sync == A.sync || (Bq.hd.sync2 && Bq.hd.sync2 == A.hd.sync2)) return GRAPES_EINVAL;   // (one look-back scratch each)
            hipLaunchKernelGGL(compact_emit_pair_k, dim3(grid + r->grid), dim3(T1), 0, s, A, Bq, grid);
        } else {
            single(s);
        }
        GRAPES_LAUNCH_CHECK();
        return 0;
    }
    // a bitmap beyond the look-back scratch at one word per thread: eight consecutive words per thread (compact_emit_wide_k)
    const int GW = grapes_div_up(W, 1024 * COMPACT_WIDE_WPT);
    static int wide_off = -1;
    if (wide_off < 0) { const char* e = grapes_tune_env("GRAPES_COMPACT_WIDE"); wide_off = (e && atoi(e) == 0) ? 1 : 0; }
    if (sync && GW <= GRAPES_SYNC_SLOTS && !wide_off) {
        const size_t zw = (zero_a ? zero_a_words : 0) + (zero_b ? zero_b_words : 0) + (zero_c ? zero_c_words : 0);
        int GZ = (int)(zw / 16384 > 240 ? 240 : zw / 16384) - GW;
        if (GZ < 0) GZ = 0;
        GRAPES_RIDER_OTHER(s, hipLaunchKernelGGL(compact_emit_wide_k<COMPACT_WIDE_WPT>, dim3(GW + GZ), dim3(1024), 0, s, (unsigned long long*)bits,
                           (const unsigned long long*)prev_bits, W, n_cap, batch_nodes, neighbor_nodes, nb_local, node_map, counts,
                           status, ind_code, epoch, d_epoch, ind_bit, (unsigned long long*)sync, cand_pos, (uint32_t*)zero_a,
                           zero_a ? zero_a_words : 0, (uint32_t*)zero_b, zero_b ? zero_b_words : 0, (uint32_t*)zero_c,
                           zero_c ? zero_c_words : 0, crm, GW, hd));
        GRAPES_LAUNCH_CHECK();
        return 0;
    }
    if (degrees) return GRAPES_EINVAL;          // (the counted form is the one-launch form only)
    GRAPES_RIDER_OTHER(s, hipLaunchKernelGGL(compact_count_k, dim3(G), dim3(1024), 0, s, (const unsigned long long*)bits,
                       (const unsigned long long*)prev_bits, W, bsum_b, bsum_n));
    GRAPES_LAUNCH_CHECK();
    const CompactEmitArgs A2{(unsigned long long*)bits, (const unsigned long long*)prev_bits, W, (const int32_t*)bsum_b,
                             (const int32_t*)bsum_n, n_cap, batch_nodes, neighbor_nodes, nb_local, node_map, counts, status, ind_code,
                             epoch, d_epoch, ind_bit, nullptr, cand_pos, (uint32_t*)zero_a, zero_a ? zero_a_words : 0, (uint32_t*)zero_b,
                             zero_b ? zero_b_words : 0, (uint32_t*)zero_c, zero_c ? zero_c_words : 0, crm, G, grapes_hop_degree_args{}};
    GRAPES_RIDER_OTHER(s, hipLaunchKernelGGL(compact_emit_k, dim3(G), dim3(1024), 0, s, A2));
    GRAPES_LAUNCH_CHECK();
    return 0;
}
extern "C" int grapes_frontier_compact(uint64_t* bits, uint64_t* bits1, const uint64_t* prev_bits,
                                       int32_t num_nodes, int32_t n_cap, int32_t* batch_nodes,
                                       int32_t* neighbor_nodes, int32_t* nb_local, int32_t* node_map,
                                       int32_t* counts, uint32_t* ind_code, uint32_t epoch, const uint32_t* d_epoch,
                                       int32_t ind_bit, int32_t* cand_pos, void* zero_a, size_t zero_a_words, void* zero_b,
                                       size_t zero_b_words, void* zero_c, size_t zero_c_words,
                                       const grapes_slice_remark_args* remark, void* workspace,
                                       uint64_t* sync, int32_t* status, grapes_stream_t stream) {
    return grapes_frontier_compact_counted(bits, bits1, prev_bits, num_nodes, n_cap, batch_nodes, neighbor_nodes, nb_local, node_map,
                                           counts, ind_code, epoch, d_epoch, ind_bit, cand_pos, zero_a, zero_a_words, zero_b,
                                           zero_b_words, zero_c, zero_c_words, remark, workspace, sync, status, nullptr, stream);
}

// ---------------------------------------------------------------------------- A3 slice_adjacency
__global__ void slice_mark_k(int32_t* __restrict__ mult, const int32_t* __restrict__ cols, int c_host,
                             const int32_t* d_c, int unmark, unsigned long long* __restrict__ clear_bits) {
    const int c = eff_count(d_c, c_host);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < c; i += gridDim.x * blockDim.x) {
        const int id = cols[i];
        if (unmark) mult[id] = 0;
        else atomicAdd(&mult[id], 1);
        if (clear_bits) clear_bits[id >> 6] = 0ull;       // the hop's `previous` bitmap is done with (same id list)
    }
}

// One launch between two hops: previous_nodes changes from (targets + samples of hop h-1) to (targets + samples of hop
// h), and the sample sets of consecutive hops are disjoint from each other and from the targets (a hop samples among the
// nodes NOT in its previous_nodes), so un-marking the old samples and marking the new ones cannot touch the same entry.
__global__ void slice_remark_k(int32_t* __restrict__ mult, const int32_t* __restrict__ un_ids, int un_host,
                               const int32_t* d_un, const int32_t* __restrict__ mk_ids, int mk_host, const int32_t* d_mk,
                               unsigned long long* __restrict__ clear_bits, const int32_t* __restrict__ cl_ids, int cl_host,
                               const int32_t* d_cl) {
    const int stride = gridDim.x * blockDim.x, i0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (un_ids) {
        const int n = eff_count(d_un, un_host);
        for (int i = i0; i < n; i += stride) mult[un_ids[i]] = 0;
    }
    if (mk_ids) {
        const int n = eff_count(d_mk, mk_host);
        for (int i = i0; i < n; i += stride) atomicAdd(&mult[mk_ids[i]], 1);
    }
    if (cl_ids) {
        const int n = eff_count(d_cl, cl_host);
        for (int i = i0; i < n; i += stride) clear_bits[cl_ids[i] >> 6] = 0ull;
    }
}

// Ordered filter of the expanded edge list: edge t survives mult[dst[t]] times.  Stage 1: per
// workgroup (1024 edges) survivor totals; stage 2: base offset from the totals + block scan, so the
// output keeps the expansion order (row-major over rows, ascending column inside a row).
__global__ __launch_bounds__(1024) void slice_count_k(const int32_t* __restrict__ mult, const int32_t* __restrict__ dst,
                                                      int e_host, const int32_t* d_e, int32_t* __restrict__ bsum) {
    __shared__ int lds[17];
    const int e = eff_count(d_e, e_host);
    if (blockIdx.x * blockDim.x >= e) return;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = t < e ? mult[dst[t]] : 0;
    int tot;
    block_excl_scan(c, lds, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(1024) void slice_emit_k(const int32_t* __restrict__ mult, const int32_t* __restrict__ src,
                                                     const int32_t* __restrict__ dst, int e_host, const int32_t* d_e,
                                                     const int32_t* __restrict__ bsum, int out_cap,
                                                     int32_t* __restrict__ out_src, int32_t* __restrict__ out_dst,
                                                     int32_t* d_out_count, int32_t* status) {
    __shared__ int lds[17];
    const int e = eff_count(d_e, e_host);
    if (e == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && d_out_count) *d_out_count = 0;
        return;
    }
    if (blockIdx.x * blockDim.x >= e) return;
    const int base = block_prefix_of_sums(bsum, blockIdx.x, lds);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    int c = 0, s = 0, d = 0;
    if (t < e) { d = dst[t]; s = src[t]; c = mult[d]; }
    int tot;
    int pos = base + block_excl_scan(c, lds, &tot);
    bool overflow = false;
    for (int r = 0; r < c; ++r, ++pos) {
        if (pos < out_cap) { out_src[pos] = s; out_dst[pos] = d; }
        else overflow = true;
    }
    const bool last = (blockIdx.x + 1) * blockDim.x >= e;
    if (last && threadIdx.x == 0 && d_out_count) {
        const int total = base + tot;
        *d_out_count = total < out_cap ? total : out_cap;
    }
    if (overflow && status) atomicOr(status, GRAPES_STATUS_EDGE_OVERFLOW);
}

// The filter in ONE launch: four consecutive edges per thread (<= GRAPES_SYNC_SLOTS workgroups up to ~1M edges of
// capacity), workgroup totals through the look-back scratch.
#define SLICE_IPT 4
__global__ __launch_bounds__(1024) void slice_one_k(const int32_t* __restrict__ mult, const int32_t* __restrict__ src,
                                                    const int32_t* __restrict__ dst, int e_host, const int32_t* d_e,
                                                    int out_cap, int32_t* __restrict__ out_src, int32_t* __restrict__ out_dst,
                                                    int32_t* d_out_count, int32_t* status, unsigned long long* __restrict__ sync) {
    __shared__ int lds[17];
    __shared__ unsigned long long lds64;
    const int e = eff_count(d_e, e_host);
    if (e == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && d_out_count) *d_out_count = 0;
        return;
    }
    const int per = (int)blockDim.x * SLICE_IPT;
    const int live = (e + per - 1) / per;
    if ((int)blockIdx.x >= live) return;
    const int t0 = ((int)blockIdx.x * (int)blockDim.x + (int)threadIdx.x) * SLICE_IPT;
    int c[SLICE_IPT], sv[SLICE_IPT], dv[SLICE_IPT], mine = 0;
#pragma unroll
    for (int j = 0; j < SLICE_IPT; ++j) {
        const int t = t0 + j;
        c[j] = 0; sv[j] = 0; dv[j] = 0;
        if (t < e) { dv[j] = dst[t]; sv[j] = src[t]; }
    }
#pragma unroll
    for (int j = 0; j < SLICE_IPT; ++j) { if (t0 + j < e) c[j] = mult[dv[j]]; mine += c[j]; }
    int tot;
    int pos = block_excl_scan(mine, lds, &tot);
    const int base = (int)lookback_exclusive(sync, blockIdx.x, (unsigned long long)(unsigned)tot, &lds64, status);
    lookback_finish(sync, live);
    pos += base;
    bool overflow = false;
#pragma unroll
    for (int j = 0; j < SLICE_IPT; ++j) {
        for (int r = 0; r < c[j]; ++r, ++pos) {
            if (pos < out_cap) { out_src[pos] = sv[j]; out_dst[pos] = dv[j]; }
            else overflow = true;
        }
    }
    if ((int)blockIdx.x == live - 1 && threadIdx.x == 0 && d_out_count) {
        const int total = base + tot;
        *d_out_count = total < out_cap ? total : out_cap;
    }
    if (overflow && status) atomicOr(status, GRAPES_STATUS_EDGE_OVERFLOW);
}

extern "C" int grapes_slice_mark(int32_t* mult, const int32_t* cols, int32_t c, const int32_t* d_c, int32_t unmark,
                                 uint64_t* clear_bits, grapes_stream_t stream) {
    if (!mult || (!cols && c > 0) || c < 0) return GRAPES_EINVAL;
    if (c == 0) return 0;
    int grid = grapes_div_up(c, 256); if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(slice_mark_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, mult, cols, c, d_c, unmark,
                       (unsigned long long*)clear_bits);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

extern "C" int grapes_slice_remark(int32_t* mult, const int32_t* unmark_ids, int32_t n_unmark, const int32_t* d_n_unmark,
                                   const int32_t* mark_ids, int32_t n_mark, const int32_t* d_n_mark, uint64_t* clear_bits,
                                   const int32_t* clear_ids, int32_t n_clear, const int32_t* d_n_clear,
                                   grapes_stream_t stream) {
    if (!mult || n_unmark < 0 || n_mark < 0 || n_clear < 0) return GRAPES_EINVAL;
    if ((n_unmark > 0 && !unmark_ids) || (n_mark > 0 && !mark_ids) || (n_clear > 0 && (!clear_ids || !clear_bits)))
        return GRAPES_EINVAL;
    int nmax = n_unmark; if (n_mark > nmax) nmax = n_mark; if (n_clear > nmax) nmax = n_clear;
    if (nmax == 0) return 0;
    int grid = grapes_div_up(nmax, 256); if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(slice_remark_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, mult, n_unmark > 0 ? unmark_ids : nullptr,
                       n_unmark, d_n_unmark, n_mark > 0 ? mark_ids : nullptr, n_mark, d_n_mark,
                       (unsigned long long*)clear_bits, n_clear > 0 ? clear_ids : nullptr, n_clear, d_n_clear);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t grapes_slice_filter_workspace_bytes(int32_t e_cap) {
    return (size_t)(grapes_div_up(e_cap > 0 ? e_cap : 1, 1024) + 1) * sizeof(int32_t);
}

extern "C" int grapes_slice_filter(const int32_t* mult, const int32_t* src, const int32_t* dst, int32_t e,
                                   const int32_t* d_e, int32_t out_cap, int32_t* out_src, int32_t* out_dst,
                                   int32_t* d_out_count, void* workspace, uint64_t* sync, int32_t counted, int32_t* status,
                                   grapes_stream_t stream) {
    if (!mult || e < 0 || out_cap < 0 || ((!src || !dst) && e > 0) || ((!out_src || !out_dst) && out_cap > 0))
        return GRAPES_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (e == 0) {
        if (d_out_count) { hipError_t er = grapes_zero_async(d_out_count, sizeof(int32_t), s); if (er) return (int)er; }
        return 0;
    }
    if (!counted && sync && grapes_div_up(e, 1024 * SLICE_IPT) <= GRAPES_SYNC_SLOTS) {
        hipLaunchKernelGGL(slice_one_k, dim3(grapes_div_up(e, 1024 * SLICE_IPT)), dim3(1024), 0, s, mult, src, dst, e, d_e, out_cap,
                           out_src, out_dst, d_out_count, status, (unsigned long long*)sync);
        GRAPES_LAUNCH_CHECK();
        return 0;
    }
    if (!workspace) return GRAPES_EINVAL;
    const int G = grapes_div_up(e, 1024);
    int32_t* bsum = (int32_t*)workspace;
    if (!counted) {     // (counted: the expansion that produced src / dst has left the per-block survivor counts in `workspace`)
        hipLaunchKernelGGL(slice_count_k, dim3(G), dim3(1024), 0, s, mult, dst, e, d_e, bsum);
        GRAPES_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(slice_emit_k, dim3(G), dim3(1024), 0, s, mult, src, dst, e, d_e, (const int32_t*)bsum, out_cap, out_src,
                       out_dst, d_out_count, status);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------- indicators + feature gather
// advance != 0 (one workgroup): the marks carry epoch *d_epoch + 1, which the kernel then stores back — the start of a new
// step without a launch of its own for the counter
__global__ void indicator_mark_k(uint32_t* __restrict__ code, const int32_t* __restrict__ ids, int n_host,
                                 const int32_t* d_n, uint32_t epoch_host, uint32_t* d_epoch, int bit, int advance) {
    const int n = eff_count(d_n, n_host);
    uint32_t epoch = d_epoch ? (*d_epoch & 0xffffffu) : epoch_host;
    if (advance) {
        epoch = (epoch + 1u) & 0xffffffu;
        __syncthreads();                              // every wavefront has read the old value
        if (threadIdx.x == 0) *d_epoch = epoch;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int id = ids[i];
        uint32_t c = code[id];
        if ((c >> 8) != epoch) c = epoch << 8;
        code[id] = c | (1u << bit);
    }
}

template <bool VLOAD, bool VSTORE>
__global__ __launch_bounds__(256) void gather_rows_k(const float* __restrict__ X, int F,
                                                     const int32_t* __restrict__ ids, int n_host,
                                                     const int32_t* d_n, const uint32_t* __restrict__ code,
                                                     uint32_t epoch_host, const uint32_t* d_epoch, int num_ind,
                                                     float* __restrict__ out) {
    const int n = eff_count(d_n, n_host);
    const uint32_t epoch = d_epoch ? (*d_epoch & 0xffffffu) : epoch_host;
    const int Fo = F + num_ind;
    const int chunks = VLOAD ? (F >> 2) : F;        // items of the feature part per row
    const int ipr = chunks + (num_ind > 0 ? 1 : 0);  // + one item that writes the indicators
    const long long total = (long long)n * ipr;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < total;
         it += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(it / ipr);
        const int c = (int)(it - (long long)row * ipr);
        const int id = ids[row];
        float* o = out + (long long)row * Fo;
        if (c < chunks) {
            if (VLOAD) {
                const float4 v = *reinterpret_cast<const float4*>(X + (long long)id * F + c * 4);
                if (VSTORE) {
                    *reinterpret_cast<float4*>(o + c * 4) = v;
                } else {
                    o[c * 4 + 0] = v.x; o[c * 4 + 1] = v.y; o[c * 4 + 2] = v.z; o[c * 4 + 3] = v.w;
                }
            } else {
                o[c] = X[(long long)id * F + c];
            }
        } else {
            uint32_t cd = code[id];
            if ((cd >> 8) != epoch) cd = 0;
            for (int j = 0; j < num_ind; ++j) o[F + j] = ((cd >> j) & 1u) ? 1.0f : 0.0f;
        }
    }
}


extern "C" int grapes_step_begin(uint32_t* ind_code, uint32_t* d_epoch, int32_t bit, const int32_t* ids, int32_t n_ids,
                                 int32_t* d_cursor, int32_t stride, int32_t offset, int32_t batch, int32_t* targets,
                                 const int32_t* counters, int32_t counter_stride, int32_t n_counters, int64_t* totals,
                                 grapes_stream_t stream) {
    if (!ids || !d_cursor || !targets || batch <= 0 || n_ids < batch || stride <= 0 || offset < 0) return GRAPES_EINVAL;
    if (ind_code && (!d_epoch || bit < 0 || bit > 7)) return GRAPES_EINVAL;
    if (totals && (!counters || counter_stride <= 0 || n_counters < 0)) return GRAPES_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const StepBeginArgs BA{ind_code, d_epoch, bit, ids, n_ids, d_cursor, stride, offset, batch, targets, counters, counter_stride, n_counters,
                           (long long*)totals};
    auto single = [=](hipStream_t s_) {
        hipLaunchKernelGGL(step_begin_k, dim3(1), dim3(1024), 0, s_, BA.ind_code, BA.d_epoch, BA.bit, BA.ids, BA.n_ids, BA.d_cursor, BA.stride,
                           BA.offset, BA.B, BA.targets, BA.ctr, BA.ctr_stride, BA.n_ctr, BA.totals);
    };
    if (grapes_rider_recording()) { grapes_rider_record(grapes_rider_make(GRAPES_RK_BEGIN, 0, 1, 1024, BA, single)); return 0; }
    single(s);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

extern "C" int grapes_indicator_mark(uint32_t* ind_code, const int32_t* ids, int32_t n, const int32_t* d_n,
                                     uint32_t epoch, uint32_t* d_epoch, int32_t bit, int32_t advance_epoch,
                                     grapes_stream_t stream) {
    if (!ind_code || (!ids && n > 0) || n < 0 || bit < 0 || bit > 7 || epoch >= (1u << 24)) return GRAPES_EINVAL;
    if (advance_epoch && !d_epoch) return GRAPES_EINVAL;
    if (n == 0 && !advance_epoch) return 0;
    int grid = grapes_div_up(n, 256); if (grid > 1024) grid = 1024;
    if (advance_epoch || grid < 1) grid = 1;          // the workgroup that advances the counter is the only reader
    hipLaunchKernelGGL(indicator_mark_k, dim3(grid), dim3(advance_epoch ? 1024 : 256), 0, (hipStream_t)stream, ind_code, ids,
                       n, d_n, epoch, d_epoch, bit, advance_epoch ? 1 : 0);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

extern "C" int grapes_gather_rows(const float* X, int32_t F, const int32_t* ids, int32_t n, const int32_t* d_n,
                                  const uint32_t* ind_code, uint32_t epoch, const uint32_t* d_epoch,
                                  int32_t num_ind, float* out, grapes_stream_t stream) {
    if (!X || F <= 0 || n < 0 || num_ind < 0 || num_ind > 8 || (num_ind > 0 && !ind_code)) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!ids || !out) return GRAPES_EINVAL;
    const bool vload = (F % 4 == 0) && (((uintptr_t)X & 15) == 0);
    const bool vstore = vload && ((F + num_ind) % 4 == 0) && (((uintptr_t)out & 15) == 0);
    const int ipr = (vload ? F / 4 : F) + (num_ind > 0 ? 1 : 0);
    int grid = grapes_div_up((int64_t)n * ipr, 256); if (grid > 8192) grid = 8192;
    hipStream_t s = (hipStream_t)stream;
    if (vstore)
        hipLaunchKernelGGL((gather_rows_k<true, true>), dim3(grid), dim3(256), 0, s, X, F, ids, n, d_n, ind_code, epoch, d_epoch, num_ind, out);
    else if (vload)
        hipLaunchKernelGGL((gather_rows_k<true, false>), dim3(grid), dim3(256), 0, s, X, F, ids, n, d_n, ind_code, epoch, d_epoch, num_ind, out);
    else
        hipLaunchKernelGGL((gather_rows_k<false, false>), dim3(grid), dim3(256), 0, s, X, F, ids, n, d_n, ind_code, epoch, d_epoch, num_ind, out);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// Backward of the feature-row gather when the features are LEARNED (--embed_nodes, main.py:89-100: data.x is an nn.Parameter
// and autograd's index backward accumulates d x[ids] into a dense N x F gradient): dst[ids[i], 0:F] (+)= src[i, 0:F].  The id
// lists of the step (batch_nodes / all_nodes) are duplicate-free, so rows are written by plain stores; with `atomic` != 0
// (arbitrary id lists: the drop-in autograd function) float atomics are used and the result depends on the order only in the
// last bits, as torch's own index_add_ does.
__global__ __launch_bounds__(256) void scatter_rows_k(float* __restrict__ dst, long long ld_dst, const int32_t* __restrict__ ids,
                                                     const float* __restrict__ src, long long ld_src, int F, int n_host,
                                                     const int32_t* d_n, int accumulate, int atomic) {
    const int n = eff_count(d_n, n_host);
    const long long total = (long long)n * F;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(it / F), c = (int)(it - (long long)row * F);
        float* d = dst + (long long)ids[row] * ld_dst + c;
        const float v = src[(long long)row * ld_src + c];
        if (atomic) atomicAdd(d, v);
        else *d = accumulate ? *d + v : v;
    }
}
extern "C" int grapes_scatter_rows(float* dst, int64_t dst_stride, const int32_t* ids, const float* src, int64_t src_stride,
                                   int32_t F, int32_t n, const int32_t* d_n, int32_t accumulate, int32_t atomic,
                                   grapes_stream_t stream) {
    if (n < 0 || F <= 0 || dst_stride < F || src_stride < F) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!dst || !ids || !src) return GRAPES_EINVAL;
    int grid = grapes_div_up((int64_t)n * F, 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(scatter_rows_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, dst, (long long)dst_stride, ids, src,
                       (long long)src_stride, F, n, d_n, accumulate, atomic);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------- misc
extern "C" int grapes_abi_version(void) { return GRAPES_ABI_VERSION; }
extern "C" const char* grapes_build_flavor(void) {
#ifdef GRAPES_DIAG
    return "diag";
#else
    return "product";
#endif
}
extern "C" const char* grapes_target_arch(void) { return "gfx950"; }
