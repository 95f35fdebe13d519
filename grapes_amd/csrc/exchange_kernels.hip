// Kernels either side of the RCCL exchanges of the 1-D node-partitioned graph (the reference is
// single-device, main.py:180,199-204 are the calls being distributed).  Everything is sized by capacities that
// are equal on every rank, with the live counts on the device, so a hop never reads a size on the host:
//
//   adjacency rows   every rank all-gathers its query list  q = [cap ids | count];  the owner of an id serves
//                    the row into the requester's fixed reply slot
//                        reply[p] = [ len[cap] | off[cap] | columns[e_slot] ]      (len = 0 for ids it does not own)
//                    one all-to-all returns the slots; the requester reads its rows back in query order.
//   halo features    every rank all-gathers its ASCENDING id list; an owner's ids form one contiguous run of
//                    each list, which it gathers into reply[p] = rows[n_slot][F]; one all-to-all returns them.
// Integer / byte work, HBM- and latency-bound: no MFMA.
#include "common.h"
#include <string.h>

#define XCH_MAX_PEERS 64

// ---------------------------------------------------------------------------- owner side: adjacency rows
// One workgroup: lengths of the owned rows of all peers' queries + exclusive scan (chunks of 1024 with carry);
// writes the reply headers.  eoff has n_peers*cap + 1 entries (compact edge order: peer, then query).
__global__ __launch_bounds__(1024) void serve_offsets_k(const int64_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ req, int n_peers, int cap,
                                                        int lo, int hi, int32_t* __restrict__ reply,
                                                        long long stride, int32_t* __restrict__ eoff) {
    __shared__ int lds[17];
    const int M = n_peers * cap;
    long long carry = 0;
    for (int base = 0; base < M; base += blockDim.x) {
        const int i = base + threadIdx.x;
        int len = 0, p = 0, j = 0;
        if (i < M) {
            p = i / cap; j = i - p * cap;
            const int32_t* q = req + (long long)p * (cap + 1);
            int m = q[cap]; m = m < 0 ? 0 : (m > cap ? cap : m);
            const int id = q[j];
            if (j < m && id >= lo && id < hi) len = (int)(rowptr[id - lo + 1] - rowptr[id - lo]);
        }
        int tot;
        const int ex = block_excl_scan(len, lds, &tot);
        long long o = carry + ex;
        const int oi = o > 0x7fffffffLL ? 0x7fffffff : (int)o;
        if (i < M) eoff[i] = oi;
        carry += tot;
    }
    if (threadIdx.x == 0) eoff[M] = carry > 0x7fffffffLL ? 0x7fffffff : (int)carry;
    __threadfence_block();
    __syncthreads();
    // headers: len and offset inside the peer's slot (second sweep; eoff is complete and visible to the block)
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        const int p = i / cap, j = i - p * cap;
        int32_t* h = reply + (long long)p * stride;
        h[j] = eoff[i + 1] - eoff[i];
        h[cap + j] = eoff[i] - eoff[p * cap];
    }
}

__global__ __launch_bounds__(256) void serve_expand_k(const int64_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ col,
                                                      const int32_t* __restrict__ req, int n_peers, int cap, int lo,
                                                      const int32_t* __restrict__ eoff, int32_t* __restrict__ reply,
                                                      long long stride, int e_slot, int32_t* status) {
    const int M = n_peers * cap;
    const int e = eoff[M];
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < e; t += gridDim.x * blockDim.x) {
        int a = 0, b = M;            // invariant: eoff[a] <= t < eoff[b]
        while (b - a > 1) {
            const int mid = (a + b) >> 1;
            if (eoff[mid] <= t) a = mid; else b = mid;
        }
        const int p = a / cap, j = a - p * cap;
        const int in_slot = t - eoff[p * cap];
        if (in_slot >= e_slot) {
            if (status) atomicOr(status, GRAPES_STATUS_EDGE_OVERFLOW);
            continue;
        }
        const int id = req[(long long)p * (cap + 1) + j];
        reply[(long long)p * stride + 2 * cap + in_slot] = col[rowptr[id - lo] + (t - eoff[a])];
    }
}

// ---------------------------------------------------------------------------- requester side: adjacency rows
// One workgroup: this rank's queries in order; the row of query i sits in the reply of its owner.
__global__ __launch_bounds__(1024) void recv_offsets_k(const int32_t* __restrict__ back, long long stride,
                                                       const int32_t* __restrict__ nodes, int cap,
                                                       const int32_t* d_m, const int32_t* __restrict__ bounds,
                                                       int n_peers, int32_t* __restrict__ eoff,
                                                       int32_t* __restrict__ rowstart, int32_t* d_e) {
    __shared__ int lds[17];
    const int m = eff_count(d_m, cap);
    long long carry = 0;
    for (int base = 0; base < m; base += blockDim.x) {
        const int i = base + threadIdx.x;
        int len = 0;
        if (i < m) {
            const int id = nodes[i];
            int p = 0;
            while (p + 1 < n_peers && id >= bounds[p + 1]) ++p;
            const int32_t* h = back + (long long)p * stride;
            len = h[i];
            rowstart[i] = (int)((long long)p * stride + 2 * cap + h[cap + i]);
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
        if (d_e) *d_e = e;
    }
}

__global__ __launch_bounds__(256) void recv_expand_k(const int32_t* __restrict__ back,
                                                     const int32_t* __restrict__ nodes, int cap, const int32_t* d_m,
                                                     const int32_t* __restrict__ eoff,
                                                     const int32_t* __restrict__ rowstart, int e_cap,
                                                     int32_t* __restrict__ src, int32_t* __restrict__ dst,
                                                     int32_t* status) {
    const int m = eff_count(d_m, cap);
    const int e_true = eoff[m];
    const int e = e_true < e_cap ? e_true : e_cap;
    if (blockIdx.x == 0 && threadIdx.x == 0 && e_true > e_cap && status) atomicOr(status, GRAPES_STATUS_EDGE_OVERFLOW);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < e; t += gridDim.x * blockDim.x) {
        int a = 0, b = m;
        while (b - a > 1) {
            const int mid = (a + b) >> 1;
            if (eoff[mid] <= t) a = mid; else b = mid;
        }
        src[t] = nodes[a];
        dst[t] = back[rowstart[a] + (t - eoff[a])];
    }
}

// ---------------------------------------------------------------------------- halo feature rows
__device__ __forceinline__ int lower_bound_i32(const int32_t* __restrict__ v, int n, int key) {
    int a = 0, b = n;                // first index with v[idx] >= key
    while (a < b) {
        const int mid = (a + b) >> 1;
        if (v[mid] < key) a = mid + 1; else b = mid;
    }
    return a;
}

// grid (x, n_peers): the run of peer blockIdx.y's ascending list that this rank owns, gathered into its slot.
template <bool VEC>
__global__ __launch_bounds__(256) void serve_rows_k(const float* __restrict__ X, int F,
                                                    const int32_t* __restrict__ req, int cap, int lo, int hi,
                                                    float* __restrict__ reply, int n_slot, int32_t* status) {
    __shared__ int s_a, s_cnt;
    const int p = blockIdx.y;
    const int32_t* q = req + (long long)p * (cap + 1);
    if (threadIdx.x == 0) {
        int m = q[cap]; m = m < 0 ? 0 : (m > cap ? cap : m);
        const int a = lower_bound_i32(q, m, lo), b = lower_bound_i32(q, m, hi);
        int cnt = b - a;
        if (cnt > n_slot) {
            if (status && blockIdx.x == 0) atomicOr(status, GRAPES_STATUS_NODE_OVERFLOW);
            cnt = n_slot;
        }
        s_a = a; s_cnt = cnt;
    }
    __syncthreads();
    const int a = s_a, cnt = s_cnt;
    const int ipr = VEC ? (F >> 2) : F;
    const long long total = (long long)cnt * ipr;
    float* out = reply + (long long)p * n_slot * F;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < total;
         it += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(it / ipr);
        const int c = (int)(it - (long long)row * ipr);
        const int id = q[a + row];
        if (id < lo || id >= hi) continue;                     // a list that is not ascending: nothing is read
        if (VEC)
            *reinterpret_cast<float4*>(out + (long long)row * F + c * 4) =
                *reinterpret_cast<const float4*>(X + (long long)(id - lo) * F + c * 4);
        else
            out[(long long)row * F + c] = X[(long long)(id - lo) * F + c];
    }
}

// out[i, 0:F] = the halo row of ids[i] (slot of its owner, position = rank inside the owner's run),
// out[i, F+j] = indicator j of ids[i]  (same packing as gather_rows_k).
template <bool VLOAD, bool VSTORE>
__global__ __launch_bounds__(256) void halo_assemble_k(const float* __restrict__ back, int F, int n_slot,
                                                       const int32_t* __restrict__ ids, int n_host,
                                                       const int32_t* d_n, const int32_t* __restrict__ bounds,
                                                       int n_peers, const uint32_t* __restrict__ code,
                                                       uint32_t epoch_host, const uint32_t* d_epoch, int num_ind,
                                                       float* __restrict__ out) {
    __shared__ int s_cut[XCH_MAX_PEERS + 1], s_bnd[XCH_MAX_PEERS + 1];
    const int n = eff_count(d_n, n_host);
    if ((int)threadIdx.x <= n_peers) {
        s_bnd[threadIdx.x] = bounds[threadIdx.x];
        s_cut[threadIdx.x] = lower_bound_i32(ids, n, bounds[threadIdx.x]);
    }
    __syncthreads();
    const uint32_t epoch = d_epoch ? (*d_epoch & 0xffffffu) : epoch_host;
    const int Fo = F + num_ind;
    const int chunks = VLOAD ? (F >> 2) : F;
    const int ipr = chunks + (num_ind > 0 ? 1 : 0);
    const long long total = (long long)n * ipr;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < total;
         it += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(it / ipr);
        const int c = (int)(it - (long long)row * ipr);
        float* o = out + (long long)row * Fo;
        if (c < chunks) {
            int p = 0;
            while (p + 1 < n_peers && row >= s_cut[p + 1]) ++p;
            int r = row - s_cut[p];
            if (r >= n_slot) r = n_slot - 1;                   // the owner raised the overflow status
            const float* x = back + ((long long)p * n_slot + r) * F;
            if (VLOAD) {
                const float4 v = *reinterpret_cast<const float4*>(x + c * 4);
                if (VSTORE) {
                    *reinterpret_cast<float4*>(o + c * 4) = v;
                } else {
                    o[c * 4 + 0] = v.x; o[c * 4 + 1] = v.y; o[c * 4 + 2] = v.z; o[c * 4 + 3] = v.w;
                }
            } else {
                o[c] = x[c];
            }
        } else {
            uint32_t cd = code[ids[row]];
            if ((cd >> 8) != epoch) cd = 0;
            for (int j = 0; j < num_ind; ++j) o[F + j] = ((cd >> j) & 1u) ? 1.0f : 0.0f;
        }
    }
}

// One rank's query message [cap ids | live count] from an id list with a device-side count (one launch; the two framework
// kernels it replaces ran 8 times per step).
__global__ void pack_query_k(const int32_t* __restrict__ ids, int n, const int32_t* d_n, int cap, int32_t* __restrict__ q) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) q[i] = ids[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int live = n;
        if (d_n) { const int v = *d_n; live = v < n ? v : n; }
        q[cap] = live;
    }
}

// ---------------------------------------------------------------------------- C-ABI
extern "C" int grapes_exchange_pack_query(const int32_t* ids, int32_t n, const int32_t* d_n, int32_t cap, int32_t* query,
                                          grapes_stream_t stream) {
    if (!query || n < 0 || cap < n || (!ids && n > 0)) return GRAPES_EINVAL;
    int grid = grapes_div_up(n > 0 ? n : 1, 256); if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(pack_query_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, ids, n, d_n, cap, query);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

extern "C" int grapes_exchange_serve_rows(const int64_t* rowptr_local, const int32_t* col_local,
                                          const int32_t* req, int32_t n_peers, int32_t cap, int32_t lo, int32_t hi,
                                          int32_t* reply, int64_t reply_stride, int32_t e_slot, int32_t* eoff,
                                          int32_t* status, grapes_stream_t stream) {
    if (!rowptr_local || !col_local || !req || !reply || !eoff) return GRAPES_EINVAL;
    if (n_peers <= 0 || n_peers > XCH_MAX_PEERS || cap <= 0 || e_slot < 0 || lo < 0 || hi < lo) return GRAPES_EINVAL;
    if (reply_stride < 2LL * cap + e_slot) return GRAPES_EINVAL;
    if ((int64_t)n_peers * cap >= 0x7fffffffLL) return GRAPES_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(serve_offsets_k, dim3(1), dim3(1024), 0, s, rowptr_local, req, n_peers, cap, lo, hi, reply,
                       (long long)reply_stride, eoff);
    GRAPES_LAUNCH_CHECK();
    if (e_slot > 0) {
        int grid = grapes_div_up((int64_t)n_peers * e_slot, 256); if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(serve_expand_k, dim3(grid), dim3(256), 0, s, rowptr_local, col_local, req, n_peers, cap, lo,
                           eoff, reply, (long long)reply_stride, e_slot, status);
        GRAPES_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int grapes_exchange_recv_rows(const int32_t* back, int64_t reply_stride, const int32_t* nodes, int32_t cap,
                                         const int32_t* d_m, const int32_t* bounds, int32_t n_peers, int32_t e_cap,
                                         int32_t* eoff, int32_t* rowstart, int32_t* src, int32_t* dst, int32_t* d_e,
                                         int32_t* status, grapes_stream_t stream) {
    if (!back || !nodes || !bounds || !eoff || !rowstart || !src || !dst) return GRAPES_EINVAL;
    if (n_peers <= 0 || n_peers > XCH_MAX_PEERS || cap <= 0 || e_cap < 0) return GRAPES_EINVAL;
    if (reply_stride * n_peers >= 0x7fffffffLL) return GRAPES_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(recv_offsets_k, dim3(1), dim3(1024), 0, s, back, (long long)reply_stride, nodes, cap, d_m,
                       bounds, n_peers, eoff, rowstart, d_e);
    GRAPES_LAUNCH_CHECK();
    if (e_cap > 0) {
        int grid = grapes_div_up(e_cap, 256); if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(recv_expand_k, dim3(grid), dim3(256), 0, s, back, nodes, cap, d_m, eoff, rowstart, e_cap,
                           src, dst, status);
        GRAPES_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int grapes_exchange_serve_features(const float* X_local, int32_t F, const int32_t* req, int32_t n_peers,
                                              int32_t cap, int32_t lo, int32_t hi, float* reply, int32_t n_slot,
                                              int32_t* status, grapes_stream_t stream) {
    if (!X_local || !req || !reply || F <= 0 || cap <= 0 || n_slot <= 0 || lo < 0 || hi < lo) return GRAPES_EINVAL;
    if (n_peers <= 0 || n_peers > XCH_MAX_PEERS) return GRAPES_EINVAL;
    const bool vec = (F % 4 == 0) && (((uintptr_t)X_local & 15) == 0) && (((uintptr_t)reply & 15) == 0);
    const int ipr = vec ? F / 4 : F;
    int gx = grapes_div_up((int64_t)n_slot * ipr, 256); if (gx > 2048) gx = 2048;
    hipStream_t s = (hipStream_t)stream;
    if (vec)
        hipLaunchKernelGGL((serve_rows_k<true>), dim3(gx, n_peers), dim3(256), 0, s, X_local, F, req, cap, lo, hi, reply, n_slot, status);
    else
        hipLaunchKernelGGL((serve_rows_k<false>), dim3(gx, n_peers), dim3(256), 0, s, X_local, F, req, cap, lo, hi, reply, n_slot, status);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// The requester's rows WHERE THEY ARRIVED: pos[i] = row of ids[i] inside back = fp32[n_peers * n_slot][F] (slot of its owner,
// position = rank inside the owner's run) — the "feature matrix" of the fused gather-SpMM, which then reads the exchanged rows in
// place (head records built with head_ids = pos) instead of from an assembled copy; code_pos[pos[i]] = the indicator word of
// ids[i] (the gather-SpMM indexes its code table by the same row id).  Needs only the id list, not the exchanged data.
__global__ __launch_bounds__(256) void halo_positions_k(const int32_t* __restrict__ ids, int n_host, const int32_t* d_n,
                                                        const int32_t* __restrict__ bounds, int n_peers, int n_slot,
                                                        const uint32_t* __restrict__ code, int32_t* __restrict__ pos,
                                                        uint32_t* __restrict__ code_pos) {
    __shared__ int s_cut[XCH_MAX_PEERS + 1];
    const int n = eff_count(d_n, n_host);
    if ((int)threadIdx.x <= n_peers) s_cut[threadIdx.x] = lower_bound_i32(ids, n, bounds[threadIdx.x]);
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_host; i += gridDim.x * blockDim.x) {
        int q = 0;
        if (i < n) {
            int p = 0;
            while (p + 1 < n_peers && i >= s_cut[p + 1]) ++p;
            int r = i - s_cut[p];
            if (r >= n_slot) r = n_slot - 1;                   // the owner raised the overflow status
            q = p * n_slot + r;
            if (code_pos) code_pos[q] = code[ids[i]];
        }
        pos[i] = q;
    }
}

extern "C" int grapes_exchange_halo_positions(const int32_t* ids, int32_t n, const int32_t* d_n, const int32_t* bounds,
                                              int32_t n_peers, int32_t n_slot, const uint32_t* ind_code, int32_t* pos,
                                              uint32_t* code_pos, grapes_stream_t stream) {
    if (n < 0 || n_slot <= 0 || !bounds || n_peers <= 0 || n_peers > XCH_MAX_PEERS || ((ind_code == nullptr) != (code_pos == nullptr)))
        return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!ids || !pos) return GRAPES_EINVAL;
    int grid = grapes_div_up(n, 256); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(halo_positions_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, ids, n, d_n, bounds, n_peers, n_slot, ind_code,
                       pos, code_pos);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// loc[ids[i]] = base + pos[row(i)],  row(i) = node_map[ids[i]]  or  idx_b[idx_a[i]]  (see include/grapes_hip.h).  In the node_map
// form an id that is NOT a row of the batch (its map entry is stale: batch[row] != id) is left alone — an isolated target keeps its
// static location.
__global__ __launch_bounds__(256) void exchange_note_rows_k(int32_t* __restrict__ loc, const int32_t* __restrict__ ids, int n_host,
                                                            const int32_t* d_n, const int32_t* __restrict__ node_map,
                                                            const int32_t* __restrict__ batch, int nb_host, const int32_t* d_nb,
                                                            const int32_t* __restrict__ idx_a, const int32_t* __restrict__ idx_b,
                                                            const int32_t* __restrict__ pos, int base) {
    const int n = eff_count(d_n, n_host);
    const int nb = node_map ? eff_count(d_nb, nb_host) : 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int id = ids[i];
        int row;
        if (node_map) {
            row = node_map[id];
            if ((unsigned)row >= (unsigned)nb || batch[row] != id) continue;
        } else {
            row = idx_b[idx_a[i]];
        }
        loc[id] = base + pos[row];
    }
}
extern "C" int grapes_exchange_note_rows(int32_t* loc, const int32_t* ids, int32_t n, const int32_t* d_n, const int32_t* node_map,
                                         const int32_t* batch, int32_t n_batch, const int32_t* d_n_batch,
                                         const int32_t* idx_a, const int32_t* idx_b, const int32_t* pos, int32_t base,
                                         grapes_stream_t stream) {
    if (n < 0 || base < 0) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!loc || !ids || !pos || (node_map && (!batch || n_batch < 0)) || (!node_map && (!idx_a || !idx_b))) return GRAPES_EINVAL;
    int grid = grapes_div_up(n, 256); if (grid > 256) grid = 256;
    hipLaunchKernelGGL(exchange_note_rows_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, loc, ids, n, d_n, node_map, batch, n_batch, d_n_batch,
                       idx_a, idx_b, pos, base);
    GRAPES_LAUNCH_CHECK();
    return 0;
}
extern "C" int grapes_exchange_assemble_features(const float* back, int32_t F, int32_t n_slot, const int32_t* ids,
                                                 int32_t n, const int32_t* d_n, const int32_t* bounds,
                                                 int32_t n_peers, const uint32_t* ind_code, uint32_t epoch,
                                                 const uint32_t* d_epoch, int32_t num_ind, float* out,
                                                 grapes_stream_t stream) {
    if (!back || F <= 0 || n_slot <= 0 || n < 0 || !bounds || num_ind < 0 || num_ind > 8 || (num_ind > 0 && !ind_code))
        return GRAPES_EINVAL;
    if (n_peers <= 0 || n_peers > XCH_MAX_PEERS) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!ids || !out) return GRAPES_EINVAL;
    const bool vload = (F % 4 == 0) && (((uintptr_t)back & 15) == 0);
    const bool vstore = vload && ((F + num_ind) % 4 == 0) && (((uintptr_t)out & 15) == 0);
    const int ipr = (vload ? F / 4 : F) + (num_ind > 0 ? 1 : 0);
    int grid = grapes_div_up((int64_t)n * ipr, 256); if (grid > 8192) grid = 8192;
    hipStream_t s = (hipStream_t)stream;
    if (vstore)
        hipLaunchKernelGGL((halo_assemble_k<true, true>), dim3(grid), dim3(256), 0, s, back, F, n_slot, ids, n, d_n, bounds, n_peers, ind_code, epoch, d_epoch, num_ind, out);
    else if (vload)
        hipLaunchKernelGGL((halo_assemble_k<true, false>), dim3(grid), dim3(256), 0, s, back, F, n_slot, ids, n, d_n, bounds, n_peers, ind_code, epoch, d_epoch, num_ind, out);
    else
        hipLaunchKernelGGL((halo_assemble_k<false, false>), dim3(grid), dim3(256), 0, s, back, F, n_slot, ids, n, d_n, bounds, n_peers, ind_code, epoch, d_epoch, num_ind, out);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------- peer shards mapped in place (no exchange)
// Host side only: the mapping is made once per run; afterwards a peer's rows are ordinary global loads inside
// gcn_aggregate_gather_head5_k<.., PEER> (spmm_kernels.hip).  The handle names the ALLOCATION `ptr` lies in (a tensor of a
// caching allocator sits somewhere inside a larger block), so the byte offset travels with it.
extern "C" int grapes_peer_export(const void* ptr, void* handle, uint64_t* offset) {
    if (!ptr || !handle || !offset) return GRAPES_EINVAL;
    hipDeviceptr_t base = nullptr; size_t size = 0;
    hipError_t e = hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    hipIpcMemHandle_t h;
    static_assert(sizeof(hipIpcMemHandle_t) == GRAPES_PEER_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
    e = hipIpcGetMemHandle(&h, base);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    memcpy(handle, &h, sizeof(h));
    *offset = (uint64_t)((const char*)ptr - (const char*)base);
    return 0;
}
extern "C" int grapes_peer_open(const void* handle, uint64_t offset, void** ptr) {
    if (!handle || !ptr) return GRAPES_EINVAL;
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void* base = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    *ptr = (char*)base + offset;
    return 0;
}
extern "C" int grapes_peer_close(void* ptr, uint64_t offset) {
    if (!ptr) return GRAPES_EINVAL;
    const hipError_t e = hipIpcCloseMemHandle((char*)ptr - offset);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    return 0;
}
extern "C" int grapes_peer_copy(void* dst, const void* src, size_t bytes, grapes_stream_t stream) {
    if (!dst || !src) return GRAPES_EINVAL;
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    return 0;
}
