// N3 (ingest): edge_index -> device CSR with the semantics of the SciPy constructor the reference calls once per run,
//   adjacency = sp.csr_matrix((ones(E, bool), edge_index), (N, N))        (main.py:134-136)
// i.e. duplicate (row, col) pairs collapse to one entry, columns ascend inside a row, self-loops stay.
//
// Integer / byte work, HBM-bound: no sort of 64-bit keys over the whole edge list.  Rows are found by COUNTING (one atomic per
// edge into a degree table, a scan, one atomic per edge to scatter the column into its row's segment), then every row segment is
// sorted and de-duplicated where it lies:
//   * short rows (<= CSR_WIN entries, all but a few hubs): a workgroup takes a window of CSR_WIN consecutive raw slots, i.e. a
//     run of WHOLE rows of up to 2*CSR_WIN entries, sorts the 64-bit keys (row in window << 32 | column) with one bitonic
//     network in LDS — every row of the run at once —, marks first occurrences, scans them and writes each row's unique columns
//     compacted at the row's own start;
//   * long rows (hubs): one workgroup per row sorts the segment in global memory with the same comparator network (virtual
//     +inf padding, so no scratch copy) and compacts it in place;
// then a scan of the unique counts gives the final row pointer and the rows are copied to their final place.
// Scatter order inside a row depends on atomic arrival, the sorted result does not: the output is deterministic.
// Sized for ogbn-papers100M (N = 1.1e8, 3.2e9 symmetrised edges): 64-bit offsets throughout, 12.8 GB raw + 12.8 GB final columns.
#include "common.h"

#define CSR_WIN 4096                 // raw slots per window; a window's rows hold at most 2 * CSR_WIN entries
#define CSR_CAP (2 * CSR_WIN)
#define SCAN_ITEMS 16
#define SCAN_BLOCK 256
#define SCAN_TILE (SCAN_ITEMS * SCAN_BLOCK)

// ---------------------------------------------------------------------------------------------- counting
__global__ __launch_bounds__(256) void csr_hist_k(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, long long E,
                                                  int N, unsigned* __restrict__ deg, int32_t* status) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    bool bad = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += stride) {
        const int64_t s = src[i], d = dst[i];
        if (s < 0 || s >= N || d < 0 || d >= N) { bad = true; continue; }
        atomicAdd(&deg[s], 1u);
    }
    if (bad && status) atomicOr(status, GRAPES_STATUS_BAD_INDEX);
}

__global__ __launch_bounds__(256) void csr_scatter_k(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, long long E,
                                                     int N, const int64_t* __restrict__ rowptr_raw, unsigned* __restrict__ cursor,
                                                     int32_t* __restrict__ col_raw) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += stride) {
        const int64_t s = src[i], d = dst[i];
        if (s < 0 || s >= N || d < 0 || d >= N) continue;
        const unsigned p = atomicAdd(&cursor[s], 1u);
        col_raw[rowptr_raw[s] + p] = (int32_t)d;
    }
}

// ---------------------------------------------------------------------------------------------- scan (uint32 -> int64, exclusive)
__global__ __launch_bounds__(SCAN_BLOCK) void scan_tile_sums_k(const unsigned* __restrict__ v, long long n, int64_t* __restrict__ tile_sum) {
    __shared__ long long red[SCAN_BLOCK / 64];
    const long long base = (long long)blockIdx.x * SCAN_TILE;
    long long acc = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        const long long i = base + (long long)j * SCAN_BLOCK + threadIdx.x;
        acc += i < n ? (long long)v[i] : 0;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { long long t = 0; for (int w = 0; w < SCAN_BLOCK / 64; ++w) t += red[w]; tile_sum[blockIdx.x] = t; }
}
// one workgroup: exclusive scan of the tile sums in place; total -> *total_out
__global__ __launch_bounds__(1024) void scan_tile_offsets_k(int64_t* __restrict__ tile_sum, long long ntiles, int64_t* __restrict__ total_out) {
    __shared__ long long part[1024];
    __shared__ long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (long long base = 0; base < ntiles; base += 1024) {
        const long long i = base + threadIdx.x;
        const long long x = i < ntiles ? tile_sum[i] : 0;
        part[threadIdx.x] = x;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {          // Hillis-Steele inclusive scan in LDS
            const long long t = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        const long long incl = part[threadIdx.x], c = carry;
        if (i < ntiles) tile_sum[i] = c + incl - x;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry;
}
__global__ __launch_bounds__(SCAN_BLOCK) void scan_apply_k(const unsigned* __restrict__ v, long long n, const int64_t* __restrict__ tile_off,
                                                           int64_t* __restrict__ out) {
    __shared__ long long wsum[SCAN_BLOCK / 64];
    __shared__ long long run;
    const long long base = (long long)blockIdx.x * SCAN_TILE;
    if (threadIdx.x == 0) run = tile_off[blockIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int j = 0; j < SCAN_ITEMS; ++j) {            // SCAN_BLOCK consecutive items per round
        const long long i = base + (long long)j * SCAN_BLOCK + threadIdx.x;
        const long long x = i < n ? (long long)v[i] : 0;
        long long incl = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const long long t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        long long woff = 0;
        for (int w = 0; w < wid; ++w) woff += wsum[w];
        const long long r0 = run;
        if (i < n) out[i] = r0 + woff + incl - x;
        __syncthreads();
        if (threadIdx.x == SCAN_BLOCK - 1) run = r0 + woff + incl;
        __syncthreads();
    }
}
static int scan_u32_to_i64(const unsigned* v, long long n, int64_t* out /* [n] exclusive */, int64_t* total /* device */,
                           int64_t* tile_ws, hipStream_t s) {
    const long long ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(scan_tile_sums_k, dim3((unsigned)ntiles), dim3(SCAN_BLOCK), 0, s, v, n, tile_ws);
    GRAPES_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_tile_offsets_k, dim3(1), dim3(1024), 0, s, tile_ws, ntiles, total);
    GRAPES_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_apply_k, dim3((unsigned)ntiles), dim3(SCAN_BLOCK), 0, s, v, n, (const int64_t*)tile_ws, out);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------- short rows: windows in LDS
// first index r in [0, n] with a[r] >= key (a ascending, a[n] = total)
__device__ __forceinline__ long long lower_bound_i64(const int64_t* __restrict__ a, long long n, long long key) {
    long long lo = 0, hi = n + 1;
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}
__global__ __launch_bounds__(512) void csr_window_sort_k(const int64_t* __restrict__ rowptr_raw /* [N+1] */, int N,
                                                         int32_t* __restrict__ col_raw, unsigned* __restrict__ udeg,
                                                         int32_t* __restrict__ long_rows, int32_t* __restrict__ n_long, int long_cap,
                                                         int32_t* status) {
    extern __shared__ unsigned long long csr_dyn[];   // 96 KB: keys + exclusive counts of first occurrences
    unsigned long long* key = csr_dyn;
    int* excl = reinterpret_cast<int*>(csr_dyn + CSR_CAP);
    __shared__ int wtot[8];
    __shared__ long long s_r0, s_r1;
    const long long total = rowptr_raw[N];
    for (long long b = blockIdx.x; b * CSR_WIN < total; b += gridDim.x) {
        if (threadIdx.x == 0) {
            s_r0 = lower_bound_i64(rowptr_raw, N, b * (long long)CSR_WIN);
            s_r1 = lower_bound_i64(rowptr_raw, N, (b + 1) * (long long)CSR_WIN);
        }
        __syncthreads();
        long long r0 = s_r0, r1 = s_r1;             // rows [r0, r1) start inside this window (r1 <= N)
        __syncthreads();
        if (r1 > N) r1 = N;
        if (r1 > r0 && rowptr_raw[r1] - rowptr_raw[r1 - 1] > CSR_WIN) {     // a hub can only be the LAST row that starts here
            if (threadIdx.x == 0) {
                const int slot = atomicAdd(n_long, 1);
                if (slot < long_cap) long_rows[slot] = (int32_t)(r1 - 1);
                else if (status) atomicOr(status, GRAPES_STATUS_NODE_OVERFLOW);
            }
            r1 -= 1;
        }
        if (r1 <= r0) continue;
        const long long base = rowptr_raw[r0];
        const int cnt = (int)(rowptr_raw[r1] - base);                    // <= 2 * CSR_WIN
        if (cnt == 0) continue;
        int P = 64; while (P < cnt) P <<= 1;
        // ---- keys: (row - r0) << 32 | column; the row of a raw slot by binary search over the run's row pointers
        for (int i = threadIdx.x; i < P; i += 512) {
            unsigned long long k = ~0ull;
            if (i < cnt) {
                long long lo = r0, hi = r1;                                // last row r with rowptr_raw[r] <= base + i
                const long long pos = base + i;
                while (hi - lo > 1) { const long long mid = (lo + hi) >> 1; if (rowptr_raw[mid] <= pos) lo = mid; else hi = mid; }
                k = ((unsigned long long)(lo - r0) << 32) | (unsigned)col_raw[pos];
            }
            key[i] = k;
        }
        __syncthreads();
        // ---- bitonic network, every comparator ascending (lower index keeps the smaller key)
        for (int k = 2; k <= P; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = threadIdx.x; t < (P >> 1); t += 512) {
                    int i, p;
                    if (j == (k >> 1)) { const int blk = t / j, off = t - blk * j; i = blk * k + off; p = blk * k + (k - 1 - off); }
                    else { const int blk = t / j, off = t - blk * j; i = blk * 2 * j + off; p = i + j; }
                    const unsigned long long a = key[i], c = key[p];
                    if (a > c) { key[i] = c; key[p] = a; }
                }
                __syncthreads();
            }
        }
        // ---- first occurrences and their exclusive scan (512 threads x ceil(cnt / 512) consecutive items)
        const int per = (cnt + 511) / 512;
        const int i0 = threadIdx.x * per;
        int local = 0;
        for (int q = 0; q < per; ++q) {
            const int i = i0 + q;
            if (i < cnt) { const int u = (i == 0 || key[i] != key[i - 1]) ? 1 : 0; excl[i] = local; local += u; }
        }
        int incl = local;
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        if (lane == 63) wtot[wid] = incl;
        __syncthreads();
        int off = incl - local;
        for (int w = 0; w < wid; ++w) off += wtot[w];
        for (int q = 0; q < per; ++q) { const int i = i0 + q; if (i < cnt) excl[i] += off; }
        __syncthreads();
        // ---- every row's unique columns compacted at the row's own start; unique degree by the row's last entry
        for (int i = threadIdx.x; i < cnt; i += 512) {
            const unsigned long long k = key[i];
            const long long row = r0 + (long long)(k >> 32);
            const int rs = (int)(rowptr_raw[row] - base), re = (int)(rowptr_raw[row + 1] - base);
            const bool first = (i == 0 || k != key[i - 1]);
            const int e_rs = excl[rs];
            if (first) col_raw[base + rs + (excl[i] - e_rs)] = (int32_t)(unsigned)(k & 0xffffffffull);
            if (i == re - 1) udeg[row] = (unsigned)(excl[i] + (first ? 1 : 0) - e_rs);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------- long rows: one workgroup each, in place
__global__ __launch_bounds__(1024) void csr_long_sort_k(const int64_t* __restrict__ rowptr_raw, int32_t* col_raw,
                                                        unsigned* __restrict__ udeg, const int32_t* __restrict__ long_rows,
                                                        const int32_t* __restrict__ n_long, int long_cap) {
    __shared__ int wtot[16];
    __shared__ long long s_out;
    int nl = *n_long; if (nl > long_cap) nl = long_cap;
    for (int li = blockIdx.x; li < nl; li += gridDim.x) {
        const long long row = long_rows[li];
        volatile int32_t* a = col_raw + rowptr_raw[row];                   // (volatile: this workgroup's own stores must be re-read)
        const long long len = rowptr_raw[row + 1] - rowptr_raw[row];
        long long P = 64; while (P < len) P <<= 1;
        for (long long k = 2; k <= P; k <<= 1) {
            for (long long j = k >> 1; j > 0; j >>= 1) {
                for (long long t = threadIdx.x; t < (P >> 1); t += 1024) {
                    long long i, p;
                    if (j == (k >> 1)) { const long long blk = t / j, off = t - blk * j; i = blk * k + off; p = blk * k + (k - 1 - off); }
                    else { const long long blk = t / j, off = t - blk * j; i = blk * 2 * j + off; p = i + j; }
                    if (p < len) {                                          // slots >= len are +inf: such a comparator is a no-op
                        const unsigned x = (unsigned)a[i], y = (unsigned)a[p];
                        if (x > y) { a[i] = (int32_t)y; a[p] = (int32_t)x; }
                    }
                }
                __threadfence_block();
                __syncthreads();
            }
        }
        // ---- in-place compaction of first occurrences, 1024 entries at a time (outputs land at or before what was read)
        if (threadIdx.x == 0) s_out = 0;
        __syncthreads();
        for (long long base = 0; base < len; base += 1024) {
            const long long i = base + threadIdx.x;
            int32_t v = 0; int u = 0;
            if (i < len) { v = a[i]; u = (i == 0 || a[i - 1] != v) ? 1 : 0; }
            int incl = u;
            const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
            if (lane == 63) wtot[wid] = incl;
            __syncthreads();                                               // (also: every read of this chunk is done)
            int off = incl - u;
            for (int w = 0; w < wid; ++w) off += wtot[w];
            const long long o0 = s_out;
            if (u) a[o0 + off] = v;
            __threadfence_block();
            __syncthreads();
            if (threadIdx.x == 1023) s_out = o0 + off + u;
            __syncthreads();
        }
        if (threadIdx.x == 0) udeg[row] = (unsigned)s_out;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------- final placement
__global__ __launch_bounds__(256) void csr_place_k(const int64_t* __restrict__ rowptr_raw, const int64_t* __restrict__ rowptr,
                                                   const unsigned* __restrict__ udeg, const int32_t* __restrict__ col_raw,
                                                   int32_t* __restrict__ col, int N) {
    const int lane = threadIdx.x & 63;
    const long long nw = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; row < N; row += nw) {
        const unsigned d = udeg[row];
        if (d == 0) continue;
        const int32_t* s = col_raw + rowptr_raw[row];
        int32_t* o = col + rowptr[row];
        for (unsigned j = lane; j < d; j += 64) o[j] = s[j];
    }
}

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
extern "C" size_t grapes_csr_build_workspace_bytes(int64_t num_edges, int32_t num_nodes) {
    const size_t E = (size_t)(num_edges > 0 ? num_edges : 1), N = (size_t)(num_nodes > 0 ? num_nodes : 1);
    const size_t ntiles = (N + SCAN_TILE - 1) / SCAN_TILE + 1;
    return al256(E * 4) /* col_raw */ + al256((N + 1) * 8) /* rowptr_raw */ + 2 * al256(N * 4) /* deg / cursor, udeg */ +
           al256(ntiles * 8) + al256((E / CSR_WIN + 2) * 4) /* long rows */ + 512;
}

/* rowptr int64[N+1], col int32[capacity num_edges]; *d_nnz receives the number of stored entries (= rowptr[N]). */
extern "C" int grapes_csr_build(const int64_t* edge_src, const int64_t* edge_dst, int64_t num_edges, int32_t num_nodes,
                                int64_t* rowptr, int32_t* col, int64_t* d_nnz, void* workspace, int32_t* status,
                                grapes_stream_t stream) {
    if (num_edges < 0 || num_nodes <= 0 || !rowptr || !d_nnz || !workspace) return GRAPES_EINVAL;
    if (num_edges > 0 && (!edge_src || !edge_dst || !col)) return GRAPES_EINVAL;
    if (((uintptr_t)workspace & 255) != 0) return GRAPES_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const size_t E = (size_t)(num_edges > 0 ? num_edges : 1), N = (size_t)num_nodes;
    char* w = (char*)workspace;
    int32_t* col_raw = (int32_t*)w; w += al256(E * 4);
    int64_t* rowptr_raw = (int64_t*)w; w += al256((N + 1) * 8);
    unsigned* deg = (unsigned*)w; w += al256(N * 4);           // degrees, then (re-zeroed) the scatter cursors
    unsigned* udeg = (unsigned*)w; w += al256(N * 4);
    const size_t ntiles = (N + SCAN_TILE - 1) / SCAN_TILE + 1;
    int64_t* tiles = (int64_t*)w; w += al256(ntiles * 8);
    const int long_cap = (int)(E / CSR_WIN + 2);
    int32_t* long_rows = (int32_t*)w; w += al256((size_t)long_cap * 4);
    int32_t* n_long = (int32_t*)w;
    hipError_t e;
    if ((e = grapes_zero_async(deg, N * 4, s)) != hipSuccess) return (int)e;
    if ((e = grapes_zero_async(udeg, N * 4, s)) != hipSuccess) return (int)e;
    if ((e = grapes_zero_async(n_long, 64, s)) != hipSuccess) return (int)e;
    int grid = grapes_div_up(num_edges > 0 ? num_edges : 1, 256 * 8); if (grid > 16384) grid = 16384; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(csr_hist_k, dim3(grid), dim3(256), 0, s, edge_src, edge_dst, (long long)num_edges, num_nodes, deg, status);
    GRAPES_LAUNCH_CHECK();
    int rc = scan_u32_to_i64(deg, (long long)N, rowptr_raw, rowptr_raw + N, tiles, s);
    if (rc) return rc;
    if ((e = grapes_zero_async(deg, N * 4, s)) != hipSuccess) return (int)e;
    hipLaunchKernelGGL(csr_scatter_k, dim3(grid), dim3(256), 0, s, edge_src, edge_dst, (long long)num_edges, num_nodes,
                       (const int64_t*)rowptr_raw, deg, col_raw);
    GRAPES_LAUNCH_CHECK();
    long long nwin = (num_edges + CSR_WIN - 1) / CSR_WIN; if (nwin < 1) nwin = 1;
    int wgrid = (int)(nwin > 65536 ? 65536 : nwin);
    const size_t win_lds = (size_t)CSR_CAP * (sizeof(unsigned long long) + sizeof(int));
    {   // 96 KB of dynamic LDS: set on every call (cheap, per device, no shared flag), after checking that the device has it
        int dev = 0, max_lds = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return (int)hipGetLastError();
        if ((size_t)max_lds < win_lds + 256) return GRAPES_EINVAL;          // gfx950 has 160 KB per workgroup; 64 KB parts are not a target
        e = hipFuncSetAttribute((const void*)csr_window_sort_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)win_lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(csr_window_sort_k, dim3(wgrid), dim3(512), win_lds, s, (const int64_t*)rowptr_raw, num_nodes, col_raw, udeg, long_rows,
                       n_long, long_cap, status);
    GRAPES_LAUNCH_CHECK();
    hipLaunchKernelGGL(csr_long_sort_k, dim3(512), dim3(1024), 0, s, (const int64_t*)rowptr_raw, col_raw, udeg,
                       (const int32_t*)long_rows, (const int32_t*)n_long, long_cap);
    GRAPES_LAUNCH_CHECK();
    rc = scan_u32_to_i64(udeg, (long long)N, rowptr, rowptr + N, tiles, s);
    if (rc) return rc;
    int pgrid = grapes_div_up((int64_t)N, 4); if (pgrid > 65536) pgrid = 65536;
    hipLaunchKernelGGL(csr_place_k, dim3(pgrid), dim3(256), 0, s, (const int64_t*)rowptr_raw, (const int64_t*)rowptr,
                       (const unsigned*)udeg, (const int32_t*)col_raw, col, num_nodes);
    GRAPES_LAUNCH_CHECK();
    if (hipMemcpyAsync(d_nnz, rowptr + N, sizeof(int64_t), hipMemcpyDeviceToDevice, s) != hipSuccess) return (int)hipGetLastError();
    return 0;
}
