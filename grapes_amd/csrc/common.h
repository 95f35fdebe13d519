// Shared device helpers for the gfx950 kernels (64-wide wavefronts throughout).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/grapes_hip.h"

#define GRAPES_WAVE 64
#define DWS_ROWS 128            // rows per slab of the few-row weight-gradient kernels (gemm_kernels.hip) and of whoever sums them

#define GRAPES_LAUNCH_CHECK()                      \
    do {                                           \
        hipError_t _e = hipGetLastError();         \
        if (_e != hipSuccess) return (int)_e;      \
    } while (0)

static inline int grapes_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// A/B and tuning switches (GRAPES_* environment variables: forms that lost their measurement, grid sizes, thresholds) exist in the
// DIAGNOSTIC build only (make diag -> libgrapes_hip_diag.so, -DGRAPES_DIAG; profiles/ and the tests that compare forms use it).
// The product library has one configuration: it never reads the environment.
#include <stdlib.h>
static inline const char* grapes_tune_env(const char* name) {
#ifdef GRAPES_DIAG
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// Effective element count: device word (clamped to the capacity) or the host value.
__device__ __forceinline__ int eff_count(const int32_t* d_n, int n_host) {
    if (d_n == nullptr) return n_host;
    int v = *d_n;
    return v < n_host ? (v < 0 ? 0 : v) : n_host;
}

// Several live counts at once: every count is REQUESTED before the first is looked at (an absent one — NULL pointer or not wanted —
// reads `any`, a valid 4-byte-aligned word, and yields its capacity / zero).  eff_count in a loop is a branch per count: hipcc waits
// for each scalar load before requesting the next (four dependent trips at the head of gemm_dw_split_k and slab_reduce_rank1_k).
template <int N>
__device__ __forceinline__ void eff_counts(const int32_t* const (&d)[N], const int (&cap)[N], const bool (&want)[N], const void* any,
                                           int (&out)[N]) {
    int raw[N];
#pragma unroll
    for (int q = 0; q < N; ++q) raw[q] = *((want[q] && d[q]) ? d[q] : reinterpret_cast<const int32_t*>(any));
#pragma unroll
    for (int q = 0; q < N; ++q) {     // (all N requested HERE: hipcc otherwise sinks each load into the branch that uses it)
        int r = __builtin_amdgcn_readfirstlane(raw[q]);                // (a no-op where hipcc already holds the count in a scalar register)
        asm volatile("" : "+s"(r));
        raw[q] = r;
    }
#pragma unroll
    for (int q = 0; q < N; ++q)
        out[q] = !want[q] ? 0 : (d[q] ? (raw[q] < cap[q] ? (raw[q] < 0 ? 0 : raw[q]) : cap[q]) : cap[q]);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Inclusive scan over the wavefront on the DPP network (no LDS crossbar round trips: the ds_bpermute form of the same scan
// cost six dependent ~100-cycle hops).  Rows of 16 lanes scan with row_shr 1/2/4/8 (lanes without a source add 0), then
// lane 15 of row 0 / 2 is added to row 1 / 3 (row_bcast:15) and lane 31 to rows 2 and 3 (row_bcast:31).
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
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

// Workgroup barrier for data exchanged through LDS ONLY.  __syncthreads() is a workgroup-scope fence on global memory
// too: before the barrier it waits for every outstanding global load and store of the wavefront (s_waitcnt vmcnt(0)) — in
// a loop that keeps prefetches or stores in flight across iterations that wait is a full memory round trip per barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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

// Three exclusive scans at once (the same three barriers as one): a, b, c -> their exclusive prefixes; *ta / *tb / *tc the block sums.
// `lds` needs 51 ints.
__device__ __forceinline__ void block_excl_scan3(int& a, int& b, int& c, int* lds, int* ta, int* tb, int* tc) {
    const int lane = lane_id();
    const int wid = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    const int ia = wave_incl_scan(a), ib = wave_incl_scan(b), ic = wave_incl_scan(c);
    __syncthreads();  // protect lds reuse across consecutive calls
    if (lane == 63) { lds[wid] = ia; lds[17 + wid] = ib; lds[34 + wid] = ic; }
    __syncthreads();
    if (wid == 0) {
        const int xa = (lane < nw) ? lds[lane] : 0, xb = (lane < nw) ? lds[17 + lane] : 0, xc = (lane < nw) ? lds[34 + lane] : 0;
        const int sa = wave_incl_scan(xa), sb = wave_incl_scan(xb), sc = wave_incl_scan(xc);
        if (lane < nw) { lds[lane] = sa - xa; lds[17 + lane] = sb - xb; lds[34 + lane] = sc - xc; }
        if (lane == nw - 1) { lds[16] = sa; lds[33] = sb; lds[50] = sc; }
    }
    __syncthreads();
    *ta = lds[16]; *tb = lds[33]; *tc = lds[50];
    a = lds[wid] + ia - a; b = lds[17 + wid] + ib - b; c = lds[34 + wid] + ic - c;
}

// Zero fill as a KERNEL node (hipMemsetAsync becomes a memset node under stream capture; the large scratch
// clears of the hop pipeline stay ordinary kernel nodes).  bytes and ptr must be multiples of 4.
__global__ static void grapes_zero_k(uint32_t* __restrict__ p, size_t words) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t head = (((uintptr_t)p & 15) ? (16 - ((uintptr_t)p & 15)) / 4 : 0);
    const size_t h = head < words ? head : words;
    if (i0 < h) p[i0] = 0u;
    uint4* q = reinterpret_cast<uint4*>(p + h);
    const size_t nq = (words - h) >> 2;
    for (size_t i = i0; i < nq; i += stride) q[i] = make_uint4(0u, 0u, 0u, 0u);
    const size_t done = h + (nq << 2);
    if (i0 < words - done) p[done + i0] = 0u;
}
static inline hipError_t grapes_zero_async(void* p, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    const size_t words = bytes >> 2;
    size_t blocks = (words / 4 + 255) / 256; if (blocks < 1) blocks = 1; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(grapes_zero_k, dim3((unsigned)blocks), dim3(256), 0, s, (uint32_t*)p, words);
    return hipGetLastError();
}

// "Last workgroup finalises" without a device-scope fence.  On this multi-XCD part __threadfence() has to write back the
// issuing XCD's dirty L2 lines — everything the kernel has stored so far — and costs ~10 us per launch.  Only the few
// words the finalising workgroup reads need to be coherent: they are published with a device-scope atomic exchange
// (performed at the coherence point by the time it returns), the wavefront waits for that return, then takes its
// ticket; the reader uses device-scope atomic loads.
__device__ __forceinline__ void publish_f64(double* slot, double v) {
    (void)atomicExch(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)__double_as_longlong(v));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void publish_f32(float* slot, float v) {
    (void)atomicExch(reinterpret_cast<unsigned int*>(slot), __float_as_uint(v));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- one-launch ordered scans over a FEW workgroups (<= GRAPES_SYNC_SLOTS): every workgroup publishes its total, reads
// the totals of the workgroups before it (all of them are resident: the grid is tiny, and a workgroup publishes before it
// waits, so nothing waits on a workgroup that waits), then the last one to finish puts the scratch back to zero.
// `sync` is caller memory of GRAPES_SYNC_WORDS 64-bit words, zero at rest and left zero; launches that share it must be
// stream-ordered.  Word 0 = finished-workgroup counter, word 1 + b = workgroup b's total (bit 63 = published).
#define GRAPES_SYNC_SLOTS (GRAPES_SYNC_WORDS - 1)
#define GRAPES_SYNC_SPIN_LIMIT (1 << 22)        // polls before a reader gives up: no launch may wait forever
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
// Sum of the totals of workgroups 0..b-1, returned to every thread (`mine` < 2^63: several counts may be packed in it as
// long as their grid-wide sums stay inside their fields).  `lds64` = one 64-bit LDS word.  All threads must call it.
// published: the caller has already stored VALID | mine in sync[1 + b] (as early as it could: the later workgroups wait for it).
__device__ __forceinline__ unsigned long long lookback_exclusive(unsigned long long* sync, int b, unsigned long long mine,
                                                                 unsigned long long* lds64, int32_t* status,
                                                                 bool published = false) {
    const unsigned long long VALID = 1ull << 63;
    if (threadIdx.x == 0 && !published) {
        (void)atomicExch(&sync[1 + b], VALID | mine);      // (no wait for its return: the reads below are issued behind it)
    }
    if (threadIdx.x < 64) {
        // four predecessors per lane in flight (256 per round): a later workgroup reads its whole prefix in ONE round trip
        // instead of one per 64 predecessors; entries not yet published are polled again (bounded)
        unsigned long long acc = 0ull;
        for (int i0 = 0; i0 < b; i0 += 256) {
            unsigned long long v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + 64 * u + (int)threadIdx.x;
                v[u] = i < b ? __hip_atomic_load(&sync[1 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : VALID;
            }
            for (int spin = 0; spin < GRAPES_SYNC_SPIN_LIMIT; ++spin) {
                const bool pending = !(v[0] & v[1] & v[2] & v[3] & VALID);
                if (!__any(pending)) break;
                if (pending) {
                    __builtin_amdgcn_s_sleep(1);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (!(v[u] & VALID)) v[u] = __hip_atomic_load(&sync[1 + i0 + 64 * u + (int)threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (!(v[0] & v[1] & v[2] & v[3] & VALID) && status) atomicOr(status, GRAPES_STATUS_SYNC_TIMEOUT);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u] & ~VALID;
        }
        acc = wave_sum_u64(acc);
        if (threadIdx.x == 0) *lds64 = acc;
    }
    __syncthreads();
    return *lds64;
}
// The same over TWO scratch arrays at once (a second set of packed counts; published by the caller in both): both words of
// every predecessor are requested in the same round trip.  Returns the first array's sum, *pre2 the second's.
// split = 1: the FIRST WAVEFRONT's half only (polls, sums, leaves both sums in lds64; no barrier) — the caller runs it as early as
// it likes and later calls with split = 2 (barrier + read) from every thread; 0: both halves at once.
__device__ __forceinline__ unsigned long long lookback_exclusive2(unsigned long long* sync, unsigned long long* sync2, int b,
                                                                  unsigned long long* lds64 /* two words */, int32_t* status,
                                                                  unsigned long long* pre2, int split = 0) {
    const unsigned long long VALID = 1ull << 63;
    if (threadIdx.x < 64 && split != 2) {
        unsigned long long acc = 0ull, acc2 = 0ull;
        for (int i0 = 0; i0 < b; i0 += 256) {
            unsigned long long v[4], u2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + 64 * u + (int)threadIdx.x;
                v[u] = i < b ? __hip_atomic_load(&sync[1 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : VALID;
                u2[u] = i < b ? __hip_atomic_load(&sync2[1 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : VALID;
            }
            for (int spin = 0; spin < GRAPES_SYNC_SPIN_LIMIT; ++spin) {
                const bool pending = !(v[0] & v[1] & v[2] & v[3] & u2[0] & u2[1] & u2[2] & u2[3] & VALID);
                if (!__any(pending)) break;
                if (pending) {
                    __builtin_amdgcn_s_sleep(1);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (!(v[u] & VALID)) v[u] = __hip_atomic_load(&sync[1 + i0 + 64 * u + (int)threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (!(u2[u] & VALID)) u2[u] = __hip_atomic_load(&sync2[1 + i0 + 64 * u + (int)threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            if (!(v[0] & v[1] & v[2] & v[3] & u2[0] & u2[1] & u2[2] & u2[3] & VALID) && status) atomicOr(status, GRAPES_STATUS_SYNC_TIMEOUT);
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc += v[u] & ~VALID; acc2 += u2[u] & ~VALID; }
        }
        acc = wave_sum_u64(acc); acc2 = wave_sum_u64(acc2);
        if (threadIdx.x == 0) { lds64[0] = acc; lds64[1] = acc2; }
    }
    if (split == 1) return 0ull;
    __syncthreads();
    *pre2 = lds64[1];
    return lds64[0];
}
// After its last read of the scratch: the last of `live` workgroups to get here zeroes it for the next launch.
__device__ __forceinline__ void lookback_finish(unsigned long long* sync, int live, unsigned long long* sync2 = nullptr) {
    if (threadIdx.x != 0) return;
    const unsigned done = atomicAdd(reinterpret_cast<unsigned*>(sync), 1u);
    if ((int)done == live - 1) {
        for (int i = 0; i <= live; ++i) sync[i] = 0ull;
        if (sync2) for (int i = 0; i <= live; ++i) sync2[i] = 0ull;
    }
}

// ---- several dependent phases in ONE launch (cooperative kernels: every workgroup of the grid is resident — the launcher
// keeps the grid at or below the number of compute units — and every workgroup calls every barrier).
// Data that crosses workgroups between two phases never sits in an XCD's private L2: producers use agent-scope stores
// (write-through) or atomics (performed at the memory side), consumers agent-scope loads; so the barrier needs no
// device-scope fence (which would write back the whole L2, see publish_f64 above) — the __syncthreads in front of the arrival
// waits for the workgroup's outstanding stores, the arrival itself is an atomic, the poll an agent-scope load.
// `bar`: two 32-bit words, zero at rest; grid_barrier_finish (after the LAST barrier of the launch) leaves them zero again.
__device__ __forceinline__ int ld_agent(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int32_t* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent_f(const float* p) {
    return __int_as_float(__hip_atomic_load(reinterpret_cast<const int*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent_f(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<int*>(p), __float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned phase /* 1, 2, ... */, int32_t* status) {
    // EVERY thread: its own agent-scope stores of the phase have left the wavefront before the workgroup arrives (a workgroup
    // barrier alone orders LDS, not another wavefront's vector-memory queue: ADVICE r03)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned target = phase * gridDim.x;
        (void)atomicAdd(bar, 1u);
        bool ok = false;
        for (int spin = 0; spin < GRAPES_SYNC_SPIN_LIMIT; ++spin) {
            if (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok && status) atomicOr(status, GRAPES_STATUS_SYNC_TIMEOUT);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__device__ __forceinline__ void grid_barrier_finish(unsigned* bar) {
    if (threadIdx.x != 0) return;
    if (atomicAdd(bar + 1, 1u) == gridDim.x - 1) { (void)atomicExch(bar, 0u); (void)atomicExch(bar + 1, 0u); }
}

// Columns [4c, 4c+4) of the frontier feature row  feat(v) = [ X[v, 0:F] | indicator bits of v | zero padding ]  (main.py:199-204)
// for a chunk that is not wholly inside X (c >= F / 4).  X rows are `ldx` floats apart with ldx a multiple of 4 and zeros
// in the columns [F, ldx) (the resident copy of a feature matrix whose width is not a multiple of 4 is padded once), so the
// chunk that straddles the end of X is one aligned float4 load with the indicator columns laid over it.
__device__ __forceinline__ float4 feat_tail_chunk(const float* __restrict__ X, long long ldx, int F, int v, int c,
                                                  const uint32_t* __restrict__ code, uint32_t epoch,
                                                  uint32_t bit_mask = 0xffu) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (4 * c < F) t = *reinterpret_cast<const float4*>(X + (long long)v * ldx + 4 * c);
    uint32_t cd = code ? code[v] : 0u;
    if ((cd >> 8) != epoch) cd = 0;
    cd &= bit_mask;
    const int b0 = 4 * c - F;                       // indicator bit of component 0 (negative: still an X column)
    if (b0 + 0 >= 0 && b0 + 0 < 8 && ((cd >> (b0 + 0 < 0 ? 0 : b0 + 0)) & 1u)) t.x = 1.f;
    if (b0 + 1 >= 0 && b0 + 1 < 8 && ((cd >> (b0 + 1 < 0 ? 0 : b0 + 1)) & 1u)) t.y = 1.f;
    if (b0 + 2 >= 0 && b0 + 2 < 8 && ((cd >> (b0 + 2 < 0 ? 0 : b0 + 2)) & 1u)) t.z = 1.f;
    if (b0 + 3 >= 0 && b0 + 3 < 8 && ((cd >> (b0 + 3 < 0 ? 0 : b0 + 3)) & 1u)) t.w = 1.f;
    return t;
}

// The same chunk without a branch, for kernels whose lanes hold DIFFERENT chunks of a row (lanes 0..F/4-1 inside X, the next
// lane on the indicator columns): a per-lane branch around the loads makes the two sides take their memory round trips one
// after the other.  Every lane loads a chunk of X (clamped to the row's last padded chunk) and the row's code word; the
// chunk is then corrected in registers.
__device__ __forceinline__ float4 feat_chunk_load(const float* __restrict__ X, long long ldx, int v, int c) {
    const int cl = (int)(ldx >> 2) - 1;
    return *reinterpret_cast<const float4*>(X + (long long)v * ldx + 4 * (c < cl ? c : cl));
}
__device__ __forceinline__ float4 feat_chunk_fix(float4 t, uint32_t cd, int c, int F, uint32_t epoch, uint32_t bit_mask = 0xffu) {
    const int b0 = 4 * c - F;                       // indicator bit of component 0 (negative: still an X column)
    if (b0 >= 0) t = make_float4(0.f, 0.f, 0.f, 0.f);
    cd = ((cd >> 8) == epoch) ? (cd & bit_mask) : 0u;
    const uint32_t sh = b0 > -4 ? (b0 < 0 ? cd << (-b0) : (b0 < 8 ? cd >> b0 : 0u)) : 0u;   // bit j: component j is a set indicator
    const uint32_t live = b0 > -4 ? (b0 < 0 ? 0xffu << (-b0) : (b0 < 8 ? 0xffu >> b0 : 0u)) : 0u;   // bit j: component j is an indicator column
    const uint32_t on = sh & live;
    t.x = (on & 1u) ? 1.f : t.x; t.y = (on & 2u) ? 1.f : t.y; t.z = (on & 4u) ? 1.f : t.z; t.w = (on & 8u) ? 1.f : t.w;
    return t;
}

// ---- kernel clock table (measurement only; include/grapes_hip.h: grapes_kernel_clock_*).  While a table is enabled, the
// launchers of the roofline kernels reserve one (begin, end) pair of 100 MHz s_memrealtime stamps per WAVEFRONT and pass its
// address; otherwise they pass NULL and no stamp executes.  The stamps sit before the first and after the last instruction
// of a wavefront (the end stamp waits for the wavefront's own stores), never inside a loop, and leave the kernel only through
// the table, which nothing else reads.
unsigned long long* grapes_clock_reserve(const char* kernel, int grid, int waves_per_block);   // host; NULL when disabled
bool grapes_clock_enabled();                                                                    // host
__device__ __forceinline__ unsigned long long grapes_clock_begin(const unsigned long long* clk) {
    return clk ? wall_clock64() : 0ull;
}
__device__ __forceinline__ void grapes_clock_end(unsigned long long* clk, unsigned long long t0) {
    if (!clk) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        clk[2 * w] = t0; clk[2 * w + 1] = t1;
    }
}

// ---- diagnostic build only (make stamps -> libgrapes_hip_stamps.so, profiles/stamp_probe.py): phase stamps INSIDE a kernel, to
// see where a latency-bound launch spends its microseconds.  The product build compiles GRAPES_STAMP to nothing.
#ifdef GRAPES_STAMPS
static __device__ unsigned long long* grapes_stamp_ptr = nullptr;
#define GRAPES_STAMP(slot)                                                                                   \
    do {                                                                                                     \
        if (grapes_stamp_ptr && threadIdx.x == 0 && blockIdx.x < 64) {                                       \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                      \
            grapes_stamp_ptr[blockIdx.x * 16 + (slot)] = wall_clock64();                                      \
        }                                                                                                    \
    } while (0)
// (no wait: the time the wavefront ARRIVES at the point, whatever it still has in flight)
#define GRAPES_STAMP_NW(slot)                                                                                \
    do {                                                                                                     \
        if (grapes_stamp_ptr && threadIdx.x == 0 && blockIdx.x < 64)                                         \
            grapes_stamp_ptr[blockIdx.x * 16 + (slot)] = wall_clock64();                                      \
    } while (0)
#define GRAPES_STAMP_SETTER(name)                                                                            \
    extern "C" int name(unsigned long long* p) {                                                             \
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(grapes_stamp_ptr), &p, sizeof(p));                          \
    }
#else
#define GRAPES_STAMP(slot) do { } while (0)
#define GRAPES_STAMP_NW(slot) do { } while (0)
#define GRAPES_STAMP_SETTER(name)
#endif

// rows of a bf16x3 weight image (gemm_tiled_split.hip: TS_BN) — also known to the optimiser launch, which mirrors updated weights into it
#define GRAPES_TS_IMG_ROWS 256

// ---- the end of a deferred draw (include/grapes_hip.h: grapes_draw_finish_args), run by ONE workgroup of 256 threads of the caller's
// next launch: the log-prob partial sums in the order of sampler_emit_k's last workgroup — 1024 virtual threads (thread b owns partial
// b), butterfly inside a virtual wavefront, virtual wavefronts in index order — and the histogram's return to zero.
#ifdef __HIPCC__
// The draw's statistics (utils.py:47-56: min / max of p, mean and unbiased std of the entropies) from the per-workgroup partials
// [blocks][5] = (min, max, sum, sum of squares, -), in ONE order whoever runs it: 1024 virtual threads (thread b owns partial b),
// butterfly inside a virtual wavefront, virtual wavefronts in index order.  NT = the calling workgroup's threads (a multiple of 64
// that divides 1024); all of them must call it.  Partials are read with agent-scope loads (the caller may be the launch that wrote them).
template <int NT>
__device__ __forceinline__ void draw_stats_final(const double* __restrict__ parts, int blocks, int n, float* __restrict__ stats) {
    __shared__ double ds_red[4][16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
#pragma unroll
    for (int q = 0; q < 1024 / NT; ++q) {
        double mn = (double)INFINITY, mx = -(double)INFINITY, s1 = 0.0, s2 = 0.0;
        for (int b = tid + NT * q; b < blocks; b += 1024) {              // (virtual thread b: partials b, b + 1024, ... in that order)
            const long long* p = reinterpret_cast<const long long*>(parts + 5 * (size_t)b);
            mn = fmin(mn, __longlong_as_double(__hip_atomic_load(p + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
            mx = fmax(mx, __longlong_as_double(__hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
            s1 += __longlong_as_double(__hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            s2 += __longlong_as_double(__hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            mn = fmin(mn, __shfl_xor(mn, d, 64)); mx = fmax(mx, __shfl_xor(mx, d, 64));
            s1 += __shfl_xor(s1, d, 64); s2 += __shfl_xor(s2, d, 64);
        }
        if (lane == 0) { const int vw = wid + (NT / 64) * q; ds_red[0][vw] = mn; ds_red[1][vw] = mx; ds_red[2][vw] = s1; ds_red[3][vw] = s2; }
    }
    __syncthreads();
    if (tid == 0) {
        double mn = (double)INFINITY, mx = -(double)INFINITY, s1 = 0.0, s2 = 0.0;
        for (int w = 0; w < 16; ++w) { mn = fmin(mn, ds_red[0][w]); mx = fmax(mx, ds_red[1][w]); s1 += ds_red[2][w]; s2 += ds_red[3][w]; }
        const double mean = s1 / (double)n;
        double var = n > 1 ? (s2 - s1 * s1 / (double)n) / (double)(n - 1) : 0.0;   // torch.std_mean: unbiased
        if (var < 0.0) var = 0.0;
        stats[0] = (float)mn; stats[1] = (float)mx; stats[2] = (float)mean; stats[3] = (float)sqrt(var);
    }
}
__device__ __forceinline__ void draw_finish_body(const grapes_draw_finish_args& f) {
    __shared__ double df_red[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n = eff_count(f.d_n, f.n_host);
    const bool keep_all = f.sel[2] != 0u;
    const int nb = keep_all ? f.keys_blocks : (n + f.emit_block - 1) / f.emit_block;
    const double* parts = keep_all ? f.parts_keys : f.parts_emit;
    const int pstride = keep_all ? 5 : 1;
    double v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double acc = 0.0;
        for (int bb = tid + 256 * q; bb < nb; bb += 1024) acc += parts[(size_t)bb * pstride];
        v[q] = acc;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { const double w = wave_sum_d(v[q]); if (lane == 0) df_red[wid + 4 * q] = w; }
    __syncthreads();
    if (tid == 0 && f.stats) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += df_red[w];
        f.stats[4] = (float)t;
    }
    if (f.hist)
        for (int b = tid; b < f.hist_words; b += 256) f.hist[b] = 0u;
    // the one-launch draw leaves its statistics to this workgroup as well (stats_blocks partials in front of parts_keys' column)
    if (f.stats_blocks > 0 && !keep_all && f.stats) draw_stats_final<256>(f.parts_keys - 4, nb, n, f.stats);      // (nb: the live workgroups)
}
#endif

// ---- riders (include/grapes_hip.h: grapes_rider_*; riders.hip): launches recorded instead of issued, to be carried later as
// EXTRA WORKGROUPS of another launch of the same kernel ("two problems side by side in one launch": the next step's
// weight-independent index chain inside the current step's hop-1 launches).  Host-side only.
#include <functional>
#include <type_traits>
#include <string.h>
enum GrapesRiderKind { GRAPES_RK_OTHER = 0, GRAPES_RK_EXPAND, GRAPES_RK_COMPACT, GRAPES_RK_FILL, GRAPES_RK_SORT, GRAPES_RK_GATHER,
                       GRAPES_RK_BEGIN /* grapes_step_begin: rides as one extra workgroup of an expansion, also while the rest is held */,
                       GRAPES_RK_AGGBWD /* few-row backward aggregation (variant = vector width): rides in the sampler heads' backward launches */ };
struct GrapesRiderRecord {
    int kind = GRAPES_RK_OTHER, variant = 0, grid = 0, block = 0;
    size_t arg_bytes = 0;
    alignas(16) unsigned char args[768];
    std::function<void(hipStream_t)> single;          // the launch on its own (no host of its kind came, or kind OTHER)
};
bool grapes_rider_recording();
// a RECORDED launch rides beside a host on the critical path: it takes fewer workgroups than it would on its own (its loops are
// grid-stride), so that the host keeps its share of the chip (GRAPES_RIDER_GRID: default 256 workgroups; 0 = no cap)
int grapes_rider_grid(int grid);
void grapes_rider_record(GrapesRiderRecord&& r);
// host side of a pairable launch: issues the pending records that cannot ride (kind OTHER) on `s`, then returns the next
// pending record if it is (kind, variant, block) — consumed — or NULL
#define GRAPES_RIDER_ANY_VARIANT (-0x7fffffff)        // (grapes_rider_match: whatever variant the pending record of `kind` has)
const GrapesRiderRecord* grapes_rider_match(int kind, int variant, int block, hipStream_t s);
template <class Args>
static inline GrapesRiderRecord grapes_rider_make(int kind, int variant, int grid, int block, const Args& a,
                                                  std::function<void(hipStream_t)> single) {
    static_assert(sizeof(Args) <= sizeof(GrapesRiderRecord::args), "rider argument block too small");
    static_assert(std::is_trivially_copyable<Args>::value, "rider arguments must be trivially copyable");
    GrapesRiderRecord r;
    r.kind = kind; r.variant = variant; r.grid = grid; r.block = block; r.arg_bytes = sizeof(Args);
    memcpy(r.args, &a, sizeof(Args));
    r.single = std::move(single);
    return r;
}
// a launch that never rides: recorded as kind OTHER while recording (returns true), else the caller launches it
#define GRAPES_RIDER_OTHER(stream_, launch_expr_)                                                                         \
    do {                                                                                                                  \
        if (grapes_rider_recording()) {                                                                                   \
            GrapesRiderRecord r_;                                                                                         \
            r_.single = [=](hipStream_t s_) { hipStream_t stream_ = s_; (void)stream_; launch_expr_; };                   \
            grapes_rider_record(std::move(r_));                                                                           \
        } else {                                                                                                          \
            launch_expr_;                                                                                                 \
        }                                                                                                                 \
    } while (0)
