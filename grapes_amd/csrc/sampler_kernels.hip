// A2: GFlowNet sampler draw — modules/utils.py:13-71 (training) and eval.py:126-130 (greedy):
// keys -> 4-pass radix select of the k-th largest key -> position-ordered compaction -> Bernoulli
// log-probs + statistics.
//
// THIS FILE IS COMPILED WITH -ffp-contract=off: p_expf / p_logf below must execute exactly the
// operation sequence of oracle/portable_math.py (IEEE +,-,*,/ only), so that the Gumbel-top-k
// keys — and therefore the sampled index sets — are bit-identical on the CPU oracle and on gfx950.
#include "common.h"
#include "narrow.h"
GRAPES_STAMP_SETTER(grapes_stamp_set_sampler)
#include <cstdlib>

#pragma clang fp contract(off)

// ---------------------------------------------------------------------------- portable fp32 math
// p_logf / p_expf / p_scale2 follow the polynomial schemes and constants of FreeBSD msun's e_logf.c / e_expf.c (as carried by
// musl), restated with IEEE +,-,*,/ only so that the numpy oracle (oracle/portable_math.py) performs the identical operation
// sequence.  Those sources carry this notice, preserved here as it requires:
/*
 * ====================================================
 * Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
 *
 * Developed at SunPro, a Sun Microsystems, Inc. business.
 * Permission to use, copy, modify, and distribute this
 * software is freely granted, provided that this notice
 * is preserved.
 * ====================================================
 */
__device__ __forceinline__ float p_logf(float x) {
    uint32_t ix = __float_as_uint(x);
    if ((ix & 0x7fffffffu) == 0u) return -INFINITY;
    if (ix >> 31) return (x != x) ? x : __uint_as_float(0x7fc00000u);
    if ((ix & 0x7fffffffu) >= 0x7f800000u) return x + x;
    int k = 0;
    if (ix < 0x00800000u) { k = -25; x = x * 33554432.0f; ix = __float_as_uint(x); }
    ix += 0x3f800000u - 0x3f3504f3u;
    k += (int)(ix >> 23) - 127;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    x = __uint_as_float(ix);
    const float f = x - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    const float w = z * z;
    const float t1 = w * (0.40000972152f + w * 0.24279078841f);
    const float t2 = z * (0.66666662693f + w * 0.28498786688f);
    const float R = t2 + t1;
    const float hfsq = (0.5f * f) * f;
    const float dk = (float)k;
    return ((((s * (hfsq + R)) + dk * 9.0580006145e-06f) - hfsq) + f) + dk * 6.9313812256e-01f;
}

__device__ __forceinline__ float p_scale2(float y, int k) {
    int k1 = k < -100 ? -100 : (k > 100 ? 100 : k);
    int k2 = k - k1;
    k2 = k2 < -100 ? -100 : (k2 > 100 ? 100 : k2);
    const float m1 = __uint_as_float((uint32_t)(k1 + 127) << 23);
    const float m2 = __uint_as_float((uint32_t)(k2 + 127) << 23);
    return (y * m1) * m2;
}

__device__ __forceinline__ float p_expf(float x) {
    const uint32_t hx = __float_as_uint(x);
    const int sign = (int)(hx >> 31);
    const uint32_t ax = hx & 0x7fffffffu;
    if (ax > 0x7f800000u) return x;                                  // NaN
    if (ax >= 0x42b17218u && !sign) return INFINITY;                 // x >= 88.722839
    if (ax >= 0x42cff1b5u && sign) return 0.0f;                      // x <= -103.972084
    if (ax <= 0x39000000u) return 1.0f + x;                          // |x| <= 2^-13
    int k = 0;
    float hi = x, lo = 0.0f, xr = x;
    if (ax > 0x3eb17218u) {                                          // |x| > 0.5 ln2
        if (ax > 0x3f851592u) {                                      // |x| > 1.5 ln2
            const float kf = 1.4426950216e+00f * x + (sign ? -0.5f : 0.5f);
            k = (int)kf;                                             // truncation toward zero
        } else {
            k = 1 - sign - sign;
        }
        const float kfl = (float)k;
        hi = x - kfl * 6.9314575195e-01f;
        lo = kfl * 1.4286067653e-06f;
        xr = hi - lo;
    }
    const float xx = xr * xr;
    const float c = xr - xx * (1.6666625440e-1f + xx * -2.7667332906e-3f);
    const float y = 1.0f + (((xr * c) / (2.0f - c) - lo) + hi);
    return p_scale2(y, k);
}

__device__ __forceinline__ float p_sigmoid(float l) { return 1.0f / (1.0f + p_expf(-l)); }

// Gumbel(0,1) from the torch.rand value r: Uniform(tiny, 1-eps) then -log(-log(u)) (utils.py:40-41)
__device__ __forceinline__ float p_gumbel(float r) {
    const float tiny = 1.17549435e-38f;
    const float span = (1.0f - 1.1920929e-07f) - tiny;
    const float u = r * span + tiny;
    const float x1 = p_logf(u);
    const float x3 = p_logf(-x1);
    return -x3;
}

__device__ __forceinline__ uint32_t order_key(float key) {
    const uint32_t b = __float_as_uint(key);
    return (b >> 31) ? ~b : (b | 0x80000000u);
}
// Greedy draws (mode 1, eval.py:126-127) rank PROBABILITIES: every key lies in [0, 1] and most of them in one or two binades
// below 1, where the leading 12 bits of order_key() — sign, exponent, three fraction bits — tell eight values per binade apart:
// nearly every key fell into a handful of first-level bins and the draw into its scan form (62 us instead of 17 at the
// products shape).  Same order, other spacing: [0.25, 1] is stretched over half of the 32-bit range (512 first-level bins per
// binade), smaller keys keep their own bits below it.  Strictly increasing on [0, 1] (no new ties); anything else — a NaN —
// keeps order_key()'s place at one end.
__device__ __forceinline__ uint32_t order_key_prob(float p) {
    const uint32_t b = __float_as_uint(p);
    if (b <= 0x3f800000u) return b >= 0x3e800000u ? 0x40000000u + ((b - 0x3e800000u) << 7) : b;
    return order_key(p);
}
__device__ __forceinline__ uint32_t order_key_of(float key, int mode) { return mode == 1 ? order_key_prob(key) : order_key(key); }

// ---------------------------------------------------------------------------- Philox4x32-10
struct Philox4 { uint32_t v[4]; };
__device__ __forceinline__ Philox4 philox4x32_10(uint64_t ctr, uint64_t seed) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0u, c3 = 0u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox4 r; r.v[0] = c0; r.v[1] = c1; r.v[2] = c2; r.v[3] = c3;
    return r;
}
__device__ __forceinline__ float philox_uniform_at(uint64_t seed, uint64_t offset, long long i) {
    const Philox4 p = philox4x32_10(offset + (uint64_t)(i >> 2), seed);
    return (float)(p.v[i & 3] >> 8) * 5.9604644775390625e-08f;   // 2^-24
}

__global__ void philox_uniform_k(float* __restrict__ out, long long n, uint64_t seed, uint64_t offset) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        out[i] = philox_uniform_at(seed, offset, i);
}

// ---------------------------------------------------------------------------- dropout (modules/gcn.py:33,37)
// F.dropout(x, p, training=True) on the Philox stream of the sampler: element i of the live [n, f] matrix (row-major) is
// KEPT iff philox_uniform(seed, offset, i) >= p and then scaled by 1 / (1 - p); the keep flags are written as bytes for the
// backward pass.  The stream advances by ceil(n f / 4) counters per call (dropout_advance_k: its own launch, after every
// workgroup of the forward kernel has read the offset).
__global__ __launch_bounds__(256) void dropout_fwd_k(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ keep,
                                                     int n_host, const int32_t* d_n, int f, float p, uint64_t seed,
                                                     uint64_t offset, const uint64_t* d_offset) {
    const long long total = (long long)eff_count(d_n, n_host) * f;
    const uint64_t off = d_offset ? *d_offset : offset;
    const float scale = p < 1.0f ? 1.0f / (1.0f - p) : 0.f;          // p == 1: everything dropped (F.dropout returns zeros)
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; 4 * g < total; g += (long long)gridDim.x * blockDim.x) {
        const Philox4 r = philox4x32_10(off + (uint64_t)g, seed);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long i = 4 * g + u;
            if (i < total) {
                const bool k = (float)(r.v[u] >> 8) * 5.9604644775390625e-08f >= p;
                keep[i] = k ? 1 : 0;
                y[i] = k ? x[i] * scale : 0.f;
            }
        }
    }
}
__global__ void dropout_advance_k(uint64_t* d_offset, int n_host, const int32_t* d_n, int f) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *d_offset += (uint64_t)(((long long)eff_count(d_n, n_host) * f + 3) >> 2);
}
__global__ __launch_bounds__(256) void dropout_bwd_k(const float* __restrict__ dy, const uint8_t* __restrict__ keep,
                                                     float* __restrict__ dx, int n_host, const int32_t* d_n, int f, float p) {
    const long long total = (long long)eff_count(d_n, n_host) * f;
    const float scale = p < 1.0f ? 1.0f / (1.0f - p) : 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        dx[i] = keep[i] ? dy[i] * scale : 0.f;
}
extern "C" int grapes_dropout_fwd(const float* x, float* y, uint8_t* keep, int32_t n, const int32_t* d_n, int32_t f, float p,
                                  uint64_t philox_seed, uint64_t philox_offset, uint64_t* d_philox_offset,
                                  grapes_stream_t stream) {
    if (n < 0 || f <= 0 || !(p >= 0.f && p <= 1.f)) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!x || !y || !keep) return GRAPES_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int grid = grapes_div_up(((long long)n * f + 3) / 4, 256); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(dropout_fwd_k, dim3(grid), dim3(256), 0, s, x, y, keep, n, d_n, f, p, philox_seed, philox_offset,
                       (const uint64_t*)d_philox_offset);
    GRAPES_LAUNCH_CHECK();
    if (d_philox_offset) {
        hipLaunchKernelGGL(dropout_advance_k, dim3(1), dim3(64), 0, s, d_philox_offset, n, d_n, f);
        GRAPES_LAUNCH_CHECK();
    }
    return 0;
}
extern "C" int grapes_dropout_bwd(const float* dy, const uint8_t* keep, float* dx, int32_t n, const int32_t* d_n, int32_t f,
                                  float p, grapes_stream_t stream) {
    if (n < 0 || f <= 0 || !(p >= 0.f && p <= 1.f)) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!dy || !keep || !dx) return GRAPES_EINVAL;
    int grid = grapes_div_up((long long)n * f, 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(dropout_bwd_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, keep, dx, n, d_n, f, p);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

extern "C" int grapes_philox_uniform(float* out, int64_t n, uint64_t seed, uint64_t offset,
                                     grapes_stream_t stream) {
    if (n < 0 || (!out && n > 0)) return GRAPES_EINVAL;
    if (n == 0) return 0;
    int grid = grapes_div_up(n, 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(philox_uniform_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, (long long)n, seed, offset);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------- log-sigmoid (tolerance path)
// log_prob is compared at 1e-5 (not bit-exact): stable BCE-with-logits form, finite for every
// finite logit, denormals kept (utils.py:71).
__device__ __forceinline__ float log_sigmoid_f(float x) {
    return fminf(x, 0.0f) - log1pf(expf(-fabsf(x)));
}

// ---------------------------------------------------------------------------- the draw
// Three short launches (only the threshold search is a single workgroup):
//   sampler_keys_k       (many workgroups) keys with the portable math, order keys, log-sigmoid, radix pass 1
//                                          histogram, per-workgroup statistics partials
//   sampler_threshold_k  (one workgroup)   radix select of the k-th largest order key: pass 1 from the histogram,
//                                          passes 2-4 on the LDS-resident candidates of the selected bin
//   sampler_emit_k       (many workgroups) position-ordered compaction (each workgroup recounts its prefix), mask, kept ids,
//                                          Bernoulli log-probs
//                                          the last workgroup (ticket) finalises: kept count, sum of log-probs, Philox counter
struct SamplerArgs {
    const float* logits; const int32_t* logit_index; const float* uniforms;
    uint64_t seed; uint64_t offset; uint64_t* d_offset;
    int n_host; const int32_t* d_n; int k; int mode;
    const int32_t* cand_ids; float* mask; int32_t* kept_pos; int32_t* kept_ids; int32_t* d_kept_count;
    float* log_prob; float* keys_out; float* stats;
    // optional: union_ids = [prefix_ids (prefix_n) | kept ids], *d_union_count = prefix_n + kept count — the next
    // hop's query list (main.py:236-238: batch_nodes = cat(target_nodes, sampled nodes)) without extra launches
    const int32_t* prefix_ids; int prefix_n; int32_t* union_ids; int32_t* d_union_count;
    // optional (round 5): the next query list's ROW EXTENTS beside its ids — union_ext[j] = (rowptr[id_j], rowptr[id_j + 1]) for the
    // graph the caller expands next (prefix_ext: the prefix ids' pairs), so that grapes_frontier_expand_fused_ext reads ids and
    // extents in ONE round trip instead of two dependent ones
    const long long* rowptr; const long long* prefix_ext; long long* union_ext;
    // workspace
    uint32_t* ord; float* ls; unsigned long long* gtm; unsigned long long* eqm; int32_t* eqb; int32_t* selb;
    double* part;   // [KEYS_BLOCKS][5]: pmin, pmax, sum ent, sum ent^2, sum log_sigmoid
    int32_t* hist0; // [KEYS_BLOCKS][256] per-workgroup histograms of the top byte of the order keys (no atomics, no memset)
    uint32_t* ticket_zero;   // zeroed by the keys launch: the emit launch's ticket when no selection launch runs between them
    // optional (round 3): ONE histogram of the top GH_BITS bits of the order keys for the whole draw, GH_BINS words, zero at rest —
    // the keys launch adds its workgroups' non-empty bins (integer atomics), the emit launch reads it and its last workgroup
    // puts it back to zero.  With 12 bits the bin of the k-th largest key holds ~0.3 % of the candidates instead of the
    // ~25 % of an 8-bit (sign + exponent) bin: the selection every emit workgroup works out shrinks from a scan that appends
    // thousands of keys to LDS + three passes over them to a scan that appends ~100 + three short passes.
    uint32_t* ghist;
    int norank;              // GRAPES_SAMPLER_RANK=0 (A/B): the three radix passes even over a short candidate list
    int defer_finish;        // grapes_gumbel_topk_deferred: the emit launch has no tail (see include/grapes_hip.h)
};
#define GH_BITS 12
#define GH_BINS (1 << GH_BITS)

#define KEYS_BLOCKS 512
#define PART_BLOCKS 2048        // statistics / log-prob partials: one per workgroup of the keys launch, or per block of the one-launch draw

// Histogram one digit per lane into an LDS table.  Keys cluster in a few exponent bins, and
// same-address LDS atomics serialise, so each wavefront first peels off its (up to two) most common
// digits with a ballot and adds each with ONE atomic; only the remaining lanes add individually.
// Works under partial exec masks (ballots see the active lanes only); digit < 0 = no contribution.
__device__ __forceinline__ void wave_hist_add(int* hist, int digit, int lane) {
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        const unsigned long long act = __ballot(digit >= 0);
        if (act != 0ull) {
            const int leader = __ffsll((long long)act) - 1;
            const int d0 = __shfl(digit, leader, 64);
            const unsigned long long same = __ballot(digit == d0);
            if (lane == leader) atomicAdd(&hist[d0], __popcll(same));
            if (digit == d0) digit = -1;
        }
    }
    if (digit >= 0) atomicAdd(&hist[digit], 1);
}

#define KEYS_THREADS_MAX 1024
static int keys_threads() { static int v = 0; if (!v) { const char* e = grapes_tune_env("GRAPES_KEYS_THREADS"); v = e ? atoi(e) : 1024; if (v != 256 && v != 512 && v != 1024) v = 1024; } return v; }   // measured: 1024 / 512 beat 256 (more wavefronts per SIMD hide the dependent loads)
__global__ __launch_bounds__(KEYS_THREADS_MAX) void sampler_keys_k(SamplerArgs a) {
    __shared__ double red[5][KEYS_THREADS_MAX / 64];
    __shared__ int hist[GH_BINS];                  // (the first 256 words in the per-workgroup-row form)
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    const int hbins = a.ghist ? GH_BINS : 256, hshift = a.ghist ? 32 - GH_BITS : 24;
    // the candidate count, the Philox counter and this thread's first logit index leave together: the index is read inside
    // the CAPACITY (the live count is not known yet) and only used once i < n holds — one dependent round trip less
    const int i_first = blockIdx.x * blockDim.x + tid;
    int idx_next = (a.logit_index && i_first < a.n_host) ? a.logit_index[i_first] : i_first;
    uint64_t offset = a.offset;
    if (a.d_offset) offset = *a.d_offset;
    const int n = eff_count(a.d_n, a.n_host);
    const bool keep_all = n <= a.k;                                    // utils.py:31-33
    float pmin = INFINITY, pmax = -INFINITY;
    double esum = 0.0, esq = 0.0, lsum = 0.0;
    GRAPES_STAMP(11);
    for (int b = tid; b < hbins; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    for (int i = i_first; i < n; i += gridDim.x * blockDim.x) {
        const int idx = idx_next;
        const int i_next = i + gridDim.x * blockDim.x;
        if (a.logit_index && i_next < n) idx_next = a.logit_index[i_next];
        const float l = a.logits[a.logit_index ? idx : i];
        const float lsg = log_sigmoid_f(l);
        a.ls[i] = lsg;
        if (keep_all) {
            a.mask[i] = 1.0f;
            a.kept_pos[i] = i;
            if (a.kept_ids && a.cand_ids) a.kept_ids[i] = a.cand_ids[i];
            if (a.union_ids && a.cand_ids) a.union_ids[a.prefix_n + i] = a.cand_ids[i];
            if (a.union_ext && a.rowptr && a.cand_ids) {
                const int cid = a.cand_ids[i];
                *reinterpret_cast<longlong2*>(a.union_ext + 2 * (long long)(a.prefix_n + i)) = make_longlong2(a.rowptr[cid], a.rowptr[cid + 1]);
            }
            if (a.log_prob) a.log_prob[i] = lsg;
            lsum += (double)lsg;
            continue;
        }
        const float p = p_sigmoid(l);
        float key;
        if (a.mode == 1) {
            key = p;                                                   // eval.py:126-127
        } else {
            const float r = a.uniforms ? a.uniforms[i] : philox_uniform_at(a.seed, offset, i);
            key = p_logf(p) + p_gumbel(r);                             // utils.py:42
        }
        const uint32_t ok = order_key_of(key, a.mode);
        a.ord[i] = ok;
        wave_hist_add(hist, (int)(ok >> hshift), lane);                // radix pass 1, spread over the chip
        if (a.keys_out) a.keys_out[i] = key;
        if (a.stats) {
            pmin = fminf(pmin, p); pmax = fmaxf(pmax, p);
            float ent = -(p * log2f(p) + (1.0f - p) * log2f(1.0f - p));   // utils.py:47
            if (ent != ent) ent = 0.0f;                                   // utils.py:52-54
            esum += (double)ent; esq += (double)ent * (double)ent;
        }
    }
    GRAPES_STAMP(12);
    __syncthreads();
    if (a.ghist) {                                 // non-empty bins into the draw's histogram (a few hundred atomics per workgroup)
        if (!keep_all)
            for (int b = tid; b < GH_BINS; b += blockDim.x) { const int v = hist[b]; if (v) atomicAdd(&a.ghist[b], (uint32_t)v); }
    } else if (tid < 256) a.hist0[blockIdx.x * 256 + tid] = keep_all ? 0 : hist[tid];   // summed by sampler_threshold_k
    if (a.ticket_zero && blockIdx.x == 0 && tid == 0) *a.ticket_zero = 0u;       // the emit launch's ticket (one-launch selection)
    GRAPES_STAMP(13);
    // only the sums this draw uses cross the wavefront (each double reduction is twelve dependent lane exchanges)
    if (a.stats && !keep_all) { pmin = wave_min(pmin); pmax = wave_max(pmax); esum = wave_sum_d(esum); esq = wave_sum_d(esq); }
    if (keep_all) lsum = wave_sum_d(lsum);
    if (lane == 0) { red[0][wid] = pmin; red[1][wid] = pmax; red[2][wid] = esum; red[3][wid] = esq; red[4][wid] = lsum; }
    __syncthreads();
    if (wid == 0) {   // wavefront 0, lane w = wavefront w's partial: a fixed 16-leaf tree (lanes past the last wavefront hold identities)
        const int nw = (int)(blockDim.x >> 6);
        const bool live = lane < nw;
        const int wl = live ? lane : 0;
        double mn = live ? red[0][wl] : (double)INFINITY, mx = live ? red[1][wl] : -(double)INFINITY;
        double s1 = live ? red[2][wl] : 0.0, s2 = live ? red[3][wl] : 0.0, s3 = live ? red[4][wl] : 0.0;
#pragma unroll
        for (int d = 8; d > 0; d >>= 1) {
            mn = fmin(mn, __shfl_xor(mn, d, 64)); mx = fmax(mx, __shfl_xor(mx, d, 64));
            s1 += __shfl_xor(s1, d, 64); s2 += __shfl_xor(s2, d, 64); s3 += __shfl_xor(s3, d, 64);
        }
        if (lane == 0) {
            double* o = a.part + 5 * blockIdx.x;
            o[0] = mn; o[1] = mx; o[2] = s1; o[3] = s2; o[4] = s3;
        }
    }
    GRAPES_STAMP(14);
}


// ---- the keys launch fused into the aggregation that produces the inclusion logits (main.py:210-213): the sampler net's
// 1-wide last layer is  logits = Â (act w2ᵀ) + b2  over the hop's batch rows (narrow.h), and a batch row that is a candidate
// (cand_pos[row] = its position in neighbor_nodes, from the compaction) goes straight on to its key — one launch less per
// hop, and the key arithmetic (Philox, the portable exp / log chains) runs on as many workgroups as the batch has 256-row
// blocks instead of ceil(candidates / 1024).  Same per-candidate operations as sampler_keys_k (bit-identical keys, masks and
// log-probabilities); the statistics partials are grouped by batch block instead of candidate block (double sums: the fp32
// results agree to the last bit or two).
struct NarrowAgg {
    const float* h; const int32_t* rowptr; const int32_t* csr; const float* dinv; const float* bias; float* out;
    int n_host; const int32_t* d_n; int lane_rows; const int32_t* cand_pos;
};
__global__ __launch_bounds__(256) void sampler_agg_keys_k(NarrowAgg g, SamplerArgs a) {
    __shared__ double red[5][4];
    __shared__ int hist[256];
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    const int nrows = eff_count(g.d_n, g.n_host);
    const int n = eff_count(a.d_n, a.n_host);
    const bool keep_all = n <= a.k;                                    // utils.py:31-33
    uint64_t offset = a.offset;
    if (a.d_offset) offset = *a.d_offset;
    float pmin = INFINITY, pmax = -INFINITY;
    double esum = 0.0, esq = 0.0, lsum = 0.0;
    hist[tid] = 0;
    __syncthreads();
    for (int bbase = blockIdx.x * 256; bbase < nrows; bbase += gridDim.x * 256) {   // uniform per workgroup
        narrow_block(g.h, g.rowptr, g.csr, g.dinv, g.bias, g.out, nrows, 1, 0, g.lane_rows, bbase);   // (ends with a barrier)
        const int row = bbase + tid;
        int i = row < nrows ? g.cand_pos[row] : -1;
        if (i >= n) i = -1;
        int digit = -1;
        if (i >= 0) {
            const float l = __builtin_nontemporal_load(g.out + row);   // written by this workgroup before the barrier
            const float lsg = log_sigmoid_f(l);
            a.ls[i] = lsg;
            if (keep_all) {
                a.mask[i] = 1.0f;
                a.kept_pos[i] = i;
                if (a.kept_ids && a.cand_ids) a.kept_ids[i] = a.cand_ids[i];
                if (a.union_ids && a.cand_ids) a.union_ids[a.prefix_n + i] = a.cand_ids[i];
                if (a.log_prob) a.log_prob[i] = lsg;
                lsum += (double)lsg;
            } else {
                const float p = p_sigmoid(l);
                float key;
                if (a.mode == 1) {
                    key = p;                                                   // eval.py:126-127
                } else {
                    const float r = a.uniforms ? a.uniforms[i] : philox_uniform_at(a.seed, offset, i);
                    key = p_logf(p) + p_gumbel(r);                             // utils.py:42
                }
                const uint32_t ok = order_key_of(key, a.mode);
                a.ord[i] = ok;
                digit = (int)(ok >> 24);
                if (a.keys_out) a.keys_out[i] = key;
                if (a.stats) {
                    pmin = fminf(pmin, p); pmax = fmaxf(pmax, p);
                    float ent = -(p * log2f(p) + (1.0f - p) * log2f(1.0f - p));   // utils.py:47
                    if (ent != ent) ent = 0.0f;                                   // utils.py:52-54
                    esum += (double)ent; esq += (double)ent * (double)ent;
                }
            }
        }
        wave_hist_add(hist, digit, lane);                              // radix pass 1 (digit < 0: no contribution)
    }
    __syncthreads();
    a.hist0[blockIdx.x * 256 + tid] = keep_all ? 0 : hist[tid];       // summed by the selection
    if (a.ticket_zero && blockIdx.x == 0 && tid == 0) *a.ticket_zero = 0u;
    pmin = wave_min(pmin); pmax = wave_max(pmax);
    esum = wave_sum_d(esum); esq = wave_sum_d(esq); lsum = wave_sum_d(lsum);
    if (lane == 0) { red[0][wid] = pmin; red[1][wid] = pmax; red[2][wid] = esum; red[3][wid] = esq; red[4][wid] = lsum; }
    __syncthreads();
    if (tid == 0) {   // fixed order over the wavefronts
        double mn = red[0][0], mx = red[1][0], s1 = red[2][0], s2 = red[3][0], s3 = red[4][0];
        for (int w = 1; w < 4; ++w) { mn = fmin(mn, red[0][w]); mx = fmax(mx, red[1][w]); s1 += red[2][w]; s2 += red[3][w]; s3 += red[4][w]; }
        double* o = a.part + 5 * blockIdx.x;
        o[0] = mn; o[1] = mx; o[2] = s1; o[3] = s2; o[4] = s3;
    }
}

#define SEL_BATCH 8
#define HIST_BATCH 16           // histogram rows per thread and round: one round up to 64 key workgroups (65,536 candidates)
#ifndef SCAN_BATCH
#define SCAN_BATCH 10           // 16-byte loads per thread and round of the selected-bin scan: one round up to 40,960 candidates
                                // (12 spills in sampler_emit_k: 1024 threads leave 128 registers per lane)
#endif
#define CAND_MAX 16384          // candidates of the selected top-byte bin kept in LDS (64 KiB)
#define EMIT_BLOCK 1024
#define RANK_MAX 320            // candidate lists up to this length are ranked directly (<= five wavefronts, <= 320 LDS reads each)

// suffix-scan the 256-bin histogram in one wavefront and pick the digit whose suffix count crosses kk
__device__ __forceinline__ void pick_digit(const int* hist, int lane, uint32_t prefix, int shift, uint32_t* s_prefix,
                                           int* s_kk) {
    const int b0 = 252 - 4 * lane;                   // lane 0 owns the TOP four bins
    const int h0 = hist[b0 + 3], h1 = hist[b0 + 2], h2 = hist[b0 + 1], h3 = hist[b0];
    const int local = h0 + h1 + h2 + h3;
    const int above = wave_incl_scan(local) - local; // elements in bins above this lane's four
    const int kk = *s_kk;
    int run = above;
    const int hs[4] = {h0, h1, h2, h3};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int nxt = run + hs[q];
        if (run < kk && nxt >= kk) {                 // exactly one (lane, q) satisfies this
            *s_prefix = prefix | ((uint32_t)(b0 + 3 - q) << shift);
            *s_kk = kk - run;
        }
        run = nxt;
    }
}

// Stage 2 (one workgroup): the order key T of the k-th largest candidate and how many candidates equal to T are
// taken.  Radix pass 1 comes from sampler_keys_k's histogram; the candidates of the selected top-byte bin are
// collected into LDS with one scan and passes 2-4 run on that short list.  Also finalises the statistics.
// sel[0] = T, sel[1] = take_eq, sel[2] = 1 if every candidate is kept (n <= k).
// The selection itself, by the calling workgroup (1024 threads).  `lead`: this workgroup also publishes the statistics and
// sel[0..2].  Every workgroup that runs it arrives at the same (threshold, take_eq): integer work only.
// `o`: the calling thread's order keys of the LAST scan round, four per element, element u = keys 4 (u BD + tid) .. + 3 (clamped
// to the last quad) — when n <= 4 BD SCAN_BATCH that is the whole array, and sampler_emit_k counts its prefix from these
// registers instead of reading the keys again.
// CM: capacity of the LDS list of the selected bin's keys.  FIRST = false: the caller has found the first-level bin itself
// (top = digit << tshift, kk0 = the place sought inside it) — the statistics and the first level are skipped.
template <int CM = CAND_MAX, bool FIRST = true, int SB = SCAN_BATCH>
__device__ __forceinline__ void threshold_body(SamplerArgs a, int keys_blocks, int keys_threads_dev, uint32_t* __restrict__ sel,
                                               bool lead, uint32_t* T_out, int* take_eq_out, int* keep_all_out,
                                               uint4 (&o)[SB], uint32_t top0 = 0u, int kk00 = 0) {
    __shared__ int hist[256];
    __shared__ uint32_t s_prefix;
    __shared__ int s_kk;
    __shared__ int s_cnt;
    __shared__ int tlds[17];
    __shared__ double pr[4][16];
    __shared__ uint32_t cand[CM];
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    const int BD = blockDim.x;
    const int n = eff_count(a.d_n, a.n_host);
    const int k = a.k;
    // statistics partials of sampler_keys_k, reduced in a fixed order (thread b owns partial b)
    if (FIRST && a.stats && lead) {
        double p_mn = INFINITY, p_mx = -INFINITY, p_s1 = 0.0, p_s2 = 0.0;
        if (tid < keys_blocks) { const double* p = a.part + 5 * tid; p_mn = p[0]; p_mx = p[1]; p_s1 = p[2]; p_s2 = p[3]; }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            p_mn = fmin(p_mn, __shfl_xor(p_mn, d, 64)); p_mx = fmax(p_mx, __shfl_xor(p_mx, d, 64));
        }
        p_s1 = wave_sum_d(p_s1); p_s2 = wave_sum_d(p_s2);
        if (lane == 0) { pr[0][wid] = p_mn; pr[1][wid] = p_mx; pr[2][wid] = p_s1; pr[3][wid] = p_s2; }
        __syncthreads();
        if (tid == 0) {
            double mn = INFINITY, mx = -INFINITY, s1 = 0.0, s2 = 0.0;
            for (int w = 0; w < (BD >> 6); ++w) { mn = fmin(mn, pr[0][w]); mx = fmax(mx, pr[1][w]); s1 += pr[2][w]; s2 += pr[3][w]; }
            if (n <= k) {
                a.stats[0] = 0.f; a.stats[1] = 0.f; a.stats[2] = 0.f; a.stats[3] = 0.f;
            } else {
                const double mean = s1 / (double)n;
                double var = n > 1 ? (s2 - s1 * s1 / (double)n) / (double)(n - 1) : 0.0;   // torch.std_mean: unbiased
                if (var < 0.0) var = 0.0;
                a.stats[0] = (float)mn; a.stats[1] = (float)mx; a.stats[2] = (float)mean; a.stats[3] = (float)sqrt(var);
            }
        }
    }
    if (n <= k) {   // utils.py:31-33: everything was written by sampler_keys_k
        if (tid == 0 && lead) { sel[0] = 0u; sel[1] = 0u; sel[2] = 1u; }
        *T_out = 0u; *take_eq_out = 0; *keep_all_out = 1;
        return;
    }
    __syncthreads();                                   // (the shared words below may still be read from a previous use)
    if (tid == 0) { s_prefix = FIRST ? 0u : top0; s_kk = FIRST ? k : kk00; s_cnt = 0; }
    const int tshift = a.ghist ? 32 - GH_BITS : 24;               // the first level's digit = key >> tshift
    if (!FIRST) {
    } else if (a.ghist) {
        // pass 1 from the draw's ONE histogram: thread t owns the four bins GH_BINS-1-4t .. GH_BINS-4-4t (highest first); a
        // workgroup scan of the per-thread sums gives the number of keys above them; the thread whose bins cross k publishes
        __syncthreads();                                   // (s_kk / s_prefix initialised above)
        int h4[4] = {0, 0, 0, 0};
        const int b0 = GH_BINS - 4 - 4 * tid;              // lowest of this thread's four bins (tid < GH_BINS / 4 <= blockDim: 1024 threads)
        if (b0 >= 0) {
            const uint4 v = *reinterpret_cast<const uint4*>(a.ghist + b0);
            h4[0] = (int)v.w; h4[1] = (int)v.z; h4[2] = (int)v.y; h4[3] = (int)v.x;      // descending bin order
        }
        int tot;
        int run = block_excl_scan(h4[0] + h4[1] + h4[2] + h4[3], tlds, &tot);            // keys in the bins above this thread's
        const int kk0 = k;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int nxt = run + h4[q];
            if (run < kk0 && nxt >= kk0) {                 // exactly one (thread, q)
                s_prefix = (uint32_t)(b0 + 3 - q) << tshift;
                s_kk = kk0 - run;
            }
            run = nxt;
        }
    } else {   // sum the per-workgroup histograms: 4 thread groups x 256 bins, each group a quarter of the workgroups
        const int bin = tid & 255, grp = tid >> 8;
        int h = 0;
        int rows = keys_threads_dev > 0 ? (n + keys_threads_dev - 1) / keys_threads_dev : keys_blocks;   // workgroups beyond this saw no candidate
        rows = rows < keys_blocks ? rows : keys_blocks;                   // (keys_threads_dev <= 0: every workgroup wrote its histogram)
        for (int b0 = grp; b0 < rows; b0 += 4 * HIST_BATCH) {            // HIST_BATCH independent loads in flight
            int v[HIST_BATCH];
#pragma unroll
            for (int u = 0; u < HIST_BATCH; ++u) {
                const int b = b0 + 4 * u;
                v[u] = a.hist0[(b < rows ? b : 0) * 256 + bin];          // unconditional, clamped
            }
#pragma unroll
            for (int u = 0; u < HIST_BATCH; ++u) h += (b0 + 4 * u < rows) ? v[u] : 0;
        }
        if (grp == 0) hist[bin] = h;
        __syncthreads();
        if (grp > 0) atomicAdd(&hist[bin], h);      // integer: order-free
        __syncthreads();
        GRAPES_STAMP(8);
        if (wid == 0) pick_digit(hist, lane, 0u, 24, &s_prefix, &s_kk);           // pass 1
    }
    __syncthreads();
    const uint32_t top = s_prefix;
    // one scan: candidates whose top byte is the selected one (four keys per 16-byte load, SCAN_BATCH loads in flight).
    // A single workgroup tests every key, so the scan is priced in VALU instructions per key: matches are counted per
    // WAVEFRONT (compare -> lane mask -> scalar popcount: one vector instruction per key), the wavefront takes its slots
    // with one LDS atomic, and only key slots with a match (scalar branch on the lane mask) run the few vector
    // instructions that place them.  The list's order is irrelevant: passes 2-4 only histogram it.
    {
        const uint4* ord4 = reinterpret_cast<const uint4*>(a.ord);
        const int n4 = (n + 3) >> 2;
        const uint32_t tb = top >> tshift, never = ~top;      // (`never`: a key whose first-level digit is not the selected one)
        for (int base = 0; base < n4; base += BD * SB) {
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int i = base + u * BD + tid;
                o[u] = ord4[i < n4 ? i : n4 - 1];            // unconditional, clamped
            }
            if (n & 3) {                                      // the one ragged quad: its slots past n never match
#pragma unroll
                for (int u = 0; u < SB; ++u)
                    if (base + u * BD + tid == n4 - 1) {
                        if ((n & 3) < 2) o[u].y = never;
                        if ((n & 3) < 3) o[u].z = never;
                        o[u].w = never;
                    }
            }
            int tot = 0;                                      // uniform over the wavefront
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const bool inq = base + u * BD + tid < n4;
                tot += __popcll(__ballot(inq && (o[u].x >> tshift) == tb)) + __popcll(__ballot(inq && (o[u].y >> tshift) == tb)) +
                       __popcll(__ballot(inq && (o[u].z >> tshift) == tb)) + __popcll(__ballot(inq && (o[u].w >> tshift) == tb));
            }
            if (tot != 0) {
                int wbase = 0;
                if (lane == 0) wbase = atomicAdd(&s_cnt, tot);
                wbase = __builtin_amdgcn_readfirstlane(wbase);
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const bool inq = base + u * BD + tid < n4;
                    const uint32_t kv[4] = {o[u].x, o[u].y, o[u].z, o[u].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const bool match = inq && (kv[c] >> tshift) == tb;
                        const unsigned long long mm = __ballot(match);
                        if (mm != 0ull) {
                            const int p = wbase + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
                            if (match && p < CM) cand[p] = kv[c];
                            wbase += __popcll(mm);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    GRAPES_STAMP(9);
    const int nc = s_cnt;
    const bool in_lds = nc <= CM;
    // A SHORT candidate list (the 12-bit first level leaves ~0.4 % of the keys: ~140 of 37k) is ranked directly: thread i < nc
    // counts the candidates above and equal to its own (nc broadcast reads of LDS), and the one whose rank interval holds the
    // place sought publishes (threshold, number of equal keys taken) — one barrier instead of the three passes' twelve.
    const int kk_bin = s_kk;                                                   // the place sought inside the selected bin (1-based)
    const bool ranked = a.ghist && nc <= RANK_MAX && nc > 0 && !a.norank;
    if (ranked) {
        if (tid < nc) {
            const uint32_t mine = cand[tid];
            int gt = 0, eq = 0;
            for (int j = 0; j < nc; ++j) { const uint32_t o = cand[j]; gt += o > mine ? 1 : 0; eq += o == mine ? 1 : 0; }
            if (gt < kk_bin && kk_bin <= gt + eq) { s_prefix = mine; s_kk = kk_bin - gt; }      // (equal keys write equal values)
        }
        __syncthreads();
    }
    // the remaining bits in passes of <= 8: 8 + 8 + 8 after an 8-bit first level, 8 + 8 + 4 after the 12-bit one
    const int npass = ranked ? 0 : 3;
    for (int pi = 0; pi < npass; ++pi) {                                       // passes 2-4
        const int shift = a.ghist ? (pi == 0 ? 12 : (pi == 1 ? 4 : 0)) : 16 - 8 * pi;
        const int width = (a.ghist && pi == 2) ? 4 : 8;
        const uint32_t dmask = (1u << width) - 1u;
        const uint32_t prefix = s_prefix;
        const uint32_t himask = (shift + width >= 32) ? 0u : (0xffffffffu << (shift + width));
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        if (in_lds) {
            for (int i = tid; i < nc; i += BD) {
                const uint32_t o = cand[i];
                if (((o ^ prefix) & himask) == 0u) atomicAdd(&hist[(o >> shift) & dmask], 1);
            }
        } else {   // degenerate case (a huge bin, e.g. all keys equal): scan the whole array
            for (int base = 0; base < n; base += BD * SEL_BATCH) {
                uint32_t o[SEL_BATCH];
#pragma unroll
                for (int u = 0; u < SEL_BATCH; ++u) {
                    const int i = base + u * BD + tid;
                    o[u] = a.ord[i < n ? i : n - 1];
                }
#pragma unroll
                for (int u = 0; u < SEL_BATCH; ++u) {
                    const bool match = (base + u * BD + tid < n) && ((o[u] ^ prefix) & himask) == 0u;
                    if (__ballot(match) != 0ull) wave_hist_add(hist, match ? (int)((o[u] >> shift) & dmask) : -1, lane);
                }
            }
        }
        __syncthreads();
        if (wid == 0) pick_digit(hist, lane, prefix, shift, &s_prefix, &s_kk);
        __syncthreads();
    }
    GRAPES_STAMP(10);
    if (tid == 0 && lead) { sel[0] = s_prefix; sel[1] = (uint32_t)s_kk; sel[2] = 0u; }
    *T_out = s_prefix; *take_eq_out = s_kk; *keep_all_out = 0;
}

__global__ __launch_bounds__(1024) void sampler_threshold_k(SamplerArgs a, int keys_blocks, int keys_threads_dev, uint32_t* __restrict__ sel) {
    uint32_t T; int te, ka;
    uint4 o[SCAN_BATCH];
    threshold_body(a, keys_blocks, keys_threads_dev, sel, true, &T, &te, &ka, o);
    if (threadIdx.x == 0) sel[3] = 0u;                 // the emit launch's ticket
}

// Stage 3 (many workgroups, EMIT_BLOCK candidates each): position-ordered outputs.  A candidate is kept if its key
// is > T, or == T and it is among the first take_eq such candidates (ties -> lowest positions).  Each workgroup
// first counts the keys > T / == T of the candidates BEFORE it (a re-read of at most n order keys, batched), which
// gives its output base without a separate counting launch; then it writes mask, kept_pos / kept_ids in
// candidate-position order (utils.py:57-60), the Bernoulli log-probs and a log-prob partial sum.  The workgroup
// that finishes last (ticket in sel[3]) adds the partial sums in index order (deterministic), writes the kept
// count and advances the Philox counter.
//
// emit_place: one candidate's outputs once its keep flag and output slot are known, and the workgroup's log-prob partial
// (wavefront butterfly, wavefronts in index order — every draw form sums in this order, so stats[4] is bit-identical).
template <int EB>
__device__ __forceinline__ void emit_place(const SamplerArgs& a, int n, int i, bool keep, int pos, float lsv, float lv, int cand_id,
                                           bool have_cand_id, double* __restrict__ lsum_part, int bid, double* red,
                                           bool have_ext = false, long long eb0 = 0, long long eb1 = 0) {
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    double lp_d = 0.0;
    if (i < n) {
        a.mask[i] = keep ? 1.0f : 0.0f;
        if (keep) {
            a.kept_pos[pos] = i;
            if (a.cand_ids && (a.kept_ids || a.union_ids)) {
                const int cid = have_cand_id ? cand_id : a.cand_ids[i];
                if (a.kept_ids) a.kept_ids[pos] = cid;
                if (a.union_ids) a.union_ids[a.prefix_n + pos] = cid;
                if (a.union_ext && a.rowptr) {
                    if (!have_ext) { eb0 = a.rowptr[cid]; eb1 = a.rowptr[cid + 1]; }
                    *reinterpret_cast<longlong2*>(a.union_ext + 2 * (long long)(a.prefix_n + pos)) = make_longlong2(eb0, eb1);
                }
            }
        }
        const float lp = keep ? lsv : lsv - lv;                  // -BCEWithLogits(l, m)   (utils.py:71)
        if (a.log_prob) a.log_prob[i] = lp;
        lp_d = (double)lp;
    }
    lp_d = wave_sum_d(lp_d);
    if (lane == 0) red[wid] = lp_d;
    lds_barrier();                                   // (`red` only: no wait for the stores above)
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < EB / 64; ++w) t += red[w];
        publish_f64(&lsum_part[bid], t);          // (see common.h: no device-scope fence)
    }
}

// the position-ordered outputs of workgroup `bid` from the threshold (T, take_eq): its output base by counting the keys before it
template <int EB, int SB = SCAN_BATCH>
__device__ __forceinline__ void emit_outputs(const SamplerArgs& a, int n, int bid, uint32_t T, int take_eq, bool have_keys,
                                             uint4 (&ko)[SB], uint32_t o_own, float ls_own, float l_own,
                                             double* __restrict__ lsum_part) {
    __shared__ int lds[17];
    __shared__ double red[16];
    __shared__ int s_gt[16], s_eq[16];
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    // candidates before this workgroup: [0, bid * EB)
    const int before = bid * EB;
    int c_gt = 0, c_eq = 0;
    {
        const int b4 = before >> 2;                                     // `before` is a multiple of EB (and of 4)
        if (have_keys) {    // the scan's registers: no second read of the keys.  b4 is a multiple of 64: a quad slot lies
                            // before this workgroup for a whole wavefront or not at all, and counts are per wavefront
            const int w0 = __builtin_amdgcn_readfirstlane(tid & ~63);
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                if (u * EB + w0 < b4) {
                    c_gt += __popcll(__ballot(ko[u].x > T)) + __popcll(__ballot(ko[u].y > T)) + __popcll(__ballot(ko[u].z > T)) + __popcll(__ballot(ko[u].w > T));
                    c_eq += __popcll(__ballot(ko[u].x == T)) + __popcll(__ballot(ko[u].y == T)) + __popcll(__ballot(ko[u].z == T)) + __popcll(__ballot(ko[u].w == T));
                }
            }
        } else {
            const uint4* ord4 = reinterpret_cast<const uint4*>(a.ord);
#pragma unroll 1
            for (int base = 0; base < b4; base += EB * SEL_BATCH) {
                uint4 o[SEL_BATCH];
#pragma unroll
                for (int u = 0; u < SEL_BATCH; ++u) {
                    const int i = base + u * EB + tid;
                    o[u] = ord4[i < b4 ? i : 0];                     // unconditional, clamped
                }
#pragma unroll
                for (int u = 0; u < SEL_BATCH; ++u) {
                    const bool in = base + u * EB + tid < b4;
                    const int g4 = (o[u].x > T) + (o[u].y > T) + (o[u].z > T) + (o[u].w > T);
                    const int e4 = (o[u].x == T) + (o[u].y == T) + (o[u].z == T) + (o[u].w == T);
                    c_gt += in ? g4 : 0;
                    c_eq += in ? e4 : 0;
                }
            }
            c_gt = wave_incl_scan(c_gt); c_eq = wave_incl_scan(c_eq);      // lane 63 holds the wavefront totals
            c_gt = __shfl(c_gt, 63, 64); c_eq = __shfl(c_eq, 63, 64);
        }
    }
    if (lane == 0) { s_gt[wid] = c_gt; s_eq[wid] = c_eq; }
    __syncthreads();
    GRAPES_STAMP(2);
    int gt_before = 0, eq_before = 0;
#pragma unroll
    for (int w = 0; w < EB / 64; ++w) { gt_before += s_gt[w]; eq_before += s_eq[w]; }
    const int i = bid * EB + tid;
    const bool gt = i < n && o_own > T, eq = i < n && o_own == T;
    // ONE workgroup scan for both ranks (each count <= 1024: 16 bits apiece).  Equal keys are taken in position order
    // until take_eq of them are in: the kept equal keys before this thread number min(eq before it, take_eq).
    int tot;
    const int packed = block_excl_scan((gt ? 1 : 0) | (eq ? 1 << 16 : 0), lds, &tot);
    const int eq_rank = eq_before + (packed >> 16);
    const bool keep = gt || (eq && eq_rank < take_eq);
    const int pos = gt_before + (packed & 0xffff) + (eq_rank < take_eq ? eq_rank : take_eq);
    emit_place<EB>(a, n, i, keep, pos, ls_own, l_own, 0, false, lsum_part, bid, red);
}

// the draw's tail when no later launch finishes it: the last of `arrivals` workgroups to take the ticket adds the partial sums in
// index order, writes the counts and the Philox advance, and puts ticket / histogram (/ the fused form's barrier words) back to zero
template <int EB>
__device__ __forceinline__ void emit_tail(const SamplerArgs& a, int n, bool keep_all, int keys_blocks, const double* __restrict__ lsum_part,
                                          uint32_t* ticket, unsigned arrivals, int hist_words, int stats_blocks) {
    __shared__ double red[16];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    if (tid == 0) {
        const unsigned t = atomicAdd(ticket, 1u);
        s_last = (t == arrivals - 1) ? 1 : 0;
    }
    __syncthreads();
    GRAPES_STAMP(4);
    if (!s_last) return;
    const int nb = keep_all ? keys_blocks : (n + EB - 1) / EB;
    const double* parts = keep_all ? a.part + 4 : lsum_part;
    const int pstride = keep_all ? 5 : 1;
    // 1024 VIRTUAL threads (thread b owns the partials b, b + 1024, ...), butterfly inside a virtual wavefront, virtual wavefronts in
    // index order — draw_finish_body's order (common.h), whatever this workgroup's size: stats[4] is the same bits either way
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 1024 / EB; ++q) {
        double sacc = 0.0;
        for (int bb = tid + EB * q; bb < nb; bb += 1024)
            sacc += __longlong_as_double(__hip_atomic_load((const long long*)(parts + (size_t)bb * pstride), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
        sacc = wave_sum_d(sacc);
        if (lane == 0) red[wid + (EB / 64) * q] = sacc;
    }
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w];
        if (a.stats) { a.stats[4] = (float)t; a.stats[5] = keep_all ? 0.f : 1.f; }
        if (a.d_kept_count) *a.d_kept_count = keep_all ? n : a.k;    // exactly k are drawn when n > k (utils.py:44)
        if (a.d_union_count) *a.d_union_count = a.prefix_n + (keep_all ? n : a.k);
        if (!keep_all && a.d_offset && a.mode == 0 && a.uniforms == nullptr) {
            const uint64_t off = *a.d_offset;
            *a.d_offset = off + (uint64_t)((n + 3) >> 2);
        }
        *ticket = 0u;                                                // ticket ready for the next draw
    }
    if (a.ghist)     // every workgroup has read the draw's histogram before it took its ticket: back to zero for the next draw
        for (int b = tid; b < hist_words; b += EB) if (a.ghist + b != ticket) a.ghist[b] = 0u;
    if (stats_blocks > 0 && !keep_all && a.stats) draw_stats_final<EB>(a.part, stats_blocks, n, a.stats);      // (the one-launch draw)
    GRAPES_STAMP(5);
}

// what the launch's first workgroup writes as soon as it knows the selection when the tail is deferred (the next launches read it)
__device__ __forceinline__ void emit_defer_words(const SamplerArgs& a, uint32_t* __restrict__ sel, int n, bool keep_all) {
    sel[2] = keep_all ? 1u : 0u;                                 // (for the finishing workgroup of the caller's next launch)
    if (a.stats) a.stats[5] = keep_all ? 0.f : 1.f;
    if (a.d_kept_count) *a.d_kept_count = keep_all ? n : a.k;    // exactly k are drawn when n > k (utils.py:44)
    if (a.d_union_count) *a.d_union_count = a.prefix_n + (keep_all ? n : a.k);
    if (!keep_all && a.d_offset && a.mode == 0 && a.uniforms == nullptr) {      // (every workgroup has consumed the counter)
        const uint64_t off = *a.d_offset;
        *a.d_offset = off + (uint64_t)((n + 3) >> 2);
    }
}

__global__ __launch_bounds__(EMIT_BLOCK) void sampler_emit_k(SamplerArgs a, int keys_blocks, uint32_t* __restrict__ sel,
                                                             double* __restrict__ lsum_part, int select_here,
                                                             int keys_threads_dev) {
    const int n = eff_count(a.d_n, a.n_host);
    const int tid = threadIdx.x;
    // everything this thread reads that does not depend on the selection goes out first, together with the selection
    // words themselves: one round trip instead of four dependent ones
    uint32_t sel0, sel1, sel2;
    GRAPES_STAMP(0);
    const int i_own = blockIdx.x * EMIT_BLOCK + tid;
    const int ic_own = n > 0 ? (i_own < n ? i_own : n - 1) : 0;
    uint32_t o_own = 0u; float ls_own = 0.f, l_own = 0.f;
    if (n > 0 && (int)blockIdx.x * EMIT_BLOCK < n) {
        o_own = a.ord[ic_own];
        ls_own = a.ls[ic_own];
        l_own = a.logits[a.logit_index ? a.logit_index[ic_own] : ic_own];
    }
    if (blockIdx.x == 0 && a.union_ids && a.prefix_ids)
        for (int i = tid; i < a.prefix_n; i += EMIT_BLOCK) a.union_ids[i] = a.prefix_ids[i];
    if (blockIdx.x == 0 && a.union_ext && a.prefix_ext)
        for (int i = tid; i < 2 * a.prefix_n; i += EMIT_BLOCK) a.union_ext[i] = a.prefix_ext[i];
    uint4 ko[SCAN_BATCH];
    bool have_keys = false;          // ko holds every order key this thread scanned (uniform over the workgroup)
    if (select_here) {   // ONE launch for threshold + emit: every live workgroup works the (integer) selection out itself
        uint32_t T = 0u; int te = 0, ka = 0;
        if ((int)blockIdx.x * EMIT_BLOCK < n || blockIdx.x == 0) {
            threshold_body(a, keys_blocks, keys_threads_dev, sel, blockIdx.x == 0, &T, &te, &ka, ko);
            have_keys = !ka && ((n + 3) >> 2) <= EMIT_BLOCK * SCAN_BATCH;
        }
        sel0 = T; sel1 = (uint32_t)te; sel2 = (uint32_t)ka;
    } else {
        sel0 = sel[0]; sel1 = sel[1]; sel2 = sel[2];
    }
    GRAPES_STAMP(1);
    const bool keep_all = sel2 != 0u;
    if (a.defer_finish && blockIdx.x == 0 && tid == 0) emit_defer_words(a, sel, n, keep_all);
    if (!keep_all && (int)blockIdx.x * EMIT_BLOCK < n)
        emit_outputs<EMIT_BLOCK>(a, n, (int)blockIdx.x, sel0, (int)sel1, have_keys, ko, o_own, ls_own, l_own, lsum_part);
    GRAPES_STAMP(3);
    if (a.defer_finish) return;              // the sum of the partials and the histogram's reset ride in the caller's next launch
    emit_tail<EMIT_BLOCK>(a, n, keep_all, keys_blocks, lsum_part, sel + 3, gridDim.x, a.ghist ? GH_BINS : 0, 0);
}

// ---------------------------------------------------------------------------- the draw in ONE launch (round 5)
// sampler_keys_k + sampler_emit_k were two launches of dependent round trips: the second one re-read every candidate's order key,
// log-sigmoid and logit, and EVERY workgroup scanned ALL keys for the selected first-level bin (5.8 of its 15 us).  Here a thread
// keeps its candidate's key in registers across ONE grid barrier (a workgroup per DRAW_BLOCK candidates, all resident: the
// launcher takes this form up to DRAW_MAX_WG workgroups of capacity, two per compute unit):
//   A  keys (portable math); a 12-bit histogram of the workgroup's OWN keys (LDS) gives the workgroup its CUT: the lowest
//      first-level digit such that at most DRAW_LIST of its keys lie at or above it.  Those keys — everything the workgroup can
//      contribute to the k largest unless it holds far more than its share — are published in position order with (cut, count).
//   -- the barrier --
//   B  ONE round trip brings every workgroup all (cut, count) pairs and all lists.  A histogram of the LISTED keys (LDS) gives
//      the bin b of the k-th largest listed key and the place kk sought inside it; when every cut is <= b the lists hold EVERY
//      key of the draw at or above bin b, so b, kk are the draw's own: per workgroup the number above b and the keys in b, in
//      position order ((workgroup, slot) order).  The <= RANK_MAX in-bin keys are ranked (ties go to the lowest positions, as
//      before), which gives the threshold, each workgroup's output base and its own in-bin candidates' fate; then the
//      position-ordered outputs (emit_place).  No table in global memory: the second version added every workgroup's non-empty
//      bins to the draw-wide table of the two-launch form — 25k agent-scope atomics on 16 KB, 7 us of the launch.
// Everything that crosses workgroups inside the launch is an agent-scope atomic / store / load (performed at the memory side), so
// the barrier is: wait for the wavefront's own memory operations, workgroup barrier, ONE arrival atomic, poll — no cache write-back
// or invalidate (the first version fenced per wavefront and took two barriers: 29 - 44 us per draw; profiles/r05_index_phase_stamps.txt).
// What a workgroup only owes the END of the draw — log-sigmoids, entropies, the statistics partials — runs between its arrival and
// its poll.  The statistics themselves (stats[0..3]) are formed with the log-prob sum by whoever ends the draw (the tail's last
// workgroup, or the caller's next launch: grapes_draw_finish_args.stats_blocks).
// A draw the short form cannot hold (more than RANK_MAX keys in the selected bin — greedy draws over saturated probabilities,
// constant keys — or a workgroup whose cut lies above b) takes the scan of sampler_emit_k inside the same launch: every
// workgroup counts ALL keys (a.ord) into its LDS table for the first level, then threshold_body's list + passes and
// emit_outputs: same sets, masks and log-probs, bit for bit.
#define DRAW_BLOCK 512
#define DRAW_LIST 32
#define DRAW_MAX_WG 2048           // grid: workgroups of CAPACITY (those beyond the live candidates leave at once)
#define DRAW_MAX_LIVE 384          // live workgroups that meet at the barrier, all resident: two fit a compute unit (512), with a margin;
                                   // a larger draw is worked by the first DRAW_MAX_LIVE, several blocks of candidates each (draw_many_blocks)
#define DRAW_SCAN_BATCH 4          // 16-byte key loads per thread and round of the scan form here (the short form's registers come first)
#define DRAW_CAND 4096             // the scan form's LDS list here (two workgroups per compute unit: 16 KB instead of 64)
#define GH_TAIL 64                 // words behind the histogram, zero at rest: [0] barrier arrivals, [1] the eager tail's ticket
struct DrawFused { uint32_t* bar; uint32_t* ticket; uint32_t* pubs; uint32_t* lists; double* lsum_part; uint32_t* sel; };

__device__ __forceinline__ void draw_arrive(uint32_t* bar) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wavefront's atomics / agent-scope stores have been performed
    __syncthreads();
    if (threadIdx.x == 0) (void)atomicAdd(bar, 1u);
}
__device__ __forceinline__ void draw_wait(uint32_t* bar, unsigned target, uint32_t* flag) {
    if (threadIdx.x == 0) {
        bool ok = false;
        for (int spin = 0; spin < GRAPES_SYNC_SPIN_LIMIT; ++spin) {
            if (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) *flag = 1u;                                 // (no launch waits forever; the draw is then invalid: sel[5])
    }
    lds_barrier();                                           // (thread 0 has seen the count; nobody waits for stores in flight)
}
// 16-byte agent-scope loads (the compiler's atomic loads stop at 8 bytes), all in flight together: 32 bytes at p, 16 at q0 and q1
typedef uint32_t grapes_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ld_agent_x4x2(const uint32_t* q0, const uint32_t* q1, uint4& w0, uint4& w1) {
    grapes_u32x4 r2, r3;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(r2), "=&v"(r3) : "v"(q0), "v"(q1) : "memory");
    w0 = make_uint4(r2.x, r2.y, r2.z, r2.w); w1 = make_uint4(r3.x, r3.y, r3.z, r3.w);
}
__device__ __forceinline__ uint4 ld_agent_x4(const uint32_t* p) {
    grapes_u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p) : "memory");
    return make_uint4(r.x, r.y, r.z, r.w);
}
// exclusive scan over the workgroup on ONE barrier: `buf` (NW ints) must not be in use by a scan that some wavefront may still read
template <int NW>
__device__ __forceinline__ int block_scan_1b(int v, int* buf, int* total) {
    const int lane = lane_id(), wid = threadIdx.x >> 6;
    const int incl = wave_incl_scan(v);
    if (lane == 63) buf[wid] = incl;
    lds_barrier();                                           // (LDS only: __syncthreads would wait for every store / atomic in flight)
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { const int x = buf[w]; tot += x; base += w < wid ? x : 0; }
    *total = tot;
    return base + incl - v;
}

// A draw of more than DRAW_MAX_LIVE blocks of candidates (n > 196,608): the first DRAW_MAX_LIVE workgroups take the blocks
// b = bid, bid + DRAW_MAX_LIVE, ... — keys, log-sigmoids to memory (agent-scope stores), statistics partials per BLOCK — meet at the
// barrier, then every workgroup counts all keys into its LDS table for the first level (as the scan form of the short draw does)
// and emits its blocks with sampler_emit_k's prefix counts over a.ord.  The two-launch draw in one launch: same results, bit for
// bit; correct for any n the grid covers, not fast.
__device__ __forceinline__ void draw_many_blocks(const SamplerArgs& a, const DrawFused& f, int n, int live, uint64_t offset, int* hist, int* sbuf) {
    constexpr int NW = DRAW_BLOCK / 64, BPT = GH_BINS / DRAW_BLOCK;
    __shared__ double mb_red[4][NW];
    __shared__ int mb_bin, mb_kk;
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6, bid = blockIdx.x;
    for (int b = bid; b < live; b += DRAW_MAX_LIVE) {
        const int i = b * DRAW_BLOCK + tid;
        float pmin = INFINITY, pmax = -INFINITY;
        double esum = 0.0, esq = 0.0;
        if (i < n) {
            const float l = a.logits[a.logit_index ? a.logit_index[i] : i];
            const float p = p_sigmoid(l);
            float key;
            if (a.mode == 1) {
                key = p;                                                   // eval.py:126-127
            } else {
                const float r = a.uniforms ? a.uniforms[i] : philox_uniform_at(a.seed, offset, i);
                key = p_logf(p) + p_gumbel(r);                             // utils.py:42
            }
            __hip_atomic_store(a.ord + i, order_key_of(key, a.mode), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<uint32_t*>(a.ls) + i, __float_as_uint(log_sigmoid_f(l)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.keys_out) a.keys_out[i] = key;
            if (a.stats) {
                pmin = p; pmax = p;
                float ent = -(p * log2f(p) + (1.0f - p) * log2f(1.0f - p));   // utils.py:47
                if (ent != ent) ent = 0.0f;                                   // utils.py:52-54
                esum = (double)ent; esq = (double)ent * (double)ent;
            }
        }
        if (a.stats) {   // the block's statistics partial, in the order of the short draw's (wavefront butterflies, wavefronts in index order)
            pmin = wave_min(pmin); pmax = wave_max(pmax); esum = wave_sum_d(esum); esq = wave_sum_d(esq);
            lds_barrier();
            if (lane == 0) { mb_red[0][wid] = pmin; mb_red[1][wid] = pmax; mb_red[2][wid] = esum; mb_red[3][wid] = esq; }
            lds_barrier();
            if (tid == 0) {
                double mn = INFINITY, mx = -INFINITY, s1 = 0.0, s2 = 0.0;
                for (int w = 0; w < NW; ++w) { mn = fmin(mn, mb_red[0][w]); mx = fmax(mx, mb_red[1][w]); s1 += mb_red[2][w]; s2 += mb_red[3][w]; }
                long long* o = reinterpret_cast<long long*>(a.part + 5 * b);
                __hip_atomic_store(o + 0, __double_as_longlong(mn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(o + 1, __double_as_longlong(mx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(o + 2, __double_as_longlong(s1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(o + 3, __double_as_longlong(s2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    draw_arrive(f.bar);
    draw_wait(f.bar, (unsigned)DRAW_MAX_LIVE, f.sel + 5);
    if (bid == 0 && tid == 0 && a.defer_finish) emit_defer_words(a, f.sel, n, false);    // (every workgroup has read the Philox counter)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // the first level over ALL keys, by every workgroup for itself (integer work: the same bin everywhere)
    if (tid == 0) { mb_bin = 0; mb_kk = a.k; }
    lds_barrier();                                                     // (the table was cleared at the top of the launch)
    for (int j = tid; j < n; j += DRAW_BLOCK) wave_hist_add(hist, (int)(a.ord[j] >> (32 - GH_BITS)), lane);     // (ballots see the active lanes only)
    lds_barrier();
    {
        const int b0 = GH_BINS - BPT - BPT * tid;
        const int4 ha = *reinterpret_cast<const int4*>(hist + b0), hb = *reinterpret_cast<const int4*>(hist + b0 + 4);
        const int h[BPT] = {hb.w, hb.z, hb.y, hb.x, ha.w, ha.z, ha.y, ha.x};
        int tot;
        int run = block_scan_1b<NW>(((h[0] + h[1]) + (h[2] + h[3])) + ((h[4] + h[5]) + (h[6] + h[7])), sbuf, &tot);
#pragma unroll
        for (int q = 0; q < BPT; ++q) {
            const int nxt = run + h[q];
            if (run < a.k && nxt >= a.k) { mb_bin = b0 + BPT - 1 - q; mb_kk = a.k - run; }
            run = nxt;
        }
    }
    lds_barrier();
    uint4 ko[DRAW_SCAN_BATCH];
    uint32_t T = 0u; int te = 0, ka = 0;
    threshold_body<DRAW_CAND, false, DRAW_SCAN_BATCH>(a, live, DRAW_BLOCK, f.sel, bid == 0, &T, &te, &ka, ko, (uint32_t)mb_bin << (32 - GH_BITS), mb_kk);
    for (int b = bid; b < live; b += DRAW_MAX_LIVE) {
        const int i = b * DRAW_BLOCK + tid, ic = i < n ? i : n - 1;
        const uint32_t o_own = a.ord[ic];
        const float ls_own = a.ls[ic], l_own = a.logits[a.logit_index ? a.logit_index[ic] : ic];
        __syncthreads();                                               // (emit_outputs' shared words: the previous block's readers are through)
        emit_outputs<DRAW_BLOCK, DRAW_SCAN_BATCH>(a, n, b, T, te, false, ko, o_own, ls_own, l_own, f.lsum_part);
    }
    if (a.defer_finish) return;
    emit_tail<DRAW_BLOCK>(a, n, false, live, f.lsum_part, f.ticket, (unsigned)DRAW_MAX_LIVE, GH_BINS + GH_TAIL, live);
}

__global__ __launch_bounds__(DRAW_BLOCK, 4) void sampler_draw_k(SamplerArgs a, DrawFused f) {
    constexpr int NW = DRAW_BLOCK / 64;                       // wavefronts
    constexpr int BPT = GH_BINS / DRAW_BLOCK;                 // histogram bins per thread (8)
    constexpr int QPW = DRAW_LIST / 4;                        // 16-byte quads of a workgroup's list (8)
    static_assert(BPT == 8 && QPW == 8, "sampler_draw_k: eight bins per thread, eight list quads per workgroup");
    __shared__ double red5[5][NW];
    __shared__ int hist[GH_BINS];
    __shared__ int sb[6][NW];                                 // wavefront totals of the one-barrier scans
    __shared__ double red[16];
    __shared__ int s_cut, s_lcnt, s_bin, s_kk, s_nc, s_T_kk, s_kb, s_slow, s_tie;
    __shared__ uint32_t s_T;
    __shared__ uint4 cand4[RANK_MAX / 4 + 1];
    __shared__ int s_offin[DRAW_MAX_LIVE], s_hib[DRAW_MAX_LIVE], s_cin[DRAW_MAX_LIVE], s_chi[DRAW_MAX_LIVE], s_cnt[DRAW_MAX_LIVE];
    __shared__ int s_keep[DRAW_LIST];
    __shared__ int s_gt[RANK_MAX], s_eq[RANK_MAX];
    uint32_t* cand = reinterpret_cast<uint32_t*>(cand4);
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6, bid = blockIdx.x;
    GRAPES_STAMP_NW(0);
    // the candidate count, the Philox counter, this thread's logit index and candidate id leave together (inside the CAPACITY: the
    // live count is not known yet); the LDS table is cleared under them
    const int i = bid * DRAW_BLOCK + tid;
    const int idx = (a.logit_index && i < a.n_host) ? a.logit_index[i] : i;
    const bool want_id = a.cand_ids && (a.kept_ids || a.union_ids);
    const int cid = (want_id && i < a.n_host) ? a.cand_ids[i] : 0;
    uint64_t offset = a.offset;
    if (a.d_offset) offset = *a.d_offset;
    const int n_raw = a.d_n ? *a.d_n : a.n_host;
#pragma unroll
    for (int q = 0; q < BPT; ++q) hist[tid + q * DRAW_BLOCK] = 0;
    if (tid == 0) { s_kb = 0; s_slow = 0; s_cut = 0; s_lcnt = 0; s_tie = 0; }
    const int n = n_raw < a.n_host ? (n_raw < 0 ? 0 : n_raw) : a.n_host;
    const int live = (n + DRAW_BLOCK - 1) / DRAW_BLOCK;
    const bool keep_all = n <= a.k;                                    // utils.py:31-33
    GRAPES_STAMP_NW(7);                                                // (the count has arrived)
    if (bid >= live && bid != 0 && !keep_all) return;                  // (a keep-all draw: every workgroup leaves its partial)
    if (!keep_all && live > DRAW_MAX_LIVE && bid >= DRAW_MAX_LIVE) return;
    if (bid == 0 && a.union_ids && a.prefix_ids)
        for (int j = tid; j < a.prefix_n; j += DRAW_BLOCK) a.union_ids[j] = a.prefix_ids[j];
    if (bid == 0 && a.union_ext && a.prefix_ext)
        for (int j = tid; j < 2 * a.prefix_n; j += DRAW_BLOCK) a.union_ext[j] = a.prefix_ext[j];
    float l = 0.f;
    if (i < n) l = a.logits[a.logit_index ? idx : i];
    if (keep_all) {                                                    // everything is kept: no selection, no barrier
        double lsum = 0.0;
        if (i < n) {
            const float lsg = log_sigmoid_f(l);
            a.mask[i] = 1.0f;
            a.kept_pos[i] = i;
            if (a.kept_ids && a.cand_ids) a.kept_ids[i] = cid;
            if (a.union_ids && a.cand_ids) a.union_ids[a.prefix_n + i] = cid;
            if (a.union_ext && a.rowptr && a.cand_ids)
                *reinterpret_cast<longlong2*>(a.union_ext + 2 * (long long)(a.prefix_n + i)) = make_longlong2(a.rowptr[cid], a.rowptr[cid + 1]);
            if (a.log_prob) a.log_prob[i] = lsg;
            lsum = (double)lsg;
        }
        lsum = wave_sum_d(lsum);
        if (lane == 0) red5[4][wid] = lsum;
        lds_barrier();
        if (tid == 0) {
            double s3 = 0.0;
            for (int w = 0; w < NW; ++w) s3 += red5[4][w];
            double* o = a.part + 5 * bid;
            o[0] = (double)INFINITY; o[1] = -(double)INFINITY; o[2] = 0.0; o[3] = 0.0;
            publish_f64(&o[4], s3);
        }
        if (bid == 0 && tid == 0) {
            f.sel[0] = 0u; f.sel[1] = 0u;
            if (a.stats) { a.stats[0] = 0.f; a.stats[1] = 0.f; a.stats[2] = 0.f; a.stats[3] = 0.f; }
            if (a.defer_finish) emit_defer_words(a, f.sel, n, true);
        }
        if (a.defer_finish) return;
        emit_tail<DRAW_BLOCK>(a, n, true, (int)gridDim.x, f.lsum_part, f.ticket, gridDim.x, GH_BINS + GH_TAIL, 0);
        return;
    }
    if (live > DRAW_MAX_LIVE) {                                        // more blocks of candidates than workgroups that may wait for each other
        draw_many_blocks(a, f, n, live, offset, hist, sb[0]);
        return;
    }
    // ---- A: the key, the histogram, this workgroup's list
    lds_barrier();                                                   // (the cleared table)
    uint32_t ok = 0u;
    float p = 0.f;
    int digit = -1;
    if (i < n) {
        p = p_sigmoid(l);
        float key;
        if (a.mode == 1) {
            key = p;                                                   // eval.py:126-127
        } else {
            const float r = a.uniforms ? a.uniforms[i] : philox_uniform_at(a.seed, offset, i);
            key = p_logf(p) + p_gumbel(r);                             // utils.py:42
        }
        ok = order_key_of(key, a.mode);
        __hip_atomic_store(a.ord + i, ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // (read again only by the scan form below)
        digit = (int)(ok >> (32 - GH_BITS));
        if (a.keys_out) a.keys_out[i] = key;
    }
    GRAPES_STAMP_NW(8);                                                // (logit arrived, key computed)
    wave_hist_add(hist, digit, lane);
    lds_barrier();
    GRAPES_STAMP_NW(9);                                                // (the workgroup's table is complete)
    const int b0 = GH_BINS - BPT - BPT * tid;                          // lowest of this thread's eight bins (thread 0: the top eight)
    {
        const int4 ha = *reinterpret_cast<const int4*>(hist + b0), hb = *reinterpret_cast<const int4*>(hist + b0 + 4);
        const int h[BPT] = {hb.w, hb.z, hb.y, hb.x, ha.w, ha.z, ha.y, ha.x};                       // descending bin order
        // the cut: walking down from the top bin, the first bin whose keys would make the list longer than DRAW_LIST stays out
        int tot;
        int run = block_scan_1b<NW>(((h[0] + h[1]) + (h[2] + h[3])) + ((h[4] + h[5]) + (h[6] + h[7])), sb[0], &tot);
#pragma unroll
        for (int q = 0; q < BPT; ++q) {
            const int nxt = run + h[q];
            if (run <= DRAW_LIST && nxt > DRAW_LIST) { s_cut = b0 + BPT - q; s_lcnt = run; }       // exactly one (thread, q), or none:
            run = nxt;
        }
        if (tid == 0 && tot <= DRAW_LIST) { s_cut = 0; s_lcnt = tot; }                             // ... every key fits
    }
    lds_barrier();
    GRAPES_STAMP_NW(10);                                               // (own cut found)
    {
        const int cut = s_cut;
        const bool mine = i < n && digit >= cut;
        const unsigned long long mm = __ballot(mine);
        if (lane == 0) sb[1][wid] = __popcll(mm);
        lds_barrier();
        int r = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
#pragma unroll
        for (int w = 0; w < NW; ++w) r += w < wid ? sb[1][w] : 0;
        if (mine) __hip_atomic_store(f.lists + bid * DRAW_LIST + r, ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (r < count <= DRAW_LIST)
        if (tid == 0) __hip_atomic_store(f.pubs + bid, (uint32_t)cut | ((uint32_t)s_lcnt << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    GRAPES_STAMP_NW(1);
    draw_arrive(f.bar);
    // (while the arrivals travel) what only the end of the draw needs: the log-sigmoid, the entropy, the statistics partial
    float lsg = 0.f;
    if (i < n) lsg = log_sigmoid_f(l);
    // (every candidate's row extents for the expansion that follows: two loads that travel while the barrier is polled)
    const bool with_ext = a.union_ext && a.rowptr && want_id;
    long long eb0 = 0, eb1 = 0;
    if (with_ext && i < n) { eb0 = a.rowptr[cid]; eb1 = a.rowptr[cid + 1]; }
#pragma unroll
    for (int q = 0; q < BPT; ++q) hist[tid + q * DRAW_BLOCK] = 0;       // (every wavefront has read its bins: the arrival's barrier) for the listed keys
    if (a.stats) {
        float pmin = INFINITY, pmax = -INFINITY;
        double esum = 0.0, esq = 0.0;
        if (i < n) {
            pmin = p; pmax = p;
            float ent = -(p * log2f(p) + (1.0f - p) * log2f(1.0f - p));   // utils.py:47
            if (ent != ent) ent = 0.0f;                                   // utils.py:52-54
            esum = (double)ent; esq = (double)ent * (double)ent;
        }
        pmin = wave_min(pmin); pmax = wave_max(pmax); esum = wave_sum_d(esum); esq = wave_sum_d(esq);
        if (lane == 0) { red5[0][wid] = pmin; red5[1][wid] = pmax; red5[2][wid] = esum; red5[3][wid] = esq; }
    }
    draw_wait(f.bar, (unsigned)live, f.sel + 5);
    GRAPES_STAMP_NW(2);
    // ---- B: ONE round trip: every (cut, count) pair, every list
    if (bid == 0 && tid == 0 && a.defer_finish) emit_defer_words(a, f.sel, n, false);    // (every workgroup has read the Philox counter)
    if (a.stats && tid == 0) {   // this workgroup's statistics partial (wavefronts in index order), for whoever ends the draw
        double mn = INFINITY, mx = -INFINITY, s1 = 0.0, s2 = 0.0;
        for (int w = 0; w < NW; ++w) { mn = fmin(mn, red5[0][w]); mx = fmax(mx, red5[1][w]); s1 += red5[2][w]; s2 += red5[3][w]; }
        long long* o = reinterpret_cast<long long*>(a.part + 5 * bid);
        __hip_atomic_store(o + 0, __double_as_longlong(mn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 1, __double_as_longlong(mx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 2, __double_as_longlong(s1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 3, __double_as_longlong(s2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int nquad = live * QPW;                                      // list quads; quad e belongs to workgroup e / QPW
    const uint32_t pub = __hip_atomic_load(f.pubs + (tid < live ? tid : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint4 q0, q1;
    ld_agent_x4x2(f.lists + 4 * (tid < nquad ? tid : 0), f.lists + 4 * (tid + DRAW_BLOCK < nquad ? tid + DRAW_BLOCK : 0), q0, q1);
    if (tid < live) s_cnt[tid] = (int)(pub >> 16);
    lds_barrier();
    // the 12-bit table of the LISTED keys (cleared while the barrier was polled); quads 0 .. 1023 (128 live workgroups) came with
    // the round trip above, a larger draw fetches the rest one round at a time
    auto quad_of = [&](int u, int e) -> uint4 { return u == 0 ? q0 : (u == 1 ? q1 : ld_agent_x4(f.lists + 4 * (e < nquad ? e : 0))); };
    for (int u = 0; u * DRAW_BLOCK < nquad; ++u) {                      // (uniform over the workgroup)
        const int e = tid + u * DRAW_BLOCK;
        const uint4 q = quad_of(u, e);
        const int s0 = (e % QPW) * 4, cw = e < nquad ? s_cnt[e / QPW] : 0;
        const uint32_t kv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) wave_hist_add(hist, s0 + c < cw ? (int)(kv[c] >> (32 - GH_BITS)) : -1, lane);
    }
    lds_barrier();
    {
        const int4 ha = *reinterpret_cast<const int4*>(hist + b0), hb = *reinterpret_cast<const int4*>(hist + b0 + 4);
        const int h[BPT] = {hb.w, hb.z, hb.y, hb.x, ha.w, ha.z, ha.y, ha.x};                       // descending bin order
        int tot;
        int run = block_scan_1b<NW>(((h[0] + h[1]) + (h[2] + h[3])) + ((h[4] + h[5]) + (h[6] + h[7])), sb[2], &tot);   // listed keys in the bins above this thread's
        if (tid == 0 && tot < a.k) s_slow = 1;                         // (fewer than k keys listed: some workgroup's list stops too early)
#pragma unroll
        for (int q = 0; q < BPT; ++q) {
            const int nxt = run + h[q];
            if (run < a.k && nxt >= a.k) { s_bin = b0 + BPT - 1 - q; s_kk = a.k - run; s_nc = h[q]; }      // at most one (thread, q)
            run = nxt;
        }
    }
    lds_barrier();
    const int bsel = s_bin, kk = s_kk, nc = s_nc;
    if (tid < live && (int)(pub & 0xffffu) > bsel) s_slow = 1;         // a workgroup whose list stops above the selected bin
    const bool hi = i < n && digit > bsel, inb = i < n && digit == bsel;
    // per list quad: its keys above / in the bin (slots below the workgroup's count), summed over the eight quads of a workgroup
    // -> this quad's in-bin flags, the rank of its first in-bin key inside its workgroup's list, the workgroup's totals (every lane of the segment)
    auto quad_counts = [&](const uint4& q, int e, uint32_t& inq, int& seg_in, int& tot_in, int& tot_hi) {
        const int w = e < nquad ? e / QPW : 0, s0 = (e % QPW) * 4, cw = e < nquad ? s_cnt[w] : 0;
        const uint32_t kv[4] = {q.x, q.y, q.z, q.w};
        int chi = 0, cin = 0;
        inq = 0u;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int d = (int)(kv[c] >> (32 - GH_BITS));
            const bool v = s0 + c < cw;
            chi += (v && d > bsel) ? 1 : 0;
            if (v && d == bsel) { cin += 1; inq |= 1u << c; }
        }
        const int x = cin | (chi << 8);
        const int incl = wave_incl_scan(x), excl = incl - x;
        const int seg0 = __shfl(excl, lane & ~(QPW - 1), 64), segt = __shfl(incl, lane | (QPW - 1), 64) - seg0;
        seg_in = (excl - seg0) & 0xff; tot_in = segt & 0xff; tot_hi = segt >> 8;
    };
    for (int u = 0; u * DRAW_BLOCK < nquad; ++u) {                      // (uniform over the workgroup)
        const int e = tid + u * DRAW_BLOCK;
        const uint4 q = quad_of(u, e);
        uint32_t inq; int seg_in, tin, thi;
        quad_counts(q, e, inq, seg_in, tin, thi);
        if ((lane & (QPW - 1)) == 0 && e < nquad) { s_cin[e / QPW] = tin; s_chi[e / QPW] = thi; }
    }
    lds_barrier();
    // offsets over the workgroups and this thread's own in-bin rank
    int totw, toto;
    const int cw_in = tid < live ? s_cin[tid] : 0, cw_hi = tid < live ? s_chi[tid] : 0;
    const int exw = block_scan_1b<NW>(cw_in | (cw_hi << 14), sb[3], &totw);            // (sums: <= 512 * 32 = 2^14, < k <= n <= 2^18 - 1)
    const int my_r = block_scan_1b<NW>(inb ? 1 : 0, sb[4], &toto);
    const bool short_list = nc <= RANK_MAX && (totw & 0x3fff) == nc && bsel != 0;      // (bin 0: the rank loop's padding key lives there)
    if (tid < live) { s_offin[tid] = exw & 0x3fff; s_hib[tid] = (int)((unsigned)exw >> 14); }
    if (!short_list && tid == 0) s_slow = 1;
    lds_barrier();
    GRAPES_STAMP_NW(3);
    if (!s_slow) {
        if (tid < 4) cand[(nc & ~3) + tid] = 0u;                       // (the ragged quad's tail; its slots below nc are written next)
        lds_barrier();
        for (int u = 0; u * DRAW_BLOCK < nquad; ++u) {
            const int e = tid + u * DRAW_BLOCK;
            const uint4 q = quad_of(u, e);
            uint32_t inq; int seg_in, tin, thi;
            quad_counts(q, e, inq, seg_in, tin, thi);
            if (inq && e < nquad) {
                const uint32_t kv[4] = {q.x, q.y, q.z, q.w};
                int o = s_offin[e / QPW] + seg_in;
#pragma unroll
                for (int c = 0; c < 4; ++c) if (inq & (1u << c)) cand[o++] = kv[c];
            }
        }
        lds_barrier();
        GRAPES_STAMP_NW(11);
        // rank the in-bin keys (list order = position order): the one whose rank interval holds the place sought is the threshold.
        // All 512 threads share the nc x nc comparisons: thread (j, s) counts candidate j against slice s of the list (the first
        // version left them to the nc threads that own a candidate: 38 dependent LDS reads + ~50 vector instructions each, 4.5 us).
        // The list is padded to whole quads with a key of ANOTHER first-level bin (0: never above, never equal — the selected bin is
        // not bin 0 on this path), so the loop carries no validity tests; ties (exact 32-bit equality) are rare and get their
        // position rank from a second loop that runs only when one exists.
        const int jpad = (nc + 63) & ~63, S = DRAW_BLOCK / jpad, nq = (nc + 3) >> 2;        // (nc >= 1: the bin holds the k-th key)
        const int rj = tid % jpad, rs = tid / jpad;
        if (tid < nc) { s_gt[tid] = 0; s_eq[tid] = 0; }
        lds_barrier();
        const uint32_t mine = rj < nc ? cand[rj] : 0xffffffffu;
        if (rs < S) {
            const int q0i = (int)(((long long)nq * rs) / S), q1i = (int)(((long long)nq * (rs + 1)) / S);
            int gt = 0, eq = 0;
            for (int jq = q0i; jq < q1i; ++jq) {
                const uint4 o = cand4[jq];
                gt += (o.x > mine) + (o.y > mine) + (o.z > mine) + (o.w > mine);
                eq += (o.x == mine) + (o.y == mine) + (o.z == mine) + (o.w == mine);
            }
            if (rj < nc) { if (gt) atomicAdd(&s_gt[rj], gt); if (eq && atomicAdd(&s_eq[rj], eq) + eq > 1) s_tie = 1; }
        }
        lds_barrier();
        int gt = 0, eq = 0, eqb = 0;
        const uint32_t own = tid < nc ? cand[tid] : 0u;
        if (tid < nc) { gt = s_gt[tid]; eq = s_eq[tid]; }
        if (s_tie) {                                                               // a tie: equal keys rank by position
            if (tid < nc && eq > 1) for (int j = 0; j < tid; ++j) eqb += cand[j] == own ? 1 : 0;
        }
        if (tid < nc && gt < kk && kk <= gt + eq) { s_T = own; s_T_kk = kk - gt; }          // (equal keys write equal values)
        lds_barrier();
        GRAPES_STAMP_NW(12);
        const uint32_t T = s_T; const int take_eq = s_T_kk;
        if (bid == 0 && tid == 0) { f.sel[0] = T; f.sel[1] = (uint32_t)take_eq; }
        const int my0 = s_offin[bid];
        const bool keep_j = tid < nc && (own > T || (own == T && eqb < take_eq));
        {   // kept in-bin keys of the workgroups before this one; this workgroup's own in-bin keys' fate
            const unsigned long long mb = __ballot(keep_j && tid < my0);
            if (lane == 0 && mb) atomicAdd(&s_kb, __popcll(mb));
            if (tid < nc && tid >= my0 && tid < my0 + s_cin[bid]) s_keep[tid - my0] = keep_j ? 1 : 0;
        }
        lds_barrier();
        GRAPES_STAMP_NW(5);
        const bool keep = hi || (inb && s_keep[my_r < DRAW_LIST ? my_r : 0] != 0);
        int tot4;
        const int pos = s_hib[bid] + s_kb + block_scan_1b<NW>(keep ? 1 : 0, sb[5], &tot4);
        emit_place<DRAW_BLOCK>(a, n, i, keep, pos, lsg, l, cid, want_id, f.lsum_part, bid, red, with_ext, eb0, eb1);
    } else {      // the scan form: sampler_emit_k's list + passes over a.ord (agent-scope stores; every workgroup has passed the barrier)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // the first level over ALL keys, by every workgroup for itself (integer work: the same bin everywhere)
        lds_barrier();
#pragma unroll
        for (int q = 0; q < BPT; ++q) hist[tid + q * DRAW_BLOCK] = 0;
        if (tid == 0) { s_bin = 0; s_kk = a.k; }
        lds_barrier();
        for (int j = tid; j < n; j += DRAW_BLOCK) wave_hist_add(hist, (int)(a.ord[j] >> (32 - GH_BITS)), lane);     // (ballots see the active lanes only)
        lds_barrier();
        {
            const int4 ha = *reinterpret_cast<const int4*>(hist + b0), hb = *reinterpret_cast<const int4*>(hist + b0 + 4);
            const int h[BPT] = {hb.w, hb.z, hb.y, hb.x, ha.w, ha.z, ha.y, ha.x};
            int tot;
            int run = block_scan_1b<NW>(((h[0] + h[1]) + (h[2] + h[3])) + ((h[4] + h[5]) + (h[6] + h[7])), sb[0], &tot);
#pragma unroll
            for (int q = 0; q < BPT; ++q) {
                const int nxt = run + h[q];
                if (run < a.k && nxt >= a.k) { s_bin = b0 + BPT - 1 - q; s_kk = a.k - run; }
                run = nxt;
            }
        }
        lds_barrier();
        uint4 ko[DRAW_SCAN_BATCH];
        uint32_t T = 0u; int te = 0, ka = 0;
        threshold_body<DRAW_CAND, false, DRAW_SCAN_BATCH>(a, live, DRAW_BLOCK, f.sel, bid == 0, &T, &te, &ka, ko, (uint32_t)s_bin << (32 - GH_BITS), s_kk);
        const bool have_keys = ((n + 3) >> 2) <= DRAW_BLOCK * DRAW_SCAN_BATCH;
        emit_outputs<DRAW_BLOCK, DRAW_SCAN_BATCH>(a, n, bid, T, te, have_keys, ko, ok, lsg, l, f.lsum_part);
    }
    GRAPES_STAMP_NW(6);
    if (a.defer_finish) return;
    emit_tail<DRAW_BLOCK>(a, n, false, live, f.lsum_part, f.ticket, (unsigned)live, GH_BINS + GH_TAIL, live);
}

static inline size_t align8(size_t x) { return (x + 15) & ~(size_t)15; }   // 16 B: the key array is read as uint4

extern "C" size_t grapes_sampler_workspace_bytes(int32_t n_cap) {
    const size_t n = (size_t)(n_cap > 0 ? n_cap : 1), nb = (n + DRAW_BLOCK - 1) / DRAW_BLOCK;      // (partials per workgroup of the smaller block)
    return align8(n * 4) * 2 + (size_t)PART_BLOCKS * 5 * 8 + (size_t)KEYS_BLOCKS * 256 * 4 + 64 + align8(nb * 4) * 2 + align8(nb * 8) + 64 +
           (size_t)DRAW_MAX_LIVE * 4 + (size_t)DRAW_MAX_LIVE * DRAW_LIST * 4;            // (the one-launch draw's published pairs and segments)
}

static int gumbel_topk_impl(const NarrowAgg* agg, const float* logits, const int32_t* logit_index, const float* uniforms,
                                  uint64_t philox_seed, uint64_t philox_offset, uint64_t* d_philox_offset, int32_t n,
                                  const int32_t* d_n, int32_t k, int32_t mode, const int32_t* candidate_ids,
                                  float* mask, int32_t* kept_pos, int32_t* kept_ids, int32_t* d_kept_count,
                                  float* log_prob, float* keys_out, float* stats, const int32_t* prefix_ids,
                                  int32_t prefix_n, int32_t* union_ids, int32_t* d_union_count, void* workspace,
                                  uint32_t* d_hist, grapes_stream_t stream, grapes_draw_finish_args* finish = nullptr,
                                  const int64_t* ext_rowptr = nullptr, const int64_t* prefix_ext = nullptr, int64_t* union_ext = nullptr) {
    if (n < 0 || k <= 0 || (mode != 0 && mode != 1)) return GRAPES_EINVAL;   // utils.py:35 assert k > 0
    if (n > 0 && (!logits || !mask || !kept_pos || !workspace)) return GRAPES_EINVAL;
    if (((uintptr_t)workspace & 15) != 0) return GRAPES_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    SamplerArgs a;
    a.logits = logits; a.logit_index = logit_index; a.uniforms = uniforms;
    a.seed = philox_seed; a.offset = philox_offset; a.d_offset = d_philox_offset;
    a.n_host = n; a.d_n = d_n; a.k = k; a.mode = mode;
    static int norank = -1;
    if (norank < 0) { const char* e = grapes_tune_env("GRAPES_SAMPLER_RANK"); norank = (e && atoi(e) == 0) ? 1 : 0; }
    a.norank = norank;
    a.defer_finish = finish ? 1 : 0;
    a.cand_ids = candidate_ids; a.mask = mask; a.kept_pos = kept_pos; a.kept_ids = kept_ids;
    a.d_kept_count = d_kept_count; a.log_prob = log_prob; a.keys_out = keys_out; a.stats = stats;
    if (prefix_n < 0 || (prefix_n > 0 && (!prefix_ids || !union_ids))) return GRAPES_EINVAL;
    a.prefix_ids = prefix_ids; a.prefix_n = prefix_n; a.union_ids = union_ids; a.d_union_count = d_union_count;
    if (union_ext && (!ext_rowptr || !union_ids || !candidate_ids || (prefix_n > 0 && !prefix_ext) || (((uintptr_t)union_ext) & 15) != 0)) return GRAPES_EINVAL;
    a.rowptr = (const long long*)ext_rowptr; a.prefix_ext = (const long long*)prefix_ext; a.union_ext = (long long*)union_ext;
    a.gtm = nullptr; a.eqm = nullptr; a.eqb = nullptr; a.selb = nullptr;
    a.ghist = agg ? nullptr : d_hist;             // (the fused aggregation + keys launch keeps the per-workgroup rows)
    const size_t nn = (size_t)(n > 0 ? n : 1), nb = (nn + EMIT_BLOCK - 1) / EMIT_BLOCK;
    char* w = (char*)workspace;
    a.part = (double*)w; w += (size_t)PART_BLOCKS * 5 * 8;
    double* lsum_part = (double*)w; w += align8(((nn + DRAW_BLOCK - 1) / DRAW_BLOCK) * 8);      // (one partial per workgroup of either form)
    a.hist0 = (int32_t*)w; w += (size_t)KEYS_BLOCKS * 256 * 4;
    uint32_t* sel = (uint32_t*)w; w += 64;
    a.ord = (uint32_t*)w; w += align8(nn * 4);
    a.ls = (float*)w; w += align8(nn * 4);
    static int fused_on = -1;       // GRAPES_SAMPLER_FUSED=0 (A/B): the two-launch draw (sampler_keys_k + sampler_emit_k) at every size
    if (fused_on < 0) { const char* e = grapes_tune_env("GRAPES_SAMPLER_FUSED"); fused_on = (e && atoi(e) == 0) ? 0 : 1; }
    const size_t nbd = (nn + DRAW_BLOCK - 1) / DRAW_BLOCK;
    if (fused_on && a.ghist && !agg && n > 0 && nbd <= DRAW_MAX_WG) {
        // the draw in ONE launch: a workgroup per DRAW_BLOCK candidates of CAPACITY; the live ones are all resident (two per compute unit)
        a.hist0 = nullptr; a.ticket_zero = nullptr;
        DrawFused f;
        f.bar = a.ghist + GH_BINS; f.ticket = a.ghist + GH_BINS + 1;
        f.pubs = (uint32_t*)w; w += (size_t)DRAW_MAX_LIVE * 4;
        f.lists = (uint32_t*)w; w += (size_t)DRAW_MAX_LIVE * DRAW_LIST * 4;
        f.lsum_part = lsum_part; f.sel = sel;
        hipLaunchKernelGGL(sampler_draw_k, dim3((unsigned)nbd), dim3(DRAW_BLOCK), 0, s, a, f);
        GRAPES_LAUNCH_CHECK();
        if (finish)
            *finish = grapes_draw_finish_args{a.part + 4, lsum_part, sel, (int)nbd, DRAW_BLOCK, n, d_n, stats, a.ghist, GH_BINS + GH_TAIL,
                                              stats ? (int)nbd : 0};
        return 0;
    }
    const int kt = keys_threads();
    int kb = grapes_div_up(n > 0 ? n : 1, kt); if (kb > KEYS_BLOCKS) kb = KEYS_BLOCKS;   // one candidate per thread
    a.ticket_zero = sel + 3;
    int kt_dev = kt;
    if (agg) {                     // keys from the fused aggregation: one 256-row block of the batch per workgroup
        // (at most 192 workgroups, grid-stride beyond 49k batch rows: the launch is sized by the row CAPACITY, several times
        // the live count, and every selection workgroup sums one histogram row per workgroup launched here)
        kb = grapes_div_up(agg->n_host > 0 ? agg->n_host : 1, 256); if (kb > 192) kb = 192;
        kt_dev = 0;                // every workgroup writes its histogram row
        hipLaunchKernelGGL(sampler_agg_keys_k, dim3(kb), dim3(256), 0, s, *agg, a);
        GRAPES_LAUNCH_CHECK();
    } else if (n > 0) {
        hipLaunchKernelGGL(sampler_keys_k, dim3(kb), dim3(kt), 0, s, a);
        GRAPES_LAUNCH_CHECK();
    } else {
        kb = 0;
    }
    static int fuse_sel = -1;       // GRAPES_SAMPLER_TWO_LAUNCHES=0: the selection as a launch of its own (three launches)
    if (fuse_sel < 0) { const char* e = grapes_tune_env("GRAPES_SAMPLER_TWO_LAUNCHES"); fuse_sel = e ? atoi(e) : 1; }
    const int select_here = (fuse_sel && n > 0) ? 1 : 0;
    if (!select_here) {
        hipLaunchKernelGGL(sampler_threshold_k, dim3(1), dim3(1024), 0, s, a, kb, kt_dev, sel);
        GRAPES_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(sampler_emit_k, dim3((unsigned)nb), dim3(EMIT_BLOCK), 0, s, a, kb, sel, lsum_part, select_here, kt_dev);
    GRAPES_LAUNCH_CHECK();
    if (finish)
        *finish = grapes_draw_finish_args{a.part + 4, lsum_part, sel, kb, EMIT_BLOCK, n, d_n, stats, a.ghist, a.ghist ? GH_BINS : 0};
    return 0;
}

extern "C" int grapes_gumbel_topk(const float* logits, const int32_t* logit_index, const float* uniforms,
                                  uint64_t philox_seed, uint64_t philox_offset, uint64_t* d_philox_offset, int32_t n,
                                  const int32_t* d_n, int32_t k, int32_t mode, const int32_t* candidate_ids,
                                  float* mask, int32_t* kept_pos, int32_t* kept_ids, int32_t* d_kept_count,
                                  float* log_prob, float* keys_out, float* stats, const int32_t* prefix_ids,
                                  int32_t prefix_n, int32_t* union_ids, int32_t* d_union_count, void* workspace,
                                  grapes_stream_t stream) {
    return gumbel_topk_impl(nullptr, logits, logit_index, uniforms, philox_seed, philox_offset, d_philox_offset, n, d_n, k, mode,
                            candidate_ids, mask, kept_pos, kept_ids, d_kept_count, log_prob, keys_out, stats, prefix_ids, prefix_n,
                            union_ids, d_union_count, workspace, nullptr, stream);
}
/* grapes_gumbel_topk with the draw-wide first-level histogram: d_hist = grapes_sampler_hist_words() 32-bit words of caller memory,
 * ZERO before the first use and left zero; draws that share it must be stream-ordered.  Same results, bit for bit. */
extern "C" int32_t grapes_sampler_hist_words(void) { return GH_BINS + GH_TAIL; }
extern "C" int grapes_gumbel_topk_hist(const float* logits, const int32_t* logit_index, const float* uniforms,
                                       uint64_t philox_seed, uint64_t philox_offset, uint64_t* d_philox_offset, int32_t n,
                                       const int32_t* d_n, int32_t k, int32_t mode, const int32_t* candidate_ids,
                                       float* mask, int32_t* kept_pos, int32_t* kept_ids, int32_t* d_kept_count,
                                       float* log_prob, float* keys_out, float* stats, const int32_t* prefix_ids,
                                       int32_t prefix_n, int32_t* union_ids, int32_t* d_union_count, void* workspace,
                                       uint32_t* d_hist, grapes_stream_t stream) {
    if (d_hist && (((uintptr_t)d_hist) & 15) != 0) return GRAPES_EALIGN;
    return gumbel_topk_impl(nullptr, logits, logit_index, uniforms, philox_seed, philox_offset, d_philox_offset, n, d_n, k, mode,
                            candidate_ids, mask, kept_pos, kept_ids, d_kept_count, log_prob, keys_out, stats, prefix_ids, prefix_n,
                            union_ids, d_union_count, workspace, d_hist, stream);
}
/* grapes_gumbel_topk_deferred that ALSO writes the next query list's row extents (include/grapes_hip.h). */
extern "C" int grapes_gumbel_topk_deferred_ext(const float* logits, const int32_t* logit_index, const float* uniforms,
                                           uint64_t philox_seed, uint64_t philox_offset, uint64_t* d_philox_offset, int32_t n,
                                           const int32_t* d_n, int32_t k, int32_t mode, const int32_t* candidate_ids,
                                           float* mask, int32_t* kept_pos, int32_t* kept_ids, int32_t* d_kept_count,
                                           float* log_prob, float* keys_out, float* stats, const int32_t* prefix_ids,
                                           int32_t prefix_n, int32_t* union_ids, int32_t* d_union_count, void* workspace,
                                           uint32_t* d_hist, grapes_draw_finish_args* finish, const int64_t* rowptr,
                                           const int64_t* prefix_ext, int64_t* union_ext, grapes_stream_t stream) {
    if (!finish || n <= 0) return GRAPES_EINVAL;
    if (d_hist && (((uintptr_t)d_hist) & 15) != 0) return GRAPES_EALIGN;
    return gumbel_topk_impl(nullptr, logits, logit_index, uniforms, philox_seed, philox_offset, d_philox_offset, n, d_n, k, mode,
                            candidate_ids, mask, kept_pos, kept_ids, d_kept_count, log_prob, keys_out, stats, prefix_ids, prefix_n,
                            union_ids, d_union_count, workspace, d_hist, stream, finish, rowptr, prefix_ext, union_ext);
}
/* The draw with its logits produced on the way:  logits_out[r] = (Â head_in)[r] + *bias  over the hop's n_rows batch rows
 * (the 1-wide last layer of the sampler net), candidates = the batch rows with cand_pos[r] >= 0 (cand_pos / logit_index =
 * the compaction's cand_pos / nb_local).  Two launches (aggregation + keys, selection + emit) instead of three. */
extern "C" int grapes_gumbel_topk_from_aggregate(const float* head_in, const int32_t* rowptr_t, const int32_t* csr_src,
                                                 const float* dinv, const float* bias, float* logits_out, int32_t n_rows,
                                                 const int32_t* d_n_rows, const int32_t* cand_pos,
                                                 const int32_t* logit_index, const float* uniforms,
                                                 uint64_t philox_seed, uint64_t philox_offset, uint64_t* d_philox_offset, int32_t n,
                                                 const int32_t* d_n, int32_t k, int32_t mode, const int32_t* candidate_ids,
                                                 float* mask, int32_t* kept_pos, int32_t* kept_ids, int32_t* d_kept_count,
                                                 float* log_prob, float* keys_out, float* stats, const int32_t* prefix_ids,
                                                 int32_t prefix_n, int32_t* union_ids, int32_t* d_union_count, void* workspace,
                                                 grapes_stream_t stream) {
    if (n_rows <= 0 || !head_in || !rowptr_t || !dinv || !logits_out || !cand_pos || !logit_index) return GRAPES_EINVAL;
    static int lane_rows = -1;
    if (lane_rows < 0) { const char* e = grapes_tune_env("GRAPES_NARROW_LANE_ROWS"); lane_rows = e ? atoi(e) : 8; if (lane_rows < 0) lane_rows = 0; }
    NarrowAgg g{head_in, rowptr_t, csr_src, dinv, bias, logits_out, n_rows, d_n_rows, lane_rows, cand_pos};
    return gumbel_topk_impl(&g, logits_out, logit_index, uniforms, philox_seed, philox_offset, d_philox_offset, n, d_n, k, mode,
                            candidate_ids, mask, kept_pos, kept_ids, d_kept_count, log_prob, keys_out, stats, prefix_ids, prefix_n,
                            union_ids, d_union_count, workspace, nullptr, stream);
}

// d logits = g * (mask - sigmoid(l))     (d/dl of -BCEWithLogits(l, m); also of logsigmoid when m = 1)
// sum_out (optional): the sum of the written values — the bias gradient of a 1-wide head whose output these logits
// are — as per-workgroup partials combined in index order by the last workgroup to finish (ticket).
__global__ __launch_bounds__(256) void bernoulli_logprob_bwd_k(const float* __restrict__ logits, const int32_t* __restrict__ logit_index,
                                        const float* __restrict__ mask, const float* __restrict__ grad_vec,
                                        const float* d_grad_scale, float* __restrict__ dlogits, int n_host,
                                        const int32_t* d_n, float* __restrict__ sum_out, int accumulate_sum,
                                        float* __restrict__ partials, unsigned* __restrict__ ticket) {
    __shared__ float red[4];
    __shared__ int s_last;
    const int n = eff_count(d_n, n_host);
    const float gs = d_grad_scale ? *d_grad_scale : 1.0f;
    float local = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int li = logit_index ? logit_index[i] : i;
        const float l = logits[li];
        const float sg = 1.0f / (1.0f + expf(-l));
        const float g = gs * (grad_vec ? grad_vec[i] : 1.0f);
        const float v = g * (mask[i] - sg);
        dlogits[li] = v;
        local += v;
    }
    if (!sum_out) return;
    local = wave_sum(local);
    if (lane_id() == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        publish_f32(&partials[blockIdx.x], (red[0] + red[1]) + (red[2] + red[3]));
        s_last = (atomicAdd(ticket, 1u) == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    float acc = 0.f;                                  // fixed order: thread t owns partials t, t+256, ...; xor tree; waves in order
    for (unsigned b = threadIdx.x; b < gridDim.x; b += blockDim.x)
        acc += __int_as_float(__hip_atomic_load((const int*)(partials + b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    acc = wave_sum(acc);
    __syncthreads();
    if (lane_id() == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (red[0] + red[1]) + (red[2] + red[3]);
        *sum_out = accumulate_sum ? *sum_out + t : t;
        *ticket = 0u;
    }
}

extern "C" int grapes_bernoulli_logprob_bwd(const float* logits, const int32_t* logit_index, const float* mask,
                                            const float* grad_vec, const float* d_grad_scale, float* dlogits,
                                            int32_t n, const int32_t* d_n, float* sum_out, int32_t accumulate_sum,
                                            float* partials, uint32_t* d_ticket, grapes_stream_t stream) {
    if (n < 0) return GRAPES_EINVAL;
    if (sum_out && (!partials || !d_ticket)) return GRAPES_EINVAL;
    if (n == 0) {
        if (sum_out && !accumulate_sum) { hipError_t e = grapes_zero_async(sum_out, sizeof(float), (hipStream_t)stream); if (e) return (int)e; }
        return 0;
    }
    if (!logits || !mask || !dlogits) return GRAPES_EINVAL;
    int grid = grapes_div_up(n, 256); if (grid > 2048) grid = 2048;
    if (sum_out) {   // few workgroups: every one takes a ticket on ONE address, and same-address atomics serialise
        grid = grapes_div_up(n, 1024); if (grid > 96) grid = 96;
    }
    hipLaunchKernelGGL(bernoulli_logprob_bwd_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, logit_index, mask,
                       grad_vec, d_grad_scale, dlogits, n, d_n, sum_out, accumulate_sum, partials, (unsigned*)d_ticket);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------- small reductions
__global__ __launch_bounds__(1024) void reduce_sum_k(const float* __restrict__ x, int n_host, const int32_t* d_n,
                                                     int mean, float* __restrict__ out) {
    __shared__ double red[16];
    const int n = eff_count(d_n, n_host);
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += (double)x[i];
    s = wave_sum_d(s);
    if (lane_id() == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
        if (mean) t = n > 0 ? t / (double)n : 0.0 / 0.0;
        *out = (float)t;
    }
}

extern "C" int grapes_reduce_sum(const float* x, int32_t n, const int32_t* d_n, int32_t mean, float* out,
                                 grapes_stream_t stream) {
    if (n < 0 || !out || (!x && n > 0)) return GRAPES_EINVAL;
    hipLaunchKernelGGL(reduce_sum_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, d_n, mean, out);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

__global__ void fill_k(float* __restrict__ x, int n_host, const int32_t* d_n, float value, const float* d_value,
                       float scale_by_inv_n, float* __restrict__ sum_out, int accumulate_sum) {
    const int n = eff_count(d_n, n_host);
    float v = d_value ? *d_value : value;
    if (scale_by_inv_n != 0.0f) v = v * scale_by_inv_n / (float)(n > 0 ? n : 1);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] = v;
    if (sum_out && blockIdx.x == 0 && threadIdx.x == 0) {           // sum of the n equal values written
        const float t = v * (float)n;
        *sum_out = accumulate_sum ? *sum_out + t : t;
    }
}

extern "C" int grapes_fill(float* x, int32_t n, const int32_t* d_n, float value, const float* d_value,
                           float scale_by_inv_n, float* sum_out, int32_t accumulate_sum, grapes_stream_t stream) {
    if (n < 0 || (!x && n > 0)) return GRAPES_EINVAL;
    if (n == 0) return 0;
    int grid = grapes_div_up(n, 256); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(fill_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, d_n, value, d_value, scale_by_inv_n,
                       sum_out, accumulate_sum);
    GRAPES_LAUNCH_CHECK();
    return 0;
}
