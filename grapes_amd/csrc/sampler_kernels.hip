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
        const uint32_t ok = order_key(key);
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
                const uint32_t ok = order_key(key);
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
__device__ __forceinline__ void threshold_body(SamplerArgs a, int keys_blocks, int keys_threads_dev, uint32_t* __restrict__ sel,
                                               bool lead, uint32_t* T_out, int* take_eq_out, int* keep_all_out,
                                               uint4 (&o)[SCAN_BATCH]) {
    __shared__ int hist[256];
    __shared__ uint32_t s_prefix;
    __shared__ int s_kk;
    __shared__ int s_cnt;
    __shared__ int tlds[17];
    __shared__ double pr[4][16];
    __shared__ uint32_t cand[CAND_MAX];
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    const int BD = blockDim.x;
    const int n = eff_count(a.d_n, a.n_host);
    const int k = a.k;
    // statistics partials of sampler_keys_k, reduced in a fixed order (thread b owns partial b)
    if (a.stats && lead) {
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
    if (tid == 0) { s_prefix = 0u; s_kk = k; s_cnt = 0; }
    const int tshift = a.ghist ? 32 - GH_BITS : 24;               // the first level's digit = key >> tshift
    if (a.ghist) {
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
        for (int base = 0; base < n4; base += BD * SCAN_BATCH) {
#pragma unroll
            for (int u = 0; u < SCAN_BATCH; ++u) {
                const int i = base + u * BD + tid;
                o[u] = ord4[i < n4 ? i : n4 - 1];            // unconditional, clamped
            }
            if (n & 3) {                                      // the one ragged quad: its slots past n never match
#pragma unroll
                for (int u = 0; u < SCAN_BATCH; ++u)
                    if (base + u * BD + tid == n4 - 1) {
                        if ((n & 3) < 2) o[u].y = never;
                        if ((n & 3) < 3) o[u].z = never;
                        o[u].w = never;
                    }
            }
            int tot = 0;                                      // uniform over the wavefront
#pragma unroll
            for (int u = 0; u < SCAN_BATCH; ++u) {
                const bool inq = base + u * BD + tid < n4;
                tot += __popcll(__ballot(inq && (o[u].x >> tshift) == tb)) + __popcll(__ballot(inq && (o[u].y >> tshift) == tb)) +
                       __popcll(__ballot(inq && (o[u].z >> tshift) == tb)) + __popcll(__ballot(inq && (o[u].w >> tshift) == tb));
            }
            if (tot != 0) {
                int wbase = 0;
                if (lane == 0) wbase = atomicAdd(&s_cnt, tot);
                wbase = __builtin_amdgcn_readfirstlane(wbase);
#pragma unroll
                for (int u = 0; u < SCAN_BATCH; ++u) {
                    const bool inq = base + u * BD + tid < n4;
                    const uint32_t kv[4] = {o[u].x, o[u].y, o[u].z, o[u].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const bool match = inq && (kv[c] >> tshift) == tb;
                        const unsigned long long mm = __ballot(match);
                        if (mm != 0ull) {
                            const int p = wbase + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
                            if (match && p < CAND_MAX) cand[p] = kv[c];
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
    const bool in_lds = nc <= CAND_MAX;
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
__global__ __launch_bounds__(EMIT_BLOCK) void sampler_emit_k(SamplerArgs a, int keys_blocks, uint32_t* __restrict__ sel,
                                                             double* __restrict__ lsum_part, int select_here,
                                                             int keys_threads_dev) {
    __shared__ int lds[17];
    __shared__ double red[16];
    __shared__ int s_gt[16], s_eq[16];
    __shared__ int s_last;
    const int n = eff_count(a.d_n, a.n_host);
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
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
    if (a.defer_finish && blockIdx.x == 0 && tid == 0) {
        // everything the last workgroup used to write that does not depend on the other workgroups: the next launches read it
        sel[2] = sel2;                                               // (for the finishing workgroup of the caller's next launch)
        if (a.stats) a.stats[5] = keep_all ? 0.f : 1.f;
        if (a.d_kept_count) *a.d_kept_count = keep_all ? n : a.k;    // exactly k are drawn when n > k (utils.py:44)
        if (a.d_union_count) *a.d_union_count = a.prefix_n + (keep_all ? n : a.k);
        if (!keep_all && a.d_offset && a.mode == 0 && a.uniforms == nullptr) {      // (the keys launch has consumed the counter)
            const uint64_t off = *a.d_offset;
            *a.d_offset = off + (uint64_t)((n + 3) >> 2);
        }
    }
    if (!keep_all && (int)blockIdx.x * EMIT_BLOCK < n) {
        const uint32_t T = sel0;
        const int take_eq = (int)sel1;
        // candidates before this workgroup: [0, blockIdx.x * EMIT_BLOCK)
        const int before = blockIdx.x * EMIT_BLOCK;
        int c_gt = 0, c_eq = 0;
        {
            const int b4 = before >> 2;                                     // `before` is a multiple of EMIT_BLOCK (and of 4)
            if (have_keys) {    // the scan's registers: no second read of the keys.  b4 is a multiple of 64: a quad slot lies
                                // before this workgroup for a whole wavefront or not at all, and counts are per wavefront
                const int w0 = __builtin_amdgcn_readfirstlane(tid & ~63);
#pragma unroll
                for (int u = 0; u < SCAN_BATCH; ++u) {
                    if (u * EMIT_BLOCK + w0 < b4) {
                        c_gt += __popcll(__ballot(ko[u].x > T)) + __popcll(__ballot(ko[u].y > T)) + __popcll(__ballot(ko[u].z > T)) + __popcll(__ballot(ko[u].w > T));
                        c_eq += __popcll(__ballot(ko[u].x == T)) + __popcll(__ballot(ko[u].y == T)) + __popcll(__ballot(ko[u].z == T)) + __popcll(__ballot(ko[u].w == T));
                    }
                }
            } else {
                const uint4* ord4 = reinterpret_cast<const uint4*>(a.ord);
#pragma unroll 1
                for (int base = 0; base < b4; base += EMIT_BLOCK * SEL_BATCH) {
                    uint4 o[SEL_BATCH];
#pragma unroll
                    for (int u = 0; u < SEL_BATCH; ++u) {
                        const int i = base + u * EMIT_BLOCK + tid;
                        o[u] = ord4[i < b4 ? i : 0];                     // unconditional, clamped
                    }
#pragma unroll
                    for (int u = 0; u < SEL_BATCH; ++u) {
                        const bool in = base + u * EMIT_BLOCK + tid < b4;
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
        for (int w = 0; w < EMIT_BLOCK / 64; ++w) { gt_before += s_gt[w]; eq_before += s_eq[w]; }
        const int i = i_own;
        const uint32_t o = o_own;
        const float lsv = ls_own;
        const float lv = l_own;
        const bool gt = i < n && o > T, eq = i < n && o == T;
        // ONE workgroup scan for both ranks (each count <= 1024: 16 bits apiece).  Equal keys are taken in position order
        // until take_eq of them are in: the kept equal keys before this thread number min(eq before it, take_eq).
        int tot;
        const int packed = block_excl_scan((gt ? 1 : 0) | (eq ? 1 << 16 : 0), lds, &tot);
        const int eq_rank = eq_before + (packed >> 16);
        const bool keep = gt || (eq && eq_rank < take_eq);
        const int pos = gt_before + (packed & 0xffff) + (eq_rank < take_eq ? eq_rank : take_eq);
        double lp_d = 0.0;
        if (i < n) {
            a.mask[i] = keep ? 1.0f : 0.0f;
            if (keep) {
                a.kept_pos[pos] = i;
                if (a.kept_ids && a.cand_ids) a.kept_ids[pos] = a.cand_ids[i];
                if (a.union_ids && a.cand_ids) a.union_ids[a.prefix_n + pos] = a.cand_ids[i];
            }
            const float lp = keep ? lsv : lsv - lv;                  // -BCEWithLogits(l, m)   (utils.py:71)
            if (a.log_prob) a.log_prob[i] = lp;
            lp_d = (double)lp;
        }
        lp_d = wave_sum_d(lp_d);
        if (lane == 0) red[wid] = lp_d;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < EMIT_BLOCK / 64; ++w) t += red[w];
            publish_f64(&lsum_part[blockIdx.x], t);          // (see common.h: no device-scope fence)
        }
    }
    GRAPES_STAMP(3);
    if (a.defer_finish) return;              // the sum of the partials and the histogram's reset ride in the caller's next launch
    // ---- ticket: the last workgroup to arrive finalises
    if (tid == 0) {
        const unsigned t = atomicAdd(&sel[3], 1u);
        s_last = (t == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    GRAPES_STAMP(4);
    if (!s_last) return;
    const int nb = keep_all ? keys_blocks : (n + EMIT_BLOCK - 1) / EMIT_BLOCK;
    const double* parts = keep_all ? a.part + 4 : lsum_part;
    const int pstride = keep_all ? 5 : 1;
    double sacc = 0.0;      // thread b owns partial b (nb <= EMIT_BLOCK workgroups of either kind); then a fixed-order tree
    for (int bb = tid; bb < nb; bb += EMIT_BLOCK)
        sacc += __longlong_as_double(__hip_atomic_load((const long long*)(parts + (size_t)bb * pstride), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT));
    sacc = wave_sum_d(sacc);
    __syncthreads();
    if (lane == 0) red[wid] = sacc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < EMIT_BLOCK / 64; ++w) t += red[w];
        if (a.stats) { a.stats[4] = (float)t; a.stats[5] = keep_all ? 0.f : 1.f; }
        if (a.d_kept_count) *a.d_kept_count = keep_all ? n : a.k;    // exactly k are drawn when n > k (utils.py:44)
        if (a.d_union_count) *a.d_union_count = a.prefix_n + (keep_all ? n : a.k);
        if (!keep_all && a.d_offset && a.mode == 0 && a.uniforms == nullptr) {
            const uint64_t off = *a.d_offset;
            *a.d_offset = off + (uint64_t)((n + 3) >> 2);
        }
        sel[3] = 0u;                                                 // ticket ready for the next draw
    }
    if (a.ghist)     // every workgroup has read the draw's histogram before it took its ticket: back to zero for the next draw
        for (int b = tid; b < GH_BINS; b += EMIT_BLOCK) a.ghist[b] = 0u;
    GRAPES_STAMP(5);
}

static inline size_t align8(size_t x) { return (x + 15) & ~(size_t)15; }   // 16 B: the key array is read as uint4

extern "C" size_t grapes_sampler_workspace_bytes(int32_t n_cap) {
    const size_t n = (size_t)(n_cap > 0 ? n_cap : 1), nb = (n + EMIT_BLOCK - 1) / EMIT_BLOCK;
    return align8(n * 4) * 2 + (size_t)KEYS_BLOCKS * 5 * 8 + (size_t)KEYS_BLOCKS * 256 * 4 + 64 + align8(nb * 4) * 2 + align8(nb * 8) + 64;
}

static int gumbel_topk_impl(const NarrowAgg* agg, const float* logits, const int32_t* logit_index, const float* uniforms,
                                  uint64_t philox_seed, uint64_t philox_offset, uint64_t* d_philox_offset, int32_t n,
                                  const int32_t* d_n, int32_t k, int32_t mode, const int32_t* candidate_ids,
                                  float* mask, int32_t* kept_pos, int32_t* kept_ids, int32_t* d_kept_count,
                                  float* log_prob, float* keys_out, float* stats, const int32_t* prefix_ids,
                                  int32_t prefix_n, int32_t* union_ids, int32_t* d_union_count, void* workspace,
                                  uint32_t* d_hist, grapes_stream_t stream, grapes_draw_finish_args* finish = nullptr) {
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
    a.gtm = nullptr; a.eqm = nullptr; a.eqb = nullptr; a.selb = nullptr;
    a.ghist = agg ? nullptr : d_hist;             // (the fused aggregation + keys launch keeps the per-workgroup rows)
    const size_t nn = (size_t)(n > 0 ? n : 1), nb = (nn + EMIT_BLOCK - 1) / EMIT_BLOCK;
    char* w = (char*)workspace;
    a.part = (double*)w; w += (size_t)KEYS_BLOCKS * 5 * 8;
    double* lsum_part = (double*)w; w += align8(nb * 8);
    a.hist0 = (int32_t*)w; w += (size_t)KEYS_BLOCKS * 256 * 4;
    uint32_t* sel = (uint32_t*)w; w += 64;
    a.ord = (uint32_t*)w; w += align8(nn * 4);
    a.ls = (float*)w; w += align8(nn * 4);
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
extern "C" int32_t grapes_sampler_hist_words(void) { return GH_BINS; }
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
extern "C" int grapes_gumbel_topk_deferred(const float* logits, const int32_t* logit_index, const float* uniforms,
                                           uint64_t philox_seed, uint64_t philox_offset, uint64_t* d_philox_offset, int32_t n,
                                           const int32_t* d_n, int32_t k, int32_t mode, const int32_t* candidate_ids,
                                           float* mask, int32_t* kept_pos, int32_t* kept_ids, int32_t* d_kept_count,
                                           float* log_prob, float* keys_out, float* stats, const int32_t* prefix_ids,
                                           int32_t prefix_n, int32_t* union_ids, int32_t* d_union_count, void* workspace,
                                           uint32_t* d_hist, grapes_draw_finish_args* finish, grapes_stream_t stream) {
    if (!finish || n <= 0) return GRAPES_EINVAL;           // (an empty capacity launches nothing: there would be nothing to finish)
    if (d_hist && (((uintptr_t)d_hist) & 15) != 0) return GRAPES_EALIGN;
    return gumbel_topk_impl(nullptr, logits, logit_index, uniforms, philox_seed, philox_offset, d_philox_offset, n, d_n, k, mode,
                            candidate_ids, mask, kept_pos, kept_ids, d_kept_count, log_prob, keys_out, stats, prefix_ids, prefix_n,
                            union_ids, d_union_count, workspace, d_hist, stream, finish);
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
