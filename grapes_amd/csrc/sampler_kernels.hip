// A2: GFlowNet sampler draw — modules/utils.py:13-71 (training) and eval.py:126-130 (greedy).
// One workgroup does the whole draw: keys -> 4-pass radix select of the k-th largest key ->
// position-ordered compaction -> Bernoulli log-probs + statistics.
//
// THIS FILE IS COMPILED WITH -ffp-contract=off: p_expf / p_logf below must execute exactly the
// operation sequence of oracle/portable_math.py (IEEE +,-,*,/ only), so that the Gumbel-top-k
// keys — and therefore the sampled index sets — are bit-identical on the CPU oracle and on gfx950.
#include "common.h"

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
struct SamplerArgs {
    const float* logits; const int32_t* logit_index; const float* uniforms;
    uint64_t seed; uint64_t offset; uint64_t* d_offset;
    int n_host; const int32_t* d_n; int k; int mode;
    const int32_t* cand_ids; float* mask; int32_t* kept_pos; int32_t* kept_ids; int32_t* d_kept_count;
    float* log_prob; float* keys_out; float* stats; uint32_t* ord;
};

__global__ __launch_bounds__(1024) void gumbel_topk_k(SamplerArgs a) {
    __shared__ int lds[17];
    __shared__ int hist[256];
    __shared__ int suffix[257];
    __shared__ uint32_t s_prefix;
    __shared__ int s_kk;
    __shared__ float red_f[2][16];
    __shared__ double red_d[3][16];
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    const int n = eff_count(a.d_n, a.n_host);
    const int k = a.k;
    uint64_t offset = a.offset;
    if (a.d_offset) offset = *a.d_offset;

    if (n <= k) {   // utils.py:31-33 — keep every candidate, no noise consumed
        double lsum = 0.0;
        for (int i = tid; i < n; i += blockDim.x) {
            const float l = a.logits[a.logit_index ? a.logit_index[i] : i];
            const float lp = log_sigmoid_f(l);
            a.mask[i] = 1.0f;
            a.kept_pos[i] = i;
            if (a.kept_ids && a.cand_ids) a.kept_ids[i] = a.cand_ids[i];
            if (a.log_prob) a.log_prob[i] = lp;
            lsum += lp;
        }
        if (a.stats) {
            lsum = wave_sum_d(lsum);
            if (lane == 0) red_d[0][wid] = lsum;
            __syncthreads();
            if (tid == 0) {
                double t = 0.0;
                for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red_d[0][w];
                a.stats[0] = 0.f; a.stats[1] = 0.f; a.stats[2] = 0.f; a.stats[3] = 0.f;
                a.stats[4] = (float)t; a.stats[5] = 0.f;
            }
        }
        if (tid == 0 && a.d_kept_count) *a.d_kept_count = n;
        return;
    }

    // ---- pass 0: keys (portable math), order keys, statistics
    float pmin = INFINITY, pmax = -INFINITY;
    double esum = 0.0, esq = 0.0;
    for (int i = tid; i < n; i += blockDim.x) {
        const float l = a.logits[a.logit_index ? a.logit_index[i] : i];
        const float p = p_sigmoid(l);
        float key;
        if (a.mode == 1) {
            key = p;                                                   // eval.py:126-127
        } else {
            const float r = a.uniforms ? a.uniforms[i] : philox_uniform_at(a.seed, offset, i);
            key = p_logf(p) + p_gumbel(r);                             // utils.py:42
        }
        a.ord[i] = order_key(key);
        if (a.keys_out) a.keys_out[i] = key;
        if (a.stats) {
            pmin = fminf(pmin, p); pmax = fmaxf(pmax, p);
            float ent = -(p * log2f(p) + (1.0f - p) * log2f(1.0f - p));   // utils.py:47
            if (ent != ent) ent = 0.0f;                                   // utils.py:52-54
            esum += (double)ent; esq += (double)ent * (double)ent;
        }
    }
    if (a.stats) {
        pmin = wave_min(pmin); pmax = wave_max(pmax);
        esum = wave_sum_d(esum); esq = wave_sum_d(esq);
        if (lane == 0) { red_f[0][wid] = pmin; red_f[1][wid] = pmax; red_d[0][wid] = esum; red_d[1][wid] = esq; }
    }
    if (tid == 0) { s_prefix = 0u; s_kk = k; }
    __syncthreads();

    // ---- radix select: after 4 passes s_prefix = order key of the k-th largest, s_kk = how many
    //      elements equal to it are taken
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix;
        const uint32_t himask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
        // Keys cluster in a few exponent bins: same-address LDS atomics would serialise, so each
        // wavefront first peels off its (up to two) most common digits with a ballot and adds them
        // with ONE atomic each; only the remaining lanes issue individual atomics.
        const int n_round = (n + (int)blockDim.x - 1) / (int)blockDim.x * (int)blockDim.x;
        for (int i = tid; i < n_round; i += blockDim.x) {
            int digit = -1;
            if (i < n) {
                const uint32_t o = a.ord[i];
                if (((o ^ prefix) & himask) == 0u) digit = (int)((o >> shift) & 255u);
            }
#pragma unroll
            for (int round = 0; round < 2; ++round) {
                const unsigned long long act = __ballot(digit >= 0);
                if (act == 0ull) break;
                const int leader = __ffsll((long long)act) - 1;
                const int d0 = __shfl(digit, leader, 64);
                const unsigned long long same = __ballot(digit == d0);
                if (lane == leader) atomicAdd(&hist[d0], __popcll(same));
                if (digit == d0) digit = -1;
            }
            if (digit >= 0) atomicAdd(&hist[digit], 1);
        }
        __syncthreads();
        // suffix[b] = number of matching elements with digit >= b
        if (tid < 256) suffix[tid] = hist[tid];
        if (tid == 0) suffix[256] = 0;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            int v = 0;
            if (tid < 256) v = suffix[tid] + ((tid + d < 256) ? suffix[tid + d] : 0);
            __syncthreads();
            if (tid < 256) suffix[tid] = v;
            __syncthreads();
        }
        const int kk = s_kk;
        __syncthreads();
        if (tid < 256 && suffix[tid] >= kk && suffix[tid + 1] < kk) {
            s_prefix = prefix | ((uint32_t)tid << shift);
            s_kk = kk - suffix[tid + 1];
        }
        __syncthreads();
    }
    const uint32_t T = s_prefix;
    const int take_eq = s_kk;

    // ---- ordered selection: thread t owns the contiguous candidates [lo, hi)
    const int ipt = (n + blockDim.x - 1) / blockDim.x;
    const int lo = tid * ipt < n ? tid * ipt : n;
    const int hi = lo + ipt < n ? lo + ipt : n;
    int cgt = 0, ceq = 0;
    for (int i = lo; i < hi; ++i) {
        const uint32_t o = a.ord[i];
        cgt += o > T; ceq += o == T;
    }
    int tot;
    int eq_rank = block_excl_scan(ceq, lds, &tot);
    int eq_taken = take_eq - eq_rank; eq_taken = eq_taken < 0 ? 0 : (eq_taken > ceq ? ceq : eq_taken);
    int pos = block_excl_scan(cgt + eq_taken, lds, &tot);
    double lsum = 0.0;
    for (int i = lo; i < hi; ++i) {
        const uint32_t o = a.ord[i];
        bool sel = o > T;
        if (o == T) { sel = eq_rank < take_eq; ++eq_rank; }
        a.mask[i] = sel ? 1.0f : 0.0f;
        if (sel) {
            a.kept_pos[pos] = i;
            if (a.kept_ids && a.cand_ids) a.kept_ids[pos] = a.cand_ids[i];
            ++pos;
        }
        if (a.log_prob || a.stats) {
            const float l = a.logits[a.logit_index ? a.logit_index[i] : i];
            const float ls = log_sigmoid_f(l);
            const float lp = sel ? ls : ls - l;     // -BCEWithLogits(l, m)   (utils.py:71)
            if (a.log_prob) a.log_prob[i] = lp;
            lsum += lp;
        }
    }
    if (tid == 0 && a.d_kept_count) *a.d_kept_count = tot;
    if (a.stats) {
        lsum = wave_sum_d(lsum);
        if (lane == 0) red_d[2][wid] = lsum;
        __syncthreads();
        if (tid == 0) {
            const int nw = blockDim.x >> 6;
            float mn = INFINITY, mx = -INFINITY; double s1 = 0.0, s2 = 0.0, s3 = 0.0;
            for (int w = 0; w < nw; ++w) {
                mn = fminf(mn, red_f[0][w]); mx = fmaxf(mx, red_f[1][w]);
                s1 += red_d[0][w]; s2 += red_d[1][w]; s3 += red_d[2][w];
            }
            const double mean = s1 / (double)n;
            double var = n > 1 ? (s2 - s1 * s1 / (double)n) / (double)(n - 1) : 0.0;   // torch.std_mean: unbiased
            if (var < 0.0) var = 0.0;
            a.stats[0] = mn; a.stats[1] = mx; a.stats[2] = (float)mean; a.stats[3] = (float)sqrt(var);
            a.stats[4] = (float)s3; a.stats[5] = 1.0f;
        }
    }
    if (tid == 0 && a.d_offset && a.mode == 0 && a.uniforms == nullptr) *a.d_offset = offset + (uint64_t)((n + 3) >> 2);
}

extern "C" size_t grapes_sampler_workspace_bytes(int32_t n_cap) {
    return (size_t)(n_cap > 0 ? n_cap : 1) * sizeof(uint32_t);
}

extern "C" int grapes_gumbel_topk(const float* logits, const int32_t* logit_index, const float* uniforms,
                                  uint64_t philox_seed, uint64_t philox_offset, uint64_t* d_philox_offset, int32_t n,
                                  const int32_t* d_n, int32_t k, int32_t mode, const int32_t* candidate_ids,
                                  float* mask, int32_t* kept_pos, int32_t* kept_ids, int32_t* d_kept_count,
                                  float* log_prob, float* keys_out, float* stats, void* workspace,
                                  grapes_stream_t stream) {
    if (n < 0 || k <= 0 || (mode != 0 && mode != 1)) return GRAPES_EINVAL;   // utils.py:35 assert k > 0
    if (n > 0 && (!logits || !mask || !kept_pos || !workspace)) return GRAPES_EINVAL;
    SamplerArgs a;
    a.logits = logits; a.logit_index = logit_index; a.uniforms = uniforms;
    a.seed = philox_seed; a.offset = philox_offset; a.d_offset = d_philox_offset;
    a.n_host = n; a.d_n = d_n; a.k = k; a.mode = mode;
    a.cand_ids = candidate_ids; a.mask = mask; a.kept_pos = kept_pos; a.kept_ids = kept_ids;
    a.d_kept_count = d_kept_count; a.log_prob = log_prob; a.keys_out = keys_out; a.stats = stats;
    a.ord = (uint32_t*)workspace;
    hipLaunchKernelGGL(gumbel_topk_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// d logits = g * (mask - sigmoid(l))     (d/dl of -BCEWithLogits(l, m); also of logsigmoid when m = 1)
__global__ void bernoulli_logprob_bwd_k(const float* __restrict__ logits, const int32_t* __restrict__ logit_index,
                                        const float* __restrict__ mask, const float* __restrict__ grad_vec,
                                        const float* d_grad_scale, float* __restrict__ dlogits, int n_host,
                                        const int32_t* d_n) {
    const int n = eff_count(d_n, n_host);
    const float gs = d_grad_scale ? *d_grad_scale : 1.0f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int li = logit_index ? logit_index[i] : i;
        const float l = logits[li];
        const float sg = 1.0f / (1.0f + expf(-l));
        const float g = gs * (grad_vec ? grad_vec[i] : 1.0f);
        dlogits[li] = g * (mask[i] - sg);
    }
}

extern "C" int grapes_bernoulli_logprob_bwd(const float* logits, const int32_t* logit_index, const float* mask,
                                            const float* grad_vec, const float* d_grad_scale, float* dlogits,
                                            int32_t n, const int32_t* d_n, grapes_stream_t stream) {
    if (n < 0) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!logits || !mask || !dlogits) return GRAPES_EINVAL;
    int grid = grapes_div_up(n, 256); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(bernoulli_logprob_bwd_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, logit_index, mask,
                       grad_vec, d_grad_scale, dlogits, n, d_n);
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
                       float scale_by_inv_n) {
    const int n = eff_count(d_n, n_host);
    float v = d_value ? *d_value : value;
    if (scale_by_inv_n != 0.0f) v = v * scale_by_inv_n / (float)(n > 0 ? n : 1);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] = v;
}

extern "C" int grapes_fill(float* x, int32_t n, const int32_t* d_n, float value, const float* d_value,
                           float scale_by_inv_n, grapes_stream_t stream) {
    if (n < 0 || (!x && n > 0)) return GRAPES_EINVAL;
    if (n == 0) return 0;
    int grid = grapes_div_up(n, 256); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(fill_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, d_n, value, d_value, scale_by_inv_n);
    GRAPES_LAUNCH_CHECK();
    return 0;
}
