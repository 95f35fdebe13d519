// A7, first layers with F_in >= F_out (Reddit 602 + 3 -> 256, Cora 1433 + 3 -> 256) on the bf16 matrix pipe at fp32 accuracy.
//
// gfx950 runs v_mfma_f32_32x32x2_f32 at 1/16 of the bf16 rate, and the two GEMMs of such a layer — the transform
// H = feat(ids) Wᵀ and its weight gradient dW = dHᵀ feat(ids), 18.6 GFLOP each on Reddit's 77k-row frontier — were 80 % of
// that step on the fp32 pipe.  Same arithmetic as gemm_wsplit_f32_k (gemm_kernels.hip): every fp32 operand is split EXACTLY into
// three bf16 terms x = h + m + l and a product is the sum of the six cross terms hh, hm, mh, mm, hl, lh — each exact in the
// MFMA datapath — accumulated in fp32 (dropped: ml, lm, ll < 2^-26 |a·b|).  Here K is long (608 .. 1440), so nothing is
// register-stationary: both operands stream through double-buffered LDS images of split planes,
//     image[plane 3][k-group 4 (8 k each)][row][8 bf16]         (a lane's MFMA fragment = one conflict-free ds_read_b128)
// in K steps of 32, for a 128 x 256 output tile per workgroup (8 wavefronts, 2 x 4, each 64 x 64 = four 32x32 accumulators):
// 12 fragment reads feed 48 MFMAs per wavefront and K step, which keeps LDS traffic at ~45 % of the matrix pipe's time.
//   forward : A = feat(ids[r]) = [X[ids[r], 0:F] | indicator bits | 0] read through the id list (a thread owns one row of the
//             tile for the whole K loop: its id and indicator code are fetched once) and split on the way into LDS;
//             B = W, pre-split once per step into an image laid out exactly as the LDS stage (grapes_weight_split_image):
//             one contiguous 48 KB block per K step, copied linearly.
//   dW      : the contraction index is the ROW index, so both operands are staged transposed: a thread loads 8 consecutive
//             rows of one column quad and writes, per column, the 8 k-values as one 16-byte vector (as gemm_dw_split_k);
//             A = dH (k-major), B = feat(ids[r]) (k-major, gathered); split-K over the rows, slabs summed in index order.
// Fixed summation order (k steps in order, the six products in a fixed order, slabs in index order): deterministic.
#include "common.h"
#include <cstdlib>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define TS_BM 128
#define TS_BN 256
#define TS_BK 32
#define TS_A_U4 (3 * 4 * TS_BM)            // uint4 per A image  (24 KB)
#define TS_B_U4 (3 * 4 * TS_BN)            // uint4 per B image  (48 KB)
#define TS_STAGE (TS_A_U4 + TS_B_U4)       // 72 KB; two stages = 144 KB of the 160 KB LDS
static_assert(TS_BN == GRAPES_TS_IMG_ROWS, "the optimiser launch mirrors weights into the image by this row count (loss_kernels.hip)");

__device__ __forceinline__ void ts_split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

// ts_split3 of TWO values at once: v_cvt_pk_bf16_f32 converts a pair per instruction (hipcc uses it with one live half for a scalar
// conversion) and the packed results are the adjacent bf16 pair the images store.  Same operations per value: bit-identical.
typedef __bf16 ts_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ts_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t ts_cvt_pk(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(ts_f32x2{a, b}, ts_bf16x2));
}
__device__ __forceinline__ void ts_split3_pair(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = ts_cvt_pk(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = ts_cvt_pk(ra, rb);
    l = ts_cvt_pk(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u));
}
// The MFMAs of one K step (two k16 sub-steps) of a 64 x 64 wavefront tile.  Small cross terms first.
__device__ __forceinline__ void ts_mfma_stage(const uint4* __restrict__ st, int wm, int wn, int li, int h, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int kg = 2 * ks + h;
        bf16x8 a[3][2], b[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[p][i] = __builtin_bit_cast(bf16x8, st[(p * 4 + kg) * TS_BM + wm * 64 + i * 32 + (li ^ (2 * kg))]);
                b[p][i] = __builtin_bit_cast(bf16x8, st[TS_A_U4 + (p * 4 + kg) * TS_BN + wn * 64 + i * 32 + li]);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], c, 0, 0, 0);   // l h
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], c, 0, 0, 0);   // h l
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], c, 0, 0, 0);   // m m
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], c, 0, 0, 0);   // m h
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], c, 0, 0, 0);   // h m
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], c, 0, 0, 0);   // h h
                acc[i][j] = c;
            }
        }
    }
}

// The barriers of the producer / consumer kernels hand over LDS buffers only.  __syncthreads() would also drain every
// outstanding GLOBAL load of the wavefront (s_waitcnt vmcnt(0)) — i.e. throw away the producers' multi-step prefetch of the
// gathered rows at every K step and expose a full HBM round trip per step (GRAPES_TS_FULL_BARRIER=1 at build time restores it).
#ifdef GRAPES_TS_FULL_BARRIER
__device__ __forceinline__ void ts_barrier() { __syncthreads(); }
#else
__device__ __forceinline__ void ts_barrier() { lds_barrier(); }
#endif

struct TsGather {
    const float* X; int ldx; int F; const int32_t* ids; const uint32_t* code; const uint32_t* d_epoch; uint32_t epoch;
    uint32_t mask;
    // (round 4) the resident X split ONCE into its three bf16 planes (grapes_feature_split_planes): 24 bytes per 4-column chunk —
    // [h0..h3 | m0..m3 | l0..l3] — at a row pitch of 6 * ldx bytes.  X is a run-long constant, and splitting the gathered rows
    // inside the K loop cost ~19 vector instructions per element (profiles/r03_tsplit_ablation.txt: 41 of 189 us forward, 91 of
    // 275 us weight gradient at 77k rows).  NULL: split in the loop (X that changes — learned embeddings — or no room).
    const unsigned char* planes;
};
struct TsChunk3 { uint2 h, m, l; };          // the three planes of one 4-column chunk
__device__ __forceinline__ TsChunk3 ts_plane_load(const TsGather& ga, int g, int k) {
    const int kk = k + 4 <= ga.ldx ? k : ga.ldx - 4;
    const uint2* p = reinterpret_cast<const uint2*>(ga.planes + (long long)g * (6LL * ga.ldx) + 6LL * kk);
    TsChunk3 c; c.h = p[0]; c.m = p[1]; c.l = p[2];
    return c;
}
// ts_feat_fix on planes: 1.0f = bf16 0x3F80 with zero middle / low terms; columns beyond the storage are zero
__device__ __forceinline__ TsChunk3 ts_plane_fix(TsChunk3 c, const TsGather& ga, int k, uint32_t cd) {
    if (k >= ga.ldx) { c.h = make_uint2(0u, 0u); c.m = c.h; c.l = c.h; }
    const int b0 = k - ga.F;
    const uint32_t sh = b0 > 0 ? (b0 < 31 ? (uint32_t)b0 : 31u) : 0u;
    const uint32_t bits = b0 >= 0 ? (cd >> sh) : (cd << (b0 >= -3 ? -b0 : 3));
    if (b0 > -4 && b0 < 8) {
        if (bits & 1u) { c.h.x = (c.h.x & 0xffff0000u) | 0x3F80u; c.m.x &= 0xffff0000u; c.l.x &= 0xffff0000u; }
        if (bits & 2u) { c.h.x = (c.h.x & 0x0000ffffu) | 0x3F800000u; c.m.x &= 0x0000ffffu; c.l.x &= 0x0000ffffu; }
        if (bits & 4u) { c.h.y = (c.h.y & 0xffff0000u) | 0x3F80u; c.m.y &= 0xffff0000u; c.l.y &= 0xffff0000u; }
        if (bits & 8u) { c.h.y = (c.h.y & 0x0000ffffu) | 0x3F800000u; c.m.y &= 0x0000ffffu; c.l.y &= 0x0000ffffu; }
    }
    return c;
}
// planes[r][chunk q] = split3 of X[r][4q .. 4q+3]  (one thread per chunk)
__global__ __launch_bounds__(256) void ts_split_planes_k(const float* __restrict__ X, long long n, int ldx, unsigned char* __restrict__ planes) {
    const long long total = n * (ldx >> 2);
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const float4 v = *reinterpret_cast<const float4*>(X + 4 * t);
        uint2 ph, pm, pl;
        ts_split3_pair(v.x, v.y, ph.x, pm.x, pl.x);
        ts_split3_pair(v.z, v.w, ph.y, pm.y, pl.y);
        uint2* o = reinterpret_cast<uint2*>(planes + 24 * t);
        o[0] = ph; o[1] = pm; o[2] = pl;
    }
}
// Feature chunk [k, k+4) of row g in two BRANCH-FREE halves: an unconditional (column-clamped) load, and a fix-up by selects
// once the data is used.  (A per-lane branch around a load makes hipcc wait for every load separately — s_waitcnt vmcnt(0)
// at each join: the eight row fetches of a K step then run one after the other instead of together.)
__device__ __forceinline__ float4 ts_feat_load(const TsGather& ga, int g, int k) {
    const int kk = k + 4 <= ga.ldx ? k : ga.ldx - 4;
    return *reinterpret_cast<const float4*>(ga.X + (long long)g * ga.ldx + kk);
}
// cd: the row's indicator word, already reduced to the bits that count (epoch-checked, masked); X storage is zero in [F, ldx)
__device__ __forceinline__ float4 ts_feat_fix(float4 t, const TsGather& ga, int k, uint32_t cd) {
    if (k >= ga.ldx) t = make_float4(0.f, 0.f, 0.f, 0.f);
    const int b0 = k - ga.F;
    const uint32_t sh = b0 > 0 ? (b0 < 31 ? (uint32_t)b0 : 31u) : 0u;
    const uint32_t bits = b0 >= 0 ? (cd >> sh) : (cd << (b0 >= -3 ? -b0 : 3));     // bit j of `bits` = indicator of column k + j
    if (b0 > -4 && b0 < 8) {
        if (bits & 1u) t.x = 1.f;
        if (bits & 2u) t.y = 1.f;
        if (bits & 4u) t.z = 1.f;
        if (bits & 8u) t.w = 1.f;
    }
    return t;
}
__device__ __forceinline__ uint32_t ts_code_bits(const TsGather& ga, uint32_t cd, uint32_t epoch) {
    return ((cd >> 8) == epoch) ? (cd & ga.mask) : 0u;
}

// ---------------------------------------------------------------------------------------------- weight image
// img[k-step j][plane][k-group][n 0..255][8 bf16] <- split3(W[n][32 j + 8 kg + 0..7])   (zeros beyond K and beyond N)
// w_pad (optional) [N, ld_pad]: a zero-padded fp32 copy of W written on the way (the few-row fp32 kernels read it)
// (blockIdx.y = which of up to GRAPES_MAX_WEIGHT_IMAGES weights: the step's three first layers in one launch)
struct TsImages { const float* W[GRAPES_MAX_WEIGHT_IMAGES]; int ldw[GRAPES_MAX_WEIGHT_IMAGES]; int N[GRAPES_MAX_WEIGHT_IMAGES];
                  int K[GRAPES_MAX_WEIGHT_IMAGES]; uint4* img[GRAPES_MAX_WEIGHT_IMAGES]; int nk[GRAPES_MAX_WEIGHT_IMAGES];
                  float* w_pad[GRAPES_MAX_WEIGHT_IMAGES]; int ld_pad[GRAPES_MAX_WEIGHT_IMAGES]; };
__global__ __launch_bounds__(256) void ts_weight_image_k(TsImages im) {
    const int q = blockIdx.y;
    const float* __restrict__ W = im.W[q];
    uint4* __restrict__ img = im.img[q];
    float* __restrict__ w_pad = im.w_pad[q];
    const int ldw = im.ldw[q], N = im.N[q], K = im.K[q], nk = im.nk[q], ld_pad = im.ld_pad[q];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nk * 4 * TS_BN) return;
    const int nn = t & (TS_BN - 1), kg = (t >> 8) & 3, j = t >> 10;
    const int k0 = 32 * j + 8 * kg;
    bf16x8 ph, pm, pl;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const float x = (nn < N && k0 + u < K) ? W[(long long)nn * ldw + k0 + u] : 0.f;
        if (w_pad && nn < N && k0 + u < ld_pad) w_pad[(long long)nn * ld_pad + k0 + u] = x;
        __bf16 a, b, c; ts_split3(x, a, b, c);
        ph[u] = a; pm[u] = b; pl[u] = c;
    }
    uint4* o = img + (size_t)j * TS_B_U4;
    o[(0 * 4 + kg) * TS_BN + nn] = __builtin_bit_cast(uint4, ph);
    o[(1 * 4 + kg) * TS_BN + nn] = __builtin_bit_cast(uint4, pm);
    o[(2 * 4 + kg) * TS_BN + nn] = __builtin_bit_cast(uint4, pl);
}

// ---------------------------------------------------------------------------------------------- forward
// Split-K form (nslab > 1; few rows — the classifier's <= B + hops K, a small graph: a handful of 128-row tiles would otherwise
// walk the whole K alone): work unit t = (tile t % ntiles, K steps [kper (t / ntiles), + kper)), its partial tile goes to slab
// t / ntiles at out + slab * slab_stride; ts_fwd_slab_sum_k adds the slabs in index order.  nslab = 1: the plain kernel.
// One work unit of the forward GEMM: the 128-row tile at m0 over the K steps [j0, j1), written to outp (row pitch ldo).
template <bool PLANES>
__device__ __forceinline__ void ts_fwd_unit(const TsGather& ga, const uint4* __restrict__ wimg, int nk, float* __restrict__ outp, int ldo,
                                            int n, int N, int m0, int j0, int j1, uint32_t epoch, int dbg, uint4* ts_smem) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int wm = wid >> 2, wn = wid & 3;
    const int arow = tid >> 3, ac = tid & 7;                        // A staging: rows arow and arow + 64, chunk ac of the K step
    int g[2]; uint32_t cdb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) { const int r = m0 + arow + 64 * q; g[q] = ga.ids[r < n ? r : n - 1]; }
#pragma unroll
    for (int q = 0; q < 2; ++q) cdb[q] = ga.code ? ts_code_bits(ga, ga.code[g[q]], epoch) : 0u;
    f32x16 acc[2][2] = {{{0}, {0}}, {{0}, {0}}};
    float4 ra0, ra1; uint4 rb0, rb1, rb2, rb3, rb4, rb5;     // (named, not arrays: hipcc kept `rb[6]` in scratch)
    TsChunk3 pa0, pa1;                                        // PLANES: the chunk's three bf16 planes as stored
    auto load = [&](int j) __attribute__((always_inline)) {
        // dbg (grapes_debug_tsplit_fwd, diagnosis only): 2 = the gathered rows come from ONE row (cache-resident), 4 = no W loads
        if (PLANES) {
            pa0 = ts_plane_load(ga, (dbg & 2) ? 0 : g[0], 32 * j + 4 * ac);
            pa1 = ts_plane_load(ga, (dbg & 2) ? 0 : g[1], 32 * j + 4 * ac);
        } else {
            ra0 = ts_feat_load(ga, (dbg & 2) ? 0 : g[0], 32 * j + 4 * ac);
            ra1 = ts_feat_load(ga, (dbg & 2) ? 0 : g[1], 32 * j + 4 * ac);
        }
        const uint4* wj = wimg + (size_t)((dbg & 4) ? 0 : j) * TS_B_U4 + tid;
        rb0 = wj[0]; rb1 = wj[512]; rb2 = wj[1024]; rb3 = wj[1536]; rb4 = wj[2048]; rb5 = wj[2560];
    };
    auto stage_a = [&](uint4* st, const float4 r, int row) __attribute__((always_inline)) {
        uint2 q0, q1, q2;
        ts_split3_pair(r.x, r.y, q0.x, q1.x, q2.x);
        ts_split3_pair(r.z, r.w, q0.y, q1.y, q2.y);
        const bf16x4 p0 = __builtin_bit_cast(bf16x4, q0), p1 = __builtin_bit_cast(bf16x4, q1), p2 = __builtin_bit_cast(bf16x4, q2);
        // row r of k-group kg lives in slot r ^ (2 kg): the sixteen lanes of one LDS pass (two rows x eight chunks) then
        // write 128 different bytes instead of four times the same 32
        char* base = reinterpret_cast<char*>(st) + ((size_t)((ac >> 1) * TS_BM + (row ^ (ac & 6)))) * 16 + (ac & 1) * 8;
        *reinterpret_cast<bf16x4*>(base) = p0;
        *reinterpret_cast<bf16x4*>(base + (size_t)4 * TS_BM * 16) = p1;
        *reinterpret_cast<bf16x4*>(base + (size_t)8 * TS_BM * 16) = p2;
    };
    auto stage_p = [&](uint4* st, const TsChunk3& c, int row) __attribute__((always_inline)) {      // the planes go in as they are
        char* base = reinterpret_cast<char*>(st) + ((size_t)((ac >> 1) * TS_BM + (row ^ (ac & 6)))) * 16 + (ac & 1) * 8;
        *reinterpret_cast<uint2*>(base) = c.h;
        *reinterpret_cast<uint2*>(base + (size_t)4 * TS_BM * 16) = c.m;
        *reinterpret_cast<uint2*>(base + (size_t)8 * TS_BM * 16) = c.l;
    };
    auto stage = [&](int buf, int j) __attribute__((always_inline)) {
        uint4* st = ts_smem + (size_t)buf * TS_STAGE;
        if (PLANES) {
            if (32 * j + TS_BK > ga.F) {
                stage_p(st, ts_plane_fix(pa0, ga, 32 * j + 4 * ac, cdb[0]), arow);
                stage_p(st, ts_plane_fix(pa1, ga, 32 * j + 4 * ac, cdb[1]), arow + 64);
            } else {
                stage_p(st, pa0, arow);
                stage_p(st, pa1, arow + 64);
            }
        } else if (32 * j + TS_BK > ga.F) {       // (uniform) the K steps that hold the end of X: indicator columns, padding
            stage_a(st, ts_feat_fix(ra0, ga, 32 * j + 4 * ac, cdb[0]), arow);
            stage_a(st, ts_feat_fix(ra1, ga, 32 * j + 4 * ac, cdb[1]), arow + 64);
        } else {
            stage_a(st, ra0, arow);
            stage_a(st, ra1, arow + 64);
        }
        uint4* sb = st + TS_A_U4 + tid;
        sb[0] = rb0; sb[512] = rb1; sb[1024] = rb2; sb[1536] = rb3; sb[2048] = rb4; sb[2560] = rb5;
    };
    load(j0 < nk ? j0 : nk - 1);
    stage(0, j0 < nk ? j0 : nk - 1);
    __syncthreads();
    for (int j = j0; j < j1; ++j) {
        load(j + 1 < j1 ? j + 1 : j);                             // unconditional (clamped): stays ahead of the MFMAs
        if (!(dbg & 1)) ts_mfma_stage(ts_smem + (size_t)((j - j0) & 1) * TS_STAGE, wm, wn, li, h, acc);     // dbg 1: no MFMAs
        if (j + 1 < j1 && !(dbg & 8)) stage((j + 1 - j0) & 1, j + 1);                             // dbg 8: no staging
        __syncthreads();
    }
    // D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const int col = wn * 64 + jn * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < n && col < N) outp[(long long)row * ldo + col] = acc[i][jn][r];
            }
        }
    }
}

// Split-K form (nslab > 1; few rows — the classifier's <= B + hops K, a small graph: a handful of 128-row tiles would otherwise
// walk the whole K alone): work unit t = (tile t % ntiles, K steps [kper (t / ntiles), + kper)), its partial tile goes to slab
// t / ntiles at out + slab * slab_stride; ts_fwd_slab_sum_k adds the slabs in index order.  nslab = 1: the plain kernel.
template <bool PLANES = false>
__global__ __launch_bounds__(512, 1) void gemm_tsplit_fwd_k(TsGather ga, const uint4* __restrict__ wimg, int nk,
                                                            float* __restrict__ out, int ldo, int n_host,
                                                            const int32_t* d_n, int N, unsigned long long* clk, int dbg = 0,
                                                            int nslab = 1, int kper = 0x7fffffff, long long slab_stride = 0) {
    extern __shared__ uint4 ts_smem[];
    const unsigned long long clk0 = grapes_clock_begin(clk);
    const int n = eff_count(d_n, n_host);
    const int ntiles = (n + TS_BM - 1) / TS_BM;
    const uint32_t epoch = ga.d_epoch ? (*ga.d_epoch & 0xffffffu) : ga.epoch;
    for (int t = blockIdx.x; t < ntiles * nslab; t += gridDim.x) {
        const int tile = t % ntiles, sl = t / ntiles;
        const int j0 = sl * kper < nk ? sl * kper : nk, j1 = (nk - j0 > kper) ? j0 + kper : nk;       // this unit's K steps
        ts_fwd_unit<PLANES>(ga, wimg, nk, out + (long long)sl * slab_stride, ldo, n, N, tile * TS_BM, j0, j1, epoch, dbg, ts_smem);
    }
    grapes_clock_end(clk, clk0);
}

// ---- the forward GEMM of one or two nets over the SAME gathered rows (the sampler net and the log-Z net at hop 0, main.py:210,227)
// with a SPLIT TAIL.  Units u = tile * nprob + problem (the two problems of a tile on neighbouring workgroups: the second finds
// the rows in L2).  With G resident workgroups the first floor(units / G) G units are whole tiles; the R = units mod G that are
// left would cost a full round for R / G of the chip (Reddit's hop 1: 597 tiles on 256 CUs = three rounds for 2.33; hop 0: two
// nets x 182 tiles = two rounds for 1.42), so each of them is cut along K into S = floor(G / R) pieces (at least two K steps each),
// one workgroup per piece, partial tiles into `ws` [S][R][128][ldo]; ts_fwd_tail_sum_k adds the pieces in index order.  S = 1:
// nothing is cut, nothing is summed.  All sizes follow from the DEVICE row count; both kernels derive them the same way.
struct TsTail { int T_full, R, S, kper; };
__device__ __forceinline__ TsTail ts_tail_of(int units, int G, int nk) {
    TsTail t;
    t.T_full = (units / G) * G;
    t.R = units - t.T_full;
    int S = t.R > 0 ? G / t.R : 1;
    if (S > nk / 2) S = nk / 2;
    if (S > 8) S = 8;
    if (S < 1) S = 1;
    t.kper = (nk + S - 1) / S;
    t.S = (nk + t.kper - 1) / t.kper;
    return t;
}
struct TsProb2 { TsGather ga[2]; const uint4* wimg[2]; float* out[2]; };
__global__ __launch_bounds__(512, 1) void gemm_tsplit_fwd_tail_k(TsProb2 pr, int nprob, int nk, int ldo, int n_host, const int32_t* d_n, int N,
                                                                 float* __restrict__ ws, unsigned long long* clk) {
    extern __shared__ uint4 ts_smem[];
    const unsigned long long clk0 = grapes_clock_begin(clk);
    const int n = eff_count(d_n, n_host);
    const int ntiles = (n + TS_BM - 1) / TS_BM;
    const int units = ntiles * nprob, G = gridDim.x;
    const TsTail tl = ts_tail_of(units, G, nk);
    const uint32_t epoch0 = pr.ga[0].d_epoch ? (*pr.ga[0].d_epoch & 0xffffffu) : pr.ga[0].epoch;
    const uint32_t epoch1 = nprob > 1 ? (pr.ga[1].d_epoch ? (*pr.ga[1].d_epoch & 0xffffffu) : pr.ga[1].epoch) : 0u;
    const int total = tl.T_full + tl.R * tl.S;
    for (int t = blockIdx.x; t < total; t += G) {
        int u = t, j0 = 0, j1 = nk;
        float* wsp = nullptr;
        if (t >= tl.T_full) {
            const int q = t - tl.T_full, r = q % tl.R, sl = q / tl.R;
            u = tl.T_full + r;
            j0 = sl * tl.kper < nk ? sl * tl.kper : nk; j1 = (nk - j0 > tl.kper) ? j0 + tl.kper : nk;
            if (tl.S > 1) wsp = ws + ((long long)sl * tl.R + r) * TS_BM * ldo;
        }
        const int tile = u / nprob, pb = u - tile * nprob;
        const int m0 = tile * TS_BM;
        // a piece writes its partial tile at rows 0 .. 127 of its own block of ws: shift the base so that row m0 lands there
        if (pb == 0)
            ts_fwd_unit<false>(pr.ga[0], pr.wimg[0], nk, wsp ? wsp - (long long)m0 * ldo : pr.out[0], ldo, n, N, m0, j0, j1, epoch0, 0, ts_smem);
        else
            ts_fwd_unit<false>(pr.ga[1], pr.wimg[1], nk, wsp ? wsp - (long long)m0 * ldo : pr.out[1], ldo, n, N, m0, j0, j1, epoch1, 0, ts_smem);
    }
    grapes_clock_end(clk, clk0);
}
// out[rows of the cut units] = sum of their S pieces in index order (ldo % 4 == 0)
__global__ __launch_bounds__(256) void ts_fwd_tail_sum_k(const float4* __restrict__ ws, float* out0, float* out1, int nprob, int nk, int ldo,
                                                         int n_host, const int32_t* d_n, int G) {
    const int n = eff_count(d_n, n_host);
    const int ntiles = (n + TS_BM - 1) / TS_BM;
    const TsTail tl = ts_tail_of(ntiles * nprob, G, nk);
    if (tl.S <= 1 || tl.R <= 0) return;
    const int f4 = ldo >> 2;
    const long long per = (long long)TS_BM * f4, total = (long long)tl.R * per;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / per);
        const long long rem = i - (long long)r * per;
        const int rr = (int)(rem / f4), c4 = (int)(rem - (long long)rr * f4);
        const int u = tl.T_full + r, tile = u / nprob, pb = u - tile * nprob;
        const int row = tile * TS_BM + rr;
        if (row >= n) continue;
        float4 acc = ws[i];
        for (int sl = 1; sl < tl.S; ++sl) {
            const float4 v = ws[(long long)sl * total + i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        float* o = (pb == 0 ? out0 : out1) + (long long)row * ldo + 4 * c4;
        *reinterpret_cast<float4*>(o) = acc;
    }
}

// ---------------------------------------------------------------------------------------------- forward, producer / consumer form
// The kernel above runs all eight wavefronts in lockstep — load, 48 MFMAs, split + stage, barrier — so the matrix pipe idles
// while the step is staged: 3.3 us per K step for 1.3 us of MFMA time (Reddit's 77k-row frontier: 187 us where the pipe needs
// 60).  Here, as in the dW kernel below, wavefronts 0-3 (one per SIMD) only issue MFMAs — 64 x 128 of the 128 x 256 tile each,
// 96 MFMAs per step back to back — and wavefronts 4-7 only load, split and stage the NEXT step into the other LDS buffer:
// VALU / LDS writes and matrix pipe of a SIMD work at the same time.  Same LDS image, same MFMA order per accumulator as the
// lockstep kernel (k steps in order, the six products in the same order): results are bit-identical to it.
__device__ __forceinline__ void ts_mfma_stage_wide_fwd(const uint4* __restrict__ st, int wm, int wn2, int li, int h, f32x16 (&acc)[2][4]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int kg = 2 * ks + h;
        bf16x8 a[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) a[p][i] = __builtin_bit_cast(bf16x8, st[(p * 4 + kg) * TS_BM + wm * 64 + i * 32 + (li ^ (2 * kg))]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16x8 b[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) b[p] = __builtin_bit_cast(bf16x8, st[TS_A_U4 + (p * 4 + kg) * TS_BN + wn2 * 128 + j * 32 + li]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x16 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0], c, 0, 0, 0);   // l h
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2], c, 0, 0, 0);   // h l
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1], c, 0, 0, 0);   // m m
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0], c, 0, 0, 0);   // m h
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1], c, 0, 0, 0);   // h m
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0], c, 0, 0, 0);   // h h
                acc[i][j] = c;
            }
        }
    }
}
__global__ __launch_bounds__(512, 1) void gemm_tsplit_fwd_pc_k(TsGather ga, const uint4* __restrict__ wimg, int nk,
                                                               float* __restrict__ out, int ldo, int n_host,
                                                               const int32_t* d_n, int N, unsigned long long* clk) {
    extern __shared__ uint4 ts_smem[];
    const unsigned long long clk0 = grapes_clock_begin(clk);
    const int n = eff_count(d_n, n_host);
    const int ntiles = (n + TS_BM - 1) / TS_BM;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const uint32_t epoch = ga.d_epoch ? (*ga.d_epoch & 0xffffffu) : ga.epoch;
    // Protocol: this workgroup's (tile, K step) pairs form one sequence q = 0 .. Q-1, step q lives in LDS buffer q & 1.
    //   consumers:  for q: barrier_q; MFMAs(q)                       producers:  stage(0); for q: barrier_q; stage(q + 1)
    // Passing barrier_q tells the producers that the MFMAs of step q - 1 are done (its buffer is the one stage(q + 1) writes) and
    // the consumers that step q is staged.  Both sides execute exactly Q barriers.
    const int my_tiles = (int)blockIdx.x < ntiles ? (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int Q = my_tiles * nk;
    if (wid < 4) {
        // ------------------------------------------------------------------ consumers: MFMAs only
        const int wm = wid >> 1, wn2 = wid & 1;
        int q = 0;
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const int m0 = tile * TS_BM;
            f32x16 acc[2][4] = {{{0}, {0}, {0}, {0}}, {{0}, {0}, {0}, {0}}};
            for (int j = 0; j < nk; ++j, ++q) {
                ts_barrier();                                     // barrier_q: step q is staged
                ts_mfma_stage_wide_fwd(ts_smem + (size_t)(q & 1) * TS_STAGE, wm, wn2, li, h, acc);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int jn = 0; jn < 4; ++jn) {
                    const int col = wn2 * 128 + jn * 32 + li;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (row < n && col < N) out[(long long)row * ldo + col] = acc[i][jn][r];
                    }
                }
            }
        }
        grapes_clock_end(clk, clk0);
        return;
    }
    // ---------------------------------------------------------------------- producers: load, split, stage
    const int pt = tid - 256;
    const int arow = pt >> 3, ac = pt & 7;                          // A staging: rows arow + 32 u (u = 0..3), chunk ac of the K step
    auto stage_a = [&](uint4* st, const float4 r, int row) __attribute__((always_inline)) {
        uint2 q0, q1, q2;
        ts_split3_pair(r.x, r.y, q0.x, q1.x, q2.x);
        ts_split3_pair(r.z, r.w, q0.y, q1.y, q2.y);
        const bf16x4 p0 = __builtin_bit_cast(bf16x4, q0), p1 = __builtin_bit_cast(bf16x4, q1), p2 = __builtin_bit_cast(bf16x4, q2);
        char* base = reinterpret_cast<char*>(st) + ((size_t)((ac >> 1) * TS_BM + (row ^ (ac & 6)))) * 16 + (ac & 1) * 8;
        *reinterpret_cast<bf16x4*>(base) = p0;
        *reinterpret_cast<bf16x4*>(base + (size_t)4 * TS_BM * 16) = p1;
        *reinterpret_cast<bf16x4*>(base + (size_t)8 * TS_BM * 16) = p2;
    };
    // The gathered rows are random 2.4 KB rows of a matrix far larger than the caches: an HBM round trip (~2-3 us) is longer than
    // a K step's 1.3 us of matrix work, so the A loads run THREE steps ahead of the staging (three register sets, 48 VGPRs —
    // the producers have the consumers' accumulator registers to spare); W's image is L2-resident: one step ahead.
    int g[4]; uint32_t cdb[4], cdb_tail[4] = {0u, 0u, 0u, 0u};
    float4 raA[4], raB[4], raC[4]; uint4 rb0, rb1, rb2, rb3, rb4, rb5, rb6, rb7, rb8, rb9, rb10, rb11;
    auto ids_of = [&](int tile) __attribute__((always_inline)) {
        const int m0 = tile * TS_BM;
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int r = m0 + arow + 32 * u; g[u] = ga.ids[r < n ? r : n - 1]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) cdb[u] = ga.code ? ts_code_bits(ga, ga.code[g[u]], epoch) : 0u;
    };
    int lt = blockIdx.x, lj = 0, cur_t = -1, issued = 0;             // (lt, lj): the next step whose A loads leave; `issued` of Q so far
    auto load_a = [&](float4 (&ra)[4]) __attribute__((always_inline)) {
        if (issued >= Q) return;
        if (lt != cur_t) { ids_of(lt); cur_t = lt; }
#pragma unroll
        for (int u = 0; u < 4; ++u) ra[u] = ts_feat_load(ga, g[u], 32 * lj + 4 * ac);
        if (lj == nk - 1) {                                          // the tile's tail step: its indicator words, kept until it is staged
#pragma unroll
            for (int u = 0; u < 4; ++u) cdb_tail[u] = cdb[u];
        }
        ++issued;
        if (++lj >= nk) { lj = 0; lt += gridDim.x; }
    };
    auto load_b = [&](int j) __attribute__((always_inline)) {
        const uint4* wj = wimg + (size_t)j * TS_B_U4 + pt;
        rb0 = wj[0]; rb1 = wj[256]; rb2 = wj[512]; rb3 = wj[768]; rb4 = wj[1024]; rb5 = wj[1280];
        rb6 = wj[1536]; rb7 = wj[1792]; rb8 = wj[2048]; rb9 = wj[2304]; rb10 = wj[2560]; rb11 = wj[2816];
    };
    auto stage = [&](int buf, int j, const float4 (&ra)[4]) __attribute__((always_inline)) {
        uint4* st = ts_smem + (size_t)buf * TS_STAGE;
        if (32 * j + TS_BK > ga.F) {              // (uniform) the K steps that hold the end of X: indicator columns, padding
#pragma unroll
            for (int u = 0; u < 4; ++u) stage_a(st, ts_feat_fix(ra[u], ga, 32 * j + 4 * ac, cdb_tail[u]), arow + 32 * u);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) stage_a(st, ra[u], arow + 32 * u);
        }
        uint4* sb = st + TS_A_U4 + pt;
        sb[0] = rb0; sb[256] = rb1; sb[512] = rb2; sb[768] = rb3; sb[1024] = rb4; sb[1280] = rb5;
        sb[1536] = rb6; sb[1792] = rb7; sb[2048] = rb8; sb[2304] = rb9; sb[2560] = rb10; sb[2816] = rb11;
    };
    if (Q > 0) {
        // (the tail step needs nk >= 2 steps between a tile's tail load and the next tile's: nk >= 4 keeps one tail in flight)
        // Loads complete in issue order (one vmcnt queue per wavefront): a step's W loads must therefore leave BEFORE the gathered
        // loads that run ahead of it, or waiting for them would wait for the whole prefetch.
        load_b(0);
        load_a(raA); load_a(raB); load_a(raC);                       // steps 0, 1, 2
        stage(0, 0, raA);                                            // step 0 (set A)
        int sj = 1 % nk;                                             // K step index of the step staged next (q + 1)
        if (Q > 1) load_b(sj);
        load_a(raA);                                                 // step 3
        for (int q = 0; q < Q; ++q) {
            ts_barrier();                                            // barrier_q
            if (q + 1 < Q) {
                const int set = (q + 1) % 3;
                const int sjn = sj + 1 >= nk ? 0 : sj + 1;
                if (set == 0) { stage((q + 1) & 1, sj, raA); if (q + 2 < Q) load_b(sjn); load_a(raA); }
                else if (set == 1) { stage((q + 1) & 1, sj, raB); if (q + 2 < Q) load_b(sjn); load_a(raB); }
                else { stage((q + 1) & 1, sj, raC); if (q + 2 < Q) load_b(sjn); load_a(raC); }
                sj = sjn;
            }
        }
    }
    grapes_clock_end(clk, clk0);
}

// ---------------------------------------------------------------------------------------------- dW (split-K over rows)
// dW[m][c] = sum_r dH[r][m] * feat(ids[r])[c];  output tile 128 (m) x 256 (c) per workgroup, rows [r_lo, r_hi) per slab.
// PRODUCER / CONSUMER wavefronts.  Both operands are transposed AND split on the way into LDS — 12,288 elements per K step,
// ~6 VALU operations each — which, done by the wavefronts that also issue the MFMAs, ran serially with them (all wavefronts
// are in the same phase between two barriers): 196 us where the matrix pipe needs 60.  Here wavefronts 0-3 (one per SIMD)
// only issue MFMAs — 64 x 128 of the tile each, eight 32x32 accumulators, 96 MFMAs per step back to back — and wavefronts
// 4-7 (the other wavefront of each SIMD) only load, split and stage the next step: VALU and matrix pipe of a SIMD work at
// the same time.  The producers keep TWO steps of global loads in flight (ids of the gathered rows three steps ahead): an
// HBM row fetch takes longer than one step's 1.3 us of matrix work.
// Both images of the dW kernel are written TRANSPOSED (a producer lane holds four consecutive columns of its rows: lane stride
// 64 bytes, every ds_write four- to sixteen-way bank conflicted); so column n of a k-group lives in slot ts_sw(n): consecutive
// lanes then land in different 16-byte bank groups, and a fragment read (32 consecutive columns) stays a permutation of the
// same 512 bytes.
__device__ __forceinline__ int ts_sw(int n) { return n ^ ((n >> 3) & 3); }
__device__ __forceinline__ void ts_mfma_stage_wide(const uint4* __restrict__ st, int wm, int wn2, int li, int h, f32x16 (&acc)[2][4]) {
    li = ts_sw(li);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int kg = 2 * ks + h;
        bf16x8 a[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) a[p][i] = __builtin_bit_cast(bf16x8, st[(p * 4 + kg) * TS_BM + wm * 64 + i * 32 + li]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16x8 b[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) b[p] = __builtin_bit_cast(bf16x8, st[TS_A_U4 + (p * 4 + kg) * TS_BN + wn2 * 128 + j * 32 + li]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x16 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0], c, 0, 0, 0);   // l h
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2], c, 0, 0, 0);   // h l
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1], c, 0, 0, 0);   // m m
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0], c, 0, 0, 0);   // m h
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1], c, 0, 0, 0);   // h m
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0], c, 0, 0, 0);   // h h
                acc[i][j] = c;
            }
        }
    }
}
// The same step for a 64 x 64 wavefront tile (EIGHT consumer wavefronts, two per SIMD: see gemm_tsplit_dw_k<8>); per
// accumulator the same six products in the same order as the wide form.
// SWAPPED: the A image holds the FEATURE columns and the B image dH's (gemm_tsplit_dw_sw_k) — the cross terms are issued so that
// an accumulator still receives dH.l x.h, dH.h x.l, m m, dH.m x.h, dH.h x.m, h h in this order.
template <bool SWAPPED = false>
__device__ __forceinline__ void ts_mfma_stage_sw(const uint4* __restrict__ st, int wm, int wn, int li, int h, f32x16 (&acc)[2][2]) {
    li = ts_sw(li);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int kg = 2 * ks + h;
        bf16x8 a[3][2], b[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[p][i] = __builtin_bit_cast(bf16x8, st[(p * 4 + kg) * TS_BM + wm * 64 + i * 32 + li]);
                b[p][i] = __builtin_bit_cast(bf16x8, st[TS_A_U4 + (p * 4 + kg) * TS_BN + wn * 64 + i * 32 + li]);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 c = acc[i][j];
                if constexpr (SWAPPED) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], c, 0, 0, 0);   // x.h dH.l
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], c, 0, 0, 0);   // x.l dH.h
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], c, 0, 0, 0);   // m m
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], c, 0, 0, 0);   // x.h dH.m
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], c, 0, 0, 0);   // x.m dH.h
                } else {
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], c, 0, 0, 0);   // l h
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], c, 0, 0, 0);   // h l
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], c, 0, 0, 0);   // m m
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], c, 0, 0, 0);   // m h
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], c, 0, 0, 0);   // h m
                }
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], c, 0, 0, 0);   // h h
                acc[i][j] = c;
            }
        }
    }
}
// one producer thread's share of a K step: a feature task (8 rows x 4 columns of the gathered operand) and a dH half-task
// (4 rows x 4 columns)
struct TsProd { float4 f[8]; float4 d[4]; uint32_t cd[8]; };
struct TsProdP { TsChunk3 p[8]; float4 d[4]; uint32_t cd[8]; };      // PLANES: the gathered rows' chunks as stored (three bf16 planes)
// CW = consumer wavefronts: 4 (64 x 128 of the tile each, ONE MFMA wavefront per SIMD) or 8 (64 x 64 each, TWO per SIMD, a
// 768-thread workgroup).  One MFMA wavefront per SIMD stalls on its own fragment reads — "MFMAs only" measured 185 us for the
// 144 GFLOP of Reddit's hop 1 against 143 us in the forward kernel, whose two wavefronts per SIMD cover each other
// (profiles/r03_tsplit_ablation.txt); the producers' four wavefronts stay as they are.  Bit-identical either way.
template <int CW, bool PLANES = false>
__global__ __launch_bounds__(64 * (CW + 4), 1) void gemm_tsplit_dw_k(const float* __restrict__ dH, int M /* f_out */, TsGather ga, int Kp,
                                                           float* __restrict__ slabs, int n_host, const int32_t* d_n,
                                                           int nslab, int mt, int ct, int dbg = 0) {
    extern __shared__ uint4 ts_smem[];
    const int n = eff_count(d_n, n_host);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const uint32_t epoch = ga.d_epoch ? (*ga.d_epoch & 0xffffffu) : ga.epoch;
    // XCD-aware map: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), and the mt * ct output tiles of
    // one slab read the SAME rows of dH and of the gathered features — so all tiles of a slab get ids that are equal mod 8:
    // the re-reads are L2 hits on that XCD instead of HBM traffic.
    const int nt = mt * ct;
    const int grp = blockIdx.x / (8 * nt), within = blockIdx.x - grp * 8 * nt;
    const int slab = grp * 8 + (within & 7), tileid = within >> 3;
    if (slab >= nslab) return;
    const int tm = tileid / ct, tc = tileid - tm * ct;
    const int m0 = tm * TS_BM, c0 = tc * TS_BN;
    const int steps = (n + TS_BK - 1) / TS_BK;                      // balanced row ranges, multiples of 32
    const int per = (steps + nslab - 1) / nslab;
    const int s_lo = slab * per, s_hi = (s_lo + per < steps) ? s_lo + per : steps;
    const int nst = s_hi > s_lo ? s_hi - s_lo : 0;
    if (CW == 8 && wid < 8) {
        // ------------------------------------------------------------------ consumers (two per SIMD): the MFMAs — and the dH image.
        // With the MFMAs on two wavefronts per SIMD the launch was bound by its four producer wavefronts (counters: 46 % of the
        // wavefront cycles parked at a barrier or a wait); the consumers take the smaller third of the staging — dH, 4,096 of a
        // step's 12,288 elements: two rows x four columns per thread, loaded two steps ahead, split and written (4-byte pairs)
        // after the step's MFMAs are issued — the producers keep the gathered feature rows.
        const int wm = wid >> 2, wn = wid & 3;
        const int hq2 = 4 * (tid >> 7) + (tid & 3), aq = (tid >> 2) & 31;     // row pair 0..15, column quad 0..31
        const int asw = (aq >> 1) & 3, mq = m0 + 4 * aq;
        const int s_last = s_hi - 1;
        auto load_d = [&](float4 (&d)[2], int s) __attribute__((always_inline)) {
            const int r0 = (s < s_last ? s : s_last) * TS_BK + 2 * hq2;
#pragma unroll
            for (int u = 0; u < 2; ++u)
                d[u] = *reinterpret_cast<const float4*>(dH + (long long)(r0 + u < n ? r0 + u : n - 1) * M + (mq + 3 < M ? mq : 0));
        };
        auto stage_d = [&](const float4 (&d)[2], int buf, int s) __attribute__((always_inline)) {
            uint4* st = ts_smem + (size_t)buf * TS_STAGE;
            const int r0 = s * TS_BK + 2 * hq2;
            float4 dd[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) dd[u] = (r0 + u >= n || mq + 3 >= M) ? make_float4(0.f, 0.f, 0.f, 0.f) : d[u];
            char* dbase = reinterpret_cast<char*>(st + (hq2 >> 2) * TS_BM + 4 * aq) + (hq2 & 3) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                uint32_t qh[1], qm[1], ql[1];
#pragma unroll
                for (int u = 0; u < 2; u += 2) {
                    const float x0 = c == 0 ? dd[u].x : (c == 1 ? dd[u].y : (c == 2 ? dd[u].z : dd[u].w));
                    const float x1 = c == 0 ? dd[u + 1].x : (c == 1 ? dd[u + 1].y : (c == 2 ? dd[u + 1].z : dd[u + 1].w));
                    ts_split3_pair(x0, x1, qh[u >> 1], qm[u >> 1], ql[u >> 1]);
                }
                const bf16x2 ph = __builtin_bit_cast(bf16x2, qh), pm = __builtin_bit_cast(bf16x2, qm), pl = __builtin_bit_cast(bf16x2, ql);
                *reinterpret_cast<bf16x2*>(dbase + (size_t)(c ^ asw) * 16) = ph;
                *reinterpret_cast<bf16x2*>(dbase + (size_t)(4 * TS_BM + (c ^ asw)) * 16) = pm;
                *reinterpret_cast<bf16x2*>(dbase + (size_t)(8 * TS_BM + (c ^ asw)) * 16) = pl;
            }
        };
        f32x16 acc[2][2] = {{{0}, {0}}, {{0}, {0}}};
        if (nst > 0) {
            float4 da[2], db[2];
            load_d(da, s_lo); load_d(db, s_lo + 1);
            stage_d(da, 0, s_lo);
            ts_barrier();                                             // stage 0 is in place (the producers' half too)
            for (int j = 0; j < nst; j += 2) {
                const int s = s_lo + j;
                // step s on buffer 0; db = dH of step s + 1 (arrived or in flight), da <- step s + 2
                load_d(da, s + 2);
                if (!(dbg & 1)) ts_mfma_stage_sw(ts_smem, wm, wn, li, h, acc);
                if (j + 1 < nst) stage_d(db, 1, s + 1);
                ts_barrier();
                if (j + 1 >= nst) break;
                // step s + 1 on buffer 1; da = step s + 2, db <- step s + 3
                load_d(db, s + 3);
                if (!(dbg & 1)) ts_mfma_stage_sw(ts_smem + (size_t)TS_STAGE, wm, wn, li, h, acc);
                if (j + 2 < nst) stage_d(da, 0, s + 2);
                ts_barrier();
            }
        }
        float* C = slabs + (long long)slab * M * Kp;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                const int col = c0 + wn * 64 + jn * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < M && col < Kp) C[(long long)row * Kp + col] = acc[i][jn][r];
                }
            }
        }
        return;
    }
    if (CW == 4 && wid < 4) {
        // ------------------------------------------------------------------ consumers: MFMAs only
        const int wm = wid >> 1, wn2 = wid & 1;
        f32x16 acc[2][4] = {{{0}, {0}, {0}, {0}}, {{0}, {0}, {0}, {0}}};
        if (nst > 0) ts_barrier();                                // stage 0 is in place
        for (int j = 0; j < nst; ++j) {
            if (!(dbg & 1)) ts_mfma_stage_wide(ts_smem + (size_t)(j & 1) * TS_STAGE, wm, wn2, li, h, acc);     // (dbg: diagnosis only)
            ts_barrier();
        }
        float* C = slabs + (long long)slab * M * Kp;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) {
                const int col = c0 + wn2 * 128 + jn * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < M && col < Kp) C[(long long)row * Kp + col] = acc[i][jn][r];
                }
            }
        }
        return;
    }
    // ---------------------------------------------------------------------- producers: load, split, stage
    const int pt = tid - 64 * CW;
    const int fb = pt >> 6, fq = pt & 63;                           // feature task: k-group fb, column quad fq
    // dH half-task: 4-row group hq (0..7), column quad aq; the two halves of a 16-byte slot are neighbouring lanes, so that
    // sixteen lanes' 8-byte writes cover 128 different bytes
    const int hq = 2 * (pt >> 6) + (pt & 1), aq = (pt >> 1) & 31;
    const int fsw = (fq >> 1) & 3, asw = (aq >> 1) & 3;              // ts_sw(4 q + c) = 4 q + (c ^ ((q >> 1) & 3))
    const int s_last = s_hi - 1;
    const int cq = c0 + 4 * fq, mq = m0 + 4 * aq;
    int gidA[8], gidB[8];
    auto ids_of = [&](int (&gid)[8], int s) __attribute__((always_inline)) {
        const int r0 = (s < s_last ? s : s_last) * TS_BK + 8 * fb;
#pragma unroll
        for (int u = 0; u < 8; ++u) gid[u] = ga.ids[r0 + u < n ? r0 + u : n - 1];
    };
    // the indicator words are needed by the ONE column tile that holds the end of X only (workgroup-uniform)
    const bool tail_tile = ga.code != nullptr && c0 + TS_BN > ga.F;
    using Prod = typename std::conditional<PLANES, TsProdP, TsProd>::type;
    auto load = [&](Prod& v, const int (&gid)[8], int s) __attribute__((always_inline)) {
        const int r0 = (s < s_last ? s : s_last) * TS_BK;
        if constexpr (PLANES) {
#pragma unroll
            for (int u = 0; u < 8; ++u) v.p[u] = ts_plane_load(ga, (dbg & 2) ? 0 : gid[u], cq);
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) v.f[u] = ts_feat_load(ga, (dbg & 2) ? 0 : gid[u], cq);    // unconditional, clamped
        }
        if (tail_tile) {
#pragma unroll
            for (int u = 0; u < 8; ++u) v.cd[u] = ga.code[gid[u]];
        }
        if (CW == 4) {        // (CW == 8: the consumers load and stage dH)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + 4 * hq + u;
                v.d[u] = *reinterpret_cast<const float4*>(dH + (long long)(r < n ? r : n - 1) * M + (mq + 3 < M ? mq : 0));
            }
        }
    };
    auto stage = [&](const Prod& v, int buf, int s) __attribute__((always_inline)) {
        if (dbg & 8) return;
        uint4* st = ts_smem + (size_t)buf * TS_STAGE;
        uint4* fbase = st + TS_A_U4 + fb * TS_BN + 4 * fq;
        const int r0 = s * TS_BK;
        float4 dd[4];
        // interior steps of interior tiles (workgroup-uniform test) need no fix-up at all; the others fix by selects:
        // tail columns, rows beyond n, columns beyond Kp / M
        const bool edge = tail_tile || r0 + TS_BK > n || c0 + TS_BN > Kp || m0 + TS_BM > M;
        if constexpr (PLANES) {
            // the rows arrive split: a column's k-vector is the same 16-bit field of eight rows' plane words — moves, no arithmetic
            TsChunk3 pp[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                TsChunk3 t = v.p[u];
                if (edge) {
                    if (tail_tile) t = ts_plane_fix(t, ga, cq, ts_code_bits(ga, v.cd[u], epoch));
                    if (r0 + 8 * fb + u >= n || cq >= Kp) { t.h = make_uint2(0u, 0u); t.m = t.h; t.l = t.h; }
                }
                pp[u] = t;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                uint32_t wh[4], wm[4], wl[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {          // rows 2q (low half) and 2q + 1 (high half) of the k-vector
                    const uint32_t a0 = c < 2 ? pp[2 * q].h.x : pp[2 * q].h.y, a1 = c < 2 ? pp[2 * q + 1].h.x : pp[2 * q + 1].h.y;
                    const uint32_t b0 = c < 2 ? pp[2 * q].m.x : pp[2 * q].m.y, b1 = c < 2 ? pp[2 * q + 1].m.x : pp[2 * q + 1].m.y;
                    const uint32_t e0 = c < 2 ? pp[2 * q].l.x : pp[2 * q].l.y, e1 = c < 2 ? pp[2 * q + 1].l.x : pp[2 * q + 1].l.y;
                    if (c & 1) { wh[q] = (a0 >> 16) | (a1 & 0xffff0000u); wm[q] = (b0 >> 16) | (b1 & 0xffff0000u); wl[q] = (e0 >> 16) | (e1 & 0xffff0000u); }
                    else { wh[q] = (a0 & 0xffffu) | (a1 << 16); wm[q] = (b0 & 0xffffu) | (b1 << 16); wl[q] = (e0 & 0xffffu) | (e1 << 16); }
                }
                fbase[c ^ fsw] = make_uint4(wh[0], wh[1], wh[2], wh[3]);
                fbase[4 * TS_BN + (c ^ fsw)] = make_uint4(wm[0], wm[1], wm[2], wm[3]);
                fbase[8 * TS_BN + (c ^ fsw)] = make_uint4(wl[0], wl[1], wl[2], wl[3]);
            }
            if (CW == 8) return;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                dd[u] = v.d[u];
                if (edge && (r0 + 4 * hq + u >= n || mq + 3 >= M)) dd[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
        float4 ff[8];
        if (!edge) {
#pragma unroll
            for (int u = 0; u < 8; ++u) ff[u] = v.f[u];
#pragma unroll
            for (int u = 0; u < 4; ++u) dd[u] = v.d[u];
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float4 t = tail_tile ? ts_feat_fix(v.f[u], ga, cq, ts_code_bits(ga, v.cd[u], epoch)) : v.f[u];
                if (r0 + 8 * fb + u >= n || cq >= Kp) t = make_float4(0.f, 0.f, 0.f, 0.f);
                ff[u] = t;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                dd[u] = v.d[u];
                if (r0 + 4 * hq + u >= n || mq + 3 >= M) dd[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t qh[4], qm[4], ql[4];
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                const float x0 = c == 0 ? ff[u].x : (c == 1 ? ff[u].y : (c == 2 ? ff[u].z : ff[u].w));
                const float x1 = c == 0 ? ff[u + 1].x : (c == 1 ? ff[u + 1].y : (c == 2 ? ff[u + 1].z : ff[u + 1].w));
                ts_split3_pair(x0, x1, qh[u >> 1], qm[u >> 1], ql[u >> 1]);
            }
            const bf16x8 ph = __builtin_bit_cast(bf16x8, qh), pm = __builtin_bit_cast(bf16x8, qm), pl = __builtin_bit_cast(bf16x8, ql);
            fbase[c ^ fsw] = __builtin_bit_cast(uint4, ph);
            fbase[4 * TS_BN + (c ^ fsw)] = __builtin_bit_cast(uint4, pm);
            fbase[8 * TS_BN + (c ^ fsw)] = __builtin_bit_cast(uint4, pl);
        }
        }
        if (CW == 8) return;      // the dH image is the consumers'
        char* dbase = reinterpret_cast<char*>(st + (hq >> 1) * TS_BM + 4 * aq) + (hq & 1) * 8;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t qh[2], qm[2], ql[2];
#pragma unroll
            for (int u = 0; u < 4; u += 2) {
                const float x0 = c == 0 ? dd[u].x : (c == 1 ? dd[u].y : (c == 2 ? dd[u].z : dd[u].w));
                const float x1 = c == 0 ? dd[u + 1].x : (c == 1 ? dd[u + 1].y : (c == 2 ? dd[u + 1].z : dd[u + 1].w));
                ts_split3_pair(x0, x1, qh[u >> 1], qm[u >> 1], ql[u >> 1]);
            }
            const bf16x4 ph = __builtin_bit_cast(bf16x4, qh), pm = __builtin_bit_cast(bf16x4, qm), pl = __builtin_bit_cast(bf16x4, ql);
            *reinterpret_cast<bf16x4*>(dbase + (size_t)(c ^ asw) * 16) = ph;
            *reinterpret_cast<bf16x4*>(dbase + (size_t)(4 * TS_BM + (c ^ asw)) * 16) = pm;
            *reinterpret_cast<bf16x4*>(dbase + (size_t)(8 * TS_BM + (c ^ asw)) * 16) = pl;
        }
    };
    if (nst > 0) {
        Prod va, vb;
        ids_of(gidA, s_lo); ids_of(gidB, s_lo + 1);
        load(va, gidA, s_lo);
        ids_of(gidA, s_lo + 2);
        load(vb, gidB, s_lo + 1);
        ids_of(gidB, s_lo + 3);
        stage(va, 0, s_lo);
        ts_barrier();                                             // stage 0 is in place
        for (int j = 0; j < nst; j += 2) {
            const int s = s_lo + j;
            // step s: consumers work on buffer 0; vb = step s + 1 (in flight), gidA = ids of s + 2, gidB = ids of s + 3
            load(va, gidA, s + 2);
            ids_of(gidA, s + 4);
            if (j + 1 < nst) stage(vb, 1, s + 1);
            ts_barrier();
            if (j + 1 >= nst) break;
            // step s + 1: buffer 1; va = step s + 2 (in flight), gidB = ids of s + 3, gidA = ids of s + 4
            load(vb, gidB, s + 3);
            ids_of(gidB, s + 5);
            if (j + 2 < nst) stage(va, 0, s + 2);
            ts_barrier();
        }
    }
}

// ---------------------------------------------------------------------------------------------- dW, operand roles SWAPPED
// The tile of gemm_tsplit_dw_k is 128 (m) x 256 (c): Reddit's 608 feature columns take THREE column tiles, the third 37 % full —
// 6 workgroup tiles per slab where 4.75 are needed, and a tile costs its full time whatever it holds (201 us on the 77k-row hop).
// Here the SAME two LDS images change roles: the 128-wide A image holds feature columns, the 256-wide B image ALL of dH's
// (f_out <= 256), i.e. the output tile is 128 (c) x 256 (m): ceil(608 / 128) = 5 tiles per slab.  The host takes this form when
// it needs fewer tiles (kp mod 256 in (0, 128], f_out > 128); same slabs, same K order, same six products in the same order per
// accumulator: bit-identical to gemm_tsplit_dw_k<8>.
//   consumers (eight wavefronts, two per SIMD): the MFMAs, and dH's columns 0..127 -> B image (two rows x four columns per thread)
//   producers (four wavefronts): a feature half-task (4 gathered rows x 4 columns -> A image) and a dH half-task (4 rows x 4
//   columns of dH's columns 128..255 -> B image) per thread — 32 elements per step, as in the other form.
struct TsProdS { float4 f[4]; float4 d[4]; uint32_t cd[4]; };
// One (slab, column tile) of a problem: rows [32 s_lo, 32 s_hi) of the n live ones, feature columns [c0, c0 + 128) -> C (the slab's
// [M, Kp] matrix).  Shared by the one-problem launch and the several-problem launch below.
__device__ __forceinline__ void ts_dw_sw_tile(const float* __restrict__ dH, const int M, const TsGather& ga, const int Kp,
                                              float* __restrict__ C, const int n, const int s_lo, const int s_hi, const int c0,
                                              const int dbg) {
    extern __shared__ uint4 ts_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const uint32_t epoch = ga.d_epoch ? (*ga.d_epoch & 0xffffffu) : ga.epoch;
    const int nst = s_hi > s_lo ? s_hi - s_lo : 0;
    const int s_last = s_hi - 1;
    if (wid < 8) {
        // ------------------------------------------------------------------ consumers
        const int wm = wid >> 2, wn = wid & 3;
        const int hq2 = 4 * (tid >> 7) + (tid & 3), aq = (tid >> 2) & 31;     // row pair 0..15, column quad 0..31 (dH columns 0..127)
        const int asw = (aq >> 1) & 3, mq = 4 * aq;
        auto load_d = [&](float4 (&d)[2], int s) __attribute__((always_inline)) {
            const int r0 = (s < s_last ? s : s_last) * TS_BK + 2 * hq2;
#pragma unroll
            for (int u = 0; u < 2; ++u)
                d[u] = *reinterpret_cast<const float4*>(dH + (long long)(r0 + u < n ? r0 + u : n - 1) * M + (mq + 3 < M ? mq : 0));
        };
        auto stage_d = [&](const float4 (&d)[2], int buf, int s) __attribute__((always_inline)) {
            uint4* st = ts_smem + (size_t)buf * TS_STAGE;
            const int r0 = s * TS_BK + 2 * hq2;
            float4 dd[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) dd[u] = (r0 + u >= n || mq + 3 >= M) ? make_float4(0.f, 0.f, 0.f, 0.f) : d[u];
            char* dbase = reinterpret_cast<char*>(st + TS_A_U4 + (hq2 >> 2) * TS_BN + 4 * aq) + (hq2 & 3) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                uint32_t qh[1], qm[1], ql[1];
#pragma unroll
                for (int u = 0; u < 2; u += 2) {
                    const float x0 = c == 0 ? dd[u].x : (c == 1 ? dd[u].y : (c == 2 ? dd[u].z : dd[u].w));
                    const float x1 = c == 0 ? dd[u + 1].x : (c == 1 ? dd[u + 1].y : (c == 2 ? dd[u + 1].z : dd[u + 1].w));
                    ts_split3_pair(x0, x1, qh[u >> 1], qm[u >> 1], ql[u >> 1]);
                }
                const bf16x2 ph = __builtin_bit_cast(bf16x2, qh), pm = __builtin_bit_cast(bf16x2, qm), pl = __builtin_bit_cast(bf16x2, ql);
                *reinterpret_cast<bf16x2*>(dbase + (size_t)(c ^ asw) * 16) = ph;
                *reinterpret_cast<bf16x2*>(dbase + (size_t)(4 * TS_BN + (c ^ asw)) * 16) = pm;
                *reinterpret_cast<bf16x2*>(dbase + (size_t)(8 * TS_BN + (c ^ asw)) * 16) = pl;
            }
        };
        f32x16 acc[2][2] = {{{0}, {0}}, {{0}, {0}}};
        if (nst > 0) {
            float4 da[2], db[2];
            load_d(da, s_lo); load_d(db, s_lo + 1);
            stage_d(da, 0, s_lo);
            ts_barrier();
            for (int j = 0; j < nst; j += 2) {
                const int s = s_lo + j;
                load_d(da, s + 2);
                if (!(dbg & 1)) ts_mfma_stage_sw<true>(ts_smem, wm, wn, li, h, acc);      // (dbg = 0; see below)
                if (j + 1 < nst) stage_d(db, 1, s + 1);
                ts_barrier();
                if (j + 1 >= nst) break;
                load_d(db, s + 3);
                if (!(dbg & 1)) ts_mfma_stage_sw<true>(ts_smem + (size_t)TS_STAGE, wm, wn, li, h, acc);
                if (j + 2 < nst) stage_d(da, 0, s + 2);
                ts_barrier();
            }
        }
        // (dbg is always 0: the branch keeps the MFMA block a scheduling region of its own — without it the fragment reads are hoisted
        // over the staging and an accumulator tile is spilled INSIDE the loop, 216 bytes of scratch per lane; gemm_tsplit_dw_k's
        // diagnosis switch has the same effect there)
        // accumulator rows are feature columns (four consecutive ones per register quad), its column is dH's: 16-byte stores
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                const int m = wn * 64 + jn * 32 + li;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = c0 + wm * 64 + i * 32 + 8 * g + 4 * h;
                    if (m < M && col < Kp)
                        *reinterpret_cast<float4*>(C + (long long)m * Kp + col) =
                            make_float4(acc[i][jn][4 * g], acc[i][jn][4 * g + 1], acc[i][jn][4 * g + 2], acc[i][jn][4 * g + 3]);
                }
            }
        }
        return;
    }
    // ---------------------------------------------------------------------- producers
    const int pt = tid - 512;
    const int hq = 2 * (pt >> 6) + (pt & 1), aq = (pt >> 1) & 31;   // 4-row group 0..7, column quad 0..31 (both half-tasks)
    const int asw = (aq >> 1) & 3;
    const int cq = c0 + 4 * aq, mq = TS_BM + 4 * aq;                // feature columns of the tile; dH columns 128..255
    const bool tail_tile = ga.code != nullptr && c0 + TS_BM > ga.F;
    int gidA[4], gidB[4];
    auto ids_of = [&](int (&gid)[4], int s) __attribute__((always_inline)) {
        const int r0 = (s < s_last ? s : s_last) * TS_BK + 4 * hq;
#pragma unroll
        for (int u = 0; u < 4; ++u) gid[u] = ga.ids[r0 + u < n ? r0 + u : n - 1];
    };
    auto load = [&](TsProdS& v, const int (&gid)[4], int s) __attribute__((always_inline)) {
        const int r0 = (s < s_last ? s : s_last) * TS_BK + 4 * hq;
#pragma unroll
        for (int u = 0; u < 4; ++u) v.f[u] = ts_feat_load(ga, gid[u], cq);
        if (tail_tile) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v.cd[u] = ga.code[gid[u]];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            v.d[u] = *reinterpret_cast<const float4*>(dH + (long long)(r0 + u < n ? r0 + u : n - 1) * M + (mq + 3 < M ? mq : 0));
    };
    auto stage = [&](const TsProdS& v, int buf, int s) __attribute__((always_inline)) {
        uint4* st = ts_smem + (size_t)buf * TS_STAGE;
        const int r0 = s * TS_BK + 4 * hq;
        float4 ff[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float4 t = tail_tile ? ts_feat_fix(v.f[u], ga, cq, ts_code_bits(ga, v.cd[u], epoch)) : v.f[u];
            if (r0 + u >= n || cq >= Kp) t = make_float4(0.f, 0.f, 0.f, 0.f);
            ff[u] = t;
        }
        char* abase = reinterpret_cast<char*>(st + (hq >> 1) * TS_BM + 4 * aq) + (hq & 1) * 8;
        char* bbase = reinterpret_cast<char*>(st + TS_A_U4 + (hq >> 1) * TS_BN + TS_BM + 4 * aq) + (hq & 1) * 8;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t qh[2], qm[2], ql[2];
#pragma unroll
            for (int u = 0; u < 4; u += 2) {
                const float x0 = c == 0 ? ff[u].x : (c == 1 ? ff[u].y : (c == 2 ? ff[u].z : ff[u].w));
                const float x1 = c == 0 ? ff[u + 1].x : (c == 1 ? ff[u + 1].y : (c == 2 ? ff[u + 1].z : ff[u + 1].w));
                ts_split3_pair(x0, x1, qh[u >> 1], qm[u >> 1], ql[u >> 1]);
            }
            const bf16x4 ph = __builtin_bit_cast(bf16x4, qh), pm = __builtin_bit_cast(bf16x4, qm), pl = __builtin_bit_cast(bf16x4, ql);
            *reinterpret_cast<bf16x4*>(abase + (size_t)(c ^ asw) * 16) = ph;
            *reinterpret_cast<bf16x4*>(abase + (size_t)(4 * TS_BM + (c ^ asw)) * 16) = pm;
            *reinterpret_cast<bf16x4*>(abase + (size_t)(8 * TS_BM + (c ^ asw)) * 16) = pl;
        }
        float4 dd[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) dd[u] = (r0 + u >= n || mq + 3 >= M) ? make_float4(0.f, 0.f, 0.f, 0.f) : v.d[u];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t qh[2], qm[2], ql[2];
#pragma unroll
            for (int u = 0; u < 4; u += 2) {
                const float x0 = c == 0 ? dd[u].x : (c == 1 ? dd[u].y : (c == 2 ? dd[u].z : dd[u].w));
                const float x1 = c == 0 ? dd[u + 1].x : (c == 1 ? dd[u + 1].y : (c == 2 ? dd[u + 1].z : dd[u + 1].w));
                ts_split3_pair(x0, x1, qh[u >> 1], qm[u >> 1], ql[u >> 1]);
            }
            const bf16x4 ph = __builtin_bit_cast(bf16x4, qh), pm = __builtin_bit_cast(bf16x4, qm), pl = __builtin_bit_cast(bf16x4, ql);
            *reinterpret_cast<bf16x4*>(bbase + (size_t)(c ^ asw) * 16) = ph;
            *reinterpret_cast<bf16x4*>(bbase + (size_t)(4 * TS_BN + (c ^ asw)) * 16) = pm;
            *reinterpret_cast<bf16x4*>(bbase + (size_t)(8 * TS_BN + (c ^ asw)) * 16) = pl;
        }
    };
    if (nst > 0) {
        TsProdS va, vb;
        ids_of(gidA, s_lo); ids_of(gidB, s_lo + 1);
        load(va, gidA, s_lo);
        ids_of(gidA, s_lo + 2);
        load(vb, gidB, s_lo + 1);
        ids_of(gidB, s_lo + 3);
        stage(va, 0, s_lo);
        ts_barrier();
        for (int j = 0; j < nst; j += 2) {
            const int s = s_lo + j;
            load(va, gidA, s + 2);
            ids_of(gidA, s + 4);
            if (j + 1 < nst) stage(vb, 1, s + 1);
            ts_barrier();
            if (j + 1 >= nst) break;
            load(vb, gidB, s + 3);
            ids_of(gidB, s + 5);
            if (j + 2 < nst) stage(va, 0, s + 2);
            ts_barrier();
        }
    }
}

__global__ __launch_bounds__(768, 1) void gemm_tsplit_dw_sw_k(const float* __restrict__ dH, int M /* f_out <= 256 */, TsGather ga, int Kp,
                                                              float* __restrict__ slabs, int n_host, const int32_t* d_n,
                                                              int nslab, int ct, int dbg = 0) {
    const int n = eff_count(d_n, n_host);
    // XCD-aware map as in gemm_tsplit_dw_k: the ct tiles of a slab read the same rows and get ids that are equal mod 8
    const int grp = blockIdx.x / (8 * ct), within = blockIdx.x - grp * 8 * ct;
    const int slab = grp * 8 + (within & 7), tc = within >> 3;
    if (slab >= nslab) return;
    const int steps = (n + TS_BK - 1) / TS_BK;
    const int per = (steps + nslab - 1) / nslab;
    const int s_lo = slab * per, s_hi = (s_lo + per < steps) ? s_lo + per : steps;
    ts_dw_sw_tile(dH, M, ga, Kp, slabs + (long long)slab * M * Kp, n, s_lo, s_hi, tc * TS_BM, dbg);
}

// ---------------------------------------------------------------------------------------------- dW of SEVERAL problems, one launch
// A step's transform-first nets need up to three of these GEMMs at the same point of the backward pass (the sampler net at every
// hop — one weight, the hops' rows and indicator masks — and the log-Z net at hop 0): as three launches each gets its own 768
// workgroups' worth of slabs whatever its rows (Reddit: hop 0's 22k rows in 153 slabs of 4.5 K steps — prologue, epilogue and 95 MB
// of slab traffic each for 73 us of a launch), and three slab sums.  Here ONE launch covers all problems: the slab budget is dealt
// out by the LIVE row counts (every workgroup works the same partition out from the device counts), a slab belongs to one problem,
// problems with the same gradient share a slab set, and one launch sums every set in slab order.
#define TS_DW_MAXP 4
struct TsDwProb { const float* dH; const int32_t* ids; const uint32_t* code; uint32_t mask; int F; int Kp; int n_host;
                  const int32_t* d_n; float* slabs; int out; };
struct TsDwMulti { TsDwProb p[TS_DW_MAXP]; int count; int nslab_total; int ct; const float* X; int ldx; const uint32_t* d_epoch;
                   uint32_t epoch; int M; };
struct TsDwPart { int n[TS_DW_MAXP]; int base[TS_DW_MAXP + 1]; int per; };
__device__ __forceinline__ TsDwPart ts_dw_partition(const TsDwMulti& mp) {
    TsDwPart pt;
    int steps[TS_DW_MAXP], total = 0;
    {
        const int32_t* dq[TS_DW_MAXP]; int cq[TS_DW_MAXP]; bool wq[TS_DW_MAXP];
#pragma unroll
        for (int q = 0; q < TS_DW_MAXP; ++q) { dq[q] = mp.p[q].d_n; cq[q] = mp.p[q].n_host; wq[q] = q < mp.count; }
        eff_counts<TS_DW_MAXP>(dq, cq, wq, mp.X, pt.n);
    }
#pragma unroll
    for (int q = 0; q < TS_DW_MAXP; ++q) {
        steps[q] = (pt.n[q] + TS_BK - 1) / TS_BK;
        total += steps[q];
    }
    const int avail = mp.nslab_total - mp.count;                    // sum of ceil(steps / per) <= total / per + count <= nslab_total
    int per = (total + avail - 1) / avail;
    if (per < 4) per = 4;                                          // (a slab keeps at least four K steps: see ts_dw_slabs)
    pt.per = per;
    pt.base[0] = 0;
#pragma unroll
    for (int q = 0; q < TS_DW_MAXP; ++q) pt.base[q + 1] = pt.base[q] + (steps[q] + per - 1) / per;
    return pt;
}
__global__ __launch_bounds__(768, 1) void gemm_tsplit_dw_sw_multi_k(TsDwMulti mp, int dbg = 0) {
    const int ct = mp.ct;
    const int grp = blockIdx.x / (8 * ct), within = blockIdx.x - grp * 8 * ct;
    const int slab = grp * 8 + (within & 7), tc = within >> 3;
    const TsDwPart pt = ts_dw_partition(mp);
    if (slab >= pt.base[TS_DW_MAXP]) return;
    int q = 0;
#pragma unroll
    for (int u = 1; u < TS_DW_MAXP; ++u) if (slab >= pt.base[u]) q = u;
    // (problem q by selects over the by-value table: a dynamic index would put the table in scratch)
    TsDwProb pr = mp.p[0];
#pragma unroll
    for (int u = 1; u < TS_DW_MAXP; ++u) if (q == u) pr = mp.p[u];
    const int n = pt.n[0] * (q == 0) + pt.n[1] * (q == 1) + pt.n[2] * (q == 2) + pt.n[3] * (q == 3);
    const int base = pt.base[0] * (q == 0) + pt.base[1] * (q == 1) + pt.base[2] * (q == 2) + pt.base[3] * (q == 3);
    const int c0 = tc * TS_BM;
    if (c0 >= pr.Kp) return;
    const int steps = (n + TS_BK - 1) / TS_BK;
    const int s_lo = (slab - base) * pt.per, s_hi = (s_lo + pt.per < steps) ? s_lo + pt.per : steps;
    TsGather ga{mp.X, mp.ldx, pr.F, pr.ids, pr.code, mp.d_epoch, mp.epoch, pr.mask, nullptr};
    ts_dw_sw_tile(pr.dH, mp.M, ga, pr.Kp, pr.slabs + (long long)slab * mp.M * pr.Kp, n, s_lo, s_hi, c0, dbg);
}
// the slab sets of gemm_tsplit_dw_sw_multi_k -> their gradients; blockIdx.y = output.  An output's slabs are those of its problems,
// summed in slab order (eight loads in flight); layouts as in ts_slab_sum_k.
struct TsDwOuts { float* out[TS_DW_MAXP]; int kp[TS_DW_MAXP]; int out_ld[TS_DW_MAXP]; int accumulate[TS_DW_MAXP]; int count; };
__global__ __launch_bounds__(256) void ts_slab_sum_multi_k(TsDwMulti mp, TsDwOuts outs) {
    const int o = blockIdx.y;
    if (o >= outs.count) return;
    const TsDwPart pt = ts_dw_partition(mp);
    float* out_ = outs.out[0]; int kp = outs.kp[0], out_ld = outs.out_ld[0], accumulate = outs.accumulate[0];
#pragma unroll
    for (int u = 1; u < TS_DW_MAXP; ++u) if (o == u) { out_ = outs.out[u]; kp = outs.kp[u]; out_ld = outs.out_ld[u]; accumulate = outs.accumulate[u]; }
    const float* slabs = nullptr;
    int lo[TS_DW_MAXP], hi[TS_DW_MAXP];
#pragma unroll
    for (int q = 0; q < TS_DW_MAXP; ++q) {
        const bool mine = q < mp.count && mp.p[q].out == o;
        lo[q] = mine ? pt.base[q] : 0; hi[q] = mine ? pt.base[q + 1] : 0;
        if (mine) slabs = mp.p[q].slabs;
    }
    if (!slabs) return;
    const long long count = (long long)mp.M * kp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / kp), cc = (int)(i - (long long)m * kp);
        if (cc >= out_ld) continue;
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < TS_DW_MAXP; ++q) {
            int z = lo[q];
            for (; z + 8 <= hi[q]; z += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = slabs[(long long)(z + u) * count + i];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
            for (; z < hi[q]; ++z) acc += slabs[(long long)z * count + i];
        }
        float* dst = out_ + (long long)m * out_ld + cc;
        *dst = accumulate ? *dst + acc : acc;
    }
}

// slabs [nslab][count] -> out (+)= sum in slab order (count = f_out * Kp)
// kp / out_ld / out_cols: slab rows are kp wide; out rows have a pitch of out_ld floats and only their first out_cols columns
// are written (out_ld = out_cols = kp: the padded layout; = K: the parameter's own [f_out, K] gradient, no copy afterwards)
__global__ __launch_bounds__(256) void ts_slab_sum_k(const float* __restrict__ slabs, float* __restrict__ out_, long long count,
                                                     int nslab, int accumulate, int kp, int out_ld, int out_cols) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / kp), cc = (int)(i - (long long)m * kp);
        if (cc >= out_cols) continue;
        float* out = out_ + ((long long)m * out_ld + cc) - i;            // so that out[i] below is element (m, cc)
        float acc = 0.f;
        int z = 0;
        for (; z + 8 <= nslab; z += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slabs[(long long)(z + u) * count + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; z < nslab; ++z) acc += slabs[(long long)z * count + i];
        out[i] = accumulate ? out[i] + acc : acc;
    }
}

// out[i] = sum of the split-K forward's slabs in index order, the live rows only
__global__ __launch_bounds__(256) void ts_fwd_slab_sum_k(const float4* __restrict__ slabs, float4* __restrict__ out, int n_host,
                                                         const int32_t* d_n, int f4 /* f_out / 4 */, int nslab, long long stride4) {
    const long long count = (long long)eff_count(d_n, n_host) * f4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
        float4 acc = slabs[i];
        int z = 1;
        for (; z + 4 <= nslab; z += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = slabs[(long long)(z + u) * stride4 + i];
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        for (; z < nslab; ++z) { const float4 v = slabs[(long long)z * stride4 + i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        out[i] = acc;
    }
}

// ---------------------------------------------------------------------------------------------- host side
static inline bool ts_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Pre-split planes of a resident feature matrix (TsGather::planes) — DIAGNOSTIC BUILD ONLY: measured SLOWER than splitting the
// gathered rows in the K loop (Reddit 1.418 against 1.366 ms/step, same box, profiles/r04_workloads.txt): the kernels are bound by
// their MFMAs at the sustained clock, so the vector instructions the planes save were not on the critical path, while the planes
// are 1.5 x the bytes to gather, in 8-byte pieces.  Kept as an A/B form (GRAPES_FEATURE_PLANES=1 with GRAPES_DIAG=1).
#ifdef GRAPES_DIAG
namespace {
struct TsPlanesEntry { const float* X; int ldx; const unsigned char* planes; };
TsPlanesEntry g_ts_planes[8];
int g_ts_nplanes = 0;
}
static const unsigned char* ts_planes_of(const float* X, int ldx) {
    for (int i = 0; i < g_ts_nplanes; ++i)
        if (g_ts_planes[i].X == X && g_ts_planes[i].ldx == ldx) return g_ts_planes[i].planes;
    return nullptr;
}
extern "C" size_t grapes_feature_planes_bytes(int64_t n, int32_t x_stride) { return (size_t)(n > 0 ? n : 0) * 6u * (size_t)(x_stride > 0 ? x_stride : 0); }
extern "C" int grapes_feature_split_planes(const float* X, int64_t n, int32_t x_stride, void* planes, grapes_stream_t stream) {
    if (!X || !planes || n <= 0 || x_stride <= 0 || (x_stride & 3)) return GRAPES_EINVAL;
    if (!ts_aligned16(X) || (((uintptr_t)planes) & 7)) return GRAPES_EALIGN;
    long long blocks = (n * (x_stride >> 2) + 255) / 256; if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(ts_split_planes_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, X, (long long)n, x_stride,
                       (unsigned char*)planes);
    GRAPES_LAUNCH_CHECK();
    return 0;
}
// planes NULL: forget the matrix
extern "C" int grapes_feature_planes_register(const float* X, int32_t x_stride, const void* planes) {
    if (!X || x_stride <= 0) return GRAPES_EINVAL;
    for (int i = 0; i < g_ts_nplanes; ++i)
        if (g_ts_planes[i].X == X && g_ts_planes[i].ldx == x_stride) {
            if (planes) { g_ts_planes[i].planes = (const unsigned char*)planes; return 0; }
            g_ts_planes[i] = g_ts_planes[--g_ts_nplanes];
            return 0;
        }
    if (!planes) return 0;
    if (g_ts_nplanes >= 8) return GRAPES_EINVAL;
    g_ts_planes[g_ts_nplanes++] = TsPlanesEntry{X, x_stride, (const unsigned char*)planes};
    return 0;
}
#else
static inline const unsigned char* ts_planes_of(const float*, int) { return nullptr; }
#endif
static int ts_set_lds() {
    static bool done = false;
    if (done) return 0;
    const int bytes = 2 * TS_STAGE * (int)sizeof(uint4);
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tsplit_fwd_k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
#ifdef GRAPES_DIAG
    e = hipFuncSetAttribute((const void*)gemm_tsplit_fwd_k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_tsplit_dw_k<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
#endif
    e = hipFuncSetAttribute((const void*)gemm_tsplit_dw_k<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_tsplit_dw_k<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_tsplit_dw_sw_k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_tsplit_dw_sw_multi_k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_tsplit_fwd_pc_k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    done = true;
    return 0;
}
// operand roles swapped (gemm_tsplit_dw_sw_k: 128 feature columns x all of f_out per tile) when that takes fewer tiles
static inline bool ts_dw_swapped(int f_out, int kp) {
    static int sw = -1;      // GRAPES_TSPLIT_DW_SWAP = 0: never, 1: only with FEWER tiles, 2 (default): also with as many (A/B)
    if (sw < 0) { const char* e = grapes_tune_env("GRAPES_TSPLIT_DW_SWAP"); sw = e ? atoi(e) : 2; }
    // (as many tiles — Cora's 1436 columns, 12 either way — measured equal launch for launch; the swapped form is the one the
    // several-problem launch has: Cora 0.510 -> 0.493 ms/step)
    const int tn = grapes_div_up(kp, TS_BM), to = grapes_div_up(f_out, TS_BM) * grapes_div_up(kp, TS_BN);
    return sw && f_out <= TS_BN && f_out > TS_BM && (sw == 2 ? tn <= to : tn < to);
}
static inline int ts_dw_slabs(int f_out, int kp, int n_cap = 0x7fffffff) {
    const int tiles = ts_dw_swapped(f_out, kp) ? grapes_div_up(kp, TS_BM) : grapes_div_up(f_out, TS_BM) * grapes_div_up(kp, TS_BN);
    static int target = 0;     // workgroups aimed at (GRAPES_TSPLIT_DW_WGS; one workgroup per compute unit at a time: 144 KB of LDS)
    if (!target) { const char* e = grapes_tune_env("GRAPES_TSPLIT_DW_WGS"); target = e ? atoi(e) : 768; if (target < 8) target = 768; }      // (Reddit, ms/step: 256 -> 1.72, 512 -> 1.63, 640 -> 1.60, 768 -> 1.59, 896 -> 1.64, 1536 -> 1.69)
    int ns = target / tiles;
    // ... but a slab keeps at least four K steps of the row CAPACITY (Cora: 2.7k rows = 85 steps over 12 tiles — 64 slabs of one
    // step each were all prologue and slab traffic: 0.86 ms/step against 0.79 with 21)
    const int by_rows = grapes_div_up(grapes_div_up(n_cap, TS_BK), 4);
    if (ns > by_rows) ns = by_rows;
    return ns < 1 ? 1 : ns;
}

extern "C" int32_t grapes_split_gathered_available(int32_t f_out) {
    static int split = -1;
    if (split < 0) { const char* e = grapes_tune_env("GRAPES_GEMM_SPLIT"); split = e ? atoi(e) : 1; }
    return (split && f_out >= 32 && f_out <= TS_BN && f_out % 4 == 0) ? 1 : 0;
}
extern "C" size_t grapes_weight_split_image_bytes(int32_t k) {
    const size_t nk = (size_t)grapes_div_up(k > 0 ? k : 1, TS_BK);
    return nk * TS_B_U4 * sizeof(uint4);
}
/* image of W [f_out, k] (row stride ldw floats) for grapes_linear_fwd_gathered_split: written once per step */
extern "C" int grapes_weight_split_images(int32_t count, const float* const* w, const int32_t* ldw, const int32_t* f_out,
                                          const int32_t* k, void* const* image, float* const* w_pad, const int32_t* ld_pad,
                                          grapes_stream_t stream) {
    if (count < 1 || count > GRAPES_MAX_WEIGHT_IMAGES || !w || !ldw || !f_out || !k || !image) return GRAPES_EINVAL;
    TsImages im{};
    int gx = 0;
    for (int q = 0; q < count; ++q) {
        float* wp = w_pad ? w_pad[q] : nullptr;
        const int lp = (wp && ld_pad) ? ld_pad[q] : 0;
        if (!w[q] || !image[q] || f_out[q] <= 0 || f_out[q] > TS_BN || k[q] <= 0 || ldw[q] < k[q]) return GRAPES_EINVAL;
        if (wp && (lp < k[q] || lp > grapes_div_up(k[q], TS_BK) * TS_BK)) return GRAPES_EINVAL;
        if (!ts_aligned16(image[q])) return GRAPES_EALIGN;
        const int nk = grapes_div_up(k[q], TS_BK);
        im.W[q] = w[q]; im.ldw[q] = ldw[q]; im.N[q] = f_out[q]; im.K[q] = k[q]; im.img[q] = (uint4*)image[q]; im.nk[q] = nk;
        im.w_pad[q] = wp; im.ld_pad[q] = lp;
        const int g = grapes_div_up(nk * 4 * TS_BN, 256);
        gx = g > gx ? g : gx;
    }
    hipLaunchKernelGGL(ts_weight_image_k, dim3(gx, count), dim3(256), 0, (hipStream_t)stream, im);
    GRAPES_LAUNCH_CHECK();
    return 0;
}
extern "C" int grapes_weight_split_image_padded(const float* w, int32_t ldw, int32_t f_out, int32_t k, void* image,
                                                float* w_pad, int32_t ld_pad, grapes_stream_t stream) {
    return grapes_weight_split_images(1, &w, &ldw, &f_out, &k, &image, &w_pad, &ld_pad, stream);
}
extern "C" int grapes_weight_split_image(const float* w, int32_t ldw, int32_t f_out, int32_t k, void* image,
                                         grapes_stream_t stream) {
    return grapes_weight_split_image_padded(w, ldw, f_out, k, image, nullptr, 0, stream);
}
extern "C" int grapes_linear_fwd_gathered_split(const float* X, int32_t F, int32_t x_stride, const int32_t* ids,
                                                const uint32_t* ind_code, uint32_t epoch, const uint32_t* d_epoch,
                                                int32_t num_ind, const void* w_image, float* h, int32_t n, const int32_t* d_n,
                                                int32_t f_out, grapes_stream_t stream) {
    if (n < 0 || !grapes_split_gathered_available(f_out) || !w_image || !h) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!X || !ids || F <= 0 || num_ind < 0 || num_ind > 8 || x_stride < F || (x_stride & 3) || (num_ind > 0 && !ind_code)) return GRAPES_EINVAL;
    if (!ts_aligned16(X) || !ts_aligned16(w_image)) return GRAPES_EALIGN;
    int rc = ts_set_lds();
    if (rc) return rc;
    const int kp = (F + num_ind + 3) & ~3;
    const int nk = grapes_div_up(kp, TS_BK);
    TsGather ga{X, x_stride, F, ids, ind_code, d_epoch, epoch, 0xffu, ts_planes_of(X, x_stride)};
    const int ntiles = grapes_div_up(n, TS_BM);
    const int grid = ntiles > 256 ? 256 : ntiles;
    // GRAPES_TSPLIT_FWD_PC=1: the producer / consumer form — measured SLOWER (223 vs 194 us at 77k rows, profiles/r03_tsplit_ablation.txt):
    // the lockstep kernel is bound by its MFMAs at the sustained clock (MFMAs alone: 142 of 194 us), not by staging or load latency
    static int pc = -1;
    if (pc < 0) { const char* e = grapes_tune_env("GRAPES_TSPLIT_FWD_PC"); pc = e ? atoi(e) : 0; }
    if (pc)
        hipLaunchKernelGGL(gemm_tsplit_fwd_pc_k, dim3(grid), dim3(512), 2 * TS_STAGE * sizeof(uint4), (hipStream_t)stream, ga,
                           (const uint4*)w_image, nk, h, f_out, n, d_n, f_out, grapes_clock_reserve("gemm_tsplit_fwd_pc_k", grid, 8));
#ifdef GRAPES_DIAG
    else if (ga.planes)
        hipLaunchKernelGGL(gemm_tsplit_fwd_k<true>, dim3(grid), dim3(512), 2 * TS_STAGE * sizeof(uint4), (hipStream_t)stream, ga,
                           (const uint4*)w_image, nk, h, f_out, n, d_n, f_out, grapes_clock_reserve("gemm_tsplit_fwd_k", grid, 8));
#endif
    else
        hipLaunchKernelGGL(gemm_tsplit_fwd_k<false>, dim3(grid), dim3(512), 2 * TS_STAGE * sizeof(uint4), (hipStream_t)stream, ga,
                           (const uint4*)w_image, nk, h, f_out, n, d_n, f_out, grapes_clock_reserve("gemm_tsplit_fwd_k", grid, 8));
    GRAPES_LAUNCH_CHECK();
    return 0;
}
// ---- one or two nets over the same gathered rows, split tail (gemm_tsplit_fwd_tail_k): two launches
#define TS_TAIL_GRID 256
extern "C" size_t grapes_linear_fwd_gathered_split_tail_workspace_bytes(int32_t f_out) {
    return (size_t)TS_TAIL_GRID * TS_BM * (size_t)(f_out > 0 ? f_out : 1) * sizeof(float) + 16;
}
extern "C" int grapes_linear_fwd_gathered_split_tail(const float* X, int32_t F, int32_t x_stride, const int32_t* ids, int32_t nprob,
                                                     const uint32_t* const* ind_code, uint32_t epoch, const uint32_t* d_epoch,
                                                     const int32_t* num_ind, const void* const* w_image, float* const* h, int32_t n,
                                                     const int32_t* d_n, int32_t f_out, void* workspace, grapes_stream_t stream) {
    if (n < 0 || nprob < 1 || nprob > 2 || !grapes_split_gathered_available(f_out) || (f_out & 3) || !w_image || !h || !num_ind || !ind_code || !workspace)
        return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!X || !ids || F <= 0 || x_stride < F || (x_stride & 3)) return GRAPES_EINVAL;
    if (!ts_aligned16(X) || !ts_aligned16(workspace)) return GRAPES_EALIGN;
    int rc = ts_set_lds();
    if (rc) return rc;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_tsplit_fwd_tail_k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(2 * TS_STAGE * sizeof(uint4)));
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    TsProb2 pr{};
    int nk = 0;
    for (int q = 0; q < 2; ++q) {
        const int p = q < nprob ? q : 0;
        if (num_ind[p] < 0 || num_ind[p] > 8 || (num_ind[p] > 0 && !ind_code[p]) || !w_image[p] || !h[p]) return GRAPES_EINVAL;
        if (!ts_aligned16(w_image[p]) || !ts_aligned16(h[p])) return GRAPES_EALIGN;
        const int kp = (F + num_ind[p] + 3) & ~3;
        const int nkq = grapes_div_up(kp, TS_BK);
        if (q == 0) nk = nkq; else if (nkq != nk) return GRAPES_EINVAL;          // (the nets share the K steps)
        pr.ga[q] = TsGather{X, x_stride, F, ids, num_ind[p] > 0 ? ind_code[p] : nullptr, d_epoch, epoch, 0xffu, nullptr};
        pr.wimg[q] = (const uint4*)w_image[p]; pr.out[q] = h[p];
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gemm_tsplit_fwd_tail_k, dim3(TS_TAIL_GRID), dim3(512), 2 * TS_STAGE * sizeof(uint4), s, pr, nprob, nk, f_out, n, d_n,
                       f_out, (float*)workspace, grapes_clock_reserve("gemm_tsplit_fwd_k", TS_TAIL_GRID, 8));
    GRAPES_LAUNCH_CHECK();
    hipLaunchKernelGGL(ts_fwd_tail_sum_k, dim3(1024), dim3(256), 0, s, (const float4*)workspace, h[0], h[nprob > 1 ? 1 : 0], nprob, nk, f_out, n,
                       d_n, TS_TAIL_GRID);
    GRAPES_LAUNCH_CHECK();
    return 0;
}
// ... for FEW rows (n < ~8k): the same kernel split along K into slabs + their sum (two launches).  workspace: nslab * n * f_out floats.
static inline void ts_fwd_slabs(int n, int kp, int* nslab, int* kper) {
    const int ntiles = grapes_div_up(n, TS_BM), nk = grapes_div_up(kp, TS_BK);
    int want = 256 / (ntiles > 0 ? ntiles : 1); if (want < 1) want = 1;
    if (want > nk / 2) want = nk / 2 > 0 ? nk / 2 : 1;                   // at least two K steps per slab
    if (want > 16) want = 16;
    *kper = grapes_div_up(nk, want);
    *nslab = grapes_div_up(nk, *kper);
}
extern "C" size_t grapes_linear_fwd_gathered_split_k_workspace_bytes(int32_t n, int32_t kp, int32_t f_out) {
    int nslab = 1, kper = 1;
    ts_fwd_slabs(n > 0 ? n : 1, kp > 0 ? kp : 1, &nslab, &kper);
    return (size_t)nslab * (size_t)(n > 0 ? n : 1) * (size_t)f_out * sizeof(float) + 16;
}
extern "C" int grapes_linear_fwd_gathered_split_k(const float* X, int32_t F, int32_t x_stride, const int32_t* ids,
                                                  const uint32_t* ind_code, uint32_t epoch, const uint32_t* d_epoch,
                                                  int32_t num_ind, const void* w_image, float* h, int32_t n, const int32_t* d_n,
                                                  int32_t f_out, void* workspace, grapes_stream_t stream) {
    if (n < 0 || !grapes_split_gathered_available(f_out) || !w_image || !h || !workspace) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!X || !ids || F <= 0 || num_ind < 0 || num_ind > 8 || x_stride < F || (x_stride & 3) || (num_ind > 0 && !ind_code)) return GRAPES_EINVAL;
    if (!ts_aligned16(X) || !ts_aligned16(w_image) || !ts_aligned16(workspace) || !ts_aligned16(h)) return GRAPES_EALIGN;
    int rc = ts_set_lds();
    if (rc) return rc;
    const int kp = (F + num_ind + 3) & ~3;
    const int nk = grapes_div_up(kp, TS_BK);
    int nslab = 1, kper = nk;
    ts_fwd_slabs(n, kp, &nslab, &kper);
    TsGather ga{X, x_stride, F, ids, ind_code, d_epoch, epoch, 0xffu, ts_planes_of(X, x_stride)};
    const int ntiles = grapes_div_up(n, TS_BM);
    int grid = ntiles * nslab; if (grid > 512) grid = 512;
    const long long stride = (long long)n * f_out;
#ifdef GRAPES_DIAG
    if (ga.planes)
        hipLaunchKernelGGL(gemm_tsplit_fwd_k<true>, dim3(grid), dim3(512), 2 * TS_STAGE * sizeof(uint4), (hipStream_t)stream, ga,
                           (const uint4*)w_image, nk, nslab > 1 ? (float*)workspace : h, f_out, n, d_n, f_out, (unsigned long long*)nullptr, 0,
                           nslab, kper, stride);
    else
#endif
        hipLaunchKernelGGL(gemm_tsplit_fwd_k<false>, dim3(grid), dim3(512), 2 * TS_STAGE * sizeof(uint4), (hipStream_t)stream, ga,
                           (const uint4*)w_image, nk, nslab > 1 ? (float*)workspace : h, f_out, n, d_n, f_out, (unsigned long long*)nullptr, 0,
                           nslab, kper, stride);
    GRAPES_LAUNCH_CHECK();
    if (nslab > 1) {
        int g2 = grapes_div_up((long long)n * (f_out >> 2), 256); if (g2 > 2048) g2 = 2048;
        hipLaunchKernelGGL(ts_fwd_slab_sum_k, dim3(g2), dim3(256), 0, (hipStream_t)stream, (const float4*)workspace, (float4*)h, n, d_n,
                           f_out >> 2, nslab, stride >> 2);
        GRAPES_LAUNCH_CHECK();
    }
    return 0;
}
#ifdef GRAPES_DIAG
// diagnosis entry point (profiles/tsplit_ablation.py): the lockstep forward kernel with parts switched off (dbg bits: 1 no MFMAs,
// 2 every gathered row is row 0, 4 one W block for every step, 8 no staging) or the producer / consumer kernel (dbg = 16)
extern "C" int grapes_debug_tsplit_fwd(const float* X, int32_t F, int32_t x_stride, const int32_t* ids, const void* w_image, float* h,
                                       int32_t n, int32_t f_out, int32_t dbg, grapes_stream_t stream) {
    if (!X || !ids || !w_image || !h || n <= 0) return GRAPES_EINVAL;
    int rc = ts_set_lds();
    if (rc) return rc;
    const int kp = (F + 3) & ~3;
    const int nk = grapes_div_up(kp, TS_BK);
    TsGather ga{X, x_stride, F, ids, nullptr, nullptr, 0u, 0xffu, nullptr};
    const int ntiles = grapes_div_up(n, TS_BM);
    const int grid = ntiles > 256 ? 256 : ntiles;
    if (dbg & 16)
        hipLaunchKernelGGL(gemm_tsplit_fwd_pc_k, dim3(grid), dim3(512), 2 * TS_STAGE * sizeof(uint4), (hipStream_t)stream, ga,
                           (const uint4*)w_image, nk, h, f_out, n, (const int32_t*)nullptr, f_out, (unsigned long long*)nullptr);
    else
        hipLaunchKernelGGL(gemm_tsplit_fwd_k<false>, dim3(grid), dim3(512), 2 * TS_STAGE * sizeof(uint4), (hipStream_t)stream, ga,
                           (const uint4*)w_image, nk, h, f_out, n, (const int32_t*)nullptr, f_out, (unsigned long long*)nullptr, dbg);
    GRAPES_LAUNCH_CHECK();
    return 0;
}
extern "C" int grapes_debug_tsplit_dw(const float* dh, const float* X, int32_t F, int32_t x_stride, const int32_t* ids, int32_t n,
                                      int32_t f_out, void* workspace, int32_t dbg, grapes_stream_t stream) {
    if (!dh || !X || !ids || !workspace || n <= 0) return GRAPES_EINVAL;
    int rc = ts_set_lds();
    if (rc) return rc;
    const int kp = (F + 3) & ~3;
    TsGather ga{X, x_stride, F, ids, nullptr, nullptr, 0u, 0xffu, nullptr};
    const int mt = grapes_div_up(f_out, TS_BM), ct = grapes_div_up(kp, TS_BN);
    const int nslab = 512 / (mt * ct) < 1 ? 1 : 512 / (mt * ct);
    if (dbg & 32)      // the eight-consumer form
        hipLaunchKernelGGL((gemm_tsplit_dw_k<8, false>), dim3(mt * ct * grapes_div_up(nslab, 8) * 8), dim3(768), 2 * TS_STAGE * sizeof(uint4), (hipStream_t)stream,
                           dh, f_out, ga, kp, (float*)workspace, n, (const int32_t*)nullptr, nslab, mt, ct, dbg & ~32);
    else
        hipLaunchKernelGGL((gemm_tsplit_dw_k<4, false>), dim3(mt * ct * grapes_div_up(nslab, 8) * 8), dim3(512), 2 * TS_STAGE * sizeof(uint4), (hipStream_t)stream,
                           dh, f_out, ga, kp, (float*)workspace, n, (const int32_t*)nullptr, nslab, mt, ct, dbg);
    GRAPES_LAUNCH_CHECK();
    return 0;
}
#endif  // GRAPES_DIAG
extern "C" size_t grapes_linear_bwd_weight_gathered_split_workspace_bytes(int32_t k_pad, int32_t f_out) {
    return (size_t)ts_dw_slabs(f_out, k_pad) * k_pad * f_out * sizeof(float) + 64;
}
extern "C" int grapes_linear_bwd_weight_gathered_split(const float* dh, const float* X, int32_t F, int32_t x_stride,
                                                       const int32_t* ids, const uint32_t* ind_code, uint32_t epoch,
                                                       const uint32_t* d_epoch, int32_t num_ind, uint32_t ind_mask, float* dw,
                                                       int32_t n, const int32_t* d_n, int32_t f_out, int32_t accumulate,
                                                       void* workspace, grapes_stream_t stream) {
    return grapes_linear_bwd_weight_gathered_split_ld(dh, X, F, x_stride, ids, ind_code, epoch, d_epoch, num_ind, ind_mask, dw, 0, n, d_n,
                                                      f_out, accumulate, workspace, stream);
}
extern "C" int grapes_linear_bwd_weight_gathered_split_ld(const float* dh, const float* X, int32_t F, int32_t x_stride,
                                                          const int32_t* ids, const uint32_t* ind_code, uint32_t epoch,
                                                          const uint32_t* d_epoch, int32_t num_ind, uint32_t ind_mask, float* dw,
                                                          int32_t dw_ld, int32_t n, const int32_t* d_n, int32_t f_out,
                                                          int32_t accumulate, void* workspace, grapes_stream_t stream) {
    if (n < 0 || !grapes_split_gathered_available(f_out) || !dw) return GRAPES_EINVAL;
    if (dw_ld != 0 && dw_ld != F + num_ind && dw_ld != ((F + num_ind + 3) & ~3)) return GRAPES_EINVAL;
    if (!X || !ids || F <= 0 || num_ind < 0 || num_ind > 8 || x_stride < F || (x_stride & 3) || (num_ind > 0 && !ind_code)) return GRAPES_EINVAL;
    const int kp = (F + num_ind + 3) & ~3;
    hipStream_t s = (hipStream_t)stream;
    const int out_ld = dw_ld ? dw_ld : kp;            // rows of dw: kp floats (padded layout) or exactly F + num_ind (the parameter's own)
    if (n == 0) {
        if (!accumulate) { hipError_t e = grapes_zero_async(dw, (size_t)out_ld * f_out * sizeof(float), s); if (e) return (int)e; }
        return 0;
    }
    if (!dh || !workspace) return GRAPES_EINVAL;
    if (!ts_aligned16(X) || !ts_aligned16(dh)) return GRAPES_EALIGN;
    int rc = ts_set_lds();
    if (rc) return rc;
    TsGather ga{X, x_stride, F, ids, ind_code, d_epoch, epoch, ind_mask ? (ind_mask & 0xffu) : 0xffu, ts_planes_of(X, x_stride)};
    const int mt = grapes_div_up(f_out, TS_BM), ct = grapes_div_up(kp, TS_BN);
    const int nslab = ts_dw_slabs(f_out, kp, n);
    static int cw = 0;       // GRAPES_TSPLIT_DW_CW = 4 | 8 consumer wavefronts (A/B; see the kernel)
    if (!cw) { const char* e = grapes_tune_env("GRAPES_TSPLIT_DW_CW"); cw = (e && atoi(e) == 4) ? 4 : 8; }
#ifdef GRAPES_DIAG
    if (cw == 8 && ga.planes)
        hipLaunchKernelGGL((gemm_tsplit_dw_k<8, true>), dim3(mt * ct * grapes_div_up(nslab, 8) * 8), dim3(768), 2 * TS_STAGE * sizeof(uint4), s, dh, f_out, ga, kp,
                           (float*)workspace, n, d_n, nslab, mt, ct, 0);
    else
#endif
    if (cw == 8 && ts_dw_swapped(f_out, kp)) {
        const int cts = grapes_div_up(kp, TS_BM);
        hipLaunchKernelGGL(gemm_tsplit_dw_sw_k, dim3(cts * grapes_div_up(nslab, 8) * 8), dim3(768), 2 * TS_STAGE * sizeof(uint4), s, dh, f_out, ga, kp,
                           (float*)workspace, n, d_n, nslab, cts, 0);
    } else if (cw == 8)
        hipLaunchKernelGGL((gemm_tsplit_dw_k<8, false>), dim3(mt * ct * grapes_div_up(nslab, 8) * 8), dim3(768), 2 * TS_STAGE * sizeof(uint4), s, dh, f_out, ga, kp,
                           (float*)workspace, n, d_n, nslab, mt, ct, 0);
    else
        hipLaunchKernelGGL((gemm_tsplit_dw_k<4, false>), dim3(mt * ct * grapes_div_up(nslab, 8) * 8), dim3(512), 2 * TS_STAGE * sizeof(uint4), s, dh, f_out, ga, kp,
                           (float*)workspace, n, d_n, nslab, mt, ct, 0);
    GRAPES_LAUNCH_CHECK();
    const long long count = (long long)f_out * kp;
    int grid = grapes_div_up(count, 256); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(ts_slab_sum_k, dim3(grid), dim3(256), 0, s, (const float*)workspace, dw, count, nslab, accumulate, kp, out_ld, out_ld);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// ---- several problems, one launch (see gemm_tsplit_dw_sw_multi_k)
static inline int ts_dw_multi_slabs(int ct) { const int ns = 768 / ct; return ns < TS_DW_MAXP + 4 ? TS_DW_MAXP + 4 : ns; }
extern "C" int32_t grapes_linear_bwd_weight_gathered_split_multi_available(int32_t f_out, int32_t k_pad) {
    return (grapes_split_gathered_available(f_out) && k_pad > 0 && ts_dw_swapped(f_out, k_pad)) ? 1 : 0;
}
extern "C" size_t grapes_linear_bwd_weight_gathered_split_multi_workspace_bytes(int32_t outputs, int32_t k_pad_max, int32_t f_out) {
    if (outputs < 1 || k_pad_max < 4) return 0;
    return (size_t)outputs * ts_dw_multi_slabs(grapes_div_up(k_pad_max, TS_BM)) * k_pad_max * f_out * sizeof(float) + 64;
}
extern "C" int grapes_linear_bwd_weight_gathered_split_multi(int32_t count, const float* const* dh, const float* X, int32_t F,
                                                             int32_t x_stride, const int32_t* const* ids,
                                                             const uint32_t* const* ind_code, uint32_t epoch, const uint32_t* d_epoch,
                                                             const int32_t* num_ind, const uint32_t* ind_mask, float* const* dw,
                                                             const int32_t* dw_ld, const int32_t* n, const int32_t* const* d_n,
                                                             int32_t f_out, const int32_t* accumulate, void* workspace,
                                                             grapes_stream_t stream) {
    if (count < 1 || count > TS_DW_MAXP || !dh || !ids || !ind_code || !num_ind || !ind_mask || !dw || !dw_ld || !n || !d_n || !accumulate)
        return GRAPES_EINVAL;
    if (!X || F <= 0 || x_stride < F || (x_stride & 3) || !workspace || !grapes_split_gathered_available(f_out)) return GRAPES_EINVAL;
    if (!ts_aligned16(X) || !ts_aligned16(workspace)) return GRAPES_EALIGN;
    TsDwMulti mp{};
    TsDwOuts outs{};
    int kp_max = 0;
    for (int q = 0; q < count; ++q) {
        if (n[q] < 0 || !dw[q] || !ids[q] || !dh[q] || num_ind[q] < 0 || num_ind[q] > 8 || (num_ind[q] > 0 && !ind_code[q])) return GRAPES_EINVAL;
        if (!ts_aligned16(dh[q])) return GRAPES_EALIGN;
        const int kp = (F + num_ind[q] + 3) & ~3;
        if (dw_ld[q] != 0 && dw_ld[q] != F + num_ind[q] && dw_ld[q] != kp) return GRAPES_EINVAL;
        if (!ts_dw_swapped(f_out, kp)) return GRAPES_EINVAL;                 // (grapes_..._multi_available says so beforehand)
        if (kp > kp_max) kp_max = kp;
    }
    const int ct = grapes_div_up(kp_max, TS_BM), nslab = ts_dw_multi_slabs(ct);
    const size_t set_floats = (size_t)nslab * kp_max * f_out;
    for (int q = 0; q < count; ++q) {
        const int kp = (F + num_ind[q] + 3) & ~3;
        int o = -1;
        for (int r = 0; r < outs.count; ++r) if (outs.out[r] == dw[q]) o = r;
        if (o < 0) {
            o = outs.count++;
            outs.out[o] = dw[q]; outs.kp[o] = kp; outs.out_ld[o] = dw_ld[q] ? dw_ld[q] : kp; outs.accumulate[o] = accumulate[q];
        } else if (outs.kp[o] != kp || outs.out_ld[o] != (dw_ld[q] ? dw_ld[q] : kp)) {
            return GRAPES_EINVAL;                                                // one gradient, one layout
        }
        mp.p[q] = TsDwProb{dh[q], ids[q], num_ind[q] ? ind_code[q] : nullptr, ind_mask[q] ? (ind_mask[q] & 0xffu) : 0xffu, F, kp, n[q],
                           d_n[q], (float*)workspace + (size_t)o * set_floats, o};
    }
    mp.count = count; mp.nslab_total = nslab; mp.ct = ct; mp.X = X; mp.ldx = x_stride; mp.d_epoch = d_epoch; mp.epoch = epoch; mp.M = f_out;
    int rc = ts_set_lds();
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gemm_tsplit_dw_sw_multi_k, dim3(ct * grapes_div_up(nslab, 8) * 8), dim3(768), 2 * TS_STAGE * sizeof(uint4), s, mp, 0);
    GRAPES_LAUNCH_CHECK();
    int grid = grapes_div_up((long long)f_out * kp_max, 256); if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(ts_slab_sum_multi_k, dim3(grid, outs.count), dim3(256), 0, s, mp, outs);
    GRAPES_LAUNCH_CHECK();
    return 0;
}
