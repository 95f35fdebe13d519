// A7 (dense part): GCNConv.lin — H = X Wᵀ, and its backward dW = dHᵀ X, dX = dH W — on the fp32
// matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, 157 TFLOP/s peak; no TF32 on gfx950,
// and bf16 would break the 1e-5 parity target).  This is the only MFMA use on the path.
#include <type_traits>
#include "common.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GB_M 128
#define GB_N 128
#define GB_K 16
#define GB_LD 132   // LDS row stride (floats): rows stay 16 B aligned, ds_write conflicts <= 2-way (free)

// One operand tile (128 x 16) -> two float4 per thread.  KMAJOR: memory is [k][r] (r contiguous,
// the reduction index is the slow one); otherwise [r][k].
// Optional extras of the VEC path (used by the fused dW of an aggregate-first layer):
//   G        same layout as P: element kept only where G > 0 (ReLU backward applied while loading)
//   ones_at  KMAJOR only: column index that reads as 1.0 for every valid k (turns the padding column of
//            the B tile into a "ones" vector, so that output column ones_at = column sums of A = bias grad)
// A GEMM operand that is not a stored matrix but the frontier feature rows  feat(ids[r]) = [X[ids[r], 0:F] | indicator bits |
// 0-padding]  (main.py:199-204), read straight from the resident (row-padded) feature matrix: the gathered matrix the
// reference materialises on the host is never formed.  VEC tile loads only; the operand's `ld` is X's row stride.
struct GatherOp { const int32_t* ids; const uint32_t* code; const uint32_t* d_epoch; uint32_t epoch; int F; uint32_t mask; };

template <bool KMAJOR, bool VEC>
__device__ __forceinline__ void gemm_load_tile(const float* __restrict__ P, long long ld, int r0, int R, int k0,
                                               int kend, int tid, float4 (&v)[2], const float* __restrict__ G = nullptr,
                                               int ones_at = -1, const float* __restrict__ row_scale = nullptr,
                                               const float* __restrict__ col_vec = nullptr, float4* cs2 = nullptr,
                                               const GatherOp* go = nullptr, bool do_cs2 = false) {
    if (VEC) {
        // Launch-side contract of the VEC instantiation: ld % 4 == 0, 16 B aligned base, and the
        // contiguous extent equals ld, so a float4 is either wholly inside the operand or wholly
        // outside.  The loads are UNCONDITIONAL on a clamped address and masked afterwards: a branch
        // around a load makes hipcc wait for each one separately (no loads in flight together).
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int k, r;
            bool ok;
            long long off;
            if (KMAJOR) {
                k = k0 + (tid >> 5) + 8 * i;
                r = r0 + (tid & 31) * 4;
                ok = (k < kend) && (r + 3 < R);
                const int kc = k < kend ? k : kend - 1;
                const int rc = r + 3 < R ? r : 0;
                off = (long long)kc * ld + rc;
            } else {
                r = r0 + (tid >> 2) + 64 * i;
                k = k0 + (tid & 3) * 4;
                ok = (r < R) && (k + 3 < kend);
                const int rc = r < R ? r : R - 1;
                const int kc = k + 3 < kend ? k : 0;
                off = (long long)rc * ld + kc;
            }
            float4 t;
            if (go) {                    // gathered feature rows: row index = r (row-major operand) or k (k-major operand)
                const int rowi = KMAJOR ? (k < kend ? k : kend - 1) : (r < R ? r : R - 1);
                const int col0 = KMAJOR ? (r + 3 < R ? r : 0) : (k + 3 < kend ? k : 0);
                const int g = go->ids[rowi];
                if (col0 + 4 <= go->F) t = *reinterpret_cast<const float4*>(P + (long long)g * ld + col0);
                else t = feat_tail_chunk(P, ld, go->F, g, col0 >> 2, go->code,
                                         go->d_epoch ? (*go->d_epoch & 0xffffffu) : go->epoch, go->mask);
            } else if (KMAJOR && row_scale) {   // rank-1 operand: row_scale[k] * col_vec[r..r+3]
                const float rs = row_scale[k < kend ? k : kend - 1];
                const float4 cv = *reinterpret_cast<const float4*>(col_vec + (r + 3 < R ? r : 0));
                t = make_float4(rs * cv.x, rs * cv.y, rs * cv.z, rs * cv.w);
            } else {
                t = *reinterpret_cast<const float4*>(P + off);
            }
            if (G) {
                const float4 g = *reinterpret_cast<const float4*>(G + off);
                t.x = g.x > 0.f ? t.x : 0.f; t.y = g.y > 0.f ? t.y : 0.f;
                t.z = g.z > 0.f ? t.z : 0.f; t.w = g.w > 0.f ? t.w : 0.f;
                if (KMAJOR && do_cs2 && row_scale) {   // second column sum: sum_k row_scale[k] * gate[k][m]  (dW of the 1-wide head)
                    const float rs = ok ? row_scale[k < kend ? k : kend - 1] : 0.f;
                    cs2->x = fmaf(rs, g.x, cs2->x); cs2->y = fmaf(rs, g.y, cs2->y);
                    cs2->z = fmaf(rs, g.z, cs2->z); cs2->w = fmaf(rs, g.w, cs2->w);
                }
            }
            v[i].x = ok ? t.x : 0.f; v[i].y = ok ? t.y : 0.f; v[i].z = ok ? t.z : 0.f; v[i].w = ok ? t.w : 0.f;
            if (KMAJOR && ones_at >= 0 && k < kend) {
                if (r + 0 == ones_at) v[i].x = 1.f;
                if (r + 1 == ones_at) v[i].y = 1.f;
                if (r + 2 == ones_at) v[i].z = 1.f;
                if (r + 3 == ones_at) v[i].w = 1.f;
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KMAJOR) {
            const int k = k0 + (tid >> 5) + 8 * i;
            const int r = r0 + (tid & 31) * 4;
            if (k < kend) {
                const float* p = P + (long long)k * ld + r;
                if (VEC && r + 3 < R) {
                    v[i] = *reinterpret_cast<const float4*>(p);
                } else {
                    if (r + 0 < R) v[i].x = p[0];
                    if (r + 1 < R) v[i].y = p[1];
                    if (r + 2 < R) v[i].z = p[2];
                    if (r + 3 < R) v[i].w = p[3];
                }
            }
        } else {
            const int r = r0 + (tid >> 2) + 64 * i;
            const int k = k0 + (tid & 3) * 4;
            if (r < R) {
                const float* p = P + (long long)r * ld + k;
                if (VEC && k + 3 < kend) {
                    v[i] = *reinterpret_cast<const float4*>(p);
                } else {
                    if (k + 0 < kend) v[i].x = p[0];
                    if (k + 1 < kend) v[i].y = p[1];
                    if (k + 2 < kend) v[i].z = p[2];
                    if (k + 3 < kend) v[i].w = p[3];
                }
            }
        }
    }
}

template <bool KMAJOR>
__device__ __forceinline__ void gemm_store_tile(float (*S)[GB_LD], const float4 (&v)[2], int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (KMAJOR) {
            *reinterpret_cast<float4*>(&S[(tid >> 5) + 8 * i][(tid & 31) * 4]) = v[i];
        } else {
            const int r = (tid >> 2) + 64 * i, k = (tid & 3) * 4;
            S[k + 0][r] = v[i].x; S[k + 1][r] = v[i].y; S[k + 2][r] = v[i].z; S[k + 3][r] = v[i].w;
        }
    }
}

// Split-K chunk when the reduction length lives on the device (dW: K = number of frontier rows): the launch
// fixes the NUMBER of slabs (so that slabs x tiles = two workgroups per CU) and the chunk follows from K.
__host__ __device__ __forceinline__ int auto_kchunk(int K, int nslab) {
    int kc = (K + nslab - 1) / nslab;
    kc = (kc + 1) & ~1;
    return kc < 16 ? 16 : kc;
}

struct GemmEx {
    const float* bias;      // epilogue: + bias[n]                     (GCNConv bias of an aggregate-first layer)
    int relu;               // epilogue: max(., 0)
    const float* gate_a;    // A element kept only where gate_a > 0     (ReLU backward fused into the dW operand load)
    float* colsum;          // != NULL: column N of the B tile reads as ones, out column N (= sum_k A[k][m]) -> colsum[m]
    long long colsum_slab;  // split-K stride of colsum
    const float* row_scale; // with col_vec: A[k][m] = row_scale[k] * col_vec[m] (masked by gate_a) — the rank-1 gradient
    const float* col_vec;   //   dAct = dh2 (x) w2 of a 1-wide head is never materialised; A itself is not read
    int dbg;                // diagnosis only (grapes_debug_gemm_fwd): 1 no stores, 2 no operand reloads, 4 no MFMAs
    float* colsum2;         // rank-1 mode: colsum2[m] = sum_k row_scale[k] * gate_a[k][m] over this block's k range (split-K
                            //   stride colsum_slab) — dW of the 1-wide head, gathered while the gate tile is loaded anyway
    // K segments (rank-1 dW of SEVERAL hops that share the weights, in one launch): blockIdx.y = seg * slabs_per_seg + z;
    // segment 0 = the main arguments (B, gate_a, row_scale, d_K / K_host), segments 1..3 below.  Every slab is written
    // (an empty k range writes zeros), the slab reduction then sums all nseg * slabs_per_seg slabs.
    int nseg, slabs_per_seg;
    const float* seg_B[3]; const float* seg_gate[3]; const float* seg_rs[3]; const int32_t* seg_dK[3]; int seg_K[3];
    int gather;             // 1: operand A (row-major) is a GatherOp over `ga`, 2: operand B (k-major) is
    GatherOp ga;
    const float* out_scale; // epilogue: row m of the output times out_scale[m] (full-batch inference: H = X Wᵀ leaves the GEMM
                            //   already scaled by dinv[row], the operand of grapes_gcn_aggregate_fwd_prescaled)
};

// C[m][n] = sum_k Aop[m][k] * Bop[n][k].   128x128 tile / workgroup, 4 waves (2x2), each wave a
// 64x64 sub-tile = 2x2 MFMA 32x32 accumulators (64 accumulator VGPRs).  K is streamed in steps of 16
// through a double-buffered k-major LDS image (conflict-free ds_read_b32 operand fetch: lanes 0-31
// read 32 consecutive floats); the next step's global loads are in flight behind the 32 MFMAs of
// the current one.  blockIdx.x -> (tm, tn) keeps the N-tiles of one row panel on one XCD (ids
// differ by a multiple of 8) so the shared A panel is an L2 hit.
// blockIdx.y = split-K slab (dW): slab z covers k in [z*kchunk, (z+1)*kchunk) and writes C + z*slab.
template <bool A_KMAJOR, bool B_KMAJOR, bool VEC, int GATHER = 0 /* 1: operand A, 2: operand B is a GatherOp (ex.ga) */>
__global__ __launch_bounds__(256, 2) void gemm_mfma_f32_k(const float* __restrict__ A, const float* __restrict__ B,
                                                           float* __restrict__ C, int M_host, int N, int K_host,
                                                           long long lda, long long ldb, long long ldc,
                                                           const int32_t* d_M, const int32_t* d_K, int kchunk,
                                                           long long slab, int nt, int mt, GemmEx ex) {
    __shared__ __attribute__((aligned(16))) float As[2][GB_K][GB_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GB_K][GB_LD];
    const int M = eff_count(d_M, M_host);
    int K = eff_count(d_K, K_host);
    const float* gate_q = ex.gate_a;
    const float* rs_q = ex.row_scale;
    int zslab = blockIdx.y;
    if (ex.nseg > 1) {
        const int seg = blockIdx.y / ex.slabs_per_seg;
        zslab = blockIdx.y - seg * ex.slabs_per_seg;
        if (seg >= 1) {           // ternary chains, not indexed loads from the by-value struct (no scratch)
            B = seg == 1 ? ex.seg_B[0] : (seg == 2 ? ex.seg_B[1] : ex.seg_B[2]);
            gate_q = seg == 1 ? ex.seg_gate[0] : (seg == 2 ? ex.seg_gate[1] : ex.seg_gate[2]);
            rs_q = seg == 1 ? ex.seg_rs[0] : (seg == 2 ? ex.seg_rs[1] : ex.seg_rs[2]);
            const int32_t* dk = seg == 1 ? ex.seg_dK[0] : (seg == 2 ? ex.seg_dK[1] : ex.seg_dK[2]);
            const int kh = seg == 1 ? ex.seg_K[0] : (seg == 2 ? ex.seg_K[1] : ex.seg_K[2]);
            K = eff_count(dk, kh);
        }
    }
    const int bid = blockIdx.x;
    int tm, tn;
    if (gridDim.x == (unsigned)(mt * nt)) {   // few row panels (split-K dW): plain map, every launched block works
        tm = bid / nt; tn = bid - tm * nt;
    } else {                                  // XCD-aware: the N-tiles of one row panel get ids that differ by 8
        const int group = bid / (8 * nt), within = bid - group * 8 * nt;
        tn = within >> 3; tm = group * 8 + (within & 7);
    }
    const int m0 = tm * GB_M, n0 = tn * GB_N;
    if (m0 >= M) return;
    if (kchunk < 0) kchunk = auto_kchunk(K, -kchunk);      // balanced split-K over a device-side K
    const int kb = zslab * kchunk;
    int ke = kb + kchunk; if (ke > K) ke = K;
    if (kb >= K && gridDim.y > 1 && ex.nseg <= 1) return;  // (segmented: an empty slab is written as zeros)
    C += (long long)blockIdx.y * slab;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int li = lane & 31, lk = lane >> 5;

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    float4 cs2 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int nk = (ke > kb) ? (ke - kb + GB_K - 1) / GB_K : 0;
    if (nk > 0) {
        // Two register sets: while tile kt feeds the MFMAs out of LDS, tile kt+1 is (still) landing in one set
        // and the loads of tile kt+2 are issued into the other, so every global load has two K-steps of
        // matrix work (>= 64 MFMAs per wave) to cover its HBM latency.
        float4 ra0[2], rb0[2], ra1[2], rb1[2];
        const int ones_at = ex.colsum ? N : -1;
        const bool do_cs2 = ex.colsum2 && tn == 0;   // (a flag, not a nullable pointer: an address-taken local would live in scratch)
        const GatherOp* goa = GATHER == 1 ? &ex.ga : nullptr;     // (compile-time: the plain instantiations carry no trace of it)
        const GatherOp* gob = GATHER == 2 ? &ex.ga : nullptr;
#define GEMM_LOAD(RA, RB, KT)                                                                                           \
        gemm_load_tile<A_KMAJOR, VEC>(A, lda, m0, M, kb + (KT) * GB_K, ke, tid, RA, gate_q, -1, rs_q, ex.col_vec, &cs2, goa, do_cs2); \
        gemm_load_tile<B_KMAJOR, VEC>(B, ldb, n0, N, kb + (KT) * GB_K, ke, tid, RB, nullptr, ones_at, nullptr, nullptr, nullptr, gob)
#define GEMM_STEP(CUR, RNEXT_A, RNEXT_B, RFREE_A, RFREE_B, KT)                                                          \
        {                                                                                                               \
            if ((KT) + 2 < nk && !(ex.dbg & 2)) { GEMM_LOAD(RFREE_A, RFREE_B, (KT) + 2); }                              \
            _Pragma("unroll") for (int kk = 0; kk < GB_K; kk += 2) {                                                    \
                const float a0 = As[CUR][kk + lk][wm * 64 + li];                                                        \
                const float a1 = As[CUR][kk + lk][wm * 64 + 32 + li];                                                   \
                const float b0 = Bs[CUR][kk + lk][wn * 64 + li];                                                        \
                const float b1 = Bs[CUR][kk + lk][wn * 64 + 32 + li];                                                   \
                if (ex.dbg & 4) { asm volatile("" :: "v"(a0), "v"(a1), "v"(b0), "v"(b1)); continue; }                   \
                acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);                                   \
                acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);                                   \
                acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);                                   \
                acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);                                   \
            }                                                                                                           \
            if ((KT) + 1 < nk) {                                                                                        \
                gemm_store_tile<A_KMAJOR>(As[(CUR) ^ 1], RNEXT_A, tid);                                                 \
                gemm_store_tile<B_KMAJOR>(Bs[(CUR) ^ 1], RNEXT_B, tid);                                                 \
            }                                                                                                           \
            __syncthreads();                                                                                            \
        }
        GEMM_LOAD(ra0, rb0, 0);
        if (nk > 1) { GEMM_LOAD(ra1, rb1, 1); }
        gemm_store_tile<A_KMAJOR>(As[0], ra0, tid);
        gemm_store_tile<B_KMAJOR>(Bs[0], rb0, tid);
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            // even step: LDS[0] = tile kt, set 1 = tile kt+1 (in flight), set 0 is free
            GEMM_STEP(0, ra1, rb1, ra0, rb0, kt)
            if (kt + 1 < nk) {
                // odd step: LDS[1] = tile kt+1, set 0 = tile kt+2 (in flight), set 1 is free
                GEMM_STEP(1, ra0, rb0, ra1, rb1, kt + 1)
            }
        }
#undef GEMM_LOAD
#undef GEMM_STEP
    }
    if (A_KMAJOR && VEC && ex.colsum2 && tn == 0) {
        // thread (kl = tid >> 5, c4 = tid & 31) holds the partial sums of columns m0 + 4*c4 .. +3 over its k lanes:
        // combine the 8 k-lane groups in a fixed order through LDS (the tile buffers are free now)
        float* red = &As[0][0][0];
        __syncthreads();
        *reinterpret_cast<float4*>(&red[(tid >> 5) * 128 + (tid & 31) * 4]) = cs2;
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
#pragma unroll
            for (int g8 = 0; g8 < 8; ++g8) t += red[g8 * 128 + tid];
            if (m0 + tid < M) ex.colsum2[(long long)blockIdx.y * ex.colsum_slab + m0 + tid] = t;
        }
    }
    // D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int gm0 = m0 + wm * 64 + row, gm1 = gm0 + 32;
        const int gn0 = n0 + wn * 64 + li, gn1 = gn0 + 32;
        float v00 = acc00[r], v01 = acc01[r], v10 = acc10[r], v11 = acc11[r];
        if (ex.bias) {
            const float b0 = gn0 < N ? ex.bias[gn0] : 0.f, b1 = gn1 < N ? ex.bias[gn1] : 0.f;
            v00 += b0; v10 += b0; v01 += b1; v11 += b1;
        }
        if (ex.relu) { v00 = fmaxf(v00, 0.f); v01 = fmaxf(v01, 0.f); v10 = fmaxf(v10, 0.f); v11 = fmaxf(v11, 0.f); }
        if (ex.out_scale) {
            const float s0 = gm0 < M ? ex.out_scale[gm0] : 0.f, s1 = gm1 < M ? ex.out_scale[gm1] : 0.f;
            v00 *= s0; v01 *= s0; v10 *= s1; v11 *= s1;
        }
        if ((ex.dbg & 1) && v00 != 12345.678f) continue;      // diagnosis: keep the values live, skip the stores
        if (gm0 < M) {
            if (gn0 < N) C[(long long)gm0 * ldc + gn0] = v00;
            if (gn1 < N) C[(long long)gm0 * ldc + gn1] = v01;
        }
        if (gm1 < M) {
            if (gn0 < N) C[(long long)gm1 * ldc + gn0] = v10;
            if (gn1 < N) C[(long long)gm1 * ldc + gn1] = v11;
        }
        if (ex.colsum) {   // the "ones" column: column sums of A over this block's k range
            float* cs = ex.colsum + (long long)blockIdx.y * ex.colsum_slab;
            if (gn0 == N) { if (gm0 < M) cs[gm0] = acc00[r]; if (gm1 < M) cs[gm1] = acc10[r]; }
            if (gn1 == N) { if (gm0 < M) cs[gm0] = acc01[r]; if (gm1 < M) cs[gm1] = acc11[r]; }
        }
    }
}

// ---- W-stationary forward GEMM for the aggregate-first layers:  out[n, N] = act(X[n, K] Wᵀ + b),  K <= ~124, N <= 256.
// One workgroup per CU keeps ALL of W in LDS (K*N floats, loaded once) and streams 32-row panels of X through a
// double-buffered LDS image; wave w owns output columns [64w, 64w+64) of the panel (two 32x32 accumulators).  The next
// panel's global loads are issued before the current panel's 2*K/2 MFMAs.  Unlike the tiled kernel, whose 128x128
// tiles each reload a 53 KB slice of W, W is read once per CU, and rows are dealt in 32-row panels (1172 panels at
// n = 37.5k: 4.6 per CU, i.e. 8 % quantisation loss instead of 23 % with 128-row tiles).  Measured: 10-15 % faster
// than the tiled kernel at the step's shapes; the store phase of a panel is still exposed (one wavefront per SIMD).
// LDS layouts (conflict-free ds_read_b64 / ds_write_b64):  Ws[kq][h][n][s] = W[n][4kq + 2s + h],
// As[buf][kq][h][m][s] = X[m][4kq + 2s + h]  — lane (h = lane/32, i = lane%32) reads the float2 {s=0, s=1} it feeds to
// two consecutive v_mfma_f32_32x32x2_f32 (k pairs {4kq, 4kq+1} then {4kq+2, 4kq+3}: the SAME k order as the tiled
// kernel, so both kernels return bit-identical results).
#define WS_ROWS 32
#define WS_MS 33            // LDS row stride of a panel image (rows + 1): coalesced global chunks scatter without bank conflicts
// The k loop of one 32 x 64 wavefront tile, software-pipelined by hand over TWO fragment sets: the ds_read_b64s of
// quad kq+1 are issued BEFORE the four MFMAs of quad kq (sched_barrier keeps hipcc from sinking them back to their
// use, which would expose the LDS latency in every iteration: the matrix pipe then idles ~25 % of the time).
#define WS_FRAG_LOAD(A_, B0_, B1_, KQI)                                                                        \
    {                                                                                                          \
        const int kc_ = (KQI) < KQ_ ? (KQI) : KQ_ - 1;                                                         \
        A_ = *reinterpret_cast<const float2*>(Ap_ + (size_t)kc_ * as_);                                        \
        B0_ = *reinterpret_cast<const float2*>(Bp_ + (size_t)kc_ * bs_);                                       \
        B1_ = *reinterpret_cast<const float2*>(Bp_ + (size_t)kc_ * bs_ + 64);                                  \
    }
#define WS_FRAG_MFMA(A_, B0_, B1_, C0, C1)                                                                     \
    C0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.x, B0_.x, C0, 0, 0, 0);                                       \
    C1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.x, B1_.x, C1, 0, 0, 0);                                       \
    C0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.y, B0_.y, C0, 0, 0, 0);                                       \
    C1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.y, B1_.y, C1, 0, 0, 0);
#define WS_MFMA_LOOP(AP, BP, KQV, ASTEP, BSTEP, C0, C1)                                                        \
    {                                                                                                          \
        const float* Ap_ = (AP); const float* Bp_ = (BP);                                                      \
        const int KQ_ = (KQV), as_ = (ASTEP), bs_ = (BSTEP);                                                   \
        float2 fa0, fb00, fb01, fa1, fb10, fb11;                                                               \
        WS_FRAG_LOAD(fa0, fb00, fb01, 0)                                                                       \
        for (int kq_ = 0; kq_ < KQ_; kq_ += 2) {                                                               \
            WS_FRAG_LOAD(fa1, fb10, fb11, kq_ + 1)                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
            WS_FRAG_MFMA(fa0, fb00, fb01, C0, C1)                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
            WS_FRAG_LOAD(fa0, fb00, fb01, kq_ + 2)                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
            if (kq_ + 1 < KQ_) { WS_FRAG_MFMA(fa1, fb10, fb11, C0, C1) }                                       \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
        }                                                                                                      \
    }

__global__ __launch_bounds__(256, 1) void gemm_wstat_f32_k(const float* __restrict__ X, const float* __restrict__ W,
                                                           const float* __restrict__ bias, int relu,
                                                           float* __restrict__ out, int n_host, const int32_t* d_n,
                                                           int K, int N) {
    extern __shared__ __attribute__((aligned(16))) float ws_smem[];
    const int n = eff_count(d_n, n_host);
    const int npanels = (n + WS_ROWS - 1) / WS_ROWS;
    if ((int)blockIdx.x >= npanels) return;
    const int KQ = K >> 2;
    float* Ws = ws_smem;
    float* As = ws_smem + (size_t)K * N;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    // ---- panel staging: float4 chunk idx -> (row m = idx % 32, quad c = idx / 32)
    constexpr int MAXJ = 4;                       // 32 * KQ <= 1024 chunks  (K <= 128)
    float4 ra[MAXJ];
    const int nchunk = WS_ROWS * KQ;
    auto load_panel = [&](int p) {
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int idx = tid + 256 * j;
            const int m = idx & 31, c = idx >> 5;
            int gm = p * WS_ROWS + m; gm = gm < n ? gm : n - 1;                 // unconditional, clamped
            const int cc = c < KQ ? c : 0;
            ra[j] = *reinterpret_cast<const float4*>(X + (long long)gm * K + 4 * cc);
        }
    };
    auto stage_panel = [&](int buf) {
        float* Ab = As + (size_t)buf * K * WS_ROWS;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int idx = tid + 256 * j;
            if (idx < nchunk) {
                const int m = idx & 31, c = idx >> 5;
                *reinterpret_cast<float2*>(&Ab[((c * 2 + 0) * WS_ROWS + m) * 2]) = make_float2(ra[j].x, ra[j].z);
                *reinterpret_cast<float2*>(&Ab[((c * 2 + 1) * WS_ROWS + m) * 2]) = make_float2(ra[j].y, ra[j].w);
            }
        }
    };
    load_panel(blockIdx.x);
    // ---- W -> LDS (once): thread nn owns row nn of W; 8 independent float4 loads in flight per batch
    for (int nn = tid; nn < N; nn += 256) {               // ALL quads of the row in flight together: one round trip
        const float* wr = W + (long long)nn * K;
        float4 w4[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) w4[u] = *reinterpret_cast<const float4*>(wr + 4 * (u < KQ ? u : 0));
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            if (u < KQ) {
                *reinterpret_cast<float2*>(&Ws[((size_t)(u * 2 + 0) * N + nn) * 2]) = make_float2(w4[u].x, w4[u].z);
                *reinterpret_cast<float2*>(&Ws[((size_t)(u * 2 + 1) * N + nn) * 2]) = make_float2(w4[u].y, w4[u].w);
            }
        }
    }
    stage_panel(0);
    __syncthreads();
    const int n0 = wid * 64;
    const bool active = n0 < N;
    const float bias0 = (bias && active) ? bias[n0 + li] : 0.f;
    const float bias1 = (bias && active && n0 + 32 + li < N) ? bias[n0 + 32 + li] : 0.f;
    const float* Bp = Ws + ((size_t)h * N + n0 + li) * 2;
    const int astep = 2 * WS_ROWS * 2, bstep = 2 * N * 2;
    int it = 0;
    for (int p = blockIdx.x; p < npanels; p += gridDim.x, ++it) {
        const int buf = it & 1;
        const int pn = p + gridDim.x;
        // UNCONDITIONAL (clamped) prefetch: a load under a branch becomes a PHI whose copies — and therefore the
        // s_waitcnt — land right behind the load; this way it stays in flight behind this panel's MFMAs
        load_panel(pn < npanels ? pn : p);
        if (active) {
            f32x16 acc0 = {0}, acc1 = {0};
            const float* Ap = As + (size_t)buf * K * WS_ROWS + (h * WS_ROWS + li) * 2;
            WS_MFMA_LOOP(Ap, Bp, KQ, astep, bstep, acc0, acc1)
            // D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
            const bool full = (p + 1) * WS_ROWS <= n && n0 + 64 <= N;      // uniform: no per-row branches
            float* o = out + ((long long)p * WS_ROWS + 4 * h) * N + n0 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2);
                float v0 = acc0[r] + bias0, v1 = acc1[r] + bias1;
                if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                if (full) {
                    o[(long long)row * N] = v0;
                    o[(long long)row * N + 32] = v1;
                } else if (p * WS_ROWS + row + 4 * h < n) {
                    o[(long long)row * N] = v0;
                    if (n0 + 32 + li < N) o[(long long)row * N + 32] = v1;
                }
            }
        }
        if (pn < npanels) stage_panel(buf ^ 1);
        __syncthreads();
    }
}

// ---- Two panel streams per CU in anti-phase (8 wavefronts = two per SIMD): while one group of four wavefronts runs the
// 2*K/2 MFMAs of its panel, the other group stores its previous panel and stages its next one, then they swap (one
// workgroup barrier per phase).  With a single stream the store / staging phase of every panel is exposed (one
// wavefront per SIMD: nothing else to issue); here the matrix pipe only idles when the memory phase is the longer one.
// Needs W + 4 panel buffers in LDS (159.7 KB at K = 104, N = 256).  Same layouts and k order as gemm_wstat_f32_k.
__global__ __launch_bounds__(512, 1) void gemm_wstat2_f32_k(const float* __restrict__ X, const float* __restrict__ W,
                                                            const float* __restrict__ bias, int relu,
                                                            float* __restrict__ out, int n_host, const int32_t* d_n,
                                                            int K, int N) {
    extern __shared__ __attribute__((aligned(16))) float ws_smem[];
    const int n = eff_count(d_n, n_host);
    const int npanels = (n + WS_ROWS - 1) / WS_ROWS;
    if ((int)blockIdx.x >= npanels) return;
    const int KQ = K >> 2;
    float* Ws = ws_smem;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int grp = wid >> 2, wl = wid & 3, gt = tid & 255;
    const int NS = N + 1;                                                      // LDS row stride of W (see WS_MS)
    float* As = ws_smem + (size_t)K * NS + (size_t)grp * 2 * K * WS_MS;       // this group's two panel buffers
    const int li = lane & 31, h = lane >> 5;
    // this workgroup's panels: blockIdx.x + j * gridDim.x; group g owns j = g, g + 2, ...
    const int cnt_all = (npanels - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int cnt = (cnt_all - grp + 1) / 2;                                   // panels of this group
    auto panel_of = [&](int k) { return (int)blockIdx.x + (2 * k + grp) * (int)gridDim.x; };
    constexpr int MAXJ = 4;
    float4 ra[MAXJ];
    const int nchunk = WS_ROWS * KQ;
    // a panel is ONE contiguous block of 32*K floats: chunk idx -> (row m = idx / KQ, quad c = idx % KQ), coalesced
    auto load_panel = [&](int p) {
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int idx = gt + 256 * j;
            const int m = idx / KQ, c = idx - m * KQ;
            int gm = p * WS_ROWS + (m < WS_ROWS ? m : 0); gm = gm < n ? gm : n - 1;
            gm = gm < 0 ? 0 : gm;
            ra[j] = *reinterpret_cast<const float4*>(X + (long long)gm * K + 4 * c);
        }
    };
    auto stage_panel = [&](int buf) {
        float* Ab = As + (size_t)buf * K * WS_MS;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int idx = gt + 256 * j;
            if (idx < nchunk) {
                const int m = idx / KQ, c = idx - m * KQ;
                *reinterpret_cast<float2*>(&Ab[((c * 2 + 0) * WS_MS + m) * 2]) = make_float2(ra[j].x, ra[j].z);
                *reinterpret_cast<float2*>(&Ab[((c * 2 + 1) * WS_MS + m) * 2]) = make_float2(ra[j].y, ra[j].w);
            }
        }
    };
    load_panel(cnt > 0 ? panel_of(0) : (int)blockIdx.x);
    // ---- W -> LDS (once): W is one contiguous block of N*K floats; chunk f -> (row f / KQ, quad f % KQ), coalesced loads,
    //      all of a thread's chunks in flight together (one round trip)
    {
        const int wchunks = N * KQ;
        float4 w4[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int f = tid + 512 * u; w4[u] = *reinterpret_cast<const float4*>(W + 4 * (long long)(f < wchunks ? f : 0)); }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int f = tid + 512 * u;
            if (f < wchunks) {
                const int nn = f / KQ, q = f - nn * KQ;
                *reinterpret_cast<float2*>(&Ws[((size_t)(q * 2 + 0) * NS + nn) * 2]) = make_float2(w4[u].x, w4[u].z);
                *reinterpret_cast<float2*>(&Ws[((size_t)(q * 2 + 1) * NS + nn) * 2]) = make_float2(w4[u].y, w4[u].w);
            }
        }
        for (int f = tid + 512 * 16; f < wchunks; f += 512) {     // N*KQ > 8192 chunks (not reached for N <= 256, K <= 128)
            const float4 v = *reinterpret_cast<const float4*>(W + 4 * (long long)f);
            const int nn = f / KQ, q = f - nn * KQ;
            *reinterpret_cast<float2*>(&Ws[((size_t)(q * 2 + 0) * NS + nn) * 2]) = make_float2(v.x, v.z);
            *reinterpret_cast<float2*>(&Ws[((size_t)(q * 2 + 1) * NS + nn) * 2]) = make_float2(v.y, v.w);
        }
    }
    if (cnt > 0) stage_panel(0);
    __syncthreads();
    const int n0 = wl * 64;
    const bool active = n0 < N;
    const float bias0 = (bias && active) ? bias[n0 + li] : 0.f;
    const float bias1 = (bias && active && n0 + 32 + li < N) ? bias[n0 + 32 + li] : 0.f;
    const float* Bp = Ws + ((size_t)h * NS + n0 + li) * 2;
    const int astep = 2 * WS_MS * 2, bstep = 2 * NS * 2;
    const int cnt0 = (cnt_all + 1) / 2, cnt1 = cnt_all / 2;
    const int nphase = cnt1 > 0 ? (2 * cnt1 + 1 > 2 * cnt0 ? 2 * cnt1 + 1 : 2 * cnt0) : 2 * cnt0;   // uniform over the workgroup
    f32x16 acc0 = {0}, acc1 = {0};
    for (int ph = 0; ph < nphase; ++ph) {
        const int t = ph - grp;                     // group 1 runs one phase behind group 0
        if (t >= 0) {
            const int k = t >> 1;
            if (k < cnt) {
                if ((t & 1) == 0) {
                    // ---- matrix phase of panel k: next panel's loads first (unconditional, clamped), then the MFMAs
                    load_panel(panel_of(k + 1 < cnt ? k + 1 : k));
                    if (active && !(relu & 512)) {
                        acc0 = f32x16{0}; acc1 = f32x16{0};
                        const float* Ap = As + (size_t)(k & 1) * K * WS_MS + (h * WS_MS + li) * 2;
                        WS_MFMA_LOOP(Ap, Bp, KQ, astep, bstep, acc0, acc1)
                    }
                } else {
                    // ---- memory phase of panel k: its stores, then the next panel into the other buffer
                    const int p = panel_of(k);
                    if (active && !(relu & 256)) {
                        const bool full = (p + 1) * WS_ROWS <= n && n0 + 64 <= N;
                        float* o = out + ((long long)p * WS_ROWS + 4 * h) * N + n0 + li;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = (r & 3) + 8 * (r >> 2);
                            float v0 = acc0[r] + bias0, v1 = acc1[r] + bias1;
                            if (relu & 1) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                            if (full) {
                                o[(long long)row * N] = v0;
                                o[(long long)row * N + 32] = v1;
                            } else if (p * WS_ROWS + row + 4 * h < n) {
                                o[(long long)row * N] = v0;
                                if (n0 + 32 + li < N) o[(long long)row * N + 32] = v1;
                            }
                        }
                    }
                    if (k + 1 < cnt && !(relu & 1024)) stage_panel((k + 1) & 1);
                }
            }
        }
        __syncthreads();
    }
}

// ---- The same GEMM on the bf16 matrix pipe, at fp32 accuracy ("bf16x3"): gfx950 runs v_mfma_f32_32x32x2_f32 at the
// vector rate (64 FLOP/clk/SIMD, 1/16 of the bf16 rate) and has no xf32 form, so the fp32 kernel above is bound by
// its MFMAs (24 of its 26 us), not by the 54 MB it moves.  Here every fp32 operand is split EXACTLY into three bf16
// terms x = h + m + l (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m); each difference is exact in fp32, and
// |x - h - m - l| <= 2^-27 |x|), and a product a.b is the sum of the six cross terms hh, hm, mh, mm, hl, lh — each a
// bf16 x bf16 product, exact in the MFMA's fp32 datapath — accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The three
// dropped terms (ml, lm, ll) are below 2^-26 |a.b|, a quarter of the fp32 rounding of the product itself; measured
// error against fp64: tests/test_hip_parity.py (no larger than the fp32 kernel's).  Six bf16 MFMAs of 16 k each replace
// eight fp32 MFMAs of 2 k each: 2.7x less matrix-pipe time, which puts the kernel on its HBM traffic.
//   W never enters LDS: wave w keeps the split fragments of output columns [32w, 32w+32) for the WHOLE K in registers
// (3 planes x K/16 x 4 VGPRs = 84 at K = 104); X streams through 32-row panels, split into planes on the way into a
// double-buffered LDS image [plane][k/16][k/8 % 2][row][8 bf16] (a lane's fragment = one 16-byte read, a half-wave
// reads 512 contiguous bytes).  All 8 wavefronts (two per SIMD) work on the same panel: one overlaps the other's
// staging and stores.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define SP_ROWS 32
__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}
// split3 of TWO values at once: v_cvt_pk_bf16_f32 converts a pair per instruction (hipcc uses it with one live half for a scalar
// conversion), and the packed result IS the bf16 pair the images store.  Same operations per value as split3: bit-identical.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
__device__ __forceinline__ void split3_pair(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk_bf16(ra, rb);
    l = cvt_pk_bf16(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u));
}
// MFMA phase of one panel: two accumulator chains (the large cross terms, the small ones), summed at the end.
// Image layout (uint4 units): plane p, 8-column block kb = 2 ks + h, row r at  (p KS 2 + kb) SP_ROWS + (r ^ kb).  The XOR spreads
// the STAGING writes over the banks: a staging wavefront holds ~2.5 rows x 26 chunks, i.e. the same row at 13 different kb —
// 13 addresses exactly 512 bytes apart, one bank, a 13-way conflict per write instruction without it (the drain of those
// writes was ~1.5 us of a 5 us panel iteration).  The MFMA reads stay conflict-free: a permutation of 32 consecutive slots.
template <int KS>
__device__ __forceinline__ f32x16 wsplit_mfma(const uint4* __restrict__ A /* image base */, int h, int li, const bf16x8 (&wh)[KS],
                                              const bf16x8 (&wm)[KS], const bf16x8 (&wl)[KS]) {
    f32x16 acc0 = {0}, acc1 = {0};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int kb = 2 * ks + h;
        const uint4* Ak = A + (kb * SP_ROWS + (li ^ kb));
        const uint4 qh = Ak[0];
        const uint4 qm = Ak[(size_t)(KS * 2) * SP_ROWS];
        const uint4 ql = Ak[(size_t)(2 * KS * 2) * SP_ROWS];
        const bf16x8 ah = __builtin_bit_cast(bf16x8, qh), am = __builtin_bit_cast(bf16x8, qm), al = __builtin_bit_cast(bf16x8, ql);
        // W is the A operand and the panel the B operand: the tile comes out TRANSPOSED, D[n][m] — a lane holds one panel row m
        // and sixteen output columns in four runs of four, i.e. four 16-byte stores (not sixteen 4-byte ones) and a head
        // projection that is a sum inside the lane.  (The fragments' register layouts are the same for either role.)
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[ks], al, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[ks], ah, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[ks], ah, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[ks], am, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[ks], am, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[ks], ah, acc0, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] += acc1[r];
    return acc0;
}
// The MFMA phase of one panel with the STAGING of the next panel interleaved (round 5): after the six MFMAs of a k-step are issued —
// ~200 cycles of matrix-pipe time for this wavefront alone, twice that with the SIMD's other wavefront — the wavefront splits a
// slice of the panel it holds in registers (independent vector work: it co-executes with the matrix pipe) and writes finished chunks
// to the OTHER image.  The phases used to be separate — MFMAs (1.64 us) THEN staging (0.7 us), all eight wavefronts in lockstep,
// because one wavefront's two dependent chains cannot keep the pipe full while its partner stages; interleaved INSIDE a wavefront
// the staging needs no partner.  The previous panel's epilogue rides the same way (WSPLIT_DEFER_EPI).  Same operations on the same
// data: bit-identical.  sched_barrier between the k-steps pins the order (hipcc would otherwise sink all vector work behind the
// last MFMA).
template <int KS, int NCH, class Epi>
__device__ __forceinline__ f32x16 wsplit_mfma_stage(const uint4* __restrict__ A /* image read */, int h, int li, const bf16x8 (&wh)[KS],
                                                    const bf16x8 (&wm)[KS], const bf16x8 (&wl)[KS], const float4 (&ra)[NCH],
                                                    char* __restrict__ wimg /* image written */, const int (&soff)[NCH], Epi&& epi) {
    f32x16 acc0 = {0}, acc1 = {0};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int kb = 2 * ks + h;
        const uint4* Ak = A + (kb * SP_ROWS + (li ^ kb));
        const uint4 qh = Ak[0];
        const uint4 qm = Ak[(size_t)(KS * 2) * SP_ROWS];
        const uint4 ql = Ak[(size_t)(2 * KS * 2) * SP_ROWS];
        const bf16x8 ah = __builtin_bit_cast(bf16x8, qh), am = __builtin_bit_cast(bf16x8, qm), al = __builtin_bit_cast(bf16x8, ql);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[ks], al, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[ks], ah, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[ks], ah, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[ks], am, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[ks], am, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[ks], ah, acc0, 0, 0, 0);
        // chunk j of the next panel is split and written behind the MFMAs of k-step (j KS) / NCH (compile-time: the loops are unrolled)
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if (ks == (j * KS) / NCH) {
                uint2 p0, p1, p2;
                split3_pair(ra[j].x, ra[j].y, p0.x, p1.x, p2.x);
                split3_pair(ra[j].z, ra[j].w, p0.y, p1.y, p2.y);
                char* base = wimg + soff[j];
                *reinterpret_cast<uint2*>(base) = p0;
                *reinterpret_cast<uint2*>(base + (size_t)KS * 2 * SP_ROWS * 16) = p1;
                *reinterpret_cast<uint2*>(base + (size_t)2 * KS * 2 * SP_ROWS * 16) = p2;
            }
        }
        // ... and the EPILOGUE of the previous panel (its accumulators waited in registers): bias, ReLU, head products, gate bits or
        // tile stores — ~90 vector instructions that used to run after the last MFMA, with the matrix pipe idle
        if (ks == (KS > 2 ? KS - 2 : KS - 1)) epi();
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] += acc1[r];
    return acc0;
}
// Sum of each of a lane's 16 values over the 32 lanes of its half-wave, "transposing" as it goes: every step halves the
// values a lane carries (16 shuffles in all, not 80).  Returns the total of register index
// r = bit3 + 2 bit2 + 4 bit1 + 8 bit0 of the lane id; the order of the additions is fixed.
template <int HALF, int D>
__device__ __forceinline__ void rowsum_step(float (&v)[16], int lane) {     // compile-time register indices only
    const bool up = (lane & D) != 0;
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
        const float send = up ? v[i] : v[i + HALF];
        const float keep = up ? v[i + HALF] : v[i];
        v[i] = keep + __shfl_xor(send, D, 64);
    }
}
__device__ __forceinline__ float halfwave_rowsums16(float (&v)[16], int lane) {
    rowsum_step<8, 1>(v, lane);
    rowsum_step<4, 2>(v, lane);
    rowsum_step<2, 4>(v, lane);
    rowsum_step<1, 8>(v, lane);
    return v[0] + __shfl_xor(v[0], 16, 64);
}
// One transposed tile (see wsplit_mfma): lane (h, li) holds panel row li and the output columns n0 + 8 q + 4 h + {0..3},
// q = 0..3, in acc[4 q .. 4 q + 3].  PARTIAL = false: every row of the panel is live and the four stores are unconditional (no
// branches in the main loop).  HEAD: also this wavefront's share of  head[row] = sum_n out[row][n] * head_w[n]  (its 32
// columns): sixteen products inside the lane, one exchange with the other half-wave, into hpart[li].
// BITS: the tile is NOT written; instead ONE word per (row, wavefront) of ReLU gate bits — bit 16 h + 4 q + u of word
// [row][n0 / 32] is (out[row][n0 + 8 q + 4 h + u] > 0) — which is all the backward pass of a layer whose only consumer is a
// 1-wide head needs (gemm_dw_split_k): 32 bytes per row instead of 4 N.  (Measured at n = 37,500, N = 256: the 38 MB of
// activation stores are 5 us of the kernel's 14.5 in-graph — they drain at HBM write speed after the last MFMA — and the
// backward GEMM read them back.)  Both half-waves store the same word to the same address: an unconditional store
// instruction, see the vmcnt note in the kernel.
template <bool PARTIAL, bool HEAD, bool BITS = false>
__device__ __forceinline__ void wsplit_store(const f32x16& acc, const float4 (&b4)[4], int relu, float* __restrict__ o /* row li, column n0 + 4 h */,
                                             bool live /* PARTIAL: this lane's row exists */, const float4 (&hw4)[4],
                                             float* __restrict__ hpart, int lane, bool stamp = false,
                                             uint32_t* __restrict__ bo = nullptr /* BITS: the word of (row li, this wavefront) */) {
    float hs = 0.f;
    unsigned gb = 0u;
    (void)stamp;
    if (stamp) { asm volatile("" :: "v"(acc[0]), "v"(acc[15])); GRAPES_STAMP_NW(13); }     // (the accumulators have arrived)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 v = make_float4(acc[4 * q] + b4[q].x, acc[4 * q + 1] + b4[q].y, acc[4 * q + 2] + b4[q].z, acc[4 * q + 3] + b4[q].w);
        // (an INTEGER max on the bit pattern, not fmaxf: for a value that is not NaN max(as_int(v), 0) is relu(v) bit for bit (negative
        // floats are negative integers, -0 becomes +0), and it is one v_max_i32 — fmaxf, `v > 0 ? v : 0` and fmed3(v, 0, inf) all become
        // fmaxnum, in front of which hipcc canonicalises the operand: two v_max_f32 per value, 32 per panel and wavefront)
        if (relu & 1) {
            v.x = __int_as_float(max(__float_as_int(v.x), 0)); v.y = __int_as_float(max(__float_as_int(v.y), 0));
            v.z = __int_as_float(max(__float_as_int(v.z), 0)); v.w = __int_as_float(max(__float_as_int(v.w), 0));
        }
        if (BITS) gb |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u)) << (4 * q);
        else if (!PARTIAL || live) *reinterpret_cast<float4*>(o + 8 * q) = v;
        if (HEAD) { hs = fmaf(v.x, hw4[q].x, hs); hs = fmaf(v.y, hw4[q].y, hs); hs = fmaf(v.z, hw4[q].z, hs); hs = fmaf(v.w, hw4[q].w, hs); }
    }
    if (BITS) {
        const unsigned other = (unsigned)__shfl_xor((int)gb, 32, 64);
        const unsigned word = (lane & 32) ? (other | (gb << 16)) : (gb | (other << 16));
        if (!PARTIAL || live) *bo = word;
    }
    if (stamp) GRAPES_STAMP_NW(14);
    if (HEAD) {
        hs += __shfl_xor(hs, 32, 64);
        if (lane < 32) hpart[lane] = hs;
    }
}

#ifndef WSPLIT_INTERLEAVE        // 0 (make ab_wsplit -> libgrapes_hip_wsplit0.so): MFMAs, THEN staging — the separate phases of rounds 1-4, for A/B
#define WSPLIT_INTERLEAVE 1
#endif
#ifndef WSPLIT_DEFER_EPI         // 0: a panel's epilogue right behind its own MFMAs (A/B)
#define WSPLIT_DEFER_EPI 1
#endif
#ifndef WSPLIT_LAUNCH_BOUND      // diagnostic builds only (profiles/r03_fused_first_layer.txt): the register budget of a 768-thread workgroup
#define WSPLIT_LAUNCH_BOUND 512
#endif
// A SECOND, independent problem over the same rows in the same launch (alt.nwg0 < gridDim.x): workgroups [0, nwg0) run the
// kernel's own operands, workgroups [nwg0, gridDim.x) alt's (its own X view / W / bias / head, the same n, N and KS).  At hop 0
// the sampler net and the log-Z net transform the same 12.6k rows: alone each launch puts 1.5 panels on a workgroup behind an
// 8 us prologue (13.4 + 11.5 us); side by side, on half the workgroups each, the pair costs one prologue.
struct WsplitAlt { int nwg0; const float* X; const float* W; const float* bias; const float* head_w; float* head_out;
                   uint32_t* bits_out; int K; int ldx; };
template <int KS>
__global__ __launch_bounds__(WSPLIT_LAUNCH_BOUND, 1) void gemm_wsplit_f32_k(const float* X_ /* no __restrict__: hipcc treats loads through a
                                                            restrict const pointer as movable across anything */,
                                                            const float* __restrict__ W_, const float* __restrict__ bias_,
                                                            int relu, float* __restrict__ out, int n_host,
                                                            const int32_t* d_n, int K_, int N,
                                                            const float* __restrict__ head_w_, float* __restrict__ head_out_,
                                                            int ldx_ /* row stride of X in floats (>= K, multiple of 4) */,
                                                            uint32_t* __restrict__ bits_out_ /* see wsplit_store; NULL: out is written */,
                                                            unsigned long long* clk, WsplitAlt alt) {
    const unsigned long long clk0 = grapes_clock_begin(clk);
    const bool p1 = (int)blockIdx.x >= alt.nwg0;                          // (uniform) this workgroup works on the second problem
    const int bid = p1 ? (int)blockIdx.x - alt.nwg0 : (int)blockIdx.x;
    const int nwg = p1 ? (int)gridDim.x - alt.nwg0 : (alt.nwg0 < (int)gridDim.x ? alt.nwg0 : (int)gridDim.x);
    const float* X = p1 ? alt.X : X_;
    const float* __restrict__ W = p1 ? alt.W : W_;
    const float* __restrict__ bias = p1 ? alt.bias : bias_;
    const float* __restrict__ head_w = p1 ? alt.head_w : head_w_;
    float* __restrict__ head_out = p1 ? alt.head_out : head_out_;
    uint32_t* __restrict__ bits_out = p1 ? alt.bits_out : bits_out_;
    const int K = p1 ? alt.K : K_, ldx = p1 ? alt.ldx : ldx_;
    const int NW = N >> 5;
    constexpr int IMG = 3 * KS * 2 * SP_ROWS;          // uint4 per image (21 KB at KS = 7)
    constexpr int NCH = KS <= 8 ? 2 : 3;               // float4 chunks of a panel per thread (32 rows x K/4 chunks over 512 threads)
    __shared__ uint4 img[2 * IMG];                     // [buffer 2][plane 3][k-step KS][k/8 % 2][row 32]
    __shared__ float hpart[2][8][SP_ROWS];             // head partials of a panel, per wavefront (column group)
    // The live row count is REQUESTED here and consumed behind the W prologue: the bias / head words, this wavefront's W fragments and
    // the workgroup's first panel (inside the capacity) need no count, and with `if (bid >= npanels) return` in front of them the
    // count was a round trip of its own at the head of every launch.
    // (an unconditional load — without a device count a word of W stands in and is ignored: `if (d_n) n = *d_n` merges a loaded and
    // an immediate value, and the compiler waits for the load where the two paths meet)
    // ... and through an opaque per-lane offset of zero: a load the compiler knows to be uniform is moved into a scalar register
    // (a wait) right where it is issued.
    int zoff = 0;
    asm volatile("" : "+v"(zoff));
    const int n_vec = (d_n ? d_n : reinterpret_cast<const int32_t*>(W_))[zoff];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int KQ = K >> 2;
    const int n0 = wid * 32;
    const bool active = n0 < N;
    float4 b4[4], hw4[4];                              // this lane's sixteen output columns: n0 + 8 q + 4 h + {0..3}
    // (the conditions outside the loops: `(bias && active) ? load : 0` per q made the first pair of loads a round trip of its own
    // in front of the other six)
#pragma unroll
    for (int q = 0; q < 4; ++q) { b4[q] = make_float4(0.f, 0.f, 0.f, 0.f); hw4[q] = make_float4(0.f, 0.f, 0.f, 0.f); }
    if (bias && active) {
#pragma unroll
        for (int q = 0; q < 4; ++q) b4[q] = *reinterpret_cast<const float4*>(bias + n0 + 8 * q + 4 * h);
    }
    if (head_w && active) {
#pragma unroll
        for (int q = 0; q < 4; ++q) hw4[q] = *reinterpret_cast<const float4*>(head_w + n0 + 8 * q + 4 * h);
    }
    auto panel_of = [&](int j) { return bid + j * nwg; };
    // a panel is one contiguous block of 32*K floats: chunk idx -> (row idx / KQ, quad idx % KQ); 32*KQ <= 1024 chunks, two
    // per thread.  A thread without a second chunk repeats its first one — same address, same value — so that loads and
    // LDS writes are unconditional (hipcc sinks a load into the branch that uses it, i.e. behind the MFMAs).
    int goff[NCH], soff[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        int idx = tid + 512 * j;
        if (idx >= SP_ROWS * KQ) idx = tid < SP_ROWS * KQ ? tid : 0;
        const int m = idx / KQ, c = idx - m * KQ;
        goff[j] = m * ldx + 4 * c;
        soff[j] = ((c >> 1) * SP_ROWS + (m ^ (c >> 1))) * 16 + (c & 1) * 8;        // (kb = c >> 1: see wsplit_mfma)
    }
    // Two register sets for the panels in flight: the loads of panel j + 2 are issued at the TOP of iteration j (before its
    // MFMAs) into the set iteration j - 1 staged from, and consumed by the staging of iteration j + 1.  When a staging waits
    // for its panel, everything ahead of those loads in the (single, in-order) memory queue — the previous iteration's
    // stores — has had a whole MFMA phase to complete; with one set the loads sat behind the staging and in front of the
    // stores, and every iteration began by draining the previous one's stores (1 us of a 5 us iteration).
    constexpr bool PP = KS <= 8;                       // (K > 128: three chunks per thread — a second set would spill; one set, loads after the staging)
    float4 raA[NCH], raB[NCH];
    auto load_panel = [&](float4 (&ra)[NCH], int p) {  // full panels only
        const float* Xp = X + (long long)p * SP_ROWS * ldx;
#pragma unroll
        for (int j = 0; j < NCH; ++j) ra[j] = *reinterpret_cast<const float4*>(Xp + goff[j]);
    };
    auto stage_panel = [&](const float4 (&ra)[NCH], int buf) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            uint2 p0, p1, p2;
            split3_pair(ra[j].x, ra[j].y, p0.x, p1.x, p2.x);
            split3_pair(ra[j].z, ra[j].w, p0.y, p1.y, p2.y);
            char* base = reinterpret_cast<char*>(img + (size_t)buf * IMG) + soff[j];
            *reinterpret_cast<uint2*>(base) = p0;
            *reinterpret_cast<uint2*>(base + (size_t)KS * 2 * SP_ROWS * 16) = p1;
            *reinterpret_cast<uint2*>(base + (size_t)2 * KS * 2 * SP_ROWS * 16) = p2;
        }
    };
    // the first panel, in flight while W is loaded and split below — requested when it lies inside the CAPACITY (the count is not here yet)
    const bool spec0 = ((long long)bid + 1) * SP_ROWS <= (long long)n_host;
    if (spec0) load_panel(raA, bid);
    // zero the images once: the k >= K tail of the last k-step is never staged and must not hold NaN patterns
    for (int i = tid; i < 2 * IMG; i += 512) img[i] = make_uint4(0u, 0u, 0u, 0u);
    // ---- this wavefront's W fragments, split, for every k-step: lane (h, li) holds W[n0 + li][16 ks + 8 h + j], j < 8
    bf16x8 wh[KS], wm[KS], wl[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k0 = 16 * ks + 8 * h;
        float v[8];
        const float* wp = W + (long long)(active ? n0 + li : 0) * K + k0;
        const bool ldw = active && !(relu & 2048);
        const float4 a = (ldw && k0 + 4 <= K) ? *reinterpret_cast<const float4*>(wp) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 b = (ldw && k0 + 8 <= K) ? *reinterpret_cast<const float4*>(wp + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t x0, x1, x2; split3_pair(v[2 * j], v[2 * j + 1], x0, x1, x2);
            const bf16x2_t h2 = __builtin_bit_cast(bf16x2_t, x0), m2 = __builtin_bit_cast(bf16x2_t, x1), l2 = __builtin_bit_cast(bf16x2_t, x2);
            wh[ks][2 * j] = h2[0]; wh[ks][2 * j + 1] = h2[1]; wm[ks][2 * j] = m2[0]; wm[ks][2 * j + 1] = m2[1];
            wl[ks][2 * j] = l2[0]; wl[ks][2 * j + 1] = l2[1];
        }
    }
    const int n_dev = __builtin_amdgcn_readfirstlane(n_vec);
    const int n = (d_n && n_dev < n_host) ? (n_dev < 0 ? 0 : n_dev) : n_host;        // (eff_count's clamp)
    const int npanels = (n + SP_ROWS - 1) / SP_ROWS;
    if (bid >= npanels) return;                        // (uniform over the workgroup)
    const int cnt = (npanels - bid + nwg - 1) / nwg;
    // the one partial panel of the grid (n % 32 != 0) is the LAST panel of the workgroup that owns it: it runs after the
    // loop over full panels, on its own
    const bool own_partial = (n % SP_ROWS) != 0 && (npanels - 1) % nwg == bid;
    const int cntf = cnt - (own_partial ? 1 : 0);
    if (cntf > 0 && !spec0) load_panel(raA, panel_of(0));      // (cannot happen: a full live panel lies inside the capacity)
    __syncthreads();                                   // the zero fill is complete
    if (cntf > 0) {
        stage_panel(raA, 0);
        if (PP) load_panel(raB, panel_of(cntf > 1 ? 1 : 0)); else load_panel(raA, panel_of(cntf > 1 ? 1 : 0));
    }
    __syncthreads();
    // All eight wavefronts work on the same panel, one barrier per panel.  One iteration: MFMAs of panel j | staging of
    // panel j+1 (loaded an iteration ago) | loads of panel j+2 | stores of panel j.  vmcnt retires in issue order, so the
    // staging's wait for its loads must not have this panel's stores ahead of those loads: the loads are issued before
    // the stores and the wait is vmcnt(16) — sixteen stores stay in flight across it.  hipcc only emits that count if
    // every path into the loop carries the same pending sequence, hence: iteration 0 is peeled (same code, straight
    // line), nothing in the body is conditional (the last iterations re-load / re-stage a panel nobody reads: a branch
    // would also let hipcc sink the loads into it, behind the MFMAs), and there are separate copies of the loop for
    // wavefronts with output columns and (N < 256) wavefronts that only stage.  (Running the two wavefronts of a SIMD in
    // opposite half-order, or two panel streams in anti-phase, measured the same or slower: profiles/r01_split_gemm.txt.)
    const int nact = N >> 5;
    // combine of a panel's head partials: EVERY thread sums the 8 column groups of row tid % 32 (in order) and stores it —
    // sixteen identical stores per row, but one unconditional store instruction per wavefront: a store under a lane or
    // wavefront branch would hide the number of stores in flight from hipcc's vmcnt bookkeeping (see above)
    auto head_combine = [&](int buf, int p) {
        const int row = tid & (SP_ROWS - 1);
        float v[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) v[w] = hpart[buf][w][row];     // all eight reads in flight (the unused groups hold finite leftovers)
        float t = v[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) t += (w < nact) ? v[w] : 0.f;
        head_out[(long long)p * SP_ROWS + row] = t;
    };
    // WSPLIT_DEFER_EPI: panel j's accumulators wait in registers (accp) and its epilogue runs INSIDE iteration j + 1's MFMA phase; the head
    // partials of panel j are then complete after iteration j + 1 and combined at the top of iteration j + 2.  The first TWO iterations
    // are peeled (the second already carries an epilogue: every path into the loop has the same queue of loads and stores).
    constexpr bool DEFER = WSPLIT_DEFER_EPI && WSPLIT_INTERLEAVE && PP;
    f32x16 accp = {0};
    auto body = [&](int j, auto computes, auto with_head, auto first, auto odd, auto with_bits, auto has_prev) {
        constexpr bool HEAD = decltype(with_head)::value;
        constexpr bool BITS = decltype(with_bits)::value;
        constexpr bool ODD = decltype(odd)::value;         // iteration parity: loads into raA (even) / raB (odd), staging from the other
        constexpr bool EARLY = decltype(first)::value;     // a peeled iteration whose head combine has no panel yet
        constexpr bool PREV = decltype(has_prev)::value;   // DEFER: the previous panel's epilogue rides in this iteration
        const int p = panel_of(j);
        f32x16 acc;
        if (j >= 1 && j < 3) GRAPES_STAMP_NW((j - 1) * 6 + 0);
        // (the peeled iterations issue the same store — leftovers of hpart to the rows of their own panel, which a later iteration
        // overwrites from the same threads — so that every path into the loop carries the same queue of loads and stores)
        if (HEAD) {
            if (DEFER) head_combine(EARLY ? (j & 1) : (j & 1), EARLY ? p : panel_of(j - 2));          // (panel j - 2's partials: buffer (j - 2) & 1)
            else head_combine(EARLY ? (j & 1) : ((j - 1) & 1), EARLY ? p : panel_of(j - 1));
        }
        if (PP) { if (ODD) load_panel(raB, panel_of(j + 2 < cntf ? j + 2 : j)); else load_panel(raA, panel_of(j + 2 < cntf ? j + 2 : j)); }   // clamped to a full panel of this workgroup
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);                 // the loads stay ahead of the MFMAs
        if (j >= 1 && j < 3) GRAPES_STAMP_NW((j - 1) * 6 + 1);
        constexpr bool interleave = WSPLIT_INTERLEAVE && PP && decltype(computes)::value;      // (compile time: a run-time choice between the two loops cost the W fragments their registers)
        auto epilogue = [&](const f32x16& a_, int q) {     // panel q's outputs from its accumulators
            if (!(relu & 256))
                wsplit_store<false, HEAD, BITS>(a_, b4, relu, BITS ? nullptr : out + ((long long)q * SP_ROWS + li) * N + n0 + 4 * h, true, hw4,
                                                &hpart[DEFER ? ((j - 1) & 1) : (j & 1)][wid][0], lane, false,
                                                BITS ? bits_out + ((long long)q * SP_ROWS + li) * NW + wid : nullptr);
        };
        if (interleave) {
            char* wimg = reinterpret_cast<char*>(img + (size_t)((j + 1) & 1) * IMG);           // image last read in iteration j - 1 (a barrier ago)
            auto epi = [&]() { if (DEFER && PREV) epilogue(accp, panel_of(j - 1)); };
            if (ODD) acc = wsplit_mfma_stage<KS, NCH>(img + (size_t)(j & 1) * IMG, h, li, wh, wm, wl, raA, wimg, soff, epi);
            else acc = wsplit_mfma_stage<KS, NCH>(img + (size_t)(j & 1) * IMG, h, li, wh, wm, wl, raB, wimg, soff, epi);
        } else if (decltype(computes)::value && !(relu & 512)) acc = wsplit_mfma<KS>(img + (size_t)(j & 1) * IMG, h, li, wh, wm, wl);
        if (j >= 1 && j < 3) GRAPES_STAMP_NW((j - 1) * 6 + 2);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");                     // (IR-level code motion; sched_barrier only pins the machine scheduler)
        if (interleave) { }
        else if (PP) { if (ODD) stage_panel(raA, (j + 1) & 1); else stage_panel(raB, (j + 1) & 1); }     // image last read in iteration j - 1 (a barrier ago)
        else { stage_panel(raA, (j + 1) & 1); load_panel(raA, panel_of(j + 2 < cntf ? j + 2 : j)); }
        if (j >= 1 && j < 3) GRAPES_STAMP_NW((j - 1) * 6 + 3);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (j == 1) GRAPES_STAMP_NW(12);
        if (decltype(computes)::value) {
            if (DEFER && interleave) accp = acc;           // (its epilogue: inside the next iteration, or the drain below)
            else epilogue(acc, p);
        }
        if (j >= 1 && j < 3) GRAPES_STAMP_NW((j - 1) * 6 + 4);
        __syncthreads();
        if (j >= 1 && j < 3) GRAPES_STAMP_NW((j - 1) * 6 + 5);
    };
    auto loop = [&](auto computes, auto with_head, auto with_bits) {
        constexpr bool HEAD = decltype(with_head)::value;
        constexpr bool BITS = decltype(with_bits)::value;
        const std::true_type T{}; const std::false_type F{};
        if (!DEFER) {
            if (cntf > 0) body(0, computes, with_head, T, F, with_bits, F);
            int j = 1;
            for (; j + 1 < cntf; j += 2) {
                body(j, computes, with_head, F, T, with_bits, F);
                body(j + 1, computes, with_head, F, F, with_bits, F);
            }
            if (j < cntf) body(j, computes, with_head, F, T, with_bits, F);
            if (HEAD) { if (cntf > 0) head_combine((cntf - 1) & 1, panel_of(cntf - 1)); }
            return;
        }
        if (cntf > 0) body(0, computes, with_head, T, F, with_bits, F);
        if (cntf > 1) body(1, computes, with_head, T, T, with_bits, T);
        int j = 2;
        for (; j + 1 < cntf; j += 2) {
            body(j, computes, with_head, F, F, with_bits, T);
            body(j + 1, computes, with_head, F, T, with_bits, T);
        }
        if (j < cntf) body(j, computes, with_head, F, F, with_bits, T);
        if (cntf > 0) {    // the drain: panel cntf - 2's head combine, the last panel's epilogue and combine
            if (HEAD && cntf > 1) head_combine((cntf - 2) & 1, panel_of(cntf - 2));
            if (decltype(computes)::value && !(relu & 256))
                wsplit_store<false, HEAD, BITS>(accp, b4, relu, BITS ? nullptr : out + ((long long)panel_of(cntf - 1) * SP_ROWS + li) * N + n0 + 4 * h,
                                                true, hw4, &hpart[(cntf - 1) & 1][wid][0], lane, false,
                                                BITS ? bits_out + ((long long)panel_of(cntf - 1) * SP_ROWS + li) * NW + wid : nullptr);
            if (HEAD) { __syncthreads(); head_combine((cntf - 1) & 1, panel_of(cntf - 1)); __syncthreads(); }
        }
    };
    if (head_w && bits_out) { if (active) loop(std::true_type{}, std::true_type{}, std::true_type{}); else loop(std::false_type{}, std::true_type{}, std::true_type{}); }
    else if (head_w) { if (active) loop(std::true_type{}, std::true_type{}, std::false_type{}); else loop(std::false_type{}, std::true_type{}, std::false_type{}); }
    else        { if (active) loop(std::true_type{}, std::false_type{}, std::false_type{}); else loop(std::false_type{}, std::false_type{}, std::false_type{}); }
    if (own_partial) {
        const int p = npanels - 1, buf = cntf & 1;     // (buf: last read in iteration cntf - 2, two barriers ago)
#pragma unroll
        for (int j = 0; j < NCH; ++j) {                // same chunk map, rows clamped to the last live one
            int idx = tid + 512 * j;
            if (idx >= SP_ROWS * KQ) idx = tid < SP_ROWS * KQ ? tid : 0;
            const int m = idx / KQ, c = idx - m * KQ;
            int gm = p * SP_ROWS + m; gm = gm < n ? gm : n - 1;
            raA[j] = *reinterpret_cast<const float4*>(X + (long long)gm * ldx + 4 * c);
        }
        stage_panel(raA, buf);
        __syncthreads();
        if (active) {
            const f32x16 accp = wsplit_mfma<KS>(img + (size_t)buf * IMG, h, li, wh, wm, wl);
            float* o = bits_out ? nullptr : out + ((long long)p * SP_ROWS + li) * N + n0 + 4 * h;
            if (head_w && bits_out)
                wsplit_store<true, true, true>(accp, b4, relu, o, li < n - p * SP_ROWS, hw4, &hpart[buf][wid][0], lane, false,
                                               bits_out + ((long long)p * SP_ROWS + li) * NW + wid);
            else if (head_w) wsplit_store<true, true>(accp, b4, relu, o, li < n - p * SP_ROWS, hw4, &hpart[buf][wid][0], lane);
            else        wsplit_store<true, false>(accp, b4, relu, o, li < n - p * SP_ROWS, hw4, &hpart[buf][wid][0], lane);
        }
        if (head_w) {
            __syncthreads();
            const int row = tid & (SP_ROWS - 1);
            if (row < n - p * SP_ROWS) {
                float t = hpart[buf][0][row];
                for (int w = 1; w < nact; ++w) t += hpart[buf][w][row];
                head_out[(long long)p * SP_ROWS + row] = t;
            }
        }
    }
    grapes_clock_end(clk, clk0);
}
GRAPES_STAMP_SETTER(grapes_stamp_set_gemm)
static inline bool wsplit_ok(const float* x, const float* w, const void* out, int K, int N) {
    return K % 4 == 0 && K >= 4 && K <= 192 && N % 32 == 0 && N >= 32 && N <= 256 && (((uintptr_t)x) & 15) == 0 &&
           (((uintptr_t)w) & 15) == 0 && out != nullptr;
}
template <int KS>
static int launch_wsplit_ks(const float* x, const float* w, const float* bias, int relu, float* out, int n, const int32_t* d_n,
                            int K, int N, const float* head_w, float* head_out, hipStream_t s, int ldx, uint32_t* bits_out,
                            const WsplitAlt* second = nullptr) {
    const int npanels = grapes_div_up(n, SP_ROWS);
    int grid = npanels > 256 ? 256 : npanels;
    WsplitAlt alt{0x7fffffff, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
    if (second) {          // two problems over the same rows: half the workgroups each
        const int g0 = npanels > 128 ? 128 : npanels;
        alt = *second; alt.nwg0 = g0;
        grid = 2 * g0;
    }
    hipLaunchKernelGGL(gemm_wsplit_f32_k<KS>, dim3(grid), dim3(512), 0, s, x, w, bias, relu, out, n, d_n, K, N, head_w, head_out,
                       ldx, bits_out, grapes_clock_reserve("gemm_wsplit_f32_k", grid, 8), alt);
    GRAPES_LAUNCH_CHECK();
    return 0;
}
static int launch_wsplit(const float* x, const float* w, const float* bias, int relu, float* out, int n, const int32_t* d_n,
                         int K, int N, hipStream_t s, const float* head_w = nullptr, float* head_out = nullptr, int ldx = 0,
                         uint32_t* bits_out = nullptr, const WsplitAlt* second = nullptr) {
    if (ldx <= 0) ldx = K;
    if (bits_out && !head_w) return GRAPES_EINVAL;
    if (second) {          // (the pair: the instance of the LARGER K-step count runs both — a problem's steps past its K multiply the
                           // zeros of its image and weight fragments: the same sums; ogbn-arxiv: 132 and 128 columns = 9 and 8 steps)
        const int ka = (K + 15) / 16, kb = (second->K + 15) / 16;
        switch (ka > kb ? ka : kb) {
            case 7: return launch_wsplit_ks<7>(x, w, bias, relu, out, n, d_n, K, N, head_w, head_out, s, ldx, bits_out, second);
            case 8: return launch_wsplit_ks<8>(x, w, bias, relu, out, n, d_n, K, N, head_w, head_out, s, ldx, bits_out, second);
            case 9: return launch_wsplit_ks<9>(x, w, bias, relu, out, n, d_n, K, N, head_w, head_out, s, ldx, bits_out, second);
            default: return GRAPES_EINVAL;
        }
    }
    switch ((K + 15) / 16) {
        case 1: return launch_wsplit_ks<1>(x, w, bias, relu, out, n, d_n, K, N, head_w, head_out, s, ldx, bits_out);
        case 2: return launch_wsplit_ks<2>(x, w, bias, relu, out, n, d_n, K, N, head_w, head_out, s, ldx, bits_out);
        case 3: return launch_wsplit_ks<3>(x, w, bias, relu, out, n, d_n, K, N, head_w, head_out, s, ldx, bits_out);
        case 4: return launch_wsplit_ks<4>(x, w, bias, relu, out, n, d_n, K, N, head_w, head_out, s, ldx, bits_out);
        case 5: return launch_wsplit_ks<5>(x, w, bias, relu, out, n, d_n, K, N, head_w, head_out, s, ldx, bits_out);
        case 6: return launch_wsplit_ks<6>(x, w, bias, relu, out, n, d_n, K, N, head_w, head_out, s, ldx, bits_out);
        case 7: return launch_wsplit_ks<7>(x, w, bias, relu, out, n, d_n, K, N, head_w, head_out, s, ldx, bits_out);
        case 8: return launch_wsplit_ks<8>(x, w, bias, relu, out, n, d_n, K, N, head_w, head_out, s, ldx, bits_out);
        case 9: return launch_wsplit_ks<9>(x, w, bias, relu, out, n, d_n, K, N, head_w, head_out, s, ldx, bits_out);     // ogbn-arxiv: 128 + 3 -> 132
        case 10: return launch_wsplit_ks<10>(x, w, bias, relu, out, n, d_n, K, N, head_w, head_out, s, ldx, bits_out);
        case 11: return launch_wsplit_ks<11>(x, w, bias, relu, out, n, d_n, K, N, head_w, head_out, s, ldx, bits_out);
        case 12: return launch_wsplit_ks<12>(x, w, bias, relu, out, n, d_n, K, N, head_w, head_out, s, ldx, bits_out);
        default: return GRAPES_EINVAL;
    }
}

static inline size_t wstat_lds_bytes(int K, int N) { return ((size_t)K * N + 2 * (size_t)K * WS_ROWS) * sizeof(float); }
static inline bool wstat_ok(const float* x, const float* w, const float* out, int K, int N) {
    return K % 4 == 0 && K >= 4 && K <= 128 && N % 32 == 0 && N >= 32 && N <= 256 && wstat_lds_bytes(K, N) <= 160 * 1024 &&
           (((uintptr_t)x) & 15) == 0 && (((uintptr_t)w) & 15) == 0 && out != nullptr;
}
static int launch_wstat(const float* x, const float* w, const float* bias, int relu, float* out, int n, const int32_t* d_n,
                        int K, int N, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {   // opt in to > 64 KB of dynamic LDS (not a stream operation; done before any capture)
        hipError_t e = hipFuncSetAttribute((const void*)gemm_wstat_f32_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)gemm_wstat2_f32_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int npanels = grapes_div_up(n, WS_ROWS);
    int grid = npanels > 256 ? 256 : npanels;
    const size_t lds2 = ((size_t)K * (N + 1) + 4 * (size_t)K * WS_MS) * sizeof(float);
    static int mode = -1;
    if (mode < 0) { const char* e = grapes_tune_env("GRAPES_WSTAT_STREAMS"); mode = e ? atoi(e) : 2; }
    if (mode == 2 && lds2 <= 160 * 1024 && npanels >= 2 * 256) {       // two panel streams per CU in anti-phase
        static int dbgbits = -1;                                        // diagnosis: 256 no stores, 512 no MFMAs, 1024 no staging
        if (dbgbits < 0) { const char* e = grapes_tune_env("GRAPES_WSTAT_DBG"); dbgbits = e ? atoi(e) : 0; }
        hipLaunchKernelGGL(gemm_wstat2_f32_k, dim3(grid), dim3(512), lds2, s, x, w, bias, relu | dbgbits, out, n, d_n, K, N);
    } else {
        hipLaunchKernelGGL(gemm_wstat_f32_k, dim3(grid), dim3(256), wstat_lds_bytes(K, N), s, x, w, bias, relu, out, n, d_n, K, N);
    }
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// ---- "one-shot" GEMM for few rows (the classifier's <= B + hops*K rows):  C[M, N] = A[M, K] · Bop,  K <= 256.
// The tiled kernel streams K in 16-wide steps; with a handful of row panels every step is an exposed global-memory
// round trip (30-40 us for 0.13 GFLOP).  Here a workgroup owns a 32 x 64 tile and loads the WHOLE K extent of both
// operands (<= 96 KB of LDS) in ONE round trip: 512 threads, twelve 16-byte loads each, all in flight together (with two
// wavefronts a thread needed 48 loads, i.e. six dependent batches).  The eight wavefronts then split the tile's work as
// 2 column halves x 4 quarters of K (fp32 MFMAs out of LDS: K/8 per wavefront instead of K/2), and the four partial tiles
// are summed through LDS in a fixed order (k quarters 0, 1, 2, 3).
#define SK_ROWS 32
#define SK_COLS 64
#define SK_LDA (2 * SK_ROWS + 2)   // floats per (k/4, k%2) row of the A image: [32][2] + 2 of padding — rows a wavefront's transposing
#define SK_LDB (2 * SK_COLS + 2)   // (k-major) writes touch start in different banks (unpadded: an 8-way conflict per write)
#define SK_KMAX 256
#define SK_T 512
template <bool B_KMAJOR>
__device__ __forceinline__ void gemm_skinny_body(const float* __restrict__ A, const float* __restrict__ B,
                                                 float* __restrict__ C, int M_host, const int32_t* d_M, int N, int K,
                                                 long long lda, long long ldb, long long ldc,
                                                 const float* __restrict__ bias, int relu, int bx, int by, float* sk_smem) {
    const int m0 = bx * SK_ROWS, n0 = by * SK_COLS;
    const int KQ = (K + 3) >> 2;
    float* As = sk_smem;                                   // [KQ][2][32][2]
    float* Bs = sk_smem + (size_t)KQ * 2 * SK_LDA;         // [KQ][2][64][2 (+pad)]
    float* Ps = Bs + (size_t)KQ * 2 * SK_LDB;              // [4 k-quarters][2 column halves][16][64] partial tiles
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const bool avec = (lda % 4 == 0) && (K % 4 == 0) && ((((uintptr_t)A) & 15) == 0);
    const bool bvec = B_KMAJOR ? ((ldb % 4 == 0) && (N % 4 == 0) && ((((uintptr_t)B) & 15) == 0))
                               : ((ldb % 4 == 0) && (K % 4 == 0) && ((((uintptr_t)B) & 15) == 0));
    // ---- both operands' loads first (the B tile does not depend on the live row count; the A rows are clamped inside the
    // capacity and re-clamped to the live count below only where it matters: rows past M are never stored)
    constexpr int NA = SK_ROWS * (SK_KMAX / 4) / SK_T, NB = SK_COLS * (SK_KMAX / 4) / SK_T;     // 4 and 8 chunks per thread
    float4 va[NA], vb[NB];
    if (avec) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int idx = tid + SK_T * u;
            const int m = idx & 31, c = idx >> 5;
            int gm = m0 + m; gm = gm < M_host ? gm : M_host - 1;
            va[u] = *reinterpret_cast<const float4*>(A + (long long)gm * lda + 4 * (c < KQ ? c : 0));
        }
    }
    if (bvec) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int idx = tid + SK_T * u;
            if (!B_KMAJOR) {                               // B[n][k], k contiguous
                const int nn = idx & 63, c = idx >> 6;
                int gn = n0 + nn; gn = gn < N ? gn : N - 1;
                vb[u] = *reinterpret_cast<const float4*>(B + (long long)gn * ldb + 4 * (c < KQ ? c : 0));
            } else {                                       // B[k][n], n contiguous
                const int n4 = idx & 15, k = idx >> 4;
                const int gn = n0 + 4 * n4;
                vb[u] = *reinterpret_cast<const float4*>(B + (long long)(k < K ? k : 0) * ldb + (gn + 3 < N ? gn : 0));
            }
        }
    }
    const int M = eff_count(d_M, M_host);
    if (m0 >= M) return;                                   // (uniform)
    // ---- A panel: element (m, k) -> As[((k>>2)*2 + (k&1)) * 32 + m][(k>>1)&1]
    if (avec) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int idx = tid + SK_T * u;
            const int m = idx & 31, c = idx >> 5;
            if (c < KQ) {
                *reinterpret_cast<float2*>(&As[(c * 2 + 0) * SK_LDA + m * 2]) = make_float2(va[u].x, va[u].z);
                *reinterpret_cast<float2*>(&As[(c * 2 + 1) * SK_LDA + m * 2]) = make_float2(va[u].y, va[u].w);
            }
        }
    } else {
        for (int idx = tid; idx < SK_ROWS * KQ * 4; idx += SK_T) {
            const int k = idx % (KQ * 4), m = idx / (KQ * 4);
            int gm = m0 + m; gm = gm < M ? gm : M - 1;
            const float v = k < K ? A[(long long)gm * lda + k] : 0.f;
            As[((k >> 2) * 2 + (k & 1)) * SK_LDA + m * 2 + ((k >> 1) & 1)] = v;
        }
    }
    // ---- B tile: element (k, n) -> Bs[((k>>2)*2 + (k&1)) * 64 + n][(k>>1)&1]
    if (bvec) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int idx = tid + SK_T * u;
            if (!B_KMAJOR) {
                const int nn = idx & 63, c = idx >> 6;
                if (c < KQ) {
                    *reinterpret_cast<float2*>(&Bs[(c * 2 + 0) * SK_LDB + nn * 2]) = make_float2(vb[u].x, vb[u].z);
                    *reinterpret_cast<float2*>(&Bs[(c * 2 + 1) * SK_LDB + nn * 2]) = make_float2(vb[u].y, vb[u].w);
                }
            } else {
                const int n4 = idx & 15, k = idx >> 4;
                const int gn = n0 + 4 * n4;
                if (k < KQ * 4) {
                    const bool ok = k < K && gn + 3 < N;
                    const float4 v = ok ? vb[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                    float* d = &Bs[((k >> 2) * 2 + (k & 1)) * SK_LDB + 8 * n4 + ((k >> 1) & 1)];
                    d[0] = v.x; d[2] = v.y; d[4] = v.z; d[6] = v.w;
                }
            }
        }
    } else if (!B_KMAJOR) {
        for (int idx = tid; idx < SK_COLS * KQ * 4; idx += SK_T) {
            const int k = idx % (KQ * 4), nn = idx / (KQ * 4);
            int gn = n0 + nn; gn = gn < N ? gn : N - 1;
            const float v = k < K ? B[(long long)gn * ldb + k] : 0.f;
            Bs[((k >> 2) * 2 + (k & 1)) * SK_LDB + nn * 2 + ((k >> 1) & 1)] = v;
        }
    } else {
        for (int idx = tid; idx < SK_COLS * KQ * 4; idx += SK_T) {
            const int nn = idx & 63, k = idx >> 6;
            const int gn = n0 + nn;
            const float v = (k < K && gn < N) ? B[(long long)k * ldb + gn] : 0.f;
            Bs[((k >> 2) * 2 + (k & 1)) * SK_LDB + nn * 2 + ((k >> 1) & 1)] = v;
        }
    }
    __syncthreads();
    // ---- wavefront w: column half (w & 1), k quarter (w >> 1)
    {
        const int ct = wid & 1, kqr = wid >> 1;
        const int per = (KQ + 3) >> 2;
        const int kq0 = kqr * per, kq1 = (kq0 + per < KQ) ? kq0 + per : KQ;
        f32x16 acc = {0};
        const float* Ap = As + h * SK_LDA + li * 2;
        const float* Bp = Bs + h * SK_LDB + (32 * ct + li) * 2;
        for (int kq = kq0; kq < kq1; ++kq) {
            const float2 a = *reinterpret_cast<const float2*>(Ap + (size_t)kq * 2 * SK_LDA);
            const float2 b = *reinterpret_cast<const float2*>(Bp + (size_t)kq * 2 * SK_LDB);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Ps[((kqr * 2 + ct) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int o = tid + SK_T * u;                      // (column half, register index, lane) of one output element
        const int ln = o & 63, r = (o >> 6) & 15, ct = o >> 10;
        const float v0 = Ps[((0 * 2 + ct) * 16 + r) * 64 + ln], v1 = Ps[((1 * 2 + ct) * 16 + r) * 64 + ln];
        const float v2 = Ps[((2 * 2 + ct) * 16 + r) * 64 + ln], v3 = Ps[((3 * 2 + ct) * 16 + r) * 64 + ln];
        const int gn = n0 + 32 * ct + (ln & 31);
        const int gm = m0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
        float v = ((v0 + v1) + v2) + v3;
        if (bias && gn < N) v += bias[gn];
        if (relu) v = fmaxf(v, 0.f);
        if (gm < M && gn < N) C[(long long)gm * ldc + gn] = v;
    }
}
template <bool B_KMAJOR>
__global__ __launch_bounds__(SK_T) void gemm_skinny_f32_k(const float* __restrict__ A, const float* __restrict__ B,
                                                          float* __restrict__ C, int M_host, const int32_t* d_M, int N, int K,
                                                          long long lda, long long ldb, long long ldc,
                                                          const float* __restrict__ bias, int relu) {
    extern __shared__ __attribute__((aligned(16))) float sk_smem[];
    gemm_skinny_body<B_KMAJOR>(A, B, C, M_host, d_M, N, K, lda, ldb, ldc, bias, relu, blockIdx.x, blockIdx.y, sk_smem);
}

static inline bool skinny_ok(int M, int K) { return M <= 4096 && K >= 1 && K <= SK_KMAX; }
template <bool BK_>
static int launch_skinny(const float* A, const float* B, float* C, int M, const int32_t* d_M, int N, int K, long long lda,
                         long long ldb, long long ldc, const float* bias, int relu, hipStream_t s) {
    static bool lds_set = false;
    const int KQ = (K + 3) / 4;
    const size_t lds = ((size_t)KQ * 2 * (SK_LDA + SK_LDB) + 4 * 2 * 16 * 64) * sizeof(float);
    if (!lds_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_skinny_f32_k<BK_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((SK_KMAX / 4 * 2 * (SK_LDA + SK_LDB) + 4 * 2 * 16 * 64) * sizeof(float)));
        if (e != hipSuccess) return (int)e;
        lds_set = true;
    }
    dim3 grid(grapes_div_up(M, SK_ROWS), grapes_div_up(N, SK_COLS));
    hipLaunchKernelGGL((gemm_skinny_f32_k<BK_>), grid, dim3(SK_T), lds, s, A, B, C, M, d_M, N, K, lda, ldb, ldc, bias, relu);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

#ifndef GRAPES_ALIGNED16_DEFINED
#define GRAPES_ALIGNED16_DEFINED
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
#endif

template <bool AK, bool BK_>
static int launch_gemm(const float* A, const float* B, float* C, int M, int N, int K, long long lda, long long ldb,
                       long long ldc, const int32_t* d_M, const int32_t* d_K, int kchunk, int nslab, long long slab,
                       hipStream_t s, GemmEx ex = GemmEx{nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, 0}) {
    const int mt = grapes_div_up(M, GB_M), nt = grapes_div_up(N, GB_N);
    // With fewer than 8 row panels the padded XCD-aware map would leave most launched blocks idle AND put
    // all working ones on the same one or two XCDs (blocks are dealt round-robin over the 8 XCDs).
    const int grid_x = mt < 8 ? mt * nt : grapes_div_up(mt, 8) * 8 * nt;
    const bool vec = aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0) &&
                     (!ex.gate_a || aligned16(ex.gate_a));
    if (!vec && (ex.gate_a || ex.colsum || ex.gather)) return GRAPES_EALIGN;   // fused extras exist for the aligned path only
    if (ex.colsum && (N % GB_N == 0 || N % 4 != 0)) return GRAPES_EINVAL;   // needs a free padding column
    dim3 grid(grid_x, nslab);
    if (vec && ex.gather == 1)
        hipLaunchKernelGGL((gemm_mfma_f32_k<AK, BK_, true, 1>), grid, dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, d_M, d_K, kchunk, slab, nt, mt, ex);
    else if (vec && ex.gather == 2)
        hipLaunchKernelGGL((gemm_mfma_f32_k<AK, BK_, true, 2>), grid, dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, d_M, d_K, kchunk, slab, nt, mt, ex);
    else if (vec)
        hipLaunchKernelGGL((gemm_mfma_f32_k<AK, BK_, true>), grid, dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, d_M, d_K, kchunk, slab, nt, mt, ex);
    else
        hipLaunchKernelGGL((gemm_mfma_f32_k<AK, BK_, false>), grid, dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, d_M, d_K, kchunk, slab, nt, mt, ex);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// ---- f_out == 1 (the logit heads of gcn_gf / gcn_z): GEMV forms, no MFMA.
__global__ __launch_bounds__(256) void gemv_rows_k(const float* __restrict__ x, const float* __restrict__ w,
                                                   float* __restrict__ h, int n_host, const int32_t* d_n, int F) {
    const int n = eff_count(d_n, n_host);
    const int lane = lane_id();
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const bool vec = (F % 4 == 0) && ((((uintptr_t)x) & 15) == 0) && ((((uintptr_t)w) & 15) == 0);
    for (int r = wave_global; r < n; r += nwaves) {
        const float* xr = x + (long long)r * F;
        float acc = 0.f;
        if (vec) {
            for (int f = lane * 4; f < F; f += 256) {
                const float4 a = *reinterpret_cast<const float4*>(xr + f);
                const float4 b = *reinterpret_cast<const float4*>(w + f);
                acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc);
                acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
            }
        } else {
            for (int f = lane; f < F; f += 64) acc = fmaf(xr[f], w[f], acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) h[r] = acc;
    }
}

__global__ __launch_bounds__(256) void outer_rows_k(const float* __restrict__ dh, const float* __restrict__ w,
                                                    float* __restrict__ dx, int n_host, const int32_t* d_n, int F) {
    const int n = eff_count(d_n, n_host);
    const long long total = (long long)n * F;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / F);
        const int f = (int)(i - (long long)r * F);
        dx[i] = dh[r] * w[f];
    }
}

// ---- split-K slab reduction for dW: 64 outputs x 4 slab ranges per workgroup; each thread sums a
//      contiguous quarter of the slabs (several loads in flight), the quarters are combined in a
//      fixed order through LDS  => deterministic.
__global__ __launch_bounds__(256) void slab_reduce_k(const float* __restrict__ slabs, float* __restrict__ out,
                                                     long long count, int k_host, const int32_t* d_k, int kchunk,
                                                     int accumulate, const float* __restrict__ slabs2 = nullptr,
                                                     float* __restrict__ out2 = nullptr, long long count2 = 0,
                                                     const float* __restrict__ slabs3 = nullptr,
                                                     float* __restrict__ out3 = nullptr, long long count3 = 0) {
    __shared__ float part[4][64];
    const int K = eff_count(d_k, k_host);
    if (kchunk < 0) kchunk = auto_kchunk(K, -kchunk);
    const int ns = (K + kchunk - 1) / kchunk;
    const int g = threadIdx.x >> 6, c = threadIdx.x & 63;
    const int per = (ns + 3) >> 2;
    const int z0 = g * per, z1 = (z0 + per < ns) ? z0 + per : ns;
    const long long count_pad = (count + 63) & ~63LL;          // further sets of slabs (bias / head-weight gradients) ride along
    const long long count2_pad = (count2 + 63) & ~63LL;
    for (long long base = (long long)blockIdx.x * 64; base < count_pad + count2_pad + count3; base += (long long)gridDim.x * 64) {
        const int reg = base >= count_pad + count2_pad ? 2 : (base >= count_pad ? 1 : 0);
        const float* sl = reg == 2 ? slabs3 : (reg == 1 ? slabs2 : slabs);
        float* o = reg == 2 ? out3 : (reg == 1 ? out2 : out);
        const long long cnt = reg == 2 ? count3 : (reg == 1 ? count2 : count);
        const long long i = base - (reg == 2 ? count_pad + count2_pad : (reg == 1 ? count_pad : 0)) + c;
        float acc = 0.f;
        if (i < cnt) {
            int z = z0;
            for (; z + 16 <= z1; z += 16) {          // sixteen slabs in flight, added in slab order
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = sl[(long long)(z + u) * cnt + i];
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += v[u];
            }
            for (; z + 4 <= z1; z += 4) {
                const float a = sl[(long long)z * cnt + i], b = sl[(long long)(z + 1) * cnt + i];
                const float cc = sl[(long long)(z + 2) * cnt + i], d = sl[(long long)(z + 3) * cnt + i];
                acc += a; acc += b; acc += cc; acc += d;
            }
            for (; z < z1; ++z) acc += sl[(long long)z * cnt + i];
        }
        part[g][c] = acc;
        __syncthreads();
        if (g == 0 && i < cnt) {
            const float t = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
            o[i] = accumulate ? o[i] + t : t;
        }
        __syncthreads();
    }
}

// weighted column sum for the 1-wide head's dW (see spmm_kernels.hip)
int grapes_colsum_launch(const float* src, const float* gate, const float* wrow, float* dst, float* out, int n,
                         const int32_t* d_n, int F, int accumulate, float* workspace, hipStream_t s, unsigned* ticket = nullptr);
size_t grapes_colsum_workspace_bytes(int F);

// number of split-K slabs of a dW GEMM: slabs x output tiles = 512 workgroups (two per CU)
// ---- dW for FEW rows (the classifier's <= B + hops*K rows):  dW[M, N] = sum_r A[r][m] * x[r][n],  A = dOut masked by gate > 0.
// The tiled split-K launch streams its rows in 16-row steps — with ~1000 rows that is a chain of exposed round trips
// (~12 us for 0.13 GFLOP).  Here a workgroup of 512 threads owns a 32 x 64 tile of ONE 128-row slab: both operand
// blocks arrive in one round trip (<= 8 loads per thread, all in flight), the eight wavefronts split them as 2 column
// halves x 4 row quarters (fp32 MFMAs out of LDS), the quarters are summed through LDS in a fixed order, and the slab
// partials go to the existing slab reduction.  db = column sums of A rides along as an MFMA against a B of ones.
#define DWS_SMEM_FLOATS ((DWS_ROWS / 4) * 2 * (SK_LDA + SK_LDB) + 4 * SK_ROWS)
__device__ __forceinline__ void gemm_dw_small_body(const float* __restrict__ A, const float* __restrict__ gate,
                                                   const float* __restrict__ X, float* __restrict__ slabs,
                                                   float* __restrict__ cs_slabs, int n_host, const int32_t* d_n, int M,
                                                   int N, long long lda, long long ldb, int bx, int by, float* dws_smem) {
    // dws_smem: DWS_SMEM_FLOATS floats of LDS, 16-byte aligned (the caller's: static in the plain kernel, the dynamic block in the pair kernel)
    float* As = dws_smem;                                                               // [KQ][2][32][2 (+pad)]
    float* Bs = dws_smem + (DWS_ROWS / 4) * 2 * SK_LDA;                                 // [KQ][2][64][2 (+pad)]; then the partial tiles
    float* Ps = Bs;                                                                     // [4 quarters][2 halves][16][64] (same size)
    float (*Pb)[SK_ROWS] = reinterpret_cast<float (*)[SK_ROWS]>(Bs + (DWS_ROWS / 4) * 2 * SK_LDB);   // [4][32]
    constexpr int KQ = DWS_ROWS / 4;
    const int tiles_n = (N + SK_COLS - 1) / SK_COLS;
    const int tm = bx / tiles_n, tn = bx - tm * tiles_n;
    const int m0 = tm * SK_ROWS, n0 = tn * SK_COLS;
    const int k0 = by * DWS_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const bool avec = (lda % 4 == 0) && (M % 4 == 0) && ((((uintptr_t)A) & 15) == 0) && (!gate || ((((uintptr_t)gate) & 15) == 0));
    const bool bvec = (ldb % 4 == 0) && (N % 4 == 0) && ((((uintptr_t)X) & 15) == 0);
    // ---- loads first, rows clamped inside the CAPACITY (the live count arrives with them; rows past it are zeroed below)
    constexpr int NA = DWS_ROWS * (SK_ROWS / 4) / SK_T, NB = DWS_ROWS * (SK_COLS / 4) / SK_T;     // 2 and 4 chunks per thread
    float4 va[NA], vg[NA], vb[NB];
    if (avec) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int idx = tid + SK_T * u;
            const int m4 = idx & 7, k = idx >> 3;
            int r = k0 + k; r = r < n_host ? r : n_host - 1;
            const int gm = m0 + 4 * m4;
            const long long o = (long long)r * lda + (gm + 3 < M ? gm : 0);
            va[u] = *reinterpret_cast<const float4*>(A + o);
            vg[u] = gate ? *reinterpret_cast<const float4*>(gate + o) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
    }
    if (bvec) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int idx = tid + SK_T * u;
            const int n4 = idx & 15, k = idx >> 4;
            int r = k0 + k; r = r < n_host ? r : n_host - 1;
            const int gn = n0 + 4 * n4;
            vb[u] = *reinterpret_cast<const float4*>(X + (long long)r * ldb + (gn + 3 < N ? gn : 0));
        }
    }
    const int n = eff_count(d_n, n_host);
    if (k0 >= n) return;                                   // (uniform) a slab past the live rows: the reduction does not read it
    const int kc = n - k0 < DWS_ROWS ? n - k0 : DWS_ROWS;
    // ---- A block: element (m, k) -> As[((k>>2)*2 + (k&1)) * 32 + m][(k>>1)&1]
    if (avec) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int idx = tid + SK_T * u;
            const int m4 = idx & 7, k = idx >> 3;
            const bool ok = k < kc && m0 + 4 * m4 + 3 < M;
            float* d = &As[((k >> 2) * 2 + (k & 1)) * SK_LDA + 8 * m4 + ((k >> 1) & 1)];
            d[0] = (ok && vg[u].x > 0.f) ? va[u].x : 0.f; d[2] = (ok && vg[u].y > 0.f) ? va[u].y : 0.f;
            d[4] = (ok && vg[u].z > 0.f) ? va[u].z : 0.f; d[6] = (ok && vg[u].w > 0.f) ? va[u].w : 0.f;
        }
    } else {
        for (int idx = tid; idx < SK_ROWS * DWS_ROWS; idx += SK_T) {
            const int m = idx & 31, k = idx >> 5;
            const int gm = m0 + m;
            float v = 0.f;
            if (k < kc && gm < M) {
                const long long o = (long long)(k0 + k) * lda + gm;
                v = A[o];
                if (gate && !(gate[o] > 0.f)) v = 0.f;
            }
            As[((k >> 2) * 2 + (k & 1)) * SK_LDA + m * 2 + ((k >> 1) & 1)] = v;
        }
    }
    // ---- B block: element (k, n) -> Bs[((k>>2)*2 + (k&1)) * 64 + n][(k>>1)&1]
    if (bvec) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int idx = tid + SK_T * u;
            const int n4 = idx & 15, k = idx >> 4;
            const bool ok = k < kc && n0 + 4 * n4 + 3 < N;
            const float4 v = ok ? vb[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            float* d = &Bs[((k >> 2) * 2 + (k & 1)) * SK_LDB + 8 * n4 + ((k >> 1) & 1)];
            d[0] = v.x; d[2] = v.y; d[4] = v.z; d[6] = v.w;
        }
    } else {
        for (int idx = tid; idx < SK_COLS * DWS_ROWS; idx += SK_T) {
            const int nn = idx & 63, k = idx >> 6;
            const int gn = n0 + nn;
            Bs[((k >> 2) * 2 + (k & 1)) * SK_LDB + nn * 2 + ((k >> 1) & 1)] = (k < kc && gn < N) ? X[(long long)(k0 + k) * ldb + gn] : 0.f;
        }
    }
    __syncthreads();
    // ---- wavefront w: column half (w & 1), row quarter (w >> 1) of the slab
    {
        const int ct = wid & 1, kqr = wid >> 1;
        constexpr int per = KQ / 4;
        f32x16 acc = {0}, accb = {0};
        const float* Ap = As + h * SK_LDA + li * 2;
        const float* Bp = Bs + h * SK_LDB + (32 * ct + li) * 2;
        const bool colsum = cs_slabs != nullptr && tn == 0 && ct == 0;      // (uniform per wavefront)
        for (int kq = kqr * per; kq < (kqr + 1) * per; ++kq) {
            const float2 a = *reinterpret_cast<const float2*>(Ap + (size_t)kq * 2 * SK_LDA);
            const float2 b = *reinterpret_cast<const float2*>(Bp + (size_t)kq * 2 * SK_LDB);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            if (colsum) {
                accb = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, 1.0f, accb, 0, 0, 0);
                accb = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, 1.0f, accb, 0, 0, 0);
            }
        }
        __syncthreads();                                   // every wavefront has read its operands: Bs becomes Ps
#pragma unroll
        for (int r = 0; r < 16; ++r) Ps[((kqr * 2 + ct) * 16 + r) * 64 + lane] = acc[r];
        if (colsum && li == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Pb[kqr][(r & 3) + 8 * (r >> 2) + 4 * h] = accb[r];
        }
    }
    __syncthreads();
    float* C = slabs + (long long)by * M * N;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int o = tid + SK_T * u;                      // (column half, register index, lane) of one output element
        const int ln = o & 63, r = (o >> 6) & 15, ct = o >> 10;
        const float v0 = Ps[((0 * 2 + ct) * 16 + r) * 64 + ln], v1 = Ps[((1 * 2 + ct) * 16 + r) * 64 + ln];
        const float v2 = Ps[((2 * 2 + ct) * 16 + r) * 64 + ln], v3 = Ps[((3 * 2 + ct) * 16 + r) * 64 + ln];
        const int gn = n0 + 32 * ct + (ln & 31);
        const int gm = m0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
        if (gm < M && gn < N) C[(long long)gm * N + gn] = ((v0 + v1) + v2) + v3;
    }
    if (cs_slabs && tn == 0 && tid < SK_ROWS && m0 + tid < M)
        cs_slabs[(long long)by * M + m0 + tid] = ((Pb[0][tid] + Pb[1][tid]) + Pb[2][tid]) + Pb[3][tid];
}
// n rows fit the few-row form when the caller's workspace (sized by dw_nslab) holds one slab per 128 rows
__global__ __launch_bounds__(SK_T) void gemm_dw_small_k(const float* __restrict__ A, const float* __restrict__ gate,
                                                        const float* __restrict__ X, float* __restrict__ slabs,
                                                        float* __restrict__ cs_slabs, int n_host, const int32_t* d_n, int M,
                                                        int N, long long lda, long long ldb) {
    __shared__ __attribute__((aligned(16))) float dws_smem[DWS_SMEM_FLOATS];
    gemm_dw_small_body(A, gate, X, slabs, cs_slabs, n_host, d_n, M, N, lda, ldb, blockIdx.x, blockIdx.y, dws_smem);
}
// Two INDEPENDENT few-row GEMMs of a layer's backward pass side by side in one launch (both read dH = the backward aggregation's
// output; neither reads the other's result): the first na_x * na_y workgroups form the weight-gradient slabs (gemm_dw_small_body),
// the rest the input gradient dX = dH W (gemm_skinny_body<true>).  Each is two dependent round trips long whatever it computes,
// so one launch takes the time of the longer one.  Same bodies, same results as the two launches.
struct DwSmallArgs { const float* A; const float* gate; const float* X; float* slabs; float* cs_slabs; int M; int N; long long lda; long long ldb; };
struct SkinnyArgs { const float* A; const float* B; float* C; int N; int K; long long lda; long long ldb; long long ldc; };
__global__ __launch_bounds__(SK_T) void gemm_dw_small_dx_pair_k(DwSmallArgs a, SkinnyArgs b, int n_host, const int32_t* d_n,
                                                               int na_x, int na_y, int nb_x) {
    extern __shared__ __attribute__((aligned(16))) float sk_smem[];
    const int id = blockIdx.x, na = na_x * na_y;
    if (id < na) {
        gemm_dw_small_body(a.A, a.gate, a.X, a.slabs, a.cs_slabs, n_host, d_n, a.M, a.N, a.lda, a.ldb, id % na_x, id / na_x, sk_smem);
    } else {
        const int j = id - na;
        gemm_skinny_body<true>(b.A, b.B, b.C, n_host, d_n, b.N, b.K, b.lda, b.ldb, b.ldc, nullptr, 0, j % nb_x, j / nb_x, sk_smem);
    }
}
static inline int dw_nslab(int f_out, int f_in);
static inline bool dw_small_ok(int n, int f_out, int f_in) {
    return n <= 4096 && grapes_div_up(n, DWS_ROWS) <= dw_nslab(f_out, f_in);
}
// slabs: [nslab][f_out * f_in], cs_slabs: [nslab][f_out] or NULL; then slab_reduce_k(..., kchunk = DWS_ROWS, ...)
static int launch_dw_small(const float* a, const float* gate, const float* x, float* slabs, float* cs_slabs, int n,
                           const int32_t* d_n, int f_out, int f_in, long long ldx, hipStream_t s) {
    dim3 grid(grapes_div_up(f_out, SK_ROWS) * grapes_div_up(f_in, SK_COLS), grapes_div_up(n, DWS_ROWS));
    hipLaunchKernelGGL(gemm_dw_small_k, grid, dim3(SK_T), 0, s, a, gate, x, slabs, cs_slabs, n, d_n, f_out, f_in, (long long)f_out, ldx);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

static inline int dw_nslab(int f_out, int f_in) {
    const int tiles = grapes_div_up(f_out, GB_M) * grapes_div_up(f_in, GB_N);
    static int target = 0;
    if (!target) { const char* e = grapes_tune_env("GRAPES_DW_BLOCKS"); target = e ? atoi(e) : 512; if (target < 1) target = 512; }
    int ns = target / tiles;
    return ns < 1 ? 1 : ns;
}

// ---- few-row dW launches of SEVERAL layers, ONE slab reduction (the classifier's backward pass: three layers over the same
// ~1,000 rows — each slab sum used to be a launch of its own at the ~4.7 us floor of a dependent launch).
// grapes_linear_bwd_weight_slabs writes the per-128-row partial products only; grapes_slab_reduce_sets sums up to
// SR_MAX_SETS sets of slabs (all over the same row count) in slab order, 64 elements x 4 slab groups per workgroup as slab_reduce_k.
#define SR_MAX_SETS 8
struct SrSets { int nsets; const float* slabs[SR_MAX_SETS]; float* out[SR_MAX_SETS]; long long count[SR_MAX_SETS]; };
__global__ __launch_bounds__(256) void slab_reduce_sets_k(SrSets st, int k_host, const int32_t* d_k, int kchunk, int accumulate) {
    __shared__ float part[4][64];
    const int K = eff_count(d_k, k_host);
    const int ns = (K + kchunk - 1) / kchunk;
    const int g = threadIdx.x >> 6, c = threadIdx.x & 63;
    const int per = (ns + 3) >> 2;
    const int z0 = g * per, z1 = (z0 + per < ns) ? z0 + per : ns;
    long long total = 0;
    for (int q = 0; q < st.nsets; ++q) total += (st.count[q] + 63) & ~63LL;
    for (long long base = (long long)blockIdx.x * 64; base < total; base += (long long)gridDim.x * 64) {
        int q = 0; long long lo = 0;
        for (; q < st.nsets - 1; ++q) { const long long pad = (st.count[q] + 63) & ~63LL; if (base < lo + pad) break; lo += pad; }
        const float* sl = st.slabs[q]; float* o = st.out[q]; const long long cnt = st.count[q];
        const long long i = base - lo + c;
        float acc = 0.f;
        if (i < cnt) {
            int z = z0;
            for (; z + 8 <= z1; z += 8) {            // eight slabs in flight, added in slab order
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = sl[(long long)(z + u) * cnt + i];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
            for (; z < z1; ++z) acc += sl[(long long)z * cnt + i];
        }
        part[g][c] = acc;
        __syncthreads();
        if (g == 0 && i < cnt) {
            const float t = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
            o[i] = accumulate ? o[i] + t : t;
        }
        __syncthreads();
    }
}
extern "C" size_t grapes_linear_bwd_weight_slabs_bytes(int32_t n, int32_t f_in, int32_t f_out) {
    const size_t ns = (size_t)grapes_div_up(n > 0 ? n : 1, DWS_ROWS);
    return ns * ((size_t)f_in * f_out + (size_t)f_out) * sizeof(float) + 64;
}
// GRAPES_EINVAL when the shape is not one for the few-row kernel (the caller then takes grapes_linear_bwd_weight[_gated]).
// slabs: [ceil(n / 128)][f_out * f_in] at workspace, bias slabs [ceil(n / 128)][f_out] right behind them (want_bias).
extern "C" int grapes_linear_bwd_weight_slabs(const float* dout, const float* gate, const float* x, int32_t n, const int32_t* d_n,
                                              int32_t f_in, int32_t f_out, int32_t want_bias, void* workspace,
                                              grapes_stream_t stream) {
    if (n <= 0 || f_in <= 0 || f_out <= 1 || !dout || !x || !workspace) return GRAPES_EINVAL;
    if (!dw_small_ok(n, f_out, f_in)) return GRAPES_EINVAL;
    const size_t ns = (size_t)grapes_div_up(n, DWS_ROWS);
    float* w_dw = (float*)workspace;
    float* w_db = w_dw + ns * (size_t)f_in * f_out;
    return launch_dw_small(dout, gate, x, w_dw, want_bias ? w_db : nullptr, n, d_n, f_out, f_in, f_in, (hipStream_t)stream);
}
/* grapes_linear_bwd_weight_slabs (slabs of dW = (dout ⊙ [gate > 0])ᵀ x, optional bias slabs) AND grapes_linear_bwd_input
 * (dx = dh w, dh [n, f_out], w [f_out, f_in]) of the SAME layer in ONE launch — the two are independent of each other.
 * dout == dh for a layer whose gate is NULL (the caller passes the same pointer).  GRAPES_EINVAL when either shape is not one of
 * the few-row kernels (call the two entry points then). */
extern "C" int grapes_linear_bwd_weight_slabs_and_input(const float* dout, const float* gate, const float* x, const float* w,
                                                        float* dx, int32_t n, const int32_t* d_n, int32_t f_in, int32_t f_out,
                                                        int32_t want_bias, void* workspace, grapes_stream_t stream) {
    if (n <= 0 || f_in <= 0 || f_out <= 1 || !dout || !x || !w || !dx || !workspace) return GRAPES_EINVAL;
    if (!dw_small_ok(n, f_out, f_in) || !skinny_ok(n, f_out)) return GRAPES_EINVAL;
    const size_t ns = (size_t)grapes_div_up(n, DWS_ROWS);
    float* w_dw = (float*)workspace;
    float* w_db = w_dw + ns * (size_t)f_in * f_out;
    static bool lds_set = false;
    const size_t lds = ((size_t)SK_KMAX / 4 * 2 * (SK_LDA + SK_LDB) + 4 * 2 * 16 * 64) * sizeof(float);
    if (!lds_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_dw_small_dx_pair_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        lds_set = true;
    }
    const int KQ = (f_out + 3) / 4;
    size_t lds_use = ((size_t)KQ * 2 * (SK_LDA + SK_LDB) + 4 * 2 * 16 * 64) * sizeof(float);
    if (lds_use < DWS_SMEM_FLOATS * sizeof(float)) lds_use = DWS_SMEM_FLOATS * sizeof(float);
    const int na_x = grapes_div_up(f_out, SK_ROWS) * grapes_div_up(f_in, SK_COLS), na_y = (int)ns;
    const int nb_x = grapes_div_up(n, SK_ROWS), nb_y = grapes_div_up(f_in, SK_COLS);
    DwSmallArgs a{dout, gate, x, w_dw, want_bias ? w_db : nullptr, f_out, f_in, (long long)f_out, (long long)f_in};
    // dx[n, f_in] = dh[n, f_out] · w[f_out, f_in]: A = dh (k contiguous, K = f_out), B = w k-major
    SkinnyArgs b{dout, w, dx, f_in, f_out, (long long)f_out, (long long)f_in, (long long)f_in};
    hipLaunchKernelGGL(gemm_dw_small_dx_pair_k, dim3(na_x * na_y + nb_x * nb_y), dim3(SK_T), lds_use, (hipStream_t)stream, a, b, n, d_n,
                       na_x, na_y, nb_x);
    GRAPES_LAUNCH_CHECK();
    return 0;
}
extern "C" int grapes_slab_reduce_sets(int32_t nsets, const float* const* slabs, float* const* outs, const int64_t* counts,
                                       int32_t n, const int32_t* d_n, int32_t accumulate, grapes_stream_t stream) {
    if (nsets < 1 || nsets > SR_MAX_SETS || !slabs || !outs || !counts || n <= 0) return GRAPES_EINVAL;
    SrSets st; st.nsets = nsets;
    long long total = 0;
    for (int q = 0; q < SR_MAX_SETS; ++q) {
        const int r = q < nsets ? q : 0;
        if (!slabs[r] || !outs[r] || counts[r] <= 0) return GRAPES_EINVAL;
        st.slabs[q] = slabs[r]; st.out[q] = outs[r]; st.count[q] = counts[r];
        if (q < nsets) total += (counts[r] + 63) & ~63LL;
    }
    int grid = (int)(total / 64); if (grid > 8192) grid = 8192; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(slab_reduce_sets_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, st, n, d_n, DWS_ROWS, accumulate);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

extern "C" int grapes_linear_fwd(const float* x, const float* w, float* h, int32_t n, const int32_t* d_n,
                                 int32_t f_in, int32_t f_out, grapes_stream_t stream) {
    if (n < 0 || f_in <= 0 || f_out <= 0) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!x || !w || !h) return GRAPES_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (f_out == 1) {
        int grid = grapes_div_up(n, 4); if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(gemv_rows_k, dim3(grid), dim3(256), 0, s, x, w, h, n, d_n, f_in);
        GRAPES_LAUNCH_CHECK();
        return 0;
    }
    // A = x [n,f_in] (k contiguous), B = w [f_out,f_in] (k contiguous)
    if (skinny_ok(n, f_in)) return launch_skinny<false>(x, w, h, n, d_n, f_out, f_in, f_in, f_in, f_out, nullptr, 0, s);
    return launch_gemm<false, false>(x, w, h, n, f_out, f_in, f_in, f_in, f_out, d_n, nullptr, f_in + GB_K, 1, 0, s);
}

__global__ __launch_bounds__(256) void scale_rows_few_k(float* __restrict__ h, const float* __restrict__ sc, int n_host,
                                                        const int32_t* d_n, int f) {
    const int n = eff_count(d_n, n_host);
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const float s = sc[r];
    for (int c = threadIdx.x & 63; c < f; c += 64) h[(long long)r * f + c] *= s;
}

// H = diag(row_scale) X Wᵀ: the full-batch inference layer's transform with the scaling pass of grapes_scale_rows in the epilogue
// (the product is formed in the accumulators, rounded to fp32 as grapes_linear_fwd stores it, then multiplied: the same bits as
// the two launches, without writing and re-reading n x f_out floats)
extern "C" int grapes_linear_fwd_row_scaled(const float* x, const float* w, const float* row_scale, float* h, int32_t n,
                                            const int32_t* d_n, int32_t f_in, int32_t f_out, grapes_stream_t stream) {
    if (n < 0 || f_in <= 0 || f_out <= 1) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!x || !w || !h || !row_scale) return GRAPES_EINVAL;
    if (skinny_ok(n, f_in)) {     // few rows: grapes_linear_fwd's kernel for them (another summation order), then the scaling on its own
        const int rc = launch_skinny<false>(x, w, h, n, d_n, f_out, f_in, f_in, f_in, f_out, nullptr, 0, (hipStream_t)stream);
        if (rc) return rc;
        hipLaunchKernelGGL(scale_rows_few_k, dim3(grapes_div_up(n, 4)), dim3(256), 0, (hipStream_t)stream, h, row_scale, n, d_n, f_out);
        GRAPES_LAUNCH_CHECK();
        return 0;
    }
    GemmEx ex{};
    ex.out_scale = row_scale;
    return launch_gemm<false, false>(x, w, h, n, f_out, f_in, f_in, f_in, f_out, d_n, nullptr, f_in + GB_K, 1, 0, (hipStream_t)stream, ex);
}

extern "C" size_t grapes_linear_bwd_weight_workspace_bytes(int32_t n_cap, int32_t f_in, int32_t f_out) {
    if (n_cap <= 0) n_cap = 1;
    if (f_out == 1) return grapes_colsum_workspace_bytes(f_in);
    (void)n_cap;
    return (size_t)dw_nslab(f_out, f_in) * f_in * f_out * sizeof(float);
}

extern "C" int grapes_linear_bwd_weight(const float* dh, const float* x, float* dw, int32_t n, const int32_t* d_n,
                                        int32_t f_in, int32_t f_out, int32_t accumulate, void* workspace,
                                        grapes_stream_t stream) {
    if (n < 0 || f_in <= 0 || f_out <= 0 || !dw) return GRAPES_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        if (!accumulate) { hipError_t e = grapes_zero_async(dw, (size_t)f_in * f_out * sizeof(float), s); if (e) return (int)e; }
        return 0;
    }
    if (!dh || !x || !workspace) return GRAPES_EINVAL;
    if (f_out == 1)   // dW[f] = sum_r dh[r] x[r,f]
        return grapes_colsum_launch(x, nullptr, dh, nullptr, dw, n, d_n, f_in, accumulate, (float*)workspace, s);
    // dW[f_out,f_in] = sum_r dh[r,f_out] x[r,f_in] :  A = dh (k-major, M=f_out), B = x (k-major, N=f_in), K = n rows
    const int nslab = dw_nslab(f_out, f_in);
    const long long slab = (long long)f_in * f_out;
    if (dw_small_ok(n, f_out, f_in)) {
        int rc = launch_dw_small(dh, nullptr, x, (float*)workspace, nullptr, n, d_n, f_out, f_in, f_in, s);
        if (rc) return rc;
        int grid = grapes_div_up(slab, 64); if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(slab_reduce_k, dim3(grid), dim3(256), 0, s, (const float*)workspace, dw, slab, n, d_n, DWS_ROWS, accumulate);
        GRAPES_LAUNCH_CHECK();
        return 0;
    }
    int rc = launch_gemm<true, true>(dh, x, (float*)workspace, f_out, f_in, n, f_out, f_in, f_in, nullptr, d_n,
                                     -nslab, nslab, slab, s);
    if (rc) return rc;
    int grid = grapes_div_up(slab, 64); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(slab_reduce_k, dim3(grid), dim3(256), 0, s, (const float*)workspace, dw, slab, n, d_n, -nslab,
                       accumulate);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// ---- aggregate-first layers (F_in < F_out, input needs no gradient):  out = act((Â X) Wᵀ + b)
//      forward : one GEMM with the bias + ReLU epilogue;
//      backward: dW = (dOut ⊙ [out > 0])ᵀ (Â X), db = column sums of the gated dOut — ONE split-K GEMM whose
//                A-operand loads apply the ReLU mask and whose first padding column of B reads as ones.
extern "C" int grapes_linear_bias_act_fwd(const float* x, const float* w, const float* bias, int32_t relu, float* out,
                                          int32_t n, const int32_t* d_n, int32_t f_in, int32_t f_out,
                                          grapes_stream_t stream) {
    if (n < 0 || f_in <= 0 || f_out <= 1) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!x || !w || !out) return GRAPES_EINVAL;
    static int split = -1;      // GRAPES_GEMM_SPLIT=0: the fp32-MFMA kernel instead of the bf16x3 one (same accuracy class)
    if (split < 0) { const char* e = grapes_tune_env("GRAPES_GEMM_SPLIT"); split = e ? atoi(e) : 1; }
    if (split && wsplit_ok(x, w, out, f_in, f_out) && n >= 2048)
        return launch_wsplit(x, w, bias, relu ? 1 : 0, out, n, d_n, f_in, f_out, (hipStream_t)stream);
    if (wstat_ok(x, w, out, f_in, f_out) && n >= 2048)
        return launch_wstat(x, w, bias, relu ? 1 : 0, out, n, d_n, f_in, f_out, (hipStream_t)stream);
    if (skinny_ok(n, f_in))
        return launch_skinny<false>(x, w, out, n, d_n, f_out, f_in, f_in, f_in, f_out, bias, relu ? 1 : 0, (hipStream_t)stream);
    GemmEx ex{bias, relu ? 1 : 0, nullptr, nullptr, 0, nullptr, nullptr, 0};
    return launch_gemm<false, false>(x, w, out, n, f_out, f_in, f_in, f_in, f_out, d_n, nullptr, f_in + GB_K, 1, 0,
                                     (hipStream_t)stream, ex);
}

// The same layer followed by a 1-wide head without its own ReLU input:  head_out[i] = sum_n out[i][n] * head_w[n]
// (GCNConv(F_out -> 1)'s XW step, modules/gcn.py:32 on the last layer).  With the bf16x3 kernel the head is summed from the
// output tiles while they are still in registers (the n x F_out activations are not read back); otherwise two launches.
extern "C" int grapes_linear_bias_act_head_fwd(const float* x, const float* w, const float* bias, int32_t relu, float* out,
                                               const float* head_w, float* head_out, int32_t n, const int32_t* d_n,
                                               int32_t f_in, int32_t f_out, grapes_stream_t stream) {
    if (n < 0 || f_in <= 0 || f_out <= 1) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!x || !w || !out || !head_w || !head_out) return GRAPES_EINVAL;
    static int split = -1;
    if (split < 0) { const char* e = grapes_tune_env("GRAPES_GEMM_SPLIT"); split = e ? atoi(e) : 1; }
    if (split && wsplit_ok(x, w, out, f_in, f_out) && n >= 2048)
        return launch_wsplit(x, w, bias, relu ? 1 : 0, out, n, d_n, f_in, f_out, (hipStream_t)stream, head_w, head_out);
    int rc = grapes_linear_bias_act_fwd(x, w, bias, relu, out, n, d_n, f_in, f_out, stream);
    if (rc) return rc;
    return grapes_linear_fwd(out, head_w, head_out, n, d_n, f_out, 1, stream);
}

#ifdef GRAPES_DIAG
// diagnosis entry point (profiles/microbench.py): the forward GEMM with parts switched off
extern "C" int grapes_debug_gemm_fwd(const float* x, const float* w, float* out, int32_t n, int32_t f_in, int32_t f_out,
                                     int32_t dbg, grapes_stream_t stream) {
    if (dbg & 32) {   // the one-shot kernel for few rows
        if (!skinny_ok(n, f_in)) return GRAPES_EINVAL;
        return launch_skinny<false>(x, w, out, n, nullptr, f_out, f_in, f_in, f_in, f_out, nullptr, 0, (hipStream_t)stream);
    }
    if (dbg & 64) {   // the split-bf16 kernel regardless of n (dbg bits 256 / 512 / 1024: no stores / MFMAs / staging)
        if (!wsplit_ok(x, w, out, f_in, f_out)) return GRAPES_EINVAL;
        return launch_wsplit(x, w, nullptr, dbg & (256 | 512 | 1024 | 2048), out, n, nullptr, f_in, f_out, (hipStream_t)stream);
    }
    if (dbg & 16) {   // the W-stationary kernel regardless of n
        if (!wstat_ok(x, w, out, f_in, f_out)) return GRAPES_EINVAL;
        return launch_wstat(x, w, nullptr, 0, out, n, nullptr, f_in, f_out, (hipStream_t)stream);
    }
    GemmEx ex{nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, dbg & 7};
    return launch_gemm<false, false>(x, w, out, n, f_out, f_in, f_in, f_in, f_out, nullptr, nullptr, f_in + GB_K, 1, 0,
                                     (hipStream_t)stream, ex);
}
#endif  // GRAPES_DIAG

// ---- dW-stationary backward GEMM of the aggregate-first layers with a 1-wide head (rank-1 upstream gradient):
//        dW1[m][n] = sum_r (rs[r] * cv[m] * [gate[r][m] > 0]) * x[r][n],   db1[m] = sum_r (same factor),
//        dW2[m]    = sum_r rs[r] * gate[r][m]
//      over the rows of up to four hops that share the weights.  One workgroup (8 wavefronts) owns the WHOLE
//      f_out x (f_in + 1) output in registers (wave w: rows 32w..32w+31 x 128 columns = four 32x32 accumulators) and
//      streams a contiguous share of ALL hops' rows (balanced on the device: total live rows / workgroups, a share may
//      span two hops) in chunks of 32 rows through a double-buffered k-major LDS image.  Compared with 128x128 tiles
//      and 16-row steps: each gate / x row is read once (not once per column / row tile), half as many k steps, and the
//      hops no longer cost a launch and a slab set each.  Column f_in of the x image reads as ones => db1 falls out of
//      the same MFMAs.  The per-workgroup partial outputs ("slabs") are summed by slab_reduce_k in index order.
#define DW_KC 32
#define DW_MAXM 256
#define DW_NT 128
#define DW_LDA (DW_MAXM + 4)
#define DW_LDB (DW_NT + 4)
struct DwSegs {
    int nseg;
    const float* gate[4]; const float* x[4]; const float* rs[4]; const int32_t* d_n[4]; int n_cap[4];
    int ldx[4];             // row stride of x[q] in floats (gemm_dw_split_k only; the fp32 kernel needs dense rows)
    const uint32_t* bits[4];   // gemm_dw_split_k<true>: the forward pass's gate words (wsplit_store) instead of gate[q]
};
__global__ __launch_bounds__(512, 1) void gemm_dw_rank1_k(DwSegs sg, const float* __restrict__ cv, int M, int Nin,
                                                          float* __restrict__ slabs, float* __restrict__ cs_db,
                                                          float* __restrict__ cs_head) {
    extern __shared__ __attribute__((aligned(16))) float dw_smem[];
    float* As = dw_smem;                                     // [2][DW_KC][DW_LDA]
    float* Bs = dw_smem + 2 * DW_KC * DW_LDA;                // [2][DW_KC][DW_LDB]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    // ---- this workgroup's share of the concatenated row space
    int nrows[4], off[5];
    off[0] = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        nrows[q] = q < sg.nseg ? eff_count(sg.d_n[q], sg.n_cap[q]) : 0;
        off[q + 1] = off[q] + nrows[q];
    }
    const int total = off[4];
    const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int g0 = blockIdx.x * per, g1 = (g0 + per < total) ? g0 + per : total;
    // chunk iterator over (segment, first row) pairs; a chunk never straddles two segments
    int seg = 0, k0 = 0, khi = 0;
    auto seek = [&](int from_seg) {
        seg = from_seg;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q >= seg && seg == q) {
                const int lo = g0 > off[q] ? g0 - off[q] : 0;
                const int hi = (g1 < off[q + 1] ? g1 : off[q + 1]) - off[q];
                if (q < sg.nseg && lo < hi) { k0 = lo; khi = hi; return; }
                seg = q + 1;
            }
        }
        seg = 4;
    };
    seek(0);
    // ---- loaders: gate chunk 32 x M (float4 f = tid + 512 j: row f / (M/4), col4 f % (M/4)); x chunk 32 x Nin
    const int M4 = M >> 2, N4 = Nin >> 2;
    float4 rg[4], rx[2]; float rsv[4];
    const float4 cv4 = *reinterpret_cast<const float4*>(cv + 4 * (tid % M4 < M4 ? tid % M4 : 0));
    float4 cs2 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto seg_ptr = [&](int q, const float*& g, const float*& x, const float*& r) {
        g = q == 0 ? sg.gate[0] : (q == 1 ? sg.gate[1] : (q == 2 ? sg.gate[2] : sg.gate[3]));
        x = q == 0 ? sg.x[0] : (q == 1 ? sg.x[1] : (q == 2 ? sg.x[2] : sg.x[3]));
        r = q == 0 ? sg.rs[0] : (q == 1 ? sg.rs[1] : (q == 2 ? sg.rs[2] : sg.rs[3]));
    };
    int c_rows = 0;                                          // live rows of the chunk held in (rg, rx, rsv)
    auto load_chunk = [&](int q, int kk, int hi) {           // unconditional, clamped loads
        const float *g, *x, *r;
        seg_ptr(q < 4 ? q : 0, g, x, r);
        const int last = hi > 0 ? hi - 1 : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + 512 * j;
            const int row = f / M4, c4 = f - row * M4;
            const int k = kk + row < hi ? kk + row : last;
            rg[j] = *reinterpret_cast<const float4*>(g + (long long)k * M + 4 * c4);
            rsv[j] = r[k];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int f = tid + 512 * j;
            const int row = f / N4, c4 = f - row * N4;
            const int k = kk + (row < DW_KC ? row : 0) < hi ? kk + (row < DW_KC ? row : 0) : last;
            rx[j] = *reinterpret_cast<const float4*>(x + (long long)k * Nin + 4 * (c4 < N4 ? c4 : 0));
        }
        c_rows = hi - kk < DW_KC ? hi - kk : DW_KC;
    };
    auto stage_chunk = [&](int buf) {
        float* Ab = As + buf * DW_KC * DW_LDA;
        float* Bb = Bs + buf * DW_KC * DW_LDB;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + 512 * j;
            const int row = f / M4, c4 = f - row * M4;
            if (row < DW_KC) {
                const bool ok = row < c_rows;
                const float rs = ok ? rsv[j] : 0.f;
                const float4 g = rg[j];
                const float4 cvv = *reinterpret_cast<const float4*>(cv + 4 * c4);
                float4 t;
                t.x = g.x > 0.f ? rs * cvv.x : 0.f; t.y = g.y > 0.f ? rs * cvv.y : 0.f;
                t.z = g.z > 0.f ? rs * cvv.z : 0.f; t.w = g.w > 0.f ? rs * cvv.w : 0.f;
                if (!ok) t = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(&Ab[row * DW_LDA + 4 * c4]) = t;
                if (ok) {   // dW of the head: column sums of rs * gate (this thread always owns the same 4 columns)
                    cs2.x = fmaf(rs, g.x, cs2.x); cs2.y = fmaf(rs, g.y, cs2.y);
                    cs2.z = fmaf(rs, g.z, cs2.z); cs2.w = fmaf(rs, g.w, cs2.w);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int f = tid + 512 * j;
            const int row = f / N4, c4 = f - row * N4;
            if (row < DW_KC) {
                float4 v = rx[j];
                if (row >= c_rows) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(&Bb[row * DW_LDB + 4 * c4]) = v;
            }
        }
        // columns Nin .. DW_NT-1 of the x image: a column of ones (bias gradient), then zeros
        for (int f = tid; f < DW_KC * (DW_NT - Nin); f += 512) {
            const int row = f / (DW_NT - Nin), c = Nin + f - row * (DW_NT - Nin);
            Bb[row * DW_LDB + c] = (c == Nin && row < c_rows) ? 1.f : 0.f;
        }
    };
    (void)cv4;
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    const int m_w = 32 * wid;                                // this wavefront's output rows
    if (seg < 4) {
        load_chunk(seg, k0, khi);
        stage_chunk(0);
    }
    __syncthreads();
    int buf = 0;
    while (seg < 4) {
        // next chunk: advance the iterator, issue its loads, then the MFMAs of the current one
        int nseg_ = seg, nk0 = k0 + DW_KC, nhi = khi;
        if (nk0 >= khi) { const int cs = seg, ck = k0, ch = khi; seek(seg + 1); nseg_ = seg; nk0 = k0; nhi = khi; seg = cs; k0 = ck; khi = ch; }
        const bool more = nseg_ < 4;
        const int cur_rows = c_rows;
        load_chunk(more ? nseg_ : seg, more ? nk0 : k0, more ? nhi : khi);
        const int nxt_rows = c_rows;
        (void)cur_rows;
        if (m_w < M) {
            const float* Ab = As + buf * DW_KC * DW_LDA + m_w + li;
            const float* Bb = Bs + buf * DW_KC * DW_LDB + li;
#pragma unroll 4
            for (int kk = 0; kk < DW_KC; kk += 2) {
                const float a = Ab[(kk + h) * DW_LDA];
                const float b0 = Bb[(kk + h) * DW_LDB], b1 = Bb[(kk + h) * DW_LDB + 32];
                const float b2 = Bb[(kk + h) * DW_LDB + 64], b3 = Bb[(kk + h) * DW_LDB + 96];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b3, acc[3], 0, 0, 0);
            }
        }
        c_rows = nxt_rows;
        if (more) stage_chunk(buf ^ 1);
        __syncthreads();
        buf ^= 1;
        seg = nseg_; k0 = nk0; khi = nhi;
    }
    // ---- this workgroup's slab: dW1 (columns < Nin), db1 (column Nin), dW2 (cs2)
    float* C = slabs + (long long)blockIdx.x * M * Nin;
    if (m_w < M) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_w + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int n = 32 * j + li;
                if (n < Nin) C[(long long)m * Nin + n] = acc[j][r];
                else if (n == Nin && cs_db) cs_db[(long long)blockIdx.x * M + m] = acc[j][r];
            }
        }
    }
    if (cs_head) {   // combine the row groups that share a column quad (fixed order) through LDS
        __syncthreads();
        float* red = dw_smem;                                // [512 / M4][M]
        if (tid / M4 < 512 / M4 && tid < (512 / M4) * M4) *reinterpret_cast<float4*>(&red[(tid / M4) * M + 4 * (tid % M4)]) = cs2;
        __syncthreads();
        if (tid < M) {
            float t = 0.f;
            for (int g8 = 0; g8 < 512 / M4; ++g8) t += red[g8 * M + tid];
            cs_head[(long long)blockIdx.x * M + tid] = t;
        }
    }
}

// ---- The same dW-stationary backward on the bf16 matrix pipe (see gemm_wsplit_f32_k for the exact 3-way split).  The
// rank-1 structure makes it cheaper still:   dW1[m][n] = cv[m] * sum_r [gate[r][m] > 0] * (rs[r] * x[r][n])
// — the A operand is the 0/1 ReLU mask, EXACT in one bf16 plane, so a product needs three MFMAs (mask x the three
// planes of rs*x), not six, and cv[m] is applied once in the epilogue.  Column f_in of the B image holds rs[r] itself
// (=> db1), the head's dW2[m] = sum_r rs[r] * gate[r][m] stays on the vector ALU while the gate tile is staged.
// 32-row chunks; the contraction index is the ROW index, so the staging transposes: a thread loads 8 (A) or 4 (B)
// consecutive rows of one column quad and writes, per column, the 8 / 4 k-values as one 16- / 8-byte bf16 vector into
// the images  A[k/8][m][8],  B[plane][k/8][n][8]  (fragment = one ds_read_b128).  Wavefronts 0-3 stage the mask,
// 4-7 the rs*x planes and the rs column.  MFMA time per chunk drops from 8192 to 1536 cycles per SIMD: the kernel moves
// from the fp32 matrix pipe to its HBM traffic (gate + x rows, read once).
// BITS: the mask comes as the forward pass's gate words (32 bytes per row instead of the 4 M-byte activation row), and the
// head's gradient is derived from the accumulators: with S[m][n] = sum_r mask rs x (the accumulator before cv) and
// T[m] = sum_r mask rs (its column Nin),
//     dW2[m] = sum_r rs[r] relu(x[r] . W1[m] + b1[m]) = sum_n S[m][n] W1[m][n] + b1[m] T[m]
// — no activation is read at all (slab_reduce_rank1_k, from the summed slabs).
#define DS_NT 128
// diagnostic stamps of gemm_dw_split_k<true> (profiles/dw_split_stamps.py): thread T0's arrival at a point, steady-state iteration IT
#ifdef GRAPES_STAMPS
#define DWS_STAMP(slot, T0) do { if (grapes_stamp_ptr && threadIdx.x == (T0) && blockIdx.x < 64) grapes_stamp_ptr[blockIdx.x * 16 + (slot)] = wall_clock64(); } while (0)
#define DWS_STAMP_IT(it_, slot) do { if ((it_) == 6) { DWS_STAMP(slot, 0); DWS_STAMP((slot) + 6, 256); } } while (0)
#else
#define DWS_STAMP(slot, T0) do { } while (0)
#define DWS_STAMP_IT(it_, slot) do { } while (0)
#endif
#define DS_MINROWS 128
__host__ __device__ __forceinline__ int dw_share(int total, int nwg, bool min_rows) {
    const int per = (total + nwg - 1) / nwg;
    return (min_rows && per < DS_MINROWS) ? DS_MINROWS : per;
}
// A SECOND, independent problem in the same launch (alt.nwg0 < gridDim.x): workgroups [0, nwg0) own segments [0, first_seg)
// and the kernel's own Nin / slabs / cs_db; workgroups [nwg0, gridDim.x) own segments [first_seg, nseg) and alt's.  The log-Z
// net's backward GEMM (10k rows, its own weights) is a latency-bound launch of its own otherwise — 16 us + 6 us of slab sums
// behind the sampler net's 32 + 9; side by side on disjoint CUs the pair costs little more than the longer one.
struct DwAlt { int nwg0; int first_seg; int Nin; const float* cv; float* slabs; float* cs_db; };
// NT = width of the x image in columns (f_in + the rs column <= NT): 128, or 160 for f_in up to 156 (ogbn-arxiv / papers100M:
// 128 features + 3-4 indicators = 132; the log-Z net's 128) — a fifth 32-column accumulator tile per wavefront, and more
// staging tasks than the 256 threads of wavefronts 4-7: wavefront 4 takes a second task per chunk (its own copy of the loop).
// ---- staging of gemm_dw_split_k<true> as free functions: the images a chunk is READ from (by the MFMAs) and WRITTEN to (the next chunk)
// arrive as __restrict__ parameters, so hipcc may order the LDS stores of the staging between the LDS reads of the MFMA phase
// (through one extern __shared__ array with run-time image indices every store "may alias" every read and keeps its source position).
struct DwsRegs { uint32_t gw[8]; float4 xb[4]; float rsb[4]; float rso; int rows; float4 xb2[4]; float rsb2[4]; float rso2; };
struct DwsCtx { int M, Nin, kb, ac4, bshift, hb, bc4, ok_, hb2, bc42, ok2_, lane; bool a_role, b_role, o_role, b2_role, o2_role; };
__device__ __forceinline__ int dws_sw(int n) { return n ^ ((n >> 3) & 3); }
// one slice: output column G of the thread's column quad (G < 4), or the rs column (G = 4).  Values are formed by EVERY lane and only
// the ADDRESS of the LDS store depends on the lane's role (lanes without one store to `dummy`): inside a divergent `if (role)` the
// waits for the set's loads sat on a path the wavefront may skip — hipcc then assumed at the loop head that they were still in flight
// and put vmcnt(0..4) in front of the next chunk's address arithmetic — and a guarded store splits the basic block the MFMAs are in.
template <int ROLE, int NT, int G>
__device__ __forceinline__ void dws_stage_slice(const DwsRegs& R, const DwsCtx& c, uint4* __restrict__ WA, char* __restrict__ WB,
                                                uint4* __restrict__ dummy) {
    constexpr int B_PL = 4 * NT;
    char* dm = reinterpret_cast<char*>(&dummy[c.lane]);
    if constexpr (G < 4) {
        constexpr int u = G;
        if constexpr (ROLE == 0) {
            // rows in pairs: the nibbles of rows 2 p and 2 p + 1 at bits 0-3 and 16-19 of one word w; bit u of both, times bf16(1.0) =
            // 0x3F80, is (w & (0x00010001 << u)) * (0x3F80 >> u) — no carry between the halves (a select per row and column before)
            uint32_t e[4];
#pragma unroll
            for (int pq = 0; pq < 4; ++pq) {
                const uint32_t n0 = (8 * c.kb + 2 * pq < R.rows) ? (R.gw[2 * pq] >> c.bshift) & 15u : 0u;
                const uint32_t n1 = (8 * c.kb + 2 * pq + 1 < R.rows) ? (R.gw[2 * pq + 1] >> c.bshift) & 15u : 0u;
                const uint32_t w = n0 | (n1 << 16);
                e[pq] = (w & (0x00010001u << u)) * (0x3F80u >> u);
            }
            const uint4 q = make_uint4(e[0], e[1], e[2], e[3]);
            *(c.a_role ? &WA[c.kb * c.M + dws_sw(4 * c.ac4 + u)] : &dummy[c.lane]) = q;
        } else {
            {
                uint2 pl[3];
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool live = 4 * c.hb + j < R.rows;
                    const float xv = u == 0 ? R.xb[j].x : (u == 1 ? R.xb[j].y : (u == 2 ? R.xb[j].z : R.xb[j].w));
                    v[j] = live ? R.rsb[j] * xv : 0.f;
                }
                split3_pair(v[0], v[1], pl[0].x, pl[1].x, pl[2].x);
                split3_pair(v[2], v[3], pl[0].y, pl[1].y, pl[2].y);
                char* base = WB + ((size_t)((c.hb >> 1) * NT) * 16 + (c.hb & 1) * 8);
#pragma unroll
                for (int pp = 0; pp < 3; ++pp)
                    *reinterpret_cast<uint2*>(c.b_role ? base + (size_t)pp * B_PL * 16 + dws_sw(4 * c.bc4 + u) * 16 : dm) = pl[pp];
            }
            if constexpr (ROLE == 2) {       // the second task: the same work on task bt + 256
                uint2 pl[3];
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool live = 4 * c.hb2 + j < R.rows;
                    const float xv = u == 0 ? R.xb2[j].x : (u == 1 ? R.xb2[j].y : (u == 2 ? R.xb2[j].z : R.xb2[j].w));
                    v[j] = live ? R.rsb2[j] * xv : 0.f;
                }
                split3_pair(v[0], v[1], pl[0].x, pl[1].x, pl[2].x);
                split3_pair(v[2], v[3], pl[0].y, pl[1].y, pl[2].y);
                char* base = WB + ((size_t)((c.hb2 >> 1) * NT) * 16 + (c.hb2 & 1) * 8);
#pragma unroll
                for (int pp = 0; pp < 3; ++pp)
                    *reinterpret_cast<uint2*>(c.b2_role ? base + (size_t)pp * B_PL * 16 + dws_sw(4 * c.bc42 + u) * 16 : dm) = pl[pp];
            }
        }
    } else if constexpr (G == 4 && ROLE != 0) {
        {
            __bf16 x0, x1, x2; split3(c.ok_ < R.rows ? R.rso : 0.f, x0, x1, x2);
            char* base = WB + ((size_t)((c.ok_ >> 3) * NT + dws_sw(c.Nin)) * 16 + (c.ok_ & 7) * 2);
            *reinterpret_cast<__bf16*>(c.o_role ? base : dm) = x0;
            *reinterpret_cast<__bf16*>(c.o_role ? base + (size_t)B_PL * 16 : dm) = x1;
            *reinterpret_cast<__bf16*>(c.o_role ? base + (size_t)2 * B_PL * 16 : dm) = x2;
        }
        if constexpr (ROLE == 2) {
            __bf16 x0, x1, x2; split3(c.ok2_ < R.rows ? R.rso2 : 0.f, x0, x1, x2);
            char* base = WB + ((size_t)((c.ok2_ >> 3) * NT + dws_sw(c.Nin)) * 16 + (c.ok2_ & 7) * 2);
            *reinterpret_cast<__bf16*>(c.o2_role ? base : dm) = x0;
            *reinterpret_cast<__bf16*>(c.o2_role ? base + (size_t)B_PL * 16 : dm) = x1;
            *reinterpret_cast<__bf16*>(c.o2_role ? base + (size_t)2 * B_PL * 16 : dm) = x2;
        }
    }
}
template <int ROLE, int NT>
__device__ __forceinline__ void dws_stage_set(const DwsRegs& R, const DwsCtx& c, uint4* __restrict__ WA, char* __restrict__ WB,
                                              uint4* __restrict__ dummy) {
    dws_stage_slice<ROLE, NT, 0>(R, c, WA, WB, dummy);
    dws_stage_slice<ROLE, NT, 1>(R, c, WA, WB, dummy);
    dws_stage_slice<ROLE, NT, 2>(R, c, WA, WB, dummy);
    dws_stage_slice<ROLE, NT, 3>(R, c, WA, WB, dummy);
    dws_stage_slice<ROLE, NT, 4>(R, c, WA, WB, dummy);
}
template <bool BITS, int NT = 128>
__global__ __launch_bounds__(512, 1) void gemm_dw_split_k(DwSegs sg, const float* __restrict__ cv_, int M, int Nin_,
                                                          float* __restrict__ slabs_, float* __restrict__ cs_db_,
                                                          float* __restrict__ cs_head, DwAlt alt) {
    extern __shared__ uint4 ds_smem[];
    const bool dual = alt.nwg0 < (int)gridDim.x, p1 = dual && (int)blockIdx.x >= alt.nwg0;
    const int bid = p1 ? (int)blockIdx.x - alt.nwg0 : (int)blockIdx.x;
    const int nwg = p1 ? (int)gridDim.x - alt.nwg0 : (dual ? alt.nwg0 : (int)gridDim.x);
    const int seg_lo = p1 ? alt.first_seg : 0, seg_hi = (dual && !p1) ? alt.first_seg : sg.nseg;
    const int Nin = p1 ? alt.Nin : Nin_;
    const float* __restrict__ cv = p1 ? alt.cv : cv_;
    float* __restrict__ slabs = p1 ? alt.slabs : slabs_;
    float* __restrict__ cs_db = p1 ? alt.cs_db : cs_db_;
    // Column / row slots of both images are XOR-swizzled in their low two bits by bits 3-4 of the slot index: a staging thread
    // writes four consecutive slots of a column quad and its neighbours the next quads, i.e. a wavefront's write instruction
    // hit the same four banks from every second lane (64-byte stride); the MFMA reads still see 32 consecutive slots per
    // half-wave, permuted inside aligned groups of four.
    auto sw = [](int n) { return n ^ ((n >> 3) & 3); };
    constexpr int A_IMG = 4 * DW_MAXM, B_PL = 4 * NT, BUF = A_IMG + 3 * B_PL;          // uint4 units: 16 KB + 3 x 8 KB (NT = 128)
    constexpr int TN = NT / 32;                                                         // 32-column accumulator tiles
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    DWS_STAMP(0, 0);
    for (int i = tid; i < 2 * BUF; i += 512) ds_smem[i] = make_uint4(0u, 0u, 0u, 0u);   // pad columns stay zero
    // ---- this workgroup's share of the concatenated row space (as gemm_dw_rank1_k)
    int nrows[4], off[5];
    off[0] = 0;
    {   // (the other problem's segments: empty)
        const int32_t* const dq[4] = {sg.d_n[0], sg.d_n[1], sg.d_n[2], sg.d_n[3]};
        const int cq[4] = {sg.n_cap[0], sg.n_cap[1], sg.n_cap[2], sg.n_cap[3]};
        const bool wq[4] = {0 >= seg_lo && 0 < seg_hi, 1 >= seg_lo && 1 < seg_hi, 2 >= seg_lo && 2 < seg_hi, 3 >= seg_lo && 3 < seg_hi};
        eff_counts<4>(dq, cq, wq, slabs, nrows);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) off[q + 1] = off[q] + nrows[q];
    const int total = off[4];
    // BITS: a workgroup takes at least DS_MINROWS rows, and the workgroups beyond the last share neither run nor write a slab
    // (slab_reduce_rank1_k derives the same live count): few rows (the log-Z net's 10k at hop 0) then cost 75 slabs, not 256
    const int per = dw_share(total, nwg, BITS);
    if (BITS && (long long)bid * per >= total) return;
    const int g0 = bid * per, g1 = (g0 + per < total) ? g0 + per : total;
    int seg = 0, k0 = 0, khi = 0;
    auto seek = [&](int from_seg) {
        seg = from_seg;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q >= seg && seg == q) {
                const int lo = g0 > off[q] ? g0 - off[q] : 0;
                const int hi = (g1 < off[q + 1] ? g1 : off[q + 1]) - off[q];
                if (q < sg.nseg && lo < hi) { k0 = lo; khi = hi; return; }
                seg = q + 1;
            }
        }
        seg = 4;
    };
    seek(0);
    auto seg_bits = [&](int q) { return q == 0 ? sg.bits[0] : (q == 1 ? sg.bits[1] : (q == 2 ? sg.bits[2] : sg.bits[3])); };
    auto seg_ptr = [&](int q, const float*& g, const float*& x, const float*& r, int& ld) {
        g = q == 0 ? sg.gate[0] : (q == 1 ? sg.gate[1] : (q == 2 ? sg.gate[2] : sg.gate[3]));
        x = q == 0 ? sg.x[0] : (q == 1 ? sg.x[1] : (q == 2 ? sg.x[2] : sg.x[3]));
        r = q == 0 ? sg.rs[0] : (q == 1 ? sg.rs[1] : (q == 2 ? sg.rs[2] : sg.rs[3]));
        ld = q == 0 ? sg.ldx[0] : (q == 1 ? sg.ldx[1] : (q == 2 ? sg.ldx[2] : sg.ldx[3]));
    };
    // ---- staging roles
    const int M4 = M >> 2, N4 = Nin >> 2;
    const bool a_role = tid < 4 * M4 && wid < 4;             // mask task: 8-row block kb, column quad ac4
    const int kb = a_role ? tid / M4 : 0, ac4 = a_role ? tid - kb * M4 : 0;
    const int bt = tid - 256;
    const bool b_role = bt >= 0 && bt < 8 * N4;              // rs*x half-task: 4-row block hb, column quad bc4
    const int hb = b_role ? bt / N4 : 0, bc4 = b_role ? bt - hb * N4 : 0;
    const bool o_role = bt >= 8 * N4 && bt < 8 * N4 + DW_KC; // the rs column: one row each
    const int ok_ = o_role ? bt - 8 * N4 : 0;
    // NT = 160: the tasks beyond the 256th (8 N4 + DW_KC <= 344) are second tasks of wavefront 4's threads
    const int bt2 = bt + 256;
    const bool b2_role = NT > 128 && wid == 4 && bt2 < 8 * N4;
    const int hb2 = b2_role ? bt2 / N4 : 0, bc42 = b2_role ? bt2 - hb2 * N4 : 0;
    const bool o2_role = NT > 128 && wid == 4 && bt2 >= 8 * N4 && bt2 < 8 * N4 + DW_KC;
    const int ok2_ = o2_role ? bt2 - 8 * N4 : 0;
    float4 ga[BITS ? 1 : 8]; float rsa[BITS ? 1 : 8]; uint32_t gw[BITS ? 8 : 1]; float4 xb[4]; float rsb[4]; float rso = 0.f;
    float4 cs2 = make_float4(0.f, 0.f, 0.f, 0.f);
    int c_rows = 0;
    const int MW = M >> 5;
    // the four mask bits of column quad ac4: word ac4 / 8 of the row, bits 16 h + 4 q + {0..3} with 2 q + h = ac4 % 8 (wsplit_store)
    const int bword = ac4 >> 3, bshift = 16 * (ac4 & 1) + 4 * ((ac4 & 7) >> 1);
    auto load_chunk = [&](int q, int kk, int hi) {           // unconditional, clamped loads (wave-uniform roles)
        const float *g, *x, *r; int ld;
        seg_ptr(q < 4 ? q : 0, g, x, r, ld);
        const int last = hi > 0 ? hi - 1 : 0;
        if (wid < 4) {
            if (BITS) {
                const uint32_t* bp = seg_bits(q < 4 ? q : 0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = kk + 8 * kb + j < hi ? kk + 8 * kb + j : last;
                    gw[j] = bp[(long long)k * MW + bword];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = kk + 8 * kb + j < hi ? kk + 8 * kb + j : last;
                    ga[j] = *reinterpret_cast<const float4*>(g + (long long)k * M + 4 * ac4);
                    rsa[j] = r[k];
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kk + 4 * hb + j < hi ? kk + 4 * hb + j : last;
                xb[j] = *reinterpret_cast<const float4*>(x + (long long)k * ld + 4 * bc4);
                rsb[j] = r[k];
            }
            const int k = kk + ok_ < hi ? kk + ok_ : last;
            rso = r[k];
        }
        c_rows = hi - kk < DW_KC ? hi - kk : DW_KC;
    };
    auto stage_chunk = [&](int buf) {
        uint4* Ab = ds_smem + (size_t)buf * BUF;
        char* Bb = reinterpret_cast<char*>(ds_smem + (size_t)buf * BUF + A_IMG);
        if (a_role) {
            unsigned e[4][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool live = 8 * kb + j < c_rows;
                if (BITS) {
                    const uint32_t nib = live ? (gw[j] >> bshift) : 0u;
                    e[0][j] = (nib & 1u) ? 0x3F80u : 0u; e[1][j] = (nib & 2u) ? 0x3F80u : 0u;
                    e[2][j] = (nib & 4u) ? 0x3F80u : 0u; e[3][j] = (nib & 8u) ? 0x3F80u : 0u;
                } else {
                    const float4 g = ga[j];
                    e[0][j] = (live && g.x > 0.f) ? 0x3F80u : 0u; e[1][j] = (live && g.y > 0.f) ? 0x3F80u : 0u;
                    e[2][j] = (live && g.z > 0.f) ? 0x3F80u : 0u; e[3][j] = (live && g.w > 0.f) ? 0x3F80u : 0u;
                    if (live) {      // dW of the head: column sums of rs * gate (a thread always owns the same 4 columns)
                        const float rs = rsa[j];
                        cs2.x = fmaf(rs, g.x, cs2.x); cs2.y = fmaf(rs, g.y, cs2.y);
                        cs2.z = fmaf(rs, g.z, cs2.z); cs2.w = fmaf(rs, g.w, cs2.w);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                Ab[kb * M + sw(4 * ac4 + u)] = make_uint4(e[u][0] | (e[u][1] << 16), e[u][2] | (e[u][3] << 16),
                                                      e[u][4] | (e[u][5] << 16), e[u][6] | (e[u][7] << 16));
        }
        if (b_role) {
            bf16x4 pl[4][3];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool live = 4 * hb + j < c_rows;
                const float rs = live ? rsb[j] : 0.f;
                const float v[4] = {rs * xb[j].x, rs * xb[j].y, rs * xb[j].z, rs * xb[j].w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    __bf16 x0, x1, x2; split3(live ? v[u] : 0.f, x0, x1, x2);
                    pl[u][0][j] = x0; pl[u][1][j] = x1; pl[u][2][j] = x2;
                }
            }
            char* base = Bb + ((size_t)((hb >> 1) * NT) * 16 + (hb & 1) * 8);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int pp = 0; pp < 3; ++pp)
                    *reinterpret_cast<bf16x4*>(base + (size_t)pp * B_PL * 16 + sw(4 * bc4 + u) * 16) = pl[u][pp];
        }
        if (o_role) {
            __bf16 x0, x1, x2; split3(ok_ < c_rows ? rso : 0.f, x0, x1, x2);
            char* base = Bb + ((size_t)((ok_ >> 3) * NT + sw(Nin)) * 16 + (ok_ & 7) * 2);
            *reinterpret_cast<__bf16*>(base) = x0;
            *reinterpret_cast<__bf16*>(base + (size_t)B_PL * 16) = x1;
            *reinterpret_cast<__bf16*>(base + (size_t)2 * B_PL * 16) = x2;
        }
    };
    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[t] = f32x16{0};
    const int m_w = 32 * wid;                                // this wavefront's output rows
    auto mfma_chunk = [&](int buf) {
        if (m_w < M) {
            const uint4* Ab = ds_smem + (size_t)buf * BUF + h * M + sw(m_w + li);
            const uint4* Bb = ds_smem + (size_t)buf * BUF + A_IMG + h * NT + sw(li);     // (32 t + li: the swizzle bits of 32 t are zero)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, Ab[(size_t)ks * 2 * M]);
#pragma unroll
                for (int pp = 2; pp >= 0; --pp) {            // small planes first
#pragma unroll
                    for (int t = 0; t < TN; ++t) {
                        const bf16x8 b = __builtin_bit_cast(bf16x8, Bb[(size_t)pp * B_PL + ks * 2 * NT + 32 * t]);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
                    }
                }
            }
        }
    };
    __syncthreads();                                         // the zero fill is complete
    if constexpr (!BITS) {
    if (seg < 4) {
        load_chunk(seg, k0, khi);
        stage_chunk(0);
    }
    __syncthreads();
    int buf = 0;
    while (seg < 4) {
        // next chunk: advance the iterator, issue its loads, then the MFMAs of the current one
        int nseg_ = seg, nk0 = k0 + DW_KC, nhi = khi;
        if (nk0 >= khi) { const int cs = seg, ck = k0, ch = khi; seek(seg + 1); nseg_ = seg; nk0 = k0; nhi = khi; seg = cs; k0 = ck; khi = ch; }
        const bool more = nseg_ < 4;
        load_chunk(more ? nseg_ : seg, more ? nk0 : k0, more ? nhi : khi);
        mfma_chunk(buf);
        if (more) stage_chunk(buf ^ 1);
        __syncthreads();
        buf ^= 1;
        seg = nseg_; k0 = nk0; khi = nhi;
    }
    } else {
    // Gate words leave room for TWO register sets: the loads of chunk i + 2 are issued at the top of iteration i and staged in
    // iteration i + 1 — a whole iteration (MFMAs, staging, barrier) to arrive.  With one set every 32-row chunk paid a dependent
    // global round trip between its loads and its staging: 3.4 us per chunk against 0.64 us of MFMAs (measured: the kernel took
    // the same 38 us whether the mask came from 32-byte words or from 1 KB activation rows).
    uint4* const dummy = ds_smem + 2 * BUF;       // 64 slots behind the images: where the lanes without a staging role store
    struct It { int seg, k0, khi; };
    auto seek_it = [&](int from) {
        It r{4, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (r.seg == 4 && q >= from && q < sg.nseg) {
                const int lo = g0 > off[q] ? g0 - off[q] : 0;
                const int hi = (g1 < off[q + 1] ? g1 : off[q + 1]) - off[q];
                if (lo < hi) { r.seg = q; r.k0 = lo; r.khi = hi; }
            }
        }
        // (uniform by construction; said explicitly: with the iterator in VGPRs hipcc indexed the pointer arrays of `sg` with a
        // vector load and put a vmcnt(0) in front of every chunk's loads — no prefetch at all)
        r.seg = __builtin_amdgcn_readfirstlane(r.seg); r.k0 = __builtin_amdgcn_readfirstlane(r.k0); r.khi = __builtin_amdgcn_readfirstlane(r.khi);
        return r;
    };
    auto next_it = [&](It c) {
        if (c.seg >= 4) return c;
        It nx = c;
        nx.k0 += DW_KC;
        if (nx.k0 >= nx.khi) nx = seek_it(c.seg + 1);
        nx.seg = __builtin_amdgcn_readfirstlane(nx.seg); nx.k0 = __builtin_amdgcn_readfirstlane(nx.k0); nx.khi = __builtin_amdgcn_readfirstlane(nx.khi);
        return nx;
    };
    const It first = seek_it(0);
    using Regs = DwsRegs;
    const DwsCtx ctx{M, Nin, kb, ac4, bshift, hb, bc4, ok_, hb2, bc42, ok2_, lane, a_role, b_role, o_role, b2_role, o2_role};
    // role: 0 = mask (wavefronts 0-3), 1 = rs * x (wavefronts 4-7), 2 = rs * x with a second task (NT = 160: wavefront 4)
    auto load_set = [&](auto role, Regs& R, It c) {          // unconditional, clamped: a finished iterator re-reads the first chunk
        constexpr int ROLE = decltype(role)::value;
        using role_a_t = std::integral_constant<bool, ROLE == 0>;
        role_a_t role_a;
        const It v = c.seg < 4 ? c : first;
        const float *g, *x, *r; int ld;
        seg_ptr(v.seg, g, x, r, ld);
        const int last = v.khi - 1;
        // 32-bit byte offsets from the segment's (uniform) base pointers — the launcher checks n_cap * stride * 4 < 2^32: the address
        // of a load is then one 32-bit multiply-add instead of a 64-bit one plus two 64-bit shift-adds (quarter-rate vector ops, nine
        // loads per thread and chunk)
        const char* xb_ = reinterpret_cast<const char*>(x);
        const char* rb_ = reinterpret_cast<const char*>(r);
        const uint32_t ldb = (uint32_t)ld * 4u;
        if (decltype(role_a)::value) {
            const char* bp = reinterpret_cast<const char*>(seg_bits(v.seg));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = v.k0 + 8 * kb + j < v.khi ? v.k0 + 8 * kb + j : last;
                R.gw[j] = *reinterpret_cast<const uint32_t*>(bp + ((uint32_t)k * (uint32_t)MW + (uint32_t)bword) * 4u);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = v.k0 + 4 * hb + j < v.khi ? v.k0 + 4 * hb + j : last;
                R.xb[j] = *reinterpret_cast<const float4*>(xb_ + ((uint32_t)k * ldb + 16u * (uint32_t)bc4));
                R.rsb[j] = *reinterpret_cast<const float*>(rb_ + (uint32_t)k * 4u);
            }
            const int k = v.k0 + ok_ < v.khi ? v.k0 + ok_ : last;
            R.rso = *reinterpret_cast<const float*>(rb_ + (uint32_t)k * 4u);
            if (ROLE == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k2 = v.k0 + 4 * hb2 + j < v.khi ? v.k0 + 4 * hb2 + j : last;
                    R.xb2[j] = *reinterpret_cast<const float4*>(xb_ + ((uint32_t)k2 * ldb + 16u * (uint32_t)bc42));
                    R.rsb2[j] = *reinterpret_cast<const float*>(rb_ + (uint32_t)k2 * 4u);
                }
                const int k3 = v.k0 + ok2_ < v.khi ? v.k0 + ok2_ : last;
                R.rso2 = *reinterpret_cast<const float*>(rb_ + (uint32_t)k3 * 4u);
            }
        }
        R.rows = c.seg < 4 ? (v.khi - v.k0 < DW_KC ? v.khi - v.k0 : DW_KC) : 0;
    };
    // Staging in SLICES (one output column u of the thread's column quad per slice, the rs column as slice 4) so that the slices
    // can sit between the MFMA groups of the chunk being multiplied (mfma_stage below); stage_set = all slices back to back.
    auto stage_set = [&](auto role, const Regs& R, int buf) {
        constexpr int ROLE = decltype(role)::value;
        dws_stage_set<ROLE, NT>(R, ctx, ds_smem + (size_t)buf * BUF, reinterpret_cast<char*>(ds_smem + (size_t)buf * BUF + A_IMG), dummy);
    };
    // The MFMAs of the chunk in image `bm`, then the staging of register set R into the OTHER image.  The fragments of MFMA group
    // g + 2 (one k-step x one plane x TN tiles) are requested before group g's MFMAs are issued: hipcc keeps two ds_read_b128 in
    // flight and waits for each pair in front of its MFMAs — twelve exposed LDS latencies per chunk and wavefront.
    auto mfma_stage = [&](auto role, int bm, const Regs& R, bool do_mfma, bool do_stage, int it_ = 0) {
        if (NT > 128) {         // the 160-column form (five tiles per group, a second staging task): no registers for fragments ahead
            if (do_mfma) mfma_chunk(bm);
        } else if (m_w < M && do_mfma) {
            const uint4* Ab = ds_smem + (size_t)bm * BUF + h * M + sw(m_w + li);
            const uint4* Bb = ds_smem + (size_t)bm * BUF + A_IMG + h * NT + sw(li);
            uint4 fb[6][TN];
            auto rd = [&](auto g_) {
                constexpr int g = decltype(g_)::value;
                if constexpr (g < 6) {
                    constexpr int ks = g / 3, pp = 2 - g % 3;      // small planes first, as mfma_chunk
#pragma unroll
                    for (int t = 0; t < TN; ++t) fb[g][t] = Bb[(size_t)pp * B_PL + ks * 2 * NT + 32 * t];
                }
            };
            constexpr int DEPTH = 2;
            const uint4 qa0 = Ab[0];
            rd(std::integral_constant<int, 0>{});
            const uint4 qa1 = Ab[(size_t)2 * M];
            if constexpr (DEPTH > 1) rd(std::integral_constant<int, 1>{});
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, qa0), a1 = __builtin_bit_cast(bf16x8, qa1);
            auto group = [&](auto g_) {
                constexpr int g = decltype(g_)::value;
                rd(std::integral_constant<int, g + DEPTH>{});
#pragma unroll
                for (int t = 0; t < TN; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g < 3 ? a0 : a1, __builtin_bit_cast(bf16x8, fb[g][t]), acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            group(std::integral_constant<int, 0>{});
            group(std::integral_constant<int, 1>{});
            group(std::integral_constant<int, 2>{});
            group(std::integral_constant<int, 3>{});
            group(std::integral_constant<int, 4>{});
            group(std::integral_constant<int, 5>{});
        }
#ifdef GRAPES_STAMPS
        if (it_ == 6) { DWS_STAMP(5, 0); DWS_STAMP(11, 256); asm volatile("" :: "v"(acc[0][0]), "v"(acc[TN - 1][15])); DWS_STAMP(6, 0); DWS_STAMP(12, 256); }
#endif
        if (do_stage) stage_set(role, R, bm ^ 1);
    };
    // One loop, no prologue: half n loads chunk n into set n & 1, multiplies chunk n - 2 (image n & 1) and stages chunk n - 1 (from the
    // other set) into the other image — the first two halves skip what does not exist yet (the counter is opaque so that they are not
    // peeled back into a prologue).  What bounds the loop (round 5, knock-out builds, profiles/r05_ab.txt): NOT the loads (no loads
    // after the first two chunks: -0.8 us of 32), not the LDS (conflict-free, ~40 % of its cycles); without MFMAs the launch is 8.5 us
    // shorter, without the staging 8.1 — the two ADD UP, in every arrangement tried: staging slices between the MFMA groups of the same
    // wavefront, one MFMA : nine vector instructions through sched_group_barrier in one basic block, the halves of the workgroup in
    // anti-phase (0-3 stage while 4-7 multiply, two barriers per chunk: +1.2 us), three and four register sets (+1.4 us of prologue).
    // On one SIMD a 32x32x16 MFMA's 32 cycles and the vector instructions of either wavefront do not overlap here; what helped is
    // FEWER vector instructions: 32-bit offsets from uniform bases instead of 64-bit address arithmetic (nine loads per thread and
    // chunk), pairs of values per v_cvt_pk_bf16_f32, the mask expanded two rows per multiply — 34.6 -> 32.0 us.
    auto run = [&](auto role_a) {
        if (first.seg >= 4) return;
        Regs RA, RB;
        It cn = first, cp = first;             // chunk n, chunk n - 1
        int n = 0;
        for (;;) {
            asm volatile("" : "+s"(n));
            DWS_STAMP_IT(n, 3);
            load_set(role_a, RA, cn);
            DWS_STAMP_IT(n, 4);
            mfma_stage(role_a, 0, RB, n >= 2, n >= 1, n);
            DWS_STAMP_IT(n, 7);
            __syncthreads();
            DWS_STAMP_IT(n, 8);
            if (cp.seg >= 4) break;            // (n = 0: cp = the first chunk, live)
            cp = cn; cn = next_it(cn); ++n;
            load_set(role_a, RB, cn);
            mfma_stage(role_a, 1, RA, n >= 2, true);
            __syncthreads();
            if (cp.seg >= 4) break;
            cp = cn; cn = next_it(cn); ++n;
        }
        DWS_STAMP(2, 0);
    };
    if (wid < 4) run(std::integral_constant<int, 0>{});
    else if (NT > 128 && wid == 4) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 1>{});
    }
    DWS_STAMP(15, 0);
    // ---- this workgroup's slab: dW1 (columns < Nin), db1 (column Nin), both scaled by cv[m]; dW2 (cs2).  BITS: S and T
    // unscaled — slab_reduce_rank1_k applies cv and derives dW2 from the summed S, T
    // BITS: the slabs lie [output row m][workgroup][column] — slab_reduce_rank1_k's workgroup m then streams ONE contiguous block
    // (nwg x Nin floats) instead of nwg pieces 100 KB apart; the writer's rows are 400-byte pieces either way.
    float* C = BITS ? slabs + (long long)bid * Nin : slabs + (long long)bid * M * Nin;
    const long long crow = BITS ? (long long)nwg * Nin : (long long)Nin;
    if (m_w < M) {
        float cvm[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) cvm[r] = BITS ? 1.f : cv[m_w + (r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_w + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int n = 32 * j + li;
                const float v = BITS ? acc[j][r] : acc[j][r] * cvm[r];
                if (n < Nin) C[(long long)m * crow + n] = v;
                else if (n == Nin && cs_db) cs_db[(long long)bid * M + m] = v;
            }
        }
    }
#ifdef GRAPES_STAMPS
    if (BITS) GRAPES_STAMP(1);
#endif
    if (BITS) {
    } else if (cs_head) {   // combine the four row blocks that share a column quad (fixed order) through LDS
        __syncthreads();
        float* red = reinterpret_cast<float*>(ds_smem);      // [4][M]
        if (a_role) *reinterpret_cast<float4*>(&red[kb * M + 4 * ac4]) = cs2;
        __syncthreads();
        if (tid < M) cs_head[(long long)bid * M + tid] = ((red[tid] + red[M + tid]) + red[2 * M + tid]) + red[3 * M + tid];
    }
}
// Slab sums of gemm_dw_split_k<true>: S[m][n] = sum of the live slabs in index order (eight groups of slabs per workgroup,
// combined in order), then   dw[m][n] (+)= cv[m] S,   db[m] (+)= cv[m] T,   dwh[m] (+)= sum_n S[m][n] W1[m][n] + b1[m] T[m]
// (the row sum in a fixed tree order).  One workgroup per output row m, 8 x 128 threads; a thread's (at most 32) slab
// elements are all in flight together.
#define SR1_G 8
struct Sr1Prob {
    const float* slabs; const float* tslabs; const float* cv; const float* W1; const float* b1;
    float* dw; float* db; float* dwh; int nwg, Nin, seg_lo, seg_hi;
    int dw_cols;        // dw is [M, dw_cols], dw_cols <= Nin: the parameter's own layout when Nin is its 4-padded width
};
// NC = columns handled (f_in + 1 <= NC): 128 with SR1_G = 8 slab groups, or 192 with 5 (the NT = 160 form of gemm_dw_split_k)
template <int NC, int G>
__global__ __launch_bounds__(NC * G) void slab_reduce_rank1_k(DwSegs sg, Sr1Prob pa, Sr1Prob pb, int M, int accumulate) {
    __shared__ float red[NC / 64];
    const Sr1Prob P = blockIdx.y ? pb : pa;                // (gridDim.y == 2: the second problem of a dual launch)
    const int Nin = P.Nin;
    const int m = blockIdx.x, g = threadIdx.x / NC, n = threadIdx.x - g * NC;
    const bool is_s = n < Nin, is_t = n == Nin;
    // everything that does not depend on the slabs first: the counts, the weight row, the previous gradient
    // (the four counts through unconditional loads — an absent one reads a word of cv and is replaced: with a branch per count hipcc
    // waited for each before requesting the next, four dependent scalar trips at the head of the launch)
    int cnt[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool has = q < sg.nseg && sg.d_n[q];
        const int32_t raw = *(has ? sg.d_n[q] : reinterpret_cast<const int32_t*>(P.cv));
        cnt[q] = has ? raw : 0x7fffffff;
    }
    const float c = P.cv[m];
    const float wrow = (g == 0 && P.dwh) ? (is_s ? P.W1[(long long)m * Nin + n] : (is_t ? P.b1[m] : 0.f)) : 0.f;
    float* o = is_s ? (n < P.dw_cols ? P.dw + (long long)m * P.dw_cols + n : nullptr) : ((is_t && P.db) ? P.db + m : nullptr);
    const float prev = (g == 0 && accumulate && o) ? *o : 0.f;
    int total = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        total += (q >= P.seg_lo && q < P.seg_hi) ? (cnt[q] < sg.n_cap[q] ? (cnt[q] > 0 ? cnt[q] : 0) : sg.n_cap[q]) : 0;
    const int per = dw_share(total, P.nwg, true);
    const int live = total > 0 ? (total + per - 1) / per : 0;
    // Row m of every live slab is ONE contiguous block [live][Nin] (gemm_dw_split_k<true> writes [m][workgroup][n]): a thread owns a
    // column QUAD and every (NC * G / (NC / 4))-th slab — 16-byte loads, eight of them in flight per thread (<= 256 slabs over 32
    // groups); the T column (tslabs: [workgroup][M]) is the quad behind the last one.  One order of additions: a group's slabs in index
    // order, then the groups in order.
    constexpr int Q = NC / 4, NG = NC * G / Q;                          // 32 quads x 32 groups (NC = 128, G = 8); 48 x 20 (192, 5)
    __shared__ float part4[NG][NC];
    const int q4 = threadIdx.x % Q, g4 = threadIdx.x / Q;
    const bool q_s = 4 * q4 < Nin, q_t = 4 * q4 == Nin;                 // (Nin is a multiple of 4)
    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q_s || q_t) {
        const float* base = P.slabs + (long long)m * P.nwg * Nin + 4 * q4;
        for (int zb = g4; zb < live; zb += 8 * NG) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int z = zb + u * NG < live ? zb + u * NG : live - 1;                  // unconditional, clamped
                v[u] = q_t ? make_float4(P.tslabs[(long long)z * M + m], 0.f, 0.f, 0.f)
                           : *reinterpret_cast<const float4*>(base + (long long)z * Nin);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (zb + u * NG < live) { a4.x += v[u].x; a4.y += v[u].y; a4.z += v[u].z; a4.w += v[u].w; }
        }
    }
    if (4 * q4 + 3 < NC) *reinterpret_cast<float4*>(&part4[g4][4 * q4]) = a4;
    __syncthreads();
    if (g == 0) {
        float sum = part4[0][n];
#pragma unroll 8
        for (int q = 1; q < NG; ++q) sum += part4[q][n];
        if (o) *o = prev + c * sum;
        if (P.dwh) {
            const float p = wave_sum(sum * wrow);
            if ((n & 63) == 0) red[n >> 6] = p;
        }
    }
    __syncthreads();
    if (P.dwh && threadIdx.x == 0) {
        float t = red[0] + red[1];
        if (NC > 128) t += red[2];
        P.dwh[m] = accumulate ? P.dwh[m] + t : t;
    }
}

#define DS_NT_WIDE 160
// (gate words: f_in + 1 <= 160 columns and at most 256 + 64 staging tasks; gate activations: the 128-column form only)
static inline bool dw_split_ok(int f_in, int f_out, bool bits = false) {
    const bool shape = f_out % 32 == 0 && f_out >= 32 && f_out <= DW_MAXM && f_in % 4 == 0 && f_in >= 4;
    if (shape && 8 * (f_in / 4) + DW_KC <= 256) return true;
    return bits && shape && f_in + 1 <= DS_NT_WIDE && 8 * (f_in / 4) + DW_KC <= 256 + 64;
}
static inline bool dw_split_narrow(int f_in) { return 8 * (f_in / 4) + DW_KC <= 256; }

static inline bool dw_rank1_ok(int f_in, int f_out) {
    return f_out % 32 == 0 && f_out >= 32 && f_out <= DW_MAXM && f_in % 4 == 0 && f_in >= 4 && f_in + 1 <= DW_NT &&
           (512 % (f_out / 4)) == 0;
}

static inline bool fused_dw_ok(const float* dout, const float* gate, const float* x, int f_in, int f_out) {
    return aligned16(dout) && aligned16(x) && (!gate || aligned16(gate)) && f_in % 4 == 0 && f_out % 4 == 0 &&
           f_in % GB_N != 0;
}

#define DW_BLOCKS 256
#define DW_BLOCKS_SECOND 32          // workgroups of the second problem of a dual launch (the rest go to the first)
struct DwSecond { const float* col_vec; const float* w1; const float* b1; float* dw; float* dbias; float* dw_head; int f_in; };
static int launch_dw_rank1(int nseg, const float* const* gate, const float* const* x, const float* const* row_scale,
                           const int32_t* const* d_n, const int32_t* n_cap, const float* col_vec, float* dw, float* dbias,
                           float* dw_head, int f_in, int f_out, int accumulate, void* workspace, hipStream_t s,
                           const int32_t* x_stride = nullptr /* per segment; NULL = dense */,
                           const uint32_t* const* bits = nullptr /* gate words instead of gate (bf16x3 kernel only) */,
                           const float* w1 = nullptr, const float* b1 = nullptr,
                           const struct DwSecond* second = nullptr /* bits only: the LAST segment is a problem of its own */,
                           int dw_cols = 0 /* bits only: dw is [f_out, dw_cols <= f_in] (0 = f_in) */) {
    static bool attr_set = false;
    const size_t lds = (size_t)(2 * DW_KC * DW_LDA + 2 * DW_KC * DW_LDB) * sizeof(float);
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_dw_rank1_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    DwSegs sg;
    sg.nseg = nseg;
    for (int h = 0; h < 4; ++h) {
        const int q = h < nseg ? h : 0;
        sg.gate[h] = gate ? gate[q] : nullptr; sg.x[h] = x[q]; sg.rs[h] = row_scale[q]; sg.d_n[h] = d_n[q]; sg.n_cap[h] = h < nseg ? n_cap[q] : 0;
        sg.ldx[h] = (x_stride && x_stride[q] > 0) ? x_stride[q] : f_in;
        sg.bits[h] = bits ? bits[q] : nullptr;
    }
    bool strided = false;
    for (int h = 0; h < nseg; ++h) strided = strided || sg.ldx[h] != f_in;
    const long long slab = (long long)f_in * f_out;
    float* w_dw = (float*)workspace;
    float* w_db = w_dw + (size_t)DW_BLOCKS * slab;
    float* w_dh = w_db + (size_t)DW_BLOCKS * f_out;
    static int split = -1;      // GRAPES_GEMM_SPLIT=0: the fp32-MFMA kernel
    static bool attr2_set = false;
    if (split < 0) { const char* e = grapes_tune_env("GRAPES_GEMM_SPLIT"); split = e ? atoi(e) : 1; }
    if ((strided || bits) && !(split && dw_split_ok(f_in, f_out, bits != nullptr))) return GRAPES_EINVAL;     // only the bf16x3 kernel takes strided rows / gate words
    if (bits && dw_head && (!w1 || !b1)) return GRAPES_EINVAL;
    if (bits)       // gemm_dw_split_k<true> addresses a row set with 32-bit byte offsets from its base
        for (int h = 0; h < nseg; ++h)
            if ((unsigned long long)n_cap[h] * (unsigned long long)sg.ldx[h] * 4ull >= (1ull << 32)) return GRAPES_EINVAL;
    if (split && dw_split_ok(f_in, f_out, bits != nullptr)) {
        // the 160-column form (gate words only) when either problem's f_in needs it
        const bool wide = bits && (!dw_split_narrow(f_in) || (second && !dw_split_narrow(second->f_in)));
        if (second && !dw_split_ok(second->f_in, f_out, true)) return GRAPES_EINVAL;
        const size_t lds2 = (size_t)(2 * (4 * DW_MAXM + 3 * 4 * (wide ? DS_NT_WIDE : DS_NT)) + 64) * sizeof(uint4);     // (+ the dummy slots)
        if (!attr2_set) {
            const size_t l128 = (size_t)(2 * (4 * DW_MAXM + 3 * 4 * DS_NT) + 64) * sizeof(uint4), l160 = (size_t)(2 * (4 * DW_MAXM + 3 * 4 * DS_NT_WIDE) + 64) * sizeof(uint4);
            hipError_t e = hipFuncSetAttribute((const void*)gemm_dw_split_k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l128);
            if (e != hipSuccess) return (int)e;
            e = hipFuncSetAttribute((const void*)gemm_dw_split_k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l128);
            if (e != hipSuccess) return (int)e;
            e = hipFuncSetAttribute((const void*)gemm_dw_split_k<true, DS_NT_WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l160);
            if (e != hipSuccess) return (int)e;
            attr2_set = true;
        }
        if (bits) {
            // dual: the last segment is the second problem (its own f_in <= f_in, weights and outputs), on DW_BLOCKS_SECOND of the workgroups
            static int second_wgs = -1;   // GRAPES_DW_SECOND_WGS: tuning knob (profiles/): 8 .. 128
            if (second_wgs < 0) { const char* e = grapes_tune_env("GRAPES_DW_SECOND_WGS"); second_wgs = e ? atoi(e) : DW_BLOCKS_SECOND;
                                  if (second_wgs < 8 || second_wgs > 128) second_wgs = DW_BLOCKS_SECOND; }
            const int nwg0 = second ? DW_BLOCKS - second_wgs : DW_BLOCKS;
            float* w_dw2 = w_dw + (size_t)nwg0 * slab;
            float* w_db2 = w_db + (size_t)nwg0 * f_out;
            const DwAlt alt{nwg0, second ? nseg - 1 : nseg, second ? second->f_in : f_in, second ? second->col_vec : col_vec, w_dw2, w_db2};
            if (wide)
                hipLaunchKernelGGL((gemm_dw_split_k<true, DS_NT_WIDE>), dim3(DW_BLOCKS), dim3(512), lds2, s, sg, col_vec, f_out, f_in, w_dw, w_db,
                                   (float*)nullptr, alt);
            else
                hipLaunchKernelGGL(gemm_dw_split_k<true>, dim3(DW_BLOCKS), dim3(512), lds2, s, sg, col_vec, f_out, f_in, w_dw, w_db,
                                   (float*)nullptr, alt);
            GRAPES_LAUNCH_CHECK();
            const Sr1Prob pa{w_dw, w_db, col_vec, w1, b1, dw, dbias, dw_head, nwg0, f_in, 0, second ? nseg - 1 : nseg,
                             dw_cols > 0 ? dw_cols : f_in};
            const Sr1Prob pb = second ? Sr1Prob{w_dw2, w_db2, second->col_vec, second->w1, second->b1, second->dw, second->dbias,
                                                second->dw_head, DW_BLOCKS - nwg0, second->f_in, nseg - 1, nseg, second->f_in} : pa;
            if (wide)
                hipLaunchKernelGGL((slab_reduce_rank1_k<192, 5>), dim3(f_out, second ? 2 : 1), dim3(192 * 5), 0, s, sg, pa, pb, f_out, accumulate);
            else
                hipLaunchKernelGGL((slab_reduce_rank1_k<128, SR1_G>), dim3(f_out, second ? 2 : 1), dim3(128 * SR1_G), 0, s, sg, pa, pb, f_out, accumulate);
            GRAPES_LAUNCH_CHECK();
            return 0;
        }
        const DwAlt none{DW_BLOCKS, nseg, f_in, col_vec, w_dw, w_db};
        hipLaunchKernelGGL(gemm_dw_split_k<false>, dim3(DW_BLOCKS), dim3(512), lds2, s, sg, col_vec, f_out, f_in, w_dw,
                           dbias ? w_db : nullptr, dw_head ? w_dh : nullptr, none);
    } else {
        hipLaunchKernelGGL(gemm_dw_rank1_k, dim3(DW_BLOCKS), dim3(512), lds, s, sg, col_vec, f_out, f_in, w_dw,
                           dbias ? w_db : nullptr, dw_head ? w_dh : nullptr);
    }
    GRAPES_LAUNCH_CHECK();
    int grid = grapes_div_up(slab, 64); if (grid > 4096) grid = 4096;
    const int g2 = grid + (dbias ? grapes_div_up(f_out, 64) : 0) + (dw_head ? grapes_div_up(f_out, 64) : 0);
    hipLaunchKernelGGL(slab_reduce_k, dim3(g2), dim3(256), 0, s, (const float*)w_dw, dw, slab, DW_BLOCKS, (const int32_t*)nullptr, 1,
                       accumulate, (const float*)(dbias ? w_db : nullptr), dbias, (long long)(dbias ? f_out : 0),
                       (const float*)(dw_head ? w_dh : nullptr), dw_head, (long long)(dw_head ? f_out : 0));
    GRAPES_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t grapes_linear_bwd_weight_gated_workspace_bytes(int32_t n_cap, int32_t f_in, int32_t f_out) {
    if (n_cap <= 0) n_cap = 1;
    size_t nslab = (size_t)dw_nslab(f_out, f_in);
    if (nslab < DW_BLOCKS) nslab = DW_BLOCKS;          // (the bf16x3 kernels write one slab per workgroup whatever the tile count)
    // slabs of dW + slabs of db + slabs of the head's dW; the unfused fallback additionally materialises the gated dOut
    return (nslab * ((size_t)f_in * f_out + 2 * (size_t)f_out) + (size_t)n_cap * f_out) * sizeof(float) + grapes_colsum_workspace_bytes(f_out);
}

extern "C" int grapes_linear_bwd_weight_gated(const float* dout, const float* gate, const float* x, float* dw,
                                              float* dbias, int32_t n, const int32_t* d_n, int32_t f_in, int32_t f_out,
                                              int32_t accumulate, const float* row_scale, const float* col_vec,
                                              float* dw_head, void* workspace, grapes_stream_t stream) {
    if (n < 0 || f_in <= 0 || f_out <= 1 || !dw) return GRAPES_EINVAL;
    if ((row_scale == nullptr) != (col_vec == nullptr)) return GRAPES_EINVAL;
    if (dw_head && !row_scale) return GRAPES_EINVAL;
    const bool rank1 = row_scale != nullptr;    // dout = row_scale (x) col_vec, not materialised (dout may be NULL)
    if (rank1 && !gate) return GRAPES_EINVAL;
    if (rank1) dout = gate;                     // any valid [n,f_out] pointer: its values are never read
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        if (!accumulate) {
            hipError_t e = grapes_zero_async(dw, (size_t)f_in * f_out * sizeof(float), s); if (e) return (int)e;
            if (dbias) { e = grapes_zero_async(dbias, (size_t)f_out * sizeof(float), s); if (e) return (int)e; }
            if (dw_head) { e = grapes_zero_async(dw_head, (size_t)f_out * sizeof(float), s); if (e) return (int)e; }
        }
        return 0;
    }
    if (!dout || !x || !workspace) return GRAPES_EINVAL;
    const int nslab = dw_nslab(f_out, f_in);
    const long long slab = (long long)f_in * f_out;
    float* w_dw = (float*)workspace;
    float* w_db = w_dw + (size_t)nslab * slab;
    float* w_dh = w_db + (size_t)nslab * f_out;
    float* w_dpre = w_dh + (size_t)nslab * f_out;
    float* w_cs = w_dpre + (size_t)n * f_out;
    int grid = grapes_div_up(slab, 64); if (grid > 4096) grid = 4096;
    if (rank1 && !(fused_dw_ok(dout, gate, x, f_in, f_out) && aligned16(col_vec))) return GRAPES_EALIGN;
    if (rank1 && dw_rank1_ok(f_in, f_out) && DW_BLOCKS <= nslab) {       // dW-stationary kernel (one hop)
        const float* g1[1] = {gate}; const float* x1[1] = {x}; const float* r1[1] = {row_scale};
        const int32_t* d1[1] = {d_n}; const int32_t c1[1] = {n};
        return grapes_linear_bwd_weight_gated_multi(1, g1, x1, r1, d1, c1, col_vec, dw, dbias, dw_head, f_in, f_out, accumulate,
                                                    workspace, stream);
    }
    if (!rank1 && dw_small_ok(n, f_out, f_in)) {          // few rows: gate and bias sum inside the one-round-trip kernel
        int rc = launch_dw_small(dout, gate, x, w_dw, dbias ? w_db : nullptr, n, d_n, f_out, f_in, f_in, s);
        if (rc) return rc;
        const int g2 = grid + (dbias ? grapes_div_up(f_out, 64) : 0);
        hipLaunchKernelGGL(slab_reduce_k, dim3(g2), dim3(256), 0, s, (const float*)w_dw, dw, slab, n, d_n, DWS_ROWS, accumulate,
                           (const float*)(dbias ? w_db : nullptr), dbias, (long long)(dbias ? f_out : 0));
        GRAPES_LAUNCH_CHECK();
        return 0;
    }
    if (fused_dw_ok(dout, gate, x, f_in, f_out)) {
        GemmEx ex{nullptr, 0, gate, dbias ? w_db : nullptr, (long long)f_out, row_scale, col_vec, 0, dw_head ? w_dh : nullptr};
        int rc = launch_gemm<true, true>(dout, x, w_dw, f_out, f_in, n, f_out, f_in, f_in, nullptr, d_n, -nslab, nslab,
                                         slab, s, ex);
        if (rc) return rc;
        const int g2 = grid + (dbias ? grapes_div_up(f_out, 64) : 0) + (dw_head ? grapes_div_up(f_out, 64) : 0);
        hipLaunchKernelGGL(slab_reduce_k, dim3(g2), dim3(256), 0, s, (const float*)w_dw, dw, slab, n, d_n, -nslab, accumulate,
                           (const float*)(dbias ? w_db : nullptr), dbias, (long long)(dbias ? f_out : 0),
                           (const float*)(dw_head ? w_dh : nullptr), dw_head, (long long)(dw_head ? f_out : 0));
        GRAPES_LAUNCH_CHECK();
        return 0;
    }
    // unfused fallback (unaligned operands or no free padding column): gate + bias-sum pass, then the plain GEMM
    const float* a = dout;
    if (gate || dbias) {
        int rc = grapes_colsum_launch(dout, gate, nullptr, gate ? w_dpre : nullptr, dbias, n, d_n, f_out, accumulate, w_cs, s);
        if (rc) return rc;
        if (gate) a = w_dpre;
    }
    int rc = launch_gemm<true, true>(a, x, w_dw, f_out, f_in, n, f_out, f_in, f_in, nullptr, d_n, -nslab, nslab, slab, s);
    if (rc) return rc;
    hipLaunchKernelGGL(slab_reduce_k, dim3(grid), dim3(256), 0, s, (const float*)w_dw, dw, slab, n, d_n, -nslab, accumulate);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// The same backward for up to four hops that share the weights (the sampler GCN is applied at every hop), in ONE
// split-K launch + ONE slab reduction: hop h contributes its own rows (gate[h], x[h], row_scale[h], *d_n[h] of n_cap[h]).
// Rank-1 mode only (dAct = row_scale ⊗ col_vec).  The slabs are dealt evenly to the hops.
extern "C" int grapes_linear_bwd_weight_gated_multi(int32_t nseg, const float* const* gate, const float* const* x,
                                                    const float* const* row_scale, const int32_t* const* d_n,
                                                    const int32_t* n_cap, const float* col_vec, float* dw, float* dbias,
                                                    float* dw_head, int32_t f_in, int32_t f_out, int32_t accumulate,
                                                    void* workspace, grapes_stream_t stream) {
    if (nseg < 1 || nseg > 4 || !gate || !x || !row_scale || !d_n || !n_cap || !col_vec || !dw || !workspace) return GRAPES_EINVAL;
    if (f_in <= 0 || f_out <= 1) return GRAPES_EINVAL;
    for (int h = 0; h < nseg; ++h) {
        if (!gate[h] || !x[h] || !row_scale[h] || n_cap[h] <= 0) return GRAPES_EINVAL;
        if (!fused_dw_ok(gate[h], gate[h], x[h], f_in, f_out)) return GRAPES_EALIGN;
    }
    if (!aligned16(col_vec)) return GRAPES_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    static int use_stationary = -1;
    if (use_stationary < 0) { const char* e = grapes_tune_env("GRAPES_DW_STATIONARY"); use_stationary = e ? atoi(e) : 1; }
    if (use_stationary && dw_rank1_ok(f_in, f_out) && DW_BLOCKS <= dw_nslab(f_out, f_in))
        return launch_dw_rank1(nseg, gate, x, row_scale, d_n, n_cap, col_vec, dw, dbias, dw_head, f_in, f_out, accumulate, workspace, s);
    const int per = dw_nslab(f_out, f_in) / nseg > 0 ? dw_nslab(f_out, f_in) / nseg : 1;
    const int ntot = per * nseg;
    const long long slab = (long long)f_in * f_out;
    float* w_dw = (float*)workspace;
    float* w_db = w_dw + (size_t)ntot * slab;
    float* w_dh = w_db + (size_t)ntot * f_out;
    GemmEx ex{nullptr, 0, gate[0], dbias ? w_db : nullptr, (long long)f_out, row_scale[0], col_vec, 0, dw_head ? w_dh : nullptr};
    ex.nseg = nseg; ex.slabs_per_seg = per;
    for (int h = 1; h < 4; ++h) {
        const int q = h < nseg ? h : 0;
        ex.seg_B[h - 1] = x[q]; ex.seg_gate[h - 1] = gate[q]; ex.seg_rs[h - 1] = row_scale[q];
        ex.seg_dK[h - 1] = d_n[q]; ex.seg_K[h - 1] = n_cap[q];
    }
    int rc = launch_gemm<true, true>(gate[0], x[0], w_dw, f_out, f_in, n_cap[0], f_out, f_in, f_in, nullptr, d_n[0], -per, ntot,
                                     slab, s, ex);
    if (rc) return rc;
    int grid = grapes_div_up(slab, 64); if (grid > 4096) grid = 4096;
    const int g2 = grid + (dbias ? grapes_div_up(f_out, 64) : 0) + (dw_head ? grapes_div_up(f_out, 64) : 0);
    hipLaunchKernelGGL(slab_reduce_k, dim3(g2), dim3(256), 0, s, (const float*)w_dw, dw, slab, ntot, (const int32_t*)nullptr, 1,
                       accumulate, (const float*)(dbias ? w_db : nullptr), dbias, (long long)(dbias ? f_out : 0),
                       (const float*)(dw_head ? w_dh : nullptr), dw_head, (long long)(dw_head ? f_out : 0));
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// ---- first layers applied to data rows when F_in >= F_out (Reddit 602 + 3 -> 256, Cora 1433 + 3 -> 256): the reference
// order, transform then aggregate, is the cheap one (the aggregation then runs on F_out-wide rows), and the transform reads
// its operand  [X[ids] | indicators(ids)]  (main.py:199-204) through the id list — the gathered matrix is never written.
//   h[n, f_out]       = feat(ids) · Wᵀ            W [f_out, Kp], Kp = ceil4(F + num_ind), padding columns of W ignored (x 0)
//   dW[f_out, Kp] (+)= dhᵀ · feat(ids)            (padding columns of dW receive zeros)
// Few rows (the classifier's <= B + hops K rows, or a small graph): the forward K loop is cut into slabs as well, otherwise
// a handful of workgroups walk K = 1436 in 90 dependent steps.
static inline int fwd_gathered_nslab(int n, int f_out, int kp) {
    const int tiles = grapes_div_up(n, GB_M) * grapes_div_up(f_out, GB_N);
    if (tiles >= 128 || kp < 256) return 1;
    int ns = 256 / tiles; const int cap = kp / 64;
    if (ns > cap) ns = cap; if (ns > 16) ns = 16;
    return ns < 1 ? 1 : ns;
}
extern "C" size_t grapes_linear_gathered_workspace_bytes(int32_t n_cap, int32_t k_pad, int32_t f_out) {
    if (n_cap <= 0) n_cap = 1;
    const size_t fwd = (size_t)fwd_gathered_nslab(n_cap, f_out, k_pad) * n_cap * f_out;
    const size_t bwd = (size_t)dw_nslab(f_out, k_pad) * k_pad * f_out;
    return (fwd > bwd ? fwd : bwd) * sizeof(float) + 64;
}
static inline int gathered_args_ok(const float* X, int F, int x_stride, const int32_t* ids, const uint32_t* code, int num_ind) {
    if (!X || !ids || F <= 0 || num_ind < 0 || num_ind > 8 || x_stride < F || (x_stride & 3) || (num_ind > 0 && !code)) return GRAPES_EINVAL;
    if (!aligned16(X)) return GRAPES_EALIGN;
    return 0;
}
extern "C" int grapes_linear_fwd_gathered(const float* X, int32_t F, int32_t x_stride, const int32_t* ids,
                                          const uint32_t* ind_code, uint32_t epoch, const uint32_t* d_epoch, int32_t num_ind,
                                          const float* w, float* h, int32_t n, const int32_t* d_n, int32_t f_out,
                                          void* workspace, grapes_stream_t stream) {
    if (n < 0 || f_out <= 0 || !w || !h) return GRAPES_EINVAL;
    if (n == 0) return 0;
    int rc = gathered_args_ok(X, F, x_stride, ids, ind_code, num_ind);
    if (rc) return rc;
    const int kp = (F + num_ind + 3) & ~3;
    if (!aligned16(w) || !aligned16(h)) return GRAPES_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    GemmEx ex{nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, 0};
    ex.gather = 1; ex.ga = GatherOp{ids, ind_code, d_epoch, epoch, F, 0xffu};
    const int ns = fwd_gathered_nslab(n, f_out, kp);
    if (ns <= 1)
        return launch_gemm<false, false>(X, w, h, n, f_out, kp, x_stride, kp, f_out, d_n, nullptr, kp + GB_K, 1, 0, s, ex);
    if (!workspace) return GRAPES_EINVAL;
    int kchunk = grapes_div_up(kp, ns); kchunk = (kchunk + GB_K - 1) / GB_K * GB_K;
    const int nsl = grapes_div_up(kp, kchunk);
    const long long slab = (long long)n * f_out;
    rc = launch_gemm<false, false>(X, w, (float*)workspace, n, f_out, kp, x_stride, kp, f_out, d_n, nullptr, kchunk, nsl, slab, s, ex);
    if (rc) return rc;
    int grid = grapes_div_up(slab, 64); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(slab_reduce_k, dim3(grid), dim3(256), 0, s, (const float*)workspace, h, slab, kp, (const int32_t*)nullptr,
                       kchunk, 0);
    GRAPES_LAUNCH_CHECK();
    return 0;
}
extern "C" int grapes_linear_bwd_weight_gathered(const float* dh, const float* X, int32_t F, int32_t x_stride,
                                                 const int32_t* ids, const uint32_t* ind_code, uint32_t epoch,
                                                 const uint32_t* d_epoch, int32_t num_ind, uint32_t ind_mask, float* dw,
                                                 int32_t n, const int32_t* d_n, int32_t f_out, int32_t accumulate,
                                                 void* workspace, grapes_stream_t stream) {
    if (n < 0 || f_out <= 0 || !dw) return GRAPES_EINVAL;
    int rc = gathered_args_ok(X, F, x_stride, ids, ind_code, num_ind);
    if (rc) return rc;
    const int kp = (F + num_ind + 3) & ~3;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        if (!accumulate) { hipError_t e = grapes_zero_async(dw, (size_t)kp * f_out * sizeof(float), s); if (e) return (int)e; }
        return 0;
    }
    if (!dh || !workspace) return GRAPES_EINVAL;
    if (!aligned16(dh) || (f_out & 3)) return GRAPES_EALIGN;
    GemmEx ex{nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, 0};
    ex.gather = 2; ex.ga = GatherOp{ids, ind_code, d_epoch, epoch, F, ind_mask ? (ind_mask & 0xffu) : 0xffu};
    const int nslab = dw_nslab(f_out, kp);
    const long long slab = (long long)kp * f_out;
    rc = launch_gemm<true, true>(dh, X, (float*)workspace, f_out, kp, n, f_out, x_stride, kp, nullptr, d_n, -nslab, nslab, slab, s, ex);
    if (rc) return rc;
    int grid = grapes_div_up(slab, 64); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(slab_reduce_k, dim3(grid), dim3(256), 0, s, (const float*)workspace, dw, slab, n, d_n, -nslab, accumulate);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

extern "C" int grapes_linear_bwd_input(const float* dh, const float* w, float* dx, int32_t n, const int32_t* d_n,
                                       int32_t f_in, int32_t f_out, grapes_stream_t stream) {
    if (n < 0 || f_in <= 0 || f_out <= 0) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!dh || !w || !dx) return GRAPES_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (f_out == 1) {
        int grid = grapes_div_up((int64_t)n * f_in, 256); if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(outer_rows_k, dim3(grid), dim3(256), 0, s, dh, w, dx, n, d_n, f_in);
        GRAPES_LAUNCH_CHECK();
        return 0;
    }
    // dX[n,f_in] = dh[n,f_out] · W[f_out,f_in] : A = dh (k contiguous, K=f_out), B = W (k-major: rows are k)
    if (skinny_ok(n, f_out)) return launch_skinny<true>(dh, w, dx, n, d_n, f_in, f_out, f_out, f_in, f_in, nullptr, 0, s);
    return launch_gemm<false, true>(dh, w, dx, n, f_in, f_out, f_out, f_in, f_in, d_n, nullptr, f_out + GB_K, 1, 0, s);
}

// Strided-input forms (the bf16x3 kernels only): x rows are x_stride floats apart (x_stride >= f_in, a multiple of 4) — a
// layer can then read the leading f_in columns of a wider, already aggregated matrix (the log-Z net's first layer reads the
// feature columns of the sampler net's  Â [X | indicators]  at hop 0: main.py:227 feeds both nets the same rows).
extern "C" int32_t grapes_split_gemm_available(int32_t n, int32_t f_in, int32_t f_out) {
    static int split = -1;
    if (split < 0) { const char* e = grapes_tune_env("GRAPES_GEMM_SPLIT"); split = e ? atoi(e) : 1; }
    const bool fwd = f_in % 4 == 0 && f_in >= 4 && f_in <= 192 && f_out % 32 == 0 && f_out >= 32 && f_out <= 256 && n >= 2048;
    return (split && fwd && dw_split_ok(f_in, f_out, true)) ? 1 : 0;
}
extern "C" int grapes_linear_bias_act_head_fwd_strided(const float* x, int32_t x_stride, const float* w, const float* bias,
                                                       int32_t relu, float* out, const float* head_w, float* head_out,
                                                       int32_t n, const int32_t* d_n, int32_t f_in, int32_t f_out,
                                                       grapes_stream_t stream) {
    if (n <= 0 || !x || !w || !out || x_stride < f_in || (x_stride & 3)) return GRAPES_EINVAL;
    if ((head_w == nullptr) != (head_out == nullptr)) return GRAPES_EINVAL;
    if (!grapes_split_gemm_available(n, f_in, f_out) || !wsplit_ok(x, w, out, f_in, f_out)) return GRAPES_EINVAL;
    return launch_wsplit(x, w, bias, relu ? 1 : 0, out, n, d_n, f_in, f_out, (hipStream_t)stream, head_w, head_out, x_stride);
}
extern "C" int grapes_linear_bwd_weight_gated_strided(const float* gate, const float* x, int32_t x_stride,
                                                      const float* row_scale, int32_t n, const int32_t* d_n,
                                                      const float* col_vec, float* dw, float* dbias, float* dw_head,
                                                      int32_t f_in, int32_t f_out, int32_t accumulate, void* workspace,
                                                      grapes_stream_t stream) {
    if (n <= 0 || !gate || !x || !row_scale || !col_vec || !dw || !workspace || x_stride < f_in || (x_stride & 3)) return GRAPES_EINVAL;
    if (!grapes_split_gemm_available(n, f_in, f_out)) return GRAPES_EINVAL;
    if (!aligned16(gate) || !aligned16(x) || !aligned16(col_vec)) return GRAPES_EALIGN;
    const float* g1[1] = {gate}; const float* x1[1] = {x}; const float* r1[1] = {row_scale};
    const int32_t* d1[1] = {d_n}; const int32_t c1[1] = {n}; const int32_t s1[1] = {x_stride};
    return launch_dw_rank1(1, g1, x1, r1, d1, c1, col_vec, dw, dbias, dw_head, f_in, f_out, accumulate, workspace,
                           (hipStream_t)stream, s1);
}

// Gate-word forms of the layer + 1-wide head pair (bf16x3 kernels only: grapes_split_gemm_available): the forward pass
// writes head_out and 32 bytes of ReLU gate bits per row INSTEAD of the n x f_out activations, and the backward pass works
// from those bits (see wsplit_store / gemm_dw_split_k<true>).  Valid when the head is the activations' only consumer
// (the sampler net and the log-Z net: modules/gcn.py:31-36 with hidden_dims = [H, 1]).
extern "C" int grapes_linear_relu_head_fwd_bits(const float* x, int32_t x_stride, const float* w, const float* bias,
                                                const float* head_w, uint32_t* gate_bits, float* head_out, int32_t n,
                                                const int32_t* d_n, int32_t f_in, int32_t f_out, grapes_stream_t stream) {
    if (n <= 0 || !x || !w || !head_w || !gate_bits || !head_out || x_stride < f_in || (x_stride & 3)) return GRAPES_EINVAL;
    if (!grapes_split_gemm_available(n, f_in, f_out) || !wsplit_ok(x, w, gate_bits, f_in, f_out)) return GRAPES_EINVAL;
    return launch_wsplit(x, w, bias, 1, nullptr, n, d_n, f_in, f_out, (hipStream_t)stream, head_w, head_out, x_stride, gate_bits);
}
extern "C" int grapes_linear_relu_head_fwd_bits_pair(const float* x, int32_t x_stride, const float* w, const float* bias,
                                                     const float* head_w, uint32_t* gate_bits, float* head_out,
                                                     const float* x_b, int32_t x_stride_b, const float* w_b, const float* bias_b,
                                                     const float* head_w_b, uint32_t* gate_bits_b, float* head_out_b, int32_t f_in_b,
                                                     int32_t n, const int32_t* d_n, int32_t f_in, int32_t f_out,
                                                     grapes_stream_t stream) {
    if (n <= 0 || !x || !w || !head_w || !gate_bits || !head_out || x_stride < f_in || (x_stride & 3)) return GRAPES_EINVAL;
    if (!x_b || !w_b || !head_w_b || !gate_bits_b || !head_out_b || x_stride_b < f_in_b || (x_stride_b & 3)) return GRAPES_EINVAL;
    if (!grapes_split_gemm_available(n, f_in, f_out) || !wsplit_ok(x, w, gate_bits, f_in, f_out)) return GRAPES_EINVAL;
    if (!grapes_split_gemm_available(n, f_in_b, f_out) || !wsplit_ok(x_b, w_b, gate_bits_b, f_in_b, f_out)) return GRAPES_EINVAL;
    const WsplitAlt second{0, x_b, w_b, bias_b, head_w_b, head_out_b, gate_bits_b, f_in_b, x_stride_b};
    return launch_wsplit(x, w, bias, 1, nullptr, n, d_n, f_in, f_out, (hipStream_t)stream, head_w, head_out, x_stride, gate_bits, &second);
}
extern "C" int grapes_linear_bwd_weight_bits_multi_cols(int32_t nseg, const uint32_t* const* gate_bits, const float* const* x,
                                                        const int32_t* x_stride, const float* const* row_scale,
                                                        const int32_t* const* d_n, const int32_t* n_cap, const float* col_vec,
                                                        const float* w1, const float* b1, float* dw, int32_t dw_cols, float* dbias,
                                                        float* dw_head, int32_t f_in, int32_t f_out, int32_t accumulate,
                                                        void* workspace, grapes_stream_t stream) {
    if (nseg < 1 || nseg > 4 || !gate_bits || !x || !row_scale || !d_n || !n_cap || !col_vec || !dw || !workspace) return GRAPES_EINVAL;
    if (dw_cols < 0 || dw_cols > f_in || (dw_cols > 0 && f_in - dw_cols > 3)) return GRAPES_EINVAL;
    if (dw_head && (!w1 || !b1)) return GRAPES_EINVAL;
    int nmax = 0;
    for (int h = 0; h < nseg; ++h) {
        if (!gate_bits[h] || !x[h] || !row_scale[h] || n_cap[h] <= 0) return GRAPES_EINVAL;
        if (x_stride && x_stride[h] > 0 && (x_stride[h] < f_in || (x_stride[h] & 3))) return GRAPES_EINVAL;
        if (!aligned16(x[h])) return GRAPES_EALIGN;
        nmax = n_cap[h] > nmax ? n_cap[h] : nmax;
    }
    if (!aligned16(col_vec)) return GRAPES_EALIGN;
    if (!grapes_split_gemm_available(nmax > 2048 ? nmax : 2048, f_in, f_out)) return GRAPES_EINVAL;
    if (dw_cols > 0 && dw_cols != f_in && !dw_split_ok(f_in, f_out, true)) return GRAPES_EINVAL;      // (only the slab-sum form writes a pitch)
    return launch_dw_rank1(nseg, nullptr, x, row_scale, d_n, n_cap, col_vec, dw, dbias, dw_head, f_in, f_out, accumulate, workspace,
                           (hipStream_t)stream, x_stride, gate_bits, w1, b1, nullptr, dw_cols);
}
// ... and with ONE more row set that belongs to a DIFFERENT layer of the same f_out (its own f_in_b <= f_in, weights and
// gradient buffers): the log-Z net's backward beside the sampler net's (main.py:287 backpropagates through both) — one
// GEMM launch on disjoint workgroups + one slab reduction for the two.  Row set index nseg (the last entry of the operand
// arrays, which hold nseg + 1 <= 4 entries) is layer b's.
extern "C" int grapes_linear_bwd_weight_bits_pair_cols(int32_t nseg, const uint32_t* const* gate_bits, const float* const* x,
                                                       const int32_t* x_stride, const float* const* row_scale,
                                                       const int32_t* const* d_n, const int32_t* n_cap, const float* col_vec,
                                                       const float* w1, const float* b1, float* dw, int32_t dw_cols, float* dbias,
                                                       float* dw_head, int32_t f_in, const float* col_vec_b, const float* w1_b,
                                                       const float* b1_b, float* dw_b, float* dbias_b, float* dw_head_b,
                                                       int32_t f_in_b, int32_t f_out, int32_t accumulate, void* workspace,
                                                       grapes_stream_t stream) {
    if (nseg < 1 || nseg > 3 || !gate_bits || !x || !row_scale || !d_n || !n_cap || !col_vec || !dw || !workspace) return GRAPES_EINVAL;
    if (dw_cols < 0 || dw_cols > f_in || (dw_cols > 0 && f_in - dw_cols > 3)) return GRAPES_EINVAL;
    if (!col_vec_b || !dw_b || f_in_b <= 0 || f_in_b > f_in) return GRAPES_EINVAL;
    if ((dw_head && (!w1 || !b1)) || (dw_head_b && (!w1_b || !b1_b))) return GRAPES_EINVAL;
    int nmax = 0;
    for (int h = 0; h <= nseg; ++h) {
        const int fi = h < nseg ? f_in : f_in_b;
        if (!gate_bits[h] || !x[h] || !row_scale[h] || n_cap[h] <= 0) return GRAPES_EINVAL;
        if (x_stride && x_stride[h] > 0 && (x_stride[h] < fi || (x_stride[h] & 3))) return GRAPES_EINVAL;
        if (!aligned16(x[h])) return GRAPES_EALIGN;
        nmax = n_cap[h] > nmax ? n_cap[h] : nmax;
    }
    if (!aligned16(col_vec) || !aligned16(col_vec_b)) return GRAPES_EALIGN;
    if (!grapes_split_gemm_available(nmax > 2048 ? nmax : 2048, f_in, f_out) ||
        !grapes_split_gemm_available(nmax > 2048 ? nmax : 2048, f_in_b, f_out)) return GRAPES_EINVAL;
    // the second problem's rows are read with ITS stride: a dense x of layer b has rows f_in_b apart
    int32_t strides[4];
    for (int h = 0; h <= nseg; ++h) strides[h] = (x_stride && x_stride[h] > 0) ? x_stride[h] : (h < nseg ? f_in : f_in_b);
    const DwSecond sec{col_vec_b, w1_b, b1_b, dw_b, dbias_b, dw_head_b, f_in_b};
    return launch_dw_rank1(nseg + 1, nullptr, x, row_scale, d_n, n_cap, col_vec, dw, dbias, dw_head, f_in, f_out, accumulate, workspace,
                           (hipStream_t)stream, strides, gate_bits, w1, b1, &sec, dw_cols);
}

