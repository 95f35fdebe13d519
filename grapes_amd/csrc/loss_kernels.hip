// Losses and optimiser update of the step, fused on the device:
//   main.py:260     loss_c   = CrossEntropy / BCEWithLogits over the target rows of the classifier output
//   main.py:267     d loss_c / d logits (dense, zero outside the target rows) — what loss_c.backward() feeds gcn_c
//   main.py:272-282 cost_gfn = loss_c.detach(); loss_gfn = (log_z + sum(log_probs) + loss_coef * cost_gfn)^2
//                   (trajectory balance) or -sum(log_probs) * cost_gfn (REINFORCE, main.py:279)
//   main.py:268,289 Adam updates of both optimisers (torch.optim.Adam semantics, one launch for all tensors)
// One workgroup each for the two loss kernels (a few hundred rows); everything is fp32 like the reference.
#include "common.h"

#define LOSS_THREADS 1024
#define LOSS_MAX_B 4096

// deterministic sum of v[0..n) (n <= LOSS_MAX_B) by the whole workgroup: thread t owns t, t+T, ...; xor tree; waves in order
__device__ __forceinline__ float block_sum_fixed(const float* v, int n, float* red) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += v[i];
    acc = wave_sum(acc);
    __syncthreads();
    if (lane_id() == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    return t;
}

// main.py:272-282 for one thread.  stats: [hops][stride] floats, stats[h*stride + 4] = sum of hop h's log-probs
__device__ __forceinline__ void gflownet_loss_eval(bool has_z, float log_z_raw, float log_z_init,
                                                   const float* __restrict__ stats, int hops, int stride, float cost,
                                                   float loss_coef, int reinforce, float* __restrict__ out) {
    float tot = stats[4];
    for (int h = 1; h < hops; ++h) tot = __fadd_rn(tot, stats[h * stride + 4]);   // main.py:276
    const float lz = has_z ? __fsub_rn(log_z_raw, log_z_init) : 0.f;
    float loss, scale;
    if (reinforce) {
        loss = __fmul_rn(-tot, cost);                                             // main.py:279
        scale = -cost;
    } else {
        const float inner = __fadd_rn(__fadd_rn(lz, tot), __fmul_rn(loss_coef, cost));
        loss = __fmul_rn(inner, inner);                                           // main.py:282
        scale = __fmul_rn(2.0f, inner);
    }
    out[0] = loss; out[1] = scale; out[2] = lz; out[3] = tot;
}

// What grapes_step_losses adds to the classifier loss in the same workgroup: the mean of the log-Z head's output
// (main.py:228, the sum exactly as reduce_sum_k forms it) and the GFlowNet loss that needs both (main.py:272-282).
struct GfnTail {
    float* out4;                    // NULL = classifier loss only
    const float* zout; int nz; const int32_t* d_nz; float log_z_init;
    const float* stats; int hops, stride; float loss_coef; int reinforce;
    const float* loss_extra;        // NULL, or one float added to the classifier loss (the regulariser of main.py:260-261)
};

// One target row by one wavefront: loss of the row (every lane), its gradient row written.  CrossEntropy or BCE.
__device__ __forceinline__ float loss_row(const float* __restrict__ logits, float* __restrict__ dlogits, int n_rows, int C,
                                          int row, long long gid, const int64_t* __restrict__ labels,
                                          const float* __restrict__ labels_f, int multilabel, float inv, int lane) {
    float loss = 0.f;
    if ((unsigned)row < (unsigned)n_rows) {
        const float* x = logits + (long long)row * C;
        float* dx = dlogits + (long long)row * C;
        if (!multilabel) {
            const int y = (int)labels[gid];
            float m = -INFINITY;
            for (int c = lane; c < C; c += 64) m = fmaxf(m, x[c]);
            m = wave_max(m);
            float se = 0.f;
            for (int c = lane; c < C; c += 64) se += expf(x[c] - m);
            se = wave_sum(se);
            const float lse = logf(se);
            for (int c = lane; c < C; c += 64) {
                const float lsm = (x[c] - m) - lse;                            // log_softmax
                if (c == y) loss = -lsm;
                dx[c] = (expf(lsm) - (c == y ? 1.0f : 0.0f)) * inv;
            }
            loss = wave_sum(loss);
        } else {
            const float* yv = labels_f + gid * C;
            for (int c = lane; c < C; c += 64) {
                const float v = x[c], t = yv[c];
                loss += fmaxf(v, 0.f) - v * t + log1pf(expf(-fabsf(v)));       // stable BCE-with-logits
                dx[c] = (1.0f / (1.0f + expf(-v)) - t) * inv;
            }
            loss = wave_sum(loss);
        }
    }
    return loss;
}

// multilabel == 0: labels = int64 class ids (CrossEntropyLoss, mean over B)
// multilabel == 1: labels_f = fp32 [*, C] targets (BCEWithLogitsLoss, mean over B*C)        (main.py:120-123)
__global__ __launch_bounds__(LOSS_THREADS) void classifier_loss_k(
    const float* __restrict__ logits, int n_rows, int C, const int32_t* __restrict__ local_rows,
    const int32_t* __restrict__ node_map /* local_rows == NULL: row = node_map[target id] */,
    const int32_t* __restrict__ target_ids, const int64_t* __restrict__ labels, const float* __restrict__ labels_f,
    int B, int multilabel, float* __restrict__ dlogits, float* __restrict__ loss_out, GfnTail gt) {
    __shared__ float row_loss[LOSS_MAX_B];
    __shared__ float red[LOSS_THREADS / 64];
    __shared__ double zred[LOSS_THREADS / 64];
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6, nw = blockDim.x >> 6;
    const long long total = (long long)n_rows * C;
    double zs = 0.0;
    int nzv = 0;
    if (gt.out4 && gt.zout) {                         // issued first: its loads overlap the zero fill below
        nzv = eff_count(gt.d_nz, gt.nz);
        for (int i = tid; i < nzv; i += blockDim.x) zs += (double)gt.zout[i];
    }
    if ((((uintptr_t)dlogits) & 15) == 0) {
        float4* d4 = reinterpret_cast<float4*>(dlogits);
        const long long n4 = total >> 2;
        for (long long i = tid; i < n4; i += blockDim.x) d4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (long long i = (n4 << 2) + tid; i < total; i += blockDim.x) dlogits[i] = 0.f;
    } else {
        for (long long i = tid; i < total; i += blockDim.x) dlogits[i] = 0.f;
    }
    __syncthreads();
    const float inv = multilabel ? 1.0f / ((float)B * (float)C) : 1.0f / (float)B;
    if (!multilabel && C <= 64) {
        // class logits of one row fit one wavefront: ROWS_IN_FLIGHT rows per wavefront and pass, every load of the
        // pass issued before the first use (row index -> label / logit), then ALU + cross-lane work only
        constexpr int RIF = 16;        // B = 256 targets over 16 wavefronts: every row of the batch in ONE pass
        for (int b0 = wid * RIF; b0 < B; b0 += nw * RIF) {
            int row[RIF]; long long gid[RIF];
#pragma unroll
            for (int u = 0; u < RIF; ++u) {
                const int b = b0 + u < B ? b0 + u : B - 1;             // unconditional, clamped
                gid[u] = target_ids[b];
                row[u] = local_rows ? local_rows[b] : node_map[gid[u]];
            }
            float xv[RIF]; int yv[RIF];
#pragma unroll
            for (int u = 0; u < RIF; ++u) {
                const bool ok = (unsigned)row[u] < (unsigned)n_rows;
                const int r = ok ? row[u] : 0;
                xv[u] = logits[(long long)r * C + (lane < C ? lane : 0)];
                yv[u] = (int)labels[gid[u]];
            }
#pragma unroll
            for (int u = 0; u < RIF; ++u) {
                if (b0 + u >= B) continue;                              // uniform per wavefront
                const bool ok = (unsigned)row[u] < (unsigned)n_rows;
                const float x = lane < C ? xv[u] : -INFINITY;
                const float m = wave_max(x);
                const float se = wave_sum(lane < C ? expf(x - m) : 0.f);
                const float lsm = (x - m) - logf(se);
                const float l = wave_sum((lane == yv[u] && lane < C) ? -lsm : 0.f);
                if (ok && lane < C) dlogits[(long long)row[u] * C + lane] = (expf(lsm) - (lane == yv[u] ? 1.0f : 0.0f)) * inv;
                if (lane == 0) row_loss[b0 + u] = ok ? l : 0.f;
            }
        }
    } else
    for (int b = wid; b < B; b += nw) {                     // one wavefront per target row
        const long long gid = target_ids[b];
        const int row = local_rows ? local_rows[b] : node_map[gid];
        const float loss = loss_row(logits, dlogits, n_rows, C, row, gid, labels, labels_f, multilabel, inv, lane);
        if (lane == 0) row_loss[b] = loss;
    }
    __syncthreads();
    const float s = block_sum_fixed(row_loss, B, red);
    const float lc = s * inv + (gt.loss_extra ? *gt.loss_extra : 0.f);
    if (tid == 0) *loss_out = lc;
    if (!gt.out4) return;
    zs = wave_sum_d(zs);
    if (lane == 0) zred[wid] = zs;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < nw; ++w) t += zred[w];
        const float zmean = (float)(nzv > 0 ? t / (double)nzv : 0.0 / 0.0);
        gflownet_loss_eval(gt.zout != nullptr, zmean, gt.log_z_init, gt.stats, gt.hops, gt.stride, lc, gt.loss_coef,
                           gt.reinforce, gt.out4);
    }
}

// The same losses over MANY workgroups (the single workgroup spends most of its 27 us zero-filling the dense gradient
// and draining those stores).  Workgroup 0: the log-Z mean, summed exactly as reduce_sum_k does (1024 virtual threads).
// Workgroups 1..G-1: each builds the bitmap of target rows in LDS, zero-fills its share of the NON-target rows and
// handles its share of the targets (one wavefront per row, same arithmetic as the single-workgroup kernel); row losses
// are published (device-scope exchange), the last workgroup to finish sums them in the single-workgroup kernel's
// order (1024 virtual threads) and evaluates the GFlowNet loss => bit-identical results.
#define LOSS_MB_THREADS 256
#define LOSS_MB_MAXROWS 65536
__global__ __launch_bounds__(LOSS_MB_THREADS) void step_losses_mb_k(
    const float* __restrict__ logits, int n_rows, int C, const int32_t* __restrict__ node_map,
    const int32_t* __restrict__ target_ids, const int64_t* __restrict__ labels, const float* __restrict__ labels_f,
    int B, int multilabel, float* __restrict__ dlogits, float* __restrict__ loss_out, GfnTail gt,
    float* __restrict__ row_loss /* [B] + one double (8-byte aligned) */, unsigned* __restrict__ ticket) {
    __shared__ unsigned bm[LOSS_MB_MAXROWS / 32];
    __shared__ double dred[16];
    __shared__ float fred[16];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    const int G = gridDim.x;
    double* zslot = reinterpret_cast<double*>(row_loss + ((B + 1) & ~1));
    const float inv = multilabel ? 1.0f / ((float)B * (float)C) : 1.0f / (float)B;
    if (blockIdx.x == 0) {
        if (gt.out4 && gt.zout) {     // virtual thread v = tid + 256 q sums x[v], x[v + 1024], ...; virtual wave = v / 64
            const int nz = eff_count(gt.d_nz, gt.nz);
            double zs[4] = {0.0, 0.0, 0.0, 0.0};
            for (int i0 = 0; i0 < nz; i0 += 8 * 1024) {          // 32 independent loads in flight, added in index order
                float xv[8][4];
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int i = i0 + 1024 * u + tid + 256 * q;
                        xv[u][q] = gt.zout[i < nz ? i : nz - 1];   // unconditional, clamped
                    }
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { if (i0 + 1024 * u + tid + 256 * q < nz) zs[q] += (double)xv[u][q]; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { const double w = wave_sum_d(zs[q]); if (lane == 0) dred[wid + 4 * q] = w; }
            __syncthreads();
            if (tid == 0) {
                double t = 0.0;
                for (int w = 0; w < 16; ++w) t += dred[w];
                publish_f64(zslot, t);
            }
        }
    } else {
        const int nwords = (n_rows + 31) >> 5;
        // this wavefront's first target row is looked up together with the bitmap's ids (two dependent round trips once, not twice)
        const int nwv = (G - 1) * (blockDim.x >> 6);
        const int b_first = (blockIdx.x - 1) * (blockDim.x >> 6) + wid;
        const long long gid_first = target_ids[b_first < B ? b_first : 0];
        const int tb = tid < B ? tid : 0;
        const long long gid_t = target_ids[tb];
        const int row_first = node_map[gid_first];
        const int row_t = node_map[gid_t];
        for (int i = tid; i < nwords; i += blockDim.x) bm[i] = 0u;
        __syncthreads();
        for (int b = tid; b < B; b += blockDim.x) {
            const int row = b == tid ? row_t : node_map[target_ids[b]];
            if ((unsigned)row < (unsigned)n_rows) atomicOr(&bm[row >> 5], 1u << (row & 31));
        }
        __syncthreads();
        // zero fill of the non-target rows: float4 items of the flat matrix, dealt to workgroups 1..G-1
        const long long total = (long long)n_rows * C;
        const long long n4 = ((((uintptr_t)dlogits) & 15) == 0) ? (total >> 2) : 0;
        const long long stride = (long long)(G - 1) * blockDim.x;
        for (long long i = (long long)(blockIdx.x - 1) * blockDim.x + tid; i < n4; i += stride) {
            const int r0 = (int)((4 * i) / C), r1 = (int)((4 * i + 3) / C);
            const bool t0 = (bm[r0 >> 5] >> (r0 & 31)) & 1u, t1 = (bm[r1 >> 5] >> (r1 & 31)) & 1u;
            if (!t0 && !t1) {
                reinterpret_cast<float4*>(dlogits)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int r = (int)((4 * i + u) / C); if (!((bm[r >> 5] >> (r & 31)) & 1u)) dlogits[4 * i + u] = 0.f; }
            }
        }
        for (long long i = (n4 << 2) + (long long)(blockIdx.x - 1) * blockDim.x + tid; i < total; i += stride) {
            const int r = (int)(i / C);
            if (!((bm[r >> 5] >> (r & 31)) & 1u)) dlogits[i] = 0.f;
        }
        // target rows: wavefront (workgroup - 1, wid) takes b = its index, + number of wavefronts, ...
        for (int b = b_first; b < B; b += nwv) {
            const long long gid = b == b_first ? gid_first : (long long)target_ids[b];
            const int row = b == b_first ? row_first : node_map[gid];
            const float l = loss_row(logits, dlogits, n_rows, C, row, gid, labels, labels_f, multilabel, inv, lane);
            if (lane == 0) publish_f32(&row_loss[b], l);
        }
    }
    __syncthreads();
    if (tid == 0) s_last = (atomicAdd(ticket, 1u) == (unsigned)G - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    // ---- last workgroup: sum of the row losses in block_sum_fixed's order for 1024 threads (virtual thread v owns rows
    // v, v + 1024, ...; butterfly inside a virtual wave; virtual waves in index order)
    float ls[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i0 = 0; i0 < B; i0 += 1024) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + tid + 256 * q;
            if (i < B) ls[q] += __int_as_float(__hip_atomic_load((const int*)(row_loss + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float w = wave_sum(ls[q]); if (lane == 0) fred[wid + 4 * q] = w; }
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += fred[w];
        const float loss = t * inv + (gt.loss_extra ? *gt.loss_extra : 0.f);
        *loss_out = loss;
        *ticket = 0u;
        if (gt.out4) {
            float zmean = 0.f;
            if (gt.zout) {
                const int nz = eff_count(gt.d_nz, gt.nz);
                const double zt = __longlong_as_double(__hip_atomic_load((const long long*)zslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                zmean = (float)(nz > 0 ? zt / (double)nz : 0.0 / 0.0);
            }
            gflownet_loss_eval(gt.zout != nullptr, zmean, gt.log_z_init, gt.stats, gt.hops, gt.stride, loss, gt.loss_coef,
                               gt.reinforce, gt.out4);
        }
    }
}

__global__ void gflownet_loss_k(const float* __restrict__ log_z_raw, float log_z_init, const float* __restrict__ stats,
                                int hops, int stride, const float* __restrict__ loss_c, float loss_coef, int reinforce,
                                float* __restrict__ out /* [4]: loss_gfn, grad scale, log_z, sum log-probs */) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    gflownet_loss_eval(log_z_raw != nullptr, log_z_raw ? *log_z_raw : 0.f, log_z_init, stats, hops, stride, *loss_c,
                       loss_coef, reinforce, out);
}

// ---------------------------------------------------------------------------- Adam (torch.optim.Adam, amsgrad off)
struct AdamTensor {            // 96 bytes, built on the host, lives in device memory
    float* p; const float* g; float* m; float* v; float* step;
    long long n;
    double lr, beta1, beta2, eps, weight_decay;      // doubles as torch passes them: 1 - beta2 must not be formed in fp32
    int maximize; int pad_;
    // MIRRORS of a [rows, K] weight, written with the update (the launches that used to refresh them every step are gone):
    //   w_pad [rows, ld_pad] fp32, columns 0 .. K-1: the zero-padded copy the few-row / fp32 kernels read (NULL: none);
    //   img: the bf16x3 split image of gemm_tiled_split.hip (grapes_weight_split_image's layout; NULL: none).
    // The padding of both is written once by the caller (a first grapes_weight_split_image / copy) and never touched here.
    float* w_pad; void* img; int K; int ld_pad;
};
typedef __bf16 adam_bf16;
__device__ __forceinline__ void adam_mirror(const AdamTensor& d, long long i, float pnew) {
    // (a mirrored weight has < 2^31 elements — checked where the descriptor is built: 32-bit division, not the 64-bit routine)
    const unsigned iu = (unsigned)i, Ku = (unsigned)d.K;
    const int nrow = (int)(iu / Ku), k = (int)(iu - (unsigned)nrow * Ku);
    if (d.w_pad) d.w_pad[(long long)nrow * d.ld_pad + k] = pnew;
    if (d.img) {
        // ts_split3 (gemm_tiled_split.hip) and ts_weight_image_k's layout: img[k / 32][plane][k % 32 / 8][row 0..255][k % 8]
        const adam_bf16 h = (adam_bf16)pnew;
        const float r1 = pnew - (float)h;
        const adam_bf16 m = (adam_bf16)r1;
        const adam_bf16 l = (adam_bf16)(r1 - (float)m);
        adam_bf16* o = reinterpret_cast<adam_bf16*>(d.img) +
                       ((((long long)(k >> 5) * 3 * 4 + ((k >> 3) & 3)) * GRAPES_TS_IMG_ROWS + nrow) * 8 + (k & 7));
        o[0] = h; o[(long long)4 * GRAPES_TS_IMG_ROWS * 8] = m; o[(long long)8 * GRAPES_TS_IMG_ROWS * 8] = l;
    }
}

// Pending slab sums (grapes_linear_bwd_weight_slabs: the classifier's few-row weight gradients): a tensor whose gradient is the
// `out` of a set gets it summed HERE, element by element, in grapes_slab_reduce_sets' order — ((g0 + g1) + (g2 + g3)) over four
// consecutive slab groups — written back to .grad and used at once: the separate reduction launch in front of the update is gone.
#define ADAM_MAX_SETS 8
struct AdamSlabs { int nsets; int k_host; const int32_t* d_k; int accumulate; const float* slabs[ADAM_MAX_SETS]; const float* out[ADAM_MAX_SETS]; };
// ns <= 8 slabs whose values are already in registers: the additions in grapes_slab_reduce_sets' order
__device__ __forceinline__ float adam_slab_combine8(const float (&v)[8], int ns) {
    const int per = (ns + 3) >> 2;
    float part[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int z0 = g * per, z1 = (z0 + per < ns) ? z0 + per : ns;
        float acc = 0.f;
#pragma unroll
        for (int z = 0; z < 8; ++z) acc = (z >= z0 && z < z1) ? acc + v[z] : acc;
        part[g] = acc;
    }
    return (part[0] + part[1]) + (part[2] + part[3]);
}
__device__ __forceinline__ float adam_slab_sum(const float* __restrict__ sl, long long cnt, long long i, int ns) {
    if (ns <= 0) return 0.f;       // no live row: no slab was written (and slab -1 is in front of the workspace: ADVICE r03)
    const int per = (ns + 3) >> 2;
    float part[4];
    if (ns <= 8) {       // (<= 1024 rows: the classifier's sampled subgraph) all slab loads in flight, then the same order of additions
        float v[8];
#pragma unroll
        for (int z = 0; z < 8; ++z) v[z] = sl[(long long)(z < ns ? z : ns - 1) * cnt + i];
        return adam_slab_combine8(v, ns);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int z0 = g * per, z1 = (z0 + per < ns) ? z0 + per : ns;
        float acc = 0.f;
        for (int z = z0; z < z1; ++z) acc += sl[(long long)z * cnt + i];
        part[g] = acc;
    }
    return (part[0] + part[1]) + (part[2] + part[3]);
}
__global__ __launch_bounds__(256) void adam_step_k(const AdamTensor* __restrict__ desc, int n_tensors,
                                                   unsigned* __restrict__ ticket, AdamSlabs sb) {
    __shared__ int s_last;
    const AdamTensor d = desc[blockIdx.y];
    const float* sl = nullptr;                                   // this tensor's pending slabs (uniform over the workgroup)
#pragma unroll
    for (int q = 0; q < ADAM_MAX_SETS; ++q) if (q < sb.nsets && sb.out[q] == d.g) sl = sb.slabs[q];
    const int ns = sl ? (eff_count(sb.d_k, sb.k_host) + DWS_ROWS - 1) / DWS_ROWS : 0;
    // Two dependent memory round trips in all: the descriptor, then — together — the step counter and up to four elements
    // of each of the four arrays per thread (all sixteen loads in flight).  Every thread forms the bias corrections itself
    // (a handful of instructions) instead of waiting for thread 0 and a barrier; a thread used to walk its elements one
    // dependent round trip at a time (eight for the 256 x 256 weight).
    const float stepv = *d.step;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const float omb1 = (float)(1.0 - d.beta1), b2 = (float)d.beta2, omb2 = (float)(1.0 - d.beta2);
    const float eps = (float)d.eps, wd = (float)d.weight_decay;
    float step_size = 0.f, bc2_sqrt = 1.f;
    bool have = false;
    for (long long base = i0; base < d.n; base += 4 * stride) {
        float g[4], p[4], m[4], v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long i = base + u * stride;
            const long long ic = i < d.n ? i : base;
            // (.grad read unconditionally: a select below, no branch between the loads; the descriptor's pointers named as GLOBAL ones:
            // as generic pointers they are FLAT loads, which the compiler always waits for to the last one before anything else is requested)
            typedef const __attribute__((address_space(1))) float* gcf;
            g[u] = ((gcf)d.g)[ic]; p[u] = ((gcf)d.p)[ic]; m[u] = ((gcf)d.m)[ic]; v[u] = ((gcf)d.v)[ic];
        }
        const bool gzero = sl && !sb.accumulate;
        if (sl && ns > 0 && ns <= 8) {
            // the classifier's few-row gradients (<= 8 slabs): the slab words of all four elements are requested TOGETHER with the
            // sixteen loads above (clamped indices, no branch between the loads) — element by element each sum was a round trip of
            // its own behind them
            float sv[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long i = base + u * stride;
                const long long ic = i < d.n ? i : base;
#pragma unroll
                for (int z = 0; z < 8; ++z) sv[u][z] = sl[(long long)(z < ns ? z : ns - 1) * d.n + ic];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long i = base + u * stride;
                const float t = adam_slab_combine8(sv[u], ns);
                g[u] = sb.accumulate ? g[u] + t : t;
                if (i < d.n) const_cast<float*>(d.g)[i] = g[u];   // .grad holds the summed gradient, as after the separate launch
            }
        } else if (sl) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long i = base + u * stride;
                if (gzero) g[u] = 0.f;
                if (i < d.n) {
                    const float t = adam_slab_sum(sl, d.n, i, ns);
                    g[u] = sb.accumulate ? g[u] + t : t;
                    const_cast<float*>(d.g)[i] = g[u];            // .grad holds the summed gradient, as after the separate launch
                }
            }
        }
        if (!have) {
            // beta^step with the hardware exp2 / log2 (fp32: relative error ~1e-7 where beta^step matters, i.e. small steps; the
            // double-precision pow / exp / log routines are several KB of code that arrive cold in the instruction cache)
            const float fs = (float)((double)stepv + 1.0);
            const double bc1 = 1.0 - (double)exp2f(fs * log2f((float)d.beta1));
            const double bc2 = 1.0 - (double)exp2f(fs * log2f((float)d.beta2));
            step_size = (float)(d.lr / bc1);
            bc2_sqrt = (float)sqrt(bc2);
            have = true;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long i = base + u * stride;
            if (i < d.n) {
                float gg = d.maximize ? -g[u] : g[u];
                if (wd != 0.f) gg = __fmaf_rn(wd, p[u], gg);
                const float mm = m[u] + (gg - m[u]) * omb1;              // exp_avg.lerp_(grad, 1 - beta1)
                const float vv = b2 * v[u] + omb2 * (gg * gg);          // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
                const float denom = sqrtf(vv) / bc2_sqrt + eps;
                const float pnew = p[u] - step_size * (mm / denom);
                d.p[i] = pnew;
                d.m[i] = mm; d.v[i] = vv;
                if (d.w_pad || d.img) adam_mirror(d, i, pnew);
            }
        }
    }
    // the last workgroup of THIS tensor advances its step counter (all others have read it already); one ticket word per
    // tensor, so the atomics of different tensors do not serialise on one address
    __syncthreads();
    if (threadIdx.x == 0) {
        // no fence: the ticket only has to follow this workgroup's READ of the step counter, and that value has been
        // consumed above (a device-scope fence would wait for the L2 write-back of every store of the workgroup)
        const unsigned t = atomicAdd(&ticket[blockIdx.y], 1u);
        s_last = (t == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (s_last && threadIdx.x == 0) {                       // every tensor has its OWN step counter (checked on the host)
        *d.step = *d.step + 1.0f;
        ticket[blockIdx.y] = 0u;
    }
}

// ---------------------------------------------------------------------------- C-ABI
extern "C" int grapes_classifier_loss(const float* logits, int32_t n_rows, int32_t C, const int32_t* local_rows,
                                      const int32_t* target_ids, const int64_t* labels, const float* labels_f,
                                      int32_t B, float* dlogits, float* loss_out, grapes_stream_t stream) {
    if (!logits || !local_rows || !target_ids || !dlogits || !loss_out) return GRAPES_EINVAL;
    if ((labels == nullptr) == (labels_f == nullptr)) return GRAPES_EINVAL;
    if (n_rows <= 0 || C <= 0 || B <= 0 || B > LOSS_MAX_B) return GRAPES_EINVAL;
    GfnTail gt{};
    hipLaunchKernelGGL(classifier_loss_k, dim3(1), dim3(LOSS_THREADS), 0, (hipStream_t)stream, logits, n_rows, C,
                       local_rows, (const int32_t*)nullptr, target_ids, labels, labels_f, B, labels_f ? 1 : 0, dlogits,
                       loss_out, gt);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// eval.py:154-155 in one launch: predictions of the target nodes = argmax over the C logits of row node_map[target] (the FIRST
// largest; a NaN counts as the largest, as torch.argmax has it), with the rows themselves copied out.  One wavefront per target.
__global__ __launch_bounds__(256) void eval_predict_k(const float* __restrict__ logits, int n_rows, int C,
                                                      const int32_t* __restrict__ node_map, const int32_t* __restrict__ targets, int B,
                                                      long long* __restrict__ pred, float* __restrict__ rows_out, int32_t* status) {
    const int b = blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const int r = node_map[targets[b]];
    if ((unsigned)r >= (unsigned)n_rows) {          // a target that is not one of all_nodes: cannot happen in a clean step
        if (lane == 0) { pred[b] = 0; if (status) atomicOr(status, GRAPES_STATUS_BAD_INDEX); }
        return;
    }
    float best = 0.f; int bi = 0x7fffffff; bool bnan = false;
    for (int c = lane; c < C; c += 64) {
        const float v = logits[(long long)r * C + c];
        if (rows_out) rows_out[(long long)b * C + c] = v;
        const bool vn = v != v;
        const bool take = bi == 0x7fffffff || (vn && !bnan) || (!bnan && !vn && v > best);     // (ascending c: an equal value never replaces)
        if (take) { best = v; bi = c; bnan = vn; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64); const int on = __shfl_xor((int)bnan, o, 64);
        const bool better = oi != 0x7fffffff && (bi == 0x7fffffff || (on && !bnan) || (on == (int)bnan && (on ? oi < bi : (ov > best || (ov == best && oi < bi)))));
        if (better) { best = ov; bi = oi; bnan = on != 0; }
    }
    if (lane == 0) pred[b] = bi == 0x7fffffff ? 0 : bi;
}

extern "C" int grapes_eval_predict(const float* logits, int32_t n_rows, int32_t C, const int32_t* node_map, const int32_t* targets,
                                   int32_t B, int64_t* pred, float* rows_out, int32_t* status, grapes_stream_t stream) {
    if (n_rows < 0 || C <= 0 || B < 0) return GRAPES_EINVAL;
    if (B == 0) return 0;
    if (!logits || !node_map || !targets || !pred) return GRAPES_EINVAL;
    hipLaunchKernelGGL(eval_predict_k, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, n_rows, C, node_map, targets, B,
                       (long long*)pred, rows_out, status);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t grapes_step_losses_workspace_bytes(int32_t B) { return ((size_t)((B + 1) & ~1) + 2) * sizeof(float); }

extern "C" int grapes_step_losses(const float* logits, int32_t n_rows, int32_t C, const int32_t* node_map,
                                  const int32_t* target_ids, const int64_t* labels, const float* labels_f, int32_t B,
                                  float* dlogits, float* loss_out, const float* z_out, int32_t nz, const int32_t* d_nz,
                                  float log_z_init, const float* hop_stats, int32_t hops, int32_t stats_stride,
                                  float loss_coef, int32_t reinforce, float* out4, const float* loss_extra, void* workspace,
                                  uint32_t* d_ticket, grapes_stream_t stream) {
    if (!logits || !node_map || !target_ids || !dlogits || !loss_out || !out4) return GRAPES_EINVAL;
    if ((labels == nullptr) == (labels_f == nullptr)) return GRAPES_EINVAL;
    if (n_rows <= 0 || C <= 0 || B <= 0 || B > LOSS_MAX_B) return GRAPES_EINVAL;
    if (!hop_stats || hops <= 0 || stats_stride < 5 || nz < 0 || (z_out && nz == 0)) return GRAPES_EINVAL;
    GfnTail gt{out4, z_out, nz, d_nz, log_z_init, hop_stats, hops, stats_stride, loss_coef, reinforce, loss_extra};
    if (workspace && d_ticket && n_rows <= LOSS_MB_MAXROWS && (((uintptr_t)workspace) & 7) == 0) {
        int G = 1 + grapes_div_up(B, LOSS_MB_THREADS / 64 * 2);        // two target rows per wavefront
        if (G > 65) G = 65;
        hipLaunchKernelGGL(step_losses_mb_k, dim3(G), dim3(LOSS_MB_THREADS), 0, (hipStream_t)stream, logits, n_rows, C, node_map,
                           target_ids, labels, labels_f, B, labels_f ? 1 : 0, dlogits, loss_out, gt, (float*)workspace, d_ticket);
        GRAPES_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(classifier_loss_k, dim3(1), dim3(LOSS_THREADS), 0, (hipStream_t)stream, logits, n_rows, C,
                       (const int32_t*)nullptr, node_map, target_ids, labels, labels_f, B, labels_f ? 1 : 0, dlogits,
                       loss_out, gt);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// ---- reg_param * sum_r var(logits[r, :])  (main.py:260-261; torch.var: unbiased, over the class dimension, every row of the
// batch's logits).  mode 0: out[0] = that term (before grapes_step_losses, which adds it to the classifier loss and hence
// to the GFlowNet cost); mode 1: dlogits[r][c] += reg * 2 (x - mean_r) / (C - 1)  (after it).  One workgroup, a row per
// wavefront and pass; wavefront w sums its rows in row order, the sixteen partial sums are added in wavefront order.
__global__ __launch_bounds__(1024) void logit_var_reg_k(const float* __restrict__ logits, int n_host, const int32_t* d_n, int C,
                                                        float reg, float* __restrict__ out, float* __restrict__ dlogits) {
    __shared__ float red[16];
    const int n = eff_count(d_n, n_host);
    const int lane = lane_id(), wid = threadIdx.x >> 6;
    const float invc = 1.0f / (float)C, invc1 = C > 1 ? 1.0f / (float)(C - 1) : 0.0f / 0.0f;      // (C = 1: torch.var gives nan)
    float acc = 0.f;
    for (int r = wid; r < n; r += 16) {
        const float* x = logits + (long long)r * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += x[c];
        const float mean = wave_sum(s) * invc;
        float q = 0.f;
        for (int c = lane; c < C; c += 64) { const float d = x[c] - mean; q = fmaf(d, d, q); }
        acc += wave_sum(q) * invc1;
        if (dlogits) {
            float* dx = dlogits + (long long)r * C;
            for (int c = lane; c < C; c += 64) dx[c] += reg * 2.0f * (x[c] - mean) * invc1;
        }
    }
    if (lane == 0) red[wid] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && out) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        out[0] = reg * t;
    }
}
extern "C" int grapes_logit_var_reg(const float* logits, int32_t n, const int32_t* d_n, int32_t C, float reg, float* out,
                                    float* dlogits, grapes_stream_t stream) {
    if (!logits || n <= 0 || C <= 0 || (!out && !dlogits)) return GRAPES_EINVAL;
    hipLaunchKernelGGL(logit_var_reg_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, n, d_n, C, reg, out, dlogits);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

extern "C" int grapes_gflownet_loss(const float* log_z_raw, float log_z_init, const float* hop_stats, int32_t hops,
                                    int32_t stats_stride, const float* loss_c, float loss_coef, int32_t reinforce,
                                    float* out4, grapes_stream_t stream) {
    if (!hop_stats || hops <= 0 || stats_stride < 5 || !loss_c || !out4) return GRAPES_EINVAL;
    hipLaunchKernelGGL(gflownet_loss_k, dim3(1), dim3(64), 0, (hipStream_t)stream, log_z_raw, log_z_init, hop_stats, hops,
                       stats_stride, loss_c, loss_coef, reinforce, out4);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

extern "C" int32_t grapes_adam_desc_bytes(void) { return (int32_t)sizeof(AdamTensor); }

static int adam_launch(const void* d_desc, int32_t n_tensors, int64_t max_numel, uint32_t* d_ticket, const AdamSlabs& sb,
                       grapes_stream_t stream) {
    if (!d_desc || !d_ticket || n_tensors <= 0 || n_tensors > 256 || max_numel <= 0) return GRAPES_EINVAL;
    int gx = grapes_div_up(max_numel, 256 * 8); if (gx < 1) gx = 1; if (gx > 64) gx = 64;      // (256 workgroups per tensor: every launch +1 us of ticket atomics)
    hipLaunchKernelGGL(adam_step_k, dim3(gx, n_tensors), dim3(256), 0, (hipStream_t)stream, (const AdamTensor*)d_desc,
                       n_tensors, d_ticket, sb);
    GRAPES_LAUNCH_CHECK();
    return 0;
}
extern "C" int grapes_adam_step(const void* d_desc, int32_t n_tensors, int64_t max_numel, uint32_t* d_ticket,
                                grapes_stream_t stream) {
    AdamSlabs sb{}; sb.nsets = 0;
    return adam_launch(d_desc, n_tensors, max_numel, d_ticket, sb, stream);
}
/* grapes_adam_step with the slab sums of grapes_slab_reduce_sets folded in: a tensor whose gradient pointer equals grads[q] first
 * gets  grad (+)= sum of its slabs  (same order, same result as grapes_slab_reduce_sets(nsets, slabs, grads, counts, n, d_n,
 * accumulate)), written back to the gradient, then the Adam update.  EVERY grads[q] must be the gradient of one of the tensors
 * (else GRAPES_EINVAL is not detectable here: the caller checks) and counts[q] its element count. */
extern "C" int grapes_adam_step_slabs(const void* d_desc, int32_t n_tensors, int64_t max_numel, uint32_t* d_ticket, int32_t nsets,
                                      const float* const* slabs, const float* const* grads, int32_t n, const int32_t* d_n,
                                      int32_t accumulate, grapes_stream_t stream) {
    if (nsets < 0 || nsets > ADAM_MAX_SETS || (nsets > 0 && (!slabs || !grads || n <= 0))) return GRAPES_EINVAL;
    AdamSlabs sb{}; sb.nsets = nsets; sb.k_host = n; sb.d_k = d_n; sb.accumulate = accumulate;
    for (int q = 0; q < ADAM_MAX_SETS; ++q) {
        sb.slabs[q] = q < nsets ? slabs[q] : nullptr; sb.out[q] = q < nsets ? grads[q] : nullptr;
        if (q < nsets && (!slabs[q] || !grads[q])) return GRAPES_EINVAL;
    }
    return adam_launch(d_desc, n_tensors, max_numel, d_ticket, sb, stream);
}
