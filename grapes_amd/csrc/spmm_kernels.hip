// A7 (sparse part): the GCNConv aggregation  out = Â H (+ bias, ReLU)  and its transpose for the
// backward pass, as a gather-SpMM over the per-hop CSR (HBM-bound: 4·F+4 bytes per aggregated edge).
//
//   gcn_aggregate_k<VEC>        one wavefront per destination row (rows of at most GRAPES_LONG_ROW
//                               entries); a lane owns VEC consecutive features (VEC=4: one dwordx4 per
//                               lane = a whole 1 KiB row of 256 fp32 per wave-instruction); indices and
//                               weights are wave-uniform scalar loads; 8 neighbour rows in flight.
//   gcn_aggregate_chunks_k      longer rows (hub nodes: 10^2..10^4 entries in the backward CSR) are cut
//                               into items of GRAPES_LONG_ROW entries by gcn_prepare; one workgroup per
//                               item, 16 rows in flight per wavefront, so a hub is spread over the chip.
//   gcn_aggregate_combine_k     sums a long row's item partials in chunk order (+ self-loop, bias, ReLU).
//   gcn_aggregate_narrow_k      F <= 16 (the 1-wide logit heads): lanes run across ROWS; rows longer
//                               than 64 entries are reduced by the whole wavefront.
//   colsum_*                    bias gradient / ReLU backward / 1-wide dW: deterministic two-stage sums.
// Weights follow PyG: w_rc = dinv[r]·dinv[c] per edge, the unit self-loop (weight dinv[c]²) is added
// last, then the bias.  Every sum has a fixed order: results are bit-reproducible.
#include "common.h"
#include <cstdlib>

template <int VEC>
__device__ __forceinline__ void ld_vec(const float* __restrict__ p, float (&v)[VEC]) {
    if (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1 % VEC] = t.y; v[2 % VEC] = t.z; v[3 % VEC] = t.w;
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = p[i];
    }
}

// accumulate entries [beg,end) of one CSR row into acc (features f0..f0+VEC), U rows in flight
// PRE (full-batch inference only): the rows of h arrive PRE-SCALED by their own dinv (hs[r] = dinv[r] h[r], grapes_scale_rows), so
// an entry's weight is 1 and the row's own factor dinv[c] is applied once at the end:  out[c] = dinv[c] (sum hs[s] + hs[c]) + b.
// The per-edge gather of dinv[s] — a random 4-byte access, one more memory request per aggregated edge on top of the row
// itself — disappears.  Rounding differs from the w = dinv[s] dinv[c] form by an ulp or two (two roundings either way).
// MODE 2 (R1: backward of a transform-first layer under a 1-wide head, Reddit / Cora): the aggregated matrix is never stored —
// row r of it is  dpre[r][m] = [act[r][m] > 0] * (dh2[r] * w2[m])  (the ReLU-masked outer product of the head's gradient and
// weight: modules/gcn.py:32,36 differentiated), formed from the activation row as it is gathered.  Same products, same order as
// outer_rows_k + the masking pass + this aggregation did in three launches and 5 x n x H floats of traffic.
// MODE 4: MODE 2 with the activation rows replaced by their ReLU GATE BITS (written by the forward aggregation, MODE 3): row r =
// eight 32-bit words, element e = bit e % 32 of word e / 32 (F <= 256) — 32 bytes per gathered row instead of 4 F (a lane reads
// the one word that holds its four elements), and the same products in the same order (a set bit stands for "act > 0").
struct R1 { const float* dh2; const float* w2; };
#define R1_MODE(M) ((M) == 2 || (M) == 4)
template <int VEC, int MODE>
__device__ __forceinline__ void ld_row(const float* __restrict__ h, long long s, int F, int f0, float (&v)[VEC]) {
    if (MODE == 4) {
        const uint32_t wv = reinterpret_cast<const uint32_t*>(h)[8 * s + (f0 >> 5)] >> (f0 & 31);
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = ((wv >> i) & 1u) ? 1.f : 0.f;
    } else {
        ld_vec<VEC>(h + s * F + f0, v);
    }
}
template <int VEC, int MODE>
__device__ __forceinline__ void r1_gate(float (&val)[VEC], const float (&w2v)[VEC], float d) {
    if (R1_MODE(MODE)) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) val[v] = val[v] > 0.f ? d * w2v[v] : 0.f;
    }
}
template <int VEC, int U, int MODE = 0>
__device__ __forceinline__ void row_accumulate(const float* __restrict__ h, const int32_t* __restrict__ csr,
                                               const float* __restrict__ dinv, int beg, int end, float dc, int F,
                                               int f0, float (&acc)[VEC], const float* __restrict__ dh2 = nullptr,
                                               const float (*w2p)[VEC] = nullptr) {
    constexpr bool PRE = MODE == 1;
    float w2v[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) w2v[v] = R1_MODE(MODE) ? (*w2p)[v] : 0.f;
    int j = beg;
    for (; j + U <= end; j += U) {
        int s[U]; float w[U]; float val[U][VEC]; float dd[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { s[u] = csr[j + u]; w[u] = PRE ? 1.0f : dinv[s[u]] * dc; dd[u] = R1_MODE(MODE) ? dh2[s[u]] : 0.f; }
#pragma unroll
        for (int u = 0; u < U; ++u) ld_row<VEC, MODE>(h, s[u], F, f0, val[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) r1_gate<VEC, MODE>(val[u], w2v, dd[u]);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w[u], val[u][v], acc[v]);
    }
    if (U > 2 && j + 2 <= end) {   // pairs, then a single: short rows dominate the forward CSR
        for (; j + 2 <= end; j += 2) {
            const int s0 = csr[j], s1 = csr[j + 1];
            const float w0 = PRE ? 1.0f : dinv[s0] * dc, w1 = PRE ? 1.0f : dinv[s1] * dc;
            float v0[VEC], v1[VEC];
            ld_row<VEC, MODE>(h, s0, F, f0, v0);
            ld_row<VEC, MODE>(h, s1, F, f0, v1);
            if (R1_MODE(MODE)) { r1_gate<VEC, MODE>(v0, w2v, dh2[s0]); r1_gate<VEC, MODE>(v1, w2v, dh2[s1]); }
#pragma unroll
            for (int v = 0; v < VEC; ++v) { acc[v] = fmaf(w0, v0[v], acc[v]); acc[v] = fmaf(w1, v1[v], acc[v]); }
        }
    }
    for (; j < end; ++j) {
        const int s = csr[j];
        const float w = PRE ? 1.0f : dinv[s] * dc;
        float val[VEC];
        ld_row<VEC, MODE>(h, s, F, f0, val);
        if (R1_MODE(MODE)) r1_gate<VEC, MODE>(val, w2v, dh2[s]);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w, val[v], acc[v]);
    }
}

// ---- summation order of a HUB row (more than GRAPES_HUB_ROW entries), shared by every aggregation kernel of the per-hop graphs
// (gcn_aggregate_k, gcn_aggregate_gather_k, gcn_aggregate_gather_head_k) so that their results stay bit-identical:
//     sum = ((((s_0 + s_1) + s_2) + ... ) + s_7),   s_g = sum over the entries q = g, g + 8, g + 16, ... in increasing q,
// then the unit self-loop, then the bias.  Eight independent chains instead of one: a wavefront that owns the row keeps eight
// loads in flight, and the fused kernel can spread the chains over the eight row groups of a workgroup — a 100-entry hub of
// the frontier was 50 serial round trips (the tail that set the launch time of the step's gather-SpMM), now ~7.
// Rows of at most GRAPES_HUB_ROW entries keep the plain sequential order.
#define GRAPES_HUB_ROW 16
template <int VEC, int MODE = 0>
__device__ __forceinline__ void row_accumulate_hub(const float* __restrict__ h, const int32_t* __restrict__ csr,
                                                   const float* __restrict__ dinv, int beg, int end, float dc, int F,
                                                   int f0, float (&acc)[VEC], const float* __restrict__ dh2 = nullptr,
                                                   const float (*w2p)[VEC] = nullptr) {
    constexpr bool PRE = MODE == 1;
    float w2v[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) w2v[v] = R1_MODE(MODE) ? (*w2p)[v] : 0.f;
    float a8[8][VEC];
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int v = 0; v < VEC; ++v) a8[g][v] = 0.f;
    for (int j = beg; j < end; j += 8) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {           // four rows in flight at a time (registers: occupancy of the common path)
            int s[4]; float w[4]; float val[4][VEC];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int jj = j + 4 * half + u; s[u] = csr[jj < end ? jj : end - 1]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = PRE ? 1.0f : dinv[s[u]] * dc;
#pragma unroll
            for (int u = 0; u < 4; ++u) ld_row<VEC, MODE>(h, s[u], F, f0, val[u]);
            if (R1_MODE(MODE)) {
#pragma unroll
                for (int u = 0; u < 4; ++u) r1_gate<VEC, MODE>(val[u], w2v, dh2[s[u]]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j + 4 * half + u < end)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) a8[4 * half + u][v] = fmaf(w[u], val[u][v], a8[4 * half + u][v]);
        }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        float t = a8[0][v];
#pragma unroll
        for (int g = 1; g < 8; ++g) t += a8[g][v];
        acc[v] = t;
    }
}

// self-loop + bias + ReLU + store
template <int VEC, int MODE = 0>
__device__ __forceinline__ void row_finish(const float* __restrict__ h, const float* __restrict__ bias,
                                           float* __restrict__ out, int row, float dc, int F, int f0, int relu,
                                           float (&acc)[VEC], float (&self)[VEC] /* h[row][f0..] */, float dself /* MODE 2: dh2[row] */,
                                           const float (*w2p)[VEC] = nullptr, float* hdot = nullptr,
                                           unsigned* nib = nullptr /* MODE 3: this lane's four gate bits */) {
    constexpr bool PRE = MODE == 1;
    const float w = dc * dc;
    if (R1_MODE(MODE)) r1_gate<VEC, MODE>(self, *w2p, dself);
    float r[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        r[v] = PRE ? dc * (acc[v] + self[v]) : fmaf(w, self[v], acc[v]);
        if (bias) r[v] += bias[f0 + v];
        if (relu) r[v] = fmaxf(r[v], 0.f);
        if (MODE == 3) *hdot = fmaf(r[v], (*w2p)[v], *hdot);      // this lane's share of the 1-wide head that follows
    }
    if (MODE == 3) {
        unsigned b = 0u;
#pragma unroll
        for (int v = 0; v < VEC; ++v) b |= (r[v] > 0.f ? 1u : 0u) << v;
        *nib = b;
    }
    float* o = out + (long long)row * F + f0;
    if (VEC == 4) {
        *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1 % VEC], r[2 % VEC], r[3 % VEC]);
    } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) o[v] = r[v];
    }
}

template <int VEC, int MODE>
__device__ __forceinline__ void gcn_aggregate_body(const float* __restrict__ h, const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ csr, const float* __restrict__ dinv,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       int n_host, const int32_t* d_n, int F, int relu, int skip_long,
                                                       unsigned long long* clk, R1 r1,
                                                       float* __restrict__ head_out,
                                                       uint32_t* __restrict__ gate_bits, const int BID, const int NBLK) {
    // MODE 3: the forward aggregation that ALSO returns head_out[row] = out[row] . r1.w2 — the X W step of the 1-wide layer that
    // follows (modules/gcn.py:36 on main.py:210's [H, 1] layer) from the row while it is in registers: per lane the products in
    // column order, then a fixed exchange tree over the wavefront.
    const unsigned long long clk0 = grapes_clock_begin(clk);
    const int lane = lane_id();
    const int wave_global = __builtin_amdgcn_readfirstlane((BID * blockDim.x + threadIdx.x) >> 6);
    const int nwaves = (NBLK * blockDim.x) >> 6;
    // The live count and the wavefront's FIRST row's extent and dinv (inside the capacity) are requested together, through an opaque
    // per-lane offset of zero: loads the compiler knows to be uniform are moved into scalar registers — waited for — one by one where
    // they are issued (count, then row extent: two dependent round trips at the head of a launch whose rows are a round trip each).
    int zoff = 0;
    asm volatile("" : "+v"(zoff));
    const int n_vec = (d_n ? d_n : rowptr)[zoff];
    const int r0c = wave_global < n_host ? wave_global : 0;
    const int pb_vec = rowptr[r0c + zoff], pe_vec = rowptr[r0c + 1 + zoff];
    const float pd_vec = dinv[r0c + zoff];
    const int n_dev = __builtin_amdgcn_readfirstlane(n_vec);
    const int n = (d_n && n_dev < n_host) ? (n_dev < 0 ? 0 : n_dev) : n_host;          // (eff_count's clamp)
    const int pb0 = __builtin_amdgcn_readfirstlane(pb_vec), pe0 = __builtin_amdgcn_readfirstlane(pe_vec);
    const float pd0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(pd_vec)));
    // (this lane's bias and head weights are the same for every row of a one-pass width, but loading them once per wavefront
    // instead of once per row measured SLOWER — 21 -> 37 us at 23k rows: a wavefront owns ~1 row, and the loads then sit in
    // front of its chain instead of beside it)
    for (int row = wave_global; row < n; row += nwaves) {
        const bool first = row == wave_global;
        const int beg = first ? pb0 : rowptr[row], end = first ? pe0 : rowptr[row + 1];
        if (skip_long && end - beg > GRAPES_LONG_ROW) continue;   // chunk + combine kernels own it
        const float dc = first ? pd0 : dinv[row];
        float hdot = 0.f;
        unsigned nib = 0u;                                       // (lanes past the row's width keep zero bits)
        for (int f0 = lane * VEC; f0 < F; f0 += 64 * VEC) {
            // the row's own entry (the unit self-loop) does not depend on the row's extent: requested before the entries are
            // walked — most rows of a by-source hop graph hold nothing else, and their chain is then one round trip, not two
            float self[VEC];
            ld_row<VEC, MODE>(h, row, F, f0, self);
            const float dself = R1_MODE(MODE) ? r1.dh2[row] : 0.f;
            float acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
            float w2v[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) w2v[v] = (MODE >= 2) ? r1.w2[f0 + v] : 0.f;
            constexpr int AM = MODE == 3 ? 0 : MODE;             // (the head form gathers like the plain forward)
            if (end - beg > GRAPES_HUB_ROW) row_accumulate_hub<VEC, AM>(h, csr, dinv, beg, end, dc, F, f0, acc, r1.dh2, &w2v);
            else row_accumulate<VEC, 8, AM>(h, csr, dinv, beg, end, dc, F, f0, acc, r1.dh2, &w2v);
            row_finish<VEC, MODE>(h, bias, out, row, dc, F, f0, relu, acc, self, dself, &w2v, &hdot, &nib);
        }
        if (MODE == 3) {
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) hdot += __shfl_xor(hdot, d, 64);
            if (lane == 0) head_out[row] = hdot;
            if (gate_bits) {                                     // (F <= 256: one pass) eight lanes' nibbles make a word
                unsigned x = nib << (4 * (lane & 7));
                x |= (unsigned)__shfl_xor((int)x, 1, 64); x |= (unsigned)__shfl_xor((int)x, 2, 64); x |= (unsigned)__shfl_xor((int)x, 4, 64);
                if ((lane & 7) == 0) gate_bits[8 * (long long)row + (lane >> 3)] = x;
            }
        }
    }
    grapes_clock_end(clk, clk0);
}

template <int VEC, int MODE = 0>
__global__ __launch_bounds__(256) void gcn_aggregate_k(const float* __restrict__ h, const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ csr, const float* __restrict__ dinv,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       int n_host, const int32_t* d_n, int F, int relu, int skip_long,
                                                       unsigned long long* clk, R1 r1 = R1{nullptr, nullptr},
                                                       float* __restrict__ head_out = nullptr,
                                                       uint32_t* __restrict__ gate_bits = nullptr) {
    gcn_aggregate_body<VEC, MODE>(h, rowptr, csr, dinv, bias, out, n_host, d_n, F, relu, skip_long, clk, r1, head_out, gate_bits,
                                  (int)blockIdx.x, (int)gridDim.x);
}


// ---- The forward aggregation of a transform-first layer (rows of f <= 256 floats: one float4 per lane) driven by the graph
// build's per-row HEAD RECORDS with LOCAL ids (the build is given head_ids = 0, 1, 2, ...: 12 words per row — length, own id,
// dinv^2, dinv, the first four (source, weight) entries).  gcn_aggregate_k walks  rowptr -> csr -> (dinv[s], h[s]) -> store:
// three dependent round trips per row with ~1 KB of a wavefront's loads in flight, one row per wavefront — over a graph with
// NO entries it took 47 us for 81 MB in + 81 MB out (a copy of the same bytes: 20 us; Reddit's hop 2, 76k rows: 62 us = 0.51 of
// 8 TB/s by the algorithmic bytes).  Here a resident wavefront loops over PAIRS of rows: the pair's records were requested one
// iteration ahead, so a row is ONE dependent trip (record -> up to five row chunks, all requested together, both rows of the
// pair before the first FMA: ~4.7 KB of a wavefront's loads in flight, 16 wavefronts per CU); bias / head weights once per
// wavefront.  Rows longer than the record (len > 4) take the classic walk (CSR order; hub order beyond GRAPES_HUB_ROW), the
// same helpers as gcn_aggregate_k => every output is bit-identical to that kernel's.  MODE 0: out = act(Â h + b).  MODE 3: also
// head_out[row] = out[row] . w2 and the row's ReLU gate bits (see gcn_aggregate_k).
template <int MODE>
__global__ __launch_bounds__(256) void gcn_aggregate_rec_k(const float* __restrict__ h, const int4* __restrict__ rec,
                                                           const int32_t* __restrict__ rowptr, const int32_t* __restrict__ csr,
                                                           const float* __restrict__ dinv, const float* __restrict__ bias,
                                                           float* __restrict__ out, int n_host, const int32_t* d_n, int F, int relu,
                                                           unsigned long long* clk, const float* __restrict__ w2,
                                                           float* __restrict__ head_out, uint32_t* __restrict__ gate_bits) {
    const unsigned long long clk0 = grapes_clock_begin(clk);
    constexpr int R = 2;
    const int lane = lane_id();
    const int f0 = lane * 4;
    const bool live = f0 < F;
    const int fc = live ? f0 : 0;                     // (idle lanes of a narrower row read column 0 and store nothing)
    const int wave_global = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), w24 = b4;
    if (bias && live) b4 = *reinterpret_cast<const float4*>(bias + f0);
    if (MODE == 3 && live) w24 = *reinterpret_cast<const float4*>(w2 + f0);
    // a row's record as twelve wave-uniform words: every lane loads the same 48 bytes (one request) and the words move to
    // scalar registers at once (held in vector registers, the two pairs' records alone were 48 of them)
    struct Rec { int len, g0, g1, g2, g3; float dc, w0, w1, w2, w3; };
    auto to_rec = [](const int4& a, const int4& b, const int4& c) {
        Rec q;
        q.len = __builtin_amdgcn_readfirstlane(a.x); q.dc = __int_as_float(__builtin_amdgcn_readfirstlane(a.w));
        q.g0 = __builtin_amdgcn_readfirstlane(b.x); q.w0 = __int_as_float(__builtin_amdgcn_readfirstlane(b.y));
        q.g1 = __builtin_amdgcn_readfirstlane(b.z); q.w1 = __int_as_float(__builtin_amdgcn_readfirstlane(b.w));
        q.g2 = __builtin_amdgcn_readfirstlane(c.x); q.w2 = __int_as_float(__builtin_amdgcn_readfirstlane(c.y));
        q.g3 = __builtin_amdgcn_readfirstlane(c.z); q.w3 = __int_as_float(__builtin_amdgcn_readfirstlane(c.w));
        return q;
    };
    Rec ra[R];
    int row0 = wave_global * R;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int rr = row0 + r < n_host ? row0 + r : (n_host > 0 ? n_host - 1 : 0);     // inside the CAPACITY: before the live count
        ra[r] = to_rec(rec[3 * (long long)rr], rec[3 * (long long)rr + 1], rec[3 * (long long)rr + 2]);
    }
    const int n = eff_count(d_n, n_host);
    for (; row0 < n; row0 += nwaves * R) {
        // ---- requests: five chunks per short row, both rows
        float4 t[R][5];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r;
            if (row < n && ra[r].len <= 4) {
                const int g[5] = {ra[r].g0, ra[r].g1, ra[r].g2, ra[r].g3, row};
#pragma unroll
                for (int u = 0; u < 5; ++u) t[r][u] = *reinterpret_cast<const float4*>(h + (long long)g[u] * F + fc);
            } else {
#pragma unroll
                for (int u = 0; u < 5; ++u) t[r][u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        // ---- the next pair's records travel while this pair is summed
        int4 nr[R][3];
        const int next0 = row0 + nwaves * R;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int rr = next0 + r < n ? next0 + r : row0;
            nr[r][0] = rec[3 * (long long)rr]; nr[r][1] = rec[3 * (long long)rr + 1]; nr[r][2] = rec[3 * (long long)rr + 2];
        }
        // self-loop + bias + ReLU (+ head product, gate bits) and the store, as row_finish
        auto finish = [&](int row, float dc, const float (&acc)[4], const float (&self)[4]) {
            const float ws = dc * dc;
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, ww[4] = {w24.x, w24.y, w24.z, w24.w};
            float o[4];
            float hdot = 0.f;
            unsigned nib = 0u;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                o[v] = fmaf(ws, self[v], acc[v]);
                if (bias) o[v] += bb[v];
                if (relu) o[v] = fmaxf(o[v], 0.f);
                if (MODE == 3) { hdot = fmaf(o[v], ww[v], hdot); nib |= (o[v] > 0.f ? 1u : 0u) << v; }
            }
            if (!live) { hdot = 0.f; nib = 0u; }
            if (live) *reinterpret_cast<float4*>(out + (long long)row * F + f0) = make_float4(o[0], o[1], o[2], o[3]);
            if (MODE == 3) {
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) hdot += __shfl_xor(hdot, d, 64);
                if (lane == 0) head_out[row] = hdot;
                if (gate_bits) {
                    unsigned x = nib << (4 * (lane & 7));
                    x |= (unsigned)__shfl_xor((int)x, 1, 64); x |= (unsigned)__shfl_xor((int)x, 2, 64); x |= (unsigned)__shfl_xor((int)x, 4, 64);
                    if ((lane & 7) == 0) gate_bits[8 * (long long)row + (lane >> 3)] = x;
                }
            }
        };
#pragma unroll
        for (int r = 0; r < R; ++r) {                  // the rows that are whole in their record
            const int row = row0 + r;
            if (row >= n || ra[r].len > 4) continue;   // uniform
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            const float w[4] = {ra[r].w0, ra[r].w1, ra[r].w2, ra[r].w3};
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < ra[r].len) {                   // uniform; CSR order, as row_accumulate
                    acc[0] = fmaf(w[u], t[r][u].x, acc[0]); acc[1] = fmaf(w[u], t[r][u].y, acc[1]);
                    acc[2] = fmaf(w[u], t[r][u].z, acc[2]); acc[3] = fmaf(w[u], t[r][u].w, acc[3]);
                }
            const float self[4] = {t[r][4].x, t[r][4].y, t[r][4].z, t[r][4].w};
            finish(row, ra[r].dc, acc, self);
        }
#pragma unroll 1
        for (int r = 0; r < R; ++r) {                  // (rare) long rows afterwards, the pair's chunks no longer held: the classic
            const int row = row0 + r;                  // walk with the helpers and the order of gcn_aggregate_k
            const int ln = r == 0 ? ra[0].len : ra[R - 1].len;
            if (row >= n || ln <= 4) continue;         // uniform
            const float dc = r == 0 ? ra[0].dc : ra[R - 1].dc;
            const int beg = rowptr[row], end = rowptr[row + 1];
            float acc[4] = {0.f, 0.f, 0.f, 0.f}, self[4];
            ld_row<4, 0>(h, row, F, fc, self);
            if (end - beg > GRAPES_HUB_ROW) row_accumulate_hub<4, 0>(h, csr, dinv, beg, end, dc, F, fc, acc);
            else row_accumulate<4, 8, 0>(h, csr, dinv, beg, end, dc, F, fc, acc);
            finish(row, dc, acc, self);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) ra[r] = to_rec(nr[r][0], nr[r][1], nr[r][2]);
    }
    grapes_clock_end(clk, clk0);
}

// ---- gcn_aggregate_r1bits_k, below: the entries of one row (that kernel's rows with entries are the sampled sources of low out-degree; longer rows are the chunk
// kernel's): the wavefront fetches up to 64 entries' ids, factors and head gradients with ONE lane-parallel load each, then the
// entries' bit words R1B_BATCH at a time (requested together with the factors: they need the ids only) — a 64-entry row is 3
// round trips, not the ~32 of the eight-at-a-time walk that set the launch time (39 us at Reddit's hop 2 for a 17 us stream; a
// round trip under the launch's own store traffic is ~2.5 us).  Same products, same order as row_accumulate / row_accumulate_hub:
// rows of more than GRAPES_HUB_ROW entries in eight chains q % 8, summed ((a0 + a1) + ...) + a7.
#define R1B_BATCH 32
__device__ __forceinline__ void r1bits_row(const uint32_t* __restrict__ bits, const int32_t* __restrict__ csr,
                                           const float* __restrict__ dinv, const float* __restrict__ dh2, int beg, int end, float dc,
                                           int lane, const float (&w2v)[4], float (&acc)[4]) {
    const bool hub = end - beg > GRAPES_HUB_ROW;
    float a8[8][4];
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int v = 0; v < 4; ++v) a8[g][v] = 0.f;
    const int wsel = lane >> 3, nsh = 4 * (lane & 7);
    for (int b = beg; b < end; b += 64) {
        const int len = end - b < 64 ? end - b : 64;
        const int sl = lane < len ? csr[b + lane] : 0;
        const float wl = lane < len ? dinv[sl] * dc : 0.f;
        const float dl = lane < len ? dh2[sl] : 0.f;
        for (int q0 = 0; q0 < len; q0 += R1B_BATCH) {
            uint32_t wd[R1B_BATCH];
#pragma unroll
            for (int u = 0; u < R1B_BATCH; ++u) {
                const int sq = __builtin_amdgcn_readlane(sl, q0 + u);          // (lanes past len hold row 0: a valid address)
                wd[u] = bits[8 * (long long)sq + wsel];
            }
#pragma unroll
            for (int u = 0; u < R1B_BATCH; ++u) {
                if (q0 + u < len) {                                            // uniform
                    const float wq = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wl), q0 + u));
                    const float dq = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dl), q0 + u));
                    const uint32_t nb = wd[u] >> nsh;
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float val = ((nb >> v) & 1u) ? dq * w2v[v] : 0.f;
                        if (hub) a8[u & 7][v] = fmaf(wq, val, a8[u & 7][v]);   // (q0 and b - beg are multiples of 8: chain = u % 8)
                        else a8[0][v] = fmaf(wq, val, a8[0][v]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        float t = a8[0][v];
        if (hub) {
#pragma unroll
            for (int g = 1; g < 8; ++g) t += a8[g][v];
        }
        acc[v] = t;
    }
}
// ---- MODE 4 as its own kernel (the backward aggregation of a transform-first layer under a 1-wide head, by source, from gate
// bits): almost every row of a hop's by-source graph holds nothing but its unit self-loop (only the <= B + K sampled sources have
// entries), so the launch is a STREAM — per row 32 bytes of bits + two scalars in, F floats out.  A wavefront takes FOUR
// consecutive rows per round: their extents, factors and head gradients come in by ONE vector load each (lane q holds row q's;
// broadcast by readlane — as scalar loads this stream went through the scalar cache and the launch took 60 us instead of the
// 44 of the activation-row form), their bit words by one 32-byte load per row; the four output rows are 4 F contiguous floats.
// Rows with entries walk them as gcn_aggregate_k<4, 4> would (same helpers, same order): bit-identical to the MODE 2 launch.
__device__ __forceinline__ void gcn_aggregate_r1bits_body(const uint32_t* __restrict__ bits,
                                                          const int32_t* __restrict__ rowptr, const int32_t* __restrict__ csr,
                                                          const float* __restrict__ dinv, float* __restrict__ out, int n_host,
                                                          const int32_t* d_n, int F, int skip_long, R1 r1, int BID, int NBLK) {
    const int n = eff_count(d_n, n_host);
    const int lane = lane_id();
    const int wave_global = __builtin_amdgcn_readfirstlane((BID * 256 + (int)threadIdx.x) >> 6);
    const int nwaves = (NBLK * 256) >> 6;
    const int f0 = lane * 4;
    const bool livef = f0 < F;
    float w2v[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) w2v[v] = livef ? r1.w2[f0 + v] : 0.f;
    const int ngroups = (n + 3) >> 2;
    for (int g = wave_global; g < ngroups; g += nwaves) {
        const int r0 = 4 * g;
        const int rq = r0 + (lane & 7);
        const int rpv = rowptr[rq < n ? rq : n];                                  // lanes 0..4: the five extents
        const int rc = r0 + (lane & 3) < n ? r0 + (lane & 3) : n - 1;
        const float dcv = dinv[rc], ddv = r1.dh2[rc];                             // lanes 0..3: the four rows' factors
        uint32_t wq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) wq[q] = bits[8 * (long long)(r0 + q < n ? r0 + q : n - 1) + (lane >> 3)];
        int rp[5]; float dc[4], dd[4];
#pragma unroll
        for (int q = 0; q < 5; ++q) rp[q] = __builtin_amdgcn_readlane(rpv, q);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            dc[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dcv), q));
            dd[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ddv), q));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = r0 + q, beg = rp[q], end = rp[q + 1];
            if (row >= n) break;
            if (skip_long && end - beg > GRAPES_LONG_ROW) continue;   // chunk + combine kernels own it
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            if (end > beg) r1bits_row(bits, csr, dinv, r1.dh2, beg, end, dc[q], lane, w2v, acc);   // (every lane: the entries are fetched lane-parallel)
            if (!livef) continue;
            const float w = dc[q] * dc[q];
            const uint32_t nb = wq[q] >> (4 * (lane & 7));
            float r[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float sv = ((nb >> v) & 1u) ? dd[q] * w2v[v] : 0.f;
                r[v] = fmaf(w, sv, acc[v]);
            }
            *reinterpret_cast<float4*>(out + (long long)row * F + f0) = make_float4(r[0], r[1], r[2], r[3]);
        }
    }
}
__global__ __launch_bounds__(256) void gcn_aggregate_r1bits_k(const uint32_t* __restrict__ bits,
                                                              const int32_t* __restrict__ rowptr, const int32_t* __restrict__ csr,
                                                              const float* __restrict__ dinv, float* __restrict__ out, int n_host,
                                                              const int32_t* d_n, int F, int skip_long, R1 r1) {
    gcn_aggregate_r1bits_body(bits, rowptr, csr, dinv, out, n_host, d_n, F, skip_long, r1, (int)blockIdx.x, (int)gridDim.x);
}

// ---- rows narrower than a wavefront's 1 KiB (F <= 128: the full-batch passes at F = 100 and F = 48, eval.py:47-70).  With one
// dwordx4 per lane a row of F floats occupies F/4 lanes: at F = 100 the kernel above leaves 39 of 64 lanes idle (4.1 TB/s where
// the 256-wide pass reaches 5.3), at F = 47 it falls back to scalar loads.  Here a row still belongs to ONE wavefront, but its
// lanes form 64/LPR SLOTS of LPR lanes (LPR = 8 / 16 / 32 >= F/4): slot q takes the entries q, q + 64/LPR, ... of the row, four
// loads in flight per slot, and the slots' partial sums are combined by a fixed shuffle tree — every wave instruction moves up
// to 1 KiB again.  The summation order differs from the sequential kernels' (deterministic, within fp32 rounding), so this form
// is used for the item-scheduled (full-graph) aggregations only; the per-hop frontier graphs keep the sequential order.
template <int LPR, bool PRE = false>
__device__ __forceinline__ float4 lpr_accumulate(const float* __restrict__ h, const int32_t* __restrict__ csr,
                                                 const float* __restrict__ dinv, int beg, int end, float dc, int F, int sub,
                                                 int slot, bool live) {
    constexpr int EPW = 64 / LPR;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (beg >= end) return acc;
    for (int j = beg + slot; j < end; j += 4 * EPW) {
        int sidx[4]; float w[4]; float4 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int jj = j + u * EPW; sidx[u] = csr[jj < end ? jj : end - 1]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = (j + u * EPW < end) ? (PRE ? 1.0f : dinv[sidx[u]] * dc) : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = live ? *reinterpret_cast<const float4*>(h + (long long)sidx[u] * F + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (j + u * EPW < end) {
                acc.x = fmaf(w[u], t[u].x, acc.x); acc.y = fmaf(w[u], t[u].y, acc.y);
                acc.z = fmaf(w[u], t[u].z, acc.z); acc.w = fmaf(w[u], t[u].w, acc.w);
            }
        }
    }
    return acc;
}
template <int LPR>
__device__ __forceinline__ float4 lpr_combine_slots(float4 acc) {
#pragma unroll
    for (int d = LPR; d < 64; d <<= 1) {
        acc.x += __shfl_xor(acc.x, d, 64); acc.y += __shfl_xor(acc.y, d, 64);
        acc.z += __shfl_xor(acc.z, d, 64); acc.w += __shfl_xor(acc.w, d, 64);
    }
    return acc;
}
template <int LPR, bool PRE = false>
__global__ __launch_bounds__(256) void gcn_aggregate_lpr_k(const float* __restrict__ h, const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ csr, const float* __restrict__ dinv,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           int n_host, const int32_t* d_n, int F, int relu, int skip_long,
                                                           unsigned long long* clk) {
    const unsigned long long clk0 = grapes_clock_begin(clk);
    const int n = eff_count(d_n, n_host);
    const int lane = lane_id();
    const int slot = lane / LPR, sub = lane % LPR;
    const bool live = sub < (F >> 2);
    const int wave_global = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int row = wave_global; row < n; row += nwaves) {
        const int beg = rowptr[row], end = rowptr[row + 1];
        if (skip_long && end - beg > GRAPES_LONG_ROW) continue;   // chunk + combine kernels own it
        const float dc = dinv[row];
        float4 acc = lpr_combine_slots<LPR>(lpr_accumulate<LPR, PRE>(h, csr, dinv, beg, end, dc, F, sub, slot, live));
        if (slot == 0 && live) {
            const float4 sv = *reinterpret_cast<const float4*>(h + (long long)row * F + 4 * sub);
            const float w = dc * dc;
            float4 r = PRE ? make_float4(dc * (acc.x + sv.x), dc * (acc.y + sv.y), dc * (acc.z + sv.z), dc * (acc.w + sv.w))
                           : make_float4(fmaf(w, sv.x, acc.x), fmaf(w, sv.y, acc.y), fmaf(w, sv.z, acc.z), fmaf(w, sv.w, acc.w));
            if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + 4 * sub); r.x += b.x; r.y += b.y; r.z += b.z; r.w += b.w; }
            if (relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
            *reinterpret_cast<float4*>(out + (long long)row * F + 4 * sub) = r;
        }
    }
    grapes_clock_end(clk, clk0);
}
// one workgroup (4 wavefronts x 16 entries) per item of a long row, slots as above; partials[item][F]
template <int LPR, bool PRE = false>
__global__ __launch_bounds__(256) void gcn_aggregate_chunks_lpr_k(const float* __restrict__ h, const int32_t* __restrict__ rowptr,
                                                                  const int32_t* __restrict__ csr, const float* __restrict__ dinv,
                                                                  int F, const int32_t* __restrict__ items,
                                                                  const int32_t* __restrict__ d_n_items, int item_cap,
                                                                  float* __restrict__ partials) {
    __shared__ float4 part[4][32];
    int n_items = *d_n_items; if (n_items > item_cap) n_items = item_cap;
    const int lane = lane_id(), wid = threadIdx.x >> 6;
    const int slot = lane / LPR, sub = lane % LPR;
    const bool live = sub < (F >> 2);
    for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
        const int row = items[2 * it], chunk = items[2 * it + 1];
        const int rbeg = rowptr[row], rend = rowptr[row + 1];
        const int beg = rbeg + chunk * GRAPES_LONG_ROW;
        const int end = beg + GRAPES_LONG_ROW < rend ? beg + GRAPES_LONG_ROW : rend;
        const float dc = dinv[row];
        const int per = GRAPES_LONG_ROW / 4;
        const int wb = beg + wid * per;
        const int we = wb + per < end ? wb + per : end;
        const float4 acc = lpr_combine_slots<LPR>(lpr_accumulate<LPR, PRE>(h, csr, dinv, wb, we, dc, F, sub, slot, live));
        if (slot == 0 && sub < 32) part[wid][sub] = acc;
        __syncthreads();
        if (wid == 0 && slot == 0 && live) {
            const float4 a = part[0][sub], b = part[1][sub], c = part[2][sub], d = part[3][sub];
            *reinterpret_cast<float4*>(partials + (long long)it * F + 4 * sub) =
                make_float4(((a.x + b.x) + c.x) + d.x, ((a.y + b.y) + c.y) + d.y, ((a.z + b.z) + c.z) + d.z, ((a.w + b.w) + c.w) + d.w);
        }
        __syncthreads();
    }
}

// Aggregate-first first layer, fused with the feature gather of main.py:199-204:
//   out[c, :] = sum_{s in row c} (dinv[s] dinv[c]) feat(ids[s]) + dinv[c]^2 feat(ids[c]),
//   feat(v) = [ X[v, 0:F], indicator bits of v ]   (F + num_ind floats, a multiple of 4)
// i.e. Â · [X | ind] computed straight from the resident feature matrix — the gathered frontier
// feature matrix is never materialised.  LPR lanes serve one destination row (16 B per lane): with
// F + num_ind = 104 a row needs 26 lanes, so a wavefront carries two rows (LPR = 32).
template <int LPR>
__global__ __launch_bounds__(256) void gcn_aggregate_gather_k(const float* __restrict__ X, int F, int ldx,
                                                              const int32_t* __restrict__ ids,
                                                              const uint32_t* __restrict__ code, uint32_t epoch_host,
                                                              const uint32_t* d_epoch, int num_ind,
                                                              const int32_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ csr,
                                                              const float* __restrict__ dinv, float* __restrict__ out,
                                                              int n_host, const int32_t* d_n, unsigned long long* clk) {
    const unsigned long long clk0 = grapes_clock_begin(clk);
    const int n = eff_count(d_n, n_host);
    const uint32_t epoch = d_epoch ? (*d_epoch & 0xffffffu) : epoch_host;
    const int Fo = (F + num_ind + 3) & ~3;                         // output rows are padded to whole float4 chunks (zeros)
    const int chunks = Fo >> 2, xchunks = F >> 2;
    const int sub = threadIdx.x & (LPR - 1);                       // lane inside the row group
    const int rows_per_block = blockDim.x / LPR;
    const int rg = threadIdx.x / LPR;
    for (int row = blockIdx.x * rows_per_block + rg; row < n; row += gridDim.x * rows_per_block) {
        const int beg = rowptr[row], end = rowptr[row + 1];
        const float dc = dinv[row];
        const int len = end - beg;
        const int safe = beg < end ? beg : (beg > 0 ? beg - 1 : 0);   // a valid csr slot even for an empty row
        for (int c = sub; c < chunks; c += LPR) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (len > GRAPES_HUB_ROW) {            // hub row: eight strided chains (see row_accumulate_hub), then the self-loop
                float4 a8[8];
#pragma unroll
                for (int g = 0; g < 8; ++g) a8[g] = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int q0 = 0; q0 < len; q0 += 8) {
                    int sidx[8]; float w8[8]; int v8[8]; float4 t8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) sidx[u] = csr[beg + (q0 + u < len ? q0 + u : len - 1)];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { w8[u] = dinv[sidx[u]] * dc; v8[u] = ids[sidx[u]]; }
                    if (c < xchunks) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) t8[u] = *reinterpret_cast<const float4*>(X + (long long)v8[u] * ldx + c * 4);
                    } else {
#pragma unroll
                        for (int u = 0; u < 8; ++u) t8[u] = feat_tail_chunk(X, ldx, F, v8[u], c, code, epoch);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (q0 + u < len) {
                            a8[u].x = fmaf(w8[u], t8[u].x, a8[u].x); a8[u].y = fmaf(w8[u], t8[u].y, a8[u].y);
                            a8[u].z = fmaf(w8[u], t8[u].z, a8[u].z); a8[u].w = fmaf(w8[u], t8[u].w, a8[u].w);
                        }
                    }
                }
                acc = a8[0];
#pragma unroll
                for (int g = 1; g < 8; ++g) { acc.x += a8[g].x; acc.y += a8[g].y; acc.z += a8[g].z; acc.w += a8[g].w; }
                const int vs = ids[row];
                const float4 ts = (c < xchunks) ? *reinterpret_cast<const float4*>(X + (long long)vs * ldx + c * 4)
                                                : feat_tail_chunk(X, ldx, F, vs, c, code, epoch);
                const float wss = dinv[row] * dc;
                acc.x = fmaf(wss, ts.x, acc.x); acc.y = fmaf(wss, ts.y, acc.y);
                acc.z = fmaf(wss, ts.z, acc.z); acc.w = fmaf(wss, ts.w, acc.w);
                *reinterpret_cast<float4*>(out + (long long)row * Fo + c * 4) = acc;
                continue;
            }
            // items 0..len-1 = the row's entries, item len = the unit self-loop (added last).  Four items per
            // batch: their index, id and row loads are UNCONDITIONAL (clamped) and issued together, so a typical
            // frontier row (1-3 entries) costs three dependent memory round trips in total, not three per entry.
            for (int q0 = 0; q0 <= len; q0 += 4) {
                int s[4]; float w[4]; int v[4]; float4 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = q0 + u;
                    const int cj = csr[q < len ? beg + q : safe];
                    s[u] = q < len ? cj : row;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    w[u] = (q0 + u <= len) ? dinv[s[u]] * dc : 0.f;
                    v[u] = ids[s[u]];
                }
                if (c < xchunks) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const float4*>(X + (long long)v[u] * ldx + c * 4);
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) t[u] = feat_tail_chunk(X, ldx, F, v[u], c, code, epoch);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc.x = fmaf(w[u], t[u].x, acc.x); acc.y = fmaf(w[u], t[u].y, acc.y);
                    acc.z = fmaf(w[u], t[u].z, acc.z); acc.w = fmaf(w[u], t[u].w, acc.w);
                }
            }
            *reinterpret_cast<float4*>(out + (long long)row * Fo + c * 4) = acc;
        }
    }
    grapes_clock_end(clk, clk0);
}

// Same product from the per-row HEAD records gcn_prepare writes (12 words: len, gid_self, w_self, dinv, and the first
// four (gid, weight) entries): a frontier row needs two dependent memory round trips — head, then up to five feature
// rows in flight together — instead of four (rowptr -> csr -> ids / dinv -> rows).  Entries beyond the fourth (rare:
// the by-target rows of a frontier graph are short) continue through the CSR.  Summation order = CSR order, then
// the self-loop: bit-identical to gcn_aggregate_gather_k.
// Indicator bit `b` (0..7) of a code word for the running epoch, as 0.f / 1.f  (main.py:199-204: the one-hot hop columns)
__device__ __forceinline__ float code_bit_f(uint32_t cd, uint32_t epoch, int b) {
    return ((cd >> 8) == epoch && ((cd >> b) & 1u)) ? 1.f : 0.f;
}

template <int LPR>
__global__ __launch_bounds__(256) void gcn_aggregate_gather_head_k(const float* __restrict__ X, int F, int ldx,
                                                                   const int32_t* __restrict__ ids,
                                                                   const uint32_t* __restrict__ code, uint32_t epoch_host,
                                                                   const uint32_t* d_epoch, int num_ind,
                                                                   const int32_t* __restrict__ rowptr,
                                                                   const int32_t* __restrict__ csr,
                                                                   const float* __restrict__ dinv,
                                                                   const int4* __restrict__ head, float* __restrict__ out,
                                                                   int n_host, const int32_t* d_n, unsigned long long* clk) {
    const unsigned long long clk0 = grapes_clock_begin(clk);
    const int Fo = (F + num_ind + 3) & ~3;
    const int chunks = Fo >> 2;
    const int xch = F >> 2;                  // chunks wholly inside X: one lane each, four columns, no correction
    const int tcols = Fo - 4 * xch;          // the columns after them (rest of X, indicators, padding; <= 12): one lane each
    const int sub = threadIdx.x & (LPR - 1);
    constexpr int RPB = 256 / LPR;                                  // row groups per workgroup (8 or 4)
    const int rg = threadIdx.x / LPR;
    __shared__ int s_hub[3][RPB];
    __shared__ int s_nhub[3];               // hub-row counters of three consecutive batches: ONE barrier per batch (see below)
    __shared__ float4 s_part[8][LPR];
    GRAPES_STAMP(0);
    // The kernel is priced in vector instructions per row as much as in bytes (a frontier row is ~1 KB of loads), and in
    // dependent memory round trips per batch of RPB rows: one.  The workgroups are resident and loop.  In a batch every
    // load is unconditional (clamped) and issued together — feature chunks, the tail lanes' scalars and code words, then
    // the NEXT batch's records — one wait, arithmetic, stores; the records a batch needs are in registers when it starts,
    // and the previous batch's stores are acknowledged while this batch's loads travel.  The columns that are not a whole
    // chunk of X (indicator columns, the last F % 4 features, padding) are one LANE each (lanes 0..tcols-1 of the row group)
    // instead of a per-lane correction of every chunk: the correction cost more instructions than the product itself.
    // The first batch's records are requested before the live row count is known (inside the CAPACITY; a stale record past
    // n is dropped), together with the count and the epoch.
    int4 h0 = make_int4(0, 0, 0, 0), h1 = h0, h2 = h0;
    int stamp_it = 0; (void)stamp_it;
    {
        const int row0 = blockIdx.x * RPB + rg;
        const int r0 = row0 < n_host ? row0 : 0;
        h0 = head[3 * (long long)r0]; h1 = head[3 * (long long)r0 + 1]; h2 = head[3 * (long long)r0 + 2];
    }
    const int n = eff_count(d_n, n_host);
    const uint32_t epoch = d_epoch ? (*d_epoch & 0xffffffu) : epoch_host;
    // Hub rows of a batch are collected in LDS and shared by the whole workgroup after ONE LDS-only barrier per batch
    // (lds_barrier: no wait on global memory).  Counter slot b % 3 belongs to batch b; it is cleared right after batch
    // b - 2's barrier, which every wavefront passes before it can add to the slot (batch b, after barrier b - 1) and after
    // it has read the slot's previous use (batch b - 3).
    if (threadIdx.x < 3) s_nhub[threadIdx.x] = 0;
    lds_barrier();
    if (n > 0 && (int)blockIdx.x * RPB + rg >= n) {          // a row group past the last row: it repeats row n - 1's loads
        const long long rl = n - 1;                           // (whose record is valid; the one read above is stale)
        h0 = head[3 * rl]; h1 = head[3 * rl + 1]; h2 = head[3 * rl + 2];
    }
    asm volatile("" :: "v"(h0.x), "v"(h1.x), "v"(h2.x));    // the first records are waited for HERE on every path into the loop
    int slot = 0;                                            // (else the loop head inherits "maybe pending" and drains the queue every batch)
    const int tcol = 4 * xch + sub;                          // this lane's tail column (lanes sub < tcols)
    const int tcol_x = tcol < ldx ? tcol : ldx - 1;          // its column of X when tcol < F (clamped: always a valid load)
    for (int base = blockIdx.x * RPB; base < n; base += gridDim.x * RPB) {       // uniform per workgroup
        const bool own = base + rg < n;              // row groups past the last row repeat row n - 1's loads, store nothing
        const int row = own ? base + rg : n - 1;
        if (stamp_it < 4) GRAPES_STAMP(stamp_it * 3 + 1);
        const int len = h0.x;
        const bool hub = len > GRAPES_HUB_ROW;       // hub row: all row groups of the workgroup share it below
        if (hub && own && sub == 0) s_hub[slot][atomicAdd(&s_nhub[slot], 1)] = row;
        const int g[5] = {h1.x, h1.z, h2.x, h2.z, h0.y};                             // four entries, then self
        const float w[5] = {__int_as_float(h1.y), __int_as_float(h1.w), __int_as_float(h2.y), __int_as_float(h2.w),
                            __int_as_float(h0.z)};
        const float dc = __int_as_float(h0.w);
        const float* xr[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) xr[u] = X + (size_t)(uint32_t)g[u] * (size_t)(uint32_t)ldx;
        const int nrow = base + gridDim.x * RPB + rg;
        const int nsafe = nrow < n ? nrow : n - 1;
        const bool store_ok = own && !hub;
        float tx[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        uint32_t cw[5] = {0u, 0u, 0u, 0u, 0u};
        for (int cb = 0; cb < xch || cb == 0; cb += LPR) {
            const int c = cb + sub;
            const bool live = c < xch;
            const int cc = live ? c : (xch > 0 ? xch - 1 : 0);          // lanes past the last whole chunk repeat its load
            float4 t[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) t[u] = *reinterpret_cast<const float4*>(xr[u] + 4 * cc);
            if (cb == 0) {                           // (uniform) first pass: the tail lanes' operands and the next records
                if (F & 3) {
#pragma unroll
                    for (int u = 0; u < 5; ++u) tx[u] = xr[u][tcol_x];
                }
                if (num_ind > 0) {
#pragma unroll
                    for (int u = 0; u < 5; ++u) cw[u] = code[g[u]];
                }
                // h0..h2 are dead from here: every field was copied out above
                h0 = head[3 * (long long)nsafe]; h1 = head[3 * (long long)nsafe + 1]; h2 = head[3 * (long long)nsafe + 2];
            }
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc.x = fmaf(w[u], t[u].x, acc.x); acc.y = fmaf(w[u], t[u].y, acc.y);
                acc.z = fmaf(w[u], t[u].z, acc.z); acc.w = fmaf(w[u], t[u].w, acc.w);
            }
            if (len > 4 && !hub) {           // entries 5 .. GRAPES_HUB_ROW through the CSR, two per round trip, in CSR order
                const int beg = rowptr[row];
#pragma unroll 1
                for (int q = 4; q < len; q += 2) {
                    const int q1 = q + 1 < len ? q + 1 : q;
                    const int s0 = csr[beg + q], s1 = csr[beg + q1];
                    const float w0 = dinv[s0] * dc, w1 = dinv[s1] * dc;
                    const int v0 = ids[s0], v1 = ids[s1];
                    const float4 t0 = *reinterpret_cast<const float4*>(X + (size_t)(uint32_t)v0 * (size_t)(uint32_t)ldx + 4 * cc);
                    const float4 t1 = *reinterpret_cast<const float4*>(X + (size_t)(uint32_t)v1 * (size_t)(uint32_t)ldx + 4 * cc);
                    acc.x = fmaf(w0, t0.x, acc.x); acc.y = fmaf(w0, t0.y, acc.y);
                    acc.z = fmaf(w0, t0.z, acc.z); acc.w = fmaf(w0, t0.w, acc.w);
                    if (q + 1 < len) {
                        acc.x = fmaf(w1, t1.x, acc.x); acc.y = fmaf(w1, t1.y, acc.y);
                        acc.z = fmaf(w1, t1.z, acc.z); acc.w = fmaf(w1, t1.w, acc.w);
                    }
                }
            }
            acc.x = fmaf(w[4], t[4].x, acc.x); acc.y = fmaf(w[4], t[4].y, acc.y);   // unit self-loop last
            acc.z = fmaf(w[4], t[4].z, acc.z); acc.w = fmaf(w[4], t[4].w, acc.w);
            // the next batch's records are consumed HERE, before the stores: they arrived with the features, and a first use
            // after a store would wait for the store's acknowledgement as well (one counter, in order)
            asm volatile("" :: "v"(h0.x), "v"(h1.x), "v"(h2.x));
            if (live && store_ok) *reinterpret_cast<float4*>(out + (size_t)row * Fo + 4 * c) = acc;
        }
        if (sub < tcols && store_ok) {       // tail columns, one lane each: same entry order (head entries, CSR, self-loop)
            const int ib = tcol - F;         // indicator bit of this column (< 0: a column of X; >= num_ind: padding)
            const bool isx = ib < 0, isb = ib >= 0 && ib < num_ind;
            const int ibc = isb ? ib : 0;
            float acc = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = fmaf(w[u], isx ? tx[u] : (isb ? code_bit_f(cw[u], epoch, ibc) : 0.f), acc);
            if (len > 4) {
                const int beg = rowptr[row];
#pragma unroll 1
                for (int q = 4; q < len; ++q) {
                    const int s0 = csr[beg + q];
                    const float w0 = dinv[s0] * dc;
                    const int v0 = ids[s0];
                    const float f0 = isx ? X[(size_t)(uint32_t)v0 * (size_t)(uint32_t)ldx + tcol_x]
                                         : (isb ? code_bit_f(code[v0], epoch, ibc) : 0.f);
                    acc = fmaf(w0, f0, acc);
                }
            }
            acc = fmaf(w[4], isx ? tx[4] : (isb ? code_bit_f(cw[4], epoch, ibc) : 0.f), acc);
            out[(size_t)row * Fo + tcol] = acc;
        }
        if (stamp_it < 4) GRAPES_STAMP(stamp_it * 3 + 2);
        lds_barrier();
        if (stamp_it < 4) GRAPES_STAMP(stamp_it * 3 + 3);
        ++stamp_it;
        const int cur = slot;
        slot = slot == 2 ? 0 : slot + 1;
        if (threadIdx.x == 0) s_nhub[slot == 2 ? 0 : slot + 1] = 0;        // the slot of the batch after next
        // ---- hub rows of this batch of rows: the row groups take the eight strided chains of row_accumulate_hub between them
        // (group rg: chains rg, rg + RPB, ...), the chains are combined in chain order, then the self-loop — by the row's OWN
        // group: its earlier store to the same addresses (above) is then older in the same wavefront
        const int nh = s_nhub[cur];
        for (int i = 0; i < nh; ++i) {
            const int hrow = s_hub[cur][i];      // list order varies run to run; each row's result does not depend on it
            const int beg = rowptr[hrow], hlen = rowptr[hrow + 1] - beg;
            const float hdc = dinv[hrow];
            for (int cb = 0; cb < chunks; cb += LPR) {
                const int c = cb + sub;
                const bool live = c < chunks;
                for (int r = rg; r < 8; r += RPB) {
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (live) {
#pragma unroll 1
                        for (int q = r; q < hlen; q += 16) {          // two entries of the chain per round trip
                            const int q1 = q + 8 < hlen ? q + 8 : q;
                            const int s0 = csr[beg + q], s1 = csr[beg + q1];
                            const float w0 = dinv[s0] * hdc, w1 = dinv[s1] * hdc;
                            const int v0 = ids[s0], v1 = ids[s1];
                            float4 t0 = feat_chunk_load(X, ldx, v0, c), t1 = feat_chunk_load(X, ldx, v1, c);
                            uint32_t c0 = 0u, c1 = 0u;
                            if (num_ind > 0) { c0 = code[v0]; c1 = code[v1]; }
                            t0 = feat_chunk_fix(t0, c0, c, F, epoch); t1 = feat_chunk_fix(t1, c1, c, F, epoch);
                            acc.x = fmaf(w0, t0.x, acc.x); acc.y = fmaf(w0, t0.y, acc.y);
                            acc.z = fmaf(w0, t0.z, acc.z); acc.w = fmaf(w0, t0.w, acc.w);
                            if (q + 8 < hlen) {
                                acc.x = fmaf(w1, t1.x, acc.x); acc.y = fmaf(w1, t1.y, acc.y);
                                acc.z = fmaf(w1, t1.z, acc.z); acc.w = fmaf(w1, t1.w, acc.w);
                            }
                        }
                    }
                    s_part[r][sub] = acc;
                }
                __syncthreads();
                if (rg == hrow - base && live) {
                    float4 acc = s_part[0][sub];
#pragma unroll
                    for (int r = 1; r < 8; ++r) { const float4 p = s_part[r][sub]; acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w; }
                    const int vs = ids[hrow];
                    uint32_t cs = 0u;
                    if (num_ind > 0) cs = code[vs];
                    const float4 ts = feat_chunk_fix(feat_chunk_load(X, ldx, vs, c), cs, c, F, epoch);
                    const float wss = hdc * hdc;
                    acc.x = fmaf(wss, ts.x, acc.x); acc.y = fmaf(wss, ts.y, acc.y);
                    acc.z = fmaf(wss, ts.z, acc.z); acc.w = fmaf(wss, ts.w, acc.w);
                    *reinterpret_cast<float4*>(out + (long long)hrow * Fo + c * 4) = acc;
                }
                __syncthreads();
            }
        }
    }
    GRAPES_STAMP(14);
    grapes_clock_end(clk, clk0);
}

__device__ __forceinline__ void gr_fma4(float4& acc, float w, const float4& t) {
    acc.x = fmaf(w, t.x, acc.x); acc.y = fmaf(w, t.y, acc.y); acc.z = fmaf(w, t.z, acc.z); acc.w = fmaf(w, t.w, acc.w);
}

// ---- the production form of the head-record kernel.  What the measurements said (profiles/gather_bound_probe.py): a
// stripped gather over the same records — every lane of a 32-lane row group loads the 48-byte record itself, five feature
// rows, one FMA pass, one store, ~40 registers — runs the hop-2 shape in 7.6 us warm / 14 us cold, HALF the time of the
// earlier forms, and the number of row loads (two or five) or who loads the record hardly matters.  What cost the other
// half was everything around it: a workgroup barrier and LDS traffic per batch for hub rows, per-lane chunk correction for
// the indicator columns (more vector instructions than the product), 85-115 registers (5 wavefronts per SIMD instead of
// 7-8) — and the rows LONGER than the record: 0.4 % of a frontier's rows, each a chain of ~10-25 dependent memory round
// trips inside the loop, i.e. the tail of the launch (4.4 us of 14.8 warm, 7 of 24 cold).  So the launch has two roles:
//   * SHORT rows (all entries in the record, len <= 4) — workgroups NL.. : the stripped gather and nothing else.  Feature
//     chunks by the lanes below F/4; the columns after them (indicators, the last F % 4 features, padding) one lane each,
//     and only those lanes read code words.  Resident workgroups, no barrier, no LDS.
//   * LONG rows — workgroups 0..NL-1, from the first microsecond of the launch: each scans a slice of the records for
//     len > 4, and the whole workgroup does one such row at a time: its entries are fetched by the row groups in PARALLEL
//     (ids -> weights and feature-row ids -> chunks: three round trips for up to RPB entries), staged in LDS, and summed in
//     CSR order by one group (5..16 entries), or in the shared eight-chain order (hub rows).
// Summation order per output element is unchanged (CSR order or the hub order, then the self-loop).
#define GATHER_LONG_TILE 1024
// PEER: X is 1-D row-partitioned over the GPUs of the node and every shard is mapped into this process (hipIpc: the owner's
// HBM, reached over xGMI by ordinary loads) — row v lives in the shard q with bound[q] <= v < bound[q + 1], at
// vbase[q] + v * ldx (vbase[q] = the shard's base minus bound[q] rows, so the row arithmetic below is the single-GPU one).
// Picking the shard is a compare/select chain over at most 8 bounds in registers: no lookup load in front of the row load.
// The halo rows of a hop are then read where they live — no request/reply exchange, no collective, no staging copy.
struct PeerX { const float* vbase[GRAPES_MAX_PEER_SHARDS]; int32_t bound[GRAPES_MAX_PEER_SHARDS]; };
template <bool PEER>
__device__ __forceinline__ const float* peer_rows(const float* X, const PeerX& px, int v) {
    if (!PEER) return X;
    const float* b = px.vbase[0];
#pragma unroll
    for (int q = 1; q < GRAPES_MAX_PEER_SHARDS; ++q) b = v >= px.bound[q] ? px.vbase[q] : b;      // (unused shards: bound = INT_MAX)
    return b;
}
template <int LPR, bool PEER>
__device__ __forceinline__ void gcn_aggregate_gather_head5_body(const float* __restrict__ X, int F, int ldx,
                                                                    const int32_t* __restrict__ ids,
                                                                    const uint32_t* __restrict__ code, uint32_t epoch_host,
                                                                    const uint32_t* d_epoch, int num_ind,
                                                                    const int32_t* __restrict__ rowptr,
                                                                    const int32_t* __restrict__ csr,
                                                                    const float* __restrict__ dinv,
                                                                    const int4* __restrict__ head, float* __restrict__ out,
                                                                    int n_host, const int32_t* d_n, int NL, unsigned long long* clk,
                                                                    const PeerX& px, const int BID, const int NBLK) {
    const unsigned long long clk0 = grapes_clock_begin(clk);
    const int Fo = (F + num_ind + 3) & ~3;
    const int chunks = Fo >> 2;
    const int xch = F >> 2;                  // chunks wholly inside X: one lane each, four columns
    const int sub = threadIdx.x & (LPR - 1);
    constexpr int RPB = 256 / LPR;           // row groups per workgroup (8 or 4)
    const int rg = threadIdx.x / LPR;
    if (BID >= NL) {
        // ================= short rows
        const int tcols = Fo - 4 * xch;      // the columns after the whole chunks (<= 12): lanes 0..tcols-1 of the row group
        const unsigned ldx4 = (unsigned)ldx * 4u, fo4 = (unsigned)Fo * 4u;
        const bool istail = sub < tcols;
        const int tcol = 4 * xch + sub;      // this lane's tail column
        const bool tisx = istail && tcol < F;                 // ... a column of X; else indicator bit tcol - F (or padding)
        const int tib = tcol - F;
        const bool tisb = istail && tib >= 0 && tib < num_ind;
        const int tibc = tisb ? tib : 0;
        const int stride = (NBLK - NL) * RPB;
        int row = (BID - NL) * RPB + rg;
        // first batch: the records are requested inside the CAPACITY before the live count is known (a stale record is dropped)
        int4 h0 = make_int4(0, 0, 0, 0), h1 = h0, h2 = h0;
        if (row < n_host) { h0 = head[3 * (long long)row]; h1 = head[3 * (long long)row + 1]; h2 = head[3 * (long long)row + 2]; }
        const int n = eff_count(d_n, n_host);
        const uint32_t epoch = d_epoch ? (*d_epoch & 0xffffffu) : epoch_host;
        for (int base = (BID - NL) * RPB; base < n; base += stride) {       // uniform per workgroup
            if (row < n && h0.x <= 4) {
                const int g[5] = {h1.x, h1.z, h2.x, h2.z, h0.y};                     // four entries, then self
                const float w[5] = {__int_as_float(h1.y), __int_as_float(h1.w), __int_as_float(h2.y), __int_as_float(h2.w),
                                    __int_as_float(h0.z)};
                float tv[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
                if (tisx) {
#pragma unroll
                    for (int u = 0; u < 5; ++u) tv[u] = peer_rows<PEER>(X, px, g[u])[(size_t)(uint32_t)g[u] * (size_t)(uint32_t)ldx + tcol];
                }
                if (tisb) {
#pragma unroll
                    for (int u = 0; u < 5; ++u) tv[u] = code_bit_f(code[g[u]], epoch, tibc);
                }
                for (int cb = 0; cb < xch; cb += LPR) {
                    const int c = cb + sub;
                    if (c < xch) {
                        const char* xc = reinterpret_cast<const char*>(X + 4 * c);
                        float4 t[5];
#pragma unroll
                        for (int u = 0; u < 5; ++u) {
                            const char* xr = PEER ? reinterpret_cast<const char*>(peer_rows<PEER>(X, px, g[u]) + 4 * c) : xc;
                            t[u] = *reinterpret_cast<const float4*>(xr + (size_t)(uint32_t)g[u] * (size_t)ldx4);
                        }
                        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int u = 0; u < 5; ++u) gr_fma4(acc, w[u], t[u]);
                        *reinterpret_cast<float4*>(reinterpret_cast<char*>(out + 4 * c) + (size_t)(uint32_t)row * (size_t)fo4) = acc;
                    }
                }
                if (istail) {
                    float acc = 0.f;
#pragma unroll
                    for (int u = 0; u < 5; ++u) acc = fmaf(w[u], tv[u], acc);
                    out[(size_t)row * Fo + tcol] = acc;
                }
            }
            row += stride;
            if (row < n) { h0 = head[3 * (long long)row]; h1 = head[3 * (long long)row + 1]; h2 = head[3 * (long long)row + 2]; }
        }
        grapes_clock_end(clk, clk0);
        return;
    }
    // ================= long rows
    // A tile of records is scanned into TWO lists: rows of 5 .. GRAPES_HUB_ROW entries and hub rows.  The first kind goes one row
    // per ROW GROUP, RPB rows of the workgroup at a time, without a barrier or LDS: lane j of the group fetches entry j (column ->
    // weight and feature-row id -> code word: three round trips for the whole row), the lanes then walk the entries in CSR order,
    // four feature chunks in flight (more would cost the short rows' loop a wavefront per SIMD), operands by wavefront shuffles.  (Until the second session of round 5 such a row had the whole
    // workgroup — its groups fetched the entries in parallel, staged them in LDS, one group summed: two barriers per row and one
    // row at a time per workgroup.  Fine for the 0.4 % of a training frontier; a greedy evaluation frontier has a tenth of its
    // rows here and its gathers took 27 - 31 us.)  Hub rows keep the whole workgroup.  Same sums in the same order.
    __shared__ int s_long[GATHER_LONG_TILE];
    __shared__ int s_hubs[GATHER_LONG_TILE];
    __shared__ int s_nlong, s_nhubs;
    __shared__ float4 s_buf[8][LPR];         // the eight chain sums of a hub row
    const int n = eff_count(d_n, n_host);
    const uint32_t epoch = d_epoch ? (*d_epoch & 0xffffffu) : epoch_host;
    const int per = n > 0 ? (n + NL - 1) / NL : 1;           // rows per workgroup (scanned in tiles of GATHER_LONG_TILE)
    const int r_end = (BID + 1) * per < n ? (BID + 1) * per : n;
    for (int t0 = BID * per; t0 < r_end; t0 += GATHER_LONG_TILE) {       // uniform per workgroup
        if (threadIdx.x == 0) { s_nlong = 0; s_nhubs = 0; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < GATHER_LONG_TILE / 256; ++k) {
            const int r = t0 + k * 256 + (int)threadIdx.x;
            const int l = reinterpret_cast<const int32_t*>(head)[12 * (long long)(r < r_end ? r : r_end - 1)];
            if (r < r_end && l > 4) {
                if (l <= GRAPES_HUB_ROW) s_long[atomicAdd(&s_nlong, 1)] = r; else s_hubs[atomicAdd(&s_nhubs, 1)] = r;
            }
        }
        __syncthreads();
        const int nl = s_nlong, nh = s_nhubs;
        // ---- rows of 5 .. GRAPES_HUB_ROW entries: one per row group (list order varies run to run; a row's result does not depend on it)
        for (int i0 = 0; i0 < nl; i0 += RPB) {
            const int i = i0 + rg;
            if (i < nl) {
                const int hrow = s_long[i];
                const int hbeg = rowptr[hrow];
                int hlen = rowptr[hrow + 1] - hbeg;
                hlen = hlen < GRAPES_HUB_ROW ? hlen : GRAPES_HUB_ROW;        // (= the record's length; never more lanes than entries fetched)
                const float hdc = dinv[hrow];
                const int vs = ids[hrow];
                const int sj = csr[hbeg + (sub < hlen ? sub : hlen - 1)];     // lane j: entry j
                const float wj = dinv[sj] * hdc;
                const int vj = ids[sj];
                uint32_t cj = 0u, cs = 0u;
                if (num_ind > 0) { cj = code[vj]; cs = code[vs]; }
                for (int cb = 0; cb < chunks; cb += LPR) {
                    const int c = cb + sub;
                    const bool live = c < chunks;
                    const int cc = live ? c : chunks - 1;
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int q0 = 0; q0 < hlen; q0 += 4) {
                        float4 t[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int q = q0 + u < hlen ? q0 + u : hlen - 1;
                            const int vq = __shfl(vj, q, LPR);
                            t[u] = feat_chunk_load(peer_rows<PEER>(X, px, vq), ldx, vq, cc);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int q = q0 + u < hlen ? q0 + u : hlen - 1;
                            const float wq = __shfl(wj, q, LPR);
                            const uint32_t cq = (uint32_t)__shfl((int)cj, q, LPR);
                            if (q0 + u < hlen) gr_fma4(acc, wq, feat_chunk_fix(t[u], cq, cc, F, epoch));
                        }
                    }
                    gr_fma4(acc, hdc * hdc, feat_chunk_fix(feat_chunk_load(peer_rows<PEER>(X, px, vs), ldx, vs, cc), cs, cc, F, epoch));
                    if (live) *reinterpret_cast<float4*>(out + (long long)hrow * Fo + c * 4) = acc;
                }
            }
        }
        // ---- hub rows: the whole workgroup, one row at a time
        for (int i = 0; i < nh; ++i) {
            const int hrow = s_hubs[i];
            const int hbeg = rowptr[hrow], hlen = rowptr[hrow + 1] - hbeg;
            const float hdc = dinv[hrow];
            for (int cb = 0; cb < chunks; cb += LPR) {
                const int c = cb + sub;
                const bool live = c < chunks;
                // the row groups take the eight strided chains of row_accumulate_hub between them (group rg: chains
                // rg, rg + RPB, ...), the chains are combined in chain order, then the self-loop
                for (int r = rg; r < 8; r += RPB) {
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (live) {
#pragma unroll 1
                        for (int q = r; q < hlen; q += 16) {          // two entries of the chain per round trip
                            const int q1 = q + 8 < hlen ? q + 8 : q;
                            const int s0 = csr[hbeg + q], s1 = csr[hbeg + q1];
                            const float w0 = dinv[s0] * hdc, w1 = dinv[s1] * hdc;
                            const int v0 = ids[s0], v1 = ids[s1];
                            float4 u0 = feat_chunk_load(peer_rows<PEER>(X, px, v0), ldx, v0, c), u1 = feat_chunk_load(peer_rows<PEER>(X, px, v1), ldx, v1, c);
                            uint32_t c0 = 0u, c1 = 0u;
                            if (num_ind > 0) { c0 = code[v0]; c1 = code[v1]; }
                            u0 = feat_chunk_fix(u0, c0, c, F, epoch); u1 = feat_chunk_fix(u1, c1, c, F, epoch);
                            gr_fma4(acc, w0, u0);
                            if (q + 8 < hlen) gr_fma4(acc, w1, u1);
                        }
                    }
                    s_buf[r][sub] = acc;
                }
                __syncthreads();
                if (rg == 0 && live) {
                    float4 acc = s_buf[0][sub];
#pragma unroll
                    for (int r = 1; r < 8; ++r) { const float4 p = s_buf[r][sub]; acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w; }
                    const int vs = ids[hrow];
                    uint32_t cs = 0u;
                    if (num_ind > 0) cs = code[vs];
                    gr_fma4(acc, hdc * hdc, feat_chunk_fix(feat_chunk_load(peer_rows<PEER>(X, px, vs), ldx, vs, c), cs, c, F, epoch));
                    *reinterpret_cast<float4*>(out + (long long)hrow * Fo + c * 4) = acc;
                }
                __syncthreads();
            }
        }
        __syncthreads();
    }
    grapes_clock_end(clk, clk0);
}

template <int LPR, bool PEER = false>
__global__ __launch_bounds__(256) void gcn_aggregate_gather_head5_k(const float* __restrict__ X, int F, int ldx,
                                                                    const int32_t* __restrict__ ids,
                                                                    const uint32_t* __restrict__ code, uint32_t epoch_host,
                                                                    const uint32_t* d_epoch, int num_ind,
                                                                    const int32_t* __restrict__ rowptr,
                                                                    const int32_t* __restrict__ csr,
                                                                    const float* __restrict__ dinv,
                                                                    const int4* __restrict__ head, float* __restrict__ out,
                                                                    int n_host, const int32_t* d_n, int NL, unsigned long long* clk,
                                                                    PeerX px = PeerX()) {
    gcn_aggregate_gather_head5_body<LPR, PEER>(X, F, ldx, ids, code, epoch_host, d_epoch, num_ind, rowptr, csr, dinv, head, out, n_host,
                                               d_n, NL, clk, px, (int)blockIdx.x, (int)gridDim.x);
}
// two gather-SpMMs side by side in one launch (riders: common.h): workgroups [0, nA) work on `a`, the rest on `b`; no clock stamps
struct GatherHead5Args {
    const float* X; int F; int ldx; const int32_t* ids; const uint32_t* code; uint32_t epoch_host; const uint32_t* d_epoch; int num_ind;
    const int32_t* rowptr; const int32_t* csr; const float* dinv; const int4* head; float* out; int n_host; const int32_t* d_n; int NL;
};
struct GatherPeerArgs { GatherHead5Args a; PeerX px; };      // a recorded peer-form gather: the shard table rides behind the arguments
template <int LPR, bool PEER>
__global__ __launch_bounds__(256) void gcn_aggregate_gather_head5_pair_k(GatherHead5Args a, GatherHead5Args b, int nA, PeerX px) {
    const bool first = (int)blockIdx.x < nA;
    const GatherHead5Args& p = first ? a : b;
    gcn_aggregate_gather_head5_body<LPR, PEER>(p.X, p.F, p.ldx, p.ids, p.code, p.epoch_host, p.d_epoch, p.num_ind, p.rowptr, p.csr, p.dinv,
                                               p.head, p.out, p.n_host, p.d_n, p.NL, nullptr, px, first ? (int)blockIdx.x : (int)blockIdx.x - nA,
                                               first ? nA : (int)gridDim.x - nA);
}

// ---- cross-kernel riders (common.h): the classifier's forward launches have a handful of workgroups on an idle chip, and the
// next step's recorded prelude rides THERE (its own hop-1 launches compete with the rider for the same resource — atomics, the
// look-back — and the pair ran at nearly the sum of the two; these hosts are elsewhere-bound and nearly empty):
//   the classifier's gather-SpMM  (host) + the row order of the next step's hop-0 graph build (rider; prep_kernels.hip),
//   the classifier's 256-wide aggregation (host) + the next step's hop-0 gather-SpMM (rider).
// Needs index_ / prep_ / spmm_kernels.hip in ONE translation unit (hop_unity.hip).
#ifdef GRAPES_HOP_UNITY
template <int LPR, bool PEER>
__global__ __launch_bounds__(256) void gather_head5_sort_pair_k(GatherHead5Args a, SortRowsArgs b, int nA, PeerX px) {
    if ((int)blockIdx.x < nA)
        gcn_aggregate_gather_head5_body<LPR, PEER>(a.X, a.F, a.ldx, a.ids, a.code, a.epoch_host, a.d_epoch, a.num_ind, a.rowptr, a.csr, a.dinv,
                                                   a.head, a.out, a.n_host, a.d_n, a.NL, nullptr, px, (int)blockIdx.x, nA);
    else
        SORT_ROWS_CALL(b, (int)blockIdx.x - nA, (int)gridDim.x - nA);
}
#endif
struct AggregateArgs {
    const float* h; const int32_t* rowptr; const int32_t* csr; const float* dinv; const float* bias; float* out; int n_host;
    const int32_t* d_n; int F; int relu;
};
template <int LPR, bool PEER>
__global__ __launch_bounds__(256) void gcn_aggregate_gather_pair_k(AggregateArgs a, GatherHead5Args b, int nA, PeerX px) {
    if ((int)blockIdx.x < nA)
        gcn_aggregate_body<4, 0>(a.h, a.rowptr, a.csr, a.dinv, a.bias, a.out, a.n_host, a.d_n, a.F, a.relu, 0, nullptr, R1{nullptr, nullptr},
                                 nullptr, nullptr, (int)blockIdx.x, nA);
    else
        gcn_aggregate_gather_head5_body<LPR, PEER>(b.X, b.F, b.ldx, b.ids, b.code, b.epoch_host, b.d_epoch, b.num_ind, b.rowptr, b.csr, b.dinv,
                                                   b.head, b.out, b.n_host, b.d_n, b.NL, nullptr, px, (int)blockIdx.x - nA,
                                                   (int)gridDim.x - nA);
}

#ifdef GRAPES_HOP_UNITY
// ... or a recorded row order (transform-first nets have no gather-SpMM at hop 0: the prelude's last record is the graph build's sort)
__global__ __launch_bounds__(256) void gcn_aggregate_sort_pair_k(AggregateArgs a, SortRowsArgs b, int nA) {
    if ((int)blockIdx.x < nA)
        gcn_aggregate_body<4, 0>(a.h, a.rowptr, a.csr, a.dinv, a.bias, a.out, a.n_host, a.d_n, a.F, a.relu, 0, nullptr, R1{nullptr, nullptr},
                                 nullptr, nullptr, (int)blockIdx.x, nA);
    else
        SORT_ROWS_CALL(b, (int)blockIdx.x - nA, (int)gridDim.x - nA);
}
#endif

#ifdef GRAPES_DIAG
// ---- measurement only (profiles/gather_bound_probe.py): stripped-down gathers over the same head records, to price the
// ingredients of the production kernel one at a time.  NOT a product path: results are only correct for rows of <= 1 entry
// without indicator columns.  variant bit 0: five feature-row loads (else two: entry 0 and self); bit 1: resident
// workgroups that loop (records of the next batch prefetched, no barrier); bit 2: 64 lanes per row (else 32).
template <int LPR, int NLOAD, bool LOOP>
__global__ __launch_bounds__(256) void gather_probe_k(const float* __restrict__ X, int ldx, const int4* __restrict__ head,
                                                      float* __restrict__ out, int n, int Fo, unsigned long long* clk) {
    const unsigned long long clk0 = grapes_clock_begin(clk);
    constexpr int RPB = 256 / LPR;
    const int sub = threadIdx.x & (LPR - 1), rg = threadIdx.x / LPR;
    const int xch = ldx >> 2;
    const unsigned ldx4 = (unsigned)ldx * 4u;
    int row = blockIdx.x * RPB + rg;
    int4 h0 = make_int4(0, 0, 0, 0), h1 = h0, h2 = h0;
    if (row < n) { h0 = head[3 * (long long)row]; h1 = head[3 * (long long)row + 1]; h2 = head[3 * (long long)row + 2]; }
    for (;;) {
        const bool own = row < n;
        const int nrow = row + gridDim.x * RPB;
        int4 n0 = h0, n1 = h1, n2 = h2;
        if (LOOP && nrow < n) { n0 = head[3 * (long long)nrow]; n1 = head[3 * (long long)nrow + 1]; n2 = head[3 * (long long)nrow + 2]; }
        if (own && sub < xch) {
            const char* xc = reinterpret_cast<const char*>(X + 4 * sub);
            const int g[5] = {h1.x, h1.z, h2.x, h2.z, h0.y};
            const float w[5] = {__int_as_float(h1.y), __int_as_float(h1.w), __int_as_float(h2.y), __int_as_float(h2.w), __int_as_float(h0.z)};
            float4 t[5];
#pragma unroll
            for (int u = 0; u < 5; ++u)
                if (u == 4 || u < NLOAD - 1) t[u] = *reinterpret_cast<const float4*>(xc + (size_t)(uint32_t)g[u] * (size_t)ldx4);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 5; ++u)
                if (u == 4 || u < NLOAD - 1) gr_fma4(acc, w[u], t[u]);
            *reinterpret_cast<float4*>(out + (size_t)row * Fo + 4 * sub) = acc;
        }
        if (!LOOP) break;
        row = nrow; h0 = n0; h1 = n1; h2 = n2;
        if (__syncthreads_and(row >= n)) break;        // (uniform exit; a barrier per batch like the production kernel)
    }
    grapes_clock_end(clk, clk0);
}

extern "C" int grapes_debug_gather_probe(const float* X, int32_t ldx, const int32_t* row_head, float* out, int32_t n, int32_t Fo,
                                         int32_t variant, int32_t grid_cap, grapes_stream_t stream) {
    if (!X || !row_head || !out || n <= 0 || ldx <= 0 || ldx % 4 || Fo < ldx) return GRAPES_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int4* hd = (const int4*)row_head;
    const bool five = variant & 1, loop = variant & 2, wide = variant & 4;
    const int rpb = wide ? 4 : 8;
    int grid = grapes_div_up(n, rpb);
    if (loop && grid > grid_cap && grid_cap > 0) grid = grid_cap;
    unsigned long long* clk = grapes_clock_reserve("gather_probe_k", grid, 4);
#define GP_LAUNCH(L, N, LP) hipLaunchKernelGGL((gather_probe_k<L, N, LP>), dim3(grid), dim3(256), 0, s, X, ldx, hd, out, n, Fo, clk)
    if (!wide) { if (five) { if (loop) GP_LAUNCH(32, 5, true); else GP_LAUNCH(32, 5, false); } else { if (loop) GP_LAUNCH(32, 2, true); else GP_LAUNCH(32, 2, false); } }
    else       { if (five) { if (loop) GP_LAUNCH(64, 5, true); else GP_LAUNCH(64, 5, false); } else { if (loop) GP_LAUNCH(64, 2, true); else GP_LAUNCH(64, 2, false); } }
#undef GP_LAUNCH
    GRAPES_LAUNCH_CHECK();
    return 0;
}
#endif  // GRAPES_DIAG

GRAPES_STAMP_SETTER(grapes_stamp_set_spmm)
static int gather_fwd_impl(const float* X, int32_t F, int32_t x_stride, const int32_t* ids,
                           const uint32_t* ind_code, uint32_t epoch, const uint32_t* d_epoch,
                           int32_t num_ind, const int32_t* rowptr_t, const int32_t* csr_src,
                           const float* dinv, const int32_t* row_head, float* out, int32_t n,
                           const int32_t* d_n, grapes_stream_t stream, const PeerX* px) {
    if (x_stride <= 0) x_stride = F;
    if (n < 0 || F <= 0 || num_ind < 0 || num_ind > 8 || x_stride < F || x_stride % 4 != 0) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!X || !ids || !rowptr_t || !dinv || !out || (num_ind > 0 && !ind_code)) return GRAPES_EINVAL;
    if ((((uintptr_t)X) & 15) || (((uintptr_t)out) & 15)) return GRAPES_EALIGN;
    const int chunks = (F + num_ind + 3) / 4;
    const int ldx = x_stride;
    hipStream_t s = (hipStream_t)stream;
    if (row_head) {
        if ((((uintptr_t)row_head) & 15) || !csr_src) return GRAPES_EALIGN;
        const int4* hd = (const int4*)row_head;
        static int gcap = 0;
        // resident workgroups that loop (256 CUs x 5-6 per CU) rather than one per 8 rows of the CAPACITY: the loop carries the
        // next record fetch, and a launch sized by the capacity spends its tail dispatching workgroups that find no row
        if (!gcap) { const char* e = grapes_tune_env("GRAPES_GATHER_GRID"); gcap = e ? atoi(e) : 1536; if (gcap < 64) gcap = 1536; }
        // 32 lanes per row whenever the WHOLE chunks of X fit them (F <= 131): the columns after them — indicators, the last
        // F % 4 features, padding: at most 12 — are the tail lanes' scalar columns either way.  (F = 128 + 3 indicators is 33
        // chunks: by the chunk count it went to the 64-lane form with 31 of a row group's lanes idle — arxiv, papers100M.)
        static int narrow_rule = -1;
        if (narrow_rule < 0) { const char* e = grapes_tune_env("GRAPES_GATHER_NARROW_BY_CHUNKS"); narrow_rule = (e && atoi(e)) ? 1 : 0; }
        const bool narrow = narrow_rule ? chunks <= 32 : (F >> 2) <= 32;
        static int form = -1, nlong = 0;
        if (form < 0) {
            const char* e = grapes_tune_env("GRAPES_GATHER_FORM"); form = e ? atoi(e) : 5;      // 5 (default); 2: the earlier form
            const char* l = grapes_tune_env("GRAPES_GATHER_LONG_WGS"); nlong = l ? atoi(l) : 128; if (nlong < 1) nlong = 128;
        }
        if (px) {       // rows read from the peers' shards (always the production form)
            const int rpb = narrow ? 8 : 4;
            int grid = grapes_div_up(n, rpb); if (grid > gcap) grid = gcap;
            int NL = nlong; if (NL > grid) NL = grid;
            if (grapes_rider_recording()) { grid = grapes_rider_grid(grid) * 4; NL = grapes_rider_grid(NL) / 4; if (NL < 1) NL = 1; }
            grid += NL;
            const GatherHead5Args GA{X, F, ldx, ids, ind_code, epoch, d_epoch, num_ind, rowptr_t, csr_src, dinv, hd, out, n, d_n, NL};
            const PeerX pxv = *px;
            // (a rider of this launch — or this launch as a rider — reads through the SAME shard table: the variant carries its hash)
            uint32_t hv = 2166136261u;
            for (size_t q = 0; q < sizeof(PeerX); ++q) hv = (hv ^ reinterpret_cast<const unsigned char*>(&pxv)[q]) * 16777619u;
            const int variant = (int)((narrow ? 0x40000000u : 0x20000000u) | (hv & 0x0fffffffu));
            auto single = [=](hipStream_t s_) {
                if (narrow)
                    hipLaunchKernelGGL((gcn_aggregate_gather_head5_k<32, true>), dim3(grid), dim3(256), 0, s_, X, F, ldx, ids, ind_code, epoch, d_epoch,
                                       num_ind, rowptr_t, csr_src, dinv, hd, out, n, d_n, NL, grapes_clock_reserve("gcn_aggregate_gather_head5_k<32>", grid, 4), pxv);
                else
                    hipLaunchKernelGGL((gcn_aggregate_gather_head5_k<64, true>), dim3(grid), dim3(256), 0, s_, X, F, ldx, ids, ind_code, epoch, d_epoch,
                                       num_ind, rowptr_t, csr_src, dinv, hd, out, n, d_n, NL, grapes_clock_reserve("gcn_aggregate_gather_head5_k<64>", grid, 4), pxv);
            };
            if (grapes_rider_recording()) {
                // (the record carries the shard table behind the arguments: a host of another kernel launches the rider with it)
                const GatherPeerArgs GP{GA, pxv};
                grapes_rider_record(grapes_rider_make(GRAPES_RK_GATHER, variant, grid, 256, GP, single));
                return 0;
            }
#ifdef GRAPES_HOP_UNITY
            if (const GrapesRiderRecord* rs = grapes_clock_enabled() ? nullptr : grapes_rider_match(GRAPES_RK_SORT, 0, 256, s)) {
                SortRowsArgs Sq; memcpy(&Sq, rs->args, sizeof Sq);
                if (narrow) hipLaunchKernelGGL((gather_head5_sort_pair_k<32, true>), dim3(grid + rs->grid), dim3(256), 0, s, GA, Sq, grid, pxv);
                else hipLaunchKernelGGL((gather_head5_sort_pair_k<64, true>), dim3(grid + rs->grid), dim3(256), 0, s, GA, Sq, grid, pxv);
                GRAPES_LAUNCH_CHECK();
                return 0;
            }
#endif
            const GrapesRiderRecord* r = grapes_clock_enabled() ? nullptr : grapes_rider_match(GRAPES_RK_GATHER, variant, 256, s);
            if (r) {
                GatherHead5Args Bq; memcpy(&Bq, r->args, sizeof Bq);
                if (narrow) hipLaunchKernelGGL((gcn_aggregate_gather_head5_pair_k<32, true>), dim3(grid + r->grid), dim3(256), 0, s, GA, Bq, grid, pxv);
                else hipLaunchKernelGGL((gcn_aggregate_gather_head5_pair_k<64, true>), dim3(grid + r->grid), dim3(256), 0, s, GA, Bq, grid, pxv);
            } else {
                single(s);
            }
        } else if (form != 2) {
            // resident workgroups that loop over the short rows + `NL` workgroups for the long rows
            const int rpb = narrow ? 8 : 4;
            int grid = grapes_div_up(n, rpb); if (grid > gcap) grid = gcap;
            int NL = nlong; if (NL > grid) NL = grid;
            if (grapes_rider_recording()) { grid = grapes_rider_grid(grid) * 4; NL = grapes_rider_grid(NL) / 4; if (NL < 1) NL = 1; }
            grid += NL;
            const GatherHead5Args GA{X, F, ldx, ids, ind_code, epoch, d_epoch, num_ind, rowptr_t, csr_src, dinv, hd, out, n, d_n, NL};
            const int variant = narrow ? 1 : 2;
            auto single = [=](hipStream_t s_) {
                if (narrow)
                    hipLaunchKernelGGL((gcn_aggregate_gather_head5_k<32>), dim3(grid), dim3(256), 0, s_, X, F, ldx, ids, ind_code, epoch, d_epoch,
                                       num_ind, rowptr_t, csr_src, dinv, hd, out, n, d_n, NL, grapes_clock_reserve("gcn_aggregate_gather_head5_k<32>", grid, 4));
                else
                    hipLaunchKernelGGL((gcn_aggregate_gather_head5_k<64>), dim3(grid), dim3(256), 0, s_, X, F, ldx, ids, ind_code, epoch, d_epoch,
                                       num_ind, rowptr_t, csr_src, dinv, hd, out, n, d_n, NL, grapes_clock_reserve("gcn_aggregate_gather_head5_k<64>", grid, 4));
            };
            if (grapes_rider_recording()) { grapes_rider_record(grapes_rider_make(GRAPES_RK_GATHER, variant, grid, 256, GA, single)); return 0; }
            // (with the kernel clock table enabled every launch keeps its own stamps: nothing rides)
#ifdef GRAPES_HOP_UNITY
            if (const GrapesRiderRecord* rs = grapes_clock_enabled() ? nullptr : grapes_rider_match(GRAPES_RK_SORT, 0, 256, s)) {
                SortRowsArgs Sq; memcpy(&Sq, rs->args, sizeof Sq);
                if (narrow) hipLaunchKernelGGL((gather_head5_sort_pair_k<32, false>), dim3(grid + rs->grid), dim3(256), 0, s, GA, Sq, grid, PeerX());
                else hipLaunchKernelGGL((gather_head5_sort_pair_k<64, false>), dim3(grid + rs->grid), dim3(256), 0, s, GA, Sq, grid, PeerX());
                GRAPES_LAUNCH_CHECK();
                return 0;
            }
#endif
            const GrapesRiderRecord* r = grapes_clock_enabled() ? nullptr : grapes_rider_match(GRAPES_RK_GATHER, variant, 256, s);
            if (r) {
                GatherHead5Args Bq; memcpy(&Bq, r->args, sizeof Bq);
                if (narrow) hipLaunchKernelGGL((gcn_aggregate_gather_head5_pair_k<32, false>), dim3(grid + r->grid), dim3(256), 0, s, GA, Bq, grid, PeerX());
                else hipLaunchKernelGGL((gcn_aggregate_gather_head5_pair_k<64, false>), dim3(grid + r->grid), dim3(256), 0, s, GA, Bq, grid, PeerX());
            } else {
                single(s);
            }
        } else if (chunks <= 32) {
            int grid = grapes_div_up(n, 8); if (grid > gcap) grid = gcap;
            hipLaunchKernelGGL((gcn_aggregate_gather_head_k<32>), dim3(grid), dim3(256), 0, s, X, F, ldx, ids, ind_code, epoch, d_epoch,
                               num_ind, rowptr_t, csr_src, dinv, hd, out, n, d_n, grapes_clock_reserve("gcn_aggregate_gather_head_k<32>", grid, 4));
        } else {
            int grid = grapes_div_up(n, 4); if (grid > 2048) grid = 2048;
            hipLaunchKernelGGL((gcn_aggregate_gather_head_k<64>), dim3(grid), dim3(256), 0, s, X, F, ldx, ids, ind_code, epoch, d_epoch,
                               num_ind, rowptr_t, csr_src, dinv, hd, out, n, d_n, grapes_clock_reserve("gcn_aggregate_gather_head_k<64>", grid, 4));
        }
        GRAPES_LAUNCH_CHECK();
        return 0;
    }
    if (px) return GRAPES_EINVAL;            // (peer shards: head records only)
    if (chunks <= 32) {
        int grid = grapes_div_up(n, 8); if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL((gcn_aggregate_gather_k<32>), dim3(grid), dim3(256), 0, s, X, F, ldx, ids, ind_code, epoch, d_epoch, num_ind,
                           rowptr_t, csr_src, dinv, out, n, d_n, grapes_clock_reserve("gcn_aggregate_gather_k<32>", grid, 4));
    } else {
        int grid = grapes_div_up(n, 4); if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL((gcn_aggregate_gather_k<64>), dim3(grid), dim3(256), 0, s, X, F, ldx, ids, ind_code, epoch, d_epoch, num_ind,
                           rowptr_t, csr_src, dinv, out, n, d_n, grapes_clock_reserve("gcn_aggregate_gather_k<64>", grid, 4));
    }
    GRAPES_LAUNCH_CHECK();
    return 0;
}
extern "C" int grapes_gcn_aggregate_gather_fwd(const float* X, int32_t F, int32_t x_stride, const int32_t* ids,
                                               const uint32_t* ind_code, uint32_t epoch, const uint32_t* d_epoch,
                                               int32_t num_ind, const int32_t* rowptr_t, const int32_t* csr_src,
                                               const float* dinv, const int32_t* row_head, float* out, int32_t n,
                                               const int32_t* d_n, grapes_stream_t stream) {
    return gather_fwd_impl(X, F, x_stride, ids, ind_code, epoch, d_epoch, num_ind, rowptr_t, csr_src, dinv, row_head, out, n, d_n,
                           stream, nullptr);
}
extern "C" int grapes_gcn_aggregate_gather_fwd_peers(const float* const* shard_base, const int32_t* shard_bounds, int32_t n_shards,
                                                     int32_t F, int32_t x_stride, const int32_t* ids,
                                                     const uint32_t* ind_code, uint32_t epoch, const uint32_t* d_epoch,
                                                     int32_t num_ind, const int32_t* rowptr_t, const int32_t* csr_src,
                                                     const float* dinv, const int32_t* row_head, float* out, int32_t n,
                                                     const int32_t* d_n, grapes_stream_t stream) {
    if (!shard_base || !shard_bounds || n_shards < 1 || n_shards > GRAPES_MAX_PEER_SHARDS || !row_head) return GRAPES_EINVAL;
    if (x_stride <= 0) x_stride = F;
    PeerX px;
    for (int q = 0; q < GRAPES_MAX_PEER_SHARDS; ++q) {
        const bool live = q < n_shards;
        if (live && (!shard_base[q] || (((uintptr_t)shard_base[q]) & 15) || shard_bounds[q] < 0 || shard_bounds[q + 1] < shard_bounds[q]))
            return GRAPES_EINVAL;
        // (pointer arithmetic on integers: the virtual base of a shard may lie below its allocation)
        px.vbase[q] = live ? (const float*)((uintptr_t)shard_base[q] - (uintptr_t)shard_bounds[q] * (uintptr_t)x_stride * sizeof(float)) : nullptr;
        px.bound[q] = live ? shard_bounds[q] : 0x7fffffff;
    }
    if (shard_bounds[0] != 0) return GRAPES_EINVAL;
    return gather_fwd_impl(shard_base[0], F, x_stride, ids, ind_code, epoch, d_epoch, num_ind, rowptr_t, csr_src, dinv, row_head, out, n,
                           d_n, stream, &px);
}

// one workgroup (4 wavefronts) per item = GRAPES_LONG_ROW consecutive entries of a long row
template <int VEC, int MODE = 0>
__device__ __forceinline__ void gcn_aggregate_chunks_body(const float* __restrict__ h, const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ csr, const float* __restrict__ dinv,
                                                          int F, const int32_t* __restrict__ items,
                                                          const int32_t* __restrict__ d_n_items, int item_cap,
                                                          float* __restrict__ partials, R1 r1, int BID, int NBLK) {
    __shared__ float part[4][64 * VEC];
    int n_items = *d_n_items; if (n_items > item_cap) n_items = item_cap;
    const int lane = lane_id(), wid = threadIdx.x >> 6;
    for (int it = BID; it < n_items; it += NBLK) {
        const int row = items[2 * it], chunk = items[2 * it + 1];
        const int rbeg = rowptr[row], rend = rowptr[row + 1];
        const int beg = rbeg + chunk * GRAPES_LONG_ROW;
        const int end = beg + GRAPES_LONG_ROW < rend ? beg + GRAPES_LONG_ROW : rend;
        const float dc = dinv[row];
        const int per = GRAPES_LONG_ROW / 4;
        const int wb = beg + wid * per;
        const int we = wb + per < end ? wb + per : end;
        for (int fbase = 0; fbase < F; fbase += 64 * VEC) {
            const int f0 = fbase + lane * VEC;
            float acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
            float w2v[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) w2v[v] = (R1_MODE(MODE) && f0 < F) ? r1.w2[f0 + v] : 0.f;
            if (f0 < F && wb < we) row_accumulate<VEC, 16, MODE>(h, csr, dinv, wb, we, dc, F, f0, acc, r1.dh2, &w2v);
#pragma unroll
            for (int v = 0; v < VEC; ++v) part[wid][lane * VEC + v] = acc[v];
            __syncthreads();
            if (wid == 0 && f0 < F) {
                float* o = partials + (long long)it * F + f0;
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    o[v] = ((part[0][lane * VEC + v] + part[1][lane * VEC + v]) + part[2][lane * VEC + v]) + part[3][lane * VEC + v];
            }
            __syncthreads();
        }
    }
}
template <int VEC, int MODE = 0>
__global__ __launch_bounds__(256) void gcn_aggregate_chunks_k(const float* __restrict__ h, const int32_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ csr, const float* __restrict__ dinv,
                                                              int F, const int32_t* __restrict__ items,
                                                              const int32_t* __restrict__ d_n_items, int item_cap,
                                                              float* __restrict__ partials, R1 r1 = R1{nullptr, nullptr}) {
    gcn_aggregate_chunks_body<VEC, MODE>(h, rowptr, csr, dinv, F, items, d_n_items, item_cap, partials, r1, (int)blockIdx.x, (int)gridDim.x);
}

// the item with chunk 0 leads its row: its nc items are contiguous and in chunk order
__device__ __forceinline__ void gcn_aggregate_combine_body(const float* __restrict__ h, const int32_t* __restrict__ rowptr,
                                                           const float* __restrict__ dinv, const float* __restrict__ bias,
                                                           float* __restrict__ out, int F, int relu,
                                                           const int32_t* __restrict__ items,
                                                           const int32_t* __restrict__ d_n_items, int item_cap,
                                                           const float* __restrict__ partials, int prescaled,
                                                           R1 r1, int h_is_bits, int BID, int NBLK) {
    __shared__ float part[4][256];
    int n_items = *d_n_items; if (n_items > item_cap) n_items = item_cap;
    for (int it = BID; it < n_items; it += NBLK) {
        if (items[2 * it + 1] != 0) continue;
        const int row = items[2 * it];
        const int len = rowptr[row + 1] - rowptr[row];
        int nc = (len + GRAPES_LONG_ROW - 1) / GRAPES_LONG_ROW;
        if (it + nc > n_items) nc = n_items - it;
        const float dc = dinv[row];
        // 4 wavefronts each sum a contiguous quarter of the chunks (4 loads in flight), the quarters
        // are added in a fixed order
        const int g = threadIdx.x >> 6, l = threadIdx.x & 63;
        const int per = (nc + 3) >> 2;
        const int c0 = g * per < nc ? g * per : nc;
        const int c1 = c0 + per < nc ? c0 + per : nc;
        for (int fbase = 0; fbase < F; fbase += 256) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int f = fbase + v * 64 + l;
                float acc = 0.f;
                if (f < F) {
                    int c = c0;
                    for (; c + 4 <= c1; c += 4) {
                        const float p0 = partials[(long long)(it + c) * F + f], p1 = partials[(long long)(it + c + 1) * F + f];
                        const float p2 = partials[(long long)(it + c + 2) * F + f], p3 = partials[(long long)(it + c + 3) * F + f];
                        acc += p0; acc += p1; acc += p2; acc += p3;
                    }
                    for (; c < c1; ++c) acc += partials[(long long)(it + c) * F + f];
                }
                part[g][v * 64 + l] = acc;
            }
            __syncthreads();
            if (g == 0) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int f = fbase + v * 64 + l;
                    if (f < F) {
                        const int q = v * 64 + l;
                        const float acc = ((part[0][q] + part[1][q]) + part[2][q]) + part[3][q];
                        float hv;
                        if (h_is_bits) hv = ((reinterpret_cast<const uint32_t*>(h)[8 * (long long)row + (f >> 5)] >> (f & 31)) & 1u) ? 1.f : 0.f;
                        else hv = h[(long long)row * F + f];
                        if (r1.dh2) hv = hv > 0.f ? r1.dh2[row] * r1.w2[f] : 0.f;      // rank-1 gated row (see R1)
                        float r = prescaled ? dc * (acc + hv) : fmaf(dc * dc, hv, acc);
                        if (bias) r += bias[f];
                        if (relu) r = fmaxf(r, 0.f);
                        out[(long long)row * F + f] = r;
                    }
                }
            }
            __syncthreads();
        }
    }
}
__global__ __launch_bounds__(256) void gcn_aggregate_combine_k(const float* __restrict__ h, const int32_t* __restrict__ rowptr,
                                                               const float* __restrict__ dinv, const float* __restrict__ bias,
                                                               float* __restrict__ out, int F, int relu,
                                                               const int32_t* __restrict__ items,
                                                               const int32_t* __restrict__ d_n_items, int item_cap,
                                                               const float* __restrict__ partials, int prescaled,
                                                               R1 r1 = R1{nullptr, nullptr}, int h_is_bits = 0) {
    gcn_aggregate_combine_body(h, rowptr, dinv, bias, out, F, relu, items, d_n_items, item_cap, partials, prescaled, r1, h_is_bits,
                               (int)blockIdx.x, (int)gridDim.x);
}

#include "narrow.h"

__global__ __launch_bounds__(256) void gcn_aggregate_narrow_k(const float* __restrict__ h, const int32_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ csr, const float* __restrict__ dinv,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int n_host, const int32_t* d_n, int F, int relu,
                                                              int narrow_lane_rows) {
    narrow_body(h, rowptr, csr, dinv, bias, out, n_host, d_n, F, relu, narrow_lane_rows, blockIdx.x, gridDim.x);
}
// Up to four independent graphs in one launch (blockIdx.y = graph): the backward aggregations of the hops' 1-wide heads.
struct NarrowSegs {
    int count;
    const float* h[4]; const int32_t* rowptr[4]; const int32_t* csr[4]; const float* dinv[4]; float* out[4];
    const int32_t* d_n[4]; int n_cap[4];
    const float* bias[4];            // per graph, may be NULL
};
__global__ __launch_bounds__(256) void gcn_aggregate_narrow_multi_k(NarrowSegs sg, int F, int narrow_lane_rows) {
    const int q = blockIdx.y;        // ternary chains, not indexed loads from the by-value struct (no scratch)
#define NSEL(f) (q == 0 ? sg.f[0] : (q == 1 ? sg.f[1] : (q == 2 ? sg.f[2] : sg.f[3])))
    narrow_body(NSEL(h), NSEL(rowptr), NSEL(csr), NSEL(dinv), NSEL(bias), NSEL(out), NSEL(n_cap), NSEL(d_n), F, 0,
                narrow_lane_rows, blockIdx.x, gridDim.x);
#undef NSEL
}

#ifndef GRAPES_ALIGNED16_DEFINED
#define GRAPES_ALIGNED16_DEFINED
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
#endif

static int narrow_lane_rows_cfg() {     // rows up to this length are walked by ONE lane (GRAPES_NARROW_LANE_ROWS)
    static int lane_rows = -1;
    if (lane_rows < 0) { const char* e = grapes_tune_env("GRAPES_NARROW_LANE_ROWS"); lane_rows = e ? atoi(e) : 8; if (lane_rows < 0) lane_rows = 0; }
    return lane_rows;
}

template <bool PRE, int WMODE = (PRE ? 1 : 0)>
static int launch_aggregate_t(const float* h, const int32_t* rowptr, const int32_t* csr, const float* dinv,
                              const float* bias, float* out, int n, const int32_t* d_n, int f, int relu,
                              const int32_t* items, const int32_t* d_n_items, int item_cap, float* partials,
                              hipStream_t s) {
    if (f <= 16) {
        if (PRE) return GRAPES_EINVAL;
        int grid = grapes_div_up(n, 256); if (grid > 4096) grid = 4096;
        const int lane_rows = narrow_lane_rows_cfg();
        hipLaunchKernelGGL(gcn_aggregate_narrow_k, dim3(grid), dim3(256), 0, s, h, rowptr, csr, dinv, bias, out, n, d_n, f, relu, lane_rows);
        GRAPES_LAUNCH_CHECK();
        return 0;
    }
    int grid = grapes_div_up(n, 4); if (grid > 16384) grid = 16384;
    const bool vec = (f % 4 == 0) && aligned16(h) && aligned16(out) && (!bias || aligned16(bias)) && (!partials || aligned16(partials));
    if (PRE && !vec) return GRAPES_EALIGN;
    const int skip = (items && d_n_items && partials && item_cap > 0) ? 1 : 0;
    // item-scheduled (full-graph) aggregation of rows narrower than 1 KiB: lanes in slots (gcn_aggregate_lpr_k)
    const int lpr = (skip && vec && f <= 128) ? (f <= 32 ? 8 : (f <= 64 ? 16 : 32)) : 0;
    if (lpr == 8)
        hipLaunchKernelGGL((gcn_aggregate_lpr_k<8, PRE>), dim3(grid), dim3(256), 0, s, h, rowptr, csr, dinv, bias, out, n, d_n, f, relu, skip, (unsigned long long*)nullptr);
    else if (lpr == 16)
        hipLaunchKernelGGL((gcn_aggregate_lpr_k<16, PRE>), dim3(grid), dim3(256), 0, s, h, rowptr, csr, dinv, bias, out, n, d_n, f, relu, skip, (unsigned long long*)nullptr);
    else if (lpr == 32)
        hipLaunchKernelGGL((gcn_aggregate_lpr_k<32, PRE>), dim3(grid), dim3(256), 0, s, h, rowptr, csr, dinv, bias, out, n, d_n, f, relu, skip,
                           f >= 64 ? grapes_clock_reserve("gcn_aggregate_lpr_k<32>", grid, 4) : nullptr);
    else if (vec) {
        const GrapesRiderRecord* r = nullptr;
        if (!PRE && !skip && !grapes_clock_enabled())         // a small graph's aggregation may carry a recorded gather-SpMM (riders)
            r = grapes_rider_match(GRAPES_RK_GATHER, GRAPES_RIDER_ANY_VARIANT, 256, s);
#ifdef GRAPES_HOP_UNITY
        static int sort_host = -1;          // A/B (diagnostic build): GRAPES_AGG_SORT_HOST=0
        if (sort_host < 0) { const char* e = grapes_tune_env("GRAPES_AGG_SORT_HOST"); sort_host = e ? atoi(e) : 1; }
        const GrapesRiderRecord* rs = (sort_host && !r && !PRE && !skip && !grapes_clock_enabled()) ? grapes_rider_match(GRAPES_RK_SORT, 0, 256, s) : nullptr;
        if (rs) {
            SortRowsArgs Sq; memcpy(&Sq, rs->args, sizeof Sq);
            const AggregateArgs A{h, rowptr, csr, dinv, bias, out, n, d_n, f, relu};
            hipLaunchKernelGGL(gcn_aggregate_sort_pair_k, dim3(grid + rs->grid), dim3(256), 0, s, A, Sq, grid);
        } else
#endif
        if (r) {
            // variant: 1 / 2 = 32 / 64 lanes per row over the local matrix; 0x4... / 0x2... = the same through a shard table,
            // which then follows the arguments in the record
            GatherHead5Args Bq; memcpy(&Bq, r->args, sizeof Bq);
            PeerX pq = PeerX();
            const bool peer = (r->variant & 0x60000000) != 0;
            const bool narrow = peer ? (r->variant & 0x40000000) != 0 : r->variant == 1;
            if (peer) { GatherPeerArgs gp; memcpy(&gp, r->args, sizeof gp); pq = gp.px; }
            const AggregateArgs A{h, rowptr, csr, dinv, bias, out, n, d_n, f, relu};
            if (peer && narrow) hipLaunchKernelGGL((gcn_aggregate_gather_pair_k<32, true>), dim3(grid + r->grid), dim3(256), 0, s, A, Bq, grid, pq);
            else if (peer) hipLaunchKernelGGL((gcn_aggregate_gather_pair_k<64, true>), dim3(grid + r->grid), dim3(256), 0, s, A, Bq, grid, pq);
            else if (narrow) hipLaunchKernelGGL((gcn_aggregate_gather_pair_k<32, false>), dim3(grid + r->grid), dim3(256), 0, s, A, Bq, grid, pq);
            else hipLaunchKernelGGL((gcn_aggregate_gather_pair_k<64, false>), dim3(grid + r->grid), dim3(256), 0, s, A, Bq, grid, pq);
        } else {
            hipLaunchKernelGGL((gcn_aggregate_k<4, WMODE>), dim3(grid), dim3(256), 0, s, h, rowptr, csr, dinv, bias, out, n, d_n, f, relu, skip,
                               f >= 64 ? grapes_clock_reserve("gcn_aggregate_k<4>", grid, 4) : nullptr);
        }
    }
    else
        hipLaunchKernelGGL((gcn_aggregate_k<1, 0>), dim3(grid), dim3(256), 0, s, h, rowptr, csr, dinv, bias, out, n, d_n, f, relu, skip,
                           (unsigned long long*)nullptr);
    GRAPES_LAUNCH_CHECK();
    if (skip) {
        int g2 = item_cap < 2048 ? item_cap : 2048;
        if (lpr == 8)
            hipLaunchKernelGGL((gcn_aggregate_chunks_lpr_k<8, PRE>), dim3(g2), dim3(256), 0, s, h, rowptr, csr, dinv, f, items, d_n_items, item_cap, partials);
        else if (lpr == 16)
            hipLaunchKernelGGL((gcn_aggregate_chunks_lpr_k<16, PRE>), dim3(g2), dim3(256), 0, s, h, rowptr, csr, dinv, f, items, d_n_items, item_cap, partials);
        else if (lpr == 32)
            hipLaunchKernelGGL((gcn_aggregate_chunks_lpr_k<32, PRE>), dim3(g2), dim3(256), 0, s, h, rowptr, csr, dinv, f, items, d_n_items, item_cap, partials);
        else if (vec)
            hipLaunchKernelGGL((gcn_aggregate_chunks_k<4, WMODE>), dim3(g2), dim3(256), 0, s, h, rowptr, csr, dinv, f, items, d_n_items, item_cap, partials);
        else
            hipLaunchKernelGGL((gcn_aggregate_chunks_k<1, 0>), dim3(g2), dim3(256), 0, s, h, rowptr, csr, dinv, f, items, d_n_items, item_cap, partials);
        GRAPES_LAUNCH_CHECK();
        hipLaunchKernelGGL(gcn_aggregate_combine_k, dim3(g2), dim3(256), 0, s, h, rowptr, dinv, bias, out, f, relu, items, d_n_items,
                           item_cap, (const float*)partials, PRE ? 1 : 0);
        GRAPES_LAUNCH_CHECK();
    }
    return 0;
}
static int launch_aggregate(const float* h, const int32_t* rowptr, const int32_t* csr, const float* dinv,
                            const float* bias, float* out, int n, const int32_t* d_n, int f, int relu,
                            const int32_t* items, const int32_t* d_n_items, int item_cap, float* partials,
                            hipStream_t s, bool prescaled = false) {
    return prescaled ? launch_aggregate_t<true>(h, rowptr, csr, dinv, bias, out, n, d_n, f, relu, items, d_n_items, item_cap, partials, s)
                     : launch_aggregate_t<false>(h, rowptr, csr, dinv, bias, out, n, d_n, f, relu, items, d_n_items, item_cap, partials, s);
}

// hs[r, :] = dinv[r] * h[r, :]  (in place when hs == h): the pre-scaled operand of grapes_gcn_aggregate_fwd_prescaled
__global__ __launch_bounds__(256) void scale_rows_k(const float4* __restrict__ h, const float* __restrict__ dinv, float4* __restrict__ hs,
                                                    long long n, int f4) {
    const long long total = n * f4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const float d = dinv[i / f4];
        float4 v = h[i];
        v.x *= d; v.y *= d; v.z *= d; v.w *= d;
        hs[i] = v;
    }
}
extern "C" int grapes_scale_rows(const float* h, const float* dinv, float* hs, int64_t n, int32_t f, grapes_stream_t stream) {
    if (n < 0 || f <= 0 || (f & 3)) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!h || !dinv || !hs) return GRAPES_EINVAL;
    if (!aligned16(h) || !aligned16(hs)) return GRAPES_EALIGN;
    long long blocks = (n * (f >> 2) + 255) / 256; if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(scale_rows_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)h, dinv, (float4*)hs,
                       (long long)n, f >> 2);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// Two 1-wide vectors over the SAME graph in one launch: the sampler net's and the log-Z net's heads at hop 0
// (out_a = Â h_a + bias_a, out_b = Â h_b + bias_b; modules/gcn.py:36 on the [H, 1] layers of main.py:210 and main.py:227).
extern "C" int grapes_gcn_aggregate_narrow_pair(const float* h_a, const float* h_b, const int32_t* rowptr_t, const int32_t* csr_src,
                                                const float* dinv, const float* bias_a, const float* bias_b, float* out_a,
                                                float* out_b, int32_t n, const int32_t* d_n, grapes_stream_t stream) {
    if (n < 0 || !h_a || !h_b || !rowptr_t || !dinv || !out_a || !out_b) return GRAPES_EINVAL;
    if (n == 0) return 0;
    NarrowSegs ns; ns.count = 2;
    for (int q = 0; q < 4; ++q) {
        const bool b = (q & 1) != 0;
        ns.h[q] = b ? h_b : h_a; ns.rowptr[q] = rowptr_t; ns.csr[q] = csr_src; ns.dinv[q] = dinv; ns.out[q] = b ? out_b : out_a;
        ns.d_n[q] = d_n; ns.n_cap[q] = q < 2 ? n : 0; ns.bias[q] = b ? bias_b : bias_a;
    }
    int g2 = grapes_div_up(n, 256); if (g2 > 4096) g2 = 4096;
    hipLaunchKernelGGL(gcn_aggregate_narrow_multi_k, dim3(g2, 2), dim3(256), 0, (hipStream_t)stream, ns, 1, narrow_lane_rows_cfg());
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// ============================================================================ column sums
// out[c] (+)= sum_r wrow[r] * val(r,c);  val = src[r][c], optionally gated by (gate[r][c] > 0) (ReLU
// backward); the gated values are optionally written to dst (dpre).  Stage 1: CS_BLOCKS workgroups,
// each owning a fixed, strided set of 32-row chunks; stage 2: fixed-order combine.  Deterministic.
#define CS_BLOCKS 512
#define CS_ROWS 32
__global__ __launch_bounds__(256) void colsum_partial_k(const float* __restrict__ src, const float* __restrict__ gate,
                                                        const float* __restrict__ wrow, float* __restrict__ dst,
                                                        float* __restrict__ partial, int n_host, const int32_t* d_n,
                                                        int F) {
    const int n = eff_count(d_n, n_host);
    for (int c = threadIdx.x; c < F; c += blockDim.x) {
        float acc = 0.f;
        for (int r0 = blockIdx.x * CS_ROWS; r0 < n; r0 += CS_BLOCKS * CS_ROWS) {
            const int r1 = r0 + CS_ROWS < n ? r0 + CS_ROWS : n;
            int r = r0;
            for (; r + 8 <= r1; r += 8) {      // eight rows (sixteen 4-byte loads with a gate) in flight; additions in row order
                float v[8], g[8], w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const long long o = (long long)(r + u) * F + c;
                    v[u] = src[o];
                    g[u] = gate ? gate[o] : 1.f;
                    w[u] = wrow ? wrow[r + u] : 1.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float x = g[u] > 0.f ? v[u] : 0.f;
                    if (dst) dst[(long long)(r + u) * F + c] = x;
                    acc += wrow ? w[u] * x : x;
                }
            }
            for (; r + 4 <= r1; r += 4) {      // four rows in flight
                float v[4], g[4], w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long long o = (long long)(r + u) * F + c;
                    v[u] = src[o];
                    g[u] = gate ? gate[o] : 1.f;
                    w[u] = wrow ? wrow[r + u] : 1.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float x = g[u] > 0.f ? v[u] : 0.f;
                    if (dst) dst[(long long)(r + u) * F + c] = x;
                    acc += wrow ? w[u] * x : x;
                }
            }
            for (; r < r1; ++r) {
                const long long o = (long long)r * F + c;
                float x = src[o];
                if (gate) x = gate[o] > 0.f ? x : 0.f;
                if (dst) dst[o] = x;
                acc += wrow ? wrow[r] * x : x;
            }
        }
        partial[(long long)blockIdx.x * F + c] = acc;
    }
}

// narrow F (<= 16): threads run across rows, block tree-reduce per column
__global__ __launch_bounds__(256) void colsum_partial_narrow_k(const float* __restrict__ src, const float* __restrict__ gate,
                                                               const float* __restrict__ wrow, float* __restrict__ dst,
                                                               float* __restrict__ partial, int n_host,
                                                               const int32_t* d_n, int F) {
    __shared__ float red[4];
    const int n = eff_count(d_n, n_host);
    for (int c = 0; c < F; ++c) {
        float acc = 0.f;
        for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += CS_BLOCKS * blockDim.x) {
            const long long o = (long long)r * F + c;
            float v = src[o];
            if (gate) v = gate[o] > 0.f ? v : 0.f;
            if (dst) dst[o] = v;
            acc += wrow ? wrow[r] * v : v;
        }
        acc = wave_sum(acc);
        if (lane_id() == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) partial[(long long)blockIdx.x * F + c] = (red[0] + red[1]) + (red[2] + red[3]);
        __syncthreads();
    }
}

// One thread's share of a final sum: the CS_BLOCKS / 4 partials of range g of column c, added in ascending block order — with the
// loads of 64 blocks in flight at a time (the additions keep their order: same bits).  A loop that waits for four loads per trip
// is 32 dependent round trips: colsum_final2_multi_k took 70 us of Reddit's step that way, for 6 MB.
__device__ __forceinline__ float colsum_range_sum(const float* __restrict__ partial, int b0, int F, int c) {
    constexpr int CS_FINAL_BATCH = 64;
    static_assert((CS_BLOCKS / 4) % CS_FINAL_BATCH == 0, "the final sums read whole batches");
    float acc = 0.f;
#pragma unroll 1
    for (int bb = b0; bb < b0 + CS_BLOCKS / 4; bb += CS_FINAL_BATCH) {
        float v[CS_FINAL_BATCH];
#pragma unroll
        for (int u = 0; u < CS_FINAL_BATCH; ++u) v[u] = partial[(long long)(bb + u) * F + c];
#pragma unroll
        for (int u = 0; u < CS_FINAL_BATCH; ++u) acc += v[u];
    }
    return acc;
}
// 64 columns x 4 partial ranges per workgroup
__global__ __launch_bounds__(256) void colsum_final_k(const float* __restrict__ partial, float* __restrict__ out, int F,
                                                      int accumulate) {
    __shared__ float part[4][64];
    const int g = threadIdx.x >> 6, cl = threadIdx.x & 63;
    const int c = blockIdx.x * 64 + cl;
    float acc = 0.f;
    if (c < F) acc = colsum_range_sum(partial, g * (CS_BLOCKS / 4), F, c);
    part[g][cl] = acc;
    __syncthreads();
    if (g == 0 && c < F) {
        const float t = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
        out[c] = accumulate ? out[c] + t : t;
    }
}

// Few rows (the classifier's sampled subgraph): ONE launch.  Workgroup (cb, rb) sums 64 columns over the rb-th of R row
// ranges (thread (g, cl): rows lo + g, lo + g + 4, ...; the four row lanes combined in a fixed order), publishes its 64
// partials (device-scope exchange) and takes a ticket of its column block; the last of the R workgroups adds the
// partials in range order.  Deterministic.
#define CS_SMALL_ROWS 8192
#define CS_SMALL_R 16
__device__ __forceinline__ void colsum_ticket_body(const float* __restrict__ src, const float* __restrict__ gate,
                                                       const float* __restrict__ wrow, float* __restrict__ dst,
                                                       float* __restrict__ partial, float* __restrict__ out, int n_host,
                                                       const int32_t* d_n, int F, int accumulate, unsigned* __restrict__ ticket,
                                                       int cbx, int rb, int R) {
    __shared__ float part[4][64];
    __shared__ int s_last;
    const int n = eff_count(d_n, n_host);
    const int g = threadIdx.x >> 6, cl = threadIdx.x & 63;
    const int c = cbx * 64 + cl;
    const int per = ((n + R - 1) / R + 3) & ~3;
    const int lo = rb * per, hi = lo + per < n ? lo + per : n;
    float acc = 0.f;
    if (c < F) {
        // (no `gate ? … : …` / `wrow ? … : …` inside the load loops: an absent operand is read from `src` instead and ignored by a
        // select — with the uniform conditions between the loads hipcc waited for every row's pair before it requested the next
        // one's: "eight rows in flight" were eight dependent round trips, 60 per thread on a classifier layer)
        const bool has_g = gate != nullptr, has_w = wrow != nullptr;
        const float* __restrict__ gp = has_g ? gate : src;
        const float* __restrict__ wp = has_w ? wrow : src;
        int r = lo + g;
        for (; r + 28 < hi; r += 32) {        // eight rows in flight
            float v[8], gt[8], w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long long o = (long long)(r + 4 * u) * F + c;
                v[u] = src[o]; gt[u] = gp[o]; w[u] = wp[has_w ? (long long)(r + 4 * u) : o];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float x = (!has_g || gt[u] > 0.f) ? v[u] : 0.f;
                if (dst) dst[(long long)(r + 4 * u) * F + c] = x;
                acc += has_w ? w[u] * x : x;
            }
        }
        for (; r + 12 < hi; r += 16) {        // four rows in flight
            float v[4], gt[4], w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long o = (long long)(r + 4 * u) * F + c;
                v[u] = src[o]; gt[u] = gp[o]; w[u] = wp[has_w ? (long long)(r + 4 * u) : o];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float x = (!has_g || gt[u] > 0.f) ? v[u] : 0.f;
                if (dst) dst[(long long)(r + 4 * u) * F + c] = x;
                acc += has_w ? w[u] * x : x;
            }
        }
        for (; r < hi; r += 4) {
            const long long o = (long long)r * F + c;
            float x = src[o];
            if (gate) x = gate[o] > 0.f ? x : 0.f;
            if (dst) dst[o] = x;
            acc += wrow ? wrow[r] * x : x;
        }
    }
    part[g][cl] = acc;
    __syncthreads();
    if (!out) return;
    if (g == 0 && c < F) publish_f32(&partial[(long long)rb * F + c], (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]));
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(&ticket[cbx], 1u) == (unsigned)R - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    if (g == 0 && c < F) {
        float t = 0.f;
        for (int b = 0; b < R; ++b)
            t += __int_as_float(__hip_atomic_load((const int*)(partial + (long long)b * F + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        out[c] = accumulate ? out[c] + t : t;
    }
    if (threadIdx.x == 0) ticket[cbx] = 0u;
}

__global__ __launch_bounds__(256) void colsum_ticket_k(const float* __restrict__ src, const float* __restrict__ gate,
                                                       const float* __restrict__ wrow, float* __restrict__ dst,
                                                       float* __restrict__ partial, float* __restrict__ out, int n_host,
                                                       const int32_t* d_n, int F, int accumulate, unsigned* __restrict__ ticket) {
    colsum_ticket_body(src, gate, wrow, dst, partial, out, n_host, d_n, F, accumulate, ticket, blockIdx.x, blockIdx.y, gridDim.y);
}

// Backward aggregation of a FEW-row layer in one launch:  dh = Â^T (dout (.) [relu_out > 0]),  dbias (+)= column sums of the
// gated dout.  Two independent jobs share the launch: the first `ncs` workgroups are colsum_ticket_k's (column block x row
// range, last workgroup of a column block combines), the others aggregate — one wavefront per row, the ReLU mask applied
// to the gathered rows on the fly (dpre is never written), the additions of a row in CSR order as in gcn_aggregate_k.
struct AggBwdSmallArgs {
    const float* dout; const float* gate; const int32_t* rowptr; const int32_t* csr; const float* dinv; float* dh;
    int n_host; const int32_t* d_n; int F; float* dbias; int accumulate_bias; float* partials; unsigned* ticket; int ncb; int R;
};
// (BID, NBLK: this problem's workgroup index and count — the launch's own, or its share of a pair launch)
template <int VEC>
__device__ __forceinline__ void gcn_aggregate_bwd_small_body(const AggBwdSmallArgs& a, int BID, int NBLK) {
    const float* __restrict__ dout = a.dout; const float* __restrict__ gate = a.gate;
    const int32_t* __restrict__ rowptr = a.rowptr; const int32_t* __restrict__ csr = a.csr; const float* __restrict__ dinv = a.dinv;
    float* __restrict__ dh = a.dh;
    const int F = a.F;
    const int ncs = a.dbias ? a.ncb * a.R : 0;
    if (BID < ncs) {
        colsum_ticket_body(dout, gate, nullptr, nullptr, a.partials, a.dbias, a.n_host, a.d_n, F, a.accumulate_bias, a.ticket,
                           BID % a.ncb, BID / a.ncb, a.R);
        return;
    }
    const int n = eff_count(a.d_n, a.n_host);
    const int lane = lane_id();
    const int wave_global = __builtin_amdgcn_readfirstlane(((BID - ncs) * 256 + (int)threadIdx.x) >> 6);
    const int nwaves = ((NBLK - ncs) * 256) >> 6;
    const int f0 = lane * VEC;
    if (f0 >= F) return;
    auto gated = [&](int r, float (&val)[VEC]) {
        ld_vec<VEC>(dout + (long long)r * F + f0, val);
        if (gate) {
            float g[VEC];
            ld_vec<VEC>(gate + (long long)r * F + f0, g);
#pragma unroll
            for (int v = 0; v < VEC; ++v) val[v] = g[v] > 0.f ? val[v] : 0.f;
        }
    };
    for (int row = wave_global; row < n; row += nwaves) {
        const int beg = rowptr[row], end = rowptr[row + 1];
        const float dc = dinv[row];
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        int j = beg;
        for (; j + 4 <= end; j += 4) {                 // four gathered rows in flight, added in CSR order
            int s4[4]; float w4[4]; float val[4][VEC];
#pragma unroll
            for (int u = 0; u < 4; ++u) { s4[u] = csr[j + u]; w4[u] = dinv[s4[u]] * dc; }
#pragma unroll
            for (int u = 0; u < 4; ++u) gated(s4[u], val[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w4[u], val[u][v], acc[v]);
        }
        for (; j < end; ++j) {
            const int s1 = csr[j];
            const float w1 = dinv[s1] * dc;
            float val[VEC];
            gated(s1, val);
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w1, val[v], acc[v]);
        }
        float self[VEC];
        gated(row, self);
        float* o = dh + (long long)row * F + f0;
#pragma unroll
        for (int v = 0; v < VEC; ++v) o[v] = fmaf(dc * dc, self[v], acc[v]);
    }
}
template <int VEC>
__global__ __launch_bounds__(256) void gcn_aggregate_bwd_small_k(AggBwdSmallArgs a) {
    gcn_aggregate_bwd_small_body<VEC>(a, (int)blockIdx.x, (int)gridDim.x);
}

size_t grapes_colsum_workspace_bytes(int F) { return (size_t)CS_BLOCKS * (F > 0 ? F : 1) * sizeof(float); }

// ticket (optional): GRAPES_COLSUM_TICKETS zero words, left zero — few-row inputs then take one launch
int grapes_colsum_launch(const float* src, const float* gate, const float* wrow, float* dst, float* out, int n,
                         const int32_t* d_n, int F, int accumulate, float* workspace, hipStream_t s, unsigned* ticket) {
    if (ticket && F > 16 && n <= CS_SMALL_ROWS && grapes_div_up(F, 64) <= 16) {
        int R = grapes_div_up(n, 64); if (R > CS_SMALL_R) R = CS_SMALL_R; if (R < 1) R = 1;
        hipLaunchKernelGGL(colsum_ticket_k, dim3(grapes_div_up(F, 64), R), dim3(256), 0, s, src, gate, wrow, dst, workspace, out, n,
                           d_n, F, accumulate, ticket);
        GRAPES_LAUNCH_CHECK();
        return 0;
    }
    if (F <= 16)
        hipLaunchKernelGGL(colsum_partial_narrow_k, dim3(CS_BLOCKS), dim3(256), 0, s, src, gate, wrow, dst, workspace, n, d_n, F);
    else
        hipLaunchKernelGGL(colsum_partial_k, dim3(CS_BLOCKS), dim3(256), 0, s, src, gate, wrow, dst, workspace, n, d_n, F);
    GRAPES_LAUNCH_CHECK();
    if (out) {
        hipLaunchKernelGGL(colsum_final_k, dim3(grapes_div_up(F, 64)), dim3(256), 0, s, (const float*)workspace, out, F, accumulate);
        GRAPES_LAUNCH_CHECK();
    }
    return 0;
}

// ============================================================================ C-ABI
extern "C" size_t grapes_gcn_aggregate_workspace_bytes(int32_t item_cap, int32_t f) {
    return (size_t)(item_cap > 0 ? item_cap : 0) * (f > 0 ? f : 1) * sizeof(float) + 16;
}

extern "C" int grapes_gcn_aggregate_fwd(const float* h, const int32_t* rowptr_t, const int32_t* csr_src,
                                        const float* dinv, const float* bias, float* out, int32_t n,
                                        const int32_t* d_n, int32_t f, int32_t relu, const int32_t* long_items,
                                        const int32_t* d_n_items, int32_t item_cap, void* workspace,
                                        grapes_stream_t stream) {
    if (n < 0 || f <= 0) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!h || !rowptr_t || !dinv || !out) return GRAPES_EINVAL;
    return launch_aggregate(h, rowptr_t, csr_src, dinv, bias, out, n, d_n, f, relu, long_items, d_n_items, item_cap,
                            (float*)workspace, (hipStream_t)stream);
}

/* ... that also returns head_out[r] = out[r] . head_w (see gcn_aggregate_k MODE 3): f > 16, f % 4 == 0, 16-byte aligned rows,
 * rows of any length walked by their own wavefront (no long-row items). */
extern "C" int grapes_gcn_aggregate_fwd_head(const float* h, const int32_t* rowptr_t, const int32_t* csr_src, const float* dinv,
                                             const float* bias, float* out, int32_t n, const int32_t* d_n, int32_t f,
                                             int32_t relu, const float* head_w, float* head_out, uint32_t* gate_bits,
                                             grapes_stream_t stream) {
    if (n < 0 || f <= 16 || (f & 3)) return GRAPES_EINVAL;
    if (gate_bits && (f > 256 || !relu)) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!h || !rowptr_t || !dinv || !out || !head_w || !head_out) return GRAPES_EINVAL;
    if (!aligned16(h) || !aligned16(out) || (bias && !aligned16(bias)) || !aligned16(head_w)) return GRAPES_EALIGN;
    int grid = grapes_div_up(n, 4); if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL((gcn_aggregate_k<4, 3>), dim3(grid), dim3(256), 0, (hipStream_t)stream, h, rowptr_t, csr_src, dinv, bias, out,
                       n, d_n, f, relu, 0, f >= 64 ? grapes_clock_reserve("gcn_aggregate_k<4>", grid, 4) : nullptr,
                       R1{nullptr, head_w}, head_out, (uint32_t*)gate_bits);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

// The record-driven form (gcn_aggregate_rec_k): row_head = the graph build's head records over LOCAL ids (head_ids = 0 .. n-1).
// head_w NULL: out only (MODE 0).  f <= 256, f % 4 == 0.  Bit-identical to grapes_gcn_aggregate_fwd / _fwd_head.
extern "C" int grapes_gcn_aggregate_fwd_rec(const float* h, const int32_t* row_head, const int32_t* rowptr_t, const int32_t* csr_src,
                                            const float* dinv, const float* bias, float* out, int32_t n, const int32_t* d_n,
                                            int32_t f, int32_t relu, const float* head_w, float* head_out, uint32_t* gate_bits,
                                            grapes_stream_t stream) {
    if (n < 0 || f <= 16 || (f & 3) || f > 256) return GRAPES_EINVAL;
    if (gate_bits && (!relu || !head_w)) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!h || !row_head || !rowptr_t || !csr_src || !dinv || !out || (head_w && !head_out)) return GRAPES_EINVAL;
    if (!aligned16(h) || !aligned16(out) || !aligned16(row_head) || (bias && !aligned16(bias)) || (head_w && !aligned16(head_w))) return GRAPES_EALIGN;
    // resident wavefronts that loop over pairs of rows: up to 2048 workgroups of 4 wavefronts (sweep 768 .. 4096 on Reddit: profiles/r04_workloads.txt)
    static int gcap = 0;
    if (!gcap) { const char* e = grapes_tune_env("GRAPES_AGG_REC_GRID"); gcap = e ? atoi(e) : 2048; if (gcap < 32) gcap = 2048; }
    int grid = grapes_div_up(n, 8); if (grid > gcap) grid = gcap;
    unsigned long long* clk = f >= 64 ? grapes_clock_reserve("gcn_aggregate_rec_k", grid, 4) : nullptr;
    if (head_w)
        hipLaunchKernelGGL((gcn_aggregate_rec_k<3>), dim3(grid), dim3(256), 0, (hipStream_t)stream, h, (const int4*)row_head, rowptr_t, csr_src,
                           dinv, bias, out, n, d_n, f, relu, clk, head_w, head_out, gate_bits);
    else
        hipLaunchKernelGGL((gcn_aggregate_rec_k<0>), dim3(grid), dim3(256), 0, (hipStream_t)stream, h, (const int4*)row_head, rowptr_t, csr_src,
                           dinv, bias, out, n, d_n, f, relu, clk, (const float*)nullptr, (float*)nullptr, (uint32_t*)nullptr);
    GRAPES_LAUNCH_CHECK();
    return 0;
}

/* The same aggregation over rows that are PRE-SCALED by their own dinv (hs = grapes_scale_rows(h)):
 * out[c] = dinv[c] (sum_s hs[s] + hs[c]) + bias — no per-edge gather of dinv.  f > 16, f % 4 == 0, 16-byte aligned rows. */
extern "C" int grapes_gcn_aggregate_fwd_prescaled(const float* hs, const int32_t* rowptr_t, const int32_t* csr_src,
                                                  const float* dinv, const float* bias, float* out, int32_t n,
                                                  const int32_t* d_n, int32_t f, int32_t relu, const int32_t* long_items,
                                                  const int32_t* d_n_items, int32_t item_cap, void* workspace,
                                                  grapes_stream_t stream) {
    if (n < 0 || f <= 16 || (f & 3)) return GRAPES_EINVAL;
    if (n == 0) return 0;
    if (!hs || !rowptr_t || !dinv || !out) return GRAPES_EINVAL;
    return launch_aggregate(hs, rowptr_t, csr_src, dinv, bias, out, n, d_n, f, relu, long_items, d_n_items, item_cap,
                            (float*)workspace, (hipStream_t)stream, true);
}

// ---- backward of  first layer (transform-first) -> ReLU -> 1-wide head  without the n x H temporaries (R1 above):
//        dW2[m] = sum_r dh2[r] act[r][m]            db1[m] = sum_r [act[r][m] > 0] dh2[r] w2[m]
//        dH[s]  = sum_{r in out(s)} w_sr dpre[r] + w_ss dpre[s],   dpre[r][m] = [act[r][m] > 0] dh2[r] w2[m]
// one streaming pass over act for the two column sums (blocking and order of colsum_partial_k / colsum_final_k: the sums are
// those of the three-launch path bit for bit) + the by-source aggregation with gated gathers.
__device__ __forceinline__ void colsum_rank1_partial_body(const float* __restrict__ act, const float* __restrict__ dh2,
                                                          const float* __restrict__ w2, float* __restrict__ partial_a,
                                                          float* __restrict__ partial_b, int n_host, const int32_t* d_n, int F, int BID) {
    const int n = eff_count(d_n, n_host);
    for (int c = threadIdx.x; c < F; c += 256) {
        const float wc = w2[c];
        float acc_a = 0.f, acc_b = 0.f;
        for (int r0 = BID * CS_ROWS; r0 < n; r0 += CS_BLOCKS * CS_ROWS) {
            const int r1 = r0 + CS_ROWS < n ? r0 + CS_ROWS : n;
            int r = r0;
            // sixteen rows in flight (a thread's loads are 4 bytes each: with four, a 77k x 256 pass — 79 MB — ran at 3 TB/s);
            // the additions stay in row order
            for (; r + 16 <= r1; r += 16) {
                float v[16], d[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { v[u] = act[(long long)(r + u) * F + c]; d[u] = dh2[r + u]; }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    acc_a += d[u] * v[u];
                    acc_b += v[u] > 0.f ? d[u] * wc : 0.f;
                }
            }
            for (; r + 4 <= r1; r += 4) {      // four rows in flight
                float v[4], d[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { v[u] = act[(long long)(r + u) * F + c]; d[u] = dh2[r + u]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc_a += d[u] * v[u];
                    acc_b += v[u] > 0.f ? d[u] * wc : 0.f;
                }
            }
            for (; r < r1; ++r) {
                const float v = act[(long long)r * F + c], d = dh2[r];
                acc_a += d * v;
                acc_b += v > 0.f ? d * wc : 0.f;
            }
        }
        partial_a[(long long)BID * F + c] = acc_a;
        partial_b[(long long)BID * F + c] = acc_b;
    }
}
__global__ __launch_bounds__(256) void colsum_rank1_partial_k(const float* __restrict__ act, const float* __restrict__ dh2,
                                                              const float* __restrict__ w2, float* __restrict__ partial_a,
                                                              float* __restrict__ partial_b, int n_host, const int32_t* d_n, int F) {
    colsum_rank1_partial_body(act, dh2, w2, partial_a, partial_b, n_host, d_n, F, (int)blockIdx.x);
}
// colsum_final_k for two partial tables (blockIdx.y)
__global__ __launch_bounds__(256) void colsum_final2_k(const float* __restrict__ partial_a, const float* __restrict__ partial_b,
                                                       float* __restrict__ out_a, float* __restrict__ out_b, int F, int accumulate) {
    __shared__ float part[4][64];
    const float* partial = blockIdx.y ? partial_b : partial_a;
    float* out = blockIdx.y ? out_b : out_a;
    if (!out) return;
    const int g = threadIdx.x >> 6, cl = threadIdx.x & 63;
    const int c = blockIdx.x * 64 + cl;
    float acc = 0.f;
    if (c < F) acc = colsum_range_sum(partial, g * (CS_BLOCKS / 4), F, c);
    part[g][cl] = acc;
    __syncthreads();
    if (g == 0 && c < F) {
        const float t = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
        out[c] = accumulate ? out[c] + t : t;
    }
}
extern "C" size_t grapes_gcn_aggregate_bwd_rank1_workspace_bytes(int32_t item_cap, int32_t f) {
    return 2 * grapes_colsum_workspace_bytes(f) + grapes_gcn_aggregate_workspace_bytes(item_cap, f);
}
static int bwd_rank1_impl(const float* act, const uint32_t* gate_bits, const float* dh2, const float* w2, const int32_t* rowptr_s,
                         const int32_t* csr_dst, const float* dinv, float* dh, float* dw2, float* db1,
                         int32_t accumulate, int32_t n, const int32_t* d_n, int32_t f,
                         const int32_t* long_items, const int32_t* d_n_items, int32_t item_cap,
                         void* workspace, grapes_stream_t stream) {
    if (n < 0 || f <= 16 || (f & 3) || (gate_bits && f > 256)) return GRAPES_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        if (!accumulate) {
            hipError_t e;
            if (dw2 && (e = grapes_zero_async(dw2, (size_t)f * sizeof(float), s))) return (int)e;
            if (db1 && (e = grapes_zero_async(db1, (size_t)f * sizeof(float), s))) return (int)e;
        }
        return 0;
    }
    if (!act || !dh2 || !w2 || !rowptr_s || !dinv || !dh || !workspace) return GRAPES_EINVAL;
    if (!aligned16(act) || !aligned16(dh)) return GRAPES_EALIGN;
    float* pa = (float*)workspace;
    float* pb = pa + (size_t)CS_BLOCKS * f;
    float* partials = pb + (size_t)CS_BLOCKS * f;
    if (dw2 || db1) {
        hipLaunchKernelGGL(colsum_rank1_partial_k, dim3(CS_BLOCKS), dim3(256), 0, s, act, dh2, w2, pa, pb, n, d_n, f);
        GRAPES_LAUNCH_CHECK();
        hipLaunchKernelGGL(colsum_final2_k, dim3(grapes_div_up(f, 64), 2), dim3(256), 0, s, (const float*)pa, (const float*)pb, dw2, db1, f,
                           accumulate);
        GRAPES_LAUNCH_CHECK();
    }
    const R1 r1{dh2, w2};
    int grid = grapes_div_up(n, 4); if (grid > 16384) grid = 16384;
    const int skip = (long_items && d_n_items && item_cap > 0) ? 1 : 0;
    if (skip && !aligned16(partials)) return GRAPES_EALIGN;
    if (gate_bits) {     // the gates from 32 bytes of bits per row (MODE 4) instead of the activation rows
        static int cap = 0;
        if (!cap) { const char* e = grapes_tune_env("GRAPES_R1BITS_GRID"); cap = e ? atoi(e) : 2048; if (cap < 1) cap = 2048; }
        const float* hb = reinterpret_cast<const float*>(gate_bits);
        int gb = grapes_div_up(grapes_div_up(n, 4), 4); if (gb > cap) gb = cap;
        hipLaunchKernelGGL(gcn_aggregate_r1bits_k, dim3(gb), dim3(256), 0, s, (const uint32_t*)gate_bits, rowptr_s, csr_dst, dinv,
                           dh, n, d_n, f, skip, r1);
        GRAPES_LAUNCH_CHECK();
        if (skip) {
            const int g2 = item_cap < 2048 ? item_cap : 2048;
            hipLaunchKernelGGL((gcn_aggregate_chunks_k<4, 4>), dim3(g2), dim3(256), 0, s, hb, rowptr_s, csr_dst, dinv, f, long_items,
                               d_n_items, item_cap, partials, r1);
            GRAPES_LAUNCH_CHECK();
            hipLaunchKernelGGL(gcn_aggregate_combine_k, dim3(g2), dim3(256), 0, s, hb, rowptr_s, dinv, (const float*)nullptr, dh, f, 0,
                               long_items, d_n_items, item_cap, (const float*)partials, 0, r1, 1);
            GRAPES_LAUNCH_CHECK();
        }
        return 0;
    }
    hipLaunchKernelGGL((gcn_aggregate_k<4, 2>), dim3(grid), dim3(256), 0, s, act, rowptr_s, csr_dst, dinv, (const float*)nullptr, dh, n,
                       d_n, f, 0, skip, (unsigned long long*)nullptr, r1);
    GRAPES_LAUNCH_CHECK();
    if (skip) {
        const int g2 = item_cap < 2048 ? item_cap : 2048;
        hipLaunchKernelGGL((gcn_aggregate_chunks_k<4, 2>), dim3(g2), dim3(256), 0, s, act, rowptr_s, csr_dst, dinv, f, long_items,
                           d_n_items, item_cap, partials, r1);
        GRAPES_LAUNCH_CHECK();
        hipLaunchKernelGGL(gcn_aggregate_combine_k, dim3(g2), dim3(256), 0, s, act, rowptr_s, dinv, (const float*)nullptr, dh, f, 0,
                           long_items, d_n_items, item_cap, (const float*)partials, 0, r1, 0);
        GRAPES_LAUNCH_CHECK();
    }
    return 0;
}
extern "C" int grapes_gcn_aggregate_bwd_rank1(const float* act, const float* dh2, const float* w2, const int32_t* rowptr_s,
                                              const int32_t* csr_dst, const float* dinv, float* dh, float* dw2, float* db1,
                                              int32_t accumulate, int32_t n, const int32_t* d_n, int32_t f,
                                              const int32_t* long_items, const int32_t* d_n_items, int32_t item_cap,
                                              void* workspace, grapes_stream_t stream) {
    return bwd_rank1_impl(act, nullptr, dh2, w2, rowptr_s, csr_dst, dinv, dh, dw2, db1, accumulate, n, d_n, f, long_items, d_n_items,
                          item_cap, workspace, stream);
}
/* ... with the ReLU gates of the aggregation taken from gate_bits (grapes_gcn_aggregate_fwd_head) instead of from act; act still
 * feeds the two column sums.  f <= 256. */
extern "C" int grapes_gcn_aggregate_bwd_rank1_bits(const float* act, const uint32_t* gate_bits, const float* dh2, const float* w2,
                                                   const int32_t* rowptr_s, const int32_t* csr_dst, const float* dinv, float* dh,
                                                   float* dw2, float* db1, int32_t accumulate, int32_t n, const int32_t* d_n,
                                                   int32_t f, const int32_t* long_items, const int32_t* d_n_items,
                                                   int32_t item_cap, void* workspace, grapes_stream_t stream) {
    if (!gate_bits) return GRAPES_EINVAL;
    return bwd_rank1_impl(act, gate_bits, dh2, w2, rowptr_s, csr_dst, dinv, dh, dw2, db1, accumulate, n, d_n, f, long_items, d_n_items,
                          item_cap, workspace, stream);
}

// ---- the same for up to three INDEPENDENT problems in the same five launches (the hops of the sampler net and the log-Z net:
// their backward chains depend on the losses only, and each launch of a chain is a few dependent round trips long whatever it
// moves).  blockIdx.y = problem; fields picked by ternary chains (indexing a by-value struct would go through scratch).
#define R1M_MAX 3
struct R1Multi {
    int count;
    const float* act[R1M_MAX]; const uint32_t* bits[R1M_MAX]; const float* dh2[R1M_MAX]; const float* w2[R1M_MAX];
    const int32_t* rowptr[R1M_MAX]; const int32_t* csr[R1M_MAX]; const float* dinv[R1M_MAX]; float* dh[R1M_MAX];
    int n[R1M_MAX]; const int32_t* d_n[R1M_MAX]; const int32_t* items[R1M_MAX]; const int32_t* d_n_items[R1M_MAX]; int item_cap[R1M_MAX];
    float* pa[R1M_MAX]; float* pb[R1M_MAX]; float* partials[R1M_MAX];
};
// output groups of the column sums: the members of a group name the same dw2 / db1 and are added in member order, each with its
// own accumulate flag — exactly what the calls one after the other leave there
struct R1Groups { int ngroups; int nmem[R1M_MAX]; int mem[R1M_MAX][R1M_MAX]; int acc[R1M_MAX][R1M_MAX]; float* dw2[R1M_MAX]; float* db1[R1M_MAX]; };
#define R1SEL(m, f, q) ((q) == 0 ? (m).f[0] : ((q) == 1 ? (m).f[1] : (m).f[2]))

__global__ __launch_bounds__(256) void colsum_rank1_partial_multi_k(R1Multi m, int F) {
    const int q = blockIdx.y;
    colsum_rank1_partial_body(R1SEL(m, act, q), R1SEL(m, dh2, q), R1SEL(m, w2, q), R1SEL(m, pa, q), R1SEL(m, pb, q), R1SEL(m, n, q),
                              R1SEL(m, d_n, q), F, (int)blockIdx.x);
}
// grid (F / 64, 2, groups): colsum_final2_k over the members of a group, one after the other
__global__ __launch_bounds__(256) void colsum_final2_multi_k(R1Multi m, R1Groups gr, int F) {
    __shared__ float part[4][64];
    const int gi = blockIdx.z;
    float* out = blockIdx.y ? R1SEL(gr, db1, gi) : R1SEL(gr, dw2, gi);
    if (!out) return;
    const int nm = R1SEL(gr, nmem, gi);
    const int g = threadIdx.x >> 6, cl = threadIdx.x & 63;
    const int c = blockIdx.x * 64 + cl;
    // the members' column sums are independent: their loads go out together (each still summed in colsum_final2_k's order),
    // then they are combined one after the other
    const float* pm[R1M_MAX]; int am[R1M_MAX];
#pragma unroll
    for (int k = 0; k < R1M_MAX; ++k) {
        const int q = gi == 0 ? R1SEL(gr, mem[0], k) : (gi == 1 ? R1SEL(gr, mem[1], k) : R1SEL(gr, mem[2], k));
        am[k] = gi == 0 ? R1SEL(gr, acc[0], k) : (gi == 1 ? R1SEL(gr, acc[1], k) : R1SEL(gr, acc[2], k));
        pm[k] = blockIdx.y ? R1SEL(m, pb, q) : R1SEL(m, pa, q);
    }
    float acc[R1M_MAX] = {0.f, 0.f, 0.f};
    if (c < F) {
        const int b0 = g * (CS_BLOCKS / 4);
        constexpr int CS_FINAL_BATCH = 32;          // (three members' batches in registers: 96 of them)
        static_assert((CS_BLOCKS / 4) % CS_FINAL_BATCH == 0, "whole batches");
#pragma unroll 1
        for (int bb = b0; bb < b0 + CS_BLOCKS / 4; bb += CS_FINAL_BATCH) {       // (colsum_range_sum for the members at once)
            float v[R1M_MAX][CS_FINAL_BATCH];
#pragma unroll
            for (int k = 0; k < R1M_MAX; ++k)
                if (k < nm) {
#pragma unroll
                    for (int u = 0; u < CS_FINAL_BATCH; ++u) v[k][u] = pm[k][(long long)(bb + u) * F + c];
                }
#pragma unroll
            for (int k = 0; k < R1M_MAX; ++k)
                if (k < nm) {
#pragma unroll
                    for (int u = 0; u < CS_FINAL_BATCH; ++u) acc[k] += v[k][u];
                }
        }
    }
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < R1M_MAX; ++k) {
        if (k < nm) {                                        // (uniform)
            part[g][cl] = acc[k];
            __syncthreads();
            if (g == 0 && c < F) {
                const float t = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
                if (k == 0 && am[0]) v = out[c];
                v = am[k] ? v + t : t;
            }
            __syncthreads();
        }
    }
    if (g == 0 && c < F) out[c] = v;
}
__global__ __launch_bounds__(256) void gcn_aggregate_r1bits_multi_k(R1Multi m, int F) {
    const int q = blockIdx.y;
    const int skip = R1SEL(m, items, q) != nullptr;
    gcn_aggregate_r1bits_body(R1SEL(m, bits, q), R1SEL(m, rowptr, q), R1SEL(m, csr, q), R1SEL(m, dinv, q), R1SEL(m, dh, q), R1SEL(m, n, q),
                              R1SEL(m, d_n, q), F, skip, R1{R1SEL(m, dh2, q), R1SEL(m, w2, q)}, (int)blockIdx.x, (int)gridDim.x);
}
__global__ __launch_bounds__(256) void gcn_aggregate_chunks_r1bits_multi_k(R1Multi m, int F) {
    const int q = blockIdx.y;
    if (R1SEL(m, items, q) == nullptr) return;                       // (uniform over the workgroup)
    gcn_aggregate_chunks_body<4, 4>(reinterpret_cast<const float*>(R1SEL(m, bits, q)), R1SEL(m, rowptr, q), R1SEL(m, csr, q), R1SEL(m, dinv, q), F,
                                    R1SEL(m, items, q), R1SEL(m, d_n_items, q), R1SEL(m, item_cap, q), R1SEL(m, partials, q),
                                    R1{R1SEL(m, dh2, q), R1SEL(m, w2, q)}, (int)blockIdx.x, (int)gridDim.x);
}
__global__ __launch_bounds__(256) void gcn_aggregate_combine_r1bits_multi_k(R1Multi m, int F) {
    const int q = blockIdx.y;
    if (R1SEL(m, items, q) == nullptr) return;
    gcn_aggregate_combine_body(reinterpret_cast<const float*>(R1SEL(m, bits, q)), R1SEL(m, rowptr, q), R1SEL(m, dinv, q), nullptr, R1SEL(m, dh, q), F, 0,
                               R1SEL(m, items, q), R1SEL(m, d_n_items, q), R1SEL(m, item_cap, q), R1SEL(m, partials, q), 0,
                               R1{R1SEL(m, dh2, q), R1SEL(m, w2, q)}, 1, (int)blockIdx.x, (int)gridDim.x);
}

extern "C" int grapes_gcn_aggregate_bwd_rank1_bits_multi(int32_t count, const float* const* act, const uint32_t* const* gate_bits,
                                                         const float* const* dh2, const float* const* w2,
                                                         const int32_t* const* rowptr_s, const int32_t* const* csr_dst,
                                                         const float* const* dinv, float* const* dh, float* const* dw2,
                                                         float* const* db1, const int32_t* accumulate, const int32_t* n,
                                                         const int32_t* const* d_n, int32_t f, const int32_t* const* long_items,
                                                         const int32_t* const* d_n_items, const int32_t* item_cap,
                                                         void* const* workspace, grapes_stream_t stream) {
    if (count < 1 || count > R1M_MAX || f <= 16 || (f & 3) || f > 256) return GRAPES_EINVAL;
    if (!act || !gate_bits || !dh2 || !w2 || !rowptr_s || !csr_dst || !dinv || !dh || !dw2 || !db1 || !accumulate || !n || !d_n ||
        !long_items || !d_n_items || !item_cap || !workspace)
        return GRAPES_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    R1Multi m{}; R1Groups gr{};
    m.count = count;
    int nmax = 0, capmax = 0, any_sum = 0;
    for (int q = 0; q < R1M_MAX; ++q) {
        const int p = q < count ? q : 0;
        if (q < count) {
            if (n[p] <= 0 || !act[p] || !gate_bits[p] || !dh2[p] || !w2[p] || !rowptr_s[p] || !dinv[p] || !dh[p] || !workspace[p]) return GRAPES_EINVAL;
            if (!aligned16(act[p]) || !aligned16(dh[p]) || !aligned16(workspace[p])) return GRAPES_EALIGN;
            if ((dw2[p] == nullptr) != (db1[p] == nullptr)) return GRAPES_EINVAL;
        }
        m.act[q] = act[p]; m.bits[q] = gate_bits[p]; m.dh2[q] = dh2[p]; m.w2[q] = w2[p]; m.rowptr[q] = rowptr_s[p]; m.csr[q] = csr_dst[p];
        m.dinv[q] = dinv[p]; m.dh[q] = dh[p]; m.n[q] = q < count ? n[p] : 0; m.d_n[q] = d_n[p];
        const bool skip = long_items[p] && d_n_items[p] && item_cap[p] > 0;
        m.items[q] = skip ? long_items[p] : nullptr; m.d_n_items[q] = d_n_items[p]; m.item_cap[q] = skip ? item_cap[p] : 0;
        m.pa[q] = (float*)workspace[p]; m.pb[q] = m.pa[q] + (size_t)CS_BLOCKS * f; m.partials[q] = m.pb[q] + (size_t)CS_BLOCKS * f;
        if (q < count) {
            if (n[p] > nmax) nmax = n[p];
            if (m.item_cap[q] > capmax) capmax = m.item_cap[q];
            if (dw2[p]) {
                any_sum = 1;
                int g = 0;
                for (; g < gr.ngroups; ++g) if (gr.dw2[g] == dw2[p]) break;
                if (g == gr.ngroups) { gr.dw2[g] = dw2[p]; gr.db1[g] = db1[p]; gr.ngroups++; }
                else if (gr.db1[g] != db1[p]) return GRAPES_EINVAL;
                gr.mem[g][gr.nmem[g]] = q; gr.acc[g][gr.nmem[g]] = accumulate[p] ? 1 : 0; gr.nmem[g]++;
            }
        }
    }
    if (any_sum) {
        hipLaunchKernelGGL(colsum_rank1_partial_multi_k, dim3(CS_BLOCKS, count), dim3(256), 0, s, m, f);
        GRAPES_LAUNCH_CHECK();
        hipLaunchKernelGGL(colsum_final2_multi_k, dim3(grapes_div_up(f, 64), 2, gr.ngroups), dim3(256), 0, s, m, gr, f);
        GRAPES_LAUNCH_CHECK();
    }
    static int cap = 0;
    if (!cap) { const char* e = grapes_tune_env("GRAPES_R1BITS_GRID"); cap = e ? atoi(e) : 2048; if (cap < 1) cap = 2048; }
    int gb = grapes_div_up(grapes_div_up(nmax, 4), 4); if (gb > cap) gb = cap;
    hipLaunchKernelGGL(gcn_aggregate_r1bits_multi_k, dim3(gb, count), dim3(256), 0, s, m, f);
    GRAPES_LAUNCH_CHECK();
    if (capmax > 0) {
        const int g2 = capmax < 2048 ? capmax : 2048;
        hipLaunchKernelGGL(gcn_aggregate_chunks_r1bits_multi_k, dim3(g2, count), dim3(256), 0, s, m, f);
        GRAPES_LAUNCH_CHECK();
        hipLaunchKernelGGL(gcn_aggregate_combine_r1bits_multi_k, dim3(g2, count), dim3(256), 0, s, m, f);
        GRAPES_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" size_t grapes_gcn_aggregate_bwd_workspace_bytes(int32_t item_cap, int32_t f) {
    return grapes_colsum_workspace_bytes(f) + grapes_gcn_aggregate_workspace_bytes(item_cap, f);
}

extern "C" int grapes_gcn_aggregate_bwd(const float* dout, const float* relu_out, const int32_t* rowptr_s,
                                        const int32_t* csr_dst, const float* dinv, float* dpre_buf, float* dh,
                                        float* dbias, int32_t accumulate_bias, int32_t n, const int32_t* d_n,
                                        int32_t f, const int32_t* long_items, const int32_t* d_n_items,
                                        int32_t item_cap, void* workspace, uint32_t* d_ticket, grapes_stream_t stream) {
    if (n < 0 || f <= 0) return GRAPES_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        if (dbias && !accumulate_bias) { hipError_t e = grapes_zero_async(dbias, (size_t)f * sizeof(float), s); if (e) return (int)e; }
        return 0;
    }
    if (!dout || !rowptr_s || !dinv || !dh || !dpre_buf) return GRAPES_EINVAL;
    const bool need_pass = (relu_out != nullptr) || (dbias != nullptr) || (dpre_buf != dout);
    if ((need_pass || long_items) && !workspace) return GRAPES_EINVAL;
    {   // few rows: mask, bias gradient and aggregation in ONE launch (dpre_buf is then not written)
        const bool vec = (f % 4 == 0) && aligned16(dout) && aligned16(dh) && (!relu_out || aligned16(relu_out));
        const int VECW = vec ? 4 : 1;
        if (d_ticket && !long_items && n <= CS_SMALL_ROWS && f > 16 && f <= 64 * VECW && grapes_div_up(f, 64) <= 16 && (relu_out || dbias)) {
            const int ncb = grapes_div_up(f, 64);
            int R = grapes_div_up(n, 64); if (R > CS_SMALL_R) R = CS_SMALL_R; if (R < 1) R = 1;
            const int grid = (dbias ? ncb * R : 0) + grapes_div_up(n, 4);
            const AggBwdSmallArgs A{dout, relu_out, rowptr_s, csr_dst, dinv, dh, n, d_n, f, dbias, accumulate_bias, (float*)workspace,
                                    (unsigned*)d_ticket, ncb, R};
            auto single = [=](hipStream_t st) {
                if (vec) hipLaunchKernelGGL((gcn_aggregate_bwd_small_k<4>), dim3(grid), dim3(256), 0, st, A);
                else hipLaunchKernelGGL((gcn_aggregate_bwd_small_k<1>), dim3(grid), dim3(256), 0, st, A);
            };
            // recorded (riders): the launch is two dependent round trips long whatever it moves — it may ride in a launch of the
            // sampler heads' backward pass, which does not depend on it (grapes_sampler_head_bwd_multi_phase)
            if (grapes_rider_recording()) { grapes_rider_record(grapes_rider_make(GRAPES_RK_AGGBWD, VECW, grid, 256, A, single)); return 0; }
            single(s);
            GRAPES_LAUNCH_CHECK();
            return 0;
        }
    }
    if (need_pass) {
        float* dst = (relu_out != nullptr || dpre_buf != dout) ? dpre_buf : nullptr;
        int rc = grapes_colsum_launch(dout, relu_out, nullptr, dst, dbias, n, d_n, f, accumulate_bias, (float*)workspace, s, d_ticket);
        if (rc) return rc;
    }
    float* partials = workspace ? (float*)((char*)workspace + grapes_colsum_workspace_bytes(f)) : nullptr;
    return launch_aggregate(dpre_buf, rowptr_s, csr_dst, dinv, nullptr, dh, n, d_n, f, 0, long_items, d_n_items, item_cap,
                            partials, s);
}


// ============================================================================ backward of the sampler's 1-wide head, all hops
// d log_prob / d logit of hop q, DENSE over the hop's batch rows (zero for rows that were not candidates; cand_pos is
// the inverse of nb_local from grapes_frontier_compact), and the sum of all of them (the head's bias gradient): one
// launch for up to four hops, partials published and combined by the last workgroup in (hop, workgroup) order.
struct BernSegs {
    int count;
    const float* logits[4]; const float* mask[4]; const int32_t* cand_pos[4]; float* dlog[4]; const int32_t* d_n[4]; int n_cap[4];
};
struct BernArgs { const float* d_grad_scale; float* sum_out; int accumulate_sum; float* partials; unsigned* ticket; float* mean_sum_out; };
// (BX, GX: workgroup index / count along x of THIS problem; blockIdx.y = segment in every form)
__device__ __forceinline__ void bernoulli_dense_multi_body(const BernSegs& sg, const BernArgs& ba, int BX, int GX) {
    const float* d_grad_scale = ba.d_grad_scale; float* __restrict__ sum_out = ba.sum_out; const int accumulate_sum = ba.accumulate_sum;
    float* __restrict__ partials = ba.partials; unsigned* __restrict__ ticket = ba.ticket; float* __restrict__ mean_sum_out = ba.mean_sum_out;
    __shared__ float red[4];
    __shared__ int s_last;
    const int q = blockIdx.y;
#define BSEL(f) (q == 0 ? sg.f[0] : (q == 1 ? sg.f[1] : (q == 2 ? sg.f[2] : sg.f[3])))
    const float* logits = BSEL(logits); const float* mask = BSEL(mask); const int32_t* cp = BSEL(cand_pos);
    float* dlog = BSEL(dlog);
    const int n = eff_count(BSEL(d_n), BSEL(n_cap));
#undef BSEL
    const float gs = d_grad_scale ? *d_grad_scale : 1.0f;
    float local = 0.f;
    if (mask == nullptr) {       // a "mean" segment (the log-Z head, main.py:228): d mean / d x = scale / n on every live row
        const float v = gs * 1.0f / (float)(n > 0 ? n : 1);
        for (int r = BX * 256 + (int)threadIdx.x; r < n; r += GX * 256) dlog[r] = v;
        if (BX == 0 && threadIdx.x == 0 && mean_sum_out) *mean_sum_out = v * (float)n;   // its own bias gradient
    } else {
        for (int r = BX * 256 + (int)threadIdx.x; r < n; r += GX * 256) {
            const int c = cp[r];
            const float l = logits[r];
            float v = 0.f;
            if (c >= 0) v = gs * (mask[c] - 1.0f / (1.0f + expf(-l)));      // utils.py:71 differentiated (as bernoulli_logprob_bwd_k)
            dlog[r] = v;
            local += v;
        }
    }
    local = wave_sum(local);
    if (lane_id() == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    const unsigned nblk = (unsigned)GX * gridDim.y;
    if (threadIdx.x == 0) {
        publish_f32(&partials[blockIdx.y * GX + BX], (red[0] + red[1]) + (red[2] + red[3]));
        s_last = (atomicAdd(ticket, 1u) == nblk - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    float acc = 0.f;                                  // fixed order: thread t owns partials t, t+256, ...; xor tree; waves in order
    for (unsigned b = threadIdx.x; b < nblk; b += blockDim.x)
        acc += __int_as_float(__hip_atomic_load((const int*)(partials + b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    acc = wave_sum(acc);
    __syncthreads();
    if (lane_id() == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (red[0] + red[1]) + (red[2] + red[3]);
        if (sum_out) *sum_out = accumulate_sum ? *sum_out + t : t;
        *ticket = 0u;
    }
}
__global__ __launch_bounds__(256) void bernoulli_dense_multi_k(BernSegs sg, BernArgs ba) {
    bernoulli_dense_multi_body(sg, ba, (int)blockIdx.x, (int)gridDim.x);
}
// The heads' launches carrying a recorded few-row backward aggregation of the classifier (riders): columns x >= gx of the grid
// are the rider's workgroups, numbered (x - gx) + y * (gridDim.x - gx).  Same bodies, same results as the launches on their own.
template <int VEC>
__global__ __launch_bounds__(256) void bernoulli_aggbwd_pair_k(BernSegs sg, BernArgs ba, int gx, AggBwdSmallArgs b) {
    if ((int)blockIdx.x < gx) bernoulli_dense_multi_body(sg, ba, (int)blockIdx.x, gx);
    else {
        const int rx = (int)gridDim.x - gx;
        gcn_aggregate_bwd_small_body<VEC>(b, ((int)blockIdx.x - gx) + (int)blockIdx.y * rx, rx * (int)gridDim.y);
    }
}
template <int VEC>
__global__ __launch_bounds__(256) void narrow_multi_aggbwd_pair_k(NarrowSegs sg, int F, int narrow_lane_rows, int gx, AggBwdSmallArgs b) {
    if ((int)blockIdx.x < gx) {
        const int q = blockIdx.y;
#define NSEL(f) (q == 0 ? sg.f[0] : (q == 1 ? sg.f[1] : (q == 2 ? sg.f[2] : sg.f[3])))
        narrow_body(NSEL(h), NSEL(rowptr), NSEL(csr), NSEL(dinv), NSEL(bias), NSEL(out), NSEL(n_cap), NSEL(d_n), F, 0,
                    narrow_lane_rows, (int)blockIdx.x, gx);
#undef NSEL
    } else {
        const int rx = (int)gridDim.x - gx;
        gcn_aggregate_bwd_small_body<VEC>(b, ((int)blockIdx.x - gx) + (int)blockIdx.y * rx, rx * (int)gridDim.y);
    }
}

extern "C" size_t grapes_sampler_head_bwd_multi_workspace_bytes(void) { return (size_t)4 * 32 * sizeof(float); }

extern "C" int grapes_sampler_head_bwd_multi_phase(int32_t count, const float* const* logits, const float* const* mask,
                                             const int32_t* const* cand_pos, const int32_t* n_cap, const int32_t* const* d_n,
                                             const float* d_grad_scale, const int32_t* const* rowptr_s,
                                             const int32_t* const* csr_dst, const float* const* dinv, float* const* dlogits,
                                             float* const* dh, float* sum_out, int32_t accumulate_sum, float* mean_sum_out,
                                             void* workspace, uint32_t* d_ticket, int32_t phase, grapes_stream_t stream) {
    if (phase < 0 || phase > 2) return GRAPES_EINVAL;
    if (count < 1 || count > 4 || !logits || !mask || !cand_pos || !n_cap || !d_n || !rowptr_s || !csr_dst || !dinv || !dlogits ||
        !dh || !workspace || !d_ticket)
        return GRAPES_EINVAL;
    BernSegs bs{}; NarrowSegs ns{};
    bs.count = ns.count = count;
    int nmax = 0;
    for (int q = 0; q < 4; ++q) {
        const int p = q < count ? q : 0;
        if (q < count && (!rowptr_s[p] || !dinv[p] || !dlogits[p] || !dh[p] || n_cap[p] <= 0)) return GRAPES_EINVAL;
        if (q < count && mask[p] && (!logits[p] || !cand_pos[p])) return GRAPES_EINVAL;      // mask NULL = a mean segment
        bs.logits[q] = logits[p]; bs.mask[q] = mask[p]; bs.cand_pos[q] = cand_pos[p]; bs.dlog[q] = dlogits[p]; bs.d_n[q] = d_n[p];
        bs.n_cap[q] = q < count ? n_cap[p] : 0;
        ns.h[q] = dlogits[p]; ns.rowptr[q] = rowptr_s[p]; ns.csr[q] = csr_dst[p]; ns.dinv[q] = dinv[p]; ns.out[q] = dh[p];
        ns.d_n[q] = d_n[p]; ns.n_cap[q] = q < count ? n_cap[p] : 0; ns.bias[q] = nullptr;
        if (q < count && n_cap[p] > nmax) nmax = n_cap[p];
    }
    hipStream_t s = (hipStream_t)stream;
    int g1 = grapes_div_up(nmax, 1024); if (g1 > 32) g1 = 32; if (g1 < 1) g1 = 1;     // few workgroups: one ticket address
    // a pending recorded few-row backward aggregation (riders) rides as extra columns of the grid
    auto rider = [&](AggBwdSmallArgs& B, int& vecw, int& rx) -> bool {
        const GrapesRiderRecord* r = grapes_rider_match(GRAPES_RK_AGGBWD, 4, 256, s);
        vecw = 4;
        if (!r) { r = grapes_rider_match(GRAPES_RK_AGGBWD, 1, 256, s); vecw = 1; }
        if (!r) return false;
        memcpy(&B, r->args, sizeof B);
        rx = grapes_div_up(r->grid, count);
        return true;
    };
    AggBwdSmallArgs B; int vecw = 0, rx = 0;
    if (phase != 2) {
        const BernArgs ba{d_grad_scale, sum_out, accumulate_sum, (float*)workspace, (unsigned*)d_ticket, mean_sum_out};
        if (rider(B, vecw, rx)) {
            if (vecw == 4) hipLaunchKernelGGL((bernoulli_aggbwd_pair_k<4>), dim3(g1 + rx, count), dim3(256), 0, s, bs, ba, g1, B);
            else hipLaunchKernelGGL((bernoulli_aggbwd_pair_k<1>), dim3(g1 + rx, count), dim3(256), 0, s, bs, ba, g1, B);
        } else
            hipLaunchKernelGGL(bernoulli_dense_multi_k, dim3(g1, count), dim3(256), 0, s, bs, ba);
        GRAPES_LAUNCH_CHECK();
    }
    if (phase != 1) {
        int g2 = grapes_div_up(nmax, 256); if (g2 > 4096) g2 = 4096;
        if (rider(B, vecw, rx)) {
            if (vecw == 4) hipLaunchKernelGGL((narrow_multi_aggbwd_pair_k<4>), dim3(g2 + rx, count), dim3(256), 0, s, ns, 1, narrow_lane_rows_cfg(), g2, B);
            else hipLaunchKernelGGL((narrow_multi_aggbwd_pair_k<1>), dim3(g2 + rx, count), dim3(256), 0, s, ns, 1, narrow_lane_rows_cfg(), g2, B);
        } else
            hipLaunchKernelGGL(gcn_aggregate_narrow_multi_k, dim3(g2, count), dim3(256), 0, s, ns, 1, narrow_lane_rows_cfg());
        GRAPES_LAUNCH_CHECK();
    }
    return 0;
}
