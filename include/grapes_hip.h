/*
 * grapes_hip.h — C-ABI of libgrapes_hip.so: the MI355X (gfx950) implementation of GRAPES'
 * per-layer sample-then-aggregate training step.
 *
 * Every entry point replaces one piece of the reference's hot path (dfdazac/grapes; citations
 * are file:line in that repository).  The reference is pure Python, so the binding a maintainer
 * adds is a ctypes stub; grapes_amd/_lib.py is exactly that stub.
 *
 * Conventions
 *  - All pointers are DEVICE pointers (hipMalloc / torch caching allocator) unless named h_*.
 *  - Node ids / local ids / CSR column indices are int32; CSR row pointers of the *graph* are
 *    int64 (ogbn-papers100M has 3.2e9 directed edges); per-hop CSRs use int32 row pointers.
 *  - Feature matrices are dense row-major fp32.
 *  - Dynamic sizes: a `const int32_t* d_x` argument, when non-NULL, points to the true element
 *    count on the device; the host argument next to it is then only the CAPACITY used to size the
 *    launch.  When NULL the host argument is the exact size.  No entry point synchronises the
 *    device or allocates memory: all of them only enqueue work on `stream` and are therefore
 *    legal inside hipGraph capture.
 *  - `status` (optional, may be NULL) is a device int32 word; kernels OR error bits into it
 *    (GRAPES_STATUS_*).  The caller reads it at its own synchronisation point.
 *  - Return value: 0 on success, a negative GRAPES_E* code for an invalid argument, or a positive
 *    hipError_t if a launch failed.  Nothing aborts.
 */
#ifndef GRAPES_HIP_H
#define GRAPES_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version = 100 * MAJOR + MINOR.
 *   MAJOR changes when an entry point declared here changes its signature or meaning, or leaves the product surface: a binding
 *         written against another MAJOR must refuse to load the library.
 *   MINOR counts additions within a MAJOR: a binding written against MINOR m loads any library with MINOR >= m.
 * History: 1 (rounds 1-3, never bumped while entry points were added); 200 (round 4): the measurement-only entry points
 * (grapes_debug_*) left the product library for the diagnostic build (GRAPES_DIAG), the rider entry points were added, and the
 * product library stopped reading GRAPES_* environment switches;
 * 205: step chains (grapes_graph_chain_*); 206: grapes_linear_bwd_weight_gathered_split_multi;
 * 300 (round 5): grapes_draw_finish_args grew a field (stats_blocks) and grapes_sampler_hist_words() words now include the one-launch
 * draw's barrier words behind the histogram (callers that size d_hist by that call need no change; a binding that mirrors the struct
 * does); added: grapes_gumbel_topk_deferred_ext, grapes_frontier_expand_fused_ext; REMOVED (each a special case of an entry point that stays:
 * profiles/r05_entry_point_census.txt): grapes_frontier_expand_fused_counted / _finish (-> _ext), grapes_gumbel_topk_deferred (-> _deferred_ext),
 * grapes_linear_bwd_weight_bits_multi / _pair (-> _multi_cols / _pair_cols with dw_cols = 0), grapes_sampler_head_bwd_multi (-> _multi_phase, phase 0),
 * grapes_gate_bits_words;
 * 301: grapes_linear_fwd_row_scaled (full-batch inference: the dinv row scaling in the transform GEMM's epilogue);
 * 302: grapes_eval_predict. */
#define GRAPES_ABI_VERSION 302

#define GRAPES_EINVAL (-1)   /* bad size / NULL pointer / unsupported shape */
#define GRAPES_EALIGN (-2)   /* pointer not aligned as required */

#define GRAPES_STATUS_EDGE_OVERFLOW 1   /* frontier produced more edges than e_cap */
#define GRAPES_STATUS_NODE_OVERFLOW 2   /* compaction produced more nodes than n_cap */
#define GRAPES_STATUS_BAD_INDEX 4       /* an index was outside its table */
#define GRAPES_STATUS_SYNC_TIMEOUT 8    /* a one-launch scan gave up waiting for another workgroup (results invalid) */
/* `sync` arguments: GRAPES_SYNC_WORDS 64-bit words of caller memory, ZERO before the first use; the kernels leave it
 * zero.  It carries the workgroup totals of the one-launch ordered scans; launches that share one must be
 * stream-ordered.  NULL selects the two-launch form of the same operation. */
#define GRAPES_SYNC_WORDS 256

typedef void* grapes_stream_t; /* hipStream_t */

int grapes_abi_version(void);
/* "gfx950" — the only architecture the code object is built for. */
const char* grapes_target_arch(void);
/* "product" (libgrapes_hip.so: one configuration, no environment switches, no measurement-only entry points) or "diag"
 * (libgrapes_hip_diag.so, built with -DGRAPES_DIAG: the same kernels + the A/B switches and probes profiles/ uses). */
const char* grapes_build_flavor(void);

/* ------------------------------------------------------------------ A4: TensorMap
 * modules/utils.py:115-117  map_tensor[keys] = arange(len(keys)) */
int grapes_tensormap_update(int32_t* map, const int32_t* keys, int32_t n, const int32_t* d_n,
                            grapes_stream_t stream);
/* modules/utils.py:119-120  out = map_tensor[keys]  (also the edge relabel of main.py:195,254) */
int grapes_tensormap_map(const int32_t* map, const int32_t* keys, int32_t* out, int64_t n,
                         const int32_t* d_n, grapes_stream_t stream);

/* ------------------------------------------------------------------ A1: get_neighborhoods
 * modules/utils.py:74-82.  Two launches: offsets (row lengths + exclusive scan), then expand.
 * eoff[m+1]: eoff[i] = first output slot of nodes[i]; eoff[m] = e.  d_e_out receives e. */
int grapes_frontier_offsets(const int64_t* rowptr, const int32_t* nodes, int32_t m,
                            const int32_t* d_m, int32_t* eoff, int32_t* d_e_out,
                            grapes_stream_t stream);
/* src[t] = nodes[i] (queried node, global id), dst[t] = neighbour; order = query order then
 * ascending column.  src_pos (optional) = i.  Edges beyond e_cap are dropped and
 * GRAPES_STATUS_EDGE_OVERFLOW is raised. */
int grapes_frontier_expand(const int64_t* rowptr, const int32_t* col, const int32_t* nodes,
                           int32_t m, const int32_t* d_m, const int32_t* eoff, int32_t e_cap,
                           int32_t* src, int32_t* dst, int32_t* src_pos, int32_t* status,
                           grapes_stream_t stream);

/* Both steps in ONE launch for m <= 2048 queried nodes and e_cap < 2^23 - 1 (the step's <= B + K previous nodes): every workgroup
 * rebuilds the short row-length scan itself; eoff[m+1] and *d_e_out are published as by grapes_frontier_offsets. */
/* mark_bits (optional; with num_nodes): the expansion also does the hop's marks of grapes_bitmap_mark_hop below — queried
 * nodes -> mark_prev_bits (may be NULL), queried nodes with at least one edge and every neighbour -> mark_bits.
 * remark (optional, host struct): grapes_slice_remark (A3 below) in the same launch — its clear_bits must be a DIFFERENT
 * bitmap than mark_prev_bits (the step alternates two previous-node bitmaps from hop to hop). */
typedef struct {
    int32_t* mult;
    const int32_t* unmark_ids; int32_t n_unmark; const int32_t* d_n_unmark;
    const int32_t* mark_ids; int32_t n_mark; const int32_t* d_n_mark;
    uint64_t* clear_bits; const int32_t* clear_ids; int32_t n_clear; const int32_t* d_n_clear;
} grapes_slice_remark_args;
int grapes_frontier_expand_fused(const int64_t* rowptr, const int32_t* col, const int32_t* nodes, int32_t m,
                                 const int32_t* d_m, int32_t e_cap, int32_t* eoff, int32_t* d_e_out,
                                 int32_t* src, int32_t* dst, int32_t* status, uint64_t* mark_prev_bits,
                                 uint64_t* mark_bits, int32_t num_nodes, const grapes_slice_remark_args* remark,
                                 const int32_t* count_mult, int32_t* count_bsum, int32_t* slice_stage,
                                 grapes_stream_t stream);
/* slice_stage (optional, with count_mult; grapes_slice_stage_words(e_cap) int32 words, no clearing needed): the WHOLE edge side
 * of slice_adjacency (modules/utils.py:85-95, call main.py:241-243) over the edges this launch produces.  With W = ceil(e_cap/64):
 *   stage[wb]            number of surviving edges among the 64 edges t in [64 wb, 64 wb + 64)      (wb < ceil(e / 64))
 *   stage[W + wb]        their summed multiplicity count_mult[dst[t]]
 *   stage[2W + 64 wb + i], stage[2W + e_cap + ...], stage[2W + 2 e_cap + ...]   src / dst / multiplicity of the i-th survivor
 * in edge order.  grapes_gcn_prepare_small_batch(slice_stage = ...) assembles the filtered edge list from it inside the
 * classifier's graph build: no grapes_slice_filter launch. */
size_t grapes_slice_stage_words(int32_t e_cap);
/* count_mult + count_bsum (optional): the first half of grapes_slice_filter over the edges this launch produces —
 * count_bsum[t / 1024] += count_mult[dst[t]] (count_bsum zero on entry: the workspace of a later
 * grapes_slice_filter(counted = 1)); `remark` must then not re-mark count_mult (its clear part is fine: the two id lists
 * belong in grapes_frontier_compact's `remark`, one launch earlier). */

/* ------------------------------------------------------------------ A8 (K2/K3): frontier compaction
 * main.py:183-195.  Replaces the reference's O(N) boolean masks (one byte per node, rebuilt and scanned per hop)
 * by a bitmap over node ids, bits[(N+63)/64] (all-zero at rest; the compaction clears what it consumes): N/8 bytes
 * streamed twice per hop.  `bits1` (a summary level used by earlier versions) is optional everywhere and ignored
 * by the compaction; pass NULL.
 * mark: set the bits of ids[0..n).  */
int grapes_bitmap_mark(uint64_t* bits, uint64_t* bits1 /* may be NULL: level-0 only */,
                       const int32_t* ids, int64_t n, const int32_t* d_n, int32_t num_nodes,
                       int32_t* status, grapes_stream_t stream);
/* mark_rows: set the bit of nodes[i] for every queried node with at least one out-edge
 * (eoff[i+1] > eoff[i], eoff from grapes_frontier_offsets): the source endpoints of main.py:186, one
 * atomic per node instead of one per edge. */
int grapes_bitmap_mark_rows(uint64_t* bits, uint64_t* bits1, const int32_t* nodes, int32_t m,
                            const int32_t* d_m, const int32_t* eoff, int32_t num_nodes,
                            int32_t* status, grapes_stream_t stream);
/* clear: zero the words holding ids[0..n) (level-0 only bitmaps, e.g. the `previous` set). */
int grapes_bitmap_clear(uint64_t* bits, const int32_t* ids, int64_t n, const int32_t* d_n,
                        grapes_stream_t stream);
size_t grapes_frontier_compact_workspace_bytes(int32_t n_cap, int32_t num_nodes);
/* Emits, in ASCENDING GLOBAL ID order (main.py:189-190):
 *   batch_nodes[0..nb)      = ids set in `bits`
 *   neighbor_nodes[0..nn)   = ids set in `bits` and not in `prev_bits`      (main.py:187)
 *   nb_local[0..nn)         = rank of each neighbour inside batch_nodes     (main.py:213)
 *   node_map[id] = rank     for every batch node (main.py:194; node_map may be NULL)
 *   counts[0] = nb, counts[1] = nn.
 * Consumes (zeroes) `bits`/`bits1`; `prev_bits` (may be NULL) is left untouched. */
/* ind_code (optional): also sets indicator bit `ind_bit` of every emitted neighbour (main.py:191), with the
 * epoch convention of grapes_indicator_mark. */
int grapes_frontier_compact(uint64_t* bits, uint64_t* bits1, const uint64_t* prev_bits,
                            int32_t num_nodes, int32_t n_cap, int32_t* batch_nodes,
                            int32_t* neighbor_nodes, int32_t* nb_local, int32_t* node_map,
                            int32_t* counts, uint32_t* ind_code, uint32_t epoch, const uint32_t* d_epoch,
                            int32_t ind_bit, int32_t* cand_pos, void* zero_a, size_t zero_a_words, void* zero_b,
                            size_t zero_b_words, void* zero_c, size_t zero_c_words,
                            const grapes_slice_remark_args* remark, void* workspace, uint64_t* sync,
                            int32_t* status, grapes_stream_t stream);
/* remark (optional): the un-mark / mark id lists of grapes_slice_remark applied by this launch (its clear part is ignored
 * here) — the slice marks of the hop that starts with this compaction. */
/* zero_a / zero_b / zero_c (optional): three ranges of 32-bit words cleared by the same launch — the scratch the NEXT operation wants
 * zeroed (grapes_gcn_prepare with GRAPES_PREP_PREZEROED: its counters and csr_dst; the count_bsum of the hop's expansion), so that it needs no clearing launch. */
/* cand_pos (optional) int32[n_cap]: the inverse of nb_local — position of batch node j in neighbor_nodes, -1 if it is a
 * previous node (used by the sampler's backward pass to write d log_prob / d logit densely). */
/* sync != NULL and at most GRAPES_SYNC_WORDS - 1 workgroups (num_nodes <= 255 * 65536): ONE launch. */
/* ---- The hop graph's degree counting folded into the launches either side of it (main.py:180-195 feeding modules/gcn.py:32's
 * gcn_norm): grapes_gcn_prepare spends two of its four launches on an in-degree histogram over the relabelled edges and a scan
 * of it.  Local ids are assigned in ascending GLOBAL id order, so both can ride on launches that exist anyway:
 *   grapes_frontier_expand_fused_ext(count) per produced edge u -> v (u != v): slot[t] = atomicAdd(indeg[v], 1) — the entry's place
 *       in v's by-target row — and +1 on the in-degree sum of v's bitmap word; per queried node its edge segment (first, length)
 *       and its out-degree on the word sums of the by-source side; self-loops: slot -1, loops[u] += 1.
 *   grapes_frontier_compact_counted        reads the two word sums next to the bitmap words it scans anyway (two more scanned
 *       quantities, a second look-back word), and per emitted node its in-degree (and, for a queried node, its segment): writes
 *       rowptr_t, rowptr_s, dinv, seg_first, row_loops, the long-row items and the edge count — what grapes_gcn_prepare's first
 *       two launches produce — and puts the counters back to zero.
 *   grapes_gcn_prepare_counted             the remaining two launches: entries placed at rowptr_t[dst] + slot (no atomics),
 *       by-source rows from the segments, canonical row order + head records (identical arrays to grapes_gcn_prepare's).
 * All counter tables are indexed by GLOBAL node id, zero at rest and left zero:  indeg int32[N], loops int32[N],
 * seginfo int32[2 N] (written for every queried node, never cleared), wsum int32[2 ceil(N / 64)].  sync2: a second look-back
 * scratch of GRAPES_SYNC_WORDS words (zero at rest).  n_long: the build's counters (words 0, 1 zeroed by the expansion,
 * word 2 = aggregated edges written by the compaction). */
typedef struct {
    int32_t* indeg; int32_t* loops; int32_t* seginfo; int32_t* wsum;
    int32_t* slot;            /* expansion output, int32[e_cap]; NULL: the in-degree atomics return nothing (cursor form below) */
    int32_t* n_long;          /* counters of the build that follows (may be NULL) */
} grapes_hop_count_args;
typedef struct {
    int32_t* indeg; int32_t* loops; const int32_t* seginfo; int32_t* wsum;
    int32_t* rowptr_t; int32_t* rowptr_s; float* dinv; int32_t* seg_first; int32_t* row_loops;   /* [n_cap + 1] x 2, [n_cap] x 3 */
    int32_t* long_items; int32_t* n_long; int32_t item_cap;    /* as grapes_gcn_prepare (may be NULL / NULL / 0) */
    uint64_t* sync2;
    int32_t* cursor;          /* optional int32[n_cap]: a copy of rowptr_t for the CURSOR form of grapes_gcn_prepare_counted (slot = NULL:
                                 entries take their place with an atomic on the row's cursor, as grapes_gcn_prepare's fill does) */
} grapes_hop_degree_args;
/* ... and the END of the draw that produced `nodes` (grapes_gumbel_topk_deferred): the draw's last launch then has no tail — no
 * ticket, no last workgroup — and ONE extra workgroup of this expansion, which runs after it on the stream anyway, adds the
 * draw's log-prob partial sums in the draw's own fixed order into stats[4] and puts the draw-wide histogram back to zero
 * (main.py:276 reads that sum only when the GFlowNet loss is formed).  finish: filled by grapes_gumbel_topk_deferred (host struct,
 * device pointers into that draw's workspace, which must still be alive); NULL: no draw is finished here. */
typedef struct {
    const double* parts_keys;     /* keep-all draws (n <= k): the keys launch's partials, stride 5 doubles, keys_blocks of them */
    const double* parts_emit;     /* exact-k draws: one partial per emit workgroup */
    const uint32_t* sel;          /* sel[2] != 0: the draw kept everything */
    int32_t keys_blocks, emit_block, n_host; const int32_t* d_n;
    float* stats;                 /* stats[4] <- the sum (may be NULL) */
    uint32_t* hist; int32_t hist_words;     /* zeroed (may be NULL / 0) */
    int32_t stats_blocks;         /* > 0 (the one-launch draw): stats[0..3] are formed HERE from stats_blocks per-workgroup partials
                                     (min, max, sum, sum of squares) at parts_keys - 4, stride 5 doubles; 0: the draw wrote them */
} grapes_draw_finish_args;
/* ... and the queried rows' EXTENTS from memory (round 5): node_ext[2 i], node_ext[2 i + 1] = rowptr[nodes[i]], rowptr[nodes[i] + 1],
 * 16-byte aligned, written by whoever wrote `nodes` (grapes_gumbel_topk_deferred_ext for a hop's  cat(targets, kept)  list) — the
 * launch then reads ids, extents and the live count in ONE round trip instead of two dependent ones (modules/utils.py:78: the row
 * lookup of get_neighborhoods).  node_ext_out (optional): the extents this launch worked out itself (node_ext == NULL), for the
 * later lists that begin with the same ids (main.py:236: every hop's list starts with the targets).  NULL / NULL:
 * the plain launch with the draw's end / the degree counting riding in it. */
int grapes_frontier_expand_fused_ext(const int64_t* rowptr, const int32_t* col, const int32_t* nodes, int32_t m,
                                     const int32_t* d_m, int32_t e_cap, int32_t* eoff, int32_t* d_e_out,
                                     int32_t* src, int32_t* dst, int32_t* status, uint64_t* mark_prev_bits,
                                     uint64_t* mark_bits, int32_t num_nodes, const grapes_slice_remark_args* remark,
                                     const int32_t* count_mult, int32_t* count_bsum, int32_t* slice_stage,
                                     const grapes_hop_count_args* count, const grapes_draw_finish_args* finish,
                                     const int64_t* node_ext, int64_t* node_ext_out, grapes_stream_t stream);
int grapes_frontier_compact_counted(uint64_t* bits, uint64_t* bits1, const uint64_t* prev_bits,
                                    int32_t num_nodes, int32_t n_cap, int32_t* batch_nodes,
                                    int32_t* neighbor_nodes, int32_t* nb_local, int32_t* node_map,
                                    int32_t* counts, uint32_t* ind_code, uint32_t epoch, const uint32_t* d_epoch,
                                    int32_t ind_bit, int32_t* cand_pos, void* zero_a, size_t zero_a_words, void* zero_b,
                                    size_t zero_b_words, void* zero_c, size_t zero_c_words,
                                    const grapes_slice_remark_args* remark, void* workspace, uint64_t* sync,
                                    int32_t* status, const grapes_hop_degree_args* degrees, grapes_stream_t stream);
/* edge_src / edge_dst: GLOBAL ids as the expansion wrote them, node_map the compaction's relabel table; tmp_src: int32[e]
 * scratch; head_ids / row_head, prefetch_* as grapes_gcn_prepare_prefetching (helpers ride in the first of the two launches). */
int grapes_gcn_prepare_counted(const int32_t* edge_src, const int32_t* edge_dst, const int32_t* slot, int32_t e,
                               const int32_t* d_e, const int32_t* node_map, int32_t n, const int32_t* d_n,
                               const int32_t* rowptr_t, const int32_t* rowptr_s, const int32_t* seg_first,
                               const int32_t* row_loops, const float* dinv, int32_t* csr_src, int32_t* csr_dst,
                               int32_t* tmp_src, const int32_t* head_ids, int32_t* row_head, int32_t* status,
                               const float* prefetch_X, int64_t prefetch_pitch, int32_t prefetch_row_floats,
                               int32_t* cursor, grapes_stream_t stream);

/* The three marks of one hop in one launch (main.py:183-187): previous -> prev_bits; queried nodes with at least
 * one edge (eoff) and every neighbour dst[0..e) -> bits / bits1. */
int grapes_bitmap_mark_hop(uint64_t* prev_bits, uint64_t* bits, uint64_t* bits1, const int32_t* previous,
                           int32_t m, const int32_t* d_m, const int32_t* eoff, const int32_t* dst, int32_t e,
                           const int32_t* d_e, int32_t num_nodes, int32_t* status, grapes_stream_t stream);
/* Up to four id lists into one bitmap in one launch (main.py:221,252: all_nodes = targets + every hop's samples);
 * a list with n == 0 is skipped.  unmark_mult (optional): also zero the slice multiplicity table (A3 below) at every
 * listed id — the last un-mark of a step whose hops used grapes_slice_remark. */
int grapes_bitmap_mark_lists(uint64_t* bits, uint64_t* bits1, const int32_t* ids0, int32_t n0,
                             const int32_t* d_n0, const int32_t* ids1, int32_t n1, const int32_t* d_n1,
                             const int32_t* ids2, int32_t n2, const int32_t* d_n2, const int32_t* ids3,
                             int32_t n3, const int32_t* d_n3, int32_t num_nodes, int32_t* status,
                             int32_t* unmark_mult, grapes_stream_t stream);

/* all_nodes of a step in ONE launch when the lists are small (main.py:221,252): the ascending, duplicate-free union of up
 * to four id lists with n0 + n1 + n2 + n3 <= 4096 (host capacities) -> out_ids[0..counts[0]); node_map[id] = rank
 * (optional); unmark_mult as in grapes_bitmap_mark_lists.  Same result as marking the lists into a bitmap and compacting it.
 * More than n_cap distinct ids raise GRAPES_STATUS_NODE_OVERFLOW. */
int grapes_union_sorted(const int32_t* ids0, int32_t n0, const int32_t* d_n0, const int32_t* ids1, int32_t n1,
                        const int32_t* d_n1, const int32_t* ids2, int32_t n2, const int32_t* d_n2, const int32_t* ids3,
                        int32_t n3, const int32_t* d_n3, int32_t num_nodes, int32_t n_cap, int32_t* out_ids,
                        int32_t* node_map, int32_t* counts, int32_t* unmark_mult, int32_t* status, grapes_stream_t stream);

/* ------------------------------------------------------------------ A3: slice_adjacency
 * modules/utils.py:85-95.  `mult` is an int32[N] scratch table, all-zero at rest.
 * Step 1 grapes_slice_mark(+1 per entry of cols), step 2 frontier_offsets/expand over `rows`,
 * step 3 grapes_slice_filter keeps edge t mult[dst[t]] times, step 4 grapes_slice_mark(unmark).
 * Output order: row-major over rows, ascending global column id inside a row. */
/* clear_bits (optional): also zero the bitmap words holding cols[0..c) — in the step, `cols` is the hop's
 * previous_nodes list whose prev_bits marks are no longer needed (== grapes_bitmap_clear in the same launch). */
int grapes_slice_mark(int32_t* mult, const int32_t* cols, int32_t c, const int32_t* d_c,
                      int32_t unmark, uint64_t* clear_bits, grapes_stream_t stream);
/* Steps 4 (of hop h-1) and 1 (of hop h) in one launch: previous_nodes goes from targets + samples(h-1) to targets +
 * samples(h) (main.py:236-247), and a hop samples only nodes outside its previous_nodes, so the list to un-mark (the
 * older samples) and the list to mark (the newer ones) are disjoint from each other and from the targets, whose marks
 * stay.  clear_ids/clear_bits: the bitmap words of a third list are zeroed as in grapes_slice_mark.  Any list may be
 * empty (n == 0).  The marks left at the end of the step are removed by grapes_bitmap_mark_lists(unmark_mult). */
int grapes_slice_remark(int32_t* mult, const int32_t* unmark_ids, int32_t n_unmark, const int32_t* d_n_unmark,
                        const int32_t* mark_ids, int32_t n_mark, const int32_t* d_n_mark, uint64_t* clear_bits,
                        const int32_t* clear_ids, int32_t n_clear, const int32_t* d_n_clear, grapes_stream_t stream);
size_t grapes_slice_filter_workspace_bytes(int32_t e_cap);
/* sync (optional, see GRAPES_SYNC_WORDS): one launch instead of two for e <= 255 * 4096. */
int grapes_slice_filter(const int32_t* mult, const int32_t* src, const int32_t* dst, int32_t e,
                        const int32_t* d_e, int32_t out_cap, int32_t* out_src, int32_t* out_dst,
                        int32_t* d_out_count, void* workspace, uint64_t* sync, int32_t counted, int32_t* status,
                        grapes_stream_t stream);
/* counted != 0: `workspace` already holds the survivor count of every 1024-edge block (grapes_frontier_expand_fused's
 * count_bsum): only the emitting launch runs. */

/* ------------------------------------------------------------------ A8 (K4): feature gather
 * main.py:168,191,199-204.  ind_code[N] packs (epoch << 8 | indicator bits); a node whose epoch
 * differs from `epoch` has all indicators 0, so nothing is zeroed per batch (main.py:167). */
/* epoch: host value, or *d_epoch when d_epoch != NULL (a captured hipGraph replays with a new epoch).
 * advance_epoch != 0 (needs d_epoch): a new batch starts (main.py:167) — the marks carry *d_epoch + 1 and the kernel
 * stores that value back, so the counter needs no launch of its own. */
int grapes_indicator_mark(uint32_t* ind_code, const int32_t* ids, int32_t n, const int32_t* d_n,
                          uint32_t epoch, uint32_t* d_epoch, int32_t bit, int32_t advance_epoch,
                          grapes_stream_t stream);
/* The first launch of a captured step that feeds itself (main.py:126,157-168 without the host): targets[0..batch) =
 * ids[start..start+batch) with start = ((*d_cursor * stride + offset) * batch) mod max(1, n_ids - batch) — the unshuffled
 * sequential chunks of the reference's DataLoader, a different stripe per rank —, then *d_cursor += 1; with ind_code: the
 * grapes_indicator_mark(advance_epoch) of those targets; with totals: totals[j] += counters[j * counter_stride] (the
 * previous step's per-graph edge counters, about to be overwritten), skipped while *d_cursor == 0. */
int grapes_step_begin(uint32_t* ind_code, uint32_t* d_epoch, int32_t bit, const int32_t* ids, int32_t n_ids,
                      int32_t* d_cursor, int32_t stride, int32_t offset, int32_t batch, int32_t* targets,
                      const int32_t* counters, int32_t counter_stride, int32_t n_counters, int64_t* totals,
                      grapes_stream_t stream);
/* out[i, 0:F] = X[ids[i], :], out[i, F+j] = indicator j of ids[i]  (num_ind may be 0). */
int grapes_gather_rows(const float* X, int32_t F, const int32_t* ids, int32_t n,
                       const int32_t* d_n, const uint32_t* ind_code, uint32_t epoch,
                       const uint32_t* d_epoch,
                       int32_t num_ind, float* out, grapes_stream_t stream);

/* ------------------------------------------------------------------ A6: gcn_norm + per-hop CSRs
 * PyG gcn_norm (torch_geometric 2.5.2, not in the reference tree): self-loops are
 * dropped, one unit self-loop per node is implied, deg = in-degree on the target index + 1,
 * dinv = deg^-1/2.  Builds the CSR by TARGET (rowptr_t/csr_src: forward aggregation) and by
 * SOURCE (rowptr_s/csr_dst: backward), neighbour ids ascending inside each row (deterministic
 * summation order). */
/* flags: GRAPES_PREP_SRC_GROUPED — the edge list is in the order frontier_expand / slice_filter
 * emit (one contiguous segment per source, destinations ascending inside it): source degrees come
 * from the segment bounds and the by-source CSR is written directly — no same-address atomics and
 * no sorting on hub sources.  A list that is not grouped raises GRAPES_STATUS_BAD_INDEX.
 * long_items / n_long (optional, both or neither): work items for rows with more than
 * GRAPES_LONG_ROW entries.  An item is the int32 pair (row, chunk) = GRAPES_LONG_ROW consecutive
 * entries of that row; the items of one row occupy consecutive slots in chunk order.
 * long_items = int32[2][cap][2] (first half: by-target rows, second half: by-source rows,
 * cap = grapes_gcn_long_items_capacity(e)); n_long is int32[4]: [0],[1] = number of items per half,
 * [2] = number of aggregated (non-self-loop) edges, [3] reserved. */
/* head_ids / row_head (optional, both or neither): head_ids[local id] = row of the resident feature matrix (the hop's
 * batch_nodes); row_head = int32[n][12] receives one record per by-target row
 *   { len, head_ids[r], dinv[r]^2, dinv[r], (head_ids[src_j], dinv[src_j]*dinv[r]) for the first 4 entries }
 * (floats stored by bit pattern; unused entry slots = (head_ids[r], 0)) for grapes_gcn_aggregate_gather_fwd. */
#define GRAPES_PREP_SRC_GROUPED 1
#define GRAPES_PREP_PREZEROED 2   /* see grapes_gcn_prepare_zero_words */
#define GRAPES_LONG_ROW 64
size_t grapes_gcn_prepare_workspace_bytes(int32_t n_cap, int32_t e_cap);
int32_t grapes_gcn_long_items_capacity(int32_t e_cap);
/* node_map (optional): edge_src / edge_dst hold GLOBAL ids and are relabelled through it first — the TensorMap
 * lookup of main.py:195,254 (local_neighborhoods / local_edge_indices) folded into the build.
 * Graphs of at most 2048 nodes in grouped mode are built by ONE workgroup in one launch.
 * GRAPES_PREP_PREZEROED: an earlier launch of the caller (grapes_frontier_compact's zero_* arguments) has zeroed the first
 * grapes_gcn_prepare_zero_words(n) 32-bit words of `workspace` and, in grouped mode, csr_dst[0..e): the build then skips
 * its own clearing launch and applies node_map inside its per-edge kernels.
 * sync (optional, see GRAPES_SYNC_WORDS): the row-pointer scan of larger graphs (n <= 255 * 1024) takes one launch
 * instead of two. */
size_t grapes_gcn_prepare_zero_words(int32_t n);
int grapes_gcn_prepare(const int32_t* edge_src, const int32_t* edge_dst, int32_t e,
                       const int32_t* d_e, const int32_t* node_map, int32_t n, const int32_t* d_n, int32_t flags,
                       int32_t* rowptr_t, int32_t* csr_src, int32_t* rowptr_s, int32_t* csr_dst,
                       float* dinv, int32_t* long_items, int32_t* n_long, const int32_t* head_ids,
                       int32_t* row_head, void* workspace, uint64_t* sync, int32_t* status, grapes_stream_t stream);
/* grapes_gcn_prepare that ALSO touches the feature rows  prefetch_X[head_ids[r], 0:prefetch_row_floats]  (row pitch prefetch_pitch
 * floats; general path with head records) from extra workgroups of its first launch — the rows the hop's gather-SpMM
 * (grapes_gcn_aggregate_gather_fwd; reference main.py:199-204 + modules/gcn.py:32) reads next — so that they are Infinity-Cache
 * hits by then.  prefetch_X == NULL: exactly grapes_gcn_prepare.  No state is kept between calls. */
int grapes_gcn_prepare_prefetching(const int32_t* edge_src, const int32_t* edge_dst, int32_t e, const int32_t* d_e,
                                   const int32_t* node_map, int32_t n, const int32_t* d_n, int32_t flags,
                                   int32_t* rowptr_t, int32_t* csr_src, int32_t* rowptr_s, int32_t* csr_dst,
                                   float* dinv, int32_t* long_items, int32_t* n_long, const int32_t* head_ids,
                                   int32_t* row_head, void* workspace, uint64_t* sync, int32_t* status, const float* prefetch_X,
                                   int64_t prefetch_pitch, int32_t prefetch_row_floats, grapes_stream_t stream);

/* count (<= 8) small graphs over the SAME n <= 2048 nodes (the classifier's per-layer sampled subgraphs,
 * main.py:252-256), grouped edge lists, in ONE launch (one workgroup per graph).  The pointer arguments are HOST arrays
 * of device pointers; workspaces[i] as for grapes_gcn_prepare(n, e[i]). */
int grapes_gcn_prepare_small_batch(int32_t count, const int32_t* const* edge_src, const int32_t* const* edge_dst,
                                   const int32_t* e, const int32_t* const* d_e, const int32_t* node_map,
                                   int32_t n, const int32_t* d_n, int32_t* const* rowptr_t,
                                   int32_t* const* csr_src, int32_t* const* rowptr_s, int32_t* const* csr_dst,
                                   float* const* dinv, int32_t* const* long_items, int32_t* const* n_long,
                                   const int32_t* head_ids, int32_t* const* row_head, void* const* workspaces,
                                   const int32_t* const* slice_stage, const int32_t* const* d_fe, const int32_t* fe_cap,
                                   int32_t* status, grapes_stream_t stream);
/* slice_stage (optional host array; entry i may be NULL): graph i's edge list is first ASSEMBLED, by its workgroup, from the
 * stage a grapes_frontier_expand_fused(slice_stage = ...) launch left (fe_cap[i] = that launch's e_cap, d_fe[i] = its edge
 * count): edge_src[i] / edge_dst[i] (capacity e[i]) and *d_e[i] are then OUTPUTS of this call — the list and count
 * grapes_slice_filter would have produced — before they are read as the build's input. */

/* Full-graph variant (evaluation over the whole adjacency, eval.py:47-70): `rowptr` is an int32 CSR
 * by target with ascending columns and NO self-loops; only dinv and the hub-row work items are computed. */
int grapes_gcn_prepare_from_csr(const int32_t* rowptr, int32_t n, float* dinv, int32_t* items,
                                int32_t* n_items, int32_t item_cap, grapes_stream_t stream);

/* ------------------------------------------------------------------ A7: GCNConv arithmetic
 * H = X Wᵀ (GCNConv.lin, no bias) — fp32 MFMA (v_mfma_f32_32x32x2_f32), exact fp32 fma chain. */
int grapes_linear_fwd(const float* x, const float* w, float* h, int32_t n, const int32_t* d_n,
                      int32_t f_in, int32_t f_out, grapes_stream_t stream);
/* H = diag(row_scale) X Wᵀ — the transform of a full-batch inference layer (modules/gcn.py:32 via eval.py:50; N1) with
 * grapes_scale_rows' pass in the GEMM's epilogue: the rows leave the launch pre-scaled by dinv, the operand
 * grapes_gcn_aggregate_fwd_prescaled reads.  Bit-identical to grapes_linear_fwd followed by grapes_scale_rows; f_out > 1. */
int grapes_linear_fwd_row_scaled(const float* x, const float* w, const float* row_scale, float* h, int32_t n,
                                 const int32_t* d_n, int32_t f_in, int32_t f_out, grapes_stream_t stream);
size_t grapes_linear_bwd_weight_workspace_bytes(int32_t n_cap, int32_t f_in, int32_t f_out);
/* dW[f_out,f_in] (+)= dHᵀ X  (split over rows, slabs reduced in fixed order: deterministic). */
int grapes_linear_bwd_weight(const float* dh, const float* x, float* dw, int32_t n,
                             const int32_t* d_n, int32_t f_in, int32_t f_out, int32_t accumulate,
                             void* workspace, grapes_stream_t stream);
/* Aggregate-first form of a GCNConv whose input needs no gradient and has F_in < F_out
 * (out = act((Â X) Wᵀ + b): same result as PyG's transform-then-aggregate up to fp32 rounding, with the
 * SpMM on the narrow side):  forward = GEMM with fused bias + ReLU epilogue. */
int grapes_linear_bias_act_fwd(const float* x, const float* w, const float* bias, int32_t relu,
                               float* out, int32_t n, const int32_t* d_n, int32_t f_in,
                               int32_t f_out, grapes_stream_t stream);
/* The same layer followed by the XW step of a 1-wide GCNConv head (modules/gcn.py:32 with out_channels = 1):
 * head_out[i] = sum_n out[i][n] * head_w[n], summed from the output tiles of the GEMM (one launch) where the bf16x3
 * kernel applies, else computed by a second launch.  The summation order is fixed but is not grapes_linear_fwd's. */
int grapes_linear_bias_act_head_fwd(const float* x, const float* w, const float* bias, int32_t relu, float* out,
                                    const float* head_w, float* head_out, int32_t n, const int32_t* d_n,
                                    int32_t f_in, int32_t f_out, grapes_stream_t stream);
/* Strided-input forms (bf16x3 kernels only; GRAPES_EINVAL where grapes_split_gemm_available() is 0): x rows are x_stride
 * floats apart (x_stride >= f_in, a multiple of 4), so a layer can read the leading f_in columns of a wider matrix that is
 * already there — at hop 0 the log-Z net's first layer (main.py:227: data.x[batch_nodes], the same rows as the sampler
 * net's input minus the indicator columns) reads the feature columns of the sampler net's aggregated input  Â [X | ind]
 * instead of aggregating the same rows again.  head_w / head_out may both be NULL.  The workspace of the dW form is
 * grapes_linear_bwd_weight_gated_workspace_bytes(n, f_in, f_out). */
int32_t grapes_split_gemm_available(int32_t n, int32_t f_in, int32_t f_out);
int grapes_linear_bias_act_head_fwd_strided(const float* x, int32_t x_stride, const float* w, const float* bias,
                                            int32_t relu, float* out, const float* head_w, float* head_out, int32_t n,
                                            const int32_t* d_n, int32_t f_in, int32_t f_out, grapes_stream_t stream);
int grapes_linear_bwd_weight_gated_strided(const float* gate, const float* x, int32_t x_stride, const float* row_scale,
                                           int32_t n, const int32_t* d_n, const float* col_vec, float* dw, float* dbias,
                                           float* dw_head, int32_t f_in, int32_t f_out, int32_t accumulate,
                                           void* workspace, grapes_stream_t stream);
/* Few-row weight gradients of SEVERAL layers with ONE slab reduction (the classifier's backward pass, main.py:267: its
 * layers all see the same ~B + hops*K rows).  grapes_linear_bwd_weight_slabs launches only the partial products of
 * dW = (dout ⊙ [gate > 0])ᵀ x (gate may be NULL) per 128 rows — slabs [ceil(n/128)][f_out * f_in] at `workspace`, and with
 * want_bias the column sums of the gated dout, [ceil(n/128)][f_out], right behind them; GRAPES_EINVAL when (n, f_in, f_out)
 * is not a shape of the few-row kernel (use grapes_linear_bwd_weight / _gated then).  grapes_slab_reduce_sets then sums up
 * to 8 such sets over the same n in slab order: outs[q][i] (+)= sum_z slabs[q][z * counts[q] + i]. */
size_t grapes_linear_bwd_weight_slabs_bytes(int32_t n, int32_t f_in, int32_t f_out);
int grapes_linear_bwd_weight_slabs(const float* dout, const float* gate, const float* x, int32_t n, const int32_t* d_n,
                                   int32_t f_in, int32_t f_out, int32_t want_bias, void* workspace, grapes_stream_t stream);
/* ... and the same slabs launch with the layer's input gradient dx = dout w  (modules/gcn.py:32,36 backward: the weight
 * and input gradients of one GCNConv's linear map read the same dout and do not depend on each other) as a second problem
 * of the SAME launch: workgroups [0, A) run the slab products, the rest the few-row dx GEMM.  w [f_out, f_in], dx [n, f_in].
 * GRAPES_EINVAL when either half's shape is not one of its few-row kernel (call the two entry points separately then). */
int grapes_linear_bwd_weight_slabs_and_input(const float* dout, const float* gate, const float* x, const float* w, float* dx,
                                             int32_t n, const int32_t* d_n, int32_t f_in, int32_t f_out, int32_t want_bias,
                                             void* workspace, grapes_stream_t stream);
int grapes_slab_reduce_sets(int32_t nsets, const float* const* slabs, float* const* outs, const int64_t* counts, int32_t n,
                            const int32_t* d_n, int32_t accumulate, grapes_stream_t stream);
/* Gate-word forms of  layer -> ReLU -> 1-wide head  (reference modules/gcn.py:31-36 with hidden_dims = [H, 1]: the sampler
 * net main.py:112-113,210 and the log-Z net main.py:114,227), for callers whose only use of the hidden activations is that
 * head: the forward pass writes head_out [n] and gate_bits [n][f_out / 32] (bit 16 h + 4 q + u of word [r][c / 32] is
 * act[r][32 (c / 32) + 8 q + 4 h + u] > 0) INSTEAD of the n x f_out activations, and the backward pass forms
 *     dW1 (+)= col_vec ⊙ S,  db1 (+)= col_vec ⊙ T,  dW2 (+)= rowwise <S, w1> + b1 ⊙ T,
 *     S[m][:] = sum_r [bit(r, m)] row_scale[r] x[r][:],   T[m] = sum_r [bit(r, m)] row_scale[r]
 * over up to four row sets that share the weights (what torch autograd computes for d(head)/d(W1, b1, W2) given
 * d head_out = row_scale and W2 = col_vec; w1 / b1 are the layer's current parameters, [f_out][f_in] dense and [f_out]).
 * bf16x3 kernels only: GRAPES_EINVAL where grapes_split_gemm_available(n, f_in, f_out) is 0.  x rows may be strided
 * (x_stride[h] floats, 0 / NULL = dense).  Workspace: grapes_linear_bwd_weight_gated_workspace_bytes(1, f_in, f_out). */
int grapes_linear_relu_head_fwd_bits(const float* x, int32_t x_stride, const float* w, const float* bias,
                                     const float* head_w, uint32_t* gate_bits, float* head_out, int32_t n,
                                     const int32_t* d_n, int32_t f_in, int32_t f_out, grapes_stream_t stream);
/* TWO such layers over the same n rows in one launch (same f_out, f_in / f_in_b in the same multiple of 16): the sampler net's
 * and the log-Z net's first layers at hop 0 (main.py:199-210 and 223-228 read the same batch rows) — each alone leaves most of
 * its launch to its prologue; side by side on half the workgroups each the pair costs one.  Outputs bit-identical to two
 * grapes_linear_relu_head_fwd_bits calls.  GRAPES_EINVAL where either shape is outside the kernel (call them separately). */
int grapes_linear_relu_head_fwd_bits_pair(const float* x, int32_t x_stride, const float* w, const float* bias,
                                          const float* head_w, uint32_t* gate_bits, float* head_out,
                                          const float* x_b, int32_t x_stride_b, const float* w_b, const float* bias_b,
                                          const float* head_w_b, uint32_t* gate_bits_b, float* head_out_b, int32_t f_in_b,
                                          int32_t n, const int32_t* d_n, int32_t f_in, int32_t f_out, grapes_stream_t stream);
/* ... plus ONE more row set that belongs to a different layer of the same f_out (entry nseg of the operand arrays, which then
 * hold nseg + 1 <= 4 entries; its own f_in_b <= f_in, head weight, parameters and gradient buffers): the log-Z net's first
 * layer beside the sampler net's (main.py:287 backpropagates through both) in one GEMM launch on disjoint workgroups and one
 * slab reduction.  Same workspace. */
/* The two entries above with dw in the PARAMETER's layout: dw is [f_out, dw_cols], f_in - 3 <= dw_cols <= f_in, when f_in is the
 * layer's input width rounded up to a multiple of 4 (ogbn-arxiv: 128 features + 3 indicators = 131 -> 132) and x / w1 carry the
 * zero pad column — the slab sum writes the parameter's gradient itself instead of a padded buffer that a strided copy_ then
 * publishes (modules/gcn.py:32 backward).  dw_cols = 0 means f_in.  A row set is addressed with 32-bit byte offsets from its base:
 * GRAPES_EINVAL when n_cap[h] * x_stride[h] * 4 >= 2^32 (a 4 GiB operand; the hop capacities of BASELINE's configs are 2 - 3 orders
 * of magnitude below). */
int grapes_linear_bwd_weight_bits_multi_cols(int32_t nseg, const uint32_t* const* gate_bits, const float* const* x,
                                             const int32_t* x_stride, const float* const* row_scale,
                                             const int32_t* const* d_n, const int32_t* n_cap, const float* col_vec,
                                             const float* w1, const float* b1, float* dw, int32_t dw_cols, float* dbias,
                                             float* dw_head, int32_t f_in, int32_t f_out, int32_t accumulate, void* workspace,
                                             grapes_stream_t stream);
int grapes_linear_bwd_weight_bits_pair_cols(int32_t nseg, const uint32_t* const* gate_bits, const float* const* x,
                                            const int32_t* x_stride, const float* const* row_scale,
                                            const int32_t* const* d_n, const int32_t* n_cap, const float* col_vec,
                                            const float* w1, const float* b1, float* dw, int32_t dw_cols, float* dbias,
                                            float* dw_head, int32_t f_in, const float* col_vec_b, const float* w1_b,
                                            const float* b1_b, float* dw_b, float* dbias_b, float* dw_head_b, int32_t f_in_b,
                                            int32_t f_out, int32_t accumulate, void* workspace, grapes_stream_t stream);
/* backward of the same layer in ONE split-K GEMM: dW (+)= (dout ⊙ [gate > 0])ᵀ x,
 * dbias (+)= column sums of the gated dout (gate = the layer's ReLU output, or NULL; dbias may be NULL). */
size_t grapes_linear_bwd_weight_gated_workspace_bytes(int32_t n_cap, int32_t f_in, int32_t f_out);
/* row_scale/col_vec (both or neither): dout is the rank-1 matrix row_scale[r]·col_vec[m] (the gradient a
 * 1-wide head sends back, dh2 ⊗ w2) and is formed while loading — `dout` itself is then ignored.
 * dw_head (optional, rank-1 mode only): dw_head[m] (+)= Σ_r row_scale[r]·gate[r][m] = the head's own weight gradient
 * dh2ᵀ·act (gate = the ReLU output act >= 0), summed while the gate tiles stream through the same GEMM. */
int grapes_linear_bwd_weight_gated(const float* dout, const float* gate, const float* x, float* dw,
                                   float* dbias, int32_t n, const int32_t* d_n, int32_t f_in,
                                   int32_t f_out, int32_t accumulate, const float* row_scale,
                                   const float* col_vec, float* dw_head, void* workspace, grapes_stream_t stream);
/* The rank-1 form for up to four hops that share the weights, in ONE split-K launch + ONE slab reduction: hop h
 * contributes gate[h], x[h], row_scale[h] with *d_n[h] live rows of n_cap[h] (host arrays of device pointers).
 * Workspace: grapes_linear_bwd_weight_gated_workspace_bytes(1, f_in, f_out) suffices. */
int grapes_linear_bwd_weight_gated_multi(int32_t nseg, const float* const* gate, const float* const* x,
                                         const float* const* row_scale, const int32_t* const* d_n,
                                         const int32_t* n_cap, const float* col_vec, float* dw, float* dbias,
                                         float* dw_head, int32_t f_in, int32_t f_out, int32_t accumulate,
                                         void* workspace, grapes_stream_t stream);
#ifdef GRAPES_DIAG   /* the diagnostic build only (make -C grapes_amd/csrc diag -> libgrapes_hip_diag.so): not a product surface */
/* diagnosis only: forward GEMM with parts switched off (dbg bits: 1 no stores, 2 no operand reloads, 4 no MFMAs) */
int grapes_debug_gemm_fwd(const float* x, const float* w, float* out, int32_t n, int32_t f_in,
                          int32_t f_out, int32_t dbg, grapes_stream_t stream);
/* measurement only (profiles/gather_bound_probe.py): stripped-down gathers over gcn_prepare's head records, pricing the
 * ingredients of grapes_gcn_aggregate_gather_fwd one at a time; results are NOT the product's (rows of <= 1 entry only).
 * variant bits: 1 five row loads (else two), 2 resident looping workgroups (grid_cap), 4 a row per wavefront. */
int grapes_debug_gather_probe(const float* X, int32_t x_stride, const int32_t* row_head, float* out, int32_t n,
                              int32_t f_out, int32_t variant, int32_t grid_cap, grapes_stream_t stream);
/* measurement only (profiles/tsplit_ablation.py): the gathered-operand bf16x3 GEMMs with parts switched off */
int grapes_debug_tsplit_fwd(const float* X, int32_t F, int32_t x_stride, const int32_t* ids, const void* w_image, float* h,
                            int32_t n, int32_t f_out, int32_t dbg, grapes_stream_t stream);
int grapes_debug_tsplit_dw(const float* dh, const float* X, int32_t F, int32_t x_stride, const int32_t* ids, int32_t n,
                           int32_t f_out, void* workspace, int32_t dbg, grapes_stream_t stream);
#endif
/* dX = dH W */
int grapes_linear_bwd_input(const float* dh, const float* w, float* dx, int32_t n,
                            const int32_t* d_n, int32_t f_in, int32_t f_out,
                            grapes_stream_t stream);
/* out[c] = Σ_{r in row c} (dinv[r]·dinv[c])·H[r] + dinv[c]²·H[c] + bias ; optional ReLU
 * (modules/gcn.py:32).  Gather-SpMM over the by-target CSR: one wavefront per destination row,
 * 16 B per lane coalesced row loads, 8 rows in flight; rows longer than GRAPES_LONG_ROW are
 * processed item by item (one workgroup each) into `workspace` and combined in chunk order.
 * long_items / d_n_items: one half of gcn_prepare's item table and its count (or NULL: every row
 * is handled by a single wavefront).  bias may be NULL. */
size_t grapes_gcn_aggregate_workspace_bytes(int32_t item_cap, int32_t f);
/* two 1-wide vectors over the same graph in one launch: out_a = Â h_a + bias_a, out_b = Â h_b + bias_b (the sampler net's
 * and the log-Z net's heads at hop 0: modules/gcn.py:36 on the [H, 1] layers of main.py:210,227); biases may be NULL */
int grapes_gcn_aggregate_narrow_pair(const float* h_a, const float* h_b, const int32_t* rowptr_t, const int32_t* csr_src,
                                     const float* dinv, const float* bias_a, const float* bias_b, float* out_a, float* out_b,
                                     int32_t n, const int32_t* d_n, grapes_stream_t stream);
int grapes_gcn_aggregate_fwd(const float* h, const int32_t* rowptr_t, const int32_t* csr_src,
                             const float* dinv, const float* bias, float* out, int32_t n,
                             const int32_t* d_n, int32_t f, int32_t relu,
                             const int32_t* long_items, const int32_t* d_n_items, int32_t item_cap,
                             void* workspace, grapes_stream_t stream);
/* grapes_gcn_aggregate_fwd that also returns head_out[r] = out[r] . head_w [f] — the X W step of a 1-wide layer that follows
 * (modules/gcn.py:36 applied to main.py:210's [H, 1] layer), taken from the row while it is in registers instead of a launch
 * that reads the n x f activations back.  f > 16, f % 4 == 0, 16-byte aligned rows; every row is walked by its own wavefront.
 * gate_bits (optional; relu != 0, f <= 256): uint32[n][8], the ReLU gates of out — element e of row r is bit e % 32 of
 * gate_bits[8 r + e / 32] — for grapes_gcn_aggregate_bwd_rank1_bits. */
int grapes_gcn_aggregate_fwd_head(const float* h, const int32_t* rowptr_t, const int32_t* csr_src, const float* dinv,
                                  const float* bias, float* out, int32_t n, const int32_t* d_n, int32_t f, int32_t relu,
                                  const float* head_w, float* head_out, uint32_t* gate_bits, grapes_stream_t stream);
/* The same two aggregations (head_w NULL: grapes_gcn_aggregate_fwd's, else grapes_gcn_aggregate_fwd_head's) driven by the graph
 * build's per-row head records taken over LOCAL ids (row_head of grapes_gcn_prepare* with head_ids = 0, 1, ..., n - 1): one
 * dependent round trip per row instead of three, pairs of rows per resident wavefront (modules/gcn.py:32,36 on Reddit-shaped
 * first layers: f = hidden_dim <= 256, f % 4 == 0).  Results are bit-identical to those entry points'. */
int grapes_gcn_aggregate_fwd_rec(const float* h, const int32_t* row_head, const int32_t* rowptr_t, const int32_t* csr_src,
                                 const float* dinv, const float* bias, float* out, int32_t n, const int32_t* d_n, int32_t f,
                                 int32_t relu, const float* head_w, float* head_out, uint32_t* gate_bits, grapes_stream_t stream);
/* Full-batch inference form (eval.py:47-70; N1): the rows of `hs` are PRE-SCALED by their own dinv (grapes_scale_rows), so an
 * aggregated entry needs no gather of dinv[source]:  out[c] = dinv[c] (sum_s hs[s] + hs[c]) + bias (+ReLU).  Same graph
 * arguments as grapes_gcn_aggregate_fwd; f > 16 and a multiple of 4, 16-byte aligned rows.  Rounding differs from the
 * w = dinv[s] dinv[c] form by an ulp or two.  Replaces: the same GCNConv call sites, modules/gcn.py:32,36 via eval.py:50. */
int grapes_gcn_aggregate_fwd_prescaled(const float* hs, const int32_t* rowptr_t, const int32_t* csr_src,
                                       const float* dinv, const float* bias, float* out, int32_t n,
                                       const int32_t* d_n, int32_t f, int32_t relu, const int32_t* long_items,
                                       const int32_t* d_n_items, int32_t item_cap, void* workspace,
                                       grapes_stream_t stream);
/* Backward of  GCNConv (transform first: out = Â (X Wᵀ) + b) -> ReLU -> GCNConv to ONE output  (the sampler / log-Z nets of
 * main.py:112-114 when F_in >= hidden: Reddit, Cora; call sites modules/gcn.py:32,36) given dh2 = Âᵀ d(head output) [n]:
 *   dw2[m] (+)= sum_r dh2[r] act[r][m]       db1[m] (+)= sum_r [act[r][m] > 0] dh2[r] w2[m]
 *   dh[s][:] = sum_{r in out(s)} w_sr dpre[r] + w_ss dpre[s]   with   dpre[r][m] = [act[r][m] > 0] dh2[r] w2[m]
 * — what outer product + ReLU-mask / bias-gradient pass + grapes_gcn_aggregate_bwd computed through two n x f temporaries,
 * with the same products and summation orders (bit-identical results).  act: the layer's ReLU output [n, f]; w2 [f]; f > 16,
 * f % 4 == 0.  by-source CSR and work items as grapes_gcn_aggregate_bwd.  dw2 / db1 may be NULL. */
size_t grapes_gcn_aggregate_bwd_rank1_workspace_bytes(int32_t item_cap, int32_t f);
int grapes_gcn_aggregate_bwd_rank1(const float* act, const float* dh2, const float* w2, const int32_t* rowptr_s,
                                   const int32_t* csr_dst, const float* dinv, float* dh, float* dw2, float* db1,
                                   int32_t accumulate, int32_t n, const int32_t* d_n, int32_t f,
                                   const int32_t* long_items, const int32_t* d_n_items, int32_t item_cap,
                                   void* workspace, grapes_stream_t stream);
/* ... with the gates of the aggregated matrix read from gate_bits (32 bytes per row, written by grapes_gcn_aggregate_fwd_head)
 * instead of from the rows of act: the same products in the same order, bit-identical dh; act still feeds the two column sums
 * (dw2, db1).  f <= 256. */
int grapes_gcn_aggregate_bwd_rank1_bits(const float* act, const uint32_t* gate_bits, const float* dh2, const float* w2,
                                        const int32_t* rowptr_s, const int32_t* csr_dst, const float* dinv, float* dh,
                                        float* dw2, float* db1, int32_t accumulate, int32_t n, const int32_t* d_n,
                                        int32_t f, const int32_t* long_items, const int32_t* d_n_items, int32_t item_cap,
                                        void* workspace, grapes_stream_t stream);
/* grapes_gcn_aggregate_bwd_rank1_bits for up to three INDEPENDENT problems of one width f in the same five launches (the hops of
 * the sampler net and the log-Z net at main.py:287: each launch of a chain is a few dependent round trips long whatever it moves).
 * Arrays are HOST arrays of `count` entries; workspace[q]: grapes_gcn_aggregate_bwd_rank1_workspace_bytes(item_cap[q], f) each.
 * Results equal, bit for bit, the calls one after the other in index order: problems that name the same dw2 / db1 are added in
 * that order, each with its own accumulate flag (dw2[q] and db1[q] are both given or both NULL). */
int grapes_gcn_aggregate_bwd_rank1_bits_multi(int32_t count, const float* const* act, const uint32_t* const* gate_bits,
                                              const float* const* dh2, const float* const* w2, const int32_t* const* rowptr_s,
                                              const int32_t* const* csr_dst, const float* const* dinv, float* const* dh,
                                              float* const* dw2, float* const* db1, const int32_t* accumulate, const int32_t* n,
                                              const int32_t* const* d_n, int32_t f, const int32_t* const* long_items,
                                              const int32_t* const* d_n_items, const int32_t* item_cap, void* const* workspace,
                                              grapes_stream_t stream);
/* hs[r, :] = dinv[r] * h[r, :] (hs may alias h); f a multiple of 4. */
int grapes_scale_rows(const float* h, const float* dinv, float* hs, int64_t n, int32_t f, grapes_stream_t stream);
/* Â · [X | indicators] straight from the resident feature matrix (fuses the feature gather of
 * main.py:199-204 into the aggregation): out[c] = Σ_s w_sc feat(ids[s]) + dinv[c]² feat(ids[c]),
 * feat(v) = [X[v,0:F], indicator bits of v, zero padding];  out is [n, Kp], Kp = F + num_ind rounded up to a multiple of 4
 * (reference widths: F + hops + 1 = 131 on ogbn-arxiv, 605 on Reddit, main.py:111-113).  x_stride = floats between rows of X,
 * a multiple of 4 (0 = F); when F is not a multiple of 4 the resident matrix is a copy padded to x_stride with ZERO columns. */
int grapes_gcn_aggregate_gather_fwd(const float* X, int32_t F, int32_t x_stride, const int32_t* ids,
                                    const uint32_t* ind_code, uint32_t epoch, const uint32_t* d_epoch,
                                    int32_t num_ind, const int32_t* rowptr_t, const int32_t* csr_src,
                                    const float* dinv, const int32_t* row_head, float* out, int32_t n,
                                    const int32_t* d_n, grapes_stream_t stream);
size_t grapes_gcn_aggregate_bwd_workspace_bytes(int32_t item_cap, int32_t f);
/* dpre = dout ⊙ (out > 0) if relu_out != NULL else dout;  dbias (+)= Σ_c dpre[c];
 * dh[r] = Σ_{c in row r of by-source CSR} (dinv[c]·dinv[r])·dpre[c] + dinv[r]²·dpre[r].
 * dpre is materialised in `dpre_buf` [n,f] (may alias dout when the caller owns dout).
 * d_ticket (optional): GRAPES_COLSUM_TICKETS zero words, left zero — with n <= 8192 the ReLU mask + bias gradient pass
 * is then one launch (last-workgroup combine in a fixed order) instead of two; with f <= 256 and no long_items the WHOLE
 * operation is one launch: the mask is applied to the gathered rows on the fly and dpre_buf is NOT written. */
#define GRAPES_COLSUM_TICKETS 16
int grapes_gcn_aggregate_bwd(const float* dout, const float* relu_out, const int32_t* rowptr_s,
                             const int32_t* csr_dst, const float* dinv, float* dpre_buf,
                             float* dh, float* dbias, int32_t accumulate_bias, int32_t n,
                             const int32_t* d_n, int32_t f, const int32_t* long_items,
                             const int32_t* d_n_items, int32_t item_cap, void* workspace,
                             uint32_t* d_ticket, grapes_stream_t stream);

/* ------------------------------------------------------------------ A2: sampler
 * modules/utils.py:13-71.  One launch: keys = log(sigmoid(l)) + Gumbel(u) with the portable
 * fp32 exp/log of oracle/portable_math.py (bit-identical on CPU and GPU), exact top-k by 4-pass
 * radix select (ties -> lowest position), position-ordered compaction, Bernoulli log-prob,
 * sampler statistics.
 *   logits: if logit_index != NULL the candidate's logit is logits[logit_index[i]] (main.py:213)
 *   uniforms: the torch.rand(n) of the reference's Gumbel draw; if NULL, Philox4x32-10 with
 *             (philox_seed, offset) generates them; offset = *d_philox_offset when non-NULL
 *             (and the kernel advances it), else philox_offset.
 *   mode: 0 = Gumbel-top-k (training, utils.py:37-44); 1 = greedy top-k of probs (eval.py:126-130)
 *   n <= k: every candidate is kept, log_prob = logsigmoid(l), no noise consumed (utils.py:31-33).
 * Outputs: mask[n] (1.0 / 0.0), kept_pos[min(n,k)] ascending candidate positions,
 *   kept_ids = candidate_ids[kept_pos] (optional), d_kept_count, log_prob[n] (optional),
 *   keys_out[n] (optional), stats[6] = {min_prob, max_prob, mean_entropy, std_entropy,
 *   sum(log_prob), valid(1/0)} (optional).
 *   union_ids (optional) = [prefix_ids[0:prefix_n] | kept ids], *d_union_count = prefix_n + kept count: the next
 *   hop's batch_nodes = cat(target_nodes, sampled nodes) (main.py:236-238) written by the same launches. */
size_t grapes_sampler_workspace_bytes(int32_t n_cap);
int grapes_gumbel_topk(const float* logits, const int32_t* logit_index, const float* uniforms,
                       uint64_t philox_seed, uint64_t philox_offset, uint64_t* d_philox_offset,
                       int32_t n, const int32_t* d_n, int32_t k, int32_t mode,
                       const int32_t* candidate_ids, float* mask, int32_t* kept_pos,
                       int32_t* kept_ids, int32_t* d_kept_count, float* log_prob, float* keys_out,
                       float* stats, const int32_t* prefix_ids, int32_t prefix_n, int32_t* union_ids,
                       int32_t* d_union_count, void* workspace, grapes_stream_t stream);
/* The same draw with ONE histogram of the order keys' top 12 bits for the whole draw (instead of per-workgroup rows of their top
 * byte): d_hist = grapes_sampler_hist_words() 32-bit words of caller memory, 16-byte aligned, ZERO before the first use and left
 * zero; draws sharing it must be stream-ordered.  The bin of the k-th largest key then holds ~0.3 % of the candidates, and the
 * selection every emit workgroup derives is a short scan + three short passes.  Results are bit-identical to grapes_gumbel_topk.
 * Replaces the same reference lines (modules/utils.py:37-71). */
int32_t grapes_sampler_hist_words(void);
int grapes_gumbel_topk_hist(const float* logits, const int32_t* logit_index, const float* uniforms,
                            uint64_t philox_seed, uint64_t philox_offset, uint64_t* d_philox_offset,
                            int32_t n, const int32_t* d_n, int32_t k, int32_t mode,
                            const int32_t* candidate_ids, float* mask, int32_t* kept_pos,
                            int32_t* kept_ids, int32_t* d_kept_count, float* log_prob, float* keys_out,
                            float* stats, const int32_t* prefix_ids, int32_t prefix_n, int32_t* union_ids,
                            int32_t* d_union_count, void* workspace, uint32_t* d_hist, grapes_stream_t stream);
/* grapes_gumbel_topk_hist whose last launch has NO TAIL: the kept count, the next query list's count, stats[5] and the Philox
 * advance are written by the launch's first workgroup as soon as it knows the selection; the sum of the log-probs (stats[4]) and
 * the histogram's return to zero are LEFT to the caller's next launch — *finish receives what grapes_frontier_expand_fused_ext
 * needs for them (pointers into `workspace`, which must stay alive until that launch has run).  Same results, bit for bit. */
/* grapes_gumbel_topk_deferred that also writes the next query list's ROW EXTENTS beside its ids (round 5): union_ext[2 j], [2 j + 1] =
 * rowptr[union_ids[j]], rowptr[union_ids[j] + 1]  (int64 pairs, 16-byte aligned, prefix_n + min(k, n) of them; prefix_ext: the prefix
 * ids' pairs, copied) for the adjacency `rowptr` the caller expands next — grapes_frontier_expand_fused_ext(node_ext = union_ext) then
 * needs one round trip less.  The draw requests every candidate's pair while it waits at its barrier; only the kept ones are stored.
 * Same draw, bit for bit (modules/utils.py:37-71; main.py:236-238 for the list). */
int grapes_gumbel_topk_deferred_ext(const float* logits, const int32_t* logit_index, const float* uniforms,
                            uint64_t philox_seed, uint64_t philox_offset, uint64_t* d_philox_offset,
                            int32_t n, const int32_t* d_n, int32_t k, int32_t mode,
                            const int32_t* candidate_ids, float* mask, int32_t* kept_pos,
                            int32_t* kept_ids, int32_t* d_kept_count, float* log_prob, float* keys_out,
                            float* stats, const int32_t* prefix_ids, int32_t prefix_n, int32_t* union_ids,
                            int32_t* d_union_count, void* workspace, uint32_t* d_hist, grapes_draw_finish_args* finish,
                            const int64_t* rowptr, const int64_t* prefix_ext, int64_t* union_ext, grapes_stream_t stream);
/* d logits[i] = g · (mask[i] − sigmoid(l_i)),  g = *d_grad_scale (device scalar) × grad_vec[i]
 * (either may be NULL = 1).  If dlogits_index != NULL the result is scattered:
 * dlogits[dlogits_index[i]] = value (destination pre-zeroed by the caller). */
/* sum_out (optional): (+)= Σ of the written values = the bias gradient of the 1-wide head producing these logits;
 * needs partials (fp32[2048] scratch) and d_ticket (a device word that is ZERO at rest; the kernel leaves it zero). */
int grapes_bernoulli_logprob_bwd(const float* logits, const int32_t* logit_index,
                                 const float* mask, const float* grad_vec,
                                 const float* d_grad_scale, float* dlogits, int32_t n,
                                 const int32_t* d_n, float* sum_out, int32_t accumulate_sum,
                                 float* partials, uint32_t* d_ticket, grapes_stream_t stream);
/* Philox4x32-10 uniforms in [0,1): out[i] from counter offset + i/4, lane i%4. */
int grapes_philox_uniform(float* out, int64_t n, uint64_t seed, uint64_t offset,
                          grapes_stream_t stream);

/* ------------------------------------------------------------------ small helpers on the path */
/* out = Σ_{i<n} x[i] / (mean ? n : 1)   (main.py:228 log_z, main.py:276 tot_log_prob) */
int grapes_reduce_sum(const float* x, int32_t n, const int32_t* d_n, int32_t mean, float* out,
                      grapes_stream_t stream);
/* x[i] = value for i < n (count-aware fill; used for d(mean) broadcasts) */
/* sum_out (optional): (+)= n · value, the sum of what was written */
int grapes_fill(float* x, int32_t n, const int32_t* d_n, float value, const float* d_value,
                float scale_by_inv_n, float* sum_out, int32_t accumulate_sum, grapes_stream_t stream);

/* Backward of the sampler net's 1-wide head for up to four hops in TWO launches (the hops' weights are shared, and only
 * the backward passes of different hops are independent of each other):
 *   dlogits[q][r] = d_grad_scale * (mask[q][cand_pos[q][r]] - sigmoid(logits[q][r]))  for candidate rows, 0 otherwise —
 *                   utils.py:71 differentiated, written densely over the hop's *d_n[q] batch rows (no zero fill needed);
 *   sum_out (+)=    the sum of all of them (bias gradient of the head), combined in a fixed order;
 *   dh[q]         = Â_qᵀ dlogits[q]  (by-source CSR of grapes_gcn_prepare: rowptr_s / csr_dst / dinv).
 * Arrays are HOST arrays of `count` device pointers; logits[q] is the hop's [n_cap[q]] logit vector (per batch row),
 * mask[q] its draw (per candidate position).  A segment with mask[q] == NULL is a MEAN head instead (the log-Z net,
 * main.py:228): dlogits[q][r] = d_grad_scale / *d_n[q] on every live row, not part of sum_out; *mean_sum_out = their sum
 * (that head's bias gradient).  workspace: grapes_sampler_head_bwd_multi_workspace_bytes(); d_ticket: one zero word,
 * left zero. */
size_t grapes_sampler_head_bwd_multi_workspace_bytes(void);
/* The same in two calls: phase 1 = the d logits launch only, phase 2 = the aggregation launch only (0 = both, as above).
 * Between them the caller may issue independent launches of its own; each of the two launches carries a pending recorded
 * few-row backward aggregation of the classifier (grapes_gcn_aggregate_bwd while grapes_rider_record_begin is open) as extra
 * columns of its grid — main.py:267 and main.py:287 depend on the losses only, not on each other. */
int grapes_sampler_head_bwd_multi_phase(int32_t count, const float* const* logits, const float* const* mask,
                                        const int32_t* const* cand_pos, const int32_t* n_cap, const int32_t* const* d_n,
                                        const float* d_grad_scale, const int32_t* const* rowptr_s,
                                        const int32_t* const* csr_dst, const float* const* dinv, float* const* dlogits,
                                        float* const* dh, float* sum_out, int32_t accumulate_sum, float* mean_sum_out,
                                        void* workspace, uint32_t* d_ticket, int32_t phase, grapes_stream_t stream);

/* ------------------------------------------------------------------ losses + optimiser update
 * main.py:260,267: loss_c = CrossEntropyLoss (labels: int64 class of every node) or BCEWithLogitsLoss (labels_f:
 * fp32 [N, C]) over rows local_rows[0..B) of logits[n_rows, C], labels looked up at target_ids[b] (global ids);
 * dlogits[n_rows, C] = d loss_c / d logits (zero outside those rows — what loss_c.backward() feeds gcn_c).
 * Exactly one of labels / labels_f is non-NULL.  B <= 4096.  The target rows must be distinct. */
int grapes_classifier_loss(const float* logits, int32_t n_rows, int32_t C, const int32_t* local_rows,
                           const int32_t* target_ids, const int64_t* labels, const float* labels_f,
                           int32_t B, float* dlogits, float* loss_out, grapes_stream_t stream);
/* eval.py:154-155 (mini-batch evaluation, N1): pred[b] = argmax_c logits[node_map[targets[b]], c] — the first largest, a NaN the
 * largest (torch.argmax) — and rows_out[b, :] (optional) = that row.  A target whose map entry is not a row of logits raises
 * GRAPES_STATUS_BAD_INDEX in *status (optional).  Replaces: node_map.map + index_select + argmax (four framework launches). */
int grapes_eval_predict(const float* logits, int32_t n_rows, int32_t C, const int32_t* node_map, const int32_t* targets,
                        int32_t B, int64_t* pred, float* rows_out, int32_t* status, grapes_stream_t stream);
/* main.py:272-282.  hop_stats[h*stats_stride + 4] = sum of hop h's log-probs (the statistics row grapes_gumbel_topk
 * writes); log_z = *log_z_raw - log_z_init (log_z_raw may be NULL = 0).  out4 = {loss_gfn, d loss_gfn / d (log_z or
 * sum log-probs) [= 2·inner for trajectory balance, = -cost for REINFORCE (main.py:279)], log_z, sum log-probs}. */
int grapes_gflownet_loss(const float* log_z_raw, float log_z_init, const float* hop_stats, int32_t hops,
                         int32_t stats_stride, const float* loss_c, float loss_coef, int32_t reinforce,
                         float* out4, grapes_stream_t stream);
/* main.py:259-282 in ONE launch (one workgroup): grapes_classifier_loss with the target rows looked up as
 * node_map[target_ids[b]] (main.py:259), log_z = mean(z_out[0..nz)) - log_z_init (main.py:228; z_out may be NULL = 0;
 * the sum is formed exactly as grapes_reduce_sum forms it) and grapes_gflownet_loss on the loss just computed. */
/* workspace (grapes_step_losses_workspace_bytes(B), 8-byte aligned) + d_ticket (one zero word, left zero): the work is
 * spread over up to 65 workgroups (n_rows <= 65536), results bit-identical to the single-workgroup form that NULLs
 * select. */
size_t grapes_step_losses_workspace_bytes(int32_t B);
int grapes_step_losses(const float* logits, int32_t n_rows, int32_t C, const int32_t* node_map,
                       const int32_t* target_ids, const int64_t* labels, const float* labels_f, int32_t B,
                       float* dlogits, float* loss_out, const float* z_out, int32_t nz, const int32_t* d_nz,
                       float log_z_init, const float* hop_stats, int32_t hops, int32_t stats_stride,
                       float loss_coef, int32_t reinforce, float* out4, const float* loss_extra, void* workspace,
                       uint32_t* d_ticket, grapes_stream_t stream);
/* F.dropout(x, p, training=True) of modules/gcn.py:33,37 on the sampler's Philox stream: element i of the live [n, f] matrix is
 * kept iff philox_uniform(seed, offset, i) >= p (grapes_philox_uniform's generator) and scaled by 1 / (1 - p); keep[i] = 1 / 0.
 * With d_philox_offset the offset is read from the device and advanced by ceil(n f / 4) afterwards.  The backward pass is
 * dx = keep ? dy / (1 - p) : 0.  (The reference draws its mask from torch's generator: parity is statistical there and exact
 * against the oracle's Philox.) */
int grapes_dropout_fwd(const float* x, float* y, uint8_t* keep, int32_t n, const int32_t* d_n, int32_t f, float p,
                       uint64_t philox_seed, uint64_t philox_offset, uint64_t* d_philox_offset, grapes_stream_t stream);
int grapes_dropout_bwd(const float* dy, const uint8_t* keep, float* dx, int32_t n, const int32_t* d_n, int32_t f, float p,
                       grapes_stream_t stream);
/* The regulariser of main.py:260-261,  reg * sum_r var(logits[r, :])  (torch.var: unbiased, over the classes, every row):
 * with `out` the term itself (one float: pass it to grapes_step_losses as loss_extra — it is then part of loss_out and of
 * the GFlowNet cost, main.py:274), with `dlogits` its gradient ADDED to dlogits (after grapes_step_losses has written them):
 * dlogits[r][c] += reg * 2 (logits[r][c] - mean_r) / (C - 1).  Either pointer may be NULL. */
int grapes_logit_var_reg(const float* logits, int32_t n, const int32_t* d_n, int32_t C, float reg, float* out, float* dlogits,
                         grapes_stream_t stream);
/* main.py:268,289: torch.optim.Adam (amsgrad off) for n_tensors tensors in ONE launch.  d_desc = device array of
 *   struct { float* p; const float* g; float* m; float* v; float* step; int64_t n;
 *            double lr, beta1, beta2, eps, weight_decay; int32_t maximize, pad;
 *            float* w_pad; void* img; int32_t K, ld_pad; } (grapes_adam_desc_bytes() each)
 * step = the optimiser's per-tensor step counter (fp32 scalar, as torch keeps it for capturable=True); the launch
 * uses step+1 and advances every distinct counter once.  d_ticket: n_tensors device words, zero at rest.
 * MIRRORS (round 4; both may be NULL): p is a contiguous [rows, K] weight (< 2^31 elements) and the launch writes every updated
 * element also into w_pad [rows, ld_pad] (the zero-padded fp32 copy a first layer with K % 4 != 0 computes on) and / or into img
 * (the bf16x3 split image of grapes_weight_split_image, rows <= 256) — the copies a step used to rebuild with launches of their own
 * follow the weights by themselves; their padding is written once by the caller and never touched. */
int32_t grapes_adam_desc_bytes(void);
int grapes_adam_step(const void* d_desc, int32_t n_tensors, int64_t max_numel, uint32_t* d_ticket,
                     grapes_stream_t stream);
/* The same launch with grapes_slab_reduce_sets folded in (the classifier's few-row weight gradients, main.py:267 -> :268): a
 * tensor whose gradient pointer equals grads[q] first gets  grad (+)= sum of its slabs (slabs[q]: [ceil(n/128)][numel], summed
 * in grapes_slab_reduce_sets' order: bit-identical), written back to the gradient, then the update.  Every grads[q] must be
 * the gradient of one of the n_tensors tensors (the caller checks).  nsets <= 8; n / d_n: the rows the slabs were formed over. */
int grapes_adam_step_slabs(const void* d_desc, int32_t n_tensors, int64_t max_numel, uint32_t* d_ticket, int32_t nsets,
                           const float* const* slabs, const float* const* grads, int32_t n, const int32_t* d_n,
                           int32_t accumulate, grapes_stream_t stream);

/* ------------------------------------------------------------------ 1-D node partition: either side of the RCCL
 * exchanges (distributes main.py:180 get_neighborhoods and main.py:199-204 x[batch_nodes]).
 * Rank r owns global ids [lo, hi) = [bounds[r], bounds[r+1]); its CSR rows are rebased to 0, its columns are
 * global ids.  Every capacity is equal on all ranks and every live count is on the device, so none of these
 * reads a size on the host (they sit between fixed-size all-gather / all-to-all calls).
 *
 * Adjacency rows.  req = the all-gathered query lists, int32[n_peers][cap+1] (cap ids, then the live count).
 * The owner writes, for each peer p, reply[p] = int32[reply_stride]:
 *   [0,cap) row length of query j (0 if not owned / beyond the count), [cap,2cap) offset of that row inside the
 *   slot's column area, [2cap, 2cap+e_slot) the columns (query order, ascending inside a row).
 * eoff int32[n_peers*cap+1] is scratch.  A slot that would exceed e_slot raises GRAPES_STATUS_EDGE_OVERFLOW. */
/* This rank's query message: query[0..n) = ids, query[cap] = min(*d_n, n) (n when d_n is NULL); query[n..cap) is padding
 * and is left as it is. */
int grapes_exchange_pack_query(const int32_t* ids, int32_t n, const int32_t* d_n, int32_t cap, int32_t* query,
                               grapes_stream_t stream);
int grapes_exchange_serve_rows(const int64_t* rowptr_local, const int32_t* col_local,
                               const int32_t* req, int32_t n_peers, int32_t cap, int32_t lo, int32_t hi,
                               int32_t* reply, int64_t reply_stride, int32_t e_slot, int32_t* eoff,
                               int32_t* status, grapes_stream_t stream);
/* Requester: back = the all-to-all'ed replies int32[n_peers][reply_stride]; nodes[cap] (first *d_m live) are this
 * rank's queries.  Output = the contract of grapes_frontier_offsets + grapes_frontier_expand on the full graph:
 * eoff[cap+1], (src, dst)[e_cap] in query order then ascending column, *d_e = edge count.  rowstart[cap] is scratch. */
int grapes_exchange_recv_rows(const int32_t* back, int64_t reply_stride, const int32_t* nodes, int32_t cap,
                              const int32_t* d_m, const int32_t* bounds, int32_t n_peers, int32_t e_cap,
                              int32_t* eoff, int32_t* rowstart, int32_t* src, int32_t* dst, int32_t* d_e,
                              int32_t* status, grapes_stream_t stream);
/* Halo feature rows.  req = all-gathered ASCENDING id lists int32[n_peers][cap+1]; the ids of [lo,hi) form one run of
 * each list; reply[p] = fp32[n_slot][F] holds the rows of peer p's run (GRAPES_STATUS_NODE_OVERFLOW if longer). */
int grapes_exchange_serve_features(const float* X_local, int32_t F, const int32_t* req, int32_t n_peers,
                                   int32_t cap, int32_t lo, int32_t hi, float* reply, int32_t n_slot,
                                   int32_t* status, grapes_stream_t stream);
/* Requester: out[i, 0:F] = row of ids[i] taken from back = fp32[n_peers][n_slot][F], out[i, F+j] = indicator j
 * (same packing as grapes_gather_rows); ids ascending, first *d_n live. */
int grapes_exchange_assemble_features(const float* back, int32_t F, int32_t n_slot, const int32_t* ids,
                                      int32_t n, const int32_t* d_n, const int32_t* bounds,
                                      int32_t n_peers, const uint32_t* ind_code, uint32_t epoch,
                                      const uint32_t* d_epoch, int32_t num_ind, float* out,
                                      grapes_stream_t stream);

/* ---- The same halo rows WITHOUT an exchange: peer shards read in place (replaces the all-gather + all-to-all of
 * grapes_exchange_serve_features / _assemble_features for aggregate-first first layers, main.py:199-204 + modules/gcn.py:32).
 * Every rank maps every other rank's shard of X into its address space once (hipIpc memory handles: the owner's HBM, reached
 * over xGMI by ordinary loads) and the fused gather-SpMM picks the shard of each row it reads from a table of at most
 * GRAPES_MAX_PEER_SHARDS (base pointer, first row) pairs held in registers.  No collective, no request/reply buffers, no
 * copy: a hop's halo costs the rows it touches, once, over the links they live behind.
 *   grapes_peer_export  handle[64] + byte offset of `ptr` inside its allocation, to be sent to the other ranks of the node
 *   grapes_peer_open    maps a peer's allocation into this process (lazily enabling peer access) -> its address of `ptr`
 *   grapes_peer_close   unmaps it (pass what grapes_peer_open returned and the same offset)
 * shard_base / shard_bounds of grapes_gcn_aggregate_gather_fwd_peers are HOST arrays (n_shards pointers; n_shards + 1 ascending
 * first-row numbers, shard_bounds[0] = 0): shard q holds rows [bounds[q], bounds[q + 1]) at pitch x_stride; row_head is
 * required (head records over GLOBAL row ids); everything else as grapes_gcn_aggregate_gather_fwd — results are bit-identical
 * to it on the concatenated matrix. */
#define GRAPES_MAX_PEER_SHARDS 8
#define GRAPES_PEER_HANDLE_BYTES 64
int grapes_peer_export(const void* ptr, void* handle, uint64_t* offset);
int grapes_peer_open(const void* handle, uint64_t offset, void** ptr);
int grapes_peer_close(void* ptr, uint64_t offset);
/* bytes from a mapped peer shard (or anywhere on a device) into local memory, on `stream` — start-up self-checks and tests */
int grapes_peer_copy(void* dst, const void* src, size_t bytes, grapes_stream_t stream);
int grapes_gcn_aggregate_gather_fwd_peers(const float* const* shard_base, const int32_t* shard_bounds, int32_t n_shards,
                                          int32_t F, int32_t x_stride, const int32_t* ids,
                                          const uint32_t* ind_code, uint32_t epoch, const uint32_t* d_epoch,
                                          int32_t num_ind, const int32_t* rowptr_t, const int32_t* csr_src,
                                          const float* dinv, const int32_t* row_head, float* out, int32_t n,
                                          const int32_t* d_n, grapes_stream_t stream);

/* Requester, in-place form: pos[i] = row of ids[i] inside back viewed as fp32[n_peers * n_slot][F] (rows past *d_n: 0) and,
 * with ind_code, code_pos[pos[i]] = ind_code[ids[i]] — grapes_gcn_aggregate_gather_fwd(X = back, ids = pos, ind_code =
 * code_pos, head records built with head_ids = pos) then aggregates the exchanged rows where they arrived.  dist.py. */
int grapes_exchange_halo_positions(const int32_t* ids, int32_t n, const int32_t* d_n, const int32_t* bounds,
                                   int32_t n_peers, int32_t n_slot, const uint32_t* ind_code, int32_t* pos,
                                   uint32_t* code_pos, grapes_stream_t stream);

/* Requester: where the rows of ids[0 .. *d_n) sit among the rows this rank has ALREADY received in this step —
 *   loc[ids[i]] = base + pos[row(i)],   row(i) = node_map[ids[i]]  (node_map != NULL)  or  idx_b[idx_a[i]]
 * with pos = grapes_exchange_halo_positions of the hop's batch and base = the hop's offset in one buffer holding every hop's
 * `back`.  node_map form: batch[0 .. *d_n_batch) are the hop's batch nodes; an id whose map entry does not point at itself there is
 * not a batch row (a target without any edge) and its entry of loc — written once from a replicated side table — is left alone.  all_nodes (main.py:252) are the targets and the kept nodes of the hops — batch rows of earlier fetches — so the
 * classifier's features (main.py:256) are a local gather through loc instead of a fourth request / reply.  loc: int32[N]. */
int grapes_exchange_note_rows(int32_t* loc, const int32_t* ids, int32_t n, const int32_t* d_n, const int32_t* node_map,
                              const int32_t* batch, int32_t n_batch, const int32_t* d_n_batch,
                              const int32_t* idx_a, const int32_t* idx_b, const int32_t* pos, int32_t base, grapes_stream_t stream);

/* A2 + A7: the draw with its logits produced on the way.  logits_out[r] = (Â head_in)[r] + *bias over the hop's n_rows batch
 * rows — the 1-wide last layer of the sampler net (main.py:210; head_in = act · w2ᵀ) — and every batch row that is a
 * candidate (cand_pos[r] >= 0: its position in neighbor_nodes; logit_index = nb_local, both from grapes_frontier_compact)
 * goes straight on to its key.  Two launches (aggregation + keys | selection + emit) instead of three; keys, masks and
 * log-probabilities are bit-identical to grapes_gcn_aggregate_fwd + grapes_gumbel_topk.  Other arguments as grapes_gumbel_topk. */
int grapes_gumbel_topk_from_aggregate(const float* head_in, const int32_t* rowptr_t, const int32_t* csr_src,
                                      const float* dinv, const float* bias, float* logits_out, int32_t n_rows,
                                      const int32_t* d_n_rows, const int32_t* cand_pos,
                                      const int32_t* logit_index, const float* uniforms,
                                      uint64_t philox_seed, uint64_t philox_offset, uint64_t* d_philox_offset, int32_t n,
                                      const int32_t* d_n, int32_t k, int32_t mode, const int32_t* candidate_ids,
                                      float* mask, int32_t* kept_pos, int32_t* kept_ids, int32_t* d_kept_count,
                                      float* log_prob, float* keys_out, float* stats, const int32_t* prefix_ids,
                                      int32_t prefix_n, int32_t* union_ids, int32_t* d_union_count, void* workspace,
                                      grapes_stream_t stream);

/* ------------------------------------------------------------------ A7, first layers with F_in >= F_out (Reddit, Cora)
 * The reference order — transform, then aggregate (PyG GCNConv; modules/gcn.py:32) — with the transform reading its operand
 * feat(ids[r]) = [X[ids[r], 0:F] | indicator bits | 0-padding] (main.py:199-204) through the id list: the gathered matrix is
 * never written.  Kp = F + num_ind rounded up to a multiple of 4; w and dw are [f_out, Kp] (padding columns: ignored / zero).
 * X / x_stride as in grapes_gcn_aggregate_gather_fwd.  fp32 MFMA (v_mfma_f32_32x32x2_f32), fixed summation order. */
size_t grapes_linear_gathered_workspace_bytes(int32_t n_cap, int32_t k_pad, int32_t f_out);
int grapes_linear_fwd_gathered(const float* X, int32_t F, int32_t x_stride, const int32_t* ids,
                               const uint32_t* ind_code, uint32_t epoch, const uint32_t* d_epoch, int32_t num_ind,
                               const float* w, float* h, int32_t n, const int32_t* d_n, int32_t f_out,
                               void* workspace, grapes_stream_t stream);
int grapes_linear_bwd_weight_gathered(const float* dh, const float* X, int32_t F, int32_t x_stride,
                                      const int32_t* ids, const uint32_t* ind_code, uint32_t epoch,
                                      const uint32_t* d_epoch, int32_t num_ind, uint32_t ind_mask, float* dw,
                                      int32_t n, const int32_t* d_n, int32_t f_out, int32_t accumulate,
                                      void* workspace, grapes_stream_t stream);
/* ind_mask (0 = all): the indicator bits that count.  The indicator columns accumulate from hop to hop within a batch
 * (main.py:191), and a backward pass that re-reads them at the END of the step must see what its hop's forward pass saw:
 * bits 0..hop and the target bit (the reference keeps the hop's x tensor alive instead). */

/* The same two GEMMs on the bf16 matrix pipe at fp32 accuracy (exact 3-way bf16 operand splits, six cross products per
 * product, fp32 accumulation; csrc/gemm_tiled_split.hip): 128 x 256 output tiles, both operands streamed through split-plane
 * LDS images in K steps of 32.  The forward takes W as a pre-split IMAGE (grapes_weight_split_image, once per step, from the
 * [f_out, k] parameter itself: no padded copy); f_out <= 256, a multiple of 4.  GRAPES_GEMM_SPLIT=0 disables them
 * (grapes_split_gathered_available). */
int32_t grapes_split_gathered_available(int32_t f_out);
size_t grapes_weight_split_image_bytes(int32_t k);
int grapes_weight_split_image(const float* w, int32_t ldw, int32_t f_out, int32_t k, void* image, grapes_stream_t stream);
/* ... that also writes w_pad [f_out, ld_pad] (k <= ld_pad <= k rounded up to 32), a zero-padded fp32 copy of w for the few-row
 * fp32 kernels, in the same launch (instead of a strided copy of its own per step). */
int grapes_weight_split_image_padded(const float* w, int32_t ldw, int32_t f_out, int32_t k, void* image, float* w_pad,
                                     int32_t ld_pad, grapes_stream_t stream);
/* ... for up to GRAPES_MAX_WEIGHT_IMAGES weights in ONE launch (host arrays of `count` entries; w_pad / ld_pad may be NULL, a
 * w_pad[q] may be NULL): the sampler net's, the log-Z net's and the classifier's first layers (main.py:199-205,227,245) are
 * refreshed together at the top of a step. */
#define GRAPES_MAX_WEIGHT_IMAGES 4
int grapes_weight_split_images(int32_t count, const float* const* w, const int32_t* ldw, const int32_t* f_out, const int32_t* k,
                               void* const* image, float* const* w_pad, const int32_t* ld_pad, grapes_stream_t stream);
/* grapes_linear_bwd_weight_gathered_split with the row pitch of dw given: dw_ld = 0 or ceil4(F + num_ind) — the padded layout —
 * or exactly F + num_ind — the parameter's own [f_out, F + num_ind] gradient (modules/gcn.py:32 backward), written by the slab
 * sum itself instead of a strided copy out of a padded buffer afterwards. */
int grapes_linear_bwd_weight_gathered_split_ld(const float* dh, const float* X, int32_t F, int32_t x_stride,
                                               const int32_t* ids, const uint32_t* ind_code, uint32_t epoch,
                                               const uint32_t* d_epoch, int32_t num_ind, uint32_t ind_mask, float* dw,
                                               int32_t dw_ld, int32_t n, const int32_t* d_n, int32_t f_out,
                                               int32_t accumulate, void* workspace, grapes_stream_t stream);
/* Several of these weight-gradient GEMMs in ONE launch + one slab sum (new design; main.py:271-283's backward reaches the sampler
 * net's first layer once per hop and the log-Z net's once, all over rows of the same X).  Problems q = 0..count-1 (count <= 4): HOST
 * arrays of dh / ids / ind_code / num_ind / ind_mask / dw / dw_ld / n / d_n / accumulate with the meaning of the one-problem entry
 * point.  Problems that name the same dw are summed into it together (same num_ind and layout; accumulate = the first one's).  The
 * slab budget is dealt out on the DEVICE by the live row counts.  Available when grapes_..._multi_available(f_out, k_pad) says so
 * for every problem's k_pad (f_out in (128, 256], k_pad mod 256 in (0, 128]: the tile shape this launch has); workspace:
 * ..._multi_workspace_bytes(distinct dw pointers, largest k_pad, f_out), 16-byte aligned. */
int32_t grapes_linear_bwd_weight_gathered_split_multi_available(int32_t f_out, int32_t k_pad);
size_t grapes_linear_bwd_weight_gathered_split_multi_workspace_bytes(int32_t outputs, int32_t k_pad_max, int32_t f_out);
int grapes_linear_bwd_weight_gathered_split_multi(int32_t count, const float* const* dh, const float* X, int32_t F,
                                                  int32_t x_stride, const int32_t* const* ids,
                                                  const uint32_t* const* ind_code, uint32_t epoch, const uint32_t* d_epoch,
                                                  const int32_t* num_ind, const uint32_t* ind_mask, float* const* dw,
                                                  const int32_t* dw_ld, const int32_t* n, const int32_t* const* d_n,
                                                  int32_t f_out, const int32_t* accumulate, void* workspace,
                                                  grapes_stream_t stream);
int grapes_linear_fwd_gathered_split(const float* X, int32_t F, int32_t x_stride, const int32_t* ids,
                                     const uint32_t* ind_code, uint32_t epoch, const uint32_t* d_epoch, int32_t num_ind,
                                     const void* w_image, float* h, int32_t n, const int32_t* d_n, int32_t f_out,
                                     grapes_stream_t stream);
/* The same transform for ONE or TWO nets over the same gathered rows (nprob = 2: the sampler net and the log-Z net at hop 0,
 * main.py:210 and main.py:227 — the second net finds the rows in L2) with a split tail: on 256 resident workgroups the units
 * (tiles x nets) that do not fill a whole round are cut along K into as many pieces as there are idle workgroups and their partial
 * tiles summed by a second launch, in piece order (Reddit: 597 tiles = three rounds for 2.33 -> 2.4; two nets x 182 tiles = two
 * rounds -> 1.5).  HOST arrays of nprob entries: ind_code / num_ind / w_image / h; the nets' K must cover the same number of
 * 32-wide steps.  A tile that is not cut equals grapes_linear_fwd_gathered_split's bit for bit; a cut tile differs by the order of
 * its K steps' additions (pieces summed in order: deterministic).  workspace: …_tail_workspace_bytes(f_out), 16-byte aligned. */
size_t grapes_linear_fwd_gathered_split_tail_workspace_bytes(int32_t f_out);
int grapes_linear_fwd_gathered_split_tail(const float* X, int32_t F, int32_t x_stride, const int32_t* ids, int32_t nprob,
                                          const uint32_t* const* ind_code, uint32_t epoch, const uint32_t* d_epoch,
                                          const int32_t* num_ind, const void* const* w_image, float* const* h, int32_t n,
                                          const int32_t* d_n, int32_t f_out, void* workspace, grapes_stream_t stream);
/* The same GEMM for FEW rows (the classifier's <= B + hops K rows, a small graph's frontier): 128-row tiles x slabs of K steps
 * on ~256 workgroups + the slabs' sum in index order (two launches) — a handful of tiles would otherwise walk the whole K alone
 * (Cora: 2.7k rows x K = 1436).  workspace: grapes_linear_fwd_gathered_split_k_workspace_bytes(n, ceil4(F + num_ind), f_out). */
size_t grapes_linear_fwd_gathered_split_k_workspace_bytes(int32_t n, int32_t kp, int32_t f_out);
int grapes_linear_fwd_gathered_split_k(const float* X, int32_t F, int32_t x_stride, const int32_t* ids,
                                       const uint32_t* ind_code, uint32_t epoch, const uint32_t* d_epoch, int32_t num_ind,
                                       const void* w_image, float* h, int32_t n, const int32_t* d_n, int32_t f_out,
                                       void* workspace, grapes_stream_t stream);
size_t grapes_linear_bwd_weight_gathered_split_workspace_bytes(int32_t k_pad, int32_t f_out);
int grapes_linear_bwd_weight_gathered_split(const float* dh, const float* X, int32_t F, int32_t x_stride,
                                            const int32_t* ids, const uint32_t* ind_code, uint32_t epoch,
                                            const uint32_t* d_epoch, int32_t num_ind, uint32_t ind_mask, float* dw,
                                            int32_t n, const int32_t* d_n, int32_t f_out, int32_t accumulate,
                                            void* workspace, grapes_stream_t stream);

/* ------------------------------------------------------------------ N3: ingest, edge_index -> CSR
 * main.py:134-136  adjacency = sp.csr_matrix((ones(E, bool), edge_index), (N, N)): duplicate (row, col) pairs collapse, the
 * columns of a row ascend, self-loops stay.  Counting placement (one atomic per edge) + per-row sort / de-duplication in
 * LDS windows (hub rows in place in global memory) + a scan; 64-bit offsets (ogbn-papers100M: 3.2e9 symmetrised edges).
 * edge_src / edge_dst: int64[num_edges] (torch.long edge_index rows); rowptr int64[num_nodes + 1]; col int32 with capacity
 * num_edges; *d_nnz (device) = stored entries.  Ids outside [0, num_nodes) are skipped and raise GRAPES_STATUS_BAD_INDEX.
 * workspace: grapes_csr_build_workspace_bytes, 256-byte aligned.  Not for use inside a stream capture. */
size_t grapes_csr_build_workspace_bytes(int64_t num_edges, int32_t num_nodes);
int grapes_csr_build(const int64_t* edge_src, const int64_t* edge_dst, int64_t num_edges, int32_t num_nodes,
                     int64_t* rowptr, int32_t* col, int64_t* d_nnz, void* workspace, int32_t* status,
                     grapes_stream_t stream);

#ifdef GRAPES_DIAG
/* ------------------------------------------------------------------ pre-split feature planes (round 4; DIAGNOSTIC BUILD ONLY:
 * measured slower than splitting in the K loop — Reddit 1.418 against 1.366 ms/step — and kept as an A/B form)
 * The gathered-operand bf16x3 GEMMs of the transform-first first layers (grapes_linear_fwd_gathered_split[_k],
 * grapes_linear_bwd_weight_gathered_split[_ld]; modules/gcn.py:32 on main.py:199-204's rows) split every gathered fp32 row into
 * its three bf16 terms inside their K loops.  X is constant for a whole run (main.py:66): the caller may split it ONCE —
 * planes: grapes_feature_planes_bytes(n, x_stride) bytes, 24 per 4-column chunk [h0..h3 | m0..m3 | l0..l3], 8-byte aligned —
 * and REGISTER the planes for that matrix (host-side table keyed by the X pointer and row stride; planes NULL forgets it);
 * those entry points then read the planes of the rows they gather.  Results are bit-identical (the same split, done earlier).
 * The planes must be rebuilt when X changes (learned embeddings: do not register). */
size_t grapes_feature_planes_bytes(int64_t n, int32_t x_stride);
int grapes_feature_split_planes(const float* X, int64_t n, int32_t x_stride, void* planes, grapes_stream_t stream);
int grapes_feature_planes_register(const float* X, int32_t x_stride, const void* planes);
#endif

/* ------------------------------------------------------------------ learned node embeddings (--embed_nodes)
 * main.py:89-100,116: data.x is an nn.Parameter optimised by optimizer_c; the backward of `data.x[all_nodes]` (main.py:256)
 * accumulates the rows' gradients into a dense [N, F] gradient.  dst[ids[i], 0:F] (+)= src[i, 0:F]; accumulate = 0 overwrites
 * the addressed rows (the step's id lists are duplicate-free and the gradient was zeroed); atomic != 0 adds with float
 * atomics (id lists with duplicates: the drop-in autograd function). */
int grapes_scatter_rows(float* dst, int64_t dst_stride, const int32_t* ids, const float* src, int64_t src_stride, int32_t F,
                        int32_t n, const int32_t* d_n, int32_t accumulate, int32_t atomic, grapes_stream_t stream);

/* ------------------------------------------------------------------ measurement: kernel clock table
 * bench.py's roofline numbers are taken INSIDE the replayed hipGraph (HIP events cannot bracket a graph node on this ROCm):
 * while a table is enabled, every launch of the roofline kernels (gcn_aggregate_k<4>, gcn_aggregate_gather*_k,
 * gemm_wsplit_f32_k, ...) reserves one (begin, end) pair of 100 MHz s_memrealtime stamps per wavefront — at enqueue / capture
 * time, in launch order — and its wavefronts write them on every execution.  Launch i's duration = (max end - min begin) /
 * rate over its pairs.  With no table (the default, and the timed region of bench.py) no stamp executes.
 * table: device memory of `words` 64-bit words (NULL disables and forgets the launch log). */
int grapes_kernel_clock_enable(uint64_t* table, int64_t words);
int32_t grapes_kernel_clock_launches(void);
/* kernel64: caller buffer of 64 chars (NUL-terminated name); offset_words into the table; pairs = wavefronts of the launch */
int grapes_kernel_clock_entry(int32_t i, char* kernel64, int64_t* offset_words, int32_t* pairs);
int32_t grapes_kernel_clock_rate_khz(void);

/* ------------------------------------------------------------------ riders: two problems side by side in one launch
 * (new design; no reference counterpart — reference main.py:157-291 runs its steps strictly one after the other)
 * The step's small index kernels leave most of the chip idle, and on this stack neither a second stream nor a branch of a
 * hipGraph overlaps them for free (profiles/r04_overlap_probe.txt: a second hardware queue taxes every dispatch of the first).
 * What does overlap is work of the SAME kernel carried as extra workgroups of one launch.  While RECORDING, the launches of
 * grapes_step_begin, grapes_frontier_expand_fused[_ext], grapes_frontier_compact[_counted], grapes_gcn_prepare_counted and
 * grapes_gcn_aggregate_gather_fwd[_peers] are not issued but appended to a PROGRAM (their arguments are kept by value: the
 * buffers must stay alive).  While a program is ATTACHED, the next launch of the same kernel (same variant and workgroup
 * size) by those entry points carries the program's next record as additional workgroups ("host" and "rider" never share
 * scratch: give the rider's compaction its own sync words); records that cannot ride (grapes_step_begin, other kernel
 * variants) are issued on their own, in recorded order, as soon as they are at the head of the program; detach issues what
 * is left.  Every record is issued exactly once per attach, in recorded order.  step_graph uses this to run the NEXT training
 * step's weight-independent prelude (next batch, hop 0's expansion / compaction / graph build / gather-SpMM) inside the
 * current step's hop-1 launches.  Host-side state, one recording / one attachment at a time, not thread-safe. */
int grapes_rider_record_begin(void);
int32_t grapes_rider_record_end(void);                              /* -> program handle (>= 0) or -1 */
int32_t grapes_rider_count(int32_t program);                        /* records in the program, -1: no such program */
/* hold != 0: the early phase — only a recorded grapes_step_begin may be taken (as one more workgroup of the next expansion);
 * everything else waits for grapes_rider_release.  hold == 0: all records may ride from now on (the leading ones that cannot
 * ride are issued at once). */
int grapes_rider_attach(int32_t program, int32_t hold, grapes_stream_t stream);
int grapes_rider_release(grapes_stream_t stream);
/* -> records issued on their own during this attachment (>= 0), or a negative error; *paired = records that rode */
int grapes_rider_detach(grapes_stream_t stream, int32_t* paired);
int grapes_rider_launch(int32_t program, grapes_stream_t stream);   /* the whole program on its own, in order */
int grapes_rider_free(int32_t program);

/* ------------------------------------------------------------------ step chains: several captured steps, one hipGraphLaunch
 * (new design; reference main.py:157 `for batch in loader` launches its steps one by one — here the loop body is a captured
 * hipGraph, and between two hipGraphLaunch calls the queue idles for ~20 us: 4 % of a 0.5 ms step)
 * graphs: n hipGraph_t handles (HOST array; e.g. torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()) of captured steps.
 * The chain is a NEW executable graph holding copies of their kernel (/ memset / empty) nodes with the SAME arguments, segment
 * after segment in the given order, the whole sequence `repeat` times; a segment starts when the one before has finished.
 * The source graphs are not changed and stay launchable; their buffers must outlive the chain.  A graph with any other node
 * kind (copies, host functions, events) -> GRAPES_EINVAL.  *chain_out: opaque handle; *nodes_out (optional): nodes in it. */
int grapes_graph_chain_create(void* const* graphs, int32_t n, int32_t repeat, void** chain_out, int32_t* nodes_out);
int grapes_graph_chain_launch(void* chain, grapes_stream_t stream);
int grapes_graph_chain_destroy(void* chain);

#ifdef __cplusplus
}
#endif
#endif /* GRAPES_HIP_H */
