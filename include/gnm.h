/*
 * gnm.h -- C ABI of libgnm.so, the MI355X (gfx950) GatedGCN edge-logit engine.
 *
 * The reference (lvrcek/GNNome-assembly) has no FFI of its own: its hot path is ~120
 * lines of Python that call torch and DGL (un-vendored).  Each entry point below replaces
 * one group of those implicit kernels; the reference call site is cited per function
 * (paths relative to the reference repository root).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the function says "host";
 *  - all float tensors are fp32, row-major, contiguous unless an explicit leading
 *    dimension (ld*, in elements) is given; all indices are int32;
 *  - the library never allocates, frees or keeps a pointer across calls; scratch space
 *    is a caller-provided workspace (gnm_*_workspace_bytes tells how much);
 *  - kernels are enqueued on `stream` (a hipStream_t passed as void*), no internal
 *    synchronisation, re-entrant;
 *  - return value: 0 ok, <0 invalid argument, >0 a hipError_t; gnm_last_error() returns a
 *    thread-local message for the last non-zero return.
 *  - "internal edge order" = edges stably sorted by destination (gnm_graph_build_index);
 *    all [E,*] tensors inside the layer stack are in that order, the model boundary
 *    (e_raw in, scores out, labels) is in the caller's edge-id order.
 *  - H (hidden width) must be one of 32, 64, 128, 256.
 */
#ifndef GNM_H
#define GNM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNM_ABI_VERSION 7   /* 2: per-call max_blocks_per_cu (edge_bwd_src, node_proj_bwd_tn), locality order; 3: sweep plans, two-sided sweeps; 4: H = 256 fused kernels; 5: LayerNorm entry points take the real width, node_bgrad its row pitch, fused node-side entry points, composite layer entry points; 6: matmul mode 2 (f16x2, the default), the pre-split image entry points (gnm_*_s3) and the Bs argument of gnm_tn128_bgrad removed, gnm_ln_edge_gate2_fwd; 7: gnm_edge_bwd_fused_gt (gt given: the LayerNorm backward's fused edge pass), gnm_ln_edge_bwd_top / gnm_ln_edge_bwd_src_fix (its two-sided sweep), gnm_ln_edge_bwd_chain (its chained form), gnm_debug_set_variant("gate2_wg") */

/* GEMM operand modes: C[M,N] = op(A) * op(B) (+bias +resid, relu) */
#define GNM_GEMM_NT 0 /* A[M,K] row-major, B[N,K] row-major  : y = x W^T   (nn.Linear forward)   */
#define GNM_GEMM_NN 1 /* A[M,K] row-major, B[K,N] row-major  : gx = gy W    (nn.Linear input grad) */
#define GNM_GEMM_TN 2 /* A[K,M] row-major, B[K,N] row-major  : gW = gy^T x  (nn.Linear weight grad)*/

int gnm_abi_version(void);
const char* gnm_last_error(void);
/* number of compute units of the current device (grid sizing / partial-buffer sizing) */
int gnm_num_cus(void);
/* upper bound on the number of per-block partial rows any kernel writes (see *_partials) */
int gnm_max_partial_blocks(void);
/* Process-wide configuration knobs (NOT per-call state; set them before the first launch, never while another
 * thread is inside the library).  Everything a launch needs beyond them travels in its arguments, so the entry
 * points themselves are re-entrant: two host threads, or one host on two streams, may call concurrently.
 *   gnm_set_matmul_mode     (below) which matrix-core arithmetic the fused kernels use;
 *   gnm_set_occupancy_cap   tools/ only: a default cap on workgroups per CU for every persistent kernel
 *                           (0 = none).  A host that co-schedules two kernels on two streams does NOT use it: the
 *                           entry points it needs take the cap per call (max_blocks_per_cu; 0 = none).       */
int gnm_set_occupancy_cap(int blocks_per_cu);

/* ---- graph index (HOST pointers; replaces DGL's lazy CSR/CSC build + dgl.reverse,
 *      layers/gated_gcn_full.py:115, graph_parser.py:297 edge-id order) -------------------
 * perm[j]    = caller edge id stored at internal position j (stable sort by dst)
 * isrc/idst  = endpoints in internal order; in_ptr[N+1] = CSC row pointer over internal order
 * out_ptr[N+1], out_pos[E], out_dst[E]: out-edges grouped by source (ascending internal
 *              position inside a source): internal position and destination of each. */
int gnm_graph_build_index(const int32_t* src, const int32_t* dst, int64_t N, int64_t E,
                          int32_t* perm, int32_t* isrc, int32_t* idst, int32_t* in_ptr,
                          int32_t* out_ptr, int32_t* out_pos, int32_t* out_dst);

/* ---- locality order of the nodes (HOST pointers).  The reference never sorts reads by position
 *      (pipeline.py:46-61,160-169 keep the simulator's read order; graph_parser.py:297-304 numbers nodes by
 *      read id), while the gather kernels rely on node ids that follow the genome (the node rows of the edges in
 *      flight must fit the 4 MB per-XCD L2).  The host renumbers the nodes INTERNALLY, once per graph:
 * edge_locality:  *frac_out = fraction of the edges with |src - dst| <= window in the caller's numbering
 *                 (decides whether a renumbering is needed at all);
 * locality_order: order[new] = old, rank[old] = new: breadth-first over the symmetrised graph restricted to the
 *                 triangle-supported edges (true overlaps are transitive; repeat-induced shortcuts would fold
 *                 the order) when those are at least half of the edges, every component started from a
 *                 pseudo-peripheral node; *core_frac_out (optional) = fraction of triangle-supported edges.   */
int gnm_graph_edge_locality(const int32_t* src, const int32_t* dst, int64_t N, int64_t E, int64_t window,
                            double* frac_out);
int gnm_graph_locality_order(const int32_t* src, const int32_t* dst, int64_t N, int64_t E, int32_t* order,
                             int32_t* rank, double* core_frac_out);

/* ---- sweep plan (HOST pointers): lets ONE destination-sorted sweep also form the by-SOURCE sums the reference takes
 *      on dgl.reverse(g) (gated_gcn_full.py:115,133-143 and their autograd duals) instead of a second pass over the
 *      [E,H] tensors.  Input: the internal index (isrc / idst / in_ptr of gnm_graph_build_index) and the partition of
 *      the sweep kernels (gnm_sweep_partition).  Output, per internal row j:
 *        sinfo[j] != 0 iff row j is the first row of its source inside its tile AND the sweep serves that source:
 *          bits 0-15 tile rows sharing the source | bits 16-21 accumulator slot | bit 22 first tile | bit 23 last tile
 *        dinfo[j]: the same for the row's destination (contiguous rows, two alternating slots);
 *      fix_nodes[0 .. *nfix_out): the nodes the sweep does not serve (out-edges in more than one workgroup, no free
 *      slot, farther than `margin` ids from the workgroup's node range, or no out-edges at all): the *_fix kernels
 *      cover them.  tile_rows <= 16, nslots <= 64; *peak_live_out (optional) = most slots ever in use.
 *      Returns 3 (nothing written, gnm_last_error() says which workgroup) when the node partition gives one workgroup more
 *      rows than the sweep kernels' 32-bit row offsets reach (> 2^21 - 64: a hub-heavy graph): such a graph has no plan and
 *      runs the separate by-source passes.                                                                              */
int gnm_graph_build_sweep_plan(const int32_t* isrc, const int32_t* idst, const int32_t* in_ptr, int64_t N, int64_t E,
                               int64_t nodes_per_block, int tile_rows, int nslots, int64_t margin, uint32_t* sinfo,
                               uint32_t* dinfo, int32_t* fix_nodes, int64_t* nfix_out, int32_t* peak_live_out);
/* The same plan built ON THE DEVICE (device pointers, kernels queued on `stream`, no host synchronisation), bit for bit the
 * words of gnm_graph_build_sweep_plan: for graphs whose index never visits the host (the induced sub-graphs of the mini-batch
 * mode, train.py:288-343).  served[N] (bytes): 1 for the sources the sweep serves; fix_nodes[N] (optional): v for the nodes it does
 * not serve, -1 for the others -- the *_fix kernels skip negative entries, so the list is used as it is with nfix = N.
 * first / last [N] int32: scratch; peak [1] int32 (optional): most slots in use.  tile_rows = 16, nslots <= 64.          */
int gnm_graph_build_sweep_plan_device(const int32_t* isrc, const int32_t* idst, const int32_t* in_ptr, int64_t N, int64_t E,
                                      int64_t nodes_per_block, int tile_rows, int nslots, int64_t margin, uint32_t* sinfo,
                                      uint32_t* dinfo, uint8_t* served, int32_t* fix_nodes, int32_t* first, int32_t* last,
                                      int32_t* peak, void* stream);
/* the partition of a sweep kernel on the current device: workgroup w owns the destination nodes
 * [w * nodes_per_block, (w+1) * nodes_per_block); *grid_out (optional) = workgroups launched.
 * wg_per_cu: 1 for gnm_edge_bwd_chain_src, 2 for gnm_edge_gate2_fwd (a plan serves ONE partition)            */
int gnm_sweep_partition(int64_t N, int wg_per_cu, int64_t* nodes_per_block, int* grid_out);

/* ---- greedy decode (HOST pointers, sequential CPU work; inference.py:31-77,182-253) ----------
 * build_adjacency: successors / predecessors of every node in edge-id order, as the reference's
 *   succ / pred dicts (graph_parser.py:13-73); *_eid[p] = edges[(node, nbr)] of its edges dict, i.e. the
 *   LAST edge id of a duplicated (src, dst) pair.  ptr arrays hold N+1, nbr / eid arrays E entries.
 * decode_iteration: one pass of get_contigs' loop body for `nb` sampled start edges
 *   (start_src[i] -> start_dst[i]): greedy forward walk from the head, backward walk from the tail
 *   (forced single-neighbour moves, otherwise the best-scored neighbour that is neither in visited[]
 *   nor consumed by this walk; a node consumes its reverse complement node ^ 1), the walk with the
 *   greatest reconstructed length (sum of prefix_length over its edges + read_length of its last node)
 *   wins.  Returns its node count, the nodes in walk_out, the length in *best_length_out; when the count is
 *   >= len_threshold, visited[] (N bytes, in/out) absorbs the walk, the complements and the jumped-over
 *   nodes.  < 0: error (-3: a cycle of forced moves, on which the reference does not terminate).     */
int gnm_decode_build_adjacency(const int32_t* src, const int32_t* dst, int64_t N, int64_t E,
                               int32_t* succ_ptr, int32_t* succ_nbr, int32_t* succ_eid,
                               int32_t* pred_ptr, int32_t* pred_nbr, int32_t* pred_eid);
int64_t gnm_decode_iteration(int64_t N, const float* scores, const int64_t* prefix_length,
                             const int64_t* read_length, const int32_t* succ_ptr, const int32_t* succ_nbr,
                             const int32_t* succ_eid, const int32_t* pred_ptr, const int32_t* pred_nbr,
                             const int32_t* pred_eid, uint8_t* visited, int nb, const int32_t* start_src,
                             const int32_t* start_dst, int len_threshold, int32_t* walk_out,
                             int64_t walk_cap, int64_t* best_length_out);

/* ---- dense (fp32 MFMA v_mfma_f32_32x32x2_f32): nn.Linear call sites
 *      gated_gcn_full.py:107-113, full_graph.py:23-26, score_predictor.py:15-17 and their
 *      autograd duals.  TN mode reduces over K with split-K partials in the workspace. */
size_t gnm_gemm_f32_workspace_bytes(int mode, int64_t M, int64_t N, int64_t K);
int gnm_gemm_f32(int mode, int64_t M, int64_t N, int64_t K,
                 const float* A, int64_t lda, const float* B, int64_t ldb,
                 float* C, int64_t ldc,
                 const float* bias,                 /* [N] or NULL */
                 const float* resid, int64_t ldr,   /* [M,N] added to the result, or NULL */
                 int relu, void* ws, size_t ws_bytes, void* stream);

/* column sums out[c] = sum_m X[m*ld + c], c < W (bias gradients). ws: gnm_colsum_workspace_bytes */
size_t gnm_colsum_workspace_bytes(int64_t M, int64_t W);
int gnm_colsum_f32(int64_t M, int64_t W, const float* X, int64_t ld, float* out,
                   void* ws, size_t ws_bytes, void* stream);

/* Weight AND bias gradient of a Linear in one call (autograd of nn.Linear under loss.backward(), train.py:257):
 * C[M,N] = A[K,M]^T B[K,N], colsum[m] = sum_k A[k][m].  One pass over A where the split-mode kernel applies
 * (bf16x3 matmul mode, M and N multiples of 128, K >= 4096), gnm_gemm_f32(TN) + gnm_colsum_f32 otherwise. */
size_t gnm_gemm_tn_colsum_workspace_bytes(int64_t M, int64_t N, int64_t K);
int gnm_gemm_tn_colsum(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                       float* C, int64_t ldc, float* colsum, void* ws, size_t ws_bytes, void* stream);

/* out[j, 0:W] = X[idx[j], 0:W]  (edge features: caller edge-id order -> internal order) */
int gnm_gather_rows_f32(int64_t M, int64_t W, const float* X, const int32_t* idx, float* out,
                        void* stream);
/* x = (ref > 0) ? x : 0, elementwise over n floats (relu backward for the tiny encoders) */
int gnm_relu_mask_f32(int64_t n, float* x, const float* ref, void* stream);

/* ---- BatchNorm1d(track_running_stats=False) statistics: gated_gcn_full.py:122,147 ------
 * partials: double[nblk][2][H] = per-block (sum x, sum x^2) written by the *_stats kernels.
 *           The buffer must hold (gnm_max_partial_blocks() + 1) * 2 * 256 doubles: the row
 *           after the last possible partial row is the finalisers' reduction scratch.
 * stat out: float[4][H] = mean, rstd, scale = gamma*rstd, shift = beta - mean*scale.      */
int gnm_bn_finalize(const double* partials, int nblk, int64_t count, int H,
                    const float* gamma, const float* beta, float eps, float* stat, void* stream);
/* backward: partials = per-block (sum gy, sum gy*xhat).  bstat: float[2][H] = mean(gy),
 * mean(gy*xhat); ggamma[H] = sum gy*xhat, gbeta[H] = sum gy. */
int gnm_bn_bwd_finalize(const double* partials, int nblk, int64_t count, int H,
                        float* bstat, float* ggamma, float* gbeta, void* stream);

/* ---- GatedGCN layer, forward (gated_gcn_full.py:120-152) --------------------------------
 * P = [A1h|A2h|A3h|B1h|B2h] is the [N,5H] node projection (ld 5H).
 * edge_t_stats: t[j] += B1h[isrc j] + B2h[idst j]  (t holds B3e on entry), per-block
 *               (sum t, sum t^2) -> partials.                                   (:120-121) */
int gnm_edge_t_stats_fwd(int64_t E, int H, float* t, const float* P, const int32_t* isrc,
                         const int32_t* idst, double* partials, int* nblk_out, void* stream);
/* (e_in == NULL / h_in == NULL in the four forward kernels below and their LayerNorm twins: no residual --
 *  GatedGCN_1d(residual=False), or in_channels != out_channels which drops it: gated_gcn_full.py:41-42,124-125,151-152)
 * edge_gate: e_out = relu(t*scale+shift) + e_in; sigma = sigmoid(e_out);
 *            hf[v] = sum_{in(v)} sigma*A2h[src] / (sum sigma + 1e-6); inv_f = 1/(sum sigma + 1e-6)
 *                                                                      (:122-130) */
int gnm_edge_gate_fwd(int64_t N, int64_t E, int H, const float* t, const float* e_in,
                      const float* stat_e, const float* P, const int32_t* isrc,
                      const int32_t* in_ptr, float* e_out, float* hf, float* inv_f, void* stream);
/* edge_gate2 (H = 128; H = 256: the sweep is column-separable and runs once per 128-column half with row pitch 256):
 *   gnm_edge_gate_fwd AND gnm_node_agg_src_fwd as ONE two-sided sweep over the destination-sorted
 *   rows (same outputs: e_out, hf, inv_f, hb, inv_b, z, BatchNorm_h partials; e_out is not re-read): the by-source
 *   sums through the sweep plan (sinfo / dinfo / fix_nodes of gnm_graph_build_sweep_plan over
 *   gnm_sweep_partition(N, 2)); the plan's fix_nodes are covered by gathers.  inv_f == inv_b == NULL: a forward
 *   without a backward (torch.no_grad: inference.py:444) does not store what only the backward reads.  (:122-147) */
int gnm_edge_gate2_fwd(int64_t N, int64_t E, int H, const float* t, const float* e_in, const float* stat_e,
                       const float* P, const int32_t* isrc, const int32_t* idst, const int32_t* in_ptr,
                       const uint32_t* sinfo, const uint32_t* dinfo, int64_t plan_nodes_per_block, int64_t nfix,
                       const int32_t* fix_nodes, const int32_t* out_ptr, const int32_t* out_pos, const int32_t* out_dst,
                       float* e_out, float* hf, float* inv_f, float* hb, float* inv_b, float* z, double* partials,
                       int* nblk_out, void* stream);
/* the same sweep for a LayerNorm layer (batch_norm = False; H = 128; width = the layer's real out_channels, see the gnm_ln_*
 * entry points): gnm_ln_edge_gate_fwd AND gnm_node_agg_src_fwd in one pass, e_out bit-identical to gnm_ln_edge_gate_fwd's.
 * `partials` receives column sums nobody needs (z is formed by the same kernel as in the BatchNorm form).          (:122-147) */
int gnm_ln_edge_gate2_fwd(int64_t N, int64_t E, int H, const float* t, const float* e_in, const float* gamma_e,
                          const float* beta_e, int width, const float* P, const int32_t* isrc, const int32_t* idst,
                          const int32_t* in_ptr, const uint32_t* sinfo, const uint32_t* dinfo, int64_t plan_nodes_per_block,
                          int64_t nfix, const int32_t* fix_nodes, const int32_t* out_ptr, const int32_t* out_pos,
                          const int32_t* out_dst, float* e_out, float* hf, float* inv_f, float* hb, float* inv_b, float* z,
                          double* partials, int* nblk_out, void* stream);
/* node_agg_src: hb[v] = sum_{out(v)} sigma*A3h[dst] / (sum sigma + 1e-6), inv_b likewise;
 *               z = A1h + hf + hb; per-block (sum z, sum z^2) -> partials   (:133-145) */
int gnm_node_agg_src_fwd(int64_t N, int64_t E, int H, const float* e_out, const float* P,
                         const int32_t* out_ptr, const int32_t* out_pos, const int32_t* out_dst,
                         const float* hf, float* hb, float* inv_b, float* z, double* partials,
                         int* nblk_out, void* stream);
/* node_update: h_out = relu(z*scale+shift) + h_in                             (:147-152) */
int gnm_node_update_fwd(int64_t N, int H, const float* z, const float* stat_h, const float* h_in,
                        float* h_out, void* stream);
/* ---- GatedGCN layer, backward (autograd of the above) -------
 * node_bwd_stats: gw = gh_out*[relu(bn(z))>0]; partials (sum gw, sum gw*zhat)            */
int gnm_node_bwd_stats(int64_t N, int H, const float* z, const float* stat_h, const float* gh_out,
                       double* partials, int* nblk_out, void* stream);
/* node_bwd_apply: gz = gamma*rstd*(gw - m1 - zhat*m2) -> gP[:,0:H];
 *                 Q[N,2H] = Qf | Qb = gz*inv_f | gz*inv_b                                 */
int gnm_node_bwd_apply(int64_t N, int H, const float* z, const float* stat_h, const float* bstat_h,
                       const float* gamma_h, const float* gh_out, const float* inv_f, const float* inv_b,
                       float* gP, float* Q, void* stream);
/* edge_bwd_dst (internal order, by destination), Rf = Qf*hf, Rb = Qb*hb formed from the saved hf / hb rows:
 *   gsigma = Qf[d]*A2h[s] - Rf[d] + Qb[s]*A3h[d] - Rb[s];  ge <- ge + gsigma*sigma*(1-sigma)
 *   gu = ge*[t*scale+shift > 0];  partials (sum gu, sum gu*that)
 *   gP[:,2H:3H][d] = sum sigma*Qb[s];  Ud[d] = sum gu;  Td[d] = sum that                 */
int gnm_edge_bwd_dst(int64_t N, int64_t E, int H, const float* e_out, const float* t,
                     const float* stat_e, float* ge, const float* P, const float* Q,
                     const float* hf, const float* hb,
                     const int32_t* isrc, const int32_t* in_ptr, float* gP, float* Ud, float* Td,
                     double* partials, int* nblk_out, void* stream);
/* edge_bwd_src (by source, after bn_bwd_finalize):
 *   gP[:,H:2H][v]  = sum_{out(v)} sigma*Qf[dst]
 *   gP[:,3H:4H][v] = c*(sum_{out(v)} gu - outdeg*m1 - m2*sum_{out(v)} that)   (= sum gt)
 *   gP[:,4H:5H][v] = c*(Ud[v] - indeg*m1 - m2*Td[v]),  c = gamma_e*rstd_e                */
int gnm_edge_bwd_src(int64_t N, int64_t E, int H, const float* e_out, const float* t,
                     const float* stat_e, const float* bstat_e, const float* gamma_e,
                     const float* ge, const float* Q, const int32_t* in_ptr,
                     const int32_t* out_ptr, const int32_t* out_pos, const int32_t* out_dst,
                     const float* Ud, const float* Td, float* gP, int max_blocks_per_cu, void* stream);
/* edge_bwd_gt: gt = gamma*rstd*(gu - m1 - that*m2), gu = ge*[t*scale+shift > 0]          */
int gnm_edge_bwd_gt(int64_t E, int H, const float* ge, const float* t, const float* stat_e,
                    const float* bstat_e, const float* gamma_e, float* gt, void* stream);

/* ---- LayerNorm mode (batch_norm=False: nn.LayerNorm(H) for bn_h / bn_e, gated_gcn_full.py:57-59) ---
 * Row-wise normalisation, no global barrier.  The projections, t (gnm_edge_t_stats_fwd or the fused
 * form) and gnm_node_agg_src_fwd are shared with BatchNorm mode (their column partials are unused).
 * ln_edge_gate_fwd   e_out = relu(LN(t)*gamma+beta) + e_in, sigma, by-destination gated mean
 * ln_node_update_fwd h_out = relu(LN(z)*gamma+beta) + h_in
 * ln_node_bwd        gz -> gP[:,0:H], Q[N,4H]; partials (sum gw, sum gw*zhat) -> gnm_bn_bwd_finalize
 * ln_edge_bwd_dst    ge <- ge + gsigma*sigma'; gt = LNbwd(ge*[u>0]) -> gt[E,H]; gP[:,2H:3H] = sum
 *                    sigma*Qb[s]; gP[:,4H:5H] = sum_dst gt; partials (sum gu, sum gu*that)
 * ln_edge_bwd_src    gP[:,H:2H] = sum_src sigma*Qf[d]; gP[:,3H:4H] = sum_src gt
 * (B_3 gradients and ge_in then come from gnm_gemm_f32 TN/NN + gnm_colsum_f32 on gt.)
 * width (ABI 5): the layer's REAL channel count, 1 <= width <= H.  nn.LayerNorm(out_channels) takes its mean and
 * variance over out_channels; a layer that runs on the next kernel width up (H > width: the channels width .. H-1 are
 * dead -- zero weights, zero inputs) must not count them: they get xhat = 0 and are left out of every row mean.  */
int gnm_ln_edge_gate_fwd(int64_t N, int64_t E, int H, const float* t, const float* e_in,
                         const float* gamma, const float* beta, const float* P, const int32_t* isrc,
                         const int32_t* in_ptr, float* e_out, float* hf, float* inv_f, int width, void* stream);
int gnm_ln_node_update_fwd(int64_t N, int H, const float* z, const float* gamma, const float* beta,
                           const float* h_in, float* h_out, int width, void* stream);
int gnm_ln_node_bwd(int64_t N, int H, const float* z, const float* gamma, const float* beta,
                    const float* gh_out, const float* hf, const float* inv_f, const float* hb,
                    const float* inv_b, float* gP, float* Q, double* partials, int* nblk_out,
                    int width, void* stream);
int gnm_ln_edge_bwd_dst(int64_t N, int64_t E, int H, const float* e_out, const float* t,
                        const float* gamma, const float* beta, float* ge, const float* P, const float* Q,
                        const int32_t* isrc, const int32_t* in_ptr, float* gP, float* gt,
                        double* partials, int* nblk_out, int width, void* stream);
int gnm_ln_edge_bwd_src(int64_t N, int64_t E, int H, const float* e_out, const float* gt, const float* Q,
                        const int32_t* out_ptr, const int32_t* out_pos, const int32_t* out_dst,
                        float* gP, void* stream);
/* round 6 (ABI 7), H = 128 with a sweep plan: gnm_ln_edge_bwd_dst + gnm_ln_edge_bwd_src as ONE two-sided sweep (the LayerNorm form of
 * gnm_edge_bwd_top): ge <- ge + gsigma sigma' in place, gt [E,H] written for gnm_edge_bwd_fused_gt, gP[:, H:5H] = gA2h | gA3h | gB1h | gB2h
 * of the nodes the plan serves, partials = (sum gu, sum gu that); then the plan's unserved sources by gathers.  Q = gnm_ln_node_bwd's [N,4H].
 * ws >= gnm_edge_bwd_fused_workspace_bytes().                        autograd of gated_gcn_full.py:120-143 under nn.LayerNorm (:57-59) */
int gnm_ln_edge_bwd_top(int64_t N, int64_t E, int H, float* ge, const float* e_out, const float* t, const float* gamma_e,
                        const float* beta_e, int width, const float* P, const float* Q, const float* hf, const float* hb,
                        const int32_t* isrc, const int32_t* idst, const int32_t* in_ptr, float* gP, float* gt,
                        double* partials, const uint32_t* sinfo, int64_t plan_nodes_per_block, int* nblk_out, void* ws,
                        size_t ws_bytes, void* stream);
int gnm_ln_edge_bwd_src_fix(int64_t nfix, const int32_t* fix_nodes, int64_t N, int64_t E, int H, const float* e_out,
                            const float* gt, const float* Q, const int32_t* out_ptr, const int32_t* out_pos,
                            const int32_t* out_dst, float* gP, void* stream);
/* The chained form (split matmul modes): gnm_edge_bwd_fused_gt of layer i (gt_hi: the gt layer i's own sweep wrote) AND gnm_ln_edge_bwd_top of
 * layer i-1 in one sweep: ge (in place) holds d loss / d e_out(i) on entry and ge_tot(i-1) on exit, gt_lo receives layer i-1's gt; gW3_hi / gb3_hi
 * the B_3 gradients of layer i.  partials_hi [grid][128] and partials_lo [grid][2][128] must differ.      train.py:257 under nn.LayerNorm */
int gnm_ln_edge_bwd_chain(int64_t N, int64_t E, int H, float* ge, const float* gt_hi, const float* e_mid, const float* W3_hi,
                          float* gW3_hi, float* gb3_hi, double* partials_hi, const float* t_lo, const float* gamma_lo,
                          const float* beta_lo, int width, const float* P_lo, const float* Q_lo, const float* hf_lo,
                          const float* hb_lo, const int32_t* isrc, const int32_t* idst, const int32_t* in_ptr, float* gP_lo,
                          float* gt_lo, double* partials_lo, const uint32_t* sinfo, int64_t plan_nodes_per_block,
                          int* nblk_out, void* ws, size_t ws_bytes, void* stream);

/* ---- fused W-stationary MFMA kernels (H = 128 only; other H use the unfused entry points) ---
 * edge_t_fused_fwd: t = e_in W3^T + b3 + B1h[isrc] + B2h[idst] and the BatchNorm partials in ONE
 *                   pass (gemm NT + gnm_edge_t_stats_fwd).            gated_gcn_full.py:113,120-122
 * node_proj_fwd:    Pout[N,ncols] = h W^T + b, W [ncols,128] row-major, ncols % 128 == 0 (:107-112)
 * node_proj_bwd:    gh_in = gh_out + gP W;  gW = gP^T h_in;  gb = sum gP   (gP [N,ncols])
 *                   (gemm NN + gemm TN + colsum).                        autograd of :107-112
 * edge_bwd_fused:   gt = gamma*rstd*(gu - m1 - that*m2), gu = ge*[t*scale+shift > 0];
 *                   ge_out = ge + gt W3 (ge_out may alias ge);  gW3 = gt^T e_in;  gb3 = sum gt
 *                   (gnm_edge_bwd_gt + gemm TN + gemm NN + colsum).  autograd of :113,:122
 * ws: gnm_rowtile_workspace_bytes(ncols) / gnm_node_proj_bwd_workspace_bytes(ncols) /
 *     gnm_edge_bwd_fused_workspace_bytes().  partials: the BatchNorm partials buffer.      */
size_t gnm_rowtile_workspace_bytes(int ncols);
/* How the fused kernels multiply a fp32 tile by a fp32 weight block (process-wide, default 2):
 *   0  v_mfma_f32_32x32x2_f32 -- fp32 operands on the matrix cores;
 *   1  "bf16x3": each fp32 operand is split EXACTLY into three bf16 terms (3 x 8 = 24 significand
 *      bits) and the product is formed from six v_mfma_f32_32x32x16_bf16 (all partial products
 *      above 2^-24 |x w|), accumulated in fp32;
 *   2  "f16x2" (round 5): x s = h1 + h2 with two fp16 terms (2 x 11 = 22 significand bits) of a
 *      POWER-OF-TWO multiple -- s puts the largest magnitude of the operand's row (of a weight's
 *      output column) at 2^14, so nothing overflows or leaves the fp16 exponent range, and the fp32
 *      accumulator is multiplied by the exact 1 / s afterwards -- and THREE v_mfma_f32_32x32x16_f16
 *      per product (h1 w1, h1 w2, h2 w1), accumulated in fp32.  In every matrix kernel of a 128-wide
 *      layer (node_proj_fwd, edge_t_fused_fwd, node_proj_bwd_nn(_stats), tn128*, edge_bwd_chain*,
 *      edge_bwd_fused), in the fused H = 256 edge kernels and in the general rows / weight-gradient
 *      GEMMs of the wide models.
 *      The step runs at the package power cap: half the matrix instructions come back as time;
 *      distance to fp64 per mode: profiles/r05_f16x2_accuracy.txt (mode 2 <= mode 1 <= mode 0).
 * Applies to the NT / NN contractions of edge_t_fused_fwd, node_proj_fwd/bwd, edge_bwd_fused.   */
int gnm_debug_set_variant(const char* what, int v);   /* A/B switches between kernel generations (tests, tools) */
int gnm_set_matmul_mode(int mode);
int gnm_get_matmul_mode(void);
/* H = 128 in every matmul mode; H = 256 (the reference's default dim_latent, hyperparameters.py:8) in the split modes (1, 2; bf16x3 arithmetic): a
 * workgroup of eight waves keeps one 128-column half of W3 stationary, the two halves of the contraction meet in LDS
 * (ws >= gnm_rowtile_workspace_bytes(5 * H)).                                    gated_gcn_full.py:113,120-122 */
int gnm_edge_t_fused_fwd(int64_t E, int H, const float* e_in, const float* W3, const float* b3,
                         const float* P, const int32_t* isrc, const int32_t* idst, float* t,
                         double* partials, int* nblk_out, void* ws, size_t ws_bytes, void* stream);
/* H = 256, bf16x3 mode: gt = gamma*rstd*(gu - m1 - that*m2) (what gnm_edge_bwd_gt writes) AND ge_out = ge + gt W3 in one
 * pass over ge and t; gt is kept for the weight-gradient GEMM (gW3 = gt^T e_in).  ge_out and gt must not alias ge.
 * ws >= gnm_rowtile_workspace_bytes(5 * H).                                 autograd of gated_gcn_full.py:113,122 */
int gnm_edge_bwd_gt_nn(int64_t E, int H, const float* ge, const float* t, const float* stat_e, const float* bstat_e,
                       const float* gamma_e, const float* W3, float* gt, float* ge_out, void* ws, size_t ws_bytes,
                       void* stream);
int gnm_node_proj_fwd(int64_t N, int H, int ncols, const float* h, const float* W, const float* b,
                      float* Pout, void* ws, size_t ws_bytes, void* stream);
/* edge_bwd_chain (bf16x3 mode, H = 128): gnm_edge_bwd_fused of layer i ("hi": ge, t_hi, e_mid = e_in(i) = e_out(i-1),
 * stat/bstat/gamma/W3 of layer i -> gW3_hi, gb3_hi) CHAINED with gnm_edge_bwd_dst of layer i-1 ("lo": t_lo, stat_lo,
 * P_lo, Q_lo, hf_lo, hb_lo -> gP_lo[:,2H:3H], Ud_lo, Td_lo, BatchNorm partials in partials_lo, *nblk_out rows) in
 * one sweep: ge_out (may alias ge) receives what gnm_edge_bwd_dst would have written after gnm_edge_bwd_fused, the
 * intermediate d loss / d e_out(i-1) never leaves the chip.  partials_hi: scratch, >= gnm_max_partial_blocks()*128
 * doubles, distinct from partials_lo.  ws as gnm_edge_bwd_fused.  autograd of gated_gcn_full.py:113,120-130      */
int gnm_edge_bwd_chain(int64_t N, int64_t E, int H, const float* ge, float* ge_out, const float* t_hi,
                       const float* e_mid, const float* stat_hi, const float* bstat_hi, const float* gamma_hi,
                       const float* W3_hi, float* gW3_hi, float* gb3_hi, double* partials_hi,
                       const float* t_lo, const float* stat_lo, const float* P_lo, const float* Q_lo,
                       const float* hf_lo, const float* hb_lo, const int32_t* isrc, const int32_t* idst,
                       const int32_t* in_ptr, float* gP_lo, float* Ud_lo, float* Td_lo, double* partials_lo,
                       int* nblk_out, void* ws, size_t ws_bytes, void* stream);
/* edge_bwd_chain_src: gnm_edge_bwd_chain as a TWO-SIDED sweep -- layer i-1's by-SOURCE sums (what gnm_edge_bwd_src
 * re-reads e_out, t and ge for) are formed in the same pass from the per-edge terms that are on chip, through the
 * sweep plan (sinfo of gnm_graph_build_sweep_plan, built for plan_nodes_per_block = gnm_sweep_partition(N, 1)'s), RAW:
 *   gP_lo[:,H:2H] = sum_out sigma*Qf[dst],  UT_lo[N,2H] = [ sum_out gu | sum_out that ]    (served sources only)
 * gnm_edge_bwd_src_fix then writes the same three sums for the plan's fix_nodes (gathers; needs layer i-1's
 * e_out, t, stat_e, Q and ge = this call's ge_out), and once layer i-1's BatchNorm-backward means are known
 * gnm_node_bgrad turns the raw sums into gP[:,3H:4H] = gB1h = c (Us - outdeg m1 - m2 Ts) and gP[:,4H:5H] = gB2h =
 * c (Ud - indeg m1 - m2 Td).  Together = gnm_edge_bwd_src.  Ud_lo / Td_lo: two [N,H] arrays or the halves of one [N,2H]
 * array (Td_lo == Ud_lo + H); gnm_node_bgrad takes the row pitch of Ud / Td (H or 2H, in floats) as ud_pitch (ABI 5).
 *                                                                     autograd of gated_gcn_full.py:133-143 */
int gnm_edge_bwd_chain_src(int64_t N, int64_t E, int H, const float* ge, float* ge_out, const float* t_hi,
                           const float* e_mid, const float* stat_hi, const float* bstat_hi, const float* gamma_hi,
                           const float* W3_hi, float* gW3_hi, float* gb3_hi, double* partials_hi,
                           const float* t_lo, const float* stat_lo, const float* P_lo, const float* Q_lo,
                           const float* hf_lo, const float* hb_lo, const int32_t* isrc, const int32_t* idst,
                           const int32_t* in_ptr, float* gP_lo, float* Ud_lo, float* Td_lo, double* partials_lo,
                           const uint32_t* sinfo, int64_t plan_nodes_per_block, float* UT_lo,
                           int* nblk_out, void* ws, size_t ws_bytes, void* stream);
/* edge_bwd_top (H = 128; H = 256 with a sweep plan: once per 128-column half with row pitch 256 -- the by-destination and
 * by-source backward of EVERY layer of a 256-wide stack, followed by gnm_edge_bwd_src_fix / gnm_node_bgrad at H = 256):
 * the top layer of the stack (no layer above to chain with): gnm_edge_bwd_dst on the chained kernel's sweep
 * (ge updated in place to ge + gsigma*sigma', gP[:,2H:3H], Ud, Td, BatchNorm_e backward partials) plus, with sinfo, the
 * by-source sums exactly as gnm_edge_bwd_chain_src leaves them (then gnm_edge_bwd_src_fix, gnm_node_bgrad).          */
int gnm_edge_bwd_top(int64_t N, int64_t E, int H, float* ge, const float* e_out, const float* t, const float* stat_e,
                     const float* P, const float* Q, const float* hf, const float* hb, const int32_t* isrc,
                     const int32_t* idst, const int32_t* in_ptr, float* gP, float* Ud, float* Td, double* partials,
                     const uint32_t* sinfo, int64_t plan_nodes_per_block, float* UT, int* nblk_out, void* ws,
                     size_t ws_bytes, void* stream);
int gnm_edge_bwd_src_fix(int64_t nfix, const int32_t* fix_nodes, int64_t N, int64_t E, int H, const float* e_out,
                         const float* t, const float* stat_e, const float* ge, const float* Q, const int32_t* out_ptr,
                         const int32_t* out_pos, const int32_t* out_dst, float* gP, float* UT, void* stream);
int gnm_node_bgrad(int64_t N, int H, const float* stat_e, const float* bstat_e, const float* gamma_e,
                   const int32_t* in_ptr, const int32_t* out_ptr, const float* UT, const float* Ud, const float* Td,
                   int64_t ud_pitch, float* gP, void* stream);
size_t gnm_node_proj_bwd_workspace_bytes(int ncols);
int gnm_node_proj_bwd(int64_t N, int H, int ncols, const float* gP, const float* h_in, const float* W,
                      const float* gh_out, float* gh_in, float* gW, float* gb, double* partials,
                      void* ws, size_t ws_bytes, void* stream);
/* the two kernels behind gnm_node_proj_bwd on their own (same ws layout and size): gh_in = gh_out + gP W, and
 * gW = gP^T h_in, gb = sum gP                                                  autograd of :107-112 */
int gnm_node_proj_bwd_nn(int64_t N, int H, int ncols, const float* gP, const float* W, const float* gh_out,
                         float* gh_in, void* ws, size_t ws_bytes, void* stream);
int gnm_node_proj_bwd_tn(int64_t N, int H, int ncols, const float* gP, const float* h_in, float* gW, float* gb,
                         double* partials, void* ws, size_t ws_bytes, int max_blocks_per_cu, void* stream);
/* ---- the node side of the chained backward without its elementwise launches (ABI 5; bf16x3 mode, H = 128) ----
 * node_proj_bwd_nn_stats = gnm_node_proj_bwd_nn + gnm_node_bwd_stats of the layer BELOW in its epilogue: gh_in IS that layer's
 *   gh_out, so (sum gw, sum gw zhat), gw = gh_in [bn_h(z_lo) > 0], are taken from the rows on chip and one read of z_lo;
 *   partials / *nblk_out as gnm_node_bwd_stats leaves them (-> gnm_bn_bwd_finalize).         autograd of :107-112 and :147
 * tn128_bgrad = gnm_tn128 over the column groups gB1h | gB2h of gP (out = gW5[3H:5H], colsum = gb5[3H:5H]) with
 *   gnm_node_bgrad in its operand load: the groups are FORMED from the raw sums the two-sided sweep left (UT = [Us | Ts] with
 *   pitch 2H; Ud, Td with pitch ud_pitch = H or 2H) and the BatchNorm_e backward means, and WRITTEN to gP[:, 3H:5H] (row
 *   pitch 5H) for gnm_node_proj_bwd_nn*, which must run behind this call.  B = the layer's h_in [M,128].
 *   ws >= gnm_tn128_workspace_bytes().                                                  autograd of :107-112,120-122 */
int gnm_node_proj_bwd_nn_stats(int64_t N, int H, int ncols, const float* gP, const float* W, const float* gh_out,
                               float* gh_in, const float* z_lo, const float* stat_h_lo, double* partials, int* nblk_out,
                               void* ws, size_t ws_bytes, void* stream);
int gnm_tn128_bgrad(int64_t M, int H, const float* UT, const float* Ud, const float* Td, int64_t ud_pitch,
                    const float* stat_e, const float* bstat_e, const float* gamma_e, const int32_t* in_ptr,
                    const int32_t* out_ptr, float* gP, const float* B, float* out, float* colsum,
                    double* partials, void* ws, size_t ws_bytes, void* stream);
size_t gnm_edge_bwd_fused_workspace_bytes(void);
int gnm_edge_bwd_fused(int64_t E, int H, const float* ge, float* ge_out, const float* t, const float* e_in,
                       const float* stat_e, const float* bstat_e, const float* gamma_e,
                       const float* W3, float* gW3, float* gb3, double* partials, void* ws,
                       size_t ws_bytes, void* stream);
/* The same pass with gt GIVEN (round 6; H = 128, split matmul modes): the LayerNorm backward (layers/gated_gcn_full.py:58-59 selects
 * nn.LayerNorm) forms gt in its by-destination pass; this replaces its gemm_tn_colsum(gt, e_in) + gemm NN(gt, W3) + residual add:
 * gW3 = gt^T e_in, gb3 = sum gt, ge_out = ge + gt W3 (ge_out may alias ge, not gt).   autograd of gated_gcn_full.py:113 */
int gnm_edge_bwd_fused_gt(int64_t E, int H, const float* ge, float* ge_out, const float* gt, const float* e_in,
                          const float* W3, float* gW3, float* gb3, double* partials, void* ws, size_t ws_bytes, void* stream);

/* ---- edge-feature encoder (full_graph.py:24-26), H = 128 or 256, edge_features F = 2, hidden Q = 16 ----
 * fwd: e0[j] = W2 relu(W1 e_raw[perm j] + b1) + b2, internal order, one pass writing [E,H]
 *      (gnm_gather_rows + 2 gemm NT).
 * bwd: from ge0 [E,H]: gW2 [H,Q], gb2 [H], gW1 [Q,F], gb1 [Q] in one pass reading ge0 once
 *      (2 gemm TN + gemm NN + relu mask + 2 colsum).  ws: gnm_edge_encoder_bwd_workspace_bytes(). */
int gnm_edge_encoder_fwd(int64_t E, int H, int F, int Q, const float* e_raw, const int32_t* perm,
                         const float* W1, const float* b1, const float* W2, const float* b2,
                         float* e0, void* stream);
size_t gnm_edge_encoder_bwd_workspace_bytes(void);
int gnm_edge_encoder_bwd(int64_t E, int H, int F, int Q, const float* ge0, const float* e_raw,
                         const int32_t* perm, const float* W1, const float* b1, const float* W2,
                         float* gW1, float* gb1, float* gW2, float* gb2, void* ws, size_t ws_bytes,
                         void* stream);

/* ---- ScorePredictor (score_predictor.py:12-25), split-W1 form ---------------------------
 * hid[j] += Ps[isrc j] + Pd[idst j] (hid holds e*W1e^T+b1 on entry; Pn=[Ps|Pd] is [N,2*HS]);
 * score[perm j] = W2 . relu(hid[j]) + b2                                                 */
int gnm_predictor_score_fwd(int64_t E, int HS, float* hid, const float* Pn, const int32_t* isrc,
                            const int32_t* idst, const float* W2, const float* b2,
                            const int32_t* perm, float* scores, void* stream);
/* ghid[j] = gscore[perm j]*W2*[hid>0] (in place over hid);
 * partials double[nblk][2][HS]: (sum gscore*relu(hid)) -> gW2, and [.,1,0] = sum gscore -> gb2 */
int gnm_predictor_score_bwd(int64_t E, int HS, float* hid, const float* gscore, const float* W2,
                            const int32_t* perm, double* partials, int* nblk_out, void* stream);
/* Fused single-pass forms for H = 128, HS = 64 (fp32 MFMA, W1e stationary in registers):
 * fwd: hid = e W1e^T + b1 + Ps[isrc] + Pd[idst] (written only if hid != NULL), score[perm j] = W2 . relu(hid[j]) + b2.
 *      W1e = &W1[0][2H] with row stride ldw (= 3H): the e-columns of predictor.W1 (score_predictor.py:15-17).
 * bwd: ghid = gscore[perm j] W2 [hid>0] (in place over hid), ge = ghid W1e, gW1e[HS,H] = ghid^T e,
 *      gsums[0:HS] = gW2, gsums[HS:2HS] = gb1, gsums[2HS] = gb2 (gsums holds 3*HS floats).
 *      partials: the BatchNorm partials buffer.  ws: gnm_predictor_fused_workspace_bytes().          */
size_t gnm_predictor_fused_workspace_bytes(void);   /* sized for H = 256; the two kernels are built for H = 128 and 256, HS = 64 */
int gnm_predictor_fused_fwd(int64_t E, int H, int HS, const float* e, const float* W1e, int64_t ldw,
                            const float* b1, const float* Pn, const int32_t* isrc, const int32_t* idst,
                            const int32_t* perm, const float* W2, const float* b2, float* hid,
                            float* scores, void* ws, size_t ws_bytes, void* stream);
int gnm_predictor_fused_bwd(int64_t E, int H, int HS, float* hid, const float* gscore,
                            const int32_t* perm, const float* W2, const float* e, const float* W1e,
                            int64_t ldw, float* ge, float* gW1e, float* gsums, double* partials,
                            void* ws, size_t ws_bytes, void* stream);
/* Weight-gradient shape of a Linear with a 128-wide input: out[cg*128+n][c] = sum_r A[r][cg*128+n] B[r][c],
 * colsum[cg*128+n] = sum_r A[r][cg*128+n];  A [M, lda >= ncg*128], B [M,128], ncg <= 16.              */
size_t gnm_tn128_workspace_bytes(void);
int gnm_tn128(int64_t M, const float* A, int64_t lda, int ncg, const float* B, float* out, float* colsum,
              double* partials, void* ws, size_t ws_bytes, void* stream);
/* reduce double partials [nblk][rows][W] -> float out[rows][W] */
int gnm_reduce_partials(const double* partials, int nblk, int rows, int W, float* out, void* stream);
/* out[v*ldo + c] = sum_{m in [ptr[v],ptr[v+1])} X[(pos ? pos[m] : m)*W + c], c < W        */
int gnm_seg_sum_rows(int64_t N, int W, const float* X, const int32_t* ptr, const int32_t* pos,
                     float* out, int64_t ldo, void* stream);

/* ---- input feature preparation ("next" row: utils.py:67-74, 97-138; train.py:245-251) -------------
 * pagerank_pe: pe[N, 2+pe_dim] = in_deg | out_deg | pe_dim PageRank steps (alpha = 0.95 in the
 *              reference), fp64 iterate, from the graph index.  ws: gnm_pagerank_pe_workspace_bytes(N).
 * edge_feats_zscore: e[E,2] = z-scored (overlap_length, overlap_similarity), unbiased std, in the
 *              caller's edge-id order.  ws: 4 * gnm_max_partial_blocks() doubles.                    */
size_t gnm_pagerank_pe_workspace_bytes(int64_t N);
int gnm_pagerank_pe(int64_t N, int64_t E, const int32_t* isrc, const int32_t* in_ptr,
                    const int32_t* out_ptr, int pe_dim, double alpha, float* pe, void* ws,
                    size_t ws_bytes, void* stream);
int gnm_edge_feats_zscore(int64_t E, const float* overlap_length, const float* overlap_similarity,
                          float* e, void* ws, size_t ws_bytes, void* stream);

/* ---- loss (train.py:210-211,253-255): BCEWithLogitsLoss(pos_weight), mean ---------------
 * loss_out[0] = mean_k(pw*y*softplus(-x) + (1-y)*softplus(x)); gscore = dloss/dx.
 * ws: double[gnm_max_partial_blocks()] */
int gnm_bce_fwd_bwd(int64_t E, const float* scores, const float* y, float pos_weight,
                    float* loss_out, float* gscore, void* ws, size_t ws_bytes, void* stream);

/* ---- composite entry points (ABI 5): the measured path's launch sequences behind one call each -------------------------
 * For hosts that do not want to schedule the kernels themselves (a C++ trainer, a Go / Rust / Java binding).  Host code only:
 * every arithmetic step is one of the entry points above, called in the order the Python engine calls them on ONE stream, so
 * the results are the engine's bit for bit (tests/cabi/host_step.cpp runs both and compares).  H = 128, BatchNorm mode.  Memory
 * stays the caller's: every output, intermediate and scratch buffer is passed in (device pointers unless noted).
 *   gnm_layer_forward    GatedGCN_1d.forward (gated_gcn_full.py:99-157): projections, t + BatchNorm_e statistics, the
 *                        two-sided gate sweep (or, without a forward plan, gate + by-source pass), BatchNorm_h, node update.
 *   gnm_stack_backward   autograd of L stacked layers (processor.py:15-20 reversed; train.py:257): gh = dL/dh_out(L-1) (read),
 *                        ge = dL/de_out(L-1), UPDATED IN PLACE to dL/de_in(0); gh_in = dL/dh_in(0); parameter gradients into
 *                        gr[i].  The chained two-sided schedule (bf16x3 mode; needs the backward sweep plan).
 * gnm_graph_view: the internal index (gnm_graph_build_index) and the sweep plans (gnm_graph_build_sweep_plan over
 * gnm_sweep_partition(N, 2) for the forward, (N, 1) for the backward; fwd_sinfo == NULL: no forward plan).
 * gnm_layer_state: a layer's inputs and everything its forward leaves for its backward (h_out / e_out of layer i are
 * h_in / e_in of layer i+1).  gnm_scratch: partials* hold gnm_compose_partials_doubles() doubles each, ws / ws2 hold
 * gnm_compose_workspace_bytes(H) bytes each (partials2, partials3, ws2: gnm_stack_backward only).                        */
typedef struct gnm_graph_view {
  int64_t N, E;
  const int32_t *isrc, *idst, *in_ptr, *out_ptr, *out_pos, *out_dst;
  const uint32_t *fwd_sinfo, *fwd_dinfo; int64_t fwd_nodes_per_block, fwd_nfix; const int32_t* fwd_fix_nodes;
  const uint32_t* bwd_sinfo; int64_t bwd_nodes_per_block, bwd_nfix; const int32_t* bwd_fix_nodes;
} gnm_graph_view;
typedef struct gnm_layer_weights { const float *W5, *b5, *W3, *b3, *gamma_e, *beta_e, *gamma_h, *beta_h; } gnm_layer_weights;
typedef struct gnm_layer_state {
  const float *h_in, *e_in;                                   /* [N,H], [E,H] (internal edge order) */
  float *P, *t, *e_out;                                       /* [N,5H], [E,H], [E,H] */
  float *hf, *inv_f, *hb, *inv_b, *z, *h_out;                 /* [N,H] each */
  float *stat_e, *stat_h;                                     /* [4,H] each */
} gnm_layer_state;
typedef struct gnm_layer_grads { float *gW5, *gb5, *gW3, *gb3, *g_gamma_e, *g_beta_e, *g_gamma_h, *g_beta_h; } gnm_layer_grads;
typedef struct gnm_backward_work {
  float* gP[2];                                               /* [N,5H] each */
  float *Q, *UT, *DT;                                         /* [N,2H] each */
  float* gh_tmp[2];                                           /* [N,H] each */
  float* bstat_e[2];                                          /* [2,H] each */
  float* bstat_h;                                             /* [2,H] */
} gnm_backward_work;
typedef struct gnm_scratch { double *partials, *partials2, *partials3; void* ws; size_t ws_bytes; void* ws2; size_t ws2_bytes; } gnm_scratch;
size_t gnm_compose_workspace_bytes(int H);
size_t gnm_compose_partials_doubles(void);
int gnm_layer_forward(const gnm_graph_view* g, int H, const gnm_layer_weights* w, const gnm_layer_state* s, const gnm_scratch* sc,
                      void* stream);
int gnm_stack_backward(const gnm_graph_view* g, int H, int L, const gnm_layer_weights* w, const gnm_layer_state* s,
                       const gnm_layer_grads* gr, const float* gh, float* ge, float* gh_in, const gnm_backward_work* wk,
                       const gnm_scratch* sc, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GNM_H */
