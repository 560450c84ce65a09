/*
 * dss2_hip.h -- C ABI of libdss2_hip.so: the MI355X (gfx950) kernels for the DSS2 hot path.
 *
 * The reference (TU-Delft-AI-Energy-Lab/Deep-Statistical-Solver-for-Distribution-System-State-
 * Estimation) has no FFI layer: the path is PyTorch-eager Python (networks.py, data.py) on top of
 * torch_geometric.  Each entry point below replaces a chain of ATen/PyG ops at the cited
 * reference lines; the Python mirror of the reference interface (networks.MPN etc.,
 * data.gsp_wls_edge / get_pflow) calls them through ctypes.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in _host; the caller (PyTorch's
 *    caching allocator) owns every buffer, the library allocates nothing and keeps no state
 *    except a thread-local error string;
 *  - fp32 row-major matrices with an explicit leading dimension `ld*` (in floats), so the
 *    reference's column slices (data.x[:, :8], data.edge_attr[:, 6:]) are passed without copies;
 *  - all launches are asynchronous on `stream` (a hipStream_t passed as void*); no host sync;
 *  - return 0 on success, non-zero on error; message via dss2_last_error().
 *
 * Graph structure ("topology", built once per distinct edge_index and cached by the caller):
 *   directed edge list d in [0,E2): E2 = 2E when the reference's undirect_graph() doubles the
 *   graph (networks.py:240-258; d >= E is the reverse of stored edge d-E with edge_attr columns
 *   0 and 2 negated), else E2 = E.
 *   CSR by target:  rowptr[N+1], col[E2] = source node, ent[E2] = stored edge id | flip<<31,
 *                   w[E2] = gcn_norm weight deg^-1/2[src] * deg^-1/2[tgt]  (PyG TAGConv).
 *   CSR by source (transpose): rowptrT, colT (= target node), entT, wT.
 *   Within a row, entries are sorted by d (the order index_add visits them on the CPU).
 *   tile_start[ntiles+1]: consecutive node ranges that contain only whole graphs (no edge
 *   crosses a tile) with at most 32*nrb rows each; one workgroup processes one tile in LDS.
 */
#ifndef DSS2_HIP_H
#define DSS2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* dss2_last_error(void);
int dss2_version(void);

/* ---- topology ------------------------------------------------------------------------- */

/* Content hash + the reference's directedness rule in ONE pass over edge_index[2,E] (int64):
 *   out3[0], out3[1] = two independent 64-bit position-dependent hashes (wrapping integer sums: independent of the
 *                      order of evaluation) -- together the 128-bit key of the caller's structure cache;
 *   out3[2]          = number of edges (v0 -> u0) that reverse the batch's FIRST edge (u0 -> v0):
 *                      MPN.is_directed(edge_index) <=> out3[2] == 0  (networks.py:236-238, first edge only).
 * out3 must be zeroed by the caller; one 24-byte device-to-host copy replaces the reference's per-forward sync. */
int dss2_topology_probe(const int64_t* edge_index, int64_t n_edges, uint64_t* out3, void* stream);

/* ---- launch plans (round 5): one C call per step where a step cannot be captured into a hipGraph ----------------------------- *
 * The reference's training loop runs its step from Python batch after batch (/root/reference/dss2_run.py:134-144).  Here a step is *
 * ~14-60 launches through this C ABI, and the Python around every launch (argument structs, tensor bookkeeping, autograd) costs     *
 * ~20 us of host time: small configurations are host-bound (C1: 0.29 ms eager against 0.10 ms as a hipGraph replay).  A plan is the *
 * library's own record of a step: between dss2_plan_begin and dss2_plan_end EVERY launch entry point of this header (all threads:    *
 * autograd runs the backward on its own) still launches, and also appends itself -- bound to copies of its host-side argument        *
 * structs and tables -- to the plan; dss2_plan_run(plan, stream) then re-issues the same launches, in order, on `stream`, from ONE  *
 * C call.  Same contract as a hipGraph replay: the recorded device pointers must stay valid (record the step on tensors that        *
 * live as long as the plan, e.g. inside a private memory pool), by-value scalars are frozen (use the device-side step counter of    *
 * dss2_adamax_step_flat / _dev and use_host_seed = 0 of dss2_rng_next, as under capture).  Only launches of this library are         *
 * recorded.  One plan records at a time (process-wide).  Every entry point that launches a STEP's work records itself (model, loss, *
 * dss2_get_pflow / dss2_eval_batch, dropout masks, optimizer, dss2_collate / dss2_collate_cursor, dss2_accum_scalar); the entry     *
 * points that are not launches of a step -- structure build (dss2_topology_probe, dss2_csr_build, dss2_csr_build_graphs,             *
 * dss2_tiles_*, dss2_ell_tiles_build, dss2_deg_pows), measurement model, z-score, ragged collations -- return 3 while a plan records *
 * instead of being silently left out of it.                                                                                         */
typedef struct dss2_plan dss2_plan;
int dss2_plan_begin(dss2_plan** out);
int dss2_plan_end(dss2_plan* plan);
int dss2_plan_size(const dss2_plan* plan);                /* recorded launches */
int dss2_plan_run(const dss2_plan* plan, void* stream);
void dss2_plan_destroy(dss2_plan* plan);

/* ---- dss2_csr_build: the whole per-topology structure on the device, no host round trip.
 * Replaces MPN.undirect_graph's concatenations (networks.py:240-258), PyG gcn_norm / degree (per TAGConv call) and the
 * index handling of PyG's scatter: counting sort of the directed edge list by target, by source, and of the stored
 * edges by incident bus (integer atomics for counts and slot claims, then a per-row sort by directed edge id, so the
 * result is independent of the order the atomics land in).  doubled != 0: directed list d in [0, 2E), d >= E is the
 * reverse of stored edge d - E (flip flag in ent); 0: the list is used as given.
 * Outputs (device, caller-allocated): the arrays of the header comment (E2 = doubled ? 2E : E), perm / permT
 * (directed edge id of every CSR entry), efrom / eto, deg[N] (in-degree as float), and
 *   lastcut[N+1]: largest q' <= q such that no edge spans the boundary before row q' (a legal tile cut);
 *   meta[16] (int32): 0 max in-degree, 1 max out-degree, 2 longest / 4 shortest whole-graph segment, 3 number of
 *   segments, 5 error flag (1: endpoint outside [0, N); 2: ELL width smaller than a row), 6 / 7 max CSR entries per
 *   tile (by target / by source, written by dss2_ell_tiles_build), 8.. ntiles per candidate (dss2_tiles_walk).
 * work: dss2_csr_build_work_ints(N, E, doubled) int32 of device scratch. */
typedef struct dss2_csr_build_args {
  const int64_t* edge_index; int64_t n_edges; int64_t n_nodes; int32_t doubled;
  int32_t no_flip;   /* doubled only: 1 = reverse edges carry NO sign-flip flag in ent / entT (the Multi* / MaskEmbd* variants of
                      * the reference duplicate edge_attr unchanged, networks.py:440-444; MPN negates columns 0 and 2) */
  int32_t* rowptr; int32_t* col; int32_t* ent; int32_t* perm; float* w;
  int32_t* rowptrT; int32_t* colT; int32_t* entT; int32_t* permT; float* wT;
  int32_t* inc_rowptr; int32_t* inc_ent; int32_t* efrom; int32_t* eto;
  float* deg; int32_t* lastcut; int32_t* meta; int32_t* work;
} dss2_csr_build_args;
int dss2_csr_build(const dss2_csr_build_args* args_host, void* stream);
int64_t dss2_csr_build_work_ints(int64_t n_nodes, int64_t n_edges, int doubled);
/* The same arrays (args.work unused) in ONE launch for a batch of equal-size graphs whose ranges in the stored edge list are known (round 6):
 * graph g = rows [g * nodes_per_graph, ..) and stored edges [edge_ptr[g], edge_ptr[g + 1]) (DEVICE int64 [G + 1]; NULL: edges_per_graph
 * each).  Every global offset is then a graph-local prefix sum plus (1 or 2) * edge_ptr[g], so one wave per graph builds counts, row
 * pointers, slot claims, per-row order, gcn_norm weights, incidence lists, legal cuts and statistics in LDS -- and, if deg_pows != NULL,
 * the [N][4] folded-bias row scales of dss2_deg_pows -- with the definitions, and therefore the bits, of dss2_csr_build (an edge that leaves
 * its graph's rows sets meta[5] = 1).  What the device loader uses for every batch it assembles: a batch with a NEW structure per step
 * (BASELINE config C5) goes from ~26 launches to 5.  dss2_csr_build_graphs_supported: nodes_per_graph <= 192, max_edges_per_graph <= 1024
 * and the per-wave scratch within LDS. */
int dss2_csr_build_graphs(const dss2_csr_build_args* args_host, int32_t nodes_per_graph, const int64_t* edge_ptr, int32_t edges_per_graph,
                          int32_t max_edges_per_graph, float* deg_pows, void* stream);
int dss2_csr_build_graphs_supported(int32_t nodes_per_graph, int32_t max_edges_per_graph);

/* tile_start[ntiles+1] for batches of equal-size graphs (closed form): tile t starts at min(t * rows_per_tile, N). */
int dss2_tiles_uniform(int32_t* tile_start, int32_t ntiles, int32_t rows_per_tile, int64_t n_nodes, void* stream);
/* general batches: greedy packing of whole segments for n_cand (<= 8) candidate row budgets tm_host[c] at once, from
 * lastcut; tile_starts_host[c]: device array of cap + 1 ints; ntiles_dev[c] <- number of tiles, or -1 when a segment
 * exceeds the budget.  Sequential per candidate (each start depends on the previous one). */
int dss2_tiles_walk(const int32_t* lastcut, int64_t n_nodes, const int32_t* tm_host, int32_t n_cand,
                    int32_t* const* tile_starts_host, int32_t cap, int32_t* ntiles_dev, void* stream);

/* per-tile ELL slices of both CSRs, [ntiles][width][tm] int2 each (width 0: skipped, pointers may be NULL):
 *   ell_tiles / ellT_tiles          {local other node, weight bits}, empty slot {own row, 0}     (tile kernels)
 *   ell_ent_tiles / ellT_ent_tiles  {local other node, stored edge id | flip}, empty slot {0, -1} (edge-MLP kernels)
 * also meta[6], meta[7] <- max CSR entries of one tile (atomic max). */
typedef struct dss2_ell_build_args {
  const int32_t* rowptr; const int32_t* col; const int32_t* ent; const float* w;
  const int32_t* rowptrT; const int32_t* colT; const int32_t* entT; const float* wT;
  const int32_t* tile_start; int32_t ntiles; int32_t tm; int32_t ell_width; int32_t ellT_width;
  void* ell_tiles; void* ell_ent_tiles; void* ellT_tiles; void* ellT_ent_tiles; int32_t* meta;
  int32_t uniform_rows;   /* round 6.  > 0 (with n_nodes): tiles of equal-size graphs, tile t = rows [t * uniform_rows, (t + 1) * uniform_rows) cut at
                           * n_nodes -- tile_start is then an OUTPUT of this launch (dss2_tiles_uniform folded in); 0: tile_start is read */
  int64_t n_nodes;
} dss2_ell_build_args;
int dss2_ell_tiles_build(const dss2_ell_build_args* args_host, void* stream);

/* out[N,4] <- columns [deg, A deg, A^2 deg, A^3 deg] (A = the gcn_norm propagation matrix of the CSR), accumulated in
 * float64 in CSR order: the row scales of a bias folded through m propagations.  work: 2 N device doubles. */
int dss2_deg_pows(const int32_t* rowptr, const int32_t* col, const float* w, const float* deg, int64_t n_nodes,
                  float* out, double* work, void* stream);

/* ---- K6: standalone CSR segmented sum (the scatter-add of MessagePassing aggr='add',
 *      networks.py:164,206, measured on its own).  out[i,:] = sum_{e in row i} msg[ent[e],:]
 *      `ent` holds row indices into msg (directed edge ids).  h in {32, 64, 128, 256} with 16-byte aligned operands
 *      takes the HBM-rate kernel (16-byte lanes); any other width / alignment a one-thread-per-element kernel.
 *      Summation order inside a row = CSR order (ascending directed edge id): deterministic, no atomics. */
int dss2_segment_sum(const float* msg, int64_t ldm, const int32_t* rowptr, const int32_t* ent,
                     float* out, int64_t ldo, int64_t n_rows, int h, void* stream);

/* out[r,:] = src[idx[r],:], r < n_rows: the gather half of PyG MessagePassing.propagate (x_j = x[edge_index[0]],
 * x_i = x[edge_index[1]]; call site networks.py:206) and the backward of the segmented sum. */
int dss2_gather_rows(const float* src, int64_t lds, const int32_t* idx, float* out, int64_t ldo, int64_t n_rows, int h,
                     void* stream);

/* ---- weight packing: nn.Linear weights -> MFMA B-operand fragment order ----------------- */

typedef struct dss2_pack_desc {
  const float* src;   /* W, row-major [rows, cols] with leading dimension ld            */
  float* dst;         /* packed matrix base: [ncg][kpad/8][64 lanes][4]                 */
  int32_t rows, cols, ld;
  int32_t transpose;  /* bit 0: 1: B[k][j] = W[j][k] (forward, K=cols); 0: B[k][j] = W[k][j].
                       * bit 1: write the bf16x3 layout [ncg][kpad/16][3 planes][64 lanes][8 bf16] (kpad a multiple of 16):
                       * every weight split into three bf16 pieces h + m + l for the fp32-accurate bf16 MFMA path
                       * bit 2: ignored by the kernels (the host side marks descriptors whose src the fold of the same step writes)
                       * bit 3: the f16x2 layout of the f16x3 chains (round 5): dst = the GROUP's buffer [matrix][ncg][kpad/16][2 planes]
                       * [64 lanes][8 fp16] + one int32 per matrix and packed column ([matrix][ncg * 32]: the exponent s of the power-
                       * of-two scale applied to that column before the split, 2^s max_k |B[k][j]| in [2^14, 2^15)); koff = matrices in
                       * the group, joff = this matrix's index (not offsets); the matrix is written whole, padding included          */
  int32_t koff;       /* k offset of this block inside the packed matrix (any value)   */
  int32_t kpad;       /* padded K of the packed matrix (multiple of 8)                  */
  int32_t ncg;        /* number of 32-column groups of the packed matrix                */
  int32_t joff;       /* column offset of this block inside the packed matrix           */
} dss2_pack_desc;

/* descs: device array of n_desc descriptors.  Several descriptors may fill disjoint k / column
 * ranges of one packed matrix (stacked or side-by-side blocks).  The kernel writes exactly the
 * elements B[koff + k][joff + j], k < K, j < J; the caller zero-fills dst once so that padding
 * (k >= total K, columns >= total J) stays zero. */
int dss2_pack_weights(const dss2_pack_desc* descs, int n_desc, int max_elems, void* stream);

/* ---- K1: EdgeAggregation (networks.py:159-209) ------------------------------------------ */

/* S[i,:] = sum_{e: tgt(e)=i} relu(W1 . [x_i | x_src(e) | ea'(e)] + b1)      (first Linear + ReLU
 * of networks.py:170-174,181, aggregated before the second Linear, which is linear:
 * out = S . W2^T + deg * b2 is finished by dss2_gemm_prop with nmat = 1).
 * fn must be 8 and fe must be 6 (the only dims the reference's data produces). h <= 256. */
int dss2_edge_hidden_fwd(const float* x, int64_t ldx, const float* ea, int64_t ldea,
                         const float* W1, const float* b1,
                         const int32_t* rowptr, const int32_t* col, const int32_t* ent,
                         float* S, int64_t n_nodes, int h, int fn, int fe, void* stream);

/* Backward of the above.  dS[N,h] is the gradient w.r.t. S.
 * by_source = 0: walks the CSR by target; accumulates per-workgroup partial dW1[h,fn*2+fe] and
 *               db1[h] into slab[n_slabs][h*(2fn+fe) + h] (reduced by dss2_reduce_slabs) and,
 *               if U != NULL, writes U[i,:] = sum_{e->i} dz_e          (ldu floats per row).
 * by_source = 1: walks the transposed CSR (pass rowptrT/colT/entT); writes
 *               U[i,:] = sum_{e: src(e)=i} dz_e; slab is not touched.
 * Returns the number of slabs it will write through *n_slabs_out (host) when slab == NULL. */
int dss2_edge_hidden_bwd(const float* x, int64_t ldx, const float* ea, int64_t ldea,
                         const float* W1, const float* b1, const float* dS,
                         const int32_t* rowptr, const int32_t* col, const int32_t* ent,
                         float* slab, int n_slabs, float* U, int64_t ldu,
                         int64_t n_nodes, int h, int fn, int fe, int by_source, void* stream);

/* EdgeAggregation with node features of any width (networks.py:159-209 as MultiMPN / MaskEmbdMultiMPN instantiate it on the
 * hidden activation, networks.py:486-498).  AB[N, 2h] = X [W1[:, :d] ; W1[:, d:2d]]^T (one dss2_gemm_prop); W1c = &W1[0][2d]
 * with row stride ldw = 2d + fe; fe <= 8; any h (hidden units are independent: more than 256 run as several launches).
 *   fwd: S[i,:] = sum_{e: tgt(e)=i} relu(AB[i, :h] + AB[src(e), h:] + W1c ea'(e) + b1)
 *   bwd: dz_e = dS[tgt(e)] (z_e > 0).  by_source = 0 (CSR by target): dAB[i, :h] = sum dz, slab[n_slabs][h*fe + h] =
 *        partial dW1c, db1 (finish with dss2_reduce_slabs); by_source = 1 (CSR by source): dAB[j, h:] = sum dz.
 * Reverse edges flagged in `ent` negate edge_attr columns 0 and 2 (the reference's MPN doubling); the Multi* variants double
 * without sign flips, i.e. their structure carries no flags. */
int dss2_edge_combine_fwd(const float* AB, int64_t ldab, const float* ea, int64_t ldea, const float* W1c, int64_t ldw,
                          const float* b1, const int32_t* rowptr, const int32_t* col, const int32_t* ent, float* S,
                          int64_t n_nodes, int h, int fe, void* stream);
int dss2_edge_combine_bwd(const float* AB, int64_t ldab, const float* ea, int64_t ldea, const float* W1c, int64_t ldw,
                          const float* b1, const float* dS, const int32_t* rowptr, const int32_t* col, const int32_t* ent,
                          float* dAB, float* slab, int n_slabs, int64_t n_nodes, int h, int fe, int by_source, void* stream);

/* Tile-based variants of the two functions above (same arithmetic, same outputs): one workgroup per
 * tile with the tile's x rows, an ELL slice carrying edge ids and the gathered edge_attr rows staged
 * in LDS.  ell_ent: int2 {local other node, stored edge id | flip<<31 (-1 = empty slot)}
 * [ntiles][ell_width][32*nrb], built once per topology from the CSR by target (forward, backward with
 * by_source = 0) or by source (by_source = 1).  min(n_slabs, ntiles) persistent workgroups walk the
 * tiles and write one partial slab each. */
int dss2_edge_tile_fwd(const float* x, int64_t ldx, const float* ea, int64_t ldea, const float* W1,
                       const float* b1, const int32_t* tile_start, const void* ell_ent, int ell_width,
                       int nrb, int ntiles, float* S, int h, int fn, int fe, void* stream);
/* dss2_edge_tile_fwd for a caller that announces its backward.  bwd_with_u != 0: the caller's dss2_edge_tile_bwd will be asked for U
 * (the gradient w.r.t. x): the forward then takes the arithmetic that backward recomputes its ReLU gates with (on 96-row tiles the
 * fp32 form: the bf16x6 backward with U is not built there).  bwd_with_u == 0: exactly dss2_edge_tile_fwd. */
int dss2_edge_tile_fwd_paired(const float* x, int64_t ldx, const float* ea, int64_t ldea, const float* W1,
                              const float* b1, const int32_t* tile_start, const void* ell_ent, int ell_width,
                              int nrb, int ntiles, float* S, int h, int fn, int fe, int bwd_with_u, void* stream);
int dss2_edge_tile_bwd(const float* x, int64_t ldx, const float* ea, int64_t ldea, const float* W1,
                       const float* b1, const float* dS, const int32_t* tile_start, const void* ell_ent,
                       int ell_width, int nrb, int ntiles, float* slab, int n_slabs, float* U, int64_t ldu,
                       int h, int fn, int fe, int by_source, void* stream);

/* ---- K2/K4: fused tile GEMM + Horner graph propagation --------------------------------- *
 * Y = epilogue( sum_{m=0..nmat-1} P^m (X . B_m) ),  P = A_hat given by (rowptr, col, w).
 *  forward TAGConv (PyG TAGConv; call sites networks.py:267,271):
 *        X = h, B_m = lins[m].weight^T, P = A_hat (CSR by target), bias, optional dropout mask
 *        and ReLU;  mathematically sum_m lins[m](A_hat^m h) + b, evaluated as
 *        G0 + A_hat (G1 + A_hat G2) with G_m = h W_m^T (propagation at OUTPUT width).
 *  data-gradient of TAGConv: X = dOut, B_m = lins[m].weight, P = A_hat^T (CSR by source).
 *  plain Linear (second Linear of the edge MLP): nmat = 1, bias scaled per row by rowscale.
 * Epilogue order: + bias[col]*(rowscale?rowscale[row]:1); * dmask; relu; * (relu_src > 0);
 * + add_src.  One workgroup per tile; requires nrb in {1,2,4,8} and nmat in 1..4.          */
typedef struct dss2_gemm_prop_args {
  const float* X; int64_t ldx; int32_t kreal; int32_t kpad;
  const float* Bp;                 /* packed [nmat][ncg][kpad/8][64][4]                   */
  const float* bias; const float* rowscale;
  const float* relu_src; int64_t ld_relu;
  const float* dmask; int64_t ld_dmask;
  const float* add_src; int64_t ld_add;
  float* Y; int64_t ldy; int32_t hout; int32_t ncg;
  int32_t relu; int32_t nmat; int32_t nrb; int32_t ntiles;
  const int32_t* tile_start;
  const int32_t* rowptr; const int32_t* col; const float* w;
  int32_t max_nnz;                 /* max CSR entries of one tile (CSR staging)            */
  int32_t ell_width;               /* > 0: max row degree of the batch; the tile's graph    *
                                    * slice is staged as ELL [ell_width][rows] (fixed trip  *
                                    * count); 0: stage the CSR slice (any degree)           */
  int32_t prop_in;                 /* > 0: INPUT-side propagation (narrow inputs): X has    *
                                    * kreal/(prop_in+1) real columns; the kernel appends    *
                                    * P X, P^2 X, ... in LDS and runs ONE GEMM against the  *
                                    * k-stacked matrix [B_0; B_1; ...] (nmat must be 1)     */
  int32_t narrow_h;                /* > 0: OUTPUT-side propagation for narrow outputs: the  *
                                    * nmat matrices (narrow_h columns each, nmat*narrow_h   *
                                    * <= 32) sit side by side in ONE 32-column group; the   *
                                    * Horner recurrence runs across column blocks in LDS    */
  const float* prebias;            /* optional [nmat][hout] with pre_rowscale [N][4]: adds the  *
                                    * rank-nmat term sum_m pre_rowscale[row][m] * prebias[m][col] *
                                    * before the activation.  With pre_rowscale rows             *
                                    * [s, P s, P^2 s, P^3 s] this is sum_m P^m (s (x) prebias_m): *
                                    * a row-scaled bias inside every X B_m, i.e. the edge MLP's   *
                                    * second Linear folded in (s = deg, prebias_m = W_m b2)       */
  const float* pre_rowscale;
  const void* ell_tiles;           /* optional: per-tile ELL slices precomputed once per     *
                                    * topology, int2 {local src, weight bits}[ntiles]        *
                                    * [ell_width][32*nrb] (rows >= tile rows: {row, 0});      *
                                    * NULL: the kernel derives the slice from the CSR         */
  /* in-kernel dropout (nn.Dropout of networks.py:268, regenerated instead of stored): drop_state = device {seed,   *
   * offset} of this forward call (dss2_rng_next) or NULL; the epilogue multiplies element (row, col) by 0 or       *
   * drop_scale = 1/(1-p) according to Philox4x32-10(seed, offset, drop_id, row, col/4) >= drop_thr = p * 2^32,     *
   * where drop_id > 0 names the layer whose mask this is (0: no dropout; in a chain: per layer).  Applied where     *
   * `dmask` (an explicit [N, hout] multiplier tensor, still supported) is applied.                                 */
  const uint64_t* drop_state; uint32_t drop_thr; float drop_scale; int32_t drop_id;
  int32_t b_format;   /* layout of the packed weights: 0 = fp32 fragments; 1 = bf16x3 fragments (dss2_gemm_prop_chain, and
                         dss2_gemm_prop where dss2_gemm_prop16_supported(...) != 0; kpad is then a multiple of 16) */
  int32_t max_tile_rows;   /* 0: unknown (any tile may hold up to 32 * nrb rows); > 0: an upper bound of EVERY tile's row count --
                              the tall-tile chains then leave out the row pieces that are padding in every tile (70-bus graphs in
                              96-row tiles: rows 72..95, a quarter of the hops and of the epilogue) */
} dss2_gemm_prop_args;

int dss2_gemm_prop(const dss2_gemm_prop_args* args_host, void* stream);
/* Policy constant shared with the host: 96- / 192-row tiles with ONE column group (hout <= 32) take the split-plane chain -- single-wave
 * workgroups -- only from this many tiles on (a launch with fewer runs the multi-wave chain of that shape); -1: never
 * (DSS2_CHAIN_SP6_NCG1=0).  The *_supported queries below answer for the capability, without a tile count. */
int dss2_chain_sp6_single_group_min_tiles(void);

/* ---- layer chain: n_layers (<= 8) H -> H layers of dss2_gemm_prop in ONE launch, the activation
 *      tile staying in LDS from layer to layer (tiles hold whole graphs, so a tile's next layer depends on that tile
 *      only).  args describes the first layer's input X and everything the layers share (shapes, leading
 *      dimensions ldy / ld_relu / ld_dmask / ld_add, topology, pre_rowscale); per layer: packed weights, bias,
 *      epilogue operands and the output Y (every layer's output is still written once: the backward pass and the
 *      weight gradients read it).  Forward chain of TAGConv layers (/root/reference/networks.py:266-269 iterated)
 *      or, with the transposed graph and packs, the chain of their data-gradients.
 *      Supported: dss2_gemm_prop_chain_supported(...) != 0 (ELL slices, kreal == hout <= 256, hout % 4 == 0,
 *      16-byte aligned operands); otherwise call dss2_gemm_prop per layer.                                          */
typedef struct dss2_chain_layer {
  const float* Bp; const float* bias; const float* relu_src; const float* dmask; const float* add_src;
  const float* prebias; float* Y; int32_t relu; int32_t drop_id;   /* drop_id: as in dss2_gemm_prop_args, per layer */
  /* Optional, only where dss2_gemm_prop_chain_gate_words(...) > 0 (else leave NULL: the kernels of other shapes ignore both).
   * y_bits: the chain also writes one bit per stored element, Y > 0, as ntiles x gate_words 64-bit words in the kernel's own
   * order (per lane of the wave that owns the element: opaque to the caller).  gate_bits: such a buffer, written by a chain launch over the SAME tiles, hout and nmat, replaces the reads of
   * relu_src (which must still be given: it defines the gate) -- 1/32 of the bytes and no latency-exposed vector loads. */
  const uint64_t* gate_bits; uint64_t* y_bits;
} dss2_chain_layer;
/* 64-bit words per tile of y_bits / gate_bits for this shape; 0: the chain kernel of this shape has no bit form */
int dss2_gemm_prop_chain_gate_words(int nrb, int nmat, int kreal, int hout, int ell_width);
int dss2_gemm_prop_chain(const dss2_gemm_prop_args* args_host, const dss2_chain_layer* layers_host, int n_layers, void* stream);
/* The chain with the narrow head TAGConv fused in (replaces a dss2_gemm_prop launch next to the chain that re-reads [N, hid];
 * /root/reference/networks.py:266-275 -- the last TAGConv(dim_hid, dim_out) of MPN / SkipMPN and its data gradient).
 * mode 1 (forward chain): after the last chained layer, Y[N][nout] = bias + sum_m P^m (h W_m^T) (+ add_src), h = that layer's
 *   output (still written to its own Y for the backward).  mode 2 (transposed chain = data gradients): the chain's input tile is
 *   X = (gate > 0) * dropout(drop_id) * sum_m (P^T)^m G W_m with G[N][nout] the gradient w.r.t. the head's output, gate the
 *   head's input activation; X is also written to Xout (the weight-gradient kernels read it); args.X is ignored.
 * W[m]: the head's weights [nout][hid] row-major, m < nmat.  nout <= 4.  Same tiles / ELL slices as the chain. */
typedef struct dss2_chain_head {
  const float* W[4];
  const float* bias; const float* add_src; float* Y;      /* mode 1 */
  const float* G; const float* gate; float* Xout;         /* mode 2 */
  int64_t ld_add, ldy, ldg, ld_gate, ldxo;
  int32_t nout, mode, drop_id, pad;
  float* wg_slab;   /* mode 2, optional (round 5): the head's WEIGHT gradient from the same staging -- one slab per tile,
                     * [ntiles][nmat * nout * hid + nout] floats = [dW_0 .. dW_{nmat-1} ([nout][hid] each) | db], to be summed over the tiles
                     * (dss2_reduce_slabs*); `pad` > 0: the stride between the tiles' slabs in floats (a multiple of 4 lets the reduction use
                     * 16-byte lanes).  Needs gate (the head's input rows) and nout <= 2 on 64-, 96- or 192-row tiles
                     * (dss2_gemm_prop_chain_head_wgrad_supported); NULL: not computed */
} dss2_chain_head;
int dss2_gemm_prop_chain_head(const dss2_gemm_prop_args* args_host, const dss2_chain_layer* layers_host, int n_layers,
                              const dss2_chain_head* head_host, void* stream);
/* mask of the head modes dss2_gemm_prop_chain_head runs for this shape: bit 0 = mode 1 (forward), bit 1 = mode 2 (backward).
 * 3 on the split-plane chain of 64-row tiles (hid >= 96, bf16x6 weights), 2 on its 96- / 192-row form, 0 otherwise. */
int dss2_gemm_prop_chain_head_supported(int nrb, int nmat, int kreal, int hout, int ell_width, int nout);
/* != 0: a mode-2 launch of this shape also forms the head's weight gradient (dss2_chain_head.wg_slab) */
int dss2_gemm_prop_chain_head_wgrad_supported(int nrb, int nmat, int kreal, int hout, int ell_width, int nout);
int dss2_gemm_prop_chain_supported(int nrb, int nmat, int kreal, int hout, int ell_width);
/* != 0: the chain can also run with args.b_format = 1 -- weights packed as bf16x3 fragments, the tile GEMM as six
 * v_mfma_f32_32x32x16_bf16 per fp32 product term set (h/m/l splits of both operands, fp32 accumulation): fp32-accurate
 * results at 12 instead of 32 MFMA cycles per unit of k.  Two-row-block tiles, H <= 128. */
int dss2_gemm_prop_chain16_supported(int nrb, int nmat, int kreal, int hout, int ell_width);
/* ... and with args.b_format = 2 as f16x3 (round 5): every layer's Bp = a group buffer in the f16x2 layout (dss2_pack_desc, transpose
 * bit 3: two fp16 planes per matrix + scale exponents), three fp16 MFMAs per product, the activation tile scaled per tile and layer by
 * an exact power of two (csrc/dss2_gemm_chain_sp.hip, MS = 2).  64-row tiles, hout = kreal a multiple of 32; layers gated by fp32
 * activations (relu_src without gate_bits) are refused -- callers keep bf16x3 weights for those.  Also through
 * dss2_gemm_prop_chain_head.  Errors of the size of fp32 arithmetic's own rounding. */
int dss2_gemm_prop_chain_f16_supported(int nrb, int nmat, int kreal, int hout, int ell_width);
/* != 0: dss2_gemm_prop (one layer) accepts args.b_format = 1 for this shape -- the tall tiles (128 / 192 rows) that run
 * matrix-sequentially with the X tile staged in two K halves; same bf16x6 arithmetic as the chain. */
int dss2_gemm_prop16_supported(int nrb, int nmat, int kreal, int hout, int max_nnz, int ell_width);

/* Diagnostic: the clock the chip holds under the dominant kernel.  probe != NULL (16 uint64 of device memory): every later split-plane
 * chain launch on 64-row tiles leaves {s_memtime, s_memrealtime} at the start and at the end of its workgroups 0, 256, 512, 768;
 * shader cycles over 100 MHz ticks = the in-kernel clock (bench.py: roofline.held_clock_ghz).  NULL: off (the default). */
void dss2_debug_chain_clock_probe(unsigned long long* probe);

/* ---- dropout random state.  state[2] = persistent device {seed, offset}; snapshot[2] <- the pair this forward call's
 * kernels (forward AND backward) read.  use_host_seed != 0: snapshot = {host_seed, 0} (eager mode: the host draws the seed
 * from torch's generator, so torch.manual_seed reproduces); 0: snapshot = {state.seed, state.offset++} (inside a hipGraph
 * capture, where a by-value seed would be frozen into the graph: every replay advances the device-side offset). */
int dss2_rng_next(uint64_t* state, uint64_t* snapshot, uint64_t host_seed, int use_host_seed, void* stream);
/* the mask a kernel with (snapshot, drop_id, p) applies, written out as [n_rows, h] multipliers (0 or 1/(1-p)): lets a
 * test feed the very same mask to the CPU oracle. */
int dss2_dropout_mask(const uint64_t* snapshot, int32_t drop_id, float p, int64_t n_rows, int h, float* out, int64_t ldo,
                      void* stream);
/* gradient through "dropout, then ReLU" of a contiguous [n_rows, h] layer output y (networks.py:268-269):
 * out = g * mask * (y > 0), mask = the (snapshot, drop_id, p) mask (snapshot NULL: no dropout); relu == 0: mask only. */
int dss2_gate_grad(const float* g, const float* y, float* out, int64_t n_rows, int h, const uint64_t* snapshot,
                   int32_t drop_id, float p, int relu, void* stream);
/* host helper: drop_thr / drop_scale for a dropout rate p (the one definition both sides use) */
void dss2_dropout_params(float p, uint32_t* thr, float* scale);

/* ---- one propagation hop in global memory: out[i,:] = epi(add[i,:] + sum_{e in CSR row i} w[e] T[col[e],:]).  The TAGConv
 *      path for graphs whose connected components exceed the LDS-resident tiles (> 192 buses): a layer is then one plain
 *      tile GEMM (dss2_gemm_prop, nmat = 1, the K+1 matrices side by side) and K of these hops; epilogue fields as in
 *      dss2_gemm_prop_args, applied in the same order (bias, dropout, relu, relu_src gate, add_src).  float4 lanes when h % 4 == 0 and
 *      every operand is 16-byte aligned, scalar lanes otherwise.                                                          */
typedef struct dss2_csr_axpy_args {
  const int32_t* rowptr; const int32_t* col; const float* w;
  const float* T; int64_t ldt;             /* gathered operand [N, >= h]                        */
  const float* add; int64_t ld_add;        /* optional addend (G_m of the Horner recurrence)     */
  float* out; int64_t ldo;
  const float* bias; const float* relu_src; int64_t ld_relu; const float* add_src; int64_t ld_src;
  const uint64_t* drop_state; uint32_t drop_thr; float drop_scale; int32_t drop_id; int32_t relu;
  int64_t n_rows; int32_t h; int32_t pad_;
} dss2_csr_axpy_args;
int dss2_csr_axpy(const dss2_csr_axpy_args* args_host, void* stream);

/* ---- K4: weight gradient of TAGConv / Linear -------------------------------------------- *
 * dW_m[o,i] = sum_n (P^m G)[n,o] * X[n,i]   (P = A_hat^T via the CSR by source), m < nmat,
 * db[o] = sum_n G[n,o] * (rowscale ? rowscale[n] : 1).
 * Partial sums per workgroup go to slab[n_split][nmat*hout*hin + hout]; finish with
 * dss2_reduce_slabs.  Deterministic (no float atomics).                                    */
typedef struct dss2_wgrad_args {
  const float* G; int64_t ldg; int32_t hout;
  const float* X; int64_t ldx; int32_t hin;
  const float* rowscale;
  float* slab; int32_t n_split; int32_t nmat; int32_t nrb; int32_t ntiles;
  const int32_t* tile_start;
  const int32_t* rowptrT; const int32_t* colT; const float* wT;
  int32_t max_nnz; int32_t ell_width;   /* as in dss2_gemm_prop_args, for the transposed CSR */
  const void* ell_tiles;                /* per-tile ELL slices of the transposed graph, or NULL   */
  const float* rowscale2;               /* optional [N][4] (16-byte aligned): additionally emit, for *
                                         * every m, sum_n rowscale2[n][m] G[n, o] (the gradient of  *
                                         * dss2_gemm_prop's prebias_m when rowscale2 is its          *
                                         * pre_rowscale); slab = [nmat*hout*hin][hout][nmat*hout]    */
  int32_t narrow; int32_t mfma_bf16;    /* mfma_bf16 != 0: products as bf16x6 on the bf16 matrix *
                                         * pipe (fp32-accurate) where that kernel covers the     *
                                         * shape (64-row tiles, K <= 2, ELL slices, 16-byte      *
                                         * aligned operands), else the fp32 MFMA kernel.         *
                                         * narrow != 0 (needs nmat*hout <= 32): the propagated  *
                                         * copies P^m G are appended as extra COLUMNS of one    *
                                         * 32-wide block instead of nmat separate blocks; same  *
                                         * slab layout [nmat*hout*hin + hout].                  *
                                         * (mfma_bf16 & 255) == 2: as f16x3 where that kernel    *
                                         * covers the shape (32-, 96- and 192-row tiles with ELL *
                                         * slices), bits 8..15 = headroom bits for the gain of   *
                                         * the propagation hops, ceil(log2(max row sum of        *
                                         * |P^T|^K)); elsewhere the bf16x6 / fp32 kernels        */
} dss2_wgrad_args;

int dss2_wgrad(const dss2_wgrad_args* args_host, void* stream);
/* dynamic LDS of the kernel dss2_wgrad launches for this shape (callers size n_split by it); _ex: with args.mfma_bf16 */
size_t dss2_wgrad_lds_bytes_ex(int nrb, int nmat, int hout, int hin, int max_nnz, int ell_width, int mfma_bf16);
/* workgroups the launch puts on each of the n_split tile-list slices (> 1 only for the bf16x6 kernel at H > 128): a caller
 * that wants one workgroup per CU divides its n_split by it                                                                */
int dss2_wgrad_y_slices(int nrb, int nmat, int hout, int hin, int ell_width, int mfma_bf16, int has_rowscale2);
/* workgroup groups along grid.z of dss2_wgrad_batched for n_layers layers (one per layer; (n_layers + 1) / 2 where the f16x3 tall-tile kernel
 * walks two 32-column layers per workgroup, round 6): n_split x y_slices x this = the launch's workgroups */
int dss2_wgrad_batched_groups(int nrb, int hout, int hin, int mfma_bf16, int n_layers);

/* The same for n_layers (<= 8) layers of IDENTICAL shape and leading dimensions in one launch (one grid
 * slice per layer): Gs / Xs / slabs are HOST arrays of device pointers replacing args->G / X / slab;
 * rowscale2s (NULL, or a HOST array with NULL entries for plain layers) replaces args->rowscale2 per layer;
 * every other field of args applies to all layers.  slab_stride (elements, 0 = the dense per-layer
 * stride): distance between consecutive workgroups' partial results inside slabs[l]; with slabs[l] =
 * base + (offset of layer l) and slab_stride = the sum of the layers' lengths, ONE dss2_reduce_slabs call
 * finishes all layers (or one call per destination buffer).  The layers of a block are independent once
 * their output gradients exist; at small H a single layer cannot fill the chip.                          */
int dss2_wgrad_batched(const dss2_wgrad_args* args_host, const float* const* Gs, const float* const* Xs,
                       float* const* slabs, const float* const* rowscale2s, int64_t slab_stride, int n_layers,
                       void* stream);

/* out[j] = sum_{s < n_slabs} slab[s*stride + j], j < len, fixed order. */
int dss2_reduce_slabs(const float* slab, int n_slabs, int64_t stride, float* out, int64_t len, void* stream);

/* up to 8 such reductions in ONE launch (same fixed order per output); descs_host: HOST array, passed by value */
typedef struct dss2_reduce_desc {
  const float* slab; float* out; int64_t stride; int64_t len; int32_t n_slabs; int32_t pad_;
} dss2_reduce_desc;
int dss2_reduce_slabs_multi(const dss2_reduce_desc* descs_host, int n_desc, void* stream);

/* ---- K3: gsp_wls_edge + get_pflow, forward and backward (data.py:328-459) --------------- *
 * Phase 1 (dss2_wls_loss_partials): applies theta *= (1 - slack) IN PLACE on output[:,1]
 * (data.py:413), computes the batch-global V_hv / V_lv (data.py:335-336), the AC branch flows,
 * bus injections and the five batch sums
 *     sums[0] = sum_nodes sum_c delta^2 R^-1 lambda_c     sums[1] = sum_edges (same, edges)
 *     sums[2] = sum_nodes relu(v-1.1)+relu(0.9-v)         sums[3] = sum_edges relu(|th_ij|-0.5)
 *     sums[4] = sum_edges relu(loading-1.5)               sums[5] = N, sums[6] = E  (doubles)
 * and the per-bus residual coefficients apq[N,2] = dJ/dp_i, dJ/dq_i (times N).
 * Between the phases a data-parallel caller all-reduces sums[0..6].
 * Phase 2 (dss2_wls_loss_grad): loss scalar (data.py:450-459) and d loss / d output[N,2]
 * (gradient w.r.t. the output BEFORE the in-place masking, as autograd gives in the reference).
 * inc_rowptr/inc_ent: incidence CSR over the STORED edges: row i lists e | end<<31 for every
 * stored edge with from(e)=i (end 0) or to(e)=i (end 1), sorted by (end, e).              */
typedef struct dss2_wls_args {
  const float* input; int64_t ld_in;            /* [N,8]  data.x[:, :8]                   */
  const float* edge_input; int64_t ld_ein;      /* [E,6]  data.edge_attr[:, :6]           */
  float* output; int64_t ld_out;                /* [N,2]  model output (theta masked in place) */
  const float* node_param; int64_t ld_np;       /* [N,3]  vn_kv, slack, zero_inj          */
  const float* edge_param; int64_t ld_ep;       /* [E,7]  G,B,Gs,Bs,closed,shift,imax     */
  const float* x_mean; const float* x_std;      /* [8]                                     */
  const float* edge_mean; const float* edge_std;/* [6]                                     */
  const int32_t* efrom; const int32_t* eto;     /* [E] stored edge endpoints               */
  const int32_t* inc_rowptr; const int32_t* inc_ent;
  int64_t n_nodes; int64_t n_edges;
  float lam_v, lam_p, lam_pf, lam_reg;
  double* sums;            /* [8] device                                                  */
  double* partials;        /* [n_blocks_max*5] device scratch, n_blocks_max = 1024         */
  float* vminmax;          /* [130] device scratch: [2+2b], [3+2b] = (min, max) of vn_kv over workgroup b < 64 */
  float* apq;              /* [N,2] device scratch                                         */
  float* loss;             /* [1] device                                                   */
  float* grad_output;      /* [N,2] device, contiguous                                     */
  float* pflow;            /* optional [E,8]: get_pflow's 8 outputs per stored edge, or NULL */
  int32_t flags;           /* DSS2_WLS_* below                                             */
  uint32_t* counter;       /* DSS2_WLS_FUSED_FINISH: one device word, zero before the first use (the kernel re-zeroes it) */
  const float* gscale;     /* dss2_wls_loss_grad: optional DEVICE scalar, the upstream gradient of the loss (autograd's
                              grad_output): the kernel scales d loss / d output by it instead of a separate multiply */
} dss2_wls_args;
#define DSS2_WLS_VMM_CACHED 1     /* vminmax[] already holds this node_param's partial (min, max) pairs: no vminmax launch */
#define DSS2_WLS_FUSED_FINISH 2   /* the LAST workgroup of the partials kernel to finish sums the workgroup partials in a
                                     fixed order into sums[0..7] and writes the (local-batch) loss: no finish launch (same
                                     summation order, same bits).  The partials are handed over through memory (write-through
                                     stores, a relaxed agent-scope arrival count, agent-scope loads in the last workgroup): no
                                     release fence -- on MI355X that would write back an XCD's whole L2 per workgroup (the
                                     first form of this path: 28 us at N = 61 440 against 10 + 4.7 us for two launches) */
#define DSS2_WLS_NO_LOSS_WRITE 4  /* dss2_wls_loss_grad leaves loss[0] alone */

int dss2_wls_loss_partials(const dss2_wls_args* args_host, void* stream);
int dss2_wls_loss_grad(const dss2_wls_args* args_host, void* stream);
/* loss[0] <- the loss of the batch sums[0..6] describe (after a data-parallel caller has all-reduced them) */
int dss2_wls_loss_value(const dss2_wls_args* args_host, void* stream);

/* get_pflow alone (data.py:328-390; evaluation path dss2_run.py:193-194): y[N,2] = (v, theta)
 * in physical units; writes pflow[E,8] = loading_lines, loading_trafo, P_from, Q_from, P_to,
 * Q_to, I_from, I_to.  vminmax[130] is device scratch.  apply_shift != 0: the angle difference is
 * theta_i - theta_j - edge_param[:, 5] (the reference's phase_shift=False, data.py:364-365); 0: shift = 0
 * (phase_shift=True, the default and what gsp_wls_edge uses). */
int dss2_get_pflow(const float* y, int64_t ldy, const float* node_param, int64_t ld_np,
                   const float* edge_param, int64_t ld_ep, const int32_t* efrom, const int32_t* eto,
                   int64_t n_nodes, int64_t n_edges, float* vminmax, float* pflow, int apply_shift, void* stream);

/* ---- evaluation metrics of one test batch (/root/reference/dss2_run.py:178-208), kept on the
 *      device: yhat[N,2] <- (out[:,0] * x_std[0] + x_mean[0], out[:,1] * (1 - slack)); get_pflow of the labels y and
 *      of yhat (pf_true / pf_out [E,8] as dss2_get_pflow); then
 *      acc[0..9] += rmse_v, mae_v, rmse_th, mae_th, rmse_loading, mae_loading, rmse_loading_trafos,
 *                   mae_loading_trafos, prop_std_v, prop_std_th
 *      (loadings compared where the true loading != 0; std unbiased like torch.std): the ten per-batch
 *      quantities the reference sums over its test loader, so an epoch's evaluation needs ONE device-to-host copy.
 *      scratch: dss2_eval_scratch_doubles() device doubles; vminmax: 130 device floats.                          */
int dss2_eval_batch(const float* out, int64_t ldo, const float* y, int64_t ldy, const float* node_param, int64_t ld_np,
                    const float* edge_param, int64_t ld_ep, const int32_t* efrom, const int32_t* eto,
                    int64_t n_nodes, int64_t n_edges, const float* x_mean, const float* x_std, float* yhat,
                    float* pf_true, float* pf_out, float* vminmax, double* scratch, double* acc, void* stream);
int64_t dss2_eval_scratch_doubles(void);

/* ---- batched small dense products in weight space (folding the edge MLP's second Linear into conv 0
 *      and the chain rule back): C[M,N] (+)= sum_b op(A_b)[M,K] . op(B_b)[K,N]  (+ u[i] * v[j]).     */
typedef struct dss2_sgemm_desc {
  const float* A[4]; const float* B[4];    /* nbatch <= 4 operand pairs, summed                 */
  float* C;
  const float* u; const float* v;          /* optional rank-1 term                              */
  int64_t c_off;                           /* >= 0: C = base_out + c_off (elements); < 0: use C */
  int32_t M, N, K, lda, ldb, ldc, transA, transB, nbatch, accumulate;
} dss2_sgemm_desc;

/* max_tiles = max over the descriptors of ceil(M/32) * ceil(N/32); descs is a DEVICE array */
int dss2_small_gemm(const dss2_sgemm_desc* descs, int n_desc, int max_tiles, float* base_out, void* stream);

/* ---- dataset side: what sits in front of the path in every training step.
 *      Replaces the arithmetic of data_from_pickles (/root/reference/data.py:119-190) and the
 *      torch_geometric DataLoader collation the driver relies on (/root/reference/dss2_run.py:68-69,134).
 *      Samples of one case have uniform shapes (n buses, e closed branches).                            */

/* measurement model, nodes (data.py:119-141).  nodes [rows][7] float64 = (vm_pu, va_rad, p_mw, q_mvar, vn_kv,
 * bool_slack, bool_zero_inj); meas_v_mask [n_per_sample] bytes (1 = voltage magnitude measured at that bus);
 * z [rows][4] float64 standard-normal draws (the reference's np.random.normal(0, |std|) = z * |std|).
 * x [rows][11] float32 <- (V, 1/var, theta, 1/var, P, 1/var, Q, 1/var, vn_kv, bool_slack, bool_zero_inj),
 * NOT yet normalised.  float64 arithmetic rounded once, like numpy.                                      */
int dss2_measure_nodes(const double* nodes, const uint8_t* meas_v_mask, int32_t n_per_sample, const double* z,
                       double v_noise, double pm_noise, double p_noise, double zero_inj_coef, float* x, int64_t rows,
                       void* stream);

/* measurement model, closed branches (data.py:144-167).  edges [rows][11] float64 = (from_bus, to_bus, p_from_mw,
 * q_from_mvar, G, B, Gs, Bs, closed line, phase shift, imax or sn); z [rows][2].
 * edge_attr [rows][13] float32 <- (P, 1/var, Q, 1/var, G, B | G, B, Gs, Bs, closed, shift, imax_or_sn).     */
int dss2_measure_edges(const double* edges, const uint8_t* meas_pflow_mask, int32_t e_per_sample, const double* z,
                       double p_noise, float* edge_attr, int64_t rows, void* stream);

/* masked z-score (data.py:179-190): for the first num_feat (<= 16) columns, mean / std over the NON-ZERO entries,
 * out = nan_to_num((t - mean) * (t != 0) / std); the other columns are copied.  mean/stdv [num_feat] are written
 * (nan -> 0 like the reference).  In place allowed (out == t, ld_out == ld).  scratch: device doubles,
 * dss2_masked_zscore_scratch_doubles(rows) of them.  Deterministic (fixed-order double partial sums).     */
int dss2_masked_zscore(const float* t, int64_t rows, int32_t ld, int32_t num_feat, float* out, int32_t ld_out,
                       float* mean, float* stdv, double* scratch, void* stream);
int64_t dss2_masked_zscore_scratch_doubles(int64_t rows);

/* batch collation (PyG Batch semantics): for slot b of the batch and sample s = sample_ids[b] (NULL: s = b),
 * kind 0: dst[b][0..chunk) = src[s][0..chunk)                    (float rows of x / edge_attr / y, chunk floats)
 * kind 1: dst[r][b*chunk + j] = src[s][r][j] + b * nodes_per_sample   (int64 edge_index, chunk = e; shared != 0:
 *         every sample uses the one [2][e] list at src).   descs_host: HOST array of 1..4 descriptors, passed
 *         to the kernel by value (no descriptor copy); sample_ids: DEVICE int64 [batch].                     */
typedef struct dss2_collate_desc {
  const void* src; void* dst;
  int32_t chunk; int32_t kind; int32_t shared; int32_t pad_;
  int64_t nodes_per_sample;
} dss2_collate_desc;
int dss2_collate(const dss2_collate_desc* descs_host, int32_t n_desc, const int64_t* sample_ids, int64_t batch, void* stream);
/* The same collation with the batch's position on the DEVICE (host-free epochs: dss2_run.py:131-147's loop as N replays of one
 * recorded step).  cursor: DEVICE int64[2] = {position, epoch length}; slot b reads sample_ids[(cursor[0] + b) mod cursor[1]]
 * (cursor[1] <= 0: no wrap).  advance != 0: a one-thread launch behind the collation moves cursor[0] forward by `batch` (wrapping
 * at cursor[1]), so the identical pair of launches, replayed, walks the epoch -- which is why this entry point is recorded into
 * launch plans and captured into hipGraphs like every other launch of a step.  The caller re-draws sample_ids (a device
 * permutation) and zeroes cursor[0] between epochs; nothing is read back. */
int dss2_collate_cursor(const dss2_collate_desc* descs_host, int32_t n_desc, const int64_t* sample_ids, int64_t* cursor,
                        int64_t batch, int advance, void* stream);
/* acc[0] += value[0]; acc[1] += 1 (DEVICE double[2], DEVICE float): the running sum of the step losses of an epoch
 * (dss2_run.py:146-147: `total_loss += loss.item()`, without the host read) as a launch of this library, so that a recorded step
 * carries it. */
int dss2_accum_scalar(double* acc, const float* value, void* stream);

/* Ragged collation for batches that mix cases with different closed-branch counts per sample (BASELINE config C5):
 * one call per case; item j < count is the j-th sample of that case in the batch: sample samp[j] of the case's store,
 * written at node row node_off[j] / edge row edge_off[j] of the batch (DEVICE int64 arrays, uploaded by the host, which
 * decides the batch composition and reads nothing back).  Descriptor fields are re-used:
 *   kind 0: chunk floats per sample; nodes_per_sample = floats per ROW of dst; shared != 0: an edge-row operand
 *           (offset by edge_off), else a node-row operand (node_off);
 *   kind 1: edge_index, chunk = e of this case; dst[r][edge_off[j] + k] = src[samp[j]][r][k] + node_off[j] with dst row
 *           stride e_total (shared != 0: one [2][e] list for every sample). */
int dss2_collate_ragged(const dss2_collate_desc* descs_host, int32_t n_desc, const int64_t* samp, const int64_t* node_off,
                        const int64_t* edge_off, int64_t count, int64_t e_total, void* stream);
/* The same for up to 4 cases in ONE launch (round 6): descs_host[n_cases][n_desc], per case its three DEVICE tables and its item count
 * (HOST arrays of pointers / counts).  A mixed batch is then one upload and one launch. */
int dss2_collate_ragged_multi(const dss2_collate_desc* descs_host, int32_t n_cases, int32_t n_desc, const int64_t* const* samp,
                              const int64_t* const* node_off, const int64_t* const* edge_off, const int64_t* count, int64_t e_total,
                              void* stream);

/* ---- optimizer step (/root/reference/dss2_run.py:91-92,143: Adamax, lr 3e-3) ---
 * torch.optim.Adamax semantics on n_desc tensors, 96 tensors per launch.  descs_host: HOST array, passed to the kernels by
 * value (no descriptor copy, nothing to keep alive, safe inside a hipGraph capture).  `step` is the 1-based step count
 * (bias correction 1 - beta1^step).  grad pointers may be views of the flat gradient bucket the backward produces. */
typedef struct dss2_adamax_desc {
  float* param; const float* grad; float* exp_avg; float* exp_inf; int64_t n;
} dss2_adamax_desc;

int dss2_adamax_step(const dss2_adamax_desc* descs_host, int n_desc, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int step, void* stream);
/* the same with the step count on the device (float, like torch's capturable optimizers): *step_dev is advanced by one and
 * then used for the bias correction, so the launches can be captured into a hipGraph with the rest of the step. */
int dss2_adamax_step_dev(const dss2_adamax_desc* descs_host, int n_desc, float lr, float beta1, float beta2, float eps,
                         float weight_decay, float* step_dev, void* stream);

/* ---- whole-stack kernels ("keep a graph resident in LDS across all L blocks") -------------------
 * /root/reference/networks.py:340-388 (PFN / SkipPFN: PFN-L chained MPN / SkipMPN blocks on the same edge inputs) and
 * :212-338 (one block), for the reference driver's own model line (dss2_run.py:72-88: dim_hid 32, K = 2, 8 layers,
 * 5 blocks, dropout 0.3).  One launch walks a tile of whole graphs through EVERY block of the stack: edge MLP ->
 * aggregation -> the H -> H TAGConv layers (conv 0 on the aggregated hidden with the edge MLP's second Linear folded
 * in) -> the narrow head -> residual, the 8-wide block output handed to the next block in LDS.  The backward launch
 * walks the same tiles through the blocks in reverse with data- and weight-gradients fused (weight gradients stay in
 * MFMA accumulators / LDS across a workgroup's tiles; one slab per workgroup and block, summed in a fixed order by
 * dss2_stack_reduce, which also applies the chain rule of the fold).  Covers dim_hid == 32, K == 2, dim_featn == 8,
 * dim_feate == 6, block output widths <= 8, 2..8 layers per block, tiles of <= 64 rows with ELL slices; every other shape
 * runs the per-block kernels above.
 *
 * params: DEVICE table of the blocks' parameter pointers in MPN._params() order, per block
 *   [W1 (hid x 22), b1, W2 (hid x hid), b2, then per conv: bias, W_0, W_1, W_2]  = 4 + 4 * (n_hh + 1) pointers.
 * wpack: uint32 scratch of dss2_stack_wpack_words() elements, rewritten by dss2_stack_pack every step (weights change).
 * Gradient layout of dss2_stack_reduce's output = the blocks' flat layouts back to back, per block
 *   [W1 | b1] [W2 | b2] then per conv [W_0 W_1 W_2 | bias]   (networks.MPN._flat_offsets). */
typedef struct dss2_stack_dims {
  int32_t n_blocks;        /* MPN blocks in the stack (1: a single MPN / SkipMPN) */
  int32_t n_hh;            /* H -> H TAGConv layers per block = n_gnn_layers - 1, 1..7 */
  int32_t dout_inner;      /* output width of blocks 0 .. n_blocks-2 (= dim_featn = 8) */
  int32_t dout_last;       /* output width of the last block, <= 8 */
  int32_t skip_inner;      /* SkipMPN residual on the inner blocks */
  int32_t skip_last;       /* ... on the last block (a standalone SkipMPN) */
} dss2_stack_dims;

typedef struct dss2_stack_args {
  dss2_stack_dims dims;
  const float* x; int64_t ldx;               /* [N, 8] input of block 0 */
  const float* ea; int64_t ldea;             /* [E, 6] stored edge features */
  const uint32_t* wpack;
  const int32_t* tile_start; int32_t ntiles; int32_t tm;      /* tm = 32 * nrb: row stride of the ELL tile slices */
  const int32_t* ell_w; const int32_t* ell_e; int32_t ell_width;        /* by target: {local col, weight} / {local col, ent} */
  const int32_t* ellT_w; const int32_t* ellT_e; int32_t ellT_width;     /* by source (backward only) */
  const float* deg_pows;                     /* [N, 4] column m = A^m deg */
  float* eacache;                            /* [ntiles][ell_width + ellT_width][64][8]: per-tile gathered, sign-corrected edge_attr
                                                rows; written by dss2_stack_pack (tiles != NULL), read by forward and backward */
  float* xs;                                 /* [n_blocks][N][8]: slot b = input of block b (slot 0 unused) */
  float* acts;                               /* [n_blocks][n_hh + 1][N][32]: slot 0 = aggregated hidden S, slot l = h_l */
  float* out; int64_t ldo;                   /* forward: [N, dout_last] */
  const float* gout; int64_t ldg;            /* backward: gradient of out */
  float* dxbuf;                              /* backward scratch [N][8]: gradient between blocks */
  float* dx_out;                             /* backward, optional: gradient of x [N][8] */
  float* slab; int64_t slab_stride; int32_t n_wg;     /* backward: n_wg persistent workgroups, one slab each */
  const uint64_t* drop_state; uint32_t drop_thr; float drop_scale; int32_t drop_stride;   /* mask id = b * drop_stride + l + 1 */
  int64_t n_nodes;
} dss2_stack_args;

int64_t dss2_stack_wpack_words(const dss2_stack_dims* dims);
int64_t dss2_stack_flat_floats(const dss2_stack_dims* dims);
/* 1 when the shape is covered (see above); 0 otherwise (the caller then runs the per-block kernels) */
int dss2_stack_supported(const dss2_stack_dims* dims, int hid, int nmat, int fn, int fe, int nrb, int ell_width, int ellT_width);
/* fold (Wf_m = W_m W2, bf_m = W_m b2 of conv 0), bf16x3 fragment packing of every matrix for the forward and the
 * data-gradient GEMMs, copies of W1 / b1 / biases -- ONE launch per step.  rng_state / rng_snapshot (optional): the
 * dropout state hand-over of dss2_rng_next done by the same launch.  tick (optional): *tick += 1 (the device-side step
 * count of a capturable optimizer). */
int dss2_stack_pack(const dss2_stack_dims* dims, const float* const* params, uint32_t* wpack, uint64_t* rng_state,
                    uint64_t* rng_snapshot, uint64_t host_seed, int use_host_seed, float* tick,
                    const dss2_stack_args* tiles, void* stream);
/* tiles (optional, the forward's argument struct): the same launch also gathers args->eacache from args->ea through the two
 * ELL slices (edge_attr changes per batch, the weights per step: both are per-forward work).  NULL: weights only (a backward
 * that has to re-pack because another forward ran in between). */
int dss2_stack_forward(const dss2_stack_args* args, void* stream);
int dss2_stack_backward(const dss2_stack_args* args, void* stream);
/* flat[i] = sum over the n_slabs slabs (fixed order; slab_stride a multiple of 4), then the chain rule of the fold: conv 0's
 * and W2 / b2's gradients from the folded ones (dW_m = dWf_m W2^T + dbf_m (x) b2; dW2 = sum_m W_m^T dWf_m;
 * db2 = sum_m W_m^T dbf_m).  Two launches. */
int dss2_stack_reduce(const dss2_stack_dims* dims, const float* slab, int32_t n_slabs, int64_t slab_stride,
                      const float* const* params, float* flat, float* fold_scratch, void* stream);
int64_t dss2_stack_fold_scratch_floats(const dss2_stack_dims* dims);      /* size of fold_scratch (device floats) */

/* The same update as ONE launch for any number of tensors whose gradients are views of ONE flat bucket (what every
 * backward of this library produces: networks._MPNFn / _PFNFn / stack._FusedStackFn): descs_dev is a DEVICE table that
 * holds, per tensor, the element offset of its gradient inside the bucket -- constant from step to step, so the table is
 * written once -- and grad_base, the bucket's address of THIS step, travels by value.  step >= 1: host-side step count;
 * step == 0: the count lives in *step_dev (float) and is advanced by the launch itself (its last workgroup to finish;
 * counter = one device word, zero before the first use), so the launch can be captured into a hipGraph. */
typedef struct dss2_adamax_flat_desc {
  float* param; int64_t grad_off; float* exp_avg; float* exp_inf; int64_t n;
} dss2_adamax_flat_desc;
int dss2_adamax_step_flat(const dss2_adamax_flat_desc* descs_dev, int n_desc, int64_t max_n, const float* grad_base, float lr,
                          float beta1, float beta2, float eps, float weight_decay, int step, float* step_dev,
                          uint32_t* counter, void* stream);

/* LDS bytes a dss2_gemm_prop / dss2_wgrad launch will request (host-side helper; lets the
 * caller reject configurations that do not fit the 160 KiB LDS before launching). */
size_t dss2_gemm_prop_lds_bytes(int nrb, int nmat, int kpad, int ncg, int max_nnz, int ell_width);
size_t dss2_wgrad_lds_bytes(int nrb, int nmat, int hout, int hin, int max_nnz, int ell_width);

#ifdef __cplusplus
}
#endif
#endif /* DSS2_HIP_H */
