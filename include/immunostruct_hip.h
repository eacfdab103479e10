/* immunostruct_hip.h -- C ABI of libimmunostruct_hip.so (gfx950 / MI355X).
 *
 * The reference (KrishnaswamyLab/ImmunoStruct) has no FFI layer: its hot path
 * reaches native code through the Python operator APIs of DGL / PyG / torch.
 * Each entry point below replaces the native kernels reached by one of those
 * call sites (paths relative to /root/reference/immunostruct):
 *
 *   is_egnn_edge_fwd / _bwd   dgl.nn.EGNNConv.forward + its autograd
 *                             (models/hybrid_models.py:323-324; SDDMM u_sub_v,
 *                             edges.src/dst gathers, edge/coord MLP, SpMM
 *                             copy_e sum/mean -- SURVEY.md section 2, K1-K5, K7)
 *   is_gather_segment_sum     the scatter-add to SOURCE rows in that backward,
 *                             expressed as a CSR-by-source gather (K7)
 *   is_segment_pool_fwd/_bwd  torch_geometric.nn.global_mean_pool /
 *                             global_max_pool (models/hybrid_models.py:331,
 *                             models/ablation_models.py:296-297; K9)
 *   is_vae_loss               Losses.regression_loss / BCE_loss
 *                             (utils/loss.py:13-31; K11)
 *
 * Conventions: plain pointers and sizes only (no torch types).  All buffers are
 * device memory owned by the caller, fp32 / int32, row-major, contiguous unless
 * a leading dimension `ld_*` (in elements) is given.  Kernels are enqueued on
 * `stream` (a hipStream_t passed as void*) and are asynchronous.  Every function
 * returns 0 on success, -22 (EINVAL) on a bad argument, -5 (EIO) if the launch
 * failed; it never throws, allocates or keeps pointers.  Hidden width is 64.
 */
#ifndef IMMUNOSTRUCT_HIP_H
#define IMMUNOSTRUCT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* library version: major*10000 + minor*100 + patch */
int is_version(void);

/* MFMA operand/accumulator layout self-tests (one wave).
 *   is_mfma_selftest:       out[32x64] = A[32x64] * W[64x64]^T
 *   is_mfma_outer_selftest: out[64x64] = G[32x64]^T * M[32x64]                */
int is_mfma_selftest(const float* A, const float* W, float* out, void* stream);
int is_mfma_outer_selftest(const float* G, const float* M, float* out, void* stream);

/* Fused EGNN edge pass, forward, for one EGNNConv layer.
 *   ps, pd   [N, ld_p]  node pre-projections of edge_mlp.0:  Ps = h W1s^T,
 *                       Pd = h W1d^T + b1  (64 columns each are read)
 *   x        [N, 3]     coordinates            ea [E, Fe] edge features, CSR slot order
 *   rowptr   [N+1], srcs [E]   CSR by destination (in-edges of node v are the
 *                       slots rowptr[v] .. rowptr[v+1])
 *   W1 [64, ldw]        the NATIVE edge_mlp.0.weight, ldw = 2*din + 1 + Fe, columns
 *                       [h_src (din) | h_dst (din) | radial | edge feats]; only the radial
 *                       and edge-feature columns are read here
 *   W2,b2 = edge_mlp.2 ; Wc1,bc1 = coord_mlp.0 ; wc2 [64] = coord_mlp.2.weight
 *   h_neigh  [N, ld_hn] out: sum of messages     x_out [N, 3] out: x + mean coord message
 *   z2s, z3s [E, 64]    out (may be NULL): pre-activations saved for the backward
 *   Fe in [0, 8].                                                              */
int is_egnn_edge_fwd(const float* ps, const float* pd, int ld_p, const float* x, const float* ea,
                     const int32_t* rowptr, const int32_t* srcs, const float* W1, int ldw, int din,
                     const float* W2, const float* b2, const float* Wc1, const float* bc1,
                     const float* wc2, float* h_neigh, int ld_hn, float* x_out, float* z2s,
                     float* z3s, int N, int Fe, void* stream);

/* number of floats of the `partials` scratch buffer is_egnn_edge_bwd needs for `grid` workgroups */
int is_egnn_edge_bwd_partials_floats(int grid);

/* Fused EGNN edge pass, backward.  Inputs as in the forward plus
 *   g_hn [N, ld_ghn] = dL/dh_neigh, g_xout [N,3] = dL/dx_out.
 * Outputs: dZ1 [E,64] and dD [E,3] (per-edge gradients of the first edge-MLP
 * pre-activation and of x_src - x_dst, CSR slot order, consumed by
 * is_gather_segment_sum), dPd [N, ld_dpd], dx [N,3] (destination-side part,
 * incl. the identity path), and ONE partial weight-gradient record per workgroup
 * (`grid` persistent workgroups, <= number of 32-node tiles) laid out as
 *   dW2 [64,64] | dWc1 [64,64] | db2 | dbc1 | dwc2 | dw_r [64] | dW_a [64,8]
 * to be summed by is_reduce_partials.                                             */
int is_egnn_edge_bwd(const float* ps, const float* pd, int ld_p, const float* x, const float* ea,
                     const int32_t* rowptr, const int32_t* srcs, const float* W1, int ldw, int din,
                     const float* W2, const float* Wc1, const float* wc2, const float* z2s,
                     const float* z3s, const float* g_hn, int ld_ghn, const float* g_xout, float* dZ1,
                     float* dD, float* dPd, int ld_dpd, float* dx, float* partials, int grid, int N,
                     int Fe, void* stream);

/* Second mapping of the same two edge passes (identical arguments, results and partial-record
 * layout): 16-edge tiles on v_mfma_f32_16x16x4_f32, 2-4 waves per SIMD; the default for Fe <= 1.
 * is_egnn_edge_bwd_v2 takes grid <= number of tiles; `tiles` is NULL (tiles of 16 consecutive
 * nodes) or the greedy tile list [count, b_0, ..., b_count, ...] (int32, <= 64 in-edges and <= 24
 * nodes per tile, immunostruct_amd/graph.py greedy_node_tiles) that fills the 64-edge windows.
 * is_egnn_edge_bwd_v2 with g_xout == NULL: no gradient arrives at the layer's coordinate output; the coordinate-MLP half of the pass is
 * skipped (z3s / Wc1 / wc2 are not read, their entries of the partial record are zero).                          */
int is_egnn_edge_fwd_v2(const float* ps, const float* pd, int ld_p, const float* x, const float* ea,
                        const int32_t* rowptr, const int32_t* srcs, const float* W1, int ldw, int din,
                        const float* W2, const float* b2, const float* Wc1, const float* bc1,
                        const float* wc2, float* h_neigh, int ld_hn, float* x_out, float* z2s,
                        float* z3s, int N, int Fe, void* stream);
int is_egnn_edge_bwd_v2(const float* ps, const float* pd, int ld_p, const float* x, const float* ea,
                        const int32_t* rowptr, const int32_t* srcs, const float* W1, int ldw, int din,
                        const float* W2, const float* Wc1, const float* wc2, const float* z2s,
                        const float* z3s, const float* g_hn, int ld_ghn, const float* g_xout, float* dZ1,
                        float* dD, float* dPd, int ld_dpd, float* dx, float* partials,
                        const int32_t* tiles, int grid, int N,
                        int Fe, void* stream);

/* Third mapping of the forward edge pass (results bit-identical to v2): wave-autonomous and software-
 * pipelined.  `dsts` [E] = destination of every CSR slot; `chunk_ptr` [nchunks+1][2] = rows (b_j, rowptr[b_j]) of the
 * node-aligned, edge-balanced cut of the destination nodes (b_0 = 0, b_nchunks = N, non-decreasing): one wave walks
 * one chunk in full 16-edge tiles, prefetching the next tile's rows while the current one is on the
 * matrix cores.  E = number of CSR slots (rowptr[N] <= E); z2s / z3s (when not NULL) need at least
 * max(E, 16) rows: tiles are always stored at full width.
 * x_out == NULL: the layer's coordinate output is not wanted (the last layer of a stack whose final coordinates are
 * unused, reference hybrid_models.py:323-324): the coordinate MLP is not evaluated, z3s is not written (may be NULL). */
int is_egnn_edge_fwd_v3(const float* ps, const float* pd, int ld_p, const float* x, const float* ea,
                        const int32_t* rowptr, const int32_t* srcs, const int32_t* dsts,
                        const int32_t* chunk_ptr, int nchunks, const float* W1, int ldw, int din,
                        const float* W2, const float* b2, const float* Wc1, const float* bc1,
                        const float* wc2, float* h_neigh, int ld_hn, float* x_out, float* z2s,
                        float* z3s, int N, int E, int Fe, void* stream);
/* is_egnn_edge_fwd_v3 with the two 64 x 64 layers on split-bf16 MFMA (x = hi + lo in bf16, three
 * v_mfma_f32_16x16x32_bf16 per product, fp32 accumulation; with IMMUNOSTRUCT_SPLIT_PIECES=3 in the environment
 * x = hi + mid + lo and six MFMAs per product: fp32-class accuracy).  Opt-in (IMMUNOSTRUCT_EDGE_FWD=v3x), Fe <= 1;
 * not bit-identical to the fp32 kernels (two pieces: ~2^-16 relative error per product).               */
int is_egnn_edge_fwd_v3x(const float* ps, const float* pd, int ld_p, const float* x, const float* ea,
                         const int32_t* rowptr, const int32_t* srcs, const int32_t* dsts,
                         const int32_t* chunk_ptr, int nchunks, const float* W1, int ldw, int din,
                         const float* W2, const float* b2, const float* Wc1, const float* bc1,
                         const float* wc2, float* h_neigh, int ld_hn, float* x_out, float* z2s,
                         float* z3s, int N, int E, int Fe, void* stream);

/* Node-level kernels of an EGNNConv layer (replace the torch/hipBLASLt Linear, cat and SiLU
 * launches around the edge pass; node_mlp of dgl.nn.EGNNConv).
 *   is_node_proj_fwd : psd [N,128] = [h W1s^T + b0 | h W1d^T + b1], h [N, ld_h] with din in {20, 64};
 *                      b0 may be NULL (EGNN); with W1 = [Wq | Wk] it is also the fused query/key
 *                      projection of the node attention (models/layers.py:13-16,68)
 *   is_egnn_node_fwd : zn1 [N,64] = [h | h_neigh] Wn1^T + bn1 (saved, may be NULL);
 *                      h_out = SiLU(zn1) Wn2^T + bn2; if W1n != NULL also the NEXT layer's
 *                      psd_next [N,128] from h_out (W1n [64, ldw_n] = next edge_mlp.0.weight, b1n its bias)
 *   is_node_proj_bwd : dh_total [N,64] = g_h + g_psd W1sd (either may be NULL: g_h treated as 0,
 *                      dh_total skipped); partial record dW1sd [128,64] | db1 [64] | db0 [64]
 *   is_egnn_node_bwd : d_h [N,64] (may be NULL) and d_hneigh [N,64] from g_hout; partial record
 *                      dWn1 [64,128] (h part padded to 64 columns | h_neigh part) | dWn2 [64,64] | dbn1 | dbn2
 * The *_floats functions give the size of the `partials` buffer for `grid` workgroups.          */
int is_node_proj_fwd(const float* h, int ld_h, int din, const float* W1, int ldw, const float* b0,
                     const float* b1, float* psd, int N, void* stream);
int is_egnn_node_fwd(const float* h, int ld_h, int din, const float* h_neigh, int ld_hn, const float* Wn1,
                     const float* bn1, const float* Wn2, const float* bn2, const float* W1n, int ldw_n,
                     const float* b1n, float* zn1, float* h_out, float* psd_next, int N, void* stream);
int is_node_proj_bwd_floats(int grid);
int is_node_proj_bwd(const float* g_h, const float* g_psd, const float* h, int ld_h, int din,
                     const float* W1, int ldw, float* dh_total, float* partials, int grid, int N,
                     void* stream);
int is_egnn_node_bwd_floats(int grid);
int is_egnn_node_bwd(const float* g_hout, const float* h, int ld_h, int din, const float* h_neigh, int ld_hn,
                     const float* zn1, const float* Wn1, const float* Wn2, float* d_h, float* d_hneigh,
                     float* partials, int grid, int N, void* stream);

/* dst[map[i]] = sum_p partials[p*stride + i] for i < count, fixed summation order (deterministic);
 * map may be NULL (identity), entries < 0 are skipped.  scratch: is_reduce_partials_scratch_floats(count). */
int is_reduce_partials_scratch_floats(int stride);
int is_reduce_partials(const float* partials, int nparts, int stride, int count, const int32_t* map,
                       float* dst, float* scratch, void* stream);

/* Second mapping of the node block (default): one workgroup per 32-row tile, weights fetched by the
 * lanes straight from the native parameter tensors (L2), no LDS weight staging.
 *   is_egnn_node_fwd_v2   : arguments / results of is_egnn_node_fwd plus b0n (NULL or [64]): bias of the FIRST
 *                           half of the pre-projection -- the last layer of a stack can so emit another
 *                           128-wide projection of h' (the fused query / key projection of the node
 *                           attention: W1n = [Wq | Wk] column blocks, ldw_n = 128, b0n = bq, b1n = bk)
 *   is_egnn_node_bwd_data : dh_total = g_h + g_psd W1sd (when g_psd != NULL; else dh := g_h and dh_total
 *                           is not written), dzn1 [N,64] = (dh Wn2) * SiLU'(zn1), d_h [N,64] (first
 *                           din columns valid; may be NULL), d_hneigh [N,64].  No weight gradients.
 *   is_egnn_node_wgrad    : the layer's weight gradients as streaming outer products over the rows:
 *                           partial record per workgroup = [dW1sd 128x64 | db1 | db0] (g_psd^T h_out,
 *                           present when g_psd != NULL; is_egnn_node_wgrad_proj_floats floats) followed by
 *                           [dWn1 64x128 | dWn2 64x64 | dbn1 | dbn2]; record stride is_egnn_node_wgrad_stride. */
/* Operand packs of the node kernels: one launch per step rewrites the node-MLP / pre-projection weights of every layer
 * in the order the lanes of is_egnn_node_fwd_v2 (fpack) and is_egnn_node_bwd_data (bpack) consume them, so each of
 * their register operand loads is one coalesced 1 KB access (NULL pack: the kernels read the native tensors).
 * jobs: host array of njobs (<= 8) records
 *   { const float *Wn1, *Wn2, *W1n; float *fpack, *bpack; int din, ldw_n, pad0, pad1; }      (W1n may be NULL)
 * with fpack / bpack of is_node_pack_floats() floats each.                                                  */
int is_node_pack_floats(void);
int is_node_pack_weights(const void* jobs, int njobs, void* stream);

/* The EGNN stack's prologue as one launch: is_node_proj_fwd for layer 0 (h [N, ld_h] with din = 20 | 64 feature columns,
 * W1 = edge_mlp.0.weight [64, ldw], b0 may be NULL, b1 = edge_mlp.0.bias -> psd [N, 128]) next to is_node_pack_weights
 * (same jobs array): the packs depend on the weights only, the two run side by side.                               */
int is_stack_prologue(const void* jobs, int njobs, const float* h, int ld_h, int din, const float* W1, int ldw,
                      const float* b0, const float* b1, float* psd, int N, void* stream);
int is_egnn_node_fwd_v2(const float* h, int ld_h, int din, const float* h_neigh, int ld_hn, const float* Wn1,
                        const float* bn1, const float* Wn2, const float* bn2, const float* W1n, int ldw_n,
                        const float* b0n, const float* b1n, float* zn1, float* h_out, float* psd_next, int N,
                        const float* fpack, void* stream);
int is_egnn_node_bwd_data(const float* g_h, const float* g_psd, const float* W1n, int ldw_n, const float* zn1,
                          int din, const float* Wn1, const float* Wn2, float* dh_total, float* dzn1,
                          float* d_h, float* d_hneigh, int N, const float* bpack, void* stream);
int is_egnn_node_wgrad_stride(void);
int is_egnn_node_wgrad_proj_floats(void);
int is_egnn_node_wgrad(const float* g_psd, const float* h_out, const float* dh, const float* zn1,
                       const float* dzn1, const float* h, int ld_h, int din, const float* h_neigh,
                       int ld_hn, float* partials, int grid, int N, void* stream);

/* Batched forms (one launch for all layers of a stack / for all pending reductions).
 *   is_egnn_node_wgrad_batched: `layers` = host array of nlayers (<= 8) records
 *       { const float *g_psd, *h_out, *dh, *zn1, *dzn1, *h, *h_neigh; float* partials;
 *         int ld_h, din, ld_hn, ld_ho, dho, pad; }
 *     each processed like is_egnn_node_wgrad with `grid` workgroups; h_out has row stride ld_ho and dho (<= 64)
 *     valid columns; dzn1 == NULL marks a projection-only job (only the dW1sd part, e.g. the layer-0
 *     pre-projection of the raw node features), g_psd == NULL a job without projection part.
 *   is_reduce_partials_batched: `jobs` = host array of njobs (<= 24) records
 *       { const float* partials; const int32_t* map; float* dst; float* scratch; int nparts, stride, count, pad; }
 *     each processed exactly like is_reduce_partials (scratch: is_reduce_partials_scratch_floats(count) floats). */
int is_egnn_node_wgrad_batched(const void* layers, int nlayers, int grid, int N, void* stream);
int is_reduce_partials_batched(const void* jobs, int njobs, void* stream);

/* out_rows[v, 0:64] = sum_{p in [ptr[v], ptr[v+1])} rows[pos[p], 0:64]   (written)
 * out_vec3[v, 0:3] += sum_{p} vec3[pos[p], 0:3]                          (accumulated; vec3 may be NULL) */
int is_gather_segment_sum(const float* rows, const float* vec3, const int32_t* ptr, const int32_t* pos,
                          float* out_rows, int ld_out, float* out_vec3, int N, void* stream);

/* Paired cancer / wild-type contrastive loss (utils/contrastive.py:18-83), forward and backward.
 * emb_c, emb_w [B, ld_e] (E valid columns; E = 104), pos [B] (1.0 where the pair is immunogenic), projector
 * W1 [128, E] (Linear, no bias), gamma / beta [128] (BatchNorm1d on batch statistics), W2 [128, 128]; lambda = weight of
 * the off-diagonal terms.  loss [1].  scratch (is_contrastive_scratch_floats(B) floats) is kept for the backward,
 * which needs a work buffer of is_contrastive_work_floats(B) floats and the upstream gradient g_loss [1] on the device,
 * and writes d loss / d emb_c, d loss / d emb_w [B, ld_d] (the projector is frozen: no parameter gradients).
 * 2 <= B <= 256, E <= 256.                                                                                     */
long long is_contrastive_scratch_floats(int B);
long long is_contrastive_work_floats(int B);
int is_contrastive_fwd(const float* emb_c, const float* emb_w, int ld_e, int E, const float* pos, const float* W1,
                       const float* gamma, const float* beta, const float* W2, float lambda, float* scratch,
                       float* loss, int B, void* stream);
int is_contrastive_bwd(const float* pos, const float* W1, const float* gamma, const float* W2, float lambda,
                       const float* scratch, float* work, const float* g_loss, float* demb_c, float* demb_w,
                       int ld_d, int E, int B, void* stream);

/* target [B] -> pos [B] = (target > mean(target)) as 1.0 / 0.0 (reference utils/contrastive.py:45) and gate [1] = 1.0 when
 * the target holds exactly two distinct values, else 0.0: the reference's host-side early-out (utils/contrastive.py:38-43)
 * as a device-side factor, for captured graphs.  1 <= B <= 1024.                                                       */
int is_contrastive_targets(const float* target, float* pos, float* gate, int B, void* stream);

/* Two-layer per-sample MLP for the small dense heads (classifier Linear(F,32)-ReLU-Dropout-Linear(32,1),
 * models/hybrid_models.py:288-295; property embedding :280-286; the pooled attention's W_v / w_concat tail,
 * models/layers.py:74-77):
 *   a1 = act1(W1 X + b1), hid = a1 * mask (mask [B,hid] = scaled dropout keep-mask or NULL), y = act2(W2 hid + b2)
 * x [B, ld_x]; with hgroup > 0 every group of hgroup hidden units reads its own `in`-wide slice of the row
 * (x holds (hid / hgroup) * in valid columns: per-head value projection).  in <= 256, hid, out <= 64.
 * act: 0 identity, 1 ReLU.  a1_out [B,hid] (may be NULL) and y are what the backward needs.
 * Backward: gx [B, ld_x] (may be NULL) and is_mlp2_bwd_records(B) partial records of
 * is_mlp2_bwd_record_floats(in, hid, out) floats [dW1 (hid x in) | db1 | dW2 (out x hid) | db2], to be summed
 * with is_reduce_partials.                                                                                   */
int is_mlp2_fwd(const float* x, int ld_x, const float* W1, const float* b1, const float* W2, const float* b2,
                const float* mask, float* a1_out, float* y, int B, int in, int hid, int out, int hgroup,
                int act1, int act2, void* stream);
int is_mlp2_bwd_records(int B);
int is_mlp2_bwd_record_floats(int in, int hid, int out);
int is_mlp2_bwd(const float* x, int ld_x, const float* W1, const float* W2, const float* mask, const float* a1,
                const float* y, const float* gy, float* gx, float* partials, int B, int in, int hid, int out,
                int hgroup, int act1, int act2, void* stream);

/* Weight / bias gradient of a Linear layer whose contraction dimension is the (small) batch: dW [N, K] = gy^T x,
 * db [N] = sum_b gy (db may be NULL) for gy [B, ld_g] (N columns), x [B, ld_x] (K columns).  Used for the two large
 * matrices of the sequence VAE (vae_fc1 512 x 5943, vae_fc4 5943 x 512; models/hybrid_models.py:297-308); forward and
 * input gradient stay on the library GEMMs.                                                                   */
int is_linear_wgrad(const float* gy, int ld_g, const float* x, int ld_x, float* dW, float* db, int B, int N, int K,
                    void* stream);

/* Multi-tensor Adam / AdamW step with torch.optim semantics (reference: torch.optim.Adam in train_IEDB_wFT.py:69-74,
 * torch.optim.AdamW in train_Cancer_wFT.py:76-92).  `chunks` = DEVICE array of nchunks records
 * { float* p; const float* g; float* m; float* v; long long n; } (one workgroup each), `state` = device float[3]
 * {step count, derived step size, derived sqrt(1 - beta2^t)} updated by the call, `hyper` = device float[8]
 * {lr, beta1, beta2, eps, weight_decay, decoupled (AdamW) flag, gradient scale (1 = none), unused}.
 * Capturable in a HIP graph.                                                                                   */
int is_adam_step(const void* chunks, int nchunks, float* state, const float* hyper, void* stream);

/* Debug aid: one single-thread launch that writes the device wall clock (100 MHz) to *slot; can be captured in a
 * HIP graph to time-stamp points of a replayed step without a profiler attached.                             */
int is_debug_timestamp(long long* slot, void* stream);

/* Batched device-to-device copy (hand-over of a device-resident batch into the static buffers of a captured
 * graph): `jobs` = host array of njobs (<= 16) records { const void* src; void* dst; long long bytes; },
 * bytes a multiple of 4.                                                                                   */
int is_multi_copy(const void* jobs, int njobs, void* stream);

/* On-device batcher (reference data/utils.py:160-176: `collate` -> dgl.batch): assemble the block-diagonal batch of the
 * B graphs idx[0..B) (int64, device) from a device-resident dataset of per-graph CSR pieces, all graphs padded to n
 * nodes: x_all [G][n][F]; eoff [G+1] edge offsets; rowptr_dst_all / rowptr_src_all [G][n+1] (0-based per graph);
 * src_all / dst_all [Etot] local node ids in destination order; pos_all [Etot] local slot ids in source order;
 * ea_all [Etot][Fe].  Writes the batch arrays (node ids + slot*n, edge slots + running edge offset); the destination
 * edge arrays must hold the selected graphs' edges (B * max edges per graph is always enough).                   */
int is_batch_gather(const long long* idx, int B, int n, int F, int Fe, const float* x_all, const int32_t* eoff,
                    const int32_t* rowptr_dst_all, const int32_t* rowptr_src_all, const int32_t* src_all,
                    const int32_t* dst_all, const int32_t* pos_all, const float* ea_all, float* x,
                    int32_t* rowptr_dst, int32_t* rowptr_src, int32_t* src_sorted, int32_t* dst_sorted,
                    int32_t* pos_by_src, float* ea, void* stream);

/* Per-segment mean and/or max over rows seg_ptr[s] .. seg_ptr[s+1] of x [rows, ld_x] (C channels).
 * out_mean / out_max [num_segments, C] may each be NULL.  Empty segment: mean 0, max 0.            */
int is_segment_pool_fwd(const float* x, int ld_x, const int32_t* seg_ptr, float* out_mean, float* out_max,
                        int num_segments, int C, void* stream);
/* dx[row, c] = g_mean[s,c]/count + (x[row,c]==out_max[s,c] ? g_max[s,c]/ties : 0); g_mean/g_max may be NULL. */
int is_segment_pool_bwd(const float* x, int ld_x, const int32_t* seg_ptr, const float* out_max,
                        const float* g_mean, const float* g_max, float* dx, int ld_dx, int num_segments,
                        int C, void* stream);

/* Node self-attention reduced to what the mean-pooled readout needs (models/layers.py:13-22,67-78 followed by
 * global_mean_pool, models/hybrid_models.py:326-331): qk [B*n,128] = [Q | K] (from is_node_proj_fwd with
 * W1 = [Wq | Wk]), x [B*n,64]; heads in {1, 8}; n <= 256 padded nodes per graph.
 *   ctx [B, heads, 64] = sum_j abar_h[j] x_j with abar_h = column mean of softmax(Q_h K_h^T / sqrt(d));
 *   the pooled attention output is then W_v,h ctx_h + b_v,h (and w_concat) on B x 64 vectors.
 * abar [B,heads,n] and probs [is_attn_colmean_probs_floats(B, n, heads) floats: the attention probabilities in
 * MFMA accumulator-tile order] are saved for the backward (both may be NULL), which reads the probabilities back
 * instead of recomputing the scores.
 * Backward: dqk [B*n,128] (fully written) and dx [B*n,64] (direct term through ctx).                       */
long long is_attn_colmean_probs_floats(int B, int n, int heads);
int is_attn_colmean_fwd(const float* qk, const float* x, float* ctx, float* abar, float* probs, int B, int n,
                        int heads, void* stream);

/* Single head: is_attn_colmean_fwd followed, in the same launch, by the pooled tail hid = W_v ctx + b_v (a1_out [B,64], may
 * be NULL), y = W_c hid + b_c (y_out [B,64]): value projection and w_concat of models/layers.py:74-77 on the mean-pooled
 * vector.  wv, wc [64,64] row-major (out, in); 1 <= n <= 256.                                                          */
int is_attn_colmean_fwd_tail(const float* qk, const float* x, float* ctx, float* abar, float* probs, const float* wv,
                             const float* bv, const float* wc, const float* bc, float* a1_out, float* y_out, int B, int n,
                             void* stream);
int is_attn_colmean_bwd(const float* qk, const float* x, const float* abar, const float* probs,
                        const float* g_ctx, float* dqk, float* dx, int B, int n, int heads, void* stream);

/* "Combined attention" of the fusion head in closed form (models/hybrid_models.py:344-347 with
 * MultiHeadAttention(F, 8 heads, input_dim = 1), models/layers.py:51-106): x [B,T] scalar tokens ->
 * z [B,T] = mean over the F features of the attention block's output.  F in {16, 32}, T <= 256.
 * wq,bq,wk,wv,bv [F] (Linear(1,F) weights / biases; the key bias does not influence the result),
 * Wc [F,F], bc [F] = w_concat.  stats (is_comb_attn_stats_floats) is written by the forward for the
 * backward (may be NULL).  The backward returns dx [B,T] and the parameter gradients packed as
 * dwq|dbq|dwk|dbk|dwv|dbv [F each] | dWc [F*F] | dbc [F]  (is_comb_attn_grad_floats).              */
int is_comb_attn_stats_floats(int B, int T);
int is_comb_attn_partials_floats(int B);
int is_comb_attn_grad_floats(int F);
int is_comb_attn_fwd(const float* x, const float* wq, const float* bq, const float* wk, const float* wv,
                     const float* bv, const float* Wc, const float* bc, float* z, float* stats, int B,
                     int T, int F, void* stream);
int is_comb_attn_bwd(const float* x, const float* stats, const float* dz, const float* wq, const float* bq,
                     const float* wk, const float* wv, const float* bv, const float* Wc, const float* bc,
                     float* dx, float* partials, float* grads, int B, int T, int F, void* stream);

/* floats of scratch is_vae_loss needs */
int is_loss_partials_floats(void);
/* mode 0: c_pred*MSE(logit,y), mode 1: c_pred*BCEWithLogits(logit,y,pos_weight);
 * + c_mse*MSE(recon,x) + c_kld*(-0.5*mean(1+logvar-mu^2-exp(logvar))).
 * recon/x/d_recon may be NULL with recon_total = 0, mu/logvar likewise with latent_total = 0.
 * out[4] = {total, prediction term, recon MSE, KLD}; d_* receive d total / d input.           */
int is_vae_loss(const float* recon, const float* x, float* d_recon, long long recon_total,
                const float* mu, const float* logvar, float* d_mu, float* d_logvar, int latent_total,
                const float* logit, const float* y, float* d_logit, int batch, int mode,
                float pos_weight, float c_pred, float c_mse, float c_kld, float* partials, float* out,
                void* stream);

#ifdef __cplusplus
}
#endif
#endif
