/* immunostruct_hip.h -- C ABI of libimmunostruct_hip.so (gfx950 / MI355X).
 *
 * The reference (KrishnaswamyLab/ImmunoStruct) has no FFI layer: its hot path
 * reaches native code through the Python operator APIs of DGL / PyG / torch.
 * Each entry point below replaces the native kernels reached by one of those
 * call sites (paths relative to /root/reference/immunostruct):
 *
 *   is_egnn_layer_fwd / _bwd  dgl.nn.EGNNConv.forward + its autograd, one launch per layer
 *                             (models/hybrid_models.py:323-324; SDDMM u_sub_v,
 *                             edges.src/dst gathers, edge/coord/node MLP, SpMM
 *                             copy_e sum/mean -- SURVEY.md section 2, K1-K7)
 *   is_gather_segment_sum     the scatter-add to SOURCE rows in that backward,
 *                             expressed as a CSR-by-source gather (K7): its own launch
 *                             for the lowest layer, inside is_egnn_layer_bwd otherwise
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
 * returns 0 on success, -22 (EINVAL) on a bad argument, -38 (ENOSYS) for a shape
 * this build's form of the kernel does not cover, -5 (EIO) if the launch failed;
 * it never throws, allocates or keeps pointers -- is_last_error_string() has the
 * text behind the calling thread's last non-zero code.  Hidden width is 64.
 */
#ifndef IMMUNOSTRUCT_HIP_H
#define IMMUNOSTRUCT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* library version: major*10000 + minor*100 + patch */
int is_version(void);

/* The calling thread's most recent failure as text: "<entry point>: <reason> (<code>)", with HIP's error name and string for a
 * failed launch; "" while nothing failed on this thread.  Thread-local, never NULL, valid for the thread's lifetime (the text is
 * replaced by the thread's next failure).  SURVEY.md 8(b): "no global mutable state except an error string (thread-local)".   */
const char* is_last_error_string(void);

/* MFMA operand/accumulator layout self-tests (one wave).
 *   is_mfma_selftest:       out[32x64] = A[32x64] * W[64x64]^T
 *   is_mfma_outer_selftest: out[64x64] = G[32x64]^T * M[32x64]                */
int is_mfma_selftest(const float* A, const float* W, float* out, void* stream);
int is_mfma_outer_selftest(const float* G, const float* M, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * One EGNNConv layer per launch (dgl.nn.EGNNConv.forward and its autograd; models/hybrid_models.py:261-263,
 * 323-324).  Algebra: edge_mlp.0 applied to [h_src | h_dst | radial | a] = Ps[src] + Pd[dst] + radial w_r + a W_a
 * with the node-level pre-projections Ps = h W1s^T, Pd = h W1d^T + b1 ("psd" [N,128] = [Ps | Pd]).
 *
 * is_stack_prologue   psd of layer 0 from the raw node features (h [N, ld_h] with din = 20 | 64 feature columns,
 *                     W1 = edge_mlp.0.weight [64, ldw], b0 may be NULL, b1 = edge_mlp.0.bias) NEXT TO the lane-ordered
 *                     operand packs of every layer's node half (forward + backward order; is_node_pack_floats()
 *                     floats per layer and direction), one launch.  jobs: host array of njobs (<= 8) records
 *                       { const float *Wn1, *Wn2, *W1n, *W1nb; float *fpack, *bpack; int din, ldw_n; }
 *                     (Wn1 / Wn2 = node_mlp.0 / .2 weights; W1n / W1nb = first row of the source / destination half of the
 *                     NEXT pre-projection, row stride ldw_n: edge_mlp.0.weight and edge_mlp.0.weight + 64 of the next layer,
 *                     or the Wq / Wk matrices of a projection head where they are; both NULL: none).  x_dst != NULL: the
 *                     launch also writes the dense [N,3] copy of the coordinates x_src [N, ld_x].
 *
 * is_egnn_layer_fwd   edge pass (gather Ps[src] + Pd[dst], geometry, SiLU, edge_mlp.2, coord_mlp, running segment
 *                     sum / mean by destination -> h_neigh [N, ld_hn], x_out [N,3]) followed in the same workgroup by
 *                     the node MLP of its own nodes (zn1 = [h | h_neigh] Wn1^T + bn1, h_out = SiLU(zn1) Wn2^T + bn2)
 *                     and the next pre-projection psd_next = [h_out W1s'^T + b0n | h_out W1d'^T + b1n].
 *     ps, pd [N, ld_p]   this layer's pre-projections (64 columns each)     x [N,3]   ea [E,Fe] (CSR slot order)
 *     rowptr [N+1], srcs [E], dsts [E]   CSR by destination
 *     chunk_ptr [nchunks+1][2]   rows (b_j, rowptr[b_j]) of the node-aligned, edge-balanced cut of the destination nodes
 *                        (b_0 = 0, b_nchunks = N, non-decreasing; nchunks a multiple of 4): one wave walks one chunk
 *     W1 [64, ldw]       NATIVE edge_mlp.0.weight, ldw = 2*din + 1 + Fe; only the radial / edge-feature columns are read
 *     W2,b2 = edge_mlp.2 ; Wc1,bc1 = coord_mlp.0 ; wc2 [64] = coord_mlp.2.weight
 *     h [N, ld_h]        the layer's input node features (din = 20 | 64 columns); fpack = its forward operand pack
 *     z2s, z3s [max(E,16), 64], zn1 [N,64]   out: pre-activations saved for the backward, opaque to the caller (z2s, z3s and
 *                        m1s keep channel 16 nt + r at column 4 r + nt of a row: the MFMA accumulator layout then moves 16
 *                        contiguous bytes per lane; only is_egnn_layer_bwd reads them) (z2s == NULL: none saved;
 *                        z3s == NULL: z3 is not saved -- is_egnn_layer_bwd then recomputes it from z2, the default)
 *     wg_clock           NULL, or [nchunks / 4][2] int64: every workgroup's start / end device wall clock (100 MHz)
 *     x_out == NULL      the coordinate branch is not evaluated (last layer of a stack whose coordinates are unused,
 *                        hybrid_models.py:323-324); z3s is then unused.   psd_next == NULL: no next projection.
 *
 * is_egnn_layer_bwd   per tile of destination nodes: (optional) source-side gather of the layer ABOVE -> node data path
 *                     -> edge pass backward.
 *     edge half: forward arguments + z2s / z3s (z3s == NULL: z3 = SiLU(z2) Wc1^T + bc1 is recomputed per tile, bc1 =
 *       coord_mlp.0.bias required then; otherwise bc1 is unused); out: dZ1 [E,64], dD [E,3] (per-edge gradients of the first edge-MLP
 *       pre-activation and of x_src - x_dst, CSR slot order: gathered by source by the NEXT call or by
 *       is_gather_segment_sum), dPd [N, ld_dpd] and dx [N,3] (destination-side parts, identity path included), ONE
 *       partial weight-gradient record per workgroup (`grid` persistent workgroups, 8960 floats):
 *         dW2 [64,64] | dWc1 [64,64] | db2 | dbc1 | dwc2 | dw_r [64] | dW_a [64,8]
 *       A tile = 16 consecutive destination nodes (the greedy tile list of rounds 2 - 5 was removed in round 6).
 *     node half: g_h [N,64] direct gradient of the layer's output h (may be NULL); g_psd [N,128] gradient of the next
 *       pre-projection of h (NULL: none, then dh = g_h); zn1; bpack; out: dh_total (with g_psd), dzn1, d_h (first din
 *       columns, may be NULL), d_hn [N,64] (scratch: dL/dh_neigh, read back by the edge half).
 *     dZ1n != NULL: dZ1n / dDn / dxn = dZ1 / dD / dx of the layer above, rowptr_src / pos_by_src = CSR by source; the call
 *       completes g_psd[:, :64] = gather(dZ1n) (WRITTEN: the weight-gradient launch reads it) and uses dxn + gather(dDn)
 *       (gxtot [N,3], scratch) as the coordinate gradient; g_xout must be NULL.  Otherwise g_xout [N,3] or NULL (no
 *       coordinate gradient: the coordinate-MLP half is skipped, z3s / Wc1 / wc2 are not read).                       */
int is_node_pack_floats(void);
int is_stack_prologue(const void* jobs, int njobs, const float* h, int ld_h, int din, const float* W1, int ldw,
                      const float* b0, const float* b1, float* psd, const float* x_src, int ld_x, float* x_dst, int N,
                      void* stream);
int is_egnn_layer_fwd(const float* ps, const float* pd, int ld_p, const float* x, const float* ea,
                      const int32_t* rowptr, const int32_t* srcs, const int32_t* dsts,
                      const int32_t* chunk_ptr, int nchunks, const float* W1, int ldw, int din,
                      const float* W2, const float* b2, const float* Wc1, const float* bc1,
                      const float* wc2, float* h_neigh, int ld_hn, float* x_out, float* z2s,
                      float* z3s, int N, int E, int Fe, const float* h, int ld_h, const float* bn1,
                      const float* bn2, const float* b0n, const float* b1n, const float* fpack,
                      float* zn1, float* h_out, float* psd_next, long long* wg_clock, float* m1s, float* dy1s,
                      float* geos, void* stream);
int is_egnn_layer_bwd(const float* ps, const float* pd, int ld_p, const float* x, const float* ea,
                      const int32_t* rowptr, const int32_t* srcs, const float* W1, int ldw, int din,
                      const float* W2, const float* Wc1, const float* bc1, const float* wc2, const float* z2s,
                      const float* z3s, const float* g_xout, float* dZ1, float* dD, float* dPd, int ld_dpd,
                      float* dx, float* partials, int grid, int N, int Fe,
                      const float* dZ1n, const float* dDn, const float* dxn, const int32_t* rowptr_src,
                      const int32_t* pos_by_src, const float* g_h, float* g_psd, const float* zn1,
                      const float* bpack, float* dh_total, float* dzn1, float* d_h, float* d_hn, float* gxtot,
                      long long* wg_clock, const float* m1s, const float* dy1s, const float* geos, void* stream);
/* The PAIRED form of is_egnn_layer_bwd (csrc/egnn_layer_bwd8.hip): `grid` workgroups of 512 threads -- two groups of four waves that
 * are what two co-resident 256-thread workgroups of is_egnn_layer_bwd were (group g of workgroup b owns the tiles of its workgroup
 * b + g * grid), sharing one staged copy of the weight tiles (requested at the kernel's start, under the node phase) and writing ONE
 * partial record per workgroup (`partials` holds `grid` records: half the bytes into is_reduce_partials_batched).  Same arguments,
 * outputs and per-tile arithmetic as is_egnn_layer_bwd.  Covers the default build (z1 / geometry / z3 read back) and Fe <= 1:
 * is_egnn_layer_bwd_paired_supported(Fe) tells, -38 otherwise; grid <= ceil(N/16).                                                   */
int is_egnn_layer_bwd_paired_supported(int Fe);
int is_egnn_layer_bwd_paired(const float* ps, const float* pd, int ld_p, const float* x, const float* ea,
                             const int32_t* rowptr, const int32_t* srcs, const float* W1, int ldw, int din,
                             const float* W2, const float* Wc1, const float* bc1, const float* wc2, const float* z2s,
                             const float* z3s, const float* g_xout, float* dZ1, float* dD, float* dPd, int ld_dpd,
                             float* dx, float* partials, int grid, int N, int Fe,
                             const float* dZ1n, const float* dDn, const float* dxn, const int32_t* rowptr_src,
                             const int32_t* pos_by_src, const float* g_h, float* g_psd, const float* zn1,
                             const float* bpack, float* dh_total, float* dzn1, float* d_h, float* d_hn, float* gxtot,
                             long long* wg_clock, const float* m1s, const float* dy1s, const float* geos, void* stream);
/* How this library's backward layer kernel gets the first edge-MLP activation (a build-time choice, csrc/Makefile M1=0|1|2):
 * 2 (default): it READS the pre-activation z1 from m1s [max(E,16), 64] -- pass the same array to is_egnn_layer_fwd, which fills
 * it (dy1s = NULL in both calls); 1: it reads m1 = SiLU(z1) from m1s and SiLU'(z1) from dy1s (both filled by the forward);
 * 0: it recomputes z1 from gathered rows (m1s / dy1s may be NULL).  The backward's windows are issue-bound, HBM is not.   */
int is_layer_saves_m1(void);
/* 1 (default build, csrc/Makefile GEO=0|1): the backward layer kernel READS the edge geometry (x_src - x_dst, |.|^2) from geos
 * [max(E,16), 4], which is_egnn_layer_fwd fills (pass the same array to both); 0: it recomputes it from the coordinates.  */
int is_layer_saves_geo(void);

/* Node pre-projection on its own (any caller of a 128-wide two-bias projection of 64-wide rows: with W1 = [Wq | Wk]
 * it is the fused query/key projection of the node attention, models/layers.py:13-16,68) and its backward:
 *   is_node_proj_fwd : psd [N,128] = [h W1s^T + b0 | h W1d^T + b1], h [N, ld_h] with din in {20, 64}; b0 may be NULL
 *   is_node_proj_bwd : dh_total [N,64] = g_h + g_psd W1sd (either may be NULL: g_h treated as 0, dh_total skipped);
 *                      partial record per workgroup dW1sd [128,64] | db1 [64] | db0 [64]  (8320 floats)            */
int is_node_proj_fwd(const float* h, int ld_h, int din, const float* W1, int ldw, const float* b0,
                     const float* b1, float* psd, int N, void* stream);
int is_node_proj_bwd(const float* g_h, const float* g_psd, const float* h, int ld_h, int din,
                     const float* W1, int ldw, float* dh_total, float* partials, int grid, int N,
                     void* stream);

/* dst[map[i]] = sum_p partials[p*stride + i] for i < count, fixed summation order (deterministic);
 * map may be NULL (identity), entries < 0 are skipped.  scratch: is_reduce_partials_scratch_floats(count). */
int is_reduce_partials_scratch_floats(int stride);
int is_reduce_partials(const float* partials, int nparts, int stride, int count, const int32_t* map,
                       float* dst, float* scratch, void* stream);

/* Batched forms (one launch for all layers of a stack / for all pending reductions).
 *   is_egnn_node_wgrad_batched: the weight gradients of the node blocks as streaming outer products over the rows;
 *     `layers` = host array of nlayers (<= 8) records
 *       { const float *g_psd, *h_out, *dh, *zn1, *dzn1, *h, *h_neigh; float* partials;
 *         int ld_h, din, ld_hn, ld_ho, dho, pad; }
 *     partial record = [dW1sd 128x64 | db1 | db0] (g_psd^T h_out, is_egnn_node_wgrad_proj_floats floats: the PROJ part,
 *     written by `grid_proj` workgroups = records 0 .. grid_proj - 1) followed by [dWn1 64x128 | dWn2 64x64 | dbn1 | dbn2] (the
 *     NODE part, `grid_node` workgroups = records 0 .. grid_node - 1); record stride is_egnn_node_wgrad_stride; `partials`
 *     holds max(grid_node, grid_proj) records.  h_out has row stride ld_ho and dho (<= 64) valid columns; dzn1 == NULL marks a
 *     projection-only job (only the dW1sd part, e.g. the layer-0 pre-projection of the raw node features), g_psd ==
 *     NULL a job without projection part.
 *   is_reduce_partials_batched: `jobs` = host array of njobs (<= 24) records
 *       { const float* partials; const int32_t* map; float* dst; float* scratch; int nparts, stride, count, pad; }
 *     each processed exactly like is_reduce_partials (scratch: is_reduce_partials_scratch_floats(count) floats). */
int is_egnn_node_wgrad_stride(void);
int is_egnn_node_wgrad_proj_floats(void);
int is_egnn_node_wgrad_batched(const void* layers, int nlayers, int grid_node, int grid_proj, int N, void* stream);
int is_reduce_partials_batched(const void* jobs, int njobs, void* stream);

/* Latent block of the sequence VAE (models/hybrid_models.py:297-308,334-340): a1 [B,Hd] = vae_fc1(x) (pre-activation) ->
 *   mu = W21 relu(a1) + b21, logvar = W22 relu(a1) + b22 [B,32]; z = mu + eps * exp(0.5 logvar) (eps [B,32] from the caller);
 *   zp [B, 32 + P] = [z | p] (p [B,P] the property embedding, P <= 16, NULL when P == 0); h3 [B,Hd] = relu(W3 zp + b3).
 * L must be 32, Hd a multiple of 16 <= 2048.  Backward: upstream g_h3 / g_mu / g_lv / g_zp (each may be NULL) -> d_a1 [B,Hd],
 * d_p [B,P], wgrad = [dW21 (32 x Hd) | dW22 | db21 | db22 | dW3 (Hd x (32 + P)) | db3] (is_vae_latent_grad_floats floats,
 * contraction over the batch in a fixed order); d_a3 [B,Hd], dmu, dlv [B,32] are scratch.                              */
int is_vae_latent_fwd(const float* a1, const float* W21, const float* b21, const float* W22, const float* b22,
                      const float* eps, const float* p, int P, const float* W3, const float* b3, float* mu,
                      float* logvar, float* zp, float* h3, int B, int Hd, int L, void* stream);
int is_vae_latent_grad_floats(int Hd, int P);
int is_vae_latent_bwd(const float* g_h3, const float* h3, const float* g_mu, const float* g_lv, const float* g_zp,
                      const float* eps, const float* logvar, const float* a1, const float* zp, const float* W21,
                      const float* W22, int P, const float* W3, float* d_a3, float* dmu, float* dlv, float* d_p,
                      float* d_a1, float* wgrad, int B, int Hd, int L, void* stream);
/* the two halves of is_vae_latent_bwd as their own entry points (data path: one launch; weight pass: one), for callers
 * that put other work of the data path between them (functional.VaeLatentFn: the weight gradient of vae_fc1) */
int is_vae_latent_bwd_data(const float* g_h3, const float* h3, const float* g_mu, const float* g_lv, const float* g_zp,
                           const float* eps, const float* logvar, const float* a1, const float* W21, const float* W22,
                           int P, const float* W3, float* d_a3, float* dmu, float* dlv, float* d_p, float* d_a1, int B,
                           int Hd, int L, void* stream);
int is_vae_latent_bwd_wgrad(const float* a1, const float* dmu, const float* dlv, const float* zp, const float* d_a3,
                            int P, float* wgrad, int B, int Hd, int L, void* stream);

/* out_rows[v, 0:64] = sum_{p in [ptr[v], ptr[v+1])} rows[pos[p], 0:64]   (written)
 * out_vec3[v, 0:3] += sum_{p} vec3[pos[p], 0:3]                          (accumulated; vec3 may be NULL)
 * wg_clock: NULL, or [(N + 15) / 16][2] int64 -- every workgroup's start / end device wall clock (100 MHz), as for the two
 * layer kernels: the in-situ launch timing bench.py's roofline uses (works inside a replayed HIP graph) */
int is_gather_segment_sum(const float* rows, const float* vec3, const int32_t* ptr, const int32_t* pos,
                          float* out_rows, int ld_out, float* out_vec3, int N, long long* wg_clock, void* stream);

/* Paired cancer / wild-type contrastive loss (utils/contrastive.py:18-83), forward and backward.
 * emb_c, emb_w [B, ld_e] (E valid columns; E = 104), pos [B] (1.0 where the pair is immunogenic), projector
 * W1 [128, E] (Linear, no bias), gamma / beta [128] (BatchNorm1d on batch statistics), W2 [128, 128]; lambda = weight of
 * the off-diagonal terms.  loss [1].  scratch (is_contrastive_scratch_floats(B) floats) is kept for the backward,
 * which needs a work buffer of is_contrastive_work_floats(B) floats and the upstream gradient g_loss [1] on the device,
 * and writes d loss / d emb_c, d loss / d emb_w [B, ld_d] (the projector is frozen: no parameter gradients).
 * 2 <= B <= 256, E <= 256.                                                                                     */
long long is_contrastive_scratch_floats(int B);
long long is_contrastive_work_floats(int B);
/* gate [1] (device; NULL: 1) and scale: loss = scale * gate * L and the backward multiplies g_loss alike -- the caller's loss
 * coefficient and the early-out factor of is_contrastive_targets folded into the launches.                          */
int is_contrastive_fwd(const float* emb_c, const float* emb_w, int ld_e, int E, const float* pos, const float* W1,
                       const float* gamma, const float* beta, const float* W2, float lambda, float* scratch,
                       float* loss, const float* gate, float scale, int B, void* stream);
int is_contrastive_bwd(const float* pos, const float* W1, const float* gamma, const float* W2, float lambda,
                       const float* scratch, float* work, const float* g_loss, const float* gate, float scale,
                       float* demb_c, float* demb_w, int ld_d, int E, int B, void* stream);

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

/* Input gradient of a Linear layer with a long contraction and a small output (vae_fc4: 5943 -> 512 going backward;
 * models/hybrid_models.py:306-308): gx [B, K] = gy [B, ld_g] (N columns) W [N, ld_w] (K columns; the weight as nn.Linear
 * stores it).  The contraction is cut into chunks of 96, one workgroup per (64 batch rows, 64 output columns, chunk), partial results summed
 * in chunk order by a second launch.  scratch: is_linear_dgrad_scratch_floats(B, N, K) floats.                          */
/* ... and the forward of a layer with a long contraction and a small output (vae_fc1: 5943 -> 512; models/hybrid_models.py:297):
 * y [B, K] = x [B, ld_x] (N columns) W^T + bias, W [K, ld_w] as nn.Linear stores it, bias [K] or NULL; same split of the
 * contraction, same scratch size (is_linear_dgrad_scratch_floats(B, N, K)).                                            */
int is_linear_fwd_long(const float* x, int ld_x, const float* W, int ld_w, const float* bias, float* y, float* scratch,
                       int B, int N, int K, void* stream);
long long is_linear_dgrad_scratch_floats(int B, int N, int K);
int is_linear_dgrad(const float* gy, int ld_g, const float* W, int ld_w, float* gx, float* scratch, int B, int N, int K,
                    void* stream);

/* Multi-tensor Adam / AdamW step with torch.optim semantics (reference: torch.optim.Adam in train_IEDB_wFT.py:69-74,
 * torch.optim.AdamW in train_Cancer_wFT.py:76-92).  `chunks` = DEVICE array of nchunks records
 * { float* p; const float* g; float* m; float* v; long long n; } (one workgroup each), `state` = device float[3]
 * {step count, derived step size, derived sqrt(1 - beta2^t)} updated by the call, `hyper` = device float[8]
 * {lr, beta1, beta2, eps, weight_decay, decoupled (AdamW) flag, gradient scale (1 = none), unused}.
 * Capturable in a HIP graph.                                                                                   */
int is_adam_step(const void* chunks, int nchunks, float* state, const float* hyper, void* stream);
/* The same step in parts (is_adam_step == is_adam_prepare + is_adam_apply): prepare advances the step count and derives the
 * step's scalars once; apply updates the parameters of ONE chunk table with them -- for a caller that updates a group's
 * parameters in several launches (those whose gradients are final early beside the tail of the backward, the rest after it). */
int is_adam_prepare(float* state, const float* hyper, void* stream);
int is_adam_apply(const void* chunks, int nchunks, const float* state, const float* hyper, void* stream);

/* Debug aid: one single-thread launch that writes the device wall clock (100 MHz) to *slot; can be captured in a
 * HIP graph to time-stamp points of a replayed step without a profiler attached.                             */
int is_debug_timestamp(long long* slot, void* stream);

/* Debug aid (tools/dp_overlap_emulation.py): a stand-in for an RCCL all-reduce on a single GPU -- `grid` persistent
 * workgroups of 512 threads that stream buf[0, n) `passes` times in place (values unchanged) and hold their CU slots for
 * `ticks` of the 100 MHz device clock: occupies slots AND HBM bandwidth like a collective's kernel does.  elapsed: NULL or
 * [grid] int64, the ticks every workgroup spent streaming (longer than `ticks`: the stand-in overran).        */
int is_debug_emulated_collective(float* buf, long long n, int grid, int passes, long long ticks, long long* elapsed,
                                 void* stream);

/* Measurement aid (bench.py: the copy-bandwidth ceiling HBM fractions are also quoted against):
 * dst[0, n16) = src[0, n16) in 16-byte words, `grid` workgroups of 256 threads, non-temporal accesses; moves 32 * n16 bytes. */
int is_debug_stream_copy(const void* src, void* dst, long long n16, int grid, void* stream);

/* Batched device-to-device copy (hand-over of a device-resident batch into the static buffers of a captured
 * graph): `jobs` = host array of njobs (<= 24) records { const void* src; void* dst; long long bytes; },
 * bytes a multiple of 4.                                                                                   */
int is_multi_copy(const void* jobs, int njobs, void* stream);

/* The random tensors of one train step in ONE launch from a device-resident generator state (captured steps: no host-side
 * generator, hence no seed / offset fills in front of every graph replay): scaled keep-masks of nn.Dropout(p) in training mode
 * (reference models/hybrid_models.py:277-295) and N(0, 1) noise for the reparameterisation (hybrid_models.py:301-304).
 * `jobs` = host array of njobs (<= 8) records { float* out; long long n; int kind; float p; } (kind 0: normal, 1: keep-mask
 * scaled by 1 / (1 - p)); state = 3 x uint64 of device memory { seed, step counter, ticket }: Philox4x32-10, every value a
 * function of (seed, step, job, element); the launch advances the step counter.                                              */
int is_step_random(const void* jobs, int njobs, unsigned long long* state, void* stream);

/* On-device batcher (reference data/utils.py:160-176: `collate` -> dgl.batch): assemble the block-diagonal batch of the
 * B graphs idx[0..B) (int64, device) from a device-resident dataset of per-graph CSR pieces, all graphs padded to n
 * nodes: x_all [G][n][F]; eoff [G+1] edge offsets; rowptr_dst_all / rowptr_src_all [G][n+1] (0-based per graph);
 * src_all / dst_all [Etot] local node ids in destination order; pos_all [Etot] local slot ids in source order;
 * ea_all [Etot][Fe].  Writes the batch arrays (node ids + slot*n, edge slots + running edge offset); the destination
 * edge arrays must hold the selected graphs' edges (B * max edges per graph is always enough).
 * rows: host array of nrows (<= 4) records { const float* src; float* dst; int floats, pad; }: per-sample rows that ride
 * along in the same launch, dst[i][:] = src[idx[i]][:] (sequence one-hots, property vectors, targets).            */
int is_batch_gather(const long long* idx, int B, int n, int F, int Fe, const float* x_all, const int32_t* eoff,
                    const int32_t* rowptr_dst_all, const int32_t* rowptr_src_all, const int32_t* src_all,
                    const int32_t* dst_all, const int32_t* pos_all, const float* ea_all, float* x,
                    int32_t* rowptr_dst, int32_t* rowptr_src, int32_t* src_sorted, int32_t* dst_sorted,
                    int32_t* pos_by_src, float* ea, const void* rows, int nrows, void* stream);
/* rowptr [N+1] (device) -> chunk_ptr [k+1][2] int32 = (b_j, rowptr[b_j]), b_0 = 0, b_k = N, b_j = first node whose first
 * in-edge index is >= j * E / k: the edge-balanced node partition the layer kernels walk, recomputed on the device after
 * the batcher wrote a new rowptr (one launch, no host sync).  shares = 1 (the caller: only for the full grid of 2048 wave
 * chunks): when P = ceil(edges per wave pair / 16) is odd and the node-aligned chunks have slack, the chunks of the first /
 * second half of the workgroups get (P + 1) / 2 : (P - 1) / 2 parts of the edges (the b_j follow E * W(j) / W(k)) -- workgroups
 * i and i + 256 share a CU, so each SIMD then walks P instead of P + 1 tiles (immunostruct_amd/graph.py chunk_shares).      */
int is_chunk_partition(const int32_t* rowptr, int N, int k, int shares, int32_t* chunk_ptr, void* stream);

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
/* Backward of is_attn_colmean_fwd_tail as ONE launch: gy [B,64] = gradient of y; every graph's workgroup derives
 * g_ctx = W_v^T W_c^T gy itself and runs the attention backward; one extra workgroup contracts the B samples (ascending
 * order) into gtail = dW_v [64,64] | db_v [64] | dW_c [64,64] | db_c [64] from pooled (= ctx) [B,64] and a1 (= hid) [B,64]. */
int is_attn_colmean_bwd_tail(const float* qk, const float* x, const float* abar, const float* probs, const float* gy,
                             const float* wv, const float* wc, const float* pooled, const float* a1, float* dqk, float* dx,
                             float* gtail, int B, int n, void* stream);

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
/* parts: host array of nparts (<= 4) records { const float* x; float* dx; int width, ld; }: the T = sum(width) scalar tokens of
 * graph b are the rows b of these pieces side by side ([x_gat | z_vae], or the four pieces of a pair) -- read where they are, and
 * the backward writes each piece's gradient into its own tensor dx (no concatenation in front, no slice copies behind).   */
int is_comb_attn_fwd(const void* parts, int nparts, const float* wq, const float* bq, const float* wk, const float* wv,
                     const float* bv, const float* Wc, const float* bc, float* z, float* stats, int B,
                     int T, int F, void* stream);
int is_comb_attn_bwd(const void* parts, int nparts, const float* stats, const float* dz, const float* wq, const float* bq,
                     const float* wk, const float* wv, const float* bv, const float* Wc, const float* bc,
                     float* partials, float* grads, int B, int T, int F, void* stream);

/* The combined attention with the classifier y = act2(W2 (mask * ReLU(W1 z + b1)) + b2) behind it (models/hybrid_models.py:
 * 288-295, 344-350: Flatten, Linear(T, hid), ReLU, Dropout, Linear(hid, out)) as ONE launch forward and ONE (+ the finish
 * launch) backward.  W1 [hid, T], W2 [out, hid]; mask [B, hid] = scaled dropout keep-mask or NULL; hid <= 32, out <= 64.
 * Forward outputs: z [B, T], stats (as is_comb_attn_fwd), a1 [B, hid] (ReLU output), y [B, out].
 * Backward: gy [B, out] -> parts[].dx, grads (layout of is_comb_attn_bwd), gcls = dW1 [hid*T] | db1 | dW2 [out*hid] | db2
 * (is_comb_attn_cls_grad_floats); samples are contracted in ascending order by one extra workgroup.                      */
int is_comb_attn_cls_fwd(const void* parts, int nparts, const float* wq, const float* bq, const float* wk, const float* wv,
                         const float* bv, const float* Wc, const float* bc, const float* W1, const float* b1,
                         const float* W2, const float* b2, const float* mask, float* z, float* stats, float* a1, float* y,
                         int B, int T, int F, int hid, int out, int act2, void* stream);
int is_comb_attn_cls_grad_floats(int T, int hid, int out);
int is_comb_attn_cls_bwd(const void* parts, int nparts, const float* stats, const float* gy, const float* wq, const float* bq,
                         const float* wk, const float* wv, const float* bv, const float* Wc, const float* bc,
                         const float* W1, const float* W2, const float* mask, const float* z, const float* a1,
                         const float* y, float* partials, float* grads, float* gcls, int B, int T, int F, int hid, int out,
                         int act2, void* stream);

/* floats of scratch is_vae_loss needs */
int is_loss_partials_floats(void);
/* mode 0: c_pred*MSE(logit,y), mode 1: c_pred*BCEWithLogits(logit,y,pos_weight);
 * + c_mse*MSE(recon,x) + c_kld*(-0.5*mean(1+logvar-mu^2-exp(logvar))).
 * recon/x/d_recon may be NULL with recon_total = 0, mu/logvar likewise with latent_total = 0.
 * out[4] = {total, prediction term, recon MSE, KLD}; total (may be NULL) receives out[0] as well (its own buffer:
 * the differentiable result); d_* receive d total / d input.
 * recon = NULL with recon_total > 0: partials[] already holds stage 1 of the reconstruction term, produced by
 * is_recon_mse (partial sums of (recon - x)^2 and d_recon = gscale * (recon - x), gscale = c_mse * 2 / recon_total) --
 * for callers that run stage 1 where recon is produced (the sequence branch's stream), ahead of the prediction.      */
int is_recon_mse(const float* recon, const float* x, float* d_recon, long long recon_total, float gscale,
                 float* partials, void* stream);
int is_vae_loss(const float* recon, const float* x, float* d_recon, long long recon_total,
                const float* mu, const float* logvar, float* d_mu, float* d_logvar, int latent_total,
                const float* logit, const float* y, float* d_logit, int batch, int mode,
                float pos_weight, float c_pred, float c_mse, float c_kld, float* partials, float* out, float* total,
                void* stream);


#ifdef __cplusplus
}
#endif
#endif
