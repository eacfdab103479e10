// One EGNNConv layer, forward, as ONE launch: the fused edge pass (gather, edge MLP, coordinate MLP, segment sums by
// destination) followed, inside the same workgroup, by the node MLP of the workgroup's own nodes and the NEXT layer's
// node pre-projection (or the node attention's query / key projection after the last layer).
//
// Edge half -- wave-autonomous and software-pipelined:
//   * the destination nodes are cut into `nchunks` contiguous, NODE-ALIGNED ranges with (nearly) equal
//     edge counts (chunk_ptr rows = (first node, its first edge), built once per batch next to the CSR index).  One WAVE owns one chunk and
//     walks its edges in 16-edge tiles that ignore node boundaries, so tiles are full (the node-tiled
//     kernels left ~25 % of the matrix rows empty on degree-3 graphs) and no wave ever waits for another
//     one: the only workgroup barrier is the one-time weight staging;
//   * the segment sums by destination are a running, wave-uniform scan over the tile's rows (flush
//     points come from the destination ids held in scalar registers) -- fixed edge order, no atomics;
//   * the index loads of tile j+2 and the row gathers of tile j+1 are issued before tile j is computed,
//     so the dependent memory latencies (edge ids -> rows -> use) overlap the arithmetic of a tile;
//   * measured on gfx950 (tools/ubench/mfma_rate.hip): v_mfma_f32_16x16x4_f32 and plain VALU fp32 ops do
//     NOT overlap -- fp32 matrix math runs at the vector-ALU rate -- so the kernel is bound by the SUM of
//     its MFMA and VALU cycles.  Hence the instruction diet: every gather / store is a raw buffer access
//     with a scalar row offset (no 64-bit vector address arithmetic), every tile is computed at full
//     width (no predication), reductions use DPP.
//
// Every tile is computed at full width.  The last tile of a chunk is shifted back so that it ends at the
// chunk's last edge: its leading rows repeat edges that were already processed (by this wave or by the
// wave owning the previous chunk); they are recomputed bit-identically, re-stored with the same values,
// and skipped by the segment scan.  Only a graph with fewer than 16 edges in total has rows past the end
// (clamped loads, stores into the >= 16-row padding of z2s / z3s).
//
// Node half.  The four chunks of a workgroup are consecutive, so together its waves hold the complete h_neigh of one
// contiguous node range: after ONE workgroup barrier (h_neigh rows are re-read from L2 -- waves of a workgroup share
// their CU's L1) the workgroup runs zn1 = [h | h_neigh] Wn1^T + bn1, h' = SiLU(zn1) Wn2^T + bn2 and
// psd' = [h' W1s'^T | h' W1d'^T + b1'] for those nodes, up to 64 rows per pass (empty 16-row tiles are skipped): wave w produces output columns [16w, 16w + 16)
// with its MFMA B operands in registers, fetched from the lane-ordered operand pack (node16.h) after the edge loop has
// released its registers; the activation tiles live where the edge half kept its weight tiles.  This replaces the
// separate node kernel (one launch and one h_neigh round trip through HBM per layer).
#include "common.h"
#include "node16.h"

namespace is {

constexpr int W3 = 4;  // waves per workgroup (they only share the LDS weight tiles)

#ifdef IS_STAGE_STAMPS
__device__ long long g_stamps3[64];
#define STAMP3(k) do { if (blockIdx.x == 300 && threadIdx.x == 0) g_stamps3[k] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMP3N() do { STAMP3(stamp_k); ++stamp_k; } while (0)
#else
#define STAMP3(k) do { } while (0)
#define STAMP3N() do { } while (0)
#endif

struct Fwd3Weights { float w2[H * LD]; float wc1[H * LD]; };

template <int FE_MAX>
struct Fwd3Smem {
  Fwd3Weights w;
  float act[W3][TE16 * LD];
  float e_rad[W3][TE16];
  float e_xd[W3][3][TE16];
  float e_s[W3][TE16];
  float e_a[W3][FE_MAX][TE16];
  float e_xdst[W3][3][TE16];      // the tile's destination coordinates (the coordinate update's base, read at the flush points)
};

template <int FE_MAX>
struct EdgeIds {     // lanes 16..63 mirror lanes 0..15 (lane & 15 = edge of the tile)
  int s, d;          // source, destination
  int flush;         // 1 if this edge closes its destination node inside this tile's scan range
  float a[FE_MAX];
};

struct FwdRows {     // prefetched operands of one tile
  float gs[TE16];    // Ps[src] rows, lane = channel
  float gd[TE16];    // Pd[dst] rows, lane = channel
  float xs[3], xd[3];  // coordinates, lane & 15 = edge
};

struct FwdBufs {
  rsrc_t ps, pd, x, srcs, dsts, ea, hn, z2, z3, m1, d1, geo, xo;
};

__device__ __forceinline__ int tile_start(int cb, int e1) { return (cb + TE16 <= e1) ? cb : max(e1 - TE16, 0); }

// All prefetch loads are UNCONDITIONAL (indices clamped to the E slots of the batch): with predicated loads
// the compiler cannot count the loads in flight and falls back to s_waitcnt vmcnt(0) in front of the first
// use -- which would serialise the pipeline again.
template <int FE_MAX>
__device__ __forceinline__ EdgeIds<FE_MAX> load_edge_ids(const FwdBufs& B, int Fe, int cb, int e1, int E, int lane) {
  EdgeIds<FE_MAX> id;
  const int ts = tile_start(cb, e1);
  const int el = ts + (lane & (TE16 - 1));
  const int ec = min(el, E - 1);
  id.s = buf_load_i(B.srcs, ec * 4, 0);
  id.d = buf_load_i(B.dsts, ec * 4, 0);
  const int dn = buf_load_i(B.dsts, min(el + 1, E - 1) * 4, 0);
  const bool mine = el >= cb && el < e1;     // rows the segment scan of this tile accumulates
  id.flush = (mine && (el + 1 >= e1 || dn != id.d)) ? 1 : 0;
#pragma unroll
  for (int f = 0; f < FE_MAX; ++f) {
    // ea is a valid buffer even when Fe == 0; masked by multiplication so that the load stays unconditional
    id.a[f] = buf_load(B.ea, (ec * Fe + (f < Fe ? f : 0)) * 4, 0) * (f < Fe ? 1.0f : 0.0f);
  }
  return id;
}

template <int FE_MAX>
__device__ __forceinline__ void load_fwd_rows(FwdRows& rw, const EdgeIds<FE_MAX>& id, const FwdBufs& B, int ld_p_bytes,
                                              int lane) {
#pragma unroll
  for (int i = 0; i < TE16; ++i) {
    rw.gs[i] = buf_load(B.ps, lane * 4, __builtin_amdgcn_readlane(id.s, i) * ld_p_bytes);
    rw.gd[i] = buf_load(B.pd, lane * 4, __builtin_amdgcn_readlane(id.d, i) * ld_p_bytes);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    rw.xs[k] = buf_load(B.x, id.s * 12 + k * 4, 0);
    rw.xd[k] = buf_load(B.x, id.d * 12 + k * 4, 0);
  }
}

// COORD = false: the layer's coordinate output is not wanted (last layer of a stack whose final coordinates are unused):
// the coordinate MLP (z3 = mh Wc1^T + bc1, s = SiLU(z3) . wc2), the z3s store and the x_out update are skipped.
template <int FE_MAX, bool SAVE, bool COORD, int DIN>
__global__ __launch_bounds__(256, 2) void egnn_layer_fwd_kernel(
    const float* __restrict__ ps, const float* __restrict__ pd, int ld_p,
    const float* __restrict__ x, const float* __restrict__ ea,
    const int* __restrict__ rowptr, const int* __restrict__ srcs, const int* __restrict__ dsts,
    const int* __restrict__ chunk_ptr, int nchunks,
    const float* __restrict__ W1, int ldw, int din,
    const float* __restrict__ W2, const float* __restrict__ b2,
    const float* __restrict__ Wc1, const float* __restrict__ bc1, const float* __restrict__ wc2,
    float* __restrict__ h_neigh, int ld_hn, float* __restrict__ x_out,
    float* __restrict__ z2s, float* __restrict__ z3s, int E, int Fe,
    const float* __restrict__ h, int ld_h, const float* __restrict__ bn1, const float* __restrict__ bn2,
    const float* __restrict__ b0n, const float* __restrict__ b1n, const float* __restrict__ fpack,
    float* __restrict__ zn1, float* __restrict__ h_out, float* __restrict__ psd_next, long long* __restrict__ wg_clock,
    float* __restrict__ m1s, float* __restrict__ dy1s, float* __restrict__ geos) {
  __shared__ Fwd3Smem<FE_MAX> sm;
  wg_clock_start(wg_clock);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform => scalar registers, scalar address math
  const int r = lane & 15, q = lane >> 4;
  STAMP3(0);
#ifdef IS_STAGE_STAMPS
  int stamp_k = 2;
#endif
#ifdef IS_ABL_STORES
  constexpr bool ABL_SAVE = false;
#else
  constexpr bool ABL_SAVE = true;
#endif
  const bool save3 = ABL_SAVE && SAVE && COORD && z3s != nullptr;      // kernel-uniform: z3 is saved only for a backward that does not recompute it
  // kernel-uniform: the first edge-MLP activation m1 = SiLU(z1) and its derivative are saved for a backward that reads them
  // back instead of recomputing z1 from gathered rows (the backward's windows are issue-bound: HBM has room, the SIMDs do not)
  const bool save1 = ABL_SAVE && SAVE && m1s != nullptr;

  // ---- chunk of this wave ----
  const int c = blockIdx.x * W3 + wave;
  int va = 0, vb = 0, e0 = 0, e1 = 0;
  if (c < nchunks) {      // chunk_ptr rows = (node boundary, first edge of that node): one load level for both ranges
    va = __builtin_amdgcn_readfirstlane(chunk_ptr[2 * c]);
    e0 = __builtin_amdgcn_readfirstlane(chunk_ptr[2 * c + 1]);
    vb = __builtin_amdgcn_readfirstlane(chunk_ptr[2 * c + 2]);
    e1 = __builtin_amdgcn_readfirstlane(chunk_ptr[2 * c + 3]);
    if (va >= vb) { e0 = 0; e1 = 0; }
  }
  FwdBufs B;
  B.ps = make_rsrc(ps); B.pd = make_rsrc(pd); B.x = make_rsrc(x); B.srcs = make_rsrc(srcs); B.dsts = make_rsrc(dsts);
  B.ea = make_rsrc(ea); B.hn = make_rsrc(h_neigh); B.z2 = make_rsrc(z2s); B.z3 = make_rsrc(z3s != nullptr ? z3s : z2s);
  B.m1 = make_rsrc(m1s != nullptr ? m1s : z2s); B.d1 = make_rsrc(dy1s != nullptr ? dy1s : z2s);
  B.geo = make_rsrc(geos != nullptr ? geos : z2s);
  B.xo = make_rsrc(x_out != nullptr ? x_out : h_neigh);
  const bool save_geo = ABL_SAVE && SAVE && geos != nullptr;      // kernel-uniform: (x_src - x_dst, |.|^2) per edge slot, for the backward
  const int ld_p_bytes = ld_p * 4, ld_hn_bytes = ld_hn * 4;

  // first index / row loads are in flight while the weights are staged
  const bool has_edges = e0 < e1;     // wave-uniform
  EdgeIds<FE_MAX> id0 = {}, id1 = {};
  if (has_edges) {
    id0 = load_edge_ids<FE_MAX>(B, Fe, e0, e1, E, lane);
    id1 = load_edge_ids<FE_MAX>(B, Fe, e0 + TE16, e1, E, lane);
  }
  {
    f32x4 wreg[8];
    constexpr int NW = COORD ? 8 : 4;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int idx = tid + (j & 3) * 256;                      // 1024 float4 per matrix, 256 threads
      const float* src = (j < 4) ? W2 : Wc1;
      wreg[j] = *reinterpret_cast<const f32x4*>(src + idx * 4);
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int idx = tid + (j & 3) * 256;
      const int row = idx / (H / 4), c4 = (idx % (H / 4)) * 4;
      float* dst = (j < 4) ? sm.w.w2 : sm.w.wc1;
      *reinterpret_cast<f32x4*>(dst + row * LD + c4) = wreg[j];
    }
  }
  const float wr_c = W1[lane * ldw + 2 * din];
  float wa_c[FE_MAX];
#pragma unroll
  for (int f = 0; f < FE_MAX; ++f) wa_c[f] = (f < Fe) ? W1[lane * ldw + 2 * din + 1 + f] : 0.0f;
  float b2_c[4], bc1_c[4], wc2_c[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    b2_c[nt] = b2[nt * 16 + r];
    bc1_c[nt] = COORD ? bc1[nt * 16 + r] : 0.0f;
    wc2_c[nt] = COORD ? wc2[nt * 16 + r] : 0.0f;
  }
  FwdRows r0 = {};
  if (has_edges) load_fwd_rows<FE_MAX>(r0, id0, B, ld_p_bytes, lane);
  __syncthreads();          // the only workgroup barrier of the edge half: weights are staged
  STAMP3(1);
  if (va < vb) {            // wave-uniform

  float* act = sm.act[wave];
  float acc_h = 0.0f, acc_x = 0.0f;
  int cnt = 0;         // edges accumulated for the open node
  int vnext = va;      // first node of the chunk that has not been written yet
  const int tile_off4 = (4 * q * H + 4 * r) * 4;     // byte offset of this lane's 16 bytes in row 4 q of a saved tile; row 4 q + t: + t * 256

  for (int cb = e0; cb < e1; cb += TE16) {
    const int ts = tile_start(cb, e1);
    const int lo = cb - ts;                       // leading rows that repeat already-processed edges
    const int hi = min(TE16, e1 - ts);            // == 16 unless the whole graph has fewer than 16 edges
    FwdRows r1;
    load_fwd_rows<FE_MAX>(r1, id1, B, ld_p_bytes, lane);                                  // gathers of tile j+1
    EdgeIds<FE_MAX> id2 = load_edge_ids<FE_MAX>(B, Fe, cb + 2 * TE16, e1, E, lane);       // ids of tile j+2

    // ---- S0: geometry, lane = edge ----
    if (lane < TE16) {
      const float d0 = r0.xs[0] - r0.xd[0], d1 = r0.xs[1] - r0.xd[1], d2 = r0.xs[2] - r0.xd[2];
      const float rad = radial3(d0, d1, d2);
      const float inv = 1.0f / (sqrtf(rad) + 1e-30f);
      if (save_geo) buf_store4(f32x4{d0, d1, d2, rad}, B.geo, lane * 16 + ts * 16);
      sm.e_rad[wave][lane] = rad;
      if constexpr (COORD) { sm.e_xdst[wave][0][lane] = r0.xd[0]; sm.e_xdst[wave][1][lane] = r0.xd[1]; sm.e_xdst[wave][2][lane] = r0.xd[2]; }
      sm.e_xd[wave][0][lane] = d0 * inv;
      sm.e_xd[wave][1][lane] = d1 * inv;
      sm.e_xd[wave][2][lane] = d2 * inv;
#pragma unroll
      for (int f = 0; f < FE_MAX; ++f) sm.e_a[wave][f][lane] = id0.a[f];
    }
    __builtin_amdgcn_wave_barrier();
    STAMP3N();

    // ---- SA: first edge-MLP layer, lane = channel.  Written stage by stage over the 16 rows so that the
    //      LDS broadcasts, the exp and the rcp of different rows overlap (a row-by-row chain exposes every latency
    //      when only one or two waves share the SIMD) ----
    {
      float z[TE16];
#pragma unroll
      for (int i4 = 0; i4 < TE16; i4 += 4) {
        const f32x4 rv = *reinterpret_cast<const f32x4*>(&sm.e_rad[wave][i4]);     // broadcast read
#pragma unroll
        for (int k = 0; k < 4; ++k) z[i4 + k] = (r0.gs[i4 + k] + r0.gd[i4 + k]) + rv[k] * wr_c;
      }
#pragma unroll
      for (int f = 0; f < FE_MAX; ++f)
#pragma unroll
        for (int i4 = 0; i4 < TE16; i4 += 4) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(&sm.e_a[wave][f][i4]);
#pragma unroll
          for (int k = 0; k < 4; ++k) z[i4 + k] += av[k] * wa_c[f];
        }
      float ex[TE16];
#ifdef IS_ABL_SILU
#pragma unroll
      for (int i = 0; i < TE16; ++i) ex[i] = 0.5f;
#else
#pragma unroll
      for (int i = 0; i < TE16; ++i) ex[i] = sigmoid_f(z[i]);
#endif
      if (save1) {
        // rows in slot order, lane = channel: one coalesced 256-byte row per store.  dy1s != NULL: m1 = SiLU(z1) and
        // SiLU'(z1) = sigma + m1 (1 - sigma) (two arrays, nothing left to evaluate in the backward); dy1s == NULL: z1 itself
        // (one array: the backward evaluates the SiLU pair but no longer gathers / recomputes z1)
        // (column order of the three saved pre-activation arrays: channel c = 16 nt + r sits at position 4 r + nt of its row, so
        //  that the MFMA accumulator layout -- lane (r, q) holds channels r, 16 + r, 32 + r, 48 + r of rows 4 q + t -- stores and
        //  loads 16 contiguous bytes per lane and row: 4 full 256-byte rows per instruction instead of 16 half cache lines.  The
        //  arrays are private to is_egnn_layer_fwd / _bwd.)
        const int row_base = ts * (H * 4);
        const int pl = (4 * (lane & 15) + (lane >> 4)) * 4;      // byte position of channel `lane` in a saved row
        if (dy1s != nullptr) {
#pragma unroll
          for (int i = 0; i < TE16; ++i) {
            const float y = z[i] * ex[i];
            buf_store(y, B.m1, pl, row_base + i * (H * 4));
            buf_store(__builtin_fmaf(y, 1.0f - ex[i], ex[i]), B.d1, pl, row_base + i * (H * 4));
          }
        } else {
#pragma unroll
          for (int i = 0; i < TE16; ++i) buf_store(z[i], B.m1, pl, row_base + i * (H * 4));
        }
      }
#pragma unroll
      for (int i = 0; i < TE16; ++i) act[i * LD + lane] = z[i] * ex[i];
    }
    __builtin_amdgcn_wave_barrier();
    STAMP3N();

    const int tile_base = ts * (H * 4);      // scalar byte offset of the tile inside z2s / z3s
    // (16-byte stores take no scalar offset -- common.h buf_store4 -- so the tile's base joins the lane offset: formed where it
    //  is used, from an opaque copy, so that the sum is not one more register live across both matrix stages: the Fe = 8
    //  instantiation sits at 256)
    // ---- MM1: z2 = m1 W2^T + b2 ; mh = SiLU(z2) ----
    {
      int tile_voff4 = tile_off4;
      asm volatile("" : "+v"(tile_voff4));
      tile_voff4 += tile_base;
      f32x4 acc[4];
      zero_acc4(acc);
#ifdef IS_ABL_NOMFMA
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = *reinterpret_cast<const f32x4*>(act + r * LD + 4 * q + 16 * nt);
#else
      mm16_rows<4, H>(acc, act, sm.w.w2, lane);
#endif
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4 zr;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) zr[nt] = acc[nt][t] + b2_c[nt];
#ifdef IS_ZP_FWD_DW
        if (SAVE && ABL_SAVE) { for (int nt = 0; nt < 4; ++nt) buf_store(zr[nt], B.z2, tile_off4 + nt * 4 + t * (H * 4), tile_base); }
#else
        if (SAVE && ABL_SAVE) buf_store4(zr, B.z2, tile_voff4 + t * (H * 4));
#endif
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) act[tile16_row(t, q) * LD + nt * 16 + r] = silu_f(zr[nt]);
      }
    }
    __builtin_amdgcn_wave_barrier();
    STAMP3N();

    // ---- MM2: z3 = mh Wc1^T + bc1 ; s = SiLU(z3) . wc2 ----
    if constexpr (COORD) {
      int tile_voff4 = tile_off4;
      asm volatile("" : "+v"(tile_voff4));
      tile_voff4 += tile_base;
      f32x4 acc[4];
      zero_acc4(acc);
#ifdef IS_ABL_NOMFMA
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = *reinterpret_cast<const f32x4*>(act + r * LD + 4 * q + 16 * nt);
#else
      mm16_rows<4, H>(acc, act, sm.w.wc1, lane);
#endif
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float part = 0.0f;
        f32x4 zr;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) zr[nt] = acc[nt][t] + bc1_c[nt];
#ifdef IS_ZP_FWD_DW
        if (save3) { for (int nt = 0; nt < 4; ++nt) buf_store(zr[nt], B.z3, tile_off4 + nt * 4 + t * (H * 4), tile_base); }
#else
        if (save3) buf_store4(zr, B.z3, tile_voff4 + t * (H * 4));
#endif
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) part += silu_f(zr[nt]) * wc2_c[nt];
        part = sum_over_r16(part);
        if (r == 0) sm.e_s[wave][tile16_row(t, q)] = part;
      }
    }
    __builtin_amdgcn_wave_barrier();
    STAMP3N();

    // ---- SEG: running segment sums by destination; flush points are wave-uniform ----
#ifdef IS_ABL_SEG
    if (cb + TE16 >= e1) { buf_store(act[lane], B.hn, lane * 4, va * ld_hn_bytes); vnext = vb; }      // (keeps the tile's results alive)
#else
    {
      const unsigned long long fm = __ballot(id0.flush != 0);
      float hv[TE16], cs[TE16], cd[TE16];
#pragma unroll
      for (int i = 0; i < TE16; ++i) {
        hv[i] = act[i * LD + lane];
        cs[i] = COORD ? sm.e_s[wave][i] : 0.0f;
        cd[i] = COORD ? sm.e_xd[wave][lane < 3 ? lane : 0][i] : 0.0f;
      }
#pragma unroll
      for (int i = 0; i < TE16; ++i) {
        if (i >= lo && i < hi) {
          acc_h += hv[i];
          if constexpr (COORD) acc_x = __builtin_fmaf(cs[i], cd[i], acc_x);
          cnt += 1;
          if ((fm >> i) & 1ull) {
            const int v = __builtin_amdgcn_readlane(id0.d, i);
            for (int u = vnext; u < v; ++u) {   // nodes without in-edges (none on residue graphs)
              h_neigh[(size_t)u * ld_hn + lane] = 0.0f;
              if (COORD && lane < 3) x_out[u * 3 + lane] = x[u * 3 + lane];
            }
            buf_store(acc_h, B.hn, lane * 4, v * ld_hn_bytes);
            if constexpr (COORD) {
              // x' = x_dst + mean of the coordinate messages.  The flush code is unrolled 16 x and runs ~5 x per tile in an
              // issue-bound loop: the mean as a product with the reciprocal of the (wave-uniform) count -- v_rcp + one Newton
              // step, within 1 ulp of the quotient -- instead of a ten-instruction division, the base from LDS instead of
              // three v_readlane + two selects, the store through a buffer view instead of 64-bit address arithmetic
              const float c = (float)cnt;
              float rc = rcp_f(c);
              rc = __builtin_fmaf(__builtin_fmaf(-c, rc, 1.0f), rc, rc);
              if (lane < 3) buf_store(sm.e_xdst[wave][lane][i] + acc_x * rc, B.xo, lane * 4, v * 12);
            }
            acc_h = 0.0f; acc_x = 0.0f; cnt = 0;
            vnext = v + 1;
          }
        }
      }
    }
#endif
    __builtin_amdgcn_wave_barrier();
    STAMP3N();
    r0 = r1;
    id0 = id1;
    id1 = id2;
  }
  for (int u = vnext; u < vb; ++u) {
    h_neigh[(size_t)u * ld_hn + lane] = 0.0f;
    if (COORD && lane < 3) x_out[u * 3 + lane] = x[u * 3 + lane];
  }
  }   // va < vb
  STAMP3(56);

  // ================= node half: the workgroup's own nodes =================
  using D = Node16Dims<DIN>;
  constexpr int MT = 4, ROWS = 16 * MT;
  static_assert(sizeof(float) * ROWS * (D::LD1 + LD) <= sizeof(Fwd3Weights) + sizeof(sm.act), "node tiles must fit the dead edge buffers");
  const int c0 = blockIdx.x * W3;
  const int n0 = __builtin_amdgcn_readfirstlane(chunk_ptr[2 * c0]);
  const int n1 = __builtin_amdgcn_readfirstlane(chunk_ptr[2 * min(c0 + W3, nchunks)]);
  const bool has_next = psd_next != nullptr;
  const int col = wave * 16 + r;                       // output column of the two node-MLP layers
  f32x4 nb1[D::KQ1 / 4], nb2[4], nb3[2][4];
  {
    const f32x4* fp = reinterpret_cast<const f32x4*>(fpack) + (size_t)wave * NODE_FWD_SLOTS * 64 + lane;
#pragma unroll
    for (int g = 0; g < D::KQ1 / 4; ++g) nb1[g] = fp[g * 64];
#pragma unroll
    for (int g = 0; g < 4; ++g) nb2[g] = fp[(D::KQ1 / 4 + g) * 64];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) nb3[nt][g] = has_next ? fp[(D::KQ1 / 4 + 4 + nt * 4 + g) * 64] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float bn1_c = bn1[col], bn2_c = bn2[col];
  float b1n_c[2] = {0.f, 0.f};
  if (has_next) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int cc = wave * 32 + nt * 16 + r;          // psd column: [0, 64) = Ps (bias b0n, may be null), [64, 128) = Pd (bias b1n)
      b1n_c[nt] = cc >= 64 ? b1n[cc - 64] : (b0n != nullptr ? b0n[cc] : 0.0f);
    }
  }
  constexpr int RPW = ROWS / W3;                       // rows staged per wave and pass
  // the first pass's rows of h do not depend on the edge half: requested before the barrier (the slowest wave of the workgroup
  // is still in its edge loop), they are there when the h_neigh rows -- which do have to wait -- are asked for
  float hv[RPW];
  if (n0 < n1) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) hv[i] = h[(size_t)min(n0 + wave * RPW + i, n1 - 1) * ld_h + min(lane, DIN - 1)];
  }
  __syncthreads();     // every wave's h_neigh rows are written (L2 / the CU's L1); weight tiles and act buffers are dead
  STAMP3(57);
  float* xs = reinterpret_cast<float*>(&sm);           // [ROWS][LD1]   X = [h | h_neigh | 0]
  float* a1s = xs + ROWS * D::LD1;                     // [ROWS][LD]    SiLU(zn1)
  float* hps = xs;                                     // [ROWS][LD]    h' (over X, which is dead by then)
#ifdef IS_ABL_NONODE
  for (int row0 = n0; row0 < n0; row0 += ROWS) {
#else
  for (int row0 = n0; row0 < n1; row0 += ROWS) {
#endif
    const int mt_used = min(MT, (n1 - row0 + 15) >> 4);     // 16-row tiles of this pass that hold nodes
    {
      float nv[RPW];                                   // all loads first, then the LDS stores
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const int row = min(row0 + wave * RPW + i, n1 - 1);
        if (row0 != n0) hv[i] = h[(size_t)row * ld_h + min(lane, DIN - 1)];
        nv[i] = h_neigh[(size_t)row * ld_hn + lane];
      }
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const int lr = wave * RPW + i;
        const bool valid = row0 + lr < n1;
        if (lane < DIN) xs[lr * D::LD1 + lane] = valid ? hv[i] : 0.0f;
        xs[lr * D::LD1 + DIN + lane] = valid ? nv[i] : 0.0f;
        if (lane < D::KP - D::KV) xs[lr * D::LD1 + D::KV + lane] = 0.0f;
      }
    }
    __syncthreads();
    STAMP3(58);
    {     // zn1 = X Wn1^T + bn1 ; a1 = SiLU(zn1)
      f32x4 acc[MT];
      zero_acc4(acc);
      mm16_regB_used<MT, D::KQ1, D::LD1>(acc, xs, nb1, lane, mt_used);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int lr = mt * 16 + tile16_row(t, q);
          const float z = acc[mt][t] + bn1_c;
          if (SAVE && row0 + lr < n1) zn1[(size_t)(row0 + lr) * H + col] = z;
          a1s[lr * LD + col] = silu_f(z);
        }
    }
    __syncthreads();
    STAMP3(59);
    {     // h' = a1 Wn2^T + bn2
      f32x4 acc[MT];
      zero_acc4(acc);
      mm16_regB_used<MT, 16, LD>(acc, a1s, nb2, lane, mt_used);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int lr = mt * 16 + tile16_row(t, q);
          const float v = acc[mt][t] + bn2_c;
          if (row0 + lr < n1) h_out[(size_t)(row0 + lr) * H + col] = v;
          hps[lr * LD + col] = v;
        }
    }
    __syncthreads();
    STAMP3(60);
    if (has_next) {     // next pre-projection: wave w produces psd columns [32w, 32w + 32)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        f32x4 acc[MT];
        zero_acc4(acc);
        mm16_regB_used<MT, 16, LD>(acc, hps, nb3[nt], lane, mt_used);
        const int cc = wave * 32 + nt * 16 + r;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int lr = mt * 16 + tile16_row(t, q);
            if (row0 + lr < n1) psd_next[(size_t)(row0 + lr) * 128 + cc] = acc[mt][t] + b1n_c[nt];
          }
      }
    }
    if (row0 + ROWS < n1) __syncthreads();     // the next pass restages X over h'
  }
  STAMP3(61);
  wg_clock_end(wg_clock);
}

}  // namespace is

#ifdef IS_STAGE_STAMPS
extern "C" int is_debug_stamps3(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(is::g_stamps3), sizeof(long long) * 64) == hipSuccess ? 0 : is::fail(__func__, -5);
}
#endif

// One EGNNConv layer forward (edge pass + node MLP + next pre-projection).  h [N, ld_h] (din = 20 | 64 columns): the
// layer's input node features; ps / pd [N, ld_p]: its pre-projections; chunk_ptr: nchunks + 1 rows (node, first edge);
// fpack: the layer's forward operand pack (is_stack_prologue); x_out == NULL: the coordinate branch is not evaluated
// (z3s unused); z2s == NULL: nothing is saved for a backward pass (z3s, zn1 unused); z3s == NULL with z2s: the coordinate
// MLP's pre-activation is not saved (is_egnn_layer_bwd recomputes it from z2); psd_next == NULL: no next
// projection (b0n / b1n unused); b0n may be NULL.  wg_clock: NULL, or [nchunks / 4][2] int64 -- every workgroup's start / end
// device wall clock (common.h wg_clock_start / _end; bench.py's in-situ launch timing).  m1s / dy1s [max(E,16), 64]: NULL, or out: the
// first edge-MLP activation SiLU(z1) and SiLU'(z1) per edge slot -- or, with dy1s == NULL, the pre-activation z1 in m1s -- for a
// backward built to read them (is_layer_saves_m1(): 1 = both arrays, 2 = z1 only).  geos [max(E,16), 4]: NULL, or out: (x_src - x_dst,
// |x_src - x_dst|^2) per edge slot, for a backward built to read it (is_layer_saves_geo()).
extern "C" int is_egnn_layer_fwd(const float* ps, const float* pd, int ld_p, const float* x, const float* ea,
                                 const int32_t* rowptr, const int32_t* srcs, const int32_t* dsts,
                                 const int32_t* chunk_ptr, int nchunks, const float* W1, int ldw, int din,
                                 const float* W2, const float* b2, const float* Wc1, const float* bc1,
                                 const float* wc2, float* h_neigh, int ld_hn, float* x_out, float* z2s,
                                 float* z3s, int N, int E, int Fe, const float* h, int ld_h, const float* bn1,
                                 const float* bn2, const float* b0n, const float* b1n, const float* fpack,
                                 float* zn1, float* h_out, float* psd_next, long long* wg_clock, float* m1s, float* dy1s,
                                 float* geos, void* stream) {
  if (N <= 0) return 0;
  const bool coord = x_out != nullptr;
  const bool save = z2s != nullptr;
  if (nchunks <= 0 || (nchunks % is::W3) != 0 || Fe < 0 || Fe > 8 || (din != 20 && din != 64) || fpack == nullptr || h == nullptr ||
      h_out == nullptr || (save && zn1 == nullptr) || (psd_next != nullptr && b1n == nullptr) || (m1s == nullptr && dy1s != nullptr))
    return is::fail(__func__, -22);
  // 32-bit byte offsets inside every buffer (raw buffer addressing)
  const long long lim = 0x7fffffffLL;
  if ((long long)N * ld_p * 4 > lim || (long long)N * ld_hn * 4 > lim || (long long)(E + 16) * 64 * 4 > lim) return is::fail(__func__, -22);
  if (Fe == 0) ea = x;   // never used as a feature, but the clamped prefetch address must be valid
  const dim3 grid(nchunks / is::W3), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
#define IS_LAUNCH_LF(FE, SV, CO, DI)                                                                                           \
  hipLaunchKernelGGL((is::egnn_layer_fwd_kernel<FE, SV, CO, DI>), grid, block, 0, st, ps, pd, ld_p, x, ea, rowptr, srcs, dsts, \
                     chunk_ptr, nchunks, W1, ldw, din, W2, b2, Wc1, bc1, wc2, h_neigh, ld_hn, x_out, z2s, z3s, E, Fe, h, ld_h, \
                     bn1, bn2, b0n, b1n, fpack, zn1, h_out, psd_next, wg_clock, m1s, dy1s, geos)
#define IS_LAUNCH_LF_D(FE, SV, CO) do { if (din == 20) IS_LAUNCH_LF(FE, SV, CO, 20); else IS_LAUNCH_LF(FE, SV, CO, 64); } while (0)
#define IS_LAUNCH_LF_C(FE, SV) do { if (coord) IS_LAUNCH_LF_D(FE, SV, true); else IS_LAUNCH_LF_D(FE, SV, false); } while (0)
  if (Fe <= 1) { if (save) IS_LAUNCH_LF_C(1, true); else IS_LAUNCH_LF_C(1, false); }
  else { if (save) IS_LAUNCH_LF_C(8, true); else IS_LAUNCH_LF_C(8, false); }
#undef IS_LAUNCH_LF_C
#undef IS_LAUNCH_LF_D
#undef IS_LAUNCH_LF
  return is::launch_status(__func__);
}
