// Fused EGNN edge pass, forward -- wave-autonomous, software-pipelined version (v3).
//
// Same algorithm, inputs, outputs and summation order as egnn_edge_fwd16.hip (results are
// bit-identical); what changes is how the work is cut and scheduled:
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
#include "common.h"
#include <stdlib.h>

namespace is {

constexpr int W3 = 4;  // waves per workgroup (they only share the LDS weight tiles)

#ifdef IS_STAGE_STAMPS
__device__ long long g_stamps3[64];
#define STAMP3(k) do { if (blockIdx.x == 300 && threadIdx.x == 0) g_stamps3[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP3(k) do { } while (0)
#endif

// ---- optional split-bf16 matrix path (X3) -----------------------------------------------------------------------
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi): x*y ~ hi*hi' + hi*lo' + lo*hi' (three v_mfma_f32_16x16x32_bf16 with
// fp32 accumulation; the dropped lo*lo' term is 2^-16 relative).  One 16x16x32 bf16 MFMA replaces eight 16x16x4 fp32
// MFMAs at 1/16 of their cycles, so a 64x64 layer costs 24 x 16 instead of 64 x 32 SIMD cycles.  Opt-in
// (IMMUNOSTRUCT_EDGE_FWD=v3x): NOT bit-identical to the fp32 kernels, measured error in tests/test_gpu_kernels.py.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int LDB = H + 8;      // bf16 row stride of the split weight tiles (144 bytes: 16-byte aligned rows)

__device__ __forceinline__ void split8_3(const f32x4 a, const f32x4 b, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float v = i < 4 ? a[i] : b[i - 4];
    hi[i] = (__bf16)v;
    const float r1 = v - (float)hi[i];
    mid[i] = (__bf16)r1;
    lo[i] = (__bf16)(r1 - (float)mid[i]);
  }
}

// three-piece variant: x = hi + mid + lo exactly (3 x 8 mantissa bits); the six largest cross terms are kept
// (hh, hm, mh, hl, lh, mm): what is dropped is O(2^-24) relative, i.e. fp32-class accuracy at 6 x 16 cycles per 32 k.
__device__ __forceinline__ void mm16_rows_x6(f32x4 (&acc)[4], const float* a_lds, const __bf16* w_hi, const __bf16* w_mid,
                                             const __bf16* w_lo, int lane) {
  const int r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float* ap = a_lds + r * LD + c * 32 + q * 8;
    bf16x8 ah, am, al;
    split8_3(*reinterpret_cast<const f32x4*>(ap), *reinterpret_cast<const f32x4*>(ap + 4), ah, am, al);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int off = (nt * 16 + r) * LDB + c * 32 + q * 8;
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(w_hi + off);
      const bf16x8 bm = *reinterpret_cast<const bf16x8*>(w_mid + off);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(w_lo + off);
      // smallest terms first
      acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc[nt], 0, 0, 0);
      acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[nt], 0, 0, 0);
      acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[nt], 0, 0, 0);
      acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc[nt], 0, 0, 0);
      acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc[nt], 0, 0, 0);
      acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[nt], 0, 0, 0);
    }
  }
}

__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hi[i] = (__bf16)a[i]; lo[i] = (__bf16)(a[i] - (float)hi[i]);
    hi[4 + i] = (__bf16)b[i]; lo[4 + i] = (__bf16)(b[i] - (float)hi[4 + i]);
  }
}

// acc[nt] (16 x 16) += A[16 x 64] * W[nt*16 .., 64]^T with A fp32 in LDS (row stride LD), W split bf16 in LDS
__device__ __forceinline__ void mm16_rows_x3(f32x4 (&acc)[4], const float* a_lds, const __bf16* w_hi, const __bf16* w_lo, int lane) {
  const int r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float* ap = a_lds + r * LD + c * 32 + q * 8;
    bf16x8 ah, al;
    split8(*reinterpret_cast<const f32x4*>(ap), *reinterpret_cast<const f32x4*>(ap + 4), ah, al);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int off = (nt * 16 + r) * LDB + c * 32 + q * 8;
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(w_hi + off);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(w_lo + off);
      acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[nt], 0, 0, 0);
      acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[nt], 0, 0, 0);
      acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[nt], 0, 0, 0);
    }
  }
}

template <int X3> struct Fwd3Weights;
template <> struct Fwd3Weights<0> { float w2[H * LD]; float wc1[H * LD]; };
template <> struct Fwd3Weights<2> { __bf16 w2h[H * LDB], w2l[H * LDB], wc1h[H * LDB], wc1l[H * LDB]; };
template <> struct Fwd3Weights<3> { __bf16 w2h[H * LDB], w2m[H * LDB], w2l[H * LDB], wc1h[H * LDB], wc1m[H * LDB], wc1l[H * LDB]; };

template <int FE_MAX, int X3 = 0>
struct Fwd3Smem {
  Fwd3Weights<X3> w;
  float act[W3][TE16 * LD];
  float e_rad[W3][TE16];
  float e_xd[W3][3][TE16];
  float e_s[W3][TE16];
  float e_a[W3][FE_MAX][TE16];
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
  rsrc_t ps, pd, x, srcs, dsts, ea, hn, z2, z3;
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
template <int FE_MAX, bool SAVE, int X3 = 0, bool COORD = true>
__global__ __launch_bounds__(256) void egnn_edge_fwd3_kernel(
    const float* __restrict__ ps, const float* __restrict__ pd, int ld_p,
    const float* __restrict__ x, const float* __restrict__ ea,
    const int* __restrict__ rowptr, const int* __restrict__ srcs, const int* __restrict__ dsts,
    const int* __restrict__ chunk_ptr, int nchunks,
    const float* __restrict__ W1, int ldw, int din,
    const float* __restrict__ W2, const float* __restrict__ b2,
    const float* __restrict__ Wc1, const float* __restrict__ bc1, const float* __restrict__ wc2,
    float* __restrict__ h_neigh, int ld_hn, float* __restrict__ x_out,
    float* __restrict__ z2s, float* __restrict__ z3s, int E, int Fe) {
  __shared__ Fwd3Smem<FE_MAX, X3> sm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  STAMP3(0);
  int stamp_k = 2;

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
  B.ea = make_rsrc(ea); B.hn = make_rsrc(h_neigh); B.z2 = make_rsrc(z2s); B.z3 = make_rsrc(z3s);
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
      if constexpr (X3 == 2) {
        __bf16* dh = (j < 4) ? sm.w.w2h : sm.w.wc1h;
        __bf16* dl = (j < 4) ? sm.w.w2l : sm.w.wc1l;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const __bf16 hv = (__bf16)wreg[j][k];
          dh[row * LDB + c4 + k] = hv;
          dl[row * LDB + c4 + k] = (__bf16)(wreg[j][k] - (float)hv);
        }
      } else if constexpr (X3 == 3) {
        __bf16* dh = (j < 4) ? sm.w.w2h : sm.w.wc1h;
        __bf16* dm = (j < 4) ? sm.w.w2m : sm.w.wc1m;
        __bf16* dl = (j < 4) ? sm.w.w2l : sm.w.wc1l;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const __bf16 hv = (__bf16)wreg[j][k];
          const float r1 = wreg[j][k] - (float)hv;
          const __bf16 mv = (__bf16)r1;
          dh[row * LDB + c4 + k] = hv;
          dm[row * LDB + c4 + k] = mv;
          dl[row * LDB + c4 + k] = (__bf16)(r1 - (float)mv);
        }
      } else {
        float* dst = (j < 4) ? sm.w.w2 : sm.w.wc1;
        *reinterpret_cast<f32x4*>(dst + row * LD + c4) = wreg[j];
      }
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
  __syncthreads();          // the only workgroup barrier: weights are staged
  STAMP3(1);
  if (va >= vb) return;     // wave-uniform

  float* act = sm.act[wave];
  float acc_h = 0.0f, acc_x = 0.0f;
  int cnt = 0;         // edges accumulated for the open node
  int vnext = va;      // first node of the chunk that has not been written yet
  const int tile_off = (4 * q * H + r) * 4;     // byte offset of D-layout element (t = 0, nt = 0); (t, nt): + (t*H + nt*16)*4

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
      sm.e_rad[wave][lane] = rad;
      sm.e_xd[wave][0][lane] = d0 * inv;
      sm.e_xd[wave][1][lane] = d1 * inv;
      sm.e_xd[wave][2][lane] = d2 * inv;
#pragma unroll
      for (int f = 0; f < FE_MAX; ++f) sm.e_a[wave][f][lane] = id0.a[f];
    }
    __builtin_amdgcn_wave_barrier();
    STAMP3(stamp_k); ++stamp_k;

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
#pragma unroll
      for (int i = 0; i < TE16; ++i) ex[i] = __expf(-z[i]);
#pragma unroll
      for (int i = 0; i < TE16; ++i) ex[i] = rcp_f(1.0f + ex[i]);
#pragma unroll
      for (int i = 0; i < TE16; ++i) act[i * LD + lane] = z[i] * ex[i];
    }
    __builtin_amdgcn_wave_barrier();
    STAMP3(stamp_k); ++stamp_k;

    const int tile_base = ts * (H * 4);      // scalar byte offset of the tile inside z2s / z3s
    // ---- MM1: z2 = m1 W2^T + b2 ; mh = SiLU(z2) ----
    {
      f32x4 acc[4];
      zero_acc4(acc);
      if constexpr (X3 == 2) mm16_rows_x3(acc, act, sm.w.w2h, sm.w.w2l, lane);
      else if constexpr (X3 == 3) mm16_rows_x6(acc, act, sm.w.w2h, sm.w.w2m, sm.w.w2l, lane);
      else mm16_rows<4, H>(acc, act, sm.w.w2, lane);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float z2 = acc[nt][t] + b2_c[nt];
          if (SAVE) buf_store(z2, B.z2, tile_off + (t * H + nt * 16) * 4, tile_base);
          act[tile16_row(t, q) * LD + nt * 16 + r] = silu_f(z2);
        }
    }
    __builtin_amdgcn_wave_barrier();
    STAMP3(stamp_k); ++stamp_k;

    // ---- MM2: z3 = mh Wc1^T + bc1 ; s = SiLU(z3) . wc2 ----
    if constexpr (COORD) {
      f32x4 acc[4];
      zero_acc4(acc);
      if constexpr (X3 == 2) mm16_rows_x3(acc, act, sm.w.wc1h, sm.w.wc1l, lane);
      else if constexpr (X3 == 3) mm16_rows_x6(acc, act, sm.w.wc1h, sm.w.wc1m, sm.w.wc1l, lane);
      else mm16_rows<4, H>(acc, act, sm.w.wc1, lane);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float part = 0.0f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const float z3 = acc[nt][t] + bc1_c[nt];
          if (SAVE) buf_store(z3, B.z3, tile_off + (t * H + nt * 16) * 4, tile_base);
          part += silu_f(z3) * wc2_c[nt];
        }
        part = sum_over_r16(part);
        if (r == 0) sm.e_s[wave][tile16_row(t, q)] = part;
      }
    }
    __builtin_amdgcn_wave_barrier();
    STAMP3(stamp_k); ++stamp_k;

    // ---- SEG: running segment sums by destination; flush points are wave-uniform ----
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
            const float x0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r0.xd[0]), i));
            const float x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r0.xd[1]), i));
            const float x2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r0.xd[2]), i));
            if (COORD && lane < 3) x_out[v * 3 + lane] = (lane == 0 ? x0 : (lane == 1 ? x1 : x2)) + acc_x / (float)cnt;
            acc_h = 0.0f; acc_x = 0.0f; cnt = 0;
            vnext = v + 1;
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    STAMP3(stamp_k); ++stamp_k;
    r0 = r1;
    id0 = id1;
    id1 = id2;
  }
  for (int u = vnext; u < vb; ++u) {
    h_neigh[(size_t)u * ld_hn + lane] = 0.0f;
    if (COORD && lane < 3) x_out[u * 3 + lane] = x[u * 3 + lane];
  }
}

}  // namespace is

#ifdef IS_STAGE_STAMPS
extern "C" int is_debug_stamps3(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(is::g_stamps3), sizeof(long long) * 64) == hipSuccess ? 0 : -5;
}
#endif

static int edge_fwd_v3_launch(int x3, const float* ps, const float* pd, int ld_p, const float* x, const float* ea,
                              const int32_t* rowptr, const int32_t* srcs, const int32_t* dsts,
                              const int32_t* chunk_ptr, int nchunks, const float* W1, int ldw, int din,
                              const float* W2, const float* b2, const float* Wc1, const float* bc1,
                              const float* wc2, float* h_neigh, int ld_hn, float* x_out, float* z2s,
                              float* z3s, int N, int E, int Fe, void* stream) {
  if (N <= 0 || nchunks <= 0) return 0;
  // x_out == nullptr: the coordinate branch is not evaluated (fp32 kernels only); z3s is then not written and may be null
  const bool coord = x_out != nullptr;
  if (Fe < 0 || Fe > 8 || (coord && (z2s == nullptr) != (z3s == nullptr)) || (!coord && x3 != 0)) return -22;
  // 32-bit byte offsets inside every buffer (raw buffer addressing)
  const long long lim = 0x7fffffffLL;
  if ((long long)N * ld_p * 4 > lim || (long long)N * ld_hn * 4 > lim || (long long)(E + 16) * 64 * 4 > lim) return -22;
  if (Fe == 0) ea = x;   // never used as a feature, but the clamped prefetch address must be valid
  const dim3 grid((nchunks + is::W3 - 1) / is::W3), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool save = z2s != nullptr;
#define IS_LAUNCH_FWD3(FE, SV, XX, CO)                                                                                       \
  hipLaunchKernelGGL((is::egnn_edge_fwd3_kernel<FE, SV, XX, CO>), grid, block, 0, st, ps, pd, ld_p, x, ea, rowptr, srcs, dsts, \
                     chunk_ptr, nchunks, W1, ldw, din, W2, b2, Wc1, bc1, wc2, h_neigh, ld_hn, x_out, z2s, z3s, E, Fe)
  if (x3 == 2) {
    if (Fe > 1) return -22;
    if (save) IS_LAUNCH_FWD3(1, true, 2, true); else IS_LAUNCH_FWD3(1, false, 2, true);
  } else if (x3 == 3) {
    if (Fe > 1) return -22;
    if (save) IS_LAUNCH_FWD3(1, true, 3, true); else IS_LAUNCH_FWD3(1, false, 3, true);
  } else if (!coord) {
    if (Fe <= 1) { if (save) IS_LAUNCH_FWD3(1, true, 0, false); else IS_LAUNCH_FWD3(1, false, 0, false); }
    else { if (save) IS_LAUNCH_FWD3(8, true, 0, false); else IS_LAUNCH_FWD3(8, false, 0, false); }
  } else if (Fe <= 1) {
    if (save) IS_LAUNCH_FWD3(1, true, 0, true); else IS_LAUNCH_FWD3(1, false, 0, true);
  } else {
    if (save) IS_LAUNCH_FWD3(8, true, 0, true); else IS_LAUNCH_FWD3(8, false, 0, true);
  }
#undef IS_LAUNCH_FWD3
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int is_egnn_edge_fwd_v3(const float* ps, const float* pd, int ld_p, const float* x, const float* ea,
                                   const int32_t* rowptr, const int32_t* srcs, const int32_t* dsts,
                                   const int32_t* chunk_ptr, int nchunks, const float* W1, int ldw, int din,
                                   const float* W2, const float* b2, const float* Wc1, const float* bc1,
                                   const float* wc2, float* h_neigh, int ld_hn, float* x_out, float* z2s,
                                   float* z3s, int N, int E, int Fe, void* stream) {
  return edge_fwd_v3_launch(0, ps, pd, ld_p, x, ea, rowptr, srcs, dsts, chunk_ptr, nchunks, W1, ldw, din, W2, b2, Wc1, bc1,
                            wc2, h_neigh, ld_hn, x_out, z2s, z3s, N, E, Fe, stream);
}

// Same pass with the two 64 x 64 layers on split-bf16 MFMA (three v_mfma_f32_16x16x32_bf16 per product term, fp32
// accumulation): opt-in, Fe <= 1, not bit-identical to the fp32 kernels (relative error of a product ~2^-16).
extern "C" int is_egnn_edge_fwd_v3x(const float* ps, const float* pd, int ld_p, const float* x, const float* ea,
                                    const int32_t* rowptr, const int32_t* srcs, const int32_t* dsts,
                                    const int32_t* chunk_ptr, int nchunks, const float* W1, int ldw, int din,
                                    const float* W2, const float* b2, const float* Wc1, const float* bc1,
                                    const float* wc2, float* h_neigh, int ld_hn, float* x_out, float* z2s,
                                    float* z3s, int N, int E, int Fe, void* stream) {
  static const int pieces = (getenv("IMMUNOSTRUCT_SPLIT_PIECES") != nullptr && atoi(getenv("IMMUNOSTRUCT_SPLIT_PIECES")) == 3) ? 3 : 2;
  return edge_fwd_v3_launch(pieces, ps, pd, ld_p, x, ea, rowptr, srcs, dsts, chunk_ptr, nchunks, W1, ldw, din, W2, b2, Wc1, bc1,
                            wc2, h_neigh, ld_hn, x_out, z2s, z3s, N, E, Fe, stream);
}
