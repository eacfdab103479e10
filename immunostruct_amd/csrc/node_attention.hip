// Node self-attention reduced to what the mean-pooled readout needs, forward + backward.
//
// Reference: models/layers.py:13-22 / :67-78 (softmax(q k^T / sqrt(d)) v, no mask -- padded nodes take
// part) followed by torch_geometric global_mean_pool over the n (padded) nodes of each graph
// (models/hybrid_models.py:326-331).  With abar_h[j] = (1/n) sum_i A_h[i][j] (column mean of the
// attention matrix of head h):
//     mean_i (A_h V_h)_i = sum_j abar_h[j] v_j,h = W_v,h (sum_j abar_h[j] x_j) + b_v,h
// so the kernel only has to produce   ctx_h = sum_j abar_h[j] x_j   (one 64-vector per graph and head);
// the value projection and w_concat then act on B x 64 vectors (host side, tiny).  Q and K come from the
// fused projection kernel as one [N,128] tensor (csrc/egnn_node.hip, is_node_proj_fwd).
//
// One workgroup per graph, NT = ceil(n/32) waves; wave w owns query rows [32w, 32w+32).
//   forward : S tile (32 x n_pad) on v_mfma_f32_32x32x2_f32 from LDS-resident Q,K; row softmax in
//             registers; column sums -> abar; ctx = abar^T X.   Saves abar and the row statistics.
//             Also saves the normalised probabilities P tile by tile in the MFMA accumulator layout
//             (probs[b][head][query block][key block][register t][lane]: every store / load is one coalesced
//             256-byte row) -- 148 KB per graph and head for n = 190.
//   backward: dabar_j = g_ctx . x_j ; dS_ij = P_ij (dabar_j - t_i)/n , t_i = sum_k P_ik dabar_k ;
//             dQ = scale dS K , dK = scale dS^T Q.  P is READ BACK instead of recomputed (the first version
//             recomputed the score tiles three times: 60 k MFMA cycles per wave, now 25 k):
//               * wave w loads the six tiles of its query block once (keys on the lanes), gets t_i by a
//                 lane reduction, then transposes each tile through a wave-private LDS tile so that the
//                 queries sit on the lanes: the dS registers ARE the MFMA A operand of dQ = dS K;
//               * then loads the six tiles of its KEY block (same layout): the dS registers are directly the
//                 A operand of dK = dS^T Q.
//             dx_j += sum_h abar_h[j] g_ctx_h (direct term).
#include "common.h"
#include <cstdlib>

namespace is {

#ifdef IS_STAGE_STAMPS
__device__ long long g_stamps_attn[16];
#define STAMPA(k) do { if (blockIdx.x == 50 && threadIdx.x == 0) g_stamps_attn[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMPA(k) do { } while (0)
#endif

template <int NT, int D>
struct AttnSmem {
  static constexpr int LDQ = D + 4;
  float qs[NT * 32 * LDQ];
  float ks[NT * 32 * LDQ];
  float wpart[NT][NT * 32];   // per-wave column partial sums (forward) / scratch
  float abar[NT * 32];
  float dab[NT * 32];         // backward: d abar_j
  float tvec[NT * 32];        // backward: t_i
  float rmax[NT * 32], rinv[NT * 32];
  float cpart[NT][64];
};

// DUAL: K and Q rows resident together (the two matrix passes run side by side, see attn_colmean_bwd_kernel)
template <int NT, int D>
constexpr bool attn_bwd_dual() {      // (at most 12 waves = 170 registers per wave)
#ifdef IS_ATTN_DUAL
  return NT <= 6 && sizeof(float) * (2 * NT * 32 * (D + 4) + NT * 32 * 33 + 3 * NT * 32) <= 150 * 1024;
#else
  return false;      // measured slower (HISTORY.md round 4): kept as a build switch
#endif
}

template <int NT, int D>
struct AttnBwdSmem {
  static constexpr int LDQ = D + 4;
  static constexpr bool DUAL = attn_bwd_dual<NT, D>();
  float kq[NT * 32 * LDQ];    // K rows (pass A); without DUAL: then the Q rows (pass B)
  float qs[DUAL ? NT * 32 * LDQ : 1];      // DUAL: Q rows
  float abar[NT * 32];
  float dab[NT * 32];         // d abar_j
  float tvec[NT * 32];        // t_i
  float tr[NT][32 * 33];      // wave-private transposition tile
};

// stage one half of qk (which = 0: Q, 1: K) of head hd into an LDQ-strided LDS tile
template <int NT, int D>
__device__ __forceinline__ void attn_stage_half(float* dst, const float* __restrict__ qk, int which, int b, int n, int hd,
                                                int tid, int nthreads) {
  constexpr int LDQ = D + 4;
  for (int idx0 = tid; idx0 < NT * 32 * D; idx0 += 8 * nthreads) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = idx0 + u * nthreads, lr = idx / D, c = idx % D;
      const bool ok = idx < NT * 32 * D && lr < n;
      v[u] = ok ? qk[(size_t)(b * n + (ok ? lr : 0)) * 128 + which * 64 + hd * D + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = idx0 + u * nthreads, lr = idx / D, c = idx % D;
      if (idx < NT * 32 * D) dst[lr * LDQ + c] = v[u];
    }
  }
}

template <int NT, int D>
__device__ __forceinline__ void attn_stage_qk(AttnSmem<NT, D>& sm, const float* __restrict__ qk, int b, int n, int hd,
                                              int tid, int nthreads) {
  constexpr int LDQ = AttnSmem<NT, D>::LDQ;
  // every thread issues 8 row-segment loads of Q and K before the first LDS store (one memory round trip per batch)
  for (int idx0 = tid; idx0 < NT * 32 * D; idx0 += 8 * nthreads) {
    float q[8], k[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = idx0 + u * nthreads, lr = idx / D, c = idx % D;
      const bool ok = idx < NT * 32 * D && lr < n;
      const size_t base = (size_t)(b * n + (ok ? lr : 0)) * 128 + hd * D + c;
      q[u] = ok ? qk[base] : 0.f;
      k[u] = ok ? qk[base + 64] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = idx0 + u * nthreads, lr = idx / D, c = idx % D;
      if (idx < NT * 32 * D) { sm.qs[lr * LDQ + c] = q[u]; sm.ks[lr * LDQ + c] = k[u]; }
    }
  }
}

template <int NT, int D>
__global__ __launch_bounds__(64 * NT) void attn_colmean_fwd_kernel(
    const float* __restrict__ qk, const float* __restrict__ x, float* __restrict__ ctx, float* __restrict__ abar_out,
    float* __restrict__ probs, int n, int heads, const float* __restrict__ wv, const float* __restrict__ bv,
    const float* __restrict__ wc, const float* __restrict__ bc, float* __restrict__ a1_out, float* __restrict__ y_out) {
  constexpr int LDQ = AttnSmem<NT, D>::LDQ;
  __shared__ AttnSmem<NT, D> sm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  const int r = lane & 31, hf = lane >> 5;
  const float scale = rsqrtf((float)D);
  for (int hd = 0; hd < heads; ++hd) {
    __syncthreads();
    attn_stage_qk<NT, D>(sm, qk, b, n, hd, tid, 64 * NT);
    __syncthreads();
    {
      float* ptile = probs != nullptr ? probs + ((size_t)(b * heads + hd) * NT + wave) * NT * 1024 : nullptr;
      f32x16 acc[NT];
      zero_acc(acc);
      mm_rows<NT, D, LDQ, LDQ>(acc, sm.qs + wave * 32 * LDQ, sm.ks, lane);
      float colsum[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) colsum[nt] = 0.f;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int i = wave * 32 + tile_row(t, hf);
        float m = -INFINITY;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float s = (nt * 32 + r < n) ? acc[nt][t] * scale : -INFINITY;
          acc[nt][t] = s;
          m = fmaxf(m, s);
        }
        m = max_over_r(m);
        float l = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float e = __expf(acc[nt][t] - m);   // exp(-inf) = 0 for masked columns
          acc[nt][t] = e;
          l += e;
        }
        l = sum_over_r(l);
        const float inv = 1.0f / l;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float p = (i < n) ? acc[nt][t] * inv : 0.0f;      // rows of padded queries are stored as zeros
          colsum[nt] += p;
          if (ptile != nullptr) ptile[(nt * 16 + t) * 64 + lane] = p;
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float v = colsum[nt] + __shfl_xor(colsum[nt], 32, 64);
        if (hf == 0) sm.wpart[wave][nt * 32 + r] = v;
      }
    }
    __syncthreads();
    for (int j = tid; j < NT * 32; j += 64 * NT) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NT; ++w) v += sm.wpart[w][j];
      v /= (float)n;
      sm.abar[j] = v;
      if (j < n && abar_out != nullptr) abar_out[(size_t)(b * heads + hd) * n + j] = v;
    }
    __syncthreads();
    {
      float a = 0.f;
      for (int j0 = wave; j0 < n; j0 += 8 * NT) {     // same summation order as a plain loop, 8 row loads in flight
        float xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + u * NT;
          xv[u] = (j < n) ? x[(size_t)(b * n + j) * 64 + lane] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + u * NT;
          if (j < n) a += sm.abar[j] * xv[u];
        }
      }
      sm.cpart[wave][lane] = a;
    }
    __syncthreads();
    if (tid < 64) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NT; ++w) v += sm.cpart[w][tid];
      ctx[(size_t)(b * heads + hd) * 64 + tid] = v;
      sm.cpart[0][tid] = v;      // (own column: read above by this thread only)
    }
  }
  if constexpr (D == 64) {
    if (wv != nullptr) {
      // pooled tail of a single head (models/layers.py:74-77 on the mean-pooled vector): hid = W_v ctx + b_v,
      // y = W_c hid + b_c, the arithmetic of csrc/mlp_head.hip (k ascending from the bias).  Both 64 x 64 matrices go to LDS
      // row-major with a pitch of 65 (lane d then reads row d conflict-free) where the dead Q / K tiles were: ALL their loads
      // are issued first -- coalesced 16-byte loads, one round trip for the whole workgroup -- and stored after the barrier.
      // (One element per thread and trip -- 11 dependent load / store trips -- made this tail 18 of the launch's 37 us.)
      constexpr int LDW = 65, NTH = 64 * NT, PER = (1024 + NTH - 1) / NTH;
      float* wvl = sm.qs;                       // [d][k]
      float* wcl = sm.qs + 64 * LDW;            // [o][h]
      f32x4 rv[PER], rc[PER];
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int idx = min(tid + u * NTH, 1023);
        rv[u] = reinterpret_cast<const f32x4*>(wv)[idx];
        rc[u] = reinterpret_cast<const f32x4*>(wc)[idx];
      }
      __syncthreads();      // every wave is done with the Q / K tiles
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int idx = tid + u * NTH;
        if (idx < 1024) {
          const int row = idx >> 4, c4 = (idx & 15) * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) { wvl[row * LDW + c4 + j] = rv[u][j]; wcl[row * LDW + c4 + j] = rc[u][j]; }
        }
      }
      __syncthreads();
      if (tid < 64) {      // one wave: its LDS accesses are in order
        float acc = bv[tid];
#pragma unroll 16
        for (int k = 0; k < 64; ++k) acc += wvl[tid * LDW + k] * sm.cpart[0][k];
        if (a1_out != nullptr) a1_out[(size_t)b * 64 + tid] = acc;
        sm.dab[tid] = acc;
        __builtin_amdgcn_wave_barrier();
        float y = bc[tid];
#pragma unroll 16
        for (int h = 0; h < 64; ++h) y += wcl[tid * LDW + h] * sm.dab[h];
        y_out[(size_t)b * 64 + tid] = y;
      }
    }
  }
}

// Backward of the pooled tail y = W_c (W_v ctx + b_v) + b_c (single head; models/layers.py:74-77 on the mean-pooled vector)
// inside the attention backward launch: every graph's workgroup derives its g_ctx = W_v^T W_c^T gy on the fly (each wave
// on its own: lane = component, the other operand broadcast by v_readlane -- no barrier), and TAIL_SLABS extra workgroups
// (blockIdx.x >= B; each owns 8 rows of both matrices) contract the B samples into the parameter gradients in ascending
// sample order (deterministic):
//   gtail = dW_v [64][64] | db_v [64] | dW_c [64][64] | db_c [64]      (the record layout of is_mlp2_bwd)
struct AttnTailBwd {
  const float *gy, *wv, *wc, *pooled, *a1;
  float* gtail;
  int B;
};

__device__ __forceinline__ float attn_tail_matvec_t(const float* __restrict__ w, float v, int lane) {
  // out[lane] = sum_o w[o][lane] * v[o]   (v[o] lives in lane o)
  float acc = 0.0f;
#pragma unroll
  for (int o0 = 0; o0 < 64; o0 += 16) {
    float wr[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) wr[u] = w[(o0 + u) * 64 + lane];
#pragma unroll
    for (int u = 0; u < 16; ++u)      // component o0 + u lives in that lane: a scalar broadcast, not a trip through the LDS crossbar
      acc += wr[u] * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), o0 + u));
  }
  return acc;
}

__device__ __forceinline__ float attn_tail_gctx(const AttnTailBwd& t, int b, int lane) {
  const float gyv = t.gy[(size_t)b * 64 + lane];
  const float ghid = attn_tail_matvec_t(t.wc, gyv, lane);      // d hid = W_c^T gy
  return attn_tail_matvec_t(t.wv, ghid, lane);                 // d ctx = W_v^T d hid
}

constexpr int TAIL_SLABS = 8;      // extra workgroups: each owns TAIL_ROWS rows of dW_c and of dW_v
constexpr int TAIL_ROWS = 64 / TAIL_SLABS;

template <int NTHREADS>
__device__ __forceinline__ void attn_tail_wgrad(float* lds, const AttnTailBwd& t, int slab, int tid) {
  constexpr int S = 16;                                      // samples per chunk
  constexpr int OUT = 2 * TAIL_ROWS * 64;                    // outputs of this workgroup (both matrices)
  constexpr int PER = (OUT + NTHREADS - 1) / NTHREADS;
  float* gys = lds;                    // [S][64]  gy
  float* a1s = gys + S * 64;           // [S][64]  hid
  float* pls = a1s + S * 64;           // [S][64]  pooled ctx
  float* ghs = pls + S * 64;           // [S][TAIL_ROWS]  d hid, this slab's components
  const int r0 = slab * TAIL_ROWS;
  float acc[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) acc[k] = 0.0f;
  float bsum = 0.0f;                   // tid < TAIL_ROWS: db_c[r0 + tid]; TAIL_ROWS <= tid < 2 TAIL_ROWS: db_v[r0 + tid - TAIL_ROWS]
  for (int s0 = 0; s0 < t.B; s0 += S) {
    __syncthreads();
    for (int i = tid; i < S * 64; i += NTHREADS) {
      const int s = s0 + i / 64;
      const bool ok = s < t.B;
      const size_t off = (size_t)(ok ? s : 0) * 64 + (i & 63);
      gys[i] = ok ? t.gy[off] : 0.0f;
      a1s[i] = ok ? t.a1[off] : 0.0f;
      pls[i] = ok ? t.pooled[off] : 0.0f;
    }
    __syncthreads();
    for (int i = tid; i < S * TAIL_ROWS; i += NTHREADS) {      // d hid[s][h] = sum_o W_c[o][h] gy[s][o], h in the slab
      const int s = i / TAIL_ROWS, h = r0 + i % TAIL_ROWS;
      float v = 0.0f;
#pragma unroll 16
      for (int o = 0; o < 64; ++o) v += t.wc[o * 64 + h] * gys[s * 64 + o];
      ghs[i] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int idx = tid + k * NTHREADS;
      if (idx < OUT) {
        const int mat = idx / (TAIL_ROWS * 64), rr = (idx / 64) % TAIL_ROWS, j = idx & 63;
        const float* av = mat == 0 ? gys + r0 + rr : ghs + rr;      // dW_c rows: gy components; dW_v rows: d hid components
        const int as = mat == 0 ? 64 : TAIL_ROWS;
        const float* bv = mat == 0 ? a1s + j : pls + j;
        float v = acc[k];
#pragma unroll
        for (int s = 0; s < S; ++s) v += av[s * as] * bv[s * 64];
        acc[k] = v;
      }
    }
    if (tid < 2 * TAIL_ROWS) {
      const bool c = tid < TAIL_ROWS;
      const float* src = c ? gys + r0 + tid : ghs + (tid - TAIL_ROWS);
      const int st = c ? 64 : TAIL_ROWS;
      for (int s = 0; s < S; ++s) bsum += src[s * st];
    }
  }
  float* dWv = t.gtail;
  float* dbv = dWv + 64 * 64;
  float* dWc = dbv + 64;
  float* dbc = dWc + 64 * 64;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int idx = tid + k * NTHREADS;
    if (idx < OUT) {
      const int mat = idx / (TAIL_ROWS * 64), rr = (idx / 64) % TAIL_ROWS, j = idx & 63;
      (mat == 0 ? dWc : dWv)[(r0 + rr) * 64 + j] = acc[k];
    }
  }
  if (tid < TAIL_ROWS) dbc[r0 + tid] = bsum;
  else if (tid < 2 * TAIL_ROWS) dbv[r0 + tid - TAIL_ROWS] = bsum;
}

constexpr int DAB_ROWS = 16;      // x rows in flight per wave in the d abar / direct-term loop (32: one round trip for n = 190, but 116 spilled registers)

// One workgroup per graph.  The backward is two matrix passes per 32-row block -- dQ = dS K for the block's queries (pass A, from
// its probability tiles transposed through LDS) and dK = dS^T Q for its keys (pass B) -- of 192 MFMAs (32x32x2: 12 k cycles) each.
// Round 3 ran them one after the other on NT waves: with NT = 6 waves on 4 SIMDs two SIMDs carry two waves, and each pass lasted
// 2 x 12 k cycles + its LDS latencies (31 k + 34 k of the launch's 100 k cycles, tools/attn_stamps.py).  DUAL (K and Q rows fit
// the LDS together: every shape of the models' defaults): 2 NT waves, waves [0, NT) run pass A and waves [NT, 2 NT) pass B AT THE
// SAME TIME -- three waves per SIMD, 12 x 12 k / 4 = 37 k cycles of matrix work for both passes; the d abar / direct-term rows
// are spread over all 2 NT waves (one round of 16 row loads per wave for n = 190).  Outputs are disjoint rows: bit-identical to
// the sequential form, which remains for shapes whose K + Q rows exceed the LDS (n > 192 with one head).
// (Round 3 also tried TWO workgroups per graph: measured slower -- duplicated staging and t_i work -- and removed in round 4.)
template <int NT, int D>
__global__ __launch_bounds__((attn_bwd_dual<NT, D>() ? 128 : 64) * NT) void attn_colmean_bwd_kernel(
    const float* __restrict__ qk, const float* __restrict__ x, const float* __restrict__ abar_in,
    const float* __restrict__ probs, const float* __restrict__ g_ctx, float* __restrict__ dqk,
    float* __restrict__ dx, int n, int heads, AttnTailBwd tail) {
  constexpr int LDQ = AttnBwdSmem<NT, D>::LDQ;
  constexpr int CT = (D + 31) / 32;
  constexpr bool DUAL = AttnBwdSmem<NT, D>::DUAL;
  constexpr int NW = DUAL ? 2 * NT : NT, NTHREADS = 64 * NW;
  __shared__ AttnBwdSmem<NT, D> sm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x;
  const bool do_a = !DUAL || wave < NT, do_b = !DUAL || wave >= NT;      // (wave-uniform) this wave's passes
  const int blk = (DUAL && wave >= NT) ? wave - NT : wave;              // its 32-row block
  if constexpr (D == 64) {
    if (tail.gy != nullptr && b0 >= tail.B) {      // the extra workgroups: parameter gradients of the pooled tail
      attn_tail_wgrad<NTHREADS>(sm.kq, tail, b0 - tail.B, tid);
      return;
    }
  }
  const int r = lane & 31, hf = lane >> 5;
  const float scale = rsqrtf((float)D);
  const float coef = scale / (float)n;
  // direct term dx_j[c] = sum_h abar_h[j] g_ctx_h[c]: thread (wave, lane = c) owns rows j = wave, wave + NW, ...
  // and accumulates over the heads in global memory (same thread, same address: no race)

  STAMPA(0);
  float g_tail = 0.0f;
  if constexpr (D == 64) {
    if (tail.gy != nullptr) g_tail = attn_tail_gctx(tail, b0, lane);
    __builtin_amdgcn_sched_barrier(0);      // keep the two mat-vecs' loads out of the register-heavy passes below
  }
  float* qrows = DUAL ? sm.qs : sm.kq;      // where pass B finds the Q rows
  for (int hd = 0; hd < heads; ++hd) {
    // (opaque per head: every row address below would otherwise be formed once in front of the loop -- some sixty 64-bit values
    //  that do not fit 170 registers and came back from scratch inside the staging and row loops: 150 k instead of 100 k cycles)
    int b = b0;
    asm volatile("" : "+s"(b));
    __syncthreads();
    attn_stage_half<NT, D>(sm.kq, qk, 1, b, n, hd, tid, NTHREADS);       // K rows
    if constexpr (DUAL) attn_stage_half<NT, D>(sm.qs, qk, 0, b, n, hd, tid, NTHREADS);       // Q rows
    STAMPA(1);
    const float* gc = g_ctx != nullptr ? g_ctx + (size_t)(b * heads + hd) * 64 : nullptr;
    const float* pbase = probs + (size_t)(b * heads + hd) * NT * NT * 1024;
    // dabar_j = g_ctx . x_j and the direct term dx_j = abar_j g_ctx: wave w owns rows j = w, w + NW, ... with
    // lane = channel (one coalesced 256-byte row per load, 16 rows in flight)
    {
      const float g = (gc != nullptr) ? gc[lane] : g_tail;
      for (int j0 = wave; j0 < NT * 32; j0 += DAB_ROWS * NW) {
        float xv[DAB_ROWS], ab[DAB_ROWS];
#pragma unroll
        for (int u = 0; u < DAB_ROWS; ++u) {
          const int j = j0 + u * NW;
          xv[u] = (j < n) ? x[(size_t)(b * n + j) * 64 + lane] : 0.f;
          ab[u] = (j < n) ? abar_in[(size_t)(b * heads + hd) * n + j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < DAB_ROWS; ++u) {
          const int j = j0 + u * NW;
          // 16-lane rows by DPP (every lane of a row then holds the row's sum), then the four rows by two broadcast adds
          // (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3: lanes 48..63 hold the wave's sum) -- no trip
          // through the LDS crossbar
          float d = sum_over_r16(xv[u] * g);
          d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x142, 0xA, 0xF, false));
          d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x143, 0xC, 0xF, false));
          if (j < NT * 32 && lane == 63) sm.dab[j] = d;
          if (j < n) {
            float* dst = dx + (size_t)(b * n + j) * 64 + lane;
            *dst = (hd == 0 ? 0.0f : *dst) + ab[u] * g;
          }
        }
      }
    }
    // pass A's query-block tiles (keys on the lanes); issued AFTER the x rows: vector loads return in order, the reduction
    // above must not wait behind these 24 KB.  They are only needed for t_i here -- held in registers across the (not unrolled)
    // dQ loop below the array was indexed dynamically and lived in SCRATCH (round 3: 400 bytes per lane); the dQ loop re-reads
    // its tiles one ahead of their use, as pass B does (L2 hits)
    float tpart[16];
    {
      constexpr int G = (DUAL || NT > 6) ? NT / 2 : NT;      // tiles held at once (170 registers per wave with 2 NT waves)
      float pq[G][16];
      if (do_a) {
#pragma unroll
        for (int nt = 0; nt < G; ++nt)
#pragma unroll
          for (int t = 0; t < 16; ++t) pq[nt][t] = pbase[((blk * NT + nt) * 16 + t) * 64 + lane];
      }
      STAMPA(2);
      __syncthreads();
      STAMPA(3);
      // ---- t_i = sum_j P_ij dabar_j for query block `blk` (keys on the lanes -> lane reduction) ----
      if (do_a) {
#pragma unroll
        for (int t = 0; t < 16; ++t) tpart[t] = 0.f;
#pragma unroll
        for (int g0 = 0; g0 < NT; g0 += G) {
          if (g0 > 0) {
#pragma unroll
            for (int nt = 0; nt < G; ++nt)
#pragma unroll
              for (int t = 0; t < 16; ++t) pq[nt][t] = pbase[((blk * NT + g0 + nt) * 16 + t) * 64 + lane];
          }
#pragma unroll
          for (int nt = 0; nt < G; ++nt) {
            const float dj = sm.dab[(g0 + nt) * 32 + r];
#pragma unroll
            for (int t = 0; t < 16; ++t) tpart[t] += pq[nt][t] * dj;
          }
        }
      }
    }
    if (do_a) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const float ti = sum_over_r(tpart[t]);
        if (r == 0) sm.tvec[blk * 32 + tile_row(t, hf)] = ti;
      }
    }
    if constexpr (DUAL) __syncthreads();      // t_i of every query block is in LDS (pass B reads them all); Q and K rows are staged
    STAMPA(4);
    // ---- pass A: dQ[i][c] = sum_j dS[i][j] K[j][c] for query block `blk`: transpose each P tile (wave-private LDS) so that
    //      the queries sit on the lanes; the dS registers are then the MFMA A operand ----
    if (do_a) {
      __builtin_amdgcn_wave_barrier();
      const float ti = sm.tvec[blk * 32 + r];       // query i = blk*32 + r on the lanes from here on
      f32x16 dq[CT];
      zero_acc(dq);
      float* tr = sm.tr[blk];
      // tiles are fetched TWO ahead of their use (a tile's transposition + 32 MFMAs last about as long as one L2 / MALL round
      // trip: one tile ahead left part of it exposed)
      float pk[16], pn[16], p2[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        pk[t] = pbase[((blk * NT + 0) * 16 + t) * 64 + lane];
        pn[t] = pbase[((blk * NT + min(1, NT - 1)) * 16 + t) * 64 + lane];
      }
#pragma unroll 1
      for (int nt = 0; nt < NT; ++nt) {
        const int nn = min(nt + 2, NT - 1);
#pragma unroll
        for (int t = 0; t < 16; ++t) p2[t] = pbase[((blk * NT + nn) * 16 + t) * 64 + lane];
#pragma unroll
        for (int t = 0; t < 16; ++t) tr[tile_row(t, hf) * 33 + r] = pk[t];      // [query row][key col]
        __builtin_amdgcn_wave_barrier();
        // the LDS operands of TB k-steps first, then their MFMAs back to back (an LDS read in front of every MFMA exposes its
        // latency 16 times per tile; all sixteen steps' operands at once are 48 registers -- too many at three waves per SIMD)
        constexpr int TB = DUAL ? 8 : 16;      // k-steps whose operands are fetched together
#pragma unroll
        for (int t0 = 0; t0 < 16; t0 += TB) {
          float ds[TB], kv[TB][CT];
#pragma unroll
          for (int u = 0; u < TB; ++u) {
            const int jl = tile_row(t0 + u, hf), j = nt * 32 + jl;
            ds[u] = tr[r * 33 + jl] * (sm.dab[j] - ti) * coef;                          // P[i = lane][j] -> dS
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              const int c = ct * 32 + r;
              kv[u][ct] = (c < D) ? sm.kq[j * LDQ + c] : 0.f;
            }
          }
          __builtin_amdgcn_sched_barrier(0);     // keep the LDS reads above, the MFMAs below (the scheduler re-interleaves them)
#pragma unroll
          for (int u = 0; u < TB; ++u)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) dq[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[u], kv[u][ct], dq[ct], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 16; ++t) { pk[t] = pn[t]; pn[t] = p2[t]; }
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int ii = blk * 32 + tile_row(t, hf), c = ct * 32 + r;
          if (ii < n && c < D) dqk[(size_t)(b * n + ii) * 128 + hd * D + c] = dq[ct][t];
        }
    }
    STAMPA(5);
    if constexpr (!DUAL) {
      __syncthreads();   // t_i of every query block is in LDS; every wave is done with the K rows
      STAMPA(6);
      attn_stage_half<NT, D>(sm.kq, qk, 0, b, n, hd, tid, NTHREADS);       // Q rows
      __syncthreads();
      STAMPA(7);
    }
    // ---- pass B: key block `blk` (keys on the lanes, queries on the registers): dK[j][c] = sum_i dS[i][j] Q[i][c] ----
    if (do_b) {      // (wave-uniform)
      f32x16 dk[CT];
      zero_acc(dk);
      const float dabj = sm.dab[blk * 32 + r];
      float pk[16], pn[16], p2[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        pk[t] = pbase[((0 * NT + blk) * 16 + t) * 64 + lane];
        pn[t] = pbase[((min(1, NT - 1) * NT + blk) * 16 + t) * 64 + lane];
      }
#pragma unroll 1
      for (int mt = 0; mt < NT; ++mt) {
        const int mn = min(mt + 2, NT - 1);                    // two tiles ahead
#pragma unroll
        for (int t = 0; t < 16; ++t) p2[t] = pbase[((mn * NT + blk) * 16 + t) * 64 + lane];
#pragma unroll
        for (int t0 = 0; t0 < 16; t0 += 8) {
          float ds[8], qv[8][CT];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = mt * 32 + tile_row(t0 + u, hf);
            ds[u] = pk[t0 + u] * (dabj - sm.tvec[i]) * coef;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              const int c = ct * 32 + r;
              qv[u][ct] = (c < D) ? qrows[i * LDQ + c] : 0.f;
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) dk[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[u], qv[u][ct], dk[ct], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) { pk[t] = pn[t]; pn[t] = p2[t]; }
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int jj = blk * 32 + tile_row(t, hf), c = ct * 32 + r;
          if (jj < n && c < D) dqk[(size_t)(b * n + jj) * 128 + 64 + hd * D + c] = dk[ct][t];
        }
    }
    STAMPA(8);
  }
}

// =====================================================================================================================
// 16-row blocks (round 4): single head (D = 64), n <= 192.  One WAVE per 16-row block on v_mfma_f32_16x16x4_f32 -- NB = 4 / 8 /
// 12 waves per graph, for n = 190 three per SIMD on all four SIMDs where the 32-row kernels above put six waves on four SIMDs
// (two SIMDs carried two waves: every matrix phase lasted twice its share).  A block's operands are small (16 accumulator
// registers per output tile row, 4 registers per stored probability tile), so twelve waves fit their 170 registers; the phases
// in front of the matrix passes (staging, d abar rows, t_i) spread over twice the waves.
//   probs layout: [b][query block][key block][register t (4)][lane (64)] -- tile (qb, kb) in the 16x16x4 accumulator layout
//   (lane (r, q), register t = P[16 qb + 4 q + t][16 kb + r]): 256-byte rows, same total size as the 32-row layout.
//   backward: K (pass A) and Q (pass B) are staged TRANSPOSED and split by key / query quarter,
//       T[(j % 16) / 4][c][4 (j / 16) + j % 4]   (row j, column c; plane stride a multiple of 64 floats, row stride 52),
//   so that the B operand of four k-steps is ONE conflict-free ds_read_b128 per output tile (the k-slots of step s are the rows
//   4 q + s of the 16-row tile), as are the A operand (from the wave's transposition tile) and the d abar / t_i values.
constexpr int A16_LDQ = 68;

template <int NB>
struct Attn16FwdSmem {
  float qs[NB * 16 * A16_LDQ];
  float ks[NB * 16 * A16_LDQ];
  float wpart[NB][NB * 16];
  float abar[NB * 16];
  float cpart[NB][64];
  float hid[64];
};

template <int NB>
__global__ __launch_bounds__(64 * NB) void attn16_fwd_kernel(
    const float* __restrict__ qk, const float* __restrict__ x, float* __restrict__ ctx, float* __restrict__ abar_out,
    float* __restrict__ probs, int n, const float* __restrict__ wv, const float* __restrict__ bv,
    const float* __restrict__ wc, const float* __restrict__ bc, float* __restrict__ a1_out, float* __restrict__ y_out) {
  static_assert(NB % 4 == 0, "column tiles are walked four at a time");
  constexpr int LDQ = A16_LDQ, NTH = 64 * NB;
  __shared__ Attn16FwdSmem<NB> sm;
  const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.x;
  const int blk = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const float scale = 0.125f;      // 1 / sqrt(64)
  // ---- stage Q and K rows (all loads of a thread's batch first) ----
  for (int idx0 = tid; idx0 < NB * 16 * 16; idx0 += 4 * NTH) {      // 16 float4 per row
    f32x4 qv[4], kv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = idx0 + u * NTH, lr = idx >> 4, c4 = idx & 15;
      const bool ok = idx < NB * 16 * 16 && lr < n;
      const float* src = qk + (size_t)(b * n + (ok ? lr : 0)) * 128 + c4 * 4;
      qv[u] = ok ? *reinterpret_cast<const f32x4*>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
      kv[u] = ok ? *reinterpret_cast<const f32x4*>(src + 64) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = idx0 + u * NTH, lr = idx >> 4, c4 = idx & 15;
      if (idx < NB * 16 * 16) {
        *reinterpret_cast<f32x4*>(sm.qs + lr * LDQ + c4 * 4) = qv[u];
        *reinterpret_cast<f32x4*>(sm.ks + lr * LDQ + c4 * 4) = kv[u];
      }
    }
  }
  __syncthreads();
  {
    // ---- S = Q_blk K^T, 16 x (16 NB), four column tiles at a time; softmax over the row; column sums ----
    f32x4 acc[NB];
#pragma unroll
    for (int cg = 0; cg < NB / 4; ++cg) {
      f32x4 a4[4];
      zero_acc4(a4);
      mm16_rows<4, 64, LDQ, LDQ>(a4, sm.qs + blk * 16 * LDQ, sm.ks + cg * 64 * LDQ, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[cg * 4 + j] = a4[j];
    }
    float* ptile = probs != nullptr ? probs + ((size_t)b * NB + blk) * NB * 256 : nullptr;
    float colsum[NB];
#pragma unroll
    for (int nt = 0; nt < NB; ++nt) colsum[nt] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = blk * 16 + 4 * q + t;
      float m = -INFINITY;
#pragma unroll
      for (int nt = 0; nt < NB; ++nt) {
        const float sv = (nt * 16 + r < n) ? acc[nt][t] * scale : -INFINITY;
        acc[nt][t] = sv;
        m = fmaxf(m, sv);
      }
      m = fmaxf(m, dpp_move<0xB1>(m));
      m = fmaxf(m, dpp_move<0x4E>(m));
      m = fmaxf(m, dpp_move<0x141>(m));
      m = fmaxf(m, dpp_move<0x140>(m));
      float l = 0.f;
#pragma unroll
      for (int nt = 0; nt < NB; ++nt) {
        const float e = __expf(acc[nt][t] - m);   // exp(-inf) = 0 for masked columns
        acc[nt][t] = e;
        l += e;
      }
      l = sum_over_r16(l);
      const float inv = 1.0f / l;
#pragma unroll
      for (int nt = 0; nt < NB; ++nt) {
        const float pv = (i < n) ? acc[nt][t] * inv : 0.0f;      // rows of padded queries are stored as zeros
        colsum[nt] += pv;
        if (ptile != nullptr) ptile[(nt * 4 + t) * 64 + lane] = pv;
      }
    }
#pragma unroll
    for (int nt = 0; nt < NB; ++nt) {      // the four row quarters of a column: lanes r, r + 16, r + 32, r + 48
      float v = colsum[nt];
      v += swap_rows16(v);
      v += __shfl_xor(v, 32, 64);
      if (q == 0) sm.wpart[blk][nt * 16 + r] = v;
    }
  }
  __syncthreads();
  for (int j = tid; j < NB * 16; j += NTH) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NB; ++w) v += sm.wpart[w][j];
    v /= (float)n;
    sm.abar[j] = v;
    if (j < n && abar_out != nullptr) abar_out[(size_t)b * n + j] = v;
  }
  __syncthreads();
  {
    float a = 0.f;
    for (int j0 = blk; j0 < n; j0 += 8 * NB) {     // a fixed order per wave, 8 row loads in flight
      float xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 + u * NB;
        xv[u] = (j < n) ? x[(size_t)(b * n + j) * 64 + lane] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 + u * NB;
        if (j < n) a += sm.abar[j] * xv[u];
      }
    }
    sm.cpart[blk][lane] = a;
  }
  __syncthreads();
  if (tid < 64) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NB; ++w) v += sm.cpart[w][tid];
    ctx[(size_t)b * 64 + tid] = v;
    sm.cpart[0][tid] = v;      // (own column: read above by this thread only)
  }
  if (wv != nullptr) {
    // pooled tail of the single head (models/layers.py:74-77 on the mean-pooled vector): hid = W_v ctx + b_v, y = W_c hid + b_c
    // (see attn_colmean_fwd_kernel): both matrices row-major with a pitch of 65 where the dead Q tile was, all loads first
    constexpr int LDW = 65, PER = (1024 + NTH - 1) / NTH;
    float* wvl = sm.qs;
    float* wcl = sm.qs + 64 * LDW;
    f32x4 rv[PER], rc[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int idx = min(tid + u * NTH, 1023);
      rv[u] = reinterpret_cast<const f32x4*>(wv)[idx];
      rc[u] = reinterpret_cast<const f32x4*>(wc)[idx];
    }
    __syncthreads();      // every wave is done with the Q / K tiles (and cpart[0] is complete)
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int idx = tid + u * NTH;
      if (idx < 1024) {
        const int row = idx >> 4, c4 = (idx & 15) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) { wvl[row * LDW + c4 + j] = rv[u][j]; wcl[row * LDW + c4 + j] = rc[u][j]; }
      }
    }
    __syncthreads();
    if (tid < 64) {      // one wave: its LDS accesses are in order
      float acc1 = bv[tid];
#pragma unroll 16
      for (int k = 0; k < 64; ++k) acc1 += wvl[tid * LDW + k] * sm.cpart[0][k];
      if (a1_out != nullptr) a1_out[(size_t)b * 64 + tid] = acc1;
      sm.hid[tid] = acc1;
      __builtin_amdgcn_wave_barrier();
      float y = bc[tid];
#pragma unroll 16
      for (int h = 0; h < 64; ++h) y += wcl[tid * LDW + h] * sm.hid[h];
      y_out[(size_t)b * 64 + tid] = y;
    }
  }
}

constexpr int A16_TLD = 52;      // floats per column of a quarter plane (NB <= 12 tiles x 4 + pad; 13 slots: odd)
template <int NB>
struct Attn16BwdSmem {
  static constexpr int PLANE = 64 * A16_TLD;      // 3328 floats = 52 x 64
  float kqT[4 * PLANE];       // K rows (pass A), then Q rows (pass B), transposed and split by row quarter
  float dab[NB * 16];
  float tvec[NB * 16];
  float tr[NB][16 * 20];      // wave-private transposition tile
};

// stage Q (which = 0) or K (1) transposed: element (row j, column c) -> T[(j % 16) / 4][c][4 (j / 16) + j % 4]
template <int NB>
__device__ __forceinline__ void attn16_stage_T(float* dst, const float* __restrict__ qk, int which, int b, int n, int tid) {
  constexpr int NTH = 64 * NB, PLANE = Attn16BwdSmem<NB>::PLANE;
  for (int idx0 = tid; idx0 < NB * 16 * 16; idx0 += 4 * NTH) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = idx0 + u * NTH, lr = idx >> 4, c4 = idx & 15;
      const bool ok = idx < NB * 16 * 16 && lr < n;
      v[u] = ok ? *reinterpret_cast<const f32x4*>(qk + (size_t)(b * n + (ok ? lr : 0)) * 128 + which * 64 + c4 * 4)
                : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = idx0 + u * NTH, lr = idx >> 4, c4 = idx & 15;
      if (idx < NB * 16 * 16) {
        float* base = dst + ((lr & 15) >> 2) * PLANE + 4 * (lr >> 4) + (lr & 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) base[(c4 * 4 + j) * A16_TLD] = v[u][j];
      }
    }
  }
}

template <int NB>
__global__ __launch_bounds__(64 * NB) void attn16_bwd_kernel(
    const float* __restrict__ qk, const float* __restrict__ x, const float* __restrict__ abar_in,
    const float* __restrict__ probs, const float* __restrict__ g_ctx, float* __restrict__ dqk,
    float* __restrict__ dx, int n, AttnTailBwd tail) {
  constexpr int NTH = 64 * NB, PLANE = Attn16BwdSmem<NB>::PLANE;
  __shared__ Attn16BwdSmem<NB> sm;
  const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.x;
  const int blk = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tail.gy != nullptr && b >= tail.B) {      // the extra workgroups: parameter gradients of the pooled tail
    attn_tail_wgrad<NTH>(sm.kqT, tail, b - tail.B, tid);
    return;
  }
  const int r = lane & 15, q = lane >> 4;
  const float coef = 0.125f / (float)n;
  STAMPA(0);
  float g_tail = 0.0f;
  if (tail.gy != nullptr) g_tail = attn_tail_gctx(tail, b, lane);
  __builtin_amdgcn_sched_barrier(0);
  attn16_stage_T<NB>(sm.kqT, qk, 1, b, n, tid);       // K rows
  STAMPA(1);
  const float* pbase = probs + (size_t)b * NB * NB * 256;
  // dabar_j = g_ctx . x_j and the direct term dx_j = abar_j g_ctx: wave w owns rows j = w, w + NB, ... (lane = channel)
  {
    const float g = (g_ctx != nullptr) ? g_ctx[(size_t)b * 64 + lane] : g_tail;
    for (int j0 = blk; j0 < NB * 16; j0 += DAB_ROWS * NB) {
      float xv[DAB_ROWS], ab[DAB_ROWS];
#pragma unroll
      for (int u = 0; u < DAB_ROWS; ++u) {
        const int j = j0 + u * NB;
        xv[u] = (j < n) ? x[(size_t)(b * n + j) * 64 + lane] : 0.f;
        ab[u] = (j < n) ? abar_in[(size_t)b * n + j] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < DAB_ROWS; ++u) {
        const int j = j0 + u * NB;
        float d = sum_over_r16(xv[u] * g);
        d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x142, 0xA, 0xF, false));
        d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x143, 0xC, 0xF, false));
        if (j < NB * 16 && lane == 63) sm.dab[j] = d;
        if (j < n) dx[(size_t)(b * n + j) * 64 + lane] = ab[u] * g;
      }
    }
  }
  // ---- t_i = sum_j P_ij dabar_j for the block's 16 queries: its NB probability tiles, keys on the lanes ----
  float tpart[4] = {0.f, 0.f, 0.f, 0.f};
  {
    f32x4 pq[NB];
#pragma unroll
    for (int nt = 0; nt < NB; ++nt)
#pragma unroll
      for (int t = 0; t < 4; ++t) pq[nt][t] = pbase[((blk * NB + nt) * 4 + t) * 64 + lane];
    STAMPA(2);
    __syncthreads();      // d abar of every row is in LDS; the K rows are staged
    STAMPA(3);
#pragma unroll
    for (int nt = 0; nt < NB; ++nt) {
      const float dj = sm.dab[nt * 16 + r];
#pragma unroll
      for (int t = 0; t < 4; ++t) tpart[t] += pq[nt][t] * dj;
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float ti = sum_over_r16(tpart[t]);
    if (r == 0) sm.tvec[blk * 16 + 4 * q + t] = ti;
  }
  __builtin_amdgcn_wave_barrier();
  STAMPA(4);
  {
    // ---- pass A: dQ[i][c] = sum_j dS[i][j] K[j][c] for the block's queries.  Each P tile is transposed through the wave's LDS
    //      tile so that the queries sit on the lanes; k-slot q of step s is key 4 q + s of the tile ----
    const float ti = sm.tvec[blk * 16 + r];
    f32x4 dq[4];
    zero_acc4(dq);
    float* tr = sm.tr[blk];
    f32x4 pk, pn, p2;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      pk[t] = pbase[((blk * NB + 0) * 4 + t) * 64 + lane];
      pn[t] = pbase[((blk * NB + 1) * 4 + t) * 64 + lane];
    }
#pragma unroll 1
    for (int kt = 0; kt < NB; ++kt) {
      const int nn = min(kt + 2, NB - 1);      // tiles are fetched two ahead of their use
#pragma unroll
      for (int t = 0; t < 4; ++t) p2[t] = pbase[((blk * NB + nn) * 4 + t) * 64 + lane];
#pragma unroll
      for (int t = 0; t < 4; ++t) tr[(4 * q + t) * 20 + r] = pk[t];      // [query row][key col]
      __builtin_amdgcn_wave_barrier();
      const f32x4 pa = *reinterpret_cast<const f32x4*>(tr + r * 20 + 4 * q);      // P[i = r][keys 4 q .. 4 q + 3]
      const f32x4 dj = *reinterpret_cast<const f32x4*>(sm.dab + kt * 16 + 4 * q);
      f32x4 kb[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) kb[nt] = *reinterpret_cast<const f32x4*>(sm.kqT + q * PLANE + (nt * 16 + r) * A16_TLD + kt * 4);
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx) {
        const float ds = pa[sidx] * (dj[sidx] - ti) * coef;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) dq[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds, kb[nt][sidx], dq[nt], 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();
      pk = pn; pn = p2;
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ii = blk * 16 + 4 * q + t;
        if (ii < n) dqk[(size_t)(b * n + ii) * 128 + nt * 16 + r] = dq[nt][t];
      }
  }
  STAMPA(5);
  __syncthreads();   // t_i of every block is in LDS; every wave is done with the K rows
  STAMPA(6);
  attn16_stage_T<NB>(sm.kqT, qk, 0, b, n, tid);       // Q rows
  __syncthreads();
  STAMPA(7);
  {
    // ---- pass B: dK[j][c] = sum_i dS[i][j] Q[i][c] for the block's keys (keys on the lanes): the tile registers ARE the A
    //      operand (k-slot q of step t is query 4 q + t of the tile) ----
    f32x4 dk[4];
    zero_acc4(dk);
    const float dabj = sm.dab[blk * 16 + r];
    f32x4 pk, pn, p2;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      pk[t] = pbase[((0 * NB + blk) * 4 + t) * 64 + lane];
      pn[t] = pbase[((1 * NB + blk) * 4 + t) * 64 + lane];
    }
#pragma unroll 1
    for (int qt = 0; qt < NB; ++qt) {
      const int nn = min(qt + 2, NB - 1);
#pragma unroll
      for (int t = 0; t < 4; ++t) p2[t] = pbase[((nn * NB + blk) * 4 + t) * 64 + lane];
      const f32x4 tv = *reinterpret_cast<const f32x4*>(sm.tvec + qt * 16 + 4 * q);
      f32x4 qb[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) qb[nt] = *reinterpret_cast<const f32x4*>(sm.kqT + q * PLANE + (nt * 16 + r) * A16_TLD + qt * 4);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float ds = pk[t] * (dabj - tv[t]) * coef;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) dk[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds, qb[nt][t], dk[nt], 0, 0, 0);
      }
      pk = pn; pn = p2;
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int jj = blk * 16 + 4 * q + t;
        if (jj < n) dqk[(size_t)(b * n + jj) * 128 + 64 + nt * 16 + r] = dk[nt][t];
      }
  }
  STAMPA(8);
}

}  // namespace is

#ifdef IS_STAGE_STAMPS
extern "C" int is_debug_stamps_attn(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(is::g_stamps_attn), sizeof(long long) * 16) == hipSuccess ? 0 : is::fail(__func__, -5);
}
#endif

#define ATTN_DISPATCH(KERNEL, ...)                                                                                   \
  do {                                                                                                               \
    const int nt = (n + 31) / 32;                                                                                    \
    hipStream_t st = static_cast<hipStream_t>(stream);                                                               \
    if (heads == 1) {                                                                                                \
      if (nt <= 2) hipLaunchKernelGGL((is::KERNEL<2, 64>), dim3(B), dim3(128), 0, st, __VA_ARGS__);                   \
      else if (nt <= 4) hipLaunchKernelGGL((is::KERNEL<4, 64>), dim3(B), dim3(256), 0, st, __VA_ARGS__);              \
      else if (nt <= 6) hipLaunchKernelGGL((is::KERNEL<6, 64>), dim3(B), dim3(384), 0, st, __VA_ARGS__);              \
      else hipLaunchKernelGGL((is::KERNEL<8, 64>), dim3(B), dim3(512), 0, st, __VA_ARGS__);                           \
    } else {                                                                                                         \
      if (nt <= 2) hipLaunchKernelGGL((is::KERNEL<2, 8>), dim3(B), dim3(128), 0, st, __VA_ARGS__);                    \
      else if (nt <= 4) hipLaunchKernelGGL((is::KERNEL<4, 8>), dim3(B), dim3(256), 0, st, __VA_ARGS__);               \
      else if (nt <= 6) hipLaunchKernelGGL((is::KERNEL<6, 8>), dim3(B), dim3(384), 0, st, __VA_ARGS__);               \
      else hipLaunchKernelGGL((is::KERNEL<8, 8>), dim3(B), dim3(512), 0, st, __VA_ARGS__);                            \
    }                                                                                                                \
  } while (0)

// the backward: one workgroup per graph, 2 NT waves when K and Q rows fit the LDS together (attn_bwd_dual), else NT
#define ATTN_BWD_LAUNCH(NTV, DV, ...)                                                                                           \
  hipLaunchKernelGGL((is::attn_colmean_bwd_kernel<NTV, DV>), dim3(B), dim3((is::attn_bwd_dual<NTV, DV>() ? 128 : 64) * NTV), 0, st, \
                     __VA_ARGS__)
#define ATTN_DISPATCH_BWD(GRAPHS, ...)                                                       \
  do {                                                                                       \
    const int nt = (n + 31) / 32;                                                            \
    hipStream_t st = static_cast<hipStream_t>(stream);                                       \
    if (heads == 1) {                                                                        \
      if (nt <= 2) ATTN_BWD_LAUNCH(2, 64, __VA_ARGS__);                                      \
      else if (nt <= 4) ATTN_BWD_LAUNCH(4, 64, __VA_ARGS__);                                 \
      else if (nt <= 6) ATTN_BWD_LAUNCH(6, 64, __VA_ARGS__);                                 \
      else ATTN_BWD_LAUNCH(8, 64, __VA_ARGS__);                                              \
    } else {                                                                                 \
      if (nt <= 2) ATTN_BWD_LAUNCH(2, 8, __VA_ARGS__);                                       \
      else if (nt <= 4) ATTN_BWD_LAUNCH(4, 8, __VA_ARGS__);                                  \
      else if (nt <= 6) ATTN_BWD_LAUNCH(6, 8, __VA_ARGS__);                                  \
      else ATTN_BWD_LAUNCH(8, 8, __VA_ARGS__);                                               \
    }                                                                                        \
  } while (0)

// qk [B*n, 128] = [Q | K], x [B*n, 64]; heads in {1, 8}; n <= 256 nodes per graph (all graphs equal, padded).
// ctx [B, heads, 64]; abar [B, heads, n] and probs [is_attn_colmean_probs_floats(B, n, heads)] (the attention
// probabilities in accumulator-tile order) are saved for the backward (both may be NULL).
extern "C" long long is_attn_colmean_probs_floats(int B, int n, int heads) {
  const long long nt = ((n + 31) / 32 <= 2) ? 2 : ((n + 31) / 32 <= 4) ? 4 : ((n + 31) / 32 <= 6) ? 6 : 8;
  return (long long)B * heads * nt * nt * 1024;
}

// single head, n <= 192: the 16-row-block kernels (NB = 4 / 8 / 12 waves per graph)
#define ATTN16_FWD(...)                                                                                                     \
  do {                                                                                                                      \
    hipStream_t st = static_cast<hipStream_t>(stream);                                                                      \
    if (n <= 64) hipLaunchKernelGGL((is::attn16_fwd_kernel<4>), dim3(B), dim3(256), 0, st, __VA_ARGS__);                     \
    else if (n <= 128) hipLaunchKernelGGL((is::attn16_fwd_kernel<8>), dim3(B), dim3(512), 0, st, __VA_ARGS__);               \
    else hipLaunchKernelGGL((is::attn16_fwd_kernel<12>), dim3(B), dim3(768), 0, st, __VA_ARGS__);                            \
  } while (0)
#define ATTN16_BWD(...)                                                                                                     \
  do {                                                                                                                      \
    hipStream_t st = static_cast<hipStream_t>(stream);                                                                      \
    if (n <= 64) hipLaunchKernelGGL((is::attn16_bwd_kernel<4>), dim3(B), dim3(256), 0, st, __VA_ARGS__);                     \
    else if (n <= 128) hipLaunchKernelGGL((is::attn16_bwd_kernel<8>), dim3(B), dim3(512), 0, st, __VA_ARGS__);               \
    else hipLaunchKernelGGL((is::attn16_bwd_kernel<12>), dim3(B), dim3(768), 0, st, __VA_ARGS__);                            \
  } while (0)
namespace is {
// IMMUNOSTRUCT_ATTN_TILES=32: the 32-row-block kernels everywhere (A/B runs, tests).  Read ONCE per process: a forward and its
// backward pick the layout of the saved probabilities independently (16- or 32-row tiles, same buffer size), so a value that
// changed between the two -- or between the capture of a forward and its replay -- would make the backward read the other layout
// and return wrong gradients without an error.  A/B runs select the kernels per process.
inline bool attn16_forced_off() {
  static const bool off = [] {
    const char* e = getenv("IMMUNOSTRUCT_ATTN_TILES");
    return e != nullptr && atoi(e) == 32;
  }();
  return off;
}
inline bool attn16_applies(int n, int heads) { return heads == 1 && n <= 192 && !attn16_forced_off(); }
}

extern "C" int is_attn_colmean_fwd(const float* qk, const float* x, float* ctx, float* abar, float* probs, int B, int n,
                                   int heads, void* stream) {
  if (B <= 0) return 0;
  if (n <= 0 || n > 256 || (heads != 1 && heads != 8)) return is::fail(__func__, -22);
  const float* none = nullptr;
  float* nout = nullptr;
  if (is::attn16_applies(n, heads)) {
    ATTN16_FWD(qk, x, ctx, abar, probs, n, none, none, none, none, nout, nout);
    return is::launch_status(__func__);
  }
  ATTN_DISPATCH(attn_colmean_fwd_kernel, qk, x, ctx, abar, probs, n, heads, none, none, none, none, nout, nout);
  return is::launch_status(__func__);
}

// Single head: the same pass followed, in the same launch, by the pooled tail hid = W_v ctx + b_v (a1_out [B,64], may be
// NULL), y = W_c hid + b_c (y_out [B,64]) -- the value projection and w_concat of models/layers.py:74-77 applied to the
// mean-pooled vector (what is_mlp2_fwd computes from ctx with hgroup = 64, as one launch less).
extern "C" int is_attn_colmean_fwd_tail(const float* qk, const float* x, float* ctx, float* abar, float* probs,
                                        const float* wv, const float* bv, const float* wc, const float* bc, float* a1_out,
                                        float* y_out, int B, int n, void* stream) {
  if (B <= 0) return 0;
  if (n <= 0 || n > 256 || wv == nullptr || bv == nullptr || wc == nullptr || bc == nullptr || y_out == nullptr) return is::fail(__func__, -22);
  const int heads = 1;
  if (is::attn16_applies(n, heads)) {
    ATTN16_FWD(qk, x, ctx, abar, probs, n, wv, bv, wc, bc, a1_out, y_out);
    return is::launch_status(__func__);
  }
  ATTN_DISPATCH(attn_colmean_fwd_kernel, qk, x, ctx, abar, probs, n, heads, wv, bv, wc, bc, a1_out, y_out);
  return is::launch_status(__func__);
}

// dqk [B*n, 128] (every entry written), dx [B*n, 64] (direct term through ctx = abar^T x).
extern "C" int is_attn_colmean_bwd(const float* qk, const float* x, const float* abar, const float* probs,
                                   const float* g_ctx, float* dqk, float* dx, int B, int n, int heads, void* stream) {
  if (B <= 0) return 0;
  if (n <= 0 || n > 256 || (heads != 1 && heads != 8)) return is::fail(__func__, -22);
  const is::AttnTailBwd none{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  if (is::attn16_applies(n, heads)) {
    ATTN16_BWD(qk, x, abar, probs, g_ctx, dqk, dx, n, none);
    return is::launch_status(__func__);
  }
  ATTN_DISPATCH_BWD(B, qk, x, abar, probs, g_ctx, dqk, dx, n, heads, none);
  return is::launch_status(__func__);
}

// Single head with the pooled tail of is_attn_colmean_fwd_tail behind it: gy [B,64] is the gradient of the tail's output y;
// the launch derives g_ctx per graph, runs the attention backward, and one extra workgroup writes the tail's parameter
// gradients gtail = dW_v [64,64] | db_v [64] | dW_c [64,64] | db_c [64] from pooled (= ctx) [B,64] and a1 (= hid) [B,64].
extern "C" int is_attn_colmean_bwd_tail(const float* qk, const float* x, const float* abar, const float* probs,
                                        const float* gy, const float* wv, const float* wc, const float* pooled,
                                        const float* a1, float* dqk, float* dx, float* gtail, int B_, int n, void* stream) {
  if (B_ <= 0) return 0;
  if (n <= 0 || n > 256 || gy == nullptr || wv == nullptr || wc == nullptr || pooled == nullptr || a1 == nullptr || gtail == nullptr)
    return is::fail(__func__, -22);
  const int heads = 1;
  const float* g_ctx = nullptr;
  const is::AttnTailBwd tail{gy, wv, wc, pooled, a1, gtail, B_};
  const int B = B_ + is::TAIL_SLABS;      // grid: the graphs + the parameter-gradient workgroups
  if (is::attn16_applies(n, heads)) {
    ATTN16_BWD(qk, x, abar, probs, g_ctx, dqk, dx, n, tail);
    return is::launch_status(__func__);
  }
  ATTN_DISPATCH_BWD(B_, qk, x, abar, probs, g_ctx, dqk, dx, n, heads, tail);
  return is::launch_status(__func__);
}
