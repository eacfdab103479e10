// Paired (cancer / wild-type) contrastive loss, forward and backward, as five launches.
//
// Reference: utils/contrastive.py:18-83 (PairedContrastiveLoss): both embeddings go through the projector
//   Linear(E -> Z, no bias) -> BatchNorm1d(Z) on BATCH statistics (the module is never put in eval mode,
//   procedures/train.py:76) -> ReLU -> Linear(Z -> Z, no bias),          Z = 128, E = 104
// are centred over the batch, and
//   loss = sum_ij w_ij (zc zw^T / Z - diag(pos))_ij^2 + sum_kl w_kl (zc^T zw / B - I)_kl^2
//          + 1/2 [ mean_k relu(1 - sqrt(var_k(zc) + 1e-4)) + (same for zw) ],     w = 1 on the diagonal, lambda elsewhere
// (the reference writes the two weighted sums with boolean-mask in-place scaling, :62-81; same values).
// The projector is random and frozen (it is not handed to the optimizer, procedures/train.py:76-83) but gradients
// flow through it to the embeddings: the backward returns d loss / d emb_c and d loss / d emb_w only.
// BatchNorm's running statistics are not maintained (nothing ever reads them: the module stays in train mode).
//
//   contr_side_fwd  y1 = emb W1^T -> batch statistics -> a1 -> z0 = a1 W2^T -> centre -> per-column std -> hinge.  BatchNorm's
//                   statistics, the centring and the std are PER OUTPUT COLUMN, so the side pass is cut by 32-column blocks:
//                   contr_side_fwd_a (grid 2 sides x 4 blocks: y1, statistics, a1 of the block's columns) and
//                   contr_side_fwd_b (same grid: z0 = a1 W2^T needs every column of a1 -- hence the launch boundary --
//                   then centre / std of the block's columns).  Round 3 ran each side as ONE workgroup of 16 waves
//                   (67 us forward, 60 us backward on 2 of 256 CUs, on the chain the attention backward waits for); the
//                   row-stripe / combine order of every column reduction is unchanged, so the values are bit-identical.
//                   Intermediates live in a global scratch that stays in L2 (the whole problem is ~0.5 MB).
//   contr_pair_fwd  pair = zc zw^T / Z and corr = zc^T zw / B tile by tile (32 x 32 per wave) + the weighted squared
//                   deviations -> one partial per tile;   contr_finish sums the partials in tile order.
//   contr_pair_bwd  dzc = dPair zw / Z + zw dCorr^T / B,  dzw = dPair^T zc / Z + zc dCorr / B   (tiles of B x Z)
//   contr_side_bwd  hinge + centring backward -> da1 = dz0 W2 -> ReLU / BatchNorm backward (batch statistics) -> d emb = dy1 W1,
//                   cut the same way: contr_side_bwd_a (2 x 4 blocks of da1 / dy1 columns; every workgroup forms the column
//                   means of ALL of dz0 for itself -- 2 x 64 KB from L2 -- and centres on the fly) and contr_side_bwd_b
//                   (2 x ceil(E / 32) column blocks of d emb).
// All matrix products run on v_mfma_f32_32x32x2_f32 with operands read straight from global memory (L1 hits), every
// reduction has a fixed order -> bitwise reproducible.  B <= 256 pairs, E <= 256, Z = 128.
#include "common.h"

namespace is {

constexpr int CZ = 128;          // projector width
constexpr int C_WAVES = 16;      // waves of a side workgroup
constexpr float BN_EPS = 1e-5f;
constexpr int C_GROUPS = 64 * C_WAVES / CZ;     // row stripes of a column pass (8)

__device__ __forceinline__ float group_sum(const float (*part)[CZ], int c) {
  float t = part[0][c];
#pragma unroll
  for (int g = 1; g < C_GROUPS; ++g) t += part[g][c];
  return t;
}

// acc (32 x 32) += sum_k a(row r, k) * b(k, col r); K rounded up to even, accessors return 0 past the end.
// The operands come straight from global memory (L1 / L2 hits): the loads of 32 k are issued together before their 16 MFMAs
// (eight k at a time left 16 dependent round trips per 128-deep product: contr_pair_bwd 40 us for 4 us of matrix work).
template <typename FA, typename FB>
__device__ __forceinline__ void tile32(f32x16& acc, int K, FA a, FB b, int lane) {
  const int r = lane & 31, hf = lane >> 5;
  for (int k0 = 0; k0 < K; k0 += 32) {
    float av[16], bv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int k = k0 + 2 * u + hf;
      av[u] = (k < K) ? a(r, k) : 0.0f;
      bv[u] = (k < K) ? b(k, r) : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
  }
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int t = 0; t < 16; ++t) z[t] = 0.0f;
  return z;
}


// C[i][c] = sum_k A(i, k) * W(k, c) for i < B (<= 256), c < N (<= 128), all 16 waves of the workgroup: 32-wide K slabs of both
// operands are staged through LDS with coalesced loads (WKC: W(k, c) is contiguous in k, else in c), each wave owns the
// 32 x 32 output tiles tl = wave, wave + 16, ... (at most two).  Same k order and MFMA mapping as tile32: same bits.
constexpr int SLAB = 32, SLD = SLAB + 1, SLAB_ROWS = 256, SLAB_COLS = 128;
struct SlabSmem { float a[SLAB_ROWS * SLD]; float w[SLAB_COLS * SLD]; };

template <bool WKC, int NW = C_WAVES, typename FA, typename FW, typename FOUT>
__device__ __forceinline__ void gemm_slabs(SlabSmem& sm, int B, int N, int K, FA a_elem, FW w_elem, FOUT out, int tid) {
  const int lane = tid & 63, wave = tid >> 6, r = lane & 31, hf = lane >> 5;
  const int row_tiles = (B + 31) / 32, col_tiles = (N + 31) / 32, ntiles = row_tiles * col_tiles;
  f32x16 acc[2] = {zero16(), zero16()};
  for (int k0 = 0; k0 < K; k0 += SLAB) {
    __syncthreads();
    // A slab: rows x 32 k (k fastest: 128-byte row segments)
    for (int idx = tid; idx < row_tiles * 32 * SLAB; idx += 64 * NW) {
      const int i = idx / SLAB, kk = idx % SLAB;
      sm.a[i * SLD + kk] = (i < B && k0 + kk < K) ? a_elem(i, k0 + kk) : 0.0f;
    }
    for (int idx = tid; idx < col_tiles * 32 * SLAB; idx += 64 * NW) {
      int c, kk;
      if (WKC) { c = idx / SLAB; kk = idx % SLAB; } else { kk = idx / (col_tiles * 32); c = idx % (col_tiles * 32); }
      sm.w[c * SLD + kk] = (c < N && k0 + kk < K) ? w_elem(k0 + kk, c) : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int own = 0; own < 2; ++own) {
      const int tl = wave + own * NW;
      if (tl < ntiles) {
        const float* ap = sm.a + ((tl / col_tiles) * 32 + r) * SLD;
        const float* wp = sm.w + ((tl % col_tiles) * 32 + r) * SLD;
#pragma unroll
        for (int kk = 0; kk < SLAB; kk += 2)
          acc[own] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk + hf], wp[kk + hf], acc[own], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int own = 0; own < 2; ++own) {
    const int tl = wave + own * NW;
    if (tl < ntiles) {
      const int i0 = (tl / col_tiles) * 32, j0 = (tl % col_tiles) * 32;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int i = i0 + tile_row(t, hf), c = j0 + r;
        if (i < B && c < N) out(i, c, acc[own][t]);
      }
    }
  }
  __syncthreads();
}

// The same product for ONE 32-column block of the output (N <= 32), four waves: every thread issues ALL of a slab's loads (32 of
// the A slab, DEPTH / 8 of the W slab) before its first LDS store -- one memory round trip per K slab.  (Staged element by
// element, as gemm_slabs does with its 16 waves, four waves paid 16 dependent round trips per slab: 35 us per launch.)  The slab is
// DEPTH = 32 k deep for up to 256 rows and 64 deep for up to 128 rows (same LDS, half the round trips).  Wave w owns the row
// tiles w and w + 4.  Same k order and MFMA mapping as gemm_slabs: same bits.
template <bool WKC, int DEPTH, typename FA, typename FW, typename FOUT>
__device__ __forceinline__ void gemm_block_d(SlabSmem& sm, int B, int N, int K, FA a_elem, FW w_elem, FOUT out, int tid) {
  constexpr int NT = 256, LDD = DEPTH + 1, ROWS = SLAB_ROWS * SLAB / DEPTH, PA = ROWS * DEPTH / NT, PW = 32 * DEPTH / NT;
  static_assert(ROWS * LDD <= SLAB_ROWS * SLD && 32 * LDD <= SLAB_COLS * SLD, "the deeper slab must fit the same LDS");
  const int lane = tid & 63, wave = tid >> 6, r = lane & 31, hf = lane >> 5;
  const int row_tiles = (B + 31) / 32;
  f32x16 acc[2] = {zero16(), zero16()};
  for (int k0 = 0; k0 < K; k0 += DEPTH) {
    float av[PA], wv[PW];
#pragma unroll
    for (int u = 0; u < PA; ++u) {
      const int idx = tid + u * NT, i = idx / DEPTH, kk = idx % DEPTH;
      av[u] = (i < B && k0 + kk < K) ? a_elem(i, k0 + kk) : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < PW; ++u) {
      const int idx = tid + u * NT;
      const int c = WKC ? idx / DEPTH : idx % 32, kk = WKC ? idx % DEPTH : idx / 32;
      wv[u] = (c < N && k0 + kk < K) ? w_elem(k0 + kk, c) : 0.0f;
    }
    __syncthreads();      // the previous slab's products are done
#pragma unroll
    for (int u = 0; u < PA; ++u) {
      const int idx = tid + u * NT, i = idx / DEPTH, kk = idx % DEPTH;
      if (i < row_tiles * 32) sm.a[i * LDD + kk] = av[u];
    }
#pragma unroll
    for (int u = 0; u < PW; ++u) {
      const int idx = tid + u * NT;
      const int c = WKC ? idx / DEPTH : idx % 32, kk = WKC ? idx % DEPTH : idx / 32;
      sm.w[c * LDD + kk] = wv[u];
    }
    __syncthreads();
#pragma unroll
    for (int own = 0; own < 2; ++own) {
      const int tl = wave + own * 4;
      if (tl < row_tiles) {
        const float* ap = sm.a + (tl * 32 + r) * LDD;
        const float* wp = sm.w + r * LDD;
#pragma unroll
        for (int kk = 0; kk < DEPTH; kk += 2)
          acc[own] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk + hf], wp[kk + hf], acc[own], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int own = 0; own < 2; ++own) {
    const int tl = wave + own * 4;
    if (tl < row_tiles) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int i = tl * 32 + tile_row(t, hf);
        if (i < B && r < N) out(i, r, acc[own][t]);
      }
    }
  }
  __syncthreads();
}

template <bool WKC, typename FA, typename FW, typename FOUT>
__device__ __forceinline__ void gemm_block(SlabSmem& sm, int B, int N, int K, FA a_elem, FW w_elem, FOUT out, int tid) {
  if (B <= 128) gemm_block_d<WKC, 64>(sm, B, N, K, a_elem, w_elem, out, tid);      // (kernel-uniform)
  else gemm_block_d<WKC, 32>(sm, B, N, K, a_elem, w_elem, out, tid);
}

// scratch layout per side (floats): Y [B][Z] | A1 [B][Z] | Zc [B][Z] | stats: mu[Z] inv[Z] colmean[Z] std[Z]
__host__ __device__ inline long long side_floats(int B) { return 3LL * B * CZ + 4 * CZ; }

constexpr int SB_COLS = 32;                 // columns per side-pass workgroup
constexpr int SB_BLOCKS = CZ / SB_COLS;     // 4
constexpr int SB_WAVES = 4;
constexpr int SB_GROUPS = C_GROUPS;         // row stripes of a column reduction: the 16-wave kernel's 8 (same sums, same order)
static_assert(SB_GROUPS * SB_COLS == 64 * SB_WAVES, "one thread per (row stripe, column)");

__device__ __forceinline__ float block_group_sum(const float (*part)[SB_COLS], int c) {
  float t = part[0][c];
#pragma unroll
  for (int g = 1; g < SB_GROUPS; ++g) t += part[g][c];
  return t;
}

// grid (SB_BLOCKS, 2 sides): columns [32 blk, 32 blk + 32) of y1 = emb W1^T, their batch statistics (biased variance),
// xhat and a1 = relu(gamma xhat + beta)
constexpr int BLD = SB_COLS + 1;      // the block's [B][32] output tile stays in LDS between the product and the column passes

__global__ __launch_bounds__(64 * SB_WAVES) void contr_side_fwd_a_kernel(
    const float* __restrict__ emb_c, const float* __restrict__ emb_w, int ld_e, int E,
    const float* __restrict__ W1, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ scratch, int B) {
  const int side = blockIdx.y, c0 = blockIdx.x * SB_COLS;
  const float* emb = side == 0 ? emb_c : emb_w;
  float* S = scratch + (size_t)side * side_floats(B);
  float* Y = S; float* A1 = Y + (size_t)B * CZ; float* st = A1 + 2 * (size_t)B * CZ;
  __shared__ float red[SB_COLS], rinv[SB_COLS];
  __shared__ SlabSmem slab;
  __shared__ float blk[SLAB_ROWS * BLD];
  __shared__ float part[SB_GROUPS][SB_COLS];
  const int tid = threadIdx.x;
  const int col = tid & (SB_COLS - 1), grp = tid / SB_COLS;
  gemm_block<true>(slab, B, SB_COLS, E, [&](int i, int k) { return emb[(size_t)i * ld_e + k]; },
                   [&](int k, int c) { return W1[(size_t)(c0 + c) * E + k]; },
                   [&](int i, int c, float v) { blk[i * BLD + c] = v; }, tid);
  {
    float s = 0.0f;
    for (int b = grp; b < B; b += SB_GROUPS) s += blk[b * BLD + col];
    part[grp][col] = s;
    __syncthreads();
    if (tid < SB_COLS) red[tid] = block_group_sum(part, tid) / (float)B;
    __syncthreads();
    const float mu = red[col];
    float v = 0.0f;
    for (int b = grp; b < B; b += SB_GROUPS) { const float d = blk[b * BLD + col] - mu; v += d * d; }
    part[grp][col] = v;
    __syncthreads();
    if (tid < SB_COLS) {
      const float inv = 1.0f / sqrtf(block_group_sum(part, tid) / (float)B + BN_EPS);
      st[c0 + tid] = red[tid];
      st[CZ + c0 + tid] = inv;
      rinv[tid] = inv;
    }
  }
  __syncthreads();
  const float gm = gamma[c0 + col], bt = beta[c0 + col];
  for (int b = grp; b < B; b += SB_GROUPS) {
    const size_t at = (size_t)b * CZ + c0 + col;
    const float xh = (blk[b * BLD + col] - red[col]) * rinv[col];
    Y[at] = xh;                                      // Y holds xhat (needed by the backward)
    A1[at] = fmaxf(gm * xh + bt, 0.0f);
  }
}

// grid (SB_BLOCKS, 2 sides): columns [32 blk, 32 blk + 32) of z0 = a1 W2^T, centred over the batch; their unbiased std
// (the hinge term is formed from the stds by contr_finish, in column order)
__global__ __launch_bounds__(64 * SB_WAVES) void contr_side_fwd_b_kernel(const float* __restrict__ W2, float* __restrict__ scratch, int B) {
  const int side = blockIdx.y, c0 = blockIdx.x * SB_COLS;
  float* S = scratch + (size_t)side * side_floats(B);
  const float* A1 = S + (size_t)B * CZ; float* Zc = S + 2 * (size_t)B * CZ; float* st = Zc + (size_t)B * CZ;
  __shared__ float red[SB_COLS];
  __shared__ SlabSmem slab;
  __shared__ float blk[SLAB_ROWS * BLD];
  __shared__ float part[SB_GROUPS][SB_COLS];
  const int tid = threadIdx.x;
  const int col = tid & (SB_COLS - 1), grp = tid / SB_COLS;
  gemm_block<true>(slab, B, SB_COLS, CZ, [&](int i, int k) { return A1[(size_t)i * CZ + k]; },
                   [&](int k, int c) { return W2[(size_t)(c0 + c) * CZ + k]; },
                   [&](int i, int c, float v) { blk[i * BLD + c] = v; }, tid);
  float s = 0.0f;
  for (int b = grp; b < B; b += SB_GROUPS) s += blk[b * BLD + col];
  part[grp][col] = s;
  __syncthreads();
  if (tid < SB_COLS) red[tid] = block_group_sum(part, tid) / (float)B;
  __syncthreads();
  const float m = red[col];
  float v = 0.0f;
  for (int b = grp; b < B; b += SB_GROUPS) {
    const float d = blk[b * BLD + col] - m;
    Zc[(size_t)b * CZ + c0 + col] = d;
    v += d * d;
  }
  part[grp][col] = v;
  __syncthreads();
  if (tid < SB_COLS) {
    st[2 * CZ + c0 + tid] = red[tid];
    st[3 * CZ + c0 + tid] = sqrtf(block_group_sum(part, tid) / (float)(B - 1) + 1e-4f);
  }
}

// tiles: [0, rt*rt) pair tiles (B x B), then 16 corr tiles (Z x Z).  PAIR [B][B], CORR [Z][Z] are kept for the backward.
__global__ __launch_bounds__(256) void contr_pair_fwd_kernel(
    const float* __restrict__ scratch, const float* __restrict__ pos, float lambda, float* __restrict__ PAIR,
    float* __restrict__ CORR, float* __restrict__ partials, int B) {
  const float* Zc = scratch + 2LL * B * CZ;
  const float* Zw = scratch + side_floats(B) + 2LL * B * CZ;
  __shared__ float red[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hf = lane >> 5;
  const int rt = (B + 31) / 32, npair = rt * rt, ntiles = npair + 16;
  const int tl = blockIdx.x * 4 + wave;
  float part = 0.0f;
  if (tl < ntiles) {
    f32x16 acc = zero16();
    if (tl < npair) {
      const int i0 = (tl / rt) * 32, j0 = (tl % rt) * 32;
      tile32(acc, CZ, [&](int rr, int k) { return (i0 + rr < B) ? Zc[(size_t)(i0 + rr) * CZ + k] : 0.0f; },
             [&](int k, int cc) { return (j0 + cc < B) ? Zw[(size_t)(j0 + cc) * CZ + k] : 0.0f; }, lane);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int i = i0 + tile_row(t, hf), j = j0 + r;
        if (i < B && j < B) {
          const float v = acc[t] * (1.0f / (float)CZ);
          PAIR[(size_t)i * B + j] = v;
          const float d = (i == j) ? v - pos[i] : v;
          part += ((i == j) ? 1.0f : lambda) * d * d;
        }
      }
    } else {
      const int c = tl - npair, i0 = (c / 4) * 32, j0 = (c % 4) * 32;
      tile32(acc, B, [&](int rr, int k) { return Zc[(size_t)k * CZ + i0 + rr]; },
             [&](int k, int cc) { return Zw[(size_t)k * CZ + j0 + cc]; }, lane);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int i = i0 + tile_row(t, hf), j = j0 + r;
        const float v = acc[t] / (float)B;
        CORR[(size_t)i * CZ + j] = v;
        const float d = (i == j) ? v - 1.0f : v;
        part += ((i == j) ? 1.0f : lambda) * d * d;
      }
    }
  }
  red[wave][lane] = part;
  __syncthreads();
  if (tid < 4) {
    float s = 0.0f;
    for (int l = 0; l < 64; ++l) s += red[tid][l];
    partials[blockIdx.x * 4 + tid] = s;
  }
}

// loss = scale * gate * (pair / correlation terms + hinge): `scale` is the caller's loss coefficient, `gate` (device, may be NULL)
// the reference's early-out as a 0 / 1 factor -- both folded in here instead of two multiply launches behind the loss
// (hinge term of a side = mean_k relu(1 - std_k), summed in column order from the stds the side pass left in `scratch`)
__global__ __launch_bounds__(256) void contr_finish_kernel(const float* __restrict__ partials, int nparts, const float* __restrict__ scratch, int B,
                                                           float* __restrict__ loss, const float* __restrict__ gate, float scale) {
  // (all loads in parallel, the sums by one thread in index order from LDS: one thread walking 8 + 256 global values itself
  //  took 8 us)
  __shared__ float hs[2][CZ];
  __shared__ float ps[256];
  const int tid = threadIdx.x;
  {
    const int side = tid / CZ, k = tid % CZ;
    hs[side][k] = fmaxf(1.0f - scratch[(size_t)side * side_floats(B) + 3 * (size_t)B * CZ + 3 * CZ + k], 0.0f);
  }
  for (int i = tid; i < nparts; i += 256) ps[i] = partials[i];      // nparts <= (8 * 8 + 16 + 3) / 4 * 4 = 84
  __syncthreads();
  if (tid == 0) {
    float s = 0.0f;
    for (int i = 0; i < nparts; ++i) s += ps[i];
    float hinge[2];
    for (int side = 0; side < 2; ++side) {
      float h = 0.0f;
      for (int k = 0; k < CZ; ++k) h += hs[side][k];
      hinge[side] = h / (float)CZ;
    }
    loss[0] = (s + 0.5f * (hinge[0] + hinge[1])) * (gate != nullptr ? gate[0] * scale : scale);
  }
}

__global__ __launch_bounds__(256) void contr_targets_kernel(const float* __restrict__ target, float* __restrict__ pos,
                                                            float* __restrict__ gate, int B) {
  __shared__ float s_sum[256], s_lo[256], s_hi[256];
  __shared__ int s_other[256];
  const int tid = threadIdx.x;
  float sum = 0.0f, lo = INFINITY, hi = -INFINITY;
  for (int b = tid; b < B; b += 256) { const float v = target[b]; sum += v; lo = fminf(lo, v); hi = fmaxf(hi, v); }
  s_sum[tid] = sum; s_lo[tid] = lo; s_hi[tid] = hi;
  __syncthreads();
  if (tid == 0) {          // fixed order: deterministic mean
    float t = 0.0f, l = INFINITY, h = -INFINITY;
    for (int i = 0; i < 256; ++i) { t += s_sum[i]; l = fminf(l, s_lo[i]); h = fmaxf(h, s_hi[i]); }
    s_sum[0] = t / (float)B; s_lo[0] = l; s_hi[0] = h;
  }
  __syncthreads();
  const float mean = s_sum[0], l = s_lo[0], h = s_hi[0];
  int other = 0;
  for (int b = tid; b < B; b += 256) {
    const float v = target[b];
    pos[b] = v > mean ? 1.0f : 0.0f;
    other |= (v != l && v != h) ? 1 : 0;
  }
  s_other[tid] = other;
  __syncthreads();
  if (tid == 0) {
    int any = 0;
    for (int i = 0; i < 256; ++i) any |= s_other[i];
    gate[0] = (any == 0 && l != h) ? 1.0f : 0.0f;
  }
}

// DZ [2][B][Z]: gradient w.r.t. the centred projections (pair / corr terms only; the hinge is added by the side kernel)
__global__ __launch_bounds__(256) void contr_pair_bwd_kernel(
    const float* __restrict__ scratch, const float* __restrict__ pos, float lambda, const float* __restrict__ PAIR,
    const float* __restrict__ CORR, float* __restrict__ DZ, int B) {
  const float* Zc = scratch + 2LL * B * CZ;
  const float* Zw = scratch + side_floats(B) + 2LL * B * CZ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hf = lane >> 5;
  const int rt = (B + 31) / 32, per_side = rt * 4;
  const int tl = blockIdx.x * 4 + wave;
  if (tl >= 2 * per_side) return;
  const int side = tl / per_side, q = tl % per_side;
  const int i0 = (q / 4) * 32, j0 = (q % 4) * 32;
  const float invz = 1.0f / (float)CZ, invb = 1.0f / (float)B;
  auto dpair = [&](int i, int j) {          // d loss / d pair_ij
    if (i >= B || j >= B) return 0.0f;
    const float v = PAIR[(size_t)i * B + j];
    return (i == j) ? 2.0f * (v - pos[i]) : 2.0f * lambda * v;
  };
  auto dcorr = [&](int k, int l) {
    const float v = CORR[(size_t)k * CZ + l];
    return (k == l) ? 2.0f * (v - 1.0f) : 2.0f * lambda * v;
  };
  f32x16 acc = zero16(), acc2 = zero16();
  if (side == 0) {
    // dzc[b][k] = sum_j dpair[b][j] zw[j][k] / Z + sum_l zw[b][l] dcorr[k][l] / B
    tile32(acc, B, [&](int rr, int j) { return dpair(i0 + rr, j); },
           [&](int j, int cc) { return Zw[(size_t)j * CZ + j0 + cc]; }, lane);
    tile32(acc2, CZ, [&](int rr, int l) { return (i0 + rr < B) ? Zw[(size_t)(i0 + rr) * CZ + l] : 0.0f; },
           [&](int l, int cc) { return dcorr(j0 + cc, l); }, lane);
  } else {
    // dzw[j][l] = sum_b dpair[b][j] zc[b][l] / Z + sum_k zc[j][k] dcorr[k][l] / B
    tile32(acc, B, [&](int rr, int b) { return dpair(b, i0 + rr); },
           [&](int b, int cc) { return Zc[(size_t)b * CZ + j0 + cc]; }, lane);
    tile32(acc2, CZ, [&](int rr, int k) { return (i0 + rr < B) ? Zc[(size_t)(i0 + rr) * CZ + k] : 0.0f; },
           [&](int k, int cc) { return dcorr(k, j0 + cc); }, lane);
  }
  float* out = DZ + (size_t)side * B * CZ;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int i = i0 + tile_row(t, hf);
    if (i < B) out[(size_t)i * CZ + j0 + r] = acc[t] * invz + acc2[t] * invb;
  }
}

// grid (SB_BLOCKS, 2 sides): hinge gradient + centring backward of ALL columns of dz0 (the contraction below needs them all:
// every workgroup forms the 128 column means for itself, the values go into the product's A operand on the fly), then columns
// [32 blk, 32 blk + 32) of da1 = dz0 W2 and their ReLU / BatchNorm backward (batch statistics) -> dy1 (`work`)
__global__ __launch_bounds__(64 * SB_WAVES) void contr_side_bwd_a_kernel(
    const float* __restrict__ gamma, const float* __restrict__ W2, const float* __restrict__ scratch,
    const float* __restrict__ DZ, float* __restrict__ work, int B) {
  const int side = blockIdx.y, c0 = blockIdx.x * SB_COLS;
  const float* S = scratch + (size_t)side * side_floats(B);
  const float* XH = S; const float* A1 = XH + (size_t)B * CZ; const float* Zc = A1 + (size_t)B * CZ;
  const float* st = Zc + (size_t)B * CZ;
  const float* dz = DZ + (size_t)side * B * CZ;
  float* wk = work + (size_t)side * B * CZ;                  // dy1
  __shared__ float coef_s[CZ], mean_s[CZ];
  __shared__ float s1[SB_COLS], s2[SB_COLS];
  __shared__ SlabSmem slab;
  __shared__ float blk[SLAB_ROWS * BLD];
  __shared__ float part[SB_GROUPS][SB_COLS], part2[SB_GROUPS][SB_COLS];
  const int tid = threadIdx.x;
  const int col = tid & (SB_COLS - 1), grp = tid / SB_COLS;
  // ---- hinge gradient + centring backward, per column, all 128 columns: four rounds of 32 columns (a thread's rows of a round
  //      are loaded together; the 16-wave kernel's row stripes and combine order), their partial sums parked in LDS, ONE barrier
  //      pair for all four rounds ----
  constexpr int RMAX = SLAB_ROWS / SB_GROUPS;      // rows of a stripe (B <= 256)
  float (*part4)[SB_GROUPS][SB_COLS] = reinterpret_cast<float (*)[SB_GROUPS][SB_COLS]>(blk);      // [4][8][32]: the block tile is free
  float coef_r[CZ / SB_COLS];
#pragma unroll
  for (int cbi = 0; cbi < CZ / SB_COLS; ++cbi) {
    const int k = cbi * SB_COLS + col;
    const float sd = st[3 * CZ + k];
    // d hinge / d var_k = -(1/2) * (1/Z) * [sd < 1] / (2 sd);  d var_k / d z_bk = 2 z_bk / (B - 1)
    const float dv = (sd < 1.0f) ? -0.5f / (float)CZ / (2.0f * sd) : 0.0f;
    const float coef = dv * 2.0f / (float)(B - 1);
    coef_r[cbi] = coef;
    float dzv[RMAX], zcv[RMAX];
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
      const int b = grp + u * SB_GROUPS;
      dzv[u] = (b < B) ? dz[(size_t)b * CZ + k] : 0.0f;
      zcv[u] = (b < B) ? Zc[(size_t)b * CZ + k] : 0.0f;
    }
    float s = 0.0f;
#pragma unroll
    for (int u = 0; u < RMAX; ++u)
      if (grp + u * SB_GROUPS < B) s += dzv[u] + coef * zcv[u];
    part4[cbi][grp][col] = s;
  }
  __syncthreads();
  if (tid < CZ) {
    const int cbi = tid / SB_COLS, c = tid % SB_COLS;
    float t = part4[cbi][0][c];
#pragma unroll
    for (int g = 1; g < SB_GROUPS; ++g) t += part4[cbi][g][c];
    mean_s[tid] = t / (float)B;
  }
  if (grp == 0) {
#pragma unroll
    for (int cbi = 0; cbi < CZ / SB_COLS; ++cbi) coef_s[cbi * SB_COLS + col] = coef_r[cbi];
  }
  __syncthreads();
  // ---- da1 = dz0 W2 (this block's columns); A operand = the centred gradient ----
  gemm_block<false>(slab, B, SB_COLS, CZ,
                    [&](int i, int k) { return (dz[(size_t)i * CZ + k] + coef_s[k] * Zc[(size_t)i * CZ + k]) - mean_s[k]; },
                    [&](int k, int c) { return W2[(size_t)k * CZ + c0 + c]; },
                    [&](int i, int c, float v) { blk[i * BLD + c] = v; }, tid);
  // ---- ReLU backward, BatchNorm backward on batch statistics ----
  const float gm = gamma[c0 + col];
  float a1v[RMAX], xhv[RMAX];
#pragma unroll
  for (int u = 0; u < RMAX; ++u) {
    const int b = grp + u * SB_GROUPS;
    a1v[u] = (b < B) ? A1[(size_t)b * CZ + c0 + col] : 0.0f;
    xhv[u] = (b < B) ? XH[(size_t)b * CZ + c0 + col] : 0.0f;
  }
  {
    float a = 0.0f, c = 0.0f;
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
      const int b = grp + u * SB_GROUPS;
      if (b < B) {
        const float dxh = (a1v[u] > 0.0f) ? blk[b * BLD + col] * gm : 0.0f;
        blk[b * BLD + col] = dxh;
        a += dxh;
        c += dxh * xhv[u];
      }
    }
    part[grp][col] = a; part2[grp][col] = c;
    __syncthreads();
    if (tid < SB_COLS) { s1[tid] = block_group_sum(part, tid); s2[tid] = block_group_sum(part2, tid); }
  }
  __syncthreads();
  const float scale = st[CZ + c0 + col] / (float)B;
#pragma unroll
  for (int u = 0; u < RMAX; ++u) {
    const int b = grp + u * SB_GROUPS;
    if (b < B) wk[(size_t)b * CZ + c0 + col] = scale * ((float)B * blk[b * BLD + col] - s1[col] - xhv[u] * s2[col]);     // dy1
  }
}

// grid (ceil(E / 32), 2 sides): columns [32 blk, ...) of d emb = dy1 W1, scaled by the upstream gradient
__global__ __launch_bounds__(64 * SB_WAVES) void contr_side_bwd_b_kernel(
    const float* __restrict__ W1, const float* __restrict__ work, const float* __restrict__ g_loss,
    const float* __restrict__ gate, float scale, float* __restrict__ demb_c, float* __restrict__ demb_w, int ld_d, int E, int B) {
  const int side = blockIdx.y, c0 = blockIdx.x * SB_COLS;
  const float* wk = work + (size_t)side * B * CZ;
  float* demb = side == 0 ? demb_c : demb_w;
  __shared__ SlabSmem slab;
  const int tid = threadIdx.x;
  const float g = g_loss[0] * (gate != nullptr ? gate[0] * scale : scale);
  gemm_block<false>(slab, B, min(SB_COLS, E - c0), CZ, [&](int i, int k) { return wk[(size_t)i * CZ + k]; },
                    [&](int k, int c) { return W1[(size_t)k * E + c0 + c]; },
                    [&](int i, int c, float v) { demb[(size_t)i * ld_d + c0 + c] = v * g; }, tid);
}

}  // namespace is

// floats of `scratch` (kept from forward to backward): two sides + PAIR [B][B] + CORR [Z][Z] + hinge[2] + partials
extern "C" long long is_contrastive_scratch_floats(int B) {
  const long long rt = (B + 31) / 32;
  return 2 * is::side_floats(B) + (long long)B * B + is::CZ * is::CZ + 2 + (rt * rt + 16 + 3) / 4 * 4;
}
// floats of the backward work buffer (DZ [2][B][Z] + work [2][B][Z])
extern "C" long long is_contrastive_work_floats(int B) { return 4LL * B * is::CZ; }

// emb_c, emb_w [B, ld_e] (E valid columns), pos [B] (1.0 = immunogenic), W1 [128, E], gamma, beta [128], W2 [128, 128];
// loss [1] = scale * gate * L (gate [1] on the device or NULL: 1).  2 <= B <= 256, E <= 256.
extern "C" int is_contrastive_fwd(const float* emb_c, const float* emb_w, int ld_e, int E, const float* pos, const float* W1,
                                  const float* gamma, const float* beta, const float* W2, float lambda, float* scratch,
                                  float* loss, const float* gate, float scale, int B, void* stream) {
  if (B < 2 || B > 256 || E <= 0 || E > 256 || ld_e < E) return is::fail(__func__, -22);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long sides = 2 * is::side_floats(B);
  float* PAIR = scratch + sides;
  float* CORR = PAIR + (long long)B * B;
  float* partials = CORR + is::CZ * is::CZ + 2;
  const int rt = (B + 31) / 32, ntiles = rt * rt + 16, nblocks = (ntiles + 3) / 4;
  const dim3 sgrid(is::SB_BLOCKS, 2), sblock(64 * is::SB_WAVES);
  hipLaunchKernelGGL(is::contr_side_fwd_a_kernel, sgrid, sblock, 0, st, emb_c, emb_w, ld_e, E, W1, gamma, beta, scratch, B);
  hipLaunchKernelGGL(is::contr_side_fwd_b_kernel, sgrid, sblock, 0, st, W2, scratch, B);
  hipLaunchKernelGGL(is::contr_pair_fwd_kernel, dim3(nblocks), dim3(256), 0, st, scratch, pos, lambda, PAIR, CORR, partials, B);
  hipLaunchKernelGGL(is::contr_finish_kernel, dim3(1), dim3(256), 0, st, partials, nblocks * 4, scratch, B, loss, gate, scale);
  return is::launch_status(__func__);
}

// target [B] -> pos [B] = (target > mean(target)) as 1.0 / 0.0 (reference utils/contrastive.py:45) and gate [1] = 1.0 if the
// target holds exactly two distinct values else 0.0 (the reference's early-out, :38-43, as a device-side factor).  One
// launch instead of ~11 elementwise / reduction launches.  1 <= B <= 1024.
extern "C" int is_contrastive_targets(const float* target, float* pos, float* gate, int B, void* stream) {
  if (B < 1 || B > 1024) return is::fail(__func__, -22);
  hipLaunchKernelGGL(is::contr_targets_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), target, pos, gate, B);
  return is::launch_status(__func__);
}

// g_loss [1] = upstream gradient of the scalar loss (multiplied by scale * gate as in the forward); demb_c, demb_w [B, ld_d]
// (E columns written).
extern "C" int is_contrastive_bwd(const float* pos, const float* W1, const float* gamma, const float* W2, float lambda,
                                  const float* scratch, float* work, const float* g_loss, const float* gate, float scale,
                                  float* demb_c, float* demb_w, int ld_d, int E, int B, void* stream) {
  if (B < 2 || B > 256 || E <= 0 || E > 256 || ld_d < E) return is::fail(__func__, -22);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long sides = 2 * is::side_floats(B);
  const float* PAIR = scratch + sides;
  const float* CORR = PAIR + (long long)B * B;
  float* DZ = work;
  float* wk = work + 2LL * B * is::CZ;
  const int rt = (B + 31) / 32, ntiles = 2 * rt * 4;
  hipLaunchKernelGGL(is::contr_pair_bwd_kernel, dim3((ntiles + 3) / 4), dim3(256), 0, st, scratch, pos, lambda, PAIR, CORR, DZ, B);
  const dim3 sblock(64 * is::SB_WAVES);
  hipLaunchKernelGGL(is::contr_side_bwd_a_kernel, dim3(is::SB_BLOCKS, 2), sblock, 0, st, gamma, W2, scratch, DZ, wk, B);
  hipLaunchKernelGGL(is::contr_side_bwd_b_kernel, dim3((E + is::SB_COLS - 1) / is::SB_COLS, 2), sblock, 0, st, W1, wk, g_loss, gate,
                     scale, demb_c, demb_w, ld_d, E, B);
  return is::launch_status(__func__);
}
