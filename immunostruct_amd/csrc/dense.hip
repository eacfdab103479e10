// Weight gradient of a Linear layer with a SMALL batch (the contraction dimension): dW [N, K] = gy^T x, db [N] = sum_b gy,
// for gy [B, N], x [B, K], B <= a few hundred.  These are the two large matrices of the sequence VAE
// (vae_fc1: 512 x 5943, vae_fc4: 5943 x 512; models/hybrid_models.py:297-308) whose weight gradients hipBLASLt runs
// with 86 us / 31 us kernels at B = 128 (K = 128 is far from its tuned shapes); a 64 x 64 output tile per workgroup with
// the two operand panels staged in LDS in one go does the same in ~12 us, and the bias gradient (another 13-60 us torch
// reduction over the batch) is a by-product.  Forward and input gradient stay on the library GEMMs.
// Fixed summation order over the batch -> bitwise reproducible.
#include "common.h"

namespace is {

constexpr int WG_B = 64;       // batch rows staged per pass

// One workgroup owns a 64 (n) x 64 (k) tile of dW.  The two operand panels of a pass (64 batch rows x 64 columns each) are
// fetched with all loads in flight at once -- lane = column, one coalesced 256-byte row segment per load, 32 loads per lane --
// and staged in LDS; wave (mt, nt) then accumulates its 32 x 32 quadrant over the staged rows on v_mfma_f32_32x32x2_f32
// (fixed order over the batch).  The NEXT pass' loads are issued before the MFMAs of the current one
// (software pipeline), and with 35 KB of LDS four workgroups share a CU: the 744 workgroups of a VAE matrix are resident at
// once and overlap each other's load latency (the one-pass, two-per-CU form of round 1 ran them in 1.5 latency-bound rounds).
__global__ __launch_bounds__(256, 4) void linear_wgrad_kernel(const float* __restrict__ gy, int ld_g, const float* __restrict__ x,
                                                              int ld_x, float* __restrict__ dW, float* __restrict__ db,
                                                              int B, int N, int K) {
  __shared__ float gs[WG_B * LD];   // gy panel: [64 batch rows][64 output columns n]
  __shared__ float xs[WG_B * LD];   // x  panel: [64 batch rows][64 input columns k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
  const int mt = wave >> 1, nt = wave & 1;            // this wave's 32 x 32 quadrant of the 64 x 64 tile
  const int r = lane & 31, hf = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = 0.0f;
  float colsum = 0.0f;                                 // lane = column n0 + lane (wave 0 of the k0 = 0 column of tiles): bias gradient
  const bool n_ok = n0 + lane < N, k_ok = k0 + lane < K;
  const int nc = min(n0 + lane, N - 1), kc = min(k0 + lane, K - 1);      // clamped: every load is unconditional
  constexpr int RPW = WG_B / 4;                        // rows staged per wave
  float gv[RPW], xv[RPW];
  auto fetch = [&](int b0) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int b = min(b0 + wave * RPW + i, B - 1);
      gv[i] = gy[(size_t)b * ld_g + nc];
      xv[i] = x[(size_t)b * ld_x + kc];
    }
  };
  fetch(0);
  for (int b0 = 0; b0 < B; b0 += WG_B) {
    if (b0 > 0) __syncthreads();                       // the previous pass' MFMAs are done with the panels
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int row = wave * RPW + i;
      const bool b_ok = b0 + row < B;
      gs[row * LD + lane] = (b_ok && n_ok) ? gv[i] : 0.0f;
      xs[row * LD + lane] = (b_ok && k_ok) ? xv[i] : 0.0f;
    }
    __syncthreads();
    if (b0 + WG_B < B) fetch(b0 + WG_B);               // in flight under this pass' MFMAs
    // acc (32 x 32) += sum over the staged rows e of gs[e][mt*32 + i] * xs[e][nt*32 + j]; half hf walks rows [32 hf, 32 hf + 32)
#pragma unroll 8
    for (int sidx = 0; sidx < WG_B / 2; ++sidx) {
      const int e = hf * (WG_B / 2) + sidx;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gs[e * LD + mt * 32 + r], xs[e * LD + nt * 32 + r], acc, 0, 0, 0);
    }
    if (db != nullptr && blockIdx.y == 0 && wave == 0) {
#pragma unroll 8
      for (int e = 0; e < WG_B; ++e) colsum += gs[e * LD + lane];
    }
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int n = n0 + mt * 32 + tile_row(t, hf), k = k0 + nt * 32 + r;
    if (n < N && k < K) dW[(size_t)n * K + k] = acc[t];
  }
  if (db != nullptr && blockIdx.y == 0 && wave == 0 && n_ok) db[n0 + lane] = colsum;
}

// ---- input gradient of a Linear layer with a LONG contraction and a small output: gx [B, K] = gy [B, N] W [N, K] --------------
// (vae_fc4: N = 5943, K = 512, B = 128: 64 Ki outputs, each a 5943-term sum.)  The library runs this as ONE workgroup per 32 x 16
// output tile walking the whole contraction (Cijk_Ailk_Bljk MT32x16x128: 33-38 us at 128 workgroups); here the contraction is
// cut into chunks of 96 -- one workgroup per (64 batch rows, 64 output columns, chunk) = 2 x 8 x 62 workgroups of 50 KB LDS and
// 84 registers: small enough to sit BESIDE the two resident workgroups of a forward layer kernel (52 KB and 128 registers
// per lane are free there) instead of waiting for one to leave, or blocking one that wants to start -- each
// staging its gy chunk [64, 96] and W chunk [96, 64] in LDS in one go and writing a [64, 64] partial on
// v_mfma_f32_32x32x2_f32; a second launch sums the partials in chunk order (fixed order -> bitwise reproducible).
constexpr int DG_KC = 96;             // contraction rows per chunk
constexpr int DG_LDA = DG_KC + 2;     // gy panel pitch: the A operand reads a column of 32 rows, 2 apart -> all 64 banks
constexpr int DG_M = 64, DG_N = 64;      // 64 x 64 partial tile per workgroup: one 32 x 32 MFMA tile per wave

__global__ __launch_bounds__(256, 2) void linear_dgrad_splitk_kernel(const float* __restrict__ gy, int ld_g,
                                                                     const float* __restrict__ W, int ld_w,
                                                                     float* __restrict__ part, int B, int N, int K) {
  __shared__ float gs[DG_M * DG_LDA];       // [DG_M batch rows][DG_KC contraction columns]
  __shared__ float ws[DG_KC * DG_N];        // [DG_KC contraction rows][64 output columns], odd rows rotated by 32 columns
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k0 = blockIdx.x * DG_N, c0 = blockIdx.y * DG_KC, b0 = blockIdx.z * DG_M;
  const int r = lane & 31, hf = lane >> 5;
  {
    // gy chunk: wave w stages rows [RG w, RG w + RG), lane = column (+ 64): coalesced row segments, all loads in flight
    constexpr int CM = (DG_KC + 63) / 64, RG = DG_M / 4;
    float v[RG][CM];
#pragma unroll
    for (int i = 0; i < RG; ++i) {
      const int b = min(b0 + wave * RG + i, B - 1);
#pragma unroll
      for (int m = 0; m < CM; ++m) v[i][m] = gy[(size_t)b * ld_g + min(c0 + lane + 64 * m, N - 1)];
    }
#pragma unroll
    for (int i = 0; i < RG; ++i) {
      const bool b_ok = b0 + wave * RG + i < B;
#pragma unroll
      for (int m = 0; m < CM; ++m)
        if (lane + 64 * m < DG_KC)
          gs[(wave * RG + i) * DG_LDA + lane + 64 * m] = (b_ok && c0 + lane + 64 * m < N) ? v[i][m] : 0.0f;
    }
  }
  {
    // W chunk: wave w stages rows [RW w, RW w + RW), lane = output column
    constexpr int RW = DG_KC / 4;
    float v[RW];
    const int kc = min(k0 + lane, K - 1);
#pragma unroll
    for (int i = 0; i < RW; ++i) v[i] = W[(size_t)min(c0 + wave * RW + i, N - 1) * ld_w + kc];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int c = wave * RW + i;
      ws[c * DG_N + ((lane + 32 * (c & 1)) & 63)] = (c0 + c < N && k0 + lane < K) ? v[i] : 0.0f;
    }
  }
  __syncthreads();
  // wave w: the 32 x 32 tile (rows 32 (w >> 1), columns 32 (w & 1)); step s contracts rows 2 s, 2 s + 1 of the chunk (half hf
  // takes 2 s + hf)
  const int mt = wave >> 1, nt = wave & 1;
  f32x16 acc;
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = 0.0f;
  const float* ap = gs + (mt * 32 + r) * DG_LDA + hf;
  const float* wp = ws + hf * DG_N;
  const int j0 = (r + 32 * nt + 32 * hf) & 63;      // un-rotate: row 2 s + hf is rotated by 32 hf
#pragma unroll 8
  for (int sidx = 0; sidx < DG_KC / 2; ++sidx)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * sidx], wp[2 * sidx * DG_N + j0], acc, 0, 0, 0);
  float* out = part + ((size_t)blockIdx.y * gridDim.z + blockIdx.z) * DG_M * K;      // [chunk][batch tile][DG_M][K]
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int k = k0 + nt * 32 + r;
    if (k < K) out[(size_t)(mt * 32 + tile_row(t, hf)) * K + k] = acc[t];
  }
}

// ---- forward of a Linear layer with a LONG contraction and a small output: y [B, K] = x [B, N] W^T + b, W [K, N] as stored
// (vae_fc1: N = 5943, K = 512).  Same split of the contraction as the input gradient above; the weight chunk is [64 output
// rows, 96 contraction columns] here (rows of W are contiguous along the contraction), staged with the same pitch as the x
// chunk, so both MFMA operands are read with the conflict-free column pattern.  The library's kernel for this shape
// (Cijk_Alik_Bljk MT16x16x128) takes 22 us alone and 40-46 us beside the layer kernels.
__global__ __launch_bounds__(256, 2) void linear_fwd_splitk_kernel(const float* __restrict__ x, int ld_x,
                                                                   const float* __restrict__ W, int ld_w,
                                                                   float* __restrict__ part, int B, int N, int K) {
  __shared__ float xs[DG_M * DG_LDA];       // [DG_M batch rows][DG_KC contraction columns]
  __shared__ float ws[DG_N * DG_LDA];       // [64 output rows][DG_KC contraction columns]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k0 = blockIdx.x * DG_N, c0 = blockIdx.y * DG_KC, b0 = blockIdx.z * DG_M;
  const int r = lane & 31, hf = lane >> 5;
  constexpr int CM = (DG_KC + 63) / 64, RG = DG_M / 4;
  {
    float v[RG][CM];
#pragma unroll
    for (int i = 0; i < RG; ++i) {
      const int b = min(b0 + wave * RG + i, B - 1);
#pragma unroll
      for (int m = 0; m < CM; ++m) v[i][m] = x[(size_t)b * ld_x + min(c0 + lane + 64 * m, N - 1)];
    }
#pragma unroll
    for (int i = 0; i < RG; ++i) {
      const bool b_ok = b0 + wave * RG + i < B;
#pragma unroll
      for (int m = 0; m < CM; ++m)
        if (lane + 64 * m < DG_KC)
          xs[(wave * RG + i) * DG_LDA + lane + 64 * m] = (b_ok && c0 + lane + 64 * m < N) ? v[i][m] : 0.0f;
    }
  }
  {
    float v[16][CM];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = min(k0 + wave * 16 + i, K - 1);
#pragma unroll
      for (int m = 0; m < CM; ++m) v[i][m] = W[(size_t)k * ld_w + min(c0 + lane + 64 * m, N - 1)];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const bool k_ok = k0 + wave * 16 + i < K;
#pragma unroll
      for (int m = 0; m < CM; ++m)
        if (lane + 64 * m < DG_KC)
          ws[(wave * 16 + i) * DG_LDA + lane + 64 * m] = (k_ok && c0 + lane + 64 * m < N) ? v[i][m] : 0.0f;
    }
  }
  __syncthreads();
  const int mt = wave >> 1, nt = wave & 1;      // this wave's 32 x 32 tile
  f32x16 acc;
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = 0.0f;
  const float* ap = xs + (mt * 32 + r) * DG_LDA + hf;
  const float* wp = ws + (nt * 32 + r) * DG_LDA + hf;
#pragma unroll 8
  for (int sidx = 0; sidx < DG_KC / 2; ++sidx)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * sidx], wp[2 * sidx], acc, 0, 0, 0);
  float* out = part + ((size_t)blockIdx.y * gridDim.z + blockIdx.z) * DG_M * K;      // [chunk][batch tile][DG_M][K]
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int k = k0 + nt * 32 + r;
    if (k < K) out[(size_t)(mt * 32 + tile_row(t, hf)) * K + k] = acc[t];
  }
}

__global__ __launch_bounds__(256) void linear_dgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ gx,
                                                                  int B, int K, int chunks, int mtiles,
                                                                  const float* __restrict__ bias = nullptr) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)B * K) return;
  const int b = (int)(e / K), k = (int)(e % K);
  const float* p = part + ((size_t)(b / DG_M) * DG_M + (b % DG_M)) * K + k;
  const size_t stride = (size_t)mtiles * DG_M * K;
  float acc = 0.0f;
  int c = 0;
  for (; c + 8 <= chunks; c += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(c + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; c < chunks; ++c) acc += p[(size_t)c * stride];
  gx[e] = bias != nullptr ? acc + bias[k] : acc;
}

}  // namespace is

// gy [B, ld_g] (N valid columns), x [B, ld_x] (K valid columns) -> dW [N, K] (row-major, the nn.Linear weight layout),
// db [N] (may be NULL).
extern "C" int is_linear_wgrad(const float* gy, int ld_g, const float* x, int ld_x, float* dW, float* db, int B, int N, int K,
                               void* stream) {
  if (B <= 0 || N <= 0 || K <= 0) return is::fail(__func__, -22);
  hipLaunchKernelGGL(is::linear_wgrad_kernel, dim3((N + 63) / 64, (K + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(stream),
                     gy, ld_g, x, ld_x, dW, db, B, N, K);
  return is::launch_status(__func__);
}

// floats of the split-contraction scratch of is_linear_dgrad
extern "C" long long is_linear_dgrad_scratch_floats(int B, int N, int K) {
  const long long chunks = (N + is::DG_KC - 1) / is::DG_KC, mtiles = (B + is::DG_M - 1) / is::DG_M;
  return chunks * mtiles * is::DG_M * K;
}

// gx [B, K] = gy [B, ld_g] (N valid columns) W [N, ld_w] (K valid columns; the nn.Linear weight of a layer with K inputs and
// N outputs, as stored).  scratch: is_linear_dgrad_scratch_floats(B, N, K) floats.  Two launches.
extern "C" int is_linear_dgrad(const float* gy, int ld_g, const float* W, int ld_w, float* gx, float* scratch, int B, int N, int K,
                               void* stream) {
  if (B <= 0 || N <= 0 || K <= 0) return is::fail(__func__, -22);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int chunks = (N + is::DG_KC - 1) / is::DG_KC, mtiles = (B + is::DG_M - 1) / is::DG_M;
  hipLaunchKernelGGL(is::linear_dgrad_splitk_kernel, dim3((K + is::DG_N - 1) / is::DG_N, chunks, mtiles), dim3(256), 0, st, gy, ld_g,
                     W, ld_w, scratch, B, N, K);
  const long long total = (long long)B * K;
  hipLaunchKernelGGL(is::linear_dgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, scratch, gx, B, K,
                     chunks, mtiles);
  return is::launch_status(__func__);
}

// y [B, K] = x [B, ld_x] (N valid columns) W^T + bias, W [K, ld_w] (N valid columns; the nn.Linear weight as stored), bias [K]
// or NULL.  scratch: is_linear_dgrad_scratch_floats(B, N, K) floats (the same split of the contraction).  Two launches.
extern "C" int is_linear_fwd_long(const float* x, int ld_x, const float* W, int ld_w, const float* bias, float* y, float* scratch,
                                  int B, int N, int K, void* stream) {
  if (B <= 0 || N <= 0 || K <= 0) return is::fail(__func__, -22);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int chunks = (N + is::DG_KC - 1) / is::DG_KC, mtiles = (B + is::DG_M - 1) / is::DG_M;
  hipLaunchKernelGGL(is::linear_fwd_splitk_kernel, dim3((K + is::DG_N - 1) / is::DG_N, chunks, mtiles), dim3(256), 0, st, x, ld_x, W,
                     ld_w, scratch, B, N, K);
  const long long total = (long long)B * K;
  hipLaunchKernelGGL(is::linear_dgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, scratch, y, B, K,
                     chunks, mtiles, bias);
  return is::launch_status(__func__);
}
