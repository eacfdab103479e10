// Weight gradient of a Linear layer with a SMALL batch (the contraction dimension): dW [N, K] = gy^T x, db [N] = sum_b gy,
// for gy [B, N], x [B, K], B <= a few hundred.  These are the two large matrices of the sequence VAE
// (vae_fc1: 512 x 5943, vae_fc4: 5943 x 512; models/hybrid_models.py:297-308) whose weight gradients hipBLASLt runs
// with 86 us / 31 us kernels at B = 128 (K = 128 is far from its tuned shapes); a 64 x 64 output tile per workgroup with
// the two operand panels staged in LDS in one go does the same in ~12 us, and the bias gradient (another 13-60 us torch
// reduction over the batch) is a by-product.  Forward and input gradient stay on the library GEMMs.
// Fixed summation order over the batch -> bitwise reproducible.
#include "common.h"

namespace is {

constexpr int WG_B = 128;      // batch rows staged per pass

// One workgroup owns a 64 (n) x 64 (k) tile of dW.  Both operand panels of a pass (128 batch rows x 64 columns each) are
// fetched with ALL loads in flight at once -- lane = column, one coalesced 256-byte row segment per load, 64 loads per lane --
// and staged in LDS behind ONE barrier pair; wave (mt, nt) then accumulates its 32 x 32 quadrant over the 128 rows on
// v_mfma_f32_32x32x2_f32 (bit-equal to an fmaf chain over b = 0 .. B-1).  2 workgroups per CU (70 KB LDS).
__global__ __launch_bounds__(256, 2) void linear_wgrad_kernel(const float* __restrict__ gy, int ld_g, const float* __restrict__ x,
                                                              int ld_x, float* __restrict__ dW, float* __restrict__ db,
                                                              int B, int N, int K) {
  __shared__ float gs[WG_B * LD];   // gy panel: [128 batch rows][64 output columns n]
  __shared__ float xs[WG_B * LD];   // x  panel: [128 batch rows][64 input columns k]
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
  for (int b0 = 0; b0 < B; b0 += WG_B) {
    constexpr int RPW = WG_B / 4;                      // rows staged per wave
    float gv[RPW], xv[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int b = min(b0 + wave * RPW + i, B - 1);
      gv[i] = gy[(size_t)b * ld_g + nc];
      xv[i] = x[(size_t)b * ld_x + kc];
    }
    if (b0 > 0) __syncthreads();                       // the previous pass' MFMAs are done with the panels
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int row = wave * RPW + i;
      const bool b_ok = b0 + row < B;
      gs[row * LD + lane] = (b_ok && n_ok) ? gv[i] : 0.0f;
      xs[row * LD + lane] = (b_ok && k_ok) ? xv[i] : 0.0f;
    }
    __syncthreads();
    // acc (32 x 32) += sum over the 128 staged rows e of gs[e][mt*32 + i] * xs[e][nt*32 + j]; half hf walks rows [64 hf, 64 hf + 64)
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

}  // namespace is

// gy [B, ld_g] (N valid columns), x [B, ld_x] (K valid columns) -> dW [N, K] (row-major, the nn.Linear weight layout),
// db [N] (may be NULL).
extern "C" int is_linear_wgrad(const float* gy, int ld_g, const float* x, int ld_x, float* dW, float* db, int B, int N, int K,
                               void* stream) {
  if (B <= 0 || N <= 0 || K <= 0) return -22;
  hipLaunchKernelGGL(is::linear_wgrad_kernel, dim3((N + 63) / 64, (K + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(stream),
                     gy, ld_g, x, ld_x, dW, db, B, N, K);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}
