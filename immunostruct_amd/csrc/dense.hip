// Weight gradient of a Linear layer with a SMALL batch (the contraction dimension): dW [N, K] = gy^T x, db [N] = sum_b gy,
// for gy [B, N], x [B, K], B <= a few hundred.  These are the two large matrices of the sequence VAE
// (vae_fc1: 512 x 5943, vae_fc4: 5943 x 512; models/hybrid_models.py:297-308) whose weight gradients hipBLASLt runs
// with 86 us / 31 us kernels at B = 128 (K = 128 is far from its tuned shapes); a 64 x 64 output tile per workgroup with
// the two operand panels streamed through LDS does the same in ~12 us, and the bias gradient (another 13-60 us torch
// reduction over the batch) is a by-product.  Forward and input gradient stay on the library GEMMs.
// Fixed summation order over the batch -> bitwise reproducible.
#include "common.h"

namespace is {

__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ gy, int ld_g, const float* __restrict__ x,
                                                           int ld_x, float* __restrict__ dW, float* __restrict__ db,
                                                           int B, int N, int K) {
  __shared__ float gs[TE * LD];   // gy chunk: [32 batch rows][64 output columns n]
  __shared__ float xs[TE * LD];   // x  chunk: [32 batch rows][64 input columns k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
  const int mt = wave >> 1, nt = wave & 1;            // this wave's 32 x 32 quadrant of the 64 x 64 tile
  f32x16 acc[1][1];
  zero_acc2(acc);
  float colsum = 0.0f;                                 // lane = column n0 + lane (waves 0 only), bias gradient
  for (int b0 = 0; b0 < B; b0 += TE) {
    __syncthreads();
    // stage: 32 rows x 64 columns of each operand; thread -> (row = tid / 8 [+ 0], 8 consecutive columns)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int row = (tid >> 4) + half * 16, c = (tid & 15) * 4;
      const int b = b0 + row;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int n = n0 + c + u, k = k0 + c + u;
        gs[row * LD + c + u] = (b < B && n < N) ? gy[(size_t)b * ld_g + n] : 0.0f;
        xs[row * LD + c + u] = (b < B && k < K) ? x[(size_t)b * ld_x + k] : 0.0f;
      }
    }
    __syncthreads();
    mm_outer<1, 1>(acc, gs + mt * 32, xs + nt * 32, lane);
    if (blockIdx.y == 0 && wave == 0) {
#pragma unroll 8
      for (int e = 0; e < TE; ++e) colsum += gs[e * LD + lane];
    }
  }
  const int r = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int n = n0 + mt * 32 + tile_row(t, hf), k = k0 + nt * 32 + r;
    if (n < N && k < K) dW[(size_t)n * K + k] = acc[0][0][t];
  }
  if (db != nullptr && blockIdx.y == 0 && wave == 0 && n0 + lane < N) db[n0 + lane] = colsum;
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
