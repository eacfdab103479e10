// The latent block of the sequence VAE as three launches (reference models/hybrid_models.py:297-308, 334-340):
//
//   h1 = relu(a1)                      a1 = vae_fc1(x) [B, Hd] (pre-activation, from the library GEMM)
//   mu = W21 h1 + b21 ; logvar = W22 h1 + b22            [B, 32]
//   z  = mu + eps * exp(0.5 logvar)                      eps supplied by the caller (torch.randn_like, the reference's draw)
//   zp = [z | p]                                         p = property embedding [B, P] (P <= 16, may be 0)
//   h3 = relu(W3 zp + b3)                                [B, Hd]  -> vae_fc4 (library GEMM)
//
// is_vae_latent_fwd       one workgroup per sample; the 64 x Hd and Hd x (32 + P) products are 52 k MACs per sample, so
//                         plain FMAs on weights streamed from L2 (every workgroup reads the same 200 KB)
// is_vae_latent_bwd_data  one workgroup per sample: d a3 = g_h3 * [h3 > 0]; d zp = g_zp + W3^T d a3; d p; d mu / d logvar
//                         totals; d a1 = (W21^T dmu + W22^T dlv) * [a1 > 0]; leaves dmu, dlv, d a3 for the weight pass
// is_vae_latent_bwd_wgrad one thread per weight-gradient entry, contraction over the batch in a fixed order:
//                         dW21 / dW22 [32, Hd], db21 / db22, dW3 [Hd, 32 + P], db3
// Replaces ~25 hipBLASLt / elementwise launches per step (Cijk MT32x32x16, exp, mul, add, cat, threshold, reduce, copy).
#include "common.h"

namespace is {

constexpr int VL = 32;           // latent width (reference vae_latent_dim = 32)
constexpr int VP_MAX = 16;       // widest property embedding
constexpr int VHD_MAX = 2048;    // widest hidden layer

__global__ __launch_bounds__(256) void vae_latent_fwd_kernel(
    const float* __restrict__ a1, const float* __restrict__ W21, const float* __restrict__ b21,
    const float* __restrict__ W22, const float* __restrict__ b22, const float* __restrict__ eps,
    const float* __restrict__ p, int P, const float* __restrict__ W3, const float* __restrict__ b3,
    float* __restrict__ mu, float* __restrict__ logvar, float* __restrict__ zp_out, float* __restrict__ h3, int Hd) {
  __shared__ float h1s[VHD_MAX];
  __shared__ float ml[2 * VL];
  __shared__ float zps[VL + VP_MAX];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int k = tid; k < Hd; k += 256) h1s[k] = fmaxf(a1[(size_t)b * Hd + k], 0.0f);
  __syncthreads();
  {
    // 64 outputs (mu | logvar) x 4 contiguous quarters of the Hd-long dot product; the 4 partial sums sit in adjacent lanes
    const int o = tid >> 2, part = tid & 3;
    const float* w = (o < VL ? W21 + (size_t)o * Hd : W22 + (size_t)(o - VL) * Hd);
    const int q = Hd / 4;
    float acc = 0.0f;
    for (int k = part * q; k < (part + 1) * q; k += 4) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
      acc = __builtin_fmaf(wv[0], h1s[k], acc);
      acc = __builtin_fmaf(wv[1], h1s[k + 1], acc);
      acc = __builtin_fmaf(wv[2], h1s[k + 2], acc);
      acc = __builtin_fmaf(wv[3], h1s[k + 3], acc);
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (part == 0) ml[o] = acc + (o < VL ? b21[o] : b22[o - VL]);
  }
  __syncthreads();
  const int W = VL + P;
  if (tid < VL) {
    const float m = ml[tid], lv = ml[VL + tid];
    const float z = m + eps[(size_t)b * VL + tid] * __expf(0.5f * lv);
    mu[(size_t)b * VL + tid] = m;
    logvar[(size_t)b * VL + tid] = lv;
    zps[tid] = z;
    zp_out[(size_t)b * W + tid] = z;
  } else if (tid < W) {
    const float v = p[(size_t)b * P + (tid - VL)];
    zps[tid] = v;
    zp_out[(size_t)b * W + tid] = v;
  }
  __syncthreads();
  for (int j = tid; j < Hd; j += 256) {
    const float* w = W3 + (size_t)j * W;
    float acc = b3[j];
    for (int c = 0; c < W; ++c) acc = __builtin_fmaf(w[c], zps[c], acc);
    h3[(size_t)b * Hd + j] = fmaxf(acc, 0.0f);
  }
}

__global__ __launch_bounds__(256) void vae_latent_bwd_data_kernel(
    const float* __restrict__ g_h3, const float* __restrict__ h3, const float* __restrict__ g_mu,
    const float* __restrict__ g_lv, const float* __restrict__ g_zp, const float* __restrict__ eps,
    const float* __restrict__ logvar, const float* __restrict__ a1, const float* __restrict__ W21,
    const float* __restrict__ W22, int P, const float* __restrict__ W3,
    float* __restrict__ d_a3, float* __restrict__ dmu, float* __restrict__ dlv, float* __restrict__ d_p,
    float* __restrict__ d_a1, int Hd) {
  __shared__ float da3s[VHD_MAX];
  __shared__ float part[4][64];
  __shared__ float dml[2 * VL];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = VL + P;
  for (int j = tid; j < Hd; j += 256) {
    const float g = (g_h3 != nullptr && h3[(size_t)b * Hd + j] > 0.0f) ? g_h3[(size_t)b * Hd + j] : 0.0f;
    da3s[j] = g;
    d_a3[(size_t)b * Hd + j] = g;
  }
  __syncthreads();
  {
    // d zp[c] = sum_j W3[j][c] d a3[j]: lane = column c (rows of W3 are contiguous), wave = quarter of the j range
    const int q = Hd / 4;
    // four independent chains (j, j+1, j+2, j+3): the loads of a row group are in flight together; fixed summation order
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f;
    if (lane < W) {
#pragma unroll 4
      for (int j = wave * q; j < (wave + 1) * q; j += 4) {
        c0 = __builtin_fmaf(W3[(size_t)j * W + lane], da3s[j], c0);
        c1 = __builtin_fmaf(W3[(size_t)(j + 1) * W + lane], da3s[j + 1], c1);
        c2 = __builtin_fmaf(W3[(size_t)(j + 2) * W + lane], da3s[j + 2], c2);
        c3 = __builtin_fmaf(W3[(size_t)(j + 3) * W + lane], da3s[j + 3], c3);
      }
    }
    part[wave][lane] = (c0 + c1) + (c2 + c3);
  }
  __syncthreads();
  if (tid < W) {
    const float v = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid] + (g_zp != nullptr ? g_zp[(size_t)b * W + tid] : 0.0f);
    if (tid < VL) {
      const float lv = logvar[(size_t)b * VL + tid];
      const float m = v + (g_mu != nullptr ? g_mu[(size_t)b * VL + tid] : 0.0f);
      const float l = v * eps[(size_t)b * VL + tid] * 0.5f * __expf(0.5f * lv) + (g_lv != nullptr ? g_lv[(size_t)b * VL + tid] : 0.0f);
      dml[tid] = m;
      dml[VL + tid] = l;
      dmu[(size_t)b * VL + tid] = m;
      dlv[(size_t)b * VL + tid] = l;
    } else {
      d_p[(size_t)b * P + (tid - VL)] = v;
    }
  }
  __syncthreads();
  for (int k = tid; k < Hd; k += 256) {
    float m = 0.0f, l = 0.0f;
#pragma unroll
    for (int o = 0; o < VL; ++o) {      // fully unrolled: the 64 (coalesced) loads of a thread are independent
      m = __builtin_fmaf(W21[(size_t)o * Hd + k], dml[o], m);
      l = __builtin_fmaf(W22[(size_t)o * Hd + k], dml[VL + o], l);
    }
    d_a1[(size_t)b * Hd + k] = a1[(size_t)b * Hd + k] > 0.0f ? (m + l) : 0.0f;
  }
}

// entry e of the concatenated gradient [dW21 (32 x Hd) | dW22 (32 x Hd) | db21 (32) | db22 (32) | dW3 (Hd x W) | db3 (Hd)]
__global__ __launch_bounds__(256) void vae_latent_bwd_wgrad_kernel(
    const float* __restrict__ a1, const float* __restrict__ dmu, const float* __restrict__ dlv,
    const float* __restrict__ zp, const float* __restrict__ d_a3, int P, int B, int Hd, float* __restrict__ out) {
  const int W = VL + P;
  const long long n_w2 = 2LL * VL * Hd, n_b2 = 2 * VL, n_w3 = (long long)Hd * W;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_w2 + n_b2 + n_w3 + Hd) return;
  float acc = 0.0f;
  if (e < n_w2) {
    const int o = (int)(e / Hd), k = (int)(e % Hd);                 // o in [0, 64): mu rows then logvar rows
    const float* d = (o < VL) ? dmu + o : dlv + (o - VL);
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 4 <= B; b += 4)
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = __builtin_fmaf(d[(size_t)(b + u) * VL], fmaxf(a1[(size_t)(b + u) * Hd + k], 0.0f), c[u]);
    for (; b < B; ++b) c[0] = __builtin_fmaf(d[(size_t)b * VL], fmaxf(a1[(size_t)b * Hd + k], 0.0f), c[0]);
    acc = (c[0] + c[1]) + (c[2] + c[3]);
  } else if (e < n_w2 + n_b2) {
    const int o = (int)(e - n_w2);
    const float* d = (o < VL) ? dmu + o : dlv + (o - VL);
    for (int b = 0; b < B; ++b) acc += d[(size_t)b * VL];
  } else if (e < n_w2 + n_b2 + n_w3) {
    const long long r = e - n_w2 - n_b2;
    const int j = (int)(r / W), c = (int)(r % W);
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 4 <= B; b += 4)
#pragma unroll
      for (int u = 0; u < 4; ++u) s4[u] = __builtin_fmaf(d_a3[(size_t)(b + u) * Hd + j], zp[(size_t)(b + u) * W + c], s4[u]);
    for (; b < B; ++b) s4[0] = __builtin_fmaf(d_a3[(size_t)b * Hd + j], zp[(size_t)b * W + c], s4[0]);
    acc = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  } else {
    const int j = (int)(e - n_w2 - n_b2 - n_w3);
    for (int b = 0; b < B; ++b) acc += d_a3[(size_t)b * Hd + j];
  }
  out[e] = acc;
}

}  // namespace is

static bool vae_dims_ok(int B, int Hd, int L, int P) {
  return B > 0 && L == is::VL && P >= 0 && P <= is::VP_MAX && Hd >= 16 && Hd <= is::VHD_MAX && (Hd % 16) == 0;
}

// mu, logvar [B,32], zp [B, 32 + P] = [z | p], h3 [B,Hd] = relu(vae_fc3(zp)) from a1 [B,Hd] = vae_fc1(x) (pre-activation),
// eps [B,32], p [B,P] (NULL when P == 0); W21 / W22 [32,Hd], W3 [Hd, 32 + P].  L must be 32, Hd a multiple of 16 <= 2048.
extern "C" int is_vae_latent_fwd(const float* a1, const float* W21, const float* b21, const float* W22, const float* b22,
                                 const float* eps, const float* p, int P, const float* W3, const float* b3, float* mu,
                                 float* logvar, float* zp, float* h3, int B, int Hd, int L, void* stream) {
  if (!vae_dims_ok(B, Hd, L, P) || (P > 0 && p == nullptr)) return -22;
  hipLaunchKernelGGL(is::vae_latent_fwd_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), a1, W21, b21, W22, b22,
                     eps, p, P, W3, b3, mu, logvar, zp, h3, Hd);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

// number of floats of the weight-gradient vector [dW21 | dW22 | db21 | db22 | dW3 | db3]
extern "C" int is_vae_latent_grad_floats(int Hd, int P) { return 2 * is::VL * Hd + 2 * is::VL + Hd * (is::VL + P) + Hd; }

// upstream gradients g_h3 [B,Hd], g_mu / g_lv [B,32], g_zp [B, 32 + P] (each may be NULL: zero); outputs d_a1 [B,Hd] (gradient of
// vae_fc1's output), d_p [B,P] (NULL when P == 0), wgrad [is_vae_latent_grad_floats]; scratch: d_a3 [B,Hd], dmu, dlv [B,32].
extern "C" int is_vae_latent_bwd(const float* g_h3, const float* h3, const float* g_mu, const float* g_lv, const float* g_zp,
                                 const float* eps, const float* logvar, const float* a1, const float* zp, const float* W21,
                                 const float* W22, int P, const float* W3, float* d_a3, float* dmu, float* dlv, float* d_p,
                                 float* d_a1, float* wgrad, int B, int Hd, int L, void* stream) {
  if (!vae_dims_ok(B, Hd, L, P) || (P > 0 && d_p == nullptr)) return -22;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(is::vae_latent_bwd_data_kernel, dim3(B), dim3(256), 0, st, g_h3, h3, g_mu, g_lv, g_zp, eps, logvar, a1, W21,
                     W22, P, W3, d_a3, dmu, dlv, d_p, d_a1, Hd);
  const int total = is_vae_latent_grad_floats(Hd, P);
  hipLaunchKernelGGL(is::vae_latent_bwd_wgrad_kernel, dim3((total + 255) / 256), dim3(256), 0, st, a1, dmu, dlv, zp, d_a3, P, B, Hd,
                     wgrad);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}
