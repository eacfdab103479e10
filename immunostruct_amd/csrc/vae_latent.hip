// The latent block of the sequence VAE as one forward and two backward launches (reference models/hybrid_models.py:297-308, 334-340):
//
//   h1 = relu(a1)                      a1 = vae_fc1(x) [B, Hd] (pre-activation; csrc/dense.hip, is_linear_fwd_long)
//   mu = W21 h1 + b21 ; logvar = W22 h1 + b22            [B, 32]
//   z  = mu + eps * exp(0.5 logvar)                      eps supplied by the caller (torch.randn_like, the reference's draw)
//   zp = [z | p]                                         p = property embedding [B, P] (P <= 16, may be 0)
//   h3 = relu(W3 zp + b3)                                [B, Hd]  -> vae_fc4 (library GEMM)
//
// is_vae_latent_fwd       one workgroup per sample; the 64 x Hd and Hd x (32 + P) products are 52 k MACs per sample, so
//                         plain FMAs on weights streamed from L2 (every workgroup reads the same 200 KB)
// is_vae_latent_bwd       data path as one MFMA launch: d a3 = g_h3 * [h3 > 0]; d zp = g_zp + W3^T d a3; d p; d mu / d logvar
//                         totals; d a1 = (W21^T dmu + W22^T dlv) * [a1 > 0]; then the weight pass:
// is_vae_latent_bwd_wgrad one wave per 16 x 16 tile of the weight gradients, contraction over the batch on MFMA in a fixed order:
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

// backward data path in ONE launch: d a3 = g_h3 * [h3 > 0] (stored), d zp = g_zp + d a3 W3 on MFMA, the reparameterisation
// (d mu / d logvar totals, d p), then d a1 = ([d mu | d logvar] [W21 ; W22]) * [a1 > 0].  Workgroup (g, c) = 16 samples x one
// quarter of d a1's columns: every workgroup of a sample group computes the (small) first half itself -- the four waves each
// contract a quarter of the Hd range for all column tiles of zp on v_mfma_f32_16x16x4_f32, operands straight from global memory
// (all loads of a wave independent), quarters summed in wave order through LDS -- and keeps [d mu | d logvar] in LDS as the A
// operand of its share of the second half; workgroup (g, 0) stores d a3, d mu, d logvar, d p.
constexpr int VWT = (VL + VP_MAX + 15) / 16;      // column tiles of zp (3)
constexpr int VSS = 8;                            // super-steps (16 contraction rows each) per batch of loads: Hd = 512 is ONE round trip
__global__ __launch_bounds__(256) void vae_latent_bwd_data_kernel(
    const float* __restrict__ g_h3, const float* __restrict__ h3, const float* __restrict__ g_mu,
    const float* __restrict__ g_lv, const float* __restrict__ g_zp, const float* __restrict__ eps,
    const float* __restrict__ logvar, const float* __restrict__ a1, const float* __restrict__ W21,
    const float* __restrict__ W22, int P, const float* __restrict__ W3,
    float* __restrict__ d_a3, float* __restrict__ dmu, float* __restrict__ dlv, float* __restrict__ d_p,
    float* __restrict__ d_a1, int B, int Hd) {
  __shared__ float part[4][16][16 * VWT + 1];
  __shared__ float dml[16][2 * VL + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int W = VL + P, b0 = blockIdx.x * 16, wt = (W + 15) / 16;
  const int brow = min(b0 + r, B - 1);
  const bool store = blockIdx.y == 0, store_a3 = store && b0 + r < B;
  const int jq = Hd / 4;                                            // a multiple of 4
  // the second half's B operands and masks do not depend on the first half: the first two column tiles of this wave are
  // fetched now (in flight under everything below); tiles beyond them (Hd > 512) are fetched when their turn comes
  const int ktiles = Hd / 16, kq = (ktiles + 3) / 4, kt_end = min(ktiles, ((int)blockIdx.y + 1) * kq);
  float pb[2][16], pm[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int k0 = min((int)blockIdx.y * kq + wave + 4 * i, ktiles - 1) * 16;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      pb[i][u] = W21[(size_t)(4 * u + q) * Hd + k0 + r];
      pb[i][8 + u] = W22[(size_t)(4 * u + q) * Hd + k0 + r];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) pm[i][t] = a1[(size_t)min(b0 + tile16_row(t, q), B - 1) * Hd + k0 + r];
  }
  f32x4 acc[VWT];
#pragma unroll
  for (int c = 0; c < VWT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  // contraction index of step (ss, t), operand slot q: j = jb + 16 ss + 4 q + t -- lane (r, q) then owns FOUR CONSECUTIVE
  // elements of its row per super-step (one 16-byte load of h3 / g_h3, one 16-byte store of d a3)
  for (int jb = wave * jq; jb < (wave + 1) * jq; jb += 16 * VSS) {
    f32x4 hv[VSS], gv[VSS];
    float bv[VWT][VSS][4];
#pragma unroll
    for (int ss = 0; ss < VSS; ++ss) {
      const int j = min(jb + 16 * ss + 4 * q, Hd - 4);
      hv[ss] = *reinterpret_cast<const f32x4*>(h3 + (size_t)brow * Hd + j);
      gv[ss] = g_h3 != nullptr ? *reinterpret_cast<const f32x4*>(g_h3 + (size_t)brow * Hd + j) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < VWT; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t) bv[c][ss][t] = (16 * c + r < W) ? W3[(size_t)(j + t) * W + 16 * c + r] : 0.0f;
    }
#pragma unroll
    for (int ss = 0; ss < VSS; ++ss) {
      const int j = jb + 16 * ss + 4 * q;
      if (j < (wave + 1) * jq) {        // (per 16-lane group: a trailing partial super-step contributes zeros)
#pragma unroll
        for (int t = 0; t < 4; ++t) gv[ss][t] = hv[ss][t] > 0.0f ? gv[ss][t] : 0.0f;
        if (store_a3) *reinterpret_cast<f32x4*>(d_a3 + (size_t)(b0 + r) * Hd + j) = gv[ss];
      } else {
        gv[ss] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (jb + 16 * ss < (wave + 1) * jq) {                // wave-uniform: the super-step exists
#pragma unroll
        for (int c = 0; c < VWT; ++c)
          if (c < wt) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[ss][t], bv[c][ss][t], acc[c], 0, 0, 0);
          }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < VWT; ++c)
#pragma unroll
    for (int t = 0; t < 4; ++t) part[wave][tile16_row(t, q)][16 * c + r] = acc[c][t];
  __syncthreads();
  {
    const int row = tid >> 4, bb = b0 + row;
#pragma unroll
    for (int ct = 0; ct < VWT; ++ct) {
      const int c = 16 * ct + (tid & 15);
      if (c < W) {
        const bool live = bb < B;
        const int bs = min(bb, B - 1);
        const float v = ((part[0][row][c] + part[1][row][c]) + part[2][row][c]) + part[3][row][c] +
                        (g_zp != nullptr ? g_zp[(size_t)bs * W + c] : 0.0f);
        if (c < VL) {
          const float lv = logvar[(size_t)bs * VL + c];
          const float m = v + (g_mu != nullptr ? g_mu[(size_t)bs * VL + c] : 0.0f);
          const float l = v * eps[(size_t)bs * VL + c] * 0.5f * __expf(0.5f * lv) + (g_lv != nullptr ? g_lv[(size_t)bs * VL + c] : 0.0f);
          dml[row][c] = live ? m : 0.0f;
          dml[row][VL + c] = live ? l : 0.0f;
          if (store && live) {
            dmu[(size_t)bb * VL + c] = m;
            dlv[(size_t)bb * VL + c] = l;
          }
        } else if (store && live) {
          d_p[(size_t)bb * P + (c - VL)] = v;
        }
      }
    }
  }
  __syncthreads();
  // second half: this workgroup's quarter of the Hd / 16 column tiles, dealt to the waves round-robin; 16 MFMA steps over the
  // 64 latent rows (A = [d mu | d logvar] from LDS, the same for every tile)
  float av[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) av[u] = dml[r][4 * u + q];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int kt = blockIdx.y * kq + wave + 4 * i;
    if (kt < kt_end) {
      f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 16; ++u) o = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], pb[i][u], o, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int bb = b0 + tile16_row(t, q);
        if (bb < B) d_a1[(size_t)bb * Hd + kt * 16 + r] = pm[i][t] > 0.0f ? o[t] : 0.0f;
      }
    }
  }
  for (int kt = blockIdx.y * kq + wave + 8; kt < kt_end; kt += 4) {
    const int k0 = kt * 16;
    float bv[16];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      bv[u] = W21[(size_t)(4 * u + q) * Hd + k0 + r];
      bv[8 + u] = W22[(size_t)(4 * u + q) * Hd + k0 + r];
    }
    float am[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) am[t] = a1[(size_t)min(b0 + tile16_row(t, q), B - 1) * Hd + k0 + r];
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 16; ++u) o = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], o, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int bb = b0 + tile16_row(t, q);
      if (bb < B) d_a1[(size_t)bb * Hd + k0 + r] = am[t] > 0.0f ? o[t] : 0.0f;
    }
  }
}

// the concatenated gradient [dW21 (32 x Hd) | dW22 (32 x Hd) | db21 (32) | db22 (32) | dW3 (Hd x W) | db3 (Hd)]: ONE WAVE per
// 16 x 16 output tile, the contraction over the batch as v_mfma_f32_16x16x4_f32 steps in batch order (fixed), operands straight
// from global memory (every load of a wave is independent).  Many one-wave workgroups of a few microseconds each: beside a
// persistent layer kernel they slip into whatever slot frees up and are gone again, instead of holding slots for a whole
// contraction loop (HISTORY.md section 6).  The bias gradients are the column sums of the A operand of the first tile column.
constexpr int WSTEPS = 32;       // MFMA steps (4 batch rows each) per batch of loads: B = 128 is ONE round trip
__global__ __launch_bounds__(64) void vae_latent_bwd_wgrad_kernel(
    const float* __restrict__ a1, const float* __restrict__ dmu, const float* __restrict__ dlv,
    const float* __restrict__ zp, const float* __restrict__ d_a3, int P, int B, int Hd, float* __restrict__ out) {
  const int W = VL + P, lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  const int kt = Hd / 16, tiles_a = 4 * kt, wt = (W + 15) / 16;
  const long long n_w2 = 2LL * VL * Hd, n_b2 = 2 * VL, n_w3 = (long long)Hd * W;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  float colsum = 0.0f;
  int tile = blockIdx.x;
  if (tile < tiles_a) {
    // rows o = 16 ot + i of [dmu | dlv]^T (o < 32: mu rows), columns k0 + j of relu(a1)
    const int ot = tile / kt, k0 = (tile % kt) * 16;
    const float* __restrict__ d = (ot < 2) ? dmu + ot * 16 : dlv + (ot - 2) * 16;
    for (int b0 = 0; b0 < B; b0 += 4 * WSTEPS) {
      float av[WSTEPS], bv[WSTEPS];
#pragma unroll
      for (int u = 0; u < WSTEPS; ++u) {
        const int b = b0 + 4 * u + q;
        const bool ok = b < B;
        av[u] = ok ? d[(size_t)b * VL + r] : 0.0f;
        bv[u] = ok ? fmaxf(a1[(size_t)b * Hd + k0 + r], 0.0f) : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < WSTEPS; ++u) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
        colsum += av[u];
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) out[(size_t)(ot * 16 + tile16_row(t, q)) * Hd + k0 + r] = acc[t];
    if (k0 == 0) {
      colsum += __shfl_xor(colsum, 16, 64);
      colsum += __shfl_xor(colsum, 32, 64);
      if (q == 0) out[n_w2 + ot * 16 + r] = colsum;
    }
  } else {
    // rows j0 + i of d_a3^T, columns c0 + j of zp
    tile -= tiles_a;
    const int j0 = (tile / wt) * 16, c0 = (tile % wt) * 16;
    const bool cok = c0 + r < W;
    for (int b0 = 0; b0 < B; b0 += 4 * WSTEPS) {
      float av[WSTEPS], bv[WSTEPS];
#pragma unroll
      for (int u = 0; u < WSTEPS; ++u) {
        const int b = b0 + 4 * u + q;
        const bool ok = b < B;
        av[u] = ok ? d_a3[(size_t)b * Hd + j0 + r] : 0.0f;
        bv[u] = (ok && cok) ? zp[(size_t)b * W + c0 + r] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < WSTEPS; ++u) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
        colsum += av[u];
      }
    }
    if (cok) {
#pragma unroll
      for (int t = 0; t < 4; ++t) out[n_w2 + n_b2 + (size_t)(j0 + tile16_row(t, q)) * W + c0 + r] = acc[t];
    }
    if (c0 == 0) {
      colsum += __shfl_xor(colsum, 16, 64);
      colsum += __shfl_xor(colsum, 32, 64);
      if (q == 0) out[n_w2 + n_b2 + n_w3 + j0 + r] = colsum;
    }
  }
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
  if (!vae_dims_ok(B, Hd, L, P) || (P > 0 && p == nullptr)) return is::fail(__func__, -22);
  hipLaunchKernelGGL(is::vae_latent_fwd_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), a1, W21, b21, W22, b22,
                     eps, p, P, W3, b3, mu, logvar, zp, h3, Hd);
  return is::launch_status(__func__);
}

// number of floats of the weight-gradient vector [dW21 | dW22 | db21 | db22 | dW3 | db3]
extern "C" int is_vae_latent_grad_floats(int Hd, int P) { return 2 * is::VL * Hd + 2 * is::VL + Hd * (is::VL + P) + Hd; }

// upstream gradients g_h3 [B,Hd], g_mu / g_lv [B,32], g_zp [B, 32 + P] (each may be NULL: zero); outputs d_a1 [B,Hd] (gradient of
// vae_fc1's output), d_p [B,P] (NULL when P == 0); leaves d_a3 [B,Hd], dmu, dlv [B,32] for the weight pass.  One launch.
extern "C" int is_vae_latent_bwd_data(const float* g_h3, const float* h3, const float* g_mu, const float* g_lv, const float* g_zp,
                                      const float* eps, const float* logvar, const float* a1, const float* W21, const float* W22,
                                      int P, const float* W3, float* d_a3, float* dmu, float* dlv, float* d_p, float* d_a1, int B,
                                      int Hd, int L, void* stream) {
  if (!vae_dims_ok(B, Hd, L, P) || (P > 0 && d_p == nullptr)) return is::fail(__func__, -22);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(is::vae_latent_bwd_data_kernel, dim3((B + 15) / 16, 4), dim3(256), 0, st, g_h3, h3, g_mu, g_lv, g_zp, eps, logvar,
                     a1, W21, W22, P, W3, d_a3, dmu, dlv, d_p, d_a1, B, Hd);
  return is::launch_status(__func__);
}

// the weight pass: wgrad [is_vae_latent_grad_floats] from a1, zp and what is_vae_latent_bwd_data left in d_a3 / dmu / dlv.
// Its own entry point so that a caller can put other work of the data path between the two (functional.VaeLatentFn launches
// the weight gradient of vae_fc1 there: the big one goes first, this one -- one-wave workgroups, no LDS -- trails).
extern "C" int is_vae_latent_bwd_wgrad(const float* a1, const float* dmu, const float* dlv, const float* zp, const float* d_a3,
                                       int P, float* wgrad, int B, int Hd, int L, void* stream) {
  if (!vae_dims_ok(B, Hd, L, P)) return is::fail(__func__, -22);
  const int wtiles = 4 * (Hd / 16) + (Hd / 16) * ((is::VL + P + 15) / 16);
  hipLaunchKernelGGL(is::vae_latent_bwd_wgrad_kernel, dim3(wtiles), dim3(64), 0, static_cast<hipStream_t>(stream), a1, dmu, dlv, zp,
                     d_a3, P, B, Hd, wgrad);
  return is::launch_status(__func__);
}

// both halves back to back; scratch: d_a3 [B,Hd], dmu, dlv [B,32].
extern "C" int is_vae_latent_bwd(const float* g_h3, const float* h3, const float* g_mu, const float* g_lv, const float* g_zp,
                                 const float* eps, const float* logvar, const float* a1, const float* zp, const float* W21,
                                 const float* W22, int P, const float* W3, float* d_a3, float* dmu, float* dlv, float* d_p,
                                 float* d_a1, float* wgrad, int B, int Hd, int L, void* stream) {
  const int rc = is_vae_latent_bwd_data(g_h3, h3, g_mu, g_lv, g_zp, eps, logvar, a1, W21, W22, P, W3, d_a3, dmu, dlv, d_p, d_a1, B, Hd,
                                        L, stream);
  return rc != 0 ? rc : is_vae_latent_bwd_wgrad(a1, dmu, dlv, zp, d_a3, P, wgrad, B, Hd, L, stream);
}
