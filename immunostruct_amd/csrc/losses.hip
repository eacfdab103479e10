// Fused training losses of the reference's `Losses` class (utils/loss.py:13-31):
//
//   regression : c_pred * MSE(logit, y) + c_mse * MSE(recon, x) + c_kld * KLD_mean(mu, logvar)
//   bce        : c_pred * BCEWithLogits(logit, y, pos_weight) + c_mse * ... + c_kld * ...
//   KLD_mean   = -0.5 * mean(1 + logvar - mu^2 - exp(logvar))      (a MEAN, not a sum)
//
// The value AND the gradients w.r.t. recon, mu, logvar and logit are produced by
// one two-kernel launch (the reconstruction term streams B x 5943 floats once:
// HBM-bound; gradients are written in the same pass), so the autograd backward
// is a scalar scale.  Reductions are two-stage with a fixed order (no atomics).
#include "common.h"

namespace is {

constexpr int LOSS_BLOCK = 256;

// stage 1: reconstruction MSE partial sums + d_recon
__global__ __launch_bounds__(LOSS_BLOCK) void recon_mse_kernel(const float* __restrict__ recon,
                                                               const float* __restrict__ x, float* __restrict__ d_recon,
                                                               float* __restrict__ partials, long long total,
                                                               float gscale) {
  __shared__ float red[LOSS_BLOCK / 64];
  float acc = 0.0f;
  const long long stride = (long long)gridDim.x * LOSS_BLOCK;
  for (long long i = (long long)blockIdx.x * LOSS_BLOCK + threadIdx.x; i < total; i += stride) {
    const float d = recon[i] - x[i];
    acc += d * d;
    d_recon[i] = gscale * d;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return ((red[0] + red[1]) + red[2]) + red[3];
}

// stage 2 (one workgroup of 1024 threads: the kernel is a latency chain on the step's critical path, so the three loops are
// kept to at most four trips): finish MSE, KLD + grads, prediction term + grads, total.
// PARTS selects the terms of ONE launch (bit 0: reconstruction MSE from stage 1's partials + the total, bit 1: KLD + its gradients,
// bit 2: prediction term + its gradient).  7 is the fused launch.  The split forms let a caller evaluate the prediction term where
// the logit is produced and the sequence terms where the reconstruction is, with no wait between the two (functional.vae_loss,
// deferred total): 4 and 2 leave their raw sums in sums[2] / sums[1], 1 reads them there and writes out[] / total.  Every sum is
// formed by the same threads in the same order in all forms: the split launches reproduce the fused one bit for bit.
constexpr int FIN_BLOCK = 1024, FIN_WAVES = FIN_BLOCK / 64;
constexpr int PART_MSE = 1, PART_KLD = 2, PART_PRED = 4;
template <int PARTS>
__global__ __launch_bounds__(FIN_BLOCK) void loss_finish_kernel(
    const float* __restrict__ partials, int nparts, long long recon_total,
    const float* __restrict__ mu, const float* __restrict__ logvar, float* __restrict__ d_mu,
    float* __restrict__ d_logvar, int latent_total,
    const float* __restrict__ logit, const float* __restrict__ y, float* __restrict__ d_logit, int batch,
    int mode, float pos_weight, float c_pred, float c_mse, float c_kld, float* __restrict__ sums, float* __restrict__ out,
    float* __restrict__ total) {
  __shared__ float red[3][FIN_WAVES];
  const int tid = threadIdx.x;
  // the three per-thread partial sums first (their loads are independent: all in flight together), then ONE pass of
  // wave reductions and one barrier for all three; fixed order throughout
  float acc_m = 0.0f, acc_k = 0.0f, acc_p = 0.0f;
  if ((PARTS & PART_MSE) && recon_total > 0)
    for (int i = tid; i < nparts; i += FIN_BLOCK) acc_m += partials[i];
  const float inv = latent_total > 0 ? 1.0f / (float)latent_total : 0.0f;
  if (PARTS & PART_KLD)
    for (int i = tid; i < latent_total; i += FIN_BLOCK) {
      // exp(lv) - 1 through expm1f: for a freshly initialised model logvar is close to 0 and (1 - exp(lv)) formed from a fast exp
      // loses most of its digits (the bias gradients of vae_fc22 are sums of these terms)
      const float m = mu[i], lv = logvar[i], em1 = expm1f(lv);
      acc_k += lv - m * m - em1;
      d_mu[i] = c_kld * m * inv;
      d_logvar[i] = c_kld * 0.5f * em1 * inv;
    }
  const float invb = 1.0f / (float)batch;
  if (PARTS & PART_PRED)
    for (int i = tid; i < batch; i += FIN_BLOCK) {
      const float z = logit[i], t = y[i];
      if (mode == 0) {
        const float d = z - t;
        acc_p += d * d;
        d_logit[i] = c_pred * 2.0f * d * invb;
      } else {
        // -[pw*t*log(sig(z)) + (1-t)*log(1-sig(z))], stable form
        const float lw = 1.0f + (pos_weight - 1.0f) * t;
        const float sp = log1pf(__expf(-fabsf(z))) + fmaxf(-z, 0.0f);  // softplus(-z)
        acc_p += (1.0f - t) * z + lw * sp;
        const float sg = 1.0f / (1.0f + __expf(-z));
        d_logit[i] = c_pred * ((1.0f - t) * sg - pos_weight * t * (1.0f - sg)) * invb;
      }
    }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    if (PARTS & PART_MSE) acc_m += __shfl_xor(acc_m, m, 64);
    if (PARTS & PART_KLD) acc_k += __shfl_xor(acc_k, m, 64);
    if (PARTS & PART_PRED) acc_p += __shfl_xor(acc_p, m, 64);
  }
  if ((tid & 63) == 0) { red[0][tid >> 6] = acc_m; red[1][tid >> 6] = acc_k; red[2][tid >> 6] = acc_p; }
  __syncthreads();
  if (tid == 0) {
    float s[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float v = 0.0f;
#pragma unroll
      for (int w = 0; w < FIN_WAVES; ++w) v += red[k][w];
      s[k] = v;
    }
    if (PARTS != 7) {
      if (PARTS & PART_KLD) sums[1] = s[1];
      if (PARTS & PART_PRED) sums[2] = s[2];
      if (!(PARTS & PART_MSE)) return;
      s[1] = sums[1];
      s[2] = sums[2];
    }
    const float mse = recon_total > 0 ? s[0] / (float)recon_total : 0.0f;
    const float kld = latent_total > 0 ? -0.5f * s[1] * inv : 0.0f;
    const float pred = s[2] * invb;
    out[0] = c_pred * pred + c_mse * mse + c_kld * kld;
    if (total != nullptr) total[0] = out[0];
    out[1] = pred; out[2] = mse; out[3] = kld;
  }
}

}  // namespace is

extern "C" int is_loss_partials_floats(void) { return 1024; }

static int recon_parts(long long recon_total) {
  const long long want = (recon_total + is::LOSS_BLOCK * 4 - 1) / (is::LOSS_BLOCK * 4);
  return (int)(want < 1024 ? want : 1024);
}

// Stage 1 of the reconstruction term on its own: partial sums of (recon - x)^2 into partials[] and
// d_recon = gscale * (recon - x) (gscale = c_mse * 2 / recon_total * upstream).  For callers that run it where recon is
// produced (the sequence branch's stream) and hand is_vae_loss the partials (recon = NULL there).
extern "C" int is_recon_mse(const float* recon, const float* x, float* d_recon, long long recon_total, float gscale,
                            float* partials, void* stream) {
  if (recon_total <= 0 || recon == nullptr || x == nullptr || d_recon == nullptr || partials == nullptr) return is::fail(__func__, -22);
  hipLaunchKernelGGL(is::recon_mse_kernel, dim3(recon_parts(recon_total)), dim3(is::LOSS_BLOCK), 0, static_cast<hipStream_t>(stream),
                     recon, x, d_recon, partials, recon_total, gscale);
  return is::launch_status(__func__);
}

// mode 0 = regression (MSE on the logit), 1 = BCE-with-logits(pos_weight).
// recon/x/d_recon may be NULL (recon_total = 0) and mu/logvar NULL (latent_total = 0)
// for `sequence=False`.  recon = NULL with recon_total > 0: partials[] already holds stage 1 (is_recon_mse).
// out[4] = {total, prediction term, recon MSE, KLD}.
extern "C" int is_vae_loss(const float* recon, const float* x, float* d_recon, long long recon_total,
                           const float* mu, const float* logvar, float* d_mu, float* d_logvar, int latent_total,
                           const float* logit, const float* y, float* d_logit, int batch, int mode,
                           float pos_weight, float c_pred, float c_mse, float c_kld, float* partials, float* out,
                           float* total, void* stream) {
  if (batch <= 0) return is::fail(__func__, -22);
  hipStream_t st = static_cast<hipStream_t>(stream);
  int nparts = 0;
  if (recon_total > 0) {
    nparts = recon_parts(recon_total);
    if (recon != nullptr)
      hipLaunchKernelGGL(is::recon_mse_kernel, dim3(nparts), dim3(is::LOSS_BLOCK), 0, st, recon, x, d_recon, partials,
                         recon_total, c_mse * 2.0f / (float)recon_total);
  }
  hipLaunchKernelGGL(is::loss_finish_kernel<7>, dim3(1), dim3(is::FIN_BLOCK), 0, st, partials, nparts, recon_total, mu,
                     logvar, d_mu, d_logvar, latent_total, logit, y, d_logit, batch, mode, pos_weight, c_pred, c_mse,
                     c_kld, (float*)nullptr, out, total);
  return is::launch_status(__func__);
}
