// "Combined attention" of the fusion head, forward + backward, in closed form.
//
// Reference (models/hybrid_models.py:344-347, models/layers.py:51-106): the fused vector
// c in R^T (T = 104, or 208 for the paired models) is treated as T scalar tokens,
// lifted by MultiHeadAttention(feature_dim F in {16, 32}, 8 heads, input_dim = 1) and the
// result is averaged over the F features:
//     q_i = wq c_i + bq, k_i = wk c_i + bk, v_i = wv c_i + bv          (F-vectors)
//     A_h = softmax_j( q_i,h . k_j,h / sqrt(d) ),  O_i = concat_h A_h v_.,h
//     z_i = mean_f ( Wc O_i + bc )_f
// Because every token is a scalar the block collapses exactly (softmax shift invariance
// removes the terms of the logits that do not depend on j):
//     logit_h(i, j) = gamma_h(i) c_j ,  gamma_h(i) = (A2_h c_i + C2_h) / sqrt(d)
//         A2_h = sum_e wq wk ,  C2_h = sum_e bq wk           (sums over the head's d features)
//     m_h(i)  = sum_j softmax_j(gamma_h(i) c_j) c_j           (attention-weighted mean of scalars)
//     z_i     = beta + sum_h alpha_h m_h(i)
//         wbar_f = sum_f' Wc[f'][f], alpha_h = (1/F) sum_e wbar wv, beta = (1/F)(wbar.bv + sum bc)
// One workgroup per graph evaluates this with T*8 softmax rows of length T (VALU only);
// the ~45 torch / hipBLASLt launches of the un-fused block (incl. K = 2 batched GEMMs)
// become three short kernels.  Gradients follow the same closed form; per-graph partial
// sums of (dA2, dC2, dalpha, dbeta) are reduced in a fixed order and pushed through the
// parameter chain rule by comb_attn_finish_kernel (deterministic).
#include "common.h"

namespace is {

constexpr int CA_HEADS = 8;
constexpr int CA_THREADS = 1024;   // one (token, head) item per thread for T <= 128: the block is pure latency (one workgroup per
                                   // graph, 128 graphs on 256 CUs), so the serial exp loops per thread are what it costs
constexpr int CA_TMAX = 256;
constexpr int CA_NSTAT = 5;   // gamma, mx, 1/den, m, second moment
constexpr int CA_PART = 3 * CA_HEADS + 1;

// The T tokens of a graph may come from up to four row-major pieces laid side by side ([x_gat | z_vae], or the four pieces of a
// (cancer, wild-type) pair): the kernels read them where they are -- no concatenation launch in front, no slice copies behind
// (the backward writes each piece's gradient into its own contiguous tensor).
constexpr int CA_MAX_PARTS = 4;
struct CaPart { const float* x; float* dx; int width, ld; };
struct CaParts { CaPart part[CA_MAX_PARTS]; int n; };

__device__ __forceinline__ float ca_load(const CaParts& P, int b, int j) {
  int off = 0;
#pragma unroll
  for (int p = 0; p < CA_MAX_PARTS; ++p) {
    if (p < P.n) {
      if (j < off + P.part[p].width) return P.part[p].x[(size_t)b * P.part[p].ld + (j - off)];
      off += P.part[p].width;
    }
  }
  return 0.0f;
}
__device__ __forceinline__ void ca_store_grad(const CaParts& P, int b, int j, float v) {
  int off = 0;
#pragma unroll
  for (int p = 0; p < CA_MAX_PARTS; ++p) {
    if (p < P.n) {
      if (j < off + P.part[p].width) { P.part[p].dx[(size_t)b * P.part[p].ld + (j - off)] = v; return; }
      off += P.part[p].width;
    }
  }
}

struct CaCoef {
  float A2[CA_HEADS], C2[CA_HEADS], alpha[CA_HEADS], beta;
};

template <int F>
__device__ __forceinline__ void ca_coefficients(CaCoef& co, const float* wq, const float* bq, const float* wk,
                                                const float* wv, const float* bv, const float* Wc, const float* bc,
                                                int tid) {
  constexpr int D = F / CA_HEADS;
  if (tid < CA_HEADS) {
    float a2 = 0.f, c2 = 0.f, al = 0.f;
    for (int e = 0; e < D; ++e) {
      const int f = tid * D + e;
      a2 += wq[f] * wk[f];
      c2 += bq[f] * wk[f];
      float wbar = 0.f;
      for (int fp = 0; fp < F; ++fp) wbar += Wc[fp * F + f];
      al += wbar * wv[f];
    }
    co.A2[tid] = a2; co.C2[tid] = c2; co.alpha[tid] = al / (float)F;
  }
  if (tid == CA_HEADS) {
    float b = 0.f;
    for (int f = 0; f < F; ++f) {
      float wbar = 0.f;
      for (int fp = 0; fp < F; ++fp) wbar += Wc[fp * F + f];
      b += wbar * bv[f] + bc[f];
    }
    co.beta = b / (float)F;
  }
}

template <int F>
__global__ __launch_bounds__(CA_THREADS) void comb_attn_fwd_kernel(
    CaParts X, const float* __restrict__ wq, const float* __restrict__ bq,
    const float* __restrict__ wk, const float* __restrict__ wv, const float* __restrict__ bv,
    const float* __restrict__ Wc, const float* __restrict__ bc, float* __restrict__ z,
    float* __restrict__ stats, int T) {
  constexpr int D = F / CA_HEADS;
  __shared__ float c[CA_TMAX];
  __shared__ CaCoef co;
  __shared__ float red[2][CA_THREADS / 64];
  const int tid = threadIdx.x, b = blockIdx.x;
  ca_coefficients<F>(co, wq, bq, wk, wv, bv, Wc, bc, tid);
  float lo = INFINITY, hi = -INFINITY;
  for (int j = tid; j < T; j += CA_THREADS) {
    const float v = ca_load(X, b, j);
    c[j] = v;
    lo = fminf(lo, v); hi = fmaxf(hi, v);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { lo = fminf(lo, __shfl_xor(lo, m, 64)); hi = fmaxf(hi, __shfl_xor(hi, m, 64)); }
  if ((tid & 63) == 0) { red[0][tid >> 6] = lo; red[1][tid >> 6] = hi; }
  __syncthreads();
  float cmin = red[0][0], cmax = red[1][0];
#pragma unroll
  for (int k = 1; k < CA_THREADS / 64; ++k) { cmin = fminf(cmin, red[0][k]); cmax = fmaxf(cmax, red[1][k]); }
  const float rs = rsqrtf((float)D);
  const int items = CA_HEADS * T;
  for (int w0 = 0; w0 < items; w0 += CA_THREADS) {
    const int w = w0 + tid;
    const bool on = w < items;
    const int i = on ? (w >> 3) : 0, hd = tid & 7;
    const float gamma = (co.A2[hd] * c[i] + co.C2[hd]) * rs;
    const float mx = gamma > 0.f ? gamma * cmax : gamma * cmin;
    float den = 0.f, num = 0.f, sq = 0.f;
    for (int j = 0; j < T; ++j) {
      const float cj = c[j];
      const float e = __expf(gamma * cj - mx);
      den += e; num += e * cj; sq += e * cj * cj;
    }
    const float inv = 1.0f / den;
    const float m = num * inv;
    float contrib = on ? co.alpha[hd] * m : 0.f;
    contrib += __shfl_xor(contrib, 1, 64);
    contrib += __shfl_xor(contrib, 2, 64);
    contrib += __shfl_xor(contrib, 4, 64);
    if (on) {
      if (hd == 0) z[(size_t)b * T + i] = co.beta + contrib;
      if (stats != nullptr) {
        float* st = stats + ((size_t)b * items + w) * CA_NSTAT;
        st[0] = gamma; st[1] = mx; st[2] = inv; st[3] = m; st[4] = sq * inv;
      }
    }
  }
}

template <int F>
__global__ __launch_bounds__(CA_THREADS) void comb_attn_bwd_kernel(
    CaParts X, const float* __restrict__ stats, const float* __restrict__ dz,
    const float* __restrict__ wq, const float* __restrict__ bq, const float* __restrict__ wk,
    const float* __restrict__ wv, const float* __restrict__ bv, const float* __restrict__ Wc,
    const float* __restrict__ bc, float* __restrict__ partials, int T) {
  constexpr int D = F / CA_HEADS;
  __shared__ float c[CA_TMAX];
  __shared__ float dxi[CA_TMAX];
  __shared__ float s_gamma[CA_HEADS * CA_TMAX], s_mx[CA_HEADS * CA_TMAX], s_inv[CA_HEADS * CA_TMAX],
      s_m[CA_HEADS * CA_TMAX], s_dm[CA_HEADS * CA_TMAX];
  __shared__ CaCoef co;
  __shared__ float acc[4][CA_THREADS];
  const int tid = threadIdx.x, b = blockIdx.x;
  ca_coefficients<F>(co, wq, bq, wk, wv, bv, Wc, bc, tid);
  for (int j = tid; j < T; j += CA_THREADS) c[j] = ca_load(X, b, j);
  __syncthreads();
  const float rs = rsqrtf((float)D);
  const int items = CA_HEADS * T, hd = tid & 7;
  float dA2 = 0.f, dC2 = 0.f, dAl = 0.f, dBe = 0.f;
  // ---- phase 1: per (token i, head) item ----
  for (int w0 = 0; w0 < items; w0 += CA_THREADS) {
    const int w = w0 + tid;
    const bool on = w < items;
    const int i = on ? (w >> 3) : 0;
    float direct = 0.f;
    if (on) {
      const float* st = stats + ((size_t)b * items + w) * CA_NSTAT;
      const float gamma = st[0], m = st[3], var = st[4] - st[3] * st[3];
      const float g = dz[(size_t)b * T + i];
      const float dm = g * co.alpha[hd];
      const float dgamma = dm * var;
      dA2 += dgamma * c[i] * rs;
      dC2 += dgamma * rs;
      dAl += g * m;
      if (hd == 0) dBe += g;
      direct = dgamma * co.A2[hd] * rs;        // through gamma_h(i) = (A2 c_i + C2)/sqrt(d)
      s_gamma[w] = gamma; s_mx[w] = st[1]; s_inv[w] = st[2]; s_m[w] = m; s_dm[w] = dm;
    }
    direct += __shfl_xor(direct, 1, 64);
    direct += __shfl_xor(direct, 2, 64);
    direct += __shfl_xor(direct, 4, 64);
    if (on && hd == 0) dxi[i] = direct;
  }
  acc[0][tid] = dA2; acc[1][tid] = dC2; acc[2][tid] = dAl; acc[3][tid] = dBe;
  __syncthreads();
  // per-head sums in a fixed order: threads with tid % 8 == h hold head h
  if (tid < 3 * CA_HEADS) {
    const int which = tid / CA_HEADS, h = tid % CA_HEADS;
    float s = 0.f;
    for (int t = h; t < CA_THREADS; t += CA_HEADS) s += acc[which][t];
    partials[(size_t)b * CA_PART + tid] = s;
  }
  if (tid == 3 * CA_HEADS) {
    float s = 0.f;
    for (int t = 0; t < CA_THREADS; t += CA_HEADS) s += acc[3][t];
    partials[(size_t)b * CA_PART + 3 * CA_HEADS] = s;
  }
  // ---- phase 2: per key token j: dc_j = sum_{i,h} dm p_{ih}(j) (1 + gamma (c_j - m)) + direct ----
  // the (i,h) items are split over SPLIT thread groups so that (nearly) all threads work; the group partial
  // sums are combined in a fixed order through LDS (deterministic)
  __syncthreads();   // acc[][] has been consumed by the per-head sums above
  {
    const int split = CA_THREADS / T >= 8 ? 8 : (CA_THREADS / T >= 4 ? 4 : CA_THREADS / T);      // T * split <= CA_THREADS
    const int grp = tid / T, j = tid - grp * T;
    float a = 0.f;
    if (grp < split) {
      const float cj = c[j];
      const int chunk = (items + split - 1) / split;
      const int w_lo = grp * chunk, w_hi = min(items, w_lo + chunk);
      for (int w = w_lo; w < w_hi; ++w) {
        const float gam = s_gamma[w];
        const float p = __expf(gam * cj - s_mx[w]) * s_inv[w];
        a += s_dm[w] * p * (1.0f + gam * (cj - s_m[w]));
      }
      (&acc[0][0])[grp * CA_TMAX + j] = a;      // acc viewed as [8][CA_TMAX]
    }
    __syncthreads();
    if (tid < T) {
      float v = dxi[tid];
      for (int k = 0; k < split; ++k) v += (&acc[0][0])[k * CA_TMAX + tid];
      ca_store_grad(X, b, tid, v);
    }
  }
}

// sums the per-graph partials and applies the parameter chain rule; one workgroup.
// out layout (floats): dwq[F] dbq[F] dwk[F] dbk[F] dwv[F] dbv[F] dWc[F*F] dbc[F]
template <int F>
__global__ __launch_bounds__(256) void comb_attn_finish_kernel(
    const float* __restrict__ partials, int B, const float* __restrict__ wq, const float* __restrict__ bq,
    const float* __restrict__ wk, const float* __restrict__ wv, const float* __restrict__ bv,
    const float* __restrict__ Wc, float* __restrict__ out) {
  constexpr int D = F / CA_HEADS;
  __shared__ float tot[CA_PART];
  __shared__ float dwbar[F];
  __shared__ float slice[8][32];
  const int tid = threadIdx.x;
  {
    // 8 slices of the batch are summed concurrently (independent loads), then combined in a fixed order
    const int k = tid & 31, sl = tid >> 5;
    float s = 0.f;
    if (k < CA_PART)
      for (int b = sl; b < B; b += 8) s += partials[(size_t)b * CA_PART + k];
    slice[sl][k] = s;
  }
  __syncthreads();
  if (tid < CA_PART) {
    float s = 0.f;
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) s += slice[sl][tid];
    tot[tid] = s;
  }
  __syncthreads();
  const float dBeta = tot[3 * CA_HEADS];
  if (tid < F) {
    const int f = tid, h = f / D;
    const float dA2 = tot[h], dC2 = tot[CA_HEADS + h], dAl = tot[2 * CA_HEADS + h];
    float wbar = 0.f;
    for (int fp = 0; fp < F; ++fp) wbar += Wc[fp * F + f];
    out[0 * F + f] = dA2 * wk[f];                       // dwq
    out[1 * F + f] = dC2 * wk[f];                       // dbq
    out[2 * F + f] = dA2 * wq[f] + dC2 * bq[f];         // dwk
    out[3 * F + f] = 0.0f;                              // dbk (logits are shift invariant in the key bias)
    out[4 * F + f] = dAl * wbar / (float)F;             // dwv
    out[5 * F + f] = dBeta * wbar / (float)F;           // dbv
    dwbar[f] = (dAl * wv[f] + dBeta * bv[f]) / (float)F;
    out[6 * F + F * F + f] = dBeta / (float)F;          // dbc
  }
  __syncthreads();
  for (int idx = tid; idx < F * F; idx += 256) out[6 * F + idx] = dwbar[idx % F];   // dWc[f'][f] = dwbar[f]
}

}  // namespace is

extern "C" int is_comb_attn_stats_floats(int B, int T) { return B * is::CA_HEADS * T * is::CA_NSTAT; }
extern "C" int is_comb_attn_partials_floats(int B) { return B * is::CA_PART; }
extern "C" int is_comb_attn_grad_floats(int F) { return 7 * F + F * F; }

static int ca_parts_from(is::CaParts& P, const void* parts, int nparts, int T, bool need_dx) {
  if (nparts <= 0 || nparts > is::CA_MAX_PARTS) return -22;
  const is::CaPart* src = static_cast<const is::CaPart*>(parts);
  int total = 0;
  for (int p = 0; p < nparts; ++p) {
    P.part[p] = src[p];
    if (src[p].x == nullptr || src[p].width <= 0 || src[p].ld < src[p].width || (need_dx && src[p].dx == nullptr)) return -22;
    total += src[p].width;
  }
  P.n = nparts;
  return total == T ? 0 : -22;
}

// parts: host array of nparts (<= 4) records { const float* x; float* dx; int width, ld; } -- the T = sum(width) scalar tokens of
// graph b are the rows b of the pieces side by side (dx: the piece's gradient, written by the backward; unused in the forward)
extern "C" int is_comb_attn_fwd(const void* parts, int nparts, const float* wq, const float* bq, const float* wk, const float* wv,
                                const float* bv, const float* Wc, const float* bc, float* z, float* stats, int B,
                                int T, int F, void* stream) {
  if (B <= 0) return 0;
  if (T <= 0 || T > is::CA_TMAX || (F != 16 && F != 32)) return -22;
  is::CaParts P;
  if (ca_parts_from(P, parts, nparts, T, false) != 0) return -22;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (F == 16) hipLaunchKernelGGL(is::comb_attn_fwd_kernel<16>, dim3(B), dim3(is::CA_THREADS), 0, st, P, wq, bq, wk, wv, bv, Wc, bc, z, stats, T);
  else hipLaunchKernelGGL(is::comb_attn_fwd_kernel<32>, dim3(B), dim3(is::CA_THREADS), 0, st, P, wq, bq, wk, wv, bv, Wc, bc, z, stats, T);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int is_comb_attn_bwd(const void* parts, int nparts, const float* stats, const float* dz, const float* wq, const float* bq,
                                const float* wk, const float* wv, const float* bv, const float* Wc, const float* bc,
                                float* partials, float* grads, int B, int T, int F, void* stream) {
  if (B <= 0) return 0;
  if (T <= 0 || T > is::CA_TMAX || (F != 16 && F != 32)) return -22;
  is::CaParts P;
  if (ca_parts_from(P, parts, nparts, T, true) != 0) return -22;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (F == 16) {
    hipLaunchKernelGGL(is::comb_attn_bwd_kernel<16>, dim3(B), dim3(is::CA_THREADS), 0, st, P, stats, dz, wq, bq, wk, wv, bv, Wc, bc, partials, T);
    hipLaunchKernelGGL(is::comb_attn_finish_kernel<16>, dim3(1), dim3(256), 0, st, partials, B, wq, bq, wk, wv, bv, Wc, grads);
  } else {
    hipLaunchKernelGGL(is::comb_attn_bwd_kernel<32>, dim3(B), dim3(is::CA_THREADS), 0, st, P, stats, dz, wq, bq, wk, wv, bv, Wc, bc, partials, T);
    hipLaunchKernelGGL(is::comb_attn_finish_kernel<32>, dim3(1), dim3(256), 0, st, partials, B, wq, bq, wk, wv, bv, Wc, grads);
  }
  return hipGetLastError() == hipSuccess ? 0 : -5;
}
