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

#ifndef CA_MOMENTS_F64
#define CA_MOMENTS_F64 1
#endif
constexpr int CA_HEADS = 8;
constexpr int CA_THREADS = 1024;   // one (token, head) item per thread for T <= 128: the block is pure latency (one workgroup per
                                   // graph, 128 graphs on 256 CUs), so the serial exp loops per thread are what it costs
constexpr int CA_TMAX = 256;
constexpr int CA_NSTAT = 5;   // gamma, mx, 1/den, m, variance (second moment about the token mean - centred mean^2)
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

// The 25 scalars above from the block's parameters.  Every parameter element is fetched by its own thread in one batch of loads
// (F * F + 6 F elements over 1024 threads: one or two memory round trips), the column sums of Wc and the head sums then come from LDS -- the
// earlier form let 9 threads walk the matrices with serial loads (the last one 256 + of them), a few microseconds of pure
// latency at the top of two kernels on the step's critical chain.  Contains two barriers; every sum keeps its ascending order.
template <int F>
struct CaCoefScratch { float Wc[F * F]; float v[6][F]; float wbar[F]; };      // v: wq bq wk wv bv bc

template <int F>
__device__ __forceinline__ void ca_coefficients(CaCoef& co, CaCoefScratch<F>& sc, const float* wq, const float* bq, const float* wk,
                                                const float* wv, const float* bv, const float* Wc, const float* bc,
                                                int tid) {
  constexpr int D = F / CA_HEADS;
  for (int i = tid; i < F * F + 6 * F; i += CA_THREADS) {      // F = 16: one pass; F = 32: two
    if (i < F * F) {
      sc.Wc[i] = Wc[i];
    } else {
      const int k = (i - F * F) / F, f = (i - F * F) % F;
      const float* src = k == 0 ? wq : (k == 1 ? bq : (k == 2 ? wk : (k == 3 ? wv : (k == 4 ? bv : bc))));
      sc.v[k][f] = src[f];
    }
  }
  __syncthreads();
  if (tid < F) {
    float wbar = 0.f;
    for (int fp = 0; fp < F; ++fp) wbar += sc.Wc[fp * F + tid];
    sc.wbar[tid] = wbar;
  }
  __syncthreads();
  if (tid < CA_HEADS) {
    float a2 = 0.f, c2 = 0.f, al = 0.f;
    for (int e = 0; e < D; ++e) {
      const int f = tid * D + e;
      a2 += sc.v[0][f] * sc.v[2][f];
      c2 += sc.v[1][f] * sc.v[2][f];
      al += sc.wbar[f] * sc.v[3][f];
    }
    co.A2[tid] = a2; co.C2[tid] = c2; co.alpha[tid] = al / (float)F;
  }
  if (tid == CA_HEADS) {
    float b = 0.f;
    for (int f = 0; f < F; ++f) b += sc.wbar[f] * sc.v[4][f] + sc.v[5][f];
    co.beta = b / (float)F;
  }
}

// Mean of the T token values c[0 .. T) (LDS), in a fixed order: the reference point of the centred moments below.  Called by all
// threads after a barrier that made c visible; contains one barrier.  red: CA_THREADS / 64 floats of LDS.
__device__ __forceinline__ float ca_token_mean(const float* c, float* red, int T, int tid) {
  float s = 0.f;
  for (int j = tid; j < T; j += CA_THREADS) s += c[j];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < CA_THREADS / 64; ++k) tot += red[k];
  return tot / (float)T;
}

// The classifier behind the combined attention (models/hybrid_models.py:288-295: Flatten, Linear(T, hid), ReLU, Dropout,
// Linear(hid, out)) rides on the same launches: y = act2(W2 (mask * ReLU(W1 z + b1)) + b2) per sample, the arithmetic of
// csrc/mlp_head.hip (k ascending from the bias).  hid <= 32, out <= 64.
constexpr int CA_CLS_HID = 32;
struct CaCls {
  const float *W1, *b1, *W2, *b2, *mask;      // W1 [hid, T], W2 [out, hid]; mask [B, hid] (scaled keep-mask) or NULL
  float *a1, *y;                              // forward outputs: ReLU(W1 z + b1) [B, hid] (may be NULL), y [B, out]
  const float *a1_in, *y_in, *gy, *z_in;      // backward inputs: the saved a1 / y / z, gy [B, out]
  float* gcls;                                // backward output: dW1 [hid * T] | db1 [hid] | dW2 [out * hid] | db2 [out]
  int hid, out, act2, B;
};

template <int F, bool CLS>
__global__ __launch_bounds__(CA_THREADS) void comb_attn_fwd_kernel(
    CaParts X, const float* __restrict__ wq, const float* __restrict__ bq,
    const float* __restrict__ wk, const float* __restrict__ wv, const float* __restrict__ bv,
    const float* __restrict__ Wc, const float* __restrict__ bc, float* __restrict__ z,
    float* __restrict__ stats, int T, CaCls cls) {
  constexpr int D = F / CA_HEADS;
  __shared__ float c[CA_TMAX];
  __shared__ CaCoef co;
  __shared__ CaCoefScratch<F> csc;
  __shared__ float red[3][CA_THREADS / 64];
  __shared__ float w1t[CLS ? CA_TMAX * (CA_CLS_HID + 1) : 1];      // W1 transposed [T][hid + 1]
  __shared__ float zs[CLS ? CA_TMAX : 1], hs[CLS ? CA_CLS_HID : 1];
  const int tid = threadIdx.x, b = blockIdx.x;
  if constexpr (CLS) {      // in flight while the attention rows are evaluated
    const int ldw = cls.hid + 1;
    for (int i0 = tid; i0 < cls.hid * T; i0 += 4 * CA_THREADS) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int i = i0 + u * CA_THREADS; v[u] = (i < cls.hid * T) ? cls.W1[i] : 0.0f; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int i = i0 + u * CA_THREADS; if (i < cls.hid * T) w1t[(i % T) * ldw + i / T] = v[u]; }
    }
  }
  float lo = INFINITY, hi = -INFINITY;
  for (int j = tid; j < T; j += CA_THREADS) {      // (issued before the coefficients' loads are waited for)
    const float v = ca_load(X, b, j);
    c[j] = v;
    lo = fminf(lo, v); hi = fmaxf(hi, v);
  }
  ca_coefficients<F>(co, csc, wq, bq, wk, wv, bv, Wc, bc, tid);      // (its barriers also publish c)
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { lo = fminf(lo, __shfl_xor(lo, m, 64)); hi = fmaxf(hi, __shfl_xor(hi, m, 64)); }
  if ((tid & 63) == 0) { red[0][tid >> 6] = lo; red[1][tid >> 6] = hi; }
  const float c0 = ca_token_mean(c, red[2], T, tid);      // (its barrier also publishes red[0], red[1] and co)
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
    // first and second moment of the row's softmax ABOUT THE TOKEN MEAN c0: the backward needs the variance sq / den - m^2, and
    // formed from raw moments it loses the digits |c0|^2 / var has (the values of a graph share an offset)
    // The long sums of the head's kernels -- the classifier's dot products, the backward's sums over the (token, head) items -- run
    // in fp64: the head's backward amplifies the round-off of what it is given ~ 100 x (its gradients are small differences), and
    // fp32 chains of 104 - 832 terms left the logit 3.8 x and the gradients at the head's inputs 2.7 x further from the fp64 oracle
    // than torch's blocked fp32 sums -- the "systematic factor" of the full-size gradient tests in rounds 3 - 4
    // (tests/tools/grad_error_probe_model.py, HISTORY.md 7.10).  The kernels are latency-bound: the wider adds cost little.
    // The softmax moments below (the saved variance is what the backward multiplies its gradients with): CA_MOMENTS_F64 = 0 plain
    // fp32 chains (rounds 1 - 3), 1 (default) blocks of 8 in fp32 + block sums in fp64, 2 every term in fp64.
    double den_d = 0.0, num_d = 0.0, sq_d = 0.0;
#if CA_MOMENTS_F64 == 2
    for (int j = 0; j < T; ++j) {      // every term into the fp64 sums (comb_attn_cls_fwd 21.8 -> 27 us)
      const float cj = c[j];
      const float e = __expf(gamma * cj - mx);
      const float dj = cj - c0;
      const double ed = (double)e * (double)dj;
      den_d += (double)e; num_d += ed; sq_d += ed * (double)dj;
    }
#elif CA_MOMENTS_F64 == 1
    // blocks of 8 terms summed in fp32, the block sums in fp64: the chain a term's round-off travels through is 8 long, not T
    int j = 0;
    for (; j + 8 <= T; j += 8) {
      float d8 = 0.f, n8 = 0.f, s8 = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float cj = c[j + u];
        const float e = __expf(gamma * cj - mx);
        const float dj = cj - c0;
        const float t = e * dj;
        d8 += e; n8 += t; s8 += t * dj;
      }
      den_d += (double)d8; num_d += (double)n8; sq_d += (double)s8;
    }
    for (; j < T; ++j) {
      const float cj = c[j];
      const float e = __expf(gamma * cj - mx);
      const float dj = cj - c0;
      const float t = e * dj;
      den_d += (double)e; num_d += (double)t; sq_d += (double)(t * dj);
    }
#else
    {
      float den = 0.f, num = 0.f, sq = 0.f;
      for (int j = 0; j < T; ++j) {
        const float cj = c[j];
        const float e = __expf(gamma * cj - mx);
        const float dj = cj - c0;
        den += e; num += e * dj; sq += e * dj * dj;
      }
      den_d = (double)den; num_d = (double)num; sq_d = (double)sq;
    }
#endif
    const double inv_d = 1.0 / den_d, mc_d = num_d * inv_d;
    const float inv = (float)inv_d;
    const float m = (float)((double)c0 + mc_d);      // (mc: the centred mean)
    float contrib = on ? co.alpha[hd] * m : 0.f;
    contrib += __shfl_xor(contrib, 1, 64);
    contrib += __shfl_xor(contrib, 2, 64);
    contrib += __shfl_xor(contrib, 4, 64);
    if (on) {
      if (hd == 0) {
        z[(size_t)b * T + i] = co.beta + contrib;
        if constexpr (CLS) zs[i] = co.beta + contrib;
      }
      if (stats != nullptr) {
        float* st = stats + ((size_t)b * items + w) * CA_NSTAT;
        st[0] = gamma; st[1] = mx; st[2] = inv; st[3] = m; st[4] = (float)(sq_d * inv_d - mc_d * mc_d);      // [4]: the row's variance
      }
    }
  }
  if constexpr (CLS) {
    __syncthreads();
    const int ldw = cls.hid + 1;
    if (tid < cls.hid) {
      // four interleaved partial sums (k mod 4), combined as ((s0 + s1) + (s2 + s3)) + bias: a quarter of the dependent chain
      // (the same order in csrc/mlp_head.hip: the two kernels give the same bits)
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      int k = 0;
      for (; k + 4 <= T; k += 4) {
        s0 += (double)w1t[k * ldw + tid] * (double)zs[k];
        s1 += (double)w1t[(k + 1) * ldw + tid] * (double)zs[k + 1];
        s2 += (double)w1t[(k + 2) * ldw + tid] * (double)zs[k + 2];
        s3 += (double)w1t[(k + 3) * ldw + tid] * (double)zs[k + 3];
      }
      for (; k < T; ++k) s0 += (double)w1t[k * ldw + tid] * (double)zs[k];
      const double accd = ((s0 + s1) + (s2 + s3)) + (double)cls.b1[tid];
      float acc = fmaxf((float)accd, 0.0f);
      if (cls.a1 != nullptr) cls.a1[(size_t)b * cls.hid + tid] = acc;
      if (cls.mask != nullptr) acc *= cls.mask[(size_t)b * cls.hid + tid];
      hs[tid] = acc;
    }
    __syncthreads();
    if (tid < cls.out) {
      double accd = (double)cls.b2[tid];
      for (int h = 0; h < cls.hid; ++h) accd += (double)cls.W2[tid * cls.hid + h] * (double)hs[h];
      float acc = (float)accd;
      if (cls.act2 == 1) acc = fmaxf(acc, 0.0f);
      cls.y[(size_t)b * cls.out + tid] = acc;
    }
  }
}

// parameter gradients of the classifier, one workgroup: samples in ascending order (deterministic)
__device__ __forceinline__ void ca_cls_wgrad(float* lds, const CaCls& cls, int T, int tid) {
  // samples per chunk.  This workgroup is a chain of B / S trips of (global loads -> barrier -> pre-activation gradients ->
  // barrier -> products): with S = 8 its 16 trips at B = 128 outlasted the per-sample workgroups of the launch by ~ 15 us -- on
  // the step's critical chain.  16 samples per trip (32 spill 250 registers into the per-sample path); every sum still runs over
  // the samples in ascending order: same bits.
  constexpr int S = 16;
  static_assert(S * (CA_TMAX + 2 * CA_CLS_HID + 64) + 64 * CA_CLS_HID <= 5 * CA_HEADS * CA_TMAX + 4 * CA_THREADS, "chunk buffers + W2 must fit the shared block");
  float* zc = lds;                                       // [S][T]
  float* ghc = zc + S * CA_TMAX;                         // [S][hid]   d loss / d pre-activation 1
  float* hc = ghc + S * CA_CLS_HID;                      // [S][hid]   hid = a1 * mask
  float* g2c = hc + S * CA_CLS_HID;                      // [S][out]
  float* w2s = g2c + S * 64;                             // [out][hid]  W2, staged once (it was re-read from global every trip)
  const int hid = cls.hid, out = cls.out;
  const int n1 = hid * T, n2 = out * hid;
  const int hgroups = (hid + 3) / 4;                     // dW1 rows in groups of four
  constexpr int PER1 = (CA_CLS_HID * CA_TMAX + CA_THREADS - 1) / CA_THREADS;                    // dW1 entries per thread
  static_assert(PER1 % 4 == 0, "dW1 entries are owned four rows at a time");
  constexpr int PER2 = (CA_CLS_HID + 64 * CA_CLS_HID + 64 + CA_THREADS - 1) / CA_THREADS;       // db1 | dW2 | db2 entries
  constexpr int ZPT = (S * CA_TMAX + CA_THREADS - 1) / CA_THREADS;                              // z entries per thread and chunk
  static_assert(S * 64 <= CA_THREADS && S * CA_CLS_HID <= CA_THREADS, "one g2 / a1 / mask entry per thread and chunk");
  float acc1[PER1], acc2[PER2];
#pragma unroll
  for (int k = 0; k < PER1; ++k) acc1[k] = 0.0f;
#pragma unroll
  for (int k = 0; k < PER2; ++k) acc2[k] = 0.0f;
  for (int i = tid; i < n2; i += CA_THREADS) w2s[i] = cls.W2[i];
  // the chunk's rows travel global -> registers -> LDS, the NEXT chunk's loads in flight while this one is reduced
  float rz[ZPT], rg = 0.0f, ra = 0.0f, rm = 1.0f;
  auto fetch = [&](int s0) {
#pragma unroll
    for (int u = 0; u < ZPT; ++u) {
      const int i = tid + u * CA_THREADS, s = s0 + i / T;
      rz[u] = (i < S * T && s < cls.B) ? cls.z_in[(size_t)s * T + i % T] : 0.0f;
    }
    rg = 0.0f;
    if (tid < S * out) {
      const int s = s0 + tid / out, o = tid % out;
      if (s < cls.B) {
        rg = cls.gy[(size_t)s * out + o];
        if (cls.act2 == 1 && !(cls.y_in[(size_t)s * out + o] > 0.0f)) rg = 0.0f;
      }
    }
    ra = 0.0f; rm = 1.0f;
    if (tid < S * hid) {
      const int s = s0 + tid / hid, h = tid % hid;
      if (s < cls.B) {
        ra = cls.a1_in[(size_t)s * hid + h];
        if (cls.mask != nullptr) rm = cls.mask[(size_t)s * hid + h];
      }
    }
  };
  fetch(0);
  for (int s0 = 0; s0 < cls.B; s0 += S) {
    __syncthreads();      // the previous chunk's products are done with the buffers (first trip: w2s is staged)
#pragma unroll
    for (int u = 0; u < ZPT; ++u) {
      const int i = tid + u * CA_THREADS;
      if (i < S * T) zc[(i / T) * CA_TMAX + i % T] = rz[u];
    }
    if (tid < S * out) g2c[tid] = rg;
    const float a_own = ra, m_own = rm;
    __syncthreads();
    if (s0 + S < cls.B) fetch(s0 + S);
    if (tid < S * hid) {
      const int sl = tid / hid, h = tid % hid;
      float g = 0.0f;
      if (s0 + sl < cls.B) {
        for (int o = 0; o < out; ++o) g += g2c[sl * out + o] * w2s[o * hid + h];
        g *= m_own;
        if (!(a_own > 0.0f)) g = 0.0f;
      }
      ghc[tid] = g;
      hc[tid] = (s0 + sl < cls.B) ? a_own * m_own : 0.0f;
    }
    __syncthreads();
    // dW1[h][kk] += sum over the chunk of ghc[sl][h] * zc[sl][kk]: a thread owns column kk of FOUR consecutive rows h, so one
    // read of zc feeds four products and the four ghc reads are the same address across the wave (one entry per thread and
    // read pair made this phase -- 16 waves on one CU -- LDS-bandwidth bound: 2 us per trip)
#pragma unroll
    for (int k = 0; k < PER1 / 4; ++k) {
      const int item = tid + k * CA_THREADS;      // (row group, column)
      if (item < hgroups * T) {
        const int h0 = 4 * (item / T), kk = item % T;
        const int o1 = min(h0 + 1, hid - 1), o2 = min(h0 + 2, hid - 1), o3 = min(h0 + 3, hid - 1);
        float a0 = acc1[4 * k], a1 = acc1[4 * k + 1], a2 = acc1[4 * k + 2], a3 = acc1[4 * k + 3];
#pragma unroll 4
        for (int sl = 0; sl < S; ++sl) {
          const float zv = zc[sl * CA_TMAX + kk];
          const float* gr = ghc + sl * hid;
          a0 += gr[h0] * zv; a1 += gr[o1] * zv; a2 += gr[o2] * zv; a3 += gr[o3] * zv;
        }
        acc1[4 * k] = a0; acc1[4 * k + 1] = a1; acc1[4 * k + 2] = a2; acc1[4 * k + 3] = a3;
      }
    }
#pragma unroll
    for (int k = 0; k < PER2; ++k) {
      const int idx = tid + k * CA_THREADS;      // db1 [hid] | dW2 [out * hid] | db2 [out]
      if (idx < hid + n2 + out) {
        // one form for the three blocks: sum over the chunk of A[sl] * Bv[sl] with Bv = 1 for the bias sums
        const bool is_b1 = idx < hid, is_w2 = !is_b1 && idx < hid + n2;
        const int o = is_w2 ? (idx - hid) / hid : (is_b1 ? 0 : idx - hid - n2);
        const int h = is_b1 ? idx : (is_w2 ? (idx - hid) % hid : 0);
        float v = acc2[k];
        for (int sl = 0; sl < S; ++sl) {
          const float a = is_b1 ? ghc[sl * hid + h] : g2c[sl * out + o];
          const float bvv = is_w2 ? hc[sl * hid + h] : 1.0f;
          v += a * bvv;
        }
        acc2[k] = v;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < PER1 / 4; ++k) {
    const int item = tid + k * CA_THREADS;
    if (item < hgroups * T) {
      const int h0 = 4 * (item / T), kk = item % T;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (h0 + u < hid) cls.gcls[(h0 + u) * T + kk] = acc1[4 * k + u];
    }
  }
#pragma unroll
  for (int k = 0; k < PER2; ++k) {
    const int idx = tid + k * CA_THREADS;
    if (idx < hid + n2 + out) cls.gcls[n1 + idx] = acc2[k];
  }
}

template <int F, bool CLS>
__global__ __launch_bounds__(CA_THREADS) void comb_attn_bwd_kernel(
    CaParts X, const float* __restrict__ stats, const float* __restrict__ dz,
    const float* __restrict__ wq, const float* __restrict__ bq, const float* __restrict__ wk,
    const float* __restrict__ wv, const float* __restrict__ bv, const float* __restrict__ Wc,
    const float* __restrict__ bc, float* __restrict__ partials, int T, CaCls cls) {
  constexpr int D = F / CA_HEADS;
  __shared__ float c[CA_TMAX];
  __shared__ float dxi[CA_TMAX];
  // s_gamma | s_mx | s_inv | s_m | s_dm  (5 x [heads * T]) followed by acc [4][threads]: one block, so that the classifier's
  // prologue (W1 rows) and the parameter-gradient workgroup (sample chunks) can use it before / instead of the phases below
  __shared__ float big[5 * CA_HEADS * CA_TMAX + 4 * CA_THREADS];
  float* s_gamma = big;
  float* s_mx = s_gamma + CA_HEADS * CA_TMAX;
  float* s_inv = s_mx + CA_HEADS * CA_TMAX;
  float* s_m = s_inv + CA_HEADS * CA_TMAX;
  float* s_dm = s_m + CA_HEADS * CA_TMAX;
  float (*acc)[CA_THREADS] = reinterpret_cast<float (*)[CA_THREADS]>(s_dm + CA_HEADS * CA_TMAX);
  __shared__ float dzs[CLS ? CA_TMAX : 1], ghs[CLS ? CA_CLS_HID : 1];
  __shared__ CaCoef co;
  __shared__ CaCoefScratch<F> csc;
  const int tid = threadIdx.x, b = blockIdx.x;
  if constexpr (CLS) {
    if (b == cls.B) {      // the extra workgroup: the classifier's parameter gradients
      ca_cls_wgrad(big, cls, T, tid);
      return;
    }
    // d z = W1^T (mask * ReLU' * (W2^T g2)): the gradient the classifier hands to the attention block, never in HBM
    const int ldw = T + 1;
    float* w1s = big;                                   // [hid][T + 1]
    for (int i0 = tid; i0 < cls.hid * T; i0 += 4 * CA_THREADS) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int i = i0 + u * CA_THREADS; v[u] = (i < cls.hid * T) ? cls.W1[i] : 0.0f; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int i = i0 + u * CA_THREADS; if (i < cls.hid * T) w1s[(i / T) * ldw + i % T] = v[u]; }
    }
    if (tid < cls.hid) {
      const float a = cls.a1_in[(size_t)b * cls.hid + tid];
      const float m = cls.mask != nullptr ? cls.mask[(size_t)b * cls.hid + tid] : 1.0f;
      double gd = 0.0;
      for (int o = 0; o < cls.out; ++o) {
        float g2 = cls.gy[(size_t)b * cls.out + o];
        if (cls.act2 == 1 && !(cls.y_in[(size_t)b * cls.out + o] > 0.0f)) g2 = 0.0f;
        gd += (double)g2 * (double)cls.W2[o * cls.hid + tid];
      }
      float g = (float)gd;
      g *= m;
      if (!(a > 0.0f)) g = 0.0f;
      ghs[tid] = g;
    }
    __syncthreads();
    for (int j = tid; j < T; j += CA_THREADS) {
      double v0 = 0.0, v1 = 0.0;
      int h = 0;
      for (; h + 2 <= cls.hid; h += 2) {
        v0 += (double)ghs[h] * (double)w1s[h * ldw + j];
        v1 += (double)ghs[h + 1] * (double)w1s[(h + 1) * ldw + j];
      }
      if (h < cls.hid) v0 += (double)ghs[h] * (double)w1s[h * ldw + j];
      dzs[j] = (float)(v0 + v1);
    }
    __syncthreads();      // w1s is dead: the phases below reuse the block
  }
  for (int j = tid; j < T; j += CA_THREADS) c[j] = ca_load(X, b, j);
  ca_coefficients<F>(co, csc, wq, bq, wk, wv, bv, Wc, bc, tid);
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
      const float gamma = st[0], m = st[3], var = st[4];
      const float g = CLS ? dzs[i] : dz[(size_t)b * T + i];
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
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;      // (four interleaved partial sums: a quarter of the dependent chain)
    for (int t = h; t < CA_THREADS; t += 4 * CA_HEADS) {
      s0 += (double)acc[which][t]; s1 += (double)acc[which][t + CA_HEADS];
      s2 += (double)acc[which][t + 2 * CA_HEADS]; s3 += (double)acc[which][t + 3 * CA_HEADS];
    }
    partials[(size_t)b * CA_PART + tid] = (float)((s0 + s1) + (s2 + s3));
  }
  if (tid == 3 * CA_HEADS) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int t = 0; t < CA_THREADS; t += 4 * CA_HEADS) {
      s0 += (double)acc[3][t]; s1 += (double)acc[3][t + CA_HEADS];
      s2 += (double)acc[3][t + 2 * CA_HEADS]; s3 += (double)acc[3][t + 3 * CA_HEADS];
    }
    partials[(size_t)b * CA_PART + 3 * CA_HEADS] = (float)((s0 + s1) + (s2 + s3));
  }
  // ---- phase 2: per key token j: dc_j = sum_{i,h} dm p_{ih}(j) (1 + gamma (c_j - m)) + direct ----
  // the (i,h) items are split over SPLIT thread groups so that (nearly) all threads work; the group partial
  // sums are combined in a fixed order through LDS (deterministic)
  __syncthreads();   // acc[][] has been consumed by the per-head sums above
  {
    const int split = CA_THREADS / T >= 8 ? 8 : (CA_THREADS / T >= 4 ? 4 : CA_THREADS / T);      // T * split <= CA_THREADS
    const int grp = tid / T, j = tid - grp * T;
    double* accd = reinterpret_cast<double*>(&acc[0][0]);      // viewed as [8][CA_TMAX] doubles (4 * CA_THREADS floats: 16 KB)
    static_assert(8 * CA_TMAX * sizeof(double) <= 4 * CA_THREADS * sizeof(float), "the group partials do not fit acc");
    if (grp < split) {
      double a = 0.0;
      const float cj = c[j];
      const int chunk = (items + split - 1) / split;
      const int w_lo = grp * chunk, w_hi = min(items, w_lo + chunk);
      int w = w_lo;
      for (; w + 8 <= w_hi; w += 8) {      // blocks of 8 terms in fp32, the block sums in fp64 (as the forward's moments)
        float a8 = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float gam = s_gamma[w + u];
          const float p = __expf(gam * cj - s_mx[w + u]) * s_inv[w + u];
          a8 += s_dm[w + u] * p * (1.0f + gam * (cj - s_m[w + u]));
        }
        a += (double)a8;
      }
      for (; w < w_hi; ++w) {
        const float gam = s_gamma[w];
        const float p = __expf(gam * cj - s_mx[w]) * s_inv[w];
        a += (double)(s_dm[w] * p * (1.0f + gam * (cj - s_m[w])));
      }
      accd[grp * CA_TMAX + j] = a;
    }
    __syncthreads();
    if (tid < T) {
      double v = (double)dxi[tid];
      for (int k = 0; k < split; ++k) v += accd[k * CA_TMAX + tid];
      ca_store_grad(X, b, tid, (float)v);
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
  if (nparts <= 0 || nparts > is::CA_MAX_PARTS) return is::fail(__func__, -22);
  const is::CaPart* src = static_cast<const is::CaPart*>(parts);
  int total = 0;
  for (int p = 0; p < nparts; ++p) {
    P.part[p] = src[p];
    if (src[p].x == nullptr || src[p].width <= 0 || src[p].ld < src[p].width || (need_dx && src[p].dx == nullptr)) return is::fail(__func__, -22);
    total += src[p].width;
  }
  P.n = nparts;
  return total == T ? 0 : is::fail(__func__, -22);
}

// parts: host array of nparts (<= 4) records { const float* x; float* dx; int width, ld; } -- the T = sum(width) scalar tokens of
// graph b are the rows b of the pieces side by side (dx: the piece's gradient, written by the backward; unused in the forward)
extern "C" int is_comb_attn_fwd(const void* parts, int nparts, const float* wq, const float* bq, const float* wk, const float* wv,
                                const float* bv, const float* Wc, const float* bc, float* z, float* stats, int B,
                                int T, int F, void* stream) {
  if (B <= 0) return 0;
  if (T <= 0 || T > is::CA_TMAX || (F != 16 && F != 32)) return is::fail(__func__, -22);
  is::CaParts P;
  if (ca_parts_from(P, parts, nparts, T, false) != 0) return is::fail(__func__, -22);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const is::CaCls none{};
  if (F == 16) hipLaunchKernelGGL((is::comb_attn_fwd_kernel<16, false>), dim3(B), dim3(is::CA_THREADS), 0, st, P, wq, bq, wk, wv, bv, Wc, bc, z, stats, T, none);
  else hipLaunchKernelGGL((is::comb_attn_fwd_kernel<32, false>), dim3(B), dim3(is::CA_THREADS), 0, st, P, wq, bq, wk, wv, bv, Wc, bc, z, stats, T, none);
  return is::launch_status(__func__);
}

extern "C" int is_comb_attn_bwd(const void* parts, int nparts, const float* stats, const float* dz, const float* wq, const float* bq,
                                const float* wk, const float* wv, const float* bv, const float* Wc, const float* bc,
                                float* partials, float* grads, int B, int T, int F, void* stream) {
  if (B <= 0) return 0;
  if (T <= 0 || T > is::CA_TMAX || (F != 16 && F != 32)) return is::fail(__func__, -22);
  is::CaParts P;
  if (ca_parts_from(P, parts, nparts, T, true) != 0) return is::fail(__func__, -22);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const is::CaCls none{};
  if (F == 16) {
    hipLaunchKernelGGL((is::comb_attn_bwd_kernel<16, false>), dim3(B), dim3(is::CA_THREADS), 0, st, P, stats, dz, wq, bq, wk, wv, bv, Wc, bc, partials, T, none);
    hipLaunchKernelGGL(is::comb_attn_finish_kernel<16>, dim3(1), dim3(256), 0, st, partials, B, wq, bq, wk, wv, bv, Wc, grads);
  } else {
    hipLaunchKernelGGL((is::comb_attn_bwd_kernel<32, false>), dim3(B), dim3(is::CA_THREADS), 0, st, P, stats, dz, wq, bq, wk, wv, bv, Wc, bc, partials, T, none);
    hipLaunchKernelGGL(is::comb_attn_finish_kernel<32>, dim3(1), dim3(256), 0, st, partials, B, wq, bq, wk, wv, bv, Wc, grads);
  }
  return is::launch_status(__func__);
}

static bool ca_cls_ok(int hid, int out, int act2) { return hid > 0 && hid <= is::CA_CLS_HID && out > 0 && out <= 64 && (act2 == 0 || act2 == 1); }

// The combined attention with the classifier y = act2(W2 (mask * ReLU(W1 z + b1)) + b2) behind it (models/hybrid_models.py:
// 288-295, 344-350) as ONE launch: W1 [hid, T], b1, W2 [out, hid], b2, mask [B, hid] (scaled dropout keep-mask) or NULL;
// outputs z [B, T] (saved for the backward), a1 [B, hid] (ReLU output, saved), y [B, out].  hid <= 32, out <= 64.
extern "C" int is_comb_attn_cls_fwd(const void* parts, int nparts, const float* wq, const float* bq, const float* wk,
                                    const float* wv, const float* bv, const float* Wc, const float* bc, const float* W1,
                                    const float* b1, const float* W2, const float* b2, const float* mask, float* z,
                                    float* stats, float* a1, float* y, int B, int T, int F, int hid, int out, int act2,
                                    void* stream) {
  if (B <= 0) return 0;
  if (T <= 0 || T > is::CA_TMAX || (F != 16 && F != 32) || !ca_cls_ok(hid, out, act2) || W1 == nullptr || b1 == nullptr ||
      W2 == nullptr || b2 == nullptr || z == nullptr || y == nullptr)
    return is::fail(__func__, -22);
  is::CaParts P;
  if (ca_parts_from(P, parts, nparts, T, false) != 0) return is::fail(__func__, -22);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const is::CaCls cls{W1, b1, W2, b2, mask, a1, y, nullptr, nullptr, nullptr, nullptr, nullptr, hid, out, act2, B};
  if (F == 16) hipLaunchKernelGGL((is::comb_attn_fwd_kernel<16, true>), dim3(B), dim3(is::CA_THREADS), 0, st, P, wq, bq, wk, wv, bv, Wc, bc, z, stats, T, cls);
  else hipLaunchKernelGGL((is::comb_attn_fwd_kernel<32, true>), dim3(B), dim3(is::CA_THREADS), 0, st, P, wq, bq, wk, wv, bv, Wc, bc, z, stats, T, cls);
  return is::launch_status(__func__);
}

extern "C" int is_comb_attn_cls_grad_floats(int T, int hid, int out) { return hid * T + hid + out * hid + out; }

// Backward of is_comb_attn_cls_fwd: gy [B, out] -> the pieces' gradients (parts[].dx), the attention block's parameter
// gradients `grads` (layout of is_comb_attn_bwd) and the classifier's gcls = dW1 [hid * T] | db1 | dW2 [out * hid] | db2
// (is_comb_attn_cls_grad_floats): every sample's workgroup derives d z itself, one extra workgroup contracts the samples
// (ascending order) into gcls; then the finish launch of is_comb_attn_bwd.
extern "C" int is_comb_attn_cls_bwd(const void* parts, int nparts, const float* stats, const float* gy, const float* wq,
                                    const float* bq, const float* wk, const float* wv, const float* bv, const float* Wc,
                                    const float* bc, const float* W1, const float* W2, const float* mask, const float* z,
                                    const float* a1, const float* y, float* partials, float* grads, float* gcls, int B,
                                    int T, int F, int hid, int out, int act2, void* stream) {
  if (B <= 0) return 0;
  if (T <= 0 || T > is::CA_TMAX || (F != 16 && F != 32) || !ca_cls_ok(hid, out, act2) || W1 == nullptr || W2 == nullptr ||
      z == nullptr || a1 == nullptr || gy == nullptr || gcls == nullptr || (act2 == 1 && y == nullptr))
    return is::fail(__func__, -22);
  is::CaParts P;
  if (ca_parts_from(P, parts, nparts, T, true) != 0) return is::fail(__func__, -22);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const is::CaCls cls{W1, nullptr, W2, nullptr, mask, nullptr, nullptr, a1, y, gy, z, gcls, hid, out, act2, B};
  const float* dz = nullptr;
  if (F == 16) {
    hipLaunchKernelGGL((is::comb_attn_bwd_kernel<16, true>), dim3(B + 1), dim3(is::CA_THREADS), 0, st, P, stats, dz, wq, bq, wk, wv, bv, Wc, bc, partials, T, cls);
    hipLaunchKernelGGL(is::comb_attn_finish_kernel<16>, dim3(1), dim3(256), 0, st, partials, B, wq, bq, wk, wv, bv, Wc, grads);
  } else {
    hipLaunchKernelGGL((is::comb_attn_bwd_kernel<32, true>), dim3(B + 1), dim3(is::CA_THREADS), 0, st, P, stats, dz, wq, bq, wk, wv, bv, Wc, bc, partials, T, cls);
    hipLaunchKernelGGL(is::comb_attn_finish_kernel<32>, dim3(1), dim3(256), 0, st, partials, B, wq, bq, wk, wv, bv, Wc, grads);
  }
  return is::launch_status(__func__);
}
