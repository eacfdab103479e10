// Two-layer per-sample MLP, forward and backward, for the small dense heads of the model (B x <=256 inputs):
//
//     a1[b][h]  = act1( sum_k W1[h][k] * X(b, h, k) + b1[h] )            act in {identity, ReLU}
//     hid[b][h] = a1[b][h] * mask[b][h]                                   (mask: scaled dropout keep-mask or NULL)
//     y[b][o]   = act2( sum_h W2[o][h] * hid[b][h] + b2[o] )
//
// X(b, h, k) = x[b][k] for an ordinary first layer, or x[b][(h / hgroup) * in + k] when hgroup > 0: every group of
// `hgroup` hidden units reads its own `in`-wide slice of the input row -- the per-head value projection of the
// pooled node attention (models/layers.py:67-78 reduced as in csrc/node_attention.hip).
//
// Replaces, with ONE launch forward and ONE launch (+ the generic partial reduction) backward, the chains of
// hipBLASLt GEMM + bias + ReLU + dropout (+ their ~10 backward launches) of
//   * the classifier            Linear(F,32)-ReLU-Dropout-Linear(32,1)        (models/hybrid_models.py:288-295)
//   * the property embedding    Linear(2,32)-ReLU-Dropout-Linear(32,8)-ReLU   (models/hybrid_models.py:280-286)
//   * the pooled attention tail  W_v (per head) then w_concat                   (models/layers.py:74-77)
// These layers are launch-latency bound (< 1 MFLOP per batch); plain FMA loops over LDS-resident weights, fixed
// summation order (k ascending, h ascending, samples ascending) -> bitwise reproducible.
//
// Workgroup = 256 threads = SPW samples x 32 units; weights are staged once per workgroup.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace is { int fail(const char* entry, int code); int launch_status(const char* entry); }      // (common.h: the thread's last failure)

namespace is {

constexpr int MLP_SPW = 8;        // samples per workgroup
constexpr int MLP_MAX_IN = 256;
constexpr int MLP_MAX_HID = 64;
constexpr int MLP_MAX_OUT = 64;
constexpr int MLP_MAX_XROW = 512; // heads * in for grouped inputs

// copy `count` floats with a per-element index map, 8 loads in flight per thread before the first LDS store
// (a plain load -> store loop pays the global latency once per element and thread)
template <typename SrcF, typename DstF>
__device__ __forceinline__ void stage8(int count, int tid, SrcF src, DstF dst) {
  for (int i0 = tid; i0 < count; i0 += 8 * 256) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = i0 + u * 256; v[u] = (i < count) ? src(i) : 0.0f; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = i0 + u * 256; if (i < count) dst(i, v[u]); }
  }
}

struct MlpDims {
  int B, in, hid, out, ld_x, xrow, hgroup, act1, act2;
};

__device__ __forceinline__ int mlp_xoff(const MlpDims& d, int h, int k) { return d.hgroup > 0 ? (h / d.hgroup) * d.in + k : k; }

__global__ __launch_bounds__(256) void mlp2_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ W1, const float* __restrict__ b1,
    const float* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ mask,
    float* __restrict__ a1_out, float* __restrict__ y, MlpDims d) {
  extern __shared__ float smem[];
  const int ldw = d.hid + 1;                       // W1 transposed: [in][hid + 1]
  float* w1t = smem;
  float* w2s = w1t + d.in * ldw;                   // [out][hid]
  float* xs = w2s + d.out * d.hid;                 // [SPW][xrow]
  float* hs = xs + MLP_SPW * d.xrow;               // [SPW][hid]
  const int tid = threadIdx.x;
  const int s0 = blockIdx.x * MLP_SPW;
  stage8(d.hid * d.in, tid, [&](int i) { return W1[i]; }, [&](int i, float v) { w1t[(i % d.in) * ldw + i / d.in] = v; });
  stage8(d.out * d.hid, tid, [&](int i) { return W2[i]; }, [&](int i, float v) { w2s[i] = v; });
  stage8(MLP_SPW * d.xrow, tid,
         [&](int i) { const int s = min(s0 + i / d.xrow, d.B - 1); return x[(size_t)s * d.ld_x + i % d.xrow]; },
         [&](int i, float v) { xs[i] = (s0 + i / d.xrow < d.B) ? v : 0.0f; });
  __syncthreads();
  const int s = tid >> 5, u = tid & 31;
  const bool live = s0 + s < d.B;
  for (int h = u; h < d.hid; h += 32) {
    const float* xr = xs + s * d.xrow + mlp_xoff(d, h, 0);
    // (fp64 accumulators: these heads feed the loss, whose gradient is a small difference of their outputs -- a 104-term fp32 chain
    //  left the logit 3.8 x further from the fp64 value than torch's blocked fp32 dot product; csrc/combined_attention.hip, HISTORY.md 7.10)
    //  four interleaved partial sums (k mod 4), combined as ((p0 + p1) + (p2 + p3)) + bias: a quarter of the dependent chain)
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
    int k = 0;
    for (; k + 4 <= d.in; k += 4) {
      p0 += (double)w1t[k * ldw + h] * (double)xr[k];
      p1 += (double)w1t[(k + 1) * ldw + h] * (double)xr[k + 1];
      p2 += (double)w1t[(k + 2) * ldw + h] * (double)xr[k + 2];
      p3 += (double)w1t[(k + 3) * ldw + h] * (double)xr[k + 3];
    }
    for (; k < d.in; ++k) p0 += (double)w1t[k * ldw + h] * (double)xr[k];
    const double accd = ((p0 + p1) + (p2 + p3)) + (double)b1[h];
    float acc = (float)accd;
    if (d.act1 == 1) acc = fmaxf(acc, 0.0f);
    if (live && a1_out != nullptr) a1_out[(size_t)(s0 + s) * d.hid + h] = acc;
    if (mask != nullptr && live) acc *= mask[(size_t)(s0 + s) * d.hid + h];
    hs[s * d.hid + h] = acc;
  }
  __syncthreads();
  for (int o = u; o < d.out; o += 32) {
    double accd = (double)b2[o];
    for (int h = 0; h < d.hid; ++h) accd += (double)w2s[o * d.hid + h] * (double)hs[s * d.hid + h];
    float acc = (float)accd;
    if (d.act2 == 1) acc = fmaxf(acc, 0.0f);
    if (live) y[(size_t)(s0 + s) * d.out + o] = acc;
  }
}

// partial record per workgroup: dW1 [hid][in] | db1 [hid] | dW2 [out][hid] | db2 [out]
__host__ __device__ inline int mlp_record_floats(int in, int hid, int out) { return hid * in + hid + out * hid + out; }

__global__ __launch_bounds__(256) void mlp2_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ W1, const float* __restrict__ W2,
    const float* __restrict__ mask, const float* __restrict__ a1, const float* __restrict__ y,
    const float* __restrict__ gy, float* __restrict__ gx, float* __restrict__ partials, MlpDims d) {
  extern __shared__ float smem[];
  float* w1s = smem;                               // [hid][in + 1]
  const int ldw = d.in + 1;
  float* w2s = w1s + d.hid * ldw;                  // [out][hid]
  float* xs = w2s + d.out * d.hid;                 // [SPW][xrow]
  float* hs = xs + MLP_SPW * d.xrow;               // [SPW][hid]   hid = a1 * mask
  float* gh = hs + MLP_SPW * d.hid;                // [SPW][hid]   d loss / d pre-activation 1
  float* g2 = gh + MLP_SPW * d.hid;                // [SPW][out]   d loss / d pre-activation 2
  const int tid = threadIdx.x;
  const int s0 = blockIdx.x * MLP_SPW;
  stage8(d.hid * d.in, tid, [&](int i) { return W1[i]; }, [&](int i, float v) { w1s[(i / d.in) * ldw + i % d.in] = v; });
  stage8(d.out * d.hid, tid, [&](int i) { return W2[i]; }, [&](int i, float v) { w2s[i] = v; });
  stage8(MLP_SPW * d.xrow, tid,
         [&](int i) { const int s = min(s0 + i / d.xrow, d.B - 1); return x[(size_t)s * d.ld_x + i % d.xrow]; },
         [&](int i, float v) { xs[i] = (s0 + i / d.xrow < d.B) ? v : 0.0f; });
  for (int idx = tid; idx < MLP_SPW * d.out; idx += 256) {
    const int s = idx / d.out, o = idx % d.out;
    float g = 0.0f;
    if (s0 + s < d.B) {
      g = gy[(size_t)(s0 + s) * d.out + o];
      if (d.act2 == 1 && !(y[(size_t)(s0 + s) * d.out + o] > 0.0f)) g = 0.0f;
    }
    g2[idx] = g;
  }
  __syncthreads();
  const int s = tid >> 5, u = tid & 31;
  const bool live = s0 + s < d.B;
  for (int h = u; h < d.hid; h += 32) {
    float a = 0.0f, m = 1.0f;
    if (live) {
      a = a1[(size_t)(s0 + s) * d.hid + h];
      if (mask != nullptr) m = mask[(size_t)(s0 + s) * d.hid + h];
    }
    float acc = 0.0f;
    for (int o = 0; o < d.out; ++o) acc += g2[s * d.out + o] * w2s[o * d.hid + h];
    acc *= m;
    if (d.act1 == 1 && !(a > 0.0f)) acc = 0.0f;
    gh[s * d.hid + h] = live ? acc : 0.0f;
    hs[s * d.hid + h] = live ? a * m : 0.0f;
  }
  __syncthreads();
  // ---- input gradient ----
  if (gx != nullptr) {
    for (int c = u; c < d.xrow; c += 32) {
      float acc = 0.0f;
      if (d.hgroup > 0) {
        const int grp = c / d.in, k = c % d.in;
        for (int h = grp * d.hgroup; h < (grp + 1) * d.hgroup; ++h) acc += gh[s * d.hid + h] * w1s[h * ldw + k];
      } else {
        for (int h = 0; h < d.hid; ++h) acc += gh[s * d.hid + h] * w1s[h * ldw + c];
      }
      if (live) gx[(size_t)(s0 + s) * d.ld_x + c] = acc;
    }
  }
  // ---- parameter gradients of this workgroup's samples (fixed sample order) ----
  float* rec = partials + (size_t)blockIdx.x * mlp_record_floats(d.in, d.hid, d.out);
  for (int idx = tid; idx < d.hid * d.in; idx += 256) {
    const int h = idx / d.in, k = idx % d.in;
    const int xo = mlp_xoff(d, h, k);
    float acc = 0.0f;
#pragma unroll
    for (int ss = 0; ss < MLP_SPW; ++ss) acc += gh[ss * d.hid + h] * xs[ss * d.xrow + xo];
    rec[idx] = acc;
  }
  float* rb1 = rec + d.hid * d.in;
  for (int h = tid; h < d.hid; h += 256) {
    float acc = 0.0f;
#pragma unroll
    for (int ss = 0; ss < MLP_SPW; ++ss) acc += gh[ss * d.hid + h];
    rb1[h] = acc;
  }
  float* rw2 = rb1 + d.hid;
  for (int idx = tid; idx < d.out * d.hid; idx += 256) {
    const int o = idx / d.hid, h = idx % d.hid;
    float acc = 0.0f;
#pragma unroll
    for (int ss = 0; ss < MLP_SPW; ++ss) acc += g2[ss * d.out + o] * hs[ss * d.hid + h];
    rw2[idx] = acc;
  }
  float* rb2 = rw2 + d.out * d.hid;
  for (int o = tid; o < d.out; o += 256) {
    float acc = 0.0f;
#pragma unroll
    for (int ss = 0; ss < MLP_SPW; ++ss) acc += g2[ss * d.out + o];
    rb2[o] = acc;
  }
}

static bool mlp_dims_ok(const MlpDims& d) {
  if (d.B <= 0 || d.in <= 0 || d.in > MLP_MAX_IN || d.hid <= 0 || d.hid > MLP_MAX_HID || d.out <= 0 || d.out > MLP_MAX_OUT) return false;
  if (d.hgroup < 0 || (d.hgroup > 0 && d.hid % d.hgroup != 0)) return false;
  if (d.xrow <= 0 || d.xrow > MLP_MAX_XROW || d.ld_x < d.xrow) return false;
  if ((d.act1 != 0 && d.act1 != 1) || (d.act2 != 0 && d.act2 != 1)) return false;
  return true;
}

}  // namespace is

// x [B, ld_x] (first `in` columns, or heads * in when hgroup > 0), W1 [hid, in], b1 [hid], W2 [out, hid], b2 [out],
// mask [B, hid] or NULL, a1_out [B, hid] or NULL (saved for the backward), y [B, out].  act: 0 identity, 1 ReLU.
extern "C" int is_mlp2_fwd(const float* x, int ld_x, const float* W1, const float* b1, const float* W2, const float* b2,
                           const float* mask, float* a1_out, float* y, int B, int in, int hid, int out, int hgroup,
                           int act1, int act2, void* stream) {
  if (B == 0) return 0;
  is::MlpDims d{B, in, hid, out, ld_x, hgroup > 0 ? (hid / (hgroup > 0 ? hgroup : 1)) * in : in, hgroup, act1, act2};
  if (!is::mlp_dims_ok(d)) return is::fail(__func__, -22);
  const size_t lds = sizeof(float) * ((size_t)in * (hid + 1) + (size_t)out * hid + (size_t)is::MLP_SPW * d.xrow + (size_t)is::MLP_SPW * hid);
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)is::mlp2_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(is::mlp2_fwd_kernel, dim3((B + is::MLP_SPW - 1) / is::MLP_SPW), dim3(256), lds, static_cast<hipStream_t>(stream),
                     x, W1, b1, W2, b2, mask, a1_out, y, d);
  return is::launch_status(__func__);
}

// number of workgroups (= partial records) and floats per record of is_mlp2_bwd
extern "C" int is_mlp2_bwd_records(int B) { return (B + is::MLP_SPW - 1) / is::MLP_SPW; }
extern "C" int is_mlp2_bwd_record_floats(int in, int hid, int out) { return is::mlp_record_floats(in, hid, out); }

// gy [B, out]; gx [B, ld_x] (may be NULL); partials: is_mlp2_bwd_records(B) records of
// [dW1 (hid x in) | db1 | dW2 (out x hid) | db2], to be summed by is_reduce_partials.
extern "C" int is_mlp2_bwd(const float* x, int ld_x, const float* W1, const float* W2, const float* mask, const float* a1,
                           const float* y, const float* gy, float* gx, float* partials, int B, int in, int hid, int out,
                           int hgroup, int act1, int act2, void* stream) {
  if (B == 0) return 0;
  is::MlpDims d{B, in, hid, out, ld_x, hgroup > 0 ? (hid / (hgroup > 0 ? hgroup : 1)) * in : in, hgroup, act1, act2};
  if (!is::mlp_dims_ok(d)) return is::fail(__func__, -22);
  const size_t lds = sizeof(float) * ((size_t)hid * (in + 1) + (size_t)out * hid + (size_t)is::MLP_SPW * d.xrow +
                                      2 * (size_t)is::MLP_SPW * hid + (size_t)is::MLP_SPW * out);
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)is::mlp2_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(is::mlp2_bwd_kernel, dim3((B + is::MLP_SPW - 1) / is::MLP_SPW), dim3(256), lds, static_cast<hipStream_t>(stream),
                     x, W1, W2, mask, a1, y, gy, gx, partials, d);
  return is::launch_status(__func__);
}
