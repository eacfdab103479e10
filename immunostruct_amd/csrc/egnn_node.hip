// Node pre-projection (VALU form, any Din <= 64), its backward, and the fixed-order reduction of per-workgroup
// partial records:
//
//   is_node_proj_fwd    psd = [h W1s^T (+ b0) | h W1d^T + b1]
//   is_node_proj_bwd    dh = g_h + g_psd W1sd (optional) ; dW1sd += g_psd^T h ; db1 += colsum
//   is_reduce_partials(_batched)  fixed-order sum of per-workgroup partial records, scattered through an index map
//                       into the native parameter-gradient layout
//
// The node MLP itself lives in egnn_node16.hip / the fused layer kernels.  All kernels read the reference's NATIVE
// parameter tensors (edge_mlp.0.weight is [64, 2*Din+1+Fe] with columns [h_src | h_dst | radial | edge feats]).
#include <algorithm>
#include "common.h"

namespace is {

constexpr int LDW2 = 132;  // row stride for 128-wide LDS tiles (132/4 odd => conflict-free)

// ---------------------------------------------------------------------------
// psd[n][c]      = sum_k h[n][k] W1[c][k]            (c < 64,  "Ps")
// psd[n][64 + c] = sum_k h[n][k] W1[c][DIN + k] + b1[c]        ("Pd")
// lane = channel c; each wave walks nodes; h row is broadcast with readlane.
template <int DIN>
__global__ __launch_bounds__(256) void node_proj_fwd_kernel(const float* __restrict__ h, int ld_h,
                                                            const float* __restrict__ W1, int ldw,
                                                            const float* __restrict__ b0, const float* __restrict__ b1,
                                                            float* __restrict__ psd, int N) {
  const int lane = threadIdx.x & 63;
  float ws[DIN], wd[DIN];
#pragma unroll
  for (int k = 0; k < DIN; ++k) {
    ws[k] = W1[lane * ldw + k];
    wd[k] = W1[lane * ldw + DIN + k];
  }
  const float bias = b1[lane];
  const float bias0 = b0 != nullptr ? b0[lane] : 0.0f;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  for (int n = wave_global; n < N; n += gridDim.x * 4) {
    const float hv = (lane < DIN) ? h[(size_t)n * ld_h + lane] : 0.0f;
    float as = bias0, ad = bias;
#pragma unroll
    for (int k = 0; k < DIN; ++k) {
      const float hk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hv), k));
      as += hk * ws[k];
      ad += hk * wd[k];
    }
    psd[(size_t)n * 128 + lane] = as;
    psd[(size_t)n * 128 + 64 + lane] = ad;
  }
}

// ---------------------------------------------------------------------------
// dh_total = g_h + g_psd W1sd ; dW1sd += g_psd^T h_out ; db1 += colsum(g_psd[:, 64:])
// record: [dW1sd 128 x 64][db1 64][db0 64]   (db0 = column sums of g_psd[:, :64], used by callers with a bias on the first half)
constexpr int PROJ_STRIDE = 128 * 64 + 128;

struct ProjBwdSmem {
  float w1sdT[64 * LDW2];  // w1sdT[i][c] = W1sd[c][i]
  float bufP[WAVES][32 * LDW2];
  float bufH[WAVES][32 * LD];
};

__global__ __launch_bounds__(256, 1) void node_proj_bwd_kernel(
    const float* __restrict__ g_h, const float* __restrict__ g_psd, const float* __restrict__ h_out, int ld_h, int din,
    const float* __restrict__ W1n, int ldw_n, float* __restrict__ dh_total, float* __restrict__ partials, int N) {
  __shared__ ProjBwdSmem sm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hf = lane >> 5;
  staged_copy<128 * 64, 256>(tid,
      [&](int idx) {
        const int c = idx / 64, i = idx % 64;
        if (i >= din) return 0.0f;
        return (c < 64) ? W1n[c * ldw_n + i] : W1n[(c - 64) * ldw_n + din + i];
      },
      [&](int idx, float v) { sm.w1sdT[(idx % 64) * LDW2 + idx / 64] = v; });
  __syncthreads();
  f32x16 dW[4][2];
  zero_acc2(dW);
  float db1_c = 0.0f, db0_c = 0.0f;  // lane = channel
  float* bufP = sm.bufP[wave];
  float* bufH = sm.bufH[wave];
  const int num_tiles = (N + 31) / 32;
  for (int tile = blockIdx.x * WAVES + wave; tile < num_tiles; tile += gridDim.x * WAVES) {
    const int row0 = tile * 32;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
      const int row = row0 + i;
      const bool valid = row < N;
      const float p0 = valid ? g_psd[(size_t)row * 128 + lane] : 0.0f;
      const float p1 = valid ? g_psd[(size_t)row * 128 + 64 + lane] : 0.0f;
      bufP[i * LDW2 + lane] = p0;
      bufP[i * LDW2 + 64 + lane] = p1;
      db1_c += p1;
      db0_c += p0;
      bufH[i * LD + lane] = (valid && lane < din) ? h_out[(size_t)row * ld_h + lane] : 0.0f;
    }
    mm_outer<4, 2, LDW2, LD>(dW, bufP, bufH, lane);
    if (dh_total != nullptr) {
      f32x16 acc[2];
      zero_acc(acc);
      mm_rows<2, 128, LDW2, LDW2>(acc, bufP, sm.w1sdT, lane);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int row = row0 + tile_row(t, hf);
          if (row < N) {
            const size_t off = (size_t)row * H + nt * 32 + r;
            dh_total[off] = acc[nt][t] + (g_h != nullptr ? g_h[off] : 0.0f);
          }
        }
    }
  }
  // ---- workgroup partial: sum the 4 waves through LDS ----
  __syncthreads();
  float* part = partials + (size_t)blockIdx.x * PROJ_STRIDE;
  float* scratch = &sm.bufP[0][0];  // 4 * 32 * 132 = 16896 floats >= 16384
  {
    f32x16 blk[2][2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) blk[mt][nt] = dW[half * 2 + mt][nt];
      wg_sum_store_64x64(blk, scratch, part + half * 64 * 64, 64, tid, wave, lane);
    }
  }
  float* vec = &sm.bufH[0][0];
  vec[wave * H + lane] = db1_c;
  vec[(WAVES + wave) * H + lane] = db0_c;
  __syncthreads();
  if (tid < 2 * H) {
    const int s = tid / H, c = tid % H;
    const float* v = vec + s * WAVES * H;
    part[128 * 64 + tid] = ((v[c] + v[H + c]) + v[2 * H + c]) + v[3 * H + c];
  }
}

// ---------------------------------------------------------------------------
// dst[map[idx]] = sum_p partials[p * stride + idx]   (map may be NULL => dst[idx]; map < 0 => skipped)
// two stages so that small records still fill the chip; order of summation is fixed.
constexpr int RED_SPLIT = 16;

__global__ __launch_bounds__(256) void reduce_partials_stage1(const float* __restrict__ partials, int nparts,
                                                              int stride, int count, float* __restrict__ scratch) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= count) return;
  // 4 independent chains keep several loads in flight; the summation order is fixed (deterministic); fp64 accumulators (see
  // reduce_partials_batched_stage1)
  double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
  int p = blockIdx.y;
  for (; p + 3 * RED_SPLIT < nparts; p += 4 * RED_SPLIT) {
    v0 += (double)partials[(size_t)p * stride + idx];
    v1 += (double)partials[(size_t)(p + RED_SPLIT) * stride + idx];
    v2 += (double)partials[(size_t)(p + 2 * RED_SPLIT) * stride + idx];
    v3 += (double)partials[(size_t)(p + 3 * RED_SPLIT) * stride + idx];
  }
  for (; p < nparts; p += RED_SPLIT) v0 += (double)partials[(size_t)p * stride + idx];
  scratch[(size_t)blockIdx.y * count + idx] = (float)((v0 + v1) + (v2 + v3));
}
__global__ __launch_bounds__(256) void reduce_partials_stage2(const float* __restrict__ scratch, int count,
                                                              const int* __restrict__ map, float* __restrict__ dst) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= count) return;
  double v = 0.0;
#pragma unroll
  for (int s = 0; s < RED_SPLIT; ++s) v += (double)scratch[(size_t)s * count + idx];  // loads are independent: issued together
  const int d = map != nullptr ? map[idx] : idx;
  if (d >= 0) dst[d] = v;
}

// nparts <= RED_SPLIT: one launch.  Same summation order as the two stages (stage 1 then only copies record p to slot p,
// stage 2 adds the slots in ascending order), hence the same bits.
__global__ __launch_bounds__(256) void reduce_partials_direct(const float* __restrict__ partials, int nparts, int stride,
                                                              int count, const int* __restrict__ map, float* __restrict__ dst) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= count) return;
  float r[RED_SPLIT];
#pragma unroll
  for (int s = 0; s < RED_SPLIT; ++s) r[s] = (s < nparts) ? partials[(size_t)s * stride + idx] : 0.0f;   // issued together
  double v = 0.0;
#pragma unroll
  for (int s = 0; s < RED_SPLIT; ++s) v += (double)r[s];
  const int d = map != nullptr ? map[idx] : idx;
  if (d >= 0) dst[d] = (float)v;
}

// Batched form: up to RED_MAX_JOBS independent reductions (one per kernel's partial records) in two launches.
struct ReduceJob {
  const float* partials;
  const int* map;
  float* dst;
  float* scratch;     // RED_SPLIT * count floats
  int nparts, stride, count, pad;
};
constexpr int RED_MAX_JOBS = 24;
struct ReduceBatch { ReduceJob job[RED_MAX_JOBS]; };

__global__ __launch_bounds__(256) void reduce_partials_batched_stage1(ReduceBatch batch) {
  const ReduceJob& J = batch.job[blockIdx.z];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= J.count) return;
  // the records are fp32; their SUM is formed in fp64 (one rounding per stage instead of one per record): weight gradients
  // such as coord_mlp.2.weight are sums over 150 k edges that cancel to a small fraction of their terms, and the 512-record
  // tail of that sum was where most of the fp32 round-off entered.  The launch is bandwidth-bound: the wider adds are free.
  double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
  int p = blockIdx.y;
  for (; p + 3 * RED_SPLIT < J.nparts; p += 4 * RED_SPLIT) {
    v0 += (double)J.partials[(size_t)p * J.stride + idx];
    v1 += (double)J.partials[(size_t)(p + RED_SPLIT) * J.stride + idx];
    v2 += (double)J.partials[(size_t)(p + 2 * RED_SPLIT) * J.stride + idx];
    v3 += (double)J.partials[(size_t)(p + 3 * RED_SPLIT) * J.stride + idx];
  }
  for (; p < J.nparts; p += RED_SPLIT) v0 += (double)J.partials[(size_t)p * J.stride + idx];
  J.scratch[(size_t)blockIdx.y * J.count + idx] = (float)((v0 + v1) + (v2 + v3));
}
// the same stage with 16-byte accesses (round 5): thread = four consecutive entries -- a quarter of the load instructions and full
// 1 KB wave accesses on a launch that does nothing but stream ~ 90 MB of records.  Entry by entry the same chains and the same
// order as the scalar form: same bits.  Used when every job's count, stride and base allow it.
__global__ __launch_bounds__(256) void reduce_partials_batched_stage1_v4(ReduceBatch batch) {
  const ReduceJob& J = batch.job[blockIdx.z];
  const int idx = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (idx >= J.count) return;
  double v0[4] = {0.0, 0.0, 0.0, 0.0}, v1[4] = {0.0, 0.0, 0.0, 0.0}, v2[4] = {0.0, 0.0, 0.0, 0.0}, v3[4] = {0.0, 0.0, 0.0, 0.0};
  auto row = [&](int p) { return *reinterpret_cast<const f32x4*>(J.partials + (size_t)p * J.stride + idx); };
  int p = blockIdx.y;
  for (; p + 3 * RED_SPLIT < J.nparts; p += 4 * RED_SPLIT) {
    const f32x4 a = row(p), b = row(p + RED_SPLIT), c = row(p + 2 * RED_SPLIT), d = row(p + 3 * RED_SPLIT);
#pragma unroll
    for (int k = 0; k < 4; ++k) { v0[k] += (double)a[k]; v1[k] += (double)b[k]; v2[k] += (double)c[k]; v3[k] += (double)d[k]; }
  }
  for (; p < J.nparts; p += RED_SPLIT) {
    const f32x4 a = row(p);
#pragma unroll
    for (int k = 0; k < 4; ++k) v0[k] += (double)a[k];
  }
  f32x4 out;
#pragma unroll
  for (int k = 0; k < 4; ++k) out[k] = (float)((v0[k] + v1[k]) + (v2[k] + v3[k]));
  *reinterpret_cast<f32x4*>(J.scratch + (size_t)blockIdx.y * J.count + idx) = out;
}
__global__ __launch_bounds__(256) void reduce_partials_batched_stage2(ReduceBatch batch) {
  const ReduceJob& J = batch.job[blockIdx.z];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= J.count) return;
  double v = 0.0;
#pragma unroll
  for (int s = 0; s < RED_SPLIT; ++s) v += (double)J.scratch[(size_t)s * J.count + idx];
  const int d = J.map != nullptr ? J.map[idx] : idx;
  if (d >= 0) J.dst[d] = (float)v;
}

}  // namespace is

#define IS_STREAM(s) static_cast<hipStream_t>(s)
#define IS_RET() return hipGetLastError() == hipSuccess ? 0 : -5

extern "C" int is_node_proj_fwd(const float* h, int ld_h, int din, const float* W1, int ldw, const float* b0,
                                const float* b1, float* psd, int N, void* stream) {
  if (N <= 0) return 0;
  const dim3 grid(std::min((N + 3) / 4, 2048)), block(256);
  if (din == 20) hipLaunchKernelGGL(is::node_proj_fwd_kernel<20>, grid, block, 0, IS_STREAM(stream), h, ld_h, W1, ldw, b0, b1, psd, N);
  else if (din == 64) hipLaunchKernelGGL(is::node_proj_fwd_kernel<64>, grid, block, 0, IS_STREAM(stream), h, ld_h, W1, ldw, b0, b1, psd, N);
  else return is::fail(__func__, -22);
  IS_RET();
}

extern "C" int is_node_proj_bwd(const float* g_h, const float* g_psd, const float* h, int ld_h, int din,
                                const float* W1, int ldw, float* dh_total, float* partials, int grid, int N,
                                void* stream) {
  if (N <= 0 || grid <= 0 || din <= 0 || din > 64) return is::fail(__func__, -22);
  hipLaunchKernelGGL(is::node_proj_bwd_kernel, dim3(grid), dim3(256), 0, IS_STREAM(stream), g_h, g_psd, h, ld_h, din, W1, ldw, dh_total, partials, N);
  IS_RET();
}

extern "C" int is_reduce_partials_scratch_floats(int stride) { return is::RED_SPLIT * stride; }

// record p starts at partials + p*stride; its first `count` floats are reduced
extern "C" int is_reduce_partials(const float* partials, int nparts, int stride, int count, const int32_t* map,
                                  float* dst, float* scratch, void* stream) {
  if (nparts <= 0 || stride <= 0 || count <= 0 || count > stride) return is::fail(__func__, -22);
  const dim3 block(256);
  if (nparts <= is::RED_SPLIT) {
    hipLaunchKernelGGL(is::reduce_partials_direct, dim3((count + 255) / 256), block, 0, IS_STREAM(stream), partials, nparts, stride, count, map, dst);
    IS_RET();
  }
  hipLaunchKernelGGL(is::reduce_partials_stage1, dim3((count + 255) / 256, is::RED_SPLIT), block, 0, IS_STREAM(stream), partials, nparts, stride, count, scratch);
  hipLaunchKernelGGL(is::reduce_partials_stage2, dim3((count + 255) / 256), block, 0, IS_STREAM(stream), scratch, count, map, dst);
  IS_RET();
}

// jobs: host array of `njobs` (<= 24) records {partials, map, dst, scratch, nparts, stride, count, pad}
extern "C" int is_reduce_partials_batched(const void* jobs, int njobs, void* stream) {
  if (njobs <= 0 || njobs > is::RED_MAX_JOBS) return is::fail(__func__, -22);
  is::ReduceBatch batch;
  const is::ReduceJob* src = static_cast<const is::ReduceJob*>(jobs);
  int maxcount = 0;
  for (int i = 0; i < njobs; ++i) {
    batch.job[i] = src[i];
    if (src[i].count <= 0 || src[i].count > src[i].stride || src[i].nparts <= 0) return is::fail(__func__, -22);
    maxcount = src[i].count > maxcount ? src[i].count : maxcount;
  }
  const dim3 block(256);
  bool v4 = true;
  for (int i = 0; i < njobs; ++i)
    v4 = v4 && (src[i].count % 4 == 0) && (src[i].stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(src[i].partials) & 15) == 0) &&
         ((reinterpret_cast<uintptr_t>(src[i].scratch) & 15) == 0);
  if (v4)
    hipLaunchKernelGGL(is::reduce_partials_batched_stage1_v4, dim3((maxcount / 4 + 255) / 256, is::RED_SPLIT, njobs), block, 0, IS_STREAM(stream), batch);
  else
    hipLaunchKernelGGL(is::reduce_partials_batched_stage1, dim3((maxcount + 255) / 256, is::RED_SPLIT, njobs), block, 0, IS_STREAM(stream), batch);
  hipLaunchKernelGGL(is::reduce_partials_batched_stage2, dim3((maxcount + 255) / 256, 1, njobs), block, 0, IS_STREAM(stream), batch);
  IS_RET();
}
