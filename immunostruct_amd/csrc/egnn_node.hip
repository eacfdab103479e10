// Node-level half of an EGNNConv layer as fused MFMA kernels (replaces the dense
// torch/hipBLASLt ops around the edge pass: cat, node_mlp Linear-SiLU-Linear and
// the node pre-projection of the next layer's edge_mlp.0 -- SURVEY.md K6):
//
//   is_node_proj_fwd    psd = [h W1s^T | h W1d^T + b1]            (any Din <= 64, VALU)
//   is_egnn_node_fwd    zn1 = [h | h_neigh] Wn1^T + bn1 ; h' = SiLU(zn1) Wn2^T + bn2 ;
//                       psd' = [h' W1s'^T | h' W1d'^T + b1']       (next layer, optional)
//   is_node_proj_bwd    dh = g_h + g_psd W1sd (optional) ; dW1sd += g_psd^T h ; db1 += colsum
//   is_egnn_node_bwd    backward of the two-layer node MLP: d[h | h_neigh], dWn1, dbn1, dWn2, dbn2
//   is_reduce_partials  fixed-order sum of per-workgroup partial records, scattered
//                       through an index map into the native parameter-gradient layout
//
// All kernels read the reference's NATIVE parameter tensors (edge_mlp.0.weight is
// [64, 2*Din+1+Fe] with columns [h_src | h_dst | radial | edge feats]); no weight
// re-layout happens on the host.  One wave owns a 32-row tile; the tile's activations
// live in that wave's private LDS buffers, so no workgroup barriers are needed inside
// the tile loop.  Weight-gradient accumulators persist in registers over a
// persistent grid and leave the kernel as ONE partial record per workgroup.
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
      const float hk = __shfl(hv, k, 64);
      as += hk * ws[k];
      ad += hk * wd[k];
    }
    psd[(size_t)n * 128 + lane] = as;
    psd[(size_t)n * 128 + 64 + lane] = ad;
  }
}

// ---------------------------------------------------------------------------
template <int DIN>
struct NodeDims {
  static constexpr int KH = (DIN <= 32) ? 32 : 64;  // padded width of the h part
  static constexpr int K1 = KH + 64;
  static constexpr int LD1 = K1 + 4;                 // 100 or 132: LD1/4 odd
};

template <int DIN>
struct NodeFwdSmem {
  float wn1[64 * NodeDims<DIN>::LD1];
  float wn2[64 * LD];
  float wsd[128 * LD];
  float act[WAVES][32 * NodeDims<DIN>::LD1];
};

template <int DIN>
__global__ __launch_bounds__(256, 1) void egnn_node_fwd_kernel(
    const float* __restrict__ h, int ld_h, const float* __restrict__ h_neigh, int ld_hn,
    const float* __restrict__ Wn1, const float* __restrict__ bn1, const float* __restrict__ Wn2,
    const float* __restrict__ bn2, const float* __restrict__ W1n, int ldw_n, const float* __restrict__ b1n,
    float* __restrict__ zn1, float* __restrict__ h_out, float* __restrict__ psd_next, int N) {
  using D = NodeDims<DIN>;
  __shared__ NodeFwdSmem<DIN> sm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hf = lane >> 5;
  const bool has_next = W1n != nullptr;

  staged_copy<64 * D::K1, 256>(tid,
      [&](int idx) {
        const int o = idx / D::K1, k = idx % D::K1;
        if (k < DIN) return Wn1[o * (DIN + 64) + k];
        if (k >= D::KH) return Wn1[o * (DIN + 64) + DIN + (k - D::KH)];
        return 0.0f;
      },
      [&](int idx, float v) { sm.wn1[(idx / D::K1) * D::LD1 + idx % D::K1] = v; });
  load_matrix_lds(sm.wn2, Wn2, H, tid, 256);
  if (has_next) {
    staged_copy<128 * 64, 256>(tid,
        [&](int idx) {
          const int c = idx / 64, k = idx % 64;
          return (c < 64) ? W1n[c * ldw_n + k] : W1n[(c - 64) * ldw_n + 64 + k];
        },
        [&](int idx, float v) { sm.wsd[(idx / 64) * LD + idx % 64] = v; });
  }
  float bn1_c[2], bn2_c[2], b1n_c[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    bn1_c[nt] = bn1[nt * 32 + r];
    bn2_c[nt] = bn2[nt * 32 + r];
    b1n_c[nt] = has_next ? b1n[nt * 32 + r] : 0.0f;
  }
  __syncthreads();

  float* act = sm.act[wave];
  const int num_tiles = (N + 31) / 32;
  for (int tile = blockIdx.x * WAVES + wave; tile < num_tiles; tile += gridDim.x * WAVES) {
    const int row0 = tile * 32;
    // ---- X = [h | h_neigh] rows -> LDS (lane = column) ----
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
      const int row = row0 + i;
      const bool valid = row < N;
      if (lane < D::KH) act[i * D::LD1 + lane] = (valid && lane < DIN) ? h[(size_t)row * ld_h + lane] : 0.0f;
      act[i * D::LD1 + D::KH + lane] = valid ? h_neigh[(size_t)row * ld_hn + lane] : 0.0f;
    }
    // ---- zn1 = X Wn1^T + bn1 ; a1 = SiLU(zn1) (written over the same buffer, stride LD) ----
    {
      f32x16 acc[2];
      zero_acc(acc);
      mm_rows<2, D::K1, D::LD1, D::LD1>(acc, act, sm.wn1, lane);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int row = tile_row(t, hf);
          const float z = acc[nt][t] + bn1_c[nt];
          if (zn1 != nullptr && row0 + row < N) zn1[(size_t)(row0 + row) * H + nt * 32 + r] = z;
          act[row * LD + nt * 32 + r] = silu_f(z);
        }
    }
    // ---- h' = a1 Wn2^T + bn2 ----
    {
      f32x16 acc[2];
      zero_acc(acc);
      mm_rows<2, H>(acc, act, sm.wn2, lane);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int row = tile_row(t, hf);
          const float v = acc[nt][t] + bn2_c[nt];
          if (row0 + row < N) h_out[(size_t)(row0 + row) * H + nt * 32 + r] = v;
          act[row * LD + nt * 32 + r] = v;
        }
    }
    // ---- next layer's node pre-projection psd' = [h' W1s'^T | h' W1d'^T + b1'] ----
    if (has_next) {
      f32x16 acc[4];
      zero_acc(acc);
      mm_rows<4, H>(acc, act, sm.wsd, lane);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int row = tile_row(t, hf);
          const float v = acc[nt][t] + (nt >= 2 ? b1n_c[nt - 2] : 0.0f);
          if (row0 + row < N) psd_next[(size_t)(row0 + row) * 128 + nt * 32 + r] = v;
        }
    }
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
// backward of h' = SiLU([h | h_neigh] Wn1^T + bn1) Wn2^T + bn2
// record: [dWn1 64 x 128 (h part padded to 64 columns | h_neigh part)][dWn2 64 x 64][dbn1 64][dbn2 64]
constexpr int NODE_STRIDE = 64 * 128 + 64 * 64 + 128;

struct NodeBwdSmem {
  float wn2t[64 * LD];    // wn2t[i][o] = Wn2[o][i]
  float wn1t[128 * LD];   // wn1t[k][o] = Wn1[o][k'] (k < 64: h column k, zero beyond DIN; k >= 64: h_neigh column k-64)
  float bufA[WAVES][32 * LD];
  float bufB[WAVES][32 * LD];
};

template <int DIN>
__global__ __launch_bounds__(256, 1) void egnn_node_bwd_kernel(
    const float* __restrict__ g_hout, const float* __restrict__ h, int ld_h, const float* __restrict__ h_neigh,
    int ld_hn, const float* __restrict__ zn1, const float* __restrict__ Wn1, const float* __restrict__ Wn2,
    float* __restrict__ d_h, float* __restrict__ d_hneigh, float* __restrict__ partials, int N) {
  __shared__ NodeBwdSmem sm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hf = lane >> 5;
  load_matrix_lds_t(sm.wn2t, Wn2, tid, 256);
  staged_copy<64 * 128, 256>(tid,
      [&](int idx) {
        const int o = idx / 128, k = idx % 128;
        if (k < 64) return (k < DIN) ? Wn1[o * (DIN + 64) + k] : 0.0f;
        return Wn1[o * (DIN + 64) + DIN + (k - 64)];
      },
      [&](int idx, float v) { sm.wn1t[(idx % 128) * LD + idx / 128] = v; });
  __syncthreads();

  f32x16 dWn1h[2][2], dWn1n[2][2], dWn2[2][2];
  zero_acc2(dWn1h); zero_acc2(dWn1n); zero_acc2(dWn2);
  float dbn1_a[2] = {0.f, 0.f}, dbn2_a[2] = {0.f, 0.f};
  float* bufA = sm.bufA[wave];
  float* bufB = sm.bufB[wave];
  const int num_tiles = (N + 31) / 32;
  for (int tile = blockIdx.x * WAVES + wave; tile < num_tiles; tile += gridDim.x * WAVES) {
    const int row0 = tile * 32;
    // ---- dh' -> bufA, a1 = SiLU(zn1) -> bufB (tile layout) ----
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int row = tile_row(t, hf);
        const bool rv = row0 + row < N;
        const size_t off = (size_t)(row0 + row) * H + nt * 32 + r;
        const float g = rv ? g_hout[off] : 0.0f;
        const float z = rv ? zn1[off] : 0.0f;
        const float a1 = silu_f(z);
        dbn2_a[nt] += g;
        bufA[row * LD + nt * 32 + r] = g;
        bufB[row * LD + nt * 32 + r] = rv ? a1 : 0.0f;
      }
    mm_outer<2, 2>(dWn2, bufA, bufB, lane);
    // ---- da1 = dh' Wn2 ; dzn1 = da1 * SiLU'(zn1) -> bufA ----
    {
      f32x16 acc[2];
      zero_acc(acc);
      mm_rows<2, H>(acc, bufA, sm.wn2t, lane);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int row = tile_row(t, hf);
          const bool rv = row0 + row < N;
          float y, dy;
          silu_fg(rv ? zn1[(size_t)(row0 + row) * H + nt * 32 + r] : 0.0f, y, dy);  // re-read: L2 hit
          const float dz = rv ? acc[nt][t] * dy : 0.0f;
          dbn1_a[nt] += dz;
          bufA[row * LD + nt * 32 + r] = dz;
        }
    }
    // ---- dWn1 (h part): X_h -> bufB ----
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
      const int row = row0 + i;
      bufB[i * LD + lane] = (row < N && lane < DIN) ? h[(size_t)row * ld_h + lane] : 0.0f;
    }
    mm_outer<2, 2>(dWn1h, bufA, bufB, lane);
    // ---- dWn1 (h_neigh part): X_n -> bufB ----
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
      const int row = row0 + i;
      bufB[i * LD + lane] = (row < N) ? h_neigh[(size_t)row * ld_hn + lane] : 0.0f;
    }
    mm_outer<2, 2>(dWn1n, bufA, bufB, lane);
    // ---- dX = dzn1 Wn1 : columns [0,64) -> d_h, [64,128) -> d_hneigh ----
    {
      f32x16 acc[4];
      zero_acc(acc);
      mm_rows<4, H>(acc, bufA, sm.wn1t, lane);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int row = row0 + tile_row(t, hf);
          if (row < N) {
            if (nt < 2) { if (d_h != nullptr) d_h[(size_t)row * H + nt * 32 + r] = acc[nt][t]; }
            else d_hneigh[(size_t)row * H + (nt - 2) * 32 + r] = acc[nt][t];
          }
        }
    }
  }
  // ---- workgroup partial record ----
  __syncthreads();
  float* part = partials + (size_t)blockIdx.x * NODE_STRIDE;
  float* scratch = reinterpret_cast<float*>(&sm);   // 16384 floats of the (now idle) LDS image
  wg_sum_store_64x64(dWn1h, scratch, part, 128, tid, wave, lane);
  wg_sum_store_64x64(dWn1n, scratch, part + 64, 128, tid, wave, lane);
  wg_sum_store_64x64(dWn2, scratch, part + 64 * 128, 64, tid, wave, lane);
  float* vec = &sm.bufA[0][0];  // [wave][2][64]
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const float v1 = dbn1_a[nt] + __shfl_xor(dbn1_a[nt], 32, 64);
    const float v2 = dbn2_a[nt] + __shfl_xor(dbn2_a[nt], 32, 64);
    if (hf == 0) {
      vec[(wave * 2 + 0) * H + nt * 32 + r] = v1;
      vec[(wave * 2 + 1) * H + nt * 32 + r] = v2;
    }
  }
  __syncthreads();
  if (tid < 2 * H) {
    const int s = tid / H, c = tid % H;
    float v = 0.0f;
    for (int w = 0; w < WAVES; ++w) v += vec[(w * 2 + s) * H + c];
    part[64 * 128 + 64 * 64 + s * H + c] = v;
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
  // 4 independent chains keep several loads in flight; the summation order is fixed (deterministic)
  float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
  int p = blockIdx.y;
  for (; p + 3 * RED_SPLIT < nparts; p += 4 * RED_SPLIT) {
    v0 += partials[(size_t)p * stride + idx];
    v1 += partials[(size_t)(p + RED_SPLIT) * stride + idx];
    v2 += partials[(size_t)(p + 2 * RED_SPLIT) * stride + idx];
    v3 += partials[(size_t)(p + 3 * RED_SPLIT) * stride + idx];
  }
  for (; p < nparts; p += RED_SPLIT) v0 += partials[(size_t)p * stride + idx];
  scratch[(size_t)blockIdx.y * count + idx] = (v0 + v1) + (v2 + v3);
}
__global__ __launch_bounds__(256) void reduce_partials_stage2(const float* __restrict__ scratch, int count,
                                                              const int* __restrict__ map, float* __restrict__ dst) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= count) return;
  float v = 0.0f;
#pragma unroll
  for (int s = 0; s < RED_SPLIT; ++s) v += scratch[(size_t)s * count + idx];  // loads are independent: issued together
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
  float v = 0.0f;
#pragma unroll
  for (int s = 0; s < RED_SPLIT; ++s) v += r[s];
  const int d = map != nullptr ? map[idx] : idx;
  if (d >= 0) dst[d] = v;
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
  float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
  int p = blockIdx.y;
  for (; p + 3 * RED_SPLIT < J.nparts; p += 4 * RED_SPLIT) {
    v0 += J.partials[(size_t)p * J.stride + idx];
    v1 += J.partials[(size_t)(p + RED_SPLIT) * J.stride + idx];
    v2 += J.partials[(size_t)(p + 2 * RED_SPLIT) * J.stride + idx];
    v3 += J.partials[(size_t)(p + 3 * RED_SPLIT) * J.stride + idx];
  }
  for (; p < J.nparts; p += RED_SPLIT) v0 += J.partials[(size_t)p * J.stride + idx];
  J.scratch[(size_t)blockIdx.y * J.count + idx] = (v0 + v1) + (v2 + v3);
}
__global__ __launch_bounds__(256) void reduce_partials_batched_stage2(ReduceBatch batch) {
  const ReduceJob& J = batch.job[blockIdx.z];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= J.count) return;
  float v = 0.0f;
#pragma unroll
  for (int s = 0; s < RED_SPLIT; ++s) v += J.scratch[(size_t)s * J.count + idx];
  const int d = J.map != nullptr ? J.map[idx] : idx;
  if (d >= 0) J.dst[d] = v;
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
  else return -22;
  IS_RET();
}

extern "C" int is_egnn_node_fwd(const float* h, int ld_h, int din, const float* h_neigh, int ld_hn, const float* Wn1,
                                const float* bn1, const float* Wn2, const float* bn2, const float* W1n, int ldw_n,
                                const float* b1n, float* zn1, float* h_out, float* psd_next, int N, void* stream) {
  if (N <= 0) return 0;
  const dim3 grid((N + 127) / 128), block(256);
  if (din == 20) hipLaunchKernelGGL(is::egnn_node_fwd_kernel<20>, grid, block, 0, IS_STREAM(stream), h, ld_h, h_neigh, ld_hn, Wn1, bn1, Wn2, bn2, W1n, ldw_n, b1n, zn1, h_out, psd_next, N);
  else if (din == 64) hipLaunchKernelGGL(is::egnn_node_fwd_kernel<64>, grid, block, 0, IS_STREAM(stream), h, ld_h, h_neigh, ld_hn, Wn1, bn1, Wn2, bn2, W1n, ldw_n, b1n, zn1, h_out, psd_next, N);
  else return -22;
  IS_RET();
}

extern "C" int is_node_proj_bwd_floats(int grid) { return grid * is::PROJ_STRIDE; }

extern "C" int is_node_proj_bwd(const float* g_h, const float* g_psd, const float* h, int ld_h, int din,
                                const float* W1, int ldw, float* dh_total, float* partials, int grid, int N,
                                void* stream) {
  if (N <= 0 || grid <= 0 || din <= 0 || din > 64) return -22;
  hipLaunchKernelGGL(is::node_proj_bwd_kernel, dim3(grid), dim3(256), 0, IS_STREAM(stream), g_h, g_psd, h, ld_h, din, W1, ldw, dh_total, partials, N);
  IS_RET();
}

extern "C" int is_egnn_node_bwd_floats(int grid) { return grid * is::NODE_STRIDE; }

extern "C" int is_egnn_node_bwd(const float* g_hout, const float* h, int ld_h, int din, const float* h_neigh, int ld_hn,
                                const float* zn1, const float* Wn1, const float* Wn2, float* d_h, float* d_hneigh,
                                float* partials, int grid, int N, void* stream) {
  if (N <= 0 || grid <= 0) return -22;
  if (din == 20) hipLaunchKernelGGL(is::egnn_node_bwd_kernel<20>, dim3(grid), dim3(256), 0, IS_STREAM(stream), g_hout, h, ld_h, h_neigh, ld_hn, zn1, Wn1, Wn2, d_h, d_hneigh, partials, N);
  else if (din == 64) hipLaunchKernelGGL(is::egnn_node_bwd_kernel<64>, dim3(grid), dim3(256), 0, IS_STREAM(stream), g_hout, h, ld_h, h_neigh, ld_hn, zn1, Wn1, Wn2, d_h, d_hneigh, partials, N);
  else return -22;
  IS_RET();
}

extern "C" int is_reduce_partials_scratch_floats(int stride) { return is::RED_SPLIT * stride; }

// record p starts at partials + p*stride; its first `count` floats are reduced
extern "C" int is_reduce_partials(const float* partials, int nparts, int stride, int count, const int32_t* map,
                                  float* dst, float* scratch, void* stream) {
  if (nparts <= 0 || stride <= 0 || count <= 0 || count > stride) return -22;
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
  if (njobs <= 0 || njobs > is::RED_MAX_JOBS) return -22;
  is::ReduceBatch batch;
  const is::ReduceJob* src = static_cast<const is::ReduceJob*>(jobs);
  int maxcount = 0;
  for (int i = 0; i < njobs; ++i) {
    batch.job[i] = src[i];
    if (src[i].count <= 0 || src[i].count > src[i].stride || src[i].nparts <= 0) return -22;
    maxcount = src[i].count > maxcount ? src[i].count : maxcount;
  }
  const dim3 block(256);
  hipLaunchKernelGGL(is::reduce_partials_batched_stage1, dim3((maxcount + 255) / 256, is::RED_SPLIT, njobs), block, 0, IS_STREAM(stream), batch);
  hipLaunchKernelGGL(is::reduce_partials_batched_stage2, dim3((maxcount + 255) / 256, 1, njobs), block, 0, IS_STREAM(stream), batch);
  IS_RET();
}
