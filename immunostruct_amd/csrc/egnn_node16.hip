// Node-level kernels, second mapping (v2): one workgroup per 32-row tile, weights straight from L2.
//
// The first mapping (egnn_node.hip) stages ~86 KB of weights into LDS per workgroup to process 128
// rows: at N = 24k rows that staging and the single wave per SIMD dominate the run time.  Here
//   * a workgroup (4 waves) owns ONE 32-row tile; wave w produces output columns [16w, 16w+16)
//     (16x16x4 MFMA, two row tiles), so the 64 x 64 layers of a tile run 4-wide;
//   * the B operands (weights) are fetched by each lane directly from the NATIVE row-major parameter
//     tensors in global memory (L2-resident, 16 B per lane per 4 k) at kernel start -- they do not
//     depend on the tile's data, so their latency overlaps the staging of the activations;
//   * LDS holds only the tile's activations (~34 KB) => 4 workgroups = 16 waves per CU;
//   * the backward data path (is_egnn_node_bwd_data) carries no weight-gradient accumulators; the
//     weight gradients of a layer are produced by ONE streaming outer-product kernel
//     (is_egnn_node_wgrad) from the tensors the data path leaves in HBM.
#include <algorithm>
#include "common.h"

namespace is {

#ifdef IS_STAGE_STAMPS
__device__ long long g_stamps_node[16];
#define STAMPN(k) do { if (blockIdx.x == 300 && threadIdx.x == 0) g_stamps_node[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMPN(k) do { } while (0)
#endif

__device__ __forceinline__ f32x4 ldg4(const float* p, int align) {
  if (align >= 4) return *reinterpret_cast<const f32x4*>(p);
  if (align == 2) {
    const float2 a = *reinterpret_cast<const float2*>(p), b = *reinterpret_cast<const float2*>(p + 2);
    return f32x4{a.x, a.y, b.x, b.y};
  }
  return f32x4{p[0], p[1], p[2], p[3]};
}
__device__ __forceinline__ int align_of(const float* base, int ld) {
  const bool a16 = ((reinterpret_cast<uintptr_t>(base) & 15) == 0) && ((ld & 3) == 0);
  const bool a8 = ((reinterpret_cast<uintptr_t>(base) & 7) == 0) && ((ld & 1) == 0);
  return a16 ? 4 : (a8 ? 2 : 1);
}

// acc[mt] (16 x 16) += A[mt*16 + i][k] * B[k][j]: A rows from LDS (stride LDA), B from registers
// (b[g] holds the 4 consecutive k of group g of this lane's quarter).
template <int MT, int KQ, int LDA>
__device__ __forceinline__ void mm16_regB(f32x4 (&acc)[MT], const float* a_lds, const f32x4 (&b)[KQ / 4], int lane) {
  const int r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int g = 0; g < KQ / 4; ++g) {
    f32x4 a[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const f32x4*>(a_lds + (mt * 16 + r) * LDA + q * KQ + 4 * g);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][j], b[g][j], acc[mt], 0, 0, 0);
  }
}

template <int DIN>
struct Node16Dims {
  static constexpr int KV = DIN + 64;                       // valid k of the node-MLP input
  static constexpr int KP = (KV + 15) / 16 * 16;            // padded: 96 (Din 20) or 128 (Din 64)
  static constexpr int LD1 = KP + 4;                        // 100 / 132 (LD/4 odd)
  static constexpr int KQ1 = KP / 4;                        // k per quarter: 24 / 32
};

// ---------------------------------------------------------------------------------------------------------------
// Operand packs.  The node kernels keep their MFMA B operands in registers; fetched from the NATIVE parameter tensors
// every wave-level load touches 64 different cache lines (16 rows x 4 k-quarters, 16 bytes used of each 64-byte line) --
// stage stamps showed that phase to be 60 % (forward) / 37 % (backward) of the kernels.  A tiny kernel run once per
// step and layer rewrites the weights in exactly the order the lanes consume them (pack[wave][slot][lane][4 floats]):
// each operand load of the node kernels is then one fully coalesced 1 KB access.
//   forward slots : b1 (KQ1/4 groups) | b2 (4) | b3 (2 x 4)                         -> NODE_FWD_SLOTS = 20
//   backward slots: bp (8 groups = 32 k) | ba (4) | bx (2 x 4)                      -> NODE_BWD_SLOTS = 20
constexpr int NODE_FWD_SLOTS = 20, NODE_BWD_SLOTS = 20;
constexpr int NODE_PACK_FLOATS = 4 * 20 * 64 * 4;     // per direction and layer

struct NodePackJob {
  const float *Wn1, *Wn2, *W1n;     // W1n may be NULL (last layer without a projection head)
  float *fpack, *bpack;
  int din, ldw_n, pad0, pad1;
};
constexpr int NODE_PACK_MAX = 8;
struct NodePackBatch { NodePackJob job[NODE_PACK_MAX]; };

template <int DIN>
__device__ __forceinline__ void node_pack_body(const NodePackJob& J, int wave, int lane) {
  using D = Node16Dims<DIN>;
  const int r = lane & 15, q = lane >> 4;
  const int col = wave * 16 + r;
  f32x4* fp = reinterpret_cast<f32x4*>(J.fpack) + (size_t)wave * NODE_FWD_SLOTS * 64 + lane;
  f32x4* bp = reinterpret_cast<f32x4*>(J.bpack) + (size_t)wave * NODE_BWD_SLOTS * 64 + lane;
  // every slot is gathered into registers first and stored afterwards: the 160 scattered loads of a lane are then in
  // flight together (interleaved with the stores they would be serialised by possible aliasing)
  const float* __restrict__ Wn1 = J.Wn1;
  const float* __restrict__ Wn2 = J.Wn2;
  const float* __restrict__ W1n = J.W1n;
  const int ldw_n = J.ldw_n;
  f32x4 fv[NODE_FWD_SLOTS], bv[NODE_BWD_SLOTS];
#pragma unroll
  for (int i = 0; i < NODE_FWD_SLOTS; ++i) fv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NODE_BWD_SLOTS; ++i) bv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // ---- forward ----
  constexpr int G1 = D::KQ1 / 4;
#pragma unroll
  for (int g = 0; g < G1; ++g)
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int k = q * D::KQ1 + 4 * g + j; fv[g][j] = (k < D::KV) ? Wn1[(size_t)col * D::KV + k] : 0.0f; }
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int j = 0; j < 4; ++j) fv[G1 + g][j] = Wn2[(size_t)col * H + q * 16 + 4 * g + j];
  if (W1n != nullptr) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int c = wave * 32 + nt * 16 + r;
      const float* row = (c < 64) ? W1n + (size_t)c * ldw_n : W1n + (size_t)(c - 64) * ldw_n + 64;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) fv[G1 + 4 + nt * 4 + g][j] = row[q * 16 + 4 * g + j];
    }
  }
  // ---- backward (transposed operands) ----
  if (W1n != nullptr) {
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = q * 32 + 4 * g + j;
        bv[g][j] = (c < 64) ? W1n[(size_t)c * ldw_n + col] : W1n[(size_t)(c - 64) * ldw_n + 64 + col];
      }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[8 + g][j] = Wn2[(size_t)(q * 16 + 4 * g + j) * H + col];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int xc = (wave * 2 + nt) * 16 + r;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[12 + nt * 4 + g][j] = (xc < D::KV) ? Wn1[(size_t)(q * 16 + 4 * g + j) * D::KV + xc] : 0.0f;
  }
  constexpr int NF = G1 + 4 + 8;      // forward slots in use (the rest of the 20 stay unwritten, as before)
#pragma unroll
  for (int i = 0; i < NF; ++i) fp[i * 64] = fv[i];
#pragma unroll
  for (int i = 0; i < NODE_BWD_SLOTS; ++i) bp[i * 64] = bv[i];
}

__global__ __launch_bounds__(64) void node_pack_kernel(NodePackBatch batch) {
  const NodePackJob& J = batch.job[blockIdx.y];
  if (J.din == 20) node_pack_body<20>(J, blockIdx.x, threadIdx.x);
  else node_pack_body<64>(J, blockIdx.x, threadIdx.x);
}

// The stack's prologue in ONE launch: blocks [0, proj_blocks) compute the layer-0 pre-projection psd = [h W1s^T + b0 |
// h W1d^T + b1] (the body of node_proj_fwd_kernel, csrc/egnn_node.hip: lane = channel, a wave walks nodes), the
// remaining njobs blocks write the operand packs (wave = former blockIdx.x of node_pack_kernel).  The two are
// independent -- the packs depend on the weights only -- so they run side by side instead of back to back.
template <int DIN>
__device__ __forceinline__ void stack_proj_body(const float* __restrict__ h, int ld_h, const float* __restrict__ W1, int ldw,
                                                const float* __restrict__ b0, const float* __restrict__ b1,
                                                float* __restrict__ psd, int N, int block, int nblocks) {
  const int lane = threadIdx.x & 63;
  float ws[DIN], wd[DIN];
#pragma unroll
  for (int k = 0; k < DIN; ++k) {
    ws[k] = W1[lane * ldw + k];
    wd[k] = W1[lane * ldw + DIN + k];
  }
  const float bias = b1[lane];
  const float bias0 = b0 != nullptr ? b0[lane] : 0.0f;
  for (int n = block * 4 + (threadIdx.x >> 6); n < N; n += nblocks * 4) {
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

__global__ __launch_bounds__(256) void stack_prologue_kernel(NodePackBatch batch, int proj_blocks, const float* __restrict__ h,
                                                             int ld_h, int din, const float* __restrict__ W1, int ldw,
                                                             const float* __restrict__ b0, const float* __restrict__ b1,
                                                             float* __restrict__ psd, int N) {
  if ((int)blockIdx.x < proj_blocks) {
    if (din == 20) stack_proj_body<20>(h, ld_h, W1, ldw, b0, b1, psd, N, blockIdx.x, proj_blocks);
    else stack_proj_body<64>(h, ld_h, W1, ldw, b0, b1, psd, N, blockIdx.x, proj_blocks);
    return;
  }
  const NodePackJob& J = batch.job[blockIdx.x - proj_blocks];
  if (J.din == 20) node_pack_body<20>(J, threadIdx.x >> 6, threadIdx.x & 63);
  else node_pack_body<64>(J, threadIdx.x >> 6, threadIdx.x & 63);
}

template <int DIN>
__global__ __launch_bounds__(256) void egnn_node_fwd16_kernel(
    const float* __restrict__ h, int ld_h, const float* __restrict__ h_neigh, int ld_hn,
    const float* __restrict__ Wn1, const float* __restrict__ bn1, const float* __restrict__ Wn2,
    const float* __restrict__ bn2, const float* __restrict__ W1n, int ldw_n, const float* __restrict__ b0n,
    const float* __restrict__ b1n, float* __restrict__ zn1, float* __restrict__ h_out, float* __restrict__ psd_next, int N,
    const float* __restrict__ fpack) {
  using D = Node16Dims<DIN>;
  __shared__ float xs[32 * D::LD1];
  __shared__ float a1s[32 * LD];
  __shared__ float hps[32 * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int row0 = blockIdx.x * 32;
  const bool has_next = W1n != nullptr;
  STAMPN(0);

  // ---- B operands of all three layers for this wave's output columns (independent of the tile data) ----
  const int col = wave * 16 + r;                       // output column of MM_a / MM_b
  f32x4 b1[D::KQ1 / 4], b2[4], b3[2][4];
  if (fpack != nullptr) {      // operand pack: every load is one coalesced 1 KB access
    const f32x4* fp = reinterpret_cast<const f32x4*>(fpack) + (size_t)wave * NODE_FWD_SLOTS * 64 + lane;
#pragma unroll
    for (int g = 0; g < D::KQ1 / 4; ++g) b1[g] = fp[g * 64];
#pragma unroll
    for (int g = 0; g < 4; ++g) b2[g] = fp[(D::KQ1 / 4 + g) * 64];
    if (has_next) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) b3[nt][g] = fp[(D::KQ1 / 4 + 4 + nt * 4 + g) * 64];
    }
  } else {
    const float* wrow = Wn1 + (size_t)col * D::KV;
    const int al = align_of(Wn1, D::KV);
#pragma unroll
    for (int g = 0; g < D::KQ1 / 4; ++g) {
      const int k = q * D::KQ1 + 4 * g;
      b1[g] = (k < D::KV) ? ldg4(wrow + k, al) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float* w2row = Wn2 + (size_t)col * H;
#pragma unroll
    for (int g = 0; g < 4; ++g) b2[g] = *reinterpret_cast<const f32x4*>(w2row + q * 16 + 4 * g);
    if (has_next) {
      const int al3 = align_of(W1n, ldw_n);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int c = wave * 32 + nt * 16 + r;          // psd column: [0,64) = Ps, [64,128) = Pd
        const float* w3row = (c < 64) ? W1n + (size_t)c * ldw_n : W1n + (size_t)(c - 64) * ldw_n + 64;
        const int al3c = (c < 64) ? al3 : ((al3 == 4 || al3 == 2) ? al3 : 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) b3[nt][g] = ldg4(w3row + q * 16 + 4 * g, al3c);
      }
    }
  }
  const float bn1_c = bn1[col], bn2_c = bn2[col];
  float b1n_c[2] = {0.f, 0.f};
  if (has_next) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int c = wave * 32 + nt * 16 + r;
      b1n_c[nt] = c >= 64 ? b1n[c - 64] : (b0n != nullptr ? b0n[c] : 0.0f);
    }
  }

  // ---- X = [h | h_neigh | 0] rows of the tile -> LDS (wave w stages rows 8w .. 8w+7) ----
  {
    // all 16 row loads first (unconditional: clamped row / column), then the LDS stores: one memory round trip instead of
    // a load -> store chain per row (the predicated form kept the compiler from hoisting the loads)
    float hv[8], nv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = min(row0 + wave * 8 + i, N - 1);
      hv[i] = h[(size_t)row * ld_h + min(lane, DIN - 1)];
      nv[i] = h_neigh[(size_t)row * ld_hn + lane];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int lr = wave * 8 + i;
      const bool valid = row0 + lr < N;
      if (lane < DIN) xs[lr * D::LD1 + lane] = valid ? hv[i] : 0.0f;
      xs[lr * D::LD1 + DIN + lane] = valid ? nv[i] : 0.0f;
      if (lane < D::KP - D::KV) xs[lr * D::LD1 + D::KV + lane] = 0.0f;
    }
  }
  STAMPN(1);
  __syncthreads();
  STAMPN(2);

  // ---- zn1 = X Wn1^T + bn1 ; a1 = SiLU(zn1) ----
  {
    f32x4 acc[2];
    zero_acc4(acc);
    mm16_regB<2, D::KQ1, D::LD1>(acc, xs, b1, lane);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int lr = mt * 16 + tile16_row(t, q);
        const float z = acc[mt][t] + bn1_c;
        if (zn1 != nullptr && row0 + lr < N) zn1[(size_t)(row0 + lr) * H + col] = z;
        a1s[lr * LD + col] = silu_f(z);
      }
  }
  STAMPN(3);
  __syncthreads();
  STAMPN(4);
  // ---- h' = a1 Wn2^T + bn2 ----
  {
    f32x4 acc[2];
    zero_acc4(acc);
    mm16_regB<2, 16, LD>(acc, a1s, b2, lane);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int lr = mt * 16 + tile16_row(t, q);
        const float v = acc[mt][t] + bn2_c;
        if (row0 + lr < N) h_out[(size_t)(row0 + lr) * H + col] = v;
        hps[lr * LD + col] = v;
      }
  }
  STAMPN(5);
  if (!has_next) return;
  __syncthreads();
  STAMPN(6);
  // ---- next layer's node pre-projection: wave w produces psd columns [32w, 32w + 32) ----
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    f32x4 acc[2];
    zero_acc4(acc);
    mm16_regB<2, 16, LD>(acc, hps, b3[nt], lane);
    const int c = wave * 32 + nt * 16 + r;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int lr = mt * 16 + tile16_row(t, q);
        if (row0 + lr < N) psd_next[(size_t)(row0 + lr) * 128 + c] = acc[mt][t] + b1n_c[nt];
      }
  }
  STAMPN(7);
}


// acc[mt] += A[mt*16 + i][k] * Bt[k][j] with Bt given per k (scalar registers: transposed weights).
template <int MT, int KQ, int LDA>
__device__ __forceinline__ void mm16_regBt(f32x4 (&acc)[MT], const float* a_lds, const float (&bt)[KQ], int lane) {
  const int r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int g = 0; g < KQ / 4; ++g) {
    f32x4 a[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const f32x4*>(a_lds + (mt * 16 + r) * LDA + q * KQ + 4 * g);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][j], bt[4 * g + j], acc[mt], 0, 0, 0);
  }
}

// Backward data path of one layer's node block for one 32-row tile:
//   dh   = g_h + g_psd W1sd          (only when g_psd != NULL; W1sd = next layer's [W1s ; W1d])
//   dzn1 = (dh Wn2) * SiLU'(zn1)
//   dX   = dzn1 Wn1   ->  d_h (first DIN columns, optional) , d_hneigh (last 64 columns)
// dh (when computed) and dzn1 are written to HBM for the weight-gradient kernel.
template <int DIN>
__global__ __launch_bounds__(256) void egnn_node_bwd_data16_kernel(
    const float* __restrict__ g_h, const float* __restrict__ g_psd, const float* __restrict__ W1n, int ldw_n,
    const float* __restrict__ zn1, const float* __restrict__ Wn1, const float* __restrict__ Wn2,
    float* __restrict__ dh_total, float* __restrict__ dzn1, float* __restrict__ d_h, float* __restrict__ d_hneigh,
    int N, const float* __restrict__ bpack) {
  using D = Node16Dims<DIN>;
  constexpr int LDP = 132;
  __shared__ float ps_[32 * LDP];
  __shared__ float gs[32 * LD];
  __shared__ float zs[32 * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int row0 = blockIdx.x * 32;
  const bool has_psd = g_psd != nullptr;
  const int col = wave * 16 + r;
  STAMPN(8);

  // ---- transposed-weight operands (independent of the tile data) ----
  float bp[32], ba[16], bx[2][16];
  if (bpack != nullptr) {      // operand pack: 20 coalesced 16-byte loads per lane instead of 80 scattered dwords
    const f32x4* pk = reinterpret_cast<const f32x4*>(bpack) + (size_t)wave * NODE_BWD_SLOTS * 64 + lane;
    if (has_psd) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const f32x4 v = pk[g * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) bp[4 * g + j] = v[j];
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = pk[(8 + g) * 64];
#pragma unroll
      for (int j = 0; j < 4; ++j) ba[4 * g + j] = v[j];
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = pk[(12 + nt * 4 + g) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) bx[nt][4 * g + j] = v[j];
      }
  } else {
    if (has_psd) {
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const int c = q * 32 + s;   // psd column; quarters 0,1 -> Ps rows, 2,3 -> Pd rows
        bp[s] = (c < 64) ? W1n[(size_t)c * ldw_n + col] : W1n[(size_t)(c - 64) * ldw_n + 64 + col];
      }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) ba[s] = Wn2[(size_t)(q * 16 + s) * H + col];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int xc = (wave * 2 + nt) * 16 + r;      // column of dX = [d_h | d_hneigh]
#pragma unroll
      for (int s = 0; s < 16; ++s) bx[nt][s] = (xc < D::KV) ? Wn1[(size_t)(q * 16 + s) * D::KV + xc] : 0.0f;
    }
  }
  // the epilogue inputs of this lane (rows mt*16 + 4q + t, column col): fetched now, consumed two / three stages later
  float zpre[2][4], gpre[2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = min(row0 + mt * 16 + tile16_row(t, q), N - 1);
      zpre[mt][t] = zn1[(size_t)row * H + col];
      gpre[mt][t] = (has_psd && g_h != nullptr) ? g_h[(size_t)row * H + col] : 0.0f;
    }

  // ---- stage g_psd (or g_h) rows: wave w stages rows 8w .. 8w+7 ----
  {
    float v0[8], v1[8];       // loads first (clamped rows), LDS stores afterwards: one memory round trip
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = min(row0 + wave * 8 + i, N - 1);
      if (has_psd) {
        v0[i] = g_psd[(size_t)row * 128 + lane];
        v1[i] = g_psd[(size_t)row * 128 + 64 + lane];
      } else {
        v0[i] = g_h[(size_t)row * H + lane];
        v1[i] = 0.0f;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int lr = wave * 8 + i;
      const bool valid = row0 + lr < N;
      if (has_psd) {
        ps_[lr * LDP + lane] = valid ? v0[i] : 0.0f;
        ps_[lr * LDP + 64 + lane] = valid ? v1[i] : 0.0f;
      } else {
        gs[lr * LD + lane] = valid ? v0[i] : 0.0f;
      }
    }
  }
  STAMPN(9);
  __syncthreads();
  STAMPN(10);
  if (has_psd) {
    f32x4 acc[2];
    zero_acc4(acc);
    mm16_regBt<2, 32, LDP>(acc, ps_, bp, lane);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int lr = mt * 16 + tile16_row(t, q), row = row0 + lr;
        float v = 0.0f;
        if (row < N) {
          v = acc[mt][t] + gpre[mt][t];
          dh_total[(size_t)row * H + col] = v;
        }
        gs[lr * LD + col] = v;
      }
    __syncthreads();
  }
  // ---- da1 = dh Wn2 ; dzn1 = da1 * SiLU'(zn1) ----
  {
    f32x4 acc[2];
    zero_acc4(acc);
    mm16_regBt<2, 16, LD>(acc, gs, ba, lane);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int lr = mt * 16 + tile16_row(t, q), row = row0 + lr;
        float dz = 0.0f;
        if (row < N) {
          float y, dy;
          silu_fg(zpre[mt][t], y, dy);
          dz = acc[mt][t] * dy;
          dzn1[(size_t)row * H + col] = dz;
        }
        zs[lr * LD + col] = dz;
      }
  }
  STAMPN(11);
  __syncthreads();
  // ---- dX = dzn1 Wn1 ----
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int xc = (wave * 2 + nt) * 16 + r;
    if ((wave * 2 + nt) * 16 < D::KV) {          // wave-uniform: column tile exists
      f32x4 acc[2];
      zero_acc4(acc);
      mm16_regBt<2, 16, LD>(acc, zs, bx[nt], lane);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = row0 + mt * 16 + tile16_row(t, q);
          if (row < N && xc < D::KV) {
            if (xc < DIN) { if (d_h != nullptr) d_h[(size_t)row * H + xc] = acc[mt][t]; }
            else d_hneigh[(size_t)row * H + (xc - DIN)] = acc[mt][t];
          }
        }
    }
  }
  STAMPN(12);
}

// Streaming weight-gradient kernel of one layer's node block (outer products over the N rows):
//   dW1sd[c][i] = sum_n g_psd[n][c] h_out[n][i]   (128 x 64, next layer's edge_mlp.0 node part; optional)
//   dWn2[o][i]  = sum_n dh[n][o] SiLU(zn1[n][i])  (64 x 64)
//   dWn1[o][k]  = sum_n dzn1[n][o] [h | h_neigh][n][k]   (64 x 128, h part padded to 64 columns)
//   + the column sums db0/db1 (g_psd), dbn2 (dh), dbn1 (dzn1).
// Workgroup g owns a contiguous slice of rows; wave w owns output row tiles (no cross-wave reduction of
// the matrices).  Partial record = [PROJ part | NODE part] with the layouts of the v1 kernels.
constexpr int WG_PROJ = 128 * 64 + 128;
constexpr int WG_NODE = 64 * 128 + 64 * 64 + 128;
constexpr int WG_STRIDE = WG_PROJ + WG_NODE;

template <int DIN>
__device__ __forceinline__ void egnn_node_wgrad16_body(
    const float* __restrict__ g_psd, const float* __restrict__ h_out, int ld_ho, int dho, const float* __restrict__ dh,
    const float* __restrict__ zn1, const float* __restrict__ dzn1, const float* __restrict__ h, int ld_h,
    const float* __restrict__ h_neigh, int ld_hn, float* __restrict__ partials, int N, int rows_per_wg) {
  constexpr int LDP = 132;
  __shared__ float Ps[16 * LDP], Xs[16 * LDP];
  __shared__ float Hs[16 * LD], Gs[16 * LD], As[16 * LD], Zs[16 * LD];
  __shared__ float vec[4][4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const bool has_psd = g_psd != nullptr;
  const bool has_node = dzn1 != nullptr;     // false: a projection-only job (layer-0 pre-projection): NODE part left untouched
  const int r_begin = blockIdx.x * rows_per_wg, r_end = min(N, r_begin + rows_per_wg);

  f32x4 dW1[2][4], dW2[4], dWn[8];
#pragma unroll
  for (int a = 0; a < 2; ++a) zero_acc4(dW1[a]);
  zero_acc4(dW2);
  zero_acc4(dWn);
  float s_p0 = 0.f, s_p1 = 0.f, s_g = 0.f, s_z = 0.f;   // lane = column partial sums

  // rows are fetched one chunk ahead into registers so that the loads of chunk c+1 are in flight
  // while the MFMAs of chunk c run
  float rp0[4], rp1[4], rho[4], rg[4], rz[4], rzn[4], rxh[4], rxn[4];
  auto fetch = [&](int c0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = c0 + wave * 4 + i;
      const bool valid = row < r_end;
      rp0[i] = (has_psd && valid) ? g_psd[(size_t)row * 128 + lane] : 0.0f;
      rp1[i] = (has_psd && valid) ? g_psd[(size_t)row * 128 + 64 + lane] : 0.0f;
      rho[i] = (has_psd && valid && lane < dho) ? h_out[(size_t)row * ld_ho + lane] : 0.0f;
      rg[i] = (has_node && valid) ? dh[(size_t)row * H + lane] : 0.0f;
      rz[i] = (has_node && valid) ? dzn1[(size_t)row * H + lane] : 0.0f;
      rzn[i] = (has_node && valid) ? zn1[(size_t)row * H + lane] : 0.0f;
      rxh[i] = (has_node && valid && lane < DIN) ? h[(size_t)row * ld_h + lane] : 0.0f;
      rxn[i] = (has_node && valid) ? h_neigh[(size_t)row * ld_hn + lane] : 0.0f;
    }
  };
  if (r_begin < r_end) fetch(r_begin);
  for (int c0 = r_begin; c0 < r_end; c0 += 16) {
    __syncthreads();
    // ---- stage 16 rows: wave w stages rows 4w .. 4w+3 (lane = column) ----
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int lr = wave * 4 + i;
      const bool valid = c0 + lr < r_end;
      if (has_psd) {
        Ps[lr * LDP + lane] = rp0[i]; Ps[lr * LDP + 64 + lane] = rp1[i];
        s_p0 += rp0[i]; s_p1 += rp1[i];
        Hs[lr * LD + lane] = rho[i];
      }
      if (has_node) {
        Gs[lr * LD + lane] = rg[i]; Zs[lr * LD + lane] = rz[i];
        s_g += rg[i]; s_z += rz[i];
        As[lr * LD + lane] = valid ? silu_f(rzn[i]) : 0.0f;
        Xs[lr * LDP + lane] = rxh[i];
        Xs[lr * LDP + 64 + lane] = rxn[i];
      }
    }
    __syncthreads();
    if (c0 + 16 < r_end) fetch(c0 + 16);
    // ---- outer products: contraction over the 16 staged rows ----
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int e = 4 * q + s;
      float bh[4];
      if (has_node) {
        const float ag = Gs[e * LD + wave * 16 + r], az = Zs[e * LD + wave * 16 + r];
        float ba1[4], bxv[8];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) ba1[nt] = As[e * LD + nt * 16 + r];
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) bxv[nt] = Xs[e * LDP + nt * 16 + r];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) dW2[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag, ba1[nt], dW2[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) dWn[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(az, bxv[nt], dWn[nt], 0, 0, 0);
      }
      if (has_psd) {
        const float ap0 = Ps[e * LDP + (2 * wave) * 16 + r], ap1 = Ps[e * LDP + (2 * wave + 1) * 16 + r];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bh[nt] = Hs[e * LD + nt * 16 + r];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          dW1[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap0, bh[nt], dW1[0][nt], 0, 0, 0);
          dW1[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap1, bh[nt], dW1[1][nt], 0, 0, 0);
        }
      }
    }
  }
  // ---- partial record ----
  float* part = partials + (size_t)blockIdx.x * WG_STRIDE;
  float* pn = part + WG_PROJ;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int lr = tile16_row(t, q);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      if (has_psd) {
        part[((2 * wave) * 16 + lr) * H + nt * 16 + r] = dW1[0][nt][t];
        part[((2 * wave + 1) * 16 + lr) * H + nt * 16 + r] = dW1[1][nt][t];
      }
      if (has_node) pn[64 * 128 + (wave * 16 + lr) * H + nt * 16 + r] = dW2[nt][t];
    }
    if (has_node) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) pn[(wave * 16 + lr) * 128 + nt * 16 + r] = dWn[nt][t];
    }
  }
  vec[wave][0][lane] = s_p1; vec[wave][1][lane] = s_p0; vec[wave][2][lane] = s_z; vec[wave][3][lane] = s_g;
  __syncthreads();
  {
    const int which = tid >> 6;   // 0: db1, 1: db0, 2: dbn1, 3: dbn2
    const float v = ((vec[0][which][lane] + vec[1][which][lane]) + vec[2][which][lane]) + vec[3][which][lane];
    if (which < 2) { if (has_psd) part[128 * 64 + which * 64 + lane] = v; }
    else if (has_node) pn[64 * 128 + 64 * 64 + (which - 2) * 64 + lane] = v;
  }
}


template <int DIN>
__global__ __launch_bounds__(256) void egnn_node_wgrad16_kernel(
    const float* __restrict__ g_psd, const float* __restrict__ h_out, const float* __restrict__ dh,
    const float* __restrict__ zn1, const float* __restrict__ dzn1, const float* __restrict__ h, int ld_h,
    const float* __restrict__ h_neigh, int ld_hn, float* __restrict__ partials, int N, int rows_per_wg) {
  egnn_node_wgrad16_body<DIN>(g_psd, h_out, H, H, dh, zn1, dzn1, h, ld_h, h_neigh, ld_hn, partials, N, rows_per_wg);
}

// All layers of a stack in ONE launch (blockIdx.y = layer): 6 x 254 workgroups instead of six launches of 254
// single-wave-per-SIMD workgroups -- the co-resident workgroups of different layers hide each other's latency.
struct WgradLayer {
  const float *g_psd, *h_out, *dh, *zn1, *dzn1, *h, *h_neigh;
  float* partials;
  int ld_h, din, ld_hn, ld_ho, dho, pad;     // h_out: row stride and number of valid columns (64 / 64 for EGNN layers)
};
constexpr int WGRAD_MAX_LAYERS = 8;
struct WgradBatch { WgradLayer layer[WGRAD_MAX_LAYERS]; };

__global__ __launch_bounds__(256, 2) void egnn_node_wgrad16_batched_kernel(WgradBatch batch, int N, int rows_per_wg) {
  const WgradLayer& L = batch.layer[blockIdx.y];
  if (L.din == 20)
    egnn_node_wgrad16_body<20>(L.g_psd, L.h_out, L.ld_ho, L.dho, L.dh, L.zn1, L.dzn1, L.h, L.ld_h, L.h_neigh, L.ld_hn, L.partials, N, rows_per_wg);
  else
    egnn_node_wgrad16_body<64>(L.g_psd, L.h_out, L.ld_ho, L.dho, L.dh, L.zn1, L.dzn1, L.h, L.ld_h, L.h_neigh, L.ld_hn, L.partials, N, rows_per_wg);
}

}  // namespace is

#ifdef IS_STAGE_STAMPS
extern "C" int is_debug_stamps_node(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(is::g_stamps_node), sizeof(long long) * 16) == hipSuccess ? 0 : -5;
}
#endif

extern "C" int is_egnn_node_fwd_v2(const float* h, int ld_h, int din, const float* h_neigh, int ld_hn, const float* Wn1,
                                   const float* bn1, const float* Wn2, const float* bn2, const float* W1n, int ldw_n,
                                   const float* b0n, const float* b1n, float* zn1, float* h_out, float* psd_next, int N,
                                   const float* fpack, void* stream) {
  if (N <= 0) return 0;
  const dim3 grid((N + 31) / 32), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (din == 20) hipLaunchKernelGGL(is::egnn_node_fwd16_kernel<20>, grid, block, 0, st, h, ld_h, h_neigh, ld_hn, Wn1, bn1, Wn2, bn2, W1n, ldw_n, b0n, b1n, zn1, h_out, psd_next, N, fpack);
  else if (din == 64) hipLaunchKernelGGL(is::egnn_node_fwd16_kernel<64>, grid, block, 0, st, h, ld_h, h_neigh, ld_hn, Wn1, bn1, Wn2, bn2, W1n, ldw_n, b0n, b1n, zn1, h_out, psd_next, N, fpack);
  else return -22;
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int is_egnn_node_bwd_data(const float* g_h, const float* g_psd, const float* W1n, int ldw_n, const float* zn1,
                                     int din, const float* Wn1, const float* Wn2, float* dh_total, float* dzn1,
                                     float* d_h, float* d_hneigh, int N, const float* bpack, void* stream) {
  if (N <= 0) return 0;
  if (g_psd == nullptr && g_h == nullptr) return -22;
  const dim3 grid((N + 31) / 32), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (din == 20) hipLaunchKernelGGL(is::egnn_node_bwd_data16_kernel<20>, grid, block, 0, st, g_h, g_psd, W1n, ldw_n, zn1, Wn1, Wn2, dh_total, dzn1, d_h, d_hneigh, N, bpack);
  else if (din == 64) hipLaunchKernelGGL(is::egnn_node_bwd_data16_kernel<64>, grid, block, 0, st, g_h, g_psd, W1n, ldw_n, zn1, Wn1, Wn2, dh_total, dzn1, d_h, d_hneigh, N, bpack);
  else return -22;
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

// jobs: host array of njobs (<= 8) records { const float *Wn1, *Wn2, *W1n; float *fpack, *bpack; int din, ldw_n, pad0, pad1; }
// (W1n NULL: no next projection); fpack / bpack: is_node_pack_floats() floats each.  One launch for a whole stack.
extern "C" int is_node_pack_floats(void) { return is::NODE_PACK_FLOATS; }
extern "C" int is_node_pack_weights(const void* jobs, int njobs, void* stream) {
  if (njobs <= 0 || njobs > is::NODE_PACK_MAX) return -22;
  is::NodePackBatch batch;
  const is::NodePackJob* src = static_cast<const is::NodePackJob*>(jobs);
  for (int i = 0; i < njobs; ++i) {
    batch.job[i] = src[i];
    if (src[i].din != 20 && src[i].din != 64) return -22;
  }
  hipLaunchKernelGGL(is::node_pack_kernel, dim3(4, njobs), dim3(64), 0, static_cast<hipStream_t>(stream), batch);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

// is_node_proj_fwd (layer-0 pre-projection: h [N, ld_h] (din = 20 | 64 columns), W1 [64, ldw], b0 (may be NULL), b1 -> psd
// [N, 128]) and is_node_pack_weights (jobs as there) as ONE launch.
extern "C" int is_stack_prologue(const void* jobs, int njobs, const float* h, int ld_h, int din, const float* W1, int ldw,
                                 const float* b0, const float* b1, float* psd, int N, void* stream) {
  if (njobs <= 0 || njobs > is::NODE_PACK_MAX || N <= 0 || (din != 20 && din != 64)) return -22;
  is::NodePackBatch batch;
  const is::NodePackJob* src = static_cast<const is::NodePackJob*>(jobs);
  for (int i = 0; i < njobs; ++i) {
    batch.job[i] = src[i];
    if (src[i].din != 20 && src[i].din != 64) return -22;
  }
  const int proj_blocks = std::min((N + 3) / 4, 2048);
  hipLaunchKernelGGL(is::stack_prologue_kernel, dim3(proj_blocks + njobs), dim3(256), 0, static_cast<hipStream_t>(stream), batch,
                     proj_blocks, h, ld_h, din, W1, ldw, b0, b1, psd, N);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int is_egnn_node_wgrad_stride(void) { return is::WG_STRIDE; }
extern "C" int is_egnn_node_wgrad_proj_floats(void) { return is::WG_PROJ; }

// grid workgroups, each owning ceil(N / grid) rows rounded up to 16; partials: grid * is_egnn_node_wgrad_stride() floats
extern "C" int is_egnn_node_wgrad(const float* g_psd, const float* h_out, const float* dh, const float* zn1,
                                  const float* dzn1, const float* h, int ld_h, int din, const float* h_neigh,
                                  int ld_hn, float* partials, int grid, int N, void* stream) {
  if (N <= 0 || grid <= 0) return -22;
  const int rows = (((N + grid - 1) / grid) + 15) / 16 * 16;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (din == 20) hipLaunchKernelGGL(is::egnn_node_wgrad16_kernel<20>, dim3(grid), dim3(256), 0, st, g_psd, h_out, dh, zn1, dzn1, h, ld_h, h_neigh, ld_hn, partials, N, rows);
  else if (din == 64) hipLaunchKernelGGL(is::egnn_node_wgrad16_kernel<64>, dim3(grid), dim3(256), 0, st, g_psd, h_out, dh, zn1, dzn1, h, ld_h, h_neigh, ld_hn, partials, N, rows);
  else return -22;
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

// layers: host array of `nlayers` (<= 8) records {g_psd, h_out, dh, zn1, dzn1, h, h_neigh, partials, ld_h, din, ld_hn, ld_ho,
// dho, pad} (pointers first, then six ints); every layer uses `grid` workgroups and the record layout of
// is_egnn_node_wgrad.  A record with dzn1 == NULL is a projection-only job (PROJ part from g_psd and the first dho
// columns of h_out, row stride ld_ho); a record with g_psd == NULL leaves the PROJ part untouched.
extern "C" int is_egnn_node_wgrad_batched(const void* layers, int nlayers, int grid, int N, void* stream) {
  if (N <= 0 || grid <= 0 || nlayers <= 0 || nlayers > is::WGRAD_MAX_LAYERS) return -22;
  is::WgradBatch batch;
  const is::WgradLayer* src = static_cast<const is::WgradLayer*>(layers);
  for (int i = 0; i < nlayers; ++i) {
    batch.layer[i] = src[i];
    if ((src[i].din != 20 && src[i].din != 64) || src[i].dho < 0 || src[i].dho > 64) return -22;
    if (src[i].g_psd == nullptr && src[i].dzn1 == nullptr) return -22;
  }
  const int rows = (((N + grid - 1) / grid) + 15) / 16 * 16;
  hipLaunchKernelGGL(is::egnn_node_wgrad16_batched_kernel, dim3(grid, nlayers), dim3(256), 0, static_cast<hipStream_t>(stream), batch, N, rows);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}
