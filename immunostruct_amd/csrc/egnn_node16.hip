// Node-level pieces that are their own launches: the stack prologue (layer-0 pre-projection + operand packs of every
// layer's node half) and the batched weight gradients of the node blocks.  The node MLP forward / backward data paths
// run inside the fused layer kernels (egnn_layer_fwd.hip / egnn_layer_bwd.hip).
//   * operand packs: the node halves keep their MFMA B operands in registers; is_stack_prologue rewrites the weights in
//     exactly the order the lanes consume them, so every operand load is one coalesced 1 KB access (node16.h);
//   * the weight gradients of a layer's node block are streaming outer products over the rows, from the tensors the data
//     path leaves in HBM; all layers of a stack run as ONE launch.
#include <algorithm>
#include "common.h"
#include "node16.h"

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

struct NodePackJob {
  const float *Wn1, *Wn2;
  const float *W1n, *W1nb;          // next pre-projection: rows of its source half / destination half (row stride ldw_n); NULL:
                                    // last layer without a projection head.  A layer's edge_mlp.0.weight: W1nb = W1n + 64; a
                                    // [Wq | Wk] head: the two matrices where they are (no concatenation in front)
  float *fpack, *bpack;
  int din, ldw_n;
};
constexpr int NODE_PACK_MAX = 8;
struct NodePackBatch { NodePackJob job[NODE_PACK_MAX]; };

template <int DIN>
__device__ __forceinline__ void node_pack_body(const NodePackJob& J, int wave, int lane) {
  using D = Node16Dims<DIN>;
  const int r = lane & 15, q = lane >> 4;
  const int col = wave * 16 + r;
  f32x4* __restrict__ fp = reinterpret_cast<f32x4*>(J.fpack) + (size_t)wave * NODE_FWD_SLOTS * 64 + lane;
  f32x4* __restrict__ bp = reinterpret_cast<f32x4*>(J.bpack) + (size_t)wave * NODE_BWD_SLOTS * 64 + lane;
  // the slots of a pack are gathered into registers first and stored afterwards: the ~ 80 scattered loads of a lane are then in
  // flight together (interleaved with the stores they would be serialised by possible aliasing).  Forward pack, then backward
  // pack: both at once need 160 registers, and this kernel's register count is what lets the projection workgroups of the
  // same launch fit four to a CU (the whole launch in one round: 14 -> 7 us)
  const float* __restrict__ Wn1 = J.Wn1;
  const float* __restrict__ Wn2 = J.Wn2;
  const float* __restrict__ W1n = J.W1n;
  const float* __restrict__ W1nb = J.W1nb;
  const int ldw_n = J.ldw_n;
  f32x4 fv[NODE_FWD_SLOTS];
#pragma unroll
  for (int i = 0; i < NODE_FWD_SLOTS; ++i) fv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
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
      const float* row = (c < 64) ? W1n + (size_t)c * ldw_n : W1nb + (size_t)(c - 64) * ldw_n;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) fv[G1 + 4 + nt * 4 + g][j] = row[q * 16 + 4 * g + j];
    }
  }
  constexpr int NF = G1 + 4 + 8;      // forward slots in use (the rest of the 20 stay unwritten, as before)
#pragma unroll
  for (int i = 0; i < NF; ++i) fp[i * 64] = fv[i];
  // ---- backward (transposed operands) ----
  f32x4 bv[NODE_BWD_SLOTS];
#pragma unroll
  for (int i = 0; i < NODE_BWD_SLOTS; ++i) bv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (W1n != nullptr) {
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = q * 32 + 4 * g + j;
        bv[g][j] = (c < 64) ? W1n[(size_t)c * ldw_n + col] : W1nb[(size_t)(c - 64) * ldw_n + col];
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
#pragma unroll
  for (int i = 0; i < NODE_BWD_SLOTS; ++i) bp[i * 64] = bv[i];
}

// The stack's prologue in ONE launch: njobs blocks write the operand packs, proj_blocks blocks compute the layer-0
// pre-projection psd = [h W1s^T + b0 | h W1d^T + b1] (lane = channel, a wave walks nodes; bit-identical to
// node_proj_fwd_kernel of csrc/egnn_node.hip: same k order) and the dense copy of the coordinates.  The two are
// independent -- the packs depend on the weights only -- so they run side by side instead of back to back.
template <int DIN>
__device__ __forceinline__ void stack_proj_body(const float* __restrict__ h, int ld_h, const float* __restrict__ W1, int ldw,
                                                const float* __restrict__ b0, const float* __restrict__ b1,
                                                float* __restrict__ psd, const float* __restrict__ x_src, int ld_x,
                                                float* __restrict__ x_dst, int N, int block, int nblocks, float* wl) {
  // the [W1s | W1d] column blocks go through LDS once per workgroup (near-coalesced row segments; a lane reading its own row
  // straight from global memory touches 64 cache lines per load), row stride odd => conflict-free column reads
  constexpr int LDWL = 2 * DIN + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // a wave walks nodes n = (block * 4 + wave) + i * stride in groups of G: the group's feature rows are in flight together, and
  // the FIRST group's (usually the only one) are requested before the weights are staged: one round trip instead of two
  constexpr int G = 6;
  const int stride = nblocks * 4;
  float hv[G], xv[G];
  auto fetch = [&](int n0) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int n = n0 + g * stride;
      hv[g] = (n < N && lane < DIN) ? h[(size_t)n * ld_h + lane] : 0.0f;
      xv[g] = (x_dst != nullptr && n < N && lane < 3) ? x_src[(size_t)n * ld_x + lane] : 0.0f;
    }
  };
  fetch(block * 4 + wave);
  {
    constexpr int CNT = 64 * 2 * DIN, PER = (CNT + 255) / 256;
    float v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = tid + j * 256;
      v[j] = (i < CNT) ? W1[(i / (2 * DIN)) * ldw + i % (2 * DIN)] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = tid + j * 256;
      if (i < CNT) wl[(i / (2 * DIN)) * LDWL + i % (2 * DIN)] = v[j];
    }
  }
  const float bias = b1[lane];
  const float bias0 = b0 != nullptr ? b0[lane] : 0.0f;
  __syncthreads();
  float ws[DIN], wd[DIN];
#pragma unroll
  for (int k = 0; k < DIN; ++k) {
    ws[k] = wl[lane * LDWL + k];
    wd[k] = wl[lane * LDWL + DIN + k];
  }
  for (int n0 = block * 4 + wave; n0 < N; n0 += G * stride) {
    if (n0 != block * 4 + wave) fetch(n0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int n = n0 + g * stride;
      if (n < N) {      // wave-uniform
        float as = bias0, ad = bias;
#pragma unroll
        for (int k = 0; k < DIN; ++k) {
          // feature k of the node lives in lane k: a scalar broadcast (v_readlane), not a trip through the LDS crossbar
          const float hk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hv[g]), k));
          as += hk * ws[k];
          ad += hk * wd[k];
        }
        psd[(size_t)n * 128 + lane] = as;
        psd[(size_t)n * 128 + 64 + lane] = ad;
        if (x_dst != nullptr && lane < 3) x_dst[n * 3 + lane] = xv[g];      // dense copy of strided coordinates
      }
    }
  }
}

// blocks [0, 4 njobs) write the operand packs (first: their scattered loads are the longest chain of the launch),
// the remaining proj_blocks compute the layer-0 pre-projection
template <int PDIN>      // the projection's input width: the 64-wide form keeps 128 weight registers per lane (two workgroups per CU)
__global__ __launch_bounds__(256, PDIN == 20 ? 4 : 2) void stack_prologue_kernel(NodePackBatch batch, int njobs, int proj_blocks, const float* __restrict__ h,
                                                             int ld_h, const float* __restrict__ W1, int ldw,
                                                             const float* __restrict__ b0, const float* __restrict__ b1,
                                                             float* __restrict__ psd, const float* __restrict__ x_src, int ld_x,
                                                             float* __restrict__ x_dst, int N) {
  __shared__ float wl[64 * 129];
  if ((int)blockIdx.x >= 4 * njobs) {
    const int blk = blockIdx.x - 4 * njobs;
    stack_proj_body<PDIN>(h, ld_h, W1, ldw, b0, b1, psd, x_src, ld_x, x_dst, N, blk, proj_blocks, wl);
    return;
  }
  // a pack is written by four workgroups of ONE wave each (wave w's quarter): a wave's scattered loads touch 64 cache lines per
  // instruction, and four such waves on one CU queue behind its single texture addresser
  if (threadIdx.x >= 64) return;
  const NodePackJob& J = batch.job[blockIdx.x >> 2];
  if (J.din == 20) node_pack_body<20>(J, blockIdx.x & 3, threadIdx.x);
  else node_pack_body<64>(J, blockIdx.x & 3, threadIdx.x);
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

// Two kinds of workgroups per layer (blockIdx.y = 2 * layer + kind), so that each kind is small enough for THREE
// workgroups per CU (45 KB LDS, <= 168 registers; one combined workgroup needed 73 KB / two per CU and ran the MFMA pipe
// at 39 %: measured, profiles/r02 SQ counters):
//   kind 0 (NODE):  dWn2 (64 x 64) and dWn1 (64 x 128) + dbn2 / dbn1     -- 12 MFMAs per 4-row step
//   kind 1 (PROJ):  dW1sd (128 x 64) + db0 / db1                          --  8 MFMAs per 4-row step
// Two sets of row tiles: chunk c+1 is staged into one set while the MFMAs of chunk c read the other -- ONE barrier per
// chunk.  Rows come in through bounded raw-buffer views (rows past N read as zero; no 64-bit address math).
struct WgradSmem {
  static constexpr int LDP = 132;
  union {
    struct { float Xs2[2][16 * LDP], Gs2[2][16 * LD], Zs2[2][16 * LD], As2[2][16 * LD]; } node;
    struct { float Ps2[2][16 * LDP], Hs2[2][16 * LD]; } proj;
  };
  float vec[4][2][64];
};

template <int DIN>
__device__ __forceinline__ void egnn_node_wgrad16_node(WgradSmem& sm, const float* __restrict__ dh, const float* __restrict__ zn1,
                                                       const float* __restrict__ dzn1, const float* __restrict__ h, int ld_h,
                                                       const float* __restrict__ h_neigh, int ld_hn, float* __restrict__ part,
                                                       int N, int rows_per_wg) {
  constexpr int LDP = WgradSmem::LDP;
  auto& Xs2 = sm.node.Xs2; auto& Gs2 = sm.node.Gs2; auto& Zs2 = sm.node.Zs2; auto& As2 = sm.node.As2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int r_begin = blockIdx.x * rows_per_wg, r_end = min(N, r_begin + rows_per_wg);
  const rsrc_t rs_g = make_rsrc_n(dh, N * H * 4), rs_z = make_rsrc_n(dzn1, N * H * 4), rs_zn = make_rsrc_n(zn1, N * H * 4);
  const rsrc_t rs_h = make_rsrc_n(h, N * ld_h * 4), rs_hn = make_rsrc_n(h_neigh, N * ld_hn * 4);
  f32x4 dW2[4], dWn[8];
  zero_acc4(dW2);
  zero_acc4(dWn);
  // staging role of a lane: row 4 wave + rr of the chunk, columns 4 c4 .. 4 c4 + 3 -- ONE 16-byte load per matrix and chunk
  // (a wave fetches its four 256-byte rows with one instruction), one 16-byte LDS store each
  const int rr = lane >> 4, c4 = lane & 15;
  const bool h4 = DIN == 64 && (ld_h & 3) == 0 && (reinterpret_cast<uintptr_t>(h) & 15) == 0;      // h rows 16-byte aligned
  f32x4 s_g = f32x4{0.f, 0.f, 0.f, 0.f}, s_z = s_g;      // partial column sums of columns 4 c4 .. + 3 over rows == rr (mod 4)
  // The rows of a chunk travel global -> registers -> LDS.  TWO register sets: the loads of chunk c + 2 are issued when chunk c
  // is computed and land during chunk c + 1 (83.0 -> 81.7 us; the ~2.9 us a 16-row chunk costs whatever the kind of workgroup
  // is NOT this latency -- HISTORY.md, round 3)
  struct Rows { f32x4 g, z, zn, xh, xn; };
  Rows R0, R1;
  auto fetch = [&](Rows& R, int c0) {
    const int row = c0 + wave * 4 + rr;
    R.g = buf_load4(rs_g, row * (H * 4) + c4 * 16, 0);
    R.z = buf_load4(rs_z, row * (H * 4) + c4 * 16, 0);
    R.zn = buf_load4(rs_zn, row * (H * 4) + c4 * 16, 0);
    R.xn = buf_load4(rs_hn, row * (ld_hn * 4) + c4 * 16, 0);
    if (h4) {
      R.xh = buf_load4(rs_h, row * (ld_h * 4) + c4 * 16, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) R.xh[j] = (4 * c4 + j < DIN) ? buf_load(rs_h, row * (ld_h * 4) + (4 * c4 + j) * 4, 0) : 0.0f;
    }
  };
  auto stage = [&](const Rows& R, int buf) {      // registers -> LDS set `buf`
    float *Xs = Xs2[buf], *Gs = Gs2[buf], *As = As2[buf], *Zs = Zs2[buf];
    const int lr = wave * 4 + rr;
    *reinterpret_cast<f32x4*>(Gs + lr * LD + 4 * c4) = R.g;
    *reinterpret_cast<f32x4*>(Zs + lr * LD + 4 * c4) = R.z;
    s_g += R.g; s_z += R.z;
    f32x4 a;
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = silu_f(R.zn[j]);      // rows past N: SiLU(0) = 0
    *reinterpret_cast<f32x4*>(As + lr * LD + 4 * c4) = a;
    *reinterpret_cast<f32x4*>(Xs + lr * LDP + 4 * c4) = R.xh;
    *reinterpret_cast<f32x4*>(Xs + lr * LDP + 64 + 4 * c4) = R.xn;
  };
  auto products = [&](int buf) {
    const float *Xs = Xs2[buf], *Gs = Gs2[buf], *As = As2[buf], *Zs = Zs2[buf];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int e = 4 * q + s;
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
  };
  if (r_begin < r_end) {
    fetch(R0, r_begin);
    fetch(R1, r_begin + 16);      // (rows past the slice are fetched but never staged; rows past N read as zero)
    stage(R0, 0);
  }
  __syncthreads();
  // two chunks per trip, so that the register sets alternate without being copied (a copy waits for the loads it copies)
  int c0 = r_begin;
  for (; c0 + 16 < r_end; c0 += 32) {
    // LDS set 0 holds chunk c0, R1 holds chunk c0 + 16; R0 receives chunk c0 + 32
    fetch(R0, c0 + 32);
    products(0);
    stage(R1, 1);
    __syncthreads();      // the other set is complete, and every wave is done reading this one
    fetch(R1, c0 + 48);
    products(1);
    if (c0 + 32 < r_end) stage(R0, 0);
    __syncthreads();
  }
  if (c0 < r_end) products(0);      // an odd last chunk: staged in set 0 by the trip before (or by the prologue)
  float* pn = part + WG_PROJ;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int lr = tile16_row(t, q);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) pn[64 * 128 + (wave * 16 + lr) * H + nt * 16 + r] = dW2[nt][t];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) pn[(wave * 16 + lr) * 128 + nt * 16 + r] = dWn[nt][t];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {      // rows (mod 4) in a fixed order: (0 + 1) + (2 + 3)
    s_z[j] += __shfl_xor(s_z[j], 16, 64); s_z[j] += __shfl_xor(s_z[j], 32, 64);
    s_g[j] += __shfl_xor(s_g[j], 16, 64); s_g[j] += __shfl_xor(s_g[j], 32, 64);
  }
  if (rr == 0) {
    *reinterpret_cast<f32x4*>(&sm.vec[wave][0][4 * c4]) = s_z;
    *reinterpret_cast<f32x4*>(&sm.vec[wave][1][4 * c4]) = s_g;
  }
  __syncthreads();
  if (tid < 128) {
    const int which = tid >> 6;   // 0: dbn1, 1: dbn2
    pn[64 * 128 + 64 * 64 + which * 64 + lane] =
        ((sm.vec[0][which][lane] + sm.vec[1][which][lane]) + sm.vec[2][which][lane]) + sm.vec[3][which][lane];
  }
}

__device__ __forceinline__ void egnn_node_wgrad16_proj(WgradSmem& sm, const float* __restrict__ g_psd, const float* __restrict__ h_out,
                                                       int ld_ho, int dho, float* __restrict__ part, int N, int rows_per_wg) {
  constexpr int LDP = WgradSmem::LDP;
  auto& Ps2 = sm.proj.Ps2; auto& Hs2 = sm.proj.Hs2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int r_begin = blockIdx.x * rows_per_wg, r_end = min(N, r_begin + rows_per_wg);
  const rsrc_t rs_p = make_rsrc_n(g_psd, N * 128 * 4), rs_ho = make_rsrc_n(h_out, N * ld_ho * 4);
  f32x4 dW1[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a) zero_acc4(dW1[a]);
  const int rr = lane >> 4, c4 = lane & 15;      // staging role: row 4 wave + rr, columns 4 c4 .. + 3 (see the node kind)
  const bool h4 = dho == 64 && (ld_ho & 3) == 0 && (reinterpret_cast<uintptr_t>(h_out) & 15) == 0;
  f32x4 s_p0 = f32x4{0.f, 0.f, 0.f, 0.f}, s_p1 = s_p0;
  struct Rows { f32x4 p0, p1, ho; };      // two register sets, prefetch distance 2 (see the node kind)
  Rows R0, R1;
  auto fetch = [&](Rows& R, int c0) {
    const int row = c0 + wave * 4 + rr;
    R.p0 = buf_load4(rs_p, row * 512 + c4 * 16, 0);
    R.p1 = buf_load4(rs_p, row * 512 + 256 + c4 * 16, 0);
    if (h4) {
      R.ho = buf_load4(rs_ho, row * (ld_ho * 4) + c4 * 16, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) R.ho[j] = (4 * c4 + j < dho) ? buf_load(rs_ho, row * (ld_ho * 4) + (4 * c4 + j) * 4, 0) : 0.0f;
    }
  };
  auto stage = [&](const Rows& R, int buf) {
    float *Ps = Ps2[buf], *Hs = Hs2[buf];
    const int lr = wave * 4 + rr;
    *reinterpret_cast<f32x4*>(Ps + lr * LDP + 4 * c4) = R.p0;
    *reinterpret_cast<f32x4*>(Ps + lr * LDP + 64 + 4 * c4) = R.p1;
    s_p0 += R.p0; s_p1 += R.p1;
    *reinterpret_cast<f32x4*>(Hs + lr * LD + 4 * c4) = R.ho;
  };
  auto products = [&](int buf) {
    const float *Ps = Ps2[buf], *Hs = Hs2[buf];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int e = 4 * q + s;
      const float ap0 = Ps[e * LDP + (2 * wave) * 16 + r], ap1 = Ps[e * LDP + (2 * wave + 1) * 16 + r];
      float bh[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) bh[nt] = Hs[e * LD + nt * 16 + r];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        dW1[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap0, bh[nt], dW1[0][nt], 0, 0, 0);
        dW1[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap1, bh[nt], dW1[1][nt], 0, 0, 0);
      }
    }
  };
  if (r_begin < r_end) {
    fetch(R0, r_begin);
    fetch(R1, r_begin + 16);
    stage(R0, 0);
  }
  __syncthreads();
  int c0 = r_begin;
  for (; c0 + 16 < r_end; c0 += 32) {
    fetch(R0, c0 + 32);
    products(0);
    stage(R1, 1);
    __syncthreads();
    fetch(R1, c0 + 48);
    products(1);
    if (c0 + 32 < r_end) stage(R0, 0);
    __syncthreads();
  }
  if (c0 < r_end) products(0);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int lr = tile16_row(t, q);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      part[((2 * wave) * 16 + lr) * H + nt * 16 + r] = dW1[0][nt][t];
      part[((2 * wave + 1) * 16 + lr) * H + nt * 16 + r] = dW1[1][nt][t];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s_p0[j] += __shfl_xor(s_p0[j], 16, 64); s_p0[j] += __shfl_xor(s_p0[j], 32, 64);
    s_p1[j] += __shfl_xor(s_p1[j], 16, 64); s_p1[j] += __shfl_xor(s_p1[j], 32, 64);
  }
  if (rr == 0) {
    *reinterpret_cast<f32x4*>(&sm.vec[wave][0][4 * c4]) = s_p1;
    *reinterpret_cast<f32x4*>(&sm.vec[wave][1][4 * c4]) = s_p0;
  }
  __syncthreads();
  if (tid < 128) {
    const int which = tid >> 6;   // 0: db1, 1: db0
    part[128 * 64 + which * 64 + lane] =
        ((sm.vec[0][which][lane] + sm.vec[1][which][lane]) + sm.vec[2][which][lane]) + sm.vec[3][which][lane];
  }
}

// All layers of a stack in ONE launch (blockIdx.y = 2 * layer + kind): the co-resident workgroups of different layers and
// kinds hide each other's latency.
struct WgradLayer {
  const float *g_psd, *h_out, *dh, *zn1, *dzn1, *h, *h_neigh;
  float* partials;
  int ld_h, din, ld_hn, ld_ho, dho, pad;     // h_out: row stride and number of valid columns (64 / 64 for EGNN layers)
};
constexpr int WGRAD_MAX_LAYERS = 8;
struct WgradBatch { WgradLayer layer[WGRAD_MAX_LAYERS]; };

__global__ __launch_bounds__(256, 3) void egnn_node_wgrad16_batched_kernel(WgradBatch batch, int N, int grid_node, int rows_node,
                                                                           int grid_proj, int rows_proj) {
  __shared__ WgradSmem sm;
  const WgradLayer& L = batch.layer[blockIdx.y >> 1];
  float* part = L.partials + (size_t)blockIdx.x * WG_STRIDE;
  if ((blockIdx.y & 1) == 0) {
    if (L.dzn1 == nullptr || (int)blockIdx.x >= grid_node) return;      // a projection-only job (layer-0 pre-projection): NODE part left untouched
    if (L.din == 20) egnn_node_wgrad16_node<20>(sm, L.dh, L.zn1, L.dzn1, L.h, L.ld_h, L.h_neigh, L.ld_hn, part, N, rows_node);
    else egnn_node_wgrad16_node<64>(sm, L.dh, L.zn1, L.dzn1, L.h, L.ld_h, L.h_neigh, L.ld_hn, part, N, rows_node);
  } else {
    if (L.g_psd == nullptr || (int)blockIdx.x >= grid_proj) return;     // no next pre-projection: PROJ part left untouched
    egnn_node_wgrad16_proj(sm, L.g_psd, L.h_out, L.ld_ho, L.dho, part, N, rows_proj);
  }
}

}  // namespace is

#ifdef IS_STAGE_STAMPS
extern "C" int is_debug_stamps_node(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(is::g_stamps_node), sizeof(long long) * 16) == hipSuccess ? 0 : is::fail(__func__, -5);
}
#endif

extern "C" int is_node_pack_floats(void) { return is::NODE_PACK_FLOATS; }

// is_node_proj_fwd (layer-0 pre-projection: h [N, ld_h] (din = 20 | 64 columns), W1 [64, ldw], b0 (may be NULL), b1 -> psd
// [N, 128]) and is_node_pack_weights (jobs as there) as ONE launch; x_dst != NULL: also the dense [N,3] copy of the
// coordinates x_src [N, ld_x] (the reference keeps them as the last three columns of ndata['x']).
extern "C" int is_stack_prologue(const void* jobs, int njobs, const float* h, int ld_h, int din, const float* W1, int ldw,
                                 const float* b0, const float* b1, float* psd, const float* x_src, int ld_x, float* x_dst,
                                 int N, void* stream) {
  if (njobs <= 0 || njobs > is::NODE_PACK_MAX || N <= 0 || (din != 20 && din != 64)) return is::fail(__func__, -22);
  if (x_dst != nullptr && (x_src == nullptr || ld_x < 3)) return is::fail(__func__, -22);
  is::NodePackBatch batch;
  const is::NodePackJob* src = static_cast<const is::NodePackJob*>(jobs);
  for (int i = 0; i < njobs; ++i) {
    batch.job[i] = src[i];
    if ((src[i].din != 20 && src[i].din != 64) || ((src[i].W1n == nullptr) != (src[i].W1nb == nullptr))) return is::fail(__func__, -22);
  }
  const int proj_blocks = std::min((N + 23) / 24, 1024);
  if (din == 20)
    hipLaunchKernelGGL(is::stack_prologue_kernel<20>, dim3(proj_blocks + 4 * njobs), dim3(256), 0, static_cast<hipStream_t>(stream), batch,
                       njobs, proj_blocks, h, ld_h, W1, ldw, b0, b1, psd, x_src, ld_x, x_dst, N);
  else
    hipLaunchKernelGGL(is::stack_prologue_kernel<64>, dim3(proj_blocks + 4 * njobs), dim3(256), 0, static_cast<hipStream_t>(stream), batch,
                       njobs, proj_blocks, h, ld_h, W1, ldw, b0, b1, psd, x_src, ld_x, x_dst, N);
  return is::launch_status(__func__);
}

extern "C" int is_egnn_node_wgrad_stride(void) { return is::WG_STRIDE; }
extern "C" int is_egnn_node_wgrad_proj_floats(void) { return is::WG_PROJ; }

// layers: host array of `nlayers` (<= 8) records {g_psd, h_out, dh, zn1, dzn1, h, h_neigh, partials, ld_h, din, ld_hn, ld_ho,
// dho, pad} (pointers first, then six ints); the NODE part of every layer is produced by `grid_node` workgroups (records 0 ..
// grid_node - 1 of its partials), the PROJ part by `grid_proj` (records 0 .. grid_proj - 1), record layout and stride as
// is_egnn_node_wgrad_stride / _proj_floats describe.  (The two kinds cost 12 and 8 MFMAs per 4-row step, but equal work per
// workgroup -- grids 3 : 2 -- measured slower than equal grids: HISTORY.md, round 3.)
// A record with dzn1 == NULL is a projection-only job (PROJ part from g_psd and the first dho columns of h_out, row stride
// ld_ho); a record with g_psd == NULL leaves the PROJ part untouched.
extern "C" int is_egnn_node_wgrad_batched(const void* layers, int nlayers, int grid_node, int grid_proj, int N, void* stream) {
  if (N <= 0 || grid_node <= 0 || grid_proj <= 0 || nlayers <= 0 || nlayers > is::WGRAD_MAX_LAYERS) return is::fail(__func__, -22);
  if ((long long)N * 128 * 4 >= 0x7ffff000LL) return is::fail(__func__, -22);      // 32-bit byte offsets of the raw-buffer row loads (4.1 M nodes)
  is::WgradBatch batch;
  const is::WgradLayer* src = static_cast<const is::WgradLayer*>(layers);
  for (int i = 0; i < nlayers; ++i) {
    batch.layer[i] = src[i];
    if ((src[i].din != 20 && src[i].din != 64) || src[i].dho < 0 || src[i].dho > 64) return is::fail(__func__, -22);
    if (src[i].g_psd == nullptr && src[i].dzn1 == nullptr) return is::fail(__func__, -22);
  }
  auto rows_of = [N](int grid) { return (((N + grid - 1) / grid) + 15) / 16 * 16; };
  hipLaunchKernelGGL(is::egnn_node_wgrad16_batched_kernel, dim3(std::max(grid_node, grid_proj), 2 * nlayers), dim3(256), 0,
                     static_cast<hipStream_t>(stream), batch, N, grid_node, rows_of(grid_node), grid_proj, rows_of(grid_proj));
  return is::launch_status(__func__);
}
