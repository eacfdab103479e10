// Fused EGNN edge pass, forward  (replaces, for one EGNNConv layer, the DGL
// kernels the reference reaches through dgl.nn.EGNNConv.forward:
// SDDMM u_sub_v, the two index_select gathers + cat, the edge/coord MLP GEMMs
// and SpMM copy_e/sum + copy_e/mean -- SURVEY.md section 2, rows K1-K5;
// reference call sites models/hybrid_models.py:323-324).
//
// Work decomposition
//   One workgroup (4 waves) owns NV = 32 consecutive destination nodes and walks
//   their in-edges (CSR by destination) in windows of 4 x 32 edges, one 32-edge
//   tile per wave.  Per tile:
//     S0  lane = edge   : src id, dst-local id, x_diff, radial  -> LDS scalars
//     SA  lane = channel: z1 = Ps[src] + Pd[dst] + radial*w_r + a.W_a ; m1 = SiLU(z1)
//                         (Ps/Pd are the node-level pre-projections of edge_mlp.0,
//                          so the gather is ONE coalesced 256-byte row per edge)
//     MM1 MFMA          : z2 = m1 W2^T + b2      (saved) ; mh = SiLU(z2) -> LDS
//     MM2 MFMA          : z3 = mh Wc1^T + bc1    (saved) ; s = SiLU(z3).wc2
//     SEG lane = channel: h_neigh[v] += mh rows of v's edges (CSR order, fixed
//                         summation order => deterministic, no float atomics),
//                         x_neigh[v] += s * x_diff
//   Intermediate E x 64 messages never reach HBM; only z2/z3 are streamed out
//   (optional) for the backward pass.
#include "common.h"

namespace is {

template <int FE_MAX>
struct FwdSmem {
  float w2[H * LD];
  float wc1[H * LD];
  float act[WAVES][TE * LD];
  int rp[NV + 1];
  int e_src[WAVES][TE];
  int e_dl[WAVES][TE];
  float e_rad[WAVES][TE];
  float e_xd[WAVES][3][TE];
  float e_s[WAVES][TE];
  float e_a[WAVES][FE_MAX][TE];
};

template <int FE_MAX>
__global__ __launch_bounds__(256) void egnn_edge_fwd_kernel(
    const float* __restrict__ ps, const float* __restrict__ pd, int ld_p,
    const float* __restrict__ x, const float* __restrict__ ea,
    const int* __restrict__ rowptr, const int* __restrict__ srcs,
    const float* __restrict__ W1, int ldw, int din,
    const float* __restrict__ W2, const float* __restrict__ b2,
    const float* __restrict__ Wc1, const float* __restrict__ bc1, const float* __restrict__ wc2,
    float* __restrict__ h_neigh, int ld_hn, float* __restrict__ x_out,
    float* __restrict__ z2s, float* __restrict__ z3s, int N, int Fe) {
  __shared__ FwdSmem<FE_MAX> sm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hf = lane >> 5;
  const int v0 = blockIdx.x * NV;
  const int nv = min(NV, N - v0);

  load_matrix_lds(sm.w2, W2, H, tid, 256);
  load_matrix_lds(sm.wc1, Wc1, H, tid, 256);
  if (tid <= NV) sm.rp[tid] = rowptr[v0 + min(tid, nv)];

  // lane = channel constants
  // radial / edge-feature columns of the native edge_mlp.0.weight [64, 2*din + 1 + Fe]
  const float wr_c = W1[lane * ldw + 2 * din];
  float wa_c[FE_MAX];
#pragma unroll
  for (int f = 0; f < FE_MAX; ++f) wa_c[f] = (f < Fe) ? W1[lane * ldw + 2 * din + 1 + f] : 0.0f;
  // lane = tile-column constants (col = nt*32 + r)
  float b2_c[2], bc1_c[2], wc2_c[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    b2_c[nt] = b2[nt * 32 + r];
    bc1_c[nt] = bc1[nt * 32 + r];
    wc2_c[nt] = wc2[nt * 32 + r];
  }
  __syncthreads();

  const int e_begin = sm.rp[0], e_end = sm.rp[nv];
  float acc_h[NV / WAVES];
  float acc_x[NV / WAVES];
#pragma unroll
  for (int i = 0; i < NV / WAVES; ++i) { acc_h[i] = 0.0f; acc_x[i] = 0.0f; }

  float* act = sm.act[wave];

  for (int win = e_begin; win < e_end; win += WAVES * TE) {
    const int cb = win + wave * TE;
    const int nvalid = max(0, min(TE, e_end - cb));

    // ---- S0: per-edge scalars (lanes 0..31 = edges of this wave's tile) ----
    if (lane < TE) {
      const bool valid = lane < nvalid;
      const int e = cb + lane;
      int s = v0, dl = 0;
      float d0 = 0.f, d1 = 0.f, d2 = 0.f, rad = 0.f;
      if (valid) {
        s = srcs[e];
        int lo = 0, hi = nv;  // largest lo with rp[lo] <= e
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (sm.rp[mid] <= e) lo = mid; else hi = mid;
        }
        dl = lo;
        const int v = v0 + dl;
        d0 = x[s * 3 + 0] - x[v * 3 + 0];
        d1 = x[s * 3 + 1] - x[v * 3 + 1];
        d2 = x[s * 3 + 2] - x[v * 3 + 2];
        rad = radial3(d0, d1, d2);
        const float inv = 1.0f / (sqrtf(rad) + 1e-30f);
        d0 *= inv; d1 *= inv; d2 *= inv;
      }
      sm.e_src[wave][lane] = s;
      sm.e_dl[wave][lane] = dl;
      sm.e_rad[wave][lane] = rad;
      sm.e_xd[wave][0][lane] = d0;
      sm.e_xd[wave][1][lane] = d1;
      sm.e_xd[wave][2][lane] = d2;
#pragma unroll
      for (int f = 0; f < FE_MAX; ++f)
        sm.e_a[wave][f][lane] = (valid && f < Fe) ? ea[(size_t)e * Fe + f] : 0.0f;
    }
    __syncthreads();

    // ---- SA: gather + first edge-MLP layer, lane = channel ----
    {
      float g[TE];
#pragma unroll
      for (int i = 0; i < TE; ++i) {
        const int s = sm.e_src[wave][i];
        const int v = v0 + sm.e_dl[wave][i];
        g[i] = ps[(size_t)s * ld_p + lane] + pd[(size_t)v * ld_p + lane];
      }
#pragma unroll
      for (int i = 0; i < TE; ++i) {
        float z1 = g[i] + sm.e_rad[wave][i] * wr_c;
#pragma unroll
        for (int f = 0; f < FE_MAX; ++f) z1 += sm.e_a[wave][f][i] * wa_c[f];
        act[i * LD + lane] = (i < nvalid) ? silu_f(z1) : 0.0f;
      }
    }
    __syncthreads();

    // ---- MM1: z2 = m1 W2^T + b2 ; mh = SiLU(z2) ----
    {
      f32x16 acc[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[nt][t] = 0.0f;
      mm_rows<2, H>(acc, act, sm.w2, lane);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int row = tile_row(t, hf);
          const float z2 = acc[nt][t] + b2_c[nt];
          if (z2s != nullptr && row < nvalid) z2s[(size_t)(cb + row) * H + nt * 32 + r] = z2;
          act[row * LD + nt * 32 + r] = (row < nvalid) ? silu_f(z2) : 0.0f;
        }
      }
    }
    __syncthreads();

    // ---- MM2: z3 = mh Wc1^T + bc1 ; s = SiLU(z3) . wc2 ----
    {
      f32x16 acc[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[nt][t] = 0.0f;
      mm_rows<2, H>(acc, act, sm.wc1, lane);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int row = tile_row(t, hf);
        float part = 0.0f;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const float z3 = acc[nt][t] + bc1_c[nt];
          if (z3s != nullptr && row < nvalid) z3s[(size_t)(cb + row) * H + nt * 32 + r] = z3;
          part += silu_f(z3) * wc2_c[nt];
        }
        part = sum_over_r(part);
        if (r == 0) sm.e_s[wave][row] = part;
      }
    }
    __syncthreads();

    // ---- SEG: deterministic segment reduction over this window ----
    {
      const int win_hi = min(win + WAVES * TE, e_end);
#pragma unroll
      for (int i = 0; i < NV / WAVES; ++i) {
        const int nl = wave + WAVES * i;
        if (nl < nv) {
          const int lo = max(sm.rp[nl], win), hi = min(sm.rp[nl + 1], win_hi);
          float ah = acc_h[i], ax = acc_x[i];
          for (int e = lo; e < hi; ++e) {
            const int rel = e - win;
            const int w = rel >> 5, row = rel & 31;
            ah += sm.act[w][row * LD + lane];
            if (lane < 3) ax += sm.e_s[w][row] * sm.e_xd[w][lane][row];
          }
          acc_h[i] = ah; acc_x[i] = ax;
        }
      }
    }
    __syncthreads();
  }

  // ---- write h_neigh (sum) and x' = x + mean(msg_x) ----
#pragma unroll
  for (int i = 0; i < NV / WAVES; ++i) {
    const int nl = wave + WAVES * i;
    if (nl < nv) {
      const int v = v0 + nl;
      h_neigh[(size_t)v * ld_hn + lane] = acc_h[i];
      if (lane < 3) {
        const int deg = sm.rp[nl + 1] - sm.rp[nl];
        const float xn = deg > 0 ? acc_x[i] / (float)deg : 0.0f;
        x_out[v * 3 + lane] = x[v * 3 + lane] + xn;
      }
    }
  }
}

}  // namespace is

extern "C" int is_egnn_edge_fwd(const float* ps, const float* pd, int ld_p, const float* x, const float* ea,
                                const int32_t* rowptr, const int32_t* srcs, const float* W1, int ldw, int din,
                                const float* W2, const float* b2, const float* Wc1, const float* bc1,
                                const float* wc2, float* h_neigh, int ld_hn, float* x_out, float* z2s,
                                float* z3s, int N, int Fe, void* stream) {
  if (N <= 0) return 0;
  if (Fe < 0 || Fe > 8) return -22;  // EINVAL: edge_feat_size in [0, 8]
  const dim3 grid((N + is::NV - 1) / is::NV), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (Fe <= 1) {
    hipLaunchKernelGGL(is::egnn_edge_fwd_kernel<1>, grid, block, 0, st, ps, pd, ld_p, x, ea, rowptr, srcs, W1, ldw, din,
                       W2, b2, Wc1, bc1, wc2, h_neigh, ld_hn, x_out, z2s, z3s, N, Fe);
  } else {
    hipLaunchKernelGGL(is::egnn_edge_fwd_kernel<8>, grid, block, 0, st, ps, pd, ld_p, x, ea, rowptr, srcs, W1, ldw, din,
                       W2, b2, Wc1, bc1, wc2, h_neigh, ld_hn, x_out, z2s, z3s, N, Fe);
  }
  return hipGetLastError() == hipSuccess ? 0 : -5;
}
