// Fused EGNN edge pass, backward (autograd of egnn_edge_fwd.hip; replaces the
// backward of DGL's SDDMM / gather / SpMM kernels plus the edge- and coord-MLP
// GEMM backward -- SURVEY.md section 2 row K7).
//
// Same decomposition as the forward: a workgroup owns NV = 32 destination
// nodes per tile and walks their in-edges in windows of 4 x 32 edges; the grid
// is persistent (workgroup w handles node tiles w, w + grid, ...) so that the
// weight-gradient accumulators stay in registers for the whole launch and only
// ONE partial per workgroup is written (deterministic two-stage reduction, no
// float atomics -- the reference runs under torch.use_deterministic_algorithms,
// utils/seed.py:18).
//
// Per 32-edge tile (E-rows MFMA layout = rows are edges, columns channels):
//   S0   geometry: d, radial, r, x_diff, gx = dL/dx_neigh[dst] / deg, gx.x_diff
//   E3   t = SiLU(z3), s = t.wc2 ; dz3 = (gx.x_diff) * wc2 * SiLU'(z3)
//   WG1  dWc1 += dz3^T mh                         (mh = SiLU(z2))
//   MM3  dmh = dz3 Wc1 + dL/dh_neigh[dst] ; dz2 = dmh * SiLU'(z2)
//   SA   z1 recomputed from the gathers (lane = channel)
//   WG2  dW2 += dz2^T m1
//   MM4  dm1 = dz2 W2 ; dz1 = dm1 * SiLU'(z1)  -> streamed to dZ1[e] (CSR order)
//   GEO  d(x_src - x_dst) from d x_diff and d radial -> streamed to dD[e]
//   SEG  dPd[v] = sum dz1 , dx[v] = g_xout[v] - sum dD   over v's in-edges
// The source-side scatter (dPs[src] += dz1, dx[src] += dD) is done afterwards by
// is_gather_segment_sum over the CSR-by-source index (segment_ops.hip).
#include "common.h"

namespace is {

template <int FE_MAX>
struct BwdSmem {
  float w2t[H * LD];   // w2t[i][o]  = W2[o][i]
  float wc1t[H * LD];  // wc1t[i][o] = Wc1[o][i]
  float actA[WAVES][TE * LD];
  float actB[WAVES][TE * LD];
  float actC[WAVES][TE * LD];  // SiLU'(z2), then SiLU'(z1)
  int rp[NV + 1];
  int e_src[WAVES][TE];
  int e_dl[WAVES][TE];
  float e_rad[WAVES][TE];
  float e_r[WAVES][TE];
  float e_inv[WAVES][TE];
  float e_d[WAVES][3][TE];
  float e_xd[WAVES][3][TE];
  float e_gx[WAVES][3][TE];
  float e_gxd[WAVES][TE];
  float e_s[WAVES][TE];
  float e_drad[WAVES][TE];
  float e_dd[WAVES][3][TE];
  float e_a[WAVES][FE_MAX][TE];
};

// layout of the per-workgroup partial record (floats)
//   [0, 4096)      dW2   [o][i]
//   [4096, 8192)   dWc1  [o][i]
//   [8192, 8256)   db2      [8256, 8320) dbc1     [8320, 8384) dwc2
//   [8384, 8448)   dw_r     [8448, 8448 + 64*8)  dW_a [c][f] (stride 8)
constexpr int PART_STRIDE = 8448 + 64 * 8;

template <int FE_MAX>
__global__ __launch_bounds__(256, 1) void egnn_edge_bwd_kernel(
    const float* __restrict__ ps, const float* __restrict__ pd, int ld_p,
    const float* __restrict__ x, const float* __restrict__ ea,
    const int* __restrict__ rowptr, const int* __restrict__ srcs,
    const float* __restrict__ W1, int ldw, int din,
    const float* __restrict__ W2, const float* __restrict__ Wc1, const float* __restrict__ wc2,
    const float* __restrict__ z2s, const float* __restrict__ z3s,
    const float* __restrict__ g_hn, int ld_ghn, const float* __restrict__ g_xout,
    float* __restrict__ dZ1, float* __restrict__ dD,
    float* __restrict__ dPd, int ld_dpd, float* __restrict__ dx,
    float* __restrict__ partials, int N, int Fe) {
  __shared__ BwdSmem<FE_MAX> sm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hf = lane >> 5;

  load_matrix_lds_t(sm.w2t, W2, tid, 256);
  load_matrix_lds_t(sm.wc1t, Wc1, tid, 256);

  const float wr_c = W1[lane * ldw + 2 * din];
  float wa_c[FE_MAX];
#pragma unroll
  for (int f = 0; f < FE_MAX; ++f) wa_c[f] = (f < Fe) ? W1[lane * ldw + 2 * din + 1 + f] : 0.0f;
  float wc2_c[2], wr_t[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    wc2_c[nt] = wc2[nt * 32 + r];
    wr_t[nt] = W1[(nt * 32 + r) * ldw + 2 * din];
  }

  // launch-persistent weight-gradient accumulators
  f32x16 dW2[2][2], dWc1[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int t = 0; t < 16; ++t) { dW2[a][b][t] = 0.0f; dWc1[a][b][t] = 0.0f; }
  float db2_a[2] = {0.f, 0.f}, dbc1_a[2] = {0.f, 0.f}, dwc2_a[2] = {0.f, 0.f};
  // lane = channel accumulators
  float dwr_c = 0.0f;
  float dwa_c[FE_MAX];
#pragma unroll
  for (int f = 0; f < FE_MAX; ++f) dwa_c[f] = 0.0f;

  float* actA = sm.actA[wave];
  float* actB = sm.actB[wave];
  float* actC = sm.actC[wave];
  const int num_tiles = (N + NV - 1) / NV;

  for (int tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    const int v0 = tile * NV;
    const int nv = min(NV, N - v0);
    __syncthreads();  // previous tile's readers of sm.rp are done
    if (tid <= NV) sm.rp[tid] = rowptr[v0 + min(tid, nv)];
    __syncthreads();
    const int e_begin = sm.rp[0], e_end = sm.rp[nv];

    float acc_h[NV / WAVES], acc_x[NV / WAVES];
#pragma unroll
    for (int i = 0; i < NV / WAVES; ++i) { acc_h[i] = 0.0f; acc_x[i] = 0.0f; }

    for (int win = e_begin; win < e_end; win += WAVES * TE) {
      const int cb = win + wave * TE;
      const int nvalid = max(0, min(TE, e_end - cb));

      // ---- S0: geometry + upstream coordinate gradient, lane = edge ----
      if (lane < TE) {
        const bool valid = lane < nvalid;
        const int e = cb + lane;
        int s = v0, dl = 0;
        float d0 = 0.f, d1 = 0.f, d2 = 0.f, rad = 0.f, rr = 0.f, inv = 0.f;
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        if (valid) {
          s = srcs[e];
          int lo = 0, hi = nv;
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
          rr = sqrtf(rad);
          inv = 1.0f / (rr + 1e-30f);
          const float invdeg = 1.0f / (float)(sm.rp[dl + 1] - sm.rp[dl]);
          g0 = g_xout[v * 3 + 0] * invdeg;
          g1 = g_xout[v * 3 + 1] * invdeg;
          g2 = g_xout[v * 3 + 2] * invdeg;
        }
        sm.e_src[wave][lane] = s;
        sm.e_dl[wave][lane] = dl;
        sm.e_rad[wave][lane] = rad;
        sm.e_r[wave][lane] = rr;
        sm.e_inv[wave][lane] = inv;
        sm.e_d[wave][0][lane] = d0; sm.e_d[wave][1][lane] = d1; sm.e_d[wave][2][lane] = d2;
        const float x0 = d0 * inv, x1 = d1 * inv, x2 = d2 * inv;
        sm.e_xd[wave][0][lane] = x0; sm.e_xd[wave][1][lane] = x1; sm.e_xd[wave][2][lane] = x2;
        sm.e_gx[wave][0][lane] = g0; sm.e_gx[wave][1][lane] = g1; sm.e_gx[wave][2][lane] = g2;
        sm.e_gxd[wave][lane] = g0 * x0 + g1 * x1 + g2 * x2;
#pragma unroll
        for (int f = 0; f < FE_MAX; ++f)
          sm.e_a[wave][f][lane] = (valid && f < Fe) ? ea[(size_t)e * Fe + f] : 0.0f;
      }
      __syncthreads();

      // ---- E3: coord-MLP tail backward; stage dz3 (actA), mh (actB), SiLU'(z2) (actC) ----
#pragma unroll 2
      for (int t = 0; t < 16; ++t) {
        const int row = tile_row(t, hf);
        const bool rv = row < nvalid;
        float tt[2], sp[2];
        float part = 0.0f;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const size_t off = (size_t)(cb + row) * H + nt * 32 + r;
          const float z3 = rv ? z3s[off] : 0.0f;
          const float z2 = rv ? z2s[off] : 0.0f;
          silu_fg(z3, tt[nt], sp[nt]);
          part += tt[nt] * wc2_c[nt];
          float mh, dy2;
          silu_fg(z2, mh, dy2);
          actB[row * LD + nt * 32 + r] = rv ? mh : 0.0f;
          actC[row * LD + nt * 32 + r] = dy2;
        }
        part = sum_over_r(part);                 // s_e = SiLU(z3) . wc2
        if (r == 0) sm.e_s[wave][row] = part;
        const float ds = sm.e_gxd[wave][row];    // dL/ds_e (0 for invalid rows)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const float dz3 = ds * wc2_c[nt] * sp[nt];
          dwc2_a[nt] += ds * tt[nt];
          dbc1_a[nt] += dz3;
          actA[row * LD + nt * 32 + r] = dz3;
        }
      }
      __syncthreads();

      // ---- WG1: dWc1 += dz3^T mh ----
      mm_outer<2, 2>(dWc1, actA, actB, lane);

      // ---- MM3: dmh = dz3 Wc1 + g_hn[dst] ; dz2 = dmh * SiLU'(z2) ----
      {
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int t = 0; t < 16; ++t) acc[nt][t] = 0.0f;
        mm_rows<2, H>(acc, actA, sm.wc1t, lane);
        __syncthreads();  // every wave finished reading actA/actB of this stage
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int row = tile_row(t, hf);
          const bool rv = row < nvalid;
          const int v = v0 + sm.e_dl[wave][row];
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const float dy = actC[row * LD + nt * 32 + r];
            const float up = rv ? g_hn[(size_t)v * ld_ghn + nt * 32 + r] : 0.0f;
            const float dz2 = rv ? (acc[nt][t] + up) * dy : 0.0f;
            db2_a[nt] += dz2;
            actA[row * LD + nt * 32 + r] = dz2;
          }
        }
      }

      // ---- SA: recompute z1 (lane = channel) into actB ----
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
          actB[i * LD + lane] = (i < nvalid) ? z1 : 0.0f;
        }
      }
      __syncthreads();

      // ---- E1: m1 = SiLU(z1) in place (actB), SiLU'(z1) -> actC ----
#pragma unroll 4
      for (int t = 0; t < 16; ++t) {
        const int row = tile_row(t, hf);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const float z1 = actB[row * LD + nt * 32 + r];
          float y, dy;
          silu_fg(z1, y, dy);
          actB[row * LD + nt * 32 + r] = (row < nvalid) ? y : 0.0f;
          actC[row * LD + nt * 32 + r] = dy;
        }
      }
      __syncthreads();

      // ---- WG2: dW2 += dz2^T m1 ----
      mm_outer<2, 2>(dW2, actA, actB, lane);

      // ---- MM4: dm1 = dz2 W2 ; dz1 = dm1 * SiLU'(z1) ----
      {
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int t = 0; t < 16; ++t) acc[nt][t] = 0.0f;
        mm_rows<2, H>(acc, actA, sm.w2t, lane);
        __syncthreads();  // all reads of actA (dz2) done before it is overwritten with dz1
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int row = tile_row(t, hf);
          const bool rv = row < nvalid;
          float part = 0.0f;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const float dz1 = rv ? acc[nt][t] * actC[row * LD + nt * 32 + r] : 0.0f;
            if (rv) dZ1[(size_t)(cb + row) * H + nt * 32 + r] = dz1;
            actA[row * LD + nt * 32 + r] = dz1;
            part += dz1 * wr_t[nt];
          }
          part = sum_over_r(part);                 // dL/d radial_e
          if (r == 0) sm.e_drad[wave][row] = part;
        }
      }
      __syncthreads();

      // ---- SB: dw_r / dW_a partial sums over this wave's tile, lane = channel ----
#pragma unroll 8
      for (int i = 0; i < TE; ++i) {
        const float v = actA[i * LD + lane];  // dz1 (0 for invalid rows)
        dwr_c += v * sm.e_rad[wave][i];
#pragma unroll
        for (int f = 0; f < FE_MAX; ++f) dwa_c[f] += v * sm.e_a[wave][f][i];
      }

      // ---- GEO: gradient wrt d = x_src - x_dst, lane = edge ----
      if (lane < TE) {
        const bool valid = lane < nvalid;
        float q0 = 0.f, q1 = 0.f, q2 = 0.f;
        if (valid) {
          const float s = sm.e_s[wave][lane];
          const float inv = sm.e_inv[wave][lane], rr = sm.e_r[wave][lane];
          const float d0 = sm.e_d[wave][0][lane], d1 = sm.e_d[wave][1][lane], d2 = sm.e_d[wave][2][lane];
          // x_diff = d * inv(r), inv = 1/(r + 1e-30):  dd = dxd*inv - d * (d.dxd) * inv^2 / r
          const float u0 = s * sm.e_gx[wave][0][lane], u1 = s * sm.e_gx[wave][1][lane], u2 = s * sm.e_gx[wave][2][lane];
          const float ddot = d0 * u0 + d1 * u1 + d2 * u2;
          const float k = rr > 0.0f ? ddot * inv * inv / rr : 0.0f;
          const float dr2 = 2.0f * sm.e_drad[wave][lane];
          q0 = u0 * inv - d0 * k + d0 * dr2;
          q1 = u1 * inv - d1 * k + d1 * dr2;
          q2 = u2 * inv - d2 * k + d2 * dr2;
          const size_t e = (size_t)(cb + lane);
          dD[e * 3 + 0] = q0; dD[e * 3 + 1] = q1; dD[e * 3 + 2] = q2;
        }
        sm.e_dd[wave][0][lane] = q0; sm.e_dd[wave][1][lane] = q1; sm.e_dd[wave][2][lane] = q2;
      }
      __syncthreads();

      // ---- SEG: destination-side segment sums (deterministic, CSR order) ----
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
              ah += sm.actA[w][row * LD + lane];
              if (lane < 3) ax += sm.e_dd[w][lane][row];
            }
            acc_h[i] = ah; acc_x[i] = ax;
          }
        }
      }
      __syncthreads();
    }

    // ---- node-side outputs of this tile ----
#pragma unroll
    for (int i = 0; i < NV / WAVES; ++i) {
      const int nl = wave + WAVES * i;
      if (nl < nv) {
        const int v = v0 + nl;
        dPd[(size_t)v * ld_dpd + lane] = acc_h[i];
        if (lane < 3) dx[v * 3 + lane] = g_xout[v * 3 + lane] - acc_x[i];
      }
    }
  }

  // ---- reduce the weight-gradient accumulators over the 4 waves, write one partial ----
  __syncthreads();
  float* part = partials + (size_t)blockIdx.x * PART_STRIDE;
  float* scratch = &sm.actA[0][0];  // actA + actB are contiguous: 2 * 4 * 32 * 68 = 17408 floats >= 16384
  wg_sum_store_64x64(dW2, scratch, part, 64, tid, wave, lane);
  wg_sum_store_64x64(dWc1, scratch, part + H * H, 64, tid, wave, lane);
  // per-column vectors: tile-layout sums (db2, dbc1, dwc2) combine the two lane halves
  // (different rows), lane = channel sums (dw_r, dW_a) are already per channel; then 4 waves.
  {
    float* vec = &sm.actB[0][0];  // [wave][slot][64]
    constexpr int SLOTS = 4 + FE_MAX;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      float vals[3] = {db2_a[nt], dbc1_a[nt], dwc2_a[nt]};
#pragma unroll
      for (int sidx = 0; sidx < 3; ++sidx) {
        const float v = vals[sidx] + __shfl_xor(vals[sidx], 32, 64);
        if (hf == 0) vec[(wave * SLOTS + sidx) * H + nt * 32 + r] = v;
      }
    }
    vec[(wave * SLOTS + 3) * H + lane] = dwr_c;
#pragma unroll
    for (int f = 0; f < FE_MAX; ++f) vec[(wave * SLOTS + 4 + f) * H + lane] = dwa_c[f];
    __syncthreads();
    for (int idx = tid; idx < SLOTS * H; idx += 256) {
      const int sidx = idx / H, c = idx % H;
      float v = 0.0f;
      for (int w = 0; w < WAVES; ++w) v += vec[(w * SLOTS + sidx) * H + c];
      if (sidx < 4) part[2 * H * H + sidx * H + c] = v;
      else part[2 * H * H + 4 * H + c * 8 + (sidx - 4)] = v;
    }
  }
}

}  // namespace is

extern "C" int is_egnn_edge_bwd_partials_floats(int grid) { return grid * is::PART_STRIDE; }

extern "C" int is_egnn_edge_bwd(const float* ps, const float* pd, int ld_p, const float* x, const float* ea,
                                const int32_t* rowptr, const int32_t* srcs, const float* W1, int ldw, int din,
                                const float* W2, const float* Wc1, const float* wc2, const float* z2s,
                                const float* z3s, const float* g_hn, int ld_ghn, const float* g_xout, float* dZ1,
                                float* dD, float* dPd, int ld_dpd, float* dx, float* partials, int grid, int N,
                                int Fe, void* stream) {
  if (N <= 0) return 0;
  if (Fe < 0 || Fe > 8 || grid <= 0) return -22;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 block(256);
  if (Fe <= 1) {
    hipLaunchKernelGGL(is::egnn_edge_bwd_kernel<1>, dim3(grid), block, 0, st, ps, pd, ld_p, x, ea, rowptr, srcs, W1,
                       ldw, din, W2, Wc1, wc2, z2s, z3s, g_hn, ld_ghn, g_xout, dZ1, dD, dPd, ld_dpd, dx, partials, N, Fe);
  } else {
    hipLaunchKernelGGL(is::egnn_edge_bwd_kernel<8>, dim3(grid), block, 0, st, ps, pd, ld_p, x, ea, rowptr, srcs, W1,
                       ldw, din, W2, Wc1, wc2, z2s, z3s, g_hn, ld_ghn, g_xout, dZ1, dD, dPd, ld_dpd, dx, partials, N, Fe);
  }
  return hipGetLastError() == hipSuccess ? 0 : -5;
}
