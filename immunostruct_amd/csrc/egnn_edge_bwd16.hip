// Fused EGNN edge pass, backward -- 16-row-tile version (v2).
//
// Same mathematics, inputs, outputs and partial-record layout as egnn_edge_bwd.hip (see its
// header for the stage list); mapping changes:
//   * workgroups of 4 waves, one 16-edge tile per wave on v_mfma_f32_16x16x4_f32, two
//     16x64 LDS buffers per wave => ~75 KB LDS, TWO independent workgroups per CU (2 waves per
//     SIMD): one workgroup's gathers / SiLU epilogues overlap the other's MFMAs;
//   * the weight-gradient outer products are distributed by OUTPUT tile instead of by edge
//     tile: wave w owns rows [16w, 16w+16) of dW2 / dWc1 and contracts over all four edge
//     tiles of the 64-edge window, so a wave carries 2 x 16 accumulator registers instead of
//     2 x 64 and no cross-wave reduction is needed at the end;
//   * a workgroup owns one node tile per pass: either NV16 = 16 consecutive destination nodes (~48 edges on
//     degree-3 graphs: the fourth wave of the 64-edge window idles in the row phases), or -- when the caller
//     passes the greedy tile list of graph.py (`tiles`: <= 64 in-edges and <= 24 nodes per tile) -- a
//     node range that fills the window (63 of 64 rows on the same graphs, 24 % fewer passes).
#include "common.h"

namespace is {

#ifdef IS_STAGE_STAMPS
__device__ long long g_stamps_b[24];
#define STAMPB(k) do { if (blockIdx.x == 300 && threadIdx.x == 0 && tile == blockIdx.x) g_stamps_b[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMPB(k) do { } while (0)
#endif

constexpr int WB16 = 4;
constexpr int NV16 = 16;   // nodes per tile without a tile list
constexpr int NVB_LISTED = 24;    // most nodes a listed tile may hold

template <int FE_MAX, int NVB>
struct Bwd16Smem {
  float w2t[H * LD];
  float wc1t[H * LD];
  float bufA[WB16][TE16 * LD];
  float bufB[WB16][TE16 * LD];
  float pdt[NVB * H];   // Pd rows of this tile's destination nodes
  int rp[NVB + 1];
  int e_src[WB16][TE16];
  int e_dl[WB16][TE16];
  float e_rad[WB16][TE16];
  float e_r[WB16][TE16];
  float e_inv[WB16][TE16];
  float e_d[WB16][3][TE16];
  float e_gx[WB16][3][TE16];
  float e_gxd[WB16][TE16];
  float e_s[WB16][TE16];
  float e_drad[WB16][TE16];
  float e_dd[WB16][3][TE16];
  float e_a[WB16][FE_MAX][TE16];
};

constexpr int PART16_STRIDE = 8448 + 64 * 8;  // identical to the v1 record

template <int FE_MAX, int NVB, bool GX>
__global__ __launch_bounds__(256, 2) void egnn_edge_bwd16_kernel(
    const float* __restrict__ ps, const float* __restrict__ pd, int ld_p,
    const float* __restrict__ x, const float* __restrict__ ea,
    const int* __restrict__ rowptr, const int* __restrict__ srcs,
    const float* __restrict__ W1, int ldw, int din,
    const float* __restrict__ W2, const float* __restrict__ Wc1, const float* __restrict__ wc2,
    const float* __restrict__ z2s, const float* __restrict__ z3s,
    const float* __restrict__ g_hn, int ld_ghn, const float* __restrict__ g_xout,
    float* __restrict__ dZ1, float* __restrict__ dD,
    float* __restrict__ dPd, int ld_dpd, float* __restrict__ dx,
    float* __restrict__ partials, const int* __restrict__ tiles, int N, int Fe) {
  __shared__ Bwd16Smem<FE_MAX, NVB> sm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;

  load_matrix_lds_t(sm.w2t, W2, tid, 256);
  if constexpr (GX) load_matrix_lds_t(sm.wc1t, Wc1, tid, 256);

  const float wr_c = W1[lane * ldw + 2 * din];
  float wa_c[FE_MAX];
#pragma unroll
  for (int f = 0; f < FE_MAX; ++f) wa_c[f] = (f < Fe) ? W1[lane * ldw + 2 * din + 1 + f] : 0.0f;
  float wc2_c[4], wr_t[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    wc2_c[nt] = GX ? wc2[nt * 16 + r] : 0.0f;
    wr_t[nt] = W1[(nt * 16 + r) * ldw + 2 * din];
  }

  // wave `wave` owns output rows [16*wave, 16*wave+16) of both weight gradients
  f32x4 dW2[4], dWc1[4];
  zero_acc4(dW2);
  zero_acc4(dWc1);
  float db2_a[4] = {0.f, 0.f, 0.f, 0.f}, dbc1_a[4] = {0.f, 0.f, 0.f, 0.f}, dwc2_a[4] = {0.f, 0.f, 0.f, 0.f};
  float dwr_c = 0.0f;
  float dwa_c[FE_MAX];
#pragma unroll
  for (int f = 0; f < FE_MAX; ++f) dwa_c[f] = 0.0f;

  float* bufA = sm.bufA[wave];
  float* bufB = sm.bufB[wave];
  const int num_tiles = (tiles != nullptr) ? tiles[0] : (N + NV16 - 1) / NV16;
  constexpr int NPW = NVB / WB16;

  for (int tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    STAMPB(0);
    const int v0 = (tiles != nullptr) ? tiles[1 + tile] : tile * NV16;
    const int nv = (tiles != nullptr) ? min(NVB, tiles[2 + tile] - v0) : min(NV16, N - v0);
    __syncthreads();
    if (tid <= NVB) sm.rp[tid] = rowptr[v0 + min(tid, nv)];
#pragma unroll
    for (int i = 0; i < NVB / WB16; ++i) {
      const int nl = wave * (NVB / WB16) + i;
      sm.pdt[nl * H + lane] = (nl < nv) ? pd[(size_t)(v0 + nl) * ld_p + lane] : 0.0f;
    }
    __syncthreads();
    const int e_begin = sm.rp[0], e_end = sm.rp[nv];
    float acc_h[NPW], acc_x[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) { acc_h[i] = 0.0f; acc_x[i] = 0.0f; }

    for (int win = e_begin; win < e_end; win += WB16 * TE16) {
      const int cb = win + wave * TE16;
      const int nvalid = max(0, min(TE16, e_end - cb));
      STAMPB(1);
      float dy[4][4];   // SiLU'(z2), later SiLU'(z1), tile layout
      float up[4][4];   // dL/dh_neigh[dst] for this tile (prefetched)
      float gth[TE16];  // Ps[src] gathers for the z1 recompute (prefetched; Pd[dst] comes from the LDS tile)
      // the saved pre-activation tiles only depend on the window position: issue their loads first so that
      // they are in flight during S0's dependent (src index -> coordinates) chain
      float z3v[4][4], z2v[4][4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          // unconditional (row clamped into the tile's edge range; rows past nvalid are masked where they are used):
          // predicated loads cost the compiler its count of loads in flight, and every later wait becomes vmcnt(0)
          const int row = tile16_row(t, q);
          const size_t off = (size_t)min(cb + row, e_end - 1) * H + nt * 16 + r;
          if constexpr (GX) z3v[t][nt] = z3s[off];
          z2v[t][nt] = z2s[off];
        }
      if (nvalid > 0) {
        // ---- S0: geometry + upstream coordinate gradient, lane = edge ----
        {
          // lanes 16..63 mirror lanes 0..15; every load is unconditional (edge index clamped into the tile's range)
          const int l16 = lane & (TE16 - 1);
          const bool valid = l16 < nvalid;
          const int e = min(cb + l16, e_end - 1);
          const int s = srcs[e];
          int lo = 0, hi = nv;
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (sm.rp[mid] <= e) lo = mid; else hi = mid;
          }
          const int dl = valid ? lo : 0;
          const int v = v0 + dl;
          const float xs0 = x[s * 3 + 0], xs1 = x[s * 3 + 1], xs2 = x[s * 3 + 2];
          const float xv0 = x[v * 3 + 0], xv1 = x[v * 3 + 1], xv2 = x[v * 3 + 2];
          float gx0 = 0.0f, gx1 = 0.0f, gx2 = 0.0f;      // GX = false: the layer's coordinate output has no gradient
          if constexpr (GX) { gx0 = g_xout[v * 3 + 0]; gx1 = g_xout[v * 3 + 1]; gx2 = g_xout[v * 3 + 2]; }
          float av[FE_MAX];
#pragma unroll
          for (int f = 0; f < FE_MAX; ++f) av[f] = (f < Fe) ? ea[(size_t)e * Fe + f] : 0.0f;      // Fe is kernel-uniform
          float d0 = xs0 - xv0, d1 = xs1 - xv1, d2 = xs2 - xv2;
          float rad = radial3(d0, d1, d2);
          float rr = sqrtf(rad);
          float inv = 1.0f / (rr + 1e-30f);
          const float invdeg = 1.0f / (float)max(sm.rp[dl + 1] - sm.rp[dl], 1);
          float g0 = gx0 * invdeg, g1 = gx1 * invdeg, g2 = gx2 * invdeg;
          if (!valid) { d0 = d1 = d2 = rad = rr = inv = g0 = g1 = g2 = 0.0f; }
          if (lane < TE16) {
            sm.e_src[wave][lane] = valid ? s : v0;
            sm.e_dl[wave][lane] = dl;
            sm.e_rad[wave][lane] = rad;
            sm.e_r[wave][lane] = rr;
            sm.e_inv[wave][lane] = inv;
            sm.e_d[wave][0][lane] = d0; sm.e_d[wave][1][lane] = d1; sm.e_d[wave][2][lane] = d2;
            if constexpr (GX) {
              sm.e_gx[wave][0][lane] = g0; sm.e_gx[wave][1][lane] = g1; sm.e_gx[wave][2][lane] = g2;
              sm.e_gxd[wave][lane] = (g0 * d0 + g1 * d1 + g2 * d2) * inv;
            }
#pragma unroll
            for (int f = 0; f < FE_MAX; ++f) sm.e_a[wave][f][lane] = valid ? av[f] : 0.0f;
          }
        }
        __builtin_amdgcn_wave_barrier();
        STAMPB(2);
        // prefetch dL/dh_neigh rows of this tile's destinations: consumed after WG1 + MM3
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = tile16_row(t, q);
          const int v = v0 + sm.e_dl[wave][row];      // e_dl = 0 for rows past nvalid: a valid node
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) up[t][nt] = g_hn[(size_t)v * ld_ghn + nt * 16 + r];
        }

        // ---- E3: coord-MLP tail backward; dz3 -> bufA, mh -> bufB, SiLU'(z2) -> registers ----
        if constexpr (GX) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int row = tile16_row(t, q);
            const bool rv = row < nvalid;
            float tt[4], sp[4];
            float part = 0.0f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
              silu_fg(z3v[t][nt], tt[nt], sp[nt]);
              part += tt[nt] * wc2_c[nt];
              float mh;
              silu_fg(z2v[t][nt], mh, dy[t][nt]);
              bufB[row * LD + nt * 16 + r] = rv ? mh : 0.0f;
            }
            part = sum_over_r16(part);
            if (r == 0) sm.e_s[wave][row] = part;
            const float ds = sm.e_gxd[wave][row];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
              const float dz3 = ds * wc2_c[nt] * sp[nt];
              dwc2_a[nt] += ds * tt[nt];
              dbc1_a[nt] += dz3;
              bufA[row * LD + nt * 16 + r] = dz3;
            }
          }
        } else {
          // no gradient arrives at the coordinate branch: dz3 = 0, so only SiLU'(z2) is needed
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
              float mh;
              silu_fg(z2v[t][nt], mh, dy[t][nt]);
            }
        }
        // prefetch the gathers of the z1 recompute (SA): in flight during WG1 + MM3
#pragma unroll
        for (int i = 0; i < TE16; ++i) {
          const int s = sm.e_src[wave][i];
          gth[i] = ps[(size_t)s * ld_p + lane];      // raw: not consumed before SA
        }
      }
      if constexpr (GX) {
        STAMPB(3);
        __syncthreads();   // every wave's dz3 / mh tiles are staged
        STAMPB(4);

        // ---- WG1: dWc1[16w.., :] += sum over the window's edge tiles of dz3^T mh ----
#pragma unroll
        for (int wt = 0; wt < WB16; ++wt)
          if (win + wt * TE16 < e_end) mm16_outer_rows(dWc1, sm.bufA[wt], sm.bufB[wt], wave, lane);
      }

      if (nvalid > 0) {
        // ---- MM3: dmh = dz3 Wc1 + g_hn[dst] ; dz2 = dmh * SiLU'(z2) ----
        f32x4 acc[4];
        zero_acc4(acc);
        if constexpr (GX) mm16_rows<4, H>(acc, bufA, sm.wc1t, lane);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = tile16_row(t, q);
          const bool rv = row < nvalid;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const float dz2 = rv ? (acc[nt][t] + up[t][nt]) * dy[t][nt] : 0.0f;
            db2_a[nt] += dz2;
            dy[t][nt] = dz2;   // parked in registers until every wave has finished reading bufA / bufB
          }
        }
      }
      if constexpr (GX) {
        STAMPB(5);
        __syncthreads();   // WG1 + MM3 reads of bufA / bufB are complete in all waves
        STAMPB(6);
      }

      if (nvalid > 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) bufA[tile16_row(t, q) * LD + nt * 16 + r] = dy[t][nt];   // dz2

        // ---- SA: recompute z1 (lane = channel) -> bufB ----
        {
#pragma unroll
          for (int i = 0; i < TE16; ++i) {
            float z1 = gth[i] + sm.pdt[sm.e_dl[wave][i] * H + lane] + sm.e_rad[wave][i] * wr_c;
#pragma unroll
            for (int f = 0; f < FE_MAX; ++f) z1 += sm.e_a[wave][f][i] * wa_c[f];
            bufB[i * LD + lane] = (i < nvalid) ? z1 : 0.0f;
          }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- E1: m1 = SiLU(z1) in place (bufB), SiLU'(z1) -> registers ----
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = tile16_row(t, q);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const float z1 = bufB[row * LD + nt * 16 + r];
            float y;
            silu_fg(z1, y, dy[t][nt]);
            bufB[row * LD + nt * 16 + r] = (row < nvalid) ? y : 0.0f;
          }
        }
      }
      STAMPB(7);
      __syncthreads();   // every wave's dz2 / m1 tiles are staged
      STAMPB(8);

      // ---- WG2: dW2[16w.., :] += sum over edge tiles of dz2^T m1 ----
#pragma unroll
      for (int wt = 0; wt < WB16; ++wt)
        if (win + wt * TE16 < e_end) mm16_outer_rows(dW2, sm.bufA[wt], sm.bufB[wt], wave, lane);

      if (nvalid > 0) {
        // ---- MM4: dm1 = dz2 W2 ; dz1 = dm1 * SiLU'(z1) ----
        f32x4 acc[4];
        zero_acc4(acc);
        mm16_rows<4, H>(acc, bufA, sm.w2t, lane);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = tile16_row(t, q);
          const bool rv = row < nvalid;
          float part = 0.0f;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const float dz1 = rv ? acc[nt][t] * dy[t][nt] : 0.0f;
            if (rv) dZ1[(size_t)(cb + row) * H + nt * 16 + r] = dz1;
            dy[t][nt] = dz1;
            part += dz1 * wr_t[nt];
          }
          part = sum_over_r16(part);
          if (r == 0) sm.e_drad[wave][row] = part;
        }
      }
      STAMPB(9);
      __syncthreads();   // WG2 + MM4 reads complete in all waves
      STAMPB(10);

      if (nvalid > 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) bufA[tile16_row(t, q) * LD + nt * 16 + r] = dy[t][nt];   // dz1
        __builtin_amdgcn_wave_barrier();

        // ---- SB: dw_r / dW_a partial sums over this wave's tile, lane = channel ----
#pragma unroll
        for (int i = 0; i < TE16; ++i) {
          const float v = bufA[i * LD + lane];
          dwr_c += v * sm.e_rad[wave][i];
#pragma unroll
          for (int f = 0; f < FE_MAX; ++f) dwa_c[f] += v * sm.e_a[wave][f][i];
        }

        // ---- GEO: gradient wrt d = x_src - x_dst, lane = edge ----
        if (lane < TE16) {
          const bool valid = lane < nvalid;
          float q0 = 0.f, q1 = 0.f, q2 = 0.f;
          if (valid) {
            const float inv = sm.e_inv[wave][lane], rr = sm.e_r[wave][lane];
            const float d0 = sm.e_d[wave][0][lane], d1 = sm.e_d[wave][1][lane], d2 = sm.e_d[wave][2][lane];
            float u0 = 0.0f, u1 = 0.0f, u2 = 0.0f;
            if constexpr (GX) {
              const float s = sm.e_s[wave][lane];
              u0 = s * sm.e_gx[wave][0][lane]; u1 = s * sm.e_gx[wave][1][lane]; u2 = s * sm.e_gx[wave][2][lane];
            }
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
      }
      __syncthreads();

      STAMPB(11);
      // ---- SEG: destination-side segment sums (deterministic, CSR order) ----
      {
        const int win_hi = min(win + WB16 * TE16, e_end);
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
          const int nl = wave + WB16 * i;
          if (nl < nv) {
            const int lo = max(sm.rp[nl], win), hi = min(sm.rp[nl + 1], win_hi);
            float ah = acc_h[i], ax = acc_x[i];
            for (int e = lo; e < hi; ++e) {
              const int rel = e - win;
              const int w = rel >> 4, row = rel & 15;
              ah += sm.bufA[w][row * LD + lane];
              if (lane < 3) ax += sm.e_dd[w][lane][row];
            }
            acc_h[i] = ah; acc_x[i] = ax;
          }
        }
      }
      __syncthreads();
    }

    STAMPB(12);
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int nl = wave + WB16 * i;
      if (nl < nv) {
        const int v = v0 + nl;
        dPd[(size_t)v * ld_dpd + lane] = acc_h[i];
        if (lane < 3) dx[v * 3 + lane] = (GX ? g_xout[v * 3 + lane] : 0.0f) - acc_x[i];
      }
    }
  }

  // ---- write the workgroup's partial record: each wave owns 16 rows of dW2 / dWc1 ----
  __syncthreads();
  float* part = partials + (size_t)blockIdx.x * PART16_STRIDE;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int o = wave * 16 + tile16_row(t, q), i = nt * 16 + r;
      part[o * H + i] = dW2[nt][t];
      part[H * H + o * H + i] = dWc1[nt][t];
    }
  {
    float* vec = &sm.bufB[0][0];  // [wave][slot][64]
    constexpr int SLOTS = 4 + FE_MAX;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      float vals[3] = {db2_a[nt], dbc1_a[nt], dwc2_a[nt]};
#pragma unroll
      for (int sidx = 0; sidx < 3; ++sidx) {
        float v = vals[sidx];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (q == 0) vec[(wave * SLOTS + sidx) * H + nt * 16 + r] = v;
      }
    }
    vec[(wave * SLOTS + 3) * H + lane] = dwr_c;
#pragma unroll
    for (int f = 0; f < FE_MAX; ++f) vec[(wave * SLOTS + 4 + f) * H + lane] = dwa_c[f];
    __syncthreads();
    for (int idx = tid; idx < SLOTS * H; idx += 256) {
      const int sidx = idx / H, c = idx % H;
      float v = 0.0f;
      for (int w = 0; w < WB16; ++w) v += vec[(w * SLOTS + sidx) * H + c];
      if (sidx < 4) part[2 * H * H + sidx * H + c] = v;
      else part[2 * H * H + 4 * H + c * 8 + (sidx - 4)] = v;
    }
  }
}

}  // namespace is

#ifdef IS_STAGE_STAMPS
extern "C" int is_debug_stamps_bwd(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(is::g_stamps_b), sizeof(long long) * 24) == hipSuccess ? 0 : -5;
}
#endif

extern "C" int is_egnn_edge_bwd_v2(const float* ps, const float* pd, int ld_p, const float* x, const float* ea,
                                   const int32_t* rowptr, const int32_t* srcs, const float* W1, int ldw, int din,
                                   const float* W2, const float* Wc1, const float* wc2, const float* z2s,
                                   const float* z3s, const float* g_hn, int ld_ghn, const float* g_xout, float* dZ1,
                                   float* dD, float* dPd, int ld_dpd, float* dx, float* partials,
                                   const int32_t* tiles, int grid, int N, int Fe, void* stream) {
  if (N <= 0) return 0;
  if (Fe < 0 || Fe > 8 || grid <= 0) return -22;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 block(256);
  // g_xout == nullptr: no gradient arrives at this layer's coordinate output (the last layer of a stack whose final
  // coordinates are not used): the coordinate-MLP half of the pass (z3s reads, dz3, dWc1, dz3 Wc1) is skipped, its
  // weight-gradient entries of the partial record are zero; z3s / Wc1 / wc2 are not read
  if (g_xout != nullptr && z3s == nullptr) return -22;
#define IS_LAUNCH_BWD16(FE, NVB, GXF)                                                                                           \
  hipLaunchKernelGGL((is::egnn_edge_bwd16_kernel<FE, NVB, GXF>), dim3(grid), block, 0, st, ps, pd, ld_p, x, ea, rowptr, srcs, W1, \
                     ldw, din, W2, Wc1, wc2, z2s, z3s, g_hn, ld_ghn, g_xout, dZ1, dD, dPd, ld_dpd, dx, partials, tiles, N, Fe)
#define IS_LAUNCH_BWD16_GX(FE, NVB) \
  do { if (g_xout != nullptr) IS_LAUNCH_BWD16(FE, NVB, true); else IS_LAUNCH_BWD16(FE, NVB, false); } while (0)
  if (Fe <= 1) {
    if (tiles != nullptr) IS_LAUNCH_BWD16_GX(1, is::NVB_LISTED); else IS_LAUNCH_BWD16_GX(1, is::NV16);
  } else {
    if (tiles != nullptr) return -22;      // listed tiles: Fe <= 1 only (the Fe = 8 instantiation has no LDS left)
    IS_LAUNCH_BWD16_GX(8, is::NV16);
  }
#undef IS_LAUNCH_BWD16_GX
#undef IS_LAUNCH_BWD16
  return hipGetLastError() == hipSuccess ? 0 : -5;
}
