// Fused EGNN edge pass, forward -- 16-row-tile version (v2).
//
// Same algorithm and outputs as egnn_edge_fwd.hip (see its header); what changes is the
// mapping onto the CU: workgroups of 8 waves, one 16-edge tile per wave on
// v_mfma_f32_16x16x4_f32, wave-private stages separated by wave-level (not workgroup)
// synchronisation.  LDS per workgroup drops to ~74 KB => 2 workgroups = 16 waves per CU
// (4 per SIMD) so one wave's gather / SiLU epilogue overlaps another wave's MFMAs, and
// the 16-edge granularity halves the padding waste on ~100-edge node tiles.
#include "common.h"

namespace is {

// Optional per-stage timestamps (debug builds only: -DIS_STAGE_STAMPS): workgroup 300, wave 0.
#ifdef IS_STAGE_STAMPS
__device__ long long g_stamps[16];
#define STAMP(k) do { if (blockIdx.x == 300 && threadIdx.x == 0) g_stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

constexpr int W16 = 8;  // waves per workgroup

template <int FE_MAX>
struct Fwd16Smem {
  float w2[H * LD];
  float wc1[H * LD];
  float act[W16][TE16 * LD];
  int rp[NV + 1];
  int e_src[W16][TE16];
  int e_dl[W16][TE16];
  float e_rad[W16][TE16];
  float e_xd[W16][3][TE16];
  float e_s[W16][TE16];
  float e_a[W16][FE_MAX][TE16];
};

template <int FE_MAX>
__global__ __launch_bounds__(512, 4) void egnn_edge_fwd16_kernel(
    const float* __restrict__ ps, const float* __restrict__ pd, int ld_p,
    const float* __restrict__ x, const float* __restrict__ ea,
    const int* __restrict__ rowptr, const int* __restrict__ srcs,
    const float* __restrict__ W1, int ldw, int din,
    const float* __restrict__ W2, const float* __restrict__ b2,
    const float* __restrict__ Wc1, const float* __restrict__ bc1, const float* __restrict__ wc2,
    float* __restrict__ h_neigh, int ld_hn, float* __restrict__ x_out,
    float* __restrict__ z2s, float* __restrict__ z3s, int N, int Fe) {
  __shared__ Fwd16Smem<FE_MAX> sm;
  STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int v0 = blockIdx.x * NV;
  const int nv = min(NV, N - v0);

  // weight tiles: issue the global loads now, park them in registers and write them to LDS only before the
  // first MFMA stage -- their latency overlaps the rowptr load and the S0 / SA stages of the first window
  f32x4 wreg[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = tid + (j & 1) * 512;                      // 1024 float4 per matrix, 512 threads
    const float* src = (j < 2) ? W2 : Wc1;
    wreg[j] = *reinterpret_cast<const f32x4*>(src + idx * 4);
  }
  bool weights_staged = false;
  if (tid <= NV) sm.rp[tid] = rowptr[v0 + min(tid, nv)];

  const float wr_c = W1[lane * ldw + 2 * din];
  float wa_c[FE_MAX];
#pragma unroll
  for (int f = 0; f < FE_MAX; ++f) wa_c[f] = (f < Fe) ? W1[lane * ldw + 2 * din + 1 + f] : 0.0f;
  float b2_c[4], bc1_c[4], wc2_c[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    b2_c[nt] = b2[nt * 16 + r];
    bc1_c[nt] = bc1[nt * 16 + r];
    wc2_c[nt] = wc2[nt * 16 + r];
  }
  __syncthreads();

  STAMP(1);
  const int e_begin = sm.rp[0], e_end = sm.rp[nv];
  constexpr int NPW = NV / W16;  // nodes per wave in the segment phase
  float acc_h[NPW], acc_x[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) { acc_h[i] = 0.0f; acc_x[i] = 0.0f; }
  float* act = sm.act[wave];

  for (int win = e_begin; win < e_end; win += W16 * TE16) {
    const int cb = win + wave * TE16;
    const int nvalid = max(0, min(TE16, e_end - cb));
    if (nvalid > 0) {   // wave-uniform: everything in here touches wave-private LDS only
      // ---- S0a: source / destination ids of the tile (lanes 0..15 = edges) ----
      if (lane < TE16) {
        const bool valid = lane < nvalid;
        const int e = cb + lane;
        int s = v0, dl = 0;
        if (valid) {
          s = srcs[e];
          int lo = 0, hi = nv;
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (sm.rp[mid] <= e) lo = mid; else hi = mid;
          }
          dl = lo;
        }
        sm.e_src[wave][lane] = s;
        sm.e_dl[wave][lane] = dl;
      }
      __builtin_amdgcn_wave_barrier();
      // ---- gathers (lane = channel) are issued as soon as the ids are known; the coordinate loads and the
      //      geometry of S0b run while they are in flight ----
      float g[TE16];
#pragma unroll
      for (int i = 0; i < TE16; ++i) {
        const int s = sm.e_src[wave][i];
        const int v = v0 + sm.e_dl[wave][i];
        g[i] = ps[(size_t)s * ld_p + lane] + pd[(size_t)v * ld_p + lane];
      }
      // ---- S0b: geometry + edge features ----
      if (lane < TE16) {
        const bool valid = lane < nvalid;
        const int e = cb + lane;
        float d0 = 0.f, d1 = 0.f, d2 = 0.f, rad = 0.f;
        if (valid) {
          const int s = sm.e_src[wave][lane];
          const int v = v0 + sm.e_dl[wave][lane];
          d0 = x[s * 3 + 0] - x[v * 3 + 0];
          d1 = x[s * 3 + 1] - x[v * 3 + 1];
          d2 = x[s * 3 + 2] - x[v * 3 + 2];
          rad = radial3(d0, d1, d2);
          const float inv = 1.0f / (sqrtf(rad) + 1e-30f);
          d0 *= inv; d1 *= inv; d2 *= inv;
        }
        sm.e_rad[wave][lane] = rad;
        sm.e_xd[wave][0][lane] = d0;
        sm.e_xd[wave][1][lane] = d1;
        sm.e_xd[wave][2][lane] = d2;
#pragma unroll
        for (int f = 0; f < FE_MAX; ++f)
          sm.e_a[wave][f][lane] = (valid && f < Fe) ? ea[(size_t)e * Fe + f] : 0.0f;
      }
      __builtin_amdgcn_wave_barrier();
      STAMP(2);

      // ---- SA: first edge-MLP layer, lane = channel ----
#pragma unroll
      for (int i = 0; i < TE16; ++i) {
        float z1 = g[i] + sm.e_rad[wave][i] * wr_c;
#pragma unroll
        for (int f = 0; f < FE_MAX; ++f) z1 += sm.e_a[wave][f][i] * wa_c[f];
        act[i * LD + lane] = (i < nvalid) ? silu_f(z1) : 0.0f;
      }
      __builtin_amdgcn_wave_barrier();
    }
    STAMP(3);
    if (!weights_staged) {     // workgroup-uniform: first window only
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = tid + (j & 1) * 512;
        float* dst = (j < 2) ? sm.w2 : sm.wc1;
        *reinterpret_cast<f32x4*>(dst + (idx / (H / 4)) * LD + (idx % (H / 4)) * 4) = wreg[j];
      }
      weights_staged = true;
      __syncthreads();
    }
    STAMP(4);
    if (nvalid > 0) {
      // ---- MM1: z2 = m1 W2^T + b2 ; mh = SiLU(z2) ----
      {
        f32x4 acc[4];
        zero_acc4(acc);
        mm16_rows<4, H>(acc, act, sm.w2, lane);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int row = tile16_row(t, q);
            const float z2 = acc[nt][t] + b2_c[nt];
            if (z2s != nullptr && row < nvalid) z2s[(size_t)(cb + row) * H + nt * 16 + r] = z2;
            act[row * LD + nt * 16 + r] = (row < nvalid) ? silu_f(z2) : 0.0f;
          }
      }
      __builtin_amdgcn_wave_barrier();
      STAMP(5);

      // ---- MM2: z3 = mh Wc1^T + bc1 ; s = SiLU(z3) . wc2 ----
      {
        f32x4 acc[4];
        zero_acc4(acc);
        mm16_rows<4, H>(acc, act, sm.wc1, lane);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = tile16_row(t, q);
          float part = 0.0f;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const float z3 = acc[nt][t] + bc1_c[nt];
            if (z3s != nullptr && row < nvalid) z3s[(size_t)(cb + row) * H + nt * 16 + r] = z3;
            part += silu_f(z3) * wc2_c[nt];
          }
          part = sum_over_r16(part);
          if (r == 0) sm.e_s[wave][row] = part;
        }
      }
    }
    STAMP(6);
    __syncthreads();
    STAMP(7);

    // ---- SEG: deterministic segment reduction over this window ----
    {
      const int win_hi = min(win + W16 * TE16, e_end);
#pragma unroll
      for (int i = 0; i < NPW; ++i) {
        const int nl = wave + W16 * i;
        if (nl < nv) {
          const int lo = max(sm.rp[nl], win), hi = min(sm.rp[nl + 1], win_hi);
          float ah = acc_h[i], ax = acc_x[i];
          for (int e = lo; e < hi; ++e) {
            const int rel = e - win;
            const int w = rel >> 4, row = rel & 15;
            ah += sm.act[w][row * LD + lane];
            if (lane < 3) ax += sm.e_s[w][row] * sm.e_xd[w][lane][row];
          }
          acc_h[i] = ah; acc_x[i] = ax;
        }
      }
    }
    __syncthreads();
  }

  STAMP(8);
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int nl = wave + W16 * i;
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

#ifdef IS_STAGE_STAMPS
extern "C" int is_debug_stamps(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(is::g_stamps), sizeof(long long) * 16) == hipSuccess ? 0 : -5;
}
#endif

extern "C" int is_egnn_edge_fwd_v2(const float* ps, const float* pd, int ld_p, const float* x, const float* ea,
                                   const int32_t* rowptr, const int32_t* srcs, const float* W1, int ldw, int din,
                                   const float* W2, const float* b2, const float* Wc1, const float* bc1,
                                   const float* wc2, float* h_neigh, int ld_hn, float* x_out, float* z2s,
                                   float* z3s, int N, int Fe, void* stream) {
  if (N <= 0) return 0;
  if (Fe < 0 || Fe > 8) return -22;
  const dim3 grid((N + is::NV - 1) / is::NV), block(512);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (Fe <= 1) {
    hipLaunchKernelGGL(is::egnn_edge_fwd16_kernel<1>, grid, block, 0, st, ps, pd, ld_p, x, ea, rowptr, srcs, W1, ldw, din,
                       W2, b2, Wc1, bc1, wc2, h_neigh, ld_hn, x_out, z2s, z3s, N, Fe);
  } else {
    hipLaunchKernelGGL(is::egnn_edge_fwd16_kernel<8>, grid, block, 0, st, ps, pd, ld_p, x, ea, rowptr, srcs, W1, ldw, din,
                       W2, b2, Wc1, bc1, wc2, h_neigh, ld_hn, x_out, z2s, z3s, N, Fe);
  }
  return hipGetLastError() == hipSuccess ? 0 : -5;
}
