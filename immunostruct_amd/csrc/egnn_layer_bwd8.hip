// One EGNNConv layer, backward -- the PAIRED form of egnn_layer_bwd.hip (round 5): ONE workgroup of 512 threads per CU instead of
// two of 256.  The two halves of the workgroup ("groups" of 4 waves) are what the two co-resident workgroups of the 256-thread
// kernel were -- group g of workgroup b owns the tiles of the old workgroup b + g * gridDim.x, runs the same node phase and the same
// 64-edge windows on its own LDS buffers, keeps its own weight-gradient accumulators -- but they
//   * share ONE staged copy of the two transposed weight tiles (w2t, wc1t: 34.8 KB).  The LDS this frees lets the node phase's
//     activation tiles (48 rows per group) live BESIDE the weight tiles instead of in their place, so the weight tiles and the
//     rowptr slices of all tiles are requested at the very start of the kernel, under the node phase's bandwidth-bound front,
//     instead of behind its last barrier (stage stamps of the 256-thread kernel: 6.5 - 13 k of 158 k cycles between the end of
//     the node phase and the first window);
//   * write ONE partial weight-gradient record per CU: both groups' accumulators go through LDS and are added in a fixed order
//     (group 0 + group 1) with coalesced stores -- half the record bytes per launch (18.4 -> 9.2 MB at B = 128) and half the
//     input of reduce_partials_batched.
// Barriers are workgroup-wide, so both groups run the same number of passes, tiles and windows (a group without work in an
// iteration only meets the barriers).  Built for the library's default forms (z1 and the edge geometry read back, z3 read back:
// IS_LAYER_M1 = 2, IS_LAYER_GEO = 1) and Fe <= 1; everything else stays on egnn_layer_bwd.hip.
// Reference: dgl.nn.EGNNConv backward (third party; constructed models/hybrid_models.py:261-263, called :323-324).
#ifndef IS_LAYER_M1
#define IS_LAYER_M1 2
#endif
#ifndef IS_LAYER_GEO
#define IS_LAYER_GEO 1
#endif
#include "common.h"
#include "node16.h"

#if IS_LAYER_M1 == 2 && IS_LAYER_GEO == 1
#define IS_BWD8_BUILT 1
#else
#define IS_BWD8_BUILT 0
#endif

namespace is {

#ifdef IS_STAGE_STAMPS
__device__ long long g_stamps_b8[24];
#define STAMP8_WG 150
#ifndef STAMP8_TK
#define STAMP8_TK 0      // which tile (round) of the stamped workgroup the window stamps belong to (-DSTAMP8_TK=1: a steady-state one)
#endif
#define STAMPB8(k) do { if (blockIdx.x == STAMP8_WG && threadIdx.x == 0 && tk == STAMP8_TK) g_stamps_b8[k] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMPP8(k) do { if (blockIdx.x == STAMP8_WG && threadIdx.x == 0) g_stamps_b8[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMPB8(k) do { } while (0)
#define STAMPP8(k) do { } while (0)
#endif

constexpr int P8_WAVES = 4;          // waves per group
constexpr int P8_NV = 16;            // nodes per tile
constexpr int P8_RP_TILES = 8;       // tiles per group whose rowptr slice is fetched ahead
constexpr int P8_PROWS = 48;         // rows of a node-phase pass per group: three tiles (B = 128: all of a group's tiles in one pass)
constexpr int P8_LDP = 132;
constexpr int P8_RECORD = 8448 + 64 * 8;      // identical to egnn_layer_bwd.hip PART16_STRIDE

constexpr int P8_VEC = P8_WAVES * 3 * H;      // per-wave bias sums on their way into the record

template <int FE_MAX>
struct alignas(16) Bwd8Group {
  union {
    struct { float ps[P8_PROWS * P8_LDP]; float gs[P8_PROWS * LD]; float zs[P8_PROWS * LD]; } node;      // 51.5 KB
    struct { float bufA[P8_WAVES][TE16 * LD]; float bufB[P8_WAVES][TE16 * LD]; } win;                    // 34.8 KB
    struct { float rec[P8_RECORD]; float vec[P8_VEC]; } out;                                              // 38.9 KB
  } u;
  int rp[P8_NV + 1];
  int rp_tab[P8_RP_TILES][P8_NV + 1];
  int e_dl[P8_WAVES][TE16];
  float e_ra[P8_WAVES][TE16 * (FE_MAX + 1)];
  float e_r[P8_WAVES][TE16];
  float e_inv[P8_WAVES][TE16];
  float e_d[P8_WAVES][3][TE16];
  float e_gx[P8_WAVES][3][TE16];
  float e_gxd[P8_WAVES][TE16];
  float e_s[P8_WAVES][TE16];
  float e_drad[P8_WAVES][TE16];
};

template <int FE_MAX>
struct alignas(16) Bwd8Smem {
  float w2t[H * LD];
  float wc1t[H * LD];
  Bwd8Group<FE_MAX> g[2];
};

struct NodeBwd8Args {      // (the NodeBwdArgs of egnn_layer_bwd.hip)
  const float* dZ1n;
  const float* dDn;
  const float* dxn;
  const int* rowptr_src;
  const int* pos_by_src;
  const float* g_h;
  float* g_psd;
  const float* zn1;
  const float* bpack;
  float* dh_total;
  float* dzn1;
  float* d_h;
  float* d_hn;
  float* gxtot;
};

#if IS_BWD8_BUILT
template <int FE_MAX, bool GX, bool GATHER, int DIN>
__global__ __launch_bounds__(512) void egnn_layer_bwd8_kernel(
    const float* __restrict__ ea, const int* __restrict__ rowptr,
    const float* __restrict__ W1,
    const float* __restrict__ W2, const float* __restrict__ Wc1, const float* __restrict__ wc2,
    const float* __restrict__ z2s, const float* __restrict__ z3s,
    const float* __restrict__ g_xout,
    float* __restrict__ dZ1, float* __restrict__ dD,
    float* __restrict__ dPd, float* __restrict__ dx,
    float* __restrict__ partials, int N, int Fe, NodeBwd8Args nb, long long* __restrict__ wg_clock,
    const float* __restrict__ m1s, const float* __restrict__ geos) {
  static_assert(!GATHER || GX, "a gathered layer always receives a coordinate gradient");
  using D = Node16Dims<DIN>;
  constexpr int MT = P8_PROWS / 16, PROWS = P8_PROWS, LDP = P8_LDP;
  constexpr int TPP = PROWS / P8_NV;                         // tiles per pass and group
  constexpr int NVB = P8_NV, WB16 = P8_WAVES;
  __shared__ Bwd8Smem<FE_MAX> sm;
  wg_clock_start(wg_clock);
  constexpr int ld_dpd = 2 * H, din = DIN;
  const int ldw = 2 * DIN + 1 + Fe;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave8 >> 2, wave = wave8 & 3;              // group of this wave, wave inside the group (both wave-uniform)
  const int gtid = tid & 255;
  const int r = lane & 15, q = lane >> 4;
  constexpr int RA_LD = FE_MAX + 1;
  const int num_tiles = (N + NVB - 1) / NVB;
  const int G = 2 * (int)gridDim.x;                          // the virtual grid: group g of workgroup b is "workgroup" b + g * gridDim.x
  const int vb = (int)blockIdx.x + grp * (int)gridDim.x;
  // trip counts of the WORKGROUP (group 0 has the lower virtual id: at least as many tiles as group 1)
  const int ntk = (num_tiles - (int)blockIdx.x + G - 1) / G;
  Bwd8Group<FE_MAX>& sg = sm.g[grp];
  const float* __restrict__ gxsrc = GATHER ? nb.gxtot : g_xout;

  STAMPP8(13);
  // ---- requested first, stored to LDS in front of the node phase's first barrier: the two weight tiles (transposed) and the
  //      rowptr slices of this group's tiles.  Nothing in the node phase touches these LDS regions.
  f32x4 wv2[2], wvc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    wv2[j] = *reinterpret_cast<const f32x4*>(W2 + (tid + j * 512) * 4);
    wvc[j] = GX ? *reinterpret_cast<const f32x4*>(Wc1 + (tid + j * 512) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  int rpv = 0;
  {
    const int k = gtid / (NVB + 1), i = gtid - k * (NVB + 1);
    const int t = vb + k * G;
    if (gtid < P8_RP_TILES * (NVB + 1) && t < num_tiles) {
      const int a0 = t * NVB;
      rpv = rowptr[a0 + min(i, min(NVB, N - a0))];
    }
  }
  auto stage_early = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = tid + j * 512;
      const int row = idx / (H / 4), c4 = (idx % (H / 4)) * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sm.w2t[(c4 + k) * LD + row] = wv2[j][k];
        if constexpr (GX) sm.wc1t[(c4 + k) * LD + row] = wvc[j][k];
      }
    }
    if (gtid < P8_RP_TILES * (NVB + 1)) (&sg.rp_tab[0][0])[gtid] = rpv;
  };

  // ================= P1 + P2: source gather and node data path of ALL tiles of this group =================
  {
    float* ps_ = sg.u.node.ps;                // [PROWS][LDP]  g_psd rows
    float* gs = sg.u.node.gs;                 // [PROWS][LD]   dh
    float* zs = sg.u.node.zs;                 // [PROWS][LD]   dzn1
    const bool has_psd = nb.g_psd != nullptr;
    const int col = wave * 16 + r;
    const f32x4* pk = reinterpret_cast<const f32x4*>(nb.bpack) + (size_t)wave * NODE_BWD_SLOTS * 64 + lane;
    const int npass = (ntk + TPP - 1) / TPP;      // (>= 1: a launched workgroup owns a tile)
    // the weight tiles / rowptr slices go to LDS before the pass requests its 100 + registers of rows: one L2 round trip (every
    // workgroup reads the same 32 KB); the node phase's first barrier publishes them
    stage_early();
#ifdef IS_ABL_NONODE
    for (int pass = npass; pass < npass; ++pass) {
#else
    for (int pass = 0; pass < npass; ++pass) {
#endif
      // tiles of this group in this pass: tk = pass * TPP + k, tile = vb + tk * G
      int ta0[TPP], tcnt[TPP];
      int ntp = 0;
#pragma unroll
      for (int k = 0; k < TPP; ++k) {
        const int tl = vb + (pass * TPP + k) * G;
        int a0 = 0, cnt = 0;
        if (tl < num_tiles) {
          a0 = tl * NVB;
          cnt = min(NVB, N - a0);
          ntp = k + 1;
        }
        ta0[k] = __builtin_amdgcn_readfirstlane(a0);
        tcnt[k] = __builtin_amdgcn_readfirstlane(cnt);
      }
      const int mt_used = __builtin_amdgcn_readfirstlane(ntp);      // (one 16-row tile per node tile)
      auto row_node = [&](int lr) {
        const int k = lr / NVB, i = lr % NVB;
        int a0 = ta0[0], cnt = tcnt[0];
#pragma unroll
        for (int kk = 1; kk < TPP; ++kk) { a0 = (k == kk) ? ta0[kk] : a0; cnt = (k == kk) ? tcnt[kk] : cnt; }
        return (i < cnt) ? a0 + i : -1;
      };
      [[maybe_unused]] const int g_sub = gtid & 3;       // gather role: 4 lanes per node, row g_lr of the pass
      [[maybe_unused]] const int g_lr = gtid >> 2;
      [[maybe_unused]] int g_v = -1, g_lo = 0, g_hi = 0;
      if constexpr (GATHER) {
        g_v = (g_lr < mt_used * 16) ? row_node(g_lr) : -1;
        const int vc = max(g_v, 0);
        g_lo = nb.rowptr_src[vc];
        g_hi = nb.rowptr_src[vc + 1];
      }
      float zpre[MT][4], gpre[MT][4];
      int vrow[MT][4];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int v = (mt < mt_used) ? row_node(mt * 16 + tile16_row(t, q)) : -1;
          vrow[mt][t] = v;
          const int vc = max(v, 0);
          zpre[mt][t] = nb.zn1[(size_t)vc * H + col];
          gpre[mt][t] = (has_psd && nb.g_h != nullptr) ? nb.g_h[(size_t)vc * H + col] : 0.0f;
        }
      float bp[32];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const f32x4 v = pk[g * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) bp[4 * g + j] = v[j];
      }
      if constexpr (GATHER) {
        // (the gather of egnn_layer_bwd.hip: 4 lanes per node, slot ids of the first 8 out-edges as one level, rows in rounds of 4)
        const int sub = g_sub, lr = g_lr, v = g_v;
        const int lo = g_lo, hi = (v >= 0) ? g_hi : g_lo;
        constexpr int VIEW = 0x7fffe000;
        const rsrc_t rs_z = make_rsrc_n(nb.dZ1n, VIEW), rs_d = make_rsrc_n(nb.dDn, VIEW), rs_p = make_rsrc_n(nb.pos_by_src, VIEW);
        constexpr int GB = 4;
        int e[2 * GB];
#pragma unroll
        for (int k = 0; k < 2 * GB; ++k) e[k] = buf_load_i(rs_p, (lo + k < hi) ? (lo + k) * 4 : BUF_OOB, 0);
        f32x4 acc[4], pdv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
          pdv[j] = *reinterpret_cast<const f32x4*>(nb.g_psd + (size_t)max(v, 0) * 128 + 64 + 16 * j + 4 * sub);
        }
        const float x_dst = (sub < 3) ? nb.dxn[max(v, 0) * 3 + sub] : 0.0f;
        float acc3 = 0.0f;
        auto round = [&](const int* ek, int p0) {
          f32x4 a[GB][4];
          float d[GB];
#pragma unroll
          for (int k = 0; k < GB; ++k) {
            const bool on = p0 + k < hi;
            const int zoff = on ? ek[k] * (H * 4) + sub * 16 : BUF_OOB;
#pragma unroll
            for (int j = 0; j < 4; ++j) a[k][j] = buf_load4(rs_z, zoff + 64 * j, 0);
            d[k] = buf_load(rs_d, (on && sub < 3) ? ek[k] * 12 + sub * 4 : BUF_OOB, 0);
          }
#pragma unroll
          for (int k = 0; k < GB; ++k) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += a[k][j];
            acc3 += d[k];
          }
        };
        round(e, lo);
        if (lo + GB < hi) round(e + GB, lo + GB);
        for (int p = lo + 2 * GB; p < hi; p += GB) {
          int e2[GB];
#pragma unroll
          for (int k = 0; k < GB; ++k) e2[k] = buf_load_i(rs_p, (p + k < hi) ? (p + k) * 4 : BUF_OOB, 0);
          round(e2, p);
        }
        if (v >= 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f32x4*>(nb.g_psd + (size_t)v * 128 + 16 * j + 4 * sub) = acc[j];
          if (sub < 3) nb.gxtot[v * 3 + sub] = x_dst + acc3;
        }
        if (lr < PROWS) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<f32x4*>(ps_ + lr * LDP + 16 * j + 4 * sub) = acc[j];
            *reinterpret_cast<f32x4*>(ps_ + lr * LDP + 64 + 16 * j + 4 * sub) = (v >= 0) ? pdv[j] : f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      } else {
        constexpr int RPW = PROWS / WB16;
        float v0r[RPW], v1r[RPW];
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
          const int vc = max(row_node(wave * RPW + i), 0);
          if (has_psd) {
            v0r[i] = nb.g_psd[(size_t)vc * 128 + lane];
            v1r[i] = nb.g_psd[(size_t)vc * 128 + 64 + lane];
          } else {
            v0r[i] = nb.g_h != nullptr ? nb.g_h[(size_t)vc * H + lane] : 0.0f;
            v1r[i] = 0.0f;
          }
        }
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
          const int lr = wave * RPW + i;
          const bool valid = row_node(lr) >= 0;
          if (has_psd) {
            ps_[lr * LDP + lane] = valid ? v0r[i] : 0.0f;
            ps_[lr * LDP + 64 + lane] = valid ? v1r[i] : 0.0f;
          } else {
            gs[lr * LD + lane] = valid ? v0r[i] : 0.0f;
          }
        }
      }
      float ba[16], bx[2][16];
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
      __syncthreads();
      STAMPP8(14);
      if (has_psd) {      // dh = g_h + g_psd W1sd
        f32x4 acc[MT];
        zero_acc4(acc);
        mm16_regBt_used<MT, 32, LDP>(acc, ps_, bp, lane, mt_used);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int lr = mt * 16 + tile16_row(t, q);
            float v = 0.0f;
            if (vrow[mt][t] >= 0) {
              v = acc[mt][t] + gpre[mt][t];
              nb.dh_total[(size_t)vrow[mt][t] * H + col] = v;
            }
            gs[lr * LD + col] = v;
          }
        __syncthreads();
      }
      STAMPP8(15);
      {     // da1 = dh Wn2 ; dzn1 = da1 * SiLU'(zn1)
        f32x4 acc[MT];
        zero_acc4(acc);
        mm16_regBt_used<MT, 16, LD>(acc, gs, ba, lane, mt_used);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int lr = mt * 16 + tile16_row(t, q);
            float dz = 0.0f;
            if (vrow[mt][t] >= 0) {
              float y, dyv;
              silu_fg(zpre[mt][t], y, dyv);
              dz = acc[mt][t] * dyv;
              nb.dzn1[(size_t)vrow[mt][t] * H + col] = dz;
            }
            zs[lr * LD + col] = dz;
          }
      }
      __syncthreads();
      STAMPP8(16);
      // [d_h | d_hneigh] = dzn1 Wn1: wave w produces columns [32w, 32w + 32) of the (DIN + 64)-wide row
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int xc = (wave * 2 + nt) * 16 + r;
        if ((wave * 2 + nt) * 16 < D::KV) {
          f32x4 acc[MT];
          zero_acc4(acc);
          mm16_regBt_used<MT, 16, LD>(acc, zs, bx[nt], lane, mt_used);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int v = vrow[mt][t];
              if (v >= 0 && xc < D::KV) {
                if (xc < DIN) { if (nb.d_h != nullptr) nb.d_h[(size_t)v * H + xc] = acc[mt][t]; }
                else nb.d_hn[(size_t)v * H + (xc - DIN)] = acc[mt][t];
              }
            }
        }
      }
      __syncthreads();     // the pass's tiles are dead; d_hn / gxtot rows of these tiles are visible to the whole workgroup
      STAMPP8(17);
    }
#ifdef IS_ABL_NONODE
    __syncthreads();
#endif
  }

  const float* __restrict__ g_hn = nb.d_hn;
  float wc2_c[4], wr_t[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    wc2_c[nt] = GX ? wc2[nt * 16 + r] : 0.0f;
    wr_t[nt] = W1[(nt * 16 + r) * ldw + 2 * din];
  }

  // wave `wave` of each group owns output rows [16*wave, 16*wave+16) of its group's weight gradients
  f32x4 dW2[4], dWc1[4];
  zero_acc4(dW2);
  zero_acc4(dWc1);
  float db2_a[4] = {0.f, 0.f, 0.f, 0.f}, dbc1_a[4] = {0.f, 0.f, 0.f, 0.f}, dwc2_a[4] = {0.f, 0.f, 0.f, 0.f};
  f32x4 dWra = f32x4{0.f, 0.f, 0.f, 0.f};

  float* bufA = sg.u.win.bufA[wave];
  float* bufB = sg.u.win.bufB[wave];
  const int voff_tile = (4 * q * H + r) * 4;
  const int voff_tile4 = (4 * q * H + 4 * r) * 4;
  const rsrc_t rs_geo = make_rsrc(geos);
  const rsrc_t rs_ghn = make_rsrc_n(g_hn, N * H * 4);
  const rsrc_t rs_gx = make_rsrc(gxsrc), rs_ea = make_rsrc(ea);

  for (int tk = 0; tk < ntk; ++tk) {
    STAMPB8(0);
    const int tile = vb + tk * G;
    const bool active = tile < num_tiles;                       // (wave-uniform; false only for group 1 in the workgroup's last round)
    const int v0 = active ? tile * NVB : 0;
    const int nv = active ? min(NVB, N - v0) : 0;
    const int* rp = (tk < P8_RP_TILES) ? sg.rp_tab[tk] : sg.rp;
    if (tk >= P8_RP_TILES) {      // (workgroup-uniform) slices past the table are staged on the fly, by both groups
      __syncthreads();
      if (gtid <= NVB) sg.rp[gtid] = active ? rowptr[v0 + min(gtid, nv)] : 0;
      __syncthreads();
    }
    const int e_begin = __builtin_amdgcn_readfirstlane(active ? rp[0] : 0), e_end = __builtin_amdgcn_readfirstlane(active ? rp[nv] : 0);
    // the OTHER group's edge count of this round: both groups run the larger number of windows
    int nwin = (e_end - e_begin + WB16 * TE16 - 1) / (WB16 * TE16);
    {
      const int to = tile + (grp == 0 ? (int)gridDim.x : -(int)gridDim.x);
      if (to < num_tiles) {
        const int* rpo = (tk < P8_RP_TILES) ? sm.g[grp ^ 1].rp_tab[tk] : sm.g[grp ^ 1].rp;
        const int nvo = min(NVB, N - to * NVB);
        const int cnt_o = __builtin_amdgcn_readfirstlane(rpo[nvo] - rpo[0]);
        nwin = max(nwin, (cnt_o + WB16 * TE16 - 1) / (WB16 * TE16));
      }
    }
    constexpr int XW = WB16 - 1;
    f32x4 seg_h[1], seg_x[1];
    zero_acc4(seg_h);
    zero_acc4(seg_x);

    for (int wi = 0; wi < nwin; ++wi) {
      const int win = e_begin + wi * (WB16 * TE16);
      const int cb = win + wave * TE16;
      const int nvalid = __builtin_amdgcn_readfirstlane(max(0, min(TE16, e_end - cb)));
      int vt = voff_tile;
      asm volatile("" : "+v"(vt));
      int vt4 = voff_tile4;
      asm volatile("" : "+v"(vt4));
      STAMPB8(1);
      float dy[4][4];   // SiLU'(z2), later SiLU'(z1), tile layout
      float up[4][4];   // dL/dh_neigh[dst] for this tile (prefetched)
      float m1v[4][4];  // saved z1 of this tile
      const rsrc_t rm1 = make_rsrc_n(m1s + (size_t)cb * H, nvalid * H * 4);
      float z3v[4][4], z2v[4][4];
      const rsrc_t rz2 = make_rsrc_n(z2s + (size_t)cb * H, nvalid * H * 4);
      const rsrc_t rz3 = make_rsrc_n(GX ? z3s + (size_t)cb * H : z2s, GX ? nvalid * H * 4 : 0);
      const rsrc_t rdz1 = make_rsrc_n(dZ1 + (size_t)cb * H, nvalid * H * 4);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if constexpr (GX) {
          const f32x4 v3 = buf_load4(rz3, vt4 + t * (H * 4), 0);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) z3v[t][nt] = v3[nt];
        }
        const f32x4 v2 = buf_load4(rz2, vt4 + t * (H * 4), 0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) z2v[t][nt] = v2[nt];
      }
      if (nvalid > 0) {
        // ---- S0: geometry + upstream coordinate gradient, lane = edge ----
        {
          const int l16 = lane & (TE16 - 1);
          const bool valid = l16 < nvalid;
          const int e = min(cb + l16, e_end - 1);
          const f32x4 geo = buf_load4(rs_geo, e * 16, 0);      // (x_src - x_dst, |.|^2) as the forward formed them
          int lo = 0, hi = nv;
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (rp[mid] <= e) lo = mid; else hi = mid;
          }
          const int dl = valid ? lo : 0;
          const int v = v0 + dl;
          float gx0 = 0.0f, gx1 = 0.0f, gx2 = 0.0f;
          if constexpr (GX) buf_load3(rs_gx, v * 12, 0, gx0, gx1, gx2);
          float av[FE_MAX];
#pragma unroll
          for (int f = 0; f < FE_MAX; ++f) av[f] = (f < Fe) ? buf_load(rs_ea, (e * Fe + f) * 4, 0) : 0.0f;
          float d0 = geo[0], d1 = geo[1], d2 = geo[2];
          float rad = geo[3];
          float rr = sqrtf(rad);
          float inv = 1.0f / (rr + 1e-30f);
          const float invdeg = 1.0f / (float)max(rp[dl + 1] - rp[dl], 1);
          float g0 = gx0 * invdeg, g1 = gx1 * invdeg, g2 = gx2 * invdeg;
          if (!valid) { d0 = d1 = d2 = rad = rr = inv = g0 = g1 = g2 = 0.0f; }
          if (lane < TE16) {
            sg.e_dl[wave][lane] = dl;
            sg.e_ra[wave][lane * RA_LD] = rad;
            sg.e_r[wave][lane] = rr;
            sg.e_inv[wave][lane] = inv;
            sg.e_d[wave][0][lane] = d0; sg.e_d[wave][1][lane] = d1; sg.e_d[wave][2][lane] = d2;
            if constexpr (GX) {
              sg.e_gx[wave][0][lane] = g0; sg.e_gx[wave][1][lane] = g1; sg.e_gx[wave][2][lane] = g2;
              sg.e_gxd[wave][lane] = (g0 * d0 + g1 * d1 + g2 * d2) * inv;
            }
#pragma unroll
            for (int f = 0; f < FE_MAX; ++f) sg.e_ra[wave][lane * RA_LD + 1 + f] = valid ? av[f] : 0.0f;
          }
        }
        __builtin_amdgcn_wave_barrier();
        STAMPB8(2);
        // prefetch dL/dh_neigh rows of this tile's destinations: consumed after WG1 + MM3
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = tile16_row(t, q);
          const int vo = (row < nvalid) ? ((v0 + sg.e_dl[wave][row]) * H + r) * 4 : BUF_OOB;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) up[t][nt] = buf_load(rs_ghn, vo + nt * 64, 0);      // written by the node phase
        }

        // ---- E3: coord-MLP tail backward; dz3 -> bufA, mh -> bufB, SiLU'(z2) -> registers ----
        if constexpr (GX) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int row = tile16_row(t, q);
            float tt[4], sp[4];
            float part = 0.0f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
              silu_fg(z3v[t][nt], tt[nt], sp[nt]);
              part += tt[nt] * wc2_c[nt];
              float mh;
              silu_fg(z2v[t][nt], mh, dy[t][nt]);
              bufB[row * LD + nt * 16 + r] = mh;      // rows past nvalid: SiLU(0) = 0
            }
            part = sum_over_r16(part);
            if (r == 0) sg.e_s[wave][row] = part;
            const float ds = sg.e_gxd[wave][row];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
              const float dz3 = ds * wc2_c[nt] * sp[nt];
              dwc2_a[nt] += ds * tt[nt];
              dbc1_a[nt] += dz3;
              bufA[row * LD + nt * 16 + r] = dz3;
            }
          }
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
              float mh;
              silu_fg(z2v[t][nt], mh, dy[t][nt]);
            }
        }
        // prefetch the saved first pre-activation (tile layout, rows past nvalid read as 0): in flight during WG1 + MM3
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f32x4 v1 = buf_load4(rm1, vt4 + t * (H * 4), 0);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) m1v[t][nt] = v1[nt];
        }
      }
      if constexpr (GX) {
        STAMPB8(3);
        __syncthreads();   // every wave's dz3 / mh tiles are staged
        STAMPB8(4);
        // ---- WG1: dWc1[16w.., :] += sum over the window's edge tiles of dz3^T mh ----
#pragma unroll
        for (int wt = 0; wt < WB16; ++wt)
#ifdef IS_ABL_NOMFMA
          if (win + wt * TE16 < e_end) dWc1[0][0] += sg.u.win.bufA[wt][lane] * sg.u.win.bufB[wt][lane];
#else
          if (win + wt * TE16 < e_end) mm16_outer_rows(dWc1, sg.u.win.bufA[wt], sg.u.win.bufB[wt], wave, lane);
#endif
      }

      if (nvalid > 0) {
        // ---- MM3: dmh = dz3 Wc1 + g_hn[dst] ; dz2 = dmh * SiLU'(z2) ----
        f32x4 acc[4];
        zero_acc4(acc);
#ifdef IS_ABL_NOMFMA
        if constexpr (GX) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[nt] = *reinterpret_cast<const f32x4*>(bufA + r * LD + 4 * q + 16 * nt);
        }
#else
        if constexpr (GX) mm16_rows<4, H>(acc, bufA, sm.wc1t, lane);
#endif
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const float dz2 = (acc[nt][t] + up[t][nt]) * dy[t][nt];      // rows past nvalid: (0 + 0) * SiLU'(0)
            db2_a[nt] += dz2;
            dy[t][nt] = dz2;   // parked in registers until every wave has finished reading bufA / bufB
          }
        }
      }
      if constexpr (GX) {
        STAMPB8(5);
        __syncthreads();   // WG1 + MM3 reads of bufA / bufB are complete in all waves
        STAMPB8(6);
      }

      if (nvalid > 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) bufA[tile16_row(t, q) * LD + nt * 16 + r] = dy[t][nt];   // dz2
        // ---- E1: m1 = SiLU(z1) -> bufB, SiLU'(z1) -> registers, from the z1 read back (rows past nvalid: z1 = 0 -> m1 = 0) ----
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            float y;
            silu_fg(m1v[t][nt], y, dy[t][nt]);
            bufB[tile16_row(t, q) * LD + nt * 16 + r] = y;
          }
      }
      STAMPB8(7);
      __syncthreads();   // every wave's dz2 / m1 tiles are staged
      STAMPB8(8);

      // ---- WG2: dW2[16w.., :] += sum over edge tiles of dz2^T m1 ----
#pragma unroll
      for (int wt = 0; wt < WB16; ++wt)
#ifdef IS_ABL_NOMFMA
        if (win + wt * TE16 < e_end) dW2[0][0] += sg.u.win.bufA[wt][lane] * sg.u.win.bufB[wt][lane];
#else
        if (win + wt * TE16 < e_end) mm16_outer_rows(dW2, sg.u.win.bufA[wt], sg.u.win.bufB[wt], wave, lane);
#endif

      if (nvalid > 0) {
        // ---- MM4: dm1 = dz2 W2 ; dz1 = dm1 * SiLU'(z1) ----
        f32x4 acc[4];
        zero_acc4(acc);
#ifdef IS_ABL_NOMFMA
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = *reinterpret_cast<const f32x4*>(bufA + r * LD + 4 * q + 16 * nt);
#else
        mm16_rows<4, H>(acc, bufA, sm.w2t, lane);
#endif
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = tile16_row(t, q);
          float part = 0.0f;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const float dz1 = acc[nt][t] * dy[t][nt];      // rows past nvalid: 0 (dz2 = 0)
#ifndef IS_ABL_STORES
            buf_store(dz1, rdz1, vt + (t * H + nt * 16) * 4, 0);      // (rows past nvalid: dropped)
#endif
            dy[t][nt] = dz1;
            part += dz1 * wr_t[nt];
          }
          part = sum_over_r16(part);
          if (r == 0) sg.e_drad[wave][row] = part;
        }
      }
      STAMPB8(9);
      __syncthreads();   // WG2 + MM4 reads complete in all waves
      STAMPB8(10);

      if (nvalid > 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) bufA[tile16_row(t, q) * LD + nt * 16 + r] = dy[t][nt];   // dz1

        // ---- GEO: gradient wrt d = x_src - x_dst, lane = edge ----
        if (lane < TE16) {
          const bool valid = lane < nvalid;
          float q0 = 0.f, q1 = 0.f, q2 = 0.f;
          if (valid) {
            const float inv = sg.e_inv[wave][lane], rr = sg.e_r[wave][lane];
            const float d0 = sg.e_d[wave][0][lane], d1 = sg.e_d[wave][1][lane], d2 = sg.e_d[wave][2][lane];
            float u0 = 0.0f, u1 = 0.0f, u2 = 0.0f;
            if constexpr (GX) {
              const float s = sg.e_s[wave][lane];
              u0 = s * sg.e_gx[wave][0][lane]; u1 = s * sg.e_gx[wave][1][lane]; u2 = s * sg.e_gx[wave][2][lane];
            }
            const float ddot = d0 * u0 + d1 * u1 + d2 * u2;
            const float k = rr > 0.0f ? ddot * inv * inv / rr : 0.0f;
            const float dr2 = 2.0f * sg.e_drad[wave][lane];
            q0 = u0 * inv - d0 * k + d0 * dr2;
            q1 = u1 * inv - d1 * k + d1 * dr2;
            q2 = u2 * inv - d2 * k + d2 * dr2;
            const size_t e = (size_t)(cb + lane);
#ifndef IS_ABL_STORES
            dD[e * 3 + 0] = q0; dD[e * 3 + 1] = q1; dD[e * 3 + 2] = q2;
#endif
          }
          sg.e_gx[wave][0][lane] = q0; sg.e_gx[wave][1][lane] = q1; sg.e_gx[wave][2][lane] = q2;
        }
      }
      __syncthreads();

      STAMPB8(11);
      // ---- WG3: [dw_r | dW_a][16w.., :] += sum over the window's edge tiles of dz1^T [radial | a] ----
#pragma unroll
      for (int wt = 0; wt < WB16; ++wt)
        if (win + wt * TE16 < e_end) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int e = 4 * q + s;
            const float b = sg.e_ra[wt][e * RA_LD + min(r, FE_MAX)];     // columns past 1 + FE_MAX are zero
            dWra = __builtin_amdgcn_mfma_f32_16x16x4f32(sg.u.win.bufA[wt][e * LD + wave * 16 + r], r <= FE_MAX ? b : 0.0f, dWra, 0, 0, 0);
          }
        }
      // ---- SEG: dPd[v] += sum over the window's in-edges of v of dz1, dx[v] -= sum of dL/dd -- MFMA products with the 0 / 1
      //      incidence of the tile (exact; fixed order).  Wave w owns columns [16 w, 16 w + 16); coordinates ride on the last wave.
#ifndef IS_ABL_SEG
#pragma unroll
      for (int wt = 0; wt < WB16; ++wt)
        if (win + wt * TE16 < e_end) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int e = 4 * q + s;
            const int dl = sg.e_dl[wt][e];
            const float bh = sg.u.win.bufA[wt][e * LD + wave * 16 + r];
            float bx = 0.0f;
            if (wave == XW) bx = sg.e_gx[wt][min(r, 2)][e];
            const float ind = (dl == r) ? 1.0f : 0.0f;
            seg_h[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ind, bh, seg_h[0], 0, 0, 0);
            if (wave == XW) seg_x[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ind, r < 3 ? bx : 0.0f, seg_x[0], 0, 0, 0);
          }
        }
#else
      seg_h[0][0] += sg.u.win.bufA[0][lane]; seg_x[0][0] += sg.e_gx[0][0][r];
#endif
      __syncthreads();
    }

    STAMPB8(12);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int nl = tile16_row(t, q);
      if (nl < nv) {
        const int v = v0 + nl;
        dPd[(size_t)v * ld_dpd + wave * 16 + r] = seg_h[0][t];
        if (wave == XW && r < 3) dx[v * 3 + r] = (GX ? gxsrc[v * 3 + r] : 0.0f) - seg_x[0][t];
      }
    }
  }

  // ---- the workgroup's ONE partial record: both groups' accumulators through LDS, added group 0 + group 1, coalesced stores ----
  STAMPP8(18);
  __syncthreads();      // every window read of the group buffers is complete (the record overlays them)
  {
    float* rec = sg.u.out.rec;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int o = wave * 16 + tile16_row(t, q), i = nt * 16 + r;
        rec[o * H + i] = dW2[nt][t];
        rec[H * H + o * H + i] = dWc1[nt][t];
      }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int o = wave * 16 + tile16_row(t, q);
      if (r == 0) rec[2 * H * H + 3 * H + o] = dWra[t];
      else if (r <= 8) rec[2 * H * H + 4 * H + o * 8 + (r - 1)] = dWra[t];
    }
    // per-wave bias sums ([wave][slot][64]), summed over the 4 waves in wave order below
    float* vec = sg.u.out.vec;
    constexpr int SLOTS = 3;
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
    __syncthreads();
    if (gtid < SLOTS * H) {
      const int sidx = gtid / H, c = gtid % H;
      float v = 0.0f;
      for (int w = 0; w < WB16; ++w) v += vec[(w * SLOTS + sidx) * H + c];
      rec[2 * H * H + sidx * H + c] = v;
    }
    __syncthreads();
    float* part = partials + (size_t)blockIdx.x * P8_RECORD;
    const float* r0 = sm.g[0].u.out.rec;
    const float* r1 = sm.g[1].u.out.rec;
    for (int i = tid; i < P8_RECORD / 4; i += 512) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(r0 + 4 * i), b = *reinterpret_cast<const f32x4*>(r1 + 4 * i);
      *reinterpret_cast<f32x4*>(part + 4 * i) = a + b;
    }
  }
  STAMPP8(19);
  wg_clock_end(wg_clock);
}
#endif  // IS_BWD8_BUILT

}  // namespace is

#ifdef IS_STAGE_STAMPS
extern "C" int is_debug_stamps_bwd8(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(is::g_stamps_b8), sizeof(long long) * 24) == hipSuccess ? 0 : is::fail(__func__, -5);
}
#endif

// 1: is_egnn_layer_bwd_paired covers a launch with Fe edge features in this build
extern "C" int is_egnn_layer_bwd_paired_supported(int Fe) {
  return (IS_BWD8_BUILT && Fe >= 0 && Fe <= 1) ? 1 : 0;
}

// is_egnn_layer_bwd on `grid` workgroups of 512 threads, each two groups of four waves that share one staged copy of the weight
// tiles and write ONE partial record (so `partials` holds `grid` records; the virtual grid of 256-thread workgroups is 2 * grid).
// Same arguments and outputs as is_egnn_layer_bwd; Fe <= 1, z3s != NULL whenever a coordinate gradient
// arrives (-38 where is_egnn_layer_bwd_paired_supported says 0).
extern "C" int is_egnn_layer_bwd_paired(const float* ps, const float* pd, int ld_p, const float* x, const float* ea,
                                        const int32_t* rowptr, const int32_t* srcs, const float* W1, int ldw, int din,
                                        const float* W2, const float* Wc1, const float* bc1, const float* wc2, const float* z2s,
                                        const float* z3s, const float* g_xout, float* dZ1, float* dD, float* dPd, int ld_dpd,
                                        float* dx, float* partials, int grid, int N, int Fe,
                                        const float* dZ1n, const float* dDn, const float* dxn, const int32_t* rowptr_src,
                                        const int32_t* pos_by_src, const float* g_h, float* g_psd, const float* zn1,
                                        const float* bpack, float* dh_total, float* dzn1, float* d_h, float* d_hn, float* gxtot,
                                        long long* wg_clock, const float* m1s, const float* dy1s, const float* geos, void* stream) {
  (void)ps; (void)pd; (void)x; (void)srcs; (void)bc1; (void)dy1s;      // (inputs of the recompute forms of is_egnn_layer_bwd)
#if !IS_BWD8_BUILT
  return is::fail(__func__, -38);
#else
  if (N <= 0) return 0;
  if (Fe < 0 || Fe > 1) return is::fail(__func__, -38);
  if ((long long)N * is::H * 4 >= 0x7ffff000LL) return is::fail(__func__, -22);
  if (m1s == nullptr || geos == nullptr) return is::fail(__func__, -22);
  const bool gather = dZ1n != nullptr;
  const bool gx = gather || g_xout != nullptr;
  if (gx && z3s == nullptr) return is::fail(__func__, -38);      // the z3-recompute form lives in egnn_layer_bwd_z3r.hip
  if (grid <= 0 || (din != 20 && din != 64) || zn1 == nullptr || bpack == nullptr || dzn1 == nullptr ||
      d_hn == nullptr || (g_psd != nullptr && dh_total == nullptr) || (g_psd == nullptr && g_h == nullptr) ||
      (gather && (g_xout != nullptr || dDn == nullptr || dxn == nullptr || rowptr_src == nullptr || pos_by_src == nullptr ||
                  g_psd == nullptr || gxtot == nullptr)) ||
      ld_p != 2 * is::H || ld_dpd != 2 * is::H || ldw != 2 * din + 1 + Fe)
    return is::fail(__func__, -22);
  if (grid > (N + 15) / 16) return is::fail(__func__, -22);      // every workgroup owns at least one tile
  hipStream_t st = static_cast<hipStream_t>(stream);
  const is::NodeBwd8Args nb{dZ1n, dDn, dxn, rowptr_src, pos_by_src, g_h, g_psd, zn1, bpack, dh_total, dzn1, d_h, d_hn, gxtot};
  const dim3 block(512);
#define IS_LAUNCH_B8(GXF, GA, DI)                                                                                            \
  hipLaunchKernelGGL((is::egnn_layer_bwd8_kernel<1, GXF, GA, DI>), dim3(grid), block, 0, st, ea, rowptr, W1, W2, Wc1, wc2, z2s, \
                     z3s, g_xout, dZ1, dD, dPd, dx, partials, N, Fe, nb, wg_clock, m1s, geos)
#define IS_LAUNCH_B8_D(GXF, GA) do { if (din == 20) IS_LAUNCH_B8(GXF, GA, 20); else IS_LAUNCH_B8(GXF, GA, 64); } while (0)
  if (gather) IS_LAUNCH_B8_D(true, true);
  else if (gx) IS_LAUNCH_B8_D(true, false);
  else IS_LAUNCH_B8_D(false, false);
#undef IS_LAUNCH_B8_D
#undef IS_LAUNCH_B8
  return is::launch_status(__func__);
#endif
}
