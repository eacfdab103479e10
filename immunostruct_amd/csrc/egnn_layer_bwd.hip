// One EGNNConv layer, backward, as ONE launch per layer: for each tile of destination nodes a workgroup runs
//   P1  (GATHER) the source-side "scatter-add" of the layer ABOVE as a gather over the CSR-by-source index:
//       dPs[v] = sum_{out-edges of v} dZ1'[slot], g_x[v] = dx'_dst[v] + sum dD'[slot] -- for its OWN nodes only;
//   P2  the node data path of this layer for those nodes: dh = g_h + g_psd W1sd' ; dzn1 = (dh Wn2) * SiLU'(zn1) ;
//       [d_h | d_hneigh] = dzn1 Wn1  (MFMA, B operands in registers from the lane-ordered operand pack, node16.h);
//   P3  the fused edge pass backward over the in-edges of those nodes (what egnn_edge_bwd16 was): recompute z1 from
//       the gathers, edge / coordinate-MLP backward, geometry backward, dPd / dx by destination, per-edge dZ1 / dD
//       for the gather of the layer BELOW, weight-gradient outer products distributed by output tile.
// Everything P1 / P2 need belongs to the tile's own nodes, so no workgroup ever waits for another one; the activation
// tiles of P2 live in the LDS buffers P3 uses later.  This replaces three launches per layer (node data kernel, edge
// kernel, source gather) and their round trips of dh / d_hneigh / dpsd through HBM.
//
// P3 mapping (unchanged):
//   * workgroups of 4 waves, one 16-edge tile per wave on v_mfma_f32_16x16x4_f32, two
//     16x64 LDS buffers per wave => ~78 KB LDS, TWO independent workgroups per CU (2 waves per
//     SIMD): one workgroup's gathers / SiLU epilogues overlap the other's MFMAs;
//   * the weight-gradient outer products are distributed by OUTPUT tile instead of by edge
//     tile: wave w owns rows [16w, 16w+16) of dW2 / dWc1 / [dw_r | dW_a] and contracts over all four edge
//     tiles of the 64-edge window, so a wave carries 3 x 16 accumulator registers and no cross-wave reduction
//     is needed at the end;
//   * a workgroup owns one node tile per pass: NV16 = 16 consecutive destination nodes (~48 edges on degree-3 graphs: the fourth
//     wave of the 64-edge window idles in the row phases).  (Rounds 2 - 5 also took a greedy tile list -- <= 64 in-edges, <= 24 nodes:
//     full windows, 24 % fewer passes; round 6's sweep has plain tiles on the paired kernel, egnn_layer_bwd8.hip, ahead at every
//     batch size and density, and the list went: HISTORY.md 9.6.)
// This file is compiled TWICE: as itself (z3 read back from HBM: the default) and, through egnn_layer_bwd_z3r.hip, with
// IS_BWD_Z3R = 1 (z3 recomputed per tile: IMMUNOSTRUCT_SAVE_Z3=0).  A preprocessor switch, not a template parameter: with both
// forms in one kernel template the discarded `if constexpr` branch still changed the register allocation of the default
// instantiations (4 -> 12 spilled registers) -- the default build must not see the other form.
#ifndef IS_BWD_Z3R
#define IS_BWD_Z3R 0
#endif
// IS_LAYER_M1 (build-time, Makefile M1=0|1|2): 1 = the first edge-MLP activation m1 = SiLU(z1) and SiLU'(z1) are READ BACK (two
// [E, 64] arrays the forward saves), 2 = z1 is read back (one array; the SiLU pair is evaluated here), 0 = z1 is recomputed from
// gathered Ps rows + the Pd tile + geometry.  The windows of this
// kernel are issue-bound (HISTORY.md: 64 extra MFMAs cost exactly their issue time), HBM runs at a third of its rate: the
// recompute's 16 row gathers by source id, the Pd tile, 16 x (3 LDS reads + 3 FMAs + LDS write) and a wave barrier per wave and
// window are traded for 16 (form 2) or 32 (form 1) coalesced loads.  Measured at B = 128: backward launch 83.2 (0) / 77.2 (1) /
// 75.0 us (2), forward 48.8 / 52.2 / 50.2 us (its extra stores), step 1.155 / 1.152 / 1.12 ms: form 2 is the default.
// is_layer_saves_m1() tells the host which form the library was built as.
#ifndef IS_LAYER_M1
#define IS_LAYER_M1 2
#endif
// IS_LAYER_GEO (build-time, Makefile GEO=0|1; needs M1 != 0): 1 = the edge geometry (x_src - x_dst, |.|^2) is READ BACK from a
// [E, 4] array the forward saves instead of recomputed from the two endpoints' coordinates: the window's first dependent chain
// (source id -> coordinates) disappears -- with z1 read back nothing in the edge phase needs the source id any more.
#ifndef IS_LAYER_GEO
#define IS_LAYER_GEO 1
#endif
#if IS_LAYER_GEO && !IS_LAYER_M1
#error "IS_LAYER_GEO needs IS_LAYER_M1 != 0 (the z1 recompute gathers by source id)"
#endif
#if IS_BWD_Z3R
#define IS_BWD_KERNEL egnn_layer_bwd_z3r_kernel
#define IS_BWD_LAUNCHER launch_layer_bwd_z3r
#else
#define IS_BWD_KERNEL egnn_layer_bwd_kernel
#define IS_BWD_LAUNCHER launch_layer_bwd
#endif
#include "common.h"
#include "node16.h"

#ifndef IS_GATHER_GB
#define IS_GATHER_GB 4      // rows per round of the node phase's source gather
#endif

namespace is {

#ifdef IS_STAGE_STAMPS
__device__ long long g_stamps_b[24];
#define STAMP_WG (gridDim.x > 300 ? 300 : 100)
#define STAMPB(k) do { if (blockIdx.x == STAMP_WG && threadIdx.x == 0 && tile == blockIdx.x) g_stamps_b[k] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMPP(k) do { if (blockIdx.x == STAMP_WG && threadIdx.x == 0) g_stamps_b[k] = __builtin_amdgcn_s_memtime(); } while (0)
// stamp once every outstanding vector-memory access of the wave has returned (perturbs the schedule: diagnosis only)
#define STAMPP_W(k) do { __builtin_amdgcn_s_waitcnt(0x0F70); STAMPP(k); } while (0)
#else
#define STAMPB(k) do { } while (0)
#define STAMPP(k) do { } while (0)
#define STAMPP_W(k) do { } while (0)
#endif

constexpr int WB16 = 4;
constexpr int NV16 = 16;   // nodes per tile
constexpr int RP_TILES = 8;       // tiles per workgroup whose rowptr slice is fetched ahead (B = 128: 3; the stress slice: 8)

// Round 5: with plain 16-node tiles the node phase runs 48 rows per pass (three tiles: all of a workgroup's tiles at B = 128) and its
// dzn1 tile overlays the g_psd tile (dead behind the dh product's barrier) -- 38.4 KB instead of 68.6 KB, which fits BESIDE the two
// weight tiles (34.8 KB) in a workgroup's half of the CU's LDS.  The weight tiles and the rowptr slices are then staged at the very
// START of the kernel, under the node phase's front, instead of behind its last barrier (stage stamps: 6.5 - 13 k of 158 k cycles
// between the end of the node phase and the first window).  The z1-recompute build (its Pd tile) keeps the overlay of rounds 2 - 4.
#ifndef IS_BWD_EARLY_STAGE
#define IS_BWD_EARLY_STAGE 1      // (0: A/B builds with the staging of rounds 2 - 4)
#endif
template <int NVB>
struct BwdLayout {
  static constexpr bool EARLY = (NVB == 16) && (IS_LAYER_M1 != 0) && (IS_BWD_EARLY_STAGE != 0);
  static constexpr int PROWS = EARLY ? 48 : 64;
  static constexpr int LDP = 132;
};

template <int FE_MAX, int NVB>
struct alignas(16) Bwd16Smem {
  float w2t[H * LD];
  float wc1t[H * LD];
  union {
    struct { float bufA[WB16][TE16 * LD]; float bufB[WB16][TE16 * LD]; } w;
    float node[BwdLayout<NVB>::EARLY ? BwdLayout<NVB>::PROWS * (BwdLayout<NVB>::LDP + LD) : 1];      // EARLY: [g_psd | dzn1] tile, dh tile
  } u;
  float pdt[IS_LAYER_M1 ? 1 : NVB * H];   // Pd rows of this tile's destination nodes (z1 recompute only)
  int rp[NVB + 1];
  int rp_tab[RP_TILES][NVB + 1];   // rowptr slices of this workgroup's first RP_TILES tiles (fetched once, in front of the edge loop)
  int e_dl[WB16][TE16];
  float e_ra[WB16][TE16 * (FE_MAX + 1)];   // per edge: [radial | edge features]: the B operand of the dw_r / dW_a outer product
  float wa[(FE_MAX > 1 && !IS_LAYER_M1) ? FE_MAX * H : 1];   // W_a columns, lane = channel (registers when FE_MAX == 1; z1 recompute only)
  float e_r[WB16][TE16];
  float e_inv[WB16][TE16];
  float e_d[WB16][3][TE16];
  float e_gx[WB16][3][TE16];   // upstream coordinate gradient / deg; GEO overwrites it with dL/dd (each lane its own entries)
  float e_gxd[WB16][TE16];
  float e_s[WB16][TE16];
  float e_drad[WB16][TE16];
};

constexpr int PART16_STRIDE = 8448 + 64 * 8;  // identical to the v1 record

struct NodeBwdArgs {
  // ---- P1 (GATHER): per-edge gradients of the layer above, by CSR-by-source ----
  const float* dZ1n;        // [E, 64]
  const float* dDn;         // [E, 3]
  const float* dxn;         // [N, 3]   destination-side dL/dx of the layer above (identity path included)
  const int* rowptr_src;    // [N + 1]
  const int* pos_by_src;    // [E]
  // ---- P2 ----
  const float* g_h;         // [N, 64]  direct gradient of this layer's output h (may be NULL: zero)
  float* g_psd;             // [N, 128] gradient of the next pre-projection of h: GATHER reads the Pd half and WRITES the Ps
                            //          half (the weight-gradient kernel reads the whole row later); NULL: no such projection
  const float* zn1;         // [N, 64]  saved node-MLP pre-activation
  const float* bpack;       // backward operand pack of this layer
  float* dh_total;          // [N, 64]  out (only with g_psd): g_h + g_psd W1sd
  float* dzn1;              // [N, 64]  out
  float* d_h;               // [N, 64]  out, first DIN columns (may be NULL)
  float* d_hn;              // [N, 64]  out: dL/dh_neigh, read back by P3 (same workgroup)
  float* gxtot;             // [N, 3]   out (GATHER): dL/dx_out of this layer = dxn + gather(dDn), read back by P3
};

// Z3R: the coordinate MLP's pre-activation z3 = SiLU(z2) Wc1^T + bc1 is RECOMPUTED per tile (one more 16 x 64 x 64 product on
// the MFMA pipe, B operand read from the transposed weight tile that MM3 needs anyway) instead of being streamed back from
// HBM: the forward then stores one [E, 64] array per layer less and this kernel reads one less (18.5 MB each way at B = 128).
template <int FE_MAX, int NVB, bool GX, bool GATHER, int DIN>
__global__ __launch_bounds__(256, 2) void IS_BWD_KERNEL(
    const float* __restrict__ ps, const float* __restrict__ pd,
    const float* __restrict__ x, const float* __restrict__ ea,
    const int* __restrict__ rowptr, const int* __restrict__ srcs,
    const float* __restrict__ W1,
    const float* __restrict__ W2, const float* __restrict__ Wc1, const float* __restrict__ wc2,
    const float* __restrict__ z2s, const float* __restrict__ z3s_or_bc1,      // Z3R: coord_mlp.0.bias; otherwise the saved z3
    const float* __restrict__ g_xout,
    float* __restrict__ dZ1, float* __restrict__ dD,
    float* __restrict__ dPd, float* __restrict__ dx,
    float* __restrict__ partials, int N, int Fe, NodeBwdArgs nb, long long* __restrict__ wg_clock,
    const float* __restrict__ m1s, const float* __restrict__ dy1s, const float* __restrict__ geos) {
  static_assert(!GATHER || GX, "a gathered layer always receives a coordinate gradient");
  using D = Node16Dims<DIN>;
  constexpr bool EARLY = BwdLayout<NVB>::EARLY;
  constexpr int PROWS = BwdLayout<NVB>::PROWS, MT = PROWS / 16, LDP = BwdLayout<NVB>::LDP;      // node phase: rows (several tiles) per pass
  constexpr int PITCH = (NVB + 15) / 16 * 16;                // rows reserved per tile in a pass
  constexpr int TPP = PROWS / PITCH;                         // tiles per pass
  static_assert(EARLY || sizeof(float) * PROWS * (LDP + 2 * LD) <= sizeof(float) * (2 * H * LD + 2 * WB16 * TE16 * LD),
                "the node phase's tiles must fit the (not yet staged) weight tiles + window buffers");
  static_assert(sizeof(Bwd16Smem<FE_MAX, NVB>) <= 80 * 1024, "two workgroups per CU");
  __shared__ Bwd16Smem<FE_MAX, NVB> sm;
  wg_clock_start(wg_clock);
  // (kernel arguments live in scalar registers for the whole launch and this kernel has ~45 of them: what can be derived is --
  //  every caller's pre-projection / gradient rows are 128 wide, din is the DIN instantiation, ldw follows from it, and the
  //  z3 recompute's bias shares the slot of the array it replaces.  Two more pointers once cost the listed-tile form 6 us.)
  constexpr bool Z3R = IS_BWD_Z3R != 0;
  [[maybe_unused]] constexpr int ld_p = 2 * H;
  constexpr int ld_dpd = 2 * H, din = DIN;
  const int ldw = 2 * DIN + 1 + Fe;
  const float* __restrict__ z3s = Z3R ? nullptr : z3s_or_bc1;
  const float* __restrict__ bc1 = Z3R ? z3s_or_bc1 : nullptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform => scalar registers, scalar address math
  const int r = lane & 15, q = lane >> 4;
  constexpr int RA_LD = FE_MAX + 1;
  const int num_tiles = (N + NV16 - 1) / NV16;
  const float* __restrict__ gxsrc = GATHER ? nb.gxtot : g_xout;

  STAMPP(13);
  // ================= P1 + P2: source gather and node data path of ALL tiles of this workgroup =================
  // (before the persistent edge loop: its weight-gradient accumulators do not exist yet, so the 80 operand registers
  //  of the node phase cost nothing, they are fetched once per workgroup, and the gathers of several tiles overlap)
  // the staging of everything the edge loop reads from LDS without knowing a gradient: the two weight tiles (transposed), the
  // rowptr slices of this workgroup's tiles.  EARLY: here, before the node phase (one L2 round trip: every workgroup reads the same
  // 32 KB), published by the node phase's barriers; otherwise behind the node phase, whose tiles overlay the weight tiles
  auto stage_edge_tables = [&]() {
    for (int idx = tid; idx < RP_TILES * (NVB + 1); idx += 256) {
      const int k = idx / (NVB + 1), i = idx - k * (NVB + 1);
      const int t = blockIdx.x + k * gridDim.x;
      if (t < num_tiles) {
        const int a0 = t * NV16;
        const int cnt = min(NV16, N - a0);
        sm.rp_tab[k][i] = rowptr[a0 + min(i, cnt)];
      }
    }
    load_matrix_lds_t(sm.w2t, W2, tid, 256);
    if constexpr (GX) load_matrix_lds_t(sm.wc1t, Wc1, tid, 256);
  };
  if constexpr (EARLY) stage_edge_tables();
  {
    float* ps_ = EARLY ? &sm.u.node[0] : &sm.w2t[0];      // [PROWS][LDP]  g_psd rows     (not EARLY: w2t | wc1t | bufA | bufB are contiguous)
    float* gs = ps_ + PROWS * LDP;                         // [PROWS][LD]   dh
    float* zs = EARLY ? ps_ : gs + PROWS * LD;             // [PROWS][LD]   dzn1 (EARLY: over the g_psd rows, dead behind the dh product's barrier)
    const bool has_psd = nb.g_psd != nullptr;
    const int col = wave * 16 + r;
    // transposed-weight operands of this wave's output columns (operand pack: coalesced 16-byte loads, L2).  Fetched per
    // pass, each part where its latency is covered and its registers are free: bp behind the first gather level, ba / bx
    // behind the gather (they are first used two barriers later)
    const f32x4* pk = reinterpret_cast<const f32x4*>(nb.bpack) + (size_t)wave * NODE_BWD_SLOTS * 64 + lane;
    STAMPP_W(20);
#ifdef IS_ABL_NONODE
    for (int t0 = num_tiles; t0 < num_tiles; t0 += TPP * gridDim.x) {
#else
    for (int t0 = blockIdx.x; t0 < num_tiles; t0 += TPP * gridDim.x) {
#endif
      const int ntp = min(TPP, (num_tiles - t0 + (int)gridDim.x - 1) / (int)gridDim.x);     // tiles of this pass
      const int mt_used = ntp * (PITCH / 16);
      // first node / node count of the pass's tiles: workgroup-uniform, so they are scalar loads into scalar registers (they
      // went through LDS behind a barrier, and every row_node() below was two LDS reads)
      int ta0[TPP], tcnt[TPP];
#pragma unroll
      for (int k = 0; k < TPP; ++k) {
        const int tl = t0 + k * (int)gridDim.x;
        int a0 = 0, cnt = 0;
        if (k < ntp) {
          a0 = tl * NV16;
          cnt = min(NV16, N - a0);
        }
        ta0[k] = __builtin_amdgcn_readfirstlane(a0);
        tcnt[k] = __builtin_amdgcn_readfirstlane(cnt);
      }
      // node of pass row lr (or -1): tile k = lr / PITCH, i = lr % PITCH
      auto row_node = [&](int lr) {
        const int k = lr / PITCH, i = lr % PITCH;
        int a0 = ta0[0], cnt = tcnt[0];
#pragma unroll
        for (int kk = 1; kk < TPP; ++kk) { a0 = (k == kk) ? ta0[kk] : a0; cnt = (k == kk) ? tcnt[kk] : cnt; }
        return (i < cnt) ? a0 + i : -1;
      };
      // the pass's front is a chain of dependent loads (the whole chip starts it at the same moment: ~1.2 us per level), so the
      // order of issue is: first level of the gather, then everything independent of it, then the dependent levels
      [[maybe_unused]] const int g_sub = tid & 3;       // gather role: 4 lanes per node, row g_lr of the pass
      [[maybe_unused]] const int g_lr = tid >> 2;
      [[maybe_unused]] int g_v = -1, g_lo = 0, g_hi = 0;
      if constexpr (GATHER) {
        g_v = (g_lr < mt_used * 16) ? row_node(g_lr) : -1;
        const int vc = max(g_v, 0);
        g_lo = nb.rowptr_src[vc];
        g_hi = nb.rowptr_src[vc + 1];
      }
      // epilogue inputs of this lane (rows mt*16 + 4q + t, column col): consumed two / three stages later
      float zpre[MT][4], gpre[MT][4];
      int vrow[MT][4];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int v = row_node(mt * 16 + tile16_row(t, q));
          vrow[mt][t] = v;
          const int vc = max(v, 0);
          zpre[mt][t] = nb.zn1[(size_t)vc * H + col];
          gpre[mt][t] = (has_psd && nb.g_h != nullptr) ? nb.g_h[(size_t)vc * H + col] : 0.0f;
        }
      float bp[32];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const f32x4 v = pk[g * 64];      // (unused without a projection; the pack slots exist either way)
#pragma unroll
        for (int j = 0; j < 4; ++j) bp[4 * g + j] = v[j];
      }
      STAMPP_W(21);
      if constexpr (GATHER) {
        // lane `sub` holds columns 16 j + 4 sub .. + 3 (j = 0..3) of its node's row, so that each load instruction reads row
        // segment j as ONE contiguous 64 bytes over the node's four lanes.  The slot ids of the first 8 out-edges are one level,
        // their rows follow 4 at a time (a round's 16 row loads are in flight together; two rows per round and the ids inside
        // the round made a node with seven out-edges eight dependent latencies, and the workgroup waits for its slowest lane).
        // Ids / rows past the node's last edge are out-of-range buffer reads: no traffic, +0.0 -- the per-element summation
        // order is is_gather_segment_sum's
        const int sub = g_sub, lr = g_lr, v = g_v;
        const int lo = g_lo, hi = (v >= 0) ? g_hi : g_lo;
        constexpr int VIEW = 0x7fffe000;      // "everything below BUF_OOB" (the entry point bounds E accordingly)
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
        STAMPP_W(22);
        auto round = [&](const int* ek, int p0) {      // rows p0 .. p0 + GB - 1 of the node's list
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
        STAMPP_W(23);
        if (v >= 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j)      // Ps half, for the weight-gradient kernel
            *reinterpret_cast<f32x4*>(nb.g_psd + (size_t)v * 128 + 16 * j + 4 * sub) = acc[j];
          if (sub < 3) nb.gxtot[v * 3 + sub] = x_dst + acc3;
        }
        if (PROWS == 64 || lr < PROWS) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<f32x4*>(ps_ + lr * LDP + 16 * j + 4 * sub) = acc[j];
            *reinterpret_cast<f32x4*>(ps_ + lr * LDP + 64 + 16 * j + 4 * sub) = (v >= 0) ? pdv[j] : f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      } else {
        // stage g_psd (or g_h) rows of the pass: all loads first, LDS stores afterwards
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
      STAMPP(14);
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
      STAMPP(15);
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
      STAMPP(16);
      // [d_h | d_hneigh] = dzn1 Wn1: wave w produces columns [32w, 32w + 32) of the (DIN + 64)-wide row
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int xc = (wave * 2 + nt) * 16 + r;
        if ((wave * 2 + nt) * 16 < D::KV) {          // wave-uniform: the column tile exists
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
      STAMPP(17);
    }
  }

#ifdef IS_ABL_NONODE
  __syncthreads();      // (timing-only build without a node phase: its barriers are what publishes the early staging)
#endif
  // P3 reads the node phase's d_hn rows (written above by this workgroup, behind a barrier) through a restrict pointer:
  // without it every prefetch of the edge loop is ordered against every store of the loop
  const float* __restrict__ g_hn = nb.d_hn;
  // the rowptr slices of this workgroup's tiles: every tile of the persistent loop is known now, so its slice is fetched here,
  // under the weight staging, instead of at the tile's start (there it was a dependent load + two barriers per tile)
  if constexpr (!EARLY) stage_edge_tables();

#if !IS_LAYER_M1
  const float wr_c = W1[lane * ldw + 2 * din];
  // W_a: one register when there is a single edge feature, LDS otherwise (eight more live registers spill this kernel)
  float wa_c = 0.0f;
  if constexpr (FE_MAX == 1) {
    wa_c = (Fe > 0) ? W1[lane * ldw + 2 * din + 1] : 0.0f;
  } else {
    for (int idx = tid; idx < FE_MAX * H; idx += 256) {
      const int f = idx / H, c = idx % H;
      sm.wa[idx] = (f < Fe) ? W1[c * ldw + 2 * din + 1 + f] : 0.0f;
    }
  }
#endif
  float wc2_c[4], wr_t[4], bc1_c[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    wc2_c[nt] = GX ? wc2[nt * 16 + r] : 0.0f;
    bc1_c[nt] = (GX && Z3R) ? bc1[nt * 16 + r] : 0.0f;
    wr_t[nt] = W1[(nt * 16 + r) * ldw + 2 * din];
  }

  // wave `wave` owns output rows [16*wave, 16*wave+16) of both weight gradients
  f32x4 dW2[4], dWc1[4];
  zero_acc4(dW2);
  zero_acc4(dWc1);
  float db2_a[4] = {0.f, 0.f, 0.f, 0.f}, dbc1_a[4] = {0.f, 0.f, 0.f, 0.f}, dwc2_a[4] = {0.f, 0.f, 0.f, 0.f};
  // dw_r / dW_a = sum_e dz1[e][c] * [radial | a][e][j] as a third outer product, distributed by output tile like dW2:
  // rows [16w, 16w+16) = channels, column 0 = radial weight, columns 1 .. Fe = edge-feature weights
  f32x4 dWra = f32x4{0.f, 0.f, 0.f, 0.f};

  float* bufA = sm.u.w.bufA[wave];
  float* bufB = sm.u.w.bufB[wave];
  // raw-buffer views: scalar bases, one lane-constant byte offset per access pattern, range-checked where rows past a
  // tile's end must read as zero / must not be written (no per-element predication, no 64-bit vector address math)
  const int voff_tile = (4 * q * H + r) * 4;            // element (row 4q [+ t], column r [+ 16 nt]) of a [16][64] tile
  // the saved pre-activation arrays (z1, z2, z3) keep channel 16 nt + r at position 4 r + nt of its row (is_egnn_layer_fwd): a
  // lane's four channels of a row are 16 contiguous bytes
  const int voff_tile4 = (4 * q * H + 4 * r) * 4;
#if !IS_LAYER_M1
  const int ld_p_bytes = ld_p * 4;
  const rsrc_t rs_ps = make_rsrc(ps), rs_pd = make_rsrc(pd);
#endif
#if IS_LAYER_GEO
  const rsrc_t rs_geo = make_rsrc(geos);
#else
  const rsrc_t rs_x = make_rsrc(x), rs_srcs = make_rsrc(srcs);
#endif
  const rsrc_t rs_ghn = make_rsrc_n(g_hn, N * H * 4);
  const rsrc_t rs_gx = make_rsrc(gxsrc), rs_ea = make_rsrc(ea);

  for (int tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    STAMPB(0);
    const int v0 = tile * NV16;
    const int nv = min(NV16, N - v0);
    const int tk = (tile - (int)blockIdx.x) / (int)gridDim.x;      // (wave-uniform) index of this tile in the workgroup's sequence
    const int* rp = (tk < RP_TILES) ? sm.rp_tab[tk] : sm.rp;
    // first tile: the tables and weight tiles are staged; later tiles need no barrier here (every window ends with one, and a
    // tile's rowptr slice has its own table row) unless a slice is staged on the fly
    if ((tk == 0 && !EARLY) || tk >= RP_TILES || !IS_LAYER_M1) __syncthreads();      // (EARLY: the node phase's barriers published them)
    if (tk >= RP_TILES && tid <= NVB) sm.rp[tid] = rowptr[v0 + min(tid, nv)];
#if !IS_LAYER_M1
#pragma unroll
    for (int i = 0; i < NVB / WB16; ++i) {
      const int nl = wave * (NVB / WB16) + i;
      sm.pdt[nl * H + lane] = (nl < nv) ? buf_load(rs_pd, lane * 4, (v0 + nl) * ld_p_bytes) : 0.0f;
    }
#endif
    if (tk >= RP_TILES || !IS_LAYER_M1) __syncthreads();      // (wave-uniform) a slice / Pd tile staged just now
    const int e_begin = __builtin_amdgcn_readfirstlane(rp[0]), e_end = __builtin_amdgcn_readfirstlane(rp[nv]);
    // destination-side segment sums as MFMA products with the 0 / 1 incidence of the tile (exact products; fixed order):
    //   seg_h[m][t]: node 16 m + 4 q + t, column 16 wave + r        seg_x (wave XW only): same node, coordinate r < 3
    constexpr int MTN = (NVB + 15) / 16, XW = WB16 - 1;
    f32x4 seg_h[MTN], seg_x[MTN];
    zero_acc4(seg_h);
    zero_acc4(seg_x);

    for (int win = e_begin; win < e_end; win += WB16 * TE16) {
      const int cb = win + wave * TE16;
      const int nvalid = __builtin_amdgcn_readfirstlane(max(0, min(TE16, e_end - cb)));
      int vt = voff_tile;      // opaque per window: the constant parts of the 48 tile accesses then stay instruction offsets
      asm volatile("" : "+v"(vt));      // (hoisted out of the loop they would be 16 more live registers)
      int vt4 = voff_tile4;
      asm volatile("" : "+v"(vt4));
      STAMPB(1);
      float dy[4][4];   // SiLU'(z2), later SiLU'(z1), tile layout
      float up[4][4];   // dL/dh_neigh[dst] for this tile (prefetched)
#if IS_LAYER_M1 == 1
      float m1v[4][4], d1v[4][4];      // saved m1 = SiLU(z1) and SiLU'(z1) of this tile (prefetched where the gathers were)
      const rsrc_t rm1 = make_rsrc_n(m1s + (size_t)cb * H, nvalid * H * 4);
      const rsrc_t rd1 = make_rsrc_n(dy1s + (size_t)cb * H, nvalid * H * 4);
#elif IS_LAYER_M1 == 2
      float m1v[4][4];                 // saved z1 of this tile (prefetched where the gathers were)
      const rsrc_t rm1 = make_rsrc_n(m1s + (size_t)cb * H, nvalid * H * 4);
#else
      float gth[TE16];  // Ps[src] gathers for the z1 recompute (prefetched; Pd[dst] comes from the LDS tile)
#endif
      // the saved pre-activation tiles only depend on the window position: issue their loads first so that
      // they are in flight during S0's dependent (src index -> coordinates) chain
      float z3v[4][4], z2v[4][4];
      // bounded views of this wave's 16 rows: rows past the tile's last edge read as z = 0 (SiLU(0) = 0: such rows then
      // contribute nothing below without any mask) and are not written
      const rsrc_t rz2 = make_rsrc_n(z2s + (size_t)cb * H, nvalid * H * 4);
      const rsrc_t rz3 = make_rsrc_n((GX && !Z3R) ? z3s + (size_t)cb * H : z2s, (GX && !Z3R) ? nvalid * H * 4 : 0);
      const rsrc_t rdz1 = make_rsrc_n(dZ1 + (size_t)cb * H, nvalid * H * 4);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if constexpr (GX && !Z3R) {
#ifdef IS_ZP_BWD_DW
          f32x4 v3; for (int nt = 0; nt < 4; ++nt) v3[nt] = buf_load(rz3, vt4 + nt * 4 + t * (H * 4), 0);
#else
          const f32x4 v3 = buf_load4(rz3, vt4 + t * (H * 4), 0);
#endif
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) z3v[t][nt] = v3[nt];
        }
#ifdef IS_ZP_BWD_DW
        f32x4 v2; for (int nt = 0; nt < 4; ++nt) v2[nt] = buf_load(rz2, vt4 + nt * 4 + t * (H * 4), 0);
#else
        const f32x4 v2 = buf_load4(rz2, vt4 + t * (H * 4), 0);
#endif
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) z2v[t][nt] = v2[nt];
      }
      [[maybe_unused]] int src_lane = 0;      // S0: source node of edge (lane & 15), kept for the gathers of the z1 recompute
      if (nvalid > 0) {
        // Two forms of the window's first half, selected by the PREPROCESSOR (see the top of the file): the plain form of rounds
        // 1 - 2, and the z3-recompute form.
#if !IS_BWD_Z3R
        // ---- S0: geometry + upstream coordinate gradient, lane = edge ----
        {
          // lanes 16..63 mirror lanes 0..15; every load is unconditional (edge index clamped into the tile's range)
          const int l16 = lane & (TE16 - 1);
          const bool valid = l16 < nvalid;
          const int e = min(cb + l16, e_end - 1);
#if IS_LAYER_GEO
          const f32x4 geo = buf_load4(rs_geo, e * 16, 0);      // (x_src - x_dst, |.|^2) as the forward formed them
#else
          const int s = buf_load_i(rs_srcs, e * 4, 0);
          src_lane = s;
#endif
          int lo = 0, hi = nv;
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (rp[mid] <= e) lo = mid; else hi = mid;
          }
          const int dl = valid ? lo : 0;
          const int v = v0 + dl;
#if !IS_LAYER_GEO
          float xs0, xs1, xs2, xv0, xv1, xv2;
          buf_load3(rs_x, s * 12, 0, xs0, xs1, xs2);
          buf_load3(rs_x, v * 12, 0, xv0, xv1, xv2);
#endif
          float gx0 = 0.0f, gx1 = 0.0f, gx2 = 0.0f;      // GX = false: the layer's coordinate output has no gradient
          if constexpr (GX) buf_load3(rs_gx, v * 12, 0, gx0, gx1, gx2);
          float av[FE_MAX];
#pragma unroll
          for (int f = 0; f < FE_MAX; ++f) av[f] = (f < Fe) ? buf_load(rs_ea, (e * Fe + f) * 4, 0) : 0.0f;      // Fe is kernel-uniform
#if IS_LAYER_GEO
          float d0 = geo[0], d1 = geo[1], d2 = geo[2];
          float rad = geo[3];
#else
          float d0 = xs0 - xv0, d1 = xs1 - xv1, d2 = xs2 - xv2;
          float rad = radial3(d0, d1, d2);
#endif
          float rr = sqrtf(rad);
          float inv = 1.0f / (rr + 1e-30f);
          const float invdeg = 1.0f / (float)max(rp[dl + 1] - rp[dl], 1);
          float g0 = gx0 * invdeg, g1 = gx1 * invdeg, g2 = gx2 * invdeg;
          if (!valid) { d0 = d1 = d2 = rad = rr = inv = g0 = g1 = g2 = 0.0f; }
          if (lane < TE16) {
            sm.e_dl[wave][lane] = dl;
            sm.e_ra[wave][lane * RA_LD] = rad;
            sm.e_r[wave][lane] = rr;
            sm.e_inv[wave][lane] = inv;
            sm.e_d[wave][0][lane] = d0; sm.e_d[wave][1][lane] = d1; sm.e_d[wave][2][lane] = d2;
            if constexpr (GX) {
              sm.e_gx[wave][0][lane] = g0; sm.e_gx[wave][1][lane] = g1; sm.e_gx[wave][2][lane] = g2;
              sm.e_gxd[wave][lane] = (g0 * d0 + g1 * d1 + g2 * d2) * inv;
            }
#pragma unroll
            for (int f = 0; f < FE_MAX; ++f) sm.e_ra[wave][lane * RA_LD + 1 + f] = valid ? av[f] : 0.0f;
          }
        }
        __builtin_amdgcn_wave_barrier();
        STAMPB(2);
        // prefetch dL/dh_neigh rows of this tile's destinations: consumed after WG1 + MM3
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = tile16_row(t, q);
          // rows past nvalid: out of range => zero => dz2 = 0 there
          const int vo = (row < nvalid) ? ((v0 + sm.e_dl[wave][row]) * H + r) * 4 : BUF_OOB;
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
#else
        if constexpr (!GX) {
          // no coordinate gradient: nothing to recompute -- not instantiated (the launcher routes GX = false to the default build)
        } else {
        // ---- S0: geometry + upstream coordinate gradient, lane = edge.  Two forms: the plain one (ids, loads, arithmetic in one
        //      block), and -- EARLY (= Z3R) -- S0a (ids and loads) / S0b (the arithmetic) with the z2 half of E3 and the z3 product
        //      BETWEEN the two: z2v was requested before the source ids (vector loads return in order), so it is there when the
        //      ids are, and its SiLU + the 64 MFMAs of the z3 recompute then cover the coordinates' round trip.  (One shared form
        //      with the arithmetic in a lambda cost the plain path registers: the listed-tile instantiation 170 -> 183 us.) ----

        float xs0 = 0.f, xs1 = 0.f, xs2 = 0.f, xv0 = 0.f, xv1 = 0.f, xv2 = 0.f, gx0 = 0.f, gx1 = 0.f, gx2 = 0.f;
        float av[FE_MAX];
        int dl_e = 0;
        bool valid_e = false;
        {
          const int l16 = lane & (TE16 - 1);
          valid_e = l16 < nvalid;
          const int e = min(cb + l16, e_end - 1);
#if IS_LAYER_GEO
          const f32x4 geo = buf_load4(rs_geo, e * 16, 0);
          xs0 = geo[0]; xs1 = geo[1]; xs2 = geo[2];      // (x_src - x_dst as saved; xv stays 0: s0b forms xs - xv)
#else
          const int s = buf_load_i(rs_srcs, e * 4, 0);
          src_lane = s;
#endif
          int lo = 0, hi = nv;
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (rp[mid] <= e) lo = mid; else hi = mid;
          }
          dl_e = valid_e ? lo : 0;
          const int v = v0 + dl_e;
          if (lane < TE16) sm.e_dl[wave][lane] = dl_e;
#if !IS_LAYER_GEO
          buf_load3(rs_x, s * 12, 0, xs0, xs1, xs2);
          buf_load3(rs_x, v * 12, 0, xv0, xv1, xv2);
#endif
          buf_load3(rs_gx, v * 12, 0, gx0, gx1, gx2);
#pragma unroll
          for (int f = 0; f < FE_MAX; ++f) av[f] = (f < Fe) ? buf_load(rs_ea, (e * Fe + f) * 4, 0) : 0.0f;
        }
        auto s0b = [&]() {      // EARLY only
          float d0 = xs0 - xv0, d1 = xs1 - xv1, d2 = xs2 - xv2;
          float rad = radial3(d0, d1, d2);
          float rr = sqrtf(rad);
          float inv = 1.0f / (rr + 1e-30f);
          const float invdeg = 1.0f / (float)max(rp[dl_e + 1] - rp[dl_e], 1);
          float g0 = gx0 * invdeg, g1 = gx1 * invdeg, g2 = gx2 * invdeg;
          if (!valid_e) { d0 = d1 = d2 = rad = rr = inv = g0 = g1 = g2 = 0.0f; }
          if (lane < TE16) {
            sm.e_ra[wave][lane * RA_LD] = rad;
            sm.e_r[wave][lane] = rr;
            sm.e_inv[wave][lane] = inv;
            sm.e_d[wave][0][lane] = d0; sm.e_d[wave][1][lane] = d1; sm.e_d[wave][2][lane] = d2;
            sm.e_gx[wave][0][lane] = g0; sm.e_gx[wave][1][lane] = g1; sm.e_gx[wave][2][lane] = g2;
            sm.e_gxd[wave][lane] = (g0 * d0 + g1 * d1 + g2 * d2) * inv;
#pragma unroll
            for (int f = 0; f < FE_MAX; ++f) sm.e_ra[wave][lane * RA_LD + 1 + f] = valid_e ? av[f] : 0.0f;
          }
        };
        __builtin_amdgcn_wave_barrier();
        STAMPB(2);
        // prefetch dL/dh_neigh rows of this tile's destinations: consumed after WG1 + MM3
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = tile16_row(t, q);
          // rows past nvalid: out of range => zero => dz2 = 0 there
          const int vo = (row < nvalid) ? ((v0 + sm.e_dl[wave][row]) * H + r) * 4 : BUF_OOB;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) up[t][nt] = buf_load(rs_ghn, vo + nt * 64, 0);      // written by the node phase
        }

        // ---- E3: coord-MLP tail backward; dz3 -> bufA, mh -> bufB, SiLU'(z2) -> registers ----
        if constexpr (GX) {
          if constexpr (Z3R) {
            // mh = SiLU(z2) -> bufB first (this wave's own tile), then z3 = mh Wc1^T + bc1 on the MFMA pipe.  Rows past
            // nvalid: mh = 0, z3 = bc1 -- finite, and multiplied by ds = 0 below
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int nt = 0; nt < 4; ++nt) {
                float mh;
                silu_fg(z2v[t][nt], mh, dy[t][nt]);
                bufB[tile16_row(t, q) * LD + nt * 16 + r] = mh;
              }
            __builtin_amdgcn_wave_barrier();
            f32x4 z3a[4];
            zero_acc4(z3a);
            mm16_rows_bt<4, H>(z3a, bufB, sm.wc1t, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int nt = 0; nt < 4; ++nt) z3v[t][nt] = z3a[nt][t] + bc1_c[nt];
            s0b();      // the coordinates have arrived by now
            __builtin_amdgcn_wave_barrier();
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int row = tile16_row(t, q);
            float tt[4], sp[4];
            float part = 0.0f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
              silu_fg(z3v[t][nt], tt[nt], sp[nt]);
              part += tt[nt] * wc2_c[nt];
              if constexpr (!Z3R) {
                float mh;
                silu_fg(z2v[t][nt], mh, dy[t][nt]);
                bufB[row * LD + nt * 16 + r] = mh;      // rows past nvalid: SiLU(0) = 0
              }
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
        }
#endif
#if IS_LAYER_M1
        // prefetch the saved first activation (tile layout, rows past nvalid read as 0): in flight during WG1 + MM3
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#ifdef IS_ZP_BWD_DW
          f32x4 v1; for (int nt = 0; nt < 4; ++nt) v1[nt] = buf_load(rm1, vt4 + nt * 4 + t * (H * 4), 0);
#else
          const f32x4 v1 = buf_load4(rm1, vt4 + t * (H * 4), 0);
#endif
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) m1v[t][nt] = v1[nt];
#if IS_LAYER_M1 == 1
          const f32x4 vd = buf_load4(rd1, vt4 + t * (H * 4), 0);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) d1v[t][nt] = vd[nt];
#endif
        }
#else
        // prefetch the gathers of the z1 recompute (SA): in flight during WG1 + MM3
#pragma unroll
        for (int i = 0; i < TE16; ++i)      // one v_readlane + s_mul + buffer_load per row; not consumed before SA
          gth[i] = buf_load(rs_ps, lane * 4, __builtin_amdgcn_readlane(src_lane, i) * ld_p_bytes);
#endif
      }
      if constexpr (GX) {
        STAMPB(3);
        __syncthreads();   // every wave's dz3 / mh tiles are staged
        STAMPB(4);

        // ---- WG1: dWc1[16w.., :] += sum over the window's edge tiles of dz3^T mh ----
#pragma unroll
        for (int wt = 0; wt < WB16; ++wt)
#ifdef IS_ABL_NOMFMA
          if (win + wt * TE16 < e_end) dWc1[0][0] += sm.u.w.bufA[wt][lane] * sm.u.w.bufB[wt][lane];
#else
          if (win + wt * TE16 < e_end) mm16_outer_rows(dWc1, sm.u.w.bufA[wt], sm.u.w.bufB[wt], wave, lane);
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
        STAMPB(5);
        __syncthreads();   // WG1 + MM3 reads of bufA / bufB are complete in all waves
        STAMPB(6);
      }

      if (nvalid > 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) bufA[tile16_row(t, q) * LD + nt * 16 + r] = dy[t][nt];   // dz2

#if IS_LAYER_M1 == 1
        // ---- E1: m1 -> bufB, SiLU'(z1) -> registers (both read back) ----
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            bufB[tile16_row(t, q) * LD + nt * 16 + r] = m1v[t][nt];
            dy[t][nt] = d1v[t][nt];
          }
#elif IS_LAYER_M1 == 2
        // ---- E1: m1 = SiLU(z1) -> bufB, SiLU'(z1) -> registers, from the z1 read back (rows past nvalid: z1 = 0 -> m1 = 0) ----
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            float y;
            silu_fg(m1v[t][nt], y, dy[t][nt]);
            bufB[tile16_row(t, q) * LD + nt * 16 + r] = y;
          }
#else
        // ---- SA: recompute z1 (lane = channel) -> bufB ----
        {
#pragma unroll
          for (int i = 0; i < TE16; ++i) {
            float z1 = gth[i] + sm.pdt[sm.e_dl[wave][i] * H + lane] + sm.e_ra[wave][i * RA_LD] * wr_c;
            if constexpr (FE_MAX == 1) {
              z1 += sm.e_ra[wave][i * RA_LD + 1] * wa_c;
            } else {
#pragma unroll
              for (int f = 0; f < FE_MAX; ++f) z1 += sm.e_ra[wave][i * RA_LD + 1 + f] * sm.wa[f * H + lane];
            }
            bufB[i * LD + lane] = z1;      // rows past nvalid: some finite value; their dz2 / dz1 are zero
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
            bufB[row * LD + nt * 16 + r] = y;
          }
        }
#endif
      }
      STAMPB(7);
      __syncthreads();   // every wave's dz2 / m1 tiles are staged
      STAMPB(8);

      // ---- WG2: dW2[16w.., :] += sum over edge tiles of dz2^T m1 ----
#pragma unroll
      for (int wt = 0; wt < WB16; ++wt)
#ifdef IS_ABL_NOMFMA
        if (win + wt * TE16 < e_end) dW2[0][0] += sm.u.w.bufA[wt][lane] * sm.u.w.bufB[wt][lane];
#else
        if (win + wt * TE16 < e_end) mm16_outer_rows(dW2, sm.u.w.bufA[wt], sm.u.w.bufB[wt], wave, lane);
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
#ifndef IS_ABL_STORES
            dD[e * 3 + 0] = q0; dD[e * 3 + 1] = q1; dD[e * 3 + 2] = q2;
#endif
          }
          sm.e_gx[wave][0][lane] = q0; sm.e_gx[wave][1][lane] = q1; sm.e_gx[wave][2][lane] = q2;
        }
      }
      __syncthreads();

      STAMPB(11);
      // ---- WG3: [dw_r | dW_a][16w.., :] += sum over the window's edge tiles of dz1^T [radial | a] ----
#pragma unroll
      for (int wt = 0; wt < WB16; ++wt)
        if (win + wt * TE16 < e_end) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int e = 4 * q + s;
            const float b = sm.e_ra[wt][e * RA_LD + min(r, FE_MAX)];     // columns past 1 + FE_MAX are zero
            dWra = __builtin_amdgcn_mfma_f32_16x16x4f32(sm.u.w.bufA[wt][e * LD + wave * 16 + r], r <= FE_MAX ? b : 0.0f, dWra, 0, 0, 0);
          }
        }
      // ---- SEG: dPd[v] += sum over the window's in-edges of v of dz1, dx[v] -= sum of dL/dd -- as [nodes x edges] x
      //      [edges x columns] products with the incidence matrix built from e_dl (rows past a tile's end: dz1 = dL/dd = 0).
      //      Wave w owns columns [16 w, 16 w + 16); the coordinate columns ride on the last wave (the one whose edge tile
      //      is empty in most windows).  Replaces a per-node loop over LDS rows (dependent reads: 4 k cycles per window).
#ifndef IS_ABL_SEG
#pragma unroll
      for (int wt = 0; wt < WB16; ++wt)
        if (win + wt * TE16 < e_end) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int e = 4 * q + s;
            const int dl = sm.e_dl[wt][e];
            const float bh = sm.u.w.bufA[wt][e * LD + wave * 16 + r];
            float bx = 0.0f;
            if (wave == XW) bx = sm.e_gx[wt][min(r, 2)][e];
#pragma unroll
            for (int m = 0; m < MTN; ++m) {
              const float ind = (dl == 16 * m + r) ? 1.0f : 0.0f;
              seg_h[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ind, bh, seg_h[m], 0, 0, 0);
              if (wave == XW) seg_x[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ind, r < 3 ? bx : 0.0f, seg_x[m], 0, 0, 0);
            }
          }
        }
#else
      seg_h[0][0] += sm.u.w.bufA[0][lane]; seg_x[0][0] += sm.e_gx[0][0][r];
#endif
      __syncthreads();
    }

    STAMPB(12);
#pragma unroll
    for (int m = 0; m < MTN; ++m)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int nl = 16 * m + tile16_row(t, q);
        if (nl < nv) {
          const int v = v0 + nl;
          dPd[(size_t)v * ld_dpd + wave * 16 + r] = seg_h[m][t];
          if (wave == XW && r < 3) dx[v * 3 + r] = (GX ? gxsrc[v * 3 + r] : 0.0f) - seg_x[m][t];
        }
      }
  }

  // ---- write the workgroup's partial record: each wave owns 16 rows of dW2 / dWc1 ----
  STAMPP(18);
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
    float* vec = &sm.u.w.bufB[0][0];  // [wave][slot][64]
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
    // the third outer product: channel o = 16 wave + 4q + t, column j = r (0: dw_r, 1..8: dW_a[o][j-1])
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int o = wave * 16 + tile16_row(t, q);
      if (r == 0) part[2 * H * H + 3 * H + o] = dWra[t];
      else if (r <= 8) part[2 * H * H + 4 * H + o * 8 + (r - 1)] = dWra[t];
    }
    __syncthreads();
    for (int idx = tid; idx < SLOTS * H; idx += 256) {
      const int sidx = idx / H, c = idx % H;
      float v = 0.0f;
      for (int w = 0; w < WB16; ++w) v += vec[(w * SLOTS + sidx) * H + c];
      part[2 * H * H + sidx * H + c] = v;
    }
  }
  STAMPP(19);
  wg_clock_end(wg_clock);
}

}  // namespace is

#if !IS_BWD_Z3R
#ifdef IS_STAGE_STAMPS
extern "C" int is_debug_stamps_bwd(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(is::g_stamps_b), sizeof(long long) * 24) == hipSuccess ? 0 : is::fail(__func__, -5);
}
#endif
#endif

// the launches of this translation unit's form (default build: z3 read back; egnn_layer_bwd_z3r.hip: z3 recomputed, GX only)
namespace is {
int launch_layer_bwd_z3r(const float* ps, const float* pd, const float* x, const float* ea, const int32_t* rowptr, const int32_t* srcs,
                         const float* W1, int din, const float* W2, const float* Wc1, const float* wc2, const float* z2s,
                         const float* z3s_or_bc1, const float* g_xout, float* dZ1, float* dD, float* dPd, float* dx, float* partials,
                         int grid, int N, int Fe, bool gather, bool gx, const NodeBwdArgs& nb,
                         long long* wg_clock, const float* m1s, const float* dy1s, const float* geos, hipStream_t st);
int IS_BWD_LAUNCHER(const float* ps, const float* pd, const float* x, const float* ea, const int32_t* rowptr, const int32_t* srcs,
                    const float* W1, int din, const float* W2, const float* Wc1, const float* wc2, const float* z2s,
                    const float* z3s_or_bc1, const float* g_xout, float* dZ1, float* dD, float* dPd, float* dx, float* partials,
                    int grid, int N, int Fe, bool gather, bool gx, const NodeBwdArgs& nb,
                    long long* wg_clock, const float* m1s, const float* dy1s, const float* geos, hipStream_t st) {
  const dim3 block(256);
#define IS_LAUNCH_LB(FE, NVB, GXF, GA, DI)                                                                                       \
  hipLaunchKernelGGL((is::IS_BWD_KERNEL<FE, NVB, GXF, GA, DI>), dim3(grid), block, 0, st, ps, pd, x, ea, rowptr, srcs, W1, W2, Wc1, \
                     wc2, z2s, z3s_or_bc1, g_xout, dZ1, dD, dPd, dx, partials, N, Fe, nb, wg_clock, m1s, dy1s, geos)
#define IS_LAUNCH_LB_D(FE, NVB, GXF, GA) do { if (din == 20) IS_LAUNCH_LB(FE, NVB, GXF, GA, 20); else IS_LAUNCH_LB(FE, NVB, GXF, GA, 64); } while (0)
#if IS_BWD_Z3R
#define IS_LAUNCH_LB_G(FE, NVB) do { if (gather) IS_LAUNCH_LB_D(FE, NVB, true, true); else IS_LAUNCH_LB_D(FE, NVB, true, false); } while (0)
  if (!gx) return is::fail(__func__, -22);
#else
#define IS_LAUNCH_LB_G(FE, NVB)                                        \
  do {                                                                 \
    if (gather) IS_LAUNCH_LB_D(FE, NVB, true, true);                   \
    else if (gx) IS_LAUNCH_LB_D(FE, NVB, true, false);                 \
    else IS_LAUNCH_LB_D(FE, NVB, false, false);                        \
  } while (0)
#endif
  if (Fe <= 1) IS_LAUNCH_LB_G(1, is::NV16);
  else IS_LAUNCH_LB_G(8, is::NV16);
#undef IS_LAUNCH_LB_G
#undef IS_LAUNCH_LB_D
#undef IS_LAUNCH_LB
  return is::launch_status(__func__);
}
}  // namespace is

#if !IS_BWD_Z3R
// One EGNNConv layer backward (source gather of the layer above + node data path + edge pass).
//   edge half: arguments of the forward plus z2s / z3s (saved pre-activations), outputs dZ1 [E,64] / dD [E,3] (per-edge
//     gradients in CSR slot order, gathered by source by the NEXT call or by is_gather_segment_sum), dPd [N, ld_dpd] and dx
//     [N,3] (destination-side parts), one partial weight-gradient record per workgroup (`grid` persistent workgroups):
//       dW2 [64,64] | dWc1 [64,64] | db2 | dbc1 | dwc2 | dw_r [64] | dW_a [64,8]
//   node half: g_h [N,64] direct gradient of the layer's output h (may be NULL); g_psd [N,128] gradient of the next
//     pre-projection of h (NULL: none; then dh = g_h); zn1 saved; bpack = the layer's backward operand pack; outputs dh_total
//     (with g_psd), dzn1, d_h (first din columns; may be NULL), d_hn (scratch [N,64] read back by the edge half).
//   dZ1n != NULL ("gather"): dZ1n / dDn / dxn are dZ1 / dD / dx of the layer ABOVE and rowptr_src / pos_by_src the CSR-by-source
//     index; the kernel then completes g_psd[:, :64] = gather(dZ1n) (written: the weight-gradient kernel reads it) and uses
//     dxn + gather(dDn) (written to gxtot [N,3], scratch) as the coordinate gradient; g_xout must be NULL.  Otherwise g_xout [N,3] (or NULL: no coordinate
//     gradient; the coordinate-MLP half of the pass is skipped, z3s / Wc1 / wc2 are not read).
//   m1s / dy1s [E, 64]: the first edge-MLP activation SiLU(z1) and SiLU'(z1) saved by is_egnn_layer_fwd -- read when the library
//     was built with IS_LAYER_M1 = 1 (is_layer_saves_m1()), ignored otherwise.
//   geos [E, 4]: (x_src - x_dst, |x_src - x_dst|^2) per edge slot saved by is_egnn_layer_fwd -- read when the library was built with
//     IS_LAYER_GEO = 1 (is_layer_saves_geo()), ignored otherwise.
//   wg_clock: NULL, or [grid][2] int64 -- every workgroup's start / end device wall clock (bench.py's in-situ launch timing).
//   z3s == NULL with a coordinate gradient: the forward did not save z3; it is recomputed per tile as SiLU(z2) Wc1^T + bc1
//     (bc1 = coord_mlp.0.bias, required then).
extern "C" int is_egnn_layer_bwd(const float* ps, const float* pd, int ld_p, const float* x, const float* ea,
                                 const int32_t* rowptr, const int32_t* srcs, const float* W1, int ldw, int din,
                                 const float* W2, const float* Wc1, const float* bc1, const float* wc2, const float* z2s,
                                 const float* z3s, const float* g_xout, float* dZ1, float* dD, float* dPd, int ld_dpd,
                                 float* dx, float* partials, int grid, int N, int Fe,
                                 const float* dZ1n, const float* dDn, const float* dxn, const int32_t* rowptr_src,
                                 const int32_t* pos_by_src, const float* g_h, float* g_psd, const float* zn1,
                                 const float* bpack, float* dh_total, float* dzn1, float* d_h, float* d_hn, float* gxtot,
                                 long long* wg_clock, const float* m1s, const float* dy1s, const float* geos, void* stream) {
  if (N <= 0) return 0;
  if ((long long)N * is::H * 4 >= 0x7ffff000LL) return is::fail(__func__, -22);      // 32-bit byte offsets of the raw-buffer views (8.3 M nodes)
  // (the gathered dZ1n / dDn rows of the layer above are addressed the same way: E * 256 bytes < 0x7fffe000 -- the layer that wrote
  //  them, and this one's own z2 / dZ1 views, already required that of the same E)
  if ((IS_LAYER_M1 != 0 && m1s == nullptr) || (IS_LAYER_M1 == 1 && dy1s == nullptr) || (IS_LAYER_GEO && geos == nullptr))
    return is::fail(__func__, -22);      // this build reads them
  const bool gather = dZ1n != nullptr;
  const bool gx = gather || g_xout != nullptr;
  if (Fe < 0 || Fe > 8 || grid <= 0 || (din != 20 && din != 64) || zn1 == nullptr || bpack == nullptr || dzn1 == nullptr ||
      d_hn == nullptr || (g_psd != nullptr && dh_total == nullptr) || (g_psd == nullptr && g_h == nullptr) ||
      (gather && (g_xout != nullptr || dDn == nullptr || dxn == nullptr || rowptr_src == nullptr || pos_by_src == nullptr ||
                  g_psd == nullptr || gxtot == nullptr)) ||
      (gx && z3s == nullptr && bc1 == nullptr) ||
      ld_p != 2 * is::H || ld_dpd != 2 * is::H || ldw != 2 * din + 1 + Fe)      // the layouts the kernels are built for
    return is::fail(__func__, -22);
  const bool z3r = gx && z3s == nullptr;      // z3 was not saved by the forward: recomputed from z2 (the other translation unit)
  hipStream_t st = static_cast<hipStream_t>(stream);
  const is::NodeBwdArgs nb{dZ1n, dDn, dxn, rowptr_src, pos_by_src, g_h, g_psd, zn1, bpack, dh_total, dzn1, d_h, d_hn, gxtot};
  if (z3r)
    return is::launch_layer_bwd_z3r(ps, pd, x, ea, rowptr, srcs, W1, din, W2, Wc1, wc2, z2s, bc1, g_xout, dZ1, dD, dPd, dx, partials,
                                    grid, N, Fe, gather, gx, nb, wg_clock, m1s, dy1s, geos, st);
  return is::launch_layer_bwd(ps, pd, x, ea, rowptr, srcs, W1, din, W2, Wc1, wc2, z2s, z3s, g_xout, dZ1, dD, dPd, dx, partials,
                              grid, N, Fe, gather, gx, nb, wg_clock, m1s, dy1s, geos, st);
}

// 1: this library's is_egnn_layer_bwd reads the edge geometry from geos (filled by is_egnn_layer_fwd); 0: it recomputes it.
extern "C" int is_layer_saves_geo(void) { return IS_LAYER_GEO; }

// 1: this library's is_egnn_layer_bwd reads the first edge-MLP activation from m1s / dy1s (the caller lets is_egnn_layer_fwd save
// them); 0: it recomputes it (m1s / dy1s ignored, the forward need not save them).  Build-time choice (Makefile M1=...).
extern "C" int is_layer_saves_m1(void) { return IS_LAYER_M1; }
#endif
