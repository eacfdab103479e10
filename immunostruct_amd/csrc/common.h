// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels.
//
// Conventions used by every kernel in this directory
//   * wavefront = 64 lanes; workgroups are 256 threads = 4 waves unless stated.
//   * hidden width H = 64 channels (reference: gat_hidden_channels = 64,
//     models/hybrid_models.py:247) -- one wave lane per channel in the
//     "lane = channel" phases, two 32-column MFMA tiles in the matrix phases.
//   * matrix work uses v_mfma_f32_32x32x2_f32 (exact fp32, bit-equal to an fmaf
//     chain).  Operand map (lane l, r = l & 31, hf = l >> 5):
//         A[i = r][k-slot = hf]   B[k-slot = hf][j = r]
//         D reg t  ->  row (t & 3) + 8 * (t >> 2) + 4 * hf ,  col r
//     The two k-slots of one instruction may be ANY two distinct k as long as
//     A and B agree, so half hf walks k in [hf*K/2, (hf+1)*K/2): each lane then
//     reads CONTIGUOUS k from LDS and one ds_read_b128 feeds four MFMAs.
//   * LDS tiles are row-major with row stride LD = 68 floats: 16-byte aligned
//     rows for ds_read_b128 and conflict-free for both the row-per-lane b128
//     reads ((4*row) mod 64 distinct per 16-lane group) and the
//     lane-per-column b32 accesses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace is {

// ---- host side: what went wrong, for the caller (is_last_error_string).  Every entry point returns 0 or a negative errno-style
// code and never throws; the text of the calling THREAD's last failure is kept beside the code: the entry point, the reason, and
// for a failed launch HIP's own error string.  (Defined in abi_misc.hip.)
int fail(const char* entry, int code);          // records "<entry>: <reason> (<code>)", returns code
int launch_status(const char* entry);           // hipGetLastError(): 0, or records "<entry>: <hip error string> (-5)" and returns -5

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int H = 64;    // hidden channels
constexpr int LD = 68;   // LDS row stride in floats
constexpr int TE = 32;   // edges (rows) per wave tile
constexpr int NV = 32;   // destination nodes owned by one workgroup pass
constexpr int WAVES = 4;

__device__ __forceinline__ float rcp_f(float v) { return __builtin_amdgcn_rcpf(v); }

// SiLU and its derivative.  sigma = 1 / (1 + exp(-z)).
// (IS_ABL_* switches: TIMING-ONLY ablation builds, results wrong on purpose -- HISTORY.md "issue-bound model, ablations")
#ifdef IS_ABL_SILU
__device__ __forceinline__ float silu_f(float z) { return z * 0.5f; }
__device__ __forceinline__ void silu_fg(float z, float& y, float& dy) { y = z * 0.5f; dy = 0.5f; }
#else
// sigma(z) = 1 / (1 + exp(-z)) as v_exp_f32 of the scaled argument + v_rcp_f32 (~|z| + 1 ulp).  IS_SILU_ACCURATE builds a ~1 ulp
// form (the rounding error of the product -z log2 e recovered with one fma and folded back, 2^(t + lo) = 2^t (1 + lo ln 2); one
// Newton step on the reciprocal; 7 more full-rate instructions).  Round 4 measured both: the accurate form does NOT move the
// full-size gradients' distance from the fp64 oracle (worst tensor 8.93 x the element-wise bound against 9.09 x: the per-term
// SiLU error is not what separates the HIP path from torch's fp32 there -- HISTORY.md), and costs the iedb step nothing measurable,
// the paired step 2.3 %, the stress stack 3.4 %.  Hence opt-in.
__device__ __forceinline__ float sigmoid_f(float z) {
#ifndef IS_SILU_ACCURATE
  return rcp_f(1.0f + __expf(-z));
#else
  constexpr float L2E = 1.44269504088896340736f;          // log2(e) rounded to fp32
  constexpr float L2E_LO = 1.92596299112661746e-8f;       // log2(e) - L2E
  const float nz = -z;
  const float t = nz * L2E;
  const float lo = __builtin_fmaf(nz, L2E_LO, __builtin_fmaf(nz, L2E, -t));     // exact product error + the constant's tail
  float e = __builtin_amdgcn_exp2f(__builtin_fminf(t, 126.0f));                 // (z < -87: sigma < 2^-126 either way; no inf below)
  e = __builtin_fmaf(e, lo * 0.693147180559945309f, e);
  const float d = 1.0f + e;
  float s = rcp_f(d);
  s = __builtin_fmaf(__builtin_fmaf(-d, s, 1.0f), s, s);  // one Newton step
  return s;
#endif
}
__device__ __forceinline__ float silu_f(float z) {
  return z * sigmoid_f(z);
}
__device__ __forceinline__ void silu_fg(float z, float& y, float& dy) {
  const float s = sigmoid_f(z);
  y = z * s;
  dy = s * (1.0f + z * (1.0f - s));
}
#endif

// |d|^2 with a FIXED contraction (the compiler otherwise picks fma chains or packed multiplies per call site,
// and the forward / backward / v2 / v3 kernels would disagree in the last bit).
__device__ __forceinline__ float radial3(float d0, float d1, float d2) {
  return __builtin_fmaf(d2, d2, __builtin_fmaf(d1, d1, d0 * d0));
}

// D-register t of a 32x32 tile -> row inside the tile.
__device__ __forceinline__ int tile_row(int t, int hf) { return (t & 3) + 8 * (t >> 2) + 4 * hf; }

// acc[nt] (32 x 32) += A[32 x K] * W[nt*32 .. nt*32+32) x K]^T
//   a_lds: tile row 0 of A (row stride LDA), w_lds: row 0 of W (row stride LDW).
//   LDA/4 and LDW/4 must be odd (conflict-free b128 rows) and K a multiple of 8.
template <int NT, int K, int LDA = LD, int LDW = LD>
__device__ __forceinline__ void mm_rows(f32x16 (&acc)[NT], const float* a_lds, const float* w_lds, int lane) {
  const int r = lane & 31, hf = lane >> 5;
  const float* ap = a_lds + r * LDA + hf * (K / 2);
  const float* wp = w_lds + r * LDW + hf * (K / 2);
  // operands of step s+4 are fetched from LDS before the MFMAs of step s are issued
  f32x4 a = *reinterpret_cast<const f32x4*>(ap);
  f32x4 b[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const f32x4*>(wp + nt * 32 * LDW);
#pragma unroll
  for (int s = 0; s < K / 2; s += 4) {
    f32x4 an = a;
    f32x4 bn[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bn[nt] = b[nt];
    if (s + 4 < K / 2) {
      an = *reinterpret_cast<const f32x4*>(ap + s + 4);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bn[nt] = *reinterpret_cast<const f32x4*>(wp + nt * 32 * LDW + s + 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[nt][j], acc[nt], 0, 0, 0);
    }
    a = an;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = bn[nt];
  }
}

// acc[mt][nt] (32 x 32) += sum over the 32 tile rows e of G[e][mt*32 + i] * M[e][nt*32 + j]
//   (outer-product accumulation: the tile ROW index is the contraction index).
template <int MT, int NT, int LDG = LD, int LDM = LD>
__device__ __forceinline__ void mm_outer(f32x16 (&acc)[MT][NT], const float* g_lds, const float* m_lds, int lane) {
  const int r = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int s = 0; s < TE / 2; ++s) {
    const int e = hf * (TE / 2) + s;
    float a[MT], b[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = g_lds[e * LDG + mt * 32 + r];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = m_lds[e * LDM + nt * 32 + r];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
  }
}

template <int N>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[i][t] = 0.0f;
}
template <int M, int N>
__device__ __forceinline__ void zero_acc2(f32x16 (&acc)[M][N]) {
#pragma unroll
  for (int i = 0; i < M; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.0f;
}

// ---------------------------------------------------------------------------
// 16-row tiles on v_mfma_f32_16x16x4_f32 (32-cycle issue, 4 accumulator registers):
//   A[i = l & 15][k-slot = l >> 4]   B[k-slot = l >> 4][j = l & 15]
//   D reg t -> row 4 * (l >> 4) + t , col l & 15
// Quarter q = l >> 4 walks k in [q*K/4, (q+1)*K/4) (contiguous => ds_read_b128).
// Half the per-wave LDS footprint of the 32-row tiles => twice the waves per CU,
// which is what hides the gather / epilogue latency on these small batches.
constexpr int TE16 = 16;
__device__ __forceinline__ int tile16_row(int t, int q) { return 4 * q + t; }

template <int NT, int K, int LDA = LD, int LDW = LD>
__device__ __forceinline__ void mm16_rows(f32x4 (&acc)[NT], const float* a_lds, const float* w_lds, int lane) {
  const int r = lane & 15, q = lane >> 4;
  const float* ap = a_lds + r * LDA + q * (K / 4);
  const float* wp = w_lds + r * LDW + q * (K / 4);
  f32x4 a = *reinterpret_cast<const f32x4*>(ap);
  f32x4 b[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const f32x4*>(wp + nt * 16 * LDW);
#pragma unroll
  for (int s = 0; s < K / 4; s += 4) {
    f32x4 an = a;
    f32x4 bn[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bn[nt] = b[nt];
    if (s + 4 < K / 4) {
      an = *reinterpret_cast<const f32x4*>(ap + s + 4);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bn[nt] = *reinterpret_cast<const f32x4*>(wp + nt * 16 * LDW + s + 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[nt][j], acc[nt], 0, 0, 0);
    }
    a = an;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = bn[nt];
  }
}

// acc[nt] (16 x 16) += A[16 x K] * Wt[K x (nt*16 .. nt*16+16)]  -- the same product as mm16_rows, but with the weight matrix
// stored TRANSPOSED (wt_lds[k][j] = W[j][k], row stride LDW): lets a kernel that keeps W^T staged for the data-gradient
// product (dX = dY W) also run the forward product (Y = X W^T) from the same LDS tile.  The B operand is then one
// ds_read_b32 per MFMA (lane (r, q) reads wt[k][nt*16 + r]); quarter q walks k = 16 g + 4 q + j (g, j = 0..3), so that the
// A operand is still one ds_read_b128 per four MFMAs and the two quarters of a 32-lane LDS group (32 banks for b32 reads)
// hit disjoint bank halves: bank = (4 k + r) mod 32 = (16 q + 4 j + r) mod 32.
template <int NT, int K, int LDA = LD, int LDW = LD>
__device__ __forceinline__ void mm16_rows_bt(f32x4 (&acc)[NT], const float* a_lds, const float* wt_lds, int lane) {
  const int r = lane & 15, q = lane >> 4;
  const float* ap = a_lds + r * LDA + 4 * q;
  const float* wp = wt_lds + (4 * q) * LDW + r;
  // the B operands of step (g, j + 1) are fetched before the MFMAs of step (g, j) are issued: 2 * NT registers in flight
  f32x4 a = *reinterpret_cast<const f32x4*>(ap);
  float b[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) b[nt] = wp[nt * 16];
#pragma unroll
  for (int g = 0; g < K / 16; ++g) {
    f32x4 an = a;
    if (g + 1 < K / 16) an = *reinterpret_cast<const f32x4*>(ap + 16 * (g + 1));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float bn[NT];
      const int kn = 16 * g + j + 1 + (j == 3 ? 12 : 0);      // k row (before the + 4 q of wp) of the next step
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bn[nt] = (g + 1 < K / 16 || j < 3) ? wp[kn * LDW + nt * 16] : 0.0f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[nt], acc[nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = bn[nt];
    }
    a = an;
  }
}

// acc[mt][nt] (16 x 16) += sum over the 16 tile rows e of G[e][mt*16 + i] * M[e][nt*16 + j]
template <int MT, int NT, int LDG = LD, int LDM = LD>
__device__ __forceinline__ void mm16_outer(f32x4 (&acc)[MT][NT], const float* g_lds, const float* m_lds, int lane) {
  const int r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int e = 4 * q + s;
    float a[MT], b[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = g_lds[e * LDG + mt * 16 + r];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = m_lds[e * LDM + nt * 16 + r];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
  }
}

// acc[nt] (16 x 16) += sum over the 16 rows e of one edge tile of G[e][mt*16 + i] * M[e][nt*16 + j]
//   for ONE output row-tile mt (the calling wave's share of a 64 x 64 weight gradient).
__device__ __forceinline__ void mm16_outer_rows(f32x4 (&acc)[4], const float* g_lds, const float* m_lds, int mt, int lane) {
  const int r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int e = 4 * q + s;
    const float a = g_lds[e * LD + mt * 16 + r];
    float b[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) b[nt] = m_lds[e * LD + nt * 16 + r];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[nt], acc[nt], 0, 0, 0);
  }
}

template <int N>
__device__ __forceinline__ void zero_acc4(f32x4 (&acc)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// sum of v over the 16 lanes that share q = lane >> 4 (one DPP row): four v_add_f32_dpp, no LDS traffic.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum_over_r16(float v) {
  v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);   // row_half_mirror
  v += dpp_move<0x140>(v);   // row_mirror
  return v;
}

// ---------------------------------------------------------------------------
// Raw buffer access: scalar (SGPR) row offset + per-lane (VGPR) offset, both in BYTES and 32-bit.
// A gathered row costs one v_readlane + one s_mul + one buffer_load -- no 64-bit vector address math.
// The compiler tracks these like ordinary loads (s_waitcnt vmcnt(N) with exact counts).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, -1, 0x00020000);
}
// bounded view: accesses at byte offsets (voff + instruction offset) >= nbytes return 0 / are dropped by the hardware -- the
// range check that replaces per-element predication (the scalar offset is NOT part of the check: put it into the base).
__device__ __forceinline__ rsrc_t make_rsrc_n(const void* p, int nbytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, nbytes, 0x00020000);
}
constexpr int BUF_OOB = 0x7ffff000;      // a voffset that is out of range for every bounded view
__device__ __forceinline__ float buf_load(rsrc_t rs, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
__device__ __forceinline__ int buf_load_i(rsrc_t rs, int voff, int soff) {
  return __builtin_bit_cast(int, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
// three consecutive floats (a node's coordinates): three dword loads off ONE offset register (the b96 builtin of this
// toolchain loads a single dword)
__device__ __forceinline__ void buf_load3(rsrc_t rs, int voff, int soff, float& a, float& b, float& c) {
  a = buf_load(rs, voff, soff);
  b = buf_load(rs, voff + 4, soff);
  c = buf_load(rs, voff + 8, soff);
}
// four consecutive floats (16-byte aligned offset): one buffer_load_dwordx4
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 buf_load4(rsrc_t rs, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
// 16-byte store.  NO scalar offset: a buffer store of more than 8 bytes WITH an SGPR offset reads its data registers late, and
// this toolchain leaves only one instruction between such a store and the next VALU write of those registers -- measured on
// gfx950 (round 4): z2 / z3 rows stored as `buffer_store_dwordx4 v[18:21], v108, s[52:55], s51 offen` followed two instructions
// later by `v_pk_add_f32 v[18:19], ...` came out corrupted at full residency (nondeterministic gradients at B = 128, identical
// ones at B = 32).  The hazard does not exist for stores without an SGPR offset: callers fold their scalar part into `voff`.
__device__ __forceinline__ void buf_store4(f32x4 v, rsrc_t rs, int voff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs, voff, 0, 0);
}
__device__ __forceinline__ void buf_store(float v, rsrc_t rs, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff, soff, 0);
}

// Sum a [64 x 64] accumulator (2 x 2 tiles of 32 x 32, one copy per wave) over the 4 waves of a
// workgroup and store it to dst[o * ld_dst + i].  Every wave first dumps its copy into its own LDS
// region (independent stores), then all 256 threads add the 4 regions in a fixed order -- no
// serialized read-modify-write chain.  scratch: 4 * 4096 floats of LDS; contains two barriers.
__device__ __forceinline__ void wg_sum_store_64x64(const f32x16 (&acc)[2][2], float* scratch, float* dst, int ld_dst,
                                                   int tid, int wave, int lane) {
  const int r = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int t = 0; t < 16; ++t)
        scratch[wave * 4096 + (mt * 32 + tile_row(t, hf)) * 64 + nt * 32 + r] = acc[mt][nt][t];
  __syncthreads();
  for (int idx = tid; idx < 4096; idx += 256) {
    const float v = ((scratch[idx] + scratch[4096 + idx]) + scratch[8192 + idx]) + scratch[12288 + idx];
    dst[(idx >> 6) * ld_dst + (idx & 63)] = v;
  }
  __syncthreads();
}

// sum / maximum of v over the 32 lanes that share hf (lanes differ in r = lane & 31): the 16 lanes of a DPP row by four DPP
// operations, the two rows of the half by ONE ds_swizzle (lane ^ 16).  (As a loop of five __shfl_xor this was five ds_bpermute
// trips through the LDS crossbar, each with its address arithmetic: 166 of them per wave and head in the node attention forward.)
__device__ __forceinline__ float swap_rows16(float v) {      // value of lane ^ 16
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));
}
__device__ __forceinline__ float sum_over_r(float v) {
  v = sum_over_r16(v);
  return v + swap_rows16(v);
}
__device__ __forceinline__ float max_over_r(float v) {
  v = fmaxf(v, dpp_move<0xB1>(v));
  v = fmaxf(v, dpp_move<0x4E>(v));
  v = fmaxf(v, dpp_move<0x141>(v));
  v = fmaxf(v, dpp_move<0x140>(v));
  return fmaxf(v, swap_rows16(v));
}

// copy a row-major [rows x 64] fp32 matrix from global into an LD-strided LDS tile.
__device__ __forceinline__ void load_matrix_lds(float* dst_lds, const float* __restrict__ src, int rows, int tid, int nthreads) {
  for (int idx = tid; idx < rows * (H / 4); idx += nthreads) {
    const int row = idx / (H / 4), c4 = idx % (H / 4);
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + row * H + c4 * 4);
    *reinterpret_cast<f32x4*>(dst_lds + row * LD + c4 * 4) = v;
  }
}
// same, but stores the transpose: dst[c][r] = src[r][c]  (src is [64 x 64]).
// loads are issued as one batch (16 B per lane each) before any LDS store.
__device__ __forceinline__ void load_matrix_lds_t(float* dst_lds, const float* __restrict__ src, int tid, int nthreads) {
  f32x4 v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = tid + j * nthreads;
    v[j] = (idx < H * H / 4) ? *reinterpret_cast<const f32x4*>(src + idx * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = tid + j * nthreads;
    if (idx < H * H / 4) {
      const int row = idx / (H / 4), c4 = (idx % (H / 4)) * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) dst_lds[(c4 + k) * LD + row] = v[j][k];
    }
  }
}

// In-kernel launch timing (bench.py's roofline): with a non-NULL `wg_clock` [gridDim.x][2] every workgroup stores the device
// wall clock (100 MHz constant-rate counter) when it starts and when it has finished; the launch lasted from the smallest start
// to the largest end.  Works inside a replayed HIP graph (where HIP events cannot bracket a launch), costs two 8-byte stores per
// workgroup and no register that lives across the kernel.
__device__ __forceinline__ void wg_clock_start(long long* wg_clock) {
#ifndef IS_NO_WG_CLOCK      // (-DIS_NO_WG_CLOCK: A/B builds that measure what the stamps cost)
  if (wg_clock != nullptr && threadIdx.x == 0) wg_clock[2 * blockIdx.x] = (long long)wall_clock64();
#endif
}
__device__ __forceinline__ void wg_clock_end(long long* wg_clock) {
#ifdef IS_NO_WG_CLOCK
  return;
#endif
  if (wg_clock != nullptr) {      // kernel-uniform
    __syncthreads();
    if (threadIdx.x == 0) wg_clock[2 * blockIdx.x + 1] = (long long)wall_clock64();
  }
}

// Generic staged copy: element i (0 <= i < COUNT) is fetched by `load(i)` and placed by `store(i, v)`;
// each of the NT threads issues all of its loads before its first store (one memory round trip).
template <int COUNT, int NT, typename LoadF, typename StoreF>
__device__ __forceinline__ void staged_copy(int tid, LoadF load, StoreF store) {
  constexpr int PER = (COUNT + NT - 1) / NT;
  float v[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = tid + j * NT;
    v[j] = (i < COUNT) ? load(i) : 0.0f;
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = tid + j * NT;
    if (i < COUNT) store(i, v[j]);
  }
}

}  // namespace is
