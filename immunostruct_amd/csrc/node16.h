// Shared pieces of the node-level MFMA code (egnn_node16.hip and the fused layer kernels): tile dimensions, the
// register-operand matrix products and the layout of the lane-ordered operand packs.
#pragma once
#include "common.h"

namespace is {

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

// same, row tiles mt >= mt_used (wave-uniform) are skipped: a pass over a partly filled stack of row tiles
template <int MT, int KQ, int LDA>
__device__ __forceinline__ void mm16_regB_used(f32x4 (&acc)[MT], const float* a_lds, const f32x4 (&b)[KQ / 4], int lane, int mt_used) {
  const int r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (mt < mt_used) {
#pragma unroll
      for (int g = 0; g < KQ / 4; ++g) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(a_lds + (mt * 16 + r) * LDA + q * KQ + 4 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[g][j], acc[mt], 0, 0, 0);
      }
    }
  }
}

template <int DIN>
struct Node16Dims {
  static constexpr int KV = DIN + 64;                       // valid k of the node-MLP input
  static constexpr int KP = (KV + 15) / 16 * 16;            // padded: 96 (Din 20) or 128 (Din 64)
  static constexpr int LD1 = KP + 4;                        // 100 / 132 (LD/4 odd)
  static constexpr int KQ1 = KP / 4;                        // k per quarter: 24 / 32
};

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

// same, row tiles mt >= mt_used (wave-uniform) are skipped
template <int MT, int KQ, int LDA>
__device__ __forceinline__ void mm16_regBt_used(f32x4 (&acc)[MT], const float* a_lds, const float (&bt)[KQ], int lane, int mt_used) {
  const int r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (mt < mt_used) {
#pragma unroll
      for (int g = 0; g < KQ / 4; ++g) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(a_lds + (mt * 16 + r) * LDA + q * KQ + 4 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bt[4 * g + j], acc[mt], 0, 0, 0);
      }
    }
  }
}

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


}  // namespace is
