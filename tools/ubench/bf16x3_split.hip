// A 3-term bf16 split of an fp32 product on the bf16 matrix pipe, PRICED, not used (micro-benchmark only -- the
// product path and the bench dtype stay exact fp32).
//
//   a = a_hi + a_mid + a_lo (three bf16, round to nearest: 24 significant bits together), b likewise;
//   a b ~= a_hi b_hi + (a_hi b_mid + a_mid b_hi) + (a_mid b_mid + a_hi b_lo + a_lo b_hi)       -- 6 bf16 MFMAs, fp32 accumulate;
//   dropped: a_mid b_lo, a_lo b_mid, a_lo b_lo (<= 2^-24 |a b| each).
//
// Part 1 (accuracy): C[16 x 16] = A[16 x 64] B[64 x 16] (the K of the layer kernels' 64 x 64 products), NT random problems, against
// the fp64 result, in units of 2^-24 sum_k |a_k b_k| -- next to the fp32 MFMA chain (tools/ubench/mfma_exact.hip: rms 0.47) and a
// 2-term split (3 MFMAs).  Two accumulation orders of the six terms: small terms first into ONE accumulator / three accumulators
// (hi hi | cross | small) added at the end.
// Part 2 (cycles): one "matrix stage" of the edge kernels -- a wave's 16 x 64 fp32 tile in LDS times a staged 64 x 64 weight tile
// -- as 64 x v_mfma_f32_16x16x4_f32 from LDS operands (the kernels' mm16_rows) against 48 x v_mfma_f32_16x16x32_bf16 with B pre-split into three bf16 planes in LDS and the A tile (i) split in registers by
// bit arithmetic, (ii) split with v_cvt_pk_bf16_f32 (both INSIDE the timed loop), (iii) found pre-split in LDS (what a producer that
// splits once per element -- each dz tile feeds two products -- would leave).  1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/bf16x3_split.hip -o tools/ubench/bf16x3_split && tools/ubench/bf16x3_split
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int K = 64, NT = 4096, LD = 68;

__device__ __forceinline__ unsigned short bf16_rn(float v) {      // round to nearest even (finite inputs)
  unsigned u = __builtin_bit_cast(unsigned, v);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
struct Split { unsigned short hi, mid, lo; };
__device__ __forceinline__ Split split3(float v) {
  Split s;
  s.hi = bf16_rn(v);
  const float r1 = v - bf16_f(s.hi);      // exact
  s.mid = bf16_rn(r1);
  const float r2 = r1 - bf16_f(s.mid);    // exact
  s.lo = bf16_rn(r2);
  return s;
}
// the same split of EIGHT values with the hardware's packed conversion (v_cvt_pk_bf16_f32, round to nearest even): per pair one
// conversion, two unpack operations (hi half << 16, and-mask) and two subtractions per level -- 5.5 VALU operations per element
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3x8(const float (&v)[8], s16x8& hi, s16x8& mid, s16x8& lo) {
  unsigned* ph = reinterpret_cast<unsigned*>(&hi);
  unsigned* pm = reinterpret_cast<unsigned*>(&mid);
  unsigned* pl = reinterpret_cast<unsigned*>(&lo);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x2 a = {v[2 * j], v[2 * j + 1]};
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2));
    f32x2 r1 = {a[0] - __builtin_bit_cast(float, h << 16), a[1] - __builtin_bit_cast(float, h & 0xFFFF0000u)};
    const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
    f32x2 r2 = {r1[0] - __builtin_bit_cast(float, m << 16), r1[1] - __builtin_bit_cast(float, m & 0xFFFF0000u)};
    const unsigned l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
    ph[j] = h; pm[j] = m; pl[j] = l;
  }
}
__device__ __forceinline__ f32x4 mfma_bf16(const s16x8& a, const s16x8& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---- part 1: accuracy -------------------------------------------------------------------------------------------------------
__global__ void acc_kernel(const float* A, const float* B, float* C32, float* C6a, float* C6b, float* C3) {
  const int p = blockIdx.x, lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  const float* a = A + (size_t)p * 16 * K;      // [16][K]
  const float* b = B + (size_t)p * K * 16;      // [K][16]
  f32x4 c32 = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 4) c32 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r * K + k0 + q], b[(k0 + q) * 16 + r], c32, 0, 0, 0);
  // bf16 fragments: lane (r, q), element j of k-block kb <-> k = 32 kb + 8 q + j, for A (row r) and B (column r) alike
  s16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
  for (int kb = 0; kb < 2; ++kb)
    for (int j = 0; j < 8; ++j) {
      const int k = 32 * kb + 8 * q + j;
      const Split sa = split3(a[r * K + k]), sb = split3(b[k * 16 + r]);
      ah[kb][j] = (short)sa.hi; am[kb][j] = (short)sa.mid; al[kb][j] = (short)sa.lo;
      bh[kb][j] = (short)sb.hi; bm[kb][j] = (short)sb.mid; bl[kb][j] = (short)sb.lo;
    }
  f32x4 one = {0.f, 0.f, 0.f, 0.f}, hh = one, cross = one, small = one, two = one;
  for (int kb = 0; kb < 2; ++kb) {      // (a) ONE accumulator, smallest terms first
    one = mfma_bf16(al[kb], bh[kb], one);
    one = mfma_bf16(ah[kb], bl[kb], one);
    one = mfma_bf16(am[kb], bm[kb], one);
  }
  for (int kb = 0; kb < 2; ++kb) { one = mfma_bf16(am[kb], bh[kb], one); one = mfma_bf16(ah[kb], bm[kb], one); }
  for (int kb = 0; kb < 2; ++kb) one = mfma_bf16(ah[kb], bh[kb], one);
  for (int kb = 0; kb < 2; ++kb) {      // (b) three accumulators by magnitude class, added small -> large at the end
    small = mfma_bf16(al[kb], bh[kb], small); small = mfma_bf16(ah[kb], bl[kb], small); small = mfma_bf16(am[kb], bm[kb], small);
    cross = mfma_bf16(am[kb], bh[kb], cross); cross = mfma_bf16(ah[kb], bm[kb], cross);
    hh = mfma_bf16(ah[kb], bh[kb], hh);
    two = mfma_bf16(am[kb], bh[kb], two); two = mfma_bf16(ah[kb], bm[kb], two);
  }
  for (int kb = 0; kb < 2; ++kb) two = mfma_bf16(ah[kb], bh[kb], two);      // 2-term split: hi hi + the two cross terms
  for (int t = 0; t < 4; ++t) {
    const size_t o = (size_t)p * 256 + (4 * q + t) * 16 + r;
    C32[o] = c32[t]; C6a[o] = one[t]; C6b[o] = (small[t] + cross[t]) + hh[t]; C3[o] = two[t];
  }
}

// ---- part 2: cycles ---------------------------------------------------------------------------------------------------------
// one wave = one 16 x 64 A tile (fp32, LDS, row stride LD) x the workgroup's 64 x 64 weight tile -> acc[4] (16 x 64), ITER times
template <int MODE>
__global__ void stage_kernel(const float* W, float* out, long long* cycles, int iters) {
  __shared__ float w_f32[64 * LD];                         // W[j][k] (mm16_rows' layout)
  __shared__ __attribute__((aligned(16))) unsigned short w_bf[3][64 * 72];      // planes hi / mid / lo: [j][k], row stride 72 bf16 = 144 B
  __shared__ float a_tiles[8][16 * LD];
  __shared__ __attribute__((aligned(16))) unsigned short a_bf[8][3][16 * 72];      // MODE 3: the A tile as it would sit in LDS had its PRODUCER split it
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  for (int i = tid; i < 64 * 64; i += blockDim.x) {
    const int j = i >> 6, k = i & 63;
    const float v = W[i];
    w_f32[j * LD + k] = v;
    const Split s = split3(v);
    w_bf[0][j * 72 + k] = s.hi; w_bf[1][j * 72 + k] = s.mid; w_bf[2][j * 72 + k] = s.lo;
  }
  float* at = a_tiles[wave];
  for (int i = lane; i < 16 * 64; i += 64) at[(i >> 6) * LD + (i & 63)] = 0.01f * (float)((i * 37 + wave * 11) % 97) - 0.4f;
  for (int i = lane; i < 16 * 64; i += 64) {
    const Split sp = split3(at[(i >> 6) * LD + (i & 63)]);
    a_bf[wave][0][(i >> 6) * 72 + (i & 63)] = sp.hi; a_bf[wave][1][(i >> 6) * 72 + (i & 63)] = sp.mid; a_bf[wave][2][(i >> 6) * 72 + (i & 63)] = sp.lo;
  }
  __syncthreads();
  f32x4 acc[4];
  for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {      // exact fp32: quarter q walks k in [16 q, 16 q + 16) (mm16_rows)
#pragma unroll
      for (int s = 0; s < 16; s += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(at + r * LD + q * 16 + s);
        f32x4 b[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) b[nt] = *reinterpret_cast<const f32x4*>(w_f32 + (nt * 16 + r) * LD + q * 16 + s);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[nt][j], acc[nt], 0, 0, 0);
      }
    } else {              // 3-term split: the A tile is read as fp32 and split HERE (its cost is inside the loop)
      s16x8 ah[2], am[2], al[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        if (MODE == 3) {
          const int o = r * 72 + 32 * kb + 8 * q;
          ah[kb] = *reinterpret_cast<const s16x8*>(&a_bf[wave][0][o]);
          am[kb] = *reinterpret_cast<const s16x8*>(&a_bf[wave][1][o]);
          al[kb] = *reinterpret_cast<const s16x8*>(&a_bf[wave][2][o]);
          continue;
        }
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(at + r * LD + 32 * kb + 8 * q);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(at + r * LD + 32 * kb + 8 * q + 4);
        if (MODE == 1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const Split s = split3(j < 4 ? v0[j] : v1[j - 4]);
            ah[kb][j] = (short)s.hi; am[kb][j] = (short)s.mid; al[kb][j] = (short)s.lo;
          }
        } else {
          const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          split3x8(v, ah[kb], am[kb], al[kb]);
        }
      }
      // six terms, smallest first, the four column blocks interleaved (four independent accumulator chains, as in the fp32 form)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        s16x8 bh[4], bm[4], bl[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int o = (nt * 16 + r) * 72 + 32 * kb + 8 * q;
          bh[nt] = *reinterpret_cast<const s16x8*>(&w_bf[0][o]);
          bm[nt] = *reinterpret_cast<const s16x8*>(&w_bf[1][o]);
          bl[nt] = *reinterpret_cast<const s16x8*>(&w_bf[2][o]);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = mfma_bf16(al[kb], bh[nt], acc[nt]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = mfma_bf16(ah[kb], bl[nt], acc[nt]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = mfma_bf16(am[kb], bm[nt], acc[nt]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = mfma_bf16(am[kb], bh[nt], acc[nt]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = mfma_bf16(ah[kb], bm[nt], acc[nt]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = mfma_bf16(ah[kb], bh[nt], acc[nt]);
      }
    }
    // the next iteration's tile depends on this one's result (no hoisting of the split or the operand reads out of the loop)
    at[r * LD + q] = acc[0][0] * 1e-30f + at[r * LD + q];
    if (MODE == 3) a_bf[wave][2][r * 72 + q] = (unsigned short)(a_bf[wave][2][r * 72 + q] + (acc[0][0] == 12345.f ? 1 : 0));
    __builtin_amdgcn_wave_barrier();
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int nt = 0; nt < 4; ++nt) s += acc[nt][0] + acc[nt][1] + acc[nt][2] + acc[nt][3];
  out[blockIdx.x * blockDim.x + tid] = s;
  if (tid == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

static double urand() { return (rand() + 0.5) / ((double)RAND_MAX + 1.0); }
static float gauss() { return (float)(std::sqrt(-2.0 * std::log(urand())) * std::cos(6.283185307179586 * urand())); }

template <int MODE>
static void time_stage(const char* name, const float* dW) {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * sizeof(float));
  (void)hipMalloc(&cyc, 8);
  const int iters = 2000;
  for (int wps : {1, 2}) {
    const int threads = 256 * wps;
    stage_kernel<MODE><<<256, threads>>>(dW, out, cyc, 10);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    stage_kernel<MODE><<<256, threads>>>(dW, out, cyc, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s waves/SIMD %d: %8.0f cycles per 16x64x64 stage and wave, %7.1f ns per stage and SIMD\n", name, wps, (double)c / iters,
           ms * 1e6 / ((double)iters * wps));
  }
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  srand(7);
  std::vector<float> A((size_t)NT * 16 * K), B((size_t)NT * K * 16);
  for (auto& v : A) v = gauss();
  for (auto& v : B) v = gauss();
  const size_t nc = (size_t)NT * 256;
  std::vector<float> C[4];
  float *dA, *dB, *dC[4];
  (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4);
  for (int i = 0; i < 4; ++i) { C[i].resize(nc); (void)hipMalloc(&dC[i], nc * 4); }
  (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  acc_kernel<<<NT, 64>>>(dA, dB, dC[0], dC[1], dC[2], dC[3]);
  for (int i = 0; i < 4; ++i) (void)hipMemcpy(C[i].data(), dC[i], nc * 4, hipMemcpyDeviceToHost);
  const char* names[4] = {"fp32 MFMA chain (16 x 16x16x4 f32)", "bf16 x 3, six terms, ONE accumulator", "bf16 x 3, six terms, three accumulators",
                          "bf16 x 2, three terms"};
  double rms[4] = {0, 0, 0, 0}, mx[4] = {0, 0, 0, 0}, bias[4] = {0, 0, 0, 0};
  for (int p = 0; p < NT; ++p)
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double ref = 0, scale = 0;
        for (int k = 0; k < K; ++k) {
          const double pab = (double)A[((size_t)p * 16 + i) * K + k] * (double)B[((size_t)p * K + k) * 16 + j];
          ref += pab; scale += std::fabs(pab);
        }
        const double unit = scale * std::ldexp(1.0, -24);
        for (int m = 0; m < 4; ++m) {
          const double e = ((double)C[m][(size_t)p * 256 + i * 16 + j] - ref) / unit;
          rms[m] += e * e; bias[m] += e; if (std::fabs(e) > mx[m]) mx[m] = std::fabs(e);
        }
      }
  printf("error of C = A[16x64] B[64x16] against fp64, in units of 2^-24 sum_k |a_k b_k| (%d outputs, N(0,1) operands):\n", (int)nc);
  for (int m = 0; m < 4; ++m) printf("  %-42s rms %7.3f   max %8.3f   mean %+.4f\n", names[m], std::sqrt(rms[m] / nc), mx[m], bias[m] / nc);
  std::vector<float> W(64 * 64);
  for (auto& v : W) v = 0.125f * gauss();
  float* dW; (void)hipMalloc(&dW, W.size() * 4);
  (void)hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
  time_stage<0>("fp32: 64 x mfma_16x16x4_f32, LDS operands", dW);
  time_stage<1>("bf16 x 3: software split + 48 x mfma_16x16x32_bf16", dW);
  time_stage<2>("bf16 x 3: v_cvt_pk_bf16_f32 split + 48 x bf16 mfma", dW);
  time_stage<3>("bf16 x 3: A pre-split (3 planes in LDS) + 48 x bf16 mfma", dW);
  return 0;
}
