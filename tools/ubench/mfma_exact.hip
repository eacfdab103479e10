// How exact is v_mfma_f32_16x16x4_f32 (and 32x32x2)?  C[16 x 16] = A[16 x K] B[K x 16], K = 64, random operands:
//   (a) on the MFMA pipe, k walked 4 (2) at a time in ascending order, accumulator chained;
//   (b) the same products as a chain of fp32 FMAs in ascending k on the vector ALU (what a CPU / torch fp32 dot product of this
//       length is, up to its blocking);
// both against the fp64 result.  Prints the RMS and maximum error of each in units of 2^-24 * sum_k |a_k b_k| (the scale a
// correctly rounded chain's error is measured in) and whether (a) is bit-equal to one of the candidate evaluation orders.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_exact.hip -o tools/ubench/mfma_exact && tools/ubench/mfma_exact
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int K = 64, NT = 4096;      // NT independent 16 x 16 problems

__global__ void k16(const float* A, const float* B, float* C_mfma, float* C_fma) {
  const int p = blockIdx.x, lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  const float* a = A + (size_t)p * 16 * K;      // [16][K]
  const float* b = B + (size_t)p * K * 16;      // [K][16]
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 4)             // k-slot q holds k0 + q
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r * K + k0 + q], b[(k0 + q) * 16 + r], acc, 0, 0, 0);
  for (int t = 0; t < 4; ++t) {
    const int row = 4 * q + t;
    C_mfma[(size_t)p * 256 + row * 16 + r] = acc[t];
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = __builtin_fmaf(a[row * K + k], b[k * 16 + r], s);
    C_fma[(size_t)p * 256 + row * 16 + r] = s;
  }
}

__global__ void k32(const float* A, const float* B, float* C_mfma) {      // 32 x 32 x 2: A [32][K], B [K][32]
  const int p = blockIdx.x, lane = threadIdx.x, r = lane & 31, hf = lane >> 5;
  const float* a = A + (size_t)p * 32 * K;
  const float* b = B + (size_t)p * K * 32;
  f32x16 acc;
  for (int t = 0; t < 16; ++t) acc[t] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r * K + k0 + hf], b[(k0 + hf) * 32 + r], acc, 0, 0, 0);
  for (int t = 0; t < 16; ++t) C_mfma[(size_t)p * 1024 + ((t & 3) + 8 * (t >> 2) + 4 * hf) * 32 + r] = acc[t];
}

static double urand() { return (rand() + 0.5) / ((double)RAND_MAX + 1.0); }
static float gauss() { return (float)(std::sqrt(-2.0 * std::log(urand())) * std::cos(6.283185307179586 * urand())); }

int main() {
  srand(7);
  {
    std::vector<float> A((size_t)NT * 16 * K), B((size_t)NT * K * 16), Cm((size_t)NT * 256), Cf((size_t)NT * 256);
    for (auto& v : A) v = gauss();
    for (auto& v : B) v = gauss();
    float *dA, *dB, *dCm, *dCf;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dCm, Cm.size() * 4); hipMalloc(&dCf, Cf.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    k16<<<NT, 64>>>(dA, dB, dCm, dCf);
    hipMemcpy(Cm.data(), dCm, Cm.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(Cf.data(), dCf, Cf.size() * 4, hipMemcpyDeviceToHost);
    double sm = 0, sf = 0, mm = 0, mf = 0, bias_m = 0, bias_f = 0;
    long eq_chain = 0, eq_group = 0, eq_pair = 0, n = 0;
    for (int p = 0; p < NT; ++p)
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          double ref = 0, scale = 0;
          float chain = 0.f, grp = 0.f, pr = 0.f;
          for (int k = 0; k < K; ++k) {
            const double pab = (double)A[((size_t)p * 16 + i) * K + k] * (double)B[((size_t)p * K + k) * 16 + j];
            ref += pab; scale += std::fabs(pab);
            chain = std::fmaf(A[((size_t)p * 16 + i) * K + k], B[((size_t)p * K + k) * 16 + j], chain);
          }
          for (int k0 = 0; k0 < K; k0 += 4) {      // candidates: each instruction adds an exactly-formed group of 4 (2 + 2) products
            double g4 = 0; double g2a = 0, g2b = 0;
            for (int u = 0; u < 4; ++u) {
              const double pab = (double)A[((size_t)p * 16 + i) * K + k0 + u] * (double)B[((size_t)p * K + k0 + u) * 16 + j];
              g4 += pab; (u < 2 ? g2a : g2b) += pab;
            }
            grp = (float)((double)grp + g4);
            pr = (float)((double)(float)((double)pr + g2a) + g2b);
          }
          const float m = Cm[(size_t)p * 256 + i * 16 + j], f = Cf[(size_t)p * 256 + i * 16 + j];
          const double u = scale * std::ldexp(1.0, -24);
          const double em = (m - ref) / u, ef = (f - ref) / u;
          sm += em * em; sf += ef * ef; mm = std::fmax(mm, std::fabs(em)); mf = std::fmax(mf, std::fabs(ef));
          bias_m += em; bias_f += ef;
          eq_chain += (m == chain); eq_group += (m == grp); eq_pair += (m == pr); ++n;
        }
    printf("16x16x4, K = %d, %ld outputs: error in units of 2^-24 * sum|a b|\n", K, n);
    printf("  MFMA      : rms %.4f  max %.3f  mean %+.4f\n", std::sqrt(sm / n), mm, bias_m / n);
    printf("  FMA chain : rms %.4f  max %.3f  mean %+.4f\n", std::sqrt(sf / n), mf, bias_f / n);
    printf("  MFMA bit-equal to: ascending fp32 FMA chain %.1f %%, one rounding per instruction (4 exact products) %.1f %%, "
           "two roundings per instruction (2 + 2) %.1f %%\n", 100.0 * eq_chain / n, 100.0 * eq_group / n, 100.0 * eq_pair / n);
  }
  {
    const int P = 1024;
    std::vector<float> A((size_t)P * 32 * K), B((size_t)P * K * 32), Cm((size_t)P * 1024);
    for (auto& v : A) v = gauss();
    for (auto& v : B) v = gauss();
    float *dA, *dB, *dCm;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dCm, Cm.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    k32<<<P, 64>>>(dA, dB, dCm);
    hipMemcpy(Cm.data(), dCm, Cm.size() * 4, hipMemcpyDeviceToHost);
    double sm = 0, sc = 0, mm = 0; long eq_chain = 0, n = 0;
    for (int p = 0; p < P; ++p)
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double ref = 0, scale = 0; float chain = 0.f;
          for (int k = 0; k < K; ++k) {
            const double pab = (double)A[((size_t)p * 32 + i) * K + k] * (double)B[((size_t)p * K + k) * 32 + j];
            ref += pab; scale += std::fabs(pab);
            chain = std::fmaf(A[((size_t)p * 32 + i) * K + k], B[((size_t)p * K + k) * 32 + j], chain);
          }
          const float m = Cm[(size_t)p * 1024 + i * 32 + j];
          const double u = scale * std::ldexp(1.0, -24);
          const double em = (m - ref) / u, ec = (chain - ref) / u;
          sm += em * em; sc += ec * ec; mm = std::fmax(mm, std::fabs(em)); eq_chain += (m == chain); ++n;
        }
    printf("32x32x2, K = %d, %ld outputs: MFMA rms %.4f max %.3f | FMA chain rms %.4f | bit-equal to the chain %.1f %%\n", K, n,
           std::sqrt(sm / n), mm, std::sqrt(sc / n), 100.0 * eq_chain / n);
  }
  return 0;
}
