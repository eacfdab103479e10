// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 / 32x32x2 and of plain VALU / transcendental ops on gfx950,
// with 1, 2 or 4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void kern(float* out, long long* cycles, int iters) {
  const int lane = threadIdx.x & 63;
  float a = lane * 0.001f, b = 1.0f + lane * 0.002f;
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
  f32x16 big[2];
  for (int i = 0; i < 2; ++i) for (int t = 0; t < 16; ++t) big[i][t] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {          // 16 MFMA 16x16x4, 4 independent accumulators
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[n], 0, 0, 0);
    } else if (MODE == 1) {   // 8 MFMA 32x32x2, 2 accumulators
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < 2; ++n) big[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[n], 0, 0, 0);
    } else if (MODE == 2) {   // 16 independent FMAs
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int n = 0; n < 8; ++n) v[n] = __builtin_fmaf(v[n], b, a);
    } else if (MODE == 3) {   // 16 exp
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int n = 0; n < 8; ++n) v[n] = __builtin_amdgcn_exp2f(v[n]);
    } else if (MODE == 5) {   // 16 MFMA 16x16x16 bf16, 4 independent accumulators
      const s16x4 ab = {(short)lane, (short)(lane + 1), (short)(lane + 2), (short)(lane + 3)};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab, ab, acc[n], 0, 0, 0);
    } else if (MODE == 6) {   // 16 MFMA 16x16x4 on ONE accumulator: a dependent chain (SrcC = the previous vDst)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0], 0, 0, 0);
    } else if (MODE == 4) {   // 8 x (mfma + 4 fma): overlap test
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        acc[n & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[n & 3], 0, 0, 0);
        v[0] = __builtin_fmaf(v[0], b, a); v[1] = __builtin_fmaf(v[1], b, a);
        v[2] = __builtin_fmaf(v[2], b, a); v[3] = __builtin_fmaf(v[3], b, a);
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 2; ++i) for (int t = 0; t < 16; ++t) s += big[i][t];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, int ops_per_iter) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * sizeof(float) * 4);
  hipMalloc(&cyc, 8);
  const int iters = 2000;
  for (int waves_per_simd : {1, 2, 4}) {
    const int threads = 64 * 4 * waves_per_simd;     // one workgroup per CU fills waves_per_simd on each SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<MODE><<<256, threads>>>(out, cyc, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<MODE><<<256, threads>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s waves/SIMD %d: %8.1f memtime-ticks per op per wave, kernel %.3f ms => %.2f ns per op per SIMD\n", name, waves_per_simd,
           (double)c / (iters * (double)ops_per_iter), ms, ms * 1e6 / (iters * (double)ops_per_iter * waves_per_simd));
  }
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("mfma_f32_16x16x4", 16);
  run<1>("mfma_f32_32x32x2", 8);
  run<2>("v_fma_f32", 16);
  run<3>("v_exp_f32", 16);
  run<4>("mfma16 + 4 fma (per group)", 8);
  run<5>("mfma_f32_16x16x16_bf16", 16);
  run<6>("mfma_f32_16x16x4, one acc", 16);
  return 0;
}
