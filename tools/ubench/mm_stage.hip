// Micro-benchmark of one "MM stage" of the edge kernels: acc = A(16x64, LDS) * W^T(64x64, LDS) on
// v_mfma_f32_16x16x4_f32 with different epilogues; cycles per stage per wave at 1 / 2 / 3 waves per SIMD.
#include "../../immunostruct_amd/csrc/common.h"
#include <stdio.h>
using namespace is;

template <int MODE>
__global__ __launch_bounds__(1024) void kern(const float* w, float* out, long long* cycles, int iters) {
  extern __shared__ float smem[];
  float* wl = smem;                       // 64 x LD
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* act = smem + H * LD + wave * TE16 * LD;
  for (int i = tid; i < H * LD; i += blockDim.x) wl[i] = w[i % (H * H)] * 0.01f;
  for (int i = lane; i < TE16 * LD; i += 64) act[i] = 0.001f * i;
  __syncthreads();
  const int r = lane & 15, q = lane >> 4;
  float keep = 0.f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    f32x4 acc[4];
    zero_acc4(acc);
    mm16_rows<4, H>(acc, act, wl, lane);
    if (MODE == 0) {            // minimal epilogue: fold into a register
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int t = 0; t < 4; ++t) keep += acc[nt][t];
    } else if (MODE == 1) {     // write back to the LDS tile (dependency chain through LDS like the real kernel)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int t = 0; t < 4; ++t) act[tile16_row(t, q) * LD + nt * 16 + r] = acc[nt][t] * 0.01f;
    } else if (MODE == 2) {     // + SiLU
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int t = 0; t < 4; ++t) act[tile16_row(t, q) * LD + nt * 16 + r] = silu_f(acc[nt][t]);
    }
    __builtin_amdgcn_wave_barrier();
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + tid] = keep + act[lane];
  if (tid == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int MODE>
void run(const char* name) {
  float *w, *out; long long* cyc;
  hipMalloc(&w, H * H * 4); hipMemset(w, 0, H * H * 4);
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
  const int iters = 500;
  for (int wps : {1, 2, 3, 4}) {
    const int threads = 256 * wps;
    const size_t lds = (H * LD + 4 * wps * TE16 * LD) * 4;
    hipFuncSetAttribute((const void*)kern<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    kern<MODE><<<256, threads, lds>>>(w, out, cyc, 5);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    kern<MODE><<<256, threads, lds>>>(w, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s waves/SIMD %d: %7.0f ticks per stage (wave 0), %.3f us per stage per SIMD (ideal 64 MFMA = 2080 ticks = ~1.0 us)\n",
           name, wps, (double)c / iters, ms * 1e3 / (iters * (double)wps));
  }
}

int main() {
  run<0>("mm16_rows, register epilogue");
  run<1>("mm16_rows, LDS write-back");
  run<2>("mm16_rows, SiLU + LDS write-back");
  return 0;
}
