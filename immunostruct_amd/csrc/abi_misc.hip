// Library identification + a tiny MFMA layout self-test used by the GPU test-suite.
#include "common.h"
#include <stdio.h>

namespace is {

// ---- the calling thread's last failure (common.h: fail / launch_status; C ABI: is_last_error_string) ----
static thread_local char g_last_error[320] = "";

int fail(const char* entry, int code) {
  const char* why = code == -22 ? "invalid argument (a size, pointer or combination the entry point does not accept)"
                  : code == -38 ? "not covered by this build of the kernel (the caller takes the other form)"
                  : code == -5  ? "the device call failed"
                                : "failed";
  snprintf(g_last_error, sizeof(g_last_error), "%s: %s (%d)", entry, why, code);
  return code;
}

int launch_status(const char* entry) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  snprintf(g_last_error, sizeof(g_last_error), "%s: %s: %s (-5)", entry, hipGetErrorName(e), hipGetErrorString(e));
  return -5;
}

// out[32 x 64] = A[32 x 64] * W[64 x 64]^T through the same mm_rows path the kernels use.
__global__ __launch_bounds__(64) void mfma_selftest_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                           float* __restrict__ out) {
  __shared__ float a_lds[TE * LD];
  __shared__ float w_lds[H * LD];
  const int lane = threadIdx.x;
  load_matrix_lds(a_lds, A, TE, lane, 64);
  load_matrix_lds(w_lds, W, H, lane, 64);
  __syncthreads();
  f32x16 acc[2];
  for (int nt = 0; nt < 2; ++nt)
    for (int t = 0; t < 16; ++t) acc[nt][t] = 0.0f;
  mm_rows<2, H>(acc, a_lds, w_lds, lane);
  const int r = lane & 31, hf = lane >> 5;
  for (int nt = 0; nt < 2; ++nt)
    for (int t = 0; t < 16; ++t) out[tile_row(t, hf) * H + nt * 32 + r] = acc[nt][t];
}

// outer[64 x 64] = G[32 x 64]^T * M[32 x 64] through mm_outer.
__global__ __launch_bounds__(64) void mfma_outer_selftest_kernel(const float* __restrict__ G, const float* __restrict__ M,
                                                                 float* __restrict__ out) {
  __shared__ float g_lds[TE * LD];
  __shared__ float m_lds[TE * LD];
  const int lane = threadIdx.x;
  load_matrix_lds(g_lds, G, TE, lane, 64);
  load_matrix_lds(m_lds, M, TE, lane, 64);
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int t = 0; t < 16; ++t) acc[a][b][t] = 0.0f;
  mm_outer<2, 2>(acc, g_lds, m_lds, lane);
  const int r = lane & 31, hf = lane >> 5;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int t = 0; t < 16; ++t) out[(a * 32 + tile_row(t, hf)) * H + b * 32 + r] = acc[a][b][t];
}

}  // namespace is

extern "C" int is_version(void) { return 100; }  // 0.1.0

// text of the calling thread's most recent failed entry point ("" when none failed yet): the entry point's name, the reason behind
// its code and, for a failed launch, HIP's own error name and string.  The pointer stays valid for the thread's lifetime; the text
// changes with the thread's next failure (a successful call leaves it alone).
extern "C" const char* is_last_error_string(void) { return is::g_last_error; }

extern "C" int is_mfma_selftest(const float* A, const float* W, float* out, void* stream) {
  hipLaunchKernelGGL(is::mfma_selftest_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), A, W, out);
  return is::launch_status(__func__);
}

extern "C" int is_mfma_outer_selftest(const float* G, const float* M, float* out, void* stream) {
  hipLaunchKernelGGL(is::mfma_outer_selftest_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), G, M, out);
  return is::launch_status(__func__);
}


// Debug aid: write the device wall clock (constant-rate counter) to *slot -- one tiny launch that can be captured in
// a HIP graph to time-stamp points of a replayed step without a profiler attached.
namespace is {
__global__ void timestamp_kernel(long long* slot) { *slot = (long long)wall_clock64(); }
}
extern "C" int is_debug_timestamp(long long* slot, void* stream) {
  hipLaunchKernelGGL(is::timestamp_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), slot);
  return is::launch_status(__func__);
}


// Debug aid (tools/dp_overlap_emulation.py): a stand-in for an RCCL all-reduce that behaves like one towards the compute
// stream -- `grid` persistent workgroups (RCCL: one per channel, NCCL_MIN/MAX_NCHANNELS) that hold their CU slots for
// `ticks` of the 100 MHz device clock and stream their slice of the bucket `passes` times in place (16-byte non-temporal loads
// and stores of every element: the local HBM traffic of reduce-scatter + all-gather), then wait out the rest of the duration.
// The data is unchanged.  `elapsed` (NULL or [grid] int64): ticks every workgroup spent streaming -- when that exceeds `ticks`
// the stand-in ran longer than asked for.  A sleeping one-thread kernel (torch.cuda._sleep) holds neither slots nor bandwidth.
namespace is {
__global__ __launch_bounds__(512) void emulated_collective_kernel(float* buf, long long n, int passes, long long ticks,
                                                                  long long* elapsed) {
  const long long t0 = (long long)wall_clock64();
  const long long n4 = n / 4;
  const long long per = (n4 + gridDim.x - 1) / gridDim.x;
  const long long lo = (long long)blockIdx.x * per, hi = min(n4, lo + per);
  f32x4* b4 = reinterpret_cast<f32x4*>(buf);
  for (int p = 0; p < passes; ++p)
    for (long long i = lo + threadIdx.x; i < hi; i += 4 * 512) {      // four 16-byte loads in flight per thread
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) if (i + u * 512 < hi) v[u] = __builtin_nontemporal_load(b4 + i + u * 512);
#pragma unroll
      for (int u = 0; u < 4; ++u) if (i + u * 512 < hi) __builtin_nontemporal_store(v[u], b4 + i + u * 512);
    }
  if (elapsed != nullptr && threadIdx.x == 0) elapsed[blockIdx.x] = (long long)wall_clock64() - t0;      // the streaming part alone
  while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}
extern "C" int is_debug_emulated_collective(float* buf, long long n, int grid, int passes, long long ticks, long long* elapsed,
                                            void* stream) {
  if (grid <= 0 || n < 0 || (reinterpret_cast<uintptr_t>(buf) & 15) != 0) return is::fail(__func__, -22);
  hipLaunchKernelGGL(is::emulated_collective_kernel, dim3(grid), dim3(512), 0, static_cast<hipStream_t>(stream), buf, n, passes, ticks,
                     elapsed);
  return is::launch_status(__func__);
}


// Measurement aid (bench.py `roofline.hbm_view.measured_peak`, SURVEY.md section 8(d): "also report against a measured
// hipMemcpy / triad ceiling"): dst[0, n) = src[0, n) as a grid-stride loop of 16-byte non-temporal accesses, four loads in
// flight per thread.  n = number of 16-byte words.  Bytes moved = 32 n (read + write).
namespace is {
__global__ __launch_bounds__(256) void stream_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long long n) {
  const long long stride = (long long)gridDim.x * 256 * 4;
  for (long long i = (long long)blockIdx.x * 256 * 4 + threadIdx.x; i < n; i += stride) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + u * 256 < n) v[u] = __builtin_nontemporal_load(src + i + u * 256);
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + u * 256 < n) __builtin_nontemporal_store(v[u], dst + i + u * 256);
  }
}
}
extern "C" int is_debug_stream_copy(const void* src, void* dst, long long n16, int grid, void* stream) {
  if (grid <= 0 || n16 < 0 || ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) != 0) return is::fail(__func__, -22);
  hipLaunchKernelGGL(is::stream_copy_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const is::f32x4*>(src), static_cast<is::f32x4*>(dst), n16);
  return is::launch_status(__func__);
}


// ---------------------------------------------------------------------------------------------
// The random tensors of one train step -- scaled dropout keep-masks (nn.Dropout in training mode: reference
// models/hybrid_models.py:277-295) and the reparameterisation noise (torch.randn_like, hybrid_models.py:301-304) -- as ONE launch
// from a device-resident generator state, for the captured step: no host-side generator, so no seed / offset fill launches in
// front of every graph replay, and one launch instead of three.  Philox4x32-10 (Salmon et al., SC'11: the generator behind
// torch's and cuRAND's default streams) keyed by the seed, counter = (element quad, job, step): every value is a function of
// (seed, step, job, element) alone -- independent of the grid.  state: [0] seed, [1] step counter (advanced by the launch's last
// workgroup), [2] ticket.
namespace is {
struct RandJob { float* out; long long n; int kind; float p; };      // kind 0: N(0, 1); 1: keep-mask of Dropout(p), scaled by 1 / (1 - p)
constexpr int RAND_MAX_JOBS = 8;
struct RandBatch { RandJob job[RAND_MAX_JOBS]; };

__device__ inline void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned out[4]) {
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ __launch_bounds__(256) void step_random_kernel(RandBatch batch, unsigned long long* __restrict__ state) {
  const RandJob& J = batch.job[blockIdx.y];
  const unsigned long long seed = state[0], step = state[1];
  const long long quads = (J.n + 3) >> 2;
  const float keep = J.p < 1.0f ? 1.0f / (1.0f - J.p) : 0.0f;
  for (long long qd = (long long)blockIdx.x * 256 + threadIdx.x; qd < quads; qd += (long long)gridDim.x * 256) {
    unsigned r[4];
    philox4x32_10((unsigned)qd, (unsigned)(qd >> 32) ^ ((unsigned)blockIdx.y << 24), (unsigned)step, (unsigned)(step >> 32), (unsigned)seed,
                  (unsigned)(seed >> 32), r);
    float u[4], v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = ((float)(r[i] >> 8) + 0.5f) * (1.0f / 16777216.0f);      // (0, 1), 24 bits
    if (J.kind == 0) {
#pragma unroll
      for (int i = 0; i < 4; i += 2) {      // Box-Muller
        const float rad = sqrtf(-2.0f * logf(u[i]));
        float sn, cs;
        sincosf(6.283185307179586f * u[i + 1], &sn, &cs);
        v[i] = rad * cs;
        v[i + 1] = rad * sn;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = u[i] >= J.p ? keep : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (4 * qd + i < J.n) J.out[4 * qd + i] = v[i];
  }
  // every thread of the workgroup has read (and used) the step counter before the barrier below; the workgroup that takes the last
  // ticket advances it for the next launch.  No fence: nothing is handed from workgroup to workgroup -- the ticket is a device-scope
  // atomic, and the counter's new value is only read by the NEXT launch (an agent-scope fence here is an L2 write-back +
  // invalidate per workgroup: it was most of this launch's 12 us)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long total = (unsigned long long)gridDim.x * gridDim.y;
    if (atomicAdd(&state[2], 1ULL) == total - 1) {
      state[1] = step + 1;
      state[2] = 0;
    }
  }
}
}  // namespace is

// jobs: host array of njobs (<= 8) records { float* out; long long n; int kind; float p; }; state: 3 x uint64 of device memory
extern "C" int is_step_random(const void* jobs, int njobs, unsigned long long* state, void* stream) {
  if (njobs <= 0 || njobs > is::RAND_MAX_JOBS || state == nullptr) return is::fail(__func__, -22);
  is::RandBatch batch;
  const is::RandJob* src = static_cast<const is::RandJob*>(jobs);
  long long maxn = 0;
  for (int i = 0; i < njobs; ++i) {
    if (src[i].n <= 0 || src[i].out == nullptr || src[i].kind < 0 || src[i].kind > 1 || !(src[i].p >= 0.0f && src[i].p <= 1.0f)) return is::fail(__func__, -22);
    batch.job[i] = src[i];
    maxn = src[i].n > maxn ? src[i].n : maxn;
  }
  int blocks = (int)((maxn / 4 + 255) / 256);
  blocks = blocks < 1 ? 1 : (blocks > 64 ? 64 : blocks);
  hipLaunchKernelGGL(is::step_random_kernel, dim3(blocks, njobs), dim3(256), 0, static_cast<hipStream_t>(stream), batch, state);
  return is::launch_status(__func__);
}
