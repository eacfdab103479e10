// Multi-tensor Adam / AdamW step (torch.optim.Adam semantics, reference procedures use Adam and AdamW with
// weight_decay 1e-6: train_IEDB_wFT.py:69-74, train_Cancer_wFT.py:76-92) as ONE streaming launch over a chunk
// table: every parameter tensor is cut into chunks of <= 16 Ki elements, one workgroup per chunk, float4 accesses.
// 7 x 4 bytes per parameter of HBM traffic -- the kernel is bandwidth-bound (6.33 M parameters = 177 MB per step).
// The step counter and the learning rate live in device memory, so the launch can be captured in a HIP graph
// and replayed while a scheduler changes the learning rate between replays.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace is { int fail(const char* entry, int code); int launch_status(const char* entry); }      // (common.h: the thread's last failure)

namespace is {

struct AdamChunk {
  float* p;
  const float* g;
  float* m;
  float* v;
  long long n;
};

// state: [0] step count (as float, exact below 2^24), [1] lr / (1 - beta1^t), [2] sqrt(1 - beta2^t)
// hyper: [0] lr, [1] beta1, [2] beta2, [3] eps, [4] weight_decay, [5] decoupled (AdamW) flag, [6] gradient scale
//        (1 = none; 1 / world size when the summed data-parallel gradient bucket is averaged here instead of by its own pass)
__global__ void adam_prepare_kernel(float* __restrict__ state, const float* __restrict__ hyper) {
  const double t = (double)state[0] + 1.0;
  state[0] = (float)t;
  state[1] = (float)((double)hyper[0] / (1.0 - pow((double)hyper[1], t)));
  state[2] = (float)sqrt(1.0 - pow((double)hyper[2], t));
}

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps,
                                         float wd, bool decoupled, float step_size, float bc2s) {
  if (decoupled) p *= (1.0f - lr * wd);
  else if (wd != 0.0f) g += wd * p;
  m += (g - m) * (1.0f - b1);                     // lerp, as torch's fused functor
  v = b2 * v + (1.0f - b2) * g * g;
  const float denom = sqrtf(v) / bc2s + eps;
  p -= step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamChunk* __restrict__ chunks, const float* __restrict__ state,
                                                        const float* __restrict__ hyper) {
  const AdamChunk c = chunks[blockIdx.x];
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4];
  const bool decoupled = hyper[5] != 0.0f;
  const float gs = hyper[6];
  const float step_size = state[1], bc2s = state[2];
  const bool vec = ((reinterpret_cast<uintptr_t>(c.p) | reinterpret_cast<uintptr_t>(c.g) | reinterpret_cast<uintptr_t>(c.m) |
                     reinterpret_cast<uintptr_t>(c.v)) & 15) == 0;
  const long long n4 = vec ? (c.n >> 2) : 0;
  for (long long i = threadIdx.x; i < n4; i += 256) {
    float4 p = reinterpret_cast<float4*>(c.p)[i];
    float4 g = reinterpret_cast<const float4*>(c.g)[i];
    g.x *= gs; g.y *= gs; g.z *= gs; g.w *= gs;
    float4 m = reinterpret_cast<float4*>(c.m)[i];
    float4 v = reinterpret_cast<float4*>(c.v)[i];
    adam_one(p.x, g.x, m.x, v.x, lr, b1, b2, eps, wd, decoupled, step_size, bc2s);
    adam_one(p.y, g.y, m.y, v.y, lr, b1, b2, eps, wd, decoupled, step_size, bc2s);
    adam_one(p.z, g.z, m.z, v.z, lr, b1, b2, eps, wd, decoupled, step_size, bc2s);
    adam_one(p.w, g.w, m.w, v.w, lr, b1, b2, eps, wd, decoupled, step_size, bc2s);
    reinterpret_cast<float4*>(c.p)[i] = p;
    reinterpret_cast<float4*>(c.m)[i] = m;
    reinterpret_cast<float4*>(c.v)[i] = v;
  }
  for (long long i = (n4 << 2) + threadIdx.x; i < c.n; i += 256) {
    float p = c.p[i], m = c.m[i], v = c.v[i];
    adam_one(p, c.g[i] * gs, m, v, lr, b1, b2, eps, wd, decoupled, step_size, bc2s);
    c.p[i] = p; c.m[i] = m; c.v[i] = v;
  }
}

}  // namespace is

// chunks: DEVICE array of nchunks records { float* p; const float* g; float* m; float* v; long long n; };
// state: device float[3] (step, derived step size, derived sqrt bias correction); hyper: device float[8]
// (lr, beta1, beta2, eps, weight_decay, decoupled flag, gradient scale, unused).  Two launches: a one-thread prepare + the streaming update.
extern "C" int is_adam_step(const void* chunks, int nchunks, float* state, const float* hyper, void* stream) {
  if (nchunks <= 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(is::adam_prepare_kernel, dim3(1), dim3(1), 0, st, state, hyper);
  hipLaunchKernelGGL(is::adam_step_kernel, dim3(nchunks), dim3(256), 0, st, static_cast<const is::AdamChunk*>(chunks), state, hyper);
  return is::launch_status(__func__);
}

// The same step in parts, for a caller that updates the parameters of one group in SEVERAL launches (the engine: the parameters
// whose gradients are final early on a side stream beside the tail of the backward pass, the rest at the end): is_adam_prepare
// advances the step count and derives the step's scalars ONCE, is_adam_apply updates the parameters of a chunk table with them.
// is_adam_step == is_adam_prepare + is_adam_apply.
extern "C" int is_adam_prepare(float* state, const float* hyper, void* stream) {
  hipLaunchKernelGGL(is::adam_prepare_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), state, hyper);
  return is::launch_status(__func__);
}
extern "C" int is_adam_apply(const void* chunks, int nchunks, const float* state, const float* hyper, void* stream) {
  if (nchunks <= 0) return 0;
  hipLaunchKernelGGL(is::adam_step_kernel, dim3(nchunks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const is::AdamChunk*>(chunks), state, hyper);
  return is::launch_status(__func__);
}
