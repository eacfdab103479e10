// CSR segment reductions: the deterministic replacement for torch_scatter /
// DGL SpMM scatter-adds on this path.
//
//   is_gather_segment_sum : out[v] = sum over p in [ptr[v], ptr[v+1]) of rows[pos[p]]
//       -- the source-side "scatter-add" of the EGNN backward (dPs[src] += dz1,
//       dx[src] += dD) expressed as a gather over the CSR-by-source index, one
//       wave per node, one coalesced 256-byte row per edge.
//   is_segment_pool_{fwd,bwd} : per-graph mean / max readout, the
//       torch_geometric.nn.global_mean_pool / global_max_pool replacement
//       (reference models/hybrid_models.py:331, models/ablation_models.py:296-297).
//       mean = sum / max(count, 1); max of an empty segment = 0 (PyG semantics).
//       Backward of max splits the gradient evenly among tied maxima (torch
//       amax semantics; padded nodes produce identical rows, so ties are real).
// Fixed summation order everywhere => bitwise reproducible.
#include "common.h"

namespace is {

// 16 lanes per node (one float4 = 4 channels per lane, a 256-byte row per 16-lane group), 16 nodes per workgroup:
// a quarter of the waves of a wave-per-node mapping, so the whole batch is resident at once and the three dependent
// latencies (rowptr -> slot ids -> rows) are paid once instead of once per round.  Per-node summation order unchanged.
__global__ __launch_bounds__(256) void gather_segment_sum_kernel(
    const float* __restrict__ rows, const float* __restrict__ vec3,
    const int* __restrict__ ptr, const int* __restrict__ pos,
    float* __restrict__ out_rows, int ld_out, float* __restrict__ out_vec3, int N, long long* __restrict__ wg_clock) {
  wg_clock_start(wg_clock);
  const int sub = threadIdx.x & 15;
  const int v = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (v < N) {
  const int lo = ptr[v], hi = ptr[v + 1];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float acc3 = 0.0f;
  const bool has3 = vec3 != nullptr && sub < 3;
  int p = lo;
  for (; p + 4 <= hi; p += 4) {
    const int e0 = pos[p], e1 = pos[p + 1], e2 = pos[p + 2], e3 = pos[p + 3];
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(rows + (size_t)e0 * H + sub * 4);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(rows + (size_t)e1 * H + sub * 4);
    const f32x4 a2 = *reinterpret_cast<const f32x4*>(rows + (size_t)e2 * H + sub * 4);
    const f32x4 a3 = *reinterpret_cast<const f32x4*>(rows + (size_t)e3 * H + sub * 4);
    acc += a0; acc += a1; acc += a2; acc += a3;
    if (has3) {
      acc3 += vec3[(size_t)e0 * 3 + sub]; acc3 += vec3[(size_t)e1 * 3 + sub];
      acc3 += vec3[(size_t)e2 * 3 + sub]; acc3 += vec3[(size_t)e3 * 3 + sub];
    }
  }
  for (; p < hi; ++p) {
    const int e0 = pos[p];
    acc += *reinterpret_cast<const f32x4*>(rows + (size_t)e0 * H + sub * 4);
    if (has3) acc3 += vec3[(size_t)e0 * 3 + sub];
  }
  float* o = out_rows + (size_t)v * ld_out + sub * 4;
  if ((ld_out & 3) == 0 && (reinterpret_cast<uintptr_t>(out_rows) & 15) == 0) {
    *reinterpret_cast<f32x4*>(o) = acc;
  } else {
    o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3];
  }
  if (has3) out_vec3[v * 3 + sub] += acc3;
  }   // v < N
  wg_clock_end(wg_clock);
}

// one workgroup per (segment, 64-channel slab)
__global__ __launch_bounds__(256) void segment_pool_fwd_kernel(
    const float* __restrict__ x, int ld_x, const int* __restrict__ seg_ptr,
    float* __restrict__ out_mean, float* __restrict__ out_max, int C) {
  __shared__ float red_s[4][64];
  __shared__ float red_m[4][64];
  const int seg = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  const int lo = seg_ptr[seg], hi = seg_ptr[seg + 1];
  float s = 0.0f, m = -INFINITY;
  if (c < C) {
    for (int row = lo + rg; row < hi; row += 4) {
      const float v = x[(size_t)row * ld_x + c];
      s += v;
      m = fmaxf(m, v);
    }
  }
  red_s[rg][threadIdx.x & 63] = s;
  red_m[rg][threadIdx.x & 63] = m;
  __syncthreads();
  if (rg == 0 && c < C) {
    const int l = threadIdx.x & 63;
    const float st = ((red_s[0][l] + red_s[1][l]) + red_s[2][l]) + red_s[3][l];
    const float mt = fmaxf(fmaxf(red_m[0][l], red_m[1][l]), fmaxf(red_m[2][l], red_m[3][l]));
    const int cnt = hi - lo;
    if (out_mean != nullptr) out_mean[(size_t)seg * C + c] = st / (float)max(cnt, 1);
    if (out_max != nullptr) out_max[(size_t)seg * C + c] = cnt > 0 ? mt : 0.0f;
  }
}

__global__ __launch_bounds__(256) void segment_pool_bwd_kernel(
    const float* __restrict__ x, int ld_x, const int* __restrict__ seg_ptr,
    const float* __restrict__ out_max, const float* __restrict__ g_mean, const float* __restrict__ g_max,
    float* __restrict__ dx, int ld_dx, int C) {
  __shared__ int ties[4][64];
  const int seg = blockIdx.x, l = threadIdx.x & 63, c = blockIdx.y * 64 + l, rg = threadIdx.x >> 6;
  const int lo = seg_ptr[seg], hi = seg_ptr[seg + 1];
  const bool on = c < C;
  const float gm = (on && g_mean != nullptr) ? g_mean[(size_t)seg * C + c] / (float)max(hi - lo, 1) : 0.0f;
  float mx = 0.0f, gx = 0.0f;
  int cnt = 0;
  if (g_max != nullptr) {
    if (on) {
      mx = out_max[(size_t)seg * C + c];
      gx = g_max[(size_t)seg * C + c];
      for (int row = lo + rg; row < hi; row += 4) cnt += (x[(size_t)row * ld_x + c] == mx) ? 1 : 0;
    }
    ties[rg][l] = cnt;
    __syncthreads();
    cnt = ties[0][l] + ties[1][l] + ties[2][l] + ties[3][l];
    gx = cnt > 0 ? gx / (float)cnt : 0.0f;
  }
  if (!on) return;
  for (int row = lo + rg; row < hi; row += 4) {
    float g = gm;
    if (g_max != nullptr && x[(size_t)row * ld_x + c] == mx) g += gx;
    dx[(size_t)row * ld_dx + c] = g;
  }
}

}  // namespace is

extern "C" int is_gather_segment_sum(const float* rows, const float* vec3, const int32_t* ptr, const int32_t* pos,
                                     float* out_rows, int ld_out, float* out_vec3, int N, long long* wg_clock, void* stream) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL(is::gather_segment_sum_kernel, dim3((N + 15) / 16), dim3(256), 0, static_cast<hipStream_t>(stream),
                     rows, vec3, ptr, pos, out_rows, ld_out, out_vec3, N, wg_clock);
  return is::launch_status(__func__);
}

extern "C" int is_segment_pool_fwd(const float* x, int ld_x, const int32_t* seg_ptr, float* out_mean, float* out_max,
                                   int num_segments, int C, void* stream) {
  if (num_segments <= 0 || C <= 0) return 0;
  hipLaunchKernelGGL(is::segment_pool_fwd_kernel, dim3(num_segments, (C + 63) / 64), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, ld_x, seg_ptr, out_mean, out_max, C);
  return is::launch_status(__func__);
}

extern "C" int is_segment_pool_bwd(const float* x, int ld_x, const int32_t* seg_ptr, const float* out_max,
                                   const float* g_mean, const float* g_max, float* dx, int ld_dx, int num_segments,
                                   int C, void* stream) {
  if (num_segments <= 0 || C <= 0) return 0;
  hipLaunchKernelGGL(is::segment_pool_bwd_kernel, dim3(num_segments, (C + 63) / 64), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, ld_x, seg_ptr, out_max, g_mean, g_max, dx, ld_dx, C);
  return is::launch_status(__func__);
}

// ---------------------------------------------------------------------------------------------
// Batched device-to-device copy: the per-step hand-over of a device-resident batch into the static
// buffers a captured HIP graph replays on (node features, CSR arrays, edge features, sequence, property,
// target) as ONE launch instead of ten hipMemcpyAsync calls.
namespace is {
struct CopyJob { const void* src; void* dst; long long bytes; };
constexpr int COPY_MAX_JOBS = 24;
struct CopyBatch { CopyJob job[COPY_MAX_JOBS]; };

__global__ __launch_bounds__(256) void multi_copy_kernel(CopyBatch batch) {
  const CopyJob& J = batch.job[blockIdx.y];
  const long long words = J.bytes >> 2;              // sizes are multiples of 4 bytes
  const int* s = static_cast<const int*>(J.src);
  int* d = static_cast<int*>(J.dst);
  const bool vec = (((uintptr_t)s | (uintptr_t)d) & 15) == 0;
  const long long stride = (long long)gridDim.x * 256;
  if (vec) {
    const long long quads = words >> 2;
    const int4* s4 = reinterpret_cast<const int4*>(s);
    int4* d4 = reinterpret_cast<int4*>(d);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < quads; i += stride) d4[i] = s4[i];
    for (long long i = (quads << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < words; i += stride) d[i] = s[i];
  } else {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < words; i += stride) d[i] = s[i];
  }
}
}  // namespace is

// jobs: host array of njobs (<= 24) records { const void* src; void* dst; long long bytes; }, bytes % 4 == 0
extern "C" int is_multi_copy(const void* jobs, int njobs, void* stream) {
  if (njobs <= 0 || njobs > is::COPY_MAX_JOBS) return is::fail(__func__, -22);
  is::CopyBatch batch;
  const is::CopyJob* src = static_cast<const is::CopyJob*>(jobs);
  long long maxb = 0;
  for (int i = 0; i < njobs; ++i) {
    if (src[i].bytes < 0 || (src[i].bytes & 3)) return is::fail(__func__, -22);
    batch.job[i] = src[i];
    maxb = src[i].bytes > maxb ? src[i].bytes : maxb;
  }
  int blocks = (int)((maxb / 16 + 255) / 256);
  blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
  hipLaunchKernelGGL(is::multi_copy_kernel, dim3(blocks, njobs), dim3(256), 0, static_cast<hipStream_t>(stream), batch);
  return is::launch_status(__func__);
}

// ---------------------------------------------------------------------------------------------
// On-device batcher: assemble the block-diagonal batch of B graphs (reference data/utils.py:160-176, dgl.batch in
// `collate`) from a device-resident dataset of per-graph CSR pieces -- no host work, no sort: every graph's edges are
// already in destination order, so the batch's CSR is the concatenation of the pieces with node / edge offsets added.
// Workgroup i assembles graph slot i (graph id idx[i]); its edge offset is the sum of the edge counts of slots < i.
namespace is {
struct BatchSrc {     // dataset, all graphs padded to n nodes
  const float* x;               // [G][n][F]
  const int* eoff;              // [G + 1]  edge offset of every graph in the concatenated edge arrays
  const int* rowptr_dst;        // [G][n + 1]  local (0-based per graph)
  const int* rowptr_src;        // [G][n + 1]
  const int* src;               // [Etot]  local node ids, destination order
  const int* dst;               // [Etot]
  const int* pos;               // [Etot]  local slot ids, source order
  const float* ea;              // [Etot][Fe]
};
struct BatchDst {
  float* x; int* rowptr_dst; int* rowptr_src; int* src; int* dst; int* pos; float* ea;
};
// per-sample rows that ride along (sequence one-hots, property vectors, targets): dst[i][:] = src[idx[i]][:]
constexpr int BATCH_ROWS_MAX = 4;
struct RowGather { const float* src; float* dst; int floats, pad; };
struct RowGathers { RowGather job[BATCH_ROWS_MAX]; int n; };

__global__ __launch_bounds__(256) void batch_gather_kernel(const long long* __restrict__ idx, int B, int n, int F, int Fe,
                                                           BatchSrc S, BatchDst D, RowGathers R) {
  __shared__ int red[256];
  const int i = blockIdx.x, tid = threadIdx.x;
  int part = 0;
  for (int j = tid; j < i; j += 256) {
    const long long gj = idx[j];
    part += S.eoff[gj + 1] - S.eoff[gj];
  }
  red[tid] = part;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const int off = red[0];
  const long long g = idx[i];
  const int e0 = S.eoff[g], cnt = S.eoff[g + 1] - e0, v0 = i * n;
  for (int k = tid; k < n * F; k += 256) D.x[(size_t)v0 * F + k] = S.x[(size_t)g * n * F + k];
  for (int v = tid; v < n; v += 256) {
    D.rowptr_dst[v0 + v] = S.rowptr_dst[(size_t)g * (n + 1) + v] + off;
    D.rowptr_src[v0 + v] = S.rowptr_src[(size_t)g * (n + 1) + v] + off;
  }
  if (i == B - 1 && tid == 0) {
    D.rowptr_dst[v0 + n] = off + cnt;
    D.rowptr_src[v0 + n] = off + cnt;
  }
  for (int e = tid; e < cnt; e += 256) {
    D.src[off + e] = S.src[e0 + e] + v0;
    D.dst[off + e] = S.dst[e0 + e] + v0;
    D.pos[off + e] = S.pos[e0 + e] + off;
  }
  for (int k = tid; k < cnt * Fe; k += 256) D.ea[(size_t)off * Fe + k] = S.ea[(size_t)e0 * Fe + k];
  for (int j = 0; j < R.n; ++j) {
    const RowGather& J = R.job[j];
    const float* src = J.src + (size_t)g * J.floats;
    float* dst = J.dst + (size_t)i * J.floats;
    for (int k = tid; k < J.floats; k += 256) dst[k] = src[k];
  }
}

// chunk_ptr [k + 1][2] = (b_j, rowptr[b_j]) with b_0 = 0, b_k = N, b_j = first node whose first in-edge index is >= j * E / k:
// the node-aligned, edge-balanced partition of graph.py `balanced_node_chunks`, straight from the rowptr in device memory
// shares != 0: the chunks of the first / second half of the workgroups get (P + 1) / 2 : (P - 1) / 2 parts of the edges when
// P = ceil(edges per wave pair / 16) is odd, the node-aligned chunks have slack and the larger workgroups need no extra pass of
// the node half (graph.py chunk_shares: same integer rule)
__global__ __launch_bounds__(256) void chunk_partition_kernel(const int* __restrict__ rowptr, int N, int k, int shares, int* __restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j > k) return;
  const long long E = rowptr[N];
  long long wa = 1, wb = 1;
  if (shares) {
    const long long p = (2 * E + 16LL * k - 1) / (16LL * k);
    if ((p % 2 == 1) && p >= 3 && (p * 16 * k - 2 * E >= 6LL * k)) {
      // ... and the larger workgroups' nodes (average + a margin of 4) must not need one more 64-row pass of the forward kernel's
      // node half than equal shares would
      const long long big = (8LL * N * ((p + 1) / 2) + p * k - 1) / (p * k), flat = (4LL * N + k - 1) / k;
      if ((big + 4 + 63) / 64 == (flat + 4 + 63) / 64) { wa = (p + 1) / 2; wb = (p - 1) / 2; }
    }
  }
  const long long half = k / 2;
  const long long wj = wa * min((long long)j, half) + wb * max((long long)j - half, 0LL);
  const long long target = (wj * E) / (wa * half + wb * (k - half));
  int lo = 0, hi = N + 1;      // first index i in [0, N] with rowptr[i] >= target (N + 1: none)
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (rowptr[mid] < target) lo = mid + 1; else hi = mid;
  }
  int b = min(lo, N);
  if (j == 0) b = 0;
  if (j == k) b = N;
  out[2 * j] = b;
  out[2 * j + 1] = rowptr[b];
}
}  // namespace is

// idx [B] int64 (device): graph ids.  Dataset arrays as documented in is::BatchSrc; destination arrays of a batch with
// B * n nodes and room for the selected graphs' edges (caller guarantees the capacity: B * max edges per graph).
extern "C" int is_batch_gather(const long long* idx, int B, int n, int F, int Fe, const float* x_all, const int32_t* eoff,
                               const int32_t* rowptr_dst_all, const int32_t* rowptr_src_all, const int32_t* src_all,
                               const int32_t* dst_all, const int32_t* pos_all, const float* ea_all, float* x,
                               int32_t* rowptr_dst, int32_t* rowptr_src, int32_t* src_sorted, int32_t* dst_sorted,
                               int32_t* pos_by_src, float* ea, const void* rows, int nrows, void* stream) {
  if (B <= 0) return 0;
  if (n <= 0 || F <= 0 || Fe < 0 || nrows < 0 || nrows > is::BATCH_ROWS_MAX || (nrows > 0 && rows == nullptr)) return is::fail(__func__, -22);
  is::BatchSrc S{x_all, eoff, rowptr_dst_all, rowptr_src_all, src_all, dst_all, pos_all, ea_all};
  is::BatchDst D{x, rowptr_dst, rowptr_src, src_sorted, dst_sorted, pos_by_src, ea};
  is::RowGathers R{};
  R.n = nrows;
  for (int j = 0; j < nrows; ++j) {
    R.job[j] = static_cast<const is::RowGather*>(rows)[j];
    if (R.job[j].src == nullptr || R.job[j].dst == nullptr || R.job[j].floats <= 0) return is::fail(__func__, -22);
  }
  hipLaunchKernelGGL(is::batch_gather_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), idx, B, n, F, Fe, S, D, R);
  return is::launch_status(__func__);
}

// rowptr [N + 1] (device) -> chunk_ptr [k + 1][2] int32: the edge-balanced node partition the layer kernels walk (graph.py
// `balanced_node_chunks`), recomputed on the device after the batcher wrote a new rowptr -- one launch, no host sync.
// shares: 0 = equal shares; 1 = the two-level shares of graph.py chunk_shares (the caller passes 1 only for the full grid).
extern "C" int is_chunk_partition(const int32_t* rowptr, int N, int k, int shares, int32_t* chunk_ptr, void* stream) {
  if (N < 0 || k <= 0 || rowptr == nullptr || chunk_ptr == nullptr) return is::fail(__func__, -22);
  hipLaunchKernelGGL(is::chunk_partition_kernel, dim3((k + 256) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), rowptr, N, k, shares, chunk_ptr);
  return is::launch_status(__func__);
}
