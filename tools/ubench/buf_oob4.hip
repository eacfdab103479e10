// bounded raw-buffer views with 16-byte accesses on gfx950: which lanes of a buffer_load_dwordx4 / buffer_store_dwordx4 that
// straddles or lies past num_records read zero / are dropped?   (tools/ubench/buf_oob.hip covers the dword forms)
#include "../../immunostruct_amd/csrc/common.h"
#include <cstdio>
#include <vector>
using namespace is;
__global__ void k(const float* src, float* out, float* st, int nbytes) {
  rsrc_t r = make_rsrc_n(src, nbytes);
  const int lane = threadIdx.x;
  int vt = lane * 16;
  asm volatile("" : "+v"(vt));
  const f32x4 v = buf_load4(r, vt, 0);
  const f32x4 w = buf_load4(r, vt, 256);      // scalar offset: not part of the range check
  for (int j = 0; j < 4; ++j) { out[lane * 4 + j] = v[j]; out[256 + lane * 4 + j] = w[j]; }
  rsrc_t ws = make_rsrc_n(st, nbytes);
  buf_store4(f32x4{1.f + lane, 2.f + lane, 3.f + lane, 4.f + lane}, ws, vt, 0);
}
int main() {
  std::vector<float> h(2048);
  for (int i = 0; i < 2048; ++i) h[i] = 1000 + i;
  float *src, *out, *st;
  hipMalloc(&src, 8192); hipMalloc(&out, 8192); hipMalloc(&st, 8192);
  hipMemcpy(src, h.data(), 8192, hipMemcpyHostToDevice);
  for (int nbytes : {1024, 512, 520, 256 * 3, 0}) {
    hipMemset(st, 0, 8192);
    k<<<1, 64>>>(src, out, st, nbytes);
    std::vector<float> o(512), s(512);
    hipMemcpy(o.data(), out, 2048, hipMemcpyDeviceToHost);
    hipMemcpy(s.data(), st, 2048, hipMemcpyDeviceToHost);
    int first_zero = -1, first_zero_s = -1, first_unwritten = -1;
    for (int i = 0; i < 256; ++i) { if (o[i] == 0 && first_zero < 0) first_zero = i; if (o[256 + i] == 0 && first_zero_s < 0) first_zero_s = i; if (s[i] == 0 && first_unwritten < 0) first_unwritten = i; }
    printf("num_records %4d bytes (%3d floats): x4 load reads zero from float %d on; with scalar offset 256 from float %d on (value there %g); x4 store drops from float %d on\n",
           nbytes, nbytes / 4, first_zero, first_zero_s, first_zero_s > 0 ? o[256 + first_zero_s - 1] : -1.f, first_unwritten);
  }
  return 0;
}
