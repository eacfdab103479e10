// semantics of bounded raw-buffer accesses on gfx950 (what egnn_layer_bwd relies on): out-of-range loads return 0,
// out-of-range stores are dropped, the instruction offset takes part in the range check, dwordx3 loads work
#include "../../immunostruct_amd/csrc/common.h"
#include <cstdio>
#include <vector>
using namespace is;
__global__ void k(const float* src, float* out, float* st, int nbytes) {
  rsrc_t r = make_rsrc_n(src + 64, nbytes);
  const int lane = threadIdx.x;
  int vt = lane * 4;
  asm volatile("" : "+v"(vt));
  out[lane] = buf_load(r, vt, 0);
  out[64 + lane] = buf_load(r, vt + 256, 0);
  out[128 + lane] = buf_load(r, (lane < 32) ? vt : BUF_OOB, 0);
  float a, b, c;
  buf_load3(make_rsrc(src), lane * 12, 0, a, b, c);
  out[192 + lane] = a + 10 * b + 100 * c;
  rsrc_t w = make_rsrc_n(st, nbytes);
  buf_store(1.0f + lane, w, vt, 0);
  buf_store(100.0f + lane, w, vt + 256, 0);
}
int main() {
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = i;
  float *src, *out, *st;
  hipMalloc(&src, 4096); hipMalloc(&out, 4096); hipMalloc(&st, 4096);
  hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
  hipMemset(st, 0, 4096);
  for (int nbytes : {512, 300, 0}) {
    hipMemset(st, 0, 4096);
    k<<<1, 64>>>(src, out, st, nbytes);
    std::vector<float> o(256), s(256);
    hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost);
    hipMemcpy(s.data(), st, 1024, hipMemcpyDeviceToHost);
    printf("nbytes %d: load[0]=%g load[63]=%g | +256: [0]=%g [10]=%g [11]=%g [63]=%g | oob-sel [31]=%g [32]=%g | x3 [5]=%g (expect %g)\n", nbytes,
           o[0], o[63], o[64], o[74], o[75], o[127], o[159], o[160], o[197], 15 + 10 * 16. + 100 * 17.);
    printf("   store: st[0]=%g st[63]=%g st[64]=%g st[74]=%g st[75]=%g st[127]=%g\n", s[0], s[63], s[64], s[74], s[75], s[127]);
  }
  return 0;
}
