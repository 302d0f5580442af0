// Probe: the hi/lo f16 split of a ReLU'd fp32 value done with v_cvt_pk_f16_f32 + v_fma_mixlo/hi_f16 (2.5 VALU ops per
// value) against the plain C++ form the compiler turns into ~3.9 ops per value -- must agree bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probe_split_mix.hip -o /tmp/probe_split_mix && /tmp/probe_split_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
__global__ void k_cur(const float* in, half8* out, int n) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  half8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = relu1(in[t * 8 + e]);
    const _Float16 h = (_Float16)v;
    hi[e] = h;
    lo[e] = (_Float16)(v - (float)h);
  }
  out[t * 2] = hi;
  out[t * 2 + 1] = lo;
}
__global__ void k_mix(const float* in, uint4* out, int n) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  unsigned hi[4], lo[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a = relu1(in[t * 8 + 2 * p]), b = relu1(in[t * 8 + 2 * p + 1]);
    unsigned h, l;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(a), "v"(b));
    asm volatile("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(a));
    asm volatile("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(b));
    hi[p] = h;
    lo[p] = l;
  }
  out[t * 2] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
  out[t * 2 + 1] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}
int main() {
  const int n = 1 << 20;
  std::vector<float> h(n * 8);
  srand(1);
  for (size_t i = 0; i < h.size(); ++i) {
    unsigned bits = ((unsigned)rand() << 16) ^ (unsigned)rand();
    const int kind = i % 5;
    float v;
    if (kind == 0) {                       // any bit pattern that is a finite number
      bits &= 0xff7fffffu;  // clear one exponent bit: never inf/nan
      memcpy(&v, &bits, 4);
    } else if (kind == 1) v = (rand() / (float)RAND_MAX - 0.5f) * 8.f;       // activations
    else if (kind == 2) v = (rand() / (float)RAND_MAX) * 1e-5f;              // f16 subnormal range
    else if (kind == 3) v = (rand() / (float)RAND_MAX) * 70000.f;            // around the f16 maximum
    else v = (rand() / (float)RAND_MAX - 0.5f) * 1e-3f;
    h[i] = v;
  }
  float* d_in;
  void *d_a, *d_b;
  hipMalloc(&d_in, h.size() * 4);
  hipMalloc(&d_a, (size_t)n * 32);
  hipMalloc(&d_b, (size_t)n * 32);
  hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  k_cur<<<n / 256, 256>>>(d_in, (half8*)d_a, n);
  k_mix<<<n / 256, 256>>>(d_in, (uint4*)d_b, n);
  std::vector<unsigned short> a((size_t)n * 16), b((size_t)n * 16);
  hipMemcpy(a.data(), d_a, a.size() * 2, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), d_b, b.size() * 2, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (size_t i = 0; i < a.size(); ++i)
    if (a[i] != b[i] && bad++ < 5) printf("mismatch at %zu: %04x vs %04x (input %g)\n", i, a[i], b[i], h[(i / 16) * 8 + i % 8]);
  printf("%zu values, %zu mismatches\n", a.size(), bad);
  return bad != 0;
}
