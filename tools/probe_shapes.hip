// MFMA-only energy probe: which MFMA shape / operand rotation sustains the most FLOP/s under the package power limit?
//   hipcc --offload-arch=gfx950 -O3 tools/probe_shapes.hip -o tools/probe_shapes.bin && tools/probe_shapes.bin
// Result (one MI355X): 32x32x16 f16 1.65 PFLOP/s, 16x16x32 f16 1.88 PFLOP/s with four rotating random operand sets.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int SHAPE, int NSETS, int NA = NSETS, int NB = NSETS>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  half8 A[4], B[4];
  unsigned s = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + 12345u;
  for (int q = 0; q < 4; ++q)
    for (int e = 0; e < 8; ++e) {
      s = s * 1664525u + 1013904223u;
      const float ra = ((s >> 8) & 0xffff) * (4.0f / 65536.0f) - 2.0f;
      s = s * 1664525u + 1013904223u;
      const float rb = ((s >> 8) & 0xffff) * (4.0f / 65536.0f) - 2.0f;
      A[q][e] = (_Float16)ra;
      B[q][e] = (_Float16)rb;
    }
  float sum = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 12; ++u)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(A[NSETS == 1 ? 0 : ((u + (u >> 2)) & (NSETS - 1))]), "v"(B[NSETS == 1 ? 0 : ((u >> 1) & (NSETS - 1))]));
      if ((it & 63) == 63) for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] *= 0.001f;
    }
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) sum += acc[a][r];
  } else {
    f32x4 acc[8];
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 24; ++u)
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[u & 7]) : "v"(A[NA == 1 ? 0 : ((u + (u >> 2)) & (NA - 1))]), "v"(B[NB == 1 ? 0 : ((u >> 1) & (NB - 1))]));
      if ((it & 63) == 63) for (int a = 0; a < 8; ++a) for (int r = 0; r < 4; ++r) acc[a][r] *= 0.001f;
    }
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 4; ++r) sum += acc[a][r];
  }
  if (sum == 123.456f) out[0] = sum;
}
template <typename F> void run(const char* name, F kern, int iters) {
  float* out; (void)hipMalloc(&out, 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, iters / 8);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double flop = 256.0 * 8 * iters * 12.0 * 32768.0;
  printf("%-44s %8.3f ms  %7.0f TFLOP/s\n", name, ms, flop / ms / 1e9);
}
int main() {
  for (int rep = 0; rep < 3; ++rep) {
    run("32x32x16 f16, 4 operand sets", k<32, 4>, 16000);
    run("32x32x16 f16, 1 operand set", k<32, 1>, 16000);
    run("32x32x16 f16, 2 operand sets", k<32, 2>, 16000);
    run("16x16x32 f16, 4 operand sets", k<16, 4>, 16000);
    run("16x16x32 f16, 1 operand set", k<16, 1>, 16000);
    run("16x16x32 f16, A fixed, B 4 sets", k<16, 4, 1, 4>, 16000);
    run("16x16x32 f16, A 4 sets, B fixed", k<16, 4, 4, 1>, 16000);
  }
}
