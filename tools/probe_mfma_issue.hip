// Micro-probe: MFMA issue rate of v_mfma_f32_32x32x16_f16 from ONE wave per SIMD vs TWO, with and without a
// ds_read_b128 in the shadow of every MFMA, and with dependent (same accumulator) vs independent chains.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_issue.hip -o /tmp/probe_issue && /tmp/probe_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)i * 1e-6f;
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  half8 A, B[4];
  for (int e = 0; e < 8; ++e) A[e] = (_Float16)(0.001f * (threadIdx.x + e));
  for (int p = 0; p < 4; ++p) B[p] = *(const half8*)&lds[(threadIdx.x & 63) * 4 + p * 256];
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B[u & 3], acc[u % NACC], 0, 0, 0);
      if (LDS && u < 8) B[u & 3] = *(const half8*)&lds[((threadIdx.x & 63) * 4 + ((it + u) & 7) * 256) & 8191];
    }
    if (LDS) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = clock64();
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) s += acc[a][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NACC, bool LDS>
void run(const char* name, int threads) {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 256 * 8 * 8);
  hipMemset(cyc, 0, 256 * 8 * 8);
  const int iters = 2000;
  hipLaunchKernelGGL((k<NACC, LDS>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-44s waves/SIMD=%d : ", name, threads / 256);
  for (int w = 0; w < threads / 64; ++w) printf("%6.1f ", (double)h[w] / (iters * 12.0));
  printf(" cycles per MFMA per wave\n");
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run<4, false>("4 independent accumulators, no LDS", 256);
  run<4, false>("4 independent accumulators, no LDS", 512);
  run<1, false>("1 accumulator (dependent chain), no LDS", 256);
  run<1, false>("1 accumulator (dependent chain), no LDS", 512);
  run<4, true>("4 acc + ds_read_b128 behind 8 of 12 MFMAs", 256);
  run<4, true>("4 acc + ds_read_b128 behind 8 of 12 MFMAs", 512);
  return 0;
}
