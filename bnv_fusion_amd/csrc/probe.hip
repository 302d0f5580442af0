// Diagnostic, not on the product path: what rate of v_mfma_f32_32x32x16_f16 does THIS GPU sustain right now?
// The two MLP kernels of the frame run at the package power limit: the clock, and with it
// the dense MFMA rate, settles well below the 2.4 GHz the 2.5 PFLOP/s peak is quoted at, and by how much depends
// on the operand data (zeros toggle nothing).  bench.py runs this beside the timed frames and reports the
// dominant kernel against both numbers: the guide's peak (roofline.peak) and this measured ceiling.
#include <hip/hip_runtime.h>

#include "../../include/bnv_fusion.h"
#include "bnv_common.hpp"

namespace bnv {

typedef _Float16 pr_half8 __attribute__((ext_vector_type(8)));
typedef float pr_f32x16 __attribute__((ext_vector_type(16)));

// 8 waves per workgroup (two per SIMD), one workgroup per CU-slot; every wave issues iters x 12 32x32x16 MFMAs on
// four accumulators (SHAPE 0) or iters x 24 16x16x32 MFMAs on eight (SHAPE 1) -- the same FLOPs --, operands from
// four register sets: zeros (operands = 0) or uniform random f16 in [-2, 2) (1)
template <int SHAPE>
__global__ __launch_bounds__(512) void k_probe_mfma(float* __restrict__ sink, int iters, int operands) {
  pr_f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  typedef float pr_f32x4 __attribute__((ext_vector_type(4)));
  pr_f32x4 acc4[8];
#pragma unroll
  for (int a = 0; a < 8; ++a) acc4[a] = pr_f32x4{0.f, 0.f, 0.f, 0.f};
  pr_half8 A[4], B[4];
  unsigned s = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s = s * 1664525u + 1013904223u;
      const float ra = ((s >> 8) & 0xffff) * (4.0f / 65536.0f) - 2.0f;
      s = s * 1664525u + 1013904223u;
      const float rb = ((s >> 8) & 0xffff) * (4.0f / 65536.0f) - 2.0f;
      A[q][e] = operands ? (_Float16)ra : (_Float16)0.f;
      B[q][e] = operands ? (_Float16)rb : (_Float16)0.f;
    }
  for (int it = 0; it < iters; ++it) {
    if (SHAPE == 0) {
#pragma unroll
      for (int u = 0; u < 12; ++u)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0"
                     : "+v"(acc[u & 3])
                     : "v"(A[(u + (u >> 2)) & 3]), "v"(B[(u >> 1) & 3]));
    } else {
#pragma unroll
      for (int u = 0; u < 24; ++u)
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0"
                     : "+v"(acc4[u & 7])
                     : "v"(A[(u + (u >> 2)) & 3]), "v"(B[(u >> 1) & 3]));
    }
    if ((it & 63) == 63) {   // keep the sums finite (rare: 1 pass in 64)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] *= 0.001f;
#pragma unroll
      for (int a = 0; a < 8; ++a) acc4[a] *= 0.001f;
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += acc[a][r];
#pragma unroll
  for (int a = 0; a < 8; ++a) sum += acc4[a][0] + acc4[a][1] + acc4[a][2] + acc4[a][3];
  if (sum == 123.456f) sink[0] = sum;   // never true: keeps the accumulators alive without a store per thread
}

}  // namespace bnv

extern "C" int bnv_probe_mfma_rate(int shape, int operands, int iters, void* stream_, double* ms_host,
                                   double* flop_host) {
  using namespace bnv;
  if (!ms_host || !flop_host || iters <= 0 || (operands != 0 && operands != 1) || (shape != 0 && shape != 1))
    return BNV_ERR_INVALID_ARGUMENT;
  hipStream_t stream = (hipStream_t)stream_;
  int dev = 0, cus = 0;
  BNV_HIP_CHECK(hipGetDevice(&dev));
  BNV_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  float* sink = nullptr;
  BNV_HIP_CHECK(hipMalloc(&sink, 16));
  hipEvent_t e0, e1;
  BNV_HIP_CHECK(hipEventCreate(&e0));
  BNV_HIP_CHECK(hipEventCreate(&e1));
  auto kern = shape == 0 ? k_probe_mfma<0> : k_probe_mfma<1>;
  hipLaunchKernelGGL(kern, dim3(cus), dim3(512), 0, stream, sink, iters / 8 + 1, operands);   // warm
  BNV_HIP_CHECK(hipEventRecord(e0, stream));
  hipLaunchKernelGGL(kern, dim3(cus), dim3(512), 0, stream, sink, iters, operands);
  BNV_HIP_CHECK(hipEventRecord(e1, stream));
  BNV_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  BNV_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  BNV_HIP_CHECK(hipEventDestroy(e0));
  BNV_HIP_CHECK(hipEventDestroy(e1));
  BNV_HIP_CHECK(hipFree(sink));
  *ms_host = ms;
  *flop_host = (double)cus * 8.0 * (double)iters * 12.0 * (2.0 * 32 * 32 * 16);
  return BNV_OK;
}

// ---- do two HIP streams of this process really run side by side?  A single-workgroup kernel that spins for a number
// of shader cycles: one on each of two streams takes as long as one alone if the streams are served by different
// hardware queues (bnv_fusion_amd/streams.py picks the frame pipeline's encode stream with it; tools/stream_probe.py).
namespace bnv {
__global__ void k_probe_spin(long long cycles, int* sink) {
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) {
  }
  if (sink && cycles < 0) *sink = 1;
}
}  // namespace bnv

extern "C" int bnv_probe_spin(int n_blocks, int64_t cycles, void* stream) {
  if (n_blocks < 1 || cycles < 0) return BNV_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(bnv::k_probe_spin, dim3(n_blocks), dim3(64), 0, (hipStream_t)stream, (long long)cycles, (int*)nullptr);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}
