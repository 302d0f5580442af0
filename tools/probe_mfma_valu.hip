// Micro-probe: do VALU instructions of one wave overlap the MFMAs of ANOTHER wave on the same SIMD?
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_valu.hip -o /tmp/probe_mv && /tmp/probe_mv
// Block of 512 threads = 2 waves per SIMD.  Role per wave: 0 = idle, 1 = MFMA stream, 2 = VALU stream
// (independent v_fma_f32 chains), 3 = VALU conversion mix (cvt_pk / cvt / sub like the split store).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, int role_old, int role_young) {
  const int w = threadIdx.x >> 6;
  const int role = w < 4 ? role_old : role_young;
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  half8 A, B;
  for (int e = 0; e < 8; ++e) {
    A[e] = (_Float16)(0.001f * (threadIdx.x + e));
    B[e] = (_Float16)(0.002f * (threadIdx.x - e));
  }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
  const unsigned long long t0 = clock64();
  if (role == 1) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 12; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc[u & 3], 0, 0, 0);
    }
  } else if (role == 4) {   // v_mfma_f32_16x16x32_f16: 4 accumulator registers, 4 passes
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x4v c[4];
    for (int a = 0; a < 4; ++a) c[a] = f32x4v{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 24; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, c[u & 3], 0, 0, 0);
    }
    for (int a = 0; a < 4; ++a) acc[0][a] += c[a][0];
  } else if (role == 5) {   // SAME wave: 12 MFMAs with 8 independent VALU ops behind each
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 12; ++u) {
        acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc[u & 3], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = __builtin_fmaf(v[q], 1.0001f, 0.5f);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (role == 6) {   // ... 16 behind each
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 12; ++u) {
        acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc[u & 3], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = __builtin_fmaf(v[q], 1.0001f, 0.5f);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (role == 2) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 96; ++u) v[u & 15] = __builtin_fmaf(v[u & 15], 1.0001f, 0.5f);
    }
  } else if (role == 3) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 24; ++u) {
        const float x = v[u & 15];
        const _Float16 hh = (_Float16)x;
        const float lo = x - (float)hh;
        v[u & 15] = lo * 3.0f + (float)(_Float16)lo;
      }
    }
  }
  const unsigned long long t1 = clock64();
  float s = 0.f;
  for (int a = 0; a < 4; ++a) s += acc[a][0];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

void run(const char* name, int ro, int ry) {
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMalloc(&cyc, 256 * 8 * 8);
  const int iters = 2000;
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, cyc, iters, ro, ry);
  (void)hipDeviceSynchronize();
  unsigned long long h[8];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-46s older wave %8.1f   younger wave %8.1f  cycles per iteration\n", name, (double)h[0] / iters, (double)h[4] / iters);
  (void)hipFree(out);
  (void)hipFree(cyc);
}

int main() {
  run("MFMA x12 alone (older)", 1, 0);
  run("VALU fma x96 alone (younger)", 0, 2);
  run("MFMA x12 (older) + VALU fma x96 (younger)", 1, 2);
  run("VALU fma x96 (older) + MFMA x12 (younger)", 2, 1);
  run("MFMA 16x16x32 x24 alone (older)", 4, 0);
  run("MFMA 16x16x32 x24 (older) + VALU fma x96 (younger)", 4, 2);
  run("VALU fma x96 (older) + MFMA 16x16x32 x24 (younger)", 2, 4);
  run("same wave: 12 x (MFMA + 8 VALU), alone", 5, 0);
  run("same wave: 12 x (MFMA + 16 VALU), alone", 6, 0);
  run("same wave: 12 x (MFMA + 8 VALU), both waves", 5, 5);
  run("cvt mix x24 alone (younger)", 0, 3);
  run("MFMA x12 (older) + cvt mix x24 (younger)", 1, 3);
  run("cvt mix x24 (older) + MFMA x12 (younger)", 3, 1);
  return 0;
}
