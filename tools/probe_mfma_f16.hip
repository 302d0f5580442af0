// Hardware probe (gfx950): v_mfma_f32_32x32x16_f16 operand-slot pairing, D layout and f16
// subnormal handling.  Build: hipcc --offload-arch=gfx950 -O2 tools/probe_mfma_f16.hip -o /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const _Float16* A, const _Float16* B, float* D) {
  const int l = threadIdx.x;
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[l * 8 + j]; b[j] = B[l * 8 + j]; }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[l * 16 + r] = c[r];
}

int main() {
  _Float16 hA[512], hB[512];
  float hD[1024];
  srand(1);
  for (int i = 0; i < 512; ++i) { hA[i] = (_Float16)((rand() % 2001 - 1000) / 512.0f); hB[i] = (_Float16)((rand() % 2001 - 1000) / 512.0f); }
  _Float16 *dA, *dB; float* dD;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  // hypothesis: D[i][n] = sum_{h,j} A(lane=(i,h), j) * B(lane=(n,h), j); D reg r of lane (n,h'): i = (r&3)+8(r>>2)+4h'
  double maxerr = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
    const int n = l & 31, hp = l >> 5, i = (r & 3) + 8 * (r >> 2) + 4 * hp;
    double s = 0;
    for (int h = 0; h < 2; ++h) for (int j = 0; j < 8; ++j) s += (double)hA[(i + 32 * h) * 8 + j] * (double)hB[(n + 32 * h) * 8 + j];
    maxerr = fmax(maxerr, fabs(s - hD[l * 16 + r]));
  }
  printf("slot-pairing + D-layout hypothesis: max |err| = %.3e (%s)\n", maxerr, maxerr < 1e-4 ? "HOLDS" : "FAILS");
  // subnormal test: A = subnormal f16 (2^-20), B = 1024 -> expect 2^-10 per product if not flushed
  for (int i = 0; i < 512; ++i) { hA[i] = (_Float16)0.f; hB[i] = (_Float16)0.f; }
  for (int l = 0; l < 64; ++l) { hA[l * 8] = (_Float16)9.5367431640625e-07f; hB[l * 8] = (_Float16)1024.f; }
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  printf("subnormal A (2^-20) x 1024, two slots: D[0] = %.6e (expected 1.953125e-03 if subnormals are kept, 0 if flushed)\n", hD[0]);
  return 0;
}
