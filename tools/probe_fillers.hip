// Micro-probe: how many non-MFMA instructions fit in the shadow of one v_mfma_f32_32x32x16_f16 on gfx950, by kind,
// with one and with two waves per SIMD; and what a dependent accumulator chain costs.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_fillers.hip -o /tmp/probe_fillers && /tmp/probe_fillers
// Every loop body is 12 x (1 MFMA + K fillers) written as asm volatile statements, so the issue order is exactly
// the source order (the older probe_mfma_valu.hip only tried 8 and 16 fillers per MFMA, beyond what fits).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B))
#define F_MAX(x) asm volatile("v_max_i32 %0, %0, 0" : "+v"(x))
#define F_FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c1))
#define F_CVT(d, a, b) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define F_MIXLO(d, hi, a) asm volatile("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hi), "v"(a))
#define F_MIXHI(d, hi, b) asm volatile("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hi), "v"(b))
#define F_DSR(d, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr))
#define F_DSW(addr, s) asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(s))

// KIND: 0 none, 1 v_max_i32, 2 v_fma_f32, 3 split mix (max,max,cvt,mixlo,mixhi repeating), 4 ds_read_b128,
//       5 ds_write_b128, 6 realistic K-loop mix (K ds_read per 3 MFMA handled separately)
template <int KIND, int K>
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* cyc, int iters, int active_waves) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int w = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  half8 A, B;
  for (int e = 0; e < 8; ++e) {
    A[e] = (_Float16)(0.001f * ((threadIdx.x & 63) + e));
    B[e] = (_Float16)(0.002f * ((threadIdx.x & 63) - e));
  }
  float v[8];
  unsigned hi[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0};
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  const float c1 = 1.0001f;
  f32x4 d[4];
  for (int i = 0; i < 4; ++i) d[i] = f32x4{1.f, 2.f, 3.f, 4.f};
  const unsigned addr = (threadIdx.x & 63) * 16 + w * 1024;
  lds[threadIdx.x] = 0.f;
  __syncthreads();
  unsigned long long t0 = 0, t1 = 0;
  if (w < active_waves) {
    t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 12; ++u) {
        MFMA(acc[u & 3]);
#pragma unroll
        for (int q = 0; q < K; ++q) {
          const int s = (u * K + q);
          if (KIND == 1) F_MAX(v[s & 7]);
          if (KIND == 2) F_FMA(v[s & 7]);
          if (KIND == 3) {
            const int ph = s % 5, p = (s / 5) & 3;
            if (ph == 0) F_MAX(v[2 * p]);
            if (ph == 1) F_MAX(v[2 * p + 1]);
            if (ph == 2) F_CVT(hi[p], v[2 * p], v[2 * p + 1]);
            if (ph == 3) F_MIXLO(lo[p], hi[p], v[2 * p]);
            if (ph == 4) F_MIXHI(lo[p], hi[p], v[2 * p + 1]);
          }
          if (KIND == 4) F_DSR(d[s & 3], addr);
          if (KIND == 5) F_DSW(addr, d[s & 3]);
        }
      }
      if (KIND == 4 || KIND == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    t1 = clock64();
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a) s += acc[a][0];
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += d[i][0] + __builtin_bit_cast(float, hi[i]) + __builtin_bit_cast(float, lo[i]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

// operand data: constant smooth values (DATA 0), random f16 in [-2, 2) from 4 rotating register sets (1), zeros (2)
template <int DATA>
__global__ __launch_bounds__(512) void mfma_data(float* out, unsigned long long* cyc, int iters, int active_waves) {
  const int w = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  half8 A[4], B[4];
  unsigned s = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + 12345u;
  for (int q = 0; q < 4; ++q)
    for (int e = 0; e < 8; ++e) {
      s = s * 1664525u + 1013904223u;
      const float ra = ((s >> 8) & 0xffff) * (4.0f / 65536.0f) - 2.0f;
      s = s * 1664525u + 1013904223u;
      const float rb = ((s >> 8) & 0xffff) * (4.0f / 65536.0f) - 2.0f;
      A[q][e] = DATA == 1 ? (_Float16)ra : (DATA == 2 ? (_Float16)0.f : (_Float16)(0.001f * ((threadIdx.x & 63) + e)));
      B[q][e] = DATA == 1 ? (_Float16)rb : (DATA == 2 ? (_Float16)0.f : (_Float16)(0.002f * ((threadIdx.x & 63) - e)));
    }
  unsigned long long t0 = 0, t1 = 0;
  if (w < active_waves) {
    t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 12; ++u)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(A[(u + (u >> 2)) & 3]), "v"(B[(u >> 1) & 3]));
      if (DATA == 1 && (it & 63) == 63)   // keep the accumulators finite and busy
        for (int a = 0; a < 4; ++a)
          for (int r = 0; r < 16; ++r) acc[a][r] *= 0.001f;
    }
    t1 = clock64();
  }
  float sum = 0.f;
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) sum += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

// dependent accumulator chains: NACC accumulators used round robin (compiler inserts whatever s_nop it needs)
template <int NACC>
__global__ __launch_bounds__(512) void chain(float* out, unsigned long long* cyc, int iters, int active_waves) {
  const int w = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  half8 A, B;
  for (int e = 0; e < 8; ++e) {
    A[e] = (_Float16)(0.001f * ((threadIdx.x & 63) + e));
    B[e] = (_Float16)(0.002f * ((threadIdx.x & 63) - e));
  }
  unsigned long long t0 = 0, t1 = 0;
  if (w < active_waves) {
    t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 12; ++u)
        acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc[u % NACC], 0, 0, 0);
    }
    t1 = clock64();
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a) s += acc[a][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

static float* g_out;
static int g_iters = 2000;
static unsigned long long* g_cyc;
template <typename F>
void run(const char* name, F kern, int waves) {
  const int iters = g_iters;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, g_out, g_cyc, 200, waves);  // warm
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, g_out, g_cyc, iters, waves);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8];
  (void)hipMemcpy(h, g_cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double per = (double)h[0] / (iters * 12.0);
  const double per_y = waves > 4 ? (double)h[4] / (iters * 12.0) : 0.0;
  const double n_mfma_simd = iters * 12.0 * (waves > 4 ? 2 : 1);
  printf("%-44s waves/SIMD %d  cyc/MFMA old %6.1f young %6.1f | wall %7.3f ms -> %6.1f ns per SIMD-MFMA\n", name,
         waves > 4 ? 2 : 1, per, per_y, ms, ms * 1e6 / n_mfma_simd);
}

#define RUNK(KIND, K, label)                       \
  run(label " x" #K " /MFMA", probe<KIND, K>, 4);  \
  run(label " x" #K " /MFMA", probe<KIND, K>, 8);

int main(int argc, char** argv) {
  if (argc > 1) g_iters = atoi(argv[1]);
  (void)hipMalloc(&g_out, 256 * 512 * 4);
  (void)hipMalloc(&g_cyc, 256 * 8 * 8);
  run("MFMA only (4 acc, asm)", probe<0, 0>, 4);
  run("MFMA only (4 acc, asm)", probe<0, 0>, 8);
  for (int rep = 0; rep < 3; ++rep) {
    run("MFMA only, smooth constant operands", mfma_data<0>, 8);
    run("MFMA only, zero operands", mfma_data<2>, 8);
    run("MFMA only, random operands (4 sets)", mfma_data<1>, 8);
    run("MFMA only, random operands (4 sets)", mfma_data<1>, 4);
  }
  run("chain 1 acc", chain<1>, 4);
  run("chain 1 acc", chain<1>, 8);
  run("chain 2 acc", chain<2>, 4);
  run("chain 2 acc", chain<2>, 8);
  run("chain 4 acc", chain<4>, 4);
  RUNK(1, 1, "v_max_i32") RUNK(1, 2, "v_max_i32") RUNK(1, 3, "v_max_i32") RUNK(1, 4, "v_max_i32")
  RUNK(1, 5, "v_max_i32") RUNK(1, 6, "v_max_i32") RUNK(1, 8, "v_max_i32")
  RUNK(2, 2, "v_fma_f32") RUNK(2, 4, "v_fma_f32") RUNK(2, 6, "v_fma_f32")
  RUNK(3, 1, "split mix") RUNK(3, 2, "split mix") RUNK(3, 3, "split mix") RUNK(3, 4, "split mix")
  RUNK(3, 5, "split mix") RUNK(3, 6, "split mix")
  RUNK(4, 1, "ds_read_b128") RUNK(4, 2, "ds_read_b128") RUNK(4, 3, "ds_read_b128")
  RUNK(5, 1, "ds_write_b128") RUNK(5, 2, "ds_write_b128")
  return 0;
}
