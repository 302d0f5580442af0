// Micro-probe: how much of the MFMA rate of the SDF-MLP tile loop is lost to LDS operand bandwidth?
//   hipcc --offload-arch=gfx950 -O3 tools/probe_lds_mfma.hip -o /tmp/probe_lds && /tmp/probe_lds
// A workgroup of 8 waves (2 per SIMD) runs K-steps of a 256x256 layer on a 128-evaluation tile with the B operands
// (activations, f16 hi and lo planes) in LDS and the A operands in registers.  Register blocking per wave:
//   NB = 4: 32 features x 128 evaluations (what k_lattice_table_h does): 4 B fragments per plane and K-step
//   NB = 2: 64 features x  64 evaluations: 2 B fragments per plane and K-step, 2 A blocks
// NPROD = 3 (split operands: hi and lo planes read, 12 MFMAs per K-step) or 1 (f16 operands: 4 MFMAs per K-step).
// LOAD_A: the weights of every K-step are fetched from an L2-resident buffer instead of staying in registers.
// Prints clock64 ticks per K-step per wave (clock64 does NOT run at the shader clock: its rate is printed, from
// wall_clock64) and the MFMA issue rate in TFLOP/s from the wall clock.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NPROD, int NB, bool LOAD_A>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, const half8* __restrict__ wts) {
  extern __shared__ float lds[];   // [plane][ks 16][h 2][col 128] x 16 B
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2 * 16 * 2 * 128 * 4; i += 512) lds[i] = 0.001f * (i & 255);
  __syncthreads();
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  half8 Ah[2], Al[2];
  for (int e = 0; e < 8; ++e) {
    Ah[0][e] = (_Float16)(0.001f * (lane + e));
    Ah[1][e] = (_Float16)(0.002f * (lane - e));
    Al[0][e] = (_Float16)(0.0001f * (lane + e));
    Al[1][e] = (_Float16)(0.0002f * (lane - e));
  }
  const int j = lane & 31, h = lane >> 5;
  const int col0 = NB == 4 ? 0 : (w & 1) * 64;
  const half8* hi = (const half8*)lds;
  const half8* lo = (const half8*)(lds + 16 * 2 * 128 * 4);
  const unsigned long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      half8 bh[NB], bl[NB];
      if (LOAD_A) {   // weights of this K-step from L2 (coalesced 16 B per lane), hi and lo, 1 or 2 feature blocks
        const int fb = NB == 4 ? w : (w >> 1) * 2;
#pragma unroll
        for (int a = 0; a < (NB == 4 ? 1 : 2); ++a) {
          Ah[a] = wts[(((it & 1) * 16 + ks) * 8 + fb + a) * 128 + lane];
          if (NPROD == 3) Al[a] = wts[(((it & 1) * 16 + ks) * 8 + fb + a) * 128 + 64 + lane];
        }
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int o = (ks * 2 + h) * 128 + col0 + b * 32 + j;
        bh[b] = hi[o];
        if (NPROD == 3) bl[b] = lo[o];
      }
      if (NB == 4) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[0], bh[b], acc[b], 0, 0, 0);
          if (NPROD == 3) {
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[0], bh[b], acc[b], 0, 0, 0);
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[0], bl[b], acc[b], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            acc[a * 2 + b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[a], bh[b], acc[a * 2 + b], 0, 0, 0);
            if (NPROD == 3) {
              acc[a * 2 + b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[a], bh[b], acc[a * 2 + b], 0, 0, 0);
              acc[a * 2 + b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[a], bl[b], acc[a * 2 + b], 0, 0, 0);
            }
          }
      }
      // one LDS read in the shadow of each MFMA, as the real kernel schedules them
      constexpr int kReads = NB * (NPROD == 3 ? 2 : 1);
#pragma unroll
      for (int g = 0; g < kReads; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
  }
  const unsigned long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int a = 0; a < 4; ++a) s += acc[a][lane & 15];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
  if (lane == 0) cyc[gridDim.x * 8 + blockIdx.x * 8 + w] = w1 - w0;   // 100 MHz ticks
}

template <int NPROD, int NB, bool LOAD_A>
static void run(const char* name) {
  const int blocks = 256, iters = 200;
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, blocks * 512 * 4);
  hipMalloc(&cyc, 2 * blocks * 8 * 8);
  const size_t shm = 2 * 16 * 2 * 128 * 16;
  half8* wts;
  hipMalloc(&wts, 2 * 16 * 8 * 128 * 16);
  hipMemset(wts, 0, 2 * 16 * 8 * 128 * 16);
  hipFuncSetAttribute((const void*)k<NPROD, NB, LOAD_A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  for (int rep = 0; rep < 2; ++rep) k<NPROD, NB, LOAD_A><<<blocks, 512, shm>>>(out, cyc, iters, wts);
  hipDeviceSynchronize();
  unsigned long long h[2 * blocks * 8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < blocks * 8; ++i) s += (double)h[i];
  const double per = s / (blocks * 8) / (iters * 16.0);
  const int mf = 4 * NPROD;
  double ws = 0;
  for (int i = 0; i < blocks * 8; ++i) ws += (double)h[blocks * 8 + i];
  const double sec = ws / (blocks * 8) / 100e6;                       // mean wall time of a wave
  const double tflops = (double)blocks * 8 * iters * 16 * mf * 32768.0 / sec / 1e12;
  printf("%-44s %6.0f clock64 ticks per K-step per wave (clock64 runs at %.2f GHz); %.0f TFLOP/s of MFMA issue = %.2f of 2.5 PF\n",
         name, per, s / ws * 0.1, tflops, tflops / 2500.0);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run<3, 4, false>("split, 32 feat x 128 eval per wave, A in regs");
    run<3, 2, false>("split, 64 feat x  64 eval per wave, A in regs");
    run<3, 4, true>("split, 32 x 128, A from L2");
    run<3, 2, true>("split, 64 x  64, A from L2");
    run<1, 4, false>("f16 operands, 32 x 128, A in regs");
    run<1, 2, false>("f16 operands, 64 x  64, A in regs");
    run<1, 4, true>("f16 operands, 32 x 128, A from L2");
    run<1, 2, true>("f16 operands, 64 x  64, A from L2");
  }
  return 0;
}
