// probe_mark.hip -- what bounds the voxel-marking kernel?  Variants of k_mark on a real frame's points
// (tools: python -c "synthetic.frame(0).tofile('/tmp/pts.bin')"), 256^3 grid, voxel 0.01.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probe_mark.hip -o tools/probe_mark && tools/probe_mark /tmp/pts.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

struct Grid { float bmin[3], lo[3], hi[3], v; int n[3]; };

__device__ __forceinline__ bool corners(const float* __restrict__ pts, int i, int n_points, const Grid& g, int (&f)[3], int (&c)[3]) {
  if (i >= n_points) return false;
  const float x = pts[(size_t)i * 6], y = pts[(size_t)i * 6 + 1], z = pts[(size_t)i * 6 + 2];
  if (!((x < g.hi[0]) && (y < g.hi[1]) && (z < g.hi[2]) && (x > g.lo[0]) && (y > g.lo[1]) && (z > g.lo[2]))) return false;
  const float p[3] = {x, y, z};
  for (int a = 0; a < 3; ++a) {
    const float t = __fdiv_rn(__fsub_rn(p[a], g.bmin[a]), g.v);
    f[a] = (int)floorf(t);
    c[a] = (int)ceilf(t);
  }
  return true;
}

// mode bits: 1 = n_valid atomic per wave, 2 = visibility loads, 4 = bitmap atomics, 8 = byte-map plain stores,
// 16 = lane dedup, 32 = per-block n_valid (LDS) instead of per wave
template <int MODE>
__global__ __launch_bounds__(256) void k_mark(const float* __restrict__ pts, int n_points, Grid g, uint32_t* __restrict__ bitmap,
                                              uint8_t* __restrict__ bytemap, int* __restrict__ n_valid, int* __restrict__ sink) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  int f[3] = {0, 0, 0}, c[3] = {0, 0, 0};
  const bool valid = corners(pts, i, n_points, g, f, c);
  const int nyz = g.n[1] * g.n[2];
  const int lane = threadIdx.x & 63;
  int acc = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int gx = (k & 1) ? c[0] : f[0], gy = (k & 2) ? c[1] : f[1];
    const bool dup = ((k & 1) && c[0] == f[0]) || ((k & 2) && c[1] == f[1]);
    int a = (valid && !dup) ? (gx * nyz + gy * g.n[2] + f[2]) : -1;
    int b = (a >= 0 && c[2] != f[2]) ? a + 1 : -1;
    if (MODE & 16) {
      const int p0 = __shfl_up(a, 1), p1 = __shfl_up(b, 1);
      if (lane > 0 && p0 == a && p1 == b) a = b = -1;
    }
    if (a < 0) continue;
    if (MODE & 8) {
      bytemap[a] = 1;
      if (b >= 0) bytemap[b] = 1;
    }
    if (MODE & 4) {
      const uint32_t bit0 = 1u << (a & 31);
      if (b >= 0 && (b >> 5) == (a >> 5)) {
        const uint32_t bits = bit0 | (1u << (b & 31));
        if (!(MODE & 2) || (bitmap[a >> 5] & bits) != bits) atomicOr(&bitmap[a >> 5], bits);
      } else {
        if (!(MODE & 2) || !(bitmap[a >> 5] & bit0)) atomicOr(&bitmap[a >> 5], bit0);
        if (b >= 0) {
          const uint32_t bit1 = 1u << (b & 31);
          if (!(MODE & 2) || !(bitmap[b >> 5] & bit1)) atomicOr(&bitmap[b >> 5], bit1);
        }
      }
    }
    acc += a;
  }
  if (MODE & 1) {
    const unsigned long long bal = __ballot(valid);
    if (lane == 0 && bal) atomicAdd(n_valid, (int)__popcll(bal));
  }
  if (MODE & 32) {
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const unsigned long long bal = __ballot(valid);
    if (lane == 0 && bal) atomicAdd(&s_cnt, (int)__popcll(bal));
    __syncthreads();
    if (threadIdx.x == 0) sink[1 + blockIdx.x] = s_cnt;
  }
  if (acc == 0x7fffffff) sink[0] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>
static void run(const char* name, const float* pts, int n, Grid g, uint32_t* bitmap, uint8_t* bytemap, int* ctr, int* sink) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f, sum = 0.f;
  for (int rep = 0; rep < 12; ++rep) {
    CK(hipMemset(bitmap, 0, 2 << 20));
    CK(hipMemset(bytemap, 0, 16 << 20));
    CK(hipMemset(ctr, 0, 4));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_mark<MODE>, dim3((n + 255) / 256), dim3(256), 0, 0, pts, n, g, bitmap, bytemap, ctr, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
  }
  int nv = 0; CK(hipMemcpy(&nv, ctr, 4, hipMemcpyDeviceToHost));
  std::vector<uint32_t> h(1 << 19);
  CK(hipMemcpy(h.data(), bitmap, 2 << 20, hipMemcpyDeviceToHost));
  long bits = 0; for (uint32_t w : h) bits += __builtin_popcount(w);
  std::vector<uint8_t> hb(16 << 20);
  CK(hipMemcpy(hb.data(), bytemap, 16 << 20, hipMemcpyDeviceToHost));
  long bytes = 0; for (uint8_t w : hb) bytes += w;
  printf("%-58s avg %7.1f us  min %7.1f us   n_valid %d  bits %ld  bytes %ld\n", name, 1e3f * sum / 10, 1e3f * best, nv, bits, bytes);
}

int main(int argc, char** argv) {
  FILE* fp = fopen(argc > 1 ? argv[1] : "/tmp/pts.bin", "rb");
  if (!fp) { printf("no points file\n"); return 1; }
  std::vector<float> h(307200 * 6);
  const size_t got = fread(h.data(), 4, h.size(), fp);
  fclose(fp);
  const int n = (int)(got / 6);
  Grid g;
  const float dims = 2.54f, v = 0.01f;
  for (int a = 0; a < 3; ++a) {
    g.bmin[a] = (float)(-(double)dims / 2 - (double)v);
    g.lo[a] = g.bmin[a] + v;
    g.hi[a] = (float)(-(double)dims / 2 - (double)v + 0.01 * 256) - v;
    g.n[a] = 256;
  }
  g.v = v;
  float* pts; uint32_t* bitmap; uint8_t* bytemap; int *ctr, *sink;
  CK(hipMalloc(&pts, h.size() * 4)); CK(hipMalloc(&bitmap, 2 << 20)); CK(hipMalloc(&bytemap, 16 << 20));
  CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&sink, 8192));
  CK(hipMemcpy(pts, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  printf("%d points\n", n);
  run<0>("loads + voxelisation only", pts, n, g, bitmap, bytemap, ctr, sink);
  run<1>("+ n_valid atomic per wave", pts, n, g, bitmap, bytemap, ctr, sink);
  run<32>("+ n_valid per block (LDS, plain store)", pts, n, g, bitmap, bytemap, ctr, sink);
  run<16 | 4>("lane dedup + bitmap atomics (no visibility loads)", pts, n, g, bitmap, bytemap, ctr, sink);
  run<16 | 4 | 2>("lane dedup + visibility loads + bitmap atomics", pts, n, g, bitmap, bytemap, ctr, sink);
  run<16 | 4 | 2 | 1>("... + n_valid atomic per wave (= current k_mark)", pts, n, g, bitmap, bytemap, ctr, sink);
  run<4>("bitmap atomics, no dedup, no visibility", pts, n, g, bitmap, bytemap, ctr, sink);
  run<8>("byte map plain stores, no dedup", pts, n, g, bitmap, bytemap, ctr, sink);
  run<16 | 8>("byte map plain stores, lane dedup", pts, n, g, bitmap, bytemap, ctr, sink);
  run<16 | 8 | 32>("byte map plain stores, lane dedup, n_valid per block", pts, n, g, bitmap, bytemap, ctr, sink);
  return 0;
}
