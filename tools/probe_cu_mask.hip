// Does hipExtStreamCreateWithCUMask confine a stream's kernels to the masked CUs on this stack, and how are the mask
// bits numbered?  A spin kernel of 2048 single-wave blocks records the HW_ID of every block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
__global__ void k_spin(long long cycles, unsigned* ids) {
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    ids[blockIdx.x] = (hw & 0xffffffu) | ((xcc & 0xf) << 24);
  }
}
int main() {
  int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  printf("CUs %d\n", ncu);
  unsigned* ids; hipMalloc(&ids, 4096 * 4);
  for (int variant = 0; variant < 3; ++variant) {
    std::vector<uint32_t> mask(ncu / 32, 0xffffffffu);
    if (variant == 1) for (auto& m : mask) m = 0x0fffffffu;          // drop the last 4 of every 32
    if (variant == 2) { for (auto& m : mask) m = 0; mask[0] = 0xffffffffu; }   // only bits 0..31
    hipStream_t s; 
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("create failed %d\n", (int)e); return 1; }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_spin, dim3(2048), dim3(64), 0, s, 20000LL, ids);
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    hipLaunchKernelGGL(k_spin, dim3(2048), dim3(64), 0, s, 200000LL, ids);
    hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned> h(2048); hipMemcpy(h.data(), ids, 2048 * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> cus; std::set<unsigned> xccs;
    for (unsigned v : h) { cus.insert(((v >> 24) << 16) | ((v >> 8) & 0xf) | (((v >> 12) & 0x3) << 4) | (((v >> 13) & 0x7) << 6)); xccs.insert(v >> 24); }
    printf("variant %d: %.3f ms, distinct (xcc, se, sh, cu) ids %zu, xccs used %zu\n", variant, ms, cus.size(), xccs.size());
  }
  return 0;
}
