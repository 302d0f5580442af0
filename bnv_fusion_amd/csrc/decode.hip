// decode.hip -- SDF decode (reference sparse_volume.py:768-833, local_point_fusion.py:265-379,
// modules.py:81-123,657-662) on gfx950.
//
// One MLP core, 17 -> 256 -> 256 -> 256 -> 256 -> 1 in exact fp32 on v_mfma_f32_32x32x2_f32:
//   * a workgroup (8 waves) evaluates a tile of 128 inputs; wave w owns output features
//     [32w, 32w+32) of every layer for all 128 inputs (4 MFMA column tiles of 32);
//   * activations live in LDS as HL[kb][h][j][4] (feature 8kb+4h+i of input j): a wave reads its
//     B operands with conflict-free ds_read_b128 and writes its D registers back with
//     ds_write_b128 -- D register 4q+i of lane (j,h) IS feature 32w+8q+4h+i, the same layout;
//   * weights stream from L2 (0.8 MB, resident in every XCD's 4 MB L2) as one coalesced
//     dwordx4 per lane per 8-deep K block, pre-permuted on the host (weights.py: pack_sdf_mlp).
// Three front/back-ends share it:
//   PTS      SparseVolume.decode_pts at arbitrary points: 8 corner evaluations per point;
//   LATTICE  per-voxel table g[row][27] = MLP(enc(l), feat[row]) * voxel, l in {-.5,0,.5}^3 --
//            on the 3x3x3 meshing lattice every (point, corner) input is one of those 27 per
//            corner voxel, so the MLP runs 27x per corner voxel instead of 216x per voxel;
//   DENSE    decode_feature_grid_w_pts on dense grids.
#include <string.h>

#include <type_traits>

#include "bnv_common.hpp"

namespace bnv {

constexpr int DM = 128;  // MLP inputs per tile

// packed SDF-MLP weights (floats)
constexpr int SD_W0 = 0;                      // [8 w][3 kb][64 lane][4]
constexpr int SD_W1 = SD_W0 + 8 * 3 * 256;    // [8 w][32 kb][64 lane][4]
constexpr int SD_W2 = SD_W1 + 65536;
constexpr int SD_W3 = SD_W2 + 65536;
constexpr int SD_B0 = SD_W3 + 65536;          // [256] x 4
constexpr int SD_WA = SD_B0 + 4 * 256;        // fc_alpha weight [256]
constexpr int SD_BA = SD_WA + 256;            // fc_alpha bias, padded to 4; [1] = certified |feature| bound (below)
constexpr int SD_TOTAL = SD_BA + 4;
// split-operand variant, appended to the same pack (units: 16-bit halves from float offset SD_TOTAL)
constexpr int SH_W0 = 0;                          // [8 w][2 ks][2 hi/lo][64 lane][8]
constexpr int SH_W1 = SH_W0 + 8 * 2 * 2 * 64 * 8; // [8 w][16 ks][2 hi/lo][64 lane][8]
constexpr int SH_W2 = SH_W1 + 8 * 16 * 2 * 64 * 8;
constexpr int SH_W3 = SH_W2 + 8 * 16 * 2 * 64 * 8;
constexpr int SH_TOTAL = SH_W3 + 8 * 16 * 2 * 64 * 8;   // 409,600 halves
constexpr int SD_PACK_FLOATS = SD_TOTAL + SH_TOTAL / 2;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// weight fragment fetch through a buffer descriptor: wave-uniform base (SGPRs) + one shared per-lane
// byte offset + a scalar offset per load -- no 64-bit address VGPR per load (with flat loads the compiler
// hoists dozens of lane-constant addresses out of the tile loop and spills them)
__device__ __forceinline__ half8 load_frag(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
  return __builtin_bit_cast(half8, v);
}


// LDS (floats)
constexpr int L_HL = 0;                       // [32 kb][2 h][128 j][4]
constexpr int L_PART = L_HL + 32 * 2 * DM * 4;  // [8 w][2 h][128]
constexpr int L_ALPHA = L_PART + 16 * DM;     // [128]
constexpr int L_WTRI = L_ALPHA + DM;          // [128] trilinear weight of the evaluation
constexpr int L_WVOL = L_WTRI + DM;           // [128] volume weight (or dense count) of its corner
constexpr int L_DELTA = L_WVOL + DM;          // [128] sdf_delta sample of its corner
constexpr int L_TOTAL = L_DELTA + DM;         // 35,328 floats = 141,312 B

// MODE_PTS runs k_decode_pts, the others k_decode.  MODE_DENSE1: the two one-evaluation-per-query branches of
// decode_feature_grid_w_pts (DecodeArgs::variant).
enum { MODE_PTS = 0, MODE_LATTICE = 1, MODE_DENSE = 2, MODE_DENSE1 = 3 };

// Phase timing of the decode tile loop (development builds only: -DBNV_PHASE_PROF, tools/phase_prof.py).
// Thread 0 of every workgroup accumulates shader-clock deltas per phase in LDS and adds them to
// g_phase_cycles at kernel end.
#ifdef BNV_PHASE_PROF
constexpr int L_PROF = L_TOTAL;  // [8 waves][32] x u64 behind the regular LDS layout
__device__ unsigned long long g_phase_cycles[8 * 32];
#define BNV_PH(i)                                                                 \
  do {                                                                            \
    if ((threadIdx.x & 63) == 0) {                                                \
      unsigned long long* _p = (unsigned long long*)(lds + L_PROF) + (threadIdx.x >> 6) * 32; \
      const unsigned long long _t = clock64();                                    \
      _p[i] += _t - _p[31];                                                       \
      _p[31] = _t;                                                                \
    }                                                                             \
  } while (0)
#else
#define BNV_PH(i)
#endif

struct DecodeArgs {
  bnv_volume_t vol;
  bnv_grid_t grid;
  const float* features;
  const float* weights;
  int64_t row_limit;
  const float* pack;
  const float* coords;
  int64_t n;
  int is_coords;
  bnv_sdf_delta_t delta;
  float* out;
  // LATTICE: work list of rows (27 evaluations each) or, if `entries` is set, of (row << 5 | l) entries
  const int32_t* list;
  const int32_t* n_list;
  float* table;
  const int32_t* entries;
  uint32_t* need_mask;
  // DENSE
  const float* feat_grid;
  const float* pts_weight;
  int32_t dims[3];
  // DENSE1: 0 = nearest voxel (interpolate_decode=False), 1 = global coordinates (trilinear features)
  int32_t variant;
  float* nf_out;     // optional [n, 8]: the features the evaluation used
  int32_t* status;   // optional: [1] = 5 when a feature leaves the certified range of the split arithmetic
  int32_t half_tail; // k_lattice_table_x: hand the last partial round out as 64-evaluation tiles
  // PTS, several ray splits of an optimiser step in ONE call (bnv_optim_step, bnv_decode_pts_splits): query q belongs
  // to split q / split_samples; bit s of split_mask[row] = split s touches the row (bnv_volume_count_optim_splits).
  // The weight the mask decision of a split-s query sees is weights[row] + 1 for every split <= s that touches the
  // row -- count_optim (sparse_volume.py:602-622) called split by split, render_utils.py:491-497.  Null: plain weights.
  const uint32_t* split_mask;
  int64_t split_samples;
};

__device__ __forceinline__ f32x16 frag256(const float* __restrict__ b, int w, int h) {
  f32x16 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 t = *(const f32x4*)&b[w * 32 + 8 * q + 4 * h];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[4 * q + i] = t[i];
  }
  return v;
}

template <int NKB>
__device__ __forceinline__ void mlp_layer(const float* __restrict__ wp, const float* __restrict__ bias,
                                          const float* __restrict__ hl, f32x16 (&acc)[4], int w, int lane,
                                          int j, int h) {
  const f32x16 b0 = frag256(bias, w, h);
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) acc[pt] = b0;
  const float* wl = wp + (size_t)w * NKB * 256 + lane * 4;
  const float* hb = hl + (h * DM + j) * 4;
#pragma unroll 4
  for (int kb = 0; kb < NKB; ++kb) {
    const f32x4 a = *(const f32x4*)(wl + kb * 256);
    f32x4 b[4];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) b[pt] = *(const f32x4*)(hb + (kb * 2 * DM + pt * 32) * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int pt = 0; pt < 4; ++pt)
        acc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[pt][i], acc[pt], 0, 0, 0);
    }
  }
}

__device__ __forceinline__ void store_relu(float* __restrict__ hl, const f32x16 (&acc)[4], int w, int j, int h) {
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = relu_bits(acc[pt][4 * q + i]);
      *(f32x4*)&hl[(((4 * w + q) * 2 + h) * DM + pt * 32 + j) * 4] = v;
    }
  }
}

// Runs the MLP on the 128 inputs staged in HL[kb 0..2]; leaves alpha[128] (raw network output).
__device__ __forceinline__ void sdf_mlp_tile(float* __restrict__ lds, const float* __restrict__ pack) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, h = lane >> 5;
  float* hl = lds + L_HL;
  f32x16 acc[4];
  mlp_layer<3>(pack + SD_W0, pack + SD_B0, hl, acc, w, lane, j, h);
  __syncthreads();
  store_relu(hl, acc, w, j, h);
  __syncthreads();
  mlp_layer<32>(pack + SD_W1, pack + SD_B0 + 256, hl, acc, w, lane, j, h);
  __syncthreads();
  store_relu(hl, acc, w, j, h);
  __syncthreads();
  mlp_layer<32>(pack + SD_W2, pack + SD_B0 + 512, hl, acc, w, lane, j, h);
  __syncthreads();
  store_relu(hl, acc, w, j, h);
  __syncthreads();
  mlp_layer<32>(pack + SD_W3, pack + SD_B0 + 768, hl, acc, w, lane, j, h);
  // fc_alpha: 256 -> 1.  Each lane reduces its 16 features, partials are summed in a fixed order.
  const f32x16 wa = frag256(pack + SD_WA, w, h);
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s = fmaf(wa[r], relu_bits(acc[pt][r]), s);
    lds[L_PART + (w * 2 + h) * DM + pt * 32 + j] = s;
  }
  __syncthreads();
  if (threadIdx.x < DM) {
    float s = pack[SD_BA];
#pragma unroll
    for (int p = 0; p < 16; ++p) s += lds[L_PART + p * DM + threadIdx.x];
    lds[L_ALPHA + threadIdx.x] = s;
  }
  __syncthreads();
}

// writes the 17 network inputs [local(3), sin(3), cos(3), feat(8)] of evaluation j into HL
__device__ __forceinline__ void stage_input(float* __restrict__ hl, int j, const float (&loc)[3],
                                            const float (&feat)[8]) {
  const float s0 = sinf(loc[0]), s1 = sinf(loc[1]), s2 = sinf(loc[2]);
  const float c0 = cosf(loc[0]), c1 = cosf(loc[1]), c2 = cosf(loc[2]);
  const f32x4 v0 = {loc[0], loc[1], loc[2], s0};
  const f32x4 v1 = {s1, s2, c0, c1};
  const f32x4 v2 = {c2, feat[0], feat[1], feat[2]};
  const f32x4 v3 = {feat[3], feat[4], feat[5], feat[6]};
  const f32x4 v4 = {feat[7], 0.f, 0.f, 0.f};
  const f32x4 v5 = {0.f, 0.f, 0.f, 0.f};
  *(f32x4*)&hl[((0 * 2 + 0) * DM + j) * 4] = v0;
  *(f32x4*)&hl[((0 * 2 + 1) * DM + j) * 4] = v1;
  *(f32x4*)&hl[((1 * 2 + 0) * DM + j) * 4] = v2;
  *(f32x4*)&hl[((1 * 2 + 1) * DM + j) * 4] = v3;
  *(f32x4*)&hl[((2 * 2 + 0) * DM + j) * 4] = v4;
  *(f32x4*)&hl[((2 * 2 + 1) * DM + j) * 4] = v5;
}


// ---- split-operand MLP core: x = hi + lo (f16), a.b ~ ah.bh + ah.bl + al.bh on the f16 MFMA ----
// LDS: HH[ks][h][j][8 halves] (hi) at L_HL, HLo (lo) 64 KB behind it; slot jj of lane half h in
// K-step ks is feature 16 ks + 8 (jj >> 2) + 4 h + (jj & 3), which makes D registers 8 ksl .. 8 ksl+7
// of wave w exactly the 8 slots of K-step 2 w + ksl.
constexpr int L_HLO = L_HL + 16 * 2 * DM * 4;  // float offset of the lo plane

__device__ __forceinline__ float relu1(float x) { return relu_bits(x); }

#ifndef BNV_A_AHEAD
#define BNV_A_AHEAD 2
#endif
// NPROD = 3: split operands (al.bh + ah.bl + ah.bh); NPROD = 1: f16 operands (ah.bh only; MLP mode 3)
template <int NKS, bool BIAS = true, int NPROD = 3>
__device__ __forceinline__ void mlp_layer_h(const _Float16* __restrict__ wp, const float* __restrict__ bias,
                                            const float* __restrict__ lds, f32x16 (&acc)[4], int w, int lane,
                                            int j, int h) {
  f32x16 b0;
  if constexpr (BIAS) {
    b0 = frag256(bias, w, h);
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) b0[r] = 0.f;
  }
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) acc[pt] = b0;
  const _Float16* wl = wp + (size_t)w * NKS * 2 * 64 * 8 + lane * 8;
  const float* hh = lds + L_HL + (h * DM + j) * 4;
  const float* hl = lds + L_HLO + (h * DM + j) * 4;
  // software pipeline over the K-steps (fully unrolled, all indices static): weight fragments come
  // from L2 kAhead steps ahead (register ring), activation fragments from LDS one step ahead
  constexpr int kAhead = BNV_A_AHEAD, kRing = kAhead + 1;
  half8 ah[kRing], al[kRing], bh[2][4], bl[2][4];
#define BNV_LOAD_A(ks)                                                                  \
  {                                                                                     \
    ah[(ks) % kRing] = *(const half8*)(wl + ((ks) * 2) * 64 * 8);                       \
    if (NPROD == 3) al[(ks) % kRing] = *(const half8*)(wl + ((ks) * 2 + 1) * 64 * 8);   \
  }
#define BNV_LOAD_B(ks)                                                                    \
  {                                                                                       \
    _Pragma("unroll") for (int pt = 0; pt < 4; ++pt) {                                    \
      bh[(ks) & 1][pt] = *(const half8*)(hh + ((ks) * 2 * DM + pt * 32) * 4);             \
      if (NPROD == 3) bl[(ks) & 1][pt] = *(const half8*)(hl + ((ks) * 2 * DM + pt * 32) * 4); \
    }                                                                                     \
  }
#pragma unroll
  for (int p = 0; p < kAhead; ++p)
    if (p < NKS) BNV_LOAD_A(p);
  BNV_LOAD_B(0);
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    if (ks + kAhead < NKS) BNV_LOAD_A(ks + kAhead);
    if (ks + 1 < NKS) BNV_LOAD_B(ks + 1);
    const half8 a_hi = ah[ks % kRing];
    if constexpr (NPROD == 3) {
      const half8 a_lo = al[ks % kRing];
#pragma unroll
      for (int pt = 0; pt < 4; ++pt)
        acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, bh[ks & 1][pt], acc[pt], 0, 0, 0);
#pragma unroll
      for (int pt = 0; pt < 4; ++pt)
        acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, bl[ks & 1][pt], acc[pt], 0, 0, 0);
    }
#pragma unroll
    for (int pt = 0; pt < 4; ++pt)
      acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, bh[ks & 1][pt], acc[pt], 0, 0, 0);
    // Issue order inside the step: every prefetch goes into the shadow of an MFMA (one memory instruction
    // behind each MFMA).  A wave then keeps the MFMA pipe busy on its own; with all the loads clustered at
    // the top of the step a lone wave reached only 55-70 % (tools/phase_prof.py).
    constexpr int kDs = NPROD == 3 ? 8 : 4, kVm = NPROD == 3 ? 2 : 1;
    if (ks + 1 < NKS) {
#pragma unroll
      for (int g = 0; g < (NPROD == 3 ? kDs : kDs - 1); ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
      }
      if (NPROD == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    if (ks + kAhead < NKS) {
#pragma unroll
      for (int g = 0; g < kVm; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#undef BNV_LOAD_A
#undef BNV_LOAD_B
}

template <int NPROD = 3>
__device__ __forceinline__ void store_relu_h(float* __restrict__ lds, const f32x16 (&acc)[4], int w, int j, int h) {
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) {
#pragma unroll
    for (int ksl = 0; ksl < 2; ++ksl) {
      half8 hi, lo;
      if (NPROD == 3) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = relu1(acc[pt][8 * ksl + e]);
        split8_f16(x, hi, lo);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) hi[e] = (_Float16)relu1(acc[pt][8 * ksl + e]);
      }
      const int o = (((2 * w + ksl) * 2 + h) * DM + pt * 32 + j) * 4;
      *(half8*)&lds[L_HL + o] = hi;
      if (NPROD == 3) *(half8*)&lds[L_HLO + o] = lo;
    }
  }
}

template <int NPROD = 3>
__device__ __forceinline__ void sdf_mlp_tile_h(float* __restrict__ lds, const float* __restrict__ pack) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, h = lane >> 5;
  const _Float16* ph = (const _Float16*)(pack + SD_TOTAL);
  f32x16 acc[4];
  mlp_layer_h<2, true, NPROD>(ph + SH_W0, pack + SD_B0, lds, acc, w, lane, j, h);
  BNV_PH(1);
  __syncthreads();
  BNV_PH(2);
  store_relu_h<NPROD>(lds, acc, w, j, h);
  BNV_PH(3);
  __syncthreads();
  BNV_PH(4);
  mlp_layer_h<16, true, NPROD>(ph + SH_W1, pack + SD_B0 + 256, lds, acc, w, lane, j, h);
  BNV_PH(5);
  __syncthreads();
  BNV_PH(6);
  store_relu_h<NPROD>(lds, acc, w, j, h);
  BNV_PH(7);
  __syncthreads();
  BNV_PH(8);
  mlp_layer_h<16, true, NPROD>(ph + SH_W2, pack + SD_B0 + 512, lds, acc, w, lane, j, h);
  BNV_PH(9);
  __syncthreads();
  BNV_PH(10);
  store_relu_h<NPROD>(lds, acc, w, j, h);
  BNV_PH(11);
  __syncthreads();
  BNV_PH(12);
  mlp_layer_h<16, true, NPROD>(ph + SH_W3, pack + SD_B0 + 768, lds, acc, w, lane, j, h);
  BNV_PH(13);
  const f32x16 wa = frag256(pack + SD_WA, w, h);
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s = fmaf(wa[r], relu_bits(acc[pt][r]), s);
    lds[L_PART + (w * 2 + h) * DM + pt * 32 + j] = s;
  }
  BNV_PH(14);
  __syncthreads();
  BNV_PH(15);
  if (threadIdx.x < DM) {
    float s = pack[SD_BA];
#pragma unroll
    for (int p = 0; p < 16; ++p) s += lds[L_PART + p * DM + threadIdx.x];
    lds[L_ALPHA + threadIdx.x] = s;
  }
  __syncthreads();
  BNV_PH(16);
}

// Range certificate of the f16-split arithmetic (MLP modes 1 and 3; weights.py: certified_input_bound): with the
// local coordinates and their sin / cos in [-1, 1] and |feature| <= pack[SD_BA + 1], no value of any layer can
// reach the f16 overflow threshold (65,520), where fp32 -- the reference's arithmetic -- would still be fine.  A
// feature row beyond the bound (or NaN) sets the volume's sticky error word to 5 instead of silently producing
// inf / NaN: the caller then switches to exact fp32 (bnv_set_mlp_mode(0)).  8 compares per EVALUATION, not per
// activation: free.
__device__ __forceinline__ void check_feature_range(const float (&feat)[8], float bound, int32_t* __restrict__ status) {
  float m = fmaxf(fabsf(feat[0]), fabsf(feat[1]));
#pragma unroll
  for (int f = 2; f < 8; ++f) m = fmaxf(m, fabsf(feat[f]));
  bool bad = !(m <= bound);
#pragma unroll
  for (int f = 0; f < 8; ++f) bad = bad || (feat[f] != feat[f]);   // fmaxf drops NaNs
  if (bad && status) status[1] = 5;
}

// inputs of evaluation j in the split layout: features 0..16 (+15 zero) over K-steps 0, 1
template <int NPROD = 3>
__device__ __forceinline__ void stage_input_h(float* __restrict__ lds, int j, const float (&loc)[3],
                                              const float (&feat)[8]) {
  float in[32];
#pragma unroll
  for (int f = 0; f < 32; ++f) in[f] = 0.f;
  in[0] = loc[0]; in[1] = loc[1]; in[2] = loc[2];
  in[3] = sinf(loc[0]); in[4] = sinf(loc[1]); in[5] = sinf(loc[2]);
  in[6] = cosf(loc[0]); in[7] = cosf(loc[1]); in[8] = cosf(loc[2]);
#pragma unroll
  for (int f = 0; f < 8; ++f) in[9 + f] = feat[f];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      half8 hi, lo;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const float x = in[16 * ks + 8 * (jj >> 2) + 4 * hh + (jj & 3)];
        const _Float16 t = (_Float16)x;
        hi[jj] = t;
        if (NPROD == 3) lo[jj] = (_Float16)(x - (float)t);
      }
      const int o = ((ks * 2 + hh) * DM + j) * 4;
      *(half8*)&lds[L_HL + o] = hi;
      if (NPROD == 3) *(half8*)&lds[L_HLO + o] = lo;
    }
  }
}


// ---- tiny-cuda-nn SDF decoder (reference default checkpoint; tcnnNeRFModel, modules.py:136-253):
// 17 inputs padded to 32 with 1.0 -> 64 -> 64 -> 64 -> 16 (output 0 used), ReLU, no bias, fp16.
// The network is small enough that ONE wave runs all layers for 32 evaluations in registers (no
// barriers between layers); waves 0..3 of the workgroup cover the tile's 128 evaluations.
// Pack (halves): W0 [2 mb][2 ks][64 lane][8] | W1, W2 [2 mb][4 g][64][8] | W3 [4 g][64][8] (rows >= 16 zero).
constexpr int ST_W0 = 0;
constexpr int ST_W1 = ST_W0 + 2 * 2 * 64 * 8;
constexpr int ST_W2 = ST_W1 + 2 * 4 * 64 * 8;
constexpr int ST_W3 = ST_W2 + 2 * 4 * 64 * 8;
constexpr int ST_TOTAL = ST_W3 + 4 * 64 * 8;  // 12,288 halves
static_assert(ST_TOTAL == 12288, "tcnn SDF pack size (weights.py: pack_sdf_tcnn)");

__device__ __forceinline__ half8 relu_half8(const f32x16& v, int base) {
  half8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (_Float16)relu1(v[base + e]);
  return r;
}

__device__ __forceinline__ void sdf_mlp_tile_t(float* __restrict__ lds, const float* __restrict__ pack) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, h = lane >> 5;
  if (w < 4) {
    const _Float16* ph = (const _Float16*)pack;
    const int col = w * 32 + j;
    f32x16 a0[2], a1[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) a0[mb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const half8 b = *(const half8*)&lds[L_HL + ((ks * 2 + h) * DM + col) * 4];
        a0[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8*)&ph[ST_W0 + ((mb * 2 + ks) * 64 + lane) * 8], b,
                                                        a0[mb], 0, 0, 0);
      }
    }
    half8 s[4];
    auto layer64 = [&](int woff, const f32x16 (&in)[2], f32x16 (&out)[2]) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        s[nb * 2] = relu_half8(in[nb], 0);
        s[nb * 2 + 1] = relu_half8(in[nb], 8);
      }
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) out[mb][r] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          out[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8*)&ph[woff + ((mb * 4 + g) * 64 + lane) * 8],
                                                          s[g], out[mb], 0, 0, 0);
      }
    };
    layer64(ST_W1, a0, a1);
    layer64(ST_W2, a1, a0);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      s[nb * 2] = relu_half8(a0[nb], 0);
      s[nb * 2 + 1] = relu_half8(a0[nb], 8);
    }
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      o = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8*)&ph[ST_W3 + (g * 64 + lane) * 8], s[g], o, 0, 0, 0);
    // output 0 = row 0 of the tile = register 0 of the lanes with h == 0; the network returns fp16
    if (h == 0) lds[L_ALPHA + col] = (float)(_Float16)o[0];
  }
  __syncthreads();
}

// inputs of evaluation j for the tcnn decoder: 17 features, padded to 32 with 1.0, f16
__device__ __forceinline__ void stage_input_t(float* __restrict__ lds, int j, const float (&loc)[3],
                                              const float (&feat)[8]) {
  float in[32];
#pragma unroll
  for (int f = 0; f < 32; ++f) in[f] = 1.0f;
  in[0] = loc[0]; in[1] = loc[1]; in[2] = loc[2];
  in[3] = sinf(loc[0]); in[4] = sinf(loc[1]); in[5] = sinf(loc[2]);
  in[6] = cosf(loc[0]); in[7] = cosf(loc[1]); in[8] = cosf(loc[2]);
#pragma unroll
  for (int f = 0; f < 8; ++f) in[9 + f] = feat[f];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      half8 v;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) v[jj] = (_Float16)in[16 * ks + 8 * (jj >> 2) + 4 * hh + (jj & 3)];
      *(half8*)&lds[L_HL + ((ks * 2 + hh) * DM + j) * 4] = v;
    }
  }
}

// F.grid_sample(mode="nearest", padding_mode="zeros", align_corners=True) of the TSDF prior at a
// corner given in voxel units (sparse_volume.py:820-829): coordinate a -> index along dims[a].
__device__ __forceinline__ float sample_delta(const bnv_sdf_delta_t& d, const bnv_grid_t& g, const float (&c)[3]) {
  int idx[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float t = __fdiv_rn(c[a], (float)(g.n_xyz[a] - 1));
    t = __fsub_rn(__fmul_rn(t, 2.f), 1.f);
    t = __fmul_rn(__fdiv_rn(__fadd_rn(t, 1.f), 2.f), (float)(d.dims[a] - 1));
    const float r = nearbyintf(t);
    if (!(r >= 0.f) || !(r <= (float)(d.dims[a] - 1))) return 0.f;
    idx[a] = (int)r;
  }
  return d.data[((size_t)idx[0] * d.dims[1] + idx[1]) * d.dims[2] + idx[2]];
}

template <int MODE, int PREC>
__global__ __launch_bounds__(512, 2) void k_decode(DecodeArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* hl = lds + L_HL;
  const float voxel = A.grid.voxel_size;
  int64_t n_tiles;
  int64_t n_evals = 0;
  if constexpr (MODE == MODE_LATTICE) {
    n_evals = A.entries ? (int64_t)A.n_list[1] : (int64_t)(*A.n_list) * 27;
    n_tiles = (n_evals + DM - 1) / DM;
  } else if constexpr (MODE == MODE_DENSE1) {
    n_tiles = (A.n + DM - 1) / DM;
  } else {
    n_tiles = (A.n + 15) / 16;
  }
#ifdef BNV_PHASE_PROF
  if (threadIdx.x < 256) ((unsigned long long*)(lds + L_PROF))[threadIdx.x] = 0;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) ((unsigned long long*)(lds + L_PROF))[(threadIdx.x >> 6) * 32 + 31] = clock64();
#endif
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    // ---------------- front end: one thread per MLP input ---------------------------------
    if (threadIdx.x < DM) {
      const int j = threadIdx.x;
      float loc[3] = {0.f, 0.f, 0.f};
      float feat[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float wtri = 0.f, wvol = 0.f, dlt = 0.f;
      if constexpr (MODE == MODE_LATTICE) {
        const int64_t e = tile * DM + j;
        if (e < n_evals) {
          int row, l;
          if (A.entries) {
            const int ent = A.entries[e];
            row = ent >> 5;
            l = ent & 31;
          } else {
            const int64_t ci = e / 27;
            l = (int)(e - ci * 27);
            row = A.list[ci];
          }
          loc[0] = (float)(l / 9 - 1) * 0.5f;
          loc[1] = (float)((l / 3) % 3 - 1) * 0.5f;
          loc[2] = (float)(l % 3 - 1) * 0.5f;
          const f32x4 f0 = *(const f32x4*)&A.features[(size_t)row * 8];
          const f32x4 f1 = *(const f32x4*)&A.features[(size_t)row * 8 + 4];
#pragma unroll
          for (int f = 0; f < 4; ++f) {
            feat[f] = f0[f];
            feat[4 + f] = f1[f];
          }
        }
      } else if constexpr (MODE == MODE_DENSE1) {
        const int64_t q = tile * DM + j;
        if (q < A.n) {
          const size_t plane = (size_t)A.dims[0] * A.dims[1] * A.dims[2];
          float c[3];
#pragma unroll
          for (int a = 0; a < 3; ++a) c[a] = A.coords[q * 3 + a];
          if (A.variant == 0) {
            // nearest voxel, one evaluation (local_point_fusion.py:288-292, 331-343): torch.round = half to even
            int v[3];
            bool in = true;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              const float r = rintf(c[a]);
              // relative_xyz = rel * voxel; decode_implicit divides it by voxel again (:335,:374)
              loc[a] = __fdiv_rn(__fmul_rn(__fsub_rn(c[a], r), voxel), voxel);
              in = in && r >= 0.f && r <= (float)(A.dims[a] - 1);
              v[a] = (int)r;
            }
            if (in) {
              const size_t o = ((size_t)v[0] * A.dims[1] + v[1]) * A.dims[2] + v[2];
#pragma unroll
              for (int f = 0; f < 8; ++f) feat[f] = A.feat_grid[f * plane + o];
              wvol = A.pts_weight[o];
            }
          } else {
            // global coordinates (:345-367): features by trilinear grid_sample (align_corners, zero padding: the
            // order of torch's grid_sampler_3d, x = last axis), weight by nearest; the MLP sees coords / (res - 1)
            float u[3], fl[3];
            bool near_in = true;
            int nr[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              const float t = __fdiv_rn(c[a], (float)(A.dims[a] - 1));
              loc[a] = t;
              const float gs = __fsub_rn(__fmul_rn(t, 2.f), 1.f);
              u[a] = __fmul_rn(__fdiv_rn(__fadd_rn(gs, 1.f), 2.f), (float)(A.dims[a] - 1));
              fl[a] = floorf(u[a]);
              const float r = nearbyintf(u[a]);
              near_in = near_in && r >= 0.f && r <= (float)(A.dims[a] - 1);
              nr[a] = (int)r;
            }
            if (near_in) wvol = A.pts_weight[((size_t)nr[0] * A.dims[1] + nr[1]) * A.dims[2] + nr[2]];
#pragma unroll
            for (int k = 0; k < 8; ++k) {   // tnw, tne, tsw, tse, bnw, bne, bsw, bse: bit 0 = x (axis 2), 1 = y, 2 = z (axis 0)
              const int d0 = (k >> 2) & 1, d1 = (k >> 1) & 1, d2 = k & 1;
              const float p0 = fl[0] + (float)d0, p1 = fl[1] + (float)d1, p2 = fl[2] + (float)d2;
              // weight of a corner = product over axes of (opposite corner - u) or (u - opposite corner)
              const float w2 = d2 ? __fsub_rn(u[2], fl[2]) : __fsub_rn(fl[2] + 1.f, u[2]);
              const float w1 = d1 ? __fsub_rn(u[1], fl[1]) : __fsub_rn(fl[1] + 1.f, u[1]);
              const float w0 = d0 ? __fsub_rn(u[0], fl[0]) : __fsub_rn(fl[0] + 1.f, u[0]);
              const float wk = __fmul_rn(__fmul_rn(w2, w1), w0);
              if (p0 >= 0.f && p1 >= 0.f && p2 >= 0.f && p0 <= (float)(A.dims[0] - 1) &&
                  p1 <= (float)(A.dims[1] - 1) && p2 <= (float)(A.dims[2] - 1)) {
                const size_t o = ((size_t)(int)p0 * A.dims[1] + (int)p1) * A.dims[2] + (int)p2;
#pragma unroll
                for (int f = 0; f < 8; ++f) feat[f] = __fadd_rn(feat[f], __fmul_rn(A.feat_grid[f * plane + o], wk));
              }
            }
          }
          if (A.nf_out) {
#pragma unroll
            for (int f = 0; f < 8; ++f) A.nf_out[q * 8 + f] = feat[f];
          }
        }
      } else {
        const int64_t q = tile * 16 + (j >> 3);
        const int cb = kCornerCeilBits[j & 7];
        if (q < A.n) {
          float c[3], corner[3];
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            c[a] = A.coords[q * 3 + a];
            corner[a] = ((cb >> a) & 1) ? ceilf(c[a]) : floorf(c[a]);
            loc[a] = __fsub_rn(c[a], corner[a]);
          }
          wtri = __fmul_rn(__fmul_rn(1.f - fabsf(loc[0]), 1.f - fabsf(loc[1])), 1.f - fabsf(loc[2]));
          {  // MODE_DENSE: nearest gather == direct index, zero outside (:296-310)
            const int x = (int)corner[0], y = (int)corner[1], z = (int)corner[2];
            if (x >= 0 && y >= 0 && z >= 0 && x < A.dims[0] && y < A.dims[1] && z < A.dims[2]) {
              const size_t plane = (size_t)A.dims[0] * A.dims[1] * A.dims[2];
              const size_t o = ((size_t)x * A.dims[1] + y) * A.dims[2] + z;
#pragma unroll
              for (int f = 0; f < 8; ++f) feat[f] = A.feat_grid[f * plane + o];
              wvol = A.pts_weight[o];
            }
            // relative_xyz = rel * voxel; decode_implicit divides it by voxel again (:321,:374)
#pragma unroll
            for (int a = 0; a < 3; ++a) loc[a] = __fdiv_rn(__fmul_rn(loc[a], voxel), voxel);
          }
        }
      }
      if constexpr (PREC == 1 || PREC == 3)
        check_feature_range(feat, A.pack[SD_BA + 1], (MODE == MODE_DENSE || MODE == MODE_DENSE1) ? A.status : A.vol.n_rows);
      if constexpr (PREC == 2) stage_input_t(lds, j, loc, feat);
      else if constexpr (PREC == 1) stage_input_h<3>(lds, j, loc, feat);
      else if constexpr (PREC == 3) stage_input_h<1>(lds, j, loc, feat);
      else stage_input(hl, j, loc, feat);
      lds[L_WTRI + j] = wtri;
      lds[L_WVOL + j] = wvol;
      lds[L_DELTA + j] = dlt;
    }
    // ---------------- MLP -----------------------------------------------------------------
    BNV_PH(0);
    __syncthreads();
    BNV_PH(18);
    if constexpr (PREC == 2) sdf_mlp_tile_t(lds, A.pack);
    else if constexpr (PREC == 1) sdf_mlp_tile_h<3>(lds, A.pack);
    else if constexpr (PREC == 3) sdf_mlp_tile_h<1>(lds, A.pack);
    else sdf_mlp_tile(lds, A.pack);
    // ---------------- back end ------------------------------------------------------------
    if constexpr (MODE == MODE_LATTICE) {
      if (threadIdx.x < DM) {
        const int64_t e = tile * DM + threadIdx.x;
        if (e < n_evals) {
          int row, l;
          if (A.entries) {
            const int ent = A.entries[e];
            row = ent >> 5;
            l = ent & 31;
            if (A.need_mask) A.need_mask[row] = 0u;  // leave the per-row masks clean for the next call
          } else {
            const int64_t ci = e / 27;
            l = (int)(e - ci * 27);
            row = A.list[ci];
          }
          float av = __fmul_rn(lds[L_ALPHA + threadIdx.x], voxel);
          if constexpr (PREC == 2) av = (float)(_Float16)av;  // half tensor * python float stays half (sparse_volume.py:813)
          A.table[(size_t)row * 27 + l] = av;
        }
      }
    } else if constexpr (MODE == MODE_DENSE1) {
      if (threadIdx.x < DM) {
        const int64_t q = tile * DM + threadIdx.x;
        if (q < A.n) {
          const float wv = lds[L_WVOL + threadIdx.x];
          const bool ok = wv >= (float)A.grid.min_pts_in_grid;
          float a = lds[L_ALPHA + threadIdx.x];
          // nearest: decode_implicit(normalize=True) scales by voxel; global: normalize=False, the raw prediction
          if (A.variant == 0) {
            a = __fmul_rn(a, voxel);
            if constexpr (PREC == 2) a = (float)(_Float16)a;
          }
          A.out[q] = ok ? a : voxel;   // forward_with_mask zero + valid_mask (:340-343) / valid_mask (:365-366)
        }
      }
    } else {
      if (threadIdx.x < 16) {
        const int64_t q = tile * 16 + threadIdx.x;
        if (q < A.n) {
          const int b = threadIdx.x * 8;
          float norm = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) norm = __fadd_rn(norm, lds[L_WTRI + b + k]);
          float acc = 0.f, dacc = 0.f, wmin = 3.4e38f, wsum = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float wk = __fdiv_rn(lds[L_WTRI + b + k], norm);
            const float wv = lds[L_WVOL + b + k];
            float a = __fmul_rn(lds[L_ALPHA + b + k], voxel);
            if constexpr (PREC == 2) a = (float)(_Float16)a;
            if constexpr (MODE == MODE_DENSE) {
              const bool ok = wv >= (float)A.grid.min_pts_in_grid;  // forward_with_mask (modules.py:774-783)
              a = ok ? a : 0.f;
              wsum = __fadd_rn(wsum, ok ? wv : 0.f);
            }
            acc = __fadd_rn(acc, __fmul_rn(a, wk));
            dacc = __fadd_rn(dacc, __fmul_rn(lds[L_DELTA + b + k], wk));
            wmin = fminf(wmin, wv);
          }
          (void)dacc;
          (void)wmin;
          A.out[q] = (wsum > 0.f) ? acc : voxel;  // MODE_DENSE: any corner valid (:328-329)
        }
      }
    }
    BNV_PH(17);
    __syncthreads();
    BNV_PH(19);
  }
#ifdef BNV_PHASE_PROF
  __syncthreads();
  if (threadIdx.x < 256 && (threadIdx.x & 31) != 31)
    atomicAdd(&g_phase_cycles[threadIdx.x], ((unsigned long long*)(lds + L_PROF))[threadIdx.x]);
  if (threadIdx.x == 0) atomicAdd(&g_phase_cycles[31], 1ull);
#endif
}

// ---------------------------------------------------------------------------------------------------
// Arbitrary query points (SparseVolume.decode_pts, sparse_volume.py:768-833) with LIVE-QUERY COMPACTION.
// A query whose 8 corners are not all observed decodes to the constant voxel_size (:809, :818) without ever
// reading its MLP outputs; the ray samples of the global optimiser are ~90 % such free-space points, but
// spread so that nearly every run of 16 consecutive queries contains a live one.  The workgroup therefore
// first CLASSIFIES a chunk of 128 queries (1,024 corner look-ups by all 512 threads; masked queries are
// finished right there), compacts the live ones into an LDS list, and runs the MLP on tiles of 16 LIVE
// queries.  Shared by the forward kernel and the two backward kernels.
// ---------------------------------------------------------------------------------------------------
constexpr int PC_Q = 128;                         // queries per chunk
constexpr int C_ROW = L_TOTAL;                    // [1024] int   row of every (query, corner) or -1
constexpr int C_WN = C_ROW + PC_Q * 8;            // [1024] float trilinear weight / sum over the 8 corners
constexpr int C_DLT = C_WN + PC_Q * 8;            // [1024] float sdf_delta sample of the corner
constexpr int C_LIST = C_DLT + PC_Q * 8;          // [128]  int   chunk-local indices of the live queries
constexpr int C_CNT = C_LIST + PC_Q;              // [4]    int   number of live queries
constexpr int C_TOTAL = C_CNT + 4;                // 38,532 floats = 154,128 B

// corner k of query point c (voxel units): corner coordinates, local offset, trilinear weight
__device__ __forceinline__ float pts_corner(const DecodeArgs& A, int64_t q, int k, float (&corner)[3], float (&loc)[3]) {
  const int cb = kCornerCeilBits[k];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float c = A.coords[q * 3 + a];
    if (!A.is_coords) c = __fdiv_rn(__fsub_rn(c, A.grid.bound_min[a]), A.grid.voxel_size);  // (:793)
    corner[a] = ((cb >> a) & 1) ? ceilf(c) : floorf(c);
    loc[a] = __fsub_rn(c, corner[a]);
  }
  return __fmul_rn(__fmul_rn(1.f - fabsf(loc[0]), 1.f - fabsf(loc[1])), 1.f - fabsf(loc[2]));
}

// Classifies chunk `chunk`; returns the number of live queries (uniform).  Masked queries are finished here:
// masked(q, value) gets their final value (forward: written to A.out; fused optimiser step: their loss term); live
// ones are listed in C_LIST in ascending order.
template <class MaskedFn>
__device__ __forceinline__ int pts_classify_chunk_fn(const DecodeArgs& A, int64_t chunk, float* __restrict__ lds,
                                                     MaskedFn masked) {
  int* c_row = (int*)(lds + C_ROW);
  int* c_list = (int*)(lds + C_LIST);
  int* c_cnt = (int*)(lds + C_CNT);
  const float voxel = A.grid.voxel_size;
  if (threadIdx.x == 0) *c_cnt = 0;
  __syncthreads();
  unsigned live_bits = 0;  // lanes with k == 0: bit i set when this thread's i-th query is live
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int ce = it * 512 + threadIdx.x;  // (query, corner) index within the chunk
    const int64_t q = chunk * PC_Q + (ce >> 3);
    const int k = ce & 7;
    float wtri = 0.f, wvol = 0.f, dlt = 0.f;
    int row = -1;
    if (q < A.n) {
      float corner[3], loc[3];
      wtri = pts_corner(A, q, k, corner, loc);
      uint64_t key;
      if (pack_key((int64_t)corner[0], (int64_t)corner[1], (int64_t)corner[2], &key))
        row = volume_find(A.vol.slot_keys, A.vol.slot_rows, (uint32_t)(A.vol.n_slots - 1), key);
      if (row >= A.row_limit) row = -1;
      if (row >= 0) {
        wvol = A.weights[row];
        if (A.split_mask) {   // count_optim of the splits up to and including this query's, one exact +1 each
          uint32_t m = A.split_mask[row] & ((2u << (uint32_t)(q / A.split_samples)) - 1u);
          while (m) {
            wvol = __fadd_rn(wvol, 1.0f);
            m &= m - 1u;
          }
        }
      }
      if (A.delta.data) dlt = sample_delta(A.delta, A.grid, corner);
    }
    // the 8 corners of a query sit in 8 consecutive lanes: sums in corner order, like the reference's dim-1 sum
    float norm = 0.f, wmin = 3.4e38f;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      norm = __fadd_rn(norm, __shfl(wtri, (threadIdx.x & 56) + kk));
      wmin = fminf(wmin, __shfl(wvol, (threadIdx.x & 56) + kk));
    }
    const float wn = __fdiv_rn(wtri, norm);
    float dacc = 0.f;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) dacc = __fadd_rn(dacc, __shfl(__fmul_rn(dlt, wn), (threadIdx.x & 56) + kk));
    c_row[ce] = row;
    lds[C_WN + ce] = wn;
    lds[C_DLT + ce] = dlt;
    const bool live = q < A.n && wmin >= (float)A.grid.min_pts_in_grid;
    if (k == 0 && q < A.n) {
      if (live) {
        live_bits |= 1u << it;
      } else {
        float o = voxel;
        if (A.delta.data) o = __fadd_rn(o, dacc);
        masked(q, o);
      }
    }
  }
  // ordered compaction of the live queries (chunk-local index = ce >> 3): ballot per wave, wave offsets via LDS
  __shared__ int wave_cnt[2][8];
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  unsigned long long b0 = __ballot(live_bits & 1u), b1 = __ballot(live_bits & 2u);
  if (ln == 0) {
    wave_cnt[0][wv] = __popcll(b0);
    wave_cnt[1][wv] = __popcll(b1);
  }
  __syncthreads();
  int base0 = 0, base1 = 0, tot0 = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i < wv) {
      base0 += wave_cnt[0][i];
      base1 += wave_cnt[1][i];
    }
    tot0 += wave_cnt[0][i];
  }
  if (live_bits & 1u) c_list[base0 + __popcll(b0 & ((1ull << ln) - 1ull))] = threadIdx.x >> 3;
  if (live_bits & 2u) c_list[tot0 + base1 + __popcll(b1 & ((1ull << ln) - 1ull))] = 64 + (threadIdx.x >> 3);
  if (threadIdx.x == 511) {
    int t1 = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t1 += wave_cnt[1][i];
    *c_cnt = tot0 + t1;
  }
  __syncthreads();
  return *c_cnt;
}

template <bool WRITE_MASKED>
__device__ __forceinline__ int pts_classify_chunk(const DecodeArgs& A, int64_t chunk, float* __restrict__ lds) {
  return pts_classify_chunk_fn(A, chunk, lds, [&](int64_t q, float o) {
    if (WRITE_MASKED) A.out[q] = o;
  });
}

// front end of one tile of 16 live queries: thread e < 128 = (live query e >> 3, corner e & 7)
template <int PREC>
__device__ __forceinline__ void pts_stage_tile(const DecodeArgs& A, int64_t chunk, int tile, int n_live,
                                               float* __restrict__ lds, int* __restrict__ row_out) {
  const int e = threadIdx.x;
  const int* c_row = (const int*)(lds + C_ROW);
  const int* c_list = (const int*)(lds + C_LIST);
  float loc[3] = {0.f, 0.f, 0.f};
  float feat[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float wn = 0.f, dlt = 0.f;
  int row = -1;
  const int li = tile * 16 + (e >> 3);
  if (li < n_live) {
    const int ql = c_list[li];
    const int ce = ql * 8 + (e & 7);
    float corner[3];
    pts_corner(A, chunk * PC_Q + ql, e & 7, corner, loc);
    row = c_row[ce];
    wn = lds[C_WN + ce];
    dlt = lds[C_DLT + ce];
    if (row >= 0) {
      const f32x4 f0 = *(const f32x4*)&A.features[(size_t)row * 8];
      const f32x4 f1 = *(const f32x4*)&A.features[(size_t)row * 8 + 4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        feat[f] = f0[f];
        feat[4 + f] = f1[f];
      }
    }
  }
  if constexpr (PREC == 1 || PREC == 3) check_feature_range(feat, A.pack[SD_BA + 1], A.vol.n_rows);
  if constexpr (PREC == 2) stage_input_t(lds, e, loc, feat);
  else if constexpr (PREC == 1) stage_input_h<3>(lds, e, loc, feat);
  else if constexpr (PREC == 3) stage_input_h<1>(lds, e, loc, feat);
  else stage_input(lds + L_HL, e, loc, feat);
  lds[L_WTRI + e] = wn;
  lds[L_DELTA + e] = dlt;
  if (row_out) row_out[e] = row;
}

template <int PREC>
__global__ __launch_bounds__(512, 2) void k_decode_pts(DecodeArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const float voxel = A.grid.voxel_size;
  const int64_t n_chunks = (A.n + PC_Q - 1) / PC_Q;
  for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const int n_live = pts_classify_chunk<true>(A, chunk, lds);
    for (int tile = 0; tile * 16 < n_live; ++tile) {
      if (threadIdx.x < DM) pts_stage_tile<PREC>(A, chunk, tile, n_live, lds, nullptr);
      __syncthreads();
      if constexpr (PREC == 2) sdf_mlp_tile_t(lds, A.pack);
      else if constexpr (PREC == 1) sdf_mlp_tile_h<3>(lds, A.pack);
      else if constexpr (PREC == 3) sdf_mlp_tile_h<1>(lds, A.pack);
      else sdf_mlp_tile(lds, A.pack);
      if (threadIdx.x < 16 && tile * 16 + threadIdx.x < n_live) {
        const int b = threadIdx.x * 8;
        float acc = 0.f, dacc = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float wk = lds[L_WTRI + b + k];
          float a = __fmul_rn(lds[L_ALPHA + b + k], voxel);
          if constexpr (PREC == 2) a = (float)(_Float16)a;  // half tensor * python float stays half (:813)
          acc = __fadd_rn(acc, __fmul_rn(a, wk));
          dacc = __fadd_rn(dacc, __fmul_rn(lds[L_DELTA + b + k], wk));
        }
        if (A.delta.data) acc = __fadd_rn(acc, dacc);
        A.out[chunk * PC_Q + ((const int*)(lds + C_LIST))[tile * 16 + threadIdx.x]] = acc;
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// k_decode_pts_bwd: d(loss)/d(volume features) of k_decode<PTS> -- what the global optimiser needs
// (run_e2e.py:111-162 makes volume.features an nn.Parameter and back-propagates the ray loss of
// render_utils.py:461-560 through SparseVolume.decode_pts, sparse_volume.py:768-833).
// Only the features carry gradient (the decoder is frozen, the query points are data).
//
// Per 128-evaluation tile: the forward MLP is recomputed in split-f16 arithmetic keeping ONE BIT per
// pre-activation (z > 0) in registers -- the lane that owns z_l[feature][evaluation] in the forward D
// layout owns the same position of W_{l+1}^T delta_{l+1} in the backward pass, so the ReLU masks never
// leave the lane.  The backward pass is the same transposed-chaining MLP run on the transposed weight
// packs: delta_3 = wa * [z3 > 0]; delta_l = (W_{l+1}^T delta_{l+1}) * [z_l > 0]; g_in = W_0^T delta_0.
// It propagates d(alpha)/d(input) with a unit seed per evaluation, so its operands stay O(1) whatever the
// scale of the loss (an f16 split of 1e-7-sized loss gradients would underflow); the evaluation's
// incoming gradient go = grad_sdf[q] * voxel * w_k / sum(w) * [mask_q] multiplies the 8 feature rows of
// g_in in fp32 at the very end, followed by float atomics into grad_features[row].  Tiles whose 16
// queries are all masked (free space: most ray samples) skip the MLP altogether.
// ---------------------------------------------------------------------------------------------------
// mlp_layer_h with the weight fragments fetched by buffer loads (used where several layers' worth of
// hoisted flat addresses would not fit the register file)
template <int NKS, bool BIAS>
__device__ __forceinline__ void mlp_layer_hb(const _Float16* __restrict__ wp, const float* __restrict__ bias,
                                             const float* __restrict__ lds, f32x16 (&acc)[4], int w, int lane,
                                             int j, int h) {
  f32x16 b0;
  if constexpr (BIAS) {
    b0 = frag256(bias, w, h);
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) b0[r] = 0.f;
  }
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) acc[pt] = b0;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, 8 * NKS * 2 * 64 * 8 * 2, 0x00020000);
  const int voff = lane * 16;
  const int sbase = w * NKS * 2 * 1024;
  const float* hh = lds + L_HL + (h * DM + j) * 4;
  const float* hl = lds + L_HLO + (h * DM + j) * 4;
  half8 ah[3], al[3], bh[2][4], bl[2][4];
#define BNV_LOAD_A(ks)                                                  \
  {                                                                     \
    ah[(ks) % 3] = load_frag(rs, voff, sbase + ((ks) * 2) * 1024);      \
    al[(ks) % 3] = load_frag(rs, voff, sbase + ((ks) * 2 + 1) * 1024);  \
  }
#define BNV_LOAD_B(ks)                                                                    \
  {                                                                                       \
    _Pragma("unroll") for (int pt = 0; pt < 4; ++pt) {                                    \
      bh[(ks) & 1][pt] = *(const half8*)(hh + ((ks) * 2 * DM + pt * 32) * 4);             \
      bl[(ks) & 1][pt] = *(const half8*)(hl + ((ks) * 2 * DM + pt * 32) * 4);             \
    }                                                                                     \
  }
  BNV_LOAD_A(0);
  if (NKS > 1) BNV_LOAD_A(1);
  BNV_LOAD_B(0);
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    if (ks + 2 < NKS) BNV_LOAD_A(ks + 2);
    if (ks + 1 < NKS) BNV_LOAD_B(ks + 1);
    __builtin_amdgcn_sched_barrier(0);
    const half8 a_hi = ah[ks % 3], a_lo = al[ks % 3];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt)
      acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, bh[ks & 1][pt], acc[pt], 0, 0, 0);
#pragma unroll
    for (int pt = 0; pt < 4; ++pt)
      acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, bl[ks & 1][pt], acc[pt], 0, 0, 0);
#pragma unroll
    for (int pt = 0; pt < 4; ++pt)
      acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, bh[ks & 1][pt], acc[pt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
#undef BNV_LOAD_A
#undef BNV_LOAD_B
}

constexpr int SB_W3T = 0;                          // [8 w][16 ks][2 hi/lo][64 lane][8]: W3^T
constexpr int SB_W2T = SB_W3T + 8 * 16 * 2 * 64 * 8;
constexpr int SB_W1T = SB_W2T + 8 * 16 * 2 * 64 * 8;
constexpr int SB_W0T = SB_W1T + 8 * 16 * 2 * 64 * 8;  // [16 ks][2][64][8]: W0^T, 17 rows padded to 32
constexpr int SB_TOTAL = SB_W0T + 16 * 2 * 64 * 8;    // 409,600 halves
constexpr int SB_PACK_FLOATS = SB_TOTAL / 2;

struct DecodeBwdArgs {
  DecodeArgs d;
  const float* bwd_pack;
  const float* grad_out;
  float* grad_features;
};

// bit (pt * 16 + r) = [acc[pt][r] > 0].  Built as a shift-or chain: with independent (cmp << k) terms the
// compiler keeps all 64 selected constants live and spills them.
__device__ __forceinline__ uint64_t positive_bits(const f32x16 (&acc)[4]) {
  uint32_t m[2] = {0u, 0u};
#pragma unroll
  for (int pt = 3; pt >= 0; --pt) {
#pragma unroll
    for (int r = 15; r >= 0; --r) m[pt >> 1] = (m[pt >> 1] << 1) | (uint32_t)(acc[pt][r] > 0.f);
  }
  return ((uint64_t)m[1] << 32) | m[0];
}

// acc <- acc where the bit is set, else 0, then split + store as the next layer's B operand
__device__ __forceinline__ void store_masked_h(float* __restrict__ lds, const f32x16 (&acc)[4], uint64_t m, int w,
                                               int j, int h) {
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) {
#pragma unroll
    for (int ksl = 0; ksl < 2; ++ksl) {
      half8 hi, lo;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = ((m >> (pt * 16 + 8 * ksl + e)) & 1) ? acc[pt][8 * ksl + e] : 0.f;
        const _Float16 t = (_Float16)x;
        hi[e] = t;
        lo[e] = (_Float16)(x - (float)t);
      }
      const int o = (((2 * w + ksl) * 2 + h) * DM + pt * 32 + j) * 4;
      *(half8*)&lds[L_HL + o] = hi;
      *(half8*)&lds[L_HLO + o] = lo;
    }
  }
}

__global__ __launch_bounds__(512, 2) void k_decode_pts_bwd(DecodeBwdArgs B) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const DecodeArgs& A = B.d;
  const float voxel = A.grid.voxel_size;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, h = lane >> 5;
  int* l_row = (int*)(lds + L_WVOL);
  const int64_t n_chunks = (A.n + PC_Q - 1) / PC_Q;
  for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
   const int n_live = pts_classify_chunk<false>(A, chunk, lds);
   for (int tile = 0; tile * 16 < n_live; ++tile) {
    // front end: 16 LIVE queries of the chunk; masked queries carry no gradient and were dropped above
    if (threadIdx.x < DM) {
      pts_stage_tile<1>(A, chunk, tile, n_live, lds, l_row);
      // incoming gradient of every evaluation: d out_q / d alpha_k = voxel * w_k / sum(w)
      const int li = tile * 16 + (threadIdx.x >> 3);
      float go = 0.f;
      if (li < n_live)
        go = B.grad_out[chunk * PC_Q + ((const int*)(lds + C_LIST))[li]] * voxel * lds[L_WTRI + threadIdx.x];
      lds[L_ALPHA + threadIdx.x] = go;
    }
    __syncthreads();
    // launder the weight pointers once per tile: otherwise the bias / fc_alpha fragments (80 VGPRs) are
    // hoisted out of the tile loop as loop invariants and the MLP spills
    const float* pack = A.pack;
    const float* bpack = B.bwd_pack;
    asm volatile("" : "+s"(pack), "+s"(bpack));
    const _Float16* ph = (const _Float16*)(pack + SD_TOTAL);
    const _Float16* pb = (const _Float16*)bpack;
    // ---------------- forward, keeping the sign bits of the pre-activations ---------------------------
    f32x16 acc[4];
    mlp_layer_hb<2, true>(ph + SH_W0, pack + SD_B0, lds, acc, w, lane, j, h);
    const uint64_t m0 = positive_bits(acc);
    __syncthreads();
    store_relu_h(lds, acc, w, j, h);
    __syncthreads();
    mlp_layer_hb<16, true>(ph + SH_W1, pack + SD_B0 + 256, lds, acc, w, lane, j, h);
    const uint64_t m1 = positive_bits(acc);
    __syncthreads();
    store_relu_h(lds, acc, w, j, h);
    __syncthreads();
    mlp_layer_hb<16, true>(ph + SH_W2, pack + SD_B0 + 512, lds, acc, w, lane, j, h);
    const uint64_t m2 = positive_bits(acc);
    __syncthreads();
    store_relu_h(lds, acc, w, j, h);
    __syncthreads();
    mlp_layer_hb<16, true>(ph + SH_W3, pack + SD_B0 + 768, lds, acc, w, lane, j, h);
    // ---------------- backward with a unit seed: delta_3 = wa * [z3 > 0] ------------------------------
    {
      const uint64_t m3 = positive_bits(acc);
      const f32x16 wa = frag256(pack + SD_WA, w, h);
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) acc[pt] = wa;
      __syncthreads();
      store_masked_h(lds, acc, m3, w, j, h);
    }
    __syncthreads();
    mlp_layer_hb<16, false>(pb + SB_W3T, nullptr, lds, acc, w, lane, j, h);
    __syncthreads();
    store_masked_h(lds, acc, m2, w, j, h);
    __syncthreads();
    mlp_layer_hb<16, false>(pb + SB_W2T, nullptr, lds, acc, w, lane, j, h);
    __syncthreads();
    store_masked_h(lds, acc, m1, w, j, h);
    __syncthreads();
    mlp_layer_hb<16, false>(pb + SB_W1T, nullptr, lds, acc, w, lane, j, h);
    __syncthreads();
    store_masked_h(lds, acc, m0, w, j, h);
    __syncthreads();
    // g_in = W0^T delta_0: 32 (17 used) x 128; wave w < 4 takes column block w
    if (w < 4) {
      f32x16 g;
#pragma unroll
      for (int r = 0; r < 16; ++r) g[r] = 0.f;
      const _Float16* wl = pb + SB_W0T + lane * 8;
      const float* hh = lds + L_HL + (h * DM + w * 32 + j) * 4;
      const float* hl = lds + L_HLO + (h * DM + w * 32 + j) * 4;
#pragma unroll 4
      for (int ks = 0; ks < 16; ++ks) {
        const half8 a_hi = *(const half8*)(wl + (ks * 2) * 64 * 8);
        const half8 a_lo = *(const half8*)(wl + (ks * 2 + 1) * 64 * 8);
        const half8 b_hi = *(const half8*)(hh + ks * 2 * DM * 4);
        const half8 b_lo = *(const half8*)(hl + ks * 2 * DM * 4);
        g = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, g, 0, 0, 0);
        g = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, g, 0, 0, 0);
        g = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, g, 0, 0, 0);
      }
      // D row (r&3) + 8 (r>>2) + 4 h is network input 9 + f for feature f: h = 0 holds f = 0, 1, 2 (r = 5, 6, 7)
      // and f = 7 (r = 8); h = 1 holds f = 3..6 (r = 4..7)
      const int col = w * 32 + j;
      const float s = lds[L_ALPHA + col];
      const int row = l_row[col];
      if (s != 0.f && row >= 0) {
        float* gf = B.grad_features + (size_t)row * 8;
        if (h == 0) {
          unsafeAtomicAdd(gf + 0, g[5] * s);
          unsafeAtomicAdd(gf + 1, g[6] * s);
          unsafeAtomicAdd(gf + 2, g[7] * s);
          unsafeAtomicAdd(gf + 7, g[8] * s);
        } else {
          unsafeAtomicAdd(gf + 3, g[4] * s);
          unsafeAtomicAdd(gf + 4, g[5] * s);
          unsafeAtomicAdd(gf + 5, g[6] * s);
          unsafeAtomicAdd(gf + 6, g[7] * s);
        }
      }
    }
    __syncthreads();
   }
  }
}

// ---------------------------------------------------------------------------------------------------
// k_optim_step (round 6): ONE launch for what an optimiser step of the reference spends five forward and five
// backward decode_pts calls on (run_e2e.py:127-153: 5,000 rays in splits of 1,000; render_utils.py:461-590).
// The L1 ray loss is elementwise -- d loss / d pred_q = sign(pred_q - target_q) * weight_q / n_valid(split) -- so the
// gradient a query sends back is known as soon as ITS forward value is: the forward (which k_decode_pts_bwd
// recomputes anyway for the ReLU masks) yields pred, the loss term and the seed of the backward in the same tile,
// and the separate forward kernel, the loss kernel and the round trip of pred / grad through memory all go.  All
// splits of a step ride in one launch: the weight a mask decision sees is reconstructed per split from
// split_mask (DecodeArgs), so every decision is the one the split-by-split sequence takes; chunks of 128 queries
// are handed out dynamically (a ray split has < 1 tile of live queries per workgroup: five launches of each kernel
// left 3/4 of every launch's time to launch latency and one-tile rounds).  Arithmetic: the split-f16 forward /
// backward of k_decode_pts_bwd (fp32 checkpoints; SDF within 1e-8 of the exact-fp32 forward, gradients to 1e-6).
// ---------------------------------------------------------------------------------------------------
struct OptimArgs {
  DecodeBwdArgs b;          // b.d: the queries (coords = the step's samples, split_mask / split_samples); b.grad_features
  const float* target;      // [n]  L1 target of every sample (bnv_ray_samples)
  const float* wgt;         // [n]  valid x ray mask
  const float* n_valid;     // [n_splits]  sum of the split's ray masks + 1e-4 (render_utils.py:553)
  float* loss;              // [0] += sum over the splits of their losses; [1]: int32 chunk counter (zeroed by the caller)
  float* pred;              // optional [n]: the decoded SDF (tests, diagnostics)
};

__global__ __launch_bounds__(512) void k_optim_step(OptimArgs O) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const DecodeBwdArgs& B = O.b;
  const DecodeArgs& A = B.d;
  const float voxel = A.grid.voxel_size;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, h = lane >> 5;
  int* l_row = (int*)(lds + L_WVOL);
  __shared__ int s_chunk;
  __shared__ float s_red[8];
  const int64_t n_chunks = (A.n + PC_Q - 1) / PC_Q;
  float loss_acc = 0.f;
  auto loss_term = [&](int64_t q, float pred) -> float {      // -> d loss / d pred_q
    const float inv = 1.f / O.n_valid[A.split_samples > 0 ? q / A.split_samples : 0];
    const float wq = O.wgt[q] * inv;
    const float d = pred - O.target[q];
    loss_acc += fabsf(d) * wq;
    if (O.pred) O.pred[q] = pred;
    return d > 0.f ? wq : (d < 0.f ? -wq : 0.f);               // d|x|/dx with torch's sign(0) = 0
  };
  for (;;) {
    __syncthreads();                                           // (s_chunk of the round before has been read)
    if (threadIdx.x == 0) s_chunk = atomicAdd((int*)(O.loss + 1), 1);
    __syncthreads();
    const int64_t chunk = s_chunk;
    if (chunk >= n_chunks) break;
    const int n_live = pts_classify_chunk_fn(A, chunk, lds, [&](int64_t q, float o) { (void)loss_term(q, o); });
#if defined(BNV_OPTIM_PHASES) && BNV_OPTIM_PHASES == 1      // development probe (tools/optim_phases.sh): classification only
    continue;
#endif
    for (int tile = 0; tile * 16 < n_live; ++tile) {
      if (threadIdx.x < DM) pts_stage_tile<1>(A, chunk, tile, n_live, lds, l_row);
      __syncthreads();
      const float* pack = A.pack;
      const float* bpack = B.bwd_pack;
      asm volatile("" : "+s"(pack), "+s"(bpack));
      const _Float16* ph = (const _Float16*)(pack + SD_TOTAL);
      const _Float16* pb = (const _Float16*)bpack;
      // ---------------- forward, keeping the sign bits of the pre-activations ---------------------------
      // (the 256-wide layers are LOOPS, not three copies of the tile code each way: unrolled, the kernel's body is ~90 KB of
      // instructions against a 64 KB instruction cache, and every tile streamed all of it through the cache)
      constexpr int LAYER_HALVES = 8 * 16 * 2 * 64 * 8;
      static_assert(SH_W2 - SH_W1 == LAYER_HALVES && SH_W3 - SH_W2 == LAYER_HALVES, "forward layers are equally spaced");
      static_assert(SB_W2T - SB_W3T == LAYER_HALVES && SB_W1T - SB_W2T == LAYER_HALVES, "backward layers too");
      f32x16 acc[4];
      mlp_layer_hb<2, true>(ph + SH_W0, pack + SD_B0, lds, acc, w, lane, j, h);
      const uint64_t m0 = positive_bits(acc);
      uint64_t m1 = 0, m2 = 0;
      __syncthreads();
      store_relu_h(lds, acc, w, j, h);
      __syncthreads();
#pragma unroll 1
      for (int l = 1;; ++l) {
        mlp_layer_hb<16, true>(ph + SH_W1 + (l - 1) * LAYER_HALVES, pack + SD_B0 + 256 * l, lds, acc, w, lane, j, h);
        if (l == 3) break;
        const uint64_t m = positive_bits(acc);
        if (l == 1) m1 = m; else m2 = m;
        __syncthreads();
        store_relu_h(lds, acc, w, j, h);
        __syncthreads();
      }
      {
        // fc_alpha (the forward's last layer) and the backward's seed delta_3 = wa * [z3 > 0] from the same fragment
        const uint64_t m3 = positive_bits(acc);
        const f32x16 wa = frag256(pack + SD_WA, w, h);
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
          float sp = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) sp = fmaf(wa[r], relu_bits(acc[pt][r]), sp);
          lds[L_PART + (w * 2 + h) * DM + pt * 32 + j] = sp;
          acc[pt] = wa;
        }
        __syncthreads();                      // layer 3 has read its operands; the partial sums are in place
        store_masked_h(lds, acc, m3, w, j, h);
      }
      if (threadIdx.x < DM) {
        // evaluation e = (live query e >> 3, corner e & 7): alpha -> the query's SDF (sums in corner order, like the
        // forward kernel) -> its loss term -> the gradient every one of its 8 evaluations starts from
        const int e = threadIdx.x;
        float al = pack[SD_BA];
#pragma unroll
        for (int p = 0; p < 16; ++p) al += lds[L_PART + p * DM + e];
        const float wk = lds[L_WTRI + e];
        const float ak = __fmul_rn(__fmul_rn(al, voxel), wk);
        const float dk = __fmul_rn(lds[L_DELTA + e], wk);
        float sum = 0.f, dsum = 0.f;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          sum = __fadd_rn(sum, __shfl(ak, (e & 56) + kk));
          dsum = __fadd_rn(dsum, __shfl(dk, (e & 56) + kk));
        }
        if (A.delta.data) sum = __fadd_rn(sum, dsum);
        const int li = tile * 16 + (e >> 3);
        float go = 0.f;
        if (li < n_live) {
          const int64_t q = chunk * PC_Q + ((const int*)(lds + C_LIST))[li];
          float g = 0.f;
          if ((e & 7) == 0) g = loss_term(q, sum);
          g = __shfl(g, e & 56);
          go = g * voxel * wk;
        }
        lds[L_ALPHA + e] = go;
      }
      __syncthreads();
#if defined(BNV_OPTIM_PHASES) && BNV_OPTIM_PHASES == 2      // development probe: classification + forward + loss only
      continue;
#endif
      // ---------------- backward with a unit seed (as k_decode_pts_bwd) ---------------------------------
#pragma unroll 1
      for (int l = 0; l < 3; ++l) {
        mlp_layer_hb<16, false>(pb + SB_W3T + l * LAYER_HALVES, nullptr, lds, acc, w, lane, j, h);
        const uint64_t m = l == 0 ? m2 : (l == 1 ? m1 : m0);
        __syncthreads();
        store_masked_h(lds, acc, m, w, j, h);
        __syncthreads();
      }
#if defined(BNV_OPTIM_PHASES) && BNV_OPTIM_PHASES == 3      // development probe: ... + the three 256-wide backward layers
      continue;
#endif
      if (w < 4) {
        f32x16 g;
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = 0.f;
        const _Float16* wl = pb + SB_W0T + lane * 8;
        const float* hh = lds + L_HL + (h * DM + w * 32 + j) * 4;
        const float* hl = lds + L_HLO + (h * DM + w * 32 + j) * 4;
#pragma unroll 4
        for (int ks = 0; ks < 16; ++ks) {
          const half8 a_hi = *(const half8*)(wl + (ks * 2) * 64 * 8);
          const half8 a_lo = *(const half8*)(wl + (ks * 2 + 1) * 64 * 8);
          const half8 b_hi = *(const half8*)(hh + ks * 2 * DM * 4);
          const half8 b_lo = *(const half8*)(hl + ks * 2 * DM * 4);
          g = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, g, 0, 0, 0);
          g = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, g, 0, 0, 0);
          g = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, g, 0, 0, 0);
        }
        const int col = w * 32 + j;
        const float sg = lds[L_ALPHA + col];
        const int row = l_row[col];
#if defined(BNV_OPTIM_PHASES) && BNV_OPTIM_PHASES == 4      // development probe: everything but the gradient's atomics
        if (sg == 12345.f && row >= 0) {
#else
        if (sg != 0.f && row >= 0) {
#endif
          float* gf = B.grad_features + (size_t)row * 8;
          if (h == 0) {
            unsafeAtomicAdd(gf + 0, g[5] * sg);
            unsafeAtomicAdd(gf + 1, g[6] * sg);
            unsafeAtomicAdd(gf + 2, g[7] * sg);
            unsafeAtomicAdd(gf + 7, g[8] * sg);
          } else {
            unsafeAtomicAdd(gf + 3, g[4] * sg);
            unsafeAtomicAdd(gf + 4, g[5] * sg);
            unsafeAtomicAdd(gf + 5, g[6] * sg);
            unsafeAtomicAdd(gf + 6, g[7] * sg);
          }
        }
      }
      __syncthreads();
    }
  }
  // the workgroup's share of the loss: one atomic
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) loss_acc += __shfl_xor(loss_acc, o);
  if (lane == 0) s_red[w] = loss_acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += s_red[i];
    if (t != 0.f) unsafeAtomicAdd(O.loss, t);
  }
}

// ---------------------------------------------------------------------------------------------------
// k_decode_pts_bwd_t: the same backward for the tiny-cuda-nn decoder (MLP mode 2; the reference's default
// checkpoint).  32 | 64 | 64 | 64 | 16, no bias, f16 operands, fp32 accumulate: one wave carries 32
// evaluations forward and backward in registers (waves 0..3 of the workgroup; no barriers inside).
// PARITY UNPINNED like the forward (tcnn's CUDA arithmetic cannot run here).  tcnn back-propagates in fp16
// with a loss scale; here the Jacobian d(alpha)/d(input) is propagated with a unit seed (f16 operands O(1),
// fp32 accumulation) and multiplied by the incoming gradient in fp32, which cannot underflow.
// Pack (halves): W2^T [2 mb][4 g][64][8] | W1^T [2 mb][4 g][64][8] | W0^T [4 g][64][8] | 128 halves holding
// row 0 of the output layer as 64 floats.
// ---------------------------------------------------------------------------------------------------
constexpr int TB_W2T = 0;
constexpr int TB_W1T = TB_W2T + 2 * 4 * 64 * 8;
constexpr int TB_W0T = TB_W1T + 2 * 4 * 64 * 8;
constexpr int TB_W3R = TB_W0T + 4 * 64 * 8;   // 64 floats
constexpr int TB_TOTAL = TB_W3R + 128;        // 10,368 halves = 5,184 floats

__device__ __forceinline__ uint32_t positive_bits32(const f32x16 (&a)[2]) {
  uint32_t m = 0u;
#pragma unroll
  for (int mb = 1; mb >= 0; --mb) {
#pragma unroll
    for (int r = 15; r >= 0; --r) m = (m << 1) | (uint32_t)(a[mb][r] > 0.f);
  }
  return m;
}

__global__ __launch_bounds__(512, 2) void k_decode_pts_bwd_t(DecodeBwdArgs B) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const DecodeArgs& A = B.d;
  const float voxel = A.grid.voxel_size;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, h = lane >> 5;
  int* l_row = (int*)(lds + L_WVOL);
  const int64_t n_chunks = (A.n + PC_Q - 1) / PC_Q;
  for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
   const int n_live = pts_classify_chunk<false>(A, chunk, lds);
   for (int tile = 0; tile * 16 < n_live; ++tile) {
    // front end: 16 LIVE queries of the chunk; masked queries carry no gradient and were dropped above
    if (threadIdx.x < DM) {
      pts_stage_tile<2>(A, chunk, tile, n_live, lds, l_row);
      // incoming gradient of every evaluation: d out_q / d alpha_k = voxel * w_k / sum(w)
      const int li = tile * 16 + (threadIdx.x >> 3);
      float go = 0.f;
      if (li < n_live)
        go = B.grad_out[chunk * PC_Q + ((const int*)(lds + C_LIST))[li]] * voxel * lds[L_WTRI + threadIdx.x];
      lds[L_ALPHA + threadIdx.x] = go;
    }
    __syncthreads();
    if (w < 4) {
      const _Float16* ph = (const _Float16*)A.pack;
      const _Float16* pb = (const _Float16*)B.bwd_pack;
      const int col = w * 32 + j;
      // ---- forward, keeping the sign bits of the three hidden pre-activations -------------------------
      f32x16 a0[2], a1[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) a0[mb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const half8 b = *(const half8*)&lds[L_HL + ((ks * 2 + h) * DM + col) * 4];
          a0[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8*)&ph[ST_W0 + ((mb * 2 + ks) * 64 + lane) * 8],
                                                          b, a0[mb], 0, 0, 0);
        }
      }
      half8 s[4];
      auto fill_relu = [&](const f32x16 (&in)[2]) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          s[nb * 2] = relu_half8(in[nb], 0);
          s[nb * 2 + 1] = relu_half8(in[nb], 8);
        }
      };
      auto fill_masked = [&](const f32x16 (&in)[2], uint32_t m) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
          for (int ksl = 0; ksl < 2; ++ksl) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              s[nb * 2 + ksl][e] = (_Float16)(((m >> (nb * 16 + ksl * 8 + e)) & 1u) ? in[nb][ksl * 8 + e] : 0.f);
          }
        }
      };
      auto layer64 = [&](const _Float16* wp, f32x16 (&out)[2]) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
          for (int r = 0; r < 16; ++r) out[mb][r] = 0.f;
#pragma unroll
          for (int g = 0; g < 4; ++g)
            out[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8*)&wp[((mb * 4 + g) * 64 + lane) * 8], s[g],
                                                            out[mb], 0, 0, 0);
        }
      };
      const uint32_t m0 = positive_bits32(a0);
      fill_relu(a0);
      layer64(ph + ST_W1, a1);
      const uint32_t m1 = positive_bits32(a1);
      fill_relu(a1);
      layer64(ph + ST_W2, a0);
      const uint32_t m2 = positive_bits32(a0);
      // ---- backward with a unit seed: delta_2 = W3[0, :] * [z2 > 0] ------------------------------------
      {
        const float* w3r = (const float*)(pb + TB_W3R);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const f32x4 t = *(const f32x4*)&w3r[mb * 32 + 8 * qd + 4 * h];
#pragma unroll
            for (int i = 0; i < 4; ++i) a0[mb][4 * qd + i] = t[i];
          }
        }
      }
      fill_masked(a0, m2);
      layer64(pb + TB_W2T, a1);
      fill_masked(a1, m1);
      layer64(pb + TB_W1T, a0);
      fill_masked(a0, m0);
      f32x16 g;
#pragma unroll
      for (int r = 0; r < 16; ++r) g[r] = 0.f;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        g = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8*)&pb[TB_W0T + (gq * 64 + lane) * 8], s[gq], g, 0, 0, 0);
      const float sc = lds[L_ALPHA + col];
      const int row = l_row[col];
      if (sc != 0.f && row >= 0) {
        float* gf = B.grad_features + (size_t)row * 8;
        if (h == 0) {
          unsafeAtomicAdd(gf + 0, g[5] * sc);
          unsafeAtomicAdd(gf + 1, g[6] * sc);
          unsafeAtomicAdd(gf + 2, g[7] * sc);
          unsafeAtomicAdd(gf + 7, g[8] * sc);
        } else {
          unsafeAtomicAdd(gf + 3, g[4] * sc);
          unsafeAtomicAdd(gf + 4, g[5] * sc);
          unsafeAtomicAdd(gf + 5, g[6] * sc);
          unsafeAtomicAdd(gf + 6, g[7] * sc);
        }
      }
    }
    __syncthreads();
   }
  }
}

// ---------------------------------------------------------------------------------------------------
// The lattice-table kernel of the split-operand modes (k_lattice_table_x below) is software-pipelined ACROSS
// tiles and layers.  Same arithmetic as k_decode<LATTICE, 1>, in another summation grouping (tables equal to
// ~1e-8); what changes is when things are fetched:
//  * the work-list entry of tile t+2 and the features of tile t+1 are loaded while tile t runs its MLP;
//    the inputs of tile t+1 are split and staged into a separate LDS buffer (PARK) in the shadow of layer
//    0's store phase, and layer 0 reads its B operands from PARK -- no gather on the critical path;
//  * the weight-fragment ring runs continuously through the 50 units of a tile and on into the next
//    tile: the first fragments of layer L+1 are requested during the last units of layer L, so they
//    arrive during the convert/store phase and its barriers;
//  * sin / cos of the lattice offsets {-.5, 0, .5} are two constants; the final reduction writes the table
//    directly; 7 barriers per tile.
// (Round 1's and 2's version of this kernel on v_mfma_f32_32x32x16_f16, k_lattice_table_h, is in the git history
// up to commit c19da39: bit-identical to k_decode<LATTICE, 1>, 5 % slower than the 16x16x32 form.)
// ---------------------------------------------------------------------------------------------------
constexpr int T_PARK_HI = L_PART + 16 * DM;            // [4 octets][128 evaluations][8 halves] = 8 KB
constexpr int T_PARK_LO = T_PARK_HI + 2 * 2 * DM * 4;  // lo plane
constexpr int T_TOTAL = T_PARK_LO + 2 * 2 * DM * 4;    // 38,912 floats = 155,648 B
constexpr int kTRing = 5, kTAhead = 3;                 // 50 units per tile: 50 % 5 == 0 keeps the ring phase

struct ARing {
  half8 hi[kTRing], lo[kTRing];
};

// (row, l) of evaluation e of the work list, or row = -1 beyond its end
__device__ __forceinline__ int lattice_entry(const DecodeArgs& A, int64_t e, int64_t n_evals) {
  if (e >= n_evals) return -1;
  if (A.entries) return A.entries[e];
  const int64_t ci = e / 27;
  return (A.list[ci] << 5) | (int)(e - ci * 27);
}

// ---------------------------------------------------------------------------------------------------
// k_lattice_table_x: the same tables on v_mfma_f32_16x16x32_f16.
// Both MLP kernels run at the package power limit (tools/power_probe.py), and under that limit the 16x16x32 form
// delivers ~14 % more FLOP/s than the 32x32x16 form (tools/probe_shapes.hip: 1.88 against 1.65 PFLOP/s with random
// f16 operands, MFMA-only streams): half the accumulator traffic per FLOP.  Same tile (128 evaluations), same
// split arithmetic (three products, fp32 accumulation), same bytes from L2 and LDS; what changes is the shape of
// a wave's work and therefore every layout:
//  * wave w still owns output features [32 w, 32 w + 32) of every layer for all 128 evaluations: 2 row blocks
//    (rb) of 16 features x 8 column blocks (cb) of 16 evaluations = 16 accumulators of 4 registers.  Lane
//    (n = l & 15, g = l >> 4), register i of acc[rb][cb] is feature 32 w + 16 rb + 4 g + i of evaluation 16 cb + n;
//  * a K-step is 32 deep: operand slot jj of a lane of K-group g is K index 8 g + jj.  The activations live in LDS
//    as OCTETS [32 octets][128 evaluations][8 halves] (hi plane, lo plane 64 KB behind): octet 4 s + g of K-step
//    s.  A lane's 8 registers {acc[0][cb][0..3], acc[1][cb][0..3]} are exactly one octet (4 w + g) of the next
//    layer's input, so the epilogue is again one ds_write_b128 per plane and column block, and K-step s of the
//    next layer consumes what wave s produced: slot jj <-> feature 32 s + 16 (jj >> 2) + 4 g + (jj & 3).  The weight
//    fragments are packed to that order on the host (weights.py: _pack_split16; SX_* below);
//  * one UNIT = half a K-step = 24 MFMAs of 16 cycles = the 384 cycles of a 32x32x16 K-step, so the weight ring
//    (5 units of one hi + one lo fragment, 3 ahead, 50 units per tile) and the activation double buffer carry
//    over unchanged (chain_layer_x).
// ---------------------------------------------------------------------------------------------------
constexpr int SX_W0 = 0;                              // [8 w][2 units][2 hi/lo][64 lane][8]
constexpr int SX_W1 = SX_W0 + 8 * 2 * 2 * 64 * 8;     // [8 w][16 units][2 hi/lo][64 lane][8]
constexpr int SX_W2 = SX_W1 + 8 * 16 * 2 * 64 * 8;
constexpr int SX_W3 = SX_W2 + 8 * 16 * 2 * 64 * 8;
constexpr int SX_TOTAL = SX_W3 + 8 * 16 * 2 * 64 * 8;  // 409,600 halves, behind the SH_* pack
static_assert(SX_TOTAL == SH_TOTAL, "16x16x32 pack size");
constexpr int SD_PACK_FLOATS_X = SD_PACK_FLOATS + SX_TOTAL / 2;

typedef __attribute__((address_space(3))) const half8 lds_half8_t;
typedef __attribute__((address_space(3))) half8 lds_half8_w_t;

// One layer.  NU = units of this layer (2 for layer 0, 16 for the others), BASE = units before it within the tile
// (ring phase).  A WEIGHT unit is (K-step s, row block rb), index 2 s + rb: one hi + one lo fragment; a COMPUTE unit
// is (K-step s, column half ch), index 2 s + ch: both row blocks x 4 column blocks x 3 products = 24 MFMAs, reading
// both weight units of its K-step and 4 + 4 activation fragments.  The ring holds weight units 0 .. kTAhead-1 on
// entry; compute unit u requests weight unit u + kTAhead (the last ones those of the NEXT layer) and the activation
// fragments of compute unit u + 1 (double buffer) -- per unit 24 MFMAs, 8 LDS reads, 2 L2 reads, like a K-step of
// the 32x32x16 kernel.  b_hi / b_lo: this lane's LDS byte address of octet g, evaluation n in the source planes.
// HALF: a 64-evaluation tile (the tail of a launch, k_lattice_table_x): only the compute units of column half 0 run;
// the weight ring keeps its schedule (every weight unit is still needed), the activation fragments of K-step s + 1
// are requested during K-step s.
template <int NU, int BASE, int NEXT_NU, int NPROD, bool HALF = false>
__device__ __forceinline__ void chain_layer_x(__amdgpu_buffer_rsrc_t rs, int voff, int off, int off_next,
                                              const float* __restrict__ bias, uint32_t b_hi, uint32_t b_lo,
                                              ARing& ring, f32x4 (&acc)[2][8], int w, int g) {
  const f32x4 bias0 = *(const f32x4*)&bias[32 * w + 4 * g];
  const f32x4 bias1 = *(const f32x4*)&bias[32 * w + 16 + 4 * g];
#pragma unroll
  for (int cb = 0; cb < 8; ++cb) {
    acc[0][cb] = bias0;
    acc[1][cb] = bias1;
  }
  const int sl = off + w * NU * 2048;
  const int sn = off_next + w * NEXT_NU * 2048;
  half8 bh[2][4], bl[2][4];
#define BNV_LOAD_B(u)                                                                                         \
  {                                                                                                           \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                           \
      const uint32_t o = (uint32_t)(((u) >> 1) * 8192 + (((u) & 1) * 4 + c) * 256);                           \
      bh[(u) & 1][c] = *(lds_half8_t*)(b_hi + o);                                                             \
      if (NPROD == 3) bl[(u) & 1][c] = *(lds_half8_t*)(b_lo + o);                                             \
    }                                                                                                         \
  }
#define BNV_LOAD_BH(s_)                                                                                       \
  {                                                                                                           \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                           \
      const uint32_t o = (uint32_t)((s_) * 8192 + c * 256);                                                   \
      bh[(s_) & 1][c] = *(lds_half8_t*)(b_hi + o);                                                            \
      if (NPROD == 3) bl[(s_) & 1][c] = *(lds_half8_t*)(b_lo + o);                                            \
    }                                                                                                         \
  }
  if constexpr (HALF) {
    BNV_LOAD_BH(0);
  } else {
    BNV_LOAD_B(0);
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int s = u >> 1, ch = u & 1;
    const int nx = u + kTAhead;   // the weight unit requested during this compute unit
    bool loads_a = false;
    if (nx < NU) {
      ring.hi[(BASE + nx) % kTRing] = load_frag(rs, voff, sl + nx * 2048);
      if (NPROD == 3) ring.lo[(BASE + nx) % kTRing] = load_frag(rs, voff, sl + nx * 2048 + 1024);
      loads_a = true;
    } else if (nx - NU < NEXT_NU && nx - NU < kTAhead) {
      ring.hi[(BASE + nx) % kTRing] = load_frag(rs, voff, sn + (nx - NU) * 2048);
      if (NPROD == 3) ring.lo[(BASE + nx) % kTRing] = load_frag(rs, voff, sn + (nx - NU) * 2048 + 1024);
      loads_a = true;
    }
    if constexpr (HALF) {
      if (ch == 1) {   // nothing to compute in this unit of a half tile; its weight request stays
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      if (2 * (s + 1) < NU) BNV_LOAD_BH(s + 1);
    } else {
      if (u + 1 < NU) BNV_LOAD_B(u + 1);
    }
    const int bb = HALF ? (s & 1) : (u & 1);
    if constexpr (NPROD == 3) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          acc[rb][4 * ch + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring.hi[(BASE + 2 * s + rb) % kTRing], bl[bb][c],
                                                                       acc[rb][4 * ch + c], 0, 0, 0);
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          acc[rb][4 * ch + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring.lo[(BASE + 2 * s + rb) % kTRing], bh[bb][c],
                                                                       acc[rb][4 * ch + c], 0, 0, 0);
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        acc[rb][4 * ch + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring.hi[(BASE + 2 * s + rb) % kTRing], bh[bb][c],
                                                                     acc[rb][4 * ch + c], 0, 0, 0);
    // issue order: every prefetch in the shadow of an MFMA (one memory instruction behind each)
    if (HALF ? (2 * (s + 1) < NU) : (u + 1 < NU)) {
#pragma unroll
      for (int q = 0; q < (NPROD == 3 ? 8 : 4); ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
      }
    }
    if (loads_a) {
#pragma unroll
      for (int q = 0; q < (NPROD == 3 ? 2 : 1); ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#undef BNV_LOAD_BH
#undef BNV_LOAD_B
}

// ReLU + hi/lo split of a wave's 32 features x 128 evaluations into octet 4 w + g of the activation planes, one
// column block at a time (conversion and ds_write_b128 interleaved), BEHIND the barrier that frees the planes.
// (Converting before that barrier -- the older wave of a SIMD wins MFMA arbitration and leaves the K-loop ~6,000
// cycles early, tools/phase_prof.py -- was measured: its VALU stream then takes issue slots from the younger
// wave's MFMAs and the tile gets 2.6 % longer; converting all blocks before the first store: +1.7 %.)
template <int NPROD, bool HALF = false>
__device__ __forceinline__ void store_relu_x(uint32_t st_hi, uint32_t st_lo, const f32x4 (&acc)[2][8]) {
#pragma unroll
  for (int cb = 0; cb < (HALF ? 4 : 8); ++cb) {
    half8 hi, lo;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = relu1(acc[e >> 2][cb][e & 3]);
    if (NPROD == 3) {
      split8_f16(x, hi, lo);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) hi[e] = (_Float16)x[e];
    }
    *(lds_half8_w_t*)(st_hi + (uint32_t)(cb * 256)) = hi;
    if (NPROD == 3) *(lds_half8_w_t*)(st_lo + (uint32_t)(cb * 256)) = lo;
  }
}

#ifdef BNV_PHASE_PROF
#define BNV_PHX(i)                                                                 \
  do {                                                                             \
    if ((threadIdx.x & 63) == 0) {                                                 \
      unsigned long long* _p = (unsigned long long*)(lds + T_TOTAL) + (threadIdx.x >> 6) * 32; \
      const unsigned long long _t = clock64();                                     \
      _p[i] += _t - _p[31];                                                        \
      _p[31] = _t;                                                                 \
    }                                                                              \
  } while (0)
#else
#define BNV_PHX(i)
#endif

template <int NPROD>
__global__ __launch_bounds__(512, 2) void k_lattice_table_x(DecodeArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const float voxel = A.grid.voxel_size;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int64_t n_evals = A.entries ? (int64_t)A.n_list[1] : (int64_t)(*A.n_list) * 27;
  // HALF TILES in the tail.  A launch of T tiles on G workgroups takes ceil(T / G) rounds, and a shard of a spatially
  // sharded volume has only 5-8 tiles per workgroup: 7.2 tiles cost 8 rounds.  When the last round holds R <= G / 2
  // tiles, they are handed out as 2 R tiles of 64 evaluations (same weights from L2, half the MFMAs: ~0.6 of a round).
  const int64_t n_full_all = (n_evals + DM - 1) / DM;
  const int64_t rem = n_full_all % (int64_t)gridDim.x;
  const int64_t n_full = (A.half_tail && rem > 0 && 2 * rem <= (int64_t)gridDim.x) ? n_full_all - rem : n_full_all;
  const int64_t half0 = n_full * DM;   // first evaluation of the half tiles
  const int64_t n_tiles = n_full + (n_evals > half0 ? (n_evals - half0 + 63) / 64 : 0);
  // entry of evaluation slot se of a tile (-1: none)
  auto tile_entry = [&](int64_t t, int slot) -> int {
    if (t >= n_tiles) return -1;
    if (t < n_full) return lattice_entry(A, t * DM + slot, n_evals);
    return slot < 64 ? lattice_entry(A, half0 + (t - n_full) * 64 + slot, n_evals) : -1;
  };
  const float* pack = A.pack;
  const _Float16* px = (const _Float16*)(pack + SD_PACK_FLOATS);
  const float s5 = sinf(0.5f), c5 = cosf(0.5f);
  // Division of the per-tile side work (tools/phase_prof.py: with everything on threads 0..127 waves 0 and 1 were
  // ~1,400 cycles behind the others at two barriers of every tile):
  //  * staging of the next tile's network inputs: thread t stages octet so = t >> 7 (inputs 8 so .. 8 so + 7) of
  //    evaluation se = t & 127; octet 3 (inputs 24..31) is zero for good and written once;
  //  * the final 16-partial reduction and the table write: threads 128..255 (waves 2 and 3).
  const int se = threadIdx.x & (DM - 1);
  const int so = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7);
  const bool writer = so == 1;

  // inputs 8 so .. 8 so + 7 of (entry ent, features f0 f1) into PARK at evaluation se
  auto stage_park = [&](int ent, const f32x4& f0, const f32x4& f1) {
    if (so == 3) return;
    float in[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) in[f] = 0.f;
    if (ent >= 0) {
      const int l = ent & 31;
      const int lx = l / 9 - 1, ly = (l / 3) % 3 - 1, lz = l % 3 - 1;
      if (so == 0) {
        const int li[3] = {lx, ly, lz};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          in[a] = (float)li[a] * 0.5f;
          in[3 + a] = li[a] == 0 ? 0.f : (li[a] > 0 ? s5 : -s5);
        }
        in[6] = lx == 0 ? 1.f : c5;
        in[7] = ly == 0 ? 1.f : c5;
      } else if (so == 1) {
        const float fe[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
        check_feature_range(fe, pack[SD_BA + 1], A.vol.n_rows);
        in[0] = lz == 0 ? 1.f : c5;
#pragma unroll
        for (int f = 0; f < 4; ++f) in[1 + f] = f0[f];
#pragma unroll
        for (int f = 0; f < 3; ++f) in[5 + f] = f1[f];
      } else {
        in[0] = f1[3];
      }
    } else if (so == 0) {
      in[6] = in[7] = 1.f;   // what k_decode stages for an empty column: cos(0)
    } else if (so == 1) {
      in[0] = 1.f;
    }
    half8 hi, lo;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const _Float16 t = (_Float16)in[jj];
      hi[jj] = t;
      if (NPROD == 3) lo[jj] = (_Float16)(in[jj] - (float)t);
    }
    *(half8*)&lds[T_PARK_HI + (so * DM + se) * 4] = hi;
    if (NPROD == 3) *(half8*)&lds[T_PARK_LO + (so * DM + se) * 4] = lo;
  };
  auto load_feats = [&](int ent, f32x4& f0, f32x4& f1) {
    if (ent >= 0 && (so == 1 || so == 2)) {
      const size_t row = (size_t)(ent >> 5);
      if (so == 1) f0 = *(const f32x4*)&A.features[row * 8];
      f1 = *(const f32x4*)&A.features[row * 8 + 4];
    }
  };

  // ---- tiles are handed out DYNAMICALLY (one atomic per tile on a counter in the workspace, fetched one tile
  // ahead): when another stream's kernel still holds some CUs at launch (the next frame's encoder, an RCCL
  // collective), the workgroups that start late simply take fewer tiles instead of stretching the kernel's tail.
  // The first two tiles of a workgroup are static (b, b + grid): 2 x 256 atomics on ONE address at the start of every
  // launch serialised in the memory-side atomic unit for ~6 us before the first MFMA; dynamic ids start at 2 x grid.
  // Every loop iteration takes exactly one id and every tile is one iteration, so the counter ends at n_tiles: the
  // thread that draws n_tiles - 1 has drawn the launch's last id and puts the counter back to 0 for the next launch
  // (no exit count, no fence).
  __shared__ int s_tile[3];
  int* tile_ctr = (int*)A.n_list + 2;
  const int64_t dyn0 = 2 * (int64_t)gridDim.x;
  if (threadIdx.x == 0) {
    s_tile[0] = (int)blockIdx.x;
    s_tile[1] = (int)(blockIdx.x + gridDim.x);
  }
  if (so == 3) {   // octet 3 of PARK: inputs 24..31, always zero
    const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    *(half8*)&lds[T_PARK_HI + (3 * DM + se) * 4] = z;
    *(half8*)&lds[T_PARK_LO + (3 * DM + se) * 4] = z;
  }
  __syncthreads();
  int64_t tile = s_tile[0], tile_nx = s_tile[1];
  int ent_cur = -1, ent_nx = -1;
  f32x4 f0 = {0.f, 0.f, 0.f, 0.f}, f1 = {0.f, 0.f, 0.f, 0.f};
  ent_cur = tile_entry(tile, se);
  ent_nx = tile_entry(tile_nx, se);
  load_feats(ent_cur, f0, f1);
  stage_park(ent_cur, f0, f1);
  ARing ring;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)px, 0, SX_TOTAL * 2, 0x00020000);
  const int voff = lane * 16;
  constexpr int O0 = SX_W0 * 2, O1 = SX_W1 * 2, O2 = SX_W2 * 2, O3 = SX_W3 * 2;  // byte offsets of the layers
  {
#pragma unroll
    for (int p = 0; p < 2; ++p) {  // layer 0 has 2 units; its third request slot belongs to layer 1
      ring.hi[p] = load_frag(rs, voff, O0 + (w * 2 + p) * 2048);
      if (NPROD == 3) ring.lo[p] = load_frag(rs, voff, O0 + (w * 2 + p) * 2048 + 1024);
    }
    ring.hi[2] = load_frag(rs, voff, O1 + (w * 16) * 2048);
    if (NPROD == 3) ring.lo[2] = load_frag(rs, voff, O1 + (w * 16) * 2048 + 1024);
  }
  // LDS byte addresses of this lane (opaque to the optimiser: base + 16-bit immediates, encode.hip: EncLds)
  uint32_t act_hi, act_lo, park_hi, park_lo, st_hi, st_lo;
  {
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
    const uint32_t lane_off = (uint32_t)(g * DM + n) * 16u;
    act_hi = lds0 + L_HL * 4 + lane_off;
    act_lo = lds0 + L_HLO * 4 + lane_off;
    park_hi = lds0 + T_PARK_HI * 4 + lane_off;
    park_lo = lds0 + T_PARK_LO * 4 + lane_off;
    st_hi = act_hi + (uint32_t)w * 8192u;     // octet 4 w + g
    st_lo = act_lo + (uint32_t)w * 8192u;
    asm volatile("" : "+v"(act_hi), "+v"(act_lo), "+v"(park_hi), "+v"(park_lo), "+v"(st_hi), "+v"(st_lo));
  }
  // fc_alpha weights of this lane's eight features
  f32x4 wa0, wa1;
  wa0 = *(const f32x4*)&pack[SD_WA + 32 * w + 4 * g];
  wa1 = *(const f32x4*)&pack[SD_WA + 32 * w + 16 + 4 * g];
  __syncthreads();
#ifdef BNV_PHASE_PROF
  if (threadIdx.x < 256) ((unsigned long long*)(lds + T_TOTAL))[threadIdx.x] = 0;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) ((unsigned long long*)(lds + T_TOTAL))[(threadIdx.x >> 6) * 32 + 31] = clock64();
#endif

  auto one_tile = [&](auto half_tag) {
    constexpr bool HALF = decltype(half_tag)::value;
    // requests for the following tiles: the id of the tile after next (thread 0 asks now and publishes it behind
    // layer 0, so that the round trip of the atomic is off its wave's critical path), features of the next tile
    int next_id = 0;
    if (threadIdx.x == 0) {
      const int drawn = atomicAdd(tile_ctr, 1);
      if ((int64_t)drawn == n_tiles - 1) *tile_ctr = 0;   // the last draw of this launch
      next_id = (int)(dyn0 + drawn);
    }
    int ent_nx2 = -1;
    f0 = f32x4{0.f, 0.f, 0.f, 0.f};
    f1 = f32x4{0.f, 0.f, 0.f, 0.f};
    load_feats(ent_nx, f0, f1);
    f32x4 acc[2][8];
    BNV_PHX(0);
    chain_layer_x<2, 0, 16, NPROD, HALF>(rs, voff, O0, O1, pack + SD_B0, park_hi, park_lo, ring, acc, w, g);
    if (threadIdx.x == 0) s_tile[2] = next_id;
    BNV_PHX(1);
    __syncthreads();
    BNV_PHX(2);
    const int64_t tile_nx2 = s_tile[2];
    store_relu_x<NPROD, HALF>(st_hi, st_lo, acc);
    stage_park(ent_nx, f0, f1);  // PARK is free: every wave is past layer 0
    ent_nx2 = tile_entry(tile_nx2, se);
    BNV_PHX(3);
    __syncthreads();
    BNV_PHX(4);
    chain_layer_x<16, 2, 16, NPROD, HALF>(rs, voff, O1, O2, pack + SD_B0 + 256, act_hi, act_lo, ring, acc, w, g);
    BNV_PHX(5);
    __syncthreads();
    BNV_PHX(6);
    store_relu_x<NPROD, HALF>(st_hi, st_lo, acc);
    BNV_PHX(7);
    __syncthreads();
    BNV_PHX(8);
    chain_layer_x<16, 18, 16, NPROD, HALF>(rs, voff, O2, O3, pack + SD_B0 + 512, act_hi, act_lo, ring, acc, w, g);
    BNV_PHX(9);
    __syncthreads();
    BNV_PHX(10);
    store_relu_x<NPROD, HALF>(st_hi, st_lo, acc);
    BNV_PHX(11);
    __syncthreads();
    BNV_PHX(12);
    chain_layer_x<16, 34, 2, NPROD, HALF>(rs, voff, O3, O0, pack + SD_B0 + 768, act_hi, act_lo, ring, acc, w, g);
    ring.hi[2] = load_frag(rs, voff, O1 + (w * 16) * 2048);  // (50 + 2) % 5: layer 1's unit 0, next tile
    if (NPROD == 3) ring.lo[2] = load_frag(rs, voff, O1 + (w * 16) * 2048 + 1024);
    BNV_PHX(13);
    // fc_alpha: 256 -> 1.  Partial over this lane's 8 features, K-groups g and g + 2 combined across the lane
    // halves, 16 partials per evaluation through LDS
#pragma unroll
    for (int cb = 0; cb < (HALF ? 4 : 8); ++cb) {
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) sum = fmaf(wa0[i], relu_bits(acc[0][cb][i]), sum);
#pragma unroll
      for (int i = 0; i < 4; ++i) sum = fmaf(wa1[i], relu_bits(acc[1][cb][i]), sum);
      sum += __shfl_xor(sum, 32, 64);
      if (g < 2) lds[L_PART + (w * 2 + g) * DM + cb * 16 + n] = sum;
    }
    BNV_PHX(14);
    __syncthreads();
    BNV_PHX(15);
    if (writer && ent_cur >= 0) {
      float sum = pack[SD_BA];
#pragma unroll
      for (int p = 0; p < 16; ++p) sum += lds[L_PART + p * DM + se];
      const int row = ent_cur >> 5;
      A.table[(size_t)row * 27 + (ent_cur & 31)] = __fmul_rn(sum, voxel);
      if (A.entries && A.need_mask) A.need_mask[row] = 0u;  // leave the per-row masks clean for the next call
    }
    ent_cur = ent_nx;
    ent_nx = ent_nx2;
    tile = tile_nx;
    tile_nx = tile_nx2;
    BNV_PHX(17);
  };
  while (tile < n_tiles) {
    if (__builtin_amdgcn_readfirstlane((int)(tile >= n_full))) one_tile(std::true_type{});
    else one_tile(std::false_type{});
  }
#ifdef BNV_PHASE_PROF
  __syncthreads();
  if (threadIdx.x < 256 && (threadIdx.x & 31) != 31)
    atomicAdd(&g_phase_cycles[threadIdx.x], ((unsigned long long*)(lds + T_TOTAL))[threadIdx.x]);
  if (threadIdx.x == 0) atomicAdd(&g_phase_cycles[31], 1ull);
#endif
}

// ---------------------------------------------------------------------------------------------------
// k_lattice_table_t: the lattice tables with the tiny-cuda-nn decoder (MLP mode 2), one WAVE per 32 entries.
// The generic k_decode<LATTICE, 2> is built around the fp32 decoder's tile: 512 threads, 141 KB of LDS (one workgroup
// per CU), the 128 staged inputs of a tile pass through LDS behind a barrier and only four of the eight waves run
// the (tiny) network: 0.162 ms per frame, 40 % of the tcnn frame's MLP time for 6 % of its FLOPs.  Here a wave
// loads its 32 entries, gathers their feature rows, builds the network's B operands in registers (positional
// encoding of a lattice offset: three values of {0, +-0.5} and their sin / cos), runs all four layers in registers
// with the weights in LDS (24.5 KB, staged once per workgroup) and writes its 32 table entries: no barrier, no LDS
// traffic for activations, 256-thread workgroups, four to five waves per SIMD.  Same operand values in the same MFMA
// order as the generic kernel: bit-identical tables (tests/test_gpu_parity.py).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lattice_table_t(DecodeArgs A) {
  __shared__ __attribute__((aligned(16))) _Float16 wh[ST_TOTAL];
  stage_to_lds<256>(A.pack, wh, ST_TOTAL * 2);
  __syncthreads();
  const float voxel = A.grid.voxel_size;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int64_t n_evals = A.entries ? (int64_t)A.n_list[1] : (int64_t)(*A.n_list) * 27;
  const int64_t n_tiles = (n_evals + 31) / 32;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < n_tiles; t += (int64_t)gridDim.x * 4) {
    const int ent = lattice_entry(A, t * 32 + j, n_evals);
    float in[32];
#pragma unroll
    for (int f = 0; f < 32; ++f) in[f] = 1.0f;       // inputs 17..31: the padding of the tcnn encoding
    int row = 0, l = 0;
    if (ent >= 0) {
      row = ent >> 5;
      l = ent & 31;
      const float loc[3] = {(float)(l / 9 - 1) * 0.5f, (float)((l / 3) % 3 - 1) * 0.5f, (float)(l % 3 - 1) * 0.5f};
      const f32x4 f0 = *(const f32x4*)&A.features[(size_t)row * 8];
      const f32x4 f1 = *(const f32x4*)&A.features[(size_t)row * 8 + 4];
      // a lattice offset is -0.5, 0 or +0.5: its encoding is one of three constants (sinf / cosf cost more than
      // the tile's MFMAs; the operands are rounded to f16 below, where sin(0.5) and cos(0.5) sit 0.23 and 0.29 of a
      // spacing away from the nearest rounding boundary: the last bit of the fp32 value cannot matter)
      constexpr float kSinHalf = 0.479425538604203f, kCosHalf = 0.8775825618903728f;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        in[a] = loc[a];
        in[3 + a] = loc[a] == 0.f ? 0.f : (loc[a] > 0.f ? kSinHalf : -kSinHalf);
        in[6 + a] = loc[a] == 0.f ? 1.f : kCosHalf;
      }
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        in[9 + f] = f0[f];
        in[13 + f] = f1[f];
      }
    } else {
#pragma unroll
      for (int f = 0; f < 17; ++f) in[f] = 0.f;      // (an empty column; its output is not written)
    }
    // operand slot jj of K-step ks of this lane half: input 16 ks + 8 (jj >> 2) + 4 h + (jj & 3)  (stage_input_t)
    half8 b[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const float lo = in[16 * ks + 8 * (jj >> 2) + (jj & 3)], hi = in[16 * ks + 8 * (jj >> 2) + 4 + (jj & 3)];
        b[ks][jj] = (_Float16)(h ? hi : lo);
      }
    f32x16 a0[2], a1[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) a0[mb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        a0[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8*)&wh[ST_W0 + ((mb * 2 + ks) * 64 + lane) * 8], b[ks],
                                                        a0[mb], 0, 0, 0);
    }
    half8 s4[4];
    auto layer64 = [&](int woff, const f32x16 (&inp)[2], f32x16 (&out)[2]) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        s4[nb * 2] = relu_half8(inp[nb], 0);
        s4[nb * 2 + 1] = relu_half8(inp[nb], 8);
      }
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) out[mb][r] = 0.f;
#pragma unroll
        for (int gk = 0; gk < 4; ++gk)
          out[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8*)&wh[woff + ((mb * 4 + gk) * 64 + lane) * 8],
                                                          s4[gk], out[mb], 0, 0, 0);
      }
    };
    layer64(ST_W1, a0, a1);
    layer64(ST_W2, a1, a0);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      s4[nb * 2] = relu_half8(a0[nb], 0);
      s4[nb * 2 + 1] = relu_half8(a0[nb], 8);
    }
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int gk = 0; gk < 4; ++gk)
      o = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8*)&wh[ST_W3 + (gk * 64 + lane) * 8], s4[gk], o, 0, 0, 0);
    // output 0 = row 0 of the tile = register 0 of the lanes with h == 0; the network returns fp16, and
    // half tensor * python float stays half (sparse_volume.py:813)
    if (h == 0 && ent >= 0) {
      float av = __fmul_rn((float)(_Float16)o[0], voxel);
      av = (float)(_Float16)av;
      A.table[(size_t)row * 27 + l] = av;
      if (A.entries && A.need_mask) A.need_mask[row] = 0u;   // leave the per-row masks clean for the next call
    }
  }
}

// ---- lattice decode: neighbour lookup + blend ------------------------------------------------
struct LatticeWs {
  int32_t* nbr_rows;  // [n][27]
  int32_t* list;      // [list_capacity] rows whose table is needed
  int32_t* n_list;    // [1]
  int32_t* stamp;     // [row_capacity]
  float* table;       // [row_capacity][27]
  uint32_t* need_mask;  // [row_capacity] bit l set: table[row][l] is read by a live lattice point
  int32_t* origin_stamp;  // [row_capacity] == epoch: the row's voxel is a decoded origin of this call
  int32_t* entries;   // [entry_capacity] (row << 5) | l
  int64_t list_capacity;
  int64_t entry_capacity;
};

static size_t lattice_ws_layout(int64_t n, int64_t row_capacity, char* base, LatticeWs* ws) {
  if (n < 1) n = 1;
  int64_t cap = 27 * n;
  if (cap > row_capacity) cap = row_capacity;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off = (off + bytes + 255) / 256 * 256;
    return p;
  };
  // stamp and table first: they persist across calls with the same row_capacity
  char* st = take(row_capacity * 4);
  char* tb = take(row_capacity * 27 * 4);
  char* nm = take(row_capacity * 4);
  char* os = take(row_capacity * 4);
  char* nl = take(256);
  char* nb = take(n * 27 * 4);
  char* li = take(cap * 4);
  int64_t ecap = 27 * cap;
  if (ecap > 216 * n) ecap = 216 * n;
  char* en = take(ecap * 4);
  if (ws) {
    ws->need_mask = (uint32_t*)nm;
    ws->origin_stamp = (int32_t*)os;
    ws->entries = (int32_t*)en;
    ws->entry_capacity = ecap;
    ws->stamp = (int32_t*)st;
    ws->table = (float*)tb;
    ws->n_list = (int32_t*)nl;
    ws->nbr_rows = (int32_t*)nb;
    ws->list = (int32_t*)li;
    ws->list_capacity = cap;
  }
  return off;
}

void lattice_ws_frame_words(void* ws_ptr, int64_t row_capacity, int32_t** origin_stamp, int32_t** ctl) {
  LatticeWs ws;
  lattice_ws_layout(1, row_capacity, (char*)ws_ptr, &ws);   // both sit in the part that depends on row_capacity only
  *origin_stamp = ws.origin_stamp;
  *ctl = ws.n_list;
}

constexpr int kOriginBit = 1 << 30;   // flag in nbr_rows entries (rows are < 2^30)

// origin_stamp[row of origin b] = epoch: which rows are decoded origins of this call
__global__ __launch_bounds__(256) void k_lattice_stamp(bnv_volume_t v, const int64_t* __restrict__ origins, int64_t n,
                                                       int64_t row_limit, int32_t* __restrict__ origin_stamp,
                                                       int32_t epoch, const int32_t* __restrict__ n_dev,
                                                       int32_t* __restrict__ n_list) {
  if (n_dev) n = (int64_t)*n_dev < n ? (int64_t)*n_dev : n;
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  // the control words of the stages behind (entries listed, tile counter of the table kernel, spare) are cleared
  // here: saves bnv_decode_lattice a memset launch per call
  if (n_list && b == 0) n_list[1] = n_list[2] = n_list[3] = 0;
  if (b >= n) return;
  const int row = volume_row(v, origins[b * 3 + 0], origins[b * 3 + 1], origins[b * 3 + 2]);
  if (row >= 0 && row < row_limit) origin_stamp[row] = epoch;
}

// row (| kOriginBit) of neighbour nb (0..26) of origin b, or -1: absent, below min_pts (such rows can only ever appear
// under a false mask) or beyond row_limit
__device__ __forceinline__ int lattice_neighbor_row(const bnv_volume_t& v, const int64_t* __restrict__ origins, int64_t b,
                                                    int nb, const float* __restrict__ weights, int64_t row_limit,
                                                    float min_pts, const int32_t* __restrict__ origin_stamp,
                                                    int32_t epoch) {
  const int64_t x = origins[b * 3 + 0] + (nb / 9 - 1);
  const int64_t y = origins[b * 3 + 1] + ((nb / 3) % 3 - 1);
  const int64_t z = origins[b * 3 + 2] + (nb % 3 - 1);
  int row = volume_row(v, x, y, z);
  if (row >= row_limit) row = -1;
  if (row < 0 || !(weights[row] >= min_pts)) return -1;
  const bool is_origin = origin_stamp && origin_stamp[row] == epoch;
  return row | (is_origin ? (1 << 30) : 0);
}

__global__ __launch_bounds__(256) void k_lattice_neighbors(bnv_volume_t v, const int64_t* __restrict__ origins,
                                                           int64_t n, const float* __restrict__ weights,
                                                           int64_t row_limit, float min_pts,
                                                           int32_t* __restrict__ nbr_rows,
                                                           int32_t* __restrict__ stamp, int32_t epoch,
                                                           int32_t* __restrict__ list, int32_t* __restrict__ n_list,
                                                           const uint8_t* __restrict__ row_skip,
                                                           int32_t* __restrict__ origin_stamp,
                                                           const int32_t* __restrict__ n_dev,
                                                           int32_t* __restrict__ ctl_clear) {
  if (n_dev) n = (int64_t)*n_dev < n ? (int64_t)*n_dev : n;  // count from device memory; n = grid capacity
  // origins stamped by the frame's upsert (bnv_volume_integrate_frame): no k_lattice_stamp launch in front of this
  // one, so the control words of the stages behind are cleared here
  if (ctl_clear && blockIdx.x == 0 && threadIdx.x == 0) ctl_clear[1] = ctl_clear[2] = ctl_clear[3] = 0;
  // grid-stride: the launch is sized for the CAPACITY (the count is on the device) but capped, so a frame that holds
  // a fraction of it (a shard's 1 / world) does not pay for ten thousand workgroups that only exit
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n * 27; t += (int64_t)gridDim.x * 256) {
    const int64_t b = t / 27;
    const int nb = (int)(t - b * 27);
    // (from the dense row index when the volume keeps one: the 27 look-ups of a voxel are 9 runs of 3 neighbouring
    // words, and neighbouring voxels share them -- against 27 hash probes that each pull a line of their own)
    const int r = lattice_neighbor_row(v, origins, b, nb, weights, row_limit, min_pts, origin_stamp, epoch);
    nbr_rows[t] = r;
    // list = rows whose table must be (re)computed here; halo rows (row_skip) get theirs by exchange
    const int row = r & ~(1 << 30);
    if (r >= 0 && list && !(row_skip && row_skip[row]) && atomicExch(&stamp[row], epoch) != epoch)
      list[atomicAdd(n_list, 1)] = row;
  }
}

// One thread per lattice point P = b + d / 2 (origin b, offset d): if all 8 corner voxels are usable (the point is
// LIVE), the (row, l) table entries it reads -- one per DISTINCT corner voxel c, l = the offset of P inside c -- go
// to the MLP work list, each exactly once.  Entries of masked points are never evaluated.
// An entry (row, l) names one physical point, and whether that point is live depends on the point alone.  So
//  * the entry of P in the origin's OWN row is appended by this thread, unconditionally: no other thread appends it;
//  * the entries in other corner rows that are ORIGINS of this call are left to those origins (P is one of their
//    27 points too, and they see the same live decision);
//  * entries in corner rows that are not decoded in this call (the fringe of the frame) belong to the origin
//    floor(P) if that voxel is decoded here; only if it is not are they contended: the first thread to flag
//    (row, l) in need_mask appends it.
// Both decisions are bit tests on two 27-bit masks per origin (usable neighbours, neighbours that are origins),
// cut out of the ballots of the staging loop.  (Until r03 every corner entry of a shared point went through a
// returning global atomicOr, and an owner rule decided which origin handled a shared point: 35 us.)
// A workgroup walks kMarkChunks chunks of 1,024 lattice points and collects the new entries in LDS; they go to the
// global list with ONE atomicAdd on the list counter per flush -- normally one per workgroup.  (Same-address
// atomics serialise in the memory-side atomic unit at ~11 ns each, tools/probe_mark.hip: one per 1,024 points was
// 29 us of serial time per frame; a decoupled look-back in its place was slower still, 57-78 us, because every
// workgroup then ends with two or three dependent memory round trips.)
#ifndef BNV_MARK_THREADS
#define BNV_MARK_THREADS 1024
#endif
#ifndef BNV_MARK_CHUNKS
#define BNV_MARK_CHUNKS 2
#endif
#ifndef BNV_MARK_SCAN
#define BNV_MARK_SCAN 0
#endif
constexpr int kMarkThreads = BNV_MARK_THREADS;
constexpr int kMarkChunks = BNV_MARK_CHUNKS;
constexpr int kMarkOrigins = kMarkThreads / 27 + 2;   // origins a chunk's lattice points can belong to
constexpr int kMarkBuf = kMarkChunks > 1 ? 16 * kMarkThreads : 8 * kMarkThreads;   // LDS entry buffer; a chunk appends at most 8 per thread
// FUSED: the neighbour rows are looked up HERE (and written to nbr_rows for the blend) instead of by a
// k_lattice_neighbors launch in front: one launch and one 10 MB round trip less per frame.  Needs the origin stamps
// of the call to be complete (k_lattice_stamp or the frame's upsert) and the control words cleared.
struct MarkFused {
  bnv_volume_t v;
  const int64_t* origins;
  const float* weights;
  int64_t row_limit;
  float min_pts;
  int32_t* nbr_rows_out;
  // Persistent tables (bnv_volume_t.lattice_have; null: none): bit l of have[row] = the entry (row, l) is in the
  // persistent table for the row's current features.  Entries in rows this call does not decode are listed only when
  // their bit is clear (and the bit is set: the table kernel behind fills them); the entries of the call's own rows
  // -- always listed, the upsert has just changed the rows -- set their bits for later frames.
  uint32_t* have;
};

template <bool FUSED>
__global__ __launch_bounds__(kMarkThreads) void k_lattice_mark(const int32_t* __restrict__ nbr_rows, int64_t n,
                                                               const int32_t* __restrict__ origin_stamp, int32_t epoch,
                                                               uint32_t* __restrict__ need_mask,
                                                               int32_t* __restrict__ entries,
                                                               int32_t* __restrict__ n_entries,
                                                               int64_t entry_capacity,
                                                               const int32_t* __restrict__ n_dev, MarkFused F) {
  if (n_dev) n = (int64_t)*n_dev < n ? (int64_t)*n_dev : n;
  // chunks per (virtual) workgroup: kMarkChunks -- or ONE when the launch's workgroups then still cover the call (a
  // shard's 1 / world of a frame): twice the workgroups at work, half the dependent chunk passes per workgroup
  const int CH = (n * 27 <= (int64_t)gridDim.x * kMarkThreads) ? 1 : kMarkChunks;
  if ((int64_t)blockIdx.x * kMarkThreads * CH >= n * 27) return;
  // (grid-stride over virtual workgroups vb: the launch is sized for the capacity, capped at two workgroups per CU)
  __shared__ int s_buf[kMarkBuf];
  __shared__ int s_nbr[kMarkOrigins * 27];
  __shared__ int s_corner[216 + 27];
  __shared__ uint32_t s_need[27];                              // the neighbours a lattice point's corners are
  constexpr int kMarkWords = (kMarkOrigins * 27 + 63) / 64 + 1;
  __shared__ unsigned long long s_ub[kMarkWords], s_ob[kMarkWords];   // bit i: s_nbr[i] usable / an origin of this call
#if BNV_MARK_SCAN
  __shared__ uint32_t s_wave[kMarkThreads / 64];
#endif
  __shared__ int s_count, s_base;
  __shared__ uint32_t s_have[kMarkOrigins];   // persistent tables: live-point bits of the chunk's origins
  if (threadIdx.x < 216) {
    const int p = threadIdx.x >> 3, k = threadIdx.x & 7;
    const int d[3] = {p / 9 - 1, (p / 3) % 3 - 1, p % 3 - 1};
    int nbi = 0, li = 0, dup = 0;   // ceil == floor on an axis with d == 0: same entry as the floor corner
    for (int a = 0; a < 3; ++a) {
      int nb_a = 0, loc2 = 0;
      if (d[a] != 0) {
        if ((k >> a) & 1) {
          nb_a = (d[a] + 1) / 2;
          loc2 = -1;
        } else {
          nb_a = (d[a] - 1) / 2;
          loc2 = 1;
        }
      } else if ((k >> a) & 1) {
        dup = 1;
      }
      nbi = nbi * 3 + (nb_a + 1);
      li = li * 3 + (loc2 + 1);
    }
    s_corner[threadIdx.x] = nbi | (li << 5) | (dup << 10);
  } else if (threadIdx.x < 216 + 27) {
    // neighbour index of the voxel floor(P) if some offset of P is negative, else -1 (the origin itself)
    const int p = threadIdx.x - 216;
    const int d[3] = {p / 9 - 1, (p / 3) % 3 - 1, p % 3 - 1};
    s_corner[threadIdx.x] = (d[0] < 0 || d[1] < 0 || d[2] < 0)
                                ? ((d[0] < 0 ? 0 : 1) * 3 + (d[1] < 0 ? 0 : 1)) * 3 + (d[2] < 0 ? 0 : 1)
                                : -1;
  }
  if (threadIdx.x < kMarkWords) s_ub[threadIdx.x] = s_ob[threadIdx.x] = 0ull;
  __syncthreads();
  if (threadIdx.x < 27) {
    uint32_t m = 0;
    for (int k = 0; k < 8; ++k) m |= 1u << (s_corner[threadIdx.x * 8 + k] & 31);
    s_need[threadIdx.x] = m;
  }
  for (int64_t vb = blockIdx.x; vb * kMarkThreads * CH < n * 27; vb += gridDim.x) {
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  for (int ch = 0; ch < CH; ++ch) {
    const int64_t t0 = (vb * CH + ch) * kMarkThreads;
    const bool last = ch == CH - 1 || t0 + kMarkThreads >= n * 27;
    // the neighbour rows of the chunk's origins: one coalesced read, then LDS
    const int64_t b0 = t0 / 27;
    for (int i = threadIdx.x; i < kMarkOrigins * 27; i += kMarkThreads) {
      const int64_t g = b0 * 27 + i;
      int r = -1;
      if (g < n * 27) {
        if constexpr (FUSED) {
          const int ob = i / 27;
          r = lattice_neighbor_row(F.v, F.origins, b0 + ob, i - ob * 27, F.weights, F.row_limit, F.min_pts, origin_stamp,
                                   epoch);
          F.nbr_rows_out[g] = r;   // (a chunk boundary inside an origin: both chunks write the same values)
        } else {
          r = nbr_rows[g];
        }
      }
      s_nbr[i] = r;
      if (i < kMarkOrigins) s_have[i] = 0u;
      const unsigned long long bu = __ballot(r >= 0), bo = __ballot(r >= 0 && (r & kOriginBit));
      if ((threadIdx.x & 63) == 0) {
        s_ub[i >> 6] = bu;
        s_ob[i >> 6] = bo;
      }
    }
    __syncthreads();
    const int64_t t = t0 + threadIdx.x;
    int ent[8];
    uint32_t keep = 0;    // bit k: ent[k] is appended by this thread
    if (t < n * 27) {
      const int64_t b = t / 27;
      const int p = (int)(t - b * 27);
      const int ob = (int)(b - b0);
      const int* nb27 = s_nbr + ob * 27;
      const int q = ob * 27, w = q >> 6, sh = q & 63;
      unsigned long long xu = s_ub[w] >> sh, xo = s_ob[w] >> sh;
      if (sh > 64 - 27) {
        xu |= s_ub[w + 1] << (64 - sh);
        xo |= s_ob[w + 1] << (64 - sh);
      }
      const uint32_t um = (uint32_t)xu & 0x7FFFFFFu, om = (uint32_t)xo & 0x7FFFFFFu, need = s_need[p];
      if ((um & need) == need) {     // live
        uint32_t rest = need & ~om;  // corner voxels nobody decodes in this call
        if (!((rest >> 13) & 1u)) {  // the origin's own row (always, but for a caller's stale stamp array)
          ent[0] = ((nb27[13] & ~kOriginBit) << 5) | p;     // P inside its origin: l = d
          keep = 1u;
          if (F.have) atomicOr(&s_have[ob], 1u << p);
        }
        // Entries in rows that are not decoded here belong to the origin floor(P) when that voxel is decoded in
        // this call (it is unique: no flag needed); else every origin that holds P asks need_mask
        const int dneg = s_corner[216 + p];
        const bool mine = dneg < 0;
        if (rest && (mine || !((om >> dneg) & 1u))) {
          // (rare) all atomics are issued before any result is looked at: one memory round trip, not up to eight
          int rowk[8], lk[8];
          uint32_t seen[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int c = s_corner[p * 8 + k];     // nbi | li << 5 | dup << 10
            rowk[k] = (!(c >> 10) && ((rest >> (c & 31)) & 1u)) ? (nb27[c & 31] & ~kOriginBit) : -1;
            lk[k] = (c >> 5) & 31;
          }
          if (F.have) {   // persistent tables: the bit outlives the call (whoever finds it clear lists the entry)
#pragma unroll
            for (int k = 0; k < 8; ++k) seen[k] = rowk[k] >= 0 ? atomicOr(&F.have[rowk[k]], 1u << lk[k]) : 0u;
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
              seen[k] = (!mine && rowk[k] >= 0) ? atomicOr(&need_mask[rowk[k]], 1u << lk[k]) : 0u;
          }
          int at = (int)keep;
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (rowk[k] >= 0 && !((seen[k] >> lk[k]) & 1u)) {
              // (at most 8 distinct corners, the own row among them: at < 8)
              ent[at & 7] = (rowk[k] << 5) | lk[k];
              keep |= 1u << (at & 7);
              ++at;
            }
        }
      }
    }
#if BNV_MARK_SCAN
    // the threads' places in the LDS buffer: one block-wide scan of the counts (no LDS atomics)
    uint32_t tot;
    const uint32_t off = block_exclusive_scan<kMarkThreads>((uint32_t)__popc(keep), s_wave, &tot);
    const int at = s_count + (int)off;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if ((keep >> k) & 1u) s_buf[at + __popc(keep & ((1u << k) - 1u))] = ent[k];
    __syncthreads();
    if (threadIdx.x == 0) s_count += (int)tot;
    __syncthreads();
#else
    {
      // ent[0] (nearly every live point has exactly this one): one LDS atomic per wave, places by ballot
      const unsigned long long bal = __ballot(keep & 1u);
      int base0 = 0;
      if ((threadIdx.x & 63) == 0 && bal) base0 = atomicAdd(&s_count, __popcll(bal));
      base0 = __builtin_amdgcn_readfirstlane(base0);
      if (keep & 1u) s_buf[base0 + __popcll(bal & ((1ull << (threadIdx.x & 63)) - 1ull))] = ent[0];
      const uint32_t extra = keep >> 1;      // (rare) entries in rows that are not decoded in this call
      if (extra) {
        const int at = atomicAdd(&s_count, __popc(extra));
#pragma unroll
        for (int k = 1; k < 8; ++k)
          if ((extra >> (k - 1)) & 1u) s_buf[at + __popc(extra & ((1u << (k - 1)) - 1u))] = ent[k];
      }
    }
    __syncthreads();
#endif
    if (F.have) {   // (kernel-uniform) the own-row bits of the chunk's origins join the persistent masks
      if (threadIdx.x < kMarkOrigins && s_have[threadIdx.x]) {
        const int r13 = s_nbr[threadIdx.x * 27 + 13];
        if (r13 >= 0) atomicOr(&F.have[r13 & ~kOriginBit], s_have[threadIdx.x]);
      }
      __syncthreads();   // s_nbr / s_have are rewritten by the next chunk
    }
    const int cnt = s_count;
    if (cnt > 0 && (last || cnt > kMarkBuf - 8 * kMarkThreads)) {   // flush (block-uniform)
      if (threadIdx.x == 0) s_base = atomicAdd(n_entries, cnt);
      __syncthreads();
      // (only now has every wave read s_count above: resetting it next to the atomicAdd let a late wave see 0, skip
      // the flush and fall out of step with the workgroup's barriers)
      if (threadIdx.x == 0) s_count = 0;
      const int base = s_base;
      for (int i = threadIdx.x; i < cnt; i += kMarkThreads)
        if (base + i < entry_capacity) entries[base + i] = s_buf[i];
      __syncthreads();
    }
    if (last) break;
  }
  __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------
// k_lattice_mark_o (round 5): the same marking with ONE THREAD PER ORIGIN instead of one per lattice point.
// What k_lattice_mark spends its time on is not arithmetic (a few bit tests per point) but the per-chunk chain of
// barriers, LDS appends and flushes over 2.7 M threads, and -- fused with the neighbour look-up, on a shard -- ONE
// dependent three-load chain per thread.  Here a thread holds its origin's two 27-bit masks (usable neighbours,
// neighbours that are origins of the call) in registers and derives the live mask of its 27 lattice points with 27 bit
// tests; the own-row entries of a workgroup's 256 origins are placed by one block scan and leave through LDS as one
// coalesced copy (<= 27 per origin: the staging area of the neighbour rows is exactly large enough), the fringe
// entries (rows that are not decoded in this call: a few per cent) through a small LDS buffer; one global atomic per
// workgroup; fused, every thread has 27 independent look-up chains in flight.  Same entries as k_lattice_mark (in
// another order, which nothing depends on), same need_mask / lattice_have bookkeeping.
// ---------------------------------------------------------------------------------------------------
constexpr int kMoThreads = 256;                 // origins per workgroup
constexpr int kMoExtra = 3072;                  // LDS room for fringe entries of a workgroup (beyond it: direct appends)
constexpr int kMoWork = 2048;                   // LDS list of a workgroup's lattice points that have fringe corners
template <bool FUSED>
__global__ __launch_bounds__(kMoThreads) void k_lattice_mark_o(const int32_t* __restrict__ nbr_rows, int64_t n,
                                                               const int32_t* __restrict__ origin_stamp, int32_t epoch,
                                                               uint32_t* __restrict__ need_mask,
                                                               int32_t* __restrict__ entries,
                                                               int32_t* __restrict__ n_entries,
                                                               int64_t entry_capacity,
                                                               const int32_t* __restrict__ n_dev, MarkFused F) {
  if (n_dev) n = (int64_t)*n_dev < n ? (int64_t)*n_dev : n;
  if ((int64_t)blockIdx.x * kMoThreads >= n) return;
  __shared__ int s_nbr[kMoThreads * 27];        // neighbour rows of the workgroup's origins; then its own-row entries
  __shared__ int s_extra[kMoExtra];
  __shared__ int s_corner[216 + 27];
  __shared__ uint32_t s_need[27];
  __shared__ uint32_t s_wave[kMoThreads / 64];
  __shared__ uint32_t s_om[kMoThreads];
  __shared__ int s_work[kMoWork];
  __shared__ int s_nx, s_nw, s_base;
  if (threadIdx.x < 216) {
    const int p = threadIdx.x >> 3, k = threadIdx.x & 7;
    const int d[3] = {p / 9 - 1, (p / 3) % 3 - 1, p % 3 - 1};
    int nbi = 0, li = 0, dup = 0;   // ceil == floor on an axis with d == 0: same entry as the floor corner
    for (int a = 0; a < 3; ++a) {
      int nb_a = 0, loc2 = 0;
      if (d[a] != 0) {
        if ((k >> a) & 1) {
          nb_a = (d[a] + 1) / 2;
          loc2 = -1;
        } else {
          nb_a = (d[a] - 1) / 2;
          loc2 = 1;
        }
      } else if ((k >> a) & 1) {
        dup = 1;
      }
      nbi = nbi * 3 + (nb_a + 1);
      li = li * 3 + (loc2 + 1);
    }
    s_corner[threadIdx.x] = nbi | (li << 5) | (dup << 10);
  } else if (threadIdx.x < 216 + 27) {
    const int p = threadIdx.x - 216;
    const int d[3] = {p / 9 - 1, (p / 3) % 3 - 1, p % 3 - 1};
    s_corner[threadIdx.x] = (d[0] < 0 || d[1] < 0 || d[2] < 0)
                                ? ((d[0] < 0 ? 0 : 1) * 3 + (d[1] < 0 ? 0 : 1)) * 3 + (d[2] < 0 ? 0 : 1)
                                : -1;
  }
  __syncthreads();
  if (threadIdx.x < 27) {
    uint32_t m = 0;
    for (int k = 0; k < 8; ++k) m |= 1u << (s_corner[threadIdx.x * 8 + k] & 31);
    s_need[threadIdx.x] = m;
  }
  // the fringe entries of ONE live lattice point p of the origin whose neighbour rows are nb27 and origin mask om: the
  // corner rows that are not decoded in this call, each listed by whoever finds its bit clear.  All atomics of the
  // point are issued before any result is looked at (one memory round trip)
  auto fringe_point = [&](const int* nb27, uint32_t om, int p) {
    const uint32_t rest = s_need[p] & ~om;
    const bool mine = s_corner[216 + p] < 0;
    int rowk[8], lk[8];
    uint32_t seen[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = s_corner[p * 8 + k];     // nbi | li << 5 | dup << 10
      rowk[k] = (!(c >> 10) && ((rest >> (c & 31)) & 1u)) ? (nb27[c & 31] & ~kOriginBit) : -1;
      lk[k] = (c >> 5) & 31;
    }
    if (F.have) {   // persistent tables: the bit outlives the call
#pragma unroll
      for (int k = 0; k < 8; ++k) seen[k] = rowk[k] >= 0 ? atomicOr(&F.have[rowk[k]], 1u << lk[k]) : 0u;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        seen[k] = (!mine && rowk[k] >= 0) ? atomicOr(&need_mask[rowk[k]], 1u << lk[k]) : 0u;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (rowk[k] < 0 || ((seen[k] >> lk[k]) & 1u)) continue;
      const int e = (rowk[k] << 5) | lk[k];
      const int at = atomicAdd(&s_nx, 1);
      if (at < kMoExtra) {
        s_extra[at] = e;
      } else {   // (rare overflow of the LDS buffer: straight to the list)
        const int g = atomicAdd(n_entries, 1);
        if (g < entry_capacity) entries[g] = e;
      }
    }
  };
  for (int64_t vb = blockIdx.x; vb * kMoThreads < n; vb += gridDim.x) {
    const int64_t b0 = vb * kMoThreads;
    if (threadIdx.x == 0) s_nx = s_nw = 0;
    // the neighbour rows of the workgroup's origins
    bool staged = false;
    if constexpr (FUSED) {
      // With the dense row index a thread looks its OWN origin's 27 neighbours up in three rounds of independent loads
      // (27 index words; then 27 weights + 27 origin stamps) instead of 27 three-load chains one behind the other
      // (the generic look-up below branches between the loads, which keeps the compiler from overlapping them).
      if (F.v.brick) {
        staged = true;
        const int64_t bb = b0 + threadIdx.x;
        int rows[27];
        if (bb < n) {
          const int64_t ox = F.origins[bb * 3 + 0], oy = F.origins[bb * 3 + 1], oz = F.origins[bb * 3 + 2];
#pragma unroll
          for (int k = 0; k < 27; ++k) {
            const int64_t x = ox + (k / 9 - 1), y = oy + ((k / 3) % 3 - 1), z = oz + (k % 3 - 1);
            int64_t idx;
            rows[k] = brick_index(F.v, x, y, z, &idx) ? F.v.brick[idx] : -2;   // -2: outside the index (the hash decides)
          }
#pragma unroll
          for (int k = 0; k < 27; ++k) {
            if (rows[k] == -2)
              rows[k] = volume_row(F.v, ox + (k / 9 - 1), oy + ((k / 3) % 3 - 1), oz + (k % 3 - 1));
            if (rows[k] >= F.row_limit) rows[k] = -1;
          }
          float wk[27];
          int sk[27];
#pragma unroll
          for (int k = 0; k < 27; ++k) {
            const int rr = rows[k] < 0 ? 0 : rows[k];
            wk[k] = F.weights[rr];
            sk[k] = origin_stamp ? origin_stamp[rr] : 0;
          }
#pragma unroll
          for (int k = 0; k < 27; ++k) {
            int r = -1;
            if (rows[k] >= 0 && wk[k] >= F.min_pts) r = rows[k] | ((origin_stamp && sk[k] == epoch) ? kOriginBit : 0);
            s_nbr[threadIdx.x * 27 + k] = r;
          }
        } else {
#pragma unroll
          for (int k = 0; k < 27; ++k) s_nbr[threadIdx.x * 27 + k] = -1;
        }
      }
    }
#pragma unroll 9
    for (int i = threadIdx.x; i < (staged ? 0 : kMoThreads * 27); i += kMoThreads) {
      const int64_t g = b0 * 27 + i;
      int r = -1;
      if (g < n * 27) {
        if constexpr (FUSED) {
          // (no global store in this loop: a store the compiler cannot prove disjoint from the volume's arrays would
          // order the iterations' look-up chains one behind the other -- 81 dependent loads instead of 3)
          const int ob = i / 27;
          r = lattice_neighbor_row(F.v, F.origins, b0 + ob, i - ob * 27, F.weights, F.row_limit, F.min_pts, origin_stamp,
                                   epoch);
        } else {
          r = nbr_rows[g];
        }
      }
      s_nbr[i] = r;
    }
    __syncthreads();
    if constexpr (FUSED) {   // the blend reads the neighbour rows from global memory
      for (int i = threadIdx.x; i < kMoThreads * 27; i += kMoThreads)
        if (b0 * 27 + i < n * 27) F.nbr_rows_out[b0 * 27 + i] = s_nbr[i];
    }
    const int64_t b = b0 + threadIdx.x;
    const int* nb27 = s_nbr + threadIdx.x * 27;      // (stride 27 words: conflict-free across the lanes of a wave)
    uint32_t um = 0, om = 0;
    if (b < n) {
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        const int r = nb27[k];
        um |= (r >= 0 ? 1u : 0u) << k;
        om |= ((r >= 0 && (r & kOriginBit)) ? 1u : 0u) << k;
      }
    }
    const int own_row = (um >> 13) & 1u ? (nb27[13] & ~kOriginBit) : -1;
    uint32_t live = 0;
#pragma unroll
    for (int p = 0; p < 27; ++p) live |= ((um & s_need[p]) == s_need[p] ? 1u : 0u) << p;
    if (b >= n) live = 0;
    // the points whose entry in the origin's OWN row this thread lists (always, but for a caller's stale stamp array)
    const uint32_t own = ((om >> 13) & 1u) ? live : 0u;
    if (own && F.have) atomicOr(&F.have[own_row], own);   // (the upsert cleared the word; nobody else sets bits of an origin's row)
    // fringe entries: corner rows that are not decoded in this call.  A thread only LISTS its points that have such
    // corners (origin << 5 | p); the whole workgroup then works the list off, one point per thread and step, the (up
    // to eight) returning atomics of a point in flight together -- an origin on the fringe has dozens of them, and
    // one thread taking them one round trip after the other held its workgroup for tens of microseconds
    s_om[threadIdx.x] = om;
    if (live) {
      for (int p = 0; p < 27; ++p) {
        if (!((live >> p) & 1u) || !(s_need[p] & ~om)) continue;
        const int dneg = s_corner[216 + p];
        if (dneg >= 0 && ((om >> dneg) & 1u)) continue;      // the origin floor(P) is decoded here: it lists them
        const int at = atomicAdd(&s_nw, 1);
        if (at < kMoWork) s_work[at] = (int)(threadIdx.x << 5) | p;
        else fringe_point(nb27, om, p);                       // (a call whose fringe dwarfs its origins: inline)
      }
    }
    __syncthreads();
    {
      const int nw = s_nw < kMoWork ? s_nw : kMoWork;
      for (int i = threadIdx.x; i < nw; i += kMoThreads) {
        const int wi = s_work[i];
        fringe_point(s_nbr + (wi >> 5) * 27, s_om[wi >> 5], wi & 31);
      }
    }
    uint32_t tot;
    const uint32_t off = block_exclusive_scan<kMoThreads>((uint32_t)__popc(own), s_wave, &tot);   // (two barriers: s_nbr is read out)
    {
      int at = (int)off;
      uint32_t m = own;
      while (m) {
        const int p = __ffs(m) - 1;
        m &= m - 1;
        s_nbr[at++] = (own_row << 5) | p;
      }
    }
    __syncthreads();
    const int nx = s_nx < kMoExtra ? s_nx : kMoExtra;
    if (threadIdx.x == 0) s_base = (tot + nx) ? atomicAdd(n_entries, (int)tot + nx) : 0;
    __syncthreads();
    const int base = s_base;
    for (int i = threadIdx.x; i < (int)tot; i += kMoThreads)
      if (base + i < entry_capacity) entries[base + i] = s_nbr[i];
    for (int i = threadIdx.x; i < nx; i += kMoThreads)
      if (base + (int)tot + i < entry_capacity) entries[base + (int)tot + i] = s_extra[i];
    __syncthreads();   // s_nbr / s_extra / s_nx are rewritten by the next round
  }
}

// DELTA = false: the streaming case (no TSDF prior): 8 table reads and a weighted sum, few registers -- it runs
// beside the persistent MLP kernels of the other streams.
#ifndef BNV_BLEND_PPT
#define BNV_BLEND_PPT 3
#endif
constexpr int kBlendPpt = BNV_BLEND_PPT;     // lattice points per thread of the streaming blend: their gathers are in flight together
template <bool DELTA>
__global__ __launch_bounds__(256) void k_lattice_blend(const int32_t* __restrict__ nbr_rows, int64_t n,
                                                       const float* __restrict__ table, bnv_grid_t g,
                                                       const int64_t* __restrict__ origins, bnv_sdf_delta_t delta,
                                                       float* __restrict__ out, const int32_t* __restrict__ n_dev) {
  if (n_dev) n = (int64_t)*n_dev < n ? (int64_t)*n_dev : n;
  constexpr int PPT = DELTA ? 1 : kBlendPpt;
  constexpr int TILE = 256 * PPT;
  // the neighbour rows of the block's origins: one coalesced read, then 8 LDS reads per lattice point
  __shared__ int s_nbr[(TILE / 27 + 2) * 27];
  // grid-stride over virtual workgroups vb (the launch is sized for the capacity, capped at 8 workgroups per CU: a
  // frame that holds a fraction of it does not pay for tens of thousands of workgroups that only exit)
  for (int64_t vb = blockIdx.x; vb * TILE < n * 27; vb += gridDim.x) {
  if (vb != (int64_t)blockIdx.x) __syncthreads();   // s_nbr of the previous round is no longer read
  const int64_t b0 = (vb * TILE) / 27;
  for (int i = threadIdx.x; i < (TILE / 27 + 2) * 27; i += 256) {
    const int64_t gidx = b0 * 27 + i;
    s_nbr[i] = gidx < n * 27 ? nbr_rows[gidx] : -1;
  }
  __syncthreads();
#pragma unroll
  for (int rep = 0; rep < PPT; ++rep) {
  const int64_t t = vb * TILE + rep * 256 + threadIdx.x;
  [&]() {
  if (t >= n * 27) return;
  const int64_t b = t / 27;
  const int p = (int)(t - b * 27);
  const int* nb27 = s_nbr + (int)(b - b0) * 27;
  const int d[3] = {p / 9 - 1, (p / 3) % 3 - 1, p % 3 - 1};  // lattice point = origin + 0.5 * d
  if constexpr (!DELTA) {
    // Every corner has the same weight 0.5^m (m = axes with a half-voxel offset) and the reference's normaliser, the
    // sequential sum of the 8 weights, is exactly 8 * 0.5^m: one pass, nothing kept in arrays; fully unrolled so
    // that the 8 gathers are in flight together (a partially unrolled 20-VGPR version took 43 us instead of 33).
    const int m = (d[0] != 0) + (d[1] != 0) + (d[2] != 0);
    const float wc = m == 0 ? 1.f : (m == 1 ? 0.5f : (m == 2 ? 0.25f : 0.125f));
    const float w = __fdiv_rn(wc, 8.f * wc);
    bool ok = true;
    int rowk[8], lk[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int cb = kCornerCeilBits[k];
      int nbi = 0, li = 0;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        int nb_a = 0, loc2 = 0;
        if (d[a] != 0) {
          if ((cb >> a) & 1) {
            nb_a = (d[a] + 1) / 2;
            loc2 = -1;
          } else {
            nb_a = (d[a] - 1) / 2;
            loc2 = 1;
          }
        }
        nbi = nbi * 3 + (nb_a + 1);
        li = li * 3 + (loc2 + 1);
      }
      rowk[k] = nb27[nbi];
      lk[k] = li;
      if (rowk[k] < 0) ok = false;
      rowk[k] &= ~kOriginBit;
    }
    if (!ok) {   // masked point (about half of them on a thin sheet): the constant, no table reads
      out[t] = g.voxel_size;
      return;
    }
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc = __fadd_rn(acc, __fmul_rn(table[(size_t)rowk[k] * 27 + lk[k]], w));
    out[t] = acc;
    return;
  }
  float wk[8];
  int rowk[8], lk[8];
  float ck[8][3];
  float norm = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int cb = kCornerCeilBits[k];
    int nbi = 0, li = 0;
    float w = 1.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      int nb_a = 0, loc2 = 0;  // neighbour offset of the corner voxel, 2 * local coordinate
      if (d[a] != 0) {
        if ((cb >> a) & 1) {
          nb_a = (d[a] + 1) / 2;
          loc2 = -1;
        } else {
          nb_a = (d[a] - 1) / 2;
          loc2 = 1;
        }
        w = __fmul_rn(w, 0.5f);
      }
      nbi = nbi * 3 + (nb_a + 1);
      li = li * 3 + (loc2 + 1);
      if (DELTA) ck[k][a] = (float)(origins[b * 3 + a] + nb_a);
    }
    wk[k] = w;
    lk[k] = li;
    rowk[k] = nb27[nbi] < 0 ? -1 : (nb27[nbi] & ~kOriginBit);
    norm = __fadd_rn(norm, w);
  }
  bool ok = true;
  float acc = 0.f, dacc = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float w = __fdiv_rn(wk[k], norm);
    if (rowk[k] < 0) ok = false;
    const float a = rowk[k] >= 0 ? table[(size_t)rowk[k] * 27 + lk[k]] : 0.f;
    acc = __fadd_rn(acc, __fmul_rn(a, w));
    if (DELTA) dacc = __fadd_rn(dacc, __fmul_rn(sample_delta(delta, g, ck[k]), w));
  }
  float o = ok ? acc : g.voxel_size;
  if (DELTA) o = __fadd_rn(o, dacc);
  out[t] = o;
  }();
  }
  }
}

// grid of a grid-stride kernel: the blocks the work needs, at most `per_cu` per CU
static unsigned capped_grid(int64_t blocks, int per_cu) {
  const int64_t cap = (int64_t)(g_num_cus > 0 ? g_num_cus : 256) * per_cu;
  return (unsigned)(blocks < 1 ? 1 : (blocks < cap ? blocks : cap));
}

std::atomic<int> g_fused_mark{-1};  // bnv_set_option("fused_mark"): 1 / 0 force, -1 (default): by the call's size
std::atomic<int> g_mark_per_origin{1};   // bnv_set_option("mark_per_origin"): 1 = k_lattice_mark_o, 0 = k_lattice_mark (one thread per lattice point)
std::atomic<int> g_half_tail{1};    // bnv_set_option("half_tail"): k_lattice_table_x hands its last partial round out as half tiles
std::atomic<int> g_lattice_pipe{1}; // 1: k_lattice_table_x (cross-tile / cross-layer pipelined, 16x16x32 MFMA); 0: k_decode<LATTICE, 1>

#ifdef BNV_PHASE_PROF
constexpr int kProfLds = 2048;
#else
constexpr int kProfLds = 0;
#endif

// `mlp`: the arithmetic mode of the call (mlp_mode_of(grid.mlp_mode))
static int launch_decode(int mode, int mlp, const DecodeArgs& args, int64_t n_tiles_hint, hipStream_t stream,
                         int max_workgroups = 0) {
  int64_t grid = g_num_cus - g_reserve_cus.load(std::memory_order_relaxed);
  if (max_workgroups > 0 && grid > max_workgroups) grid = max_workgroups;   // (persistent kernels, dynamic tile hand-out)
  if (n_tiles_hint < grid) grid = n_tiles_hint;
  if (grid < 1) grid = 1;
  const bool lattice_pipe = g_lattice_pipe.load(std::memory_order_relaxed) != 0;
  if (mode == MODE_LATTICE && (mlp == 1 || mlp == 3) && lattice_pipe) {
    ProfScope prof(PROF_DECODE_LATTICE, stream);
    DecodeArgs ax = args;
    ax.half_tail = g_half_tail.load(std::memory_order_relaxed);
    if (mlp == 1)
      hipLaunchKernelGGL(k_lattice_table_x<3>, dim3((unsigned)grid), dim3(512), T_TOTAL * 4 + kProfLds, stream, ax);
    else
      hipLaunchKernelGGL(k_lattice_table_x<1>, dim3((unsigned)grid), dim3(512), T_TOTAL * 4 + kProfLds, stream, ax);
    BNV_LAUNCH_CHECK();
    return BNV_OK;
  }
  if (mode == MODE_LATTICE && mlp == 2 && lattice_pipe) {
    ProfScope prof(PROF_DECODE_LATTICE, stream);
    int64_t gt = (int64_t)g_num_cus * 4;                    // four 4-wave workgroups per CU, grid-stride over the tiles
    const int64_t wgs = (n_tiles_hint * (DM / 32) + 3) / 4;   // (the hint counts 128-evaluation tiles)
    if (wgs < gt) gt = wgs;
    if (gt < 1) gt = 1;
    hipLaunchKernelGGL(k_lattice_table_t, dim3((unsigned)gt), dim3(256), 0, stream, args);
    BNV_LAUNCH_CHECK();
    return BNV_OK;
  }
  if (mode == MODE_PTS) {
    int64_t gp = g_num_cus;
    const int64_t chunks = (args.n + PC_Q - 1) / PC_Q;
    if (chunks < gp) gp = chunks;
    if (gp < 1) gp = 1;
    ProfScope prof(PROF_DECODE_PTS, stream);
#define BNV_LAUNCH_PTS(P) \
  hipLaunchKernelGGL((k_decode_pts<P>), dim3((unsigned)gp), dim3(512), C_TOTAL * 4, stream, args)
    if (mlp == 2) BNV_LAUNCH_PTS(2);
    else if (mlp == 1) BNV_LAUNCH_PTS(1);
    else if (mlp == 3) BNV_LAUNCH_PTS(3);
    else BNV_LAUNCH_PTS(0);
#undef BNV_LAUNCH_PTS
    BNV_LAUNCH_CHECK();
    return BNV_OK;
  }
  ProfScope prof(mode == MODE_PTS ? PROF_DECODE_PTS : mode == MODE_LATTICE ? PROF_DECODE_LATTICE : PROF_DECODE_DENSE,
                 stream);
#define BNV_LAUNCH_DECODE(M, P) \
  hipLaunchKernelGGL((k_decode<M, P>), dim3((unsigned)grid), dim3(512), L_TOTAL * 4 + kProfLds, stream, args)
  if (mlp == 2) {
    if (mode == MODE_LATTICE) BNV_LAUNCH_DECODE(MODE_LATTICE, 2);
    else if (mode == MODE_DENSE1) BNV_LAUNCH_DECODE(MODE_DENSE1, 2);
    else BNV_LAUNCH_DECODE(MODE_DENSE, 2);
  } else if (mlp == 1) {
    if (mode == MODE_LATTICE) BNV_LAUNCH_DECODE(MODE_LATTICE, 1);
    else if (mode == MODE_DENSE1) BNV_LAUNCH_DECODE(MODE_DENSE1, 1);
    else BNV_LAUNCH_DECODE(MODE_DENSE, 1);
  } else if (mlp == 3) {
    if (mode == MODE_LATTICE) BNV_LAUNCH_DECODE(MODE_LATTICE, 3);
    else if (mode == MODE_DENSE1) BNV_LAUNCH_DECODE(MODE_DENSE1, 3);
    else BNV_LAUNCH_DECODE(MODE_DENSE, 3);
  } else {
    if (mode == MODE_LATTICE) BNV_LAUNCH_DECODE(MODE_LATTICE, 0);
    else if (mode == MODE_DENSE1) BNV_LAUNCH_DECODE(MODE_DENSE1, 0);
    else BNV_LAUNCH_DECODE(MODE_DENSE, 0);
  }
#undef BNV_LAUNCH_DECODE
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

static bool vol_ok_ro(const bnv_volume_t* v) {
  return v && v->slot_keys && v->slot_rows && v->n_slots > 0 && (v->n_slots & (v->n_slots - 1)) == 0 &&
         v->n_feats == 8;
}

}  // namespace bnv

using namespace bnv;

extern "C" {

int bnv_decode_init() {
#define BNV_OPT_IN(M, P) \
  BNV_HIP_CHECK(hipFuncSetAttribute((const void*)k_decode<M, P>, hipFuncAttributeMaxDynamicSharedMemorySize, L_TOTAL * 4 + kProfLds))
  BNV_OPT_IN(MODE_LATTICE, 0);
  BNV_OPT_IN(MODE_DENSE, 0);
  BNV_OPT_IN(MODE_DENSE1, 0);
  BNV_OPT_IN(MODE_LATTICE, 1);
  BNV_OPT_IN(MODE_DENSE, 1);
  BNV_OPT_IN(MODE_DENSE1, 1);
  BNV_OPT_IN(MODE_LATTICE, 2);
  BNV_OPT_IN(MODE_DENSE, 2);
  BNV_OPT_IN(MODE_DENSE1, 2);
  BNV_OPT_IN(MODE_LATTICE, 3);
  BNV_OPT_IN(MODE_DENSE, 3);
  BNV_OPT_IN(MODE_DENSE1, 3);
#undef BNV_OPT_IN
  BNV_HIP_CHECK(hipFuncSetAttribute((const void*)k_lattice_table_x<3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    T_TOTAL * 4 + kProfLds));
  BNV_HIP_CHECK(hipFuncSetAttribute((const void*)k_lattice_table_x<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    T_TOTAL * 4 + kProfLds));
  BNV_HIP_CHECK(hipFuncSetAttribute((const void*)k_decode_pts_bwd, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    C_TOTAL * 4));
  BNV_HIP_CHECK(hipFuncSetAttribute((const void*)k_decode_pts_bwd_t, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    C_TOTAL * 4));
  BNV_HIP_CHECK(hipFuncSetAttribute((const void*)k_optim_step, hipFuncAttributeMaxDynamicSharedMemorySize, C_TOTAL * 4));
  BNV_HIP_CHECK(hipFuncSetAttribute((const void*)k_decode_pts<0>, hipFuncAttributeMaxDynamicSharedMemorySize, C_TOTAL * 4));
  BNV_HIP_CHECK(hipFuncSetAttribute((const void*)k_decode_pts<1>, hipFuncAttributeMaxDynamicSharedMemorySize, C_TOTAL * 4));
  BNV_HIP_CHECK(hipFuncSetAttribute((const void*)k_decode_pts<2>, hipFuncAttributeMaxDynamicSharedMemorySize, C_TOTAL * 4));
  BNV_HIP_CHECK(hipFuncSetAttribute((const void*)k_decode_pts<3>, hipFuncAttributeMaxDynamicSharedMemorySize, C_TOTAL * 4));
  return BNV_OK;
}

size_t bnv_sdfmlp_pack_floats(void) { return SD_PACK_FLOATS_X; }

#ifdef BNV_PHASE_PROF
// development builds only (not in include/bnv_fusion.h): read and reset the phase cycle counters
int bnv_dev_phase_read(unsigned long long* out256) {
  BNV_HIP_CHECK(hipDeviceSynchronize());
  BNV_HIP_CHECK(hipMemcpyFromSymbol(out256, HIP_SYMBOL(g_phase_cycles), 256 * sizeof(unsigned long long)));
  unsigned long long z[256] = {};
  BNV_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z)));
  return BNV_OK;
}
#endif

int bnv_set_option(const char* name, int value) {
  if (!name) return BNV_ERR_INVALID_ARGUMENT;
  if (!strcmp(name, "lattice_pipe")) {
    g_lattice_pipe.store(value, std::memory_order_relaxed);
    return BNV_OK;
  }
  if (!strcmp(name, "mark_per_origin")) {
    g_mark_per_origin.store(value != 0, std::memory_order_relaxed);
    return BNV_OK;
  }
  if (!strcmp(name, "half_tail")) {
    g_half_tail.store(value != 0, std::memory_order_relaxed);
    return BNV_OK;
  }
  if (!strcmp(name, "fused_mark")) {
    g_fused_mark.store(value, std::memory_order_relaxed);
    return BNV_OK;
  }
  if (!strcmp(name, "tcnn_block_encoder")) {
    g_tcnn_block_encoder.store(value != 0, std::memory_order_relaxed);
    return BNV_OK;
  }
  if (!strcmp(name, "finalize_blocks")) {
    if (value < 0) return BNV_ERR_INVALID_ARGUMENT;
    g_finalize_blocks.store(value, std::memory_order_relaxed);
    return BNV_OK;
  }
  if (!strcmp(name, "tcnn_shared_table")) {
    g_tcnn_shared_table.store(value != 0, std::memory_order_relaxed);
    return BNV_OK;
  }
  if (!strcmp(name, "reserve_cus")) {
    if (value < 0 || value >= g_num_cus) return BNV_ERR_INVALID_ARGUMENT;
    g_reserve_cus.store(value, std::memory_order_relaxed);
    return BNV_OK;
  }
  return BNV_ERR_INVALID_ARGUMENT;
}

// split_mask == NULL: one split (plain weights); else split_samples > 0 queries per split, at most 31 splits
static bool splits_ok(const uint32_t* split_mask, int64_t split_samples, int64_t n) {
  if (!split_mask) return true;
  return split_samples > 0 && (n + split_samples - 1) / split_samples <= 31;
}

int bnv_decode_pts_splits(const bnv_volume_t* vol, const bnv_grid_t* grid, const float* features, const float* weights,
                          int64_t row_limit, const float* sdfmlp_pack, const float* coords, int64_t n, int is_coords,
                          const bnv_sdf_delta_t* delta, const uint32_t* split_mask, int64_t split_samples,
                          float* out_sdf, bnv_stream_t stream) {
  if (g_num_cus <= 0) return BNV_ERR_NOT_INITIALISED;
  if (!vol_ok_ro(vol) || !grid || !features || !weights || !sdfmlp_pack || n < 0) return BNV_ERR_INVALID_ARGUMENT;
  if (!mlp_mode_field_ok(grid->mlp_mode) || !splits_ok(split_mask, split_samples, n)) return BNV_ERR_INVALID_ARGUMENT;
  if (n == 0) return BNV_OK;
  if (!coords || !out_sdf) return BNV_ERR_INVALID_ARGUMENT;
  DecodeArgs a = {};
  a.split_mask = split_mask;
  a.split_samples = split_mask ? split_samples : 0;
  a.vol = *vol;
  a.grid = *grid;
  a.features = features;
  a.weights = weights;
  a.row_limit = row_limit;
  a.pack = sdfmlp_pack;
  a.coords = coords;
  a.n = n;
  a.is_coords = is_coords;
  if (delta) a.delta = *delta;
  a.out = out_sdf;
  return launch_decode(MODE_PTS, mlp_mode_of(grid->mlp_mode), a, (n + 15) / 16, (hipStream_t)stream);
}

int bnv_decode_pts(const bnv_volume_t* vol, const bnv_grid_t* grid, const float* features, const float* weights,
                   int64_t row_limit, const float* sdfmlp_pack, const float* coords, int64_t n, int is_coords,
                   const bnv_sdf_delta_t* delta, float* out_sdf, bnv_stream_t stream) {
  return bnv_decode_pts_splits(vol, grid, features, weights, row_limit, sdfmlp_pack, coords, n, is_coords, delta,
                               nullptr, 0, out_sdf, stream);
}

size_t bnv_sdfmlp_bwd_pack_floats(void) { return SB_PACK_FLOATS; }
size_t bnv_sdfmlp_tcnn_bwd_pack_floats(void) { return TB_TOTAL / 2; }

int bnv_decode_pts_backward_splits(const bnv_volume_t* vol, const bnv_grid_t* grid, const float* features,
                                   const float* weights, int64_t row_limit, const float* sdfmlp_pack,
                                   const float* sdfmlp_bwd_pack, const float* coords, int64_t n, int is_coords,
                                   const uint32_t* split_mask, int64_t split_samples, const float* grad_sdf,
                                   float* grad_features, bnv_stream_t stream) {
  if (g_num_cus <= 0) return BNV_ERR_NOT_INITIALISED;
  if (!vol_ok_ro(vol) || !grid || !features || !weights || !sdfmlp_pack || !sdfmlp_bwd_pack || n < 0 ||
      !mlp_mode_field_ok(grid->mlp_mode) || !splits_ok(split_mask, split_samples, n))
    return BNV_ERR_INVALID_ARGUMENT;
  if (n == 0) return BNV_OK;
  if (!coords || !grad_sdf || !grad_features) return BNV_ERR_INVALID_ARGUMENT;
  DecodeBwdArgs b = {};
  b.d.split_mask = split_mask;
  b.d.split_samples = split_mask ? split_samples : 0;
  b.d.vol = *vol;
  b.d.grid = *grid;
  b.d.features = features;
  b.d.weights = weights;
  b.d.row_limit = row_limit;
  b.d.pack = sdfmlp_pack;
  b.d.coords = coords;
  b.d.n = n;
  b.d.is_coords = is_coords;
  b.bwd_pack = sdfmlp_bwd_pack;
  b.grad_out = grad_sdf;
  b.grad_features = grad_features;
  int64_t nblk = (n + PC_Q - 1) / PC_Q;
  if (nblk > g_num_cus) nblk = g_num_cus;
  ProfScope prof(PROF_DECODE_PTS, (hipStream_t)stream);
  if (mlp_mode_of(grid->mlp_mode) == 2)
    hipLaunchKernelGGL(k_decode_pts_bwd_t, dim3((unsigned)nblk), dim3(512), C_TOTAL * 4, (hipStream_t)stream, b);
  else
    hipLaunchKernelGGL(k_decode_pts_bwd, dim3((unsigned)nblk), dim3(512), C_TOTAL * 4, (hipStream_t)stream, b);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

int bnv_decode_pts_backward(const bnv_volume_t* vol, const bnv_grid_t* grid, const float* features,
                            const float* weights, int64_t row_limit, const float* sdfmlp_pack,
                            const float* sdfmlp_bwd_pack, const float* coords, int64_t n, int is_coords,
                            const float* grad_sdf, float* grad_features, bnv_stream_t stream) {
  return bnv_decode_pts_backward_splits(vol, grid, features, weights, row_limit, sdfmlp_pack, sdfmlp_bwd_pack, coords,
                                        n, is_coords, nullptr, 0, grad_sdf, grad_features, stream);
}

int bnv_optim_step(const bnv_volume_t* vol, const bnv_grid_t* grid, const float* features, const float* weights,
                   int64_t row_limit, const float* sdfmlp_pack, const float* sdfmlp_bwd_pack, const float* pts,
                   int64_t n, int is_coords, const bnv_sdf_delta_t* delta, const uint32_t* split_mask,
                   int64_t split_samples, const float* target, const float* sample_weight, const float* n_valid,
                   float* loss_and_counter, float* pred, float* grad_features, bnv_stream_t stream) {
  if (g_num_cus <= 0) return BNV_ERR_NOT_INITIALISED;
  if (!vol_ok_ro(vol) || !grid || !features || !weights || !sdfmlp_pack || !sdfmlp_bwd_pack || n < 0 ||
      !mlp_mode_field_ok(grid->mlp_mode) || !splits_ok(split_mask, split_samples, n))
    return BNV_ERR_INVALID_ARGUMENT;
  // the fused kernel is the split-f16 forward + backward of the fp32 decoder (modes 1 / 3 / 0 share it as
  // bnv_decode_pts_backward does); the tiny-cuda-nn decoder keeps its separate kernels
  if (mlp_mode_of(grid->mlp_mode) == 2) return BNV_ERR_INVALID_ARGUMENT;
  if (n == 0) return BNV_OK;
  if (!pts || !target || !sample_weight || !n_valid || !loss_and_counter || !grad_features) return BNV_ERR_INVALID_ARGUMENT;
  OptimArgs o = {};
  o.b.d.vol = *vol;
  o.b.d.grid = *grid;
  o.b.d.features = features;
  o.b.d.weights = weights;
  o.b.d.row_limit = row_limit;
  o.b.d.pack = sdfmlp_pack;
  o.b.d.coords = pts;
  o.b.d.n = n;
  o.b.d.is_coords = is_coords;
  if (delta) o.b.d.delta = *delta;
  o.b.d.split_mask = split_mask;
  o.b.d.split_samples = split_mask ? split_samples : 0;
  o.b.bwd_pack = sdfmlp_bwd_pack;
  o.b.grad_features = grad_features;
  o.target = target;
  o.wgt = sample_weight;
  o.n_valid = n_valid;
  o.loss = loss_and_counter;
  o.pred = pred;
  int64_t nblk = (n + PC_Q - 1) / PC_Q;
  const int64_t cus = g_num_cus - g_reserve_cus.load(std::memory_order_relaxed);
  if (nblk > cus) nblk = cus;
  ProfScope prof(PROF_DECODE_PTS, (hipStream_t)stream);
  hipLaunchKernelGGL(k_optim_step, dim3((unsigned)nblk), dim3(512), C_TOTAL * 4, (hipStream_t)stream, o);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

int bnv_decode_dense(const float* feat_grid, const float* pts_weight, const int32_t dims[3], float voxel_size,
                     int32_t min_pts_in_grid, const float* sdfmlp_pack, const float* voxel_coords, int64_t n,
                     int32_t variant, float* out_sdf, float* out_feats, int32_t* status, bnv_stream_t stream) {
  return bnv_decode_dense_mode(feat_grid, pts_weight, dims, voxel_size, min_pts_in_grid, sdfmlp_pack, voxel_coords, n,
                               variant, 0, out_sdf, out_feats, status, stream);
}

int bnv_decode_dense_mode(const float* feat_grid, const float* pts_weight, const int32_t dims[3], float voxel_size,
                          int32_t min_pts_in_grid, const float* sdfmlp_pack, const float* voxel_coords, int64_t n,
                          int32_t variant, int32_t mlp_mode, float* out_sdf, float* out_feats, int32_t* status,
                          bnv_stream_t stream) {
  if (g_num_cus <= 0) return BNV_ERR_NOT_INITIALISED;
  if (!feat_grid || !pts_weight || !dims || !sdfmlp_pack || n < 0 || !mlp_mode_field_ok(mlp_mode))
    return BNV_ERR_INVALID_ARGUMENT;
  const int mlp = mlp_mode_of(mlp_mode);
  if (variant < BNV_DENSE_CORNERS || variant > BNV_DENSE_GLOBAL) return BNV_ERR_INVALID_ARGUMENT;
  if (variant == BNV_DENSE_CORNERS && out_feats) return BNV_ERR_INVALID_ARGUMENT;
  if (dims[0] < 2 || dims[1] < 2 || dims[2] < 2) return BNV_ERR_INVALID_ARGUMENT;   // coords / (res - 1)
  if (n == 0) return BNV_OK;
  if (!voxel_coords || !out_sdf) return BNV_ERR_INVALID_ARGUMENT;
  DecodeArgs a = {};
  a.status = status;
  a.nf_out = out_feats;
  a.variant = variant == BNV_DENSE_GLOBAL ? 1 : 0;
  a.grid.voxel_size = voxel_size;
  a.grid.min_pts_in_grid = min_pts_in_grid;
  a.pack = sdfmlp_pack;
  a.coords = voxel_coords;
  a.n = n;
  a.is_coords = 1;
  a.out = out_sdf;
  a.feat_grid = feat_grid;
  a.pts_weight = pts_weight;
  a.dims[0] = dims[0];
  a.dims[1] = dims[1];
  a.dims[2] = dims[2];
  if (variant != BNV_DENSE_CORNERS) return launch_decode(MODE_DENSE1, mlp, a, (n + DM - 1) / DM, (hipStream_t)stream);
  return launch_decode(MODE_DENSE, mlp, a, (n + 15) / 16, (hipStream_t)stream);
}

size_t bnv_decode_lattice_workspace_bytes(int64_t n_voxels, int64_t row_capacity) {
  return lattice_ws_layout(n_voxels, row_capacity, nullptr, nullptr);
}

size_t bnv_decode_lattice_count_offset(int64_t row_capacity) {
  LatticeWs ws;
  lattice_ws_layout(1, row_capacity, (char*)256, &ws);
  return (size_t)((char*)ws.n_list - (char*)256);
}

size_t bnv_decode_lattice_table_offset(int64_t row_capacity) {
  LatticeWs ws;
  lattice_ws_layout(1, row_capacity, (char*)256, &ws);
  return (size_t)((char*)ws.table - (char*)256);
}

size_t bnv_decode_lattice_list_offset(int64_t n_voxels, int64_t row_capacity) {
  LatticeWs ws;
  lattice_ws_layout(n_voxels, row_capacity, (char*)256, &ws);
  return (size_t)((char*)ws.list - (char*)256);
}

static int lattice_neighbors_impl(const bnv_volume_t* vol, const bnv_grid_t* grid, const float* weights,
                                  int64_t row_limit, const int64_t* origins, int64_t n, const int32_t* n_dev,
                                  const uint8_t* row_skip, int build_list, void* ws_ptr, size_t ws_bytes, int32_t epoch,
                                  bool prestamped, bnv_stream_t stream_) {
  if (!vol_ok_ro(vol) || !grid || !weights || n < 0 || epoch == 0) return BNV_ERR_INVALID_ARGUMENT;
  if (!ws_ptr) return BNV_ERR_INVALID_ARGUMENT;
  LatticeWs ws;
  if (lattice_ws_layout(n, vol->row_capacity, (char*)ws_ptr, &ws) > ws_bytes) return BNV_ERR_WORKSPACE_TOO_SMALL;
  hipStream_t stream = (hipStream_t)stream_;
  if (build_list) BNV_HIP_CHECK(hipMemsetAsync(ws.n_list, 0, 16, stream));  // rows listed, (entries), tile counter, spare
  if (n == 0) return BNV_OK;
  if (!origins) return BNV_ERR_INVALID_ARGUMENT;
  if (!prestamped) {
    hipLaunchKernelGGL(k_lattice_stamp, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, *vol, origins, n,
                       row_limit, ws.origin_stamp, epoch, n_dev, build_list ? (int32_t*)nullptr : ws.n_list);
    BNV_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_lattice_neighbors, dim3(capped_grid((n * 27 + 255) / 256, 16)), dim3(256), 0, stream, *vol, origins,
                     n, weights, row_limit, (float)grid->min_pts_in_grid, ws.nbr_rows, ws.stamp, epoch,
                     build_list ? ws.list : (int32_t*)nullptr, ws.n_list, row_skip, ws.origin_stamp, n_dev,
                     (prestamped && !build_list) ? ws.n_list : (int32_t*)nullptr);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

int bnv_lattice_neighbors(const bnv_volume_t* vol, const bnv_grid_t* grid, const float* weights, int64_t row_limit,
                          const int64_t* origins, int64_t n, const int32_t* n_dev, const uint8_t* row_skip,
                          int build_list, void* ws_ptr, size_t ws_bytes, int32_t epoch, bnv_stream_t stream) {
  return lattice_neighbors_impl(vol, grid, weights, row_limit, origins, n, n_dev, row_skip, build_list, ws_ptr,
                                ws_bytes, epoch, false, stream);
}

// does this call work on the volume's persistent tables (include/bnv_fusion.h: bnv_volume_t.lattice_persist)?
static bool lattice_persist(const bnv_volume_t* vol) {
  return vol && vol->lattice_persist && vol->lattice_table && vol->lattice_have;
}

static int lattice_mark_impl(const bnv_volume_t* vol, int64_t n, const int32_t* n_dev, void* ws_ptr, size_t ws_bytes,
                             int32_t epoch, bool clear, bnv_stream_t stream_) {
  if (!vol_ok_ro(vol) || n < 0 || !ws_ptr || epoch == 0) return BNV_ERR_INVALID_ARGUMENT;
  LatticeWs ws;
  if (lattice_ws_layout(n, vol->row_capacity, (char*)ws_ptr, &ws) > ws_bytes) return BNV_ERR_WORKSPACE_TOO_SMALL;
  hipStream_t stream = (hipStream_t)stream_;
  // entries listed, tile counter of the table kernel, spare (bnv_decode_lattice: cleared by k_lattice_neighbors)
  if (clear) BNV_HIP_CHECK(hipMemsetAsync(ws.n_list + 1, 0, 12, stream));
  if (n == 0) return BNV_OK;
  MarkFused F = {};
  F.have = lattice_persist(vol) ? vol->lattice_have : nullptr;
  const dim3 mgrid(capped_grid((n * 27 + kMarkThreads * kMarkChunks - 1) / (kMarkThreads * kMarkChunks), 2));
  const dim3 ogrid(capped_grid((n + kMoThreads - 1) / kMoThreads, 4));
  const bool per_origin = g_mark_per_origin.load(std::memory_order_relaxed) != 0;
  if (per_origin)
    hipLaunchKernelGGL((k_lattice_mark_o<false>), ogrid, dim3(kMoThreads), 0, stream, ws.nbr_rows, n,
                       ws.origin_stamp, epoch, ws.need_mask, ws.entries, ws.n_list + 1, ws.entry_capacity, n_dev, F);
  else
    hipLaunchKernelGGL((k_lattice_mark<false>), mgrid, dim3(kMarkThreads), 0, stream, ws.nbr_rows, n,
                       ws.origin_stamp, epoch, ws.need_mask, ws.entries, ws.n_list + 1, ws.entry_capacity, n_dev, F);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

// stamp (unless the frame's upsert did it) -> neighbours + mark in ONE launch
static int lattice_neighbors_mark_fused(const bnv_volume_t* vol, const bnv_grid_t* grid, const float* weights,
                                        int64_t row_limit, const int64_t* origins, int64_t n, const int32_t* n_dev,
                                        void* ws_ptr, size_t ws_bytes, int32_t epoch, bool prestamped,
                                        bnv_stream_t stream_) {
  if (!vol_ok_ro(vol) || !grid || !weights || n < 0 || epoch == 0 || !ws_ptr) return BNV_ERR_INVALID_ARGUMENT;
  LatticeWs ws;
  if (lattice_ws_layout(n, vol->row_capacity, (char*)ws_ptr, &ws) > ws_bytes) return BNV_ERR_WORKSPACE_TOO_SMALL;
  hipStream_t stream = (hipStream_t)stream_;
  if (n == 0) return BNV_OK;
  if (!origins) return BNV_ERR_INVALID_ARGUMENT;
  if (!prestamped) {   // (also clears the control words of the stages behind)
    hipLaunchKernelGGL(k_lattice_stamp, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, *vol, origins, n,
                       row_limit, ws.origin_stamp, epoch, n_dev, ws.n_list);
    BNV_LAUNCH_CHECK();
  }   // (prestamped: the frame's upsert has cleared the control words too)
  MarkFused F = {};
  F.v = *vol;
  F.origins = origins;
  F.weights = weights;
  F.row_limit = row_limit;
  F.min_pts = (float)grid->min_pts_in_grid;
  F.nbr_rows_out = ws.nbr_rows;
  F.have = lattice_persist(vol) ? vol->lattice_have : nullptr;
  const dim3 mgrid(capped_grid((n * 27 + kMarkThreads * kMarkChunks - 1) / (kMarkThreads * kMarkChunks), 2));
  const dim3 ogrid(capped_grid((n + kMoThreads - 1) / kMoThreads, 4));
  // (a shard's call keeps the per-point kernel: measured equal to slightly better there, profiles/r05_experiments.txt [e7])
  const bool per_origin = g_mark_per_origin.load(std::memory_order_relaxed) != 0 && grid->shard_world <= 1;
  if (per_origin)
    hipLaunchKernelGGL((k_lattice_mark_o<true>), ogrid, dim3(kMoThreads), 0, stream, (const int32_t*)nullptr, n,
                       ws.origin_stamp, epoch, ws.need_mask, ws.entries, ws.n_list + 1, ws.entry_capacity, n_dev, F);
  else
    hipLaunchKernelGGL((k_lattice_mark<true>), mgrid, dim3(kMarkThreads), 0, stream, (const int32_t*)nullptr, n,
                       ws.origin_stamp, epoch, ws.need_mask, ws.entries, ws.n_list + 1, ws.entry_capacity, n_dev, F);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

int bnv_lattice_mark(const bnv_volume_t* vol, int64_t n, const int32_t* n_dev, void* ws_ptr, size_t ws_bytes,
                     int32_t epoch, bnv_stream_t stream) {
  return lattice_mark_impl(vol, n, n_dev, ws_ptr, ws_bytes, epoch, true, stream);
}

static int lattice_table_impl(const bnv_volume_t* vol, const bnv_grid_t* grid, const float* features,
                              const float* sdfmlp_pack, int64_t n_voxels, int use_entries, void* ws_ptr,
                              size_t ws_bytes, int max_workgroups, bnv_stream_t stream);

int bnv_lattice_table(const bnv_volume_t* vol, const bnv_grid_t* grid, const float* features,
                      const float* sdfmlp_pack, int64_t n_voxels, int use_entries, void* ws_ptr, size_t ws_bytes,
                      bnv_stream_t stream) {
  return lattice_table_impl(vol, grid, features, sdfmlp_pack, n_voxels, use_entries, ws_ptr, ws_bytes, 0, stream);
}

static int lattice_table_impl(const bnv_volume_t* vol, const bnv_grid_t* grid, const float* features,
                              const float* sdfmlp_pack, int64_t n_voxels, int use_entries, void* ws_ptr,
                              size_t ws_bytes, int max_workgroups, bnv_stream_t stream) {
  if (g_num_cus <= 0) return BNV_ERR_NOT_INITIALISED;
  if (!vol_ok_ro(vol) || !grid || !features || !sdfmlp_pack || !ws_ptr || !mlp_mode_field_ok(grid->mlp_mode))
    return BNV_ERR_INVALID_ARGUMENT;
  LatticeWs ws;
  if (lattice_ws_layout(n_voxels, vol->row_capacity, (char*)ws_ptr, &ws) > ws_bytes)
    return BNV_ERR_WORKSPACE_TOO_SMALL;
  DecodeArgs a = {};
  a.vol = *vol;
  a.grid = *grid;
  a.features = features;
  a.pack = sdfmlp_pack;
  a.list = ws.list;
  a.n_list = ws.n_list;
  a.table = ws.table;
  a.need_mask = ws.need_mask;
  if (lattice_persist(vol)) {
    // ONE predicate for the three stages (mark, table, blend all ask lattice_persist(vol)): a call that switches the
    // persistent tables on works on the volume's own feature rows and on listed entries, or is rejected -- the marking
    // kernel has kept its books in lattice_have and the blend will read vol->lattice_table
    if (!use_entries || features != vol->features) return BNV_ERR_INVALID_ARGUMENT;
    a.table = vol->lattice_table;   // the listed entries are the ones the persistent table lacks
    a.need_mask = nullptr;
  }
  a.entries = use_entries ? ws.entries : nullptr;
  const int64_t evals = use_entries ? ws.entry_capacity : ws.list_capacity * 27;
  return launch_decode(MODE_LATTICE, mlp_mode_of(grid->mlp_mode), a, (evals + DM - 1) / DM, (hipStream_t)stream,
                       max_workgroups);
}

int bnv_lattice_blend(const bnv_volume_t* vol, const bnv_grid_t* grid, const int64_t* origins, int64_t n,
                      const int32_t* n_dev, const bnv_sdf_delta_t* delta, void* ws_ptr, size_t ws_bytes,
                      float* out_sdf, bnv_stream_t stream) {
  if (!vol_ok_ro(vol) || !grid || n < 0 || !ws_ptr) return BNV_ERR_INVALID_ARGUMENT;
  if (n == 0) return BNV_OK;
  if (!origins || !out_sdf) return BNV_ERR_INVALID_ARGUMENT;
  LatticeWs ws;
  if (lattice_ws_layout(n, vol->row_capacity, (char*)ws_ptr, &ws) > ws_bytes) return BNV_ERR_WORKSPACE_TOO_SMALL;
  bnv_sdf_delta_t d = {};
  if (delta) d = *delta;
  const float* table = lattice_persist(vol) ? vol->lattice_table : ws.table;
  if (d.data)
    hipLaunchKernelGGL(k_lattice_blend<true>, dim3(capped_grid((n * 27 + 255) / 256, 8)), dim3(256), 0,
                       (hipStream_t)stream, ws.nbr_rows, n, table, *grid, origins, d, out_sdf, n_dev);
  else
    hipLaunchKernelGGL(k_lattice_blend<false>, dim3(capped_grid((n * 27 + 256 * kBlendPpt - 1) / (256 * kBlendPpt), 8)), dim3(256), 0,
                       (hipStream_t)stream, ws.nbr_rows, n, table, *grid, origins, d, out_sdf, n_dev);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

static int decode_lattice_impl(const bnv_volume_t* vol, const bnv_grid_t* grid, const float* features,
                               const float* weights, int64_t row_limit, const float* sdfmlp_pack,
                               const int64_t* origins, int64_t n, const int32_t* n_dev, const bnv_sdf_delta_t* delta,
                               void* ws_ptr, size_t ws_bytes, int32_t epoch, float* out_sdf, bool prestamped,
                               bnv_stream_t stream, int stages = 7) {
  // stages: 1 = neighbour rows + live entries, 2 = table MLP, 4 = blend
  if (g_num_cus <= 0) return BNV_ERR_NOT_INITIALISED;
  if (!features || !grid || n < 0 || ((stages & 2) && !sdfmlp_pack)) return BNV_ERR_INVALID_ARGUMENT;
  // persistent tables belong to the volume's own rows: have-bits set by a call that decodes other features would poison them
  if (lattice_persist(vol) && features != vol->features) return BNV_ERR_INVALID_ARGUMENT;
  if (n == 0) return BNV_OK;
  // neighbour rows -> entries read by live lattice points -> MLP on those entries only -> blend.  The marking kernel
  // looks the neighbour rows up itself (one launch and a 10 MB round trip less): always with the per-origin kernel on
  // a volume that keeps its dense row index (k_lattice_mark_o: a thread's 27 look-ups are three rounds of independent
  // loads; tiny-cuda-nn frame 0.254 -> 0.236 ms, fp32 frame unchanged, profiles/r05_experiments.txt [e7]); with the
  // per-point kernel only on small calls (a shard's 1 / world of a frame), where the 256-thread look-up kernel of its
  // own would cost more than it hides (48.7 us for the pair against 62.4 us fused on whole frames)
  int rc;
  if (stages & 1) {
    const int fused_opt = g_fused_mark.load(std::memory_order_relaxed);
    const bool per_origin = g_mark_per_origin.load(std::memory_order_relaxed) != 0;
    const bool fuse = fused_opt == 1 || (fused_opt < 0 && ((per_origin && vol->brick) || n <= 49152 || grid->shard_world > 1));   // (n may be a capacity: a shard's frame holds 1 / world of it)
    if (fuse) {
      rc = lattice_neighbors_mark_fused(vol, grid, weights, row_limit, origins, n, n_dev, ws_ptr, ws_bytes, epoch,
                                        prestamped, stream);
      if (rc != BNV_OK) return rc;
    } else {
      rc = lattice_neighbors_impl(vol, grid, weights, row_limit, origins, n, n_dev, nullptr, 0, ws_ptr, ws_bytes, epoch,
                                  prestamped, stream);
      if (rc != BNV_OK) return rc;
      rc = lattice_mark_impl(vol, n, n_dev, ws_ptr, ws_bytes, epoch, false, stream);
      if (rc != BNV_OK) return rc;
    }
  }
  if (stages & 2) {
    rc = lattice_table_impl(vol, grid, features, sdfmlp_pack, n, 1, ws_ptr, ws_bytes, 0, stream);
    if (rc != BNV_OK) return rc;
  }
  if (!(stages & 4) || !out_sdf) return BNV_OK;   // (the caller blends itself, bnv_decode_lattice_stamped_tables)
  return bnv_lattice_blend(vol, grid, origins, n, n_dev, delta, ws_ptr, ws_bytes, out_sdf, stream);
}

int bnv_decode_lattice(const bnv_volume_t* vol, const bnv_grid_t* grid, const float* features,
                       const float* weights, int64_t row_limit, const float* sdfmlp_pack, const int64_t* origins,
                       int64_t n, const int32_t* n_dev, const bnv_sdf_delta_t* delta, void* ws_ptr, size_t ws_bytes,
                       int32_t epoch, float* out_sdf, bnv_stream_t stream) {
  return decode_lattice_impl(vol, grid, features, weights, row_limit, sdfmlp_pack, origins, n, n_dev, delta, ws_ptr,
                             ws_bytes, epoch, out_sdf, false, stream);
}

int bnv_decode_lattice_stamped_tables(const bnv_volume_t* vol, const bnv_grid_t* grid, const float* features,
                                      const float* weights, int64_t row_limit, const float* sdfmlp_pack,
                                      const int64_t* origins, int64_t n, const int32_t* n_dev, void* ws_ptr,
                                      size_t ws_bytes, int32_t epoch, bnv_stream_t stream) {
  return decode_lattice_impl(vol, grid, features, weights, row_limit, sdfmlp_pack, origins, n, n_dev, nullptr, ws_ptr,
                             ws_bytes, epoch, nullptr, true, stream, 3);
}

int bnv_decode_lattice_stamped(const bnv_volume_t* vol, const bnv_grid_t* grid, const float* features,
                               const float* weights, int64_t row_limit, const float* sdfmlp_pack,
                               const int64_t* origins, int64_t n, const int32_t* n_dev, const bnv_sdf_delta_t* delta,
                               void* ws_ptr, size_t ws_bytes, int32_t epoch, float* out_sdf, bnv_stream_t stream) {
  return decode_lattice_impl(vol, grid, features, weights, row_limit, sdfmlp_pack, origins, n, n_dev, delta, ws_ptr,
                             ws_bytes, epoch, out_sdf, true, stream);
}

}  // extern "C"
