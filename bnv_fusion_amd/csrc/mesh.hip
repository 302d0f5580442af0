// Per-voxel marching cubes on the decoded 3x3x3 SDF lattices.
//
// Reference: SparseVolume.meshlize, src/models/sparse_volume.py:697-766 -- for every active voxel whose
// lattice straddles the level (:742) skimage.measure.marching_cubes(sdf[3,3,3], level, spacing=0.5), then
// verts += origin - 0.5 (:749), * voxel_size + min_coords (:756); all voxels' meshes are concatenated.
// Here: two passes over the voxels (count triangles, then emit at the prefix-summed offsets), one thread per
// voxel walking its 8 cells; the output is a triangle soup in voxel / cell / table order, so it is
// reproducible.  The 256-case table is generated on the host (bnv_fusion_amd/mc_tables.py); scikit-image is
// not available, so the triangulation inside ambiguous cells is this table's, not Lewiner's (vertex
// positions -- the level crossings of the lattice edges -- are the same for every marching-cubes variant).
//
// HBM-bound: 108 B read per voxel (twice), 36 B written per triangle.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bnv_fusion.h"
#include "bnv_common.hpp"

namespace bnv {

// cube edges as corner pairs; corner c = 4 dx + 2 dy + dz (the lattice's flatten order), as mc_tables.EDGES
__device__ const int8_t kMcEdgeA[12] = {0, 0, 0, 1, 1, 2, 2, 3, 4, 4, 5, 6};
__device__ const int8_t kMcEdgeB[12] = {1, 2, 4, 3, 5, 3, 6, 7, 5, 6, 7, 7};
constexpr int kMcRow = 16;  // table row: up to 5 triangles (15 edge ids) + terminator

__device__ __forceinline__ bool mc_gate(const float (&s)[27], float level) {
  float mx = s[0], mn = s[0];
#pragma unroll
  for (int i = 1; i < 27; ++i) {
    mx = fmaxf(mx, s[i]);
    mn = fminf(mn, s[i]);
  }
  return mx > level && mn < level;  // sparse_volume.py:742
}

__device__ __forceinline__ int mc_case(const float (&s)[27], int cx, int cy, int cz, float level) {
  int c = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float v = s[(cx + ((k >> 2) & 1)) * 9 + (cy + ((k >> 1) & 1)) * 3 + (cz + (k & 1))];
    c |= (v < level) << k;
  }
  return c;
}

__global__ __launch_bounds__(256) void k_mc_count(const float* __restrict__ sdf, int64_t n,
                                                  const int32_t* __restrict__ n_dev, float level,
                                                  const int8_t* __restrict__ table, int32_t* __restrict__ counts) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  if (n_dev && v >= *n_dev) {
    counts[v] = 0;
    return;
  }
  float s[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) s[i] = sdf[v * 27 + i];
  int t = 0;
  if (mc_gate(s, level)) {
    for (int cell = 0; cell < 8; ++cell) {
      const int8_t* row = table + mc_case(s, cell >> 2, (cell >> 1) & 1, cell & 1, level) * kMcRow;
      for (int k = 0; k < kMcRow - 1 && row[k] >= 0; k += 3) ++t;
    }
  }
  counts[v] = t;
}

__global__ __launch_bounds__(256) void k_mc_emit(const float* __restrict__ sdf, const int64_t* __restrict__ origins,
                                                 int64_t n, const int32_t* __restrict__ n_dev, float level,
                                                 float voxel, float mx, float my, float mz,
                                                 const int8_t* __restrict__ table,
                                                 const int64_t* __restrict__ tri_offsets,
                                                 float* __restrict__ vertices) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n || (n_dev && v >= *n_dev)) return;
  float s[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) s[i] = sdf[v * 27 + i];
  if (!mc_gate(s, level)) return;
  const float org[3] = {(float)origins[v * 3] - 0.5f, (float)origins[v * 3 + 1] - 0.5f, (float)origins[v * 3 + 2] - 0.5f};
  const float mn[3] = {mx, my, mz};
  float* out = vertices + tri_offsets[v] * 9;
  for (int cell = 0; cell < 8; ++cell) {
    const int cc[3] = {cell >> 2, (cell >> 1) & 1, cell & 1};
    const int8_t* row = table + mc_case(s, cc[0], cc[1], cc[2], level) * kMcRow;
    for (int k = 0; k < kMcRow - 1 && row[k] >= 0; ++k) {
      const int a = kMcEdgeA[row[k]], b = kMcEdgeB[row[k]];
      const int pa[3] = {cc[0] + ((a >> 2) & 1), cc[1] + ((a >> 1) & 1), cc[2] + (a & 1)};
      const int pb[3] = {cc[0] + ((b >> 2) & 1), cc[1] + ((b >> 1) & 1), cc[2] + (b & 1)};
      const float va = s[pa[0] * 9 + pa[1] * 3 + pa[2]], vb = s[pb[0] * 9 + pb[1] * 3 + pb[2]];
      const float t = __fdiv_rn(__fsub_rn(level, va), __fsub_rn(vb, va));
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        float p = __fadd_rn((float)pa[d], __fmul_rn(t, (float)(pb[d] - pa[d])));  // lattice index units
        p = __fmul_rn(p, 0.5f);                                                    // spacing (:719)
        p = __fadd_rn(p, org[d]);                                                  // :749
        out[d] = __fadd_rn(__fmul_rn(p, voxel), mn[d]);                            // :756
      }
      out += 3;
    }
  }
}

// ---- indexed output: per voxel a vertex list WITHOUT duplicates + faces indexing it, concatenated over the voxels
// the way SparseVolume.meshlize does (sparse_volume.py:740-756: faces + last_face_id; last_face_id += max(faces) + 1
// = the voxel's vertex count, every vertex being used).  A vertex of a voxel's mesh lies on one of the 54 edges of its
// 3x3x3 lattice, so a 54-bit mask of the sign-changing lattice edges names the voxel's vertices; they are emitted in
// ascending edge order and a triangle corner's local index is the rank of its edge's bit.
__device__ __forceinline__ int mc_lattice_edge(const int (&pa)[3], int axis) {
  // edge from node pa to pa + e_axis; 18 edges per axis
  if (axis == 0) return pa[0] * 9 + pa[1] * 3 + pa[2];
  if (axis == 1) return 18 + pa[0] * 6 + pa[1] * 3 + pa[2];
  return 36 + pa[0] * 6 + pa[1] * 2 + pa[2];
}

__device__ __forceinline__ unsigned long long mc_edge_mask(const float (&s)[27], float level) {
  unsigned long long m = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const bool in0 = s[i * 9 + j * 3 + k] < level;
        const int pa[3] = {i, j, k};
        if (i < 2 && in0 != (s[(i + 1) * 9 + j * 3 + k] < level)) m |= 1ull << mc_lattice_edge(pa, 0);
        if (j < 2 && in0 != (s[i * 9 + (j + 1) * 3 + k] < level)) m |= 1ull << mc_lattice_edge(pa, 1);
        if (k < 2 && in0 != (s[i * 9 + j * 3 + k + 1] < level)) m |= 1ull << mc_lattice_edge(pa, 2);
      }
  return m;
}

__global__ __launch_bounds__(256) void k_mc_count_indexed(const float* __restrict__ sdf, int64_t n,
                                                          const int32_t* __restrict__ n_dev, float level,
                                                          const int8_t* __restrict__ table,
                                                          int32_t* __restrict__ n_verts, int32_t* __restrict__ n_tris) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  int nv = 0, nt = 0;
  if (!(n_dev && v >= *n_dev)) {
    float s[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) s[i] = sdf[v * 27 + i];
    if (mc_gate(s, level)) {
      nv = (int)__popcll(mc_edge_mask(s, level));
      for (int cell = 0; cell < 8; ++cell) {
        const int8_t* row = table + mc_case(s, cell >> 2, (cell >> 1) & 1, cell & 1, level) * kMcRow;
        for (int k = 0; k < kMcRow - 1 && row[k] >= 0; k += 3) ++nt;
      }
    }
  }
  n_verts[v] = nv;
  n_tris[v] = nt;
}

__global__ __launch_bounds__(256) void k_mc_emit_indexed(const float* __restrict__ sdf,
                                                         const int64_t* __restrict__ origins, int64_t n,
                                                         const int32_t* __restrict__ n_dev, float level, float voxel,
                                                         float mx, float my, float mz, const int8_t* __restrict__ table,
                                                         const int64_t* __restrict__ vert_offsets,
                                                         const int64_t* __restrict__ tri_offsets,
                                                         float* __restrict__ vertices, int64_t* __restrict__ faces) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n || (n_dev && v >= *n_dev)) return;
  float s[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) s[i] = sdf[v * 27 + i];
  if (!mc_gate(s, level)) return;
  const unsigned long long mask = mc_edge_mask(s, level);
  const float org[3] = {(float)origins[v * 3] - 0.5f, (float)origins[v * 3 + 1] - 0.5f, (float)origins[v * 3 + 2] - 0.5f};
  const float mn[3] = {mx, my, mz};
  // vertices, ascending lattice-edge order
  float* vo = vertices + vert_offsets[v] * 3;
  for (int axis = 0; axis < 3; ++axis) {
    const int ni = axis == 0 ? 2 : 3, nj = axis == 1 ? 2 : 3, nk = axis == 2 ? 2 : 3;
    for (int i = 0; i < ni; ++i)
      for (int j = 0; j < nj; ++j)
        for (int k = 0; k < nk; ++k) {
          const int pa[3] = {i, j, k};
          if (!((mask >> mc_lattice_edge(pa, axis)) & 1ull)) continue;
          const int pb[3] = {i + (axis == 0), j + (axis == 1), k + (axis == 2)};
          const float va = s[pa[0] * 9 + pa[1] * 3 + pa[2]], vb = s[pb[0] * 9 + pb[1] * 3 + pb[2]];
          const float t = __fdiv_rn(__fsub_rn(level, va), __fsub_rn(vb, va));
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            float p = __fadd_rn((float)pa[d], __fmul_rn(t, (float)(pb[d] - pa[d])));  // lattice index units
            p = __fmul_rn(p, 0.5f);                                                    // spacing (:719)
            p = __fadd_rn(p, org[d]);                                                  // :749
            vo[d] = __fadd_rn(__fmul_rn(p, voxel), mn[d]);                             // :756
          }
          vo += 3;
        }
  }
  // faces: cell / table order, global indices = the voxel's vertex offset + rank of the corner's lattice edge
  int64_t* fo = faces + tri_offsets[v] * 3;
  const int64_t base = vert_offsets[v];
  for (int cell = 0; cell < 8; ++cell) {
    const int cc[3] = {cell >> 2, (cell >> 1) & 1, cell & 1};
    const int8_t* row = table + mc_case(s, cc[0], cc[1], cc[2], level) * kMcRow;
    for (int k = 0; k < kMcRow - 1 && row[k] >= 0; ++k) {
      const int a = kMcEdgeA[row[k]], b = kMcEdgeB[row[k]];   // a < b, they differ in one bit
      const int pa[3] = {cc[0] + ((a >> 2) & 1), cc[1] + ((a >> 1) & 1), cc[2] + (a & 1)};
      const int axis = (a ^ b) == 4 ? 0 : ((a ^ b) == 2 ? 1 : 2);
      const int id = mc_lattice_edge(pa, axis);
      *fo++ = base + (int64_t)__popcll(mask & ((1ull << id) - 1ull));
    }
  }
}

}  // namespace bnv

using namespace bnv;

extern "C" {

int bnv_mc_count(const float* sdf, int64_t n, const int32_t* n_dev, float level, const int8_t* tri_table,
                 int32_t* counts, bnv_stream_t stream) {
  if (n < 0 || (n > 0 && (!sdf || !tri_table || !counts))) return BNV_ERR_INVALID_ARGUMENT;
  if (n == 0) return BNV_OK;
  hipLaunchKernelGGL(k_mc_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sdf, n, n_dev,
                     level, tri_table, counts);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

int bnv_mc_emit(const float* sdf, const int64_t* origins, int64_t n, const int32_t* n_dev, float level,
                float voxel_size, const float min_coords[3], const int8_t* tri_table, const int64_t* tri_offsets,
                float* vertices, bnv_stream_t stream) {
  if (n < 0 || (n > 0 && (!sdf || !origins || !min_coords || !tri_table || !tri_offsets || !vertices)))
    return BNV_ERR_INVALID_ARGUMENT;
  if (n == 0) return BNV_OK;
  hipLaunchKernelGGL(k_mc_emit, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sdf, origins, n,
                     n_dev, level, voxel_size, min_coords[0], min_coords[1], min_coords[2], tri_table, tri_offsets,
                     vertices);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

int bnv_mc_count_indexed(const float* sdf, int64_t n, const int32_t* n_dev, float level, const int8_t* tri_table,
                         int32_t* n_verts, int32_t* n_tris, bnv_stream_t stream) {
  if (n < 0 || (n > 0 && (!sdf || !tri_table || !n_verts || !n_tris))) return BNV_ERR_INVALID_ARGUMENT;
  if (n == 0) return BNV_OK;
  hipLaunchKernelGGL(k_mc_count_indexed, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sdf, n,
                     n_dev, level, tri_table, n_verts, n_tris);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

int bnv_mc_emit_indexed(const float* sdf, const int64_t* origins, int64_t n, const int32_t* n_dev, float level,
                        float voxel_size, const float min_coords[3], const int8_t* tri_table,
                        const int64_t* vert_offsets, const int64_t* tri_offsets, float* vertices, int64_t* faces,
                        bnv_stream_t stream) {
  if (n < 0 || (n > 0 && (!sdf || !origins || !min_coords || !tri_table || !vert_offsets || !tri_offsets || !vertices ||
                          !faces)))
    return BNV_ERR_INVALID_ARGUMENT;
  if (n == 0) return BNV_OK;
  hipLaunchKernelGGL(k_mc_emit_indexed, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sdf,
                     origins, n, n_dev, level, voxel_size, min_coords[0], min_coords[1], min_coords[2], tri_table,
                     vert_offsets, tri_offsets, vertices, faces);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

}  // extern "C"
