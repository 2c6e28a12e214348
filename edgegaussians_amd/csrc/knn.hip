// nearest neighbours and the orientation regularisers, on device.
//
// Replaces, for the last 150 of 400 epochs of an ABC run (train_gaussians.py:108-131):
//   update_nearest_neighbors / k_nearest_sklearn   edge_gs.py:326-344,135-151  (CPU KD-tree + D2H of means)
//   compute_direction_loss                          edge_gs.py:346-373
//   compute_ratio_loss                              edge_gs.py:375-380
// kNN: uniform grid over the points' bounding box (cell edge chosen by the caller for ~2 points per
// cell), counting sort of the points by cell, then one thread per query point scanning growing cube
// shells of cells with a K-best list in registers; the search stops when the K-th distance is inside
// the scanned block.  Exact (not approximate) neighbours.
// The two losses are streaming per-Gaussian kernels that produce value and gradient in one pass.
#include "common.h"

namespace eg {

struct Grid {
  float ox, oy, oz, inv_cell, cell;
  int nx, ny, nz;
};

__device__ __forceinline__ int3 cell_of_point(const Grid &g, float x, float y, float z) {
  int3 c;
  c.x = min(max((int)floorf((x - g.ox) * g.inv_cell), 0), g.nx - 1);
  c.y = min(max((int)floorf((y - g.oy) * g.inv_cell), 0), g.ny - 1);
  c.z = min(max((int)floorf((z - g.oz) * g.inv_cell), 0), g.nz - 1);
  return c;
}

__global__ void __launch_bounds__(256)
knn_count_kernel(const float *__restrict__ pts, int N, Grid g, int *__restrict__ cell_of, int *__restrict__ counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int3 c = cell_of_point(g, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
  const int id = (c.z * g.ny + c.y) * g.nx + c.x;
  cell_of[i] = id;
  atomicAdd(&counts[id], 1);
}

// counts are consumed back to zero, like the tile binning (no memset between calls).  The points are stored in
// cell order WITH their coordinates (x y z index): the query loop then streams 16-byte records instead of
// chasing an index into the unsorted array per candidate (one thread walks hundreds of candidates when the
// points sit on curves: the dependent gather was 0.9 ms per call on 6 k trained Gaussians)
__global__ void __launch_bounds__(256)
knn_scatter_kernel(const float *__restrict__ pts, const int *__restrict__ cell_of, int N,
                   const int *__restrict__ cell_start, int *__restrict__ counts, float4 *__restrict__ sorted) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int id = cell_of[i];
  sorted[cell_start[id] + atomicSub(&counts[id], 1) - 1] =
      make_float4(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], __int_as_float(i));
}

// sorted insertion of candidate (d, j) into the K-best list (ties keep the lower index first, like a stable
// sort on (d, j))
template <int KMAX>
__device__ __forceinline__ void knn_insert(float (&bd)[KMAX], int (&bi)[KMAX], float d, int j) {
  if (d < bd[KMAX - 1] || (d == bd[KMAX - 1] && j < bi[KMAX - 1])) {
    bd[KMAX - 1] = d;
    bi[KMAX - 1] = j;
#pragma unroll
    for (int k = KMAX - 1; k > 0; --k) {
      const bool sw = (bd[k] < bd[k - 1]) || (bd[k] == bd[k - 1] && bi[k] < bi[k - 1]);
      const float td = sw ? bd[k - 1] : bd[k];
      const int ti = sw ? bi[k - 1] : bi[k];
      bd[k - 1] = sw ? bd[k] : bd[k - 1];
      bi[k - 1] = sw ? bi[k] : bi[k - 1];
      bd[k] = td;
      bi[k] = ti;
    }
  }
}

template <int KMAX>
__global__ void __launch_bounds__(128)
knn_query_kernel(const float *__restrict__ pts, int N, int K, Grid g, const int *__restrict__ cell_start,
                 const float4 *__restrict__ sorted, int *__restrict__ out_idx, float *__restrict__ out_d2,
                 int r_brute) {
  // thread t answers the query of the t-th point IN CELL ORDER: the lanes of a wave then sit in neighbouring
  // cells, walk (nearly) the same candidate ranges (cache lines shared, similar trip counts) and settle together
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N) return;
  const float4 me = sorted[t];
  const int i = __float_as_int(me.w);
  const float x = me.x, y = me.y, z = me.z;
  const int3 c = cell_of_point(g, x, y, z);
  float bd[KMAX];
  int bi[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) { bd[k] = 3.0e38f; bi[k] = -1; }
  // Blocks of cells of radius r = 1, then whatever the K-th distance found so far asks for, around the query's
  // cell, each scanned afresh.  Cells that are
  // consecutive in x hold consecutive runs of the cell-ordered records, so a whole ROW of the block is one
  // contiguous range: (2r+1)^2 range look-ups per block instead of (2r+1)^3 cell look-ups -- the look-up is a
  // dependent global load (~600 cycles), a candidate costs ~40, so re-evaluating the inner block is cheaper
  // than visiting its shell cell by cell.  A query that is still not settled at r_brute (an outlier far from
  // everything) scans ALL records sequentially, every lane of the wave reading the same address.
  const int rmax = max(max(g.nx, g.ny), g.nz);
  bool settled = false;
  for (int r = 1; !settled;) {
    if (r > r_brute && r < rmax) break;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { bd[k] = 3.0e38f; bi[k] = -1; }
    const int x0 = max(c.x - r, 0), x1 = min(c.x + r, g.nx - 1);
    for (int cz = max(c.z - r, 0); cz <= min(c.z + r, g.nz - 1); ++cz)
      for (int cy = max(c.y - r, 0); cy <= min(c.y + r, g.ny - 1); ++cy) {
        const int row = (cz * g.ny + cy) * g.nx;
        const int s1 = cell_start[row + x1 + 1];
        for (int s = cell_start[row + x0]; s < s1; s += 4) {
          float4 c4[4];  // four records in flight per wait: one thread's walk is a chain of dependent loads
#pragma unroll
          for (int u = 0; u < 4; ++u) c4[u] = sorted[min(s + u, s1 - 1)];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int j = __float_as_int(c4[u].w);
            if (s + u >= s1 || j == i) continue;
            const float ex = c4[u].x - x, ey = c4[u].y - y, ez = c4[u].z - z;
            knn_insert<KMAX>(bd, bi, ex * ex + ey * ey + ez * ez, j);
          }
        }
      }
    // everything outside the (2r+1)^3 block is at least r * cell away
    const float reach = (float)r * g.cell;
    float kth = 3.0e38f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) kth = (k == K - 1) ? bd[k] : kth;
    settled = (kth <= reach * reach) || (r >= rmax);
    // not settled: if K candidates are known, everything nearer than the K-th lies within sqrt(kth) of the query,
    // so ONE more block of exactly that radius settles it (doubling blindly scanned 14x the needed volume for an
    // isolated point, and such a lane holds its whole wave); with fewer than K known, double
    const int r_need = (kth < 1.0e38f) ? (int)ceilf(sqrtf(kth) * g.inv_cell) : 2 * r;
    r = min(max(r_need, r + 1), max(rmax, r + 1));
  }
  if (!settled) {
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { bd[k] = 3.0e38f; bi[k] = -1; }
    for (int s = 0; s < N; s += 4) {
      float4 c4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) c4[u] = sorted[min(s + u, N - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = __float_as_int(c4[u].w);
        if (s + u >= N || j == i) continue;
        const float ex = c4[u].x - x, ey = c4[u].y - y, ez = c4[u].z - z;
        knn_insert<KMAX>(bd, bi, ex * ex + ey * ey + ez * ez, j);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    if (k < K) {
      out_idx[(size_t)i * K + k] = bi[k];
      if (out_d2) out_d2[(size_t)i * K + k] = bd[k];
    }
}

// ---------------------------------------------------------------------------------------------
// Exhaustive search for small N (the ABC-NEF runs end at ~10 k Gaussians, three quarters of them faint floaters
// spread through the volume while the rest sit on curves: no uniform grid fits both, one isolated query held its
// whole wave for ~800 us).  N^2 distance evaluations are cheap when the whole chip takes part: workgroup (b, s)
// answers queries 256 b .. 256 b + 255 against candidate chunk s; every lane of a wave reads the SAME candidate
// (a wave-uniform address: scalar loads, no LDS), ~10 VALU instructions per pair, K-best list in registers;
// a second kernel merges the S partial lists of a query.  Same (distance, index) order as the grid search.
template <int KMAX>
__global__ void __launch_bounds__(256)
knn_exhaustive_kernel(const float *__restrict__ pts, int N, int chunk, float *__restrict__ part_d,
                      int *__restrict__ part_i) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int s = blockIdx.y;
  const int q = min(i, N - 1);
  const float x = pts[3 * q], y = pts[3 * q + 1], z = pts[3 * q + 2];
  float bd[KMAX];
  int bi[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) { bd[k] = 3.0e38f; bi[k] = -1; }
  const int c0 = s * chunk, c1 = min(N, c0 + chunk);
  int j = c0;
  for (; j + 8 <= c1; j += 8) {  // eight candidates = 24 consecutive floats per round of scalar loads
    float c[24];
#pragma unroll
    for (int u = 0; u < 24; ++u) c[u] = pts[3 * j + u];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float ex = c[3 * u] - x, ey = c[3 * u + 1] - y, ez = c[3 * u + 2] - z;
      const float d = ex * ex + ey * ey + ez * ez;
      if (j + u != i) knn_insert<KMAX>(bd, bi, d, j + u);
    }
  }
  for (; j < c1; ++j) {
    const float ex = pts[3 * j] - x, ey = pts[3 * j + 1] - y, ez = pts[3 * j + 2] - z;
    const float d = ex * ex + ey * ey + ez * ez;
    if (j != i) knn_insert<KMAX>(bd, bi, d, j);
  }
  if (i >= N) return;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    part_d[((size_t)s * KMAX + k) * N + i] = bd[k];
    part_i[((size_t)s * KMAX + k) * N + i] = bi[k];
  }
}

template <int KMAX>
__global__ void __launch_bounds__(256)
knn_merge_kernel(const float *__restrict__ part_d, const int *__restrict__ part_i, int N, int K, int S,
                 int *__restrict__ out_idx, float *__restrict__ out_d2) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  float bd[KMAX];
  int bi[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    bd[k] = part_d[(size_t)k * N + i];
    bi[k] = part_i[(size_t)k * N + i];
  }
  for (int s = 1; s < S; ++s)
    for (int k = 0; k < K; ++k) {  // the partial lists are ascending: stop at the first entry that does not enter
      const float d = part_d[((size_t)s * KMAX + k) * N + i];
      const int j = part_i[((size_t)s * KMAX + k) * N + i];
      if (j < 0 || !(d < bd[KMAX - 1] || (d == bd[KMAX - 1] && j < bi[KMAX - 1]))) break;
      knn_insert<KMAX>(bd, bi, d, j);
    }
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    if (k < K) {
      out_idx[(size_t)i * K + k] = bi[k];
      if (out_d2) out_d2[(size_t)i * K + k] = bd[k];
    }
}

// ---------------------------------------------------------------------------------------------
// direction loss (edge_gs.py:346-373): 1 - mean_i mean_k | m_i . unit(mu_i - mu_nn(i,k)) |
// with m_i = column argmax_k(scale) of R(q_i).  One thread per Gaussian: value (sum of alignments, the
// caller forms 1 - sum / (N k)) and UNSCALED gradients d(sum)/d{mu, q} (the caller multiplies by
// -lambda / (N k); lambda is data-dependent in the reference, train_gaussians.py:113).
// top_k in (0, K): the 'enforce_half' method (edge_gs.py:366-369) -- only the top_k best-aligned of the K
// listed neighbours count (sort descending, mean of the first k); otherwise every neighbour counts.
constexpr int kMaxDirNN = 32;
__global__ void __launch_bounds__(256)
direction_loss_kernel(const float *__restrict__ means, const float *__restrict__ quats,
                      const float *__restrict__ log_scales, const int *__restrict__ nn, int N, int K, int top_k,
                      float *__restrict__ g_means, float *__restrict__ g_quats, float *__restrict__ sum_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  if (i < N) {
    float w = quats[4 * i], x = quats[4 * i + 1], y = quats[4 * i + 2], z = quats[4 * i + 3];
    const float qinv = rsqrtf(w * w + x * x + y * y + z * z);
    w *= qinv; x *= qinv; y *= qinv; z *= qinv;
    const float s0 = log_scales[3 * i], s1 = log_scales[3 * i + 1], s2 = log_scales[3 * i + 2];
    const int c = (s0 >= s1 && s0 >= s2) ? 0 : ((s1 >= s2) ? 1 : 2);  // torch.argmax: first maximum
    float R[9];
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - w * z);       R[2] = 2.f * (x * z + w * y);
    R[3] = 2.f * (x * y + w * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - w * x);
    R[6] = 2.f * (x * z - w * y);       R[7] = 2.f * (y * z + w * x);       R[8] = 1.f - 2.f * (x * x + y * y);
    const float mx = R[c], my = R[3 + c], mz = R[6 + c];
    const float px = means[3 * i], py = means[3 * i + 1], pz = means[3 * i + 2];
    float vmx = 0.f, vmy = 0.f, vmz = 0.f, vpx = 0.f, vpy = 0.f, vpz = 0.f;
    // enforce_half: mark the top_k largest alignments (first pass), ties resolved towards the lower slot
    unsigned chosen = 0xffffffffu;
    if (top_k > 0 && top_k < K) {
      float al[kMaxDirNN];
      for (int k = 0; k < K; ++k) {
        const int j = nn[(size_t)i * K + k];
        float a = -1.f;
        if (j >= 0) {
          const float dx = px - means[3 * j], dy = py - means[3 * j + 1], dz = pz - means[3 * j + 2];
          const float n2 = dx * dx + dy * dy + dz * dz;
          // a coincident neighbour gives 0/0 = NaN in the reference; it is skipped here like below
          if (n2 > 0.f) a = fabsf((mx * dx + my * dy + mz * dz) * rsqrtf(n2));
        }
        al[k] = a;
      }
      chosen = 0u;
      for (int t = 0; t < top_k; ++t) {
        int best = -1;
        float bv = -2.f;
        for (int k = 0; k < K; ++k)
          if (!((chosen >> k) & 1u) && al[k] > bv) { bv = al[k]; best = k; }
        if (best >= 0) chosen |= 1u << best;
      }
    }
    for (int k = 0; k < K; ++k) {
      const int j = nn[(size_t)i * K + k];
      if (j < 0 || !((chosen >> k) & 1u)) continue;
      const float dx = px - means[3 * j], dy = py - means[3 * j + 1], dz = pz - means[3 * j + 2];
      const float n2 = dx * dx + dy * dy + dz * dz;
      if (!(n2 > 0.f)) continue;
      const float inv = rsqrtf(n2);
      const float ux = dx * inv, uy = dy * inv, uz = dz * inv;
      const float dot = mx * ux + my * uy + mz * uz;
      acc += fabsf(dot);
      const float sg = (dot > 0.f) ? 1.f : ((dot < 0.f) ? -1.f : 0.f);
      vmx += sg * ux; vmy += sg * uy; vmz += sg * uz;
      // d|dot|/du = sg * m; through the normalisation: (v - (v.u) u) / |d|
      const float vu = sg * dot;  // (sg m) . u
      const float gx = (sg * mx - vu * ux) * inv, gy = (sg * my - vu * uy) * inv, gz = (sg * mz - vu * uz) * inv;
      vpx += gx; vpy += gy; vpz += gz;
      unsafeAtomicAdd(&g_means[3 * j], -gx);
      unsafeAtomicAdd(&g_means[3 * j + 1], -gy);
      unsafeAtomicAdd(&g_means[3 * j + 2], -gz);
    }
    unsafeAtomicAdd(&g_means[3 * i], vpx);
    unsafeAtomicAdd(&g_means[3 * i + 1], vpy);
    unsafeAtomicAdd(&g_means[3 * i + 2], vpz);
    // rotation column c -> normalised quaternion -> raw quaternion
    float vR[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    vR[c] = vmx; vR[3 + c] = vmy; vR[6 + c] = vmz;
    const float nw = 2.f * (x * (vR[7] - vR[5]) + y * (vR[2] - vR[6]) + z * (vR[3] - vR[1]));
    const float nx = 2.f * (-2.f * x * (vR[4] + vR[8]) + y * (vR[1] + vR[3]) + z * (vR[2] + vR[6]) + w * (vR[7] - vR[5]));
    const float ny = 2.f * (x * (vR[1] + vR[3]) - 2.f * y * (vR[0] + vR[8]) + z * (vR[5] + vR[7]) + w * (vR[2] - vR[6]));
    const float nz = 2.f * (x * (vR[2] + vR[6]) + y * (vR[5] + vR[7]) - 2.f * z * (vR[0] + vR[4]) + w * (vR[3] - vR[1]));
    const float d = nw * w + nx * x + ny * y + nz * z;
    g_quats[4 * i] = (nw - d * w) * qinv;
    g_quats[4 * i + 1] = (nx - d * x) * qinv;
    g_quats[4 * i + 2] = (ny - d * y) * qinv;
    g_quats[4 * i + 3] = (nz - d * z) * qinv;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0 && acc != 0.f) unsafeAtomicAdd(sum_out, acc);
}

// ratio loss (edge_gs.py:375-380): mean_i second-largest / largest scale.  Value (sum of ratios) and
// unscaled gradient w.r.t. the LOG-scales: d r / d ls_second = r, d r / d ls_first = -r.
__global__ void __launch_bounds__(256)
ratio_loss_kernel(const float *__restrict__ log_scales, int N, float *__restrict__ g_scales,
                  float *__restrict__ sum_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float r = 0.f;
  if (i < N) {
    const float s[3] = {log_scales[3 * i], log_scales[3 * i + 1], log_scales[3 * i + 2]};
    int a = 0;  // largest (first maximum)
    if (s[1] > s[a]) a = 1;
    if (s[2] > s[a]) a = 2;
    int b = (a == 0) ? 1 : 0;  // second largest (first among the rest)
    for (int k = 0; k < 3; ++k)
      if (k != a && k != b && s[k] > s[b]) b = k;
    r = expf(s[b] - s[a]);
    float g[3] = {0.f, 0.f, 0.f};
    g[b] = r;
    g[a] = -r;
    g_scales[3 * i] = g[0]; g_scales[3 * i + 1] = g[1]; g_scales[3 * i + 2] = g[2];
  }
  float acc = r;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0 && acc != 0.f) unsafeAtomicAdd(sum_out, acc);
}

}  // namespace eg

using namespace eg;

extern "C" int eg_knn(const float *points, int32_t N, int32_t K, const float *origin_host /*[3]*/, float cell,
                      const int32_t *dims_host /*[3]*/, int32_t *cell_of /*[N]*/,
                      int32_t *cell_counts /*[C], zero on entry and on exit*/, int32_t *cell_start /*[C+1]*/,
                      float *sorted /*[N,4]*/, int32_t *out_idx /*[N,K]*/, float *out_d2 /*[N,K]|NULL*/,
                      eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && K >= 1 && K <= 32 && cell > 0.f && origin_host && dims_host, "bad arguments");
  if (N == 0) return EG_OK;
  EG_REQUIRE(points && cell_of && cell_counts && cell_start && sorted && out_idx, "null pointer");
  Grid g;
  g.ox = origin_host[0]; g.oy = origin_host[1]; g.oz = origin_host[2];
  g.cell = cell; g.inv_cell = 1.f / cell;
  g.nx = dims_host[0]; g.ny = dims_host[1]; g.nz = dims_host[2];
  EG_REQUIRE(g.nx > 0 && g.ny > 0 && g.nz > 0 && (int64_t)g.nx * g.ny * g.nz < (1ll << 30), "bad grid");
  const int C = g.nx * g.ny * g.nz;
  hipStream_t st = as_stream(stream);
  // a block of radius r costs (2r+1)^2 dependent look-ups (~600 cycles each), the exhaustive scan ~40 cycles per
  // point: beyond this radius the scan is the cheaper way to settle an outlier
  int r_brute = 1;
  while ((2 * (2 * r_brute) + 1) * (2 * (2 * r_brute) + 1) * 15 < N) r_brute *= 2;
  knn_count_kernel<<<cdiv(N, 256), 256, 0, st>>>(points, N, g, cell_of, cell_counts);
  int rc = eg_tile_offsets(cell_counts, C, (int64_t)1 << 40, cell_start, nullptr, nullptr, stream);
  if (rc) return rc;
  knn_scatter_kernel<<<cdiv(N, 256), 256, 0, st>>>(points, cell_of, N, cell_start, cell_counts, (float4 *)sorted);
  if (K <= 8)
    knn_query_kernel<8><<<cdiv(N, 128), 128, 0, st>>>(points, N, K, g, cell_start, (const float4 *)sorted, out_idx,
                                                         out_d2, r_brute);
  else if (K <= 16)
    knn_query_kernel<16><<<cdiv(N, 128), 128, 0, st>>>(points, N, K, g, cell_start, (const float4 *)sorted, out_idx,
                                                         out_d2, r_brute);
  else  // 'enforce_half' with dir_loss_num_nn = 10 asks for 2 k + 1 = 21 neighbours (edge_gs.py:339-340)
    knn_query_kernel<32><<<cdiv(N, 128), 128, 0, st>>>(points, N, K, g, cell_start, (const float4 *)sorted, out_idx,
                                                         out_d2, r_brute);
  return check_launch("knn");
}

// candidate chunks per query block of the exhaustive search: enough workgroups to fill the chip (~4 per CU)
static int knn_small_splits(int N) {
  const int blocks = cdiv(N, 256);
  return max(1, min(min(64, cdiv(N, 64)), cdiv(1024, blocks)));
}

extern "C" int64_t eg_knn_small_scratch_bytes(int32_t N, int32_t K) {
  if (N <= 0 || K < 1 || K > 32) return 0;
  const int kmax = K <= 8 ? 8 : (K <= 16 ? 16 : 32);
  return (int64_t)knn_small_splits(N) * kmax * N * 8;
}

extern "C" int eg_knn_small(const float *points, int32_t N, int32_t K, void *scratch, int32_t *out_idx,
                            float *out_d2, eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && K >= 1 && K <= 32, "bad arguments");
  if (N == 0) return EG_OK;
  EG_REQUIRE(points && scratch && out_idx, "null pointer");
  EG_REQUIRE(N <= (1 << 16), "eg_knn_small: N <= 65536 (use the grid search, eg_knn)");
  hipStream_t st = as_stream(stream);
  const int S = knn_small_splits(N), chunk = cdiv(N, S);
  const int kmax = K <= 8 ? 8 : (K <= 16 ? 16 : 32);
  float *pd = (float *)scratch;
  int *pi = (int *)(pd + (size_t)S * kmax * N);
  const dim3 grid(cdiv(N, 256), S);
#define EG_KNN_SMALL(KM)                                                                                     \
  do {                                                                                                       \
    knn_exhaustive_kernel<KM><<<grid, 256, 0, st>>>(points, N, chunk, pd, pi);                                 \
    knn_merge_kernel<KM><<<cdiv(N, 256), 256, 0, st>>>(pd, pi, N, K, S, out_idx, out_d2);                    \
  } while (0)
  if (kmax == 8) EG_KNN_SMALL(8);
  else if (kmax == 16) EG_KNN_SMALL(16);
  else EG_KNN_SMALL(32);
#undef EG_KNN_SMALL
  return check_launch("knn_small");
}

extern "C" int eg_direction_loss(const float *means, const float *quats, const float *log_scales,
                                 const int32_t *nn_idx /*[N,K]*/, int32_t N, int32_t K, int32_t top_k,
                                 float *g_means /*[N,3] accumulated*/, float *g_quats /*[N,4] written*/,
                                 float *sum_out /*[1] accumulated*/, eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && K >= 1 && K <= kMaxDirNN, "bad sizes (K <= 32)");
  if (N == 0) return EG_OK;
  EG_REQUIRE(means && quats && log_scales && nn_idx && g_means && g_quats && sum_out, "null pointer");
  direction_loss_kernel<<<cdiv(N, 256), 256, 0, as_stream(stream)>>>(means, quats, log_scales, nn_idx, N, K, top_k,
                                                                   g_means, g_quats, sum_out);
  return check_launch("direction_loss");
}

extern "C" int eg_ratio_loss(const float *log_scales, int32_t N, float *g_scales /*[N,3] written*/,
                             float *sum_out /*[1] accumulated*/, eg_stream_t stream) {
  EG_REQUIRE(N >= 0, "bad sizes");
  if (N == 0) return EG_OK;
  EG_REQUIRE(log_scales && g_scales && sum_out, "null pointer");
  ratio_loss_kernel<<<cdiv(N, 256), 256, 0, as_stream(stream)>>>(log_scales, N, g_scales, sum_out);
  return check_launch("ratio_loss");
}
