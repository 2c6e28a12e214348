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

// squared distance with a FIXED rounding sequence: the grid search and the exhaustive search must order
// near-equidistant candidates identically (the compiler otherwise contracts the sum differently per kernel)
__device__ __forceinline__ float dist2(float ex, float ey, float ez) {
  return __fmaf_rn(ez, ez, __fmaf_rn(ey, ey, __fmul_rn(ex, ex)));
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
            knn_insert<KMAX>(bd, bi, dist2(ex, ey, ez), j);
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
        knn_insert<KMAX>(bd, bi, dist2(ex, ey, ez), j);
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
// whole wave for ~800 us).  N^2 distance evaluations are cheap IF the K-best bookkeeping stays off the common path:
// with one query per lane, some lane of the wave inserts at nearly every candidate and the whole wave pays the
// ~65-instruction sorted insertion every time (measured: 0.5 ms at N = 10 k).  So the roles are turned round:
// a LANE holds a CANDIDATE, a wave owns Q queries (coordinates and the K-th distance so far are wave-uniform),
// and each query's K-best list is spread over the lanes -- lane k holds the k-th best (distance, index).  A
// candidate enters the list only if it beats the K-th entry (a ballot; ~K ln(N/K) times per query in total), and
// then in O(1): every lane compares the newcomer with its own entry and its left neighbour's (one DPP shift) and
// keeps, takes the newcomer, or takes the neighbour's.  ~11 VALU instructions per (query, 64 candidates) otherwise.
// Same (distance, index) order as the grid search; K <= 32 <= 64 lanes.
template <int Q>
__global__ void __launch_bounds__(256)
knn_wave_kernel(const float *__restrict__ pts, int N, int K, int *__restrict__ out_idx, float *__restrict__ out_d2) {
  const int lane = threadIdx.x & 63;
  const int q0 = __builtin_amdgcn_readfirstlane((blockIdx.x * 4 + (threadIdx.x >> 6)) * Q);
  if (q0 >= N) return;
  float qx[Q], qy[Q], qz[Q], tau[Q], ld[Q];
  int lj[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int qi = min(q0 + q, N - 1);
    qx[q] = pts[3 * qi]; qy[q] = pts[3 * qi + 1]; qz[q] = pts[3 * qi + 2];
    tau[q] = 3.0e38f; ld[q] = 3.0e38f; lj[q] = -1;
  }
  // Candidate blocks of 64 rows are visited OUTWARD from the queries' own block (b0, b0+1, b0-1, b0+2, ...; wrapping):
  // when the rows are in spatial (Morton) order -- EdgeTrainer(spatial_order=True) -- the true neighbours come
  // first, the K-th distance is tight after two or three blocks and almost nothing enters the lists afterwards.
  // (A plain 0..N scan of spatially sorted rows is the worst case: the candidates close in on the query, and
  // nearly every one of them beats the K-th so far.)
  const int nb = (N + 63) >> 6, b0 = q0 >> 6;
  int cn = min(b0 * 64 + lane, N - 1);
  float cx = pts[3 * cn], cy = pts[3 * cn + 1], cz = pts[3 * cn + 2];
  int c0 = b0 * 64;
  for (int it = 0; it < nb; ++it) {
    const int c = c0 + lane;
    const float x = cx, y = cy, z = cz;
    const int off = (it + 2) >> 1;  // block of the NEXT round: in flight while this one is evaluated
    int bn = ((it + 1) & 1) ? b0 + off : b0 - off;
    bn += (bn < 0) ? nb : 0;
    bn -= (bn >= nb) ? nb : 0;
    const int c0_next = bn * 64;
    cn = min(c0_next + lane, N - 1);
    cx = pts[3 * cn]; cy = pts[3 * cn + 1]; cz = pts[3 * cn + 2];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const float ex = x - qx[q], ey = y - qy[q], ez = z - qz[q];
      const float d = dist2(ex, ey, ez);
      unsigned long long mask = __ballot(c < N && d <= tau[q] && c != q0 + q);
      while (mask) {
        const int l = __builtin_ctzll(mask);
        const float dn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), l));
        const int jn = c0 + l;
        // left neighbour's entry (lane 0 sees (-inf, -1): the newcomer never sorts before it)
        const float pd = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-3.0e38f), __float_as_int(ld[q]),
                                                                    0x138 /* wave_shr:1 */, 0xf, 0xf, false));
        const int pj = __builtin_amdgcn_update_dpp(-1, lj[q], 0x138, 0xf, 0xf, false);
        const bool lt_mine = dn < ld[q] || (dn == ld[q] && jn < lj[q]);
        const bool lt_prev = dn < pd || (dn == pd && jn < pj);
        ld[q] = lt_prev ? pd : (lt_mine ? dn : ld[q]);
        lj[q] = lt_prev ? pj : (lt_mine ? jn : lj[q]);
        tau[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ld[q]), K - 1));
        mask &= mask - 1;
        mask &= __ballot(d <= tau[q]);
      }
    }
    c0 = c0_next;
  }
#pragma unroll
  for (int q = 0; q < Q; ++q)
    if (q0 + q < N && lane < K) {
      out_idx[(size_t)(q0 + q) * K + lane] = lj[q];
      if (out_d2) out_d2[(size_t)(q0 + q) * K + lane] = ld[q];
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
                      const float *__restrict__ log_scales, const int *__restrict__ nn, int nn_stride, int N, int K,
                      int top_k, float *__restrict__ g_means, float *__restrict__ g_quats,
                      float *__restrict__ sum_out) {
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
        const int j = nn[(size_t)i * nn_stride + k];
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
      const int j = nn[(size_t)i * nn_stride + k];
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


// The tail of a regulariser iteration (train_gaussians.py:108-131) on the device: loss value from the kernel's sum,
// lambda = (running projection-loss sum) * factor / loss, gradients scaled in place.  Same float32 operation
// order as the tensor expressions it replaces: loss = 1 + w * sum (direction, w = -1 / (N k)) or sum / N (ratio);
// g = (raw * w) * lambda or (raw / N) * lambda.
__global__ void __launch_bounds__(256)
regulariser_scale_kernel(float *__restrict__ g, size_t n, const float *__restrict__ sum,
                         const float *__restrict__ loss_sum_dev, float loss_sum_host, float factor, float w,
                         float n_gauss, int ratio, float *__restrict__ loss_out) {
  const float loss = ratio ? sum[0] / n_gauss : 1.0f + w * sum[0];
  const float lam = (loss_sum_dev ? loss_sum_dev[0] * factor : loss_sum_host * factor) / loss;
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i == 0 && loss_out) loss_out[0] = loss;
  if (i < n) g[i] = (ratio ? g[i] / n_gauss : g[i] * w) * lam;
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

extern "C" int64_t eg_knn_small_scratch_bytes(int32_t N, int32_t K) {
  (void)N; (void)K;
  return 0;  // (the lists live in registers; kept in the ABI for callers that size a buffer)
}

extern "C" int eg_knn_small(const float *points, int32_t N, int32_t K, void *scratch, int32_t *out_idx,
                            float *out_d2, eg_stream_t stream) {
  (void)scratch;
  EG_REQUIRE(N >= 0 && K >= 1 && K <= 32, "bad arguments");
  if (N == 0) return EG_OK;
  EG_REQUIRE(points && out_idx, "null pointer");
  EG_REQUIRE(N <= (1 << 17), "eg_knn_small: N <= 131072 (use the grid search, eg_knn)");
  hipStream_t st = as_stream(stream);
  // queries per wave: more of them amortise the candidate stream, fewer give the chip more waves
  if (N >= 16384)
    knn_wave_kernel<8><<<cdiv(N, 32), 256, 0, st>>>(points, N, K, out_idx, out_d2);
  else if (N >= 4096)
    knn_wave_kernel<4><<<cdiv(N, 16), 256, 0, st>>>(points, N, K, out_idx, out_d2);
  else
    knn_wave_kernel<2><<<cdiv(N, 8), 256, 0, st>>>(points, N, K, out_idx, out_d2);
  return check_launch("knn_small");
}

extern "C" int eg_direction_loss(const float *means, const float *quats, const float *log_scales,
                                 const int32_t *nn_idx /*[N,K]*/, int32_t N, int32_t K, int32_t top_k,
                                 float *g_means /*[N,3] accumulated*/, float *g_quats /*[N,4] written*/,
                                 float *sum_out /*[1] accumulated*/, eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && K >= 1 && K <= kMaxDirNN, "bad sizes (K <= 32)");
  if (N == 0) return EG_OK;
  EG_REQUIRE(means && quats && log_scales && nn_idx && g_means && g_quats && sum_out, "null pointer");
  direction_loss_kernel<<<cdiv(N, 256), 256, 0, as_stream(stream)>>>(means, quats, log_scales, nn_idx, K, N, K, top_k,
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

// One regulariser iteration of train_gaussians.py:108-131 as ONE native enqueue: zero the gradient blocks, loss
// kernel (raw gradients + sum), lambda and scaling on the device, Adam step of the means / scales / quats optimizers
// (hyper.group_steps[3] < 0: the opacity optimizer does not step).  grads: the [11 N] block layout of eg_train_step
// (means 3N | quats 4N | scales 3N | opacities N).  kind 0 = direction (nn: [N, nn_stride] neighbour table, the K
// columns from nn_offset on are used; top_k as eg_direction_loss), 1 = ratio.  loss_sum: device scalar (the running
// projection-loss sum) or NULL -> loss_sum_host.  work: 2 floats of scratch (sum, loss value = work[1] afterwards).
extern "C" int eg_regulariser_step(int32_t kind, float *means, float *quats, float *log_scales,
                                   float *logit_opacities, float *adam_m, float *adam_v, float *grads, int32_t N,
                                   const int32_t *nn, int32_t nn_stride, int32_t nn_offset, int32_t K, int32_t top_k,
                                   const float *loss_sum, float loss_sum_host, float scale_factor, float *work,
                                   eg_adam_hyper hyper, eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && (kind == 0 || kind == 1), "bad arguments");
  if (N == 0) return EG_OK;
  EG_REQUIRE(means && quats && log_scales && logit_opacities && adam_m && adam_v && grads && work, "null pointer");
  hipStream_t st = as_stream(stream);
  float *gm = grads, *gq = grads + 3 * (size_t)N, *gs = grads + 7 * (size_t)N, *go = grads + 10 * (size_t)N;
  if (hipMemsetAsync(grads, 0, sizeof(float) * 11 * (size_t)N, st) != hipSuccess ||
      hipMemsetAsync(work, 0, sizeof(float) * 2, st) != hipSuccess)
    return check_launch("regulariser_step memset");
  if (kind == 0) {
    EG_REQUIRE(nn && K >= 1 && K <= kMaxDirNN && nn_offset >= 0 && nn_offset + K <= nn_stride, "bad neighbour table");
    direction_loss_kernel<<<cdiv(N, 256), 256, 0, st>>>(means, quats, log_scales, nn + nn_offset, nn_stride, N, K,
                                                        top_k, gm, gq, work);
    const int used = (top_k > 0 && top_k < K) ? top_k : K;
    // means and quats are adjacent blocks: one scaling launch over both
    regulariser_scale_kernel<<<cdiv(7 * (int64_t)N, 256), 256, 0, st>>>(
        gm, 7 * (size_t)N, work, loss_sum, loss_sum_host, scale_factor, (float)(-1.0 / ((double)N * used)), (float)N, 0,
        work + 1);
  } else {
    ratio_loss_kernel<<<cdiv(N, 256), 256, 0, st>>>(log_scales, N, gs, work);
    regulariser_scale_kernel<<<cdiv(3 * (int64_t)N, 256), 256, 0, st>>>(gs, 3 * (size_t)N, work, loss_sum, loss_sum_host,
                                                                      scale_factor, 0.f, (float)N, 1, work + 1);
  }
  int rc = check_launch("regulariser_step");
  if (rc) return rc;
  return eg_adam_multi(means, log_scales, quats, logit_opacities, gm, gs, gq, go, adam_m, adam_v, N, hyper, nullptr,
                       nullptr, stream);
}
