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

// The bounding box on the device (eg_knn_auto: no host-side look at the points), ONE launch: floats are compared as
// order-preserving unsigned integers; the scratch words hold max(~u) for the minima and max(u) for the maxima, so that
// an all-zero scratch is the identity (the caller zeroes it once; the last workgroup -- device-scope ticket -- turns
// the box into the grid parameters and hands the words back zeroed).  mm: [6] extrema, [6] ticket.
__device__ __forceinline__ unsigned ordered_u(float f) {
  const unsigned b = (unsigned)__float_as_int(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float float_of_u(unsigned u) {
  return __int_as_float((int)((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u));
}

// D x D x D cubic cells over the box (inflated by 0.1 %: the maxima must land inside the last cell)
__global__ void __launch_bounds__(256)
knn_bbox_kernel(const float *__restrict__ pts, int N, unsigned *__restrict__ mm, int D, Grid *__restrict__ out) {
  unsigned lo[3] = {0u, 0u, 0u}, hi[3] = {0u, 0u, 0u};  // lo holds max(~u)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const unsigned u = ordered_u(pts[3 * i + k]);
      lo[k] = max(lo[k], ~u);
      hi[k] = max(hi[k], u);
    }
  // wave -> workgroup -> six atomics per workgroup (same-address atomics serialise at ~60 ns each: one per WAVE of
  // a 1024-workgroup grid cost 280 us)
  __shared__ unsigned s_lo[4][3], s_hi[4][3];
  __shared__ int s_last;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      lo[k] = max(lo[k], (unsigned)__shfl_xor((int)lo[k], d, 64));
      hi[k] = max(hi[k], (unsigned)__shfl_xor((int)hi[k], d, 64));
    }
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6][k] = lo[k]; s_hi[threadIdx.x >> 6][k] = hi[k]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int k = threadIdx.x;
    atomicMax(&mm[k], max(max(s_lo[0][k], s_lo[1][k]), max(s_lo[2][k], s_lo[3][k])));
    atomicMax(&mm[3 + k], max(max(s_hi[0][k], s_hi[1][k]), max(s_hi[2][k], s_hi[3][k])));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the extrema are at the L2 before this workgroup's ticket is
  __syncthreads();
  if (threadIdx.x == 0) s_last = (atomicAdd((int *)&mm[6], 1) == (int)gridDim.x - 1);
  __syncthreads();
  if (!s_last || threadIdx.x != 0) return;
  float lof[3], ext = 1e-6f;
  for (int k = 0; k < 3; ++k) {
    const unsigned l = ~__hip_atomic_load(&mm[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned h = __hip_atomic_load(&mm[3 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    lof[k] = float_of_u(l);
    ext = fmaxf(ext, float_of_u(h) - lof[k]);
    mm[k] = 0u;
    mm[3 + k] = 0u;
  }
  mm[6] = 0u;
  Grid g;
  g.cell = ext * 1.002f / (float)D;
  g.inv_cell = 1.f / g.cell;
  g.ox = lof[0] - 0.001f * ext; g.oy = lof[1] - 0.001f * ext; g.oz = lof[2] - 0.001f * ext;
  g.nx = g.ny = g.nz = D;
  *out = g;
}

__global__ void __launch_bounds__(256)
knn_count_kernel(const float *__restrict__ pts, int N, Grid g_, const Grid *__restrict__ gp, int *__restrict__ cell_of,
                 int *__restrict__ counts) {
  const Grid g = gp ? *gp : g_;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int3 c = cell_of_point(g, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
  const int id = (c.z * g.ny + c.y) * g.nx + c.x;
  cell_of[i] = id;
  atomicAdd(&counts[id], 1);
}

// Exclusive scan of the cell populations over several workgroups (64 k cells took 91 us in one workgroup): a
// workgroup scans 4096 cells locally, then waits for the running total of the workgroups before it -- a chain
// through `carry` (one 64-bit word per workgroup: flag << 32 | total, zeroed by the caller), each link a
// device-scope store / load; lower block indices never wait on higher ones.
constexpr int kScanPer = 16;  // cells per thread
__global__ void __launch_bounds__(256)
knn_scan_kernel(const int *__restrict__ counts, int C, int *__restrict__ start, unsigned long long *carry) {
  __shared__ int s_tmp[4];
  __shared__ int s_carry;
  const int base = (blockIdx.x * 256 + threadIdx.x) * kScanPer;
  int v[kScanPer], sum = 0;
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    v[k] = base + k < C ? counts[base + k] : 0;
    sum += v[k];
  }
  int total;
  int excl = block_excl_scan<256>(sum, s_tmp, total);
  if (threadIdx.x == 0) {
    int before = 0;
    if (blockIdx.x > 0) {
      unsigned long long w;
      while (((w = __hip_atomic_load(&carry[blockIdx.x - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) == 0)
        __builtin_amdgcn_s_sleep(1);
      before = (int)(unsigned)w;
    }
    __hip_atomic_store(&carry[blockIdx.x], (1ull << 32) | (unsigned)(before + total), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    s_carry = before;
  }
  __syncthreads();
  excl += s_carry;
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    if (base + k < C) start[base + k] = excl;
    excl += v[k];
  }
  if (base <= C && C < base + kScanPer) start[C] = excl - 0;  // (every v[k] past C is zero: excl is the grand total)
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

// ---------------------------------------------------------------------------------------------
// The grid search with the lane <-> candidate mapping of knn_wave_kernel (below): a wavefront answers Q cell-ordered
// queries one after the other; the rows of the query's block of cells are dealt to groups of lanes (7 lanes per row
// for the 9 rows of the first block, 4 per row for 16 rows at a time afterwards), every lane evaluates one
// candidate per round, and the query's K-best list lives spread over the lanes (one DPP shift per insertion).
// With one query per lane (knn_query_kernel) the 56-instruction sorted insertion runs for the whole wave whenever
// ANY of its 64 queries inserts -- nearly always -- and its cost doubles with every step of KMAX: K = 11 (Replica's
// dir_loss_num_nn = 10) took 3.6-4.8 ms on 500 k points against 0.8-1.1 ms for K = 6.  Here an insertion costs the
// same for every K <= 64.
struct WaveList {
  float d, tau, bound;  // this lane's entry; K-th distance of the list (uniform); upper bound carried across blocks
  int j;
};

__device__ __forceinline__ void wavelist_reset(WaveList &w, float bound) {
  w.d = 3.0e38f; w.j = -1; w.tau = 3.0e38f; w.bound = bound;
}

// offer one candidate per lane (d, j; valid = this lane holds one): those that beat the K-th entry enter, in lane order
__device__ __forceinline__ void wavelist_offer(WaveList &w, float d, int j, bool valid, int K) {
  unsigned long long mask = __ballot(valid && d <= fminf(w.tau, w.bound));
  while (mask) {
    const int l = __builtin_ctzll(mask);
    const float dn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), l));
    const int jn = __builtin_amdgcn_readlane(j, l);
    const float pd = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-3.0e38f), __float_as_int(w.d),
                                                                0x138 /* wave_shr:1 */, 0xf, 0xf, false));
    const int pj = __builtin_amdgcn_update_dpp(-1, w.j, 0x138, 0xf, 0xf, false);
    const bool lt_mine = dn < w.d || (dn == w.d && jn < w.j);
    const bool lt_prev = dn < pd || (dn == pd && jn < pj);
    w.d = lt_prev ? pd : (lt_mine ? dn : w.d);
    w.j = lt_prev ? pj : (lt_mine ? jn : w.j);
    w.tau = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w.d), K - 1));
    mask &= mask - 1;
    mask &= __ballot(d <= fminf(w.tau, w.bound));
  }
}

template <int Q>
__global__ void __launch_bounds__(256)
knn_grid_wave_kernel(int N, int K, Grid g_, const Grid *__restrict__ gp, const int *__restrict__ cell_start,
                     const float4 *__restrict__ sorted, int *__restrict__ out_idx, float *__restrict__ out_d2,
                     int r_brute, float *kth /*[N] or NULL: see eg_knn_auto*/, float kth_slack) {
  const Grid g = gp ? *gp : g_;
  const int lane = threadIdx.x & 63;
  const int q0 = __builtin_amdgcn_readfirstlane((blockIdx.x * 4 + (threadIdx.x >> 6)) * Q);
  const int rmax = max(max(g.nx, g.ny), g.nz);
  for (int q = 0; q < Q && q0 + q < N; ++q) {
    const float4 me = sorted[q0 + q];  // (the same address in every lane)
    const int i = __float_as_int(me.w);
    const int3 c = cell_of_point(g, me.x, me.y, me.z);
    WaveList w;
    // temporal coherence: the K-th squared distance this point had in the caller's previous search (inflated) is the
    // entry bound from the first candidate on -- ~K insertions instead of ~K ln(n / K).  Only a filter: should the
    // point's neighbourhood have thinned out, fewer than K candidates pass, the K-th distance stays infinite, the
    // block is not settled and the re-scan runs without the bound.
    const float bound0 = kth ? kth[i] * kth_slack : 3.0e38f;
    wavelist_reset(w, bound0 > 0.f ? bound0 : 3.0e38f);
    bool settled = false;
    for (int r = 1; !settled;) {
      if (r > r_brute && r < rmax) break;
      const int x0 = max(c.x - r, 0), x1 = min(c.x + r, g.nx - 1);
      const int y0 = max(c.y - r, 0), ny = min(c.y + r, g.ny - 1) - y0 + 1;
      const int z0 = max(c.z - r, 0), nz = min(c.z + r, g.nz - 1) - z0 + 1;
      const int nrows = ny * nz;
      // rows per round and lanes per row: the 9 rows of the first block get 7 lanes each, later blocks 4 lanes x 16 rows
      const int G = (nrows <= 9) ? 9 : 16, LPR = (nrows <= 9) ? 7 : 4;
      const int rho = (nrows <= 9) ? (lane * 37) >> 8 : lane >> 2;  // lane / LPR (exact for lane < 64)
      const int u = lane - rho * LPR;
      for (int row0 = 0; row0 < nrows; row0 += G) {
        int s = 0, e = 0;
        // the query's own row first (lane order = insertion order: the nearest candidates tighten the K-th distance
        // before the far rows are looked at)
        const int ci = (c.z - z0) * ny + (c.y - y0);
        int rr = row0 + rho;
        if (row0 == 0 && rr < min(G, nrows)) { rr += ci % min(G, nrows); rr -= rr >= min(G, nrows) ? min(G, nrows) : 0; }
        if (rho < G && rr < nrows) {
          const int zz = rr / ny, yy = rr - zz * ny;
          const int row = ((z0 + zz) * g.ny + (y0 + yy)) * g.nx;
          s = cell_start[row + x0];
          e = cell_start[row + x1 + 1];
        }
        int longest = e - s;
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) longest = max(longest, __shfl_xor(longest, dd, 64));
        float4 c4 = sorted[min(s + u, N - 1)];
        for (int k = u; k < longest + u; k += LPR) {  // (uniform trip count: `longest` is)
          const float4 cur = c4;
          const int idx = s + k;
          c4 = sorted[min(idx + LPR, N - 1)];  // the next round's candidate is in flight
          const int j = __float_as_int(cur.w);
          const float ex = cur.x - me.x, ey = cur.y - me.y, ez = cur.z - me.z;
          wavelist_offer(w, dist2(ex, ey, ez), j, idx < e && j != i, K);
        }
      }
      // everything outside the (2r+1)^3 block is at least r * cell away
      const float reach = (float)r * g.cell;
      settled = (w.tau <= reach * reach) || (r >= rmax);
      if (!settled) {
        // K candidates known: everything nearer than the K-th lies within sqrt(kth) of the query, ONE more block of
        // exactly that radius settles it; the block is scanned afresh with the K-th distance as the entry bound
        const int r_need = (w.tau < 1.0e38f) ? (int)ceilf(sqrtf(w.tau) * g.inv_cell) : 2 * r;
        r = min(max(r_need, r + 1), max(rmax, r + 1));
        wavelist_reset(w, w.tau);
      }
    }
    if (!settled) {  // an outlier far from everything: all records, 64 at a time
      wavelist_reset(w, w.bound);
      float4 c4 = sorted[min(lane, N - 1)];
      for (int s0 = 0; s0 < N; s0 += 64) {
        const float4 cur = c4;
        c4 = sorted[min(s0 + 64 + lane, N - 1)];
        const int j = __float_as_int(cur.w);
        const float ex = cur.x - me.x, ey = cur.y - me.y, ez = cur.z - me.z;
        wavelist_offer(w, dist2(ex, ey, ez), j, s0 + lane < N && j != i, K);
      }
    }
    if (lane < K) {
      out_idx[(size_t)i * K + lane] = w.j;
      if (out_d2) out_d2[(size_t)i * K + lane] = w.d;
      if (kth && lane == K - 1) kth[i] = w.j >= 0 ? w.d : 0.f;  // (0 = unknown: fewer than K other points)
    }
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
// sum of `acc` over a 256-thread workgroup, ONE atomic per workgroup (a same-address atomic per wavefront -- 1563 of
// them at 100 k Gaussians -- serialises at the L2: 15 of the 29 us of the ratio-loss kernel)
__device__ __forceinline__ void block_sum_add(float acc, float *__restrict__ sum_out) {
  __shared__ float s_part[4];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    if (t != 0.f) unsafeAtomicAdd(sum_out, t);
  }
}

// ---- ORDER-INDEPENDENT accumulation (data-parallel runs: replicas must stay bit-identical).  Float atomics
// add in the order the hardware happens to take them; 64-bit FIXED-POINT integer atomics (2^-32 resolution, |x| < 2^31)
// are associative: every rank quantises the same fp32 terms to the same integers and the sums are equal whatever the
// order.  The accumulators live in a caller-supplied int64 scratch [3 N + 1] (gradient of the means, then the loss sum),
// zero on entry; fixed_finish_kernel converts them to the float gradient block / sum and hands the scratch back zeroed.
constexpr float kFixedScale = 4294967296.f;               // 2^32
constexpr double kFixedInv = 2.3283064365386963e-10;     // 2^-32
__device__ __forceinline__ unsigned long long to_fixed(float x) { return (unsigned long long)__float2ll_rn(x * kFixedScale); }
__device__ __forceinline__ void block_sum_add_fixed(float acc, unsigned long long *__restrict__ sum_out) {
  __shared__ unsigned long long s_part64[4];
  unsigned long long v = to_fixed(acc);  // (quantised per thread: integer sums from here on)
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  if ((threadIdx.x & 63) == 0) s_part64[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = s_part64[0] + s_part64[1] + s_part64[2] + s_part64[3];
    if (t != 0ull) atomicAdd(sum_out, t);
  }
}
__global__ void __launch_bounds__(256)
fixed_finish_kernel(unsigned long long *__restrict__ fixed, long long n, float *__restrict__ out, float *__restrict__ sum_out) {
  const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (e > n) return;
  const long long v = (long long)fixed[e];
  fixed[e] = 0ull;
  const float f = (float)((double)v * kFixedInv);
  if (e < n) { if (out) out[e] = f; }
  else *sum_out = f;  // (element n: the loss sum)
}

constexpr int kMaxDirNN = 32;
template <bool FIXED>
__global__ void __launch_bounds__(256)
direction_loss_kernel(const float *__restrict__ means, const float *__restrict__ quats,
                      const float *__restrict__ log_scales, const int *__restrict__ nn, int nn_stride, int N, int K,
                      int top_k, float *__restrict__ g_means, float *__restrict__ g_quats,
                      float *__restrict__ sum_out, unsigned long long *__restrict__ fixed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  // Gradients w.r.t. the means of the workgroup's OWN 256 rows are collected in LDS and flushed with three global
  // atomics per row; with the rows in spatial order most neighbours of a row sit in the same workgroup, so the 18
  // device-scope float atomics per Gaussian (15 scattered to neighbours, 3 to itself) drop to 3 + the neighbours
  // that live in other workgroups
  __shared__ float s_g[FIXED ? 1 : 256 * 3];
  __shared__ unsigned long long s_g64[FIXED ? 256 * 3 : 1];
  const int b0 = blockIdx.x * 256;
  if (FIXED) { s_g64[threadIdx.x] = 0ull; s_g64[256 + threadIdx.x] = 0ull; s_g64[512 + threadIdx.x] = 0ull; }
  else { s_g[threadIdx.x] = 0.f; s_g[256 + threadIdx.x] = 0.f; s_g[512 + threadIdx.x] = 0.f; }
  __syncthreads();
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
      const unsigned jl = (unsigned)(j - b0);
      if (FIXED) {
        if (jl < 256u) {
          atomicAdd(&s_g64[3 * jl], to_fixed(-gx));
          atomicAdd(&s_g64[3 * jl + 1], to_fixed(-gy));
          atomicAdd(&s_g64[3 * jl + 2], to_fixed(-gz));
        } else {
          atomicAdd(&fixed[3 * (size_t)j], to_fixed(-gx));
          atomicAdd(&fixed[3 * (size_t)j + 1], to_fixed(-gy));
          atomicAdd(&fixed[3 * (size_t)j + 2], to_fixed(-gz));
        }
        // (this row's own term, neighbour by neighbour: a float sum over k first would be deterministic as well -- one
        // thread, fixed order -- but then the quantised terms of the two sides of an edge would no longer cancel exactly)
        atomicAdd(&s_g64[3 * threadIdx.x], to_fixed(gx));
        atomicAdd(&s_g64[3 * threadIdx.x + 1], to_fixed(gy));
        atomicAdd(&s_g64[3 * threadIdx.x + 2], to_fixed(gz));
      } else if (jl < 256u) {
        atomicAdd(&s_g[3 * jl], -gx);
        atomicAdd(&s_g[3 * jl + 1], -gy);
        atomicAdd(&s_g[3 * jl + 2], -gz);
      } else {
        unsafeAtomicAdd(&g_means[3 * j], -gx);
        unsafeAtomicAdd(&g_means[3 * j + 1], -gy);
        unsafeAtomicAdd(&g_means[3 * j + 2], -gz);
      }
    }
    if (!FIXED) {
      atomicAdd(&s_g[3 * threadIdx.x], vpx);
      atomicAdd(&s_g[3 * threadIdx.x + 1], vpy);
      atomicAdd(&s_g[3 * threadIdx.x + 2], vpz);
    }
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
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 3; ++k) {  // coalesced flush of the workgroup's 768 accumulators
    const int e = k * 256 + threadIdx.x;
    if (3 * (size_t)b0 + e >= 3 * (size_t)N) continue;
    if (FIXED) {
      const unsigned long long v = s_g64[e];
      if (v != 0ull) atomicAdd(&fixed[3 * (size_t)b0 + e], v);
    } else {
      const float v = s_g[e];
      if (v != 0.f) unsafeAtomicAdd(&g_means[3 * (size_t)b0 + e], v);
    }
  }
  if (FIXED) block_sum_add_fixed(acc, &fixed[3 * (size_t)N]);
  else block_sum_add(acc, sum_out);
}

// ratio loss (edge_gs.py:375-380): mean_i second-largest / largest scale.  Value (sum of ratios) and
// unscaled gradient w.r.t. the LOG-scales: d r / d ls_second = r, d r / d ls_first = -r.
__global__ void __launch_bounds__(256)
ratio_loss_kernel(const float *__restrict__ log_scales, int N, float *__restrict__ g_scales,
                  float *__restrict__ sum_out, unsigned long long *__restrict__ fixed_sum) {
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
  if (fixed_sum) block_sum_add_fixed(acc, fixed_sum);  // (uniform: the order-independent sum of a data-parallel run)
  else block_sum_add(acc, sum_out);
}


}  // namespace eg

using namespace eg;

// count -> scan -> scatter (points in cell order) -> wave-cooperative query; grid by value (host) or by pointer (device)
static int knn_grid_search(const float *points, int32_t N, int32_t K, const Grid &g, const Grid *gp, int C,
                           int32_t *cell_of, int32_t *cell_counts, int32_t *cell_start, float *sorted,
                           int32_t *out_idx, float *out_d2, eg_stream_t stream, float *kth = nullptr,
                           float kth_slack = 0.f) {
  hipStream_t st = as_stream(stream);
  // a block of radius r costs (2r+1)^2 row look-ups, the exhaustive scan N / 64 rounds of ~15 instructions:
  // beyond this radius the scan is the cheaper way to settle an outlier
  int r_brute = 1;
  while ((2 * (2 * r_brute) + 1) * (2 * (2 * r_brute) + 1) * 15 < N) r_brute *= 2;
  knn_count_kernel<<<cdiv(N, 256), 256, 0, st>>>(points, N, g, gp, cell_of, cell_counts);
  // (the chain words of the scan live at the head of `sorted`, which the scatter kernel fills afterwards)
  const int scan_blocks = cdiv(C + 1, 256 * kScanPer);
  EG_REQUIRE((int64_t)scan_blocks * 8 <= (int64_t)N * 16, "grid too fine for the number of points");
  if (hipMemsetAsync(sorted, 0, sizeof(unsigned long long) * scan_blocks, st) != hipSuccess)
    return check_launch("knn scan memset");
  knn_scan_kernel<<<scan_blocks, 256, 0, st>>>(cell_counts, C, cell_start, (unsigned long long *)sorted);
  knn_scatter_kernel<<<cdiv(N, 256), 256, 0, st>>>(points, cell_of, N, cell_start, cell_counts, (float4 *)sorted);
  knn_grid_wave_kernel<4><<<cdiv(N, 16), 256, 0, st>>>(N, K, g, gp, cell_start, (const float4 *)sorted, out_idx, out_d2,
                                                       r_brute, kth, kth_slack);
  return check_launch("knn");
}

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
  return knn_grid_search(points, N, K, g, nullptr, g.nx * g.ny * g.nz, cell_of, cell_counts, cell_start, sorted, out_idx,
                         out_d2, stream);
}

// cells per axis of eg_knn_auto's grid: ~4 points per cell if they fill the cube (8 when more than 12 neighbours
// are asked for).  Measured at 100 k / 500 k points, uniform and trained-like, K = 6 / 11 / 21, medians in ms:
// 2 per cell 0.18-1.04 / 0.28-1.52 / 0.40-2.0; 4: 0.17-1.00 / 0.24-1.39 / 0.43-2.3; 8: 0.19-1.21 / 0.26-1.47 /
// 0.39-2.3; 16: 0.23-1.53 / 0.31-1.80 / 0.44-2.6 -- flat around 4 (fuller cells cost rounds, emptier ones re-scans)
extern "C" int32_t eg_knn_auto_dims(int32_t N, int32_t K) {
  const int per_cell = K <= 12 ? 4 : 8;
  int D = 1;
  while ((int64_t)D * D * D * per_cell < N && D < 256) ++D;
  return D;
}

extern "C" int eg_knn_auto(const float *points, int32_t N, int32_t K, int32_t *cell_of /*[N]*/,
                           int32_t *cell_counts /*[D^3] zero on entry and on exit*/, int32_t *cell_start /*[D^3+1]*/,
                           float *sorted /*[N,4]*/, void *grid_scratch /*64 bytes, zeroed once by the caller*/,
                           int32_t *out_idx,
                           float *out_d2, float *kth, float kth_slack, eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && K >= 1 && K <= 32, "bad arguments");
  if (N == 0) return EG_OK;
  EG_REQUIRE(points && cell_of && cell_counts && cell_start && sorted && grid_scratch && out_idx, "null pointer");
  const int D = eg_knn_auto_dims(N, K);
  hipStream_t st = as_stream(stream);
  unsigned *mm = (unsigned *)grid_scratch;  // 6 extrema + ticket (all zero between calls)
  Grid *gp = (Grid *)(mm + 8);              // 8 words, 32-byte aligned when the scratch is
  knn_bbox_kernel<<<min(cdiv(N, 2048), 128), 256, 0, st>>>(points, N, mm, D, gp);
  Grid g = {};
  g.nx = g.ny = g.nz = D;  // (only the pointer's copy is read by the kernels)
  return knn_grid_search(points, N, K, g, gp, D * D * D, cell_of, cell_counts, cell_start, sorted, out_idx, out_d2,
                         stream, kth, kth_slack);
}

extern "C" int eg_knn_small(const float *points, int32_t N, int32_t K, int32_t *out_idx, float *out_d2,
                            eg_stream_t stream) {
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
  direction_loss_kernel<false><<<cdiv(N, 256), 256, 0, as_stream(stream)>>>(means, quats, log_scales, nn_idx, K, N, K, top_k,
                                                                   g_means, g_quats, sum_out, nullptr);
  return check_launch("direction_loss");
}

extern "C" int eg_ratio_loss(const float *log_scales, int32_t N, float *g_scales /*[N,3] written*/,
                             float *sum_out /*[1] accumulated*/, eg_stream_t stream) {
  EG_REQUIRE(N >= 0, "bad sizes");
  if (N == 0) return EG_OK;
  EG_REQUIRE(log_scales && g_scales && sum_out, "null pointer");
  ratio_loss_kernel<<<cdiv(N, 256), 256, 0, as_stream(stream)>>>(log_scales, N, g_scales, sum_out, nullptr);
  return check_launch("ratio_loss");
}

// One regulariser iteration of train_gaussians.py:108-131 as ONE native enqueue: zero the gradient blocks, loss
// kernel (raw gradients + sum), lambda and scaling on the device, Adam step of the means / scales / quats optimizers
// (hyper.group_steps[3] < 0: the opacity optimizer does not step).  grads: the [11 N] block layout of eg_train_step
// (means 3N | quats 4N | scales 3N | opacities N).  kind 0 = direction (nn: [N, nn_stride] neighbour table, the K
// columns from nn_offset on are used; top_k as eg_direction_loss), 1 = ratio.  loss_sum: device scalar (the running
// projection-loss sum) or NULL -> loss_sum_host.  work: 2 floats of scratch (sum, loss value = work[1] afterwards).
// fixed != nullptr (eg_regulariser_step_fixed): the neighbour gradients and the loss sum are accumulated in 64-bit fixed
// point (order-independent: bit-identical on every rank of a data-parallel run) in that scratch, [3 N + 1], zero on entry
// and on exit.
static int regulariser_step_impl(int32_t kind, float *means, float *quats, float *log_scales,
                                 float *logit_opacities, float *adam_m, float *adam_v, float *grads, int32_t N,
                                 const int32_t *nn, int32_t nn_stride, int32_t nn_offset, int32_t K, int32_t top_k,
                                 const float *loss_sum, float loss_sum_host, float scale_factor, float *work,
                                 eg_adam_hyper hyper, eg_stream_t stream, unsigned long long *fixed) {
  EG_REQUIRE(N >= 0 && (kind == 0 || kind == 1), "bad arguments");
  if (N == 0) return EG_OK;
  EG_REQUIRE(means && quats && log_scales && logit_opacities && adam_m && adam_v && grads && work, "null pointer");
  hipStream_t st = as_stream(stream);
  float *gm = grads, *gq = grads + 3 * (size_t)N, *gs = grads + 7 * (size_t)N;
  // the loss kernel leaves RAW gradients and the sum of the per-Gaussian terms; the Adam kernel forms the loss value
  // and lambda from them and scales on the fly; the blocks a loss does not touch are never read (zero gradients)
  if (hipMemsetAsync(work, 0, sizeof(float) * 2, st) != hipSuccess) return check_launch("regulariser_step memset");
  int rc;
  if (kind == 0) {
    EG_REQUIRE(nn && K >= 1 && K <= kMaxDirNN && nn_offset >= 0 && nn_offset + K <= nn_stride, "bad neighbour table");
    if (fixed) {
      direction_loss_kernel<true><<<cdiv(N, 256), 256, 0, st>>>(means, quats, log_scales, nn + nn_offset, nn_stride, N, K,
                                                               top_k, gm, gq, work, fixed);
      fixed_finish_kernel<<<cdiv(3 * (int64_t)N + 1, 256), 256, 0, st>>>(fixed, 3 * (long long)N, gm, work);
    } else {
      if (hipMemsetAsync(gm, 0, sizeof(float) * 3 * (size_t)N, st) != hipSuccess)  // (accumulated with atomics)
        return check_launch("regulariser_step memset");
      direction_loss_kernel<false><<<cdiv(N, 256), 256, 0, st>>>(means, quats, log_scales, nn + nn_offset, nn_stride, N, K,
                                                                top_k, gm, gq, work, nullptr);
    }
    rc = check_launch("regulariser_step");
    if (rc) return rc;
    const int used = (top_k > 0 && top_k < K) ? top_k : K;
    return launch_adam_regulariser(means, log_scales, quats, logit_opacities, gm, nullptr, gq, adam_m, adam_v, N, hyper,
                                   work, loss_sum, loss_sum_host, scale_factor, (float)(-1.0 / ((double)N * used)), 0,
                                   work + 1, st);
  }
  ratio_loss_kernel<<<cdiv(N, 256), 256, 0, st>>>(log_scales, N, gs, work, fixed ? fixed + 3 * (size_t)N : nullptr);
  if (fixed) fixed_finish_kernel<<<1, 256, 0, st>>>(fixed + 3 * (size_t)N, 0, nullptr, work);
  rc = check_launch("regulariser_step");
  if (rc) return rc;
  return launch_adam_regulariser(means, log_scales, quats, logit_opacities, nullptr, gs, nullptr, adam_m, adam_v, N, hyper,
                                 work, loss_sum, loss_sum_host, scale_factor, 0.f, 1, work + 1, st);
}

extern "C" int eg_regulariser_step(int32_t kind, float *means, float *quats, float *log_scales,
                                   float *logit_opacities, float *adam_m, float *adam_v, float *grads, int32_t N,
                                   const int32_t *nn, int32_t nn_stride, int32_t nn_offset, int32_t K, int32_t top_k,
                                   const float *loss_sum, float loss_sum_host, float scale_factor, float *work,
                                   eg_adam_hyper hyper, eg_stream_t stream) {
  return regulariser_step_impl(kind, means, quats, log_scales, logit_opacities, adam_m, adam_v, grads, N, nn, nn_stride,
                               nn_offset, K, top_k, loss_sum, loss_sum_host, scale_factor, work, hyper, stream, nullptr);
}

extern "C" int eg_regulariser_step_fixed(int32_t kind, float *means, float *quats, float *log_scales,
                                         float *logit_opacities, float *adam_m, float *adam_v, float *grads, int32_t N,
                                         const int32_t *nn, int32_t nn_stride, int32_t nn_offset, int32_t K, int32_t top_k,
                                         const float *loss_sum, float loss_sum_host, float scale_factor, float *work,
                                         eg_adam_hyper hyper, int64_t *fixed, eg_stream_t stream) {
  EG_REQUIRE(fixed != nullptr, "null pointer");
  return regulariser_step_impl(kind, means, quats, log_scales, logit_opacities, adam_m, adam_v, grads, N, nn, nn_stride,
                               nn_offset, K, top_k, loss_sum, loss_sum_host, scale_factor, work, hyper, stream,
                               (unsigned long long *)fixed);
}
