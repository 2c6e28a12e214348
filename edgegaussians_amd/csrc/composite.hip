// G7 / G8: per-pixel alpha compositing forward and backward.
//
// Replaces gsplat 1.0.0 rasterize_to_pixels_{fwd,bwd} as reached from the reference call at
// edgegaussians/models/edge_gs.py:250-268, with the reference's clamp
// (edge_gs.py:279) and projection loss (edge_gs.py:288-324, losses.py:5-11, in weight-map form)
// fused into the forward epilogue.
//
// Work decomposition: the reference's scenes put nearly all Gaussians into the ~13x13 central tiles
// (the object spans ~200 px), so "one workgroup per tile" leaves >80 % of the 256 CUs idle while a
// few workgroups walk thousands of Gaussians serially (measured: 1.2 ms).  Both passes therefore
// run one 256-thread workgroup per ITEM = (tile, slice of <= 128 depth-sorted Gaussians); the item
// table is the second scan produced by eg_tile_offsets (an empty tile owns one empty item, which
// finalises its pixels).
//
// Forward (unit colours), one kernel + a (usually idle) fix-up:
//   slice  : item -> per-pixel product P = prod(1 - alpha) over the slice and the index of the last
//            contributing Gaussian; Gaussians staged through LDS as packed 32-byte records, inner loop
//            = two broadcast ds_read_b128 + ~15 VALU; a conservative sigma threshold (ln(255 o) +
//            margin) skips the exp for pairs that cannot reach alpha >= 1/255, the exact test follows.
//            The LAST workgroup of a tile to finish (per-tile ticket) combines: T = prod over the tile's
//            slices in depth order, fused clamp + weighted L1 + upstream gradient.  A tile with one slice
//            (most tiles at the reference's sizes) never writes its products at all.
//   re-walk: the transmittance stop (T <= 1e-4) is only DETECTED on the slice products; the slice in which
//            it falls is put on a compact list and resolved exactly by this kernel
// (front-to-back compositing is associative: (C1,T1) o (C2,T2) = (C1 + T1 C2, T1 T2); with unit
// colours C = 1 - T, so only T travels).  General colours use the classic one-workgroup-per-tile
// kernel below.
//
// Backward (unit colours -- the reference passes colours == 1, edge_gs.py:247): with c == 1 and no
// background, pix = 1 - T_final, hence dpix/dalpha_i = T_final / (1 - alpha_i) for EVERY contributing
// Gaussian: the pass is order-independent.  That allows the wave64-native transposition
//     lane = Gaussian, loop = pixels of the tile
// in which each lane accumulates its own Gaussian's 8 partial derivatives in registers: no
// cross-lane reduction at all (a 32-lane-warp design spends 5 shuffles x 9 values per Gaussian per
// warp here) and one set of atomics per (Gaussian, tile, pixel-split) instead of one per warp.
// Pixels with zero upstream gradient are compacted away first (the reference's `bg_edge_ratio`
// strategy leaves ~1.5 % of the pixels active).
//
// General colours keep the classic order-dependent pixel-per-lane backward with wave64 butterfly
// reductions (eg_composite_bwd_colors); it is off the reference's path and not tuned.
#include <cstdlib>

#include "common.h"
#include "composite.h"

namespace eg {


template <int CH, bool UNIT>
__global__ void __launch_bounds__(256)
composite_fwd_kernel(const float4 *__restrict__ splat, const float *__restrict__ colors,
                     const int *__restrict__ offsets, const int *__restrict__ flat, int width, int height,
                     int tw, int th, float *__restrict__ render, float *__restrict__ alphas,
                     int *__restrict__ last_ids, const float *__restrict__ gt, const float *__restrict__ wmap,
                     float loss_scale, float *__restrict__ vpix, float *__restrict__ loss_out) {
  __shared__ float4 sA[kTilePix];  // x, y, a, b
  __shared__ float4 sB[kTilePix];  // c, o, sigma threshold, -
  __shared__ float sC[UNIT ? 1 : kTilePix * CH];
  __shared__ float sRed[4];

  const int tile = xcd_tile(blockIdx.x, tw * th);
  const int tid = threadIdx.x;
  const int ty = tile / tw, tx = tile - ty * tw;
  const int i = ty * kTile + (tid >> 4), j = tx * kTile + (tid & 15);
  const bool inside = (i < height) && (j < width);
  const float px = (float)j + 0.5f, py = (float)i + 0.5f;
  const int start = offsets[tile], end = offsets[tile + 1];

  float T = 1.f;
  float pix[CH];
#pragma unroll
  for (int k = 0; k < CH; ++k) pix[k] = 0.f;
  int last = 0;
  bool done = !inside;

  for (int base = start; base < end; base += kTilePix) {
    if (__syncthreads_and(done)) break;
    const int idx = base + tid;
    if (idx < end) {
      const int g = flat[idx];
      const float4 s0 = splat[2 * g], s1 = splat[2 * g + 1];
      sA[tid] = s0;
      sB[tid] = make_float4(s1.x, s1.y, __logf(255.f * s1.y) + kThrMargin, 0.f);
      if (!UNIT) {
#pragma unroll
        for (int k = 0; k < CH; ++k) sC[tid * CH + k] = colors[(size_t)g * CH + k];
      }
    }
    __syncthreads();
    const int n = min(kTilePix, end - base);
    for (int t = 0; t < n && !done; ++t) {
      const float4 A = sA[t], B = sB[t];
      const float dx = A.x - px, dy = A.y - py;
      const float sigma = 0.5f * (A.z * dx * dx + B.x * dy * dy) + A.w * dx * dy;
      if (sigma < 0.f || sigma > B.z) continue;
      const float alpha = fminf(kAlphaMax, B.y * __expf(-sigma));
      if (alpha < kAlphaMin) continue;
      const float next_T = T * (1.f - alpha);
      if (next_T <= kTStop) { done = true; break; }
      const float w = alpha * T;
      if (UNIT) {
#pragma unroll
        for (int k = 0; k < CH; ++k) pix[k] += w;
      } else {
#pragma unroll
        for (int k = 0; k < CH; ++k) pix[k] += sC[t * CH + k] * w;
      }
      last = base + t;
      T = next_T;
    }
  }

  float l = 0.f;
  if (inside) {
    const int p = i * width + j;
    alphas[p] = 1.f - T;
    last_ids[p] = last;
#pragma unroll
    for (int k = 0; k < CH; ++k) render[(size_t)p * CH + k] = pix[k];
    if (wmap) {
      const float w = wmap[p];
      const float c0 = fminf(fmaxf(pix[0], 0.f), 1.f);
      const float d = c0 - gt[p];
      l = w * fabsf(d);
      const float pass = (pix[0] >= 0.f && pix[0] <= 1.f) ? 1.f : 0.f;
      const float sgn = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
      if (vpix) vpix[p] = loss_scale * w * sgn * pass;
    }
  }
  if (wmap && loss_out) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) l += __shfl_xor(l, d, 64);
    if ((tid & 63) == 0) sRed[tid >> 6] = l;
    __syncthreads();
    if (tid == 0) {
      const float s = sRed[0] + sRed[1] + sRed[2] + sRed[3];
      if (s != 0.f) unsafeAtomicAdd(loss_out, s);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward, unit colours: lane = Gaussian, loop over the tile's active pixels
__global__ void __launch_bounds__(256)
composite_bwd_unit_kernel(const float4 *__restrict__ splat, const int *__restrict__ offsets,
                          const int *__restrict__ flat, int width, int height, int tw, int th,
                          const float *__restrict__ alphas, const int *__restrict__ last_ids,
                          const float *__restrict__ vpix, float *__restrict__ g2d) {
  __shared__ float4 sP[kTilePix];  // px, py, v*T_final, last id (int bits) -- compacted
  __shared__ int sCnt[4];
  __shared__ int sMaxLast[4];

  const int tile = xcd_tile(blockIdx.x, tw * th);
  const int start = offsets[tile], end = offsets[tile + 1];
  const int n = end - start;
  if (n <= 0) return;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int ty = tile / tw, tx = tile - ty * tw;
  const int i = ty * kTile + (tid >> 4), j = tx * kTile + (tid & 15);
  const bool inside = (i < height) && (j < width);
  float gT = 0.f;
  int last = -1;
  if (inside) {
    const int p = i * width + j;
    const float a = alphas[p];
    const float v = vpix[p];
    if (a > 0.f && v != 0.f) {  // a > 0 <=> at least one Gaussian contributed to this pixel
      gT = v * (1.f - a);
      last = last_ids[p];
    }
  }
  const bool active = (last >= 0) && (gT != 0.f);
  const unsigned long long bal = __ballot(active);
  const int wave_cnt = __popcll(bal);
  const int rank = __popcll(bal & ((1ull << lane) - 1ull));
  int wmax = last;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, __shfl_xor(wmax, d, 64));
  if (lane == 0) { sCnt[wv] = wave_cnt; sMaxLast[wv] = wmax; }
  __syncthreads();
  int pre = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) pre += (w < wv) ? sCnt[w] : 0;
  const int n_act = sCnt[0] + sCnt[1] + sCnt[2] + sCnt[3];
  const int max_last = max(max(sMaxLast[0], sMaxLast[1]), max(sMaxLast[2], sMaxLast[3]));
  if (n_act == 0) return;
  if (active) sP[pre + rank] = make_float4((float)j + 0.5f, (float)i + 0.5f, gT, __int_as_float(last));
  __syncthreads();

  // work items = (64-Gaussian chunk) x (pixel split); only Gaussians up to max_last can contribute
  const int n_live = min(n, max_last - start + 1);
  const int n_chunks = (n_live + 63) >> 6;
  const int splits = (n_chunks >= 4) ? 1 : ((n_chunks == 2) ? 2 : 4);
  const int n_items = n_chunks * splits;
  for (int item = wv; item < n_items; item += 4) {
    const int chunk = item / splits, sp = item - chunk * splits;
    const int q0 = (n_act * sp) / splits, q1 = (n_act * (sp + 1)) / splits;
    const int idx = start + (chunk << 6) + lane;
    const bool have = idx < start + n_live;
    int g = 0;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (have) {
      g = flat[idx];
      s0 = splat[2 * g];
      s1 = splat[2 * g + 1];
    }
    const float x = s0.x, y = s0.y, ca = s0.z, cb = s0.w, cc = s1.x, o = s1.y;
    const float thr = have ? __logf(255.f * o) + kThrMargin : -1.f;
    float ax = 0.f, ay = 0.f, aax = 0.f, aay = 0.f, aa = 0.f, ab = 0.f, ac = 0.f, ao = 0.f;
    bool hit = false;
    for (int q = q0; q < q1; ++q) {
      const float4 P = sP[q];
      const float dx = x - P.x, dy = y - P.y;
      const float sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
      bool valid = (idx <= __float_as_int(P.w)) && (sigma >= 0.f) && (sigma <= thr);
      if (!__any(valid)) continue;
      const float vis = __expf(-sigma);
      const float araw = o * vis;
      const float alpha = fminf(kAlphaMax, araw);
      valid = valid && (alpha >= kAlphaMin);
      if (valid) {
        hit = true;
        const float v_alpha = P.z * __builtin_amdgcn_rcpf(1.f - alpha);  // dL/dalpha = v * T_final / (1 - alpha)
        if (araw <= kAlphaMax) {
          const float v_sigma = -araw * v_alpha;
          const float gx = v_sigma * (ca * dx + cb * dy);
          const float gy = v_sigma * (cb * dx + cc * dy);
          ax += gx; ay += gy;
          aax += fabsf(gx); aay += fabsf(gy);
          aa += 0.5f * v_sigma * dx * dx;
          ab += v_sigma * dx * dy;
          ac += 0.5f * v_sigma * dy * dy;
          ao += vis * v_alpha;
        }
      }
    }
    if (hit) {
      float *dst = g2d + (size_t)g * 8;
      unsafeAtomicAdd(dst + 0, ax);
      unsafeAtomicAdd(dst + 1, ay);
      unsafeAtomicAdd(dst + 2, aax);
      unsafeAtomicAdd(dst + 3, aay);
      unsafeAtomicAdd(dst + 4, aa);
      unsafeAtomicAdd(dst + 5, ab);
      unsafeAtomicAdd(dst + 6, ac);
      unsafeAtomicAdd(dst + 7, ao);
    }
  }
}

// largest t with item_offsets[t] <= b (item_offsets[T] = n_items > b): the tile owning item b
__device__ __forceinline__ int item_tile(const int *__restrict__ item_offsets, int T, int b) {
  int lo = 0, hi = T;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (item_offsets[mid] <= b) lo = mid; else hi = mid;
  }
  return lo;
}

// The same lookup done by the whole 256-thread workgroup in two parallel probes (256 samples at a
// fixed stride, then the entries of the selected stride) instead of log2(T) dependent loads by
// everyone: the serial search was 10-13 global-memory latencies at the head of every workgroup.
// Contains two workgroup barriers; `s_tmp` is 5 ints of LDS.
__device__ __forceinline__ int item_tile_coop(const int *__restrict__ item_offsets, int T, int b, int *s_tmp) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int S = (T + 256) / 256;  // ceil((T + 1) / 256) entries per sample
  const bool le = item_offsets[min(tid * S, T)] <= b;  // monotone: true for a prefix of the threads
  const unsigned long long bal = __ballot(le);
  if (lane == 0) s_tmp[wv] = __popcll(bal);
  __syncthreads();
  const int base = (s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3] - 1) * S;  // item_offsets[0] = 0 <= b
  if (tid < S && base + tid < T && item_offsets[base + tid] <= b && b < item_offsets[base + tid + 1])
    s_tmp[4] = base + tid;  // the one tile whose (non-empty) item range holds b
  __syncthreads();
  return s_tmp[4];
}

// Per-quadrant compacted lists of one slice in LDS, stored as PAIRS of Gaussians so that the walks
// can evaluate two Gaussians per lane with packed fp32 (v_pk_*: the walks are VALU-issue-bound).
struct QuadLists {
  float4 X[4][kSlice / 2 + 2];  // x0 x1 y0 y1
  float4 C[4][kSlice / 2 + 2];  // a0/2 a1/2 b0 b1
  float4 D[4][kSlice / 2 + 2];  // c0/2 c1/2 o0 o1
  float4 E[4][kSlice / 2 + 2];  // sigma thresholds thr0 thr1, slice-local indices idx0 idx1 (int bits)
  int cnt[4][4];                // [quadrant][source wave]
};

// Thread tid < kSlice brings Gaussian (s0 = x y a b, rB = c o thr idx) and the quadrants it reaches;
// on return wave w's list (quadrant w) is complete, in slice order, padded with three rejecting
// sentinels so that a 4-way unrolled walk may read past the end.  Two workgroup barriers.  Returns
// the length of the calling wave's list.
__device__ __forceinline__ int build_quad_lists(QuadLists &ql, const bool (&hitq)[4], const float4 s0,
                                                const float4 rB, int tid) {
  const int lane = tid & 63, wv = tid >> 6;
  unsigned long long bal[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    bal[q] = __ballot(hitq[q]);
    if (lane == 0) ql.cnt[q][wv] = __popcll(bal[q]);
  }
  __syncthreads();
  int n_mine = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int c = ql.cnt[q][w];
      base += (w < wv) ? c : 0;
      tot += c;
    }
    if (hitq[q]) {
      const int pos = base + __popcll(bal[q] & ((1ull << lane) - 1ull));
      const int pr = pos >> 1, sl = pos & 1;
      float *X = (float *)&ql.X[q][pr], *Cc = (float *)&ql.C[q][pr], *D = (float *)&ql.D[q][pr],
            *E = (float *)&ql.E[q][pr];
      X[sl] = s0.x; X[2 + sl] = s0.y;
      Cc[sl] = 0.5f * s0.z; Cc[2 + sl] = s0.w;  // the walks want the half conics
      D[sl] = 0.5f * rB.x; D[2 + sl] = rB.y;
      E[sl] = rB.z; E[2 + sl] = rB.w;
    }
    if (q == wv) n_mine = tot;
    if (tid >= 64 * q && tid < 64 * q + 3) {  // sentinels: threshold < 0 => rejected
      const int e = tot + (tid - 64 * q);
      const int pr = e >> 1, sl = e & 1;
      ((float *)&ql.X[q][pr])[sl] = 0.f; ((float *)&ql.X[q][pr])[2 + sl] = 0.f;
      ((float *)&ql.C[q][pr])[sl] = 0.f; ((float *)&ql.C[q][pr])[2 + sl] = 0.f;
      ((float *)&ql.D[q][pr])[sl] = 0.f; ((float *)&ql.D[q][pr])[2 + sl] = 0.f;
      ((float *)&ql.E[q][pr])[sl] = -1.f; ((float *)&ql.E[q][pr])[2 + sl] = 0.f;
    }
  }
  __syncthreads();
  return n_mine;
}

// alpha of a pair of listed Gaussians at pixel (px, py) with packed fp32, and whether each one counts
typedef float v2f __attribute__((ext_vector_type(2)));
struct PairEval {
  float a0, a1;
  bool k0, k1;
};
__device__ __forceinline__ PairEval eval_pair(const float4 X, const float4 Cq, const float4 D, const float4 E,
                                              const v2f px2, const v2f py2) {
  const v2f x = {X.x, X.y}, y = {X.z, X.w}, ha = {Cq.x, Cq.y}, bb = {Cq.z, Cq.w};
  const v2f hc = {D.x, D.y}, o = {D.z, D.w};
  const v2f dx = x - px2, dy = y - py2;
  const v2f sigma = dx * (ha * dx + bb * dy) + hc * dy * dy;
  const v2f arg = sigma * -1.44269504088896341f;
  const v2f e = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
  const v2f araw = o * e;
  PairEval r;
  r.a0 = fminf(kAlphaMax, araw.x);
  r.a1 = fminf(kAlphaMax, araw.y);
  // '&': no short-circuit branches
  r.k0 = (sigma.x >= 0.f) & (sigma.x <= E.x) & (r.a0 >= kAlphaMin);
  r.k1 = (sigma.y >= 0.f) & (sigma.y <= E.y) & (r.a1 >= kAlphaMin);
  return r;
}

__device__ __forceinline__ void block_loss_add(float l, float *sRed, float *__restrict__ loss_out) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) l += __shfl_xor(l, d, 64);
  if ((tid & 63) == 0) sRed[tid >> 6] = l;
  __syncthreads();
  if (tid == 0) {
    const float s = sRed[0] + sRed[1] + sRed[2] + sRed[3];
    if (s != 0.f) unsafeAtomicAdd(loss_out, s);
  }
}

// Phase B for one tile (256 threads = its 256 pixels): T = product of the slice products in depth order, the
// hand-over of stopping pixels to the re-walk list, the per-pixel epilogue.  `T`, `last`, `stop_slice` come in
// holding the result of a single-slice tile; with ns > 1 the slices' records are read back -- at DEVICE scope when
// they were published inside the same launch (another XCD's L2 may hold the line), plainly after a kernel boundary.
template <int CH, bool DEVICE_SCOPE>
__device__ __forceinline__ void combine_tail(int tile, int tid, int i0, int ns, bool inside, int p, float T, int last,
                                             int stop_slice, const SliceWs &ws, const int *__restrict__ flat,
                                             float *__restrict__ render, float *__restrict__ alphas,
                                             int *__restrict__ last_ids, bool has_loss, float gt_p, float w_p,
                                             float loss_scale, float *__restrict__ vpix, float *__restrict__ loss_out,
                                             StopRec *__restrict__ gtstop, float *sRed, bool rewalk_skipped) {
  if (ns > 1) {
    for (int s8 = 0; s8 < ns && stop_slice < 0; s8 += 8) {
      // eight slices' records in flight per wait (the walk over a tile's slices is a chain of round trips)
      float Ps[8];
      int Ls[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool ok = s8 + u < ns;
        const size_t o = (size_t)(i0 + (ok ? s8 + u : 0)) * kTilePix + tid;
        if (DEVICE_SCOPE) {
          Ps[u] = __hip_atomic_load(&ws.sliceP[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          Ls[u] = __hip_atomic_load(&ws.sliceL[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          Ps[u] = ws.sliceP[o];
          Ls[u] = ws.sliceL[o];
        }
        if (!ok) { Ps[u] = 1.f; Ls[u] = -1; }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (Ls[u] < 0 || stop_slice >= 0) continue;
        const float nT = T * Ps[u];
        if (nT <= kTStop) { stop_slice = s8 + u; continue; }
        T = nT;
        last = Ls[u];
      }
    }
  }
  StopInfo si;
  si.slice = inside ? stop_slice : -1;
  si.T = T;
  si.last = last;
  if (__syncthreads_or(si.slice >= 0)) {  // only tiles that hand pixels over need the per-pixel records
    if (rewalk_skipped) {
      // the caller speculated that no pixel would stop and did not launch the re-walk: tell it (sticky word 3 of
      // the control block); it restores its state and runs the step again with the re-walk
      if (tid == 0) atomicExch(&ws.ctl[3], 1);
    } else {
      ws.stopinfo[(size_t)tile * kTilePix + tid] = si;
      // first pixel to flag a slice puts it on the list (the re-walk kernel returns the flag to zero)
      if (si.slice >= 0 && atomicExch(&ws.item_flags[i0 + si.slice], 1) == 0)
        ws.rewalk[atomicAdd(&ws.ctl[0], 1)] = make_int2(i0 + si.slice, tile);
    }
  }
  float l = 0.f;
  if (inside && stop_slice < 0)
    l = finalize_pixel<CH>(p, T, last, false, flat, render, alphas, last_ids, has_loss, gt_p, w_p, loss_scale, vpix,
                           gtstop, nullptr);  // pixels finalised here did not stop
  if (has_loss && loss_out) block_loss_add(l, sRed, loss_out);
}

// forward phase A: per (tile, slice) transmittance products -- and, in the tile's last workgroup, phase B.
// Staging: thread t fetches Gaussian t of the slice, computes its conservative alpha >= 1/255 extent
// (ex, ey) and appends the packed record to the list of every quadrant it can touch (ballot +
// popcount compaction, 4 lists x 128 records in LDS).  Each wave then walks only ITS list.
// Phase B (T = product of the slice products in depth order): the slice workgroups of a tile publish their
// products with device-scope stores and take a ticket on the tile; whoever draws the last ticket reads all
// of them back (device-scope loads: another XCD's L2 may hold the line) and finalises the pixels.  A pixel
// whose running T * P_s drops to <= 1e-4 has its transmittance stop INSIDE slice s: the item goes on the
// re-walk list with the pixel's state before it.
// FUSED = false (many items: several rounds of workgroups per CU): a workgroup of a multi-slice tile stores its
// record plainly and leaves at once -- draining device-scope stores and a returning ticket atomic cost every
// workgroup ~4 us of residency, which multiplies by the number of rounds -- and composite_combine_fwd_kernel
// does phase B after the kernel boundary.  Single-slice tiles are finalised here in both modes.
template <int CH, bool FUSED>
__global__ void __launch_bounds__(256)
composite_slice_fwd_kernel(const float4 *__restrict__ splat, const TileTable tt_, const int *__restrict__ total,
                           const int *__restrict__ flat, int width, int height, int tw, int th, const SliceWs ws_,
                           float *__restrict__ render, float *__restrict__ alphas, int *__restrict__ last_ids,
                           const float *__restrict__ gt, const float *__restrict__ wmap, float loss_scale,
                           float *__restrict__ vpix, float *__restrict__ loss_out, StopRec *__restrict__ gtstop,
                           const Batch bt, int rewalk_skipped) {
  __shared__ QuadLists ql;
  static_assert(kSlice <= kTilePix, "one staging thread per Gaussian of the slice");
  __shared__ int sTile[5];
  __shared__ int s_last;
  __shared__ float sRed[4];
  const int bv = blockIdx.y;  // view of a batched step
  const TileTable tt = view_of(tt_, bt, bv);
  const SliceWs ws = view_of(ws_, bt, bv);
  total += 4 * bv; flat += bv * bt.keys; splat += bv * bt.splat4;
  if (render) render += bv * bt.pixels * CH;
  if (alphas) alphas += bv * bt.pixels;
  if (last_ids) last_ids += bv * bt.pixels;
  if (vpix) vpix += bv * bt.pixels;
  if (gtstop) gtstop += bv * bt.pixels;
  if (bt.gt[0]) { gt = bt.gt[bv]; wmap = bt.wmap[bv]; }
  const int b = blockIdx.x;
  if (b >= total[2]) return;
  const int tile = tt.item_tile ? tt.item_tile[b] : item_tile_coop(tt.item_first, tw * th, b, sTile);
  const int tid = threadIdx.x, wv = tid >> 6;
  const int ty = tile / tw, tx = tile - ty * tw;
  int di, dj;
  quad_pixel(tid, di, dj);
  const int i = ty * kTile + di, j = tx * kTile + dj;
  const bool inside = (i < height) && (j < width);
  const float px = (float)j + 0.5f, py = (float)i + 0.5f;
  const int i0 = tt.item_first[tile], ns = tt.item_end[tile] - i0;
  const int start = tt.start[tile] + (b - i0) * kSlice;
  const int end = min(tt.end[tile], start + kSlice);
  const int idx = start + tid;
  if (tt.cursor_reset && b == i0 && tid == 0) tt.cursor_reset[tile] = 0;

  float P = 1.f;
  int L = -1;
  if (end > start) {  // (an empty tile's single item has nothing to walk)
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), rB = s0;
    bool hitq[4] = {false, false, false, false};
    if (idx < end) {
      const int g = flat[idx];
      s0 = splat[2 * g];
      const float4 s1 = splat[2 * g + 1];
      const float thr = __logf(255.f * s1.y) + kThrMargin;
      rB = make_float4(s1.x, s1.y, thr, __int_as_float(tid));
      const float det = s0.z * s1.x - s0.w * s0.w;
      if (thr > 0.f && det > 0.f) {
        const float k2 = 2.f * thr * __builtin_amdgcn_rcpf(det);  // hardware rcp / sqrt: the inflation covers 1 ulp
        const float ex = __builtin_amdgcn_sqrtf(k2 * s1.x) * 1.001f + 0.01f;
        const float ey = __builtin_amdgcn_sqrtf(k2 * s0.z) * 1.001f + 0.01f;
        const float X0 = (float)(tx * kTile), Y0 = (float)(ty * kTile);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float qx = X0 + (float)((q & 1) << 3), qy = Y0 + (float)((q >> 1) << 3);
          // pixel centres of the quadrant span [q + 0.5, q + 7.5]: AABB reject, then the exact
          // ellipse-vs-rectangle test (thin diagonal ellipses miss most of their AABB)
          hitq[q] = (s0.x - ex <= qx + 7.5f) && (s0.x + ex >= qx + 0.5f) && (s0.y - ey <= qy + 7.5f) &&
                    (s0.y + ey >= qy + 0.5f) &&
                    ellipse_hits_rect(s0.x, s0.y, s0.z, s0.w, s1.x, thr, qx + 0.5f, qy + 0.5f, qx + 7.5f, qy + 7.5f);
        }
      }
    }
    if (tid < kSlice)  // which quadrants each Gaussian of the slice reaches: reused by the exact-stop re-walk
      ws.sliceQ[(size_t)b * kSlice + tid] = (unsigned char)((int)hitq[0] | ((int)hitq[1] << 1) | ((int)hitq[2] << 2) |
                                                            ((int)hitq[3] << 3));
    const int n_mine = build_quad_lists(ql, hitq, s0, rB, tid);

    const float4 *lX = ql.X[wv], *lC = ql.C[wv], *lD = ql.D[wv], *lE = ql.E[wv];
    const v2f px2 = {px, px}, py2 = {py, py};
    // Walk, two pairs (four Gaussians) per iteration: all LDS reads of a group are issued before the
    // first use.  No branches: every listed Gaussian reaches some pixel of the quadrant (exact test
    // above), so a wave-level skip never fires and exec-mask bookkeeping is pure overhead; a rejected
    // pair multiplies by 1.
    for (int t = 0; t < n_mine; t += 4) {
      float4 X[2], Cq[2], D[2], E[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        X[u] = lX[(t >> 1) + u]; Cq[u] = lC[(t >> 1) + u]; D[u] = lD[(t >> 1) + u]; E[u] = lE[(t >> 1) + u];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const PairEval ev = eval_pair(X[u], Cq[u], D[u], E[u], px2, py2);
        P *= ev.k0 ? 1.f - ev.a0 : 1.f;  // depth order kept: (P * m0) * m1
        P *= ev.k1 ? 1.f - ev.a1 : 1.f;
        L = ev.k0 ? t + 2 * u : L;       // list position (wave-uniform value); translated once after the walk
        L = ev.k1 ? t + 2 * u + 1 : L;
      }
    }
    if (L >= 0) L = start + __float_as_int(((const float *)&lE[L >> 1])[2 + (L & 1)]);
  }

  // the pixel's target and loss weight do not depend on the slices: in flight under the ticket round trip
  const bool has_loss = wmap != nullptr;
  const float w_p = (has_loss && inside) ? wmap[i * width + j] : 0.f;
  const float gt_p = (has_loss && inside) ? gt[i * width + j] : 0.f;

  float T = 1.f;
  int last = 0, stop_slice = -1;
  if (ns == 1) {  // the tile's only slice: nothing to publish, nothing to wait for
    if (L >= 0) {
      if (P <= kTStop) stop_slice = 0; else { T = P; last = L; }
    }
  } else if (!FUSED) {
    ws.sliceP[(size_t)b * kTilePix + tid] = P;
    ws.sliceL[(size_t)b * kTilePix + tid] = L;
    return;
  } else {
    // publish at device scope (the other slices of this tile may run on other XCDs, whose L2s do not snoop
    // this one), drain, take the tile's ticket
    __hip_atomic_store(&ws.sliceP[(size_t)b * kTilePix + tid], P, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&ws.sliceL[(size_t)b * kTilePix + tid], L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const bool lastwg = atomicAdd(&ws.tile_ticket[tile], 1) == ns - 1;
      if (lastwg) atomicExch(&ws.tile_ticket[tile], 0);  // ready for the next step
      s_last = lastwg;
    }
    __syncthreads();
    if (!s_last) return;  // (whole workgroup)
  }
  combine_tail<CH, true>(tile, tid, i0, ns, inside, i * width + j, T, last, stop_slice, ws, flat, render, alphas,
                         last_ids, has_loss, gt_p, w_p, loss_scale, vpix, loss_out, gtstop, sRed, rewalk_skipped != 0);
}

// ---------------------------------------------------------------------------------------------
// Chained forward: slice products, phase B and the EXACT transmittance stop in ONE kernel (used by the training
// step whenever pixels do reach the stop, i.e. for most of a real training run).  Every slice workgroup of a tile
// publishes its per-pixel product and last contributor (device-scope stores, then a per-item flag carrying the
// caller's tag) and then LOOKS BACK: it waits for the flags of the slices in front of it -- lower block indices,
// dispatched earlier, never waiting on it, so the wait cannot deadlock (the decoupled look-back idiom of
// single-pass scans) -- and multiplies their products up to its own slice.  That tells each pixel, in every slice
// workgroup, whether its stop fell in an earlier slice (nothing to do), falls in THIS slice (resolved on the spot
// by a sequential walk of the quadrant lists that are still in LDS -- no re-staging, no second kernel, no lists)
// or lies further back.  The workgroup in whose slice a pixel stops finalises that pixel; the last slice
// finalises the pixels that never stop.  Same arithmetic, in the same order, as slice -> combine -> re-walk.
__device__ __forceinline__ int stage_slice(QuadLists &ql, const float4 *__restrict__ splat,
                                           const int *__restrict__ flat, int start, int end, int tx, int ty, int tid) {
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), rB = s0;
  bool hitq[4] = {false, false, false, false};
  if (start + tid < end) {
    const int g = flat[start + tid];
    s0 = splat[2 * g];
    const float4 s1 = splat[2 * g + 1];
    const float thr = __logf(255.f * s1.y) + kThrMargin;
    rB = make_float4(s1.x, s1.y, thr, __int_as_float(tid));
    const float det = s0.z * s1.x - s0.w * s0.w;
    if (thr > 0.f && det > 0.f) {
      const float k2 = 2.f * thr * __builtin_amdgcn_rcpf(det);
      const float ex = __builtin_amdgcn_sqrtf(k2 * s1.x) * 1.001f + 0.01f;
      const float ey = __builtin_amdgcn_sqrtf(k2 * s0.z) * 1.001f + 0.01f;
      const float X0 = (float)(tx * kTile), Y0 = (float)(ty * kTile);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float qx = X0 + (float)((q & 1) << 3), qy = Y0 + (float)((q >> 1) << 3);
        hitq[q] = (s0.x - ex <= qx + 7.5f) && (s0.x + ex >= qx + 0.5f) && (s0.y - ey <= qy + 7.5f) &&
                  (s0.y + ey >= qy + 0.5f) &&
                  ellipse_hits_rect(s0.x, s0.y, s0.z, s0.w, s1.x, thr, qx + 0.5f, qy + 0.5f, qx + 7.5f, qy + 7.5f);
      }
    }
  }
  return build_quad_lists(ql, hitq, s0, rB, tid);
}

// sequential walk of this wave's quadrant list with the stop rule (the re-walk kernel's inner loop): lanes with
// `live` look for their stop from transmittance T; returns the list position of the last contributor (-1: none)
__device__ __forceinline__ int exact_walk(const QuadLists &ql, int wv, int n_mine, const v2f px2, const v2f py2,
                                          bool &live, float &T, bool &found) {
  const float4 *lX = ql.X[wv], *lC = ql.C[wv], *lD = ql.D[wv], *lE = ql.E[wv];
  int lastpos = -1;
  for (int t = 0; t < n_mine && __ballot(live) != 0ull; t += 4) {
    float4 X[2], Cq[2], D[2], E[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      X[u] = lX[(t >> 1) + u]; Cq[u] = lC[(t >> 1) + u]; D[u] = lD[(t >> 1) + u]; E[u] = lE[(t >> 1) + u];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const PairEval ev = eval_pair(X[u], Cq[u], D[u], E[u], px2, py2);
      {
        const float nT = T * (1.f - ev.a0);
        const bool hit = live & ev.k0, stop = hit & (nT <= kTStop), upd = hit & !stop;
        T = upd ? nT : T;
        lastpos = upd ? t + 2 * u : lastpos;
        found = found | stop;
        live = live & !stop;
      }
      {
        const float nT = T * (1.f - ev.a1);
        const bool hit = live & ev.k1, stop = hit & (nT <= kTStop), upd = hit & !stop;
        T = upd ? nT : T;
        lastpos = upd ? t + 2 * u + 1 : lastpos;
        found = found | stop;
        live = live & !stop;
      }
    }
  }
  return lastpos;
}

template <int CH>
__global__ void __launch_bounds__(256)
composite_chained_fwd_kernel(const float4 *__restrict__ splat, const TileTable tt_, const int *__restrict__ total,
                             const int *__restrict__ flat, int width, int height, int tw, int th, const SliceWs ws_,
                             int tag, float *__restrict__ render, float *__restrict__ alphas,
                             int *__restrict__ last_ids, const float *__restrict__ gt, const float *__restrict__ wmap,
                             float loss_scale, float *__restrict__ vpix, float *__restrict__ loss_out,
                             StopRec *__restrict__ gtstop, const Batch bt) {
  __shared__ QuadLists ql;
  __shared__ int sTile[5];
  __shared__ float sRed[4];
  const int bv = blockIdx.y;  // view of a batched step
  const TileTable tt = view_of(tt_, bt, bv);
  const SliceWs ws = view_of(ws_, bt, bv);
  total += 4 * bv; flat += bv * bt.keys; splat += bv * bt.splat4;
  if (render) render += bv * bt.pixels * CH;
  if (alphas) alphas += bv * bt.pixels;
  if (last_ids) last_ids += bv * bt.pixels;
  if (vpix) vpix += bv * bt.pixels;
  if (gtstop) gtstop += bv * bt.pixels;
  if (bt.gt[0]) { gt = bt.gt[bv]; wmap = bt.wmap[bv]; }
  const int b = blockIdx.x;
  if (b >= total[2]) return;
  const int tile = tt.item_tile ? tt.item_tile[b] : item_tile_coop(tt.item_first, tw * th, b, sTile);
  const int tid = threadIdx.x, wv = tid >> 6;
  const int ty = tile / tw, tx = tile - ty * tw;
  int di, dj;
  quad_pixel(tid, di, dj);
  const int i = ty * kTile + di, j = tx * kTile + dj;
  const bool inside = (i < height) && (j < width);
  const float px = (float)j + 0.5f, py = (float)i + 0.5f;
  const v2f px2 = {px, px}, py2 = {py, py};
  const int i0 = tt.item_first[tile], ns = tt.item_end[tile] - i0, s_me = b - i0;
  const int t_start = tt.start[tile], t_end = tt.end[tile];
  const int start = t_start + s_me * kSlice, end = min(t_end, start + kSlice);
  if (tt.cursor_reset && s_me == 0 && tid == 0) tt.cursor_reset[tile] = 0;
  const bool has_loss = wmap != nullptr;
  const float w_p = (has_loss && inside) ? wmap[i * width + j] : 0.f;
  const float gt_p = (has_loss && inside) ? gt[i * width + j] : 0.f;

  // ---- phase A: this slice's product and last contributor (lists stay in LDS)
  float P = 1.f;
  int Lpos = -1, n_mine = 0;
  if (end > start) {
    n_mine = stage_slice(ql, splat, flat, start, end, tx, ty, tid);
    const float4 *lX = ql.X[wv], *lC = ql.C[wv], *lD = ql.D[wv], *lE = ql.E[wv];
    for (int t = 0; t < n_mine; t += 4) {
      float4 X[2], Cq[2], D[2], E[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        X[u] = lX[(t >> 1) + u]; Cq[u] = lC[(t >> 1) + u]; D[u] = lD[(t >> 1) + u]; E[u] = lE[(t >> 1) + u];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const PairEval ev = eval_pair(X[u], Cq[u], D[u], E[u], px2, py2);
        P *= ev.k0 ? 1.f - ev.a0 : 1.f;
        P *= ev.k1 ? 1.f - ev.a1 : 1.f;
        Lpos = ev.k0 ? t + 2 * u : Lpos;
        Lpos = ev.k1 ? t + 2 * u + 1 : Lpos;
      }
    }
  }
  const int L = (Lpos >= 0) ? start + __float_as_int(((const float *)&ql.E[wv][Lpos >> 1])[2 + (Lpos & 1)]) : -1;

  // ---- publish for the slices behind this one (the last slice has nobody behind it)
  if (s_me < ns - 1) {
    __hip_atomic_store(&ws.sliceP[(size_t)b * kTilePix + tid], P, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&ws.sliceL[(size_t)b * kTilePix + tid], L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&ws.ready[b], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // ---- look back: T in front of this slice; `before` = the pixel stopped in an earlier slice
  float T = 1.f;
  int last = 0;
  bool before = false;
  for (int j8 = 0; j8 < s_me; j8 += 8) {
    const int nb = min(8, s_me - j8);
    if (tid < nb)  // one poller per awaited flag; the others wait at the barrier
      while (__hip_atomic_load(&ws.ready[i0 + j8 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag)
        __builtin_amdgcn_s_sleep(2);
    __syncthreads();
    float Ps[8];
    int Ls[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool ok = u < nb;
      const size_t o = (size_t)(i0 + j8 + (ok ? u : 0)) * kTilePix + tid;
      Ps[u] = __hip_atomic_load(&ws.sliceP[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      Ls[u] = __hip_atomic_load(&ws.sliceL[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!ok) { Ps[u] = 1.f; Ls[u] = -1; }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (Ls[u] < 0 || before) continue;
      const float nT = T * Ps[u];
      if (nT <= kTStop) { before = true; continue; }
      T = nT;
      last = Ls[u];
    }
  }
  // ---- this slice: does the stop fall in it?  (same test, on the same products, as the combine of the split path)
  bool cross = false;
  if (!before && L >= 0) {
    const float nT = T * P;
    if (nT <= kTStop) cross = true; else { T = nT; last = L; }
  }
  cross = cross && inside;
  bool found = false;
  if (__syncthreads_or(cross)) {
    if (tid == 0 && ws.ctl[2] == 0) atomicMax(&ws.ctl[2], 1);  // "pixels do stop": the caller's launch-mode hint
    // exact stop from the lists still in LDS, sequentially in depth order from T; should float rounding move the
    // crossing past the slice end, the same lanes carry on through the following slices (staged afresh)
    bool live = cross;
    const int lp = exact_walk(ql, wv, n_mine, px2, py2, live, T, found);
    if (lp >= 0) last = start + __float_as_int(((const float *)&ql.E[wv][lp >> 1])[2 + (lp & 1)]);
    for (int s2 = s_me + 1; s2 < ns; ++s2) {
      if (!__syncthreads_or(cross && !found)) break;
      const int st2 = t_start + s2 * kSlice, en2 = min(t_end, st2 + kSlice);
      const int n2 = stage_slice(ql, splat, flat, st2, en2, tx, ty, tid);
      live = cross && !found;
      const int lp2 = exact_walk(ql, wv, n2, px2, py2, live, T, found);
      if (lp2 >= 0) last = st2 + __float_as_int(((const float *)&ql.E[wv][lp2 >> 1])[2 + (lp2 & 1)]);
    }
  }
  // ---- finalise: the pixels that stop here (whichever way the exact walk ended), and -- in the last slice -- the
  // pixels that never stop
  float l = 0.f;
  if (cross || (inside && !before && s_me == ns - 1))
    l = finalize_pixel<CH>(i * width + j, T, last, cross && found, flat, render, alphas, last_ids, has_loss, gt_p, w_p,
                           loss_scale, vpix, gtstop, splat);
  if (has_loss && loss_out) block_loss_add(l, sRed, loss_out);
}

// phase B as its own launch (FUSED = false): one workgroup per tile with more than one slice
template <int CH>
__global__ void __launch_bounds__(256)
composite_combine_fwd_kernel(const TileTable tt_, const int *__restrict__ flat, int width, int height, int tw,
                             const SliceWs ws_, float *__restrict__ render, float *__restrict__ alphas,
                             int *__restrict__ last_ids, const float *__restrict__ gt, const float *__restrict__ wmap,
                             float loss_scale, float *__restrict__ vpix, float *__restrict__ loss_out,
                             StopRec *__restrict__ gtstop, const Batch bt, int rewalk_skipped) {
  __shared__ float sRed[4];
  const int bv = blockIdx.y;  // view of a batched step
  const TileTable tt = view_of(tt_, bt, bv);
  const SliceWs ws = view_of(ws_, bt, bv);
  flat += bv * bt.keys;
  if (render) render += bv * bt.pixels * CH;
  if (alphas) alphas += bv * bt.pixels;
  if (last_ids) last_ids += bv * bt.pixels;
  if (vpix) vpix += bv * bt.pixels;
  if (gtstop) gtstop += bv * bt.pixels;
  if (bt.gt[0]) { gt = bt.gt[bv]; wmap = bt.wmap[bv]; }
  const int tile = blockIdx.x, tid = threadIdx.x;
  const int i0 = tt.item_first[tile], ns = tt.item_end[tile] - i0;
  if (ns <= 1) return;  // finalised by its slice workgroup
  const int ty = tile / tw, tx = tile - ty * tw;
  int di, dj;
  quad_pixel(tid, di, dj);  // same thread -> pixel map as the slice kernel
  const int i = ty * kTile + di, j = tx * kTile + dj;
  const bool inside = (i < height) && (j < width);
  const bool has_loss = wmap != nullptr;
  const float w_p = (has_loss && inside) ? wmap[i * width + j] : 0.f;
  const float gt_p = (has_loss && inside) ? gt[i * width + j] : 0.f;
  combine_tail<CH, false>(tile, tid, i0, ns, inside, i * width + j, 1.f, 0, -1, ws, flat, render, alphas, last_ids,
                          has_loss, gt_p, w_p, loss_scale, vpix, loss_out, gtstop, sRed, rewalk_skipped != 0);
}

// forward phase C: exact transmittance stop.  A small grid strides over the compact list of flagged
// (tile, slice) items: the slice's records are staged through LDS once and the pixels whose stop falls in this
// slice walk it sequentially from their known T; should float rounding move the crossing past the slice end,
// the same lanes carry on through the following slices.  In scenes without stops the launch reads one word.
template <int CH>
__global__ void __launch_bounds__(256)
composite_rewalk_fwd_kernel(const float4 *__restrict__ splat, const TileTable tt_,
                            const int *__restrict__ flat, int width, int height, int tw, int th, const SliceWs ws_,
                            float *__restrict__ render, float *__restrict__ alphas, int *__restrict__ last_ids,
                            const float *__restrict__ gt, const float *__restrict__ wmap, float loss_scale,
                            float *__restrict__ vpix, float *__restrict__ loss_out, StopRec *__restrict__ gtstop,
                            const Batch bt) {
  __shared__ QuadLists ql;
  __shared__ float sRed[4];
  const int bv = blockIdx.y;  // view of a batched step
  const TileTable tt = view_of(tt_, bt, bv);
  const SliceWs ws = view_of(ws_, bt, bv);
  flat += bv * bt.keys; splat += bv * bt.splat4;
  if (render) render += bv * bt.pixels * CH;
  if (alphas) alphas += bv * bt.pixels;
  if (last_ids) last_ids += bv * bt.pixels;
  if (vpix) vpix += bv * bt.pixels;
  if (gtstop) gtstop += bv * bt.pixels;
  if (bt.gt[0]) { gt = bt.gt[bv]; wmap = bt.wmap[bv]; }
  const int tid = threadIdx.x, wv = tid >> 6;
  int di, dj;
  quad_pixel(tid, di, dj);
  const int n_list = ws.ctl[0];
  if ((int)blockIdx.x >= n_list) return;  // the usual case: an empty list costs one load per workgroup
  for (int k = blockIdx.x; k < n_list; k += gridDim.x) {
  __syncthreads();
  const int2 it = ws.rewalk[k];
  const int b = it.x, tile = it.y;
  const int ty = tile / tw, tx = tile - ty * tw;
  const int i = ty * kTile + di, j = tx * kTile + dj;
  const float px = (float)j + 0.5f, py = (float)i + 0.5f;
  const v2f px2 = {px, px}, py2 = {py, py};
  const int ib = tt.item_first[tile];
  const int s0 = b - ib, ns = tt.item_end[tile] - ib;
  const StopInfo si = ws.stopinfo[(size_t)tile * kTilePix + tid];
  const bool mine = si.slice == s0;
  float T = si.T;
  int last = si.last;
  bool found = false;
  for (int s = s0; s < ns; ++s) {
    if (!__syncthreads_or(mine && !found)) break;
    // stage the slice exactly like the slice kernel did; the quadrant test is not repeated, the slice
    // kernel left its verdicts in sliceQ
    const int start = tt.start[tile] + s * kSlice, end = min(tt.end[tile], start + kSlice);
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), rB = g0;
    bool hitq[4] = {false, false, false, false};
    if (start + tid < end) {
      const int g = flat[start + tid];
      g0 = splat[2 * g];
      const float4 g1 = splat[2 * g + 1];
      rB = make_float4(g1.x, g1.y, __logf(255.f * g1.y) + kThrMargin, __int_as_float(tid));
      const int m = ws.sliceQ[(size_t)(ib + s) * kSlice + tid];
#pragma unroll
      for (int q = 0; q < 4; ++q) hitq[q] = (m >> q) & 1;
    }
    const int n_mine = build_quad_lists(ql, hitq, g0, rB, tid);
    const float4 *lX = ql.X[wv], *lC = ql.C[wv], *lD = ql.D[wv], *lE = ql.E[wv];
    // sequential walk with the stop rule, branch-free inside: `live` lanes are still looking for
    // their stop; a wave leaves as soon as none of its lanes is
    bool live = mine && !found;
    int lastpos = -1;
    for (int t = 0; t < n_mine && __ballot(live) != 0ull; t += 4) {
      float4 X[2], Cq[2], D[2], E[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        X[u] = lX[(t >> 1) + u]; Cq[u] = lC[(t >> 1) + u]; D[u] = lD[(t >> 1) + u]; E[u] = lE[(t >> 1) + u];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const PairEval ev = eval_pair(X[u], Cq[u], D[u], E[u], px2, py2);
        {
          const float nT = T * (1.f - ev.a0);
          const bool hit = live & ev.k0, stop = hit & (nT <= kTStop), upd = hit & !stop;
          T = upd ? nT : T;
          lastpos = upd ? t + 2 * u : lastpos;
          found = found | stop;
          live = live & !stop;
        }
        {
          const float nT = T * (1.f - ev.a1);
          const bool hit = live & ev.k1, stop = hit & (nT <= kTStop), upd = hit & !stop;
          T = upd ? nT : T;
          lastpos = upd ? t + 2 * u + 1 : lastpos;
          found = found | stop;
          live = live & !stop;
        }
      }
    }
    if (lastpos >= 0) last = start + __float_as_int(((const float *)&lE[lastpos >> 1])[2 + (lastpos & 1)]);
  }
  float l = 0.f;
  if (mine)
    l = finalize_pixel<CH>(i * width + j, T, last, found, flat, render, alphas, last_ids, wmap != nullptr,
                           wmap ? gt[i * width + j] : 0.f, wmap ? wmap[i * width + j] : 0.f, loss_scale, vpix, gtstop,
                           splat);
  if (wmap && loss_out) block_loss_add(l, sRed, loss_out);
  if (tid == 0) ws.item_flags[b] = 0;  // off the list
  }  // list loop
  // the last of the workgroups that had work empties the list for the next step: every one of them has read its
  // length by now, and a workgroup that starts later reads zero and leaves.  Only these take the exit ticket, and
  // in TWO LEVELS (64 groups on 64 cache lines, then the groups' last members on one word): returning atomics on
  // one address run at ~60 ns apiece, a flat ticket cost 70 us with 1024 idle workgroups and 30 us of the 41 us this
  // kernel took on a trained-like scene with 700 busy ones
  if (tid == 0) {
    const int P = min((int)gridDim.x, n_list), grp = blockIdx.x & 63;
    const int members = (P - grp + 63) >> 6;
    if (atomicAdd(&ws.exit_grp[grp * 16], 1) == members - 1) {
      atomicExch(&ws.exit_grp[grp * 16], 0);
      if (atomicAdd(&ws.ctl[1], 1) == min(P, 64) - 1) {
        ws.ctl[2] = max(ws.ctl[2], n_list);  // longest list since the caller last looked (its launch-shape hint)
        atomicExch(&ws.ctl[0], 0);
        atomicExch(&ws.ctl[1], 0);
      }
    }
  }
}

// backward, unit colours, one workgroup per item: lane = Gaussian of the slice, loop = active pixels
__global__ void __launch_bounds__(256)
composite_bwd_item_kernel(const float4 *__restrict__ splat, const int *__restrict__ offsets,
                          const int *__restrict__ item_offsets, const int *__restrict__ total,
                          const int *__restrict__ flat, int width, int height, int tw, int th,
                          const float *__restrict__ alphas, const int *__restrict__ last_ids,
                          const float *__restrict__ vpix, float *__restrict__ g2d) {
  __shared__ float4 sP[kTilePix];  // px, py, v*T_final, last id (int bits) -- compacted
  __shared__ int sCnt[4];
  __shared__ int sMaxLast[4];
  const int b = blockIdx.x;
  if (b >= total[2]) return;
  const int tile = item_tile(item_offsets, tw * th, b);
  const int start = offsets[tile] + (b - item_offsets[tile]) * kSlice;
  const int end = min(offsets[tile + 1], start + kSlice);

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int ty = tile / tw, tx = tile - ty * tw;
  const int i = ty * kTile + (tid >> 4), j = tx * kTile + (tid & 15);
  const bool inside = (i < height) && (j < width);
  float gT = 0.f;
  int last = -1;
  if (inside) {
    const int p = i * width + j;
    const float a = alphas[p];
    const float v = vpix[p];
    if (a > 0.f && v != 0.f) {  // a > 0 <=> at least one Gaussian contributed to this pixel
      const int l = last_ids[p];
      if (l >= start) { gT = v * (1.f - a); last = l; }  // nothing of this slice contributes otherwise
    }
  }
  const bool active = (last >= 0) && (gT != 0.f);
  const unsigned long long bal = __ballot(active);
  const int rank = __popcll(bal & ((1ull << lane) - 1ull));
  int wmax = last;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, __shfl_xor(wmax, d, 64));
  if (lane == 0) { sCnt[wv] = __popcll(bal); sMaxLast[wv] = wmax; }
  __syncthreads();
  int pre = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) pre += (w < wv) ? sCnt[w] : 0;
  const int n_act = sCnt[0] + sCnt[1] + sCnt[2] + sCnt[3];
  const int max_last = max(max(sMaxLast[0], sMaxLast[1]), max(sMaxLast[2], sMaxLast[3]));
  if (n_act == 0) return;
  if (active) sP[pre + rank] = make_float4((float)j + 0.5f, (float)i + 0.5f, gT, __int_as_float(last));
  __syncthreads();

  const int n_live = min(end, max_last + 1) - start;  // Gaussians past every pixel's stop are dead
  if (n_live <= 0) return;
  const int n_chunks = (n_live + 63) >> 6;
  const int splits = (n_chunks >= 4) ? 1 : ((n_chunks == 2) ? 2 : 4);
  const int n_items = n_chunks * splits;
  for (int item = wv; item < n_items; item += 4) {
    const int chunk = item / splits, sp = item - chunk * splits;
    const int q0 = (n_act * sp) / splits, q1 = (n_act * (sp + 1)) / splits;
    const int idx = start + (chunk << 6) + lane;
    const bool have = idx < start + n_live;
    int g = 0;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (have) {
      g = flat[idx];
      s0 = splat[2 * g];
      s1 = splat[2 * g + 1];
    }
    const float x = s0.x, y = s0.y, ca = s0.z, cb = s0.w, cc = s1.x, o = s1.y;
    const float thr = have ? __logf(255.f * o) + kThrMargin : -1.f;
    float ax = 0.f, ay = 0.f, aax = 0.f, aay = 0.f, aa = 0.f, ab = 0.f, ac = 0.f, ao = 0.f;
    bool hit = false;
    for (int q = q0; q < q1; ++q) {
      const float4 P = sP[q];
      const float dx = x - P.x, dy = y - P.y;
      const float sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
      bool valid = (idx <= __float_as_int(P.w)) && (sigma >= 0.f) && (sigma <= thr);
      if (!__any(valid)) continue;
      const float vis = __expf(-sigma);
      const float araw = o * vis;
      const float alpha = fminf(kAlphaMax, araw);
      valid = valid && (alpha >= kAlphaMin);
      if (valid) {
        hit = true;
        const float v_alpha = P.z * __builtin_amdgcn_rcpf(1.f - alpha);  // dL/dalpha = v * T_final / (1 - alpha)
        if (araw <= kAlphaMax) {
          const float v_sigma = -araw * v_alpha;
          const float gx = v_sigma * (ca * dx + cb * dy);
          const float gy = v_sigma * (cb * dx + cc * dy);
          ax += gx; ay += gy;
          aax += fabsf(gx); aay += fabsf(gy);
          aa += 0.5f * v_sigma * dx * dx;
          ab += v_sigma * dx * dy;
          ac += 0.5f * v_sigma * dy * dy;
          ao += vis * v_alpha;
        }
      }
    }
    if (hit) {
      float *dst = g2d + (size_t)g * 8;
      unsafeAtomicAdd(dst + 0, ax);
      unsafeAtomicAdd(dst + 1, ay);
      unsafeAtomicAdd(dst + 2, aax);
      unsafeAtomicAdd(dst + 3, aay);
      unsafeAtomicAdd(dst + 4, aa);
      unsafeAtomicAdd(dst + 5, ab);
      unsafeAtomicAdd(dst + 6, ac);
      unsafeAtomicAdd(dst + 7, ao);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Footprint backward for unit colours (the fused training path): no tile lists, no atomics.
//
// With colours == 1 and no background dL/dalpha_i = v_p T_final,p / (1 - alpha_i) for every Gaussian
// that contributed to pixel p, independent of the depth order.  So a Gaussian's whole 2D gradient is
// a sum over ITS OWN footprint: the pixels of gsplat's tile box that lie inside the ellipse
// sigma <= ln(255 o) (outside it alpha < 1/255 and the forward skipped the pair), read from the
// packed {v * T_final, stop id} image the forward wrote.  The transmittance stop (pixels whose
// front-to-back walk ended on T <= 1e-4) is the only order-dependent part: for those pixels the
// forward records the id of the last contributing Gaussian and a candidate contributes iff its
// (depth bits, id) <= that Gaussian's.
//
// The walk.  On pixel row i the ellipse is the interval  x_g + (b/a) dy -/+ sqrt(2 a thr - det dy^2)/a,
// dy = y_g - (i + 0.5): its centre moves linearly with the row and its half width never exceeds
// hw = sqrt(2 thr / a).  So the footprint is covered by a SHEARED box of constant width
// pw = floor(2 hw) + 1 columns starting at ceil(x_g + (b/a) dy - hw - 0.5) -- pi/4 of it lies inside
// the ellipse whatever the orientation, where the axis-aligned box of a thin diagonal edge Gaussian
// is mostly empty.  (If the sheared box is not narrower than the axis-aligned one the latter is used.)
// The rows x pw cells are numbered row-major and dealt to lanes with a stride, which needs one
// divmod per lane and stream; after that a position advances by a constant (rows, columns) step with
// a carry.
//
// Lanes.  A wavefront owns 8 consecutive Gaussians and hands its 64 lanes out in proportion to their
// cell counts (every non-empty footprint gets at least one lane), so that all lanes of the wave run
// (nearly) the same number of visits whatever the size mix -- with a fixed 8 lanes per Gaussian the
// wave waits for its largest footprint (measured 1.7x on random scenes) and rows whose width is not
// a multiple of 8 leave lanes idle.  Each lane accumulates first/second moments of w = dL/dsigma
// (the gradient is linear in them); the partial g2d records go through LDS and lane (k, component)
// adds the partials of Gaussian k in lane order: deterministic, no atomics, one coalesced 256-byte
// store per wave, g2d needs no zeroing.

}  // namespace eg
#include "footprint_dev.h"
namespace eg {

#ifndef EG_FP_SHARED_WALK
#define EG_FP_SHARED_WALK 1  // (round 5, same box: 8.78 / 20.67 / 158.1 / 188.4 -> 8.24 / 19.7 / 157.7 / 186.3 us at configs 1-4; 0 = sized per wave)
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
footprint_bwd_kernel(const float4 *__restrict__ splat, int N, int width, int height,
                     const StopRec *__restrict__ gtstop, float *__restrict__ g2d, const Batch bt,
                     float *__restrict__ loss_part, float *__restrict__ loss_out) {
  __shared__ float red[4][64 * 8];
  splat += blockIdx.y * bt.splat4; gtstop += blockIdx.y * bt.pixels; g2d += blockIdx.y * bt.splat4 * 4;  // view
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (loss_part && blockIdx.x == 0 && wv == 0) {
    // the wave-autonomous forward left this view's loss terms in 64 partial sums: fold them into the caller's
    // accumulator and hand the partials back zeroed
    float *lp = (float *)((char *)loss_part + blockIdx.y * bt.ws_bytes);
    float v = lp[lane];
    if (v != 0.f) lp[lane] = 0.f;
    v = wave_sum_dpp_f(v);
    if (lane == 63 && v != 0.f) unsafeAtomicAdd(loss_out, v);
  }
  const int wave = blockIdx.x * 4 + wv;
  const int gbase = wave * 8;
#if EG_FP_SHARED_WALK
  // Round 5: the workgroup's 32 footprints are sized ONCE, one Gaussian per lane of the first wave's lower half, and handed
  // to the four waves through LDS -- sized in the home phase of every wave, eight lanes per Gaussian, the same arithmetic
  // ran four times per workgroup: ~150 of the ~1000 vector instructions a wave of this kernel issues at config 2.
  __shared__ __attribute__((aligned(16))) int s_walk[32][12];  // i0 fh pw jlo | jhi cells thr xoff | shear - - -
  if (threadIdx.x < 32) {
    const int hg = blockIdx.x * 32 + (int)threadIdx.x;
    Walk w = walk_of(make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), width, height);
    if (hg < N) w = walk_of(splat[2 * hg], splat[2 * hg + 1], width, height);
    int4 *dst = (int4 *)s_walk[threadIdx.x];
    dst[0] = make_int4(w.i0, w.fh, w.pw, w.jlo);
    dst[1] = make_int4(w.jhi, w.cells, __float_as_int(w.thr), __float_as_int(w.xoff));
    dst[2] = make_int4(__float_as_int(w.shear), 0, 0, 0);
  }
  __syncthreads();  // (the only barrier of the kernel: in front of the whole-wave exits)
#endif
  if (gbase >= N) return;  // whole waves leave; there is no workgroup barrier below
  // descriptor of this view's record image, built from uniform values only
  const __amdgpu_buffer_rsrc_t rec_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)gtstop, 0, width * height * (int)sizeof(StopRec), 0x00020000);

  // (Eight Gaussians per wave is the measured optimum at the reference's sizes: what precedes and follows the walk cost
  // a wave ~400 VALU instructions whatever it holds -- a third of the kernel's issue slots at config 2 -- but sixteen
  // Gaussians per wave lose more to the coarser lane dealing than they save, +13 % at config 2 and +47 % at
  // 1600 x 1200; four per wave gain 8 % at 1600 x 1200 and lose 15 % at config 2.)
  // home phase: the 8 lanes of group k all size the footprint of Gaussian gbase + k
#if EG_FP_SHARED_WALK
  Walk h;
  {
    const int4 *src = (const int4 *)s_walk[wv * 8 + (lane >> 3)];
    const int4 a = src[0], b = src[1];
    h.i0 = a.x; h.fh = a.y; h.pw = a.z; h.jlo = a.w; h.jhi = b.x; h.cells = b.y;
    h.thr = __int_as_float(b.z); h.xoff = __int_as_float(b.w); h.shear = __int_as_float(src[2].x);
  }
#else
  Walk h = walk_of(make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), width, height);
  {
    const int hg = gbase + (lane >> 3);
    if (hg < N) h = walk_of(splat[2 * hg], splat[2 * hg + 1], width, height);
  }
#endif
  // Lanes per Gaussian, computed group-parallel: one lane for every live footprint, the other
  // 64 - live in proportion to the cell counts (rounded down), the slack (<= live) one each to the
  // first live groups.  A footprint of any size is handled here: a screen-filling Gaussian simply
  // takes (nearly) all lanes of its wave.
  int total = h.cells, live = h.cells > 0 ? 1 : 0;
#pragma unroll
  for (int d = 8; d < 64; d <<= 1) {
    total += __shfl_xor(total, d, 64);
    live += __shfl_xor(live, d, 64);
  }
  int h_n = 0, h_first = 0;
  Moments m = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  if (total > 0) {
    const float share = (float)(64 - live) * (1.f - 1e-5f) * __builtin_amdgcn_rcpf((float)total);
    h_n = h.cells > 0 ? 1 + (int)((float)h.cells * share) : 0;
    int incl = h_n + (h.cells > 0 ? 1 << 16 : 0);  // inclusive scan over the groups of (lanes, live flag)
#pragma unroll
    for (int d = 8; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d, 64);
      if (lane >= d) incl += up;
    }
    const int slack = 64 - (__builtin_amdgcn_readlane(incl, 63) & 0xffff);
    const int rank = (incl >> 16) - (h.cells > 0 ? 1 : 0);  // live groups before this one
    h_first = (incl & 0xffff) - h_n + min(rank, slack);
    if (h.cells > 0 && rank < slack) ++h_n;
    // lane -> Gaussian: the group whose lane range holds this lane (a lane past the last range idles)
    int k = 0;
#pragma unroll
    for (int q = 1; q < 8; ++q) k += lane >= __builtin_amdgcn_readlane(h_first, 8 * q);
    const int src = 8 * k;
    const int n = max(__shfl(h_n, src, 64), 1);
    const int r = lane - __shfl(h_first, src, 64);
    const int g = min(gbase + k, N - 1);
    const int cells = r < n ? __shfl(h.cells, src, 64) : 0;
    s0 = splat[2 * g];
    s1 = splat[2 * g + 1];
#if EG_FP_SHARED_WALK
    {  // (the walk of the lane's Gaussian: three 16-byte LDS reads instead of nine lane shuffles)
      const int4 *wk = (const int4 *)s_walk[wv * 8 + k];
      const int4 a = wk[0], b = wk[1];
      const float shear_k = __int_as_float(wk[2].x);
      footprint_walk(s0, s1, g, r, n, a.x, cells > 0 ? a.y : 0, max(a.z, 1), a.w, b.x, __int_as_float(b.z),
                     __int_as_float(b.w), shear_k, width, rec_rsrc, m);
    }
#else
    footprint_walk(s0, s1, g, r, n, __shfl(h.i0, src, 64), cells > 0 ? __shfl(h.fh, src, 64) : 0,
                   max(__shfl(h.pw, src, 64), 1), __shfl(h.jlo, src, 64), __shfl(h.jhi, src, 64), __shfl(h.thr, src, 64),
                   __shfl(h.xoff, src, 64), __shfl(h.shear, src, 64), width, rec_rsrc, m);
#endif
  }
  // partial g2d record of this lane: vx vy |vx| |vy| va vb vc vo
  float *mine = &red[wv][lane * 8];
  // (the absgrad sums were taken over |w| |h|, h = log2(e) / 2 times the gradient of sigma: footprint_visit)
  *(float4 *)mine = make_float4(s0.z * m.w_x + s0.w * m.w_y, s0.w * m.w_x + s1.x * m.w_y, (2.f / kLog2e) * m.abs_x,
                                (2.f / kLog2e) * m.abs_y);
  *(float4 *)(mine + 4) = make_float4(0.5f * m.w_xx, m.w_xy, 0.5f * m.w_yy, m.v_o);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // lane (k, component) adds the partial records of Gaussian k's lanes in lane order
  const int comp = lane & 7;
  float sum = 0.f;
  for (int t = 0; t < h_n; ++t) sum += red[wv][(h_first + t) * 8 + comp];
  if (gbase + (lane >> 3) < N) g2d[(size_t)gbase * 8 + lane] = sum;
}

// ---------------------------------------------------------------------------------------------
// backward, general colours: lane = pixel, back to front, wave64 butterfly reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

template <int CH>
__global__ void __launch_bounds__(256)
composite_bwd_colors_kernel(const float4 *__restrict__ splat, const float *__restrict__ colors,
                            const int *__restrict__ offsets, const int *__restrict__ flat, int width,
                            int height, int tw, int th, const float *__restrict__ alphas,
                            const int *__restrict__ last_ids, const float *__restrict__ v_render,
                            const float *__restrict__ v_alphas, float *__restrict__ g2d,
                            float *__restrict__ v_colors) {
  __shared__ float4 sA[kTilePix];
  __shared__ float4 sB[kTilePix];
  __shared__ float sC[kTilePix * CH];
  __shared__ int sG[kTilePix];

  const int tile = xcd_tile(blockIdx.x, tw * th);
  const int start = offsets[tile], end = offsets[tile + 1];
  if (end <= start) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int ty = tile / tw, tx = tile - ty * tw;
  const int i = ty * kTile + (tid >> 4), j = tx * kTile + (tid & 15);
  const bool inside = (i < height) && (j < width);
  const float px = (float)j + 0.5f, py = (float)i + 0.5f;
  const int p = inside ? i * width + j : 0;

  const float T_final = inside ? 1.f - alphas[p] : 1.f;
  float T = T_final;
  float buffer[CH], vr[CH];
#pragma unroll
  for (int k = 0; k < CH; ++k) { buffer[k] = 0.f; vr[k] = inside ? v_render[(size_t)p * CH + k] : 0.f; }
  const float va_pix = (inside && v_alphas) ? v_alphas[p] : 0.f;
  // a pixel nothing contributed to has alpha == 0 exactly; mark it with last = -1
  const int bin_final = (inside && alphas[p] > 0.f) ? last_ids[p] : -1;
  int wave_last = bin_final;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) wave_last = max(wave_last, __shfl_xor(wave_last, d, 64));

  const int n_batches = (end - start + kTilePix - 1) / kTilePix;
  for (int b = 0; b < n_batches; ++b) {
    __syncthreads();
    const int batch_end = end - 1 - kTilePix * b;
    const int size = min(kTilePix, batch_end + 1 - start);
    const int idx = batch_end - tid;
    if (idx >= start) {
      const int g = flat[idx];
      const float4 s0 = splat[2 * g], s1 = splat[2 * g + 1];
      sG[tid] = g;
      sA[tid] = s0;
      sB[tid] = make_float4(s1.x, s1.y, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < CH; ++k) sC[tid * CH + k] = colors ? colors[(size_t)g * CH + k] : 1.f;
    }
    __syncthreads();
    for (int t = max(0, batch_end - wave_last); t < size; ++t) {
      bool valid = inside && (batch_end - t <= bin_final);
      const float4 A = sA[t], B = sB[t];
      const float dx = A.x - px, dy = A.y - py;
      float vis = 0.f, alpha = 0.f;
      if (valid) {
        const float sigma = 0.5f * (A.z * dx * dx + B.x * dy * dy) + A.w * dx * dy;
        vis = __expf(-sigma);
        alpha = fminf(kAlphaMax, B.y * vis);
        if (sigma < 0.f || alpha < kAlphaMin) valid = false;
      }
      if (!__any(valid)) continue;
      float r_rgb[CH];
#pragma unroll
      for (int k = 0; k < CH; ++k) r_rgb[k] = 0.f;
      float gx = 0.f, gy = 0.f, ga = 0.f, gb = 0.f, gc = 0.f, go = 0.f;
      if (valid) {
        const float ra = 1.f / (1.f - alpha);
        T *= ra;
        const float fac = alpha * T;
        float v_alpha = 0.f;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          r_rgb[k] = fac * vr[k];
          v_alpha += (sC[t * CH + k] * T - buffer[k] * ra) * vr[k];
        }
        v_alpha += T_final * ra * va_pix;
        if (B.y * vis <= kAlphaMax) {
          const float v_sigma = -B.y * vis * v_alpha;
          gx = v_sigma * (A.z * dx + A.w * dy);
          gy = v_sigma * (A.w * dx + B.x * dy);
          ga = 0.5f * v_sigma * dx * dx;
          gb = v_sigma * dx * dy;
          gc = 0.5f * v_sigma * dy * dy;
          go = vis * v_alpha;
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) buffer[k] += sC[t * CH + k] * fac;
      }
      const float agx = wave_sum(fabsf(gx)), agy = wave_sum(fabsf(gy));
      gx = wave_sum(gx); gy = wave_sum(gy);
      ga = wave_sum(ga); gb = wave_sum(gb); gc = wave_sum(gc); go = wave_sum(go);
#pragma unroll
      for (int k = 0; k < CH; ++k) r_rgb[k] = wave_sum(r_rgb[k]);
      if (lane == 0) {
        const int g = sG[t];
        float *dst = g2d + (size_t)g * 8;
        unsafeAtomicAdd(dst + 0, gx);
        unsafeAtomicAdd(dst + 1, gy);
        unsafeAtomicAdd(dst + 2, agx);
        unsafeAtomicAdd(dst + 3, agy);
        unsafeAtomicAdd(dst + 4, ga);
        unsafeAtomicAdd(dst + 5, gb);
        unsafeAtomicAdd(dst + 6, gc);
        unsafeAtomicAdd(dst + 7, go);
        if (v_colors) {
#pragma unroll
          for (int k = 0; k < CH; ++k) unsafeAtomicAdd(v_colors + (size_t)g * CH + k, r_rgb[k]);
        }
      }
    }
  }
}

}  // namespace eg

using namespace eg;

extern "C" int64_t eg_composite_workspace_ctl_bytes(int64_t max_items, int64_t n_tiles) {
  if (max_items < 0 || n_tiles < 0) return 0;
  return ctl_bytes_aligned(max_items, n_tiles);
}

extern "C" int64_t eg_composite_workspace_bytes(int64_t max_items, int64_t n_tiles) {
  if (max_items < 0 || n_tiles < 0) return 0;
  return eg_composite_workspace_ctl_bytes(max_items, n_tiles) +
         max_items * kTilePix * (int64_t)(sizeof(float) + sizeof(int32_t)) +
         n_tiles * kTilePix * (int64_t)sizeof(StopInfo) + max_items * (int64_t)sizeof(int2) + max_items * (int64_t)kSlice +
         8 + ((max_items >> kAnchorShift) + 2) * kTilePix * (int64_t)sizeof(unsigned long long);
}

// unit colours: slice-parallel forward (slice products + combine by the tile's last workgroup -> exact-stop re-walk)
static int launch_sliced_fwd(const float4 *splat, const TileTable tt, int channels, const int32_t *flatten_ids,
                             int width, int height, float *render, float *alphas, int32_t *last_ids, const float *gt,
                             const float *wmap, float loss_scale, float *vpix, float *loss_out, const int32_t *total,
                             int64_t max_items, void *workspace, float *gtstop, int rewalk_hint, hipStream_t s,
                             const Batch &bt = Batch{}, int C = 1, int max_tile_hint = 0, int chain_tag = 0) {
  // fused phase B saves a launch and the idle tail of a kernel (~7 us at the reference's sizes, 14 us at 1600x1200)
  // but the last workgroup of a tile reads its slices back at device scope, one round trip per 8 slices, at the very
  // end of the tile's critical path: with a 35-slice tile (500 k Gaussians @1200x680) it cost 120 us.  Decided from
  // the largest tile population the caller has seen; without that knowledge, from the size of the launch.
  const bool fused = max_tile_hint > 0 ? max_tile_hint <= 24 * kSlice : (int64_t)max_items * C <= 3 * 2048;
  const int tw = cdiv(width, kTile), th = cdiv(height, kTile);
  const SliceWs ws = carve_workspace(workspace, max_items, tw * th);
  // the re-walk grid strides over the compact list: sized from the caller's hint (launching 1024 workgroups that
  // find an empty list costs 4.5 us, 64 cost 1.3 us); any grid is correct
  // rewalk_hint == EG_REWALK_SPECULATE: no launch at all; a pixel that does stop raises control word 3 instead
  const int skip = rewalk_hint == EG_REWALK_SPECULATE;
#ifdef EG_DEV_SWITCHES
  static const bool old_fwd = getenv("EG_FWD_OLD") && atoi(getenv("EG_FWD_OLD")) != 0;  // A/B against round 2's kernels
#else
  constexpr bool old_fwd = false;
#endif
  // The training step (no images wanted, fused loss, segmented tables): the wave-autonomous forward of
  // composite_wave.hip -- speculative while no pixel reaches the transmittance stop, chained (exact stop inside) otherwise
  if (!old_fwd && wave_forward_selected(channels, render, alphas, last_ids, vpix, gtstop, wmap, tt.item_rec, chain_tag))
    return launch_wave_fwd(splat, tt, flatten_ids, width, height, gt, wmap, loss_scale, total, max_items, workspace, gtstop,
                           !skip, (unsigned)chain_tag, max_tile_hint, s, bt, C);
  // (rewalk_hint == 0 without speculation -- a caller without a journal, e.g. the data-parallel leg, that has seen no
  // stop lately: the fused slice kernel + a 64-workgroup re-walk launch that finds an empty list is 3 us cheaper than
  // the chained kernel's look-back, and still exact should a pixel stop after all)
  if (!skip && chain_tag > 0 && rewalk_hint != 0) {
    // pixels do reach the stop: slice products, phase B and the exact stop in one kernel (decoupled look-back)
    if (channels == 1)
      composite_chained_fwd_kernel<1><<<dim3((unsigned)max_items, C), 256, 0, s>>>(
          splat, tt, total, flatten_ids, width, height, tw, th, ws, chain_tag, render, alphas, last_ids, gt, wmap,
          loss_scale, vpix, loss_out, (StopRec *)gtstop, bt);
    else
      composite_chained_fwd_kernel<3><<<dim3((unsigned)max_items, C), 256, 0, s>>>(
          splat, tt, total, flatten_ids, width, height, tw, th, ws, chain_tag, render, alphas, last_ids, gt, wmap,
          loss_scale, vpix, loss_out, (StopRec *)gtstop, bt);
    timing_mark(kMarkSlice, s);
    timing_mark(kMarkRewalk, s);
    return check_launch("composite_fwd(chained)");
  }
  int64_t want = rewalk_hint < 0 ? 256 : (rewalk_hint == 0 ? 64 : 2 * (int64_t)rewalk_hint);
  want = want < 64 ? 64 : (want > 1024 ? 1024 : want);
  const unsigned rewalk_grid = (unsigned)(max_items < want ? max_items : want);
#define EG_LAUNCH_CB(CH)                                                                                          \
  do {                                                                                                            \
    if (fused)                                                                                                    \
      composite_slice_fwd_kernel<CH, true><<<dim3((unsigned)max_items, C), 256, 0, s>>>(                          \
          splat, tt, total, flatten_ids, width, height, tw, th, ws, render, alphas, last_ids, gt, wmap,           \
          loss_scale, vpix, loss_out, (StopRec *)gtstop, bt, skip);                                               \
    else {                                                                                                        \
      composite_slice_fwd_kernel<CH, false><<<dim3((unsigned)max_items, C), 256, 0, s>>>(                         \
          splat, tt, total, flatten_ids, width, height, tw, th, ws, render, alphas, last_ids, gt, wmap,           \
          loss_scale, vpix, loss_out, (StopRec *)gtstop, bt, skip);                                               \
      composite_combine_fwd_kernel<CH><<<dim3(tw * th, C), 256, 0, s>>>(tt, flatten_ids, width, height, tw, ws,   \
                                                                        render, alphas, last_ids, gt, wmap,       \
                                                                        loss_scale, vpix, loss_out,               \
                                                                        (StopRec *)gtstop, bt, skip);             \
    }                                                                                                             \
    timing_mark(kMarkSlice, s);                                                                                   \
    if (!skip)                                                                                                    \
      composite_rewalk_fwd_kernel<CH><<<dim3(rewalk_grid, C), 256, 0, s>>>(                                       \
          splat, tt, flatten_ids, width, height, tw, th, ws, render, alphas, last_ids, gt, wmap, loss_scale,      \
          vpix, loss_out, (StopRec *)gtstop, bt);                                                                 \
    timing_mark(kMarkRewalk, s);                                                                                  \
  } while (0)
  if (channels == 1) EG_LAUNCH_CB(1); else EG_LAUNCH_CB(3);
#undef EG_LAUNCH_CB
  return check_launch("composite_fwd(sliced)");
}

extern "C" int eg_composite_fwd(const float *splat, const float *colors, int32_t channels, const int32_t *offsets,
                                const int32_t *flatten_ids, int32_t width, int32_t height, float *render,
                                float *alphas, int32_t *last_ids, const float *gt, const float *wmap,
                                float loss_scale, float *vpix, float *loss_out, const int32_t *item_offsets,
                                const int32_t *total, int64_t max_items, void *workspace, float *gtstop,
                                int32_t rewalk_hint, eg_stream_t stream) {
  EG_REQUIRE(width > 0 && height > 0, "bad sizes");
  EG_REQUIRE(channels == 1 || channels == 3, "channels must be 1 or 3");
  EG_REQUIRE(splat && offsets, "null pointer");
  EG_REQUIRE((render && alphas && last_ids) || gtstop, "render / alphas / last_ids are optional only with gtstop");
  EG_REQUIRE(!wmap || gt, "wmap needs gt");
  EG_REQUIRE(!gtstop || (!colors && item_offsets && total && workspace && max_items > 0),
             "gtstop needs the slice-parallel unit-colour mode");
  const int tw = cdiv(width, kTile), th = cdiv(height, kTile);
  hipStream_t s = as_stream(stream);
  if (!colors && item_offsets && total && workspace && max_items > 0) {
    const TileTable tt = {offsets, offsets + 1, item_offsets, item_offsets + 1, nullptr, nullptr, nullptr};
    return launch_sliced_fwd((const float4 *)splat, tt, channels, flatten_ids, width, height, render, alphas, last_ids,
                             gt, wmap, loss_scale, vpix, loss_out, total, max_items, workspace, gtstop, rewalk_hint, s);
  }
#define EG_LAUNCH_FWD(CH, UNIT)                                                                              \
  composite_fwd_kernel<CH, UNIT><<<tw * th, 256, 0, s>>>((const float4 *)splat, colors, offsets, flatten_ids, \
                                                        width, height, tw, th, render, alphas, last_ids, gt,  \
                                                        wmap, loss_scale, vpix, loss_out)
  if (channels == 1) { if (colors) EG_LAUNCH_FWD(1, false); else EG_LAUNCH_FWD(1, true); }
  else               { if (colors) EG_LAUNCH_FWD(3, false); else EG_LAUNCH_FWD(3, true); }
#undef EG_LAUNCH_FWD
  return check_launch("composite_fwd");
}

namespace eg {
// the training step's case -- no images wanted, fused loss, item records, a call tag -- takes the wave-autonomous forward
// (composite_wave.hip); step.hip asks the same question to let the sort kernel skip the empty tiles (binning.hip, skip_empty)
bool wave_forward_selected(int channels, const void *render, const void *alphas, const void *last_ids, const void *vpix,
                           const void *gtstop, const void *wmap, const void *item_rec, int chain_tag) {
#ifdef EG_DEV_SWITCHES
  static const bool old_fwd = getenv("EG_FWD_OLD") && atoi(getenv("EG_FWD_OLD")) != 0;
  if (old_fwd) return false;
#endif
  return channels == 1 && !render && !alphas && !last_ids && !vpix && gtstop && wmap && item_rec && chain_tag > 0;
}
}  // namespace eg

namespace eg {
int composite_fwd_segments_hinted(const float *splat, const int32_t *tile_start, const int32_t *tile_end,
                                  const int32_t *item_first, const int32_t *item_end, const int32_t *item_tile,
                                  const int32_t *flatten_ids, int32_t width, int32_t height, float *render, float *alphas,
                                  int32_t *last_ids, const float *gt, const float *wmap, float loss_scale, float *vpix,
                                  float *loss_out, const int32_t *total, int64_t max_items, void *workspace,
                                  float *gtstop, int32_t rewalk_hint, int32_t max_tile_hint, int32_t chain_tag,
                                  hipStream_t st, int32_t *cursor_reset, const int32_t *item_rec, int32_t seg_cap) {
  const TileTable tt = {tile_start, tile_end, item_first, item_end, item_tile, cursor_reset, (const int4 *)item_rec, seg_cap};
  return launch_sliced_fwd((const float4 *)splat, tt, 1, flatten_ids, width, height, render, alphas, last_ids, gt, wmap,
                           loss_scale, vpix, loss_out, total, max_items, workspace, gtstop, rewalk_hint, st, Batch{}, 1,
                           max_tile_hint, chain_tag);
}
}  // namespace eg

extern "C" int eg_composite_fwd_segments(const float *splat, const int32_t *tile_start, const int32_t *tile_end,
                                         const int32_t *item_first, const int32_t *item_end,
                                         const int32_t *item_tile, const int32_t *flatten_ids, int32_t width,
                                         int32_t height, float *render, float *alphas, int32_t *last_ids,
                                         const float *gt, const float *wmap, float loss_scale, float *vpix,
                                         float *loss_out, const int32_t *total, int64_t max_items, void *workspace,
                                         float *gtstop, int32_t rewalk_hint, eg_stream_t stream) {
  EG_REQUIRE(width > 0 && height > 0 && max_items > 0, "bad sizes");
  EG_REQUIRE(splat && tile_start && tile_end && item_first && item_end && item_tile && flatten_ids && total &&
                 workspace,
             "null pointer");
  EG_REQUIRE((render && alphas && last_ids) || gtstop, "render / alphas / last_ids are optional only with gtstop");
  EG_REQUIRE(!wmap || gt, "wmap needs gt");
  // (gtstop without the fused loss: the record carries T_final itself, upstream gradient 1 -- the drop-in operator's
  // backward scales it by what autograd hands it)
  const TileTable tt = {tile_start, tile_end, item_first, item_end, item_tile, nullptr, nullptr};
  return launch_sliced_fwd((const float4 *)splat, tt, 1, flatten_ids, width, height, render, alphas, last_ids, gt, wmap,
                           loss_scale, vpix, loss_out, total, max_items, workspace, gtstop, rewalk_hint,
                           as_stream(stream));
}

extern "C" int eg_composite_bwd(const float *splat, const int32_t *offsets, const int32_t *flatten_ids,
                                int32_t width, int32_t height, const float *alphas, const int32_t *last_ids,
                                const float *vpix, float *g2d, const int32_t *item_offsets, const int32_t *total,
                                int64_t max_items, eg_stream_t stream) {
  EG_REQUIRE(width > 0 && height > 0, "bad sizes");
  EG_REQUIRE(splat && offsets && alphas && last_ids && vpix && g2d, "null pointer");
  const int tw = cdiv(width, kTile), th = cdiv(height, kTile);
  if (item_offsets && total && max_items > 0)
    composite_bwd_item_kernel<<<(unsigned)max_items, 256, 0, as_stream(stream)>>>(
        (const float4 *)splat, offsets, item_offsets, total, flatten_ids, width, height, tw, th, alphas, last_ids,
        vpix, g2d);
  else
    composite_bwd_unit_kernel<<<tw * th, 256, 0, as_stream(stream)>>>((const float4 *)splat, offsets, flatten_ids,
                                                                      width, height, tw, th, alphas, last_ids,
                                                                      vpix, g2d);
  return check_launch("composite_bwd");
}

extern "C" int eg_composite_bwd_footprint(const float *splat, int32_t N, int32_t width, int32_t height,
                                          const float *gtstop, float *g2d, eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && width > 0 && height > 0, "bad sizes");
  EG_REQUIRE((int64_t)width * height * (int64_t)sizeof(StopRec) < (int64_t)kOutOfImage,
             "image above 2^31 / 12 pixels (the record image is addressed by 32-bit byte offsets)");
  if (N == 0) return EG_OK;
  EG_REQUIRE(splat && gtstop && g2d, "null pointer");
  hipStream_t st = as_stream(stream);
  footprint_bwd_kernel<<<cdiv((int64_t)N, 32), 256, 0, st>>>((const float4 *)splat, N, width, height,
                                                            (const StopRec *)gtstop, g2d, Batch{}, nullptr, nullptr);
  timing_mark(kMarkFootprint, st);
  return check_launch("composite_bwd_footprint");
}

namespace eg {
// batched step: the slice-parallel forward and the footprint backward of C views in one launch each
int launch_composite_fwd_segments(const float *splat, const int32_t *tile_start, const int32_t *tile_end,
                                  const int32_t *item_first, const int32_t *item_end, const int32_t *item_tile,
                                  const int32_t *flatten_ids, int32_t width, int32_t height, float loss_scale,
                                  float *loss_out, const int32_t *total, int64_t max_items, void *workspace,
                                  float *gtstop, int32_t rewalk_hint, const Batch &bt, int C, hipStream_t st,
                                  int32_t max_tile_hint, int32_t chain_tag, int32_t *cursor_reset,
                                  const int32_t *item_rec, int32_t seg_cap) {
  const TileTable tt = {tile_start, tile_end, item_first, item_end, item_tile, cursor_reset, (const int4 *)item_rec, seg_cap};
  return launch_sliced_fwd((const float4 *)splat, tt, 1, flatten_ids, width, height, nullptr, nullptr, nullptr,
                           bt.gt[0], bt.wmap[0], loss_scale, nullptr, loss_out, total, max_items, workspace, gtstop,
                           rewalk_hint, st, bt, C, max_tile_hint, chain_tag);
}
int launch_footprint_bwd(const float *splat, int32_t N, int32_t width, int32_t height, const float *gtstop, float *g2d,
                         const Batch &bt, int C, hipStream_t st, void *workspace, int64_t max_items, float *loss_out) {
  if ((int64_t)width * height * (int64_t)sizeof(StopRec) >= (int64_t)kOutOfImage) {
    set_error("composite_bwd_footprint: image above 2^31 / 12 pixels");
    return EG_ERR_ARG;
  }
  float *loss_part = nullptr;
  if (workspace && loss_out) loss_part = carve_workspace(workspace, max_items, cdiv(width, kTile) * cdiv(height, kTile)).loss_part;
  footprint_bwd_kernel<<<dim3(cdiv((int64_t)N, 32), C), 256, 0, st>>>((const float4 *)splat, N, width, height,
                                                                     (const StopRec *)gtstop, g2d, bt, loss_part, loss_out);
  timing_mark(kMarkFootprint, st);
  return check_launch("composite_bwd_footprint");
}
}  // namespace eg

extern "C" int eg_composite_bwd_colors(const float *splat, const float *colors, int32_t channels,
                                       const int32_t *offsets, const int32_t *flatten_ids, int32_t width,
                                       int32_t height, const float *alphas, const int32_t *last_ids,
                                       const float *v_render, const float *v_alphas, float *g2d, float *v_colors,
                                       eg_stream_t stream) {
  EG_REQUIRE(width > 0 && height > 0, "bad sizes");
  EG_REQUIRE(channels == 1 || channels == 3, "channels must be 1 or 3");
  EG_REQUIRE(splat && offsets && alphas && last_ids && v_render && g2d, "null pointer");
  const int tw = cdiv(width, kTile), th = cdiv(height, kTile);
  hipStream_t s = as_stream(stream);
  if (channels == 1)
    composite_bwd_colors_kernel<1><<<tw * th, 256, 0, s>>>((const float4 *)splat, colors, offsets, flatten_ids,
                                                          width, height, tw, th, alphas, last_ids, v_render,
                                                          v_alphas, g2d, v_colors);
  else
    composite_bwd_colors_kernel<3><<<tw * th, 256, 0, s>>>((const float4 *)splat, colors, offsets, flatten_ids,
                                                          width, height, tw, th, alphas, last_ids, v_render,
                                                          v_alphas, g2d, v_colors);
  return check_launch("composite_bwd_colors");
}
