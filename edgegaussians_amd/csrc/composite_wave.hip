// G7 forward of the training step, wave-autonomous form (round 3; hand-over rebuilt in round 4).
//
// Replaces gsplat 1.0.0 rasterize_to_pixels_fwd as reached from edgegaussians/models/edge_gs.py:250-268 with the
// clamp (edge_gs.py:279) and the projection loss (edge_gs.py:288-324, weight-map form) fused into the epilogue --
// the same semantics as composite.hip's slice / chained kernels (restated in oracle/ref_torch.py:230-319), which stay
// the path of the general C ABI (render / alphas / last_ids outputs, arbitrary colours, re-walk list).
//
//   wave (slice = one 128-Gaussian item of a tile's depth-sorted list, quadrant = 8x8 pixels of the tile)
//     head    : ONE 16-byte item record left by the sort kernel {tile, item | items << 16, call tag, end of the tile} -- valid iff
//               it carries this call's tag (round 5: XCD-aware placement with holes, no record for an empty tile: binning.hip)
//     stage   : every thread of the workgroup gathers one half record of the slice into the workgroup's LDS copy (ONE
//               barrier, the only one of the kernel), then every wave tests all 128 Gaussians against ITS quadrant (the
//               exact ellipse-vs-rectangle test) and ballot-compacts the hits into its OWN list in LDS, in slice order
//     walk    : 64 pixels x listed Gaussians, four list entries per iteration, broadcast LDS reads.  The entry holds
//               the conic premultiplied by -log2(e) and log2(opacity), so that alpha = exp2(quadratic form): 14 VALU
//               operations per (pixel, Gaussian) instead of 18, three b128 LDS reads per PAIR of entries instead of four
//     hand-over (tiles with more than one slice), per quadrant, no workgroup involved, no flag, no ticket, no drain:
//               a slice that has slices behind it publishes one DATA-TAGGED 8-byte AGGREGATE granule per pixel
//               {product, tag << 9 | slice-local index of its last contributor} with a single device-scope store and,
//               in speculative mode, is done.  A reader polls the granules themselves (the tag is the call's).
//               Speculative mode: only the tile's LAST slice looks back, multiplies the products in depth order and
//               finalises the 64 pixels (a product that crosses 1e-4 raises the sticky miss word and the caller replays
//               the step in chained mode).
//               Chained mode (round 4): EVERY slice needs T_in, the product of the slices in front of it in depth order
//               with the stop rule applied after every factor.  Every 8th slice of a tile (an ANCHOR) also publishes, once
//               it knows its T_in, an INCLUSIVE granule {fl(T_in * P) or 0 = "stopped at or in front of me", tag << 16 |
//               last contributor so far (slice << 7 | index)}.  Slice s reads the inclusive granule of the nearest anchor
//               a in front of s - 1 and the aggregates of slices a + 1 .. s - 1 -- at most 8 granules, all in flight
//               together -- and folds them in depth order: the same sequence of fp32 roundings whatever the anchor (an
//               inclusive granule IS the left fold up to its slice), so every slice of a tile takes the same view of where
//               a pixel stops.  Round 3 read ALL s aggregates in front (O(ns^2 / 2) block reads per tile, 3.9x the
//               forward's algorithmic traffic at config 2) and issued 8 loads per batch whatever the batch held.
//               DEAD SLICES: (tile, quadrant) keeps one word "first slice that lies behind every pixel's stop", raised by
//               the wave that finds out; every wave reads it (non-blocking) at its head and leaves at once if it is at or
//               behind that slice -- exact, of THIS call (round 3 guessed from the previous call's other view).  What
//               makes the look pay is the DISPATCH ORDER (binning.hip): the deep slices of all tiles are dispatched behind
//               the front slices of all tiles, so by the time a deep slice starts the word is there (500 k Gaussians
//               @1200x680: 202 -> 119 us; before that, half of the launch's wave slots held deep slices waiting for the
//               slices in front).  Waiting for the anchor before staging -- round 3's "gating" -- then buys nothing
//               (measured equal) and is gone.
//
// Slices in front have lower record indices: they were dispatched earlier and wait on nobody behind them (the
// decoupled look-back idiom; binning.hip keeps that order, tests/test_gpu_parity.py::
// test_item_records_follow_the_dispatch_order_contract).  A poll that has not seen its granule after kSpinLimit rounds
// raises bit 1 of control word 3 (sticky; the trainer's read-back raises on it) and carries on with a neutral value
// instead of hanging the GPU.  Per-pixel loss terms are summed per wave and added to one of 64 partial sums (4 x tiles
// same-address atomics would serialise at ~12 ns each); the footprint backward folds the partials into the caller's
// accumulator.
#include <cstdlib>

#include "common.h"
#include "composite.h"

namespace eg {

// one wave's compacted list: PAIRS of Gaussians side by side (three broadcast ds_read_b128 fetch two entries)
template <int CAP>
struct WaveListT {
  float4 X[CAP / 2 + 2];        // x0 x1 y0 y1
  float4 C[CAP / 2 + 2];        // A0 A1 B0 B1     A = -log2(e) a / 2, B = -log2(e) b   (conic [[a, b], [b, c]])
  float4 D[CAP / 2 + 2];        // C0 C1 lo0 lo1   C = -log2(e) c / 2, lo = log2(opacity)
  unsigned char idx[CAP + 16];  // slice-local index of every entry
};

// The workgroup's copy of its slice: every thread gathers ONE half record (the four quadrant waves used to gather the
// whole slice each: 4x the traffic, 61 MB per launch against 14 MB algorithmic), one barrier, then every wave tests
// all of them against its quadrant out of LDS.  It stays in place: a pixel that stops behind entry k of a list records
// that Gaussian's id and depth bits for the backward from here (round 2's epilogue: two dependent global loads).
// Next to the records the copy holds what the quadrant test needs of a Gaussian but not of a quadrant -- the sigma
// threshold, the extent of the alpha >= 1/255 ellipse, the slopes of its conjugate diameters -- computed ONCE, by the
// thread that gathered the record, while the workgroup waits for memory anyway (the four quadrant waves used to redo
// them: a log, three reciprocals and two square roots per test, a third of the staging phase's issue slots).
template <int CAP>
struct WgStage {
  float4 A[CAP];  // x y a b
  float4 B[CAP];  // c log2(o) depth thr      (thr = ln(255 o) + margin)
  float4 D[CAP];  // ex ey -b/a -b/c          (ex < 0: the Gaussian reaches no pixel)
  int gid[CAP];
};
static_assert(sizeof(WaveListT<128>) % 16 == 0 && sizeof(WaveListT<256>) % 16 == 0, "float4 alignment of the lists");

constexpr float kNegLog2e = -1.44269504088896341f;
constexpr int kNoContributor = 511;  // 9-bit "this slice did not contribute to the pixel"

// does Gaussian (s0 = x y a b, s1 = c o depth radius) reach the pixel centres [qx + 0.5, qx + 7.5] x [qy + 0.5, qy + 7.5]
// with alpha >= 1/255?  AABB reject and the exact ellipse-vs-rectangle test of common.h (ellipse_hits_rect: the minimum
// of the convex quadratic over the rectangle lies at the centre or on one of the four edges) -- the same arithmetic,
// written WITHOUT branches: the lanes of a wave hold unrelated Gaussians, every early exit is a divergent branch
__device__ __forceinline__ bool quad_hit(const float4 s0, const float4 s1, float qx, float qy) {
#pragma clang fp contract(off)
  const float x = s0.x, y = s0.y, a = s0.z, b = s0.w, c = s1.x;
  const float thr = __logf(255.f * s1.y) + kThrMargin;
  const float det = a * c - b * b;
  const bool valid = (thr > 0.f) & (det > 0.f);
  const float k2 = 2.f * thr * __builtin_amdgcn_rcpf(det);  // hardware rcp / sqrt: the inflation covers 1 ulp
  const float ex = __builtin_amdgcn_sqrtf(k2 * c) * 1.001f + 0.01f;
  const float ey = __builtin_amdgcn_sqrtf(k2 * a) * 1.001f + 0.01f;
  const float rx0 = qx + 0.5f, ry0 = qy + 0.5f, rx1 = qx + 7.5f, ry1 = qy + 7.5f;
  const bool aabb = (x - ex <= rx1) & (x + ex >= rx0) & (y - ey <= ry1) & (y + ey >= ry0);
  const float u0 = rx0 - x, u1 = rx1 - x, v0 = ry0 - y, v1 = ry1 - y;
  const bool centre_in = (u0 <= 0.f) & (u1 >= 0.f) & (v0 <= 0.f) & (v1 >= 0.f);
  const float nba = -b * __builtin_amdgcn_rcpf(a), nbc = -b * __builtin_amdgcn_rcpf(c);
  const float us0 = fminf(fmaxf(nba * v0, u0), u1), us1 = fminf(fmaxf(nba * v1, u0), u1);
  const float vs0 = fminf(fmaxf(nbc * u0, v0), v1), vs1 = fminf(fmaxf(nbc * u1, v0), v1);
  const float best = fminf(fminf(sigma_at(a, b, c, us0, v0), sigma_at(a, b, c, us1, v1)),
                           fminf(sigma_at(a, b, c, u0, vs0), sigma_at(a, b, c, u1, vs1)));
  return valid & aabb & (centre_in | (best <= thr * 1.001f + 1e-3f));
}

template <class WaveList>
__device__ __forceinline__ void put_entry(WaveList &wl, int pos, const float4 s0, const float4 s1, int slice_idx) {
  const int pr = pos >> 1, sl = pos & 1;
  float *X = (float *)&wl.X[pr], *Cc = (float *)&wl.C[pr], *D = (float *)&wl.D[pr];
  X[sl] = s0.x; X[2 + sl] = s0.y;
  Cc[sl] = (0.5f * kNegLog2e) * s0.z; Cc[2 + sl] = kNegLog2e * s0.w;
  D[sl] = (0.5f * kNegLog2e) * s1.x; D[2 + sl] = __builtin_amdgcn_logf(s1.y);  // v_log_f32 = log2
  wl.idx[pos] = (unsigned char)slice_idx;
}

// the same test on a record of the workgroup's copy (WgStage: the per-Gaussian part is already there)
__device__ __forceinline__ bool quad_hit_staged(const float4 s0, const float4 sb, const float4 d, float qx, float qy) {
#pragma clang fp contract(off)
  const float x = s0.x, y = s0.y, a = s0.z, b = s0.w, c = sb.x, thr = sb.w;
  const float ex = d.x, ey = d.y, nba = d.z, nbc = d.w;
  const float rx0 = qx + 0.5f, ry0 = qy + 0.5f, rx1 = qx + 7.5f, ry1 = qy + 7.5f;
  const bool aabb = (ex >= 0.f) & (x - ex <= rx1) & (x + ex >= rx0) & (y - ey <= ry1) & (y + ey >= ry0);
  const float u0 = rx0 - x, u1 = rx1 - x, v0 = ry0 - y, v1 = ry1 - y;
  const bool centre_in = (u0 <= 0.f) & (u1 >= 0.f) & (v0 <= 0.f) & (v1 >= 0.f);
  const float us0 = fminf(fmaxf(nba * v0, u0), u1), us1 = fminf(fmaxf(nba * v1, u0), u1);
  const float vs0 = fminf(fmaxf(nbc * u0, v0), v1), vs1 = fminf(fmaxf(nbc * u1, v0), v1);
  const float best = fminf(fminf(sigma_at(a, b, c, us0, v0), sigma_at(a, b, c, us1, v1)),
                           fminf(sigma_at(a, b, c, u0, vs0), sigma_at(a, b, c, u1, vs1)));
  return aabb & (centre_in | (best <= thr * 1.001f + 1e-3f));
}

// what the gathering thread derives of a Gaussian (s0 = x y a b; c, o): {ex, ey, -b/a, -b/c}, thr -- quad_hit's arithmetic
__device__ __forceinline__ float4 stage_derive(const float4 s0, float c, float thr) {
#pragma clang fp contract(off)
  const float a = s0.z, b = s0.w;
  const float det = a * c - b * b;
  const bool valid = (thr > 0.f) & (det > 0.f);
  const float k2 = 2.f * thr * __builtin_amdgcn_rcpf(det);  // hardware rcp / sqrt: the inflation covers 1 ulp
  const float ex = __builtin_amdgcn_sqrtf(k2 * c) * 1.001f + 0.01f;
  const float ey = __builtin_amdgcn_sqrtf(k2 * a) * 1.001f + 0.01f;
  return make_float4(valid ? ex : -1.f, ey, -b * __builtin_amdgcn_rcpf(a), -b * __builtin_amdgcn_rcpf(c));
}

template <class WaveList>
__device__ __forceinline__ void put_entry_staged(WaveList &wl, int pos, const float4 s0, const float4 sb, int slice_idx) {
  const int pr = pos >> 1, sl = pos & 1;
  float *X = (float *)&wl.X[pr], *Cc = (float *)&wl.C[pr], *D = (float *)&wl.D[pr];
  X[sl] = s0.x; X[2 + sl] = s0.y;
  Cc[sl] = (0.5f * kNegLog2e) * s0.z; Cc[2 + sl] = kNegLog2e * s0.w;
  D[sl] = (0.5f * kNegLog2e) * sb.x; D[2 + sl] = sb.y;  // (log2(o) is in the copy)
  wl.idx[pos] = (unsigned char)slice_idx;
}

__device__ __forceinline__ void wave_lds_fence() {
  // LDS operations of one wave complete in order; this only keeps the compiler from moving them across
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Gaussians [start, end) of the sorted ids (at most kWaveSlice) -> this wave's list of those that reach its quadrant,
// in slice order, padded with three rejecting sentinels (the walk reads four entries at a time).  Returns the length.
template <int R, class WaveList>
__device__ __forceinline__ int stage_rounds(WaveList &wl, const float4 *__restrict__ splat, const int *__restrict__ flat,
                                            int start, int end, float qx, float qy, int lane) {
  const int n = end - start;  // >= 1, <= 64 R
  int g[R];
  float4 r0[R], r1[R];
  // Unconditional loads (a lane beyond the slice re-reads its last Gaussian): with a branch around each load the
  // compiler waits for one gather before it issues the next -- four dependent round trips instead of two
#pragma unroll
  for (int r = 0; r < R; ++r) g[r] = flat[start + min(lane + 64 * r, n - 1)];
#pragma unroll
  for (int r = 0; r < R; ++r) { r0[r] = splat[2 * g[r]]; r1[r] = splat[2 * g[r] + 1]; }
  bool hit[R];
  unsigned long long bal[R];
  int base[R + 1];
  base[0] = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    hit[r] = (lane + 64 * r < n) & quad_hit(r0[r], r1[r], qx, qy);
    bal[r] = __ballot(hit[r]);
    base[r + 1] = base[r] + __popcll(bal[r]);
  }
  const int n_mine = base[R];
  const unsigned long long lt = (1ull << lane) - 1ull;
  wave_lds_fence();  // (a list that is being re-staged: the previous walk's reads come first)
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (hit[r]) put_entry(wl, base[r] + __popcll(bal[r] & lt), r0[r], r1[r], lane + 64 * r);
  if (lane < 3) {  // sentinels: log2(opacity) = -1e30 => alpha = 0
    const int e = n_mine + lane, pr = e >> 1, sl = e & 1;
    ((float *)&wl.X[pr])[sl] = 0.f; ((float *)&wl.X[pr])[2 + sl] = 0.f;
    ((float *)&wl.C[pr])[sl] = 0.f; ((float *)&wl.C[pr])[2 + sl] = 0.f;
    ((float *)&wl.D[pr])[sl] = 0.f; ((float *)&wl.D[pr])[2 + sl] = -1e30f;
  }
  wave_lds_fence();
  return n_mine;
}

// the same from the workgroup's LDS copy of the slice (n Gaussians)
template <int R, class WaveList, class Stage>
__device__ __forceinline__ int stage_from_lds(WaveList &wl, const Stage &st, int n, float qx, float qy, int lane) {
  float4 r0[R], r1[R], rd[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int k = min(lane + 64 * r, n - 1);
    r0[r] = st.A[k];
    r1[r] = st.B[k];
    rd[r] = st.D[k];
  }
  bool hit[R];
  unsigned long long bal[R];
  int base[R + 1];
  base[0] = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    hit[r] = (lane + 64 * r < n) & quad_hit_staged(r0[r], r1[r], rd[r], qx, qy);
    bal[r] = __ballot(hit[r]);
    base[r + 1] = base[r] + __popcll(bal[r]);
  }
  const int n_mine = base[R];
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (hit[r]) put_entry_staged(wl, base[r] + __popcll(bal[r] & lt), r0[r], r1[r], lane + 64 * r);
  if (lane < 3) {  // sentinels: log2(opacity) = -1e30 => alpha = 0
    const int e = n_mine + lane, pr = e >> 1, sl = e & 1;
    ((float *)&wl.X[pr])[sl] = 0.f; ((float *)&wl.X[pr])[2 + sl] = 0.f;
    ((float *)&wl.C[pr])[sl] = 0.f; ((float *)&wl.C[pr])[2 + sl] = 0.f;
    ((float *)&wl.D[pr])[sl] = 0.f; ((float *)&wl.D[pr])[2 + sl] = -1e30f;
  }
  wave_lds_fence();
  return n_mine;
}

// a slice staged by ONE wave from memory (the rare continuation of an exact stop into the following slices)
template <class WaveList>
__device__ __forceinline__ int stage_wave(WaveList &wl, const float4 *__restrict__ splat, const int *__restrict__ flat,
                                          int start, int end, float qx, float qy, int lane) {
  if (end - start <= 64) return stage_rounds<1>(wl, splat, flat, start, end, qx, qy, lane);  // (wave-uniform)
  return stage_rounds<2>(wl, splat, flat, start, end, qx, qy, lane);
}

// alpha of one list entry at pixel (px, py), and whether it counts.  s = log2(o) - log2(e) sigma is evaluated in one
// fixed sequence everywhere (phase A and the exact-stop walk must agree bit for bit):
//   sigma >= 0  <=>  s <= log2(o);   alpha = min(0.999, exp2(s));   alpha >= 1/255
struct EntryEval {
  float a;
  bool k;
};
__device__ __forceinline__ EntryEval eval_entry(float x, float y, float A, float B, float Cq, float lo, float px, float py) {
  const float dx = x - px, dy = y - py;
  const float t = __builtin_fmaf(A, dx, B * dy);
  const float u = __builtin_fmaf(Cq * dy, dy, lo);
  const float s = __builtin_fmaf(dx, t, u);
  const float e = __builtin_amdgcn_exp2f(s);
  EntryEval r;
  r.a = fminf(kAlphaMax, e);
  r.k = (s <= lo) & (e >= kAlphaMin);  // '&': no short-circuit branch
  return r;
}

// phase A: product of (1 - alpha) over the list in depth order, and (TRACK) the list position of the last contributor
template <bool TRACK, class WaveList>
__device__ __forceinline__ void walk_list(const WaveList &wl, int n_mine, float px, float py, float &P, int &Lpos) {
  for (int t = 0; t < n_mine; t += 4) {
    const int p = t >> 1;
    const float4 X0 = wl.X[p], X1 = wl.X[p + 1], C0 = wl.C[p], C1 = wl.C[p + 1], D0 = wl.D[p], D1 = wl.D[p + 1];
    const EntryEval e0 = eval_entry(X0.x, X0.z, C0.x, C0.z, D0.x, D0.z, px, py);
    const EntryEval e1 = eval_entry(X0.y, X0.w, C0.y, C0.w, D0.y, D0.w, px, py);
    const EntryEval e2 = eval_entry(X1.x, X1.z, C1.x, C1.z, D1.x, D1.z, px, py);
    const EntryEval e3 = eval_entry(X1.y, X1.w, C1.y, C1.w, D1.y, D1.w, px, py);
    P *= e0.k ? 1.f - e0.a : 1.f;  // depth order kept: ((P m0) m1) m2 ...
    P *= e1.k ? 1.f - e1.a : 1.f;
    P *= e2.k ? 1.f - e2.a : 1.f;
    P *= e3.k ? 1.f - e3.a : 1.f;
    if (TRACK) {
      Lpos = e0.k ? t : Lpos;
      Lpos = e1.k ? t + 1 : Lpos;
      Lpos = e2.k ? t + 2 : Lpos;
      Lpos = e3.k ? t + 3 : Lpos;
    }
  }
}

// Phase A with CHECKPOINTS (chained mode): the product and the last contributor after every quarter of the list
// capacity.  A pixel whose stop falls in this slice then starts its exact walk behind the last quarter its
// transmittance survives -- at ITS OWN list position (per-lane LDS addresses): the wave walks a quarter of the list,
// where a walk in lockstep from the front spans it all (the 64 pixels of a quadrant cross at different positions).
template <class WaveList>
__device__ __forceinline__ void walk_list_ck(const WaveList &wl, int n_mine, float px, float py, float &P, int &Lpos,
                                             float (&ck)[4], unsigned &ckL) {
  constexpr int kSeg = (int)(sizeof(wl.idx) - 16) / 4;
  ckL = 0xffffffffu;
#pragma unroll
  for (int seg = 0; seg < 4; ++seg) {
    const int t1 = min(n_mine, kSeg * (seg + 1));
    for (int t = kSeg * seg; t < t1; t += 4) {
      const int p = t >> 1;
      const float4 X0 = wl.X[p], X1 = wl.X[p + 1], C0 = wl.C[p], C1 = wl.C[p + 1], D0 = wl.D[p], D1 = wl.D[p + 1];
      const EntryEval e0 = eval_entry(X0.x, X0.z, C0.x, C0.z, D0.x, D0.z, px, py);
      const EntryEval e1 = eval_entry(X0.y, X0.w, C0.y, C0.w, D0.y, D0.w, px, py);
      const EntryEval e2 = eval_entry(X1.x, X1.z, C1.x, C1.z, D1.x, D1.z, px, py);
      const EntryEval e3 = eval_entry(X1.y, X1.w, C1.y, C1.w, D1.y, D1.w, px, py);
      P *= e0.k ? 1.f - e0.a : 1.f;
      P *= e1.k ? 1.f - e1.a : 1.f;
      P *= e2.k ? 1.f - e2.a : 1.f;
      P *= e3.k ? 1.f - e3.a : 1.f;
      Lpos = e0.k ? t : Lpos;
      Lpos = e1.k ? t + 1 : Lpos;
      Lpos = e2.k ? t + 2 : Lpos;
      Lpos = e3.k ? t + 3 : Lpos;
    }
    ck[seg] = P;
    ckL = (ckL & ~(0xffu << (8 * seg))) | (((unsigned)Lpos & 0xffu) << (8 * seg));  // (-1 -> 0xff: none yet)
  }
}

// the exact stop, every lane from its own (even) list position pos: returns the position of the last contributor it
// composited (-1: none); a lane that runs off the list without stopping carries on in the following slice
template <class WaveList>
__device__ __forceinline__ int exact_walk_lane(const WaveList &wl, int n_mine, int pos, float px, float py, bool &live,
                                               float &T, bool &found) {
  constexpr int kPairs = (int)(sizeof(wl.X) / sizeof(float4));
  int lastpos = -1;
  live = live & (pos < n_mine);
  while (__ballot(live) != 0ull) {
    const int p = min(pos >> 1, kPairs - 1);
    const float4 X0 = wl.X[p], C0 = wl.C[p], D0 = wl.D[p];
    const EntryEval e0 = eval_entry(X0.x, X0.z, C0.x, C0.z, D0.x, D0.z, px, py);
    const EntryEval e1 = eval_entry(X0.y, X0.w, C0.y, C0.w, D0.y, D0.w, px, py);
    {
      const float nT = T * (1.f - e0.a);
      const bool hit = live & e0.k, stop = hit & (nT <= kTStop), upd = hit & !stop;
      T = upd ? nT : T;
      lastpos = upd ? pos : lastpos;
      found = found | stop;
      live = live & !stop;
    }
    {
      const float nT = T * (1.f - e1.a);
      const bool hit = live & e1.k, stop = hit & (nT <= kTStop), upd = hit & !stop;
      T = upd ? nT : T;
      lastpos = upd ? pos + 1 : lastpos;
      found = found | stop;
      live = live & !stop;
    }
    pos += 2;
    live = live & (pos < n_mine);
  }
  return lastpos;
}

// Sequential walk with the stop rule (gsplat: stop BEFORE compositing the Gaussian that would take T to <= 1e-4):
// lanes with `live` look for their stop from transmittance T.  Returns the list position of the last contributor
// composited here (-1: none); the wave leaves as soon as none of its lanes is looking.
template <class WaveList>
__device__ __forceinline__ int exact_walk_wave(const WaveList &wl, int n_mine, float px, float py, bool &live, float &T,
                                               bool &found) {
  int lastpos = -1;
  for (int t = 0; t < n_mine && __ballot(live) != 0ull; t += 2) {
    const int p = t >> 1;
    const float4 X0 = wl.X[p], C0 = wl.C[p], D0 = wl.D[p];
    const EntryEval e0 = eval_entry(X0.x, X0.z, C0.x, C0.z, D0.x, D0.z, px, py);
    const EntryEval e1 = eval_entry(X0.y, X0.w, C0.y, C0.w, D0.y, D0.w, px, py);
    {
      const float nT = T * (1.f - e0.a);
      const bool hit = live & e0.k, stop = hit & (nT <= kTStop), upd = hit & !stop;
      T = upd ? nT : T;
      lastpos = upd ? t : lastpos;
      found = found | stop;
      live = live & !stop;
    }
    {
      const float nT = T * (1.f - e1.a);
      const bool hit = live & e1.k, stop = hit & (nT <= kTStop), upd = hit & !stop;
      T = upd ? nT : T;
      lastpos = upd ? t + 1 : lastpos;
      found = found | stop;
      live = live & !stop;
    }
  }
  return lastpos;
}

__device__ __forceinline__ int gridDim_tiles(int tw, int height) { return tw * ((height + kTile - 1) / kTile); }

// one pixel's granule of a slice: {product, tag << 9 | index of the last contributor}
__device__ __forceinline__ unsigned long long load_granule(const unsigned long long *g) {
  return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_granule(unsigned long long *g, unsigned long long v) {
  __hip_atomic_store(g, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long make_aggregate(float P, unsigned tag, int lidx) {
  return (unsigned long long)(unsigned)__float_as_int(P) | ((unsigned long long)((tag << 9) | (unsigned)lidx) << 32);
}
// inclusive granule of an anchor slice: {T after the slice (0 = the pixel stopped at or in front of it), tag << 16 | last
// contributor so far as slice << 7 | slice-local index (0xffff = none)}
constexpr unsigned kNoFront16 = 0xffffu;
constexpr int kMaxAnchoredSlices = 510;  // (slice << 7 | index) must stay below kNoFront16
__device__ __forceinline__ unsigned long long make_inclusive(float T, unsigned tag, int front_last) {
  const unsigned f16 = front_last >= 0 ? ((unsigned)(front_last >> 9) << 7) | ((unsigned)front_last & 127u) : kNoFront16;
  return (unsigned long long)(unsigned)__float_as_int(T) | ((unsigned long long)((tag << 16) | f16) << 32);
}

// A poll gives up after this many rounds (each one s_sleep + a device-scope round trip: tens of milliseconds in all)
constexpr int kSpinLimit = 1 << 16;
__device__ __forceinline__ void spin_stalled(int *ctl) { atomicOr(&ctl[3], 2); }

// Speculative mode (and tiles of more slices than an inclusive granule can name): look back over slices [j_begin, j_end)
// of the tile (in front of the calling wave's): multiply their products onto T in depth order, eight granules in flight
// per lane; a granule that does not carry this call's tag yet is asked for again.
// `before` becomes true for a pixel once the product crosses the transmittance threshold (its walk stopped in front).
constexpr int kLook = 8;
template <bool CHAINED>
__device__ __forceinline__ void look_back(const unsigned long long *gran, int i0, int j_begin, int j_end, unsigned tag,
                                          bool inside, float &T, bool &before, int &front_last, int *ctl) {
  // front_last: the last contributor among the slices looked over so far, (slice << 9 | slice-local index), -1 = none
  for (int j8 = j_begin; j8 < j_end; j8 += kLook) {
    if (CHAINED && __ballot(!before && inside) == 0ull) break;  // every pixel of the quadrant stopped further in front
    const int nb = min(kLook, j_end - j8);  // (wave-uniform: the guards below are scalar branches, the loads stay in flight)
    unsigned long long G[kLook];
#pragma unroll
    for (int u = 0; u < kLook; ++u)
      if (u < nb) G[u] = load_granule(&gran[(size_t)(i0 + j8 + u) * kTilePix + threadIdx.x]);
#pragma unroll
    for (int u = 0; u < kLook; ++u) {
      if (u < nb) {
        int spins = 0;
        while ((unsigned)(G[u] >> 41) != tag) {  // (rare once the first poll has come back)
          if (++spins > kSpinLimit) { spin_stalled(ctl); G[u] = make_aggregate(1.f, tag, kNoContributor); break; }
          __builtin_amdgcn_s_sleep(1);
          G[u] = load_granule(&gran[(size_t)(i0 + j8 + u) * kTilePix + threadIdx.x]);
        }
        const float nT = T * __int_as_float((int)(unsigned)G[u]);  // a slice without a contributor holds exactly 1
        before = before | (nT <= kTStop);
        T = before ? T : nT;
        if (CHAINED) {
          const int li = (int)((unsigned)(G[u] >> 32) & 511u);
          front_last = li != kNoContributor ? (((j8 + u) << 9) | li) : front_last;
        }
      }
    }
  }
}

// Chained mode: the inclusive granule of anchor slice a (a > 0) -> {T, before, front_last}
__device__ __forceinline__ void fold_inclusive(unsigned long long GB, float &T, bool &before, int &front_last) {
  T = __int_as_float((int)(unsigned)GB);
  before = T == 0.f;
  const unsigned f16 = (unsigned)(GB >> 32) & 0xffffu;
  front_last = f16 != kNoFront16 ? (int)(((f16 >> 7) << 9) | (f16 & 127u)) : -1;
}
// Chained mode, T_in of slice s (> 0): the inclusive granule of anchor a (0: none, then a + 1 .. means 0 ..) and the
// aggregates of slices a + 1 .. s - 1, at most kLook granules, all requested before the first is waited for.
__device__ __forceinline__ void look_back_anchored(const unsigned long long *gran, const unsigned long long *anchor, int i0,
                                                   int a, int s, unsigned tag, bool inside, float &T, bool &before,
                                                   int &front_last, int *ctl) {
  const int j0 = a > 0 ? a + 1 : 0, nb = s - j0;  // nb <= 8 (a == 0: s <= 8), wave-uniform
  const unsigned long long *slot = &anchor[(size_t)((i0 + a) >> kAnchorShift) * kTilePix + threadIdx.x];
  unsigned long long GB = 0ull, G[kLook];
  const bool need_anchor = a > 0;
  if (need_anchor) GB = load_granule(slot);
#pragma unroll
  for (int u = 0; u < kLook; ++u)
    if (u < nb) G[u] = load_granule(&gran[(size_t)(i0 + j0 + u) * kTilePix + threadIdx.x]);
  if (need_anchor) {
    int spins = 0;
    while ((unsigned)(GB >> 48) != tag) {
      if (++spins > kSpinLimit) { spin_stalled(ctl); GB = make_inclusive(0.f, tag, -1); break; }
      __builtin_amdgcn_s_sleep(1);
      GB = load_granule(slot);
    }
    fold_inclusive(GB, T, before, front_last);
  }
#pragma unroll
  for (int u = 0; u < kLook; ++u) {
    if (u < nb) {
      int spins = 0;
      while ((unsigned)(G[u] >> 41) != tag) {
        if (++spins > kSpinLimit) { spin_stalled(ctl); G[u] = make_aggregate(1.f, tag, kNoContributor); break; }
        __builtin_amdgcn_s_sleep(1);
        G[u] = load_granule(&gran[(size_t)(i0 + j0 + u) * kTilePix + threadIdx.x]);
      }
      const float nT = T * __int_as_float((int)(unsigned)G[u]);
      before = before | (nT <= kTStop);
      T = before ? T : nT;
      const int li = (int)((unsigned)(G[u] >> 32) & 511u);
      front_last = li != kNoContributor ? (((j0 + u) << 9) | li) : front_last;
    }
  }
}

// What the kernel reads, resolved on the host for a single view (the training step's case): one block of kernel
// arguments that the compiler loads in ONE batch at the head of the wave.  (Round 3's first version took the general
// tables -- TileTable, SliceWs, Batch -- by value and adjusted them per view behind null-pointer tests: seven dependent
// scalar loads and waits before the first byte of the item record was requested.)
struct WaveArgs {
  const float4 *splat;
  const int4 *item_rec;
  const int *total, *flat;
  int *cursor_reset;
  unsigned long long *gran;    // [max_items][256] aggregate granules
  unsigned long long *anchor;  // [max_items / 8 + 2][256] inclusive granules of the anchor slices (chained mode)
  int *dead, *ctl;             // dead: [T][4] first dead slice of (tile, quadrant), tag << 15 | (32767 - slice)
  float *loss_part;
  const float *gt, *wmap;      // wmap == nullptr: no fused loss (the drop-in operator): the record carries T_final itself
  float *alphas;               // optional [H,W] accumulated alpha 1 - T_final (the operator's image)
  StopRec *gtstop;
  unsigned long long *prof;
  int width, height, tw, n_tiles;
  unsigned tag;
  float loss_scale;
  int max_anchored;  // tiles of more slices look back over all aggregates (kMaxAnchoredSlices; 0 in an A/B build leg)
  const int *item_first;  // [T]: the tile's first item in the contiguous per-tile numbering (hand-over storage)
  int seg_cap;            // keys per tile segment: slice s of tile t starts at key t * seg_cap + 128 s
};

__device__ __forceinline__ int dead_key(unsigned tag, int slice) { return (int)((tag << 15) | (unsigned)(32767 - min(slice, 32767))); }
// first dead slice recorded by THIS call in word `key` (0x7fffffff: none yet)
__device__ __forceinline__ int dead_from(int key, unsigned tag) {
  return ((unsigned)key >> 15) == tag ? 32767 - (key & 32767) : 0x7fffffff;
}

// TIMED (EG_FWD_PROF=1, debugging only): shader-clock ticks per phase of every wave of the LAST launch, one 8-word
// record per wave in prof[(item * 4 + quadrant) * 8 ...] (plain stores: atomics on shared words would serialise and
// be measured themselves); word 7 = 1 marks a wave that ran (read by eg_debug_fwd_profile)
template <bool CHAINED, bool TIMED, bool BATCHED>
__device__ __forceinline__ void wave_fwd_body(WaveArgs a, const Batch &bt, WaveListT<kSlice> *lists, WgStage<kSlice> &stg) {
  typedef WaveListT<kSlice> WaveList;
  long long t_prev = TIMED ? (long long)__builtin_readcyclecounter() : 0;
  unsigned long long *my_prof = TIMED ? a.prof + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 : nullptr;
#define EG_TICK(k)                                                                                        \
  do {                                                                                                    \
    if (TIMED) {                                                                                          \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                         \
      const long long now_ = (long long)__builtin_readcyclecounter();                                     \
      if ((threadIdx.x & 63) == 0) my_prof[k] += (unsigned long long)(now_ - t_prev);                     \
      t_prev = now_;                                                                                      \
    }                                                                                                     \
  } while (0)
  if (BATCHED) {  // view blockIdx.y of a batched step: [C, ...] work buffers
    const long long bv = blockIdx.y;
    a.total += 4 * bv; a.flat += bv * bt.keys; a.splat += bv * bt.splat4; a.gtstop += bv * bt.pixels;
    a.item_rec += bv * bt.items; a.cursor_reset += bv * bt.tiles; a.item_first += bv * bt.tiles;
    a.gran = (unsigned long long *)((char *)a.gran + bv * bt.ws_bytes);
    a.anchor = (unsigned long long *)((char *)a.anchor + bv * bt.ws_bytes);
    a.dead = (int *)((char *)a.dead + bv * bt.ws_bytes); a.ctl = (int *)((char *)a.ctl + bv * bt.ws_bytes);
    a.loss_part = (float *)((char *)a.loss_part + bv * bt.ws_bytes);
    a.gt = bt.gt[bv]; a.wmap = bt.wmap[bv];
    if (a.alphas) a.alphas += bv * bt.pixels;
  }
  const float4 *__restrict__ splat = a.splat;
  const int *__restrict__ flat = a.flat;
  const float *__restrict__ gt = a.gt, *__restrict__ wmap = a.wmap;
  StopRec *__restrict__ gtstop = a.gtstop;
  const int width = a.width, height = a.height, tw = a.tw;
  const unsigned tag = a.tag;
  const float loss_scale = a.loss_scale;
  const int b = blockIdx.x;  // the workgroup's item RECORD (records are in dispatch order: front slices first, binning.hip)
  // where this item lives: ONE 16-byte record left by the sort kernel.  Requested together with the item count (the grid
  // covers max_items, the table has max_items entries: a stale record beyond the count is read and dropped) -- one
  // dependent round trip less at the head of every wave.
  // (round 5: a record is valid iff its word 2 carries THIS call's tag -- the table has holes where an XCD's list ends
  // before the longest one, and beyond the last record it holds the records of earlier calls; the item count is not needed)
  const int4 ir = a.item_rec[b];
  if ((unsigned)ir.z != a.tag) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (TIMED && lane < 8) my_prof[lane] = lane == 7 ? 1ull : 0ull;
  WaveList &wl = lists[wv];
  const int tile = ir.x, s_me = ir.y & 0xffff, ns = ir.y >> 16;
  const int start = tile * a.seg_cap + s_me * kSlice, t_end = ir.w, end = min(t_end, start + kSlice);
  // ---- chained mode, DEAD SLICES: the (tile, quadrant)'s "first slice behind every pixel's stop" of THIS call, if a
  // wave in front has found it out already (a non-blocking look: all four words, so that the workgroup can take the
  // decision to leave without talking to itself).  Requested first: the answer is waited for before anything else.
  int dead_keys = 0;
  if (CHAINED && s_me > 0) dead_keys = __hip_atomic_load(&a.dead[tile * 4 + (lane & 3)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // the tile's slices hand over through granule blocks i0 .. i0 + ns - 1 (the contiguous per-tile item numbering; this
  // workgroup's RECORD index b follows the dispatch order instead).  Needed after the walk: in flight with the ids.
  const int i0 = a.item_first[tile], t_start = start - s_me * kSlice;
  const int ty = tile / tw, tx = tile - ty * tw;
  // wave wv owns the 8x8 quadrant (wv & 1, wv >> 1) of the tile, lane l the pixel (l & 7, l >> 3) inside it
  const int qj = tx * kTile + ((wv & 1) << 3), qi = ty * kTile + ((wv >> 1) << 3);
  const int i = qi + (lane >> 3), j = qj + (lane & 7);
  const bool inside = (i < height) && (j < width);
  const float px = (float)j + 0.5f, py = (float)i + 0.5f;
  if (a.cursor_reset && s_me == 0 && threadIdx.x == 0) a.cursor_reset[tile] = 0;
  // who finalises pixels here: in speculative mode only the tile's last slice, in chained mode any slice may
  const bool finisher = CHAINED || s_me == ns - 1;
  // the pixel's target and loss weight do not depend on the slices: in flight under everything else
  const int p = i * width + j;
  const bool has_loss = wmap != nullptr;  // (uniform)
  const float w_p = (has_loss && finisher && inside) ? wmap[p] : 0.f;
  const float gt_p = (has_loss && finisher && inside) ? gt[p] : 0.f;

  unsigned long long *gran = a.gran;
  const int b_store = i0 + s_me;
  // anchors: every 8th slice (tiles of more slices than an inclusive granule can name fall back to the full look-back)
  const bool anchored = CHAINED && ns <= a.max_anchored;
  const bool publishes_anchor = anchored && s_me > 0 && (s_me & ((1 << kAnchorShift) - 1)) == 0 && s_me < ns - 1;
  const int my_anchor = (anchored && s_me > (1 << kAnchorShift)) ? ((s_me - 1) & ~((1 << kAnchorShift) - 1)) : 0;
  unsigned long long *anchor_slot = &a.anchor[(size_t)(b_store >> kAnchorShift) * kTilePix + threadIdx.x];
  // what a slice that turns out to lie behind every stop of its quadrant leaves for the slices behind it: they are dead
  // too, but one that has not seen the news may be polling
  auto publish_dead = [&]() {
    if (s_me < ns - 1) store_granule(&gran[(size_t)b_store * kTilePix + threadIdx.x], make_aggregate(1.f, tag, kNoContributor));
    if (publishes_anchor) store_granule(anchor_slot, make_inclusive(0.f, tag, -1));
  };
  bool my_dead = false;
  if (CHAINED && s_me > 0) {
    const int k0 = __builtin_amdgcn_readlane(dead_keys, 0), k1 = __builtin_amdgcn_readlane(dead_keys, 1);
    const int k2 = __builtin_amdgcn_readlane(dead_keys, 2), k3 = __builtin_amdgcn_readlane(dead_keys, 3);
    const bool d0 = s_me >= dead_from(k0, tag), d1 = s_me >= dead_from(k1, tag);
    const bool d2 = s_me >= dead_from(k2, tag), d3 = s_me >= dead_from(k3, tag);
    my_dead = wv == 0 ? d0 : (wv == 1 ? d1 : (wv == 2 ? d2 : d3));
    if (d0 & d1 & d2 & d3) {  // (workgroup-uniform) nothing of this item can reach a pixel: not even its records are wanted
      publish_dead();
      EG_TICK(0);
      return;
    }
  }

  float T = 1.f, l = 0.f;
  int front_last = -1;
  bool before = false;  // the pixel stopped in a slice in front of this one
  // the workgroup's copy of the slice: one half record per thread
  if (end > start) {
    const int n = end - start;
    const int t = threadIdx.x, k = t >> 1;
    // (the two threads of a Gaussian sit in adjacent lanes; both run the shuffles, a thread beyond the slice too)
    const int g = flat[start + min(k, n - 1)];
    const float4 rec = splat[2 * g + (t & 1)];
    // odd thread: c o depth radius -> log2(o) and the sigma threshold ln(255 o) + margin; even thread: x y a b
    const float lo = __builtin_amdgcn_logf(rec.y);
    const float thr = (lo + 7.99435343685886f) * 0.693147180559945f + kThrMargin;  // log2(255), ln 2
    const float c_o = lane_xor1_f(rec.x), thr_o = lane_xor1_f(thr);  // (DPP, not a trip through the LDS crossbar)
    if (k < n) {
      if (t & 1) stg.B[k] = make_float4(rec.x, lo, rec.z, thr);
      else { stg.A[k] = rec; stg.D[k] = stage_derive(rec, c_o, thr_o); stg.gid[k] = g; }
    }
  }
  __syncthreads();  // (the staging barrier: every wave of the workgroup is still here)
  EG_TICK(0);  // head: the item record, the slice's records (+ the pixel's gt / weight)
  if (my_dead) {  // (this quadrant only: the wave has helped to stage the slice for the others)
    publish_dead();
    return;
  }

  // (whether the caller has been told already that pixels stop: read here, in flight with everything, not on the
  // stop-resolution path)
  const int stops_seen = CHAINED ? a.ctl[2] : 1;

  // ---- phase A: this slice's product (and last contributor) over this wave's quadrant
  float P = 1.f, ck[4] = {1.f, 1.f, 1.f, 1.f};
  unsigned ckL = 0xffffffffu;
  int Lpos = -1, n_mine = 0;
  if (end > start) {  // (an empty tile's single item has nothing to walk)
    const int n = end - start;
    if (n <= 64) n_mine = stage_from_lds<1>(wl, stg, n, (float)qj, (float)qi, lane);
    else n_mine = stage_from_lds<2>(wl, stg, n, (float)qj, (float)qi, lane);
    EG_TICK(1);  // staging: quadrant tests -> list
    if (CHAINED) walk_list_ck(wl, n_mine, px, py, P, Lpos, ck, ckL);  // (checkpoints for the exact stop)
    else walk_list<false>(wl, n_mine, px, py, P, Lpos);
    if (TIMED) { float keep = P; asm volatile("" : "+v"(keep)); P = keep; }
    EG_TICK(2);  // walk
  }
  const int Lidx = (CHAINED && Lpos >= 0) ? (int)wl.idx[Lpos] : kNoContributor;

  // ---- publish for the slices behind this one: one data-tagged granule per pixel, a single store, nothing to wait for
  if (s_me < ns - 1) store_granule(&gran[(size_t)b_store * kTilePix + threadIdx.x], make_aggregate(P, tag, Lidx));
  EG_TICK(3);  // publish
  if (!finisher) return;  // (speculative mode: whole wave)

  // ---- look back over the slices in front
  if (s_me > 0) {
    if (anchored) look_back_anchored(gran, a.anchor, i0, my_anchor, s_me, tag, inside, T, before, front_last, a.ctl);
    else look_back<CHAINED>(gran, i0, 0, s_me, tag, inside, T, before, front_last, a.ctl);
  }
  // chained mode: does the stop fall in this slice?  (decided here, ahead of the exact walk: the anchor's inclusive granule
  // is on the critical path of the eight slices behind it, the exact stop is not)
  bool cross = false;
  if (CHAINED) {
    if (!before && Lidx != kNoContributor) {
      const float nT = T * P;
      if (nT <= kTStop) cross = true; else T = nT;
    }
    if (publishes_anchor)
      store_granule(anchor_slot, make_inclusive((before || cross) ? 0.f : T, tag, Lidx != kNoContributor ? ((s_me << 9) | Lidx) : front_last));
    cross = cross && inside;
  }
  // A pixel that stops on the FIRST contributor of this slice's list names a Gaussian of a slice in front as its last
  // contributor (one stopped pixel in five, i.e. nearly every wave that resolves stops): its sorted position is known
  // from the granules just read -- the id is requested now, under the exact walk, instead of after it
  const int front_pos = front_last >= 0 ? t_start + (front_last >> 9) * kSlice + (front_last & 511) : -1;
  int front_gid = -1;
  if (CHAINED && cross && front_pos >= 0) front_gid = flat[front_pos];
  if (CHAINED && s_me > 0 && __ballot(!before && inside) == 0ull && lane == 0)  // this slice turned out dead: tell the ones behind
    atomicMax(&a.dead[tile * 4 + wv], dead_key(tag, s_me));
  EG_TICK(4);  // look-back

  if (!CHAINED) {
    // ---- speculative mode (the tile's last slice): nothing is expected to stop
    const float nT = T * P;
    const bool stop_seen = before | (nT <= kTStop);
    T = before ? T : nT;
    if (__ballot(stop_seen && inside) != 0ull) {
      // the caller speculated that no pixel would stop: tell it (sticky word 3 of the control block); it restores its
      // state and runs the step again in chained mode
      if (lane == 0) atomicOr(&a.ctl[3], 1);
    }
    if (inside)
      l = finalize_pixel<1>(p, T, 0, false, flat, nullptr, a.alphas, nullptr, has_loss, gt_p, w_p, loss_scale, nullptr, gtstop,
                            splat);
  } else {
    int last = -1;           // sorted index of the last contributor in front of a stop (only when it sits in a slice in front)
    int stop_id = -1;        // ... its Gaussian id and depth bits (read from the list in LDS)
    unsigned stop_dep = 0u;
    bool found = false;
    if (__ballot(cross) != 0ull) {
      if (lane == 0 && stops_seen == 0) atomicMax(&a.ctl[2], 1);  // "pixels do stop": the caller's launch-mode hint
      // exact stop from the list still in LDS, sequentially in depth order from T; should float rounding move the
      // crossing past the slice end, the same lanes carry on through the following slices (staged afresh)
      bool live = cross;
      int lp;
      {
        // start behind the last quarter of the list the pixel's transmittance survives
        constexpr int kSeg = kSlice / 4;
        int pos0 = 0, lp0 = 0xff;
        float Ts = T;
        bool adv = true;
#pragma unroll
        for (int seg = 0; seg < 3; ++seg) {  // (static indices only: a dynamic one would put ck[] into scratch memory)
          const float nT = T * ck[seg];
          adv = adv & (nT > kTStop);
          Ts = adv ? nT : Ts;
          pos0 = adv ? kSeg * (seg + 1) : pos0;
          lp0 = adv ? (int)((ckL >> (8 * seg)) & 0xffu) : lp0;
        }
        T = cross ? Ts : T;  // (the other lanes of the wave hold a finished transmittance)
        lp = exact_walk_lane(wl, n_mine, pos0, px, py, live, T, found);
        if (lp < 0 && lp0 != 0xff) lp = lp0;
      }
      if (lp >= 0) {  // (the workgroup's copy of this slice is still in LDS)
        const int k = (int)wl.idx[lp];
        stop_id = stg.gid[k];
        stop_dep = (unsigned)__float_as_int(stg.B[k].z);
      }
      for (int s2 = s_me + 1; s2 < ns; ++s2) {
        if (__ballot(cross && !found) == 0ull) break;
        const int st2 = t_start + s2 * kSlice, en2 = min(t_end, st2 + kSlice);
        const int n2 = stage_wave(wl, splat, flat, st2, en2, (float)qj, (float)qi, lane);
        live = cross && !found;
        const int lp2 = exact_walk_wave(wl, n2, px, py, live, T, found);
        if (lp2 >= 0) { last = st2 + (int)wl.idx[lp2]; stop_id = -1; }  // (a slice staged by this wave alone: via memory below)
      }
      // the stop is the first contributor of its slice: the last contributor sits in a slice in front
      if (cross && found && stop_id < 0 && last < 0) last = front_pos;
    }
    if (s_me + 1 < ns && __ballot(inside && !(before || (cross && found))) == 0ull && lane == 0)
      atomicMax(&a.dead[tile * 4 + wv], dead_key(tag, s_me + 1));  // the slices behind are dead
    EG_TICK(5);  // exact stop
    // finalise the pixels that stop here (whichever way the exact walk ended) and -- in the last slice -- the pixels
    // that never stop
    if (cross || (inside && !before && s_me == ns - 1)) {
      if (cross && found && stop_id < 0 && last >= 0) {  // (the last contributor sits in a slice in front)
        stop_id = last == front_pos ? front_gid : flat[last];
        stop_dep = (unsigned)__float_as_int(splat[2 * stop_id + 1].z);
      }
      const float pix = 1.f - T, c0 = fminf(fmaxf(pix, 0.f), 1.f), d = c0 - gt_p;
      // (without the fused loss the record carries T_final itself: upstream gradient 1, scaled by the caller's later)
      const float v = has_loss ? loss_scale * w_p * ((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f)) : 1.f;
      if (a.alphas) a.alphas[p] = pix;
      StopRec r;  // (finalize_pixel<1> of composite.h with the stop Gaussian already in registers)
      r.gT = (T < 1.f) ? v * T : 0.f;
      r.stop_id = (cross && found) ? stop_id : -1;
      r.stop_depth = (cross && found && stop_id >= 0) ? stop_dep : kNoStopDepth;
      gtstop[p] = r;
      l = w_p * fabsf(d);
    }
  }
  // loss terms of this wave's pixels -> one of 64 partial sums
  l = wave_sum_dpp_f(l);  // (total in lane 63)
  if (lane == 63 && l != 0.f) unsafeAtomicAdd(&a.loss_part[(tile * 4 + wv) & 63], l);
  EG_TICK(6);  // epilogue
#undef EG_TICK
}

// (eight waves per SIMD -- 64 VGPRs -- are worth more than the few registers the compiler would like on top)
template <bool CHAINED, bool TIMED, bool BATCHED>
__global__ void __launch_bounds__(256, 8)
composite_wave_fwd_kernel(const WaveArgs a, const Batch bt) {
  __shared__ WaveListT<kSlice> lists[4];
  __shared__ WgStage<kSlice> stg;
  wave_fwd_body<CHAINED, TIMED, BATCHED>(a, bt, lists, stg);
}

static unsigned long long *g_prof = nullptr;
static int64_t g_prof_items = 0;

int launch_wave_fwd(const float4 *splat, const TileTable tt, const int32_t *flatten_ids, int width, int height,
                    const float *gt, const float *wmap, float loss_scale, const int32_t *total, int64_t max_items,
                    void *workspace, float *gtstop, int chained, unsigned tag, int max_tile_hint, hipStream_t s,
                    const Batch &bt, int C, float *alphas) {
  const int tw = cdiv(width, kTile), th = cdiv(height, kTile);
  const SliceWs ws = carve_workspace(workspace, max_items, tw * th);
  if (tag == 0 || tag >= kGranuleTagMask) {
    set_error("composite_fwd(wave): the call tag must lie in 1 .. EG_MAX_WS_TAG (got %u)", tag);
    return EG_ERR_ARG;
  }
  // EG_FWD_PROF=1 (a debugging aid, see eg_debug_fwd_profile; the only state this file keeps): the timed instantiation
  static const bool timed = getenv("EG_FWD_PROF") && atoi(getenv("EG_FWD_PROF")) != 0;
  if (timed && (!g_prof || g_prof_items < max_items)) {
    if (g_prof) (void)hipFree(g_prof);
    g_prof_items = max_items;
    (void)hipMalloc((void **)&g_prof, (size_t)max_items * 32 * sizeof(unsigned long long));
  }
  if (timed) (void)hipMemsetAsync(g_prof, 0, (size_t)max_items * 32 * sizeof(unsigned long long), s);
  int max_anchored = kMaxAnchoredSlices;
#ifdef EG_DEV_SWITCHES
  static const int anchor_env = getenv("EG_WAVE_ANCHOR") ? atoi(getenv("EG_WAVE_ANCHOR")) : 1;
  if (!anchor_env) max_anchored = 0;
#endif
  (void)max_tile_hint;
  WaveArgs a;
  a.splat = splat; a.item_rec = tt.item_rec; a.total = total; a.flat = flatten_ids; a.cursor_reset = tt.cursor_reset;
  a.gran = (unsigned long long *)ws.sliceP; a.anchor = ws.anchor; a.dead = ws.dead_hint; a.ctl = ws.ctl;
  a.loss_part = ws.loss_part;
  a.gt = gt; a.wmap = wmap; a.alphas = alphas; a.gtstop = (StopRec *)gtstop; a.prof = g_prof;
  a.width = width; a.height = height; a.tw = tw; a.n_tiles = tw * th;
  a.tag = tag; a.loss_scale = loss_scale; a.max_anchored = max_anchored; a.item_first = tt.item_first;
  a.seg_cap = tt.seg_cap;
  if (tt.seg_cap <= 0) {
    set_error("composite_fwd(wave): the item records need the segment capacity");
    return EG_ERR_ARG;
  }
  // one view: everything is resolved here and the kernel never looks at the batch descriptor
  const bool batched = C > 1;
  if (!batched && bt.gt[0]) { a.gt = bt.gt[0]; a.wmap = bt.wmap[0]; }
  const dim3 grid((unsigned)max_items, C);
#define EG_LAUNCH(CH_, TI_, BA_) composite_wave_fwd_kernel<CH_, TI_, BA_><<<grid, 256, 0, s>>>(a, bt)
  if (batched) { if (chained) EG_LAUNCH(true, false, true); else EG_LAUNCH(false, false, true); }
  else if (timed) { if (chained) EG_LAUNCH(true, true, false); else EG_LAUNCH(false, true, false); }
  else { if (chained) EG_LAUNCH(true, false, false); else EG_LAUNCH(false, false, false); }
#undef EG_LAUNCH
  timing_mark(kMarkSlice, s);
  timing_mark(kMarkRewalk, s);
  return check_launch("composite_fwd(wave)");
}

}  // namespace eg

// debugging aid (EG_FWD_PROF=1): the per-wave phase records of the last forward launch ([items][4 quadrants][8] words,
// see the kernel); returns the number of 8-word records copied (<= max_records) or a negative code.  Synchronises.
extern "C" int64_t eg_debug_fwd_profile(uint64_t *out, int64_t max_records) {
  if (!eg::g_prof || !out) return EG_ERR_ARG;
  const int64_t n = eg::g_prof_items * 4 < max_records ? eg::g_prof_items * 4 : max_records;
  if (hipMemcpy(out, eg::g_prof, (size_t)n * 8 * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) return EG_ERR_LAUNCH;
  return n;
}
