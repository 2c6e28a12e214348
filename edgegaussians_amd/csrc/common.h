// Shared device/host helpers for libedgegs (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/edgegs.h"

namespace eg {

constexpr int kTile = EG_TILE;            // 16x16 pixel tiles
constexpr int kTilePix = kTile * kTile;   // 256 pixels = 4 wavefronts of 64
constexpr int kWave = 64;
constexpr float kAlphaMax = 0.999f;
constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kTStop = 1e-4f;
constexpr float kFovClamp = 1.3f;

void set_error(const char *fmt, ...);
int check_launch(const char *what);

// kernel-granular timing marks of the training step (step.hip): no-ops unless a timing window is open
enum TimingMark {
  kMarkStart = 0, kMarkProjectBin, kMarkEmit, kMarkSort, kMarkSlice, kMarkRewalk, kMarkFootprint,
  kMarkProjectBwd, kNumMarks
};
void timing_mark(int mark, hipStream_t stream);

// View batching: the kernels of the training step take the view from blockIdx.y.  Every
// per-view work buffer of a batched step is a [C, ...] array; Batch holds the strides between the per-view
// copies and the per-view inputs.  A default-constructed Batch (all strides zero, no pointers) is the
// single-view launch: blockIdx.y is 0 and the kernels use their own view arguments.
// largest tile grid for which the sort kernel forms the tile prefix itself (every tile's workgroup reads the
// cursors of the tiles before it: O(T^2 / 2) loads in all)
// (round 6: 2048 -> 2560, so that the reference's NATIVE image size -- 800 x 800 = 2500 tiles, data/ABC-NEF_Edge/.../meta_data.json --
// takes this path with everything that hangs on it: no scan tail in the projection kernels, empty tiles without a record or a
// forward workgroup, XCD-aware records, the one-kernel backward)
constexpr int kPrefixHereMaxTiles = 2560;
constexpr int kPrefixBatchTiles = 1024;  // cursors one batch of the sort kernel's prefix loads covers (<= 3 batches)
static_assert(kPrefixHereMaxTiles <= 3 * kPrefixBatchTiles, "tile_sort_kernel forms the prefix in at most three batches");
constexpr int kMaxBatch = EG_MAX_BATCH;
struct Batch {
  long long splat4 = 0;    // splat / g2d: float4 units (2 N)
  long long tiles = 0;     // per-tile tables: T
  long long keys = 0;      // keys / sorted ids: T * seg_cap
  long long items = 0;     // item tables: max_items
  long long ws_bytes = 0;  // compositing workspace
  long long pixels = 0;    // per-pixel images / records: H * W
  const float *viewmat[kMaxBatch] = {}, *K[kMaxBatch] = {}, *gt[kMaxBatch] = {}, *wmap[kMaxBatch] = {};
};

// internal launchers of the batched step (defined next to their kernels, sequenced by step.hip)
int launch_project_emit(const float *means, const float *quats, const float *log_scales, const float *logit_opacities,
                        const float *viewmat, const float *K, int32_t N, int32_t width, int32_t height, uint32_t flags,
                        float *splat, int32_t *tile_cursor, int32_t seg_cap, uint64_t *keys, int32_t *item_first,
                        int32_t max_items, int32_t *total, int32_t *ticket, const Batch &bt, int C, hipStream_t st);
// Round 5, XCD-aware item records (binning.hip SegTable::xcd_shift, composite_wave.hip): the XCD a tile's workgroups are
// meant to run on -- square blocks of 2^shift tiles, dealt (block_x + 3 block_y) % 8.  tile < 2^20: the quotient is exact.
__device__ __forceinline__ int xcd_of_tile(int tile, int tw, float inv_tw, int shift) {
  const int ty = (int)(((float)tile + 0.5f) * inv_tw);
  const int tx = tile - __mul24(ty, tw);  // (24-bit multiplies: full rate)
  return ((tx >> shift) + __mul24(3, ty >> shift)) & 7;
}

int launch_sort_segments(uint64_t *keys, int32_t *tile_cursor, int32_t T, int32_t seg_cap, int32_t *flatten_ids,
                         int32_t *tile_start, int32_t *tile_end, int32_t *item_first, int32_t *item_end,
                         int32_t *item_tile, int32_t max_items, int32_t max_tile_hint, const Batch &bt, int C,
                         hipStream_t st, int32_t *total_prefix_here = nullptr, int32_t *item_rec = nullptr,
                         const int32_t *item_front = nullptr, uint32_t rec_tag = 0, int32_t tiles_per_row = 0,
                         const float *gt = nullptr, const float *wmap = nullptr, void *workspace = nullptr, int32_t width = 0,
                         int32_t height = 0, int32_t front_slices = 0, int32_t *total_flag = nullptr);
bool wave_forward_selected(int channels, const void *render, const void *alphas, const void *last_ids, const void *vpix,
                           const void *gtstop, const void *wmap, const void *item_rec, int chain_tag);
int launch_composite_fwd_segments(const float *splat, const int32_t *tile_start, const int32_t *tile_end,
                                  const int32_t *item_first, const int32_t *item_end, const int32_t *item_tile,
                                  const int32_t *flatten_ids, int32_t width, int32_t height, float loss_scale,
                                  float *loss_out, const int32_t *total, int64_t max_items, void *workspace,
                                  float *gtstop, int32_t rewalk_hint, const Batch &bt, int C, hipStream_t st,
                                  int32_t max_tile_hint, int32_t chain_tag, int32_t *cursor_reset = nullptr,
                                  const int32_t *item_rec = nullptr, int32_t seg_cap = 0);
int composite_fwd_segments_hinted(const float *splat, const int32_t *tile_start, const int32_t *tile_end,
                                  const int32_t *item_first, const int32_t *item_end, const int32_t *item_tile,
                                  const int32_t *flatten_ids, int32_t width, int32_t height, float *render, float *alphas,
                                  int32_t *last_ids, const float *gt, const float *wmap, float loss_scale, float *vpix,
                                  float *loss_out, const int32_t *total, int64_t max_items, void *workspace,
                                  float *gtstop, int32_t rewalk_hint, int32_t max_tile_hint, int32_t chain_tag,
                                  hipStream_t st, int32_t *cursor_reset = nullptr, const int32_t *item_rec = nullptr,
                                  int32_t seg_cap = 0);
// the XCD-aware record placement launch_sort_segments applies ("prefix here" grid with item records, one view): 0 = none
int record_xcd_shift(int T, bool prefix_here, bool has_item_rec, int C);
constexpr int kFrontChained = 9;  // class boundary of the forward's dispatch order when it runs in chained mode (binning.hip)
// workspace / max_items / loss_out (all three or none): the compositing workspace whose 64 partial loss sums (left by
// the wave-autonomous forward) block (0, view) folds into *loss_out
int launch_footprint_bwd(const float *splat, int32_t N, int32_t width, int32_t height, const float *gtstop, float *g2d,
                         const Batch &bt, int C, hipStream_t st, void *workspace = nullptr, int64_t max_items = 0,
                         float *loss_out = nullptr);
int launch_adam_regulariser(float *means, float *scales, float *quats, float *opacities, const float *g_means,
                            const float *g_scales, const float *g_quats, float *m, float *v, int32_t N,
                            const eg_adam_hyper &hyper, const float *sum, const float *loss_sum, float loss_sum_host,
                            float factor, float w, int ratio, float *loss_out, hipStream_t st);
int launch_project_bwd_emit(float *means, float *quats, float *scales, float *opacities, const float *viewmat,
                            const float *K, const float *next_viewmat, const float *next_K, int32_t N, int32_t width,
                            int32_t height, float eps2d, uint32_t flags, float *splat, const float *g2d, float *absgrads,
                            float *m, float *v, const eg_adam_hyper &hyper, int32_t *tile_cursor, int32_t seg_cap,
                            uint64_t *keys, int32_t *item_first, int32_t max_items, int32_t *total, int32_t *ticket,
                            hipStream_t st);
#ifndef EG_FUSED_BWD_MAX_GAUSSIANS
#define EG_FUSED_BWD_MAX_GAUSSIANS 32768
#endif
constexpr int kFusedBwdMaxGaussians = EG_FUSED_BWD_MAX_GAUSSIANS;  // (see step.hip fused_backward_pays)
// round 6 (backward_fused.hip): footprint backward + projection backward + Adam + the next view's projection and binning
// in ONE kernel, tile grids of <= kPrefixHereMaxTiles tiles; g2d optional (the record stays in LDS)
int launch_gaussian_bwd_fused(float *means, float *quats, float *scales, float *opacities, const float *viewmat,
                              const float *K, const float *next_viewmat, const float *next_K, int32_t N, int32_t width,
                              int32_t height, float eps2d, uint32_t flags, float *splat, const float *gtstop, float *g2d,
                              float *absgrads, float *m, float *v, const eg_adam_hyper &hyper, int32_t *tile_cursor,
                              int32_t seg_cap, uint64_t *keys, void *workspace, int64_t max_items, float *loss_out,
                              hipStream_t st);
int launch_project_bwd_batched(float *means, float *quats, float *scales, float *opacities, int32_t N, int32_t width,
                               int32_t height, float eps2d, uint32_t flags, const float *splat, const float *g2d,
                               float *v_means, float *v_quats, float *v_scales, float *v_opacities, float *absgrads,
                               float *m, float *v, const eg_adam_hyper *hyper_host, const Batch &bt, int C,
                               hipStream_t st);

inline hipStream_t as_stream(eg_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
#if defined(__HIPCC__)
// Inclusive scan over the 64 lanes of a wavefront with DPP (row shifts inside the rows of 16, then the two row
// broadcasts): six VALU instructions, no trip through the LDS crossbar -- `__shfl_xor` / `__shfl_up` compile to
// ds_bpermute_b32 with a computed lane address, ~100 cycles each and six of them in a dependent chain per reduction.
// Lane 63 holds the reduction over the wave.  `ident` is the operation's identity (lanes a shift leaves without a
// source read it).
#define EG_DPP_STEP(ctrl, rmask) v = op(v, __builtin_amdgcn_update_dpp(ident, v, ctrl, rmask, 0xf, false))
template <class Op>
__device__ __forceinline__ int wave_scan_dpp(int v, int ident, Op op) {
  EG_DPP_STEP(0x111, 0xf);  // row_shr:1
  EG_DPP_STEP(0x112, 0xf);  // row_shr:2
  EG_DPP_STEP(0x114, 0xf);  // row_shr:4
  EG_DPP_STEP(0x118, 0xf);  // row_shr:8
  EG_DPP_STEP(0x142, 0xa);  // row_bcast:15 -> rows 1 and 3
  EG_DPP_STEP(0x143, 0xc);  // row_bcast:31 -> rows 2 and 3
  return v;
}
#undef EG_DPP_STEP
// sum of a float over the wave, in lane 63 (the same six steps)
__device__ __forceinline__ float wave_sum_dpp_f(float x) {
#define EG_DPP_STEPF(ctrl, rmask) \
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rmask, 0xf, false))
  EG_DPP_STEPF(0x111, 0xf); EG_DPP_STEPF(0x112, 0xf); EG_DPP_STEPF(0x114, 0xf); EG_DPP_STEPF(0x118, 0xf);
  EG_DPP_STEPF(0x142, 0xa); EG_DPP_STEPF(0x143, 0xc);
#undef EG_DPP_STEPF
  return x;
}
// the value of the neighbouring lane (lane ^ 1): quad_perm [1, 0, 3, 2]
__device__ __forceinline__ float lane_xor1_f(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, false));
}
struct OpAdd { __device__ __forceinline__ int operator()(int a, int b) const { return a + b; } };
struct OpMaxI { __device__ __forceinline__ int operator()(int a, int b) const { return a > b ? a : b; } };
struct OpMinU { __device__ __forceinline__ int operator()(int a, int b) const { return (unsigned)a < (unsigned)b ? a : b; } };
struct OpMaxU { __device__ __forceinline__ int operator()(int a, int b) const { return (unsigned)a > (unsigned)b ? a : b; } };
#endif

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// blockIdx -> tile remap: the dispatcher places block b on XCD b % 8 (observed, speed only);
// give each XCD a contiguous run of tiles so neighbouring tiles (which share Gaussians) hit the
// same 4 MiB L2.  Bijective for any T.
__device__ __forceinline__ int xcd_tile(int b, int T) {
  const int q = T >> 3, r = T & 7;
  const int x = b & 7, k = b >> 3;
  return x * q + (x < r ? x : r) + k;
}

// tile box [x0,x1) x [y0,y1) of a Gaussian, fp32 arithmetic in the same order as the oracle:
// (c / 16) -+ (r / 16), floor / ceil, clamp to the grid.
__device__ __forceinline__ void tile_box(float x, float y, int radius, int tw, int th, int &x0, int &y0,
                                         int &x1, int &y1) {
  const float ts = (float)kTile;
  const float tr = (float)radius / ts;
  const float tx = x / ts, ty = y / ts;
  x0 = min(max((int)floorf(tx - tr), 0), tw);
  y0 = min(max((int)floorf(ty - tr), 0), th);
  x1 = min(max((int)ceilf(tx + tr), 0), tw);
  y1 = min(max((int)ceilf(ty + tr), 0), th);
}

// exclusive scan of one int per thread over a THREADS-wide workgroup (wave64 shuffles + one LDS hop)
template <int THREADS>
__device__ __forceinline__ int block_excl_scan(int c, int *wave_tmp, int &block_total) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int NW = THREADS / 64;
  int s = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(s, d, 64);
    if (lane >= d) s += o;
  }
  if (lane == 63) wave_tmp[wv] = s;
  __syncthreads();
  int pre = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const int v = wave_tmp[w];
    pre += (w < wv) ? v : 0;
    tot += v;
  }
  block_total = tot;
  __syncthreads();
  return pre + (s - c);
}

// Opacity-aware tight tile box (fused training path only; the gsplat-compatible API keeps the
// 3-sigma square box so that info["tiles_per_gauss"/"flatten_ids"] stay bit-identical).
// A pixel can only pass alpha = o * exp(-sigma) >= 1/255 inside the ellipse sigma <= ln(255 o);
// its axis-aligned extent is ex = sqrt(2 thr c / det), ey = sqrt(2 thr a / det) for conic
// [[a,b],[b,c]].  The box is the INTERSECTION of gsplat's box with the tiles that hold a pixel
// centre inside that extent, so every dropped (Gaussian, tile) pair contributes exactly nothing and
// rendered values / gradients are unchanged.  `thr` must be the compositing kernels' threshold.
constexpr float kThrMargin = 1e-3f;

__device__ __forceinline__ void tile_box_tight(float x, float y, int radius, float a, float b, float c, float o,
                                               int tw, int th, int &x0, int &y0, int &x1, int &y1) {
#pragma clang fp contract(off)  // (same decisions in every kernel this is inlined into: see forward_geom)
  tile_box(x, y, radius, tw, th, x0, y0, x1, y1);
  const float thr = __logf(255.f * o) + kThrMargin;
  const float det = a * c - b * b;
  if (!(thr > 0.f) || !(det > 0.f)) { x1 = x0; y1 = y0; return; }
  const float k2 = 2.f * thr * __builtin_amdgcn_rcpf(det);  // hardware rcp / sqrt: the inflation covers 1 ulp
  const float ex = __builtin_amdgcn_sqrtf(k2 * c) * 1.001f + 0.01f;
  const float ey = __builtin_amdgcn_sqrtf(k2 * a) * 1.001f + 0.01f;
  // tile t holds pixel centres 16 t + 0.5 .. 16 t + 15.5
  const float ts = (float)kTile;
  x0 = max(x0, (int)ceilf((x - ex - 15.5f) / ts));
  y0 = max(y0, (int)ceilf((y - ey - 15.5f) / ts));
  x1 = min(x1, (int)floorf((x + ex - 0.5f) / ts) + 1);
  y1 = min(y1, (int)floorf((y + ey - 0.5f) / ts) + 1);
  if (x1 <= x0 || y1 <= y0) { x1 = x0; y1 = y0; }
}

// Exact refinement of the tight box: does the ellipse sigma(p) <= thr reach any point of the tile's
// pixel-centre rectangle [X0+0.5, X0+15.5] x [Y0+0.5, Y0+15.5]?  sigma is a convex quadratic, so its
// minimum over the rectangle is 0 if the centre is inside, else it lies on one of the four edges, where
// it is a 1-D quadratic minimised in closed form (clamped to the edge).  Thin diagonal edge Gaussians
// miss many of the tiles their AABB touches.  A small relative slack keeps the test conservative.
__device__ __forceinline__ float sigma_at(float a, float b, float c, float dx, float dy) {
#pragma clang fp contract(off)  // (same decisions in every kernel this is inlined into: see forward_geom)
  return 0.5f * (a * dx * dx + c * dy * dy) + b * dx * dy;
}

__device__ __forceinline__ bool ellipse_hits_rect(float x, float y, float a, float b, float c, float thr,
                                                  float rx0, float ry0, float rx1, float ry1) {
#pragma clang fp contract(off)  // (same decisions in every kernel this is inlined into: see forward_geom)
  // offsets of the rectangle relative to the centre (dx = x - px as in the kernels; sign is irrelevant
  // for a centred quadratic, so use p - centre)
  const float u0 = rx0 - x, u1 = rx1 - x, v0 = ry0 - y, v1 = ry1 - y;
  if (u0 <= 0.f && u1 >= 0.f && v0 <= 0.f && v1 >= 0.f) return true;
  float best = 3.0e38f;
  // hardware reciprocals (1 ulp): the minimiser only moves by a relative 1e-7, a second-order change
  // of the minimum that the slack below swallows; four IEEE divisions per call were a third of the
  // staging cost of the slice kernel
  const float nba = -b * __builtin_amdgcn_rcpf(a), nbc = -b * __builtin_amdgcn_rcpf(c);
  // edges v = v0 and v = v1: minimise over u in [u0,u1]: u* = -b v / a
  {
    const float us0 = fminf(fmaxf(nba * v0, u0), u1), us1 = fminf(fmaxf(nba * v1, u0), u1);
    best = fminf(best, fminf(sigma_at(a, b, c, us0, v0), sigma_at(a, b, c, us1, v1)));
  }
  // edges u = u0 and u = u1: v* = -b u / c
  {
    const float vs0 = fminf(fmaxf(nbc * u0, v0), v1), vs1 = fminf(fmaxf(nbc * u1, v0), v1);
    best = fminf(best, fminf(sigma_at(a, b, c, u0, vs0), sigma_at(a, b, c, u1, vs1)));
  }
  return best <= thr * 1.001f + 1e-3f;
}

__device__ __forceinline__ bool splat_hits_tile(float x, float y, float a, float b, float c, float o, int tx, int ty) {
  const float thr = __logf(255.f * o) + kThrMargin;
  const float X0 = (float)(tx * kTile), Y0 = (float)(ty * kTile);
  return ellipse_hits_rect(x, y, a, b, c, thr, X0 + 0.5f, Y0 + 0.5f, X0 + 15.5f, Y0 + 15.5f);
}

}  // namespace eg

#define EG_REQUIRE(cond, msg)                         \
  do {                                                \
    if (!(cond)) {                                    \
      eg::set_error("%s: %s", __func__, msg);         \
      return EG_ERR_ARG;                              \
    }                                                 \
  } while (0)
