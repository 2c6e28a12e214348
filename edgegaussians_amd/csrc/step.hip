// Library housekeeping + the natively sequenced single-view training step.
//
// eg_train_step enqueues, from native code and without any host synchronisation, the whole per-view
// protocol of the reference's train_epoch body (train_gaussians.py:81-106):
//   model(idx)                 -> project(+exp/sigmoid, +tile counts) / offsets / emit / sort / composite
//   compute_projection_loss    -> fused in the compositing epilogue (weight-map form)
//   backward()                 -> footprint compositing VJP (no lists, no atomics) + projection VJP
//   update_absgrads()          -> fused in the projection VJP
//   4x Adam.step(), zero_grad  -> fused in the projection VJP (single-GPU) or left to eg_adam_multi after
//                                 the RCCL all-reduce (multi-GPU)
// 6 launches per step in the segmented layout (project+emit+scan, sort, slice+combine, re-walk [a bare launch
// when no pixel stops], footprint, project-bwd+Adam; a second sort launch only when a tile holds > 4096 keys)
// instead of ~60 (torch glue + gsplat + CUB passes + 4 unfused Adams).
#include <dlfcn.h>

#include <array>
#include <cstdarg>
#include <cstdio>
#include <thread>
#include <vector>

#include "common.h"

namespace eg {

static thread_local char g_err[512] = "no error";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char *what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return EG_ERR_LAUNCH;
  }
  return EG_OK;
}

}  // namespace eg

using namespace eg;

extern "C" const char *eg_last_error_string(void) { return g_err; }
extern "C" int eg_version(void) { return 100; }

extern "C" int eg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// ---- optional per-stage timing: HIP events recorded by this library between the stages of
// eg_train_step, on the launch stream, for a window of steps; one synchronisation at the end.
// (Driving the stages from Python to time them measures the Python call overhead instead: a
// 20-argument ctypes call costs ~100 us, the kernels 5-50 us.)
namespace eg {
constexpr int kStages = kNumMarks - 1;  // one stage ends at every mark after kMarkStart
static const char *kStageNames[kStages] = {"project_bin", "tile_emit", "tile_sort", "composite_slice_fwd",
                                           "composite_rewalk_fwd", "footprint_bwd", "project_bwd_adam"};
static hipEvent_t *g_ev = nullptr;  // [(kStages + 1) * g_ev_steps]
static int g_ev_steps = 0, g_ev_next = 0;
static hipEvent_t *g_ev_cur = nullptr;  // events of the step being enqueued (nullptr outside a window)

void timing_mark(int mark, hipStream_t stream) {
  if (g_ev_cur) (void)hipEventRecord(g_ev_cur[mark], stream);
}

// ---- optional roctx ranges around the stages of eg_train_step (tracing): off unless eg_roctx_enable(1) found
// the ROCTx library; rocprofv3 --marker-trace then shows project_bin / tile_sort / composite_fwd / footprint_bwd /
// project_bwd_adam ranges on the host timeline next to the kernel trace.  No link dependency: dlopen'ed on request.
typedef int (*RoctxPushFn)(const char *);
typedef int (*RoctxPopFn)(void);
static RoctxPushFn g_roctx_push = nullptr;
static RoctxPopFn g_roctx_pop = nullptr;
struct RoctxRange {
  bool on;
  explicit RoctxRange(const char *name) : on(g_roctx_push != nullptr) { if (on) (void)g_roctx_push(name); }
  ~RoctxRange() { if (on && g_roctx_pop) (void)g_roctx_pop(); }
};
}  // namespace eg

extern "C" int eg_roctx_enable(int32_t on) {
  if (!on) { g_roctx_push = nullptr; g_roctx_pop = nullptr; return EG_OK; }
  if (g_roctx_push) return EG_OK;
  void *h = nullptr;
  for (const char *name : {"librocprofiler-sdk-roctx.so", "libroctx64.so"})
    if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
  if (!h) { set_error("eg_roctx_enable: no ROCTx library found (%s)", dlerror()); return EG_ERR_ARG; }
  g_roctx_push = (RoctxPushFn)dlsym(h, "roctxRangePushA");
  g_roctx_pop = (RoctxPopFn)dlsym(h, "roctxRangePop");
  if (!g_roctx_push || !g_roctx_pop) {
    g_roctx_push = nullptr; g_roctx_pop = nullptr;
    set_error("eg_roctx_enable: roctxRangePushA / roctxRangePop not exported");
    return EG_ERR_ARG;
  }
  return EG_OK;
}

extern "C" int eg_timing_begin(int32_t n_steps) {
  EG_REQUIRE(n_steps > 0 && n_steps <= 4096, "n_steps out of range");
  if (g_ev) {
    for (int i = 0; i < (kStages + 1) * g_ev_steps; ++i) (void)hipEventDestroy(g_ev[i]);
    delete[] g_ev;
  }
  g_ev = new hipEvent_t[(kStages + 1) * n_steps];
  for (int i = 0; i < (kStages + 1) * n_steps; ++i)
    if (hipEventCreate(&g_ev[i]) != hipSuccess) {
      set_error("eg_timing_begin: hipEventCreate failed");
      return EG_ERR_LAUNCH;
    }
  g_ev_steps = n_steps;
  g_ev_next = 0;
  return EG_OK;
}

extern "C" const char *eg_timing_stage_name(int32_t i) { return (i >= 0 && i < kStages) ? kStageNames[i] : ""; }
extern "C" int eg_timing_stage_count(void) { return kStages; }

// average microseconds per stage over the recorded steps; stops the timing window
extern "C" int eg_timing_end(float *stage_us /*[kStages] host*/, int32_t *n_steps_out /*host|NULL*/) {
  EG_REQUIRE(g_ev && stage_us, "no timing window open");
  const int n = g_ev_next;
  for (int k = 0; k < kStages; ++k) stage_us[k] = 0.f;
  if (n > 0) {
    if (hipEventSynchronize(g_ev[(kStages + 1) * (n - 1) + kStages]) != hipSuccess) {
      set_error("eg_timing_end: hipEventSynchronize failed");
      return EG_ERR_LAUNCH;
    }
    for (int s = 0; s < n; ++s)
      for (int k = 0; k < kStages; ++k) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, g_ev[(kStages + 1) * s + k], g_ev[(kStages + 1) * s + k + 1]);
        stage_us[k] += 1e3f * ms / (float)n;
      }
  }
  if (n_steps_out) *n_steps_out = n;
  for (int i = 0; i < (kStages + 1) * g_ev_steps; ++i) (void)hipEventDestroy(g_ev[i]);
  delete[] g_ev;
  g_ev = nullptr;
  g_ev_steps = g_ev_next = 0;
  return EG_OK;
}

// where the one-kernel backward (backward_fused.hip) is the faster form: see launch_gaussian_bwd_fused
static bool fused_backward_pays(int n_gaussians, int n_tiles) {
  return n_tiles <= kPrefixHereMaxTiles && n_gaussians <= kFusedBwdMaxGaussians;
}
extern "C" int eg_backward_is_fused(int32_t n_gaussians, int32_t n_tiles) {
  return (n_gaussians > 0 && n_tiles > 0 && fused_backward_pays(n_gaussians, n_tiles)) ? 1 : 0;
}

extern "C" int eg_train_step(const eg_step_args *a, eg_stream_t stream) {
  EG_REQUIRE(a != nullptr, "null args");
  EG_REQUIRE(a->N > 0 && a->width > 0 && a->height > 0 && a->capacity > 0, "bad sizes");
  EG_REQUIRE(a->seg_cap <= 0 || (a->tile_end && a->item_end && a->item_tile), "segmented binning needs its tables");
  const int tw = cdiv(a->width, kTile), th = cdiv(a->height, kTile), T = tw * th;
  // (tile grids above 2048 tiles: the projection's scan also leaves the front-slice prefix in ticket[1..], from which the
  // sort kernel writes the forward's item records front slices first -- every projection of such a grid sets the flag)
  const uint32_t flags = EG_FLAG_LOG_SCALES | EG_FLAG_LOGIT_OPACITIES | EG_FLAG_ANTIALIASED | EG_FLAG_TIGHT_TILES |
                         ((T > kPrefixHereMaxTiles && a->seg_cap > 0 && a->ticket) ? EG_FLAG_FRONT_PREFIX : 0u);
  // (written only inside a timing window: eg_train_steps_multi's host threads run this function side by side)
  if (g_ev || g_ev_cur) g_ev_cur = (g_ev && g_ev_next < g_ev_steps) ? &g_ev[(kStages + 1) * g_ev_next] : nullptr;
  hipStream_t st = as_stream(stream);
#define EG_MARK(k) timing_mark(k, st)
  int rc;
  EG_MARK(kMarkStart);
  if (a->seg_cap > 0) {
    // segmented binning: projection + binning (+ the item scan by its last workgroup) in one pass, then the
    // per-tile sort, which also writes the tile / item tables; there is no emit kernel, its stage stays empty
    // (the previous step's last kernel may already have projected + binned this view: have_projection)
    // prefix_here (small tile grids): the projection kernels end without their serial tail (ticket, last workgroup,
    // tile scan); the sort kernel's workgroups form the item prefix of their tile from the cursors themselves and the
    // compositing kernel returns the cursors to zero (binning.hip, SegTable::total)
    const bool prefix_here = T <= kPrefixHereMaxTiles;
    EG_REQUIRE((int64_t)T * a->seg_cap < (1ll << 31), "T * seg_cap must fit 31 bits");
    if (!a->have_projection) {
      RoctxRange range("eg:project_bin");
      rc = prefix_here ? launch_project_emit(a->means, a->quats, a->log_scales, a->logit_opacities, a->viewmat, a->K, a->N,
                                             a->width, a->height, flags, a->splat, a->tile_counts, a->seg_cap, a->keys,
                                             a->item_offsets, (int32_t)a->max_items, a->total, nullptr, Batch{}, 1, st)
                       : eg_project_emit(a->means, a->quats, a->log_scales, a->logit_opacities, a->viewmat, a->K, a->N,
                                         a->width, a->height, flags, a->splat, a->tile_counts, a->seg_cap, a->keys,
                                         a->item_offsets, (int32_t)a->max_items, a->total, a->ticket, stream);
      if (rc) return rc;
    }
    EG_MARK(kMarkProjectBin);
    EG_MARK(kMarkEmit);
    {
    RoctxRange range("eg:tile_sort");
    // (the wave-autonomous forward with the fused loss follows: the sort kernel adds the empty tiles' loss terms and gives them no record)
    const bool wave_fwd = wave_forward_selected(1, a->render, a->alphas, a->last_ids, a->vpix, a->gtstop, a->wmap, a->item_rec, a->ws_tag);
    rc = launch_sort_segments(a->keys, a->tile_counts, T, a->seg_cap, a->flatten_ids, a->offsets, a->tile_end,
                              a->item_offsets, a->item_end, a->item_tile, (int32_t)a->max_items, a->max_tile_hint, Batch{}, 1,
                              st, prefix_here ? a->total : nullptr, a->item_rec,
                              (flags & EG_FLAG_FRONT_PREFIX) ? a->ticket + 1 : nullptr, (uint32_t)max(a->ws_tag, 0), tw,
                              wave_fwd ? a->gt : nullptr, wave_fwd ? a->wmap : nullptr, wave_fwd ? a->workspace : nullptr, a->width,
                              a->height, (wave_fwd && a->rewalk_hint != EG_REWALK_SPECULATE) ? kFrontChained : 0, a->total);
    }
    if (rc) return rc;
    EG_MARK(kMarkSort);
    EG_REQUIRE(a->splat && a->offsets && a->flatten_ids && a->total && a->workspace && a->max_items > 0 &&
                   (a->gtstop ? a->wmap != nullptr : (a->render && a->alphas && a->last_ids)) && (!a->wmap || a->gt),
               "bad compositing arguments");
    RoctxRange range("eg:composite_fwd");
    rc = composite_fwd_segments_hinted(a->splat, a->offsets, a->tile_end, a->item_offsets, a->item_end, a->item_tile,
                                       a->flatten_ids, a->width, a->height, a->render, a->alphas, a->last_ids, a->gt,
                                       a->wmap, a->loss_scale, a->vpix, a->loss, a->total, a->max_items, a->workspace,
                                       a->gtstop, a->rewalk_hint, a->max_tile_hint, a->ws_tag, st,
                                       prefix_here ? a->tile_counts : nullptr, a->item_rec, a->seg_cap);
    if (rc) return rc;
  } else {
  // tile_counts is zero on entry (caller zero-initialises it once): the projection counts it up and its
  // last workgroup scans it; the emit pass counts it back down to zero.
  rc = eg_project_bin(a->means, a->quats, a->log_scales, a->logit_opacities, a->viewmat, a->K, a->N, a->width,
                      a->height, flags, a->splat, a->tile_counts, a->tile_mask, a->capacity, a->offsets,
                      a->item_offsets, a->total, a->ticket, stream);
  if (rc) return rc;
  EG_MARK(kMarkProjectBin);
  rc = eg_tile_emit(nullptr, nullptr, nullptr, a->splat, flags, a->N, a->width, a->height, a->offsets,
                    a->tile_counts, a->capacity, a->keys, a->tile_mask, stream);
  if (rc) return rc;
  EG_MARK(kMarkEmit);
  rc = eg_sort_pairs(a->keys, a->offsets, T, a->capacity, a->flatten_ids, nullptr, a->max_tile_hint, stream);
  if (rc) return rc;
  EG_MARK(kMarkSort);
  rc = eg_composite_fwd(a->splat, nullptr, 1, a->offsets, a->flatten_ids, a->width, a->height, a->render,
                        a->alphas, a->last_ids, a->gt, a->wmap, a->loss_scale, a->vpix, a->loss, a->item_offsets,
                        a->total, a->max_items, a->workspace, a->gtstop, a->rewalk_hint, stream);
  if (rc) return rc;
  }
  // (slice / re-walk marks are recorded inside eg_composite_fwd)
  // backward: footprint compositing VJP, then projection VJP + absgrad (+ Adam)
  EG_REQUIRE(a->splat && a->gtstop && a->g2d, "null pointer");
  if (a->adam_host && a->next_viewmat && a->next_K && a->seg_cap > 0 && fused_backward_pays(a->N, T) && !a->two_kernel_backward) {
    // round 6: both in ONE kernel (backward_fused.hip) -- the projection chain of a workgroup's 64 Gaussians runs in its
    // first wave while the other workgroups' footprint walks fill the issue slots; the g2d record stays in LDS
    RoctxRange range("eg:gaussian_bwd_fused");
    rc = launch_gaussian_bwd_fused(a->means, a->quats, a->log_scales, a->logit_opacities, a->viewmat, a->K, a->next_viewmat,
                                   a->next_K, a->N, a->width, a->height, 0.3f, flags, a->splat, a->gtstop,
#ifdef EG_BF_PROF
                                   a->g2d,  // (development: the kernel's phase stamps land here)
#else
                                   nullptr,
#endif
                                   a->absgrads, a->adam_m, a->adam_v, *a->adam_host, a->tile_counts, a->seg_cap, a->keys,
                                   a->workspace, a->max_items, a->loss, st);
    EG_MARK(kMarkProjectBwd);
    if (g_ev_cur) { ++g_ev_next; g_ev_cur = nullptr; }
    return rc;
  }
  {
  RoctxRange range("eg:footprint_bwd");
  rc = launch_footprint_bwd(a->splat, a->N, a->width, a->height, a->gtstop, a->g2d, Batch{}, 1, st,
                            a->seg_cap > 0 ? a->workspace : nullptr, a->max_items, a->loss);
  }
  if (rc) return rc;
  // (the footprint mark is recorded inside eg_composite_bwd_footprint)
  RoctxRange range_bwd("eg:project_bwd_adam");
  if (a->adam_host && a->next_viewmat && a->next_K && a->seg_cap > 0)
    // projection backward + absgrad + Adam of this view, projection + binning + tile scan of the next one
    rc = launch_project_bwd_emit(a->means, a->quats, a->log_scales, a->logit_opacities, a->viewmat, a->K,
                                 a->next_viewmat, a->next_K, a->N, a->width, a->height, 0.3f, flags, a->splat, a->g2d,
                                 a->absgrads, a->adam_m, a->adam_v, *a->adam_host, a->tile_counts, a->seg_cap, a->keys,
                                 a->item_offsets, (int32_t)a->max_items, a->total,
                                 T <= kPrefixHereMaxTiles ? nullptr : a->ticket, st);
  else if (a->adam_host)
    rc = eg_project_bwd_adam(a->means, a->quats, a->log_scales, a->logit_opacities, a->viewmat, a->K, a->N,
                             a->width, a->height, 0.3f, flags, a->splat, a->g2d, a->adam_m, a->adam_v, a->absgrads,
                             *a->adam_host, stream);
  else
    rc = eg_project_bwd(a->means, a->quats, a->log_scales, a->logit_opacities, a->viewmat, a->K, a->N, a->width,
                        a->height, 0.3f, flags | EG_FLAG_ABSGRAD_WRITE, a->splat, a->g2d, nullptr, nullptr, a->v_means, a->v_quats,
                        a->v_scales, a->v_opacities, a->absgrads, stream);
  EG_MARK(kMarkProjectBwd);
#undef EG_MARK
  if (g_ev_cur) { ++g_ev_next; g_ev_cur = nullptr; }
  return rc;
}

// ---- C views per launch sequence.  The same six kernels as eg_train_step, each launched
// once with gridDim.y = C over [C, ...] work buffers, then ONE projection backward that sums the per-view
// gradients (what a C-rank data-parallel all-reduce forms) and ONE Adam step.  At the reference's sizes a single
// view fills a quarter of the chip (30 k Gaussians = 59 projection workgroups on 256 CUs): batching amortises
// every launch and every dependent-latency chain over C views.  A THROUGHPUT mode -- C views per optimizer
// step is a different trajectory from C sequential steps (the reference steps after every view).
extern "C" int64_t eg_batched_workspace_stride(int64_t max_items, int64_t n_tiles) {
  return (eg_composite_workspace_bytes(max_items, n_tiles) + 255) & ~(int64_t)255;
}

extern "C" int eg_train_step_batched(const eg_step_args *a, int32_t C, const float *const *viewmats,
                                     const float *const *Ks, const float *const *gts, const float *const *wmaps,
                                     eg_stream_t stream) {
  EG_REQUIRE(a != nullptr && viewmats && Ks && gts && wmaps, "null args");
  EG_REQUIRE(C >= 1 && C <= kMaxBatch, "1 <= C <= EG_MAX_BATCH");
  EG_REQUIRE(a->N > 0 && a->width > 0 && a->height > 0 && a->max_items > 0, "bad sizes");
  EG_REQUIRE(a->seg_cap > 0 && a->tile_end && a->item_end && a->item_tile, "the batched step uses the segmented layout");
  EG_REQUIRE(a->gtstop && a->g2d && a->splat && a->workspace && a->loss, "null buffer");
  const int tw = cdiv(a->width, kTile), th = cdiv(a->height, kTile), T = tw * th;
  EG_REQUIRE((int64_t)T * a->seg_cap < (1ll << 31), "T * seg_cap must fit 31 bits");
  const uint32_t flags = EG_FLAG_LOG_SCALES | EG_FLAG_LOGIT_OPACITIES | EG_FLAG_ANTIALIASED | EG_FLAG_TIGHT_TILES;
  Batch bt;
  bt.splat4 = 2ll * a->N;
  bt.tiles = T;
  bt.keys = (long long)T * a->seg_cap;
  bt.items = a->max_items;
  bt.ws_bytes = eg_batched_workspace_stride(a->max_items, T);
  bt.pixels = (long long)a->width * a->height;
  for (int v = 0; v < C; ++v) {
    EG_REQUIRE(viewmats[v] && Ks[v] && gts[v] && wmaps[v], "null view input");
    bt.viewmat[v] = viewmats[v]; bt.K[v] = Ks[v]; bt.gt[v] = gts[v]; bt.wmap[v] = wmaps[v];
  }
  hipStream_t st = as_stream(stream);
  // (see eg_train_step; with C views the prefix loads multiply: 8 views at 512 x 512 lost 7 us to them)
  const bool prefix_here = (int64_t)T * C <= kPrefixHereMaxTiles;
  int rc = launch_project_emit(a->means, a->quats, a->log_scales, a->logit_opacities, bt.viewmat[0], bt.K[0], a->N,
                               a->width, a->height, flags, a->splat, a->tile_counts, a->seg_cap, a->keys,
                               a->item_offsets, (int32_t)a->max_items, a->total, prefix_here ? nullptr : a->ticket, bt,
                               C, st);
  if (rc) return rc;
  const bool wave_fwd_b = wave_forward_selected(1, nullptr, nullptr, nullptr, nullptr, a->gtstop, bt.wmap[0], a->item_rec, a->ws_tag);
  rc = launch_sort_segments(a->keys, a->tile_counts, T, a->seg_cap, a->flatten_ids, a->offsets, a->tile_end,
                            a->item_offsets, a->item_end, a->item_tile, (int32_t)a->max_items, a->max_tile_hint, bt, C,
                            st, prefix_here ? a->total : nullptr, a->item_rec, nullptr, (uint32_t)max(a->ws_tag, 0), tw,
                            wave_fwd_b ? bt.gt[0] : nullptr, wave_fwd_b ? bt.wmap[0] : nullptr, wave_fwd_b ? a->workspace : nullptr,
                            a->width, a->height);
  if (rc) return rc;
  rc = launch_composite_fwd_segments(a->splat, a->offsets, a->tile_end, a->item_offsets, a->item_end, a->item_tile,
                                     a->flatten_ids, a->width, a->height, a->loss_scale, a->loss, a->total,
                                     a->max_items, a->workspace, a->gtstop, a->rewalk_hint, bt, C, st,
                                     a->max_tile_hint, a->ws_tag, prefix_here ? a->tile_counts : nullptr, a->item_rec,
                                     a->seg_cap);
  if (rc) return rc;
  rc = launch_footprint_bwd(a->splat, a->N, a->width, a->height, a->gtstop, a->g2d, bt, C, st, a->workspace, a->max_items,
                            a->loss);
  if (rc) return rc;
  return launch_project_bwd_batched(a->means, a->quats, a->log_scales, a->logit_opacities, a->N, a->width, a->height,
                                    0.3f, a->adam_host ? flags : (flags | EG_FLAG_ABSGRAD_WRITE), a->splat, a->g2d,
                                    a->v_means, a->v_quats, a->v_scales, a->v_opacities, a->absgrads, a->adam_m,
                                    a->adam_v, a->adam_host, bt, C, st);
}

// ---- K consecutive per-view steps enqueued by one native call (the reference's train_epoch body, K times:
// train_gaussians.py:71-106).  `a` describes step 0 exactly like eg_train_step (its viewmat / K / gt / wmap are
// ignored); step k takes view views_host[k] out of the [V, ...] arrays and weight map wmaps_host[k], and runs
// with every ACTIVE Adam step count advanced by k (a group whose count is < 0 stays skipped).  Saves the
// per-step host round trip through the binding (~40 us of Python per step: at the reference's sizes the host,
// not the GPU, bounds the real training loop).
// step k of a run of K (see eg_train_steps): the view's inputs, the call tag, the Adam counts and -- inside the run --
// the projection of view k + 1 fused into step k's last kernel
static int train_step_of_run(const eg_step_args *a, int k, int K, const int32_t *views_host, const float *const *wmaps_host,
                             const float *viewmats, const float *Ks, const float *gts, eg_stream_t stream) {
  const size_t hw = (size_t)a->width * a->height;
  EG_REQUIRE(views_host[k] >= 0 && wmaps_host[k], "bad view / null weight map");
  eg_step_args s = *a;
  eg_adam_hyper h;
  s.viewmat = viewmats + 16 * (size_t)views_host[k];
  s.K = Ks + 9 * (size_t)views_host[k];
  s.gt = gts + hw * (size_t)views_host[k];
  s.wmap = wmaps_host[k];
  if (a->ws_tag > 0) s.ws_tag = a->ws_tag + k;  // a fresh tag per step (the caller keeps ws_tag + K - 1 <= EG_MAX_WS_TAG)
  // inside the run the parameters change only through these steps: step k's last kernel projects view k + 1
  s.have_projection = (k > 0 && a->adam_host && a->seg_cap > 0) ? 1 : a->have_projection;
  s.next_viewmat = s.next_K = nullptr;
  if (k + 1 < K && a->adam_host && a->seg_cap > 0) {
    s.next_viewmat = viewmats + 16 * (size_t)views_host[k + 1];
    s.next_K = Ks + 9 * (size_t)views_host[k + 1];
  }
  if (a->adam_host) {
    h = *a->adam_host;
    h.step += k;
    for (int i = 0; i < 4; ++i)
      if (h.group_steps[i] > 0) h.group_steps[i] += k;
    s.adam_host = &h;
  }
  return eg_train_step(&s, stream);
}

extern "C" int eg_train_steps(const eg_step_args *a, int32_t K, const int32_t *views_host, const float *const *wmaps_host,
                              const float *viewmats /*[V,4,4]*/, const float *Ks /*[V,3,3]*/, const float *gts /*[V,H,W]*/,
                              eg_stream_t stream) {
  EG_REQUIRE(a != nullptr && K >= 0 && (K == 0 || (views_host && wmaps_host)) && viewmats && Ks && gts, "bad arguments");
  EG_REQUIRE(a->ws_tag <= 0 || (int64_t)a->ws_tag + K - 1 <= EG_MAX_WS_TAG, "ws_tag + K - 1 exceeds EG_MAX_WS_TAG: zero the workspace and start over at 1");
  for (int k = 0; k < K; ++k) {
    const int rc = train_step_of_run(a, k, K, views_host, wmaps_host, viewmats, Ks, gts, stream);
    if (rc) return rc;
  }
  return EG_OK;
}

// ---- S independent scenes side by side on one GPU (BASELINE config 5 on one device): K steps of each, enqueued by ONE
// native call, scene s on streams[s].  `n_threads` host threads share the scenes (thread j takes scenes j, j + n_threads,
// ...) and walk them ROUND-ROBIN, step k of all its scenes before step k + 1 of any, so that every stream always has
// work queued while the others are served; n_threads = 1 enqueues everything from the calling thread.  No Python
// between the steps and none around the threads (S interpreter threads each calling eg_train_steps pay the interpreter's
// thread switches on top of the launches: profiles/r04_scenes_per_gpu.txt).  The scenes share nothing: every trainer's
// result equals its solo run.
extern "C" int eg_train_steps_multi(int32_t S, const eg_step_args *const *args_host, int32_t K,
                                    const int32_t *const *views_host, const float *const *const *wmaps_host,
                                    const float *const *viewmats, const float *const *Ks, const float *const *gts,
                                    const eg_stream_t *streams, int32_t n_threads) {
  EG_REQUIRE(S >= 1 && S <= 64 && K >= 0 && args_host && views_host && wmaps_host && viewmats && Ks && gts && streams,
             "bad arguments");
  EG_REQUIRE(n_threads >= 1, "n_threads >= 1");
  EG_REQUIRE(g_ev == nullptr, "eg_train_steps_multi: close the eg_timing_begin window first (its events belong to one stream)");
  for (int s = 0; s < S; ++s) {
    const eg_step_args *a = args_host[s];
    EG_REQUIRE(a && (K == 0 || (views_host[s] && wmaps_host[s])) && viewmats[s] && Ks[s] && gts[s], "null scene argument");
    EG_REQUIRE(a->ws_tag <= 0 || (int64_t)a->ws_tag + K - 1 <= EG_MAX_WS_TAG, "ws_tag + K - 1 exceeds EG_MAX_WS_TAG");
    // (distinct streams AND distinct buffers, as the header promises: the same argument block -- or another block over the
    // same parameters / workspace -- on a second stream would have two host threads race on one scene)
    for (int q = 0; q < s; ++q)
      EG_REQUIRE(streams[q] != streams[s] && args_host[q] != a && args_host[q]->means != a->means &&
                     args_host[q]->workspace != a->workspace,
                 "the same scene (or the same stream) twice");
  }
  const int nt = n_threads < S ? n_threads : S;
  auto drive = [&](int j, char *msg, size_t msg_len) -> int {
    for (int k = 0; k < K; ++k)
      for (int s = j; s < S; s += nt) {
        const int rc = train_step_of_run(args_host[s], k, K, views_host[s], wmaps_host[s], viewmats[s], Ks[s], gts[s], streams[s]);
        if (rc) { snprintf(msg, msg_len, "scene %d, step %d: %s", s, k, g_err); return rc; }
      }
    return EG_OK;
  };
  char msg[512] = "";
  if (nt == 1) {
    const int rc = drive(0, msg, sizeof(msg));
    if (rc) set_error("%s", msg);
    return rc;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { set_error("eg_train_steps_multi: hipGetDevice failed"); return EG_ERR_LAUNCH; }
  std::vector<std::thread> th;
  std::vector<int> rcs(nt, EG_OK);
  std::vector<std::array<char, 512>> msgs(nt);
  for (int j = 1; j < nt; ++j)
    th.emplace_back([&, j]() {
      msgs[j][0] = 0;
      if (hipSetDevice(dev) != hipSuccess) { snprintf(msgs[j].data(), 512, "hipSetDevice(%d) failed", dev); rcs[j] = EG_ERR_LAUNCH; return; }
      rcs[j] = drive(j, msgs[j].data(), 512);
    });
  rcs[0] = drive(0, msgs[0].data(), 512);
  for (auto &t : th) t.join();
  for (int j = 0; j < nt; ++j)
    if (rcs[j]) { set_error("%s", msgs[j].data()); return rcs[j]; }
  return EG_OK;
}
