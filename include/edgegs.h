/*
 * edgegs.h -- C ABI of libedgegs.so, the MI355X (gfx950) edge-Gaussian rasterizer.
 *
 * This is the drop-in boundary for the ONE hot path of kunalchelani/EdgeGaussians: the call
 *     render, alpha, info = gsplat.rasterization(...)      reference edgegaussians/models/edge_gs.py:250-268
 * and its autograd backward (train_gaussians.py:101), plus the per-step glue around it
 * (edge_gs.py:278-324 loss, :603-613 absgrad, train_gaussians.py:104-106 Adam,
 * edge_gs.py:384-488,544-601 densify/cull).  The reference binds that path through Python
 * (`from gsplat import rasterization`, edge_gs.py:8); the binding a maintainer adds is the
 * ctypes stub in INTEGRATION.md (shipped as edgegaussians_amd/_lib.py + gsplat/__init__.py).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; plain pointers + sizes,
 *     no torch types; the library allocates nothing and keeps no state between calls -- with three stated
 *     exceptions, all measurement / communication plumbing and none on the default path: the HIP events of an open
 *     eg_timing_begin window, the per-wave profile buffer of the EG_FWD_PROF=1 debugging build path
 *     (eg_debug_fwd_profile), and the RCCL communicator handed to eg_dp_init (eg_train_steps_dp)
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t); no call synchronises
 *   - return 0 on success, a negative EG_ERR_* code otherwise (eg_last_error_string() explains)
 *   - fp32 arithmetic; indices int32; sort keys uint64 = (depth_bits << 32) | gaussian_id within
 *     a tile segment; isect ids int64 = (tile_id << 32) | depth_bits exactly as gsplat 1.0.0
 *   - N Gaussians, M tile intersections, T = tiles_x * tiles_y tiles of 16x16 pixels, one camera
 *     per call (the reference always passes C = 1, edge_gs.py:235-236; C > 1 is looped above)
 *
 * Packed per-Gaussian screen record ("splat", 8 floats, 32 B, one L2 sector pair):
 *     [0] x  [1] y  [2] conic a  [3] conic b  [4] conic c  [5] opacity*compensation
 *     [6] depth (fp32)  [7] radius (int32 bits; 0 = culled)
 * Packed per-Gaussian 2D gradient accumulator ("g2d", 8 floats, 32 B):
 *     [0] v_x [1] v_y [2] |v_x| [3] |v_y| [4] v_a [5] v_b [6] v_c [7] v_opacity_eff
 */
#ifndef EDGEGS_H
#define EDGEGS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *eg_stream_t; /* hipStream_t */

#define EG_OK 0
#define EG_ERR_ARG (-1)     /* bad argument (null pointer, negative size, unsupported channel count) */
#define EG_ERR_LAUNCH (-2)  /* hipGetLastError() after a launch */
#define EG_ERR_NODEVICE (-3)

#define EG_TILE 16
#define EG_MAX_BATCH 8 /* views per eg_train_step_batched call */
#define EG_REWALK_SPECULATE (-2) /* rewalk_hint: do not launch the exact-stop re-walk; control word 3 reports a miss */
#define EG_FLAG_LOG_SCALES 1u      /* `scales` holds log-scales: exp() fused (edge_gs.py:253) */
#define EG_FLAG_LOGIT_OPACITIES 2u /* `opacities` holds logits: sigmoid() fused (edge_gs.py:254) */
#define EG_FLAG_ANTIALIASED 4u     /* rasterize_mode="antialiased" (edge_gs.py:50,266) */
#define EG_FLAG_ABSGRAD_WRITE 16u   /* eg_project_bwd: absgrads[g] = increment instead of += (data-parallel leg) */
#define EG_FLAG_FRONT_PREFIX 32u     /* eg_project_emit & co.: `ticket` is [T + 2] int32 and the scan of the last workgroup    \
                                      also leaves ticket[1 + t] = sum over the tiles before t of min(items, EG_FRONT_LARGE)  \
                                      and ticket[1 + T] = that sum over all tiles (dispatch classes of the forward on tile   \
                                      grids above 2560 tiles: the sort kernel then writes the item records front slices first) */
#define EG_FLAG_GRAD_ACCUM 64u       /* eg_project_bwd: v_means / v_quats / v_scales / v_opacities += instead of = (the sum over the \
                                      cameras of eg_project_bwd_cams) */
#define EG_FRONT_LARGE 9
#define EG_FLAG_TIGHT_TILES 8u     /* bin with the opacity-aware tile box (subset of gsplat's box that \
                                      drops only (Gaussian, tile) pairs contributing exactly nothing) */

const char *eg_last_error_string(void);
int eg_version(void);
/* number of gfx950 devices visible; <= 0 means the library cannot run (callers must fail loudly) */
int eg_device_count(void);

/* ---- G1: fully fused projection forward (replaces gsplat fully_fused_projection fwd)
 * Optional outputs may be NULL.  When tile_counts != NULL the per-tile intersection counts are
 * accumulated too (gsplat isect_tiles pass 1) -- tile_counts must be zeroed by the
 * caller once (eg_tile_emit returns it to zero every step).  g2d, when non-NULL, is zeroed for the
 * backward pass of this step. */
int eg_project_fwd(const float *means, const float *quats, const float *scales, const float *opacities,
                   const float *viewmat /*[16] row-major world->cam*/, const float *K /*[9]*/,
                   int32_t N, int32_t width, int32_t height, float near_plane, float far_plane,
                   float eps2d, float radius_clip, uint32_t flags,
                   float *splat /*[N,8]*/, int32_t *radii /*[N]|NULL*/, float *means2d /*[N,2]|NULL*/,
                   float *depths /*[N]|NULL*/, float *conics /*[N,3]|NULL*/, float *compensations /*[N]|NULL*/,
                   int32_t *tiles_per_gauss /*[N]|NULL*/, int32_t *tile_counts /*[T]|NULL*/,
                   float *g2d /*[N,8]|NULL*/, eg_stream_t stream);

/* ---- G1 + G2 + G3 for the fused training step: projection of raw parameters (log-scales, logits,
 * antialiased), per-tile counting with the opacity-aware exact tile test, AND the scan -- the last
 * workgroup to finish (device-scope ticket) produces offsets / item_offsets / total, which saves the
 * eg_tile_offsets launch.  tile_mask[N] receives each Gaussian's exact tile hits (bit mask over its
 * box, 0xffffffff = box larger than 32 tiles) for eg_tile_emit.  ticket: int32[1], zero-initialised
 * once by the caller (the kernel returns it to zero). */
int eg_project_bin(const float *means, const float *quats, const float *log_scales, const float *logit_opacities,
                   const float *viewmat, const float *K, int32_t N, int32_t width, int32_t height, uint32_t flags,
                   float *splat /*[N,8]*/, int32_t *tile_counts /*[T], zero on entry*/, uint32_t *tile_mask /*[N]*/,
                   int64_t capacity, int32_t *offsets /*[T+1]*/, int32_t *item_offsets /*[T+1]*/,
                   int32_t *total /*[4]*/, int32_t *ticket /*[1]*/, eg_stream_t stream);

/* ---- G2 (per-Gaussian part) on caller-supplied screen data: tiles_per_gauss + tile_counts from
 * (means2d, radii).  Used when projection ran elsewhere (parity tests feed oracle floats). */
int eg_tile_count(const float *means2d, const int32_t *radii, int32_t N, int32_t width, int32_t height,
                  int32_t *tiles_per_gauss /*[N]|NULL*/, int32_t *tile_counts /*[T]*/, eg_stream_t stream);

/* ---- G3+G6: exclusive scan of the per-tile counts -> isect_offsets[T+1] (offsets[T] = M), which is
 * gsplat's isect_offset_encode result.  tile_counts is left intact: eg_tile_emit
 * counts it back down to zero.  item_offsets[T+1] (may be NULL) is the exclusive scan of
 * ceil(count/128): the (tile, 128-Gaussian slice) work items of the compositing kernels.
 * total[4] = { M, overflow flag (M > capacity), number of items, largest tile population }.
 * The overflow flag is STICKY: every producer (this scan, eg_project_bin, eg_project_emit) only ever
 * raises total[1]; the caller zero-initialises it and clears it after reading, so one read after K
 * steps tells whether ANY of them dropped intersections. */
int eg_tile_offsets(const int32_t *tile_counts /*[T]*/, int32_t T, int64_t capacity,
                    int32_t *offsets /*[T+1]*/, int32_t *item_offsets /*[T+1]|NULL*/, int32_t *total /*[4]*/,
                    eg_stream_t stream);

/* ---- G4: emit one (depth_bits<<32 | gaussian_id) key per (Gaussian, tile) into that tile's
 * segment [offsets[t], offsets[t+1]) (order inside a segment arbitrary; eg_sort_pairs fixes it). */
int eg_tile_emit(const float *means2d_or_null, const int32_t *radii_or_null, const float *depths_or_null,
                 const float *splat_or_null, uint32_t flags /*EG_FLAG_TIGHT_TILES needs splat*/,
                 int32_t N, int32_t width, int32_t height,
                 const int32_t *offsets /*[T+1]*/, int32_t *tile_counts /*[T]: counts on entry, zero on exit*/,
                 int64_t capacity, uint64_t *keys /*[capacity]*/,
                 const uint32_t *tile_mask /*[N]|NULL: exact tile hits left by eg_project_bin*/, eg_stream_t stream);

/* ---- G5: segmented sort -- every tile segment ascending by (depth bits, gaussian id), which equals
 * gsplat's stable radix sort on (tile, depth) of index-ordered emissions.  Writes the sorted
 * Gaussian ids (gsplat flatten_ids) and optionally the int64 isect ids (tile<<32 | depth_bits). */
int eg_sort_pairs(uint64_t *keys /*[capacity] in/out*/, const int32_t *offsets /*[T+1]*/, int32_t T,
                  int64_t capacity, int32_t *flatten_ids /*[capacity]*/, int64_t *isect_ids /*[capacity]|NULL*/,
                  int32_t max_tile_hint /*largest tile population seen so far, 0 = unknown: launch-shape hint
                  only (skips the idle launch of the large-tile variant), never affects the result*/,
                  eg_stream_t stream);

/* ---- segmented binning (training step): projection + binning in ONE pass.  Tile t owns the fixed
 * segment keys[t * seg_cap ... (t + 1) * seg_cap), so a Gaussian's keys are placed (one returning atomic
 * per workgroup and touched tile on tile_cursor[t]) without a count pass, a global scan of offsets or a
 * second (emit) kernel.  The last workgroup to finish (device-scope ticket) scans the populations:
 * item_first [T] (first 128-Gaussian slice of every tile) and total[4] = {M, overflow flag, items, largest
 * tile population}.  eg_sort_segments sorts every segment in place (same order as eg_sort_pairs) and
 * its one-workgroup-per-tile kernel writes the per-tile tables: tile_start/tile_end [T] (the keys),
 * item_end [T], item_tile [max_items] (owner of every item), and returns the cursors to zero.  A tile
 * above seg_cap (or more items than max_items) drops the excess and raises total[1].
 * eg_composite_fwd_segments = the slice-parallel eg_composite_fwd on those tables. */
int eg_project_emit(const float *means, const float *quats, const float *log_scales, const float *logit_opacities,
                    const float *viewmat, const float *K, int32_t N, int32_t width, int32_t height,
                    uint32_t flags /*must include EG_FLAG_TIGHT_TILES*/, float *splat /*[N,8]*/,
                    int32_t *tile_cursor /*[T] zero on entry*/, int32_t seg_cap, uint64_t *keys /*[T*seg_cap]*/,
                    int32_t *item_first /*[T]*/, int32_t max_items, int32_t *total /*[4]*/,
                    int32_t *ticket /*[1] zero-initialised once*/, eg_stream_t stream);
int eg_sort_segments(uint64_t *keys, int32_t *tile_cursor /*[T] in: populations, out: zero*/, int32_t T,
                     int32_t seg_cap, int32_t *flatten_ids /*[T*seg_cap]*/, int32_t *tile_start /*[T]*/,
                     int32_t *tile_end /*[T]*/, const int32_t *item_first /*[T]*/, int32_t *item_end /*[T]*/,
                     int32_t *item_tile /*[max_items]*/, int32_t max_items, int32_t max_tile_hint,
                     eg_stream_t stream);
int eg_composite_fwd_segments(const float *splat, const int32_t *tile_start, const int32_t *tile_end,
                              const int32_t *item_first, const int32_t *item_end, const int32_t *item_tile,
                              const int32_t *flatten_ids, int32_t width, int32_t height,
                              float *render /*[H,W]|NULL*/, float *alphas /*[H,W]|NULL*/,
                              int32_t *last_ids /*[H,W]|NULL*/, const float *gt, const float *wmap, float loss_scale,
                              float *vpix /*[H,W]|NULL*/, float *loss_out /*[1]*/, const int32_t *total,
                              int64_t max_items, void *workspace, float *gtstop /*[H,W,3]*/,
                              int32_t rewalk_hint, eg_stream_t stream);

/* ---- G7: alpha compositing forward (replaces gsplat rasterize_to_pixels fwd).
 * channels = 1 or 3.  colors == NULL means "all ones" (the reference's colours, edge_gs.py:247).
 * Fused weighted-L1 (edge_gs.py:279,288-324 in weight-map form): when wmap != NULL,
 * channel 0 is clamped to [0,1], loss_out[0] += sum_p wmap_p*|c0_p - gt_p| and
 * vpix[p] = loss_scale * wmap_p * sign(c0_p - gt_p) (the upstream gradient of eg_composite_bwd).
 * Slice-parallel mode (unit colours only): pass item_offsets + total from eg_tile_offsets, an upper
 * bound max_items >= total[2] (e.g. ceil(capacity/128) + T) and a workspace of
 * eg_composite_workspace_bytes(max_items, T) bytes; one workgroup runs per (tile, 128-Gaussian slice; an
 * empty tile owns one empty item).  The first eg_composite_workspace_ctl_bytes(max_items, T) bytes of the
 * workspace are control words (per-tile tickets, item flags and tags, the re-walk list counter): the caller ZEROES
 * them once when the workspace is allocated -- with these (max_items, T) -- and every call hands them back
 * zeroed.  With item_offsets == NULL (or per-Gaussian colours) one workgroup walks each tile.
 * When gtstop != NULL (the fused training step, whose backward reads nothing else) render, alphas,
 * last_ids and vpix may each be NULL and are then not materialised.  Without wmap the gtstop record carries
 * T_final itself (upstream gradient 1): the caller scales word 0 by its per-pixel upstream gradient before
 * eg_composite_bwd_footprint (what the gsplat-compatible operator's backward does). */
int64_t eg_composite_workspace_bytes(int64_t max_items, int64_t n_tiles);
int64_t eg_composite_workspace_ctl_bytes(int64_t max_items, int64_t n_tiles);
int eg_composite_fwd(const float *splat, const float *colors /*[N,channels]|NULL*/, int32_t channels,
                     const int32_t *offsets, const int32_t *flatten_ids, int32_t width, int32_t height,
                     float *render /*[H,W,channels]|NULL*/, float *alphas /*[H,W]|NULL*/,
                     int32_t *last_ids /*[H,W]|NULL*/, const float *gt /*[H,W]|NULL*/, const float *wmap /*[H,W]|NULL*/, float loss_scale,
                     float *vpix /*[H,W]|NULL*/, float *loss_out /*[1]|NULL*/,
                     const int32_t *item_offsets /*[T+1]|NULL*/, const int32_t *total /*[4]|NULL*/,
                     int64_t max_items, void *workspace,
                     float *gtstop /*[H,W,3] 32-bit words|NULL: {vpix * T_final (f32), id of the last contributor
                                     if the pixel's walk stopped on T <= 1e-4 else -1 (i32), that Gaussian's
                                     depth bits (u32)} for eg_backward_fused*/,
                     int32_t rewalk_hint /*how many (tile, slice) items needed the exact-stop re-walk lately: sizes that
                     kernel's grid, never affects the result; 0 = none seen, -1 = unknown.  The workspace's control
                     word [n_tiles + max_items + 2] holds the largest list length since the caller last zeroed it.
                     EG_REWALK_SPECULATE (-2): the caller bets that no pixel reaches the transmittance stop (true
                     until opacities have trained up) and the re-walk kernel is NOT launched; if a pixel does stop,
                     the sticky control word [n_tiles + max_items + 3] is raised and the results of that call are
                     INVALID -- the caller must check the word, restore its state and repeat the call with a
                     hint >= -1 (what EdgeTrainer's journal does), then zero the word*/,
                     eg_stream_t stream);

/* ---- G8: compositing backward for unit colours (replaces gsplat rasterize_to_pixels bwd for the
 * reference's call).  vpix[p] = sum_k dL/drender[p,k] + dL/dalpha[p].  Accumulates
 * into g2d with float atomics. */
int eg_composite_bwd(const float *splat, const int32_t *offsets, const int32_t *flatten_ids,
                     int32_t width, int32_t height, const float *alphas, const int32_t *last_ids,
                     const float *vpix, float *g2d /*[N,8] accumulated*/,
                     const int32_t *item_offsets /*[T+1]|NULL*/, const int32_t *total /*[4]|NULL*/,
                     int64_t max_items, eg_stream_t stream);

/* ---- G8 (general colours): order-dependent backward with per-Gaussian colours, channels = 3.
 * v_colors may be NULL. */
int eg_composite_bwd_colors(const float *splat, const float *colors, int32_t channels,
                            const int32_t *offsets, const int32_t *flatten_ids, int32_t width, int32_t height,
                            const float *alphas, const int32_t *last_ids, const float *v_render,
                            const float *v_alphas, float *g2d, float *v_colors /*[N,channels]|NULL*/,
                            eg_stream_t stream);

/* ---- G9: fully fused projection backward (replaces gsplat fully_fused_projection bwd).
 * Consumes g2d; writes (not accumulates) gradients w.r.t. the SAME representation the forward
 * took (flags): v_means[N,3], v_quats[N,4], v_scales[N,3], v_opacities[N].
 * absgrads (edge_gs.py:603-613), when non-NULL: absgrads[g] += hypot(g2d[g][2], g2d[g][3]).
 * External mode (v_comps_ext != NULL; gsplat's autograd layout, where `opacities * compensations`
 * is a torch op between projection and compositing): dL/dcompensation is read from v_comps_ext[N],
 * g2d[.][7] is ignored and v_opacities may be NULL.  v_depths_ext[N] (may be NULL) adds dL/ddepth. */
int eg_project_bwd(const float *means, const float *quats, const float *scales, const float *opacities,
                   const float *viewmat, const float *K, int32_t N, int32_t width, int32_t height,
                   float eps2d, uint32_t flags, const float *splat, const float *g2d,
                   const float *v_comps_ext, const float *v_depths_ext,
                   float *v_means, float *v_quats, float *v_scales, float *v_opacities,
                   float *absgrads /*[N]|NULL*/, eg_stream_t stream);

/* hyper-parameters of the four torch.optim.Adam instances (train_utils.py:50-60) */
typedef struct {
  double lr_means, lr_scales, lr_quats, lr_opacities; /* doubles: the bias-corrected scalars are */
  double beta1, beta2, eps;                            /* formed in double like torch does, then cast */
  int32_t step;           /* 1-based step count shared by the four optimizers ... */
  int32_t group_steps[4]; /* ... unless overridden per optimizer (means, scales, quats, opacities):
                             0 = use `step`, > 0 = that optimizer's own count, < 0 = it does not step in
                             this call (the regulariser steps of train_gaussians.py:108-131 advance only
                             means / scales / quats, so the counts drift apart) */
} eg_adam_hyper;

/* ---- G8, footprint form (unit colours, fused path): every Gaussian's gradient is summed over its
 * own footprint in the gtstop image written by eg_composite_fwd (the unit-colour backward is
 * order-independent); a wavefront owns 8 Gaussians and deals its 64 lanes out in proportion to
 * their footprint sizes; the footprint is walked as a sheared box that follows the ellipse.  The g2d
 * record is WRITTEN -- no tile lists, no atomics, no zeroing of g2d, deterministic.
 * Footprints of any size are handled in the one launch (a screen-filling Gaussian takes the lanes of
 * its wavefront). */
int eg_composite_bwd_footprint(const float *splat, int32_t N, int32_t width, int32_t height,
                               const float *gtstop /*[H,W,3]*/, float *g2d /*[N,8] written*/,
                               eg_stream_t stream);

/* ---- whole backward of the fused path: eg_composite_bwd_footprint, then eg_project_bwd_adam
 * (hyper_host != NULL: absgrads accumulated, Adam applied) or eg_project_bwd (hyper_host == NULL:
 * gradients written to v_*, absgrads[g] += increment). */
int eg_backward_fused(float *means, float *quats, float *scales, float *opacities,
                      const float *viewmat, const float *K, int32_t N, int32_t width, int32_t height,
                      float eps2d, uint32_t flags, const float *splat, const float *gtstop, float *g2d,
                      float *v_means, float *v_quats, float *v_scales, float *v_opacities,
                      float *absgrads /*[N]|NULL*/, float *m, float *v,
                      const eg_adam_hyper *hyper_host /*NULL = write gradients*/, eg_stream_t stream);

/* ---- a6: absgrad accumulate on its own (edge_gs.py:607-613): absgrads += ||means2d.absgrad||_2 */
int eg_absgrad_accum(const float *means2d_absgrad /*[N,2]*/, int32_t N, float *absgrads, eg_stream_t stream);

/* ---- a7: the four torch.optim.Adam steps of train_gaussians.py:104-106 in one launch
 * (train_utils.py:50-60: betas 0.9/0.999, eps 1e-8, no weight decay, no amsgrad; torch 1.13
 * update order).  `step` is the 1-based step count shared by the four optimizers. */

int eg_adam_multi(float *means, float *scales, float *quats, float *opacities,
                  const float *g_means, const float *g_scales, const float *g_quats, const float *g_opacities,
                  float *m /*[N,11]: means3|scales3|quats4|opac1 blocks of N*dim*/, float *v /*same*/,
                  int32_t N, eg_adam_hyper hyper,
                  const float *absgrad_inc /*[N]|NULL*/, float *absgrads /*[N]|NULL: += absgrad_inc (the
                                             all-reduced increment of the data-parallel leg)*/,
                  eg_stream_t stream);

/* One torch.optim.Adam step on ONE flat fp32 tensor: what each of the reference's four optimizers does
 * (train_utils.py:50-60 builds them, train_gaussians.py:104-106 / 116-118 / 128-130 step them one by one) -- the
 * native leg of the drop-in optimizer class edgegaussians_amd/optim.py:Adam.  Same arithmetic as eg_adam_multi
 * (torch's update order, bias corrections formed in double on the host); `step` is the optimizer's 1-based count
 * AFTER this step.  zero_grad != 0 also clears `grad` (opt.zero_grad(set_to_none=False) in the same pass). */
int eg_adam_tensor(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, double lr, double beta1,
                   double beta2, double eps, int32_t step, int32_t zero_grad, eg_stream_t stream);

/* The same four Adam steps (identical arithmetic) fused with eg_project_emit of the view this rank rasterises NEXT:
 * the tail of a data-parallel step (after the all-reduce of the gradients), one launch and one read of the
 * parameters instead of two.  The step that follows passes eg_step_args.have_projection = 1.  flags as
 * eg_project_emit; buffers as eg_project_emit (segmented layout). */
int eg_adam_emit(float *means, float *scales, float *quats, float *opacities,
                 const float *g_means, const float *g_scales, const float *g_quats, const float *g_opacities,
                 float *m, float *v, int32_t N, eg_adam_hyper hyper, const float *absgrad_inc, float *absgrads,
                 const float *next_viewmat, const float *next_K, int32_t width, int32_t height, uint32_t flags,
                 float *splat, int32_t *tile_cursor, int32_t seg_cap, uint64_t *keys, int32_t *item_first,
                 int32_t max_items, int32_t *total, int32_t *ticket, eg_stream_t stream);

/* ---- fused G9 + absgrad + Adam: single-GPU training step tail (no gradient exchange needed). */
int eg_project_bwd_adam(float *means, float *quats, float *scales, float *opacities,
                        const float *viewmat, const float *K, int32_t N, int32_t width, int32_t height,
                        float eps2d, uint32_t flags, const float *splat, const float *g2d,
                        float *m, float *v, float *absgrads, eg_adam_hyper hyper, eg_stream_t stream);

/* ---- a8/a9: densify / cull row movement (edge_gs.py:384-488,544-576).
 * eg_compact: out[j] = in[i] for the j-th row i with keep[i] != 0 (stable); rows of `dim` floats.
 * `positions` is an exclusive scan of keep (int32[N]) computed by eg_mask_scan; returns nothing on host.
 * eg_append: out[n_old + k*n_sel + j] = in[sel_j] (+ noise) for copies k = 0..copies-1. */
int eg_mask_scan(const uint8_t *keep /*[N]*/, int32_t N, int32_t *positions /*[N]*/, int32_t *count /*[1]*/,
                 eg_stream_t stream);
int eg_compact_rows(const float *in, const uint8_t *keep, const int32_t *positions, int32_t N, int32_t dim,
                    float *out, eg_stream_t stream);
int eg_append_rows(const float *in, const uint8_t *sel, const int32_t *positions, int32_t N, int32_t n_sel,
                   int32_t dim, int32_t copies, const float *noise /*[copies*n_sel,dim]|NULL*/, float fill_zero,
                   float *out_tail /*[copies*n_sel, dim]*/, eg_stream_t stream);

/* ---- a10: cull_gaussians_not_projecting (edge_gs.py:578-601): hits[g] += 1 for every view whose
 * rounded projection of means[g] lands inside the image on an edge pixel. */
int eg_project_hits(const float *means, int32_t N, const float *P /*[V,3,4] = K @ viewmat[:3]*/, int32_t V,
                    const uint8_t *edge_masks /*[V,H,W]*/, int32_t width, int32_t height,
                    int32_t *hits /*[N] zeroed by caller*/, eg_stream_t stream);

/* ---- 8(f) rank 3 twin: filter_by_projection (edge_extraction/filtering.py:80-123): the post-hoc filter
 * of the edge-extraction stage.  visib[g] += edge_maps[v][round(v_g), round(u_g)] (the float edge
 * strength, not a mask) for every view v in which x = K (R X + t) lands inside the image; the caller
 * divides by V and thresholds (in float64, like numpy does there).  cams: [V, 21] floats = K (9, row-major) | R (9) | t (3). */
int eg_project_visibility(const float *means, int32_t N, const float *cams /*[V,21]*/, int32_t V,
                          const float *edge_maps /*[V,H,W]*/, int32_t width, int32_t height,
                          double *visib /*[N] f64 (the reference's accumulator type), zeroed by caller*/,
                          eg_stream_t stream);

/* ---- nearest neighbours + orientation regularisers (train_gaussians.py:108-131).
 * eg_knn: exact K <= 32 nearest neighbours (self excluded, ascending distance, ties by index) of N 3D
 * points on a uniform grid: origin/cell/dims chosen by the caller (bounding box of the points, a few
 * points per cell).  A wavefront answers a few cell-ordered queries one after the other: the rows of the query's
 * block of cells are dealt to groups of lanes, a lane evaluates one candidate per round, the K-best list is spread
 * over the lanes (insertion = one DPP shift, the same cost for every K).  Scratch: cell_of[N], cell_counts[C] (zero on entry, returned to zero),
 * cell_start[C+1], sorted[N,4] (the points in cell order: x y z index) with C = dims[0]*dims[1]*dims[2].  Replaces k_nearest_sklearn
 * (edge_gs.py:135-151). */
int eg_knn(const float *points /*[N,3]*/, int32_t N, int32_t K, const float *origin_host /*[3]*/, float cell,
           const int32_t *dims_host /*[3]*/, int32_t *cell_of, int32_t *cell_counts, int32_t *cell_start,
           float *sorted /*[N,4]*/, int32_t *out_idx /*[N,K]*/, float *out_d2 /*[N,K]|NULL squared distances*/,
           eg_stream_t stream);

/* eg_knn_auto: the grid search with the grid chosen on the DEVICE: bounding box by a reduction kernel, D x D x D
 * cubic cells with D = eg_knn_auto_dims(N, K) (a function of the sizes only, so the caller can size the scratch without
 * looking at the points) -- no host sync at all.  Scratch: cell_of[N], cell_counts[D^3] (zero on entry, returned to zero),
 * cell_start[D^3 + 1], sorted[N,4], grid_scratch (64 bytes, zero on entry, returned to zero).  Same result as eg_knn / eg_knn_small. */
int32_t eg_knn_auto_dims(int32_t N, int32_t K);
int eg_knn_auto(const float *points /*[N,3]*/, int32_t N, int32_t K, int32_t *cell_of, int32_t *cell_counts,
                int32_t *cell_start, float *sorted, void *grid_scratch, int32_t *out_idx /*[N,K]*/,
                float *out_d2 /*[N,K] or NULL*/,
                float *kth /*[N] or NULL: temporal coherence for callers that search the SAME (slowly moving) points
                             again and again -- on entry each point's K-th squared distance of the previous call
                             (0 = unknown), used times kth_slack as the entry bound of its list; on exit this call's.
                             Exactness does not depend on it: a bound that turns out too small costs a re-scan */,
                float kth_slack /* e.g. 1.2 */, eg_stream_t stream);

/* eg_knn_small: the same result (exact K <= 32 neighbours, self excluded, ascending (distance, index)) by
 * exhaustive search in ONE launch, for N <= 131072: no grid, no scratch.  A lane holds a candidate, a wavefront owns a
 * few queries whose K-best lists are spread over its lanes (one DPP shift per insertion).  The fastest entry up to
 * ~5 * 10^3 points; fastest when the rows are in spatial order (the scan starts at the queries' own rows). */
int eg_knn_small(const float *points /*[N,3]*/, int32_t N, int32_t K, int32_t *out_idx /*[N,K]*/,
                 float *out_d2 /*[N,K] or NULL*/, eg_stream_t stream);
/* compute_direction_loss (edge_gs.py:346-373): sum_out[0] += sum over the counted (i,k) of
 * |m_i . unit(mu_i - mu_nn(i,k))|; g_means += and g_quats = the gradient of that SUM.  top_k <= 0 or >= K:
 * every listed neighbour counts (loss = 1 - sum/(N K), the caller scales by -lambda/(N K)); 0 < top_k < K
 * ('enforce_half', :366-369, K = 2 k listed, top_k = k): only the top_k best-aligned neighbours of each
 * Gaussian count (loss = 1 - sum/(N top_k)).  K <= 32. */
int eg_direction_loss(const float *means, const float *quats, const float *log_scales,
                      const int32_t *nn_idx /*[N,K]*/, int32_t N, int32_t K, int32_t top_k,
                      float *g_means /*[N,3] accumulated*/, float *g_quats /*[N,4] written*/, float *sum_out,
                      eg_stream_t stream);
/* compute_ratio_loss (edge_gs.py:375-380): sum_out[0] += sum_i second/largest scale (loss = sum/N);
 * g_scales = gradient of the sum w.r.t. the log-scales. */
int eg_ratio_loss(const float *log_scales, int32_t N, float *g_scales /*[N,3] written*/, float *sum_out,
                  eg_stream_t stream);

/* One regulariser iteration of train_gaussians.py:108-131 as one native enqueue: loss ('direction' kind 0 /
 * 'ratio' kind 1), lambda = loss_sum * scale_factor / loss formed on the device, backward, and the Adam step of the
 * means / scales / quats optimizers (set hyper.group_steps[3] < 0: the opacity optimizer does not step there).
 * grads: [11 N] scratch in the block layout of eg_train_step's gradient buffer.  nn: neighbour table [N, nn_stride];
 * columns nn_offset .. nn_offset + K - 1 are used (the reference drops the nearest one, edge_gs.py:342).
 * loss_sum: device scalar or NULL (then loss_sum_host).  work: [2] floats; work[1] = the loss value afterwards. */
int eg_regulariser_step(int32_t kind, float *means, float *quats, float *log_scales, float *logit_opacities,
                        float *adam_m, float *adam_v, float *grads, int32_t N, const int32_t *nn, int32_t nn_stride,
                        int32_t nn_offset, int32_t K, int32_t top_k, const float *loss_sum, float loss_sum_host,
                        float scale_factor, float *work /*[2]*/, eg_adam_hyper hyper, eg_stream_t stream);
/* The same with ORDER-INDEPENDENT sums (data-parallel runs, SURVEY 8e: every rank must take bit-identical regulariser
 * steps): the neighbour gradients of the direction loss and the loss sums are accumulated with 64-bit fixed-point integer
 * atomics (2^-32 resolution) instead of float atomics, in `fixed` -- int64 [3 N + 1], zero on entry, handed back zeroed.
 * Agrees with eg_regulariser_step to ~1e-7 relative; equal bit for bit from run to run and from rank to rank. */
int eg_regulariser_step_fixed(int32_t kind, float *means, float *quats, float *log_scales, float *logit_opacities,
                              float *adam_m, float *adam_v, float *grads, int32_t N, const int32_t *nn, int32_t nn_stride,
                              int32_t nn_offset, int32_t K, int32_t top_k, const float *loss_sum, float loss_sum_host,
                              float scale_factor, float *work /*[2]*/, eg_adam_hyper hyper, int64_t *fixed /*[3 N + 1]*/,
                              eg_stream_t stream);

/* ---- the per-pixel loss weights of the 'bg_edge_ratio' strategy (edge_gs.py:298-314 in weight-map form) built on
 * the device: out[p] = [gt_p >= thr] / n_edge + [p among perm[0 .. n_sel)] / n_sel; perm = a random permutation
 * drawn by the caller (int64, distinct, taken modulo HW like the reference's unravel, :303-310). */
int eg_ratio_wmap(const float *gt /*[H*W]*/, float thr, int32_t n_edge, const int64_t *perm, int32_t n_sel, int32_t HW,
                  float *out /*[H*W]*/, eg_stream_t stream);

/* The same map with the sample drawn inside the kernel (one launch, no permutation buffer): the n_sel sampled
 * pixels are { p < n_bg : pi(p) < n_sel } for a pseudo-random permutation pi of [0, n_bg) keyed by `seed` (Feistel
 * network + cycle walking): exactly n_sel distinct pixels, uniformly distributed; a new seed gives a new sample. */
int eg_ratio_wmap_seeded(const float *gt /*[H*W]*/, float thr, int32_t n_edge, int32_t n_bg, int32_t n_sel,
                         uint64_t seed, int32_t HW, float *out /*[H*W]*/, eg_stream_t stream);
/* C of them by one native call: map c from gts[c] with (n_edge[c], n_bg[c], n_sel[c], seeds[c]) into out + c * HW (host
 * arrays; the same kernel, the same maps). */
int eg_ratio_wmaps_seeded(int32_t C, const float *const *gts, float thr, const int32_t *n_edge, const int32_t *n_bg,
                          const int32_t *n_sel, const uint64_t *seeds, int32_t HW, float *out /*[C,H*W]*/, eg_stream_t stream);

/* ---- whole training step for one view, enqueued from native code (train_gaussians.py:81-106):
 * project+count -> offsets -> emit -> sort -> composite+loss -> composite bwd -> project bwd
 * (+absgrad, +Adam when hyper != NULL).  All buffers caller-owned. */
typedef struct {
  /* trainable state, raw representation (log-scales, logit-opacities) */
  float *means, *quats, *log_scales, *logit_opacities;
  float *adam_m, *adam_v, *absgrads;
  int32_t N;
  /* view */
  const float *viewmat, *K, *gt, *wmap;
  int32_t width, height;
  float loss_scale; /* lambda_projection (train_gaussians.py:98) */
  /* workspace */
  float *splat, *g2d;
  int32_t *tile_counts, *offsets, *item_offsets, *total; /* [T], [T+1], [T+1], [4] */
  uint32_t *tile_mask;                                   /* [N] */
  int32_t *ticket;                                       /* [T + 2], zero-initialised once ([0]: the scan's ticket; [1..T+1]:
                                                            see EG_FLAG_FRONT_PREFIX, used on tile grids above 2560 tiles) */
  void *workspace;   /* eg_composite_workspace_bytes(max_items, T) bytes, control prefix zeroed at allocation */
  int64_t max_items; /* >= ceil(capacity/128) + T */
  uint64_t *keys;
  int32_t *flatten_ids;
  int64_t capacity;
  int32_t max_tile_hint;                /* see eg_sort_pairs; 0 = unknown */
  int32_t rewalk_hint;                  /* see eg_composite_fwd; 0 = none seen, < 0 = unknown */
  int32_t seg_cap;                      /* > 0: segmented binning (eg_project_emit ...): keys / flatten_ids are
                                           [T * seg_cap], offsets / item_offsets serve as tile_start / item_first */
  int32_t *tile_end, *item_end, *item_tile; /* [T], [T], [max_items]; used when seg_cap > 0 */
  float *render, *alphas, *vpix, *loss; /* [H,W], [H,W], [H,W], [1] accumulated */
  float *gtstop;                        /* [H,W,3] */
  int32_t *last_ids;
  /* gradient outputs (used when adam == NULL, e.g. before an RCCL all-reduce) */
  float *v_means, *v_quats, *v_scales, *v_opacities;
  const eg_adam_hyper *adam_host; /* NULL = write gradients instead of stepping */
  /* tail fusion across the step boundary (segmented layout, adam_host != NULL): when next_viewmat != NULL the
   * step's last kernel also projects + bins the NEXT view with the parameters it has just updated, and the next
   * eg_train_step call on that view passes have_projection = 1 to skip its own projection.  Any change to the
   * parameters or the buffers in between voids the projection (the caller then passes 0). */
  const float *next_viewmat, *next_K;
  int32_t have_projection;
  /* > 0: when pixels reach the transmittance stop (rewalk_hint != EG_REWALK_SPECULATE) the forward runs as ONE
   * chained kernel (slice products, phase B and the exact stop by decoupled look-back; same results as the
   * slice / combine / re-walk sequence).  The tag marks the items published in THIS call: it must differ from the
   * tags of all earlier calls on the same workspace since that workspace was zeroed (count up from 1;
   * eg_train_steps uses ws_tag .. ws_tag + K - 1).  0: the slice / combine / re-walk sequence. */
  int32_t ws_tag;
  /* optional (segmented layout): the sort kernel then also leaves one 16-byte record per item -- item_rec [max_items, 4]
   * = {tile, slice | slices << 16, ws_tag of this call, end of the tile's keys} (the slice's first key is tile * seg_cap +
   * 128 * slice), in the order the forward dispatches its workgroups in.  The record INDEX is not the item number -- the
   * hand-over storage is addressed through item_first -- and the table may have HOLES: the forward takes a record for
   * this call's iff word 2 equals ws_tag, so the table must be zeroed whenever the workspace is (tag wrap).  Tile grids of
   * of 512 .. 2560 tiles (eg_record_xcd_shift), one view per launch: XCD-aware placement -- workgroup b of a launch runs on XCD b % 8; the tiles are
   * dealt to the XCDs in 2 x 2-tile blocks, xcd = (block_x + 3 block_y) % 8, and XCD x's records sit at indices 8 k + x,
   * slices 0..3 of its tiles first, their deeper slices after them; the table spans 8 x the longest of the eight lists
   * (beyond max_items: the sticky overflow word).  Larger grids / batched launches: slices [0, 9) resp. [0, 4) of all
   * tiles first, then the deeper slices, no holes.  Grids of <= 2560 tiles, fused loss (wmap != NULL, no images wanted): an
   * EMPTY tile other than the last one has NO record -- its sort workgroup adds the background's loss term
   * sum_p wmap_p |gt_p| to the forward's partial sums and leaves; its gtstop pixels and its tile / item table entries are
   * not written (no Gaussian's footprint reaches them: the exact tile test is conservative), its one empty item stays in the
   * numbering.  The step
   * (no images wanted,
   * ws_tag > 0) runs the wave-autonomous forward, whose hand-over granules carry ws_tag: 1 <= ws_tag <= EG_MAX_WS_TAG,
   * different from the tag of every earlier call on this workspace since the workspace was last zeroed (the WHOLE
   * workspace must be zero before its first use; a caller that runs out of tags zeroes it again -- every 65 534
   * steps).  A wave of that kernel that polls a hand-over granule for tens of milliseconds without seeing it gives up
   * and raises bit 1 of control word 3 of the workspace (sticky, next to bit 0 = "a pixel stopped in speculative
   * mode"): the results of that call are void and the caller must say so.  Batched step: [C, max_items, 4]. */
  int32_t *item_rec;
  /* Round 6.  Inside a run of steps (adam_host, next_viewmat, segmented layout) on a tile grid of <= 2560 tiles (round 6: 2048 -> 2560, the reference's native 800 x 800 = 2500 tiles included) a scene of
   * up to 32768 Gaussians (eg_backward_is_fused) runs the step's backward as ONE kernel: every workgroup walks the
   * footprints of its 64 Gaussians (compositing VJP), then its first wave runs their projection VJP, absgrads, Adam and the
   * next view's projection + binning -- the same functions, the same arithmetic per Gaussian, bit-identical parameters; the
   * g2d record stays on chip (args.g2d is not written).  != 0: the two kernels of rounds 1-5 (footprint backward, then
   * projection backward + Adam + next projection), g2d written.  Everywhere else -- larger scenes, larger grids, no next
   * view, the gradient form -- the field is ignored (the two kernels run). */
  int32_t two_kernel_backward;
} eg_step_args;
#define EG_MAX_WS_TAG 0xfffe

/* (Segmented layout, tile grids of <= 2560 tiles: eg_train_step / eg_train_steps / eg_train_step_batched run the
 * projection kernels without their ticket + scan tail -- `ticket` is then unused -- let every tile's sort workgroup
 * form its item prefix from the cursors, and have the compositing kernel return the cursors to zero.  Same tables,
 * same results; the stand-alone entries eg_project_emit / eg_sort_segments / eg_composite_fwd_segments keep the
 * protocol described with them.) */
int eg_train_step(const eg_step_args *args_host, eg_stream_t stream);

/* 1 when a step of a run (adam_host, next_viewmat, segmented layout, two_kernel_backward == 0) on a scene of n_gaussians
 * and a grid of n_tiles tiles runs its backward as the ONE kernel described at eg_step_args.two_kernel_backward, else 0:
 * what a caller labels its measurements with (bench.py). */
int eg_backward_is_fused(int32_t n_gaussians, int32_t n_tiles);

/* The XCD-aware placement of the item records (eg_step_args::item_rec) on a grid of n_tiles tiles with one view per launch:
 * tiles per block side = 2^result, 0 = dense records.  A caller sizes max_items for 8 x the longest of the eight lists. */
int eg_record_xcd_shift(int32_t n_tiles);

/* ---- K consecutive steps by one native call: step k = eg_train_step on view views_host[k] (taken out of the
 * [V,4,4] / [V,3,3] / [V,H,W] arrays) with weight map wmaps_host[k] and every active Adam step count of
 * args_host->adam_host advanced by k.  args_host's own viewmat / K / gt / wmap are ignored. */
int eg_train_steps(const eg_step_args *args_host, int32_t K, const int32_t *views_host,
                   const float *const *wmaps_host, const float *viewmats, const float *Ks, const float *gts,
                   eg_stream_t stream);

/* ---- S independent scenes side by side on ONE GPU (BASELINE configs[4], "one scene per GPU", with S of them sharing
 * a device: at the reference's sizes a single scene's dependent launches sit at their latency floor and leave most of
 * the chip idle).  K steps of every scene by one native call: scene s = eg_train_steps(args_host[s], K, views_host[s],
 * wmaps_host[s], viewmats[s], Ks[s], gts[s]) on streams[s] (distinct streams, distinct buffers: the scenes share
 * nothing, every scene's result equals its solo run).  n_threads host threads (1 .. S) share the scenes -- thread j
 * takes scenes j, j + n_threads, ... -- and walk them round-robin, step k of all their scenes before step k + 1 of any.
 * The reference trains one scene per process (train_gaussians.py:311); this is the throughput form of its 115-scan sweep.
 * Not inside an eg_timing_begin window.  Returns the first failing scene's code (eg_last_error_string names scene and step). */
int eg_train_steps_multi(int32_t S, const eg_step_args *const *args_host, int32_t K, const int32_t *const *views_host,
                         const float *const *const *wmaps_host, const float *const *viewmats, const float *const *Ks,
                         const float *const *gts, const eg_stream_t *streams, int32_t n_threads);

/* ---- C <= EG_MAX_BATCH views per launch sequence.  `args_host` as for eg_train_step
 * (segmented layout required; its viewmat / K / gt / wmap fields are ignored) with every per-view work buffer
 * holding C consecutive copies: splat, g2d [C,N,8]; tile_counts, offsets, tile_end, item_offsets, item_end [C,T];
 * item_tile [C,max_items]; keys, flatten_ids [C,T*seg_cap]; total [C,4]; ticket [C]; gtstop [C,H,W,3];
 * workspace = C blocks of eg_batched_workspace_stride(max_items, T) bytes (control prefix of each zeroed at
 * allocation).  render / alphas / last_ids / vpix are not materialised.  The gradients of the C views are
 * SUMMED (= what an all-reduce over C data-parallel ranks forms), absgrads += the C per-view increments, then
 * one Adam step (adam_host != NULL) or the summed gradients are written (v_means ...).  viewmats / Ks / gts /
 * wmaps: host arrays of C device pointers.  The reference steps after every view (train_gaussians.py:104-106):
 * this is a throughput mode with the semantics of data parallelism, on one GPU. */
int64_t eg_batched_workspace_stride(int64_t max_items, int64_t n_tiles);
int eg_train_step_batched(const eg_step_args *args_host, int32_t C, const float *const *viewmats,
                          const float *const *Ks, const float *const *gts, const float *const *wmaps,
                          eg_stream_t stream);

/* ---- C cameras per native call for the GENERAL operator path (gsplat.rasterization takes viewmats [C,4,4]; the reference
 * itself calls it with one camera, edge_gs.py:250-268).  Native loops over the per-camera entries above -- the same kernels,
 * the same results -- with the cameras' arrays as [C, ...] blocks, or as HOST arrays of C device pointers where the sizes
 * differ per camera (the binning arrays of M_c entries).  One native call per stage, one host read-back for all totals. */
int eg_project_fwd_cams(const float *means, const float *quats, const float *scales, const float *opacities,
                        const float *viewmats /*[C,4,4]*/, const float *Ks /*[C,3,3]*/, int32_t N, int32_t C, int32_t width,
                        int32_t height, float near_plane, float far_plane, float eps2d, float radius_clip, uint32_t flags,
                        float *splat /*[C,N,8]*/, int32_t *radii, float *means2d, float *depths, float *conics,
                        float *compensations, int32_t *tiles_per_gauss /*[C,N] each, NULL ok*/,
                        int32_t *tile_counts /*[C,T]*/, eg_stream_t stream);
/* v_means [N,3], v_quats [N,4], v_scales [N,3] = SUM over the cameras (camera 0 writes, the others add) */
int eg_project_bwd_cams(const float *means, const float *quats, const float *scales, const float *opacities,
                        const float *viewmats, const float *Ks, int32_t N, int32_t C, int32_t width, int32_t height, float eps2d,
                        uint32_t flags, const float *splat /*[C,N,8]*/, const float *g2d /*[C,N,8]*/,
                        const float *v_comps_ext /*[C,N]*/, const float *v_depths_ext /*[C,N] or NULL*/, float *v_means,
                        float *v_quats, float *v_scales, eg_stream_t stream);
int eg_tile_offsets_cams(const int32_t *tile_counts /*[C,T]*/, int32_t T, int32_t C, int64_t capacity,
                         int32_t *offsets /*[C,T+1]*/, int32_t *item_offsets /*[C,T+1]*/, int32_t *total /*[C,4]*/,
                         eg_stream_t stream);
int eg_tile_emit_sort_cams(const float *means2d /*[C,N,2]*/, const int32_t *radii /*[C,N]*/, const float *depths /*[C,N]*/,
                           int32_t N, int32_t C, int32_t width, int32_t height, const int32_t *offsets /*[C,T+1]*/,
                           int32_t *tile_counts /*[C,T], returned to zero*/, const int64_t *M_host /*[C]*/,
                           uint64_t *const *keys, int32_t *const *flatten_ids, int64_t *const *isect_ids /*NULL ok*/,
                           const int32_t *max_tile_host /*[C] or NULL*/, eg_stream_t stream);
int eg_composite_fwd_cams(int32_t C, const float *splat /*[C,N,8]*/, int32_t N, const float *colors, int32_t colors_per_camera,
                          int32_t channels, const int32_t *const *offsets, const int32_t *const *flatten_ids, int32_t width,
                          int32_t height, float *render /*[C,H,W,channels]*/, float *alphas /*[C,H,W]*/,
                          int32_t *last_ids /*[C,H,W]*/, const int32_t *const *item_offsets, const int32_t *const *total,
                          const int64_t *max_items_host, void *const *workspace, float *gtstop /*[C,H,W,3] or NULL*/,
                          eg_stream_t stream);
int eg_composite_bwd_footprint_cams(const float *splat /*[C,N,8]*/, int32_t N, int32_t C, int32_t width, int32_t height,
                                    const float *gtstop /*[C,H,W,3]*/, float *g2d /*[C,N,8]*/, eg_stream_t stream);

/* ---- the drop-in operator's fast path in two calls (edgegaussians_amd/rasterizer.py: the reference's own call of
 * gsplat.rasterization -- one camera, colours == 1 without grad, edge_gs.py:247-279 -- and its autograd backward).
 * eg_operator_fwd: projection + exact tile binning -> per-tile sort -> the training step's wave-autonomous forward in its
 * exact mode, with the accumulated-alpha image as output and T_final in the gtstop record (no fused loss); means2d is
 * copied out of the packed record; total[4] = 1 iff every entry of `colors` is 1, total[5] = control word 3 of the workspace
 * (bit 1: a look-back poll of the forward gave up: the outputs are void).  The per-call outputs (splat, alphas,
 * means2d, gtstop) are the caller's fresh tensors, everything else its cached work buffers; total is [8] int32
 * ([0..3] as everywhere, [1] the sticky overflow flag: the caller reads total[0..4] back ONCE after the call).
 * eg_operator_bwd: rec = gtstop with word 0 scaled by the upstream gradient v_alphas[p * v_stride] -> footprint backward
 * -> absgrad_out [N,2] = sum over pixels of |dL/dmean2d| (what the reference reads as means2d.absgrad, edge_gs.py:612)
 * -> (+ v_means2d when someone differentiates through info["means2d"]) -> projection backward. */
typedef struct {
  const float *means, *quats, *scales, *opacities, *colors; /* colors [N, color_channels] or NULL */
  int32_t color_channels;
  const float *viewmat, *K;
  int32_t N, width, height;
  uint32_t flags;                      /* EG_FLAG_ANTIALIASED | EG_FLAG_LOG_SCALES | EG_FLAG_LOGIT_OPACITIES */
  float *splat, *alphas, *means2d, *gtstop; /* [N,8], [H,W], [N,2]|NULL, [H,W,3] */
  int32_t *tile_counts, *tile_start, *tile_end, *item_first, *item_end, *item_tile, *item_rec;
  int32_t *total;                      /* [8] */
  int32_t *ticket;                     /* [T + 2], zero-initialised once */
  uint64_t *keys;
  int32_t *flatten_ids;
  int32_t seg_cap, max_tile_hint;
  int64_t max_items;
  void *workspace;                     /* eg_composite_workspace_bytes(max_items, T), zeroed at allocation */
  int32_t ws_tag;                      /* as eg_step_args.ws_tag: fresh per call, 1 .. EG_MAX_WS_TAG */
} eg_operator_args;
int eg_operator_fwd(const eg_operator_args *a, eg_stream_t stream);
int eg_operator_bwd(const eg_operator_args *a, const float *v_alphas, int64_t v_stride, float *rec /*[H,W,3] scratch*/,
                    float *g2d /*[N,8]*/, float *absgrad_out /*[N,2]|NULL*/, const float *v_means2d /*[N,2]|NULL*/,
                    float *v_means, float *v_quats, float *v_scales, float *v_opacities, eg_stream_t stream);

/* ---- native data-parallel run (edgegaussians_amd/dist.py drives it).  RCCL is dlopen'ed from
 * `librccl_path` (NULL / "": "librccl.so" by the loader's search path) -- the library PyTorch ships, so that the process
 * holds ONE RCCL -- and the communicator is created from a 128-byte ncclUniqueId: rank 0 calls eg_dp_unique_id, the
 * ranks exchange the bytes (torch.distributed broadcast), every rank calls eg_dp_init on ITS device.  One communicator
 * per process (library state: see the conventions above).  eg_dp_world() = its size, 0 without one. */
int eg_dp_unique_id(const char *librccl_path, void *id_out_host /*128 bytes*/);
int eg_dp_init(const char *librccl_path, const void *id_host /*128 bytes*/, int32_t rank, int32_t world);
int eg_dp_world(void);
int eg_dp_shutdown(void);
/* the communicator's size as RCCL reports it (ncclCommCount): what bench.py puts on its line as the proof that RCCL saw
 * N ranks; 0 without a communicator, negative on an error */
int eg_dp_comm_count(void);
/* test / measurement switch.  on > 0 makes eg_train_steps_dp issue its [12 N] ncclAllReduce through a ONE-rank
 * communicator too (a sum over one rank is the identity, the run skips it by default: RCCL does it with copy-engine blits
 * that stall the stream).  on < 0 (round 6, MEASUREMENT ONLY): the collective is left out with N ranks as well -- every
 * rank then steps on its own view's gradient, the replicas DIVERGE and the run is no longer the reference's data-parallel
 * step; bench.py times one such window after all its valid ones ("step time with the collective compiled out") so that a
 * scaling curve separates RCCL's latency from everything else.  0: default.  Reset by eg_dp_shutdown. */
int eg_dp_force_all_reduce(int32_t on);
/* [12 N] gradient collectives issued by eg_train_steps_dp since eg_dp_init (and the floats they carried) */
int64_t eg_dp_grad_all_reduces(int64_t *floats_out_host /*NULL ok*/);
/* measurement aid: HIP events on the launch stream around the [12 N] collective of the next n_steps steps of
 * eg_train_steps_dp; _end synchronises once: mean / max microseconds the collective occupied the launch stream
 * (everything in it is exposed at one view per rank), number of collectives timed */
int eg_dp_comm_timing_begin(int32_t n_steps);
int eg_dp_comm_timing_end(float *mean_us_host, float *max_us_host, int32_t *n_out_host);
/* in-place sum over the ranks of n floats on `stream` (small collectives that ride the same communicator) */
int eg_dp_all_reduce(float *buf, int64_t n, eg_stream_t stream);
/* measurement aid: mean HOST microseconds per step that eg_train_steps_dp spent enqueueing {the gradient step's kernels,
 * ncclAllReduce, Adam + next projection} since the last call; returns the number of steps averaged */
int64_t eg_dp_host_profile(double *us_out_host /*[3]*/);
/* K consecutive view-sharded optimizer steps by ONE native call; per step: eg_train_step in its gradient form ->
 * ncclAllReduce(sum, fp32) of the fused [12 N] gradient buffer on the launch stream -> eg_adam_emit (the four Adam steps
 * on the reduced gradient + projection / binning of this rank's next view; eg_adam_multi when there is none).  `a`: step
 * 0 as for eg_train_step with adam_host == NULL and v_means / v_quats / v_scales / v_opacities / absgrads the blocks of
 * ONE contiguous buffer [means 3N | quats 4N | log-scales 3N | logit-opacity N | absgrad increment N]; its view inputs
 * are ignored: step k takes view views_host[k] of the [V, ...] arrays and weight map wmaps_host[k].  hyper: step 0's
 * Adam state (counts advance by k); absgrads: the accumulator that receives the reduced increment; next_view_after: the
 * view this rank rasterises in the step after the run (its projection rides in the last Adam launch; the caller then
 * passes have_projection = 1), or < 0.  Same results as the three calls enqueued one by one. */
int eg_train_steps_dp(const eg_step_args *a, const eg_adam_hyper *hyper, float *absgrads, int32_t K,
                      const int32_t *views_host, const float *const *wmaps_host, const float *viewmats /*[V,4,4]*/,
                      const float *Ks /*[V,3,3]*/, const float *gts /*[V,H,W]*/, int32_t next_view_after,
                      eg_stream_t stream);

/* ---- measurement aid: between eg_timing_begin(n) and eg_timing_end(), the next n eg_train_step
 * calls record HIP events between their stages on the launch stream; eg_timing_end synchronises
 * once and returns the average microseconds per stage (eg_timing_stage_count() entries, names by
 * eg_timing_stage_name(i)).  Not thread safe; one window at a time. */
int eg_timing_begin(int32_t n_steps);
/* tracing aid: eg_roctx_enable(1) dlopen's the ROCTx library and makes eg_train_step wrap its stages in
 * roctx ranges (eg:project_bin, eg:tile_sort, eg:composite_fwd, eg:footprint_bwd, eg:project_bwd_adam) -- rocprofv3
 * --marker-trace shows them next to the kernel trace; eg_roctx_enable(0) turns them off again (the default). */
int eg_roctx_enable(int32_t on);
/* debugging aid (process started with EG_FWD_PROF=1: the wave-autonomous forward runs its timed instantiation): the
 * per-wave phase records of the LAST forward launch, [items][4 quadrants][8] 64-bit words of shader-clock ticks (head,
 * staging, walk, publish, look-back, exact stop, epilogue; word 7 = 1 marks a wave that ran).  Returns the number of
 * 8-word records copied to the HOST buffer `out_host` (<= max_records) or a negative code.  Synchronises the device. */
int64_t eg_debug_fwd_profile(uint64_t *out_host, int64_t max_records);
int eg_timing_end(float *stage_us_host, int32_t *n_steps_out_host);
int eg_timing_stage_count(void);
const char *eg_timing_stage_name(int32_t i);

#ifdef __cplusplus
}
#endif
#endif /* EDGEGS_H */
