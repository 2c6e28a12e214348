// C cameras per native call for the GENERAL operator path (round 6).
//
// gsplat.rasterization takes viewmats [C,4,4] (the reference calls it with C = 1: edge_gs.py:250-268, which takes the
// two-call fast path of operator.hip).  Until round 5 the general path looped over the cameras in Python: C x (projection,
// offsets + one host read-back, emit, sort, compositing) ctypes calls.  These entries take the cameras' arrays as [C, ...]
// blocks -- or, where the per-camera sizes differ (the binning arrays of length M_c), as host arrays of C device pointers
// -- and loop natively over the very per-camera launchers: one native call per stage, ONE host read-back for all cameras'
// totals, the same kernels and the same results as the per-camera entries (the footprint backward is one launch with
// gridDim.y = C).
#include "common.h"
#include "composite.h"

using namespace eg;

extern "C" int eg_project_fwd_cams(const float *means, const float *quats, const float *scales, const float *opacities,
                                   const float *viewmats, const float *Ks, int32_t N, int32_t C, int32_t width, int32_t height,
                                   float near_plane, float far_plane, float eps2d, float radius_clip, uint32_t flags,
                                   float *splat, int32_t *radii, float *means2d, float *depths, float *conics,
                                   float *compensations, int32_t *tiles_per_gauss, int32_t *tile_counts, eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && C >= 1 && width > 0 && height > 0, "bad sizes");
  if (N == 0) return EG_OK;
  EG_REQUIRE(viewmats && Ks && splat, "null pointer");
  const size_t n = (size_t)N, T = (size_t)cdiv(width, kTile) * cdiv(height, kTile);
  for (int c = 0; c < C; ++c) {
    const int rc = eg_project_fwd(means, quats, scales, opacities, viewmats + 16 * (size_t)c, Ks + 9 * (size_t)c, N, width,
                                  height, near_plane, far_plane, eps2d, radius_clip, flags, splat + 8 * n * c,
                                  radii ? radii + n * c : nullptr, means2d ? means2d + 2 * n * c : nullptr,
                                  depths ? depths + n * c : nullptr, conics ? conics + 3 * n * c : nullptr,
                                  compensations ? compensations + n * c : nullptr,
                                  tiles_per_gauss ? tiles_per_gauss + n * c : nullptr,
                                  tile_counts ? tile_counts + T * c : nullptr, nullptr, stream);
    if (rc) return rc;
  }
  return EG_OK;
}

// gradients SUMMED over the cameras into v_means / v_quats / v_scales (camera 0 writes, the others add: the order of the
// Python loop this replaces)
extern "C" int eg_project_bwd_cams(const float *means, const float *quats, const float *scales, const float *opacities,
                                   const float *viewmats, const float *Ks, int32_t N, int32_t C, int32_t width, int32_t height,
                                   float eps2d, uint32_t flags, const float *splat, const float *g2d, const float *v_comps_ext,
                                   const float *v_depths_ext, float *v_means, float *v_quats, float *v_scales,
                                   eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && C >= 1, "bad sizes");
  if (N == 0) return EG_OK;
  EG_REQUIRE(viewmats && Ks && splat && g2d && v_comps_ext && v_means && v_quats && v_scales, "null pointer");
  const size_t n = (size_t)N;
  for (int c = 0; c < C; ++c) {
    const int rc = eg_project_bwd(means, quats, scales, opacities, viewmats + 16 * (size_t)c, Ks + 9 * (size_t)c, N, width,
                                  height, eps2d, (flags & ~EG_FLAG_GRAD_ACCUM) | (c > 0 ? EG_FLAG_GRAD_ACCUM : 0u),
                                  splat + 8 * n * c, g2d + 8 * n * c, v_comps_ext + n * c,
                                  v_depths_ext ? v_depths_ext + n * c : nullptr, v_means, v_quats, v_scales, nullptr, nullptr,
                                  stream);
    if (rc) return rc;
  }
  return EG_OK;
}

// offsets / item_offsets [C, T + 1], total [C, 4] from tile_counts [C, T]: C scans, no host read-back in between
extern "C" int eg_tile_offsets_cams(const int32_t *tile_counts, int32_t T, int32_t C, int64_t capacity, int32_t *offsets,
                                    int32_t *item_offsets, int32_t *total, eg_stream_t stream) {
  EG_REQUIRE(T > 0 && C >= 1 && tile_counts && offsets && item_offsets && total, "bad arguments");
  for (int c = 0; c < C; ++c) {
    const int rc = eg_tile_offsets(tile_counts + (size_t)T * c, T, capacity, offsets + (size_t)(T + 1) * c,
                                   item_offsets + (size_t)(T + 1) * c, total + 4 * (size_t)c, stream);
    if (rc) return rc;
  }
  return EG_OK;
}

// key emission + per-tile sort of every camera; keys / flatten_ids / isect_ids: HOST arrays of C device pointers (camera c's
// arrays hold M_host[c] entries; isect_ids or its entries may be NULL), tile_counts [C, T] is returned to zero
extern "C" int eg_tile_emit_sort_cams(const float *means2d, const int32_t *radii, const float *depths, int32_t N, int32_t C,
                                      int32_t width, int32_t height, const int32_t *offsets, int32_t *tile_counts,
                                      const int64_t *M_host, uint64_t *const *keys, int32_t *const *flatten_ids,
                                      int64_t *const *isect_ids, const int32_t *max_tile_host, eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && C >= 1 && width > 0 && height > 0, "bad sizes");
  if (N == 0) return EG_OK;
  EG_REQUIRE(means2d && radii && depths && offsets && tile_counts && M_host && keys && flatten_ids, "null pointer");
  const size_t n = (size_t)N;
  const int T = cdiv(width, kTile) * cdiv(height, kTile);
  for (int c = 0; c < C; ++c) {
    EG_REQUIRE(M_host[c] >= 0 && (M_host[c] == 0 || (keys[c] && flatten_ids[c])), "null isect array");
    int rc = eg_tile_emit(means2d + 2 * n * c, radii + n * c, depths + n * c, nullptr, 0, N, width, height,
                          offsets + (size_t)(T + 1) * c, tile_counts + (size_t)T * c, M_host[c], keys[c], nullptr, stream);
    if (rc) return rc;
    rc = eg_sort_pairs(keys[c], offsets + (size_t)(T + 1) * c, T, M_host[c], flatten_ids[c],
                       isect_ids ? isect_ids[c] : nullptr, max_tile_host ? max_tile_host[c] : 0, stream);
    if (rc) return rc;
  }
  return EG_OK;
}

// compositing forward of every camera: splat [C, N, 8]; colors NULL (all ones), [N, D] (colors_per_camera = 0) or [C, N, D];
// images [C, H, W, ...]; offsets / flatten_ids / item_offsets / total / workspace: HOST arrays of C device pointers (the
// last three NULL or per-entry NULL = the one-workgroup-per-tile kernel); gtstop [C, H, W, 3] or NULL
extern "C" int eg_composite_fwd_cams(int32_t C, const float *splat, int32_t N, const float *colors, int32_t colors_per_camera,
                                     int32_t channels, const int32_t *const *offsets, const int32_t *const *flatten_ids,
                                     int32_t width, int32_t height, float *render, float *alphas, int32_t *last_ids,
                                     const int32_t *const *item_offsets, const int32_t *const *total,
                                     const int64_t *max_items_host, void *const *workspace, float *gtstop,
                                     eg_stream_t stream) {
  EG_REQUIRE(C >= 1 && N >= 0 && width > 0 && height > 0, "bad sizes");
  EG_REQUIRE(splat && offsets && flatten_ids && render && alphas && last_ids, "null pointer");
  const size_t n = (size_t)N, hw = (size_t)width * height;
  for (int c = 0; c < C; ++c) {
    const bool sliced = item_offsets && total && workspace && max_items_host && item_offsets[c] && total[c] && workspace[c] &&
                        max_items_host[c] > 0;
    const float *col = colors ? colors + (colors_per_camera ? n * (size_t)channels * c : 0) : nullptr;
    const int rc = eg_composite_fwd(splat + 8 * n * c, col, channels, offsets[c], flatten_ids[c], width, height,
                                    render + hw * (size_t)channels * c, alphas + hw * c, last_ids + hw * c, nullptr, nullptr,
                                    1.0f, nullptr, nullptr, sliced ? item_offsets[c] : nullptr, sliced ? total[c] : nullptr,
                                    sliced ? max_items_host[c] : 0, sliced ? workspace[c] : nullptr,
                                    (gtstop && sliced) ? gtstop + 3 * hw * c : nullptr, -1, stream);
    if (rc) return rc;
  }
  return EG_OK;
}

// footprint backward of every camera in ONE launch (gridDim.y = camera): splat / g2d [C, N, 8], gtstop [C, H, W, 3]
extern "C" int eg_composite_bwd_footprint_cams(const float *splat, int32_t N, int32_t C, int32_t width, int32_t height,
                                               const float *gtstop, float *g2d, eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && C >= 1 && C <= 65535 && width > 0 && height > 0, "bad sizes");
  if (N == 0) return EG_OK;
  EG_REQUIRE(splat && gtstop && g2d, "null pointer");
  Batch bt;
  bt.splat4 = 2ll * N;
  bt.pixels = (long long)width * height;
  return launch_footprint_bwd(splat, N, width, height, gtstop, g2d, bt, C, as_stream(stream));
}
