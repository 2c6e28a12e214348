// The drop-in operator's fast path as TWO native entries: what `gsplat.rasterization(...)` and its
// autograd backward cost on the HOST when the reference's own model class calls them (edge_gs.py:247-279: one camera,
// colours == 1 without grad) was 197-227 us + 263-377 us of Python for ~100 us of device work -- three 20-argument ctypes
// calls, six tensor allocations and an index_select per forward, five allocations, a clone, a scale and two calls per
// backward.  Here: ONE call each over a cached argument block.
//   forward : projection + exact tile binning (eg_project_emit's kernel) -> per-tile sort with item records -> the
//             wave-autonomous forward of the training step in its exact (chained) mode with the alpha image as an extra
//             output and no fused loss (the record carries T_final) -> means2d copied out of the packed record -> "are
//             the colours all ones" folded into total[4] (the caller's ONE read-back tells it together with M and the
//             overflow flag; total[5] carries the forward's "a look-back poll gave up" bit)
//   backward: record x upstream gradient -> footprint backward -> absgrad copied out -> projection backward
#include "common.h"
#include "composite.h"

namespace eg {

__global__ void colors_are_ones_kernel(const float *__restrict__ c, long long n, int *__restrict__ flag) {
  bool ok = true;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    ok = ok && (c[i] == 1.f);
  if (__ballot(!ok) != 0ull && (threadIdx.x & 63) == 0) atomicAnd(flag, 0);
}
// total[4] = 1 (colours are ones until a mismatch clears it), total[5] = control word 3 of the compositing workspace
// (bit 1: a look-back poll of the forward gave up; sticky)
__global__ void verdict_words_kernel(int *total, const int *ctl) { total[4] = 1; total[5] = ctl[3]; }

// rec[p] = {gtstop[p].gT * v[p], stop id, stop depth}
__global__ void scale_record_kernel(const StopRec *__restrict__ src, const float *__restrict__ v, long long v_stride,
                                    StopRec *__restrict__ dst, int n) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  StopRec r = src[p];
  r.gT *= v[(long long)p * v_stride];
  dst[p] = r;
}
__global__ void add_means2d_grad_kernel(float *__restrict__ g2d, const float *__restrict__ v, int n) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  g2d[8 * (size_t)g + 0] += v[2 * (size_t)g + 0];
  g2d[8 * (size_t)g + 1] += v[2 * (size_t)g + 1];
}

}  // namespace eg

using namespace eg;

extern "C" int eg_operator_fwd(const eg_operator_args *a, eg_stream_t stream) {
  EG_REQUIRE(a != nullptr, "null args");
  EG_REQUIRE(a->N > 0 && a->width > 0 && a->height > 0 && a->seg_cap > 0 && a->max_items > 0, "bad sizes");
  EG_REQUIRE(a->means && a->quats && a->scales && a->opacities && a->viewmat && a->K && a->splat && a->alphas && a->gtstop &&
                 a->tile_counts && a->tile_start && a->tile_end && a->item_first && a->item_end && a->item_tile &&
                 a->item_rec && a->total && a->ticket && a->keys && a->flatten_ids && a->workspace,
             "null pointer");
  EG_REQUIRE(a->ws_tag >= 1 && a->ws_tag <= EG_MAX_WS_TAG, "ws_tag out of range");
  const int tw = cdiv(a->width, kTile), th = cdiv(a->height, kTile), T = tw * th;
  EG_REQUIRE((int64_t)T * a->seg_cap < (1ll << 31), "T * seg_cap must fit 31 bits");
  hipStream_t st = as_stream(stream);
  const bool prefix_here = T <= kPrefixHereMaxTiles;
  const uint32_t flags = (a->flags & (EG_FLAG_ANTIALIASED | EG_FLAG_LOG_SCALES | EG_FLAG_LOGIT_OPACITIES)) | EG_FLAG_TIGHT_TILES |
                         (prefix_here ? 0u : EG_FLAG_FRONT_PREFIX);
  int rc = launch_project_emit(a->means, a->quats, a->scales, a->opacities, a->viewmat, a->K, a->N, a->width, a->height, flags,
                               a->splat, a->tile_counts, a->seg_cap, a->keys, a->item_first, (int32_t)a->max_items, a->total,
                               prefix_here ? nullptr : a->ticket, Batch{}, 1, st);
  if (rc) return rc;
  rc = launch_sort_segments(a->keys, a->tile_counts, T, a->seg_cap, a->flatten_ids, a->tile_start, a->tile_end, a->item_first,
                            a->item_end, a->item_tile, (int32_t)a->max_items, a->max_tile_hint, Batch{}, 1, st,
                            prefix_here ? a->total : nullptr, a->item_rec, prefix_here ? nullptr : a->ticket + 1,
                            (uint32_t)a->ws_tag, tw, nullptr, nullptr, nullptr, 0, 0, kFrontChained, a->total);
  if (rc) return rc;
  const TileTable tt = {a->tile_start, a->tile_end, a->item_first, a->item_end, a->item_tile,
                        prefix_here ? a->tile_counts : nullptr, (const int4 *)a->item_rec, a->seg_cap};
  rc = launch_wave_fwd((const float4 *)a->splat, tt, a->flatten_ids, a->width, a->height, nullptr, nullptr, 1.f, a->total,
                       a->max_items, a->workspace, a->gtstop, /*chained=*/1, (unsigned)a->ws_tag, a->max_tile_hint, st, Batch{}, 1,
                       a->alphas);
  if (rc) return rc;
  if (a->means2d &&
      hipMemcpy2DAsync(a->means2d, 2 * sizeof(float), a->splat, 8 * sizeof(float), 2 * sizeof(float), (size_t)a->N,
                       hipMemcpyDeviceToDevice, st) != hipSuccess)
    return check_launch("operator_fwd(means2d)");
  // total[4] = 1 iff every colour entry equals 1 (read back by the caller together with total[0..3])
  verdict_words_kernel<<<1, 1, 0, st>>>(a->total, carve_workspace(a->workspace, a->max_items, T).ctl);
  if (a->colors && a->color_channels > 0) {
    const long long n = (long long)a->N * a->color_channels;
    colors_are_ones_kernel<<<(unsigned)min((long long)1024, (n + 255) / 256), 256, 0, st>>>(a->colors, n, a->total + 4);
  }
  return check_launch("operator_fwd");
}

extern "C" int eg_operator_bwd(const eg_operator_args *a, const float *v_alphas, int64_t v_stride, float *rec, float *g2d,
                               float *absgrad_out, const float *v_means2d, float *v_means, float *v_quats, float *v_scales,
                               float *v_opacities, eg_stream_t stream) {
  EG_REQUIRE(a != nullptr && v_alphas && rec && g2d && v_means && v_quats && v_scales && v_opacities, "null pointer");
  EG_REQUIRE(a->N > 0 && a->width > 0 && a->height > 0 && v_stride >= 1, "bad sizes");
  hipStream_t st = as_stream(stream);
  const int hw = a->width * a->height;
  scale_record_kernel<<<cdiv(hw, 256), 256, 0, st>>>((const StopRec *)a->gtstop, v_alphas, v_stride, (StopRec *)rec, hw);
  int rc = launch_footprint_bwd(a->splat, a->N, a->width, a->height, rec, g2d, Batch{}, 1, st);
  if (rc) return rc;
  if (absgrad_out &&
      hipMemcpy2DAsync(absgrad_out, 2 * sizeof(float), g2d + 2, 8 * sizeof(float), 2 * sizeof(float), (size_t)a->N,
                       hipMemcpyDeviceToDevice, st) != hipSuccess)
    return check_launch("operator_bwd(absgrad)");
  if (v_means2d) add_means2d_grad_kernel<<<cdiv(a->N, 256), 256, 0, st>>>(g2d, v_means2d, a->N);
  return eg_project_bwd(a->means, a->quats, a->scales, a->opacities, a->viewmat, a->K, a->N, a->width, a->height, 0.3f,
                        a->flags & (EG_FLAG_ANTIALIASED | EG_FLAG_LOG_SCALES | EG_FLAG_LOGIT_OPACITIES), a->splat, g2d, nullptr,
                        nullptr, v_means, v_quats, v_scales, v_opacities, nullptr, stream);
}
