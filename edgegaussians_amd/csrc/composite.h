// Shared definitions of the compositing kernels (composite.hip: workgroup-per-item kernels, general API;
// composite_wave.hip: the wave-autonomous forward of the training step).
#pragma once
#include "common.h"

namespace eg {

constexpr int kWaveSlice = 256;               // most Gaussians one wave of the wave-autonomous forward stages at a time
constexpr unsigned kGranuleTagMask = 0xffffu;  // its hand-over granules carry a 16-bit call tag (composite_wave.hip)
#ifndef EG_ANCHOR_SHIFT
#define EG_ANCHOR_SHIFT 3
#endif
constexpr int kAnchorShift = EG_ANCHOR_SHIFT;  // every 8th slice of a tile publishes an inclusive granule (composite_wave.hip)
constexpr int kSlice = 128;  // Gaussians per item (half the LDS of 256 => 8 workgroups/CU, 2x the items)

// Where a tile's sorted ids and its items (128-Gaussian slices) live.  Classic layout: start = offsets,
// end = offsets + 1, item_first = item_offsets, item_end = item_offsets + 1, item_tile = nullptr (the
// owner of an item is found by search).  Segmented layout (eg_sort_segments): four explicit arrays and
// the item -> tile map.
struct TileTable {
  const int *start, *end, *item_first, *item_end, *item_tile;
  // != nullptr (the training step when the sort kernel forms the tile prefix itself, binning.hip): the binning
  // cursors [T], which the first slice workgroup of every tile returns to zero for the next projection
  int *cursor_reset;
  // optional (training step): the sort kernel's per-item records {tile, slice | slices << 16, call tag, end of the
  // tile's keys} (binning.hip, SegTable::item_rec / rec_tag)
  const int4 *item_rec;
  int seg_cap = 0;    // with item_rec: keys per tile segment (a slice's first key is tile * seg_cap + 128 slice)
};

// pixel of thread `tid` in the slice-parallel kernels: wave w owns the 8x8 quadrant (w & 1, w >> 1)
// of the tile, lane l the pixel (l & 7, l >> 3) inside it (a compact 8x8 block culls far better
// against thin ellipses than a 4x16 strip)
__device__ __forceinline__ void quad_pixel(int tid, int &di, int &dj) {
  const int w = tid >> 6, l = tid & 63;
  di = ((w >> 1) << 3) + (l >> 3);
  dj = ((w & 1) << 3) + (l & 7);
}

// per-pixel epilogue shared by the slice (combine) and re-walk kernels: outputs, fused clamp + weighted L1
// (edge_gs.py:279,288-324) and the packed record the footprint backward reads.  Returns the loss term.
// Per-pixel record the fused forward leaves for the footprint backward (12 bytes, one dwordx3 load):
// v * T_final, and -- only for pixels whose front-to-back walk ended on the transmittance rule -- the
// id and the depth bits of the last contributing Gaussian, so that a candidate decides "at or before
// the stop" from the record alone (a per-visit gather of the stop Gaussian's depth cost 12 us/step
// on a trained-like scene).
struct StopRec {
  float gT;
  int stop_id;          // -1: the walk did not stop
  unsigned stop_depth;  // depth float bits of Gaussian stop_id; 0xffffffff with stop_id == -1, so that the backward's
                        // "behind the stop" test is ONE unsigned 64-bit compare of (depth bits << 32 | id)
};
constexpr unsigned kNoStopDepth = 0xffffffffu;
static_assert(sizeof(StopRec) == 12, "gtstop is [H,W,3] 32-bit words");

template <int CH>
__device__ __forceinline__ float finalize_pixel(int p, float T, int last, bool stopped, const int *__restrict__ flat,
                                                float *__restrict__ render, float *__restrict__ alphas,
                                                int *__restrict__ last_ids, bool has_loss, float gt_p, float w,
                                                float loss_scale, float *__restrict__ vpix,
                                                StopRec *__restrict__ gtstop, const float4 *__restrict__ splat) {
  const float pix = 1.f - T;  // unit colours, no background: sum_i alpha_i T_i == 1 - T_final
  // the images are optional in the fused training step, whose backward reads only the gtstop record
  if (alphas) alphas[p] = pix;
  if (last_ids) last_ids[p] = last;
  if (render) {
#pragma unroll
    for (int k = 0; k < CH; ++k) render[(size_t)p * CH + k] = pix;
  }
  float l = 0.f, v = 1.f;  // without the fused loss the record carries T_final itself (upstream gradient 1)
  if (has_loss) {  // gt_p, w: this pixel's target and weight, loaded by the caller ahead of its own work
    const float c0 = fminf(fmaxf(pix, 0.f), 1.f);
    const float d = c0 - gt_p;
    l = w * fabsf(d);
    const float sgn = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
    v = loss_scale * w * sgn;  // pix is in [0,1): the clamp always passes the gradient
    if (vpix) vpix[p] = v;
  }
  if (gtstop) {
    // v * T_final, and -- only for pixels whose walk stopped on the transmittance rule -- the id of
    // the last contributing Gaussian
    StopRec r;
    r.gT = (T < 1.f) ? v * T : 0.f;
    r.stop_id = stopped ? flat[last] : -1;
    r.stop_depth = stopped ? (unsigned)__float_as_int(splat[2 * r.stop_id + 1].z) : kNoStopDepth;
    gtstop[p] = r;
  }
  return l;
}

// per-pixel hand-off from the combine to the re-walk kernel: the slice in which the stop falls and the state
// before it
struct StopInfo {
  int slice;     // tile-local slice index, -1 = this pixel is final
  float T;       // transmittance before that slice
  int last;      // last contributor before that slice
};

// Scratch of the slice-parallel forward.  The control words must be ZERO before the first use; every launch
// sequence hands them back zeroed (tile tickets by the combining workgroup, item flags and the list counter by
// the re-walk kernel), so a workspace is zeroed once, when it is allocated.
struct SliceWs {
  int *tile_ticket;     // [T]   slices of the tile that have finished
  int *item_flags;      // [max_items] 1 = the item is on the re-walk list
  int *ctl;             // [4]   {list length, exit ticket of the re-walk kernel, longest list since the caller looked, missed re-walk}
  int *exit_grp;        // [64 x 16] first level of the re-walk kernel's exit ticket, one cache line per group
  int *ready;           // [max_items] chained forward: the caller's tag once the item's record is published
  float *sliceP;        // [max_items][256] transmittance product of the slice (written only by tiles with > 1 slice)
  int *sliceL;          // [max_items][256] its last contributor (global index into the sorted ids, -1 none)
  StopInfo *stopinfo;   // [T][256]
  int2 *rewalk;         // [max_items] (item, tile)
  unsigned char *sliceQ;  // [max_items][128] quadrant verdicts
  // wave-autonomous forward (composite_wave.hip): every wave owns one 8x8 quadrant of one item
  int *qticket;         // [T][4] slices of the (tile, quadrant) that have published (speculative mode)
  int *qready;          // [max_items][4] chained mode: the caller's tag once the (item, quadrant) record is published
  float *loss_part;     // [64] partial loss sums (spread over 64 addresses; folded into loss_out by the footprint backward)
  int *dead_hint;       // [2][T][4] (the first [T][4] in use) chained mode: call tag << 15 | (32767 - first slice that
                        // lies behind the stop of every pixel of the quadrant), kept by atomicMax
  unsigned char *sliceL8;  // [max_items][256] slice-local index of the last contributor, 255 = none (aliases sliceL)
  unsigned long long *anchor;  // [max_items / 8 + 2][256] inclusive granules of the anchor slices (chained mode)
};

// the per-view copies of a batched step (blockIdx.y = view; strides are zero for a single view)
__device__ __forceinline__ TileTable view_of(TileTable tt, const Batch &bt, int v) {
  tt.start += v * bt.tiles; tt.end += v * bt.tiles; tt.item_first += v * bt.tiles; tt.item_end += v * bt.tiles;
  if (tt.item_tile) tt.item_tile += v * bt.items;
  if (tt.cursor_reset) tt.cursor_reset += v * bt.tiles;
  if (tt.item_rec) tt.item_rec += v * bt.items;
  return tt;
}
__device__ __forceinline__ SliceWs view_of(SliceWs ws, const Batch &bt, int v) {
  const long long o = v * bt.ws_bytes;
  ws.tile_ticket = (int *)((char *)ws.tile_ticket + o); ws.item_flags = (int *)((char *)ws.item_flags + o);
  ws.ctl = (int *)((char *)ws.ctl + o); ws.exit_grp = (int *)((char *)ws.exit_grp + o);
  ws.ready = (int *)((char *)ws.ready + o);
  ws.sliceP = (float *)((char *)ws.sliceP + o);
  ws.sliceL = (int *)((char *)ws.sliceL + o); ws.stopinfo = (StopInfo *)((char *)ws.stopinfo + o);
  ws.rewalk = (int2 *)((char *)ws.rewalk + o); ws.sliceQ = (unsigned char *)ws.sliceQ + o;
  ws.qticket = (int *)((char *)ws.qticket + o); ws.qready = (int *)((char *)ws.qready + o);
  ws.loss_part = (float *)((char *)ws.loss_part + o); ws.sliceL8 = ws.sliceL8 + o;
  ws.dead_hint = (int *)((char *)ws.dead_hint + o);
  ws.anchor = (unsigned long long *)((char *)ws.anchor + o);
  return ws;
}

// workspace layout: control words first (they must be zero before the first use, see SliceWs):
//   tile_ticket i32[T] | item_flags i32[max_items] | ctl i32[4] | exit_grp i32[64 x 16] | ready i32[max_items]
//   | qticket i32[4 T] | qready i32[4 max_items] | loss_part f32[64] | dead_hint i32[2][4 T]
// then  sliceP f32[max_items][256] | sliceL i32[max_items][256] | stopinfo {i32,f32,i32}[T][256]
//       | rewalk int2[max_items] | sliceQ u8[max_items][128] | (8-byte aligned) anchor u64[max_items / 8 + 2][256]
inline int64_t ctl_bytes_aligned(int64_t max_items, int64_t n_tiles) {
  const int64_t words = n_tiles + 2 * max_items + 4 + 64 * 16 + 4 * n_tiles + 4 * max_items + 64 + 8 * n_tiles;
  return ((words + 3) & ~(int64_t)3) * (int64_t)sizeof(int32_t);
}

inline SliceWs carve_workspace(void *workspace, int64_t max_items, int n_tiles) {
  SliceWs ws;
  ws.tile_ticket = (int *)workspace;
  ws.item_flags = ws.tile_ticket + n_tiles;
  ws.ctl = ws.item_flags + max_items;
  ws.exit_grp = ws.ctl + 4;
  ws.ready = ws.exit_grp + 64 * 16;
  ws.qticket = ws.ready + max_items;
  ws.qready = ws.qticket + 4 * (size_t)n_tiles;
  ws.loss_part = (float *)(ws.qready + 4 * (size_t)max_items);
  ws.dead_hint = (int *)(ws.loss_part + 64);
  // (the data part starts 16-byte aligned: the hand-over granules are read and written as single 8-byte words)
  ws.sliceP = (float *)((char *)workspace + ctl_bytes_aligned(max_items, n_tiles));
  ws.sliceL = (int *)(ws.sliceP + (size_t)max_items * kTilePix);
  ws.stopinfo = (StopInfo *)(ws.sliceL + (size_t)max_items * kTilePix);
  ws.rewalk = (int2 *)(ws.stopinfo + (size_t)n_tiles * kTilePix);
  ws.sliceQ = (unsigned char *)(ws.rewalk + max_items);
  ws.sliceL8 = (unsigned char *)ws.sliceL;
  ws.anchor = (unsigned long long *)(((uintptr_t)(ws.sliceQ + (size_t)max_items * kSlice) + 7) & ~(uintptr_t)7);
  return ws;
}


// internal launcher of the wave-autonomous forward (composite_wave.hip); chained = 0: speculative (no pixel is expected
// to reach the transmittance stop: a stop raises ws.ctl[3]), != 0: exact stop inside; tag: this call's granule tag
// (1 .. kGranuleTagMask - 1, different from every earlier call's on this workspace since its granules were zeroed)
int launch_wave_fwd(const float4 *splat, const TileTable tt, const int32_t *flatten_ids, int width, int height,
                    const float *gt, const float *wmap, float loss_scale, const int32_t *total, int64_t max_items,
                    void *workspace, float *gtstop, int chained, unsigned tag, int max_tile_hint, hipStream_t s,
                    const Batch &bt, int C, float *alphas = nullptr);

}  // namespace eg
