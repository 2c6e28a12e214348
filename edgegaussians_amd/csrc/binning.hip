// G2-G6: tile binning and the (tile, depth) ordering.
//
// Replaces gsplat 1.0.0 isect_tiles (2 passes) + torch.cumsum + cub::DeviceRadixSort::SortPairs +
// isect_offset_encode as reached from edgegaussians/models/edge_gs.py:250-268.
//
// MI355X-first design instead of a 6-pass global LSD radix sort over M 12-byte pairs
// (>= 18 dependent launches, each ~1.5-2 us of boundary on this chip -- longer than the
// compositing itself at the reference's sizes):
//   1. per-TILE counts by atomics (fused into the projection kernel)          -> tile_counts[T]
//   2. one single-workgroup exclusive scan over T <= ~8k tiles                -> isect_offsets[T+1]
//      (this IS gsplat's isect_offsets; no per-Gaussian cumsum is needed at all)
//   3. emit: each (Gaussian, tile) claims a slot in its tile's segment with a returning atomic that
//      counts the tile's counter back DOWN (so the counters are zero again for the next step) and
//      writes key = depth_bits << 32 | gaussian_id                            -> keys[M]
//   4. one workgroup per tile sorts its segment in LDS (bucket + rank on unique 64-bit keys;
//      bitonic network as the fallback)
// The result is bit-identical to gsplat's stable sort: inside a tile the order is by depth bits with
// ties broken by Gaussian id, which is exactly what a stable sort of index-ordered emissions gives.
// Segments larger than the LDS capacity fall back to a hybrid global/LDS bitonic sort by the
// same workgroup (slow path, correct for any size).
#include <cstdlib>

#include "common.h"
#include "composite.h"  // (carve_workspace: where the forward keeps its partial loss sums)

namespace eg {

// Per-tile counting with LDS privatisation.  All of a view's hot tile counters sit in a handful of
// cache lines (the object covers ~13x13 central tiles), so per-intersection global atomics serialise
// on those lines (measured: 250 us for 480k atomics).  Each 512-thread workgroup histograms its
// Gaussians into LDS first and flushes one global atomic per touched tile.
constexpr int kBinThreads = 512;
constexpr int kMaxLdsTiles = 16384;  // 64 KiB of counters; larger grids use the direct-atomic path

template <bool LDS>
__global__ void __launch_bounds__(kBinThreads)
tile_count_kernel(const float2 *__restrict__ means2d, const int *__restrict__ radii, int N, int width,
                  int height, int *__restrict__ tiles_per_gauss, int *__restrict__ tile_counts) {
  extern __shared__ __attribute__((aligned(16))) int s_hist[];
  const int tw = (width + kTile - 1) / kTile, th = (height + kTile - 1) / kTile, T = tw * th;
  if (LDS) {
    for (int t = threadIdx.x; t < T; t += kBinThreads) s_hist[t] = 0;
    __syncthreads();
  }
  const int g = blockIdx.x * kBinThreads + threadIdx.x;
  if (g < N) {
    const int radius = radii[g];
    int n = 0;
    if (radius > 0) {
      const float2 m = means2d[g];
      int x0, y0, x1, y1;
      tile_box(m.x, m.y, radius, tw, th, x0, y0, x1, y1);
      n = (y1 - y0) * (x1 - x0);
      for (int ty = y0; ty < y1; ++ty)
        for (int tx = x0; tx < x1; ++tx) atomicAdd(LDS ? &s_hist[ty * tw + tx] : &tile_counts[ty * tw + tx], 1);
    }
    if (tiles_per_gauss) tiles_per_gauss[g] = n;
  }
  if (LDS) {
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += kBinThreads) {
      const int c = s_hist[t];
      if (c) atomicAdd(&tile_counts[t], c);
    }
  }
}

// single workgroup, 1024 threads: exclusive scan of counts[T] -> offsets[T+1]; counts stay intact
// (the emit pass counts them back down to zero, so no memset is ever needed between steps).
// Also scans the per-tile SLICE counts ceil(n_t / 128) -> item_offsets[T+1]: the compositing kernels
// run one workgroup per (tile, 128-Gaussian slice) "item", so a tile holding thousands of Gaussians
// is spread over many CUs instead of serialising on one.
__device__ __forceinline__ int block_excl_scan_1024(int c, int *wave_sums, int &block_total) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int s = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(s, d, 64);
    if (lane >= d) s += o;
  }
  if (lane == 63) wave_sums[wv] = s;
  __syncthreads();
  if (wv == 0) {
    int ws = (lane < 16) ? wave_sums[lane] : 0;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      const int o = __shfl_up(ws, d, 64);
      if (lane >= d) ws += o;
    }
    if (lane < 16) wave_sums[lane] = ws;  // inclusive over waves
  }
  __syncthreads();
  const int wave_excl = (wv == 0) ? 0 : wave_sums[wv - 1];
  block_total = wave_sums[15];
  const int r = wave_excl + (s - c);
  __syncthreads();
  return r;
}

__global__ void __launch_bounds__(1024)
tile_offsets_kernel(const int *__restrict__ counts, int T, long long capacity, int *__restrict__ offsets,
                    int *__restrict__ item_offsets, int *__restrict__ total) {
  __shared__ int wave_sums[16];
  const int tid = threadIdx.x;
  int carry = 0, icarry = 0, cmax = 0;
  for (int base = 0; base < T; base += 1024) {
    const int i = base + tid;
    const int c = (i < T) ? counts[i] : 0;
    const int it = max(1, (c + 127) >> 7);  // 128-Gaussian slices (kSlice in composite.hip); an empty tile owns one
                                            // (empty) item: the forward finalises its pixels there
    int tot, itot;
    const int e = block_excl_scan_1024(c, wave_sums, tot);
    const int ie = block_excl_scan_1024(it, wave_sums, itot);
    if (i < T) {
      // clamped to the buffer capacity: on overflow (flagged in total[1]) every downstream range stays
      // inside the isect buffers instead of faulting
      offsets[i] = (int)min((long long)(carry + e), capacity);
      if (item_offsets) item_offsets[i] = icarry + ie;
    }
    carry += tot;
    icarry += itot;
    cmax = max(cmax, c);
  }
  // max tile population (sizes the sort variant that has to run)
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) cmax = max(cmax, __shfl_xor(cmax, d, 64));
  if ((tid & 63) == 0) wave_sums[tid >> 6] = cmax;
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w) cmax = max(cmax, wave_sums[w]);
    offsets[T] = (int)min((long long)carry, capacity);
    if (item_offsets) item_offsets[T] = icarry;
    if (total) {
      total[0] = carry;
      if ((long long)carry > capacity) total[1] = 1;  // sticky: only the host clears it
      total[2] = icarry;
      total[3] = cmax;
    }
  }
}

// emit: block-local histogram in LDS -> ONE returning global atomic per (block, touched tile) reserves
// a run of slots in the tile's segment -> block-local LDS cursors hand out the slots.
template <bool LDS>
__global__ void __launch_bounds__(kBinThreads)
tile_emit_kernel(const float2 *__restrict__ means2d, const int *__restrict__ radii,
                 const float *__restrict__ depths, const float4 *__restrict__ splat, unsigned flags, int N,
                 int width, int height, const int *__restrict__ offsets, int *__restrict__ cursor,
                 long long capacity, unsigned long long *__restrict__ keys, const unsigned *__restrict__ tile_mask) {
  extern __shared__ __attribute__((aligned(16))) int s_mem[];
  const int tw = (width + kTile - 1) / kTile, th = (height + kTile - 1) / kTile, T = tw * th;
  int *s_hist = s_mem, *s_base = s_mem + T;
  if (LDS) {
    for (int t = threadIdx.x; t < T; t += kBinThreads) s_hist[t] = 0;
    __syncthreads();
  }
  const int g = blockIdx.x * kBinThreads + threadIdx.x;
  float x = 0.f, y = 0.f, depth = 0.f;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  int radius = 0;
  if (g < N) {
    if (splat) {
      s0 = splat[2 * g]; s1 = splat[2 * g + 1];
      x = s0.x; y = s0.y; depth = s1.z; radius = __float_as_int(s1.w);
    } else {
      const float2 m = means2d[g];
      x = m.x; y = m.y; depth = depths[g]; radius = radii[g];
    }
  }
  int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
  if (radius > 0) {
    if (splat && (flags & EG_FLAG_TIGHT_TILES)) tile_box_tight(x, y, radius, s0.z, s0.w, s1.x, s1.y, tw, th, x0, y0, x1, y1);
    else tile_box(x, y, radius, tw, th, x0, y0, x1, y1);
  }
  const unsigned long long key = ((unsigned long long)(unsigned)__float_as_int(depth) << 32) | (unsigned)g;
  const bool exact = splat && (flags & EG_FLAG_TIGHT_TILES);  // must mirror the counting pass exactly
  // the counting pass left the exact hits of small boxes as a bit mask (all ones = "re-test")
  const unsigned mask = (exact && tile_mask && g < N) ? tile_mask[g] : 0xffffffffu;
  const bool use_mask = exact && mask != 0xffffffffu;
  const int bw = x1 - x0;
#define EG_TILE_OK(tx, ty)                                                           \
  (!exact || (use_mask ? ((mask >> (((ty)-y0) * bw + ((tx)-x0))) & 1u) != 0u         \
                       : splat_hits_tile(x, y, s0.z, s0.w, s1.x, s1.y, tx, ty)))
  if (!LDS) {
    for (int ty = y0; ty < y1; ++ty)
      for (int tx = x0; tx < x1; ++tx) {
        if (!EG_TILE_OK(tx, ty)) continue;
        const int t = ty * tw + tx;
        const long long idx = (long long)offsets[t] + (atomicSub(&cursor[t], 1) - 1);
        if (idx < capacity) keys[idx] = key;
      }
    return;
  }
  for (int ty = y0; ty < y1; ++ty)
    for (int tx = x0; tx < x1; ++tx)
      if (EG_TILE_OK(tx, ty)) atomicAdd(&s_hist[ty * tw + tx], 1);
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += kBinThreads) {
    const int c = s_hist[t];
    if (c) {
      s_base[t] = offsets[t] + (atomicSub(&cursor[t], c) - c);  // slots [base, base + c)
      s_hist[t] = 0;
    }
  }
  __syncthreads();
  for (int ty = y0; ty < y1; ++ty)
    for (int tx = x0; tx < x1; ++tx) {
      if (!EG_TILE_OK(tx, ty)) continue;
      const int t = ty * tw + tx;
      const long long idx = (long long)s_base[t] + atomicAdd(&s_hist[t], 1);
      if (idx < capacity) keys[idx] = key;
    }
#undef EG_TILE_OK
}

// ---------------------------------------------------------------------------------------------
// segmented sort: one workgroup per tile on unique 64-bit keys in LDS.
// Two variants are launched back to back; each tile is handled by exactly one of them:
//   small: 256 threads,  4096 keys (34 KiB LDS)   -- tiles with n <= 4096 (16 keys per thread in registers)
//   large: 1024 threads, 16384 keys (136 KiB LDS) -- tiles with n > 4096; bucket+rank / in-LDS bitonic network
//          up to 16384 keys, hybrid global/LDS network beyond (any n, slow)

// all compare-exchanges of bitonic stages k = k_lo .. k_hi restricted to strides < P, on s[0..P)
// (ascending-only network: first substage of each k mirrors inside the k-block, then half-cleaners)
template <int THREADS>
__device__ __forceinline__ void bitonic_lds(unsigned long long *s, int P, int k_lo, int k_hi, int tid) {
  // every size is a power of two: index math by shifts and masks (a runtime integer division costs
  // more than the compare-exchange it addresses)
  for (int k = k_lo; k <= k_hi; k <<= 1) {
    if (k <= P) {
      const int lhk = 31 - __clz(k) - 1;  // log2(k/2)
      const int mk = (1 << lhk) - 1;
      for (int i = tid; i < (P >> 1); i += THREADS) {
        const int blk = i >> lhk, off = i & mk;
        const int base = blk << (lhk + 1);
        const int lo = base + off, hi = base + k - 1 - off;
        unsigned long long a = s[lo], b = s[hi];
        if (a > b) { s[lo] = b; s[hi] = a; }
      }
      __syncthreads();
    }
    for (int j = min(k >> 2, P >> 1); j >= 1; j >>= 1) {
      const int lj = 31 - __clz(j);
      for (int i = tid; i < (P >> 1); i += THREADS) {
        const int lo = ((i >> lj) << (lj + 1)) + (i & (j - 1)), hi = lo + j;
        unsigned long long a = s[lo], b = s[hi];
        if (a > b) { s[lo] = b; s[hi] = a; }
      }
      __syncthreads();
    }
  }
}

// Bucket + rank sort (the fast path).  Keys inside one tile are spread over a narrow depth range, so
// a monotone map depth -> bucket (uniform over the tile's own [min, max] depth bits, THREADS buckets)
// followed by an exact rank inside each small bucket sorts the segment in O(n) LDS operations instead
// of the bitonic network's O(n log^2 n) barrier-separated stages:
//   load keys -> min/max -> LDS histogram -> block scan -> scatter by bucket -> rank inside bucket
// Keys are unique 64-bit values, so the rank (number of smaller keys in the bucket) is the final
// position: the result equals the stable (tile, depth) sort bit for bit.  A bucket that is too full
// (many equal depths) falls back to the bitonic network for that tile.
template <int THREADS>
__device__ __forceinline__ unsigned block_reduce_u32(unsigned v, bool take_max, unsigned *wave_tmp) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int NW = THREADS / 64;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned o = __shfl_xor(v, d, 64);
    v = take_max ? max(v, o) : min(v, o);
  }
  if (lane == 0) wave_tmp[wv] = v;
  __syncthreads();
  unsigned r = wave_tmp[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) r = take_max ? max(r, wave_tmp[w]) : min(r, wave_tmp[w]);
  __syncthreads();
  return r;
}

constexpr int kMaxBucketFill = 96;  // beyond this a bucket's rank pass degenerates: use the network
#ifndef EG_SORT_BM
#define EG_SORT_BM 2  // (round 5, final kernel, same box: 7.81 / 11.45 / 31.07 -> 7.59 / 11.22 / 30.14 us at configs 1 / 2 / 3; 4: config 3 36.3)
#endif
#ifndef EG_SORT_RANK_G
#define EG_SORT_RANK_G 1
#endif
constexpr int kSortBM = EG_SORT_BM;          // buckets per thread of the bucket + rank sort
constexpr int kSortRankG = EG_SORT_RANK_G;   // keys a thread ranks side by side (their LDS reads in flight together)

// Segmented layout (eg_project_emit): tile t owns keys[t * seg_cap ...), its population sits in
// cursor[t] and the first of its items (128-Gaussian slices) in item_first[t].  The small variant
// (always launched, one workgroup per tile) writes the per-tile tables exactly once -- key range, item
// range, item -> tile map -- and returns the cursor to zero; the large variant, launched after it,
// reads the ranges.  cursor == nullptr: classic layout, ranges come from `offsets`.
struct SegTable {
  int *cursor;
  int seg_cap;
  int *tile_start, *tile_end;  // [T]
  int *item_first;             // [T]: from the scan in eg_project_emit -- or, with `total`, WRITTEN here
  int *item_end;               // [T]
  int *item_tile;              // [max_items]
  int max_items;
  // != nullptr: "prefix here" (the training step, T <= 2048).  The projection kernel then skips its serial tail
  // (ticket, last workgroup, scan: ~3 us on the critical path of every step); every tile's workgroup instead sums
  // the item counts of the tiles before it from the cursors -- which therefore stay untouched during this kernel:
  // the compositing kernel returns them to zero -- and the last tile's workgroup leaves the totals
  // [4]: M, sticky overflow flag, items, largest tile population.
  int *total;
  int *total_flag = nullptr;   // the view's [4] totals for a kernel that does not form them (`total` == nullptr): only [1], the sticky overflow flag, is raised
  // optional output for the wave-autonomous forward (composite_wave.hip): one 16-byte record per item
  // {tile, slice | slices << 16, rec_tag, end of the TILE's keys}   (the slice's first key is tile * seg_cap + 128 slice)
  int4 *item_rec;
  // with `total` and `item_rec`: the records are laid out in DISPATCH order, FRONT SLICES FIRST -- slices 0..3 of every
  // tile (tile by tile, in the order of the item numbering), then the deeper slices of the tiles that have them --
  // while item_first / item_end / item_tile (and the hand-over storage the forward addresses through
  // item_first) keep the contiguous per-tile numbering.  The forward takes one workgroup per record in record order:
  // the front slices of all tiles, which are alive and carry the look-backs and the exact stops, start early, and the
  // deep slices of the heavy tiles -- mostly behind every pixel's stop -- fill the tail, by which time the granules
  // they look back on are there.  (Tile-major order put the live slices of the 6-9-slice tiles at the end of the first
  // round of workgroups, behind the heavy tiles' dead ones: they were the launch's tail.  An exactly slice-major order
  // measured -3.2 us in the forward but needs a histogram of the slice counts in every sort workgroup: +4.7 us
  // there; the two classes cost one more running sum.)
  // A slice still follows all slices in front of it: the look-back's dispatch-order guarantee holds.
  int slice_major = 0;  // the class boundary (slices), 0 = off
  // (round 4 tried a third class -- the items of the single-slice tiles behind the deep slices, so that the launch's tail
  // is made of short-lived waves: same-box A/B at config 2, forward 33.4 -> 32.6 us and sort 11.2 -> 11.0 WITHOUT it on the
  // trained-like scene, 0.7 us the other way on the initial-opacity one.  Removed.)
  // without `total` (tile grids above 2048 tiles), optional: the projection's scan of min(items, EG_FRONT_LARGE) over the
  // tiles, [T + 1] with the total at [T] (EG_FLAG_FRONT_PREFIX).  The records are then written in TWO classes -- slices
  // [0, 9) of every tile, tile by tile, then the deeper slices -- instead of in item order: a tile's deep slices used to be
  // dispatched right behind its front ones and sat waiting for their anchor's inclusive granule, half of the forward's
  // wave slots at 500 k Gaussians; dispatched after every tile's front they find the dead word set and leave at once.
  const int *item_front = nullptr;
  // round 6, without `total`, xcd_shift > 0 (and item_front, total_flag): the XCD-aware placement on grids above 2048 tiles --
  // tiles dealt to the XCDs in bands of 2^xcd_shift tile ROWS, xcd = (ty >> xcd_shift) & 7, positions from the two prefixes the
  // projection's scan leaves anyway (see the kernel); class boundary EG_FRONT_LARGE; every tile has a record (no skip_empty)
  int middle_out = 0;   // workgroup -> tile assignment of the small sort variant (see the kernel)
  // Round 5.  Every record carries the CALL TAG of the forward that will read it in word 2 (the forward validates a
  // record by its tag: the table may have holes, and a record of an earlier call is not mistaken for this call's).
  unsigned rec_tag = 0;
  // with `total`, xcd_shift > 0: XCD-AWARE record placement.  Workgroup b of a launch runs on XCD b % 8 (measured:
  // profiles/r05_sort_ab.txt); the tiles are dealt to the eight XCDs in square blocks of 2^xcd_shift tiles,
  // xcd = (block_x + 3 block_y) % 8, and the records of XCD x's tiles are written at indices 8 k + x, k counting
  // through THAT XCD's list -- front slices [0, slice_major) of its tiles first, tile by tile, then their deeper
  // slices.  All slices of a tile then run on one XCD (their hand-over stays inside one L2) and the tiles that share
  // Gaussians share an L2 (the forward's record gather filled every one of the eight L2s with the whole record array).
  // The lists differ in length: the table has holes at the end of the shorter ones (no record carries the call's tag
  // there) and spans 8 x the longest list -- beyond max_items the sticky overflow flag goes up.
  int xcd_shift = 0;
  int tw = 0;          // tiles per image row (xcd_shift > 0 / skip_empty)
  // Round 5, with `total` and `item_rec` when the caller's forward is the wave-autonomous one with the fused loss: an EMPTY
  // tile gets no record.  57 % of the 512 x 512 grid's tiles hold nothing at the reference's sizes; their forward workgroups
  // -- a fifth of the launch's waves -- only added the background's loss term sum_p w_p |gt_p| and wrote records the footprint
  // backward never reads (no Gaussian's footprint reaches an empty tile: the exact tile test is conservative).  The tile's
  // sort workgroup, which has nothing to sort, adds that term instead (to the forward's 64 partial sums) and leaves before the prefix scans and the barrier; the item numbering still counts the tile's one empty item
  // (nothing else changes), its table entries are not written.  The LAST tile keeps the full path (it leaves the totals).
  const float *gt = nullptr, *wmap = nullptr;  // [H,W] of the view (batched: Batch::gt / wmap)
  float *loss_part = nullptr;                  // the forward's 64 partial loss sums (batched: + view * Batch::ws_bytes)
  int width = 0, height = 0;
  int skip_empty = 0;
  float inv_tw = 0.f;  // 1 / tw
#ifdef EG_SORT_PROF
  // development builds (-DEG_SORT_PROF): [T][12] per-workgroup phase record of the small variant's last launch --
  // {wall clock at start / end (100 MHz), n, shader-clock ticks of: loads, range barrier + prefix, histogram, scan,
  // scatter, rank, store drain} (tools/sort_prof.py)
  unsigned long long *prof = nullptr;
#endif
};
#ifndef EG_XCD_SHIFT_DEFAULT
#define EG_XCD_SHIFT_DEFAULT 1
#endif
constexpr int kXcdMinTiles = 512;  // XCD-aware placement on grids of 512 .. 2048 tiles (the reference's 512 x 512 images: 1024)
constexpr int kXcdShiftDefault = EG_XCD_SHIFT_DEFAULT;  // XCD-aware record placement: tiles per block side = 2^shift (0 = off)
#ifndef EG_SORT_GRID_DIV
#define EG_SORT_GRID_DIV 1
#endif
constexpr int kSortGridDiv = EG_SORT_GRID_DIV;  // tiles per workgroup of the small sort variant (launch_tile_sort)
constexpr int kFrontDefault = 4;  // class boundary of the dispatch order (slices); SegTable::slice_major carries it

// THREADS = number of buckets; CAP = keys per buffer (two buffers).  n_lo < n handled here.
template <int THREADS, int CAP, bool LARGE, bool PREFIX3 = false>
// (PREFIX3, round 6: "prefix here" on grids of 2049 .. 2560 tiles -- the reference's native 800 x 800 is 2500 -- takes a third batch
// of cursor loads; an instantiation of its own: the branch alone cost the 1024-tile launches of configs 1 / 2 0.3-0.4 us)
// (512-thread variant: two workgroups per CU need 4 waves per SIMD, i.e. at most 128 VGPRs; the 256-thread variant at
// 128 VGPRs -- three dwords spilled -- fits FOUR workgroups per CU instead of three: config 3's 7500 tiles 36.9 -> 29.4 us.
// Pushing either further -- 6 or 8 waves per SIMD -- spills the keys and costs 5-12 us: measured)
#ifndef EG_SORT_WAVES
#define EG_SORT_WAVES 4
#endif
#ifndef EG_SORT_WAVES_256
#define EG_SORT_WAVES_256 4
#endif
__global__ void __launch_bounds__(THREADS, THREADS == 512 ? EG_SORT_WAVES : (THREADS == 256 ? EG_SORT_WAVES_256 : 1))
tile_sort_kernel(unsigned long long *__restrict__ keys, const int *__restrict__ offsets, int T,
                 long long capacity, int small_cap, int *__restrict__ flatten_ids,
                 long long *__restrict__ isect_ids, const SegTable seg_, const Batch bt) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long s[];
  // view of this workgroup (batched step: blockIdx.y over [C, ...] work buffers; segmented layout only)
  SegTable seg = seg_;
  {
    const int bv = blockIdx.y;
    keys += bv * bt.keys; flatten_ids += bv * bt.keys;
    if (seg.cursor) {
      seg.cursor += bv * bt.tiles; seg.tile_start += bv * bt.tiles; seg.tile_end += bv * bt.tiles;
      seg.item_first += bv * bt.tiles; seg.item_end += bv * bt.tiles; seg.item_tile += bv * bt.items;
      if (seg.total) seg.total += 4 * bv;
      if (seg.item_rec) seg.item_rec += bv * bt.items;
      if (seg.skip_empty && gridDim.y > 1) {
        seg.gt = bt.gt[bv]; seg.wmap = bt.wmap[bv];
        seg.loss_part = (float *)((char *)seg.loss_part + bv * bt.ws_bytes);
      }
    }
  }
  // NB = kSortBM buckets per thread: the rank pass costs a key one dependent LDS read per key of its bucket, so the
  // fullest tile's workgroup -- which the kernel lasts as long as -- is shortened by thinner buckets
  constexpr int BM = LARGE ? 1 : kSortBM;  // (the large variant is at the LDS limit)
  constexpr int NB = THREADS * BM;
  unsigned long long *kout = s;      // [CAP] keys scattered by bucket (the fast path keeps its input in registers)
  int *hist = (int *)(s + CAP);      // [NB] counts -> exclusive starts
  int *cursor = hist + NB;           // [NB]
  __shared__ unsigned wave_tmp[32], wave_tmp2[32];  // two hops per tile, never the same array twice in a row

  const int tid = threadIdx.x;
#ifdef EG_SORT_PROF
  // (constant indices only: the record stays in registers; no wait of its own -- a tick reads the clock where the code
  // has just waited anyway, behind a barrier or a reduction over loaded values)
  unsigned prof_t[7] = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
  const unsigned long long prof_wall0 = __builtin_amdgcn_s_memrealtime();
  long long prof_prev = (long long)__builtin_readcyclecounter();
#define EG_SP_TICK(k_)                                                           \
  do {                                                                           \
    if (!LARGE) {                                                                \
      const long long now_ = (long long)__builtin_readcyclecounter();            \
      prof_t[k_] = (unsigned)(now_ - prof_prev);                                 \
      prof_prev = now_;                                                          \
    }                                                                            \
  } while (0)
#define EG_SP_DONE(n_)                                                           \
  do {                                                                           \
    if (!LARGE && seg.prof && tid == 0 && blockIdx.y == 0) {                     \
      unsigned long long *r_ = seg.prof + (size_t)tile * 12;                     \
      r_[0] = prof_wall0; r_[1] = __builtin_amdgcn_s_memrealtime(); r_[2] = (unsigned long long)(n_); \
      r_[3] = prof_t[0]; r_[4] = prof_t[1]; r_[5] = prof_t[2]; r_[6] = prof_t[3]; r_[7] = prof_t[4]; r_[8] = prof_t[5]; r_[9] = prof_t[6]; \
      r_[10] = (unsigned long long)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15); r_[11] = blockIdx.x; /* XCC_ID, workgroup */ \
    }                                                                            \
  } while (0)
#else
#define EG_SP_TICK(k_) do {} while (0)
#define EG_SP_DONE(n_) do {} while (0)
#endif
  // The large variant runs a small grid (<= 256 workgroups of 136 KiB LDS) over the tiles that outgrew the small one.
  // Round 4: every workgroup first FINDS them -- all its threads look at the T ranges at once and collect the oversized
  // tiles in LDS (sorted by tile index, so that all workgroups hold the same list) -- and then takes entries blockIdx.x,
  // blockIdx.x + gridDim.x, ... of that list; striding over ALL tiles with a barrier and two dependent loads per tile to
  // find a handful cost 14 us at 500 k Gaussians @1200x680 (3225 tiles, a dozen above 4096 keys).
  constexpr int kBigList = 512;
  __shared__ int s_big[LARGE ? kBigList : 1], s_big_sorted[LARGE ? kBigList : 1];
  __shared__ int s_nbig;
  int n_loop = T;
  bool by_list = false;
  if (LARGE) {
    if (tid == 0) s_nbig = 0;
    __syncthreads();
    for (int t = tid; t < T; t += THREADS) {
      long long a_, b_;
      if (seg.cursor) { a_ = seg.tile_start[t]; b_ = seg.tile_end[t]; }
      else { a_ = offsets[t]; b_ = offsets[t + 1]; if (b_ > capacity) b_ = capacity; }
      if (b_ - a_ > small_cap) {
        const int k = atomicAdd(&s_nbig, 1);
        if (k < kBigList) s_big[k] = t;
      }
    }
    __syncthreads();
    const int nb = s_nbig;
    if (nb <= kBigList) {  // (else: more oversized tiles than the list holds -- stride over all tiles as before)
      for (int i = tid; i < nb; i += THREADS) {
        const int me = s_big[i];
        int rank = 0;
        for (int q = 0; q < nb; ++q) rank += s_big[q] < me ? 1 : 0;
        s_big_sorted[rank] = me;
      }
      __syncthreads();
      by_list = true;
      n_loop = nb;
    }
  }
  for (int it = blockIdx.x; it < n_loop; it += gridDim.x) {
  const int wg = by_list ? s_big_sorted[it] : it;
  // Workgroup -> tile, MIDDLE OUT over the row-major tile index (the small variant: one tile per workgroup, dispatched in
  // index order, two or three rounds of them): the tiles of the image's middle rows -- where a centred object puts its
  // thousands of keys -- start in the first round and the near-empty border tiles make up the last one, instead of the
  // image's lower half with its share of heavy tiles.  Any assignment is correct.
  const int tile = (!LARGE && seg.middle_out) ? ((wg & 1) ? T / 2 - (wg + 1) / 2 : T / 2 + wg / 2) : wg;
  __syncthreads();
  long long start, end;
  __shared__ int s_pre[5][THREADS / 64];
  bool prefix_pending = false;  // (uniform) the tile prefix still has to be finished: see SegTable::total
  int pop_here = 0;
  // after a barrier: every thread sums the waves' partials; thread 0 writes the tile's table entries (and the
  // totals of the view, if this is the last tile), the first threads the item -> tile map
  auto finish_prefix = [&](int kept_) {
    prefix_pending = false;
    if (tid >= 64) return;  // (the first wave writes the tables and the records: a tile has a few dozen items at most)
    int isum = 0, k0 = 0, l0k1 = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) { isum += s_pre[0][w]; k0 += s_pre[1][w]; l0k1 += s_pre[4][w]; }
    // of the tiles of THIS tile's XCD (all tiles without the XCD-aware placement): k0 = front-class items in front of this
    // tile; l0k1 = front-class items in all + deep-class items in front of this tile.  (Two full ints: round 5 packed them
    // into 16-bit halves of one, and a list of >= 32768 items turned `l0k1` negative -- a negative dispatch index still
    // passed `disp < max_items`.)
    const int first_ = min(isum, seg.max_items);
    const int items_ = min(max(1, (kept_ + 127) >> 7), max(0, seg.max_items - first_));
    // dispatch index of slice i: class 0 = slices [0, slice_major) of the list's tiles, class 1 = the rest; inside a class
    // tile by tile in the item numbering's order.  (Every workgroup forms the same sums from the same cursors.)
    const int myx = seg.xcd_shift > 0 ? xcd_of_tile(tile, seg.tw, seg.inv_tw, seg.xcd_shift) : 0;
    bool rec_over = false;
    for (int i = tid; i < items_; i += 64) {
      seg.item_tile[first_ + i] = tile;
      if (seg.item_rec) {
        const int pos = i < seg.slice_major ? k0 + i : l0k1 + (i - seg.slice_major);
        const int disp = seg.xcd_shift > 0 ? 8 * pos + myx : pos;
        if (disp >= 0 && disp < seg.max_items)
          seg.item_rec[disp] = make_int4(tile, i | (items_ << 16), (int)seg.rec_tag, tile * seg.seg_cap + kept_);
        else
          rec_over = true;  // (the longest XCD list does not fit the table: the caller grows it and replays)
      }
    }
    if (rec_over) seg.total[1] = 1;
    if (tid == 0) {
      seg.item_first[tile] = first_;
      seg.tile_start[tile] = tile * seg.seg_cap;
      seg.tile_end[tile] = tile * seg.seg_cap + kept_;
      seg.item_end[tile] = first_ + items_;
      if (tile == T - 1) {  // (the last tile's workgroup has summed the view's totals as well)
        int msum = 0, cmax = 0;
#pragma unroll
        for (int w = 0; w < THREADS / 64; ++w) { msum += s_pre[2][w]; cmax = max(cmax, s_pre[3][w]); }
        const int itot = isum + max(1, (kept_ + 127) >> 7);  // (the tiles in front + this one: all of them)
        seg.total[0] = msum;
        if (cmax > seg.seg_cap || itot > seg.max_items) seg.total[1] = 1;  // sticky: only the host clears it
        seg.total[2] = min(itot, seg.max_items);
        seg.total[3] = cmax;
      }
    }
    prefix_pending = false;
  };
  // Segmented layout: the tile's keys sit at a FIXED address, so the first key of every thread is requested before
  // the population is known (one round trip to memory instead of two: most tiles hold fewer keys than the workgroup
  // has threads); a slot beyond the population holds a stale key of an earlier step and is dropped below
  unsigned long long spec0 = ~0ull;
  if (!LARGE && seg.cursor && tid < seg.seg_cap) spec0 = keys[(size_t)tile * seg.seg_cap + tid];
  if (seg.cursor) {
    if (!LARGE) {
      int kept, first, items;
      if (seg.total) {
        // The populations of the tiles before this one are REQUESTED by all threads before anything is waited for
        // (they travel with the tile's own cursor and first keys); per-wave partial sums go to LDS and the prefix
        // is finished right after the FIRST barrier the sort takes anyway (finish_prefix below): no barrier of its own
        // Round 5.  What a workgroup needs of the other tiles is TWO sums -- the items of the tiles in front (the tile's first
        // item) and, over the tiles of its XCD's list (all tiles without the XCD-aware placement), the packed pair {front-class
        // items in front | front-class items in all + deep-class items in front} -- and only the LAST tile's workgroup needs
        // the view's totals (M, the largest population; the item count is its own prefix + its own items).  Round 4 formed
        // five sums in every wave of every workgroup; the per-workgroup phase record (profiles/r05_sort_phases_config2.txt,
        // r05_sort_ab.txt) prices the dependent vector chain in front of the first barrier at ~500 cycles per tile and lane and
        // ~240 per DPP scan with four waves per SIMD doing the same -- the launch's floor, since 57 % of its workgroups do
        // nothing else.  The first batch of populations is asked for together with the tile's own: buffer loads on a
        // descriptor of the T cursors, a tile beyond T reads 0 without a clamp.
        pop_here = seg.cursor[tile];
        kept = min(pop_here, seg.seg_cap);
        const __amdgpu_buffer_rsrc_t cur_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)seg.cursor, 0, T * 4, 0x00020000);
        constexpr int NE = kPrefixBatchTiles / THREADS;  // tiles per lane and batch: a batch covers 1024 tiles
        int pv0[NE];
#pragma unroll
        for (int j = 0; j < NE; ++j) pv0[j] = (int)__builtin_amdgcn_raw_buffer_load_b32(cur_rsrc, (tid + j * THREADS) * 4, 0, 0);
        // (skip_empty: the background's loss term of this tile; thread t of the first four waves takes pixel (t >> 4, t & 15)
        // of the tile: four half lines per wave and array.  The 512-thread variant -- two rounds of workgroups, the empty
        // border tiles last -- asks for it WITH everything else and uses it only if the tile turns out empty; the
        // 256-thread variant, whose workgroups all start together, asks once it knows: requested by every tile's
        // workgroup the term cost that launch 0.6 us at config 1, profiles/r05_skip_empty_ab.txt)
        constexpr bool kBgEarly = THREADS >= 512;
        auto bg_term = [&]() -> float {
          float l = 0.f;
          if (tid < kTilePix) {
            const int ty_ = (int)(((float)tile + 0.5f) * seg.inv_tw), tx_ = tile - __mul24(ty_, seg.tw);
            const int pi = ty_ * kTile + (tid >> 4), pj = tx_ * kTile + (tid & 15);
            if (pi < seg.height && pj < seg.width) {
              const int pp = pi * seg.width + pj;
              l = seg.wmap[pp] * fabsf(seg.gt[pp]);  // w_p |clamp(1 - T_final) - gt_p| with T_final = 1
            }
          }
          return l;
        };
        float l_bg = 0.f;
        if (kBgEarly && seg.skip_empty) l_bg = bg_term();
        if (seg.skip_empty && pop_here == 0 && tile != T - 1) {  // (uniform) nothing to sort, no record, no table entry
          if (!kBgEarly) l_bg = bg_term();
          if (tid < kTilePix) {
            l_bg = wave_sum_dpp_f(l_bg);  // (total in lane 63)
            if ((tid & 63) == 63 && l_bg != 0.f) unsafeAtomicAdd(&seg.loss_part[(tile * 4 + (tid >> 6)) & 63], l_bg);
          }
          EG_SP_TICK(0); EG_SP_DONE(0);
          continue;
        }
        {
          int isum = 0, front = 0, deep = 0, msum = 0, cmax = 0;
          const bool last_tile = tile == T - 1;  // (uniform)
          const int myx = seg.xcd_shift > 0 ? xcd_of_tile(tile, seg.tw, seg.inv_tw, seg.xcd_shift) : 0;
          auto sums = [&](const int (&pv)[NE], int j0) {
#pragma unroll
            for (int j = 0; j < NE; ++j) {
              const int tj = tid + (j0 + j) * THREADS;
              const bool ok = tj < T, before = tj < tile;
              const int kk = min(pv[j], seg.seg_cap), it = ok ? max(1, (kk + 127) >> 7) : 0;
              isum += before ? it : 0;
              // (records: an empty tile other than the last one has none with skip_empty; the numbering above still counts it)
              const bool mine = (seg.xcd_shift == 0 || xcd_of_tile(tj, seg.tw, seg.inv_tw, seg.xcd_shift) == myx) &&
                                !(seg.skip_empty && kk == 0 && tj != T - 1);
              const int itf = mine ? min(it, seg.slice_major) : 0;
              front += before ? itf : 0;
              deep += itf + ((mine && before) ? it - itf : 0);
              if (last_tile) { msum += kk; cmax = max(cmax, pv[j]); }
            }
          };
          sums(pv0, 0);
          if (T > kPrefixBatchTiles) {  // (uniform) a grid above 1024 tiles takes a second round trip
            int pv1[NE];
#pragma unroll
            for (int j = 0; j < NE; ++j)
              pv1[j] = (int)__builtin_amdgcn_raw_buffer_load_b32(cur_rsrc, (tid + (NE + j) * THREADS) * 4, 0, 0);
            sums(pv1, NE);
          }
          // (round 6: ... and the reference's native 800 x 800 -- 2500 tiles -- a third.  Requesting all three batches before the
          // first is summed costs the 256-thread variant six spilled registers: 15.9 against 14.5 us at 800 x 800)
          if (PREFIX3 && T > 2 * kPrefixBatchTiles) {
            int pv2[NE];
#pragma unroll
            for (int j = 0; j < NE; ++j)
              pv2[j] = (int)__builtin_amdgcn_raw_buffer_load_b32(cur_rsrc, (tid + (2 * NE + j) * THREADS) * 4, 0, 0);
            sums(pv2, 2 * NE);
          }
          // (DPP scans: the totals land in lane 63)
          isum = wave_scan_dpp(isum, 0, OpAdd());
          front = wave_scan_dpp(front, 0, OpAdd());
          deep = wave_scan_dpp(deep, 0, OpAdd());
          if (last_tile) {
            msum = wave_scan_dpp(msum, 0, OpAdd());
            cmax = wave_scan_dpp(cmax, 0, OpMaxI());
          }
          if ((tid & 63) == 63) {
            s_pre[0][tid >> 6] = isum; s_pre[1][tid >> 6] = front; s_pre[4][tid >> 6] = deep;
            if (last_tile) { s_pre[2][tid >> 6] = msum; s_pre[3][tid >> 6] = cmax; }
          }
        }
        prefix_pending = true;
        first = items = 0;
      } else {
        kept = min(seg.cursor[tile], seg.seg_cap);  // every thread reads it; reset after the barrier
        first = seg.item_first[tile];
        items = min(max(1, (kept + 127) >> 7), max(0, seg.max_items - first));
        // Round 6: XCD-aware record placement above 2048 tiles, WITHOUT another scan.  The tiles are dealt to the XCDs in
        // BANDS of 2^xcd_shift tile rows, xcd = (ty >> shift) & 7: a band is a contiguous run of tile indices, a list is
        // every eighth band, and the projection's scan already left the two exclusive prefixes over ALL tiles -- F (front-class
        // items: item_front) and A (items: item_first).  Tile t's place in its list follows from their values at t and at the
        // boundaries of the list's bands -- a dozen loads that travel with the tile's own, summed by the first wave:
        //   k0   = sum over the list's bands in front of (F(end) - F(start)) + F(t) - F(start of t's band)
        //   l0k1 = the list's front total + the same sums of G = A - F (deep-class items)
        // (2 x 2 blocks dealt round-robin, as on the small grids, need a prefix per list: sixteen more sums in the projection
        // kernel's serial tail measured +1.7 / +4.3 us per step at 800 x 800 / 1600 x 1200 -- profiles/r06_xcd_large_ab.txt.)
        int k0 = 0, l0k1 = 0, myx = -1;
        if (seg.xcd_shift > 0 && seg.item_front && seg.item_rec && seg.total_flag) {
          const int ftot = seg.item_front[T];
          if (ftot >= 0) {  // (< 0: the view overflowed the item capacity -- item order, see below)
            const int band_tiles = seg.tw << seg.xcd_shift, nb = (T + band_tiles - 1) / band_tiles;
            const int b = (int)(((float)tile + 0.5f) * seg.inv_tw) >> seg.xcd_shift;
            myx = b & 7;
            if (tid < 64) {
              int fb = 0, gb = 0, ft = 0;
              for (int bj = myx + 8 * tid; bj < nb; bj += 8 * 64) {
                const int s0 = bj * band_tiles, e0 = min(T, s0 + band_tiles);
                const int Fs = seg.item_front[s0], Fe = seg.item_front[e0];
                const int As = seg.item_first[s0], Ae = e0 < T ? seg.item_first[e0] : seg.total_flag[2];
                const int df = Fe - Fs, dg = (Ae - As) - df;
                ft += df;
                if (bj < b) { fb += df; gb += dg; }
              }
              fb = wave_scan_dpp(fb, 0, OpAdd()); gb = wave_scan_dpp(gb, 0, OpAdd()); ft = wave_scan_dpp(ft, 0, OpAdd());
              if (tid == 63) {
                const int s0 = b * band_tiles;
                const int Ft = seg.item_front[tile], Fs = seg.item_front[s0], As = seg.item_first[s0];
                s_pre[0][0] = fb + (Ft - Fs);
                s_pre[1][0] = ft + gb + ((first - Ft) - (As - Fs));
              }
            }
          }
        }
        __syncthreads();
        if (myx >= 0) { k0 = s_pre[0][0]; l0k1 = s_pre[1][0]; }
        if (tid == 0) {
          seg.cursor[tile] = 0;  // ready for the next step
          seg.tile_start[tile] = tile * seg.seg_cap;
          seg.tile_end[tile] = tile * seg.seg_cap + kept;
          seg.item_end[tile] = first + items;
        }
        bool rec_over = false;
        for (int i = tid; i < items; i += THREADS) {
          seg.item_tile[first + i] = tile;
          if (seg.item_rec) {
            int disp = first + i;
            if (myx >= 0) {
              disp = 8 * (i < EG_FRONT_LARGE ? k0 + i : l0k1 + (i - EG_FRONT_LARGE)) + myx;
              if (disp >= seg.max_items) rec_over = true;  // (the longest list does not fit the table: grow + replay)
            } else if (seg.item_front) {
              // (ftot < 0: the view overflowed the item capacity and the projection's scan said so -- item order, which
              // has no holes when tiles are truncated; the caller grows the buffers and replays)
              const int fpre = seg.item_front[tile], ftot = seg.item_front[T];
              if (ftot >= 0) disp = i < EG_FRONT_LARGE ? fpre + i : ftot + (first - fpre) + (i - EG_FRONT_LARGE);
            }
            if (disp >= 0 && disp < seg.max_items)
              seg.item_rec[disp] = make_int4(tile, i | (items << 16), (int)seg.rec_tag, tile * seg.seg_cap + kept);
          }
        }
        if (rec_over) seg.total_flag[1] = 1;  // sticky: only the host clears it
      }
      start = (long long)tile * seg.seg_cap;
      end = start + kept;
    } else {  // launched after the small variant: the ranges are in place
      start = seg.tile_start[tile];
      end = seg.tile_end[tile];
    }
  } else {
    start = offsets[tile];
    end = offsets[tile + 1];
    if (end > capacity) end = capacity;
  }
  const int n = (int)(end - start);
  if (n <= 0 || (LARGE ? (n <= small_cap) : (n > small_cap))) {  // empty, or the other variant owns this tile
    if (prefix_pending) { __syncthreads(); finish_prefix(n); }
    EG_SP_TICK(0); EG_SP_DONE(n);
    continue;
  }
  unsigned long long *segk = keys + start;
  const unsigned long long kInf = ~0ull;

  if (n <= CAP) {
    // ---- bucket + rank.  A thread keeps its (up to CAP / THREADS) keys in registers across the three passes
    // (range, histogram, scatter), so LDS only holds the scattered copy: twice the keys per byte of LDS
    constexpr int KPT = CAP / THREADS;
    unsigned long long kr[KPT];
    unsigned dmin = 0xffffffffu, dmax = 0u;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int i = tid + j * THREADS;
      kr[j] = kInf;
      if (j * THREADS < n && i < n) {
        kr[j] = (!LARGE && j == 0 && seg.cursor) ? spec0 : segk[i];
        const unsigned d = (unsigned)(kr[j] >> 32);
        dmin = min(dmin, d);
        dmax = max(dmax, d);
      }
    }
#pragma unroll
    for (int m = 0; m < BM; ++m) { hist[tid + m * THREADS] = 0; cursor[tid + m * THREADS] = 0; }
    // depth range of the tile: both reductions share one LDS hop and one barrier
    constexpr int NW = THREADS / 64;
    const int lane = tid & 63, wv = tid >> 6;
    dmin = (unsigned)wave_scan_dpp((int)dmin, -1, OpMinU());
    dmax = (unsigned)wave_scan_dpp((int)dmax, 0, OpMaxU());
    EG_SP_TICK(0);  // 0: loads (cursors, keys) + the wave-level reductions
    if (lane == 63) { wave_tmp[wv] = dmin; wave_tmp[16 + wv] = dmax; }
    __syncthreads();
    if (prefix_pending) finish_prefix(n);
    EG_SP_TICK(1);  // 1: range barrier + prefix tables / item records
#pragma unroll
    for (int w = 0; w < NW; ++w) { dmin = min(dmin, wave_tmp[w]); dmax = max(dmax, wave_tmp[16 + w]); }
    const float scale = (float)NB / ((float)(dmax - dmin) + 1.f);
#pragma unroll
    for (int j = 0; j < KPT; ++j)
      if (j * THREADS < n && tid + j * THREADS < n) {
        const unsigned d = (unsigned)(kr[j] >> 32);
        atomicAdd(&hist[min(NB - 1, (int)((float)(d - dmin) * scale))], 1);
      }
    __syncthreads();
    EG_SP_TICK(2);  // 2: histogram
    // exclusive scan of the bucket counts and their maximum, again one hop and one barrier
    // (thread tid owns buckets tid * BM ...: cnt = their sum)
    int cb[BM], cnt = 0, cmx = 0;
#pragma unroll
    for (int m = 0; m < BM; ++m) { cb[m] = hist[tid * BM + m]; cnt += cb[m]; cmx = max(cmx, cb[m]); }
    const int incl = wave_scan_dpp(cnt, 0, OpAdd());
    const int wmax = wave_scan_dpp(cmx, 0, OpMaxI());
    if (lane == 63) { wave_tmp2[wv] = (unsigned)incl; wave_tmp2[16 + wv] = (unsigned)wmax; }
    __syncthreads();
    int pre = 0;
    unsigned fill = 0u;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      pre += (w < wv) ? (int)wave_tmp2[w] : 0;
      fill = max(fill, wave_tmp2[16 + w]);
    }
    if (fill <= (unsigned)kMaxBucketFill) {
      {
        int run = pre + (incl - cnt);
#pragma unroll
        for (int m = 0; m < BM; ++m) { hist[tid * BM + m] = run; run += cb[m]; }
      }
      __syncthreads();
      EG_SP_TICK(3);  // 3: scan
#pragma unroll
      for (int j = 0; j < KPT; ++j)
        if (j * THREADS < n && tid + j * THREADS < n) {
          const unsigned d = (unsigned)(kr[j] >> 32);
          const int bk = min(NB - 1, (int)((float)(d - dmin) * scale));
          kout[hist[bk] + atomicAdd(&cursor[bk], 1)] = kr[j];
        }
      __syncthreads();
      EG_SP_TICK(4);  // 4: scatter
      if (kSortRankG == 1) {
        for (int i = tid; i < n; i += THREADS) {
          const unsigned long long k = kout[i];
          const unsigned d = (unsigned)(k >> 32);
          const int bk = min(NB - 1, (int)((float)(d - dmin) * scale));
          const int b0 = hist[bk], b1 = b0 + cursor[bk];
          int rank = 0;
          for (int q = b0; q < b1; ++q) rank += (kout[q] < k) ? 1 : 0;
          const long long o = start + b0 + rank;
          const int gid = (int)(unsigned)(k & 0xffffffffull);
          flatten_ids[o] = gid;
          if (isect_ids) isect_ids[o] = ((long long)tile << 32) | (long long)(k >> 32);
        }
      } else {
        // kSortRankG keys of a thread step through their buckets side by side: one round trip to LDS per step of all of
        // them instead of one per key and step (the fullest tile's threads rank CAP / THREADS keys each)
        constexpr int G = kSortRankG;
        for (int i0 = tid; i0 < n; i0 += G * THREADS) {
          unsigned long long k[G];
          int b0[G], f[G], rank[G], fmax = 0;
#pragma unroll
          for (int g = 0; g < G; ++g) {
            const int i = min(i0 + g * THREADS, n - 1);
            k[g] = kout[i];
            const unsigned d = (unsigned)(k[g] >> 32);
            const int bk = min(NB - 1, (int)((float)(d - dmin) * scale));
            b0[g] = hist[bk];
            f[g] = (i0 + g * THREADS < n) ? cursor[bk] : 0;
            rank[g] = 0;
            fmax = max(fmax, f[g]);
          }
          for (int q = 0; q < fmax; ++q) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
              const unsigned long long o_ = kout[min(b0[g] + q, n - 1)];
              rank[g] += (q < f[g] && o_ < k[g]) ? 1 : 0;
            }
          }
#pragma unroll
          for (int g = 0; g < G; ++g)
            if (i0 + g * THREADS < n) {
              const long long o = start + b0[g] + rank[g];
              flatten_ids[o] = (int)(unsigned)(k[g] & 0xffffffffull);
              if (isect_ids) isect_ids[o] = ((long long)tile << 32) | (long long)(k[g] >> 32);
            }
        }
      }
#ifdef EG_SORT_PROF
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      EG_SP_TICK(5);  // 5: rank (issue)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      EG_SP_TICK(6);  // 6: the stores' drain
      EG_SP_DONE(n);
#endif
      continue;
    }
    // degenerate depth distribution: bitonic network on the keys (kin := LDS buffer `kout`)
    unsigned long long *kin = kout;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KPT; ++j)
      if (j * THREADS < n && tid + j * THREADS < n) kin[tid + j * THREADS] = kr[j];
    int P = 1;
    while (P < n) P <<= 1;
    for (int i = n + tid; i < P; i += THREADS) kin[i] = kInf;
    __syncthreads();
    bitonic_lds<THREADS>(kin, P, 2, P, tid);
    for (int i = tid; i < n; i += THREADS) {
      const unsigned long long key = kin[i];
      const int gid = (int)(unsigned)(key & 0xffffffffull);
      flatten_ids[start + i] = gid;
      if (isect_ids) isect_ids[start + i] = ((long long)tile << 32) | (long long)(key >> 32);
    }
    continue;
  }

  // ---- oversized segment (n > CAP): hybrid global/LDS bitonic network.  Virtual size P (power of two),
  // indices >= n behave as +inf and never move (the network only ever moves larger keys to higher indices).
  if (prefix_pending) { __syncthreads(); finish_prefix(n); }
  constexpr int BCAP = CAP;
  long long P = BCAP;
  while (P < n) P <<= 1;
  // (a) sort every BCAP chunk completely
  for (long long c0 = 0; c0 < n; c0 += BCAP) {
    for (int i = tid; i < BCAP; i += THREADS) s[i] = (c0 + i < n) ? segk[c0 + i] : kInf;
    __syncthreads();
    bitonic_lds<THREADS>(s, BCAP, 2, BCAP, tid);
    for (int i = tid; i < BCAP; i += THREADS)
      if (c0 + i < n) segk[c0 + i] = s[i];
    __syncthreads();
  }
  // (b) merges for k > BCAP: long strides in global memory, the tail (j <= BCAP/2) in LDS
  for (long long k = 2 * (long long)BCAP; k <= P; k <<= 1) {
    const long long hk = k >> 1;
    for (long long i = tid; i < (P >> 1); i += THREADS) {
      const long long blk = i / hk, off = i - blk * hk;
      const long long lo = blk * k + off, hi = blk * k + k - 1 - off;
      if (hi < n) {
        unsigned long long a = segk[lo], b = segk[hi];
        if (a > b) { segk[lo] = b; segk[hi] = a; }
      }
    }
    __syncthreads();
    for (long long j = k >> 2; j >= BCAP; j >>= 1) {
      for (long long i = tid; i < (P >> 1); i += THREADS) {
        const long long lo = ((i / j) * (j << 1)) + (i % j), hi = lo + j;
        if (hi < n) {
          unsigned long long a = segk[lo], b = segk[hi];
          if (a > b) { segk[lo] = b; segk[hi] = a; }
        }
      }
      __syncthreads();
    }
    for (long long c0 = 0; c0 < n; c0 += BCAP) {
      for (int i = tid; i < BCAP; i += THREADS) s[i] = (c0 + i < n) ? segk[c0 + i] : kInf;
      __syncthreads();
      // only the half-cleaner substages j = BCAP/2 .. 1 (k_lo = k_hi = 2*BCAP > P skips the mirror)
      bitonic_lds<THREADS>(s, BCAP, 2 * BCAP, 2 * BCAP, tid);
      for (int i = tid; i < BCAP; i += THREADS)
        if (c0 + i < n) segk[c0 + i] = s[i];
      __syncthreads();
    }
  }
  for (int i = tid; i < n; i += THREADS) {
    const unsigned long long key = segk[i];
    const int gid = (int)(unsigned)(key & 0xffffffffull);
    flatten_ids[start + i] = gid;
    if (isect_ids) isect_ids[start + i] = ((long long)tile << 32) | (long long)(key >> 32);
  }
  }  // tile loop
}

}  // namespace eg

using namespace eg;

extern "C" int eg_tile_count(const float *means2d, const int32_t *radii, int32_t N, int32_t width, int32_t height,
                             int32_t *tiles_per_gauss, int32_t *tile_counts, eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && width > 0 && height > 0, "bad sizes");
  if (N == 0) return EG_OK;
  EG_REQUIRE(means2d && radii && tile_counts, "null pointer");
  const int T = cdiv(width, kTile) * cdiv(height, kTile);
  if (T <= kMaxLdsTiles)
    tile_count_kernel<true><<<cdiv(N, kBinThreads), kBinThreads, sizeof(int) * T, as_stream(stream)>>>(
        (const float2 *)means2d, radii, N, width, height, tiles_per_gauss, tile_counts);
  else
    tile_count_kernel<false><<<cdiv(N, kBinThreads), kBinThreads, 0, as_stream(stream)>>>(
        (const float2 *)means2d, radii, N, width, height, tiles_per_gauss, tile_counts);
  return check_launch("tile_count");
}

extern "C" int eg_tile_offsets(const int32_t *tile_counts, int32_t T, int64_t capacity, int32_t *offsets,
                               int32_t *item_offsets, int32_t *total, eg_stream_t stream) {
  EG_REQUIRE(T > 0 && tile_counts && offsets, "bad arguments");
  tile_offsets_kernel<<<1, 1024, 0, as_stream(stream)>>>(tile_counts, T, (long long)capacity, offsets, item_offsets,
                                                        total);
  return check_launch("tile_offsets");
}

extern "C" int eg_tile_emit(const float *means2d, const int32_t *radii, const float *depths, const float *splat,
                            uint32_t flags, int32_t N, int32_t width, int32_t height, const int32_t *offsets, int32_t *tile_cursor,
                            int64_t capacity, uint64_t *keys, const uint32_t *tile_mask, eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && width > 0 && height > 0 && capacity >= 0, "bad sizes");
  if (N == 0) return EG_OK;
  EG_REQUIRE(offsets && tile_cursor && (keys || capacity == 0), "null pointer");
  EG_REQUIRE(splat || (means2d && radii && depths), "need splat or (means2d, radii, depths)");
  const int T = cdiv(width, kTile) * cdiv(height, kTile);
  if (2 * T <= kMaxLdsTiles)
    tile_emit_kernel<true><<<cdiv(N, kBinThreads), kBinThreads, sizeof(int) * 2 * T, as_stream(stream)>>>(
        (const float2 *)means2d, radii, depths, (const float4 *)splat, flags, N, width, height, offsets,
        tile_cursor, (long long)capacity, (unsigned long long *)keys, tile_mask);
  else
    tile_emit_kernel<false><<<cdiv(N, kBinThreads), kBinThreads, 0, as_stream(stream)>>>(
        (const float2 *)means2d, radii, depths, (const float4 *)splat, flags, N, width, height, offsets,
        tile_cursor, (long long)capacity, (unsigned long long *)keys, tile_mask);
  return check_launch("tile_emit");
}

static bool g_sort_attr_set = false;

#ifdef EG_SORT_PROF
static unsigned long long *g_sort_prof = nullptr;
static int g_sort_prof_tiles = 0;
#endif

static int launch_tile_sort(uint64_t *keys, const int32_t *offsets, int32_t T, int64_t capacity,
                            int32_t *flatten_ids, int64_t *isect_ids, int32_t max_tile_hint, const SegTable seg_in,
                            eg_stream_t stream, const Batch &bt = Batch{}, int C = 1) {
  SegTable seg = seg_in;
#ifdef EG_SORT_PROF
  if (!g_sort_prof || g_sort_prof_tiles < T) {
    if (g_sort_prof) (void)hipFree(g_sort_prof);
    g_sort_prof_tiles = T;
    (void)hipMalloc((void **)&g_sort_prof, (size_t)T * 12 * sizeof(unsigned long long));
  }
  (void)hipMemsetAsync(g_sort_prof, 0, (size_t)T * 12 * sizeof(unsigned long long), as_stream(stream));
  seg.prof = g_sort_prof;
#endif
  // small: 256 threads / buckets, 4096 keys; large: 1024 threads / buckets, 16384 keys
  constexpr int kSmall = 4096, kLarge = 16384;
  constexpr size_t kLargeLds = kLarge * 8 + 2 * 1024 * 4;
  // workgroup of the small variant: the kernel lasts as long as its fullest tile, so tiles of thousands of keys
  // want 512 threads (config 2: 16.0 -> 13.1 us; 500 k @1200x680: 45 -> 41) -- unless the grid has thousands of
  // small tiles (200 k @1600x1200: 33 -> 44 with 512), where the extra waves cost more than the fullest tile gains
  // (a batch of C views multiplies the grid: config 2 with 4 views per launch sequence 221 -> 278 us with 512)
  bool wide = max_tile_hint > 1536 && (C == 1 ? T <= 4096 : T * C <= 2048);
#ifdef EG_DEV_SWITCHES
  static const int wide_env = getenv("EG_SORT_WIDE") ? atoi(getenv("EG_SORT_WIDE")) : -1;  // (A/B switch)
  if (wide_env >= 0) wide = wide_env != 0;
#endif
  if (!g_sort_attr_set) {
    (void)hipFuncSetAttribute((const void *)tile_sort_kernel<1024, kLarge, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLargeLds);
    g_sort_attr_set = true;
  }
  // max_tile_hint: the largest tile population the caller has seen (0 = unknown).  If even 1.25x that
  // fits the small variant, the launch of the large one (256 workgroups of 1024 threads and 136 KiB of
  // LDS that would all find nothing to do: ~3 us) is skipped and the small variant owns EVERY tile -- a
  // tile that outgrew the hint is then still sorted correctly, by the slower paths of the small variant.
  const bool small_only = max_tile_hint > 0 && (int64_t)max_tile_hint * 5 / 4 <= kSmall;
  // workgroups of the small variant: one per tile -- or (grid_div > 1) one per grid_div tiles, which the kernel's
  // grid-stride loop pairs middle-out rank b with rank b + T / grid_div: a central tile and a border tile.  Round 6 measured
  // 2 ("one workgroup retires a run of empty tiles"): 7.7 -> 10.1 us at config 1, 11.8 -> 12.4 at config 2
  // (profiles/r06_misc_ab.txt) -- the second tile's round trip to memory starts after the first tile's sort; stays 1
  int grid_div = kSortGridDiv;
#ifdef EG_DEV_SWITCHES
  static const int div_env = getenv("EG_SORT_GRID_DIV") ? atoi(getenv("EG_SORT_GRID_DIV")) : 0;  // (A/B switch)
  if (div_env > 0) grid_div = div_env;
#endif
  const int grid_x = cdiv(T, grid_div);
#define EG_SORT_SMALL(TH_, P3_)                                                                                          \
  tile_sort_kernel<TH_, kSmall, false, P3_><<<dim3(grid_x, C), TH_, kSmall * 8 + 2 * TH_ * 4 * kSortBM, as_stream(stream)>>>( \
      (unsigned long long *)keys, offsets, T, (long long)capacity, small_only ? 0x7fffffff : kSmall, flatten_ids,            \
      (long long *)isect_ids, seg, bt)
  const bool prefix3 = seg.total != nullptr && T > 2 * kPrefixBatchTiles;
  if (wide) { if (prefix3) EG_SORT_SMALL(512, true); else EG_SORT_SMALL(512, false); }
  else      { if (prefix3) EG_SORT_SMALL(256, true); else EG_SORT_SMALL(256, false); }
#undef EG_SORT_SMALL
  if (!small_only)
    tile_sort_kernel<1024, kLarge, true><<<dim3(min(T, 256), C), 1024, kLargeLds, as_stream(stream)>>>(
        (unsigned long long *)keys, offsets, T, (long long)capacity, kSmall, flatten_ids, (long long *)isect_ids,
        seg, bt);
  return check_launch("tile_sort");
}

#ifdef EG_SORT_PROF
// development builds only: the per-workgroup phase records of the last small-variant sort launch ([T][12] words)
extern "C" int64_t eg_debug_sort_profile(uint64_t *out, int64_t max_tiles) {
  if (!g_sort_prof || !out) return EG_ERR_ARG;
  const int64_t n = g_sort_prof_tiles < max_tiles ? g_sort_prof_tiles : max_tiles;
  if (hipMemcpy(out, g_sort_prof, (size_t)n * 12 * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) return EG_ERR_LAUNCH;
  return n;
}
#endif

// the XCD-aware record placement the training step / the operator apply on a grid of n_tiles tiles (one view per launch):
// tiles per block side = 2^shift, 0 = dense records (include/edgegs.h, eg_step_args::item_rec)
extern "C" int eg_record_xcd_shift(int32_t n_tiles) {
  return record_xcd_shift(n_tiles, n_tiles <= kPrefixHereMaxTiles, true, 1);
}

extern "C" int eg_sort_pairs(uint64_t *keys, const int32_t *offsets, int32_t T, int64_t capacity,
                             int32_t *flatten_ids, int64_t *isect_ids, int32_t max_tile_hint, eg_stream_t stream) {
  EG_REQUIRE(T > 0 && offsets, "bad arguments");
  if (capacity == 0) return EG_OK;
  EG_REQUIRE(keys && flatten_ids, "null pointer");
  return launch_tile_sort(keys, offsets, T, capacity, flatten_ids, isect_ids, max_tile_hint, SegTable{}, stream);
}

extern "C" int eg_sort_segments(uint64_t *keys, int32_t *tile_cursor, int32_t T, int32_t seg_cap,
                                int32_t *flatten_ids, int32_t *tile_start, int32_t *tile_end,
                                const int32_t *item_first, int32_t *item_end, int32_t *item_tile,
                                int32_t max_items, int32_t max_tile_hint, eg_stream_t stream) {
  EG_REQUIRE(T > 0 && seg_cap > 0 && max_items > 0, "bad sizes");
  EG_REQUIRE(keys && tile_cursor && flatten_ids && tile_start && tile_end && item_first && item_end && item_tile,
             "null pointer");
  SegTable seg;
  seg.cursor = tile_cursor; seg.seg_cap = seg_cap;
  seg.tile_start = tile_start; seg.tile_end = tile_end;
  seg.item_first = const_cast<int32_t *>(item_first); seg.item_end = item_end;  // (read only: total == nullptr)
  seg.item_tile = item_tile; seg.max_items = max_items;
  seg.total = nullptr;
  seg.item_rec = nullptr;
  seg.slice_major = 0;
  return launch_tile_sort(keys, nullptr, T, (int64_t)T * seg_cap, flatten_ids, nullptr, max_tile_hint, seg, stream);
}

namespace eg {
int record_xcd_shift(int T, bool prefix_here, bool has_item_rec, int C) {
  // (small grids stay dense: eight lists over a few dozen tiles are not balanced, and nothing there misses an L2)
  // Round 6 built the placement for grids ABOVE 2048 tiles as well (bands of tile rows: tile_sort_kernel) and measured it
  // (profiles/r06_xcd_large_ab.txt): the forward's fabric traffic falls from 3.4x / 3.1x to 1.6x / 1.4x of the algorithmic
  // bytes at 1600 x 1200 / 1200 x 680 -- and the forward takes 138 / 126 us instead of 133 / 116 (eight lists of unequal work;
  // with 2 x 2 blocks and a per-list scan in the projection's tail 134 / 117 and +4 us in that tail): the launch is not
  // traffic-bound.  OFF there; EG_XCD_LARGE=1 in a development build turns it on.
  int shift = (has_item_rec && C == 1 && T >= kXcdMinTiles) ? kXcdShiftDefault : 0;
  int xcd_large = 0;
#ifdef EG_DEV_SWITCHES
  static const int xcd_large_env = getenv("EG_XCD_LARGE") ? atoi(getenv("EG_XCD_LARGE")) : 0;  // (A/B switch)
  xcd_large = xcd_large_env;
#endif
  if (!prefix_here && !xcd_large) shift = 0;
#ifdef EG_DEV_SWITCHES
  static const int xcd_env = getenv("EG_XCD_SHIFT") ? atoi(getenv("EG_XCD_SHIFT")) : -1;  // (A/B switch)
  if (xcd_env >= 0 && shift > 0) shift = xcd_env;
#endif
  return shift;
}

int launch_sort_segments(uint64_t *keys, int32_t *tile_cursor, int32_t T, int32_t seg_cap, int32_t *flatten_ids,
                         int32_t *tile_start, int32_t *tile_end, int32_t *item_first, int32_t *item_end,
                         int32_t *item_tile, int32_t max_items, int32_t max_tile_hint, const Batch &bt, int C,
                         hipStream_t st, int32_t *total_prefix_here, int32_t *item_rec, const int32_t *item_front,
                         uint32_t rec_tag, int32_t tiles_per_row, const float *gt, const float *wmap, void *workspace,
                         int32_t width, int32_t height, int32_t front_slices, int32_t *total_flag) {
  SegTable seg;
  seg.total_flag = total_flag;
  seg.cursor = tile_cursor; seg.seg_cap = seg_cap;
  seg.tile_start = tile_start; seg.tile_end = tile_end;
  seg.item_first = item_first; seg.item_end = item_end;
  seg.item_tile = item_tile; seg.max_items = max_items;
  seg.total = total_prefix_here;
  // the forward's dispatch order (SegTable): front slices first; tiles taken middle-out
  seg.item_rec = (int4 *)item_rec;
  // the class boundary of the dispatch order: the caller's (the step passes 9 when its forward runs in chained mode --
  // under the XCD-aware placement 30.6 against 31.7 us at config 2, 21.1 against 22.1 with the initial opacities -- and 4
  // in speculative mode: 17.7 against 17.9; profiles/r05_front_slices_ab.txt), 4 by default
  seg.slice_major = front_slices > 0 ? min(front_slices, 15) : kFrontDefault;
  seg.middle_out = 1;
  seg.item_front = total_prefix_here ? nullptr : item_front;
  seg.rec_tag = rec_tag;
  // XCD-aware record placement ("prefix here" grids, one view per launch: the workgroup -> XCD map of a batched launch
  // depends on max_items % 8): blocks of 2 x 2 tiles
  seg.tw = tiles_per_row;
  seg.inv_tw = tiles_per_row > 0 ? 1.f / (float)tiles_per_row : 0.f;
  // the empty tiles' loss terms are added here (SegTable::skip_empty): the caller hands over the view's target / weights and
  // the compositing workspace exactly when its forward is the wave-autonomous one with the fused loss
  if (total_prefix_here && item_rec && tiles_per_row > 0 && workspace && width > 0 && height > 0 &&
      (C > 1 ? bt.gt[0] && bt.wmap[0] : gt && wmap)) {
    seg.skip_empty = 1;
    seg.gt = gt; seg.wmap = wmap; seg.width = width; seg.height = height;
    seg.loss_part = carve_workspace(workspace, max_items, T).loss_part;
#ifdef EG_DEV_SWITCHES
    static const int skip_env = getenv("EG_SKIP_EMPTY") ? atoi(getenv("EG_SKIP_EMPTY")) : 1;  // (A/B switch)
    seg.skip_empty = skip_env;
#endif
  }
  seg.xcd_shift = tiles_per_row > 0 ? record_xcd_shift(T, total_prefix_here != nullptr, item_rec != nullptr, C) : 0;
  if (!total_prefix_here) {
    // above 2048 tiles the placement works from the projection scan's two prefixes (item_first, item_front: EG_FLAG_FRONT_PREFIX)
    if (!(seg.xcd_shift > 0 && item_front && total_flag)) seg.xcd_shift = 0;
  }
#ifdef EG_DEV_SWITCHES  // A/B switches of development builds (edgegaussians_amd/build.py, EG_DEV_SWITCHES=1)
  static const int front = getenv("EG_FRONT_SLICES") ? atoi(getenv("EG_FRONT_SLICES")) : -2;  // (-2: not set)
  static const int middle_out = getenv("EG_SORT_MIDDLE_OUT") ? atoi(getenv("EG_SORT_MIDDLE_OUT")) : 1;
  if (front != -2) seg.slice_major = front < 0 ? 0 : (front > 15 ? 15 : front);
  seg.middle_out = middle_out;
  static const int front_large = getenv("EG_FRONT_LARGE") ? atoi(getenv("EG_FRONT_LARGE")) : 1;
  if (!front_large) seg.item_front = nullptr;
#endif

  return launch_tile_sort(keys, nullptr, T, (int64_t)T * seg_cap, flatten_ids, nullptr, max_tile_hint, seg,
                          (eg_stream_t)st, bt, C);
}
}  // namespace eg
