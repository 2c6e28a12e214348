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
//   4. one workgroup per tile sorts its segment in LDS (bitonic network on unique 64-bit keys)
// The result is bit-identical to gsplat's stable sort: inside a tile the order is by depth bits with
// ties broken by Gaussian id, which is exactly what a stable sort of index-ordered emissions gives.
// Segments larger than the LDS capacity fall back to a hybrid global/LDS bitonic sort by the
// same workgroup (slow path, correct for any size).
#include "common.h"

namespace eg {

__global__ void __launch_bounds__(256)
tile_count_kernel(const float2 *__restrict__ means2d, const int *__restrict__ radii, int N, int width,
                  int height, int *__restrict__ tiles_per_gauss, int *__restrict__ tile_counts) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= N) return;
  const int radius = radii[g];
  int n = 0;
  if (radius > 0) {
    const int tw = (width + kTile - 1) / kTile, th = (height + kTile - 1) / kTile;
    const float2 m = means2d[g];
    int x0, y0, x1, y1;
    tile_box(m.x, m.y, radius, tw, th, x0, y0, x1, y1);
    n = (y1 - y0) * (x1 - x0);
    for (int ty = y0; ty < y1; ++ty)
      for (int tx = x0; tx < x1; ++tx) atomicAdd(&tile_counts[ty * tw + tx], 1);
  }
  if (tiles_per_gauss) tiles_per_gauss[g] = n;
}

// single workgroup, 1024 threads: exclusive scan of counts[T] -> offsets[T+1]; counts stay intact
// (the emit pass counts them back down to zero, so no memset is ever needed between steps)
__global__ void __launch_bounds__(1024)
tile_offsets_kernel(const int *__restrict__ counts, int T, long long capacity, int *__restrict__ offsets,
                    int *__restrict__ total) {
  __shared__ int wave_sums[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < T; base += 1024) {
    const int i = base + tid;
    const int c = (i < T) ? counts[i] : 0;
    // inclusive scan inside the wave
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
    const int excl = carry + wave_excl + (s - c);
    if (i < T) offsets[i] = excl;
    __syncthreads();
    if (tid == 1023) carry = excl + c;
    __syncthreads();
  }
  if (tid == 0) {
    offsets[T] = carry;
    if (total) {
      total[0] = carry;
      total[1] = ((long long)carry > capacity) ? 1 : 0;
    }
  }
}

__global__ void __launch_bounds__(256)
tile_emit_kernel(const float2 *__restrict__ means2d, const int *__restrict__ radii,
                 const float *__restrict__ depths, const float4 *__restrict__ splat, int N, int width,
                 int height, const int *__restrict__ offsets, int *__restrict__ cursor, long long capacity,
                 unsigned long long *__restrict__ keys) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= N) return;
  float x, y, depth;
  int radius;
  if (splat) {
    const float4 s0 = splat[2 * g], s1 = splat[2 * g + 1];
    x = s0.x; y = s0.y; depth = s1.z; radius = __float_as_int(s1.w);
  } else {
    const float2 m = means2d[g];
    x = m.x; y = m.y; depth = depths[g]; radius = radii[g];
  }
  if (radius <= 0) return;
  const int tw = (width + kTile - 1) / kTile, th = (height + kTile - 1) / kTile;
  int x0, y0, x1, y1;
  tile_box(x, y, radius, tw, th, x0, y0, x1, y1);
  const unsigned long long key = ((unsigned long long)(unsigned)__float_as_int(depth) << 32) | (unsigned)g;
  for (int ty = y0; ty < y1; ++ty)
    for (int tx = x0; tx < x1; ++tx) {
      const int t = ty * tw + tx;
      const long long idx = (long long)offsets[t] + (atomicSub(&cursor[t], 1) - 1);
      if (idx < capacity) keys[idx] = key;
    }
}

// ---------------------------------------------------------------------------------------------
// segmented sort: one 256-thread workgroup per tile
constexpr int kSortCap = 4096;  // keys per LDS pass (32 KiB)

__device__ __forceinline__ void ce(unsigned long long &a, unsigned long long &b) {
  if (a > b) { const unsigned long long t = a; a = b; b = t; }
}

// all compare-exchanges of bitonic stages k = k_lo .. k_hi restricted to strides < P, on s[0..P)
// (ascending-only network: first substage of each k mirrors inside the k-block, then half-cleaners)
__device__ __forceinline__ void bitonic_lds(unsigned long long *s, int P, int k_lo, int k_hi, int tid) {
  for (int k = k_lo; k <= k_hi; k <<= 1) {
    if (k <= P) {
      const int hk = k >> 1;
      for (int i = tid; i < (P >> 1); i += 256) {
        const int blk = i / hk, off = i - blk * hk;
        const int lo = blk * k + off, hi = blk * k + k - 1 - off;
        unsigned long long a = s[lo], b = s[hi];
        if (a > b) { s[lo] = b; s[hi] = a; }
      }
      __syncthreads();
    }
    for (int j = min(k >> 2, P >> 1); j >= 1; j >>= 1) {
      for (int i = tid; i < (P >> 1); i += 256) {
        const int lo = ((i / j) * (j << 1)) + (i % j), hi = lo + j;
        unsigned long long a = s[lo], b = s[hi];
        if (a > b) { s[lo] = b; s[hi] = a; }
      }
      __syncthreads();
    }
  }
}

__global__ void __launch_bounds__(256)
tile_sort_kernel(unsigned long long *__restrict__ keys, const int *__restrict__ offsets, int T,
                 long long capacity, int *__restrict__ flatten_ids, long long *__restrict__ isect_ids) {
  __shared__ unsigned long long s[kSortCap];
  const int tile = blockIdx.x, tid = threadIdx.x;
  const long long start = offsets[tile];
  long long end = offsets[tile + 1];
  if (end > capacity) end = capacity;
  const int n = (int)(end - start);
  if (n <= 0) return;
  unsigned long long *seg = keys + start;
  const unsigned long long kInf = ~0ull;

  if (n <= kSortCap) {
    int P = 1;
    while (P < n) P <<= 1;
    for (int i = tid; i < P; i += 256) s[i] = (i < n) ? seg[i] : kInf;
    __syncthreads();
    bitonic_lds(s, P, 2, P, tid);
    for (int i = tid; i < n; i += 256) {
      const unsigned long long key = s[i];
      flatten_ids[start + i] = (int)(unsigned)(key & 0xffffffffull);
      if (isect_ids) isect_ids[start + i] = ((long long)tile << 32) | (long long)(key >> 32);
    }
    return;
  }

  // ---- slow path: segment larger than one LDS pass.  Virtual size P (power of two), indices >= n
  // behave as +inf and never move (the network only ever moves larger keys to higher indices).
  long long P = kSortCap;
  while (P < n) P <<= 1;
  // (a) sort every kSortCap chunk completely
  for (long long c0 = 0; c0 < n; c0 += kSortCap) {
    for (int i = tid; i < kSortCap; i += 256) s[i] = (c0 + i < n) ? seg[c0 + i] : kInf;
    __syncthreads();
    bitonic_lds(s, kSortCap, 2, kSortCap, tid);
    for (int i = tid; i < kSortCap; i += 256)
      if (c0 + i < n) seg[c0 + i] = s[i];
    __syncthreads();
  }
  // (b) merges for k > kSortCap: long strides in global memory, the tail (j <= kSortCap/2) in LDS
  for (long long k = 2 * (long long)kSortCap; k <= P; k <<= 1) {
    const long long hk = k >> 1;
    for (long long i = tid; i < (P >> 1); i += 256) {
      const long long blk = i / hk, off = i - blk * hk;
      const long long lo = blk * k + off, hi = blk * k + k - 1 - off;
      if (hi < n) {
        unsigned long long a = seg[lo], b = seg[hi];
        if (a > b) { seg[lo] = b; seg[hi] = a; }
      }
    }
    __syncthreads();
    for (long long j = k >> 2; j >= kSortCap; j >>= 1) {
      for (long long i = tid; i < (P >> 1); i += 256) {
        const long long lo = ((i / j) * (j << 1)) + (i % j), hi = lo + j;
        if (hi < n) {
          unsigned long long a = seg[lo], b = seg[hi];
          if (a > b) { seg[lo] = b; seg[hi] = a; }
        }
      }
      __syncthreads();
    }
    for (long long c0 = 0; c0 < n; c0 += kSortCap) {
      for (int i = tid; i < kSortCap; i += 256) s[i] = (c0 + i < n) ? seg[c0 + i] : kInf;
      __syncthreads();
      // only the half-cleaner substages j = kSortCap/2 .. 1 (k_lo = k_hi = 2*kSortCap > P skips the mirror)
      bitonic_lds(s, kSortCap, 2 * kSortCap, 2 * kSortCap, tid);
      for (int i = tid; i < kSortCap; i += 256)
        if (c0 + i < n) seg[c0 + i] = s[i];
      __syncthreads();
    }
  }
  for (int i = tid; i < n; i += 256) {
    const unsigned long long key = seg[i];
    flatten_ids[start + i] = (int)(unsigned)(key & 0xffffffffull);
    if (isect_ids) isect_ids[start + i] = ((long long)tile << 32) | (long long)(key >> 32);
  }
}

}  // namespace eg

using namespace eg;

extern "C" int eg_tile_count(const float *means2d, const int32_t *radii, int32_t N, int32_t width, int32_t height,
                             int32_t *tiles_per_gauss, int32_t *tile_counts, eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && width > 0 && height > 0, "bad sizes");
  if (N == 0) return EG_OK;
  EG_REQUIRE(means2d && radii && tile_counts, "null pointer");
  tile_count_kernel<<<cdiv(N, 256), 256, 0, as_stream(stream)>>>((const float2 *)means2d, radii, N, width, height,
                                                                tiles_per_gauss, tile_counts);
  return check_launch("tile_count");
}

extern "C" int eg_tile_offsets(const int32_t *tile_counts, int32_t T, int64_t capacity, int32_t *offsets, int32_t *total,
                               eg_stream_t stream) {
  EG_REQUIRE(T > 0 && tile_counts && offsets, "bad arguments");
  tile_offsets_kernel<<<1, 1024, 0, as_stream(stream)>>>(tile_counts, T, (long long)capacity, offsets, total);
  return check_launch("tile_offsets");
}

extern "C" int eg_tile_emit(const float *means2d, const int32_t *radii, const float *depths, const float *splat,
                            int32_t N, int32_t width, int32_t height, const int32_t *offsets, int32_t *tile_cursor,
                            int64_t capacity, uint64_t *keys, eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && width > 0 && height > 0 && capacity >= 0, "bad sizes");
  if (N == 0) return EG_OK;
  EG_REQUIRE(offsets && tile_cursor && (keys || capacity == 0), "null pointer");
  EG_REQUIRE(splat || (means2d && radii && depths), "need splat or (means2d, radii, depths)");
  tile_emit_kernel<<<cdiv(N, 256), 256, 0, as_stream(stream)>>>(
      (const float2 *)means2d, radii, depths, (const float4 *)splat, N, width, height, offsets, tile_cursor,
      (long long)capacity, (unsigned long long *)keys);
  return check_launch("tile_emit");
}

extern "C" int eg_sort_pairs(uint64_t *keys, const int32_t *offsets, int32_t T, int64_t capacity,
                             int32_t *flatten_ids, int64_t *isect_ids, eg_stream_t stream) {
  EG_REQUIRE(T > 0 && offsets, "bad arguments");
  if (capacity == 0) return EG_OK;
  EG_REQUIRE(keys && flatten_ids, "null pointer");
  tile_sort_kernel<<<T, 256, 0, as_stream(stream)>>>((unsigned long long *)keys, offsets, T, (long long)capacity,
                                                    flatten_ids, (long long *)isect_ids);
  return check_launch("tile_sort");
}
