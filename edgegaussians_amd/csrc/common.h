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

inline hipStream_t as_stream(eg_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
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

}  // namespace eg

#define EG_REQUIRE(cond, msg)                         \
  do {                                                \
    if (!(cond)) {                                    \
      eg::set_error("%s: %s", __func__, msg);         \
      return EG_ERR_ARG;                              \
    }                                                 \
  } while (0)
