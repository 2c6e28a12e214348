// Device code of the footprint backward (G8, order-independent form: DESIGN.md section 5) shared by composite.hip
// (footprint_bwd_kernel) and backward_fused.hip.  Moved out of composite.hip in round 6, unchanged; the derivation and
// the description of the walk sit in front of footprint_bwd_kernel in composite.hip.
#pragma once
#include "composite.h"

namespace eg {

struct Walk {
  int i0, fh;     // rows i0 .. i0 + fh - 1
  int pw;         // cells per row
  int jlo, jhi;   // inclusive column clip: gsplat's tile box, the image, the ellipse's extent
  int cells;      // fh * pw, 0 for an invisible Gaussian
  float thr;      // sigma threshold ln(255 o) (+ margin)
  float xoff, shear;  // first column of row i: ceil(xoff + shear * (y_g - (i + 0.5)))
};

__device__ __forceinline__ Walk walk_of(const float4 s0, const float4 s1, int width, int height) {
  Walk w;
  w.i0 = w.fh = w.pw = w.jlo = w.cells = 0;
  w.jhi = -1;
  w.thr = w.xoff = w.shear = 0.f;
  const int radius = __float_as_int(s1.w);
  // (the raw v_log_f32: 255 o >= 1 wherever the result is used, no denormal scaling needed; 1 ulp, inside the margin)
  const float thr = 0.693147180559945f * __builtin_amdgcn_logf(255.f * s1.y) + kThrMargin;
  const float det = s0.z * s1.x - s0.w * s0.w;
  if (radius <= 0 || !(thr > 0.f) || !(det > 0.f) || !(s0.z > 0.f)) return w;
  const int tw = (width + kTile - 1) / kTile, th = (height + kTile - 1) / kTile;
  int x0, y0, x1, y1;
  tile_box(s0.x, s0.y, radius, tw, th, x0, y0, x1, y1);  // pixels outside gsplat's box never see it
  // hardware sqrt / rcp (1 ulp): the 0.1 % + 0.01 px inflation swallows their error
  const float k = 2.f * thr * __builtin_amdgcn_rcpf(det);
  const float ex = __builtin_amdgcn_sqrtf(k * s1.x) * 1.001f + 0.01f;
  const float ey = __builtin_amdgcn_sqrtf(k * s0.z) * 1.001f + 0.01f;
  const int j0 = max(x0 * kTile, (int)ceilf(s0.x - ex - 0.5f));
  const int j1 = min(min(x1 * kTile, width) - 1, (int)floorf(s0.x + ex - 0.5f));
  const int i0 = max(y0 * kTile, (int)ceilf(s0.y - ey - 0.5f));
  const int i1 = min(min(y1 * kTile, height) - 1, (int)floorf(s0.y + ey - 0.5f));
  const int fw = j1 - j0 + 1, fh = i1 - i0 + 1;
  if (fw <= 0 || fh <= 0) return w;
  const float inv_a = __builtin_amdgcn_rcpf(s0.z);
  const float hw = __builtin_amdgcn_sqrtf(2.f * thr * inv_a) * 1.001f + 0.01f;
  const int pw = (int)(2.f * hw) + 1;
  w.i0 = i0; w.fh = fh; w.jlo = j0; w.jhi = j1; w.thr = thr;
  if (pw < fw) {
    w.pw = pw; w.shear = s0.w * inv_a; w.xoff = s0.x - hw - 0.5f;
  } else {
    w.pw = fw; w.shear = 0.f; w.xoff = (float)j0;
  }
  w.cells = w.pw * fh;
  return w;
}

struct Moments {
  float w_x, w_y, w_xx, w_xy, w_yy, abs_x, abs_y, v_o;
};

// One PAIR of horizontally adjacent cells of a footprint as the walk carries it from the prefetch to the visit: the two
// pixels' records (all zeros: nothing to visit) and the offset of the Gaussian's centre from the LEFT pixel's (the right
// pixel's is one less in x).
constexpr unsigned kOutOfImage = 0x80000000u;  // a byte offset no gtstop image reaches (launch_footprint_bwd checks)
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
struct Pair {
  u32x3 rec0, rec1;  // {gT bits, stop id, stop depth bits}: the pixel's last contributor, all ones if its walk did not stop
  float dx, dy;
};

// The quadratic form of a Gaussian as a visit uses it (round 5): sigma log2(e) = dx hx + dy hy with the HALF gradients
// hx = A dx + B dy, hy = B dx + C dy -- six operations give sigma and both components of d sigma / d (dx, dy), which the
// absgrad sums need anyway (|dL/dmean2d| of one pixel = |w| |a dx + b dy|, |w| |b dx + c dy|): the accepted part of a
// visit no longer forms a wx + b wy, b wx + c wy (six operations, now two), the exponential takes sigma log2(e) as it
// is, and the sums of |w| |h| are scaled by 2 / log2(e) once per lane.  One step to the right (dx - 1) takes A off hx
// and B off hy: the pair's second cell costs four operations.
struct Quad {
  float A, B, C;  // log2(e) / 2 * (a, b, c)
  float thr2;     // log2(e) * sigma threshold (> 0)
  float o;        // opacity
  bool high_o;    // (wave-uniform) some lane of the wave walks a Gaussian of opacity > 0.999
};
constexpr float kLog2e = 1.44269504088896341f;

__device__ __forceinline__ void footprint_cell(const Quad qd, unsigned g, unsigned dg, const u32x3 rec, float s2, float hx,
                                               float hy, float dx, float dy, Moments &m) {
  // One branch on the whole acceptance test (whole waves fall outside on large footprints, and on the steps whose
  // weight map is sparse), none after it: the lanes of a wave sit in up to eight footprints, accepted and rejected
  // pixels are mixed, and further branching only adds exec-mask bookkeeping.  The compares are what costs here (a
  // v_cmp takes the vector pipe ~1.5x as long as a multiply-add, tools/microbench): four of them --
  //   0 <= s2 <= thr2 as ONE unsigned compare of the bit patterns (a negative or NaN s2 has a pattern above any
  //   positive float's; s2 = -0 cannot arise: dx hx and dy hy would both have to be -0);
  //   where the walk of this pixel stopped only Gaussians at or before its last contributor count, (depth bits, id) <=
  //   (its depth bits, its id): "in front by depth" is one compare, the tie (the last contributor itself, or a twin at
  //   the very same depth) is looked at under a wave-uniform branch that is rarely taken.
  const float gT = __uint_as_float(rec.x);
  const bool in = __float_as_uint(s2) <= __float_as_uint(qd.thr2);
  bool counts = rec.z > dg;
  const bool tie = rec.z == dg;
  if (__builtin_amdgcn_ballot_w64(tie) != 0ull) {
    asm volatile("" ::: "memory");  // (a real branch: if-converted, the id compare would run on every visit)
    counts |= tie & (rec.y >= g);
  }
  if (!(gT != 0.f && in && counts)) return;
  const float vis = __builtin_amdgcn_exp2f(-s2);
  const float araw = qd.o * vis;
  // forward: skip if min(0.999, araw) < 1/255; gsplat's backward: no gradient through a clamped alpha (araw <= o:
  // only a wave that holds a Gaussian of opacity above 0.999 has to look)
  bool ok = araw >= kAlphaMin;
  if (qd.high_o) {
    asm volatile("" ::: "memory");
    ok &= araw <= kAlphaMax;
  }
  const float v_alpha = ok ? gT * __builtin_amdgcn_rcpf(1.f - araw) : 0.f;
  m.v_o += vis * v_alpha;
  const float w = -araw * v_alpha;
  const float wx = w * dx, wy = w * dy;
  m.w_x += wx; m.w_y += wy;
  m.w_xx += wx * dx; m.w_xy += wx * dy; m.w_yy += wy * dy;
  m.abs_x = fmaf(fabsf(w), fabsf(hx), m.abs_x);
  m.abs_y = fmaf(fabsf(w), fabsf(hy), m.abs_y);
}

__device__ __forceinline__ void footprint_visit(const Quad qd, unsigned g, unsigned dg, const Pair c, Moments &m) {
  const float dx = c.dx, dy = c.dy;
  const float hx = fmaf(qd.A, dx, qd.B * dy);
  const float hy = fmaf(qd.C, dy, qd.B * dx);
  const float s2 = fmaf(dx, hx, dy * hy);  // sigma log2(e)
  const float dx1 = dx - 1.f, hx1 = hx - qd.A, hy1 = hy - qd.B;
  const float s21 = fmaf(dx1, hx1, dy * hy1);
  footprint_cell(qd, g, dg, c.rec0, s2, hx, hy, dx, dy, m);
  footprint_cell(qd, g, dg, c.rec1, s21, hx1, hy1, dx1, dy, m);
}

// Lane r of n walks the PAIRS r, r + n, r + 2n, ... of Gaussian g's sheared box (its rows are cut into
// ceil(pw / 2) pairs of neighbouring cells; the odd cell out of an odd row's last pair lies outside the box, hence
// outside the ellipse: its own sigma test rejects it).  What a visit has to know about where it is -- row, first column
// of the row, the clip, the byte offset -- is worked out once per pair (round 5: it was once per cell, and together with
// the compares of the accept test it, not the accepted part, was where this kernel's time went: with the accepted part
// compiled out the launch at 1600 x 1200 took 152 of 174 us, with the loads compiled out as well 146); the two
// records of a pair are 24 contiguous bytes.  The records of the NEXT pair are prefetched while this one is evaluated;
// the loop body is written out twice with the two register sets swapping roles (a rotating copy cost ten moves per pair
// of visits), and the prefetch is unconditional: a cell with nothing to visit asks for an offset beyond the image.
__device__ __forceinline__ void footprint_walk(const float4 s0, const float4 s1, int g, int r, int n, int i0, int fh,
                                               int pw, int jlo, int jhi, float thr, float xoff,
                                               float shear, int width, const __amdgpu_buffer_rsrc_t gtstop, Moments &m) {
  const int ppr = (pw + 1) >> 1;  // pairs per row
  const float inv_ppr = __builtin_amdgcn_rcpf((float)ppr);
  // pair -> (row, pair in the row); the quotient estimate is exact for pairs < 2^21, the fix-up is free
  auto divmod = [&](int q, int &qi, int &qj) {
    qi = (int)(((float)q + 0.5f) * inv_ppr);
    qj = q - __mul24(qi, ppr);
    if (qj < 0) { qj += ppr; --qi; }
    if (qj >= ppr) { qj -= ppr; ++qi; }
  };
  const int width12 = width * (int)sizeof(StopRec);
  // (centre - 0.5 once per Gaussian: the pixel centres are at integer + 0.5)
  const float xc = s0.x - 0.5f, yc = s0.y - 0.5f;
  // The records come through buffer loads: a 32-bit byte offset on the uniform descriptor (two full-rate 24-bit
  // multiply-adds and the load's own address adder instead of a quarter-rate 64-bit multiply-add per visit), and a
  // cell with nothing to visit -- past the lane's last pair or outside the column clip -- asks for an offset beyond the
  // image: the hardware's range check returns zeros without touching memory, and gT == 0 is "skip".  "Nothing to
  // visit" is the sign bit of (j - jlo) | (jhi - j) | (pairs left - 1), moved into the offset's top bit: no compare.
  auto fetch = [&](int i, int cp, int live1) -> Pair {
    Pair p;
    p.dy = yc - (float)i;
    const int j = (int)ceilf(xoff + shear * p.dy) + 2 * cp;
    const int t1 = j - jlo, t2 = jhi - j;
    const unsigned bad0 = (unsigned)(t1 | t2 | live1) & kOutOfImage;
    const unsigned bad1 = (unsigned)((t1 + 1) | (t2 - 1) | live1) & kOutOfImage;
    const unsigned base = (unsigned)(__mul24(i, width12) + __mul24(j, 12));
    p.rec0 = __builtin_amdgcn_raw_buffer_load_b96(gtstop, (int)(bad0 | base), 0, 0);
    p.rec1 = __builtin_amdgcn_raw_buffer_load_b96(gtstop, (int)(bad1 | (base + 12u)), 0, 0);
    p.dx = xc - (float)j;
    return p;
  };
  const unsigned ug = (unsigned)g, dg = (unsigned)__float_as_int(s1.z);
  const Quad qd = {0.5f * kLog2e * s0.z, 0.5f * kLog2e * s0.w, 0.5f * kLog2e * s1.x, kLog2e * thr, s1.y,
                   __builtin_amdgcn_ballot_w64(s1.y > kAlphaMax) != 0ull};
  int di, dc;
  divmod(n, di, dc);
  int ia, ca;
  divmod(r, ia, ca);
  ia += i0;
  int left = __mul24(fh, ppr) - r;  // > 0 while the lane still has a pair
  auto advance = [&]() {
    left -= n;
    // (the carry without a compare: t = c + dc - ppr, its sign mask is -1 for "no carry")
    const int t = ca + dc - ppr, sm = t >> 31;
    ca = t + (ppr & sm);
    ia += di + 1 + sm;
  };
  Pair p0 = fetch(ia, ca, left - 1);
  while (left > 0) {
    advance();
    const Pair p1 = fetch(ia, ca, left - 1);
    footprint_visit(qd, ug, dg, p0, m);
    if (left <= 0) break;
    advance();
    p0 = fetch(ia, ca, left - 1);
    footprint_visit(qd, ug, dg, p1, m);
  }
}

}  // namespace eg
