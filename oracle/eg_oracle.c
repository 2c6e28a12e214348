/*
 * eg_oracle.c -- plain-C (C99 + OpenMP) restatement of the edge-Gaussian rasterizer path.
 * TEST INFRASTRUCTURE ONLY: built into oracle/_build/libeg_oracle.so by oracle/Makefile; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product never does.
 *
 * It restates, kernel by kernel and in the kernels' own sequential order, the algorithm behind the
 * one call  gsplat.rasterization(...)  at /root/reference/edgegaussians/models/edge_gs.py:250-268
 * (gsplat==1.0.0, /root/reference/requirements.txt:64 -- not vendored, not installable here; restated
 * from SURVEY.md 2.3 / 8a, i.e. PARITY UNPINNED for this arithmetic, see oracle/ref_torch.py) plus the
 * per-step glue: clamp + weighted L1 (edge_gs.py:279,288-324), absgrad (edge_gs.py:607-613), Adam
 * (train_utils.py:50-60, torch 1.13 update order).
 *
 * Unlike oracle/ref_torch.py (dense tensors, autograd backward) this file walks every pixel's
 * depth-sorted list one Gaussian at a time and carries the hand-derived backward with transmittance
 * recovery (T /= 1 - alpha) and the behind-colour buffer, for arbitrary per-Gaussian colours -- so the
 * three implementations (torch autograd, this C, the HIP kernels) check each other.
 */
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 16
#define ALPHA_MAX 0.999f
#define ALPHA_MIN (1.0f / 255.0f)
#define T_STOP 1e-4f
#define FOV_CLAMP 1.3f

static inline float fclampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

/* ---------------------------------------------------------------- G1: projection forward -------- */
typedef struct {
  float x, y, z, qw, qx, qy, qz, qinv, s[3], R[9], W[9], rz, rz2, tx, ty;
  int in_x, in_y;
  float J00, J02, J11, J12, p0[3], p1[3], c00, c01, c11, b00, b11, det0, det1, a, b, c, comp, u, v;
} geom_t;

static int forward_geom(const float *vm, const float *K, const float *mean, const float *quat, const float *scale,
                        int width, int height, float near_plane, float far_plane, float eps2d, geom_t *f) {
  const float Rv[9] = {vm[0], vm[1], vm[2], vm[4], vm[5], vm[6], vm[8], vm[9], vm[10]};
  const float tv[3] = {vm[3], vm[7], vm[11]};
  const float fx = K[0], cx = K[2], fy = K[4], cy = K[5];
  f->x = Rv[0] * mean[0] + Rv[1] * mean[1] + Rv[2] * mean[2] + tv[0];
  f->y = Rv[3] * mean[0] + Rv[4] * mean[1] + Rv[5] * mean[2] + tv[1];
  f->z = Rv[6] * mean[0] + Rv[7] * mean[1] + Rv[8] * mean[2] + tv[2];
  if (f->z < near_plane || f->z > far_plane) return 0;
  float w = quat[0], x = quat[1], y = quat[2], z = quat[3];
  f->qinv = 1.0f / sqrtf(w * w + x * x + y * y + z * z);
  w *= f->qinv; x *= f->qinv; y *= f->qinv; z *= f->qinv;
  f->qw = w; f->qx = x; f->qy = y; f->qz = z;
  float *R = f->R;
  R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - w * z);       R[2] = 2.f * (x * z + w * y);
  R[3] = 2.f * (x * y + w * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - w * x);
  R[6] = 2.f * (x * z - w * y);       R[7] = 2.f * (y * z + w * x);       R[8] = 1.f - 2.f * (x * x + y * y);
  for (int k = 0; k < 3; ++k) f->s[k] = scale[k];
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k)
      f->W[3 * i + k] = (Rv[3 * i] * R[k] + Rv[3 * i + 1] * R[3 + k] + Rv[3 * i + 2] * R[6 + k]) * f->s[k];
  const float lim_x = FOV_CLAMP * (0.5f * (float)width / fx), lim_y = FOV_CLAMP * (0.5f * (float)height / fy);
  f->rz = 1.f / f->z;
  f->rz2 = f->rz * f->rz;
  const float xr = f->x * f->rz, yr = f->y * f->rz;
  f->in_x = (xr <= lim_x) && (xr >= -lim_x);
  f->in_y = (yr <= lim_y) && (yr >= -lim_y);
  f->tx = f->z * fclampf(xr, -lim_x, lim_x);
  f->ty = f->z * fclampf(yr, -lim_y, lim_y);
  f->J00 = fx * f->rz; f->J11 = fy * f->rz;
  f->J02 = -fx * f->tx * f->rz2; f->J12 = -fy * f->ty * f->rz2;
  for (int k = 0; k < 3; ++k) {
    f->p0[k] = f->J00 * f->W[k] + f->J02 * f->W[6 + k];
    f->p1[k] = f->J11 * f->W[3 + k] + f->J12 * f->W[6 + k];
  }
  f->c00 = f->p0[0] * f->p0[0] + f->p0[1] * f->p0[1] + f->p0[2] * f->p0[2];
  f->c01 = f->p0[0] * f->p1[0] + f->p0[1] * f->p1[1] + f->p0[2] * f->p1[2];
  f->c11 = f->p1[0] * f->p1[0] + f->p1[1] * f->p1[1] + f->p1[2] * f->p1[2];
  f->u = fx * f->x * f->rz + cx;
  f->v = fy * f->y * f->rz + cy;
  f->det0 = f->c00 * f->c11 - f->c01 * f->c01;
  f->b00 = f->c00 + eps2d; f->b11 = f->c11 + eps2d;
  f->det1 = f->b00 * f->b11 - f->c01 * f->c01;
  if (f->det1 <= 0.f) return 0;
  f->comp = sqrtf(fmaxf(0.f, f->det0 / f->det1));
  const float inv = 1.f / f->det1;
  f->a = f->b11 * inv; f->b = -f->c01 * inv; f->c = f->b00 * inv;
  return 1;
}

void ego_project_fwd(const float *means, const float *quats, const float *scales, const float *vm, const float *K,
                     int N, int width, int height, float near_plane, float far_plane, float eps2d, float radius_clip,
                     int32_t *radii, float *means2d, float *depths, float *conics, float *comps) {
#pragma omp parallel for schedule(static)
  for (int g = 0; g < N; ++g) {
    geom_t f;
    int radius = 0;
    if (forward_geom(vm, K, means + 3 * g, quats + 4 * g, scales + 3 * g, width, height, near_plane, far_plane, eps2d, &f)) {
      const float bh = 0.5f * (f.b00 + f.b11);
      const float v1 = bh + sqrtf(fmaxf(0.01f, bh * bh - f.det1));
      const float r = ceilf(3.f * sqrtf(v1));
      if (r > radius_clip && !(f.u + r <= 0.f || f.u - r >= (float)width || f.v + r <= 0.f || f.v - r >= (float)height))
        radius = (int)r;
    }
    radii[g] = radius;
    means2d[2 * g] = radius ? f.u : 0.f; means2d[2 * g + 1] = radius ? f.v : 0.f;
    depths[g] = radius ? f.z : 0.f;
    conics[3 * g] = radius ? f.a : 0.f; conics[3 * g + 1] = radius ? f.b : 0.f; conics[3 * g + 2] = radius ? f.c : 0.f;
    comps[g] = radius ? f.comp : 0.f;
  }
}

/* ---------------------------------------------------------------- G2-G6: binning, sort, offsets -- */
static void tile_box(float x, float y, int radius, int tw, int th, int *x0, int *y0, int *x1, int *y1) {
  const float ts = (float)TILE, tr = (float)radius / ts, tx = x / ts, ty = y / ts;
  int a = (int)floorf(tx - tr), b = (int)floorf(ty - tr), c = (int)ceilf(tx + tr), d = (int)ceilf(ty + tr);
  *x0 = a < 0 ? 0 : (a > tw ? tw : a); *y0 = b < 0 ? 0 : (b > th ? th : b);
  *x1 = c < 0 ? 0 : (c > tw ? tw : c); *y1 = d < 0 ? 0 : (d > th ? th : d);
}

int64_t ego_isect_count(const float *means2d, const int32_t *radii, int N, int width, int height, int32_t *tiles_per_gauss) {
  const int tw = (width + TILE - 1) / TILE, th = (height + TILE - 1) / TILE;
  int64_t M = 0;
  for (int g = 0; g < N; ++g) {
    int n = 0;
    if (radii[g] > 0) {
      int x0, y0, x1, y1;
      tile_box(means2d[2 * g], means2d[2 * g + 1], radii[g], tw, th, &x0, &y0, &x1, &y1);
      n = (y1 - y0) * (x1 - x0);
    }
    tiles_per_gauss[g] = n;
    M += n;
  }
  return M;
}

/* emit in Gaussian order, row-major over the tile box, then a STABLE LSD radix sort on the 64-bit key */
void ego_isect_emit_sort(const float *means2d, const int32_t *radii, const float *depths, int N, int width, int height,
                         int64_t M, int64_t *isect_ids, int32_t *flatten_ids, int32_t *offsets /*[T]*/) {
  const int tw = (width + TILE - 1) / TILE, th = (height + TILE - 1) / TILE, T = tw * th;
  int64_t cur = 0;
  for (int g = 0; g < N; ++g) {
    if (radii[g] <= 0) continue;
    int x0, y0, x1, y1;
    tile_box(means2d[2 * g], means2d[2 * g + 1], radii[g], tw, th, &x0, &y0, &x1, &y1);
    int32_t dbits;
    memcpy(&dbits, depths + g, 4);
    for (int ty = y0; ty < y1; ++ty)
      for (int tx = x0; tx < x1; ++tx) {
        isect_ids[cur] = ((int64_t)(ty * tw + tx) << 32) | (int64_t)(uint32_t)dbits;
        flatten_ids[cur] = g;
        ++cur;
      }
  }
  int64_t *k2 = (int64_t *)malloc(sizeof(int64_t) * (size_t)(M > 0 ? M : 1));
  int32_t *v2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(M > 0 ? M : 1));
  int64_t *ka = isect_ids, *kb = k2;
  int32_t *va = flatten_ids, *vb = v2;
  for (int pass = 0; pass < 8; ++pass) {
    size_t cnt[257] = {0};
    const int sh = 8 * pass;
    for (int64_t i = 0; i < M; ++i) ++cnt[((uint64_t)ka[i] >> sh & 255) + 1];
    for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
    for (int64_t i = 0; i < M; ++i) {
      const size_t p = cnt[(uint64_t)ka[i] >> sh & 255]++;
      kb[p] = ka[i];
      vb[p] = va[i];
    }
    int64_t *tk = ka; ka = kb; kb = tk;
    int32_t *tv = va; va = vb; vb = tv;
  }
  /* 8 passes: the result is back in the caller's arrays */
  free(k2);
  free(v2);
  /* offsets[t] = first index of tile t's run (= number of keys with a smaller tile id) */
  int64_t i = 0;
  for (int t = 0; t < T; ++t) {
    while (i < M && (isect_ids[i] >> 32) < t) ++i;
    offsets[t] = (int32_t)i;
  }
}

/* ---------------------------------------------------------------- G7: compositing forward -------- */
void ego_composite_fwd(const float *means2d, const float *conics, const float *colors, const float *opac, int CH,
                       int width, int height, const int32_t *offsets, const int32_t *flatten_ids, int64_t M,
                       float *render, float *alphas, int32_t *last_ids) {
  const int tw = (width + TILE - 1) / TILE, th = (height + TILE - 1) / TILE, T = tw * th;
#pragma omp parallel for schedule(dynamic, 1)
  for (int t = 0; t < T; ++t) {
    const int64_t start = offsets[t], end = (t == T - 1) ? M : offsets[t + 1];
    const int ty = t / tw, tx = t % tw;
    for (int di = 0; di < TILE; ++di)
      for (int dj = 0; dj < TILE; ++dj) {
        const int i = ty * TILE + di, j = tx * TILE + dj;
        if (i >= height || j >= width) continue;
        const float px = (float)j + 0.5f, py = (float)i + 0.5f;
        float Tt = 1.f, pix[8] = {0};
        int32_t cur = 0;
        for (int64_t idx = start; idx < end; ++idx) {
          const int g = flatten_ids[idx];
          const float dx = means2d[2 * g] - px, dy = means2d[2 * g + 1] - py;
          const float a = conics[3 * g], b = conics[3 * g + 1], c = conics[3 * g + 2];
          const float sigma = 0.5f * (a * dx * dx + c * dy * dy) + b * dx * dy;
          const float alpha = fminf(ALPHA_MAX, opac[g] * expf(-sigma));
          if (sigma < 0.f || alpha < ALPHA_MIN) continue;
          const float next_T = Tt * (1.f - alpha);
          if (next_T <= T_STOP) break; /* this pixel is done: exclusive */
          const float vis = alpha * Tt;
          for (int k = 0; k < CH; ++k) pix[k] += colors[(size_t)g * CH + k] * vis;
          cur = (int32_t)idx;
          Tt = next_T;
        }
        const size_t p = (size_t)i * width + j;
        alphas[p] = 1.f - Tt;
        last_ids[p] = cur;
        for (int k = 0; k < CH; ++k) render[p * CH + k] = pix[k];
      }
  }
}

/* which pixels' walks ended on the transmittance rule (the `break` of ego_composite_fwd above, same arithmetic):
 * the fused HIP forward records the id of the last contributor only for those, so the parity tests need the set */
void ego_composite_stopped(const float *means2d, const float *conics, const float *opac, int width, int height,
                           const int32_t *offsets, const int32_t *flatten_ids, int64_t M, uint8_t *stopped) {
  const int tw = (width + TILE - 1) / TILE, th = (height + TILE - 1) / TILE, T = tw * th;
#pragma omp parallel for schedule(dynamic, 1)
  for (int t = 0; t < T; ++t) {
    const int64_t start = offsets[t], end = (t == T - 1) ? M : offsets[t + 1];
    const int ty = t / tw, tx = t % tw;
    for (int di = 0; di < TILE; ++di)
      for (int dj = 0; dj < TILE; ++dj) {
        const int i = ty * TILE + di, j = tx * TILE + dj;
        if (i >= height || j >= width) continue;
        const float px = (float)j + 0.5f, py = (float)i + 0.5f;
        float Tt = 1.f;
        uint8_t st = 0;
        for (int64_t idx = start; idx < end; ++idx) {
          const int g = flatten_ids[idx];
          const float dx = means2d[2 * g] - px, dy = means2d[2 * g + 1] - py;
          const float a = conics[3 * g], b = conics[3 * g + 1], c = conics[3 * g + 2];
          const float sigma = 0.5f * (a * dx * dx + c * dy * dy) + b * dx * dy;
          const float alpha = fminf(ALPHA_MAX, opac[g] * expf(-sigma));
          if (sigma < 0.f || alpha < ALPHA_MIN) continue;
          const float next_T = Tt * (1.f - alpha);
          if (next_T <= T_STOP) { st = 1; break; }
          Tt = next_T;
        }
        stopped[(size_t)i * width + j] = st;
      }
  }
}

/* ---------------------------------------------------------------- G8: compositing backward ------- */
static inline void atomic_addf(float *p, float v) {
#pragma omp atomic
  *p += v;
}

void ego_composite_bwd(const float *means2d, const float *conics, const float *colors, const float *opac, int CH,
                       int width, int height, const int32_t *offsets, const int32_t *flatten_ids, int64_t M,
                       const float *alphas, const int32_t *last_ids, const float *v_render, const float *v_alphas,
                       float *v_means2d, float *v_means2d_abs, float *v_conics, float *v_colors, float *v_opac) {
  const int tw = (width + TILE - 1) / TILE, th = (height + TILE - 1) / TILE, T = tw * th;
#pragma omp parallel for schedule(dynamic, 1)
  for (int t = 0; t < T; ++t) {
    const int64_t start = offsets[t], end = (t == T - 1) ? M : offsets[t + 1];
    if (end <= start) continue;
    const int ty = t / tw, tx = t % tw;
    for (int di = 0; di < TILE; ++di)
      for (int dj = 0; dj < TILE; ++dj) {
        const int i = ty * TILE + di, j = tx * TILE + dj;
        if (i >= height || j >= width) continue;
        const size_t p = (size_t)i * width + j;
        const float px = (float)j + 0.5f, py = (float)i + 0.5f;
        const float T_final = 1.f - alphas[p];
        float Tt = T_final, buffer[8] = {0};
        const int32_t bin_final = last_ids[p];
        const float va_pix = v_alphas ? v_alphas[p] : 0.f;
        for (int64_t idx = end - 1; idx >= start; --idx) { /* back to front */
          if (idx > bin_final) continue;
          const int g = flatten_ids[idx];
          const float dx = means2d[2 * g] - px, dy = means2d[2 * g + 1] - py;
          const float a = conics[3 * g], b = conics[3 * g + 1], c = conics[3 * g + 2];
          const float sigma = 0.5f * (a * dx * dx + c * dy * dy) + b * dx * dy;
          const float vis = expf(-sigma);
          const float alpha = fminf(ALPHA_MAX, opac[g] * vis);
          if (sigma < 0.f || alpha < ALPHA_MIN) continue;
          const float ra = 1.f / (1.f - alpha);
          Tt *= ra; /* transmittance BEFORE this Gaussian */
          const float fac = alpha * Tt;
          float v_alpha = 0.f;
          for (int k = 0; k < CH; ++k) {
            const float vr = v_render[p * CH + k];
            if (v_colors) atomic_addf(&v_colors[(size_t)g * CH + k], fac * vr);
            v_alpha += (colors[(size_t)g * CH + k] * Tt - buffer[k] * ra) * vr;
          }
          v_alpha += T_final * ra * va_pix;
          if (opac[g] * vis <= ALPHA_MAX) {
            const float v_sigma = -opac[g] * vis * v_alpha;
            const float gx = v_sigma * (a * dx + b * dy), gy = v_sigma * (b * dx + c * dy);
            atomic_addf(&v_conics[3 * g], 0.5f * v_sigma * dx * dx);
            atomic_addf(&v_conics[3 * g + 1], v_sigma * dx * dy);
            atomic_addf(&v_conics[3 * g + 2], 0.5f * v_sigma * dy * dy);
            atomic_addf(&v_means2d[2 * g], gx);
            atomic_addf(&v_means2d[2 * g + 1], gy);
            if (v_means2d_abs) {
              atomic_addf(&v_means2d_abs[2 * g], fabsf(gx));
              atomic_addf(&v_means2d_abs[2 * g + 1], fabsf(gy));
            }
            atomic_addf(&v_opac[g], vis * v_alpha);
          }
          for (int k = 0; k < CH; ++k) buffer[k] += colors[(size_t)g * CH + k] * fac;
        }
      }
  }
}

/* ---------------------------------------------------------------- G9: projection backward ------- */
void ego_project_bwd(const float *means, const float *quats, const float *scales, const float *vm, const float *K,
                     int N, int width, int height, float eps2d, const int32_t *radii, const float *v_means2d,
                     const float *v_depths, const float *v_conics, const float *v_comps, float *v_means, float *v_quats,
                     float *v_scales) {
  const float Rv[9] = {vm[0], vm[1], vm[2], vm[4], vm[5], vm[6], vm[8], vm[9], vm[10]};
  const float fx = K[0], fy = K[4];
#pragma omp parallel for schedule(static)
  for (int g = 0; g < N; ++g) {
    for (int k = 0; k < 3; ++k) v_means[3 * g + k] = v_scales[3 * g + k] = 0.f;
    for (int k = 0; k < 4; ++k) v_quats[4 * g + k] = 0.f;
    if (radii[g] <= 0) continue;
    geom_t f;
    forward_geom(vm, K, means + 3 * g, quats + 4 * g, scales + 3 * g, width, height, -3e38f, 3e38f, eps2d, &f);
    const float vx = v_means2d[2 * g], vy = v_means2d[2 * g + 1];
    const float va = v_conics[3 * g], vb = v_conics[3 * g + 1], vc = v_conics[3 * g + 2];
    const float hb = 0.5f * vb;
    const float av00 = f.a * va + f.b * hb, av01 = f.a * hb + f.b * vc, av10 = f.b * va + f.c * hb, av11 = f.b * hb + f.c * vc;
    float G00 = -(av00 * f.a + av01 * f.b), G01 = -(av00 * f.b + av01 * f.c), G11 = -(av10 * f.b + av11 * f.c);
    if (v_comps) { /* gsplat guards the 1/(2 comp) with +1e-6 */
      const float det_conic = f.a * f.c - f.b * f.b;
      const float vs = v_comps[g] * 0.5f / (f.comp + 1e-6f), omc = 1.f - f.comp * f.comp;
      G00 += vs * (omc * f.a - eps2d * det_conic);
      G01 += vs * (omc * f.b);
      G11 += vs * (omc * f.c - eps2d * det_conic);
    }
    float vp0[3], vp1[3], vW[9];
    for (int k = 0; k < 3; ++k) {
      vp0[k] = 2.f * (G00 * f.p0[k] + G01 * f.p1[k]);
      vp1[k] = 2.f * (G01 * f.p0[k] + G11 * f.p1[k]);
    }
    const float vJ00 = vp0[0] * f.W[0] + vp0[1] * f.W[1] + vp0[2] * f.W[2];
    const float vJ02 = vp0[0] * f.W[6] + vp0[1] * f.W[7] + vp0[2] * f.W[8];
    const float vJ11 = vp1[0] * f.W[3] + vp1[1] * f.W[4] + vp1[2] * f.W[5];
    const float vJ12 = vp1[0] * f.W[6] + vp1[1] * f.W[7] + vp1[2] * f.W[8];
    for (int k = 0; k < 3; ++k) {
      vW[k] = f.J00 * vp0[k]; vW[3 + k] = f.J11 * vp1[k]; vW[6 + k] = f.J02 * vp0[k] + f.J12 * vp1[k];
    }
    const float rz3 = f.rz2 * f.rz;
    float vtx = fx * f.rz * vx, vty = fy * f.rz * vy;
    float vtz = -(fx * f.x * vx + fy * f.y * vy) * f.rz2 + (v_depths ? v_depths[g] : 0.f);
    vtz += -fx * f.rz2 * vJ00 - fy * f.rz2 * vJ11;
    if (f.in_x) { vtx += -fx * f.rz2 * vJ02; vtz += 2.f * fx * f.tx * rz3 * vJ02; } else vtz += fx * f.tx * rz3 * vJ02;
    if (f.in_y) { vty += -fy * f.rz2 * vJ12; vtz += 2.f * fy * f.ty * rz3 * vJ12; } else vtz += fy * f.ty * rz3 * vJ12;
    v_means[3 * g] = Rv[0] * vtx + Rv[3] * vty + Rv[6] * vtz;
    v_means[3 * g + 1] = Rv[1] * vtx + Rv[4] * vty + Rv[7] * vtz;
    v_means[3 * g + 2] = Rv[2] * vtx + Rv[5] * vty + Rv[8] * vtz;
    float vR[9];
    for (int k = 0; k < 3; ++k) {
      float vs = 0.f;
      for (int i = 0; i < 3; ++i) {
        const float vM = Rv[i] * vW[k] + Rv[3 + i] * vW[3 + k] + Rv[6 + i] * vW[6 + k];
        vs += f.R[3 * i + k] * vM;
        vR[3 * i + k] = vM * f.s[k];
      }
      v_scales[3 * g + k] = vs;
    }
    const float w = f.qw, x = f.qx, y = f.qy, z = f.qz;
    const float nw = 2.f * (x * (vR[7] - vR[5]) + y * (vR[2] - vR[6]) + z * (vR[3] - vR[1]));
    const float nx = 2.f * (-2.f * x * (vR[4] + vR[8]) + y * (vR[1] + vR[3]) + z * (vR[2] + vR[6]) + w * (vR[7] - vR[5]));
    const float ny = 2.f * (x * (vR[1] + vR[3]) - 2.f * y * (vR[0] + vR[8]) + z * (vR[5] + vR[7]) + w * (vR[2] - vR[6]));
    const float nz = 2.f * (x * (vR[2] + vR[6]) + y * (vR[5] + vR[7]) - 2.f * z * (vR[0] + vR[4]) + w * (vR[3] - vR[1]));
    const float d = nw * w + nx * x + ny * y + nz * z;
    v_quats[4 * g] = (nw - d * w) * f.qinv; v_quats[4 * g + 1] = (nx - d * x) * f.qinv;
    v_quats[4 * g + 2] = (ny - d * y) * f.qinv; v_quats[4 * g + 3] = (nz - d * z) * f.qinv;
  }
}

/* ---------------------------------------------------------------- one whole training step -------- */
/* train_gaussians.py:81-106 for one view with unit colours (edge_gs.py:247): returns the projection loss.
 * params: means[N,3], log_scales[N,3], quats[N,4], logit_opacities[N]; m, v: Adam moments in the same
 * order [means 3N | scales 3N | quats 4N | opac N]; lr[4] in that order; step is 1-based. */
double ego_train_step(float *means, float *log_scales, float *quats, float *logit_opac, float *m, float *v,
                      float *absgrads, int N, const float *vm, const float *K, int width, int height,
                      const float *gt, const float *wmap, const double *lr, int step, int64_t *M_out) {
  const int HW = width * height, tw = (width + TILE - 1) / TILE, th = (height + TILE - 1) / TILE, T = tw * th;
  float *scales = (float *)malloc(sizeof(float) * 3 * N), *opac = (float *)malloc(sizeof(float) * N);
  float *means2d = (float *)malloc(sizeof(float) * 2 * N), *depths = (float *)malloc(sizeof(float) * N);
  float *conics = (float *)malloc(sizeof(float) * 3 * N), *comps = (float *)malloc(sizeof(float) * N);
  float *oeff = (float *)malloc(sizeof(float) * N), *colors = (float *)malloc(sizeof(float) * N);
  int32_t *radii = (int32_t *)malloc(sizeof(int32_t) * N), *tpg = (int32_t *)malloc(sizeof(int32_t) * N);
#pragma omp parallel for
  for (int g = 0; g < N; ++g) {
    for (int k = 0; k < 3; ++k) scales[3 * g + k] = expf(log_scales[3 * g + k]);
    opac[g] = 1.f / (1.f + expf(-logit_opac[g]));
    colors[g] = 1.f;
  }
  ego_project_fwd(means, quats, scales, vm, K, N, width, height, 0.01f, 1e10f, 0.3f, 0.f, radii, means2d, depths, conics, comps);
#pragma omp parallel for
  for (int g = 0; g < N; ++g) oeff[g] = opac[g] * comps[g];
  const int64_t M = ego_isect_count(means2d, radii, N, width, height, tpg);
  int64_t *ids = (int64_t *)malloc(sizeof(int64_t) * (size_t)(M > 0 ? M : 1));
  int32_t *flat = (int32_t *)malloc(sizeof(int32_t) * (size_t)(M > 0 ? M : 1));
  int32_t *offsets = (int32_t *)malloc(sizeof(int32_t) * T);
  ego_isect_emit_sort(means2d, radii, depths, N, width, height, M, ids, flat, offsets);
  float *render = (float *)malloc(sizeof(float) * HW), *alphas = (float *)malloc(sizeof(float) * HW);
  float *v_render = (float *)malloc(sizeof(float) * HW);
  int32_t *last = (int32_t *)malloc(sizeof(int32_t) * HW);
  ego_composite_fwd(means2d, conics, colors, oeff, 1, width, height, offsets, flat, M, render, alphas, last);
  double loss = 0.0;
#pragma omp parallel for reduction(+ : loss)
  for (int p = 0; p < HW; ++p) {
    const float c0 = fclampf(render[p], 0.f, 1.f), d = c0 - gt[p];
    loss += (double)(wmap[p] * fabsf(d));
    const float pass = (render[p] >= 0.f && render[p] <= 1.f) ? 1.f : 0.f;
    v_render[p] = wmap[p] * ((d > 0.f) - (d < 0.f)) * pass;
  }
  float *v_m2d = (float *)calloc(2 * (size_t)N, 4), *v_abs = (float *)calloc(2 * (size_t)N, 4);
  float *v_con = (float *)calloc(3 * (size_t)N, 4), *v_oeff = (float *)calloc((size_t)N, 4);
  ego_composite_bwd(means2d, conics, colors, oeff, 1, width, height, offsets, flat, M, alphas, last, v_render, NULL,
                    v_m2d, v_abs, v_con, NULL, v_oeff);
  float *v_comp = (float *)malloc(sizeof(float) * N), *g_means = (float *)malloc(sizeof(float) * 3 * N);
  float *g_quats = (float *)malloc(sizeof(float) * 4 * N), *g_scales = (float *)malloc(sizeof(float) * 3 * N);
#pragma omp parallel for
  for (int g = 0; g < N; ++g) v_comp[g] = v_oeff[g] * opac[g];
  ego_project_bwd(means, quats, scales, vm, K, N, width, height, 0.3f, radii, v_m2d, NULL, v_con, v_comp, g_means,
                  g_quats, g_scales);
  /* chain to the raw parameters, absgrad, Adam (torch 1.13 order, scalars formed in double) */
  const double b1 = 0.9, b2 = 0.999, bc1 = 1.0 - pow(b1, step), bc2s = sqrt(1.0 - pow(b2, step));
  const float fb1 = (float)b1, fob1 = (float)(1.0 - b1), fb2 = (float)b2, fob2 = (float)(1.0 - b2), eps = 1e-8f;
  const float ss[4] = {(float)(lr[0] / bc1), (float)(lr[1] / bc1), (float)(lr[2] / bc1), (float)(lr[3] / bc1)};
  const float fbc2s = (float)bc2s;
#define ADAM(P, G, IDX, GRP)                                   \
  do {                                                         \
    float mm = m[IDX] * fb1 + (G)*fob1;                        \
    float vv = v[IDX] * fb2 + ((G) * (G)) * fob2;              \
    m[IDX] = mm; v[IDX] = vv;                                  \
    (P) = (P)-ss[GRP] * (mm / (sqrtf(vv) / fbc2s + eps));      \
  } while (0)
#pragma omp parallel for
  for (int g = 0; g < N; ++g) {
    absgrads[g] += sqrtf(v_abs[2 * g] * v_abs[2 * g] + v_abs[2 * g + 1] * v_abs[2 * g + 1]);
    for (int k = 0; k < 3; ++k) ADAM(means[3 * g + k], g_means[3 * g + k], 3 * (size_t)g + k, 0);
    for (int k = 0; k < 3; ++k)
      ADAM(log_scales[3 * g + k], g_scales[3 * g + k] * scales[3 * g + k], 3 * (size_t)N + 3 * (size_t)g + k, 1);
    for (int k = 0; k < 4; ++k) ADAM(quats[4 * g + k], g_quats[4 * g + k], 6 * (size_t)N + 4 * (size_t)g + k, 2);
    const float go = v_oeff[g] * comps[g] * opac[g] * (1.f - opac[g]);
    ADAM(logit_opac[g], go, 10 * (size_t)N + g, 3);
  }
#undef ADAM
  if (M_out) *M_out = M;
  free(scales); free(opac); free(means2d); free(depths); free(conics); free(comps); free(oeff); free(colors);
  free(radii); free(tpg); free(ids); free(flat); free(offsets); free(render); free(alphas); free(v_render);
  free(last); free(v_m2d); free(v_abs); free(v_con); free(v_oeff); free(v_comp); free(g_means); free(g_quats);
  free(g_scales);
  return loss;
}

/* ---------------------------------------------------------------- float-borderline analysis -------
 * The path branches on float comparisons (radius = ceil(3 sqrt(lambda)), the near / on-screen culls,
 * alpha >= 1/255, alpha raw <= 0.999, next_T <= 1e-4).  Two correct fp32 implementations with a
 * different exp or a different operation order may take different branches where the compared value sits
 * within rounding distance of its threshold; everywhere else they must agree to the float tolerance.
 * These two routines evaluate the SAME formulas in double precision on the same fp32 inputs and mark
 * what lies within a stated relative margin of a threshold, so that the parity tests can (a) take such
 * Gaussians out of the scene and (b) give such pixels zero loss weight -- and then assert the tolerance
 * on EVERY remaining element, instead of admitting an unexplained fraction of outliers. */
void ego_project_borderline(const float *means, const float *quats, const float *scales, const float *vm,
                            const float *K, int N, int width, int height, double near_plane, double eps2d,
                            double rel, uint8_t *mask /*[N] out*/) {
  const double Rv[9] = {vm[0], vm[1], vm[2], vm[4], vm[5], vm[6], vm[8], vm[9], vm[10]};
  const double tv[3] = {vm[3], vm[7], vm[11]};
  const double fx = K[0], cx = K[2], fy = K[4], cy = K[5];
  const int tw = (width + TILE - 1) / TILE, th = (height + TILE - 1) / TILE;
#pragma omp parallel for schedule(static)
  for (int g = 0; g < N; ++g) {
    const float *mu = means + 3 * g, *q = quats + 4 * g, *sc = scales + 3 * g;
    uint8_t flag = 0;
    const double x = Rv[0] * mu[0] + Rv[1] * mu[1] + Rv[2] * mu[2] + tv[0];
    const double y = Rv[3] * mu[0] + Rv[4] * mu[1] + Rv[5] * mu[2] + tv[1];
    const double z = Rv[6] * mu[0] + Rv[7] * mu[1] + Rv[8] * mu[2] + tv[2];
    if (fabs(z - near_plane) <= rel * near_plane) flag = 1;
    if (z < near_plane) { mask[g] = flag; continue; }
    double w = q[0], qx = q[1], qy = q[2], qz = q[3];
    const double qi = 1.0 / sqrt(w * w + qx * qx + qy * qy + qz * qz);
    w *= qi; qx *= qi; qy *= qi; qz *= qi;
    const double R[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - w * qz), 2 * (qx * qz + w * qy),
                         2 * (qx * qy + w * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - w * qx),
                         2 * (qx * qz - w * qy), 2 * (qy * qz + w * qx), 1 - 2 * (qx * qx + qy * qy)};
    double W[9];
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < 3; ++k) W[3 * i + k] = (Rv[3 * i] * R[k] + Rv[3 * i + 1] * R[3 + k] + Rv[3 * i + 2] * R[6 + k]) * sc[k];
    const double lim_x = 1.3 * (0.5 * width / fx), lim_y = 1.3 * (0.5 * height / fy);
    const double rz = 1.0 / z, xr = x * rz, yr = y * rz;
    if (fabs(fabs(xr) - lim_x) <= rel * lim_x || fabs(fabs(yr) - lim_y) <= rel * lim_y) flag = 1; /* fov-clamp gate */
    const double tx = z * fmin(lim_x, fmax(-lim_x, xr)), ty = z * fmin(lim_y, fmax(-lim_y, yr));
    const double J00 = fx * rz, J11 = fy * rz, J02 = -fx * tx * rz * rz, J12 = -fy * ty * rz * rz;
    double p0[3], p1[3];
    for (int k = 0; k < 3; ++k) { p0[k] = J00 * W[k] + J02 * W[6 + k]; p1[k] = J11 * W[3 + k] + J12 * W[6 + k]; }
    const double c00 = p0[0] * p0[0] + p0[1] * p0[1] + p0[2] * p0[2], c01 = p0[0] * p1[0] + p0[1] * p1[1] + p0[2] * p1[2];
    const double c11 = p1[0] * p1[0] + p1[1] * p1[1] + p1[2] * p1[2];
    const double b00 = c00 + eps2d, b11 = c11 + eps2d, det1 = b00 * b11 - c01 * c01;
    if (det1 <= 0.0) { mask[g] = 1; continue; }
    const double bh = 0.5 * (b00 + b11), disc = bh * bh - det1;
    if (fabs(disc - 0.01) <= rel * 0.01) flag = 1; /* max(0.01, .) switches: its derivative does */
    const double v = 3.0 * sqrt(bh + sqrt(fmax(0.01, disc)));
    if (fabs(v - floor(v + 0.5)) <= rel * v) flag = 1; /* ceil() is about to flip */
    const double r = ceil(v), u = fx * x * rz + cx, vv = fy * y * rz + cy;
    const double edges[4] = {u + r, u - r - width, vv + r, vv - r - height};
    for (int e = 0; e < 4; ++e)
      if (fabs(edges[e]) <= rel * (r + width + height)) flag = 1; /* on-screen cull */
    /* tile box: floor / ceil of (c -+ r) / 16 about to flip (only matters inside the grid) */
    const double tb[4] = {(u - r) / TILE, (u + r) / TILE, (vv - r) / TILE, (vv + r) / TILE};
    const double hi[4] = {(double)tw, (double)tw, (double)th, (double)th};
    for (int e = 0; e < 4; ++e)
      /* (the fp32 error of (c -+ r) / 16 is a few ulps of c: a tenth of the margin the radius gets) */
      if (tb[e] > -0.5 && tb[e] < hi[e] + 0.5 && fabs(tb[e] - floor(tb[e] + 0.5)) <= 0.1 * rel * (fabs(tb[e]) + 1.0)) flag = 1;
    mask[g] = flag;
  }
}

/* per pixel: does its front-to-back walk (ego_composite_fwd, in double) pass within the margins of a
 * threshold?  bit 0: alpha vs 1/255 or raw alpha vs 0.999 or sigma vs 0 (rel_alpha); bit 1: next_T vs
 * 1e-4 (rel_T).  The walk goes on past a borderline stop, so a flagged pixel is flagged whichever way
 * an fp32 implementation decides. */
void ego_borderline_pixels(const float *means2d, const float *conics, const float *opac, int width, int height,
                           const int32_t *offsets, const int32_t *flatten_ids, int64_t M, double rel_alpha,
                           double rel_T, uint8_t *mask /*[H*W] out*/) {
  const int tw = (width + TILE - 1) / TILE, th = (height + TILE - 1) / TILE, T = tw * th;
  const double amin = 1.0 / 255.0, amax = 0.999, tstop = 1e-4;
#pragma omp parallel for schedule(dynamic, 1)
  for (int t = 0; t < T; ++t) {
    const int64_t start = offsets[t], end = (t == T - 1) ? M : offsets[t + 1];
    const int ty = t / tw, tx = t % tw;
    for (int di = 0; di < TILE; ++di)
      for (int dj = 0; dj < TILE; ++dj) {
        const int i = ty * TILE + di, j = tx * TILE + dj;
        if (i >= height || j >= width) continue;
        const double px = (double)j + 0.5, py = (double)i + 0.5;
        double Tt = 1.0;
        uint8_t flag = 0;
        for (int64_t idx = start; idx < end; ++idx) {
          const int g = flatten_ids[idx];
          const double dx = (double)means2d[2 * g] - px, dy = (double)means2d[2 * g + 1] - py;
          const double a = conics[3 * g], b = conics[3 * g + 1], c = conics[3 * g + 2];
          const double sigma = 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy;
          const double araw = (double)opac[g] * exp(-sigma);
          if (fabs(araw - amin) <= rel_alpha * amin || fabs(araw - amax) <= rel_alpha * amax || sigma < 1e-6) flag |= 1;
          const double alpha = fmin(amax, araw);
          if (sigma < 0.0 || alpha < amin) continue;
          const double next_T = Tt * (1.0 - alpha);
          if (fabs(next_T - tstop) <= rel_T * tstop) flag |= 2;
          if (next_T <= tstop * (1.0 - rel_T)) break; /* stopped beyond doubt */
          if (next_T <= tstop) continue;              /* borderline stop: look at what follows as well */
          Tt = next_T;
        }
        mask[(size_t)i * width + j] = flag;
      }
  }
}

int ego_num_threads(void) {
  int n = 1;
#ifdef _OPENMP
#pragma omp parallel
  {
#pragma omp single
    n = omp_get_num_threads();
  }
#endif
  return n;
}
