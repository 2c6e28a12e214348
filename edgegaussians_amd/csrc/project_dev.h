// Device code of the projection (G1 / G9) shared by project.hip and backward_fused.hip: camera, the forward
// geometry of one Gaussian, its reverse sweep, Adam.  (Moved out of project.hip in round 6, unchanged: every kernel
// that inlines forward_geom forms the SAME rounding sequence -- see the comment inside it.)
#pragma once
#include <math.h>

#include "common.h"

namespace eg {

struct Cam {
  float R[9], t[3], fx, fy, cx, cy;
};

__device__ __forceinline__ Cam load_cam(const float *__restrict__ vm, const float *__restrict__ K) {
  Cam c;
  c.R[0] = vm[0]; c.R[1] = vm[1]; c.R[2] = vm[2];  c.t[0] = vm[3];
  c.R[3] = vm[4]; c.R[4] = vm[5]; c.R[5] = vm[6];  c.t[1] = vm[7];
  c.R[6] = vm[8]; c.R[7] = vm[9]; c.R[8] = vm[10]; c.t[2] = vm[11];
  c.fx = K[0]; c.cx = K[2]; c.fy = K[4]; c.cy = K[5];
  return c;
}

// Everything the forward computes for one visible Gaussian; the backward re-derives it instead of
// reading it back from memory (recompute is ~100 flops, a reload would be ~100 B).
struct Fwd {
  float x, y, z;            // camera-space mean
  float qw, qx, qy, qz, qinv;  // normalised quaternion and 1/|q|
  float s[3];               // scales (after exp)
  float o;                  // opacity (after sigmoid)
  float Rq[9];              // rotation of the Gaussian
  float W[9];               // Rv * Rq * diag(s)
  float rz, rz2, tx, ty;    // 1/z, 1/z^2, clamped x, y
  bool in_x, in_y;          // fov clamp inactive
  float J00, J02, J11, J12;
  float p0[3], p1[3];
  float c00, c01, c11, b00, b11, det0, det1;
  float a, b, c, comp;      // conic, compensation
  float u, v;               // mean2d
};

// the raw parameters of one Gaussian (as stored: log-scales / logit-opacity when the flags say so)
struct Raw {
  float m[3], q[4], s[3], o;
};
__device__ __forceinline__ Raw load_raw(const float *__restrict__ means, const float *__restrict__ quats,
                                        const float *__restrict__ scales, const float *__restrict__ opacities, int g) {
  Raw r;
#pragma unroll
  for (int k = 0; k < 3; ++k) { r.m[k] = means[3 * g + k]; r.s[k] = scales[3 * g + k]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) r.q[k] = quats[4 * g + k];
  r.o = opacities[g];
  return r;
}

__device__ __forceinline__ bool forward_geom(const Cam &cam, const Raw &raw, int width, int height,
                                             float near_plane, float far_plane, float eps2d, uint32_t flags,
                                             Fwd &f) {
  // No FMA contraction in here (nor in the tile tests of common.h): this function is inlined into four kernels
  // (project_fwd, project_emit, the projection backward's recomputation, the tail-fused next-view projection), and
  // the compiler's contraction choices differ from one inlining context to the next -- the packed records of the
  // operator path and of the training step then differ in the last bit, and with them a float-borderline tile hit
  // or alpha threshold here and there (1600 x 1200, 200 k Gaussians: one Gaussian's gradient off by 4e-4 of the
  // maximum between the two paths).  Separate multiplies and adds are also what the C oracle (gcc, x86-64) does.
#pragma clang fp contract(off)
  const float mx = raw.m[0], my = raw.m[1], mz = raw.m[2];
  f.x = cam.R[0] * mx + cam.R[1] * my + cam.R[2] * mz + cam.t[0];
  f.y = cam.R[3] * mx + cam.R[4] * my + cam.R[5] * mz + cam.t[1];
  f.z = cam.R[6] * mx + cam.R[7] * my + cam.R[8] * mz + cam.t[2];
  if (f.z < near_plane || f.z > far_plane) return false;

  float w = raw.q[0], x = raw.q[1], y = raw.q[2], z = raw.q[3];
  f.qinv = rsqrtf(w * w + x * x + y * y + z * z);
  w *= f.qinv; x *= f.qinv; y *= f.qinv; z *= f.qinv;
  f.qw = w; f.qx = x; f.qy = y; f.qz = z;
  const float x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z;
  const float wx = w * x, wy = w * y, wz = w * z;
  f.Rq[0] = 1.f - 2.f * (y2 + z2); f.Rq[1] = 2.f * (xy - wz);       f.Rq[2] = 2.f * (xz + wy);
  f.Rq[3] = 2.f * (xy + wz);       f.Rq[4] = 1.f - 2.f * (x2 + z2); f.Rq[5] = 2.f * (yz - wx);
  f.Rq[6] = 2.f * (xz - wy);       f.Rq[7] = 2.f * (yz + wx);       f.Rq[8] = 1.f - 2.f * (x2 + y2);

#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float sv = raw.s[k];
    f.s[k] = (flags & EG_FLAG_LOG_SCALES) ? expf(sv) : sv;
  }
  const float ov = raw.o;
  f.o = (flags & EG_FLAG_LOGIT_OPACITIES) ? 1.f / (1.f + expf(-ov)) : ov;

  // W = Rv * (Rq * diag(s))
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k)
      f.W[3 * i + k] =
          (cam.R[3 * i] * f.Rq[k] + cam.R[3 * i + 1] * f.Rq[3 + k] + cam.R[3 * i + 2] * f.Rq[6 + k]) * f.s[k];

  const float lim_x = kFovClamp * (0.5f * (float)width / cam.fx);
  const float lim_y = kFovClamp * (0.5f * (float)height / cam.fy);
  f.rz = 1.f / f.z;
  f.rz2 = f.rz * f.rz;
  const float xr = f.x * f.rz, yr = f.y * f.rz;
  f.in_x = (xr <= lim_x) && (xr >= -lim_x);
  f.in_y = (yr <= lim_y) && (yr >= -lim_y);
  f.tx = f.z * fminf(lim_x, fmaxf(-lim_x, xr));
  f.ty = f.z * fminf(lim_y, fmaxf(-lim_y, yr));
  f.J00 = cam.fx * f.rz;
  f.J11 = cam.fy * f.rz;
  f.J02 = -cam.fx * f.tx * f.rz2;
  f.J12 = -cam.fy * f.ty * f.rz2;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    f.p0[k] = f.J00 * f.W[k] + f.J02 * f.W[6 + k];
    f.p1[k] = f.J11 * f.W[3 + k] + f.J12 * f.W[6 + k];
  }
  f.c00 = f.p0[0] * f.p0[0] + f.p0[1] * f.p0[1] + f.p0[2] * f.p0[2];
  f.c01 = f.p0[0] * f.p1[0] + f.p0[1] * f.p1[1] + f.p0[2] * f.p1[2];
  f.c11 = f.p1[0] * f.p1[0] + f.p1[1] * f.p1[1] + f.p1[2] * f.p1[2];
  f.u = cam.fx * f.x * f.rz + cam.cx;
  f.v = cam.fy * f.y * f.rz + cam.cy;

  f.det0 = f.c00 * f.c11 - f.c01 * f.c01;
  f.b00 = f.c00 + eps2d;
  f.b11 = f.c11 + eps2d;
  f.det1 = f.b00 * f.b11 - f.c01 * f.c01;
  if (f.det1 <= 0.f) return false;
  f.comp = sqrtf(fmaxf(0.f, f.det0 / f.det1));
  const float inv = 1.f / f.det1;
  f.a = f.b11 * inv;
  f.b = -f.c01 * inv;
  f.c = f.b00 * inv;
  return true;
}

__device__ __forceinline__ bool forward_geom(const Cam &cam, const float *__restrict__ means,
                                             const float *__restrict__ quats, const float *__restrict__ scales,
                                             const float *__restrict__ opacities, int g, int width, int height,
                                             float near_plane, float far_plane, float eps2d, uint32_t flags,
                                             Fwd &f) {
  return forward_geom(cam, load_raw(means, quats, scales, opacities, g), width, height, near_plane, far_plane, eps2d,
                      flags, f);
}

__device__ __forceinline__ int radius_of(const Fwd &f, int width, int height, float radius_clip) {
#pragma clang fp contract(off)
  const float bh = 0.5f * (f.b00 + f.b11);
  const float v1 = bh + sqrtf(fmaxf(0.01f, bh * bh - f.det1));
  const float radius = ceilf(3.f * sqrtf(v1));
  if (radius <= radius_clip) return 0;
  if (f.u + radius <= 0.f || f.u - radius >= (float)width || f.v + radius <= 0.f ||
      f.v - radius >= (float)height)
    return 0;
  return (int)radius;
}

// ---------------------------------------------------------------------------------------------
struct AdamK {
  float step_size[4];    // lr_k / (1 - beta1^t_k), means | scales | quats | opacities
  float inv_bc2_sqrt[4]; // 1 / sqrt(1 - beta2^t_k)   (each optimizer has its own step count t_k)
  int active[4];         // 0 = this optimizer does not step in this call
  float b1, omb1, b2, omb2, eps;
};

// torch.optim.Adam's update (train_utils.py:50-60: no weight decay, no amsgrad), in torch's order:
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= step_size * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// The square root and the quotient use the hardware's 1-ulp v_sqrt_f32 / v_rcp_f32 (and the bias correction is a
// product with the reciprocal formed in double on the host): the correctly rounded forms expand to ~11 instructions
// each -- 33 expansions per Gaussian, a seventh of the projection-backward kernel's dependent instruction stream --
// and the difference, <= 3 ulp of one update, is four orders of magnitude inside the 1e-4 the path is held to.
// (sqrt of a denormal v reads as 0: the denominator is then eps, as it is to within 1e-11 with the exact root.)
__device__ __forceinline__ void adam1(float &p, float g, float &m, float &v, int grp, const AdamK &h) {
#pragma clang fp contract(off)  // (one rounding sequence in every kernel: see forward_geom)
  if (!h.active[grp]) return;
  m = m * h.b1 + g * h.omb1;
  v = v * h.b2 + (g * g) * h.omb2;
  const float denom = __builtin_amdgcn_sqrtf(v) * h.inv_bc2_sqrt[grp] + h.eps;
  p = p - h.step_size[grp] * (m * __builtin_amdgcn_rcpf(denom));
}

struct Grads {
  float mean[3], quat[4], scale[3], opac;
};

__device__ __forceinline__ void backward_geom(const Cam &cam, const Fwd &f, float eps2d, uint32_t flags,
                                              const float4 ga, const float4 gb, bool ext, float v_comp_ext,
                                              float v_depth_ext, Grads &o) {
#pragma clang fp contract(off)  // (one rounding sequence in every kernel: see forward_geom)
  const float vx = ga.x, vy = ga.y;
  const float va = gb.x, vb = gb.y, vc = gb.z, vo_eff = gb.w;
  const bool aa = flags & EG_FLAG_ANTIALIASED;

  // o_eff = o * comp (fused mode); in external mode the caller owns that product (gsplat layout:
  // `opacities * compensations` is a torch op between the two autograd nodes)
  float v_o = ext ? 0.f : (aa ? vo_eff * f.comp : vo_eff);
  const float v_comp = ext ? v_comp_ext : (aa ? vo_eff * f.o : 0.f);

  // conic = B^-1  =>  G = -A V A with V = [[va, vb/2],[vb/2, vc]] (b is stored once)
  const float hb = 0.5f * vb;
  const float av00 = f.a * va + f.b * hb, av01 = f.a * hb + f.b * vc;
  const float av10 = f.b * va + f.c * hb, av11 = f.b * hb + f.c * vc;
  float G00 = -(av00 * f.a + av01 * f.b);
  float G01 = -(av00 * f.b + av01 * f.c);
  float G11 = -(av10 * f.b + av11 * f.c);
  if (aa) {
    // d comp / d cov2d = (1/(2 comp)) * ((1 - comp^2) conic - eps det(conic) I); gsplat guards the
    // division with +1e-6 and the oracle does the same
    const float det_conic = f.a * f.c - f.b * f.b;
    const float vs = v_comp * 0.5f / (f.comp + 1e-6f);
    const float omc = 1.f - f.comp * f.comp;
    G00 += vs * (omc * f.a - eps2d * det_conic);
    G01 += vs * (omc * f.b);
    G11 += vs * (omc * f.c - eps2d * det_conic);
  }
  // cov2d = P P^T  =>  v_P = 2 G P
  float vp0[3], vp1[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    vp0[k] = 2.f * (G00 * f.p0[k] + G01 * f.p1[k]);
    vp1[k] = 2.f * (G01 * f.p0[k] + G11 * f.p1[k]);
  }
  // P = J W
  const float vJ00 = vp0[0] * f.W[0] + vp0[1] * f.W[1] + vp0[2] * f.W[2];
  const float vJ02 = vp0[0] * f.W[6] + vp0[1] * f.W[7] + vp0[2] * f.W[8];
  const float vJ11 = vp1[0] * f.W[3] + vp1[1] * f.W[4] + vp1[2] * f.W[5];
  const float vJ12 = vp1[0] * f.W[6] + vp1[1] * f.W[7] + vp1[2] * f.W[8];
  float vW[9];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    vW[k] = f.J00 * vp0[k];
    vW[3 + k] = f.J11 * vp1[k];
    vW[6 + k] = f.J02 * vp0[k] + f.J12 * vp1[k];
  }
  // camera-space mean: through mean2d and through J (fov clamp freezes tx = +-lim z)
  const float rz3 = f.rz2 * f.rz;
  float vtx = cam.fx * f.rz * vx;
  float vty = cam.fy * f.rz * vy;
  float vtz = -(cam.fx * f.x * vx + cam.fy * f.y * vy) * f.rz2 + v_depth_ext;
  vtz += -cam.fx * f.rz2 * vJ00 - cam.fy * f.rz2 * vJ11;
  if (f.in_x) { vtx += -cam.fx * f.rz2 * vJ02; vtz += 2.f * cam.fx * f.tx * rz3 * vJ02; }
  else        { vtz += cam.fx * f.tx * rz3 * vJ02; }
  if (f.in_y) { vty += -cam.fy * f.rz2 * vJ12; vtz += 2.f * cam.fy * f.ty * rz3 * vJ12; }
  else        { vtz += cam.fy * f.ty * rz3 * vJ12; }
  o.mean[0] = cam.R[0] * vtx + cam.R[3] * vty + cam.R[6] * vtz;
  o.mean[1] = cam.R[1] * vtx + cam.R[4] * vty + cam.R[7] * vtz;
  o.mean[2] = cam.R[2] * vtx + cam.R[5] * vty + cam.R[8] * vtz;

  // W = Rv M, M = Rq diag(s)
  float vR[9];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float vs = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float vM = cam.R[i] * vW[k] + cam.R[3 + i] * vW[3 + k] + cam.R[6 + i] * vW[6 + k];
      vs += f.Rq[3 * i + k] * vM;
      vR[3 * i + k] = vM * f.s[k];
    }
    o.scale[k] = (flags & EG_FLAG_LOG_SCALES) ? vs * f.s[k] : vs;
  }
  // rotation -> normalised quaternion -> raw quaternion
  const float w = f.qw, x = f.qx, y = f.qy, z = f.qz;
  const float nw = 2.f * (x * (vR[7] - vR[5]) + y * (vR[2] - vR[6]) + z * (vR[3] - vR[1]));
  const float nx = 2.f * (-2.f * x * (vR[4] + vR[8]) + y * (vR[1] + vR[3]) + z * (vR[2] + vR[6]) + w * (vR[7] - vR[5]));
  const float ny = 2.f * (x * (vR[1] + vR[3]) - 2.f * y * (vR[0] + vR[8]) + z * (vR[5] + vR[7]) + w * (vR[2] - vR[6]));
  const float nz = 2.f * (x * (vR[2] + vR[6]) + y * (vR[5] + vR[7]) - 2.f * z * (vR[0] + vR[4]) + w * (vR[3] - vR[1]));
  const float d = nw * w + nx * x + ny * y + nz * z;
  o.quat[0] = (nw - d * w) * f.qinv;
  o.quat[1] = (nx - d * x) * f.qinv;
  o.quat[2] = (ny - d * y) * f.qinv;
  o.quat[3] = (nz - d * z) * f.qinv;

  if (flags & EG_FLAG_LOGIT_OPACITIES) v_o *= f.o * (1.f - f.o);
  o.opac = v_o;
}

inline AdamK make_adamk(const eg_adam_hyper &h) {
  // host side in double, exactly how torch.optim.Adam forms its scalars before casting to fp32
  AdamK k;
  const double b1 = h.beta1, b2 = h.beta2;
  const double lrs[4] = {h.lr_means, h.lr_scales, h.lr_quats, h.lr_opacities};
  for (int i = 0; i < 4; ++i) {
    const int t = h.group_steps[i] == 0 ? h.step : h.group_steps[i];  // 0: shared count, < 0: skip
    k.active[i] = t > 0;
    const double tt = t > 0 ? (double)t : 1.0;
    k.step_size[i] = (float)(lrs[i] / (1.0 - pow(b1, tt)));
    k.inv_bc2_sqrt[i] = (float)(1.0 / sqrt(1.0 - pow(b2, tt)));
  }
  k.b1 = (float)b1; k.omb1 = (float)(1.0 - b1);
  k.b2 = (float)b2; k.omb2 = (float)(1.0 - b2);
  k.eps = (float)h.eps;
  return k;
}

}  // namespace eg
