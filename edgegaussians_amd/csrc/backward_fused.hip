// G8 + G9 + Adam + the next view's G1 / G2-4 in ONE kernel (round 6): the per-Gaussian backward of the training step.
//
// Replaces, inside a native run of steps on tile grids of <= 2560 tiles and for scenes of <= 32768 Gaussians, the pair
//   footprint_bwd_kernel      (composite.hip: gsplat's rasterize_to_pixels_bwd, order-independent form)
//   project_bwd_emit_kernel   (project.hip: fully_fused_projection_bwd + update_absgrads + 4x Adam.step -- SURVEY a3.G9,
//                              a6, a7; edge_gs.py:250-268, :607-613, train_gaussians.py:104-106 -- and the NEXT view's
//                              projection + exact tile binning, a3.G1-G4)
// which are two dependent passes over the Gaussians: the second needs nothing of a Gaussian but its own g2d record.
//
// Why one kernel.  The second pass is one thread per Gaussian -- 1.5 waves per SIMD at 100 k Gaussians, every wave a
// chain of loads -> ~2000 dependent instructions -> LDS histogram -> returning atomics -> key stores that nothing hides
// (16.6 us at config 2 for 3.4 M wave instructions: 16 % of the issue peak).  Lane-splitting a Gaussian's chain over the
// 8 lanes that walk its footprint would multiply the ISSUED instructions by the idle lanes (the chain parallelises ~3x,
// not 8x), and the footprint walk is issue-bound: the chain stays one lane per Gaussian, and what changes is WHEN it runs.
// A workgroup here is 8 waves = 64 Gaussians.  Phase 1: every wave walks the footprints of its 8 Gaussians
// (footprint_dev.h, unchanged) and leaves their g2d records in LDS.  One barrier.  Phase 2: seven waves leave, the first
// one runs the projection backward, Adam and the next view's projection + binning of the workgroup's 64 Gaussians, one
// per lane, while the OTHER workgroups' footprint walks fill the SIMD's issue slots: the chain is exposed once, at the
// launch's tail, instead of in a launch of its own; the g2d record never leaves the CU; one kernel boundary is gone.
//
// Binning by one wave.  project_emit's workgroup (512 Gaussians) sweeps an LDS histogram of all T tiles; 64 Gaussians in
// Morton order touch a dozen tiles, so this kernel keeps a TOUCHED LIST next to the (zero-initialised) histogram: the
// first hit of a tile appends it, the wave reserves slots with one returning global atomic per LISTED tile and hands them
// out from LDS.  Same keys, same segments; the order inside a segment is as unspecified as before (the sort fixes it).
//
// Where it pays (profiles/r06_fused_backward_ab.txt): phase 2 needs 104 VGPRs, phase 1 sixty -- built for 8 waves per SIMD phase 2
// spills 49 registers (14.6 us per wave instead of 9), built for 4 the walks lose half their occupancy; so the kernel is built
// for 4 and taken where ONE round of workgroups covers the scene (2 per CU x 256 CUs x 64 Gaussians = 32768): 40.1 -> 36.8 us
// per step at 30 k Gaussians, 42.7 -> 44.2 at 40 k, 77.6 -> 81.9 at 100 k (step.hip: fused_backward_pays).  A workgroup keeps
// its LDS through phase 2, so phase 2 REUSES phase 1's storage (13.3 KB per workgroup; 32 KB measured 45 us at 100 k).
//
// Results: phase 1 and phase 2 inline the very functions the two kernels inline (forward_geom, backward_geom, adam1,
// footprint_walk ...), every Gaussian's arithmetic is the same sequence: a step through this kernel is bit-identical to
// a step through the pair (tests/test_gpu_parity.py::test_fused_backward_kernel_equals_the_two_kernel_path).
#include "common.h"
#include "footprint_dev.h"
#include "project_dev.h"

namespace eg {

#ifndef EG_BF_WAVES_PER_EU
#define EG_BF_WAVES_PER_EU 4
#endif
#ifndef EG_BF_HOIST
#define EG_BF_HOIST 0  // (1: phase 2's loads requested before the walks -- measured 17.8 against 17.5 us at config 1, profiles/r06_misc_ab.txt)
#endif
constexpr int kBFWaves = 8;                 // waves per workgroup
constexpr int kBFGauss = 8 * kBFWaves;      // Gaussians per workgroup = lanes of the phase-2 wave
constexpr int kBFTouched = 512;             // touched-tile list (beyond it: the wave sweeps the whole histogram)
static_assert(kBFTouched + 1 <= 8 * kBFWaves * 12, "the touched list reuses the footprint sizes' storage");

struct FusedBwdArgs {
  float *means, *quats, *scales, *opacities;      // raw parameters (log-scales, logit-opacities)
  const float *viewmat, *K, *next_viewmat, *next_K;
  int N, width, height;
  float eps2d;
  uint32_t flags;
  float4 *splat;                                  // [N,2]: view k's record in, the next view's out
  const StopRec *gtstop;
  float *g2d;                                     // optional [N,8]: the record is also written out (inspection / tests)
  float *absgrads, *am, *av;
  AdamK hyper;
  int *cursor; int seg_cap; unsigned long long *keys;
  float *loss_part, *loss_out;                    // the forward's 64 partial loss sums -> the caller's accumulator
};

// Phase 2 of the fused kernel: projection VJP + absgrads + Adam of Gaussian g of
// view k, then the projection + exact tile binning of the NEXT view with the updated parameters still in registers; one
// Gaussian per lane of ONE wave, no workgroup barrier.  raw / ag0 / mm / vv: the Gaussian's parameters, absgrad accumulator
// and Adam moments as loaded by the caller (EG_BF_HOIST) or to be loaded here; ga / gb: its g2d record.  s_hist [T] zeroed,
// s_base [T], s_touched [kBFTouched], s_ntouched zeroed: the wave's LDS.
struct TailLds {
  int *hist, *base, *touched, *ntouched;
};
#ifdef EG_BF_PROF
#define EG_BF_STAMP2(k_)                                                           \
  do {                                                                             \
    if (prof_rec) {                                                                \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                  \
      const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();              \
      if (lane == 0) prof_rec[k_] = t_;                                            \
    }                                                                              \
  } while (0)
#else
#define EG_BF_STAMP2(k_) do {} while (0)
#endif
__device__ __forceinline__ void gaussian_tail(const FusedBwdArgs &a, const int g, const int lane, Raw raw, const float ag0,
                                              float (&mm)[11], float (&vv)[11], const float4 ga, const float4 gb,
                                              const TailLds lds, unsigned long long *prof_rec) {
  const int width = a.width, height = a.height, N = a.N;
  const int tw = (width + kTile - 1) / kTile, th = (height + kTile - 1) / kTile, T = tw * th;
  const float4 *splat = a.splat;
  int *s_hist = lds.hist, *s_base = lds.base, *s_touched = lds.touched;
  int &s_ntouched = *lds.ntouched;
  const bool alive = g < N;
  if (alive) {
#if !EG_BF_HOIST
    raw = load_raw(a.means, a.quats, a.scales, a.opacities, g);
#endif
    const int radius = __float_as_int(splat[2 * g + 1].w);
    const size_t oM = 0, oS = 3 * (size_t)N, oQ = 6 * (size_t)N, oO = 10 * (size_t)N;
    const Cam cam = load_cam(a.viewmat, a.K);
    Grads gr;
#pragma unroll
    for (int k = 0; k < 3; ++k) gr.mean[k] = gr.scale[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) gr.quat[k] = 0.f;
    gr.opac = 0.f;
    if (radius > 0) {
#if EG_BF_HOIST
      const float ag = ag0;
#else
      const float ag = a.absgrads ? a.absgrads[g] : 0.f;
#endif
      Fwd f;
      forward_geom(cam, raw, width, height, -3.0e38f, 3.0e38f, a.eps2d, a.flags, f);
      backward_geom(cam, f, a.eps2d, a.flags, ga, gb, false, 0.f, 0.f, gr);
      if (a.absgrads) a.absgrads[g] = ag + sqrtf(ga.z * ga.z + ga.w * ga.w);
    }
    EG_BF_STAMP2(3);  // parameters loaded, forward recomputed, projection VJP
#if !EG_BF_HOIST
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      mm[k] = a.am[oM + 3 * g + k]; vv[k] = a.av[oM + 3 * g + k];
      mm[3 + k] = a.am[oS + 3 * g + k]; vv[3 + k] = a.av[oS + 3 * g + k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { mm[6 + k] = a.am[oQ + 4 * g + k]; vv[6 + k] = a.av[oQ + 4 * g + k]; }
    mm[10] = a.am[oO + g]; vv[10] = a.av[oO + g];
#endif
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      adam1(raw.m[k], gr.mean[k], mm[k], vv[k], 0, a.hyper);
      a.means[3 * g + k] = raw.m[k]; a.am[oM + 3 * g + k] = mm[k]; a.av[oM + 3 * g + k] = vv[k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      adam1(raw.s[k], gr.scale[k], mm[3 + k], vv[3 + k], 1, a.hyper);
      a.scales[3 * g + k] = raw.s[k]; a.am[oS + 3 * g + k] = mm[3 + k]; a.av[oS + 3 * g + k] = vv[3 + k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      adam1(raw.q[k], gr.quat[k], mm[6 + k], vv[6 + k], 2, a.hyper);
      a.quats[4 * g + k] = raw.q[k]; a.am[oQ + 4 * g + k] = mm[6 + k]; a.av[oQ + 4 * g + k] = vv[6 + k];
    }
    adam1(raw.o, gr.opac, mm[10], vv[10], 3, a.hyper);
    a.opacities[g] = raw.o; a.am[oO + g] = mm[10]; a.av[oO + g] = vv[10];
  }

  EG_BF_STAMP2(4);  // moments loaded, Adam, parameters and moments stored
  // the next view with the updated parameters (emit_body of project.hip, by ONE wave)
  const Cam ncam = load_cam(a.next_viewmat, a.next_K);
  Fwd f;
  int radius = 0;
  if (alive && forward_geom(ncam, raw, width, height, 0.01f, 1e10f, 0.3f, a.flags, f))
    radius = radius_of(f, width, height, 0.f);
  const bool aa = a.flags & EG_FLAG_ANTIALIASED;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  if (radius > 0) {
    s0 = make_float4(f.u, f.v, f.a, f.b);
    s1 = make_float4(f.c, aa ? f.o * f.comp : f.o, f.z, __int_as_float(radius));
  }
  if (alive) {
    a.splat[2 * g] = s0;
    a.splat[2 * g + 1] = s1;
  }
  int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
  if (radius > 0) tile_box_tight(s0.x, s0.y, radius, s0.z, s0.w, s1.x, s1.y, tw, th, x0, y0, x1, y1);
  const int bw = x1 - x0;
  const bool small_box = (y1 - y0) * bw <= 32;
  unsigned mask = 0u;
  {
    int bit = 0;
    for (int ty = y0; ty < y1; ++ty)
      for (int tx = x0; tx < x1; ++tx, ++bit) {
        if (!splat_hits_tile(s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, tx, ty)) continue;
        if (small_box) mask |= 1u << bit;
        const int t = ty * tw + tx;
        if (atomicAdd(&s_hist[t], 1) == 0) {  // the tile's first hit in this workgroup: list it
          const int idx = atomicAdd(&s_ntouched, 1);
          if (idx < kBFTouched) s_touched[idx] = t;
        }
      }
  }
  EG_BF_STAMP2(5);  // next view projected, exact tile tests, LDS histogram
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // slots [base, base + c) of every touched tile's segment: one returning global atomic per tile
  const int nt = s_ntouched;
  if (nt <= kBFTouched) {
    for (int i = lane; i < nt; i += 64) {
      const int t = s_touched[i];
      const int c = s_hist[t];
      s_base[t] = atomicAdd(&a.cursor[t], c);
      s_hist[t] = 0;
    }
  } else {  // (a workgroup whose Gaussians touch more tiles than the list holds: the whole histogram)
    for (int t = lane; t < T; t += 64) {
      const int c = s_hist[t];
      if (c) { s_base[t] = atomicAdd(&a.cursor[t], c); s_hist[t] = 0; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  EG_BF_STAMP2(6);  // slots reserved (returning global atomics)
  const unsigned long long key = ((unsigned long long)(unsigned)__float_as_int(s1.z) << 32) | (unsigned)g;
  for (int ty = y0; ty < y1; ++ty)
    for (int tx = x0; tx < x1; ++tx) {
      const bool hit = small_box ? ((mask >> ((ty - y0) * bw + (tx - x0))) & 1u) != 0u
                                 : splat_hits_tile(s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, tx, ty);
      if (!hit) continue;
      const int t = ty * tw + tx;
      const int slot = s_base[t] + atomicAdd(&s_hist[t], 1);
      if (slot < a.seg_cap) a.keys[(size_t)t * a.seg_cap + slot] = key;
    }
  EG_BF_STAMP2(7);  // keys stored
}

__global__ void __launch_bounds__(64 * kBFWaves) __attribute__((amdgpu_waves_per_eu(EG_BF_WAVES_PER_EU, EG_BF_WAVES_PER_EU)))
gaussian_bwd_fused_kernel(const FusedBwdArgs a) {
  // LDS: a workgroup keeps its allocation until its LAST wave ends, i.e. through phase 2, where one wave of eight is left:
  // what a workgroup holds decides how many footprint walks run next to the projection chains (a first version with
  // 32 KB per workgroup -- five workgroups per CU -- took 45 us at config 2 against 35.7 for the two kernels: the CUs sat
  // on phase-2 stragglers).  So phase 2 REUSES phase 1's storage: the histogram and the segment bases take the place of
  // the partial-record exchange (which moves 4 components at a time: 8 KB instead of 16), the touched list the place of
  // the footprint sizes.  13.3 KB at T <= 1024: twelve workgroups per CU.
  extern __shared__ __attribute__((aligned(16))) int s_dyn[];  // max(kBFWaves * 256 floats, 2 T ints)
  __shared__ __attribute__((aligned(16))) int s_walk[kBFGauss * 12];  // per Gaussian: i0 fh pw jlo | jhi cells thr xoff | shear - - -
  __shared__ __attribute__((aligned(16))) float s_g2d[kBFGauss * 8];
  float *red = (float *)s_dyn;               // phase 1: [kBFWaves][64 * 4]
  int *s_touched = s_walk;                  // phase 2: kBFTouched entries + the counter behind them
  int &s_ntouched = s_walk[kBFTouched];
  const int width = a.width, height = a.height, N = a.N;
  const int tw = (width + kTile - 1) / kTile, th = (height + kTile - 1) / kTile, T = tw * th;
  int *s_hist = s_dyn, *s_base = s_dyn + T;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float4 *splat = a.splat;
#ifdef EG_BF_PROF
  // development builds: the first wave's wall-clock stamps (100 MHz) at the phase boundaries of its workgroup, 16 words per
  // workgroup in the g2d buffer (tools/bf_prof.py); every stamp waits for the wave's outstanding memory operations first
  unsigned long long *prof_rec = (unsigned long long *)(a.g2d + 8 * (size_t)a.N) + (size_t)blockIdx.x * 16;  // (behind the [N, 8] records)
#define EG_BF_STAMP(k_)                                                            \
  do {                                                                             \
    if (a.g2d && wv == 0) {                                                        \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                  \
      const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();              \
      if (lane == 0) prof_rec[k_] = t_;                                            \
    }                                                                              \
  } while (0)
#else
#define EG_BF_STAMP(k_) do {} while (0)
#endif
  EG_BF_STAMP(0);
  // (EG_BF_HOIST: what phase 2 reads of its Gaussian -- 11 parameters, the absgrad accumulator, 22 Adam moments -- requested
  // HERE by the wave that will run it, before the walks, 35 registers held through phase 1.  Built and measured: no gain,
  // the round trip hides behind the other waves' walks anyway; off.)
  const int g2 = blockIdx.x * kBFGauss + lane;  // phase 2's Gaussian of this lane (first wave)
  Raw raw = {};
  float ag0 = 0.f, mm[11], vv[11];
#pragma unroll
  for (int k = 0; k < 11; ++k) mm[k] = vv[k] = 0.f;
#if EG_BF_HOIST
  if (wv == 0 && g2 < a.N) {
    const size_t N_ = (size_t)a.N, oS = 3 * N_, oQ = 6 * N_, oO = 10 * N_;
    raw = load_raw(a.means, a.quats, a.scales, a.opacities, g2);
    ag0 = a.absgrads ? a.absgrads[g2] : 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      mm[k] = a.am[3 * g2 + k]; vv[k] = a.av[3 * g2 + k];
      mm[3 + k] = a.am[oS + 3 * g2 + k]; vv[3 + k] = a.av[oS + 3 * g2 + k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { mm[6 + k] = a.am[oQ + 4 * g2 + k]; vv[6 + k] = a.av[oQ + 4 * g2 + k]; }
    mm[10] = a.am[oO + g2]; vv[10] = a.av[oO + g2];
  }
#endif

  // ---------------------------------------------------------------- phase 1: footprint walks (footprint_bwd_kernel's body)
  if (a.loss_part && blockIdx.x == 0 && wv == 1) {  // (the second wave: the first one sizes the footprints)
    float v = a.loss_part[lane];
    if (v != 0.f) a.loss_part[lane] = 0.f;
    v = wave_sum_dpp_f(v);
    if (lane == 63 && v != 0.f) unsafeAtomicAdd(a.loss_out, v);
  }
  const int gbase = blockIdx.x * kBFGauss + wv * 8;
  if (threadIdx.x < kBFGauss) {  // the workgroup's footprints are sized once, one Gaussian per lane of the first wave
    const int hg = blockIdx.x * kBFGauss + (int)threadIdx.x;
    Walk w = walk_of(make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), width, height);
    if (hg < N) w = walk_of(splat[2 * hg], splat[2 * hg + 1], width, height);
    int4 *dst = (int4 *)&s_walk[12 * threadIdx.x];
    dst[0] = make_int4(w.i0, w.fh, w.pw, w.jlo);
    dst[1] = make_int4(w.jhi, w.cells, __float_as_int(w.thr), __float_as_int(w.xoff));
    dst[2] = make_int4(__float_as_int(w.shear), 0, 0, 0);
  }
  __syncthreads();
  if (gbase < N) {  // (wave-uniform; a wave beyond N has nothing to walk but still meets the barrier below)
    const __amdgpu_buffer_rsrc_t rec_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)a.gtstop, 0, width * height * (int)sizeof(StopRec), 0x00020000);
    Walk h;
    {
      const int4 *src = (const int4 *)&s_walk[12 * (wv * 8 + (lane >> 3))];
      const int4 p = src[0], q = src[1];
      h.i0 = p.x; h.fh = p.y; h.pw = p.z; h.jlo = p.w; h.jhi = q.x; h.cells = q.y;
      h.thr = __int_as_float(q.z); h.xoff = __int_as_float(q.w); h.shear = __int_as_float(src[2].x);
    }
    // lanes per Gaussian in proportion to the cell counts: see footprint_bwd_kernel
    int total = h.cells, live = h.cells > 0 ? 1 : 0;
#pragma unroll
    for (int d = 8; d < 64; d <<= 1) {
      total += __shfl_xor(total, d, 64);
      live += __shfl_xor(live, d, 64);
    }
    int h_n = 0, h_first = 0;
    Moments m = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (total > 0) {
      const float share = (float)(64 - live) * (1.f - 1e-5f) * __builtin_amdgcn_rcpf((float)total);
      h_n = h.cells > 0 ? 1 + (int)((float)h.cells * share) : 0;
      int incl = h_n + (h.cells > 0 ? 1 << 16 : 0);
#pragma unroll
      for (int d = 8; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
      }
      const int slack = 64 - (__builtin_amdgcn_readlane(incl, 63) & 0xffff);
      const int rank = (incl >> 16) - (h.cells > 0 ? 1 : 0);
      h_first = (incl & 0xffff) - h_n + min(rank, slack);
      if (h.cells > 0 && rank < slack) ++h_n;
      int k = 0;
#pragma unroll
      for (int q = 1; q < 8; ++q) k += lane >= __builtin_amdgcn_readlane(h_first, 8 * q);
      const int src = 8 * k;
      const int n = max(__shfl(h_n, src, 64), 1);
      const int r = lane - __shfl(h_first, src, 64);
      const int g = min(gbase + k, N - 1);
      const int cells = r < n ? __shfl(h.cells, src, 64) : 0;
      s0 = splat[2 * g];
      s1 = splat[2 * g + 1];
      const int4 *wk = (const int4 *)&s_walk[12 * (wv * 8 + k)];
      const int4 p = wk[0], q = wk[1];
      const float shear_k = __int_as_float(wk[2].x);
      footprint_walk(s0, s1, g, r, n, p.x, cells > 0 ? p.y : 0, max(p.z, 1), p.w, q.x, __int_as_float(q.z),
                     __int_as_float(q.w), shear_k, width, rec_rsrc, m);
    }
    // partial g2d records through LDS, lane (k, component) adds the partials of Gaussian k's lanes in lane order (as
    // footprint_bwd_kernel does) -- four components at a time: vx vy |vx| |vy|, then va vb vc vo
    float *mine = red + wv * 256 + lane * 4;
    const int comp = lane & 7;
    float sum = 0.f;
    *(float4 *)mine = make_float4(s0.z * m.w_x + s0.w * m.w_y, s0.w * m.w_x + s1.x * m.w_y, (2.f / kLog2e) * m.abs_x,
                                  (2.f / kLog2e) * m.abs_y);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (comp < 4)
      for (int t = 0; t < h_n; ++t) sum += red[wv * 256 + (h_first + t) * 4 + comp];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    *(float4 *)mine = make_float4(0.5f * m.w_xx, m.w_xy, 0.5f * m.w_yy, m.v_o);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (comp >= 4)
      for (int t = 0; t < h_n; ++t) sum += red[wv * 256 + (h_first + t) * 4 + comp - 4];
    s_g2d[wv * 64 + lane] = sum;  // record of Gaussian gbase + (lane >> 3), component lane & 7
#ifndef EG_BF_PROF
    if (a.g2d && gbase + (lane >> 3) < N) a.g2d[(size_t)gbase * 8 + lane] = sum;
#endif
  }
  EG_BF_STAMP(1);  // this wave's footprints walked and reduced
  __syncthreads();
  if (wv != 0) return;
  EG_BF_STAMP(2);  // ... and every wave's
  // (the exchange buffer and the footprint sizes are dead: this wave's histogram, bases and touched list move in)
  for (int t = lane; t < T; t += 64) s_hist[t] = 0;
  if (lane == 0) s_ntouched = 0;

  // ---------------------------------------------------------------- phase 2: one Gaussian per lane of the first wave
  {
    const float4 ga = *(const float4 *)&s_g2d[8 * lane], gb = *(const float4 *)&s_g2d[8 * lane + 4];
    const TailLds lds = {s_hist, s_base, s_touched, &s_ntouched};
#ifdef EG_BF_PROF
    unsigned long long *pr = a.g2d ? prof_rec : nullptr;
#else
    unsigned long long *pr = nullptr;
#endif
    gaussian_tail(a, g2, lane, raw, ag0, mm, vv, ga, gb, lds, pr);
  }
}

int launch_gaussian_bwd_fused(float *means, float *quats, float *scales, float *opacities, const float *viewmat, const float *K,
                          const float *next_viewmat, const float *next_K, int32_t N, int32_t width, int32_t height,
                          float eps2d, uint32_t flags, float *splat, const float *gtstop, float *g2d, float *absgrads,
                          float *m, float *v, const eg_adam_hyper &hyper, int32_t *tile_cursor, int32_t seg_cap,
                          uint64_t *keys, void *workspace, int64_t max_items, float *loss_out, hipStream_t st) {
  const int T = cdiv(width, kTile) * cdiv(height, kTile);
  if (T > kPrefixHereMaxTiles) {
    set_error("backward_fused: tile grids of <= %d tiles only", kPrefixHereMaxTiles);
    return EG_ERR_ARG;
  }
  if ((int64_t)width * height * (int64_t)sizeof(StopRec) >= (int64_t)kOutOfImage) {
    set_error("backward_fused: image above 2^31 / 12 pixels");
    return EG_ERR_ARG;
  }
  FusedBwdArgs a;
  a.means = means; a.quats = quats; a.scales = scales; a.opacities = opacities;
  a.viewmat = viewmat; a.K = K; a.next_viewmat = next_viewmat; a.next_K = next_K;
  a.N = N; a.width = width; a.height = height; a.eps2d = eps2d; a.flags = flags;
  a.splat = (float4 *)splat; a.gtstop = (const StopRec *)gtstop; a.g2d = g2d;
  a.absgrads = absgrads; a.am = m; a.av = v; a.hyper = make_adamk(hyper);
  a.cursor = tile_cursor; a.seg_cap = seg_cap; a.keys = (unsigned long long *)keys;
  a.loss_part = nullptr; a.loss_out = loss_out;
  if (workspace && loss_out) a.loss_part = carve_workspace(workspace, max_items, T).loss_part;
  gaussian_bwd_fused_kernel<<<cdiv((int64_t)N, kBFGauss), 64 * kBFWaves, sizeof(int) * max(2 * T, kBFWaves * 256), st>>>(a);
  timing_mark(kMarkFootprint, st);
  return check_launch("backward_fused");
}

}  // namespace eg
