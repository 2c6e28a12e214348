// G1 / G9: fully fused projection forward + backward, absgrad accumulation and Adam.
//
// Replaces gsplat 1.0.0 fully_fused_projection_{fwd,bwd} as reached from the reference call at
// edgegaussians/models/edge_gs.py:250-268, the pre-activations at
// edge_gs.py:253-254, update_absgrads (edge_gs.py:607-613) and the four Adam steps of
// train_gaussians.py:104-106.  One thread per Gaussian: ~150 flops on 44 B in / 32 B out, so the
// kernels are pure streaming passes (HBM/L2 bound); everything per-Gaussian is fused into these
// two launches so the arrays are touched once per step.
//
// Math (written from the projection equations, not translated from the CUDA source):
//   t = Rv mu + tv;  W = Rv R(q) diag(s);  P = J W (2x3);  cov2d = P P^T;  B = cov2d + eps I
//   conic = B^-1;  comp = sqrt(max(0, det cov2d / det B));  o_eff = o * comp
// and the reverse sweep through the same chain.
#include <cstdlib>

#include "common.h"
#include "project_dev.h"

namespace eg {


// The per-tile counters are privatised in LDS (one global atomic per workgroup per touched tile):
// the hot counters of a view share a dozen cache lines and direct atomics serialise on them.
template <bool LDS_COUNT>
__global__ void __launch_bounds__(256)
project_fwd_kernel(const float *__restrict__ means, const float *__restrict__ quats,
                   const float *__restrict__ scales, const float *__restrict__ opacities,
                   const float *__restrict__ viewmat, const float *__restrict__ K, int N, int width, int height,
                   float near_plane, float far_plane, float eps2d, float radius_clip, uint32_t flags,
                   float4 *__restrict__ splat, int *__restrict__ radii, float *__restrict__ means2d,
                   float *__restrict__ depths, float *__restrict__ conics, float *__restrict__ comps,
                   int *__restrict__ tiles_per_gauss, int *__restrict__ tile_counts, float4 *__restrict__ g2d,
                   unsigned *__restrict__ tile_mask, int *__restrict__ scan_offsets,
                   int *__restrict__ scan_item_offsets, int *__restrict__ scan_total, long long scan_capacity,
                   int *__restrict__ scan_ticket) {
  extern __shared__ __attribute__((aligned(16))) int s_hist[];
  const int tw = (width + kTile - 1) / kTile, th = (height + kTile - 1) / kTile, T = tw * th;
  if (LDS_COUNT) {
    for (int t = threadIdx.x; t < T; t += 256) s_hist[t] = 0;
    __syncthreads();
  }
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = g < N;
  const Cam cam = load_cam(viewmat, K);
  Fwd f;
  int radius = 0;
  if (live &&
      forward_geom(cam, means, quats, scales, opacities, g, width, height, near_plane, far_plane, eps2d, flags, f))
    radius = radius_of(f, width, height, radius_clip);

  const bool aa = flags & EG_FLAG_ANTIALIASED;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  if (radius > 0) {
    s0 = make_float4(f.u, f.v, f.a, f.b);
    s1 = make_float4(f.c, aa ? f.o * f.comp : f.o, f.z, __int_as_float(radius));
  }
  if (live) {
    splat[2 * g] = s0;
    splat[2 * g + 1] = s1;
    if (radii) radii[g] = radius;
    if (means2d) { means2d[2 * g] = s0.x; means2d[2 * g + 1] = s0.y; }
    if (depths) depths[g] = s1.z;
    if (conics) { conics[3 * g] = s0.z; conics[3 * g + 1] = s0.w; conics[3 * g + 2] = s1.x; }
    if (comps) comps[g] = radius > 0 ? f.comp : 0.f;
    if (g2d) { g2d[2 * g] = make_float4(0.f, 0.f, 0.f, 0.f); g2d[2 * g + 1] = make_float4(0.f, 0.f, 0.f, 0.f); }
  }

  if (tiles_per_gauss || tile_counts) {
    int n = 0;
    if (radius > 0) {
      int x0, y0, x1, y1;
      if (flags & EG_FLAG_TIGHT_TILES) tile_box_tight(s0.x, s0.y, radius, s0.z, s0.w, s1.x, s1.y, tw, th, x0, y0, x1, y1);
      else tile_box(s0.x, s0.y, radius, tw, th, x0, y0, x1, y1);
      const bool exact = flags & EG_FLAG_TIGHT_TILES;  // drop tiles the ellipse itself does not reach
      n = exact ? 0 : (y1 - y0) * (x1 - x0);
      // the exact hits of boxes up to 32 tiles are remembered as a bit mask so that the emit pass does
      // not evaluate the ellipse test a second time (bit = row-major index inside the box)
      unsigned mask = 0u;
      const bool small_box = (y1 - y0) * (x1 - x0) <= 32;
      int bit = 0;
      if (tile_counts || exact)
        for (int ty = y0; ty < y1; ++ty)
          for (int tx = x0; tx < x1; ++tx, ++bit) {
            if (exact && !splat_hits_tile(s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, tx, ty)) continue;
            if (exact) ++n;
            if (small_box) mask |= 1u << bit;
            if (tile_counts) atomicAdd(LDS_COUNT ? &s_hist[ty * tw + tx] : &tile_counts[ty * tw + tx], 1);
          }
      if (tile_mask) tile_mask[g] = small_box ? mask : 0xffffffffu;  // all ones on a big box: "re-test"
    } else if (live && tile_mask) {
      tile_mask[g] = 0u;
    }
    if (live && tiles_per_gauss) tiles_per_gauss[g] = n;
  }
  if (LDS_COUNT && tile_counts) {
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += 256) {
      const int c = s_hist[t];
      if (c) atomicAdd(&tile_counts[t], c);
    }
  }
  // Fused scan (saves the tile_offsets launch of the training step): the LAST workgroup to finish --
  // found with a device-scope ticket -- scans the per-tile counters.  They are only ever written by
  // device-scope atomics (performed at the coherence point, not in an XCD-private L2 line) and are read
  // here with device-scope atomic loads, so no cache write-back / invalidate fence is needed: each wave
  // only has to drain its own outstanding atomics (vmcnt) before the workgroup takes its ticket.
  if (scan_ticket) {
    __shared__ int s_last;
    __shared__ int s_tmp[4];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(scan_ticket, 1) == (int)gridDim.x - 1);
    __syncthreads();
    if (s_last) {
      const int per = (T + 255) / 256, t0 = threadIdx.x * per, t1 = min(T, t0 + per);
      int sum = 0, isum = 0, cmax = 0;
      for (int t = t0; t < t1; ++t) {
        const int c = __hip_atomic_load(&tile_counts[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sum += c; isum += max(1, (c + 127) >> 7); cmax = max(cmax, c);  // an empty tile owns one (empty) item
      }
      int tot, itot;
      int e = block_excl_scan<256>(sum, s_tmp, tot);
      int ie = block_excl_scan<256>(isum, s_tmp, itot);
      for (int t = t0; t < t1; ++t) {
        const int c = __hip_atomic_load(&tile_counts[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        scan_offsets[t] = (int)min((long long)e, scan_capacity);
        scan_item_offsets[t] = ie;
        e += c; ie += max(1, (c + 127) >> 7);
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) cmax = max(cmax, __shfl_xor(cmax, d, 64));
      if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = cmax;
      __syncthreads();
      if (threadIdx.x == 0) {
        scan_offsets[T] = (int)min((long long)tot, scan_capacity);
        scan_item_offsets[T] = itot;
        scan_total[0] = tot;
        if ((long long)tot > scan_capacity) scan_total[1] = 1;  // sticky: only the host clears it
        scan_total[2] = itot;
        scan_total[3] = max(max(s_tmp[0], s_tmp[1]), max(s_tmp[2], s_tmp[3]));
        *scan_ticket = 0;  // ready for the next launch
      }
    }
  }
}


template <bool ADAM>
__global__ void __launch_bounds__(256)
project_bwd_kernel(float *__restrict__ means, float *__restrict__ quats, float *__restrict__ scales,
                   float *__restrict__ opacities, const float *__restrict__ viewmat, const float *__restrict__ K,
                   int N, int width, int height, float eps2d, uint32_t flags, const float4 *__restrict__ splat,
                   const float4 *__restrict__ g2d, float *__restrict__ v_means, float *__restrict__ v_quats,
                   float *__restrict__ v_scales, float *__restrict__ v_opacities, float *__restrict__ absgrads,
                   const float *__restrict__ v_comps_ext, const float *__restrict__ v_depths_ext,
                   float *__restrict__ am, float *__restrict__ av, AdamK hyper) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= N) return;
  const Cam cam = load_cam(viewmat, K);
  Grads gr;
#pragma unroll
  for (int k = 0; k < 3; ++k) gr.mean[k] = gr.scale[k] = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) gr.quat[k] = 0.f;
  gr.opac = 0.f;

  const int radius = __float_as_int(splat[2 * g + 1].w);
  if (radius > 0) {
    const float4 ga = g2d[2 * g], gb = g2d[2 * g + 1];
    Fwd f;
    // near/far and det culls already passed in the forward (radius > 0), so pass open limits
    forward_geom(cam, means, quats, scales, opacities, g, width, height, -3.0e38f, 3.0e38f, eps2d, flags, f);
    backward_geom(cam, f, eps2d, flags, ga, gb, v_comps_ext != nullptr, v_comps_ext ? v_comps_ext[g] : 0.f,
                  v_depths_ext ? v_depths_ext[g] : 0.f, gr);
    if (absgrads && !(flags & EG_FLAG_ABSGRAD_WRITE)) absgrads[g] += sqrtf(ga.z * ga.z + ga.w * ga.w);
    if (absgrads && (flags & EG_FLAG_ABSGRAD_WRITE)) absgrads[g] = sqrtf(ga.z * ga.z + ga.w * ga.w);
  } else if (absgrads && (flags & EG_FLAG_ABSGRAD_WRITE)) {
    absgrads[g] = 0.f;
  }
  if (!ADAM) {
    if (flags & EG_FLAG_GRAD_ACCUM) {  // (the sum over the cameras of eg_project_bwd_cams: camera 0 wrote, this one adds)
#pragma unroll
      for (int k = 0; k < 3; ++k) { v_means[3 * g + k] += gr.mean[k]; v_scales[3 * g + k] += gr.scale[k]; }
#pragma unroll
      for (int k = 0; k < 4; ++k) v_quats[4 * g + k] += gr.quat[k];
      if (v_opacities) v_opacities[g] += gr.opac;
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) { v_means[3 * g + k] = gr.mean[k]; v_scales[3 * g + k] = gr.scale[k]; }
#pragma unroll
      for (int k = 0; k < 4; ++k) v_quats[4 * g + k] = gr.quat[k];
      if (v_opacities) v_opacities[g] = gr.opac;
    }
  } else {
    // moment layout: [means 3N | scales 3N | quats 4N | opacities N]
    const size_t oM = 0, oS = 3 * (size_t)N, oQ = 6 * (size_t)N, oO = 10 * (size_t)N;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float p = means[3 * g + k], m = am[oM + 3 * g + k], v = av[oM + 3 * g + k];
      adam1(p, gr.mean[k], m, v, 0, hyper);
      means[3 * g + k] = p; am[oM + 3 * g + k] = m; av[oM + 3 * g + k] = v;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float p = scales[3 * g + k], m = am[oS + 3 * g + k], v = av[oS + 3 * g + k];
      adam1(p, gr.scale[k], m, v, 1, hyper);
      scales[3 * g + k] = p; am[oS + 3 * g + k] = m; av[oS + 3 * g + k] = v;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float p = quats[4 * g + k], m = am[oQ + 4 * g + k], v = av[oQ + 4 * g + k];
      adam1(p, gr.quat[k], m, v, 2, hyper);
      quats[4 * g + k] = p; am[oQ + 4 * g + k] = m; av[oQ + 4 * g + k] = v;
    }
    {
      float p = opacities[g], m = am[oO + g], v = av[oO + g];
      adam1(p, gr.opac, m, v, 3, hyper);
      opacities[g] = p; am[oO + g] = m; av[oO + g] = v;
    }
  }
}

// Batched step: the C views of the batch were composited into splat[v] / g2d[v]; one thread per Gaussian runs
// the projection VJP of every view and SUMS the gradients (the same sum a data-parallel all-reduce forms), the
// absgrad increments add up like C calls of update_absgrads (edge_gs.py:607-613), then ONE Adam step.
template <bool ADAM>
__global__ void __launch_bounds__(256)
project_bwd_batched_kernel(float *__restrict__ means, float *__restrict__ quats, float *__restrict__ scales,
                           float *__restrict__ opacities, int N, int width, int height, float eps2d, uint32_t flags,
                           const float4 *__restrict__ splat, const float4 *__restrict__ g2d,
                           float *__restrict__ v_means, float *__restrict__ v_quats, float *__restrict__ v_scales,
                           float *__restrict__ v_opacities, float *__restrict__ absgrads, float *__restrict__ am,
                           float *__restrict__ av, AdamK hyper, const Batch bt, int C) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= N) return;
  Grads acc;
#pragma unroll
  for (int k = 0; k < 3; ++k) acc.mean[k] = acc.scale[k] = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) acc.quat[k] = 0.f;
  acc.opac = 0.f;
  float absinc = 0.f;
  for (int v = 0; v < C; ++v) {
    const float4 *sp = splat + v * bt.splat4, *gd = g2d + v * bt.splat4;
    const int radius = __float_as_int(sp[2 * g + 1].w);
    if (radius <= 0) continue;
    const Cam cam = load_cam(bt.viewmat[v], bt.K[v]);
    const float4 ga = gd[2 * g], gb = gd[2 * g + 1];
    Fwd f;
    forward_geom(cam, means, quats, scales, opacities, g, width, height, -3.0e38f, 3.0e38f, eps2d, flags, f);
    Grads gr;
    backward_geom(cam, f, eps2d, flags, ga, gb, false, 0.f, 0.f, gr);
#pragma unroll
    for (int k = 0; k < 3; ++k) { acc.mean[k] += gr.mean[k]; acc.scale[k] += gr.scale[k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc.quat[k] += gr.quat[k];
    acc.opac += gr.opac;
    absinc += sqrtf(ga.z * ga.z + ga.w * ga.w);
  }
  if (absgrads) {
    if (flags & EG_FLAG_ABSGRAD_WRITE) absgrads[g] = absinc; else absgrads[g] += absinc;
  }
  if (!ADAM) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { v_means[3 * g + k] = acc.mean[k]; v_scales[3 * g + k] = acc.scale[k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) v_quats[4 * g + k] = acc.quat[k];
    v_opacities[g] = acc.opac;
  } else {
    const size_t oM = 0, oS = 3 * (size_t)N, oQ = 6 * (size_t)N, oO = 10 * (size_t)N;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float p = means[3 * g + k], m = am[oM + 3 * g + k], v = av[oM + 3 * g + k];
      adam1(p, acc.mean[k], m, v, 0, hyper);
      means[3 * g + k] = p; am[oM + 3 * g + k] = m; av[oM + 3 * g + k] = v;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float p = scales[3 * g + k], m = am[oS + 3 * g + k], v = av[oS + 3 * g + k];
      adam1(p, acc.scale[k], m, v, 1, hyper);
      scales[3 * g + k] = p; am[oS + 3 * g + k] = m; av[oS + 3 * g + k] = v;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float p = quats[4 * g + k], m = am[oQ + 4 * g + k], v = av[oQ + 4 * g + k];
      adam1(p, acc.quat[k], m, v, 2, hyper);
      quats[4 * g + k] = p; am[oQ + 4 * g + k] = m; av[oQ + 4 * g + k] = v;
    }
    {
      float p = opacities[g], m = am[oO + g], v = av[oO + g];
      adam1(p, acc.opac, m, v, 3, hyper);
      opacities[g] = p; am[oO + g] = m; av[oO + g] = v;
    }
  }
}

// stand-alone Adam over the 11 N parameters (used after an RCCL gradient all-reduce)
// Gradients of a regulariser iteration (train_gaussians.py:108-131) scaled on the fly: the loss kernels of knn.hip leave
// RAW gradients and the sum of the per-Gaussian terms; loss = 1 + w * sum (direction) or sum / N (ratio),
// lambda = (running projection-loss sum) * factor / loss, g = (raw * w) * lambda or (raw / N) * lambda -- the float32
// operation order of the tensor expressions this replaces.  A null gradient pointer is a zero gradient (the optimizers
// whose parameter is outside the loss still step: torch 1.13's zero_grad() leaves zeros, not None).
struct RegScale {
  const float *sum;       // null: plain Adam (no scaling)
  const float *loss_sum;  // device scalar or null -> loss_sum_host
  float loss_sum_host, factor, w, n_gauss;
  int ratio;
  float *loss_out;
};

__global__ void __launch_bounds__(256)
adam_multi_kernel(float *__restrict__ means, float *__restrict__ scales, float *__restrict__ quats,
                  float *__restrict__ opacities, const float *__restrict__ g_means,
                  const float *__restrict__ g_scales, const float *__restrict__ g_quats,
                  const float *__restrict__ g_opacities, float *__restrict__ am, float *__restrict__ av, int N,
                  AdamK hyper, const float *__restrict__ absgrad_inc, float *__restrict__ absgrads, RegScale rs) {
  float lam = 1.f;
  if (rs.sum) {
    const float loss = rs.ratio ? rs.sum[0] / rs.n_gauss : 1.0f + rs.w * rs.sum[0];
    lam = (rs.loss_sum ? rs.loss_sum[0] * rs.factor : rs.loss_sum_host * rs.factor) / loss;
    if (rs.loss_out && blockIdx.x == 0 && threadIdx.x == 0) rs.loss_out[0] = loss;
  }
  if (absgrads)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)N; i += (size_t)gridDim.x * blockDim.x)
      absgrads[i] += absgrad_inc[i];
  const size_t total = 11 * (size_t)N;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    float *p; const float *g; int grp; size_t j;
    if (i < 3 * (size_t)N) { p = means; g = g_means; grp = 0; j = i; }
    else if (i < 6 * (size_t)N) { p = scales; g = g_scales; grp = 1; j = i - 3 * (size_t)N; }
    else if (i < 10 * (size_t)N) { p = quats; g = g_quats; grp = 2; j = i - 6 * (size_t)N; }
    else { p = opacities; g = g_opacities; grp = 3; j = i - 10 * (size_t)N; }
    float pv = p[j], m = am[i], v = av[i];
    float gv = g ? g[j] : 0.f;
    if (rs.sum && g) gv = (rs.ratio ? gv / rs.n_gauss : gv * rs.w) * lam;
    adam1(pv, gv, m, v, grp, hyper);
    p[j] = pv; am[i] = m; av[i] = v;
  }
}

__global__ void absgrad_accum_kernel(const float2 *__restrict__ ag, int N, float *__restrict__ absgrads) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= N) return;
  const float2 a = ag[g];
  absgrads[g] += sqrtf(a.x * a.x + a.y * a.y);
}


}  // namespace eg

namespace eg {
int launch_project_bwd_batched(float *means, float *quats, float *scales, float *opacities, int32_t N, int32_t width,
                               int32_t height, float eps2d, uint32_t flags, const float *splat, const float *g2d,
                               float *v_means, float *v_quats, float *v_scales, float *v_opacities, float *absgrads,
                               float *m, float *v, const eg_adam_hyper *hyper_host, const Batch &bt, int C,
                               hipStream_t st) {
  if (hyper_host)
    project_bwd_batched_kernel<true><<<cdiv(N, 256), 256, 0, st>>>(
        means, quats, scales, opacities, N, width, height, eps2d, flags, (const float4 *)splat, (const float4 *)g2d,
        nullptr, nullptr, nullptr, nullptr, absgrads, m, v, make_adamk(*hyper_host), bt, C);
  else {
    AdamK dummy = {};
    project_bwd_batched_kernel<false><<<cdiv(N, 256), 256, 0, st>>>(
        means, quats, scales, opacities, N, width, height, eps2d, flags, (const float4 *)splat, (const float4 *)g2d,
        v_means, v_quats, v_scales, v_opacities, absgrads, nullptr, nullptr, dummy, bt, C);
  }
  return check_launch("project_bwd_batched");
}
}  // namespace eg

using namespace eg;

// ---------------------------------------------------------------------------------------------
// Projection + binning in ONE pass for the training step ("segmented" layout): every tile owns a
// fixed segment of seg_cap slots in the key array, so a Gaussian's keys can be placed without knowing
// the other tiles' totals -- no count pass, no scan, no second (emit) kernel.  Per workgroup: count
// the hits of its Gaussians in an LDS histogram, reserve a run of slots per touched tile with ONE
// returning global atomic on that tile's cursor, hand the slots out from LDS.  With the rows in
// spatial order a workgroup touches a handful of tiles.  The cursor ends up holding the tile's
// population (the sort kernel reads it and returns it to zero); a tile that outgrows its segment
// raises total[1] and drops the excess (the caller re-sizes, like for the global capacity).
// What the last workgroup of eg_project_emit leaves behind: the scan over the tiles.  (The per-tile
// tables -- key range, item range, item -> tile map, cursor reset -- are written by the sort kernel,
// one workgroup per tile, in parallel.)
struct SegOut {
  int *item_first;  // [T]: first item (128-Gaussian slice) of every tile, an exclusive scan
  int max_items;
  int *total;       // [4]: M, overflow flag, items, largest tile population
  int *ticket;      // [1]: zero on entry, zero on exit
  int *item_front;  // optional [T + 1] (EG_FLAG_FRONT_PREFIX): exclusive scan of min(items, EG_FRONT_LARGE), total at [T]
};

// Threads (= Gaussians) per workgroup of the projection + binning kernels: 512 halves the per-workgroup histogram sweeps and
// cursor atomics of 256 (config 2 16.6 -> 14.6 us, config 3 48 -> 33 in round 3; 61 against 47 us at config 3 in round 6);
// 1024 loses that again to the longer barriers.  Round 6: SMALL scenes take 256 -- at 30 k Gaussians 59 workgroups of 512
// leave three quarters of the CUs without one and every barrier waits for the slowest of eight waves: project_bwd_emit
// 12.5 -> 10.7 us at config 1, equal at 100 k (profiles/r06_wg_ab.txt).
constexpr int kPE = 512, kPESmall = 256;
constexpr int kPESmallMaxGaussians = 65536;
__host__ __device__ constexpr bool pe_small(int N) { return N <= kPESmallMaxGaussians; }

// Projection of Gaussian g (raw parameters in registers) for camera `cam`, packed record, exact tile hits, key
// emission into the fixed per-tile segments, and -- in the last workgroup of the view -- the scan over the tiles.
// Shared by project_emit_kernel and by the tail of project_bwd_emit_kernel (which projects the NEXT view with
// the parameters Adam has just updated).  s_mem: 2 T ints of LDS when LDS_HIST.
template <bool LDS_HIST, int PE>
__device__ __forceinline__ void emit_body(const Raw &raw, bool live, int g, const Cam &cam, int width, int height,
                                          uint32_t flags, float4 *__restrict__ splat, int *__restrict__ cursor,
                                          int seg_cap, unsigned long long *__restrict__ keys, const SegOut &out,
                                          int *s_mem) {
  const int tw = (width + kTile - 1) / kTile, th = (height + kTile - 1) / kTile, T = tw * th;
  int *s_hist = s_mem, *s_base = s_mem + T;
  if (LDS_HIST) {
    for (int t = threadIdx.x; t < T; t += PE) s_hist[t] = 0;
    __syncthreads();
  }
  Fwd f;
  int radius = 0;
  if (live && forward_geom(cam, raw, width, height, 0.01f, 1e10f, 0.3f, flags, f))
    radius = radius_of(f, width, height, 0.f);
  const bool aa = flags & EG_FLAG_ANTIALIASED;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  if (radius > 0) {
    s0 = make_float4(f.u, f.v, f.a, f.b);
    s1 = make_float4(f.c, aa ? f.o * f.comp : f.o, f.z, __int_as_float(radius));
  }
  if (live) {
    splat[2 * g] = s0;
    splat[2 * g + 1] = s1;
  }
  // exact tile hits (gsplat's box, the ellipse's extent, the ellipse-vs-tile test), remembered as a bit
  // mask for boxes of up to 32 tiles so that the second walk below does not repeat the test
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
        if (LDS_HIST) atomicAdd(&s_hist[ty * tw + tx], 1);
      }
  }
  const unsigned long long key = ((unsigned long long)(unsigned)__float_as_int(s1.z) << 32) | (unsigned)g;
#define EG_HIT(tx, ty) \
  (small_box ? ((mask >> (((ty)-y0) * bw + ((tx)-x0))) & 1u) != 0u : splat_hits_tile(s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, tx, ty))
  if (!LDS_HIST) {  // very large tile grids: one global atomic per hit
    for (int ty = y0; ty < y1; ++ty)
      for (int tx = x0; tx < x1; ++tx) {
        if (!EG_HIT(tx, ty)) continue;
        const int t = ty * tw + tx;
        const int slot = atomicAdd(&cursor[t], 1);
        if (slot < seg_cap) keys[(size_t)t * seg_cap + slot] = key;
      }
  } else {
    __syncthreads();
    // slots [base, base + c) of tile t's segment.  Four tiles per thread and round, their returning atomics all issued
    // before the first result is stored (round 5): one atomic per round was one round trip to the coherence point per
    // PE tiles of the grid, in a row -- 15 of them at 1600 x 1200
    for (int t0 = threadIdx.x; t0 < T; t0 += 4 * PE) {
      int c[4], b[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) c[q] = (t0 + q * PE < T) ? s_hist[t0 + q * PE] : 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) b[q] = c[q] ? atomicAdd(&cursor[t0 + q * PE], c[q]) : 0;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (c[q]) { s_base[t0 + q * PE] = b[q]; s_hist[t0 + q * PE] = 0; }
    }
    __syncthreads();
    for (int ty = y0; ty < y1; ++ty)
      for (int tx = x0; tx < x1; ++tx) {
        if (!EG_HIT(tx, ty)) continue;
        const int t = ty * tw + tx;
        const int slot = s_base[t] + atomicAdd(&s_hist[t], 1);
        if (slot < seg_cap) keys[(size_t)t * seg_cap + slot] = key;
      }
  }
#undef EG_HIT
  // The LAST workgroup to finish -- found with a device-scope ticket -- scans the tile populations.  The cursors are only ever touched by device-scope atomics (performed at the
  // coherence point, not in an XCD-private L2 line) and are read here with device-scope atomic loads,
  // so no cache write-back / invalidate fence is needed: every wave only drains its own outstanding
  // atomics (vmcnt) before the workgroup takes its ticket.
  // out.ticket == nullptr: the sort kernel forms the tile prefix and the totals itself (binning.hip, "prefix
  // here"): this kernel ends with its last key store instead of ~3 us of barrier, ticket, cursor loads and scan
  if (out.ticket == nullptr) return;
  __shared__ int s_last;
  __shared__ int s_tmp[PE / 64];
  // (every cursor atomic of this workgroup is a RETURNING atomic whose value has been consumed above, i.e. it has
  // been performed; the key / parameter stores still in flight need not be waited for: the scan reads cursors only)
  __syncthreads();
  if (threadIdx.x == 0) s_last = (atomicAdd(out.ticket, 1) == (int)gridDim.x - 1);
  __syncthreads();
  if (!s_last) return;
  // stage the T populations in LDS with coalesced, independent loads; every thread then owns a
  // contiguous run of tiles (the scan needs runs, the memory system wants strides)
  const int per = (T + PE - 1) / PE, t0 = threadIdx.x * per, t1 = min(T, t0 + per);
  if (LDS_HIST) {
    for (int t = threadIdx.x; t < T; t += PE)
      s_hist[t] = __hip_atomic_load(&cursor[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  }
  int isum = 0, msum = 0, cmax = 0, fsum = 0;
  for (int t = t0; t < t1; ++t) {
    const int pop = LDS_HIST ? s_hist[t] : __hip_atomic_load(&cursor[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int kept = min(pop, seg_cap), it = max(1, (kept + 127) >> 7);  // an empty tile owns one (empty) item
    isum += it; msum += kept; cmax = max(cmax, pop); fsum += min(it, EG_FRONT_LARGE);
  }
  int itot, mtot, ftot = 0;
  int ie = block_excl_scan<PE>(isum, s_tmp, itot);
  (void)block_excl_scan<PE>(msum, s_tmp, mtot);
  // (dispatch classes of the forward on large tile grids: where the front slices of the tiles before this one end)
  int fe = out.item_front ? block_excl_scan<PE>(fsum, s_tmp, ftot) : 0;
  for (int t = t0; t < t1; ++t) {
    const int pop = LDS_HIST ? s_hist[t] : __hip_atomic_load(&cursor[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (LDS_HIST) s_base[t] = ie; else out.item_first[t] = min(ie, out.max_items);
    if (out.item_front) out.item_front[t] = fe;
    const int it = max(1, (min(pop, seg_cap) + 127) >> 7);
    ie += it; fe += min(it, EG_FRONT_LARGE);
  }
  // (an overflowing view keeps the records in item order -- the sort kernel sees the -1: the tiles truncated by the item
  // capacity write no records, which in the two-class order would leave holes of stale records below total[2])
  if (out.item_front && threadIdx.x == 0) out.item_front[T] = itot > out.max_items ? -1 : ftot;
  if (LDS_HIST) {
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += PE) out.item_first[t] = min(s_base[t], out.max_items);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) cmax = max(cmax, __shfl_xor(cmax, d, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = cmax;
  __syncthreads();
  if (threadIdx.x == 0) {
    int m = 0;
    for (int w = 0; w < PE / 64; ++w) m = max(m, s_tmp[w]);
    out.total[0] = mtot;
    if (m > seg_cap || itot > out.max_items) out.total[1] = 1;  // sticky: only the host clears it
    out.total[2] = min(itot, out.max_items);
    out.total[3] = m;
    *out.ticket = 0;  // ready for the next launch
  }
}

template <bool LDS_HIST, int PE>
__global__ void __launch_bounds__(PE)
project_emit_kernel(const float *__restrict__ means, const float *__restrict__ quats,
                    const float *__restrict__ scales, const float *__restrict__ opacities,
                    const float *__restrict__ viewmat, const float *__restrict__ K, int N, int width, int height,
                    uint32_t flags, float4 *__restrict__ splat, int *__restrict__ cursor, int seg_cap,
                    unsigned long long *__restrict__ keys, const SegOut out_, const Batch bt) {
  extern __shared__ __attribute__((aligned(16))) int s_mem[];
  // view of this workgroup (a batched step runs its C views as blockIdx.y over [C, ...] work buffers)
  const int bv = blockIdx.y;
  SegOut out = out_;
  splat += bv * bt.splat4; cursor += bv * bt.tiles; keys += bv * bt.keys;
  out.item_first += bv * bt.tiles; out.total += 4 * bv;
  if (out.ticket) out.ticket += bv;  // (EG_FLAG_FRONT_PREFIX is a single-view feature: the batched step does not set it)
  if (bt.viewmat[0]) { viewmat = bt.viewmat[bv]; K = bt.K[bv]; }
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = g < N;
  Raw raw = {};
  if (live) raw = load_raw(means, quats, scales, opacities, g);
  emit_body<LDS_HIST, PE>(raw, live, g, load_cam(viewmat, K), width, height, flags, splat, cursor, seg_cap, keys, out, s_mem);
}

// Tail fusion across the step boundary (single-view training, consecutive steps enqueued natively): the
// projection backward + absgrad + Adam of view k and -- with the parameters still in registers -- the projection,
// binning and tile scan of view k + 1.  One launch and one read of the parameters instead of two.
template <bool LDS_HIST, bool HOIST, int PE>
__global__ void __launch_bounds__(PE)
project_bwd_emit_kernel(float *__restrict__ means, float *__restrict__ quats, float *__restrict__ scales,
                        float *__restrict__ opacities, const float *__restrict__ viewmat, const float *__restrict__ K,
                        const float *__restrict__ next_viewmat, const float *__restrict__ next_K, int N, int width,
                        int height, float eps2d, uint32_t flags, float4 *__restrict__ splat,
                        const float4 *__restrict__ g2d, float *__restrict__ absgrads, float *__restrict__ am,
                        float *__restrict__ av, AdamK hyper, int *__restrict__ cursor, int seg_cap,
                        unsigned long long *__restrict__ keys, const SegOut out) {
  extern __shared__ __attribute__((aligned(16))) int s_mem[];
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = g < N;
  Raw raw = {};
  if (live) {
    // Every load of this thread is issued HERE, before anything waits: parameters, radius, the g2d record (the
    // footprint backward writes one for every Gaussian, visible or not), the absgrad accumulator and the 22 Adam
    // moments -- one round trip to memory instead of three (the compiler does not hoist loads over the
    // `radius > 0` branch; ~100 k threads are 1.5 waves per SIMD: nothing else hides the latency)
    // (HOIST: 144 instead of 104 VGPRs, i.e. 3 instead of 4 waves per SIMD -- with 200 k Gaussians, 3 waves per SIMD
    // of work, the late loads of the plain order overlap better: 49 vs 58 us; the launcher picks by N)
    raw = load_raw(means, quats, scales, opacities, g);
    const int radius = __float_as_int(splat[2 * g + 1].w);
    float4 ga = make_float4(0.f, 0.f, 0.f, 0.f), gb = ga;
    float ag = 0.f;
    // moment layout: [means 3N | scales 3N | quats 4N | opacities N]
    const size_t oM = 0, oS = 3 * (size_t)N, oQ = 6 * (size_t)N, oO = 10 * (size_t)N;
    float mm[11], vv[11];
    auto load_moments = [&]() {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        mm[k] = am[oM + 3 * g + k]; vv[k] = av[oM + 3 * g + k];
        mm[3 + k] = am[oS + 3 * g + k]; vv[3 + k] = av[oS + 3 * g + k];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { mm[6 + k] = am[oQ + 4 * g + k]; vv[6 + k] = av[oQ + 4 * g + k]; }
      mm[10] = am[oO + g]; vv[10] = av[oO + g];
    };
    if (HOIST) {
      ga = g2d[2 * g]; gb = g2d[2 * g + 1];
      ag = absgrads ? absgrads[g] : 0.f;
      load_moments();
    }
    const Cam cam = load_cam(viewmat, K);
    Grads gr;
#pragma unroll
    for (int k = 0; k < 3; ++k) gr.mean[k] = gr.scale[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) gr.quat[k] = 0.f;
    gr.opac = 0.f;
    if (radius > 0) {
      if (!HOIST) {
        ga = g2d[2 * g]; gb = g2d[2 * g + 1];
        ag = absgrads ? absgrads[g] : 0.f;
      }
      Fwd f;
      forward_geom(cam, raw, width, height, -3.0e38f, 3.0e38f, eps2d, flags, f);
      backward_geom(cam, f, eps2d, flags, ga, gb, false, 0.f, 0.f, gr);
      if (absgrads) absgrads[g] = ag + sqrtf(ga.z * ga.z + ga.w * ga.w);
    }
    if (!HOIST) load_moments();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      adam1(raw.m[k], gr.mean[k], mm[k], vv[k], 0, hyper);
      means[3 * g + k] = raw.m[k]; am[oM + 3 * g + k] = mm[k]; av[oM + 3 * g + k] = vv[k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      adam1(raw.s[k], gr.scale[k], mm[3 + k], vv[3 + k], 1, hyper);
      scales[3 * g + k] = raw.s[k]; am[oS + 3 * g + k] = mm[3 + k]; av[oS + 3 * g + k] = vv[3 + k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      adam1(raw.q[k], gr.quat[k], mm[6 + k], vv[6 + k], 2, hyper);
      quats[4 * g + k] = raw.q[k]; am[oQ + 4 * g + k] = mm[6 + k]; av[oQ + 4 * g + k] = vv[6 + k];
    }
    adam1(raw.o, gr.opac, mm[10], vv[10], 3, hyper);
    opacities[g] = raw.o; am[oO + g] = mm[10]; av[oO + g] = vv[10];
  }
  // the next view, with the updated parameters still in registers (this thread's splat row of view k is dead now)
  emit_body<LDS_HIST, PE>(raw, live, g, load_cam(next_viewmat, next_K), width, height, flags, splat, cursor, seg_cap, keys,
                      out, s_mem);
}

// Data-parallel tail: the four Adam steps on the ALL-REDUCED gradients (what eg_adam_multi does, same arithmetic)
// and -- with the updated parameters still in registers -- the projection, binning and tile scan of the view this
// rank rasterises NEXT (the body of project_emit_kernel): the tail fusion of the single-GPU step for the
// data-parallel leg, one launch and one read of the parameters instead of two.
template <bool LDS_HIST, int PE>
__global__ void __launch_bounds__(PE)
adam_emit_kernel(float *__restrict__ means, float *__restrict__ scales, float *__restrict__ quats,
                 float *__restrict__ opacities, const float *__restrict__ g_means, const float *__restrict__ g_scales,
                 const float *__restrict__ g_quats, const float *__restrict__ g_opacities, float *__restrict__ am,
                 float *__restrict__ av, int N, AdamK hyper, const float *__restrict__ absgrad_inc,
                 float *__restrict__ absgrads, const float *__restrict__ next_viewmat,
                 const float *__restrict__ next_K, int width, int height, uint32_t flags, float4 *__restrict__ splat,
                 int *__restrict__ cursor, int seg_cap, unsigned long long *__restrict__ keys, const SegOut out) {
  extern __shared__ __attribute__((aligned(16))) int s_mem[];
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = g < N;
  Raw raw = {};
  if (live) {
    raw = load_raw(means, quats, scales, opacities, g);
    if (absgrads) absgrads[g] += absgrad_inc[g];
    // moment layout: [means 3N | scales 3N | quats 4N | opacities N]
    const size_t oM = 0, oS = 3 * (size_t)N, oQ = 6 * (size_t)N, oO = 10 * (size_t)N;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float m = am[oM + 3 * g + k], v = av[oM + 3 * g + k];
      adam1(raw.m[k], g_means[3 * g + k], m, v, 0, hyper);
      means[3 * g + k] = raw.m[k]; am[oM + 3 * g + k] = m; av[oM + 3 * g + k] = v;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float m = am[oS + 3 * g + k], v = av[oS + 3 * g + k];
      adam1(raw.s[k], g_scales[3 * g + k], m, v, 1, hyper);
      scales[3 * g + k] = raw.s[k]; am[oS + 3 * g + k] = m; av[oS + 3 * g + k] = v;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float m = am[oQ + 4 * g + k], v = av[oQ + 4 * g + k];
      adam1(raw.q[k], g_quats[4 * g + k], m, v, 2, hyper);
      quats[4 * g + k] = raw.q[k]; am[oQ + 4 * g + k] = m; av[oQ + 4 * g + k] = v;
    }
    {
      float m = am[oO + g], v = av[oO + g];
      adam1(raw.o, g_opacities[g], m, v, 3, hyper);
      opacities[g] = raw.o; am[oO + g] = m; av[oO + g] = v;
    }
  }
  emit_body<LDS_HIST, PE>(raw, live, g, load_cam(next_viewmat, next_K), width, height, flags, splat, cursor, seg_cap, keys,
                      out, s_mem);
}

static int launch_project_fwd(const float *means, const float *quats, const float *scales, const float *opacities,
                              const float *viewmat, const float *K, int32_t N, int32_t width, int32_t height,
                              float near_plane, float far_plane, float eps2d, float radius_clip, uint32_t flags,
                              float *splat, int32_t *radii, float *means2d, float *depths, float *conics,
                              float *compensations, int32_t *tiles_per_gauss, int32_t *tile_counts, float *g2d,
                              uint32_t *tile_mask, int32_t *offsets, int32_t *item_offsets, int32_t *total,
                              int64_t capacity, int32_t *ticket, eg_stream_t stream) {
  const int T = cdiv(width, kTile) * cdiv(height, kTile);
  if (tile_counts && T <= 16384)
    project_fwd_kernel<true><<<cdiv(N, 256), 256, sizeof(int) * T, as_stream(stream)>>>(
        means, quats, scales, opacities, viewmat, K, N, width, height, near_plane, far_plane, eps2d, radius_clip,
        flags, (float4 *)splat, radii, means2d, depths, conics, compensations, tiles_per_gauss, tile_counts,
        (float4 *)g2d, tile_mask, offsets, item_offsets, total, (long long)capacity, ticket);
  else
    project_fwd_kernel<false><<<cdiv(N, 256), 256, 0, as_stream(stream)>>>(
        means, quats, scales, opacities, viewmat, K, N, width, height, near_plane, far_plane, eps2d, radius_clip,
        flags, (float4 *)splat, radii, means2d, depths, conics, compensations, tiles_per_gauss, tile_counts,
        (float4 *)g2d, tile_mask, offsets, item_offsets, total, (long long)capacity, ticket);
  return check_launch("project_fwd");
}

extern "C" int eg_project_fwd(const float *means, const float *quats, const float *scales, const float *opacities,
                              const float *viewmat, const float *K, int32_t N, int32_t width, int32_t height,
                              float near_plane, float far_plane, float eps2d, float radius_clip, uint32_t flags,
                              float *splat, int32_t *radii, float *means2d, float *depths, float *conics,
                              float *compensations, int32_t *tiles_per_gauss, int32_t *tile_counts, float *g2d,
                              eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && width > 0 && height > 0, "bad sizes");
  if (N == 0) return EG_OK;
  EG_REQUIRE(means && quats && scales && opacities && viewmat && K && splat, "null pointer");
  return launch_project_fwd(means, quats, scales, opacities, viewmat, K, N, width, height, near_plane, far_plane,
                            eps2d, radius_clip, flags, splat, radii, means2d, depths, conics, compensations,
                            tiles_per_gauss, tile_counts, g2d, nullptr, nullptr, nullptr, nullptr, 0, nullptr, stream);
}

extern "C" int eg_project_bin(const float *means, const float *quats, const float *log_scales,
                              const float *logit_opacities, const float *viewmat, const float *K, int32_t N,
                              int32_t width, int32_t height, uint32_t flags, float *splat, int32_t *tile_counts,
                              uint32_t *tile_mask, int64_t capacity, int32_t *offsets, int32_t *item_offsets,
                              int32_t *total, int32_t *ticket, eg_stream_t stream) {
  EG_REQUIRE(N > 0 && width > 0 && height > 0, "bad sizes");
  EG_REQUIRE(means && quats && log_scales && logit_opacities && viewmat && K && splat && tile_counts && tile_mask &&
                 offsets && item_offsets && total && ticket,
             "null pointer");
  return launch_project_fwd(means, quats, log_scales, logit_opacities, viewmat, K, N, width, height, 0.01f, 1e10f,
                            0.3f, 0.0f, flags, splat, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, tile_counts,
                            nullptr, tile_mask, offsets, item_offsets, total, capacity, ticket, stream);
}

namespace eg {
int launch_project_emit(const float *means, const float *quats, const float *log_scales, const float *logit_opacities,
                        const float *viewmat, const float *K, int32_t N, int32_t width, int32_t height, uint32_t flags,
                        float *splat, int32_t *tile_cursor, int32_t seg_cap, uint64_t *keys, int32_t *item_first,
                        int32_t max_items, int32_t *total, int32_t *ticket, const Batch &bt, int C, hipStream_t st) {
  const int T = cdiv(width, kTile) * cdiv(height, kTile);
  SegOut out;
  out.item_first = item_first; out.max_items = max_items;
  out.total = total; out.ticket = ticket;
  out.item_front = (ticket && (flags & EG_FLAG_FRONT_PREFIX)) ? ticket + 1 : nullptr;
#define EG_EMIT(LDS, PE_, SMEM)                                                                                        \
  project_emit_kernel<LDS, PE_><<<dim3(cdiv(N, PE_), C), PE_, SMEM, st>>>(                                              \
      means, quats, log_scales, logit_opacities, viewmat, K, N, width, height, flags, (float4 *)splat, tile_cursor,     \
      seg_cap, (unsigned long long *)keys, out, bt)
  // (C views per launch multiply the workgroups: the batched step keeps 512)
  const bool small = pe_small(N) && C == 1;
  if (2 * T <= 16384) {
    if (small) EG_EMIT(true, kPESmall, sizeof(int) * 2 * T); else EG_EMIT(true, kPE, sizeof(int) * 2 * T);
  } else {
    if (small) EG_EMIT(false, kPESmall, 0); else EG_EMIT(false, kPE, 0);
  }
#undef EG_EMIT
  return check_launch("project_emit");
}
}  // namespace eg

extern "C" int eg_project_emit(const float *means, const float *quats, const float *log_scales,
                               const float *logit_opacities, const float *viewmat, const float *K, int32_t N,
                               int32_t width, int32_t height, uint32_t flags, float *splat, int32_t *tile_cursor,
                               int32_t seg_cap, uint64_t *keys, int32_t *item_first, int32_t max_items,
                               int32_t *total, int32_t *ticket, eg_stream_t stream) {
  EG_REQUIRE(N > 0 && width > 0 && height > 0 && seg_cap > 0 && max_items > 0, "bad sizes");
  EG_REQUIRE(means && quats && log_scales && logit_opacities && viewmat && K && splat && tile_cursor && keys &&
                 item_first && total && ticket,
             "null pointer");
  EG_REQUIRE((flags & EG_FLAG_TIGHT_TILES) != 0, "the segmented path bins with the exact tile test");
  const int T = cdiv(width, kTile) * cdiv(height, kTile);
  EG_REQUIRE((int64_t)T * seg_cap < (1ll << 31), "T * seg_cap must fit 31 bits");
  return launch_project_emit(means, quats, log_scales, logit_opacities, viewmat, K, N, width, height, flags, splat,
                             tile_cursor, seg_cap, keys, item_first, max_items, total, ticket, Batch{}, 1,
                             as_stream(stream));
}

extern "C" int eg_project_bwd(const float *means, const float *quats, const float *scales, const float *opacities,
                              const float *viewmat, const float *K, int32_t N, int32_t width, int32_t height,
                              float eps2d, uint32_t flags, const float *splat, const float *g2d,
                              const float *v_comps_ext, const float *v_depths_ext, float *v_means,
                              float *v_quats, float *v_scales, float *v_opacities, float *absgrads,
                              eg_stream_t stream) {
  EG_REQUIRE(N >= 0, "bad sizes");
  if (N == 0) return EG_OK;
  EG_REQUIRE(means && quats && scales && opacities && viewmat && K && splat && g2d, "null pointer");
  EG_REQUIRE(v_means && v_quats && v_scales && (v_opacities || v_comps_ext), "null gradient output");
  AdamK dummy = {};
  project_bwd_kernel<false><<<cdiv(N, 256), 256, 0, as_stream(stream)>>>(
      (float *)means, (float *)quats, (float *)scales, (float *)opacities, viewmat, K, N, width, height, eps2d,
      flags, (const float4 *)splat, (const float4 *)g2d, v_means, v_quats, v_scales, v_opacities, absgrads,
      v_comps_ext, v_depths_ext, nullptr, nullptr, dummy);
  return check_launch("project_bwd");
}

extern "C" int eg_project_bwd_adam(float *means, float *quats, float *scales, float *opacities,
                                   const float *viewmat, const float *K, int32_t N, int32_t width, int32_t height,
                                   float eps2d, uint32_t flags, const float *splat, const float *g2d, float *m,
                                   float *v, float *absgrads, eg_adam_hyper hyper, eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && hyper.step >= 1, "bad sizes / step");
  if (N == 0) return EG_OK;
  EG_REQUIRE(means && quats && scales && opacities && viewmat && K && splat && g2d && m && v, "null pointer");
  project_bwd_kernel<true><<<cdiv(N, 256), 256, 0, as_stream(stream)>>>(
      means, quats, scales, opacities, viewmat, K, N, width, height, eps2d, flags, (const float4 *)splat,
      (const float4 *)g2d, nullptr, nullptr, nullptr, nullptr, absgrads, nullptr, nullptr, m, v,
      make_adamk(hyper));
  return check_launch("project_bwd_adam");
}

namespace eg {
int launch_project_bwd_emit(float *means, float *quats, float *scales, float *opacities, const float *viewmat,
                            const float *K, const float *next_viewmat, const float *next_K, int32_t N, int32_t width,
                            int32_t height, float eps2d, uint32_t flags, float *splat, const float *g2d, float *absgrads,
                            float *m, float *v, const eg_adam_hyper &hyper, int32_t *tile_cursor, int32_t seg_cap,
                            uint64_t *keys, int32_t *item_first, int32_t max_items, int32_t *total, int32_t *ticket,
                            hipStream_t st) {
  const int T = cdiv(width, kTile) * cdiv(height, kTile);
  SegOut out;
  out.item_first = item_first; out.max_items = max_items;
  out.total = total; out.ticket = ticket;
  out.item_front = (ticket && (flags & EG_FLAG_FRONT_PREFIX)) ? ticket + 1 : nullptr;
  const bool hoist = N <= 160000;  // up to ~2.5 waves of these threads per SIMD
#define EG_BWD_EMIT(LDS, HO, PE_, SMEM)                                                                             \
  project_bwd_emit_kernel<LDS, HO, PE_><<<cdiv(N, PE_), PE_, SMEM, st>>>(                                          \
      means, quats, scales, opacities, viewmat, K, next_viewmat, next_K, N, width, height, eps2d, flags,           \
      (float4 *)splat, (const float4 *)g2d, absgrads, m, v, make_adamk(hyper), tile_cursor, seg_cap,               \
      (unsigned long long *)keys, out)
  if (pe_small(N)) {  // (small scenes: hoisted loads, 256 Gaussians per workgroup)
    if (2 * T <= 16384) EG_BWD_EMIT(true, true, kPESmall, sizeof(int) * 2 * T); else EG_BWD_EMIT(false, true, kPESmall, 0);
  } else if (2 * T <= 16384) {
    if (hoist) EG_BWD_EMIT(true, true, kPE, sizeof(int) * 2 * T); else EG_BWD_EMIT(true, false, kPE, sizeof(int) * 2 * T);
  } else {
    if (hoist) EG_BWD_EMIT(false, true, kPE, 0); else EG_BWD_EMIT(false, false, kPE, 0);
  }
#undef EG_BWD_EMIT
  return check_launch("project_bwd_emit");
}
}  // namespace eg

extern "C" int eg_backward_fused(float *means, float *quats, float *scales, float *opacities,
                                 const float *viewmat, const float *K, int32_t N, int32_t width, int32_t height,
                                 float eps2d, uint32_t flags, const float *splat, const float *gtstop, float *g2d,
                                 float *v_means, float *v_quats, float *v_scales, float *v_opacities,
                                 float *absgrads, float *m, float *v, const eg_adam_hyper *hyper_host,
                                 eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && width > 0 && height > 0, "bad sizes");
  if (N == 0) return EG_OK;
  EG_REQUIRE(means && quats && scales && opacities && viewmat && K && splat && gtstop && g2d, "null pointer");
  int rc = eg_composite_bwd_footprint(splat, N, width, height, gtstop, g2d, stream);
  if (rc) return rc;
  if (hyper_host)
    return eg_project_bwd_adam(means, quats, scales, opacities, viewmat, K, N, width, height, eps2d, flags, splat,
                               g2d, m, v, absgrads, *hyper_host, stream);
  return eg_project_bwd(means, quats, scales, opacities, viewmat, K, N, width, height, eps2d, flags, splat, g2d,
                        nullptr, nullptr, v_means, v_quats, v_scales, v_opacities, absgrads, stream);
}

extern "C" int eg_adam_multi(float *means, float *scales, float *quats, float *opacities, const float *g_means,
                             const float *g_scales, const float *g_quats, const float *g_opacities, float *m,
                             float *v, int32_t N, eg_adam_hyper hyper, const float *absgrad_inc, float *absgrads,
                             eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && hyper.step >= 1, "bad sizes / step");
  if (N == 0) return EG_OK;
  EG_REQUIRE(means && scales && quats && opacities && g_means && g_scales && g_quats && g_opacities && m && v,
             "null pointer");
  EG_REQUIRE(!absgrads || absgrad_inc, "absgrads needs absgrad_inc");
  const int blocks = min(cdiv(11 * (int64_t)N, 256), 2048);
  adam_multi_kernel<<<blocks, 256, 0, as_stream(stream)>>>(means, scales, quats, opacities, g_means, g_scales,
                                                          g_quats, g_opacities, m, v, N, make_adamk(hyper),
                                                          absgrad_inc, absgrads, RegScale{});
  return check_launch("adam_multi");
}

// One torch.optim.Adam step on ONE flat tensor (the drop-in optimizer of edgegaussians_amd/optim.py: the reference
// keeps four single-tensor optimizers, train_utils.py:50-60, and steps them one by one, train_gaussians.py:104-106):
// adam1's arithmetic, four elements per thread where the pointers allow it.
__global__ void __launch_bounds__(256)
adam_tensor_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, long long n,
                   const AdamK h, int zero_grad) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
  const long long n4 = vec ? n / 4 : 0;
  for (long long i = i0; i < n4; i += stride) {
    float4 pp = ((float4 *)p)[i], gg = ((const float4 *)g)[i], mm = ((float4 *)m)[i], vv = ((float4 *)v)[i];
    adam1(pp.x, gg.x, mm.x, vv.x, 0, h);
    adam1(pp.y, gg.y, mm.y, vv.y, 0, h);
    adam1(pp.z, gg.z, mm.z, vv.z, 0, h);
    adam1(pp.w, gg.w, mm.w, vv.w, 0, h);
    ((float4 *)p)[i] = pp; ((float4 *)m)[i] = mm; ((float4 *)v)[i] = vv;
    if (zero_grad) ((float4 *)g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (long long i = 4 * n4 + i0; i < n; i += stride) {
    float pp = p[i], mm = m[i], vv = v[i];
    adam1(pp, g[i], mm, vv, 0, h);
    p[i] = pp; m[i] = mm; v[i] = vv;
    if (zero_grad) g[i] = 0.f;
  }
}

extern "C" int eg_adam_tensor(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, double lr,
                              double beta1, double beta2, double eps, int32_t step, int32_t zero_grad,
                              eg_stream_t stream) {
  EG_REQUIRE(n >= 0 && step >= 1, "bad size / step");
  if (n == 0) return EG_OK;
  EG_REQUIRE(param && grad && exp_avg && exp_avg_sq, "null pointer");
  eg_adam_hyper hy = {};
  hy.lr_means = lr; hy.beta1 = beta1; hy.beta2 = beta2; hy.eps = eps; hy.step = step;
  hy.group_steps[1] = hy.group_steps[2] = hy.group_steps[3] = -1;
  const int64_t want = ((n + 3) / 4 + 255) / 256;
  const int blocks = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  adam_tensor_kernel<<<blocks, 256, 0, as_stream(stream)>>>(param, grad, exp_avg, exp_avg_sq, (long long)n,
                                                                  make_adamk(hy), zero_grad);
  return check_launch("adam_tensor");
}

namespace eg {
// Adam of a regulariser iteration: gradients scaled by lambda on the fly (RegScale above), null pointer = zero gradient
int launch_adam_regulariser(float *means, float *scales, float *quats, float *opacities, const float *g_means,
                            const float *g_scales, const float *g_quats, float *m, float *v, int32_t N,
                            const eg_adam_hyper &hyper, const float *sum, const float *loss_sum, float loss_sum_host,
                            float factor, float w, int ratio, float *loss_out, hipStream_t st) {
  RegScale rs;
  rs.sum = sum; rs.loss_sum = loss_sum; rs.loss_sum_host = loss_sum_host; rs.factor = factor; rs.w = w;
  rs.n_gauss = (float)N; rs.ratio = ratio; rs.loss_out = loss_out;
  const int blocks = min(cdiv(11 * (int64_t)N, 256), 2048);
  adam_multi_kernel<<<blocks, 256, 0, st>>>(means, scales, quats, opacities, g_means, g_scales, g_quats, nullptr, m, v, N,
                                            make_adamk(hyper), nullptr, nullptr, rs);
  return check_launch("adam_regulariser");
}
}  // namespace eg

extern "C" int eg_adam_emit(float *means, float *scales, float *quats, float *opacities, const float *g_means,
                            const float *g_scales, const float *g_quats, const float *g_opacities, float *m, float *v,
                            int32_t N, eg_adam_hyper hyper, const float *absgrad_inc, float *absgrads,
                            const float *next_viewmat, const float *next_K, int32_t width, int32_t height,
                            uint32_t flags, float *splat, int32_t *tile_cursor, int32_t seg_cap, uint64_t *keys,
                            int32_t *item_first, int32_t max_items, int32_t *total, int32_t *ticket,
                            eg_stream_t stream) {
  EG_REQUIRE(N > 0 && hyper.step >= 1 && width > 0 && height > 0 && seg_cap > 0 && max_items > 0, "bad sizes / step");
  EG_REQUIRE(means && scales && quats && opacities && g_means && g_scales && g_quats && g_opacities && m && v &&
                 next_viewmat && next_K && splat && tile_cursor && keys && item_first && total,
             "null pointer");
  EG_REQUIRE(!absgrads || absgrad_inc, "absgrads needs absgrad_inc");
  const int T = cdiv(width, kTile) * cdiv(height, kTile);
  // (ticket == NULL: no scan tail -- for a following eg_train_step on a tile grid of <= 2048 tiles, whose sort kernel
  // forms the tile prefix itself)
  EG_REQUIRE(ticket || T <= kPrefixHereMaxTiles, "ticket may be NULL only on tile grids of <= 2560 tiles");
  EG_REQUIRE((int64_t)T * seg_cap < (1ll << 31), "T * seg_cap must fit 31 bits");
  SegOut out;
  out.item_first = item_first; out.max_items = max_items;
  out.total = total; out.ticket = ticket;
  out.item_front = (ticket && (flags & EG_FLAG_FRONT_PREFIX)) ? ticket + 1 : nullptr;
  hipStream_t st = as_stream(stream);
#define EG_ADAM_EMIT(LDS, PE_, SMEM)                                                                                    \
  adam_emit_kernel<LDS, PE_><<<cdiv(N, PE_), PE_, SMEM, st>>>(                                                          \
      means, scales, quats, opacities, g_means, g_scales, g_quats, g_opacities, m, v, N, make_adamk(hyper), absgrad_inc, \
      absgrads, next_viewmat, next_K, width, height, flags, (float4 *)splat, tile_cursor, seg_cap,                     \
      (unsigned long long *)keys, out)
  if (2 * T <= 16384) {
    if (pe_small(N)) EG_ADAM_EMIT(true, kPESmall, sizeof(int) * 2 * T); else EG_ADAM_EMIT(true, kPE, sizeof(int) * 2 * T);
  } else {
    if (pe_small(N)) EG_ADAM_EMIT(false, kPESmall, 0); else EG_ADAM_EMIT(false, kPE, 0);
  }
#undef EG_ADAM_EMIT
  return check_launch("adam_emit");
}

extern "C" int eg_absgrad_accum(const float *means2d_absgrad, int32_t N, float *absgrads, eg_stream_t stream) {
  EG_REQUIRE(N >= 0, "bad sizes");
  if (N == 0) return EG_OK;
  EG_REQUIRE(means2d_absgrad && absgrads, "null pointer");
  absgrad_accum_kernel<<<cdiv(N, 256), 256, 0, as_stream(stream)>>>((const float2 *)means2d_absgrad, N, absgrads);
  return check_launch("absgrad_accum");
}
