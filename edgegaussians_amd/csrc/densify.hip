// a8-a10: densify / cull row movement and the not-projecting vote, on device.
//
// Replaces the torch boolean-index / cat / CPU loops of the reference:
//   cull_gaussians + remove_from_optim   edge_gs.py:384-423  -> eg_mask_scan + eg_compact_rows
//   dup_gaussians + dup_in_optim         edge_gs.py:431-474  -> eg_mask_scan + eg_append_rows
//   cull_gaussians_not_projecting        edge_gs.py:578-601  -> eg_project_hits
// All three are streaming passes (HBM bound); they run at 22 epoch boundaries out of 400 epochs.
#include "common.h"

namespace eg {

// exclusive scan of a byte mask, single workgroup of 1024 threads (N <= a few million)
__global__ void __launch_bounds__(1024)
mask_scan_kernel(const uint8_t *__restrict__ keep, int N, int *__restrict__ positions, int *__restrict__ count) {
  __shared__ int wave_sums[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < N; base += 1024) {
    const int i = base + tid;
    const int c = (i < N && keep[i]) ? 1 : 0;
    const unsigned long long bal = __ballot(c);
    const int excl_w = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_sums[wv] = __popcll(bal);
    __syncthreads();
    int pre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int s = wave_sums[w];
      pre += (w < wv) ? s : 0;
      tot += s;
    }
    if (i < N) positions[i] = carry + pre + excl_w;
    __syncthreads();
    if (tid == 0) carry += tot;
    __syncthreads();
  }
  if (tid == 0) count[0] = carry;
}

__global__ void __launch_bounds__(256)
compact_rows_kernel(const float *__restrict__ in, const uint8_t *__restrict__ keep,
                    const int *__restrict__ positions, int N, int dim, float *__restrict__ out) {
  const size_t total = (size_t)N * dim;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / dim), c = (int)(e - (size_t)r * dim);
    if (keep[r]) out[(size_t)positions[r] * dim + c] = in[e];
  }
}

__global__ void __launch_bounds__(256)
append_rows_kernel(const float *__restrict__ in, const uint8_t *__restrict__ sel, const int *__restrict__ positions,
                   int N, int n_sel, int dim, int copies, const float *__restrict__ noise, int fill_zero,
                   float *__restrict__ out_tail) {
  const size_t total = (size_t)N * dim;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / dim), c = (int)(e - (size_t)r * dim);
    if (!sel[r]) continue;
    const float v = fill_zero ? 0.f : in[e];
    for (int k = 0; k < copies; ++k) {
      const size_t o = ((size_t)k * n_sel + positions[r]) * dim + c;
      out_tail[o] = noise ? v + noise[o] : v;
    }
  }
}

// one thread per Gaussian, loop over views: P = K [R|t] (3x4), round half to even like torch.round
__global__ void __launch_bounds__(256)
project_hits_kernel(const float *__restrict__ means, int N, const float *__restrict__ P, int V,
                    const uint8_t *__restrict__ edge_masks, int width, int height, int *__restrict__ hits) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= N) return;
  const float x = means[3 * g], y = means[3 * g + 1], z = means[3 * g + 2];
  int h = 0;
  for (int v = 0; v < V; ++v) {
    const float *p = P + 12 * v;
    // same association order as the reference's matmul of [x y z 1] with P^T (fp32)
    const float a = p[0] * x + p[1] * y + p[2] * z + p[3];
    const float b = p[4] * x + p[5] * y + p[6] * z + p[7];
    const float c = p[8] * x + p[9] * y + p[10] * z + p[11];
    const float u = rintf(a / c), w = rintf(b / c);
    if (u >= 0.f && u < (float)width && w >= 0.f && w < (float)height)
      h += edge_masks[((size_t)v * height + (int)w) * width + (int)u] ? 1 : 0;
  }
  hits[g] += h;
}

// filter_by_projection twin: x = K (R X + t) in that association order, np.round (half to even),
// sum of the float edge strengths over the views the point lands in
__global__ void __launch_bounds__(256)
project_visibility_kernel(const float *__restrict__ means, int N, const float *__restrict__ cams, int V,
                          const float *__restrict__ edge_maps, int width, int height, double *__restrict__ visib) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= N) return;
  const float x = means[3 * g], y = means[3 * g + 1], z = means[3 * g + 2];
  double acc = 0.0;  // the reference accumulates the float32 edge strengths in a float64 matrix
  for (int v = 0; v < V; ++v) {
    const float *K = cams + 21 * v, *R = K + 9, *t = K + 18;
    const float cx = R[0] * x + R[1] * y + R[2] * z + t[0];
    const float cy = R[3] * x + R[4] * y + R[5] * z + t[1];
    const float cz = R[6] * x + R[7] * y + R[8] * z + t[2];
    const float a = K[0] * cx + K[1] * cy + K[2] * cz;
    const float b = K[3] * cx + K[4] * cy + K[5] * cz;
    const float c = K[6] * cx + K[7] * cy + K[8] * cz;
    const float u = rintf(a / c), w = rintf(b / c);
    if (u >= 0.f && u < (float)width && w >= 0.f && w < (float)height)
      acc += (double)edge_maps[((size_t)v * height + (int)w) * width + (int)u];
  }
  visib[g] += acc;
}

// bg_edge_ratio weight map (edge_gs.py:298-314 in weight-map form): w_p = [gt_p >= thr] / n_edge + [p sampled] / n_sel.
// The sample is the first n_sel entries of a random permutation (distinct indices), drawn by the caller.
__global__ void __launch_bounds__(256)
ratio_wmap_edge_kernel(const float *__restrict__ gt, float thr, float inv_edge, int HW, float *__restrict__ out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < HW) out[p] = (gt[p] >= thr) ? inv_edge : 0.f;
}
__global__ void __launch_bounds__(256)
ratio_wmap_sel_kernel(const long long *__restrict__ perm, int n_sel, int HW, float inv_sel, float *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_sel) out[(int)(perm[i] % HW)] += inv_sel;  // distinct pixels: no atomics
}

// The same weight map with the sample drawn inside the kernel: pixel p < n_bg is sampled iff pi(p) < n_sel for a
// keyed pseudo-random PERMUTATION pi of [0, n_bg) -- a 6-round Feistel network on 2 * half_bits >= log2(n_bg) bits,
// cycle-walked back into the domain (a bijection restricted to the values that land inside stays a bijection).
// Exactly n_sel distinct pixels, one launch, no sort (torch.randperm on the device is a 7-launch radix sort:
// ~55 us of device time and ~36 us of host time per step of the 'bg_edge_ratio' strategy).
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
__global__ void __launch_bounds__(256)
ratio_wmap_seeded_kernel(const float *__restrict__ gt, float thr, float inv_edge, uint32_t n_bg, uint32_t n_sel,
                         float inv_sel, uint32_t key_lo, uint32_t key_hi, int half_bits, int HW,
                         float *__restrict__ out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  float w = (gt[p] >= thr) ? inv_edge : 0.f;
  if ((uint32_t)p < n_bg && n_sel > 0) {
    const uint32_t mask = (1u << half_bits) - 1u;
    uint32_t v = (uint32_t)p;
    do {
      uint32_t l = v >> half_bits, r = v & mask;
#pragma unroll
      for (int round = 0; round < 6; ++round) {
        const uint32_t f = mix32(r ^ mix32(key_lo + 0x9e3779b9u * (uint32_t)(round + 1)) ^ key_hi) & mask;
        const uint32_t t = l ^ f;
        l = r;
        r = t;
      }
      v = (l << half_bits) | r;
    } while (v >= n_bg);
    if (v < n_sel) w += inv_sel;
  }
  out[p] = w;
}
}  // namespace eg

using namespace eg;

extern "C" int eg_mask_scan(const uint8_t *keep, int32_t N, int32_t *positions, int32_t *count, eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && count, "bad arguments");
  EG_REQUIRE(N == 0 || (keep && positions), "null pointer");
  mask_scan_kernel<<<1, 1024, 0, as_stream(stream)>>>(keep, N, positions, count);
  return check_launch("mask_scan");
}

extern "C" int eg_compact_rows(const float *in, const uint8_t *keep, const int32_t *positions, int32_t N,
                               int32_t dim, float *out, eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && dim > 0, "bad sizes");
  if (N == 0) return EG_OK;
  EG_REQUIRE(in && keep && positions && out, "null pointer");
  const int blocks = min(cdiv((int64_t)N * dim, 256), 2048);
  compact_rows_kernel<<<blocks, 256, 0, as_stream(stream)>>>(in, keep, positions, N, dim, out);
  return check_launch("compact_rows");
}

extern "C" int eg_append_rows(const float *in, const uint8_t *sel, const int32_t *positions, int32_t N,
                              int32_t n_sel, int32_t dim, int32_t copies, const float *noise, float fill_zero,
                              float *out_tail, eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && dim > 0 && copies >= 0 && n_sel >= 0, "bad sizes");
  if (N == 0 || n_sel == 0 || copies == 0) return EG_OK;
  EG_REQUIRE(in && sel && positions && out_tail, "null pointer");
  const int blocks = min(cdiv((int64_t)N * dim, 256), 2048);
  append_rows_kernel<<<blocks, 256, 0, as_stream(stream)>>>(in, sel, positions, N, n_sel, dim, copies, noise,
                                                            fill_zero != 0.f, out_tail);
  return check_launch("append_rows");
}

extern "C" int eg_project_hits(const float *means, int32_t N, const float *P, int32_t V, const uint8_t *edge_masks,
                               int32_t width, int32_t height, int32_t *hits, eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && V >= 0 && width > 0 && height > 0, "bad sizes");
  if (N == 0 || V == 0) return EG_OK;
  EG_REQUIRE(means && P && edge_masks && hits, "null pointer");
  project_hits_kernel<<<cdiv(N, 256), 256, 0, as_stream(stream)>>>(means, N, P, V, edge_masks, width, height, hits);
  return check_launch("project_hits");
}

extern "C" int eg_project_visibility(const float *means, int32_t N, const float *cams, int32_t V,
                                     const float *edge_maps, int32_t width, int32_t height, double *visib,
                                     eg_stream_t stream) {
  EG_REQUIRE(N >= 0 && V >= 0 && width > 0 && height > 0, "bad sizes");
  if (N == 0 || V == 0) return EG_OK;
  EG_REQUIRE(means && cams && edge_maps && visib, "null pointer");
  project_visibility_kernel<<<cdiv(N, 256), 256, 0, as_stream(stream)>>>(means, N, cams, V, edge_maps, width,
                                                                        height, visib);
  return check_launch("project_visibility");
}

// weight map of the 'bg_edge_ratio' strategy from the edge image and a caller-drawn random permutation (int64
// indices, at least n_sel of them): out[p] = [gt_p >= thr] / n_edge + [p among the first n_sel] / n_sel.
extern "C" int eg_ratio_wmap(const float *gt, float thr, int32_t n_edge, const int64_t *perm, int32_t n_sel,
                             int32_t HW, float *out, eg_stream_t stream) {
  EG_REQUIRE(HW > 0 && n_sel >= 0 && n_edge >= 0 && gt && out && (perm || n_sel == 0), "bad arguments");
  hipStream_t st = as_stream(stream);
  ratio_wmap_edge_kernel<<<cdiv(HW, 256), 256, 0, st>>>(gt, thr, 1.f / (float)(n_edge > 0 ? n_edge : 1), HW, out);
  if (n_sel > 0)
    ratio_wmap_sel_kernel<<<cdiv(n_sel, 256), 256, 0, st>>>((const long long *)perm, n_sel, HW, 1.f / (float)n_sel, out);
  return check_launch("ratio_wmap");
}

// ... and with the sample drawn on the device from a 64-bit seed: n_sel distinct pixels among the first n_bg flat
// indices (the reference's quirk, edge_gs.py:303-310: positions in the list of background pixels are unravelled
// as if they were pixel indices), uniformly at random; a different seed per call gives a fresh sample.
extern "C" int eg_ratio_wmap_seeded(const float *gt, float thr, int32_t n_edge, int32_t n_bg, int32_t n_sel,
                                    uint64_t seed, int32_t HW, float *out, eg_stream_t stream) {
  EG_REQUIRE(HW > 0 && n_sel >= 0 && n_edge >= 0 && n_bg >= 0 && n_bg <= HW && gt && out, "bad arguments");
  n_sel = min(n_sel, n_bg);
  int half_bits = 1;
  while (half_bits < 16 && (1ull << (2 * half_bits)) < (uint64_t)n_bg) ++half_bits;
  ratio_wmap_seeded_kernel<<<cdiv(HW, 256), 256, 0, as_stream(stream)>>>(
      gt, thr, 1.f / (float)(n_edge > 0 ? n_edge : 1), (uint32_t)n_bg, (uint32_t)n_sel,
      n_sel > 0 ? 1.f / (float)n_sel : 0.f, (uint32_t)seed, (uint32_t)(seed >> 32), half_bits, HW, out);
  return check_launch("ratio_wmap_seeded");
}

// C such maps by one native call (round 6: a run of steps draws one map per fifth step BEFORE it is enqueued -- 13 us of
// host time per draw through the binding, with the GPU idle behind a read-back at the head of a window): map c from
// gts[c] with (n_edge[c], n_bg[c], n_sel[c], seeds[c]) into out + c * HW.  Host arrays; the same kernel, the same maps.
extern "C" int eg_ratio_wmaps_seeded(int32_t C, const float *const *gts, float thr, const int32_t *n_edge, const int32_t *n_bg,
                                     const int32_t *n_sel, const uint64_t *seeds, int32_t HW, float *out, eg_stream_t stream) {
  EG_REQUIRE(C >= 0 && HW > 0 && (C == 0 || (gts && n_edge && n_bg && n_sel && seeds && out)), "bad arguments");
  for (int c = 0; c < C; ++c) {
    const int rc = eg_ratio_wmap_seeded(gts[c], thr, n_edge[c], n_bg[c], n_sel[c], seeds[c], HW, out + (size_t)HW * c, stream);
    if (rc) return rc;
  }
  return EG_OK;
}

