// Micro-benchmarks behind DESIGN.md's issue-side roofline (gfx950): how many cycles a wave64 VALU instruction
// of each class occupies its SIMD, measured with s_memtime around long unrolled runs at 1..8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o issue_rates issue_rates.hip && ./issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int KIND>
__global__ void __launch_bounds__(256) rate_kernel(float *out, long long *cycles, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
  float a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const float b = 1.0001f, c = 1e-4f;
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const v2f pb = {b, b}, pc = {c, c};
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {  // v_fma_f32, 8 independent chains
      REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    } else if (KIND == 1) {  // v_pk_fma_f32, 4 chains
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
    } else if (KIND == 2) {  // v_exp_f32
      REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                         "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 3) {  // v_cndmask + v_cmp pairs
      REP16(asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                         "v_cmp_gt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_gt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
    } else if (KIND == 4) {  // v_mul_f32 (non-fma plain op)
      REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (KIND == 5) {  // v_rcp_f32
      REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                         "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// LDS: broadcast ds_read_b128 (all lanes the same address), as the compositing walk issues them
__global__ void __launch_bounds__(256) lds_kernel(float *out, long long *cycles, int iters) {
  __shared__ float4 s[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) s[i] = make_float4(i, i + 1, i + 2, i + 3);
  __syncthreads();
  float acc = 0.f;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) {
      const float4 v = s[(i * 64 + u) & 1023];
      acc += v.x + v.w;
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// dependent global loads: L2-hit pointer chase (plain) and device-scope (sc1) chase; one lane per workgroup
template <bool DEV>
__global__ void chase_kernel(const int *next, int *out, long long *cycles, int hops) {
  int p = blockIdx.x * 64;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < hops; ++i)
    p = DEV ? __hip_atomic_load(&next[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : next[p];
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x] = p;
  cycles[blockIdx.x] = t1 - t0;
}

// same-address returning atomics from G workgroups (one lane each)
__global__ void atomic_kernel(int *ctr, int *out) { if (threadIdx.x == 0) out[blockIdx.x] = atomicAdd(ctr, 1); }
__global__ void atomic_spread_kernel(int *ctr, int *out) { if (threadIdx.x == 0) out[blockIdx.x] = atomicAdd(ctr + 32 * (blockIdx.x & 255), 1); }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  float *out; long long *cyc;
  CK(hipMalloc(&out, 4 << 20)); CK(hipMalloc(&cyc, 8 * 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_cmp+v_cndmask", "v_mul_f32", "v_rcp_f32"};
  const int iters = 200;
  for (int kind = 0; kind < 6; ++kind)
    for (int wgs_per_cu = 1; wgs_per_cu <= 8; wgs_per_cu *= 2) {  // 256-thread WG = one wave per SIMD
      const int grid = 256 * wgs_per_cu;
      auto launch = [&]() {
        switch (kind) {
          case 0: rate_kernel<0><<<grid, 256>>>(out, cyc, iters); break;
          case 1: rate_kernel<1><<<grid, 256>>>(out, cyc, iters); break;
          case 2: rate_kernel<2><<<grid, 256>>>(out, cyc, iters); break;
          case 3: rate_kernel<3><<<grid, 256>>>(out, cyc, iters); break;
          case 4: rate_kernel<4><<<grid, 256>>>(out, cyc, iters); break;
          default: rate_kernel<5><<<grid, 256>>>(out, cyc, iters); break;
        }
      };
      launch(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double instr_per_wave = (double)iters * 16 * 8;
      const double waves_per_simd = wgs_per_cu;
      // SIMD-cycles per wave-instruction at 2.4 GHz from the wall time (all SIMDs busy with waves_per_simd waves)
      const double cyc_per_instr = ms * 1e-3 * 2.4e9 / (instr_per_wave * waves_per_simd);
      printf("%-16s waves/SIMD %d  time %.3f ms  SIMD-cycles per wave-instruction %.2f\n", names[kind], wgs_per_cu, ms, cyc_per_instr);
    }
  for (int wgs_per_cu = 1; wgs_per_cu <= 8; wgs_per_cu *= 2) {
    const int grid = 256 * wgs_per_cu;
    lds_kernel<<<grid, 256>>>(out, cyc, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); lds_kernel<<<grid, 256>>>(out, cyc, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("ds_read_b128 broadcast  waves/SIMD %d  CU-cycles per wave-read %.2f\n", wgs_per_cu,
           ms * 1e-3 * 2.4e9 / ((double)iters * 64 * 4 * wgs_per_cu));
  }
  // pointer chase
  const int n = 1 << 20;
  std::vector<int> h(n);
  for (int i = 0; i < n; ++i) h[i] = (int)(((long long)i * 7919 + 64) % n);
  int *next, *iout; CK(hipMalloc(&next, n * 4)); CK(hipMalloc(&iout, 4096 * 4));
  CK(hipMemcpy(next, h.data(), n * 4, hipMemcpyHostToDevice));
  for (int dev = 0; dev < 2; ++dev)
    for (int rep = 0; rep < 2; ++rep) {
      if (dev) chase_kernel<true><<<256, 64>>>(next, iout, cyc, 256); else chase_kernel<false><<<256, 64>>>(next, iout, cyc, 256);
      CK(hipDeviceSynchronize());
      long long hc[256]; CK(hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost));
      double s = 0; for (int i = 0; i < 256; ++i) s += hc[i];
      if (rep) printf("dependent global load (%s, 4 MB working set): %.0f shader-clock cycles per hop (s_memtime ticks)\n",
                      dev ? "device scope" : "plain", s / 256 / 256);
    }
  int *ctr; CK(hipMalloc(&ctr, 1 << 20)); CK(hipMemset(ctr, 0, 1 << 20));
  for (int G : {256, 1024, 4096}) {
    atomic_kernel<<<G, 64>>>(ctr, iout); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); atomic_kernel<<<G, 64>>>(ctr, iout); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventRecord(e0)); atomic_spread_kernel<<<G, 64>>>(ctr, iout); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms2; CK(hipEventElapsedTime(&ms2, e0, e1));
    printf("returning atomicAdd from %d workgroups: same address %.1f us, 256 addresses %.1f us\n", G, ms * 1e3, ms2 * 1e3);
  }
  return 0;
}
