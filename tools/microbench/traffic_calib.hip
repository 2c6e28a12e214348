// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the compositing forward
// (MI355X_MICROARCH.md, HBM: "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in
// your own access pattern").  Every kernel moves a KNOWN number of bytes once; tools/traffic_calib.py runs this binary
// under rocprofv3 --pmc and prints counted / known per pattern.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/traffic_calib tools/microbench/traffic_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr size_t kBytes = 64ull << 20;  // every pattern touches 64 MiB exactly once

__global__ void cal_stream16(const float4 *__restrict__ p, float *sink, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  float acc = 0.f;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 123.456f) *sink = acc;
}
__global__ void cal_stream4(const float *__restrict__ p, float *sink, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  float acc = 0.f;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 123.456f) *sink = acc;
}
// the record gather of the forward: a PAIR of lanes reads the two 16-byte halves of a 32-byte record at a scattered index
__global__ void cal_gather32(const float4 *__restrict__ p, float *sink, unsigned nrec) {  // nrec: power of two
  size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  float acc = 0.f;
  for (; t < 2ull * nrec; t += (size_t)gridDim.x * blockDim.x) {
    const unsigned r = ((unsigned)(t >> 1) * 2654435761u) & (nrec - 1u);  // odd multiplier: a permutation of the records
    float4 v = p[2ull * r + (t & 1)];
    acc += v.x + v.w;
  }
  if (acc == 123.456f) *sink = acc;
}
// granule reads: 8 bytes per lane, device scope (sc1), coalesced over the wave
__global__ void cal_sc1_load8(const unsigned long long *p, float *sink, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  unsigned long long acc = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += __hip_atomic_load(&p[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (acc == 0x123456789ull) *sink = 1.f;
}
__global__ void cal_sc1_store8(unsigned long long *p, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) __hip_atomic_store(&p[i], (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the per-pixel record of the forward: 12 bytes per lane, contiguous over the wave
struct Rec12 { float a; int b; unsigned c; };
__global__ void cal_store12(Rec12 *p, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) { Rec12 r; r.a = (float)i; r.b = (int)i; r.c = 7u; p[i] = r; }
}
__global__ void cal_store16(float4 *p, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4((float)i, 0.f, 1.f, 2.f);
}

int main() {
  void *buf; float *sink;
  if (hipMalloc(&buf, kBytes) != hipSuccess || hipMalloc((void **)&sink, 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  (void)hipMemset(buf, 0, kBytes);
  const int grid = 256 * 8, block = 256;
  for (int rep = 0; rep < 3; ++rep) {
    cal_stream16<<<grid, block>>>((const float4 *)buf, sink, kBytes / 16);
    cal_stream4<<<grid, block>>>((const float *)buf, sink, kBytes / 4);
    cal_gather32<<<grid, block>>>((const float4 *)buf, sink, (unsigned)(kBytes / 32));
    cal_sc1_load8<<<grid, block>>>((const unsigned long long *)buf, sink, kBytes / 8);
    cal_sc1_store8<<<grid, block>>>((unsigned long long *)buf, kBytes / 8);
    cal_store12<<<grid, block>>>((Rec12 *)buf, kBytes / 12);
    cal_store16<<<grid, block>>>((float4 *)buf, kBytes / 16);
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("run failed\n"); return 1; }
  printf("known bytes per launch: %zu (store12: %zu)\n", kBytes, (kBytes / 12) * 12);
  return 0;
}
