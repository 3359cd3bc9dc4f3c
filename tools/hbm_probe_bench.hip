// Measures, on the GPU it runs on, (1) device-to-device stream copy bandwidth and (2) the rate of isolated random one-byte
// probes into a 3 GiB table with 18 independent probes in flight per thread: the access pattern of the k-mer filter.
// (2) is the practical ceiling for the filter kernel: every probe moves one 64-byte line.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_probe_bench tools/hbm_probe_bench.hip && ./hbm_probe_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = src[i];
}
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
template <int P>
__global__ void k_probe(const uint8_t *__restrict__ tab, uint64_t mask, uint32_t *__restrict__ out, uint32_t n) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  uint8_t v[P];
#pragma unroll
  for (int p = 0; p < P; ++p) v[p] = tab[mix((uint64_t)t * P + p + 1) & mask];
  uint32_t s = 0;
#pragma unroll
  for (int p = 0; p < P; ++p) s += v[p];
  out[t] = s;
}
int main() {
  const size_t bytes = (size_t)3 << 30;
  uint8_t *a, *b; uint32_t *out;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes));
  const uint32_t n = 8u << 20;
  CK(hipMalloc(&out, (size_t)n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  for (int it = 0; it < 3; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_copy, dim3(256 * 32), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, bytes / 16);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("stream copy 3 GiB: %.3f ms  read+write %.2f TB/s\n", ms, 2.0 * bytes / (ms * 1e-3) / 1e12);
  }
  const uint64_t masks[2] = {((uint64_t)2 << 30) - 1, ((uint64_t)256 << 20) - 1};   // 2 GiB span (power of two inside the table), 256 MiB span
  for (int m = 0; m < 2; ++m)
    for (int it = 0; it < 3; ++it) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_probe<18>, dim3((n + 255) / 256), dim3(256), 0, 0, a, masks[m], out, n);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      printf("random probes over %4llu MiB, 18 per thread, %u threads: %.3f ms  %.2f G probes/s  = %.2f TB/s of 64-byte lines\n",
             (unsigned long long)((masks[m] + 1) >> 20), n, ms, 18.0 * n / (ms * 1e-3) / 1e9, 64.0 * 18.0 * n / (ms * 1e-3) / 1e12);
    }
  return 0;
}
