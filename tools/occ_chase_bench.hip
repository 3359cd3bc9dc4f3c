// What the memory system gives the access pattern of the Occ search kernels: every lane follows its own chain of DEPENDENT
// 32-byte block loads (the address of step t+1 comes out of the data of step t), P independent chains per lane, W wavefronts
// per SIMD, over tables from L2-sized to HBM-sized.  Prints loads/s: the request-rate ceiling the search kernel can be held against.
//   hipcc --offload-arch=gfx950 -O3 -o occ_chase_bench tools/occ_chase_bench.hip && ./occ_chase_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int P>
__global__ void __launch_bounds__(64) k_chase(const uint4 *__restrict__ tab, uint32_t mask_blocks, int steps, uint32_t *__restrict__ out) {
  const uint32_t t = blockIdx.x * 64 + threadIdx.x;
  const uint64_t c0 = clock64(), w0 = wall_clock64();
  uint32_t x[P];
#pragma unroll
  for (int p = 0; p < P; ++p) x[p] = mix32(t * P + p + 1);
  for (int s = 0; s < steps; ++s) {
    uint4 a[P], b[P];
#pragma unroll
    for (int p = 0; p < P; ++p) { const uint4 *q = tab + 2 * (size_t)(x[p] & mask_blocks); a[p] = q[0]; b[p] = q[1]; }
#pragma unroll
    for (int p = 0; p < P; ++p) x[p] = mix32(x[p] + a[p].x + a[p].w + b[p].y + b[p].z + __popc(a[p].y) + __popc(b[p].w));
  }
  uint32_t r = 0;
#pragma unroll
  for (int p = 0; p < P; ++p) r += x[p];
  out[t] = r;
  if (t == 0) { out[1 << 20] = (uint32_t)(clock64() - c0); out[(1 << 20) + 1] = (uint32_t)(wall_clock64() - w0); }
}
template <int P>
static void run(const uint4 *tab, uint32_t *out, size_t span_bytes, int waves_per_simd) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int steps = 2000, blocks = 256 * 4 * waves_per_simd;
  float best = 1e30f;
  for (int it = 0; it < 3; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_chase<P>, dim3(blocks), dim3(64), 0, 0, tab, (uint32_t)(span_bytes / 32 - 1), steps, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double loads = (double)blocks * 64 * steps * P;
  uint32_t tk[2]; CK(hipMemcpy(tk, out + (1 << 20), 8, hipMemcpyDeviceToHost));
  printf("[clock64 %.1f ticks/us, wall_clock64 %.1f ticks/us] ", tk[0] / (best * 1e3), tk[1] / (best * 1e3));
  printf("span %7.1f MiB  %d chain(s)/lane  %d waves/SIMD: %8.3f ms  %6.2f G block loads/s  (%.2f us per dependent step)\n",
         span_bytes / 1048576.0, P, waves_per_simd, best, loads / (best * 1e-3) / 1e9, best * 1e3 / steps);
}
int main() {
  const size_t bytes = (size_t)1 << 30;
  uint4 *tab; uint32_t *out;
  CK(hipMalloc(&tab, bytes)); CK(hipMemset(tab, 0x5a, bytes));
  CK(hipMalloc(&out, (size_t)((1 << 20) + 16) * 4));
  const size_t spans[] = {(size_t)2 << 20, (size_t)8 << 20};
  for (size_t sp : spans) {
    for (int w : {4, 8}) run<1>(tab, out, sp, w);
    for (int w : {4, 8}) run<2>(tab, out, sp, w);
    run<4>(tab, out, sp, 4);
  }
  return 0;
}
