// tools/pin_bench.hip -- what pinned host memory costs to get on the box: hipHostMalloc against an anonymous mapping (huge pages asked for) made known with
// hipHostRegister; the copy rate into it; a kernel reading and writing it through the HOST pointer (what the staging kernels of csrc/fq_device.hip do).
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/pin_bench tools/pin_bench.hip && gpurun -- tools/bin/pin_bench
// Measured (256 MB): hipHostMalloc 34-51 ms, mmap + hipHostRegister 10.6 ms, both copy at 57 GB/s, the same address on the device.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main_alloc() {
  hipSetDevice(0);
  void *d; hipMalloc(&d, 512u << 20);
  hipStream_t s; hipStreamCreate(&s);
  const size_t N = 256u << 20;
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now(); void *p = nullptr; hipError_t e = hipHostMalloc(&p, N, hipHostMallocDefault); double t1 = now();
    hipMemcpyAsync(p, d, N, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); double t2 = now();
    hipMemcpyAsync(p, d, N, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); double t3 = now();
    hipHostFree(p); double t4 = now();
    printf("hipHostMalloc 256MB: %d alloc %.1f ms  copy1 %.1f ms copy2 %.1f ms (%.1f GB/s) free %.1f ms\n", (int)e, t1 - t0, t2 - t1, t3 - t2, N / (t3 - t2) / 1e6, t4 - t3);
  }
  for (int huge = 0; huge < 2; ++huge) for (int rep = 0; rep < 2; ++rep) {
    double t0 = now();
    void *p = mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (huge) madvise(p, N, MADV_HUGEPAGE);
    double t1 = now();
    hipError_t e = hipHostRegister(p, N, hipHostRegisterDefault); double t2 = now();
    hipMemcpyAsync(p, d, N, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); double t3 = now();
    hipMemcpyAsync(p, d, N, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); double t4 = now();
    hipHostUnregister(p); munmap(p, N); double t5 = now();
    printf("mmap%s + hipHostRegister 256MB: %d mmap %.1f register %.1f ms copy1 %.1f copy2 %.1f (%.1f GB/s) unregister+unmap %.1f\n", huge ? "+THP" : "", (int)e, t1 - t0, t2 - t1, t3 - t2, t4 - t3, N / (t4 - t3) / 1e6, t5 - t4);
  }
  FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r"); char b[128] = {0}; if (f) { fgets(b, 127, f); fclose(f); } printf("THP: %s", b);
  return 0;
}
__global__ void k_touch(const unsigned *src, unsigned *dst, size_t n, unsigned *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned acc = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) { acc += src[i]; dst[i] = src[i] + 1; }
  atomicAdd(out, acc);
}
int main_kernel() {
  hipSetDevice(0);
  hipStream_t s; hipStreamCreate(&s);
  unsigned *d_out; hipMalloc(&d_out, 4);
  const size_t N = 64u << 20;
  for (int mode = 0; mode < 3; ++mode) {
    void *p = nullptr;
    double t0 = now();
    if (mode == 0) hipHostMalloc(&p, N, hipHostMallocDefault);
    else { p = mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); if (mode == 2) madvise(p, N, MADV_HUGEPAGE); hipError_t e = hipHostRegister(p, N, hipHostRegisterDefault); if (e) printf("register error %d\n", (int)e); }
    double t1 = now();
    void *dp = nullptr; hipError_t e2 = hipHostGetDevicePointer(&dp, p, 0);
    unsigned *h = (unsigned *)p; const size_t n = N / 8;
    for (size_t i = 0; i < n; ++i) h[i] = (unsigned)i;
    hipMemsetAsync(d_out, 0, 4, s);
    k_touch<<<256, 256, 0, s>>>(h, h + n, n, d_out);      // the HOST pointer, as k_copy_bytes uses it
    unsigned got = 0; hipMemcpyAsync(&got, d_out, 4, hipMemcpyDeviceToHost, s); hipError_t e3 = hipStreamSynchronize(s);
    unsigned want = 0; for (size_t i = 0; i < n; ++i) want += (unsigned)i;
    bool ok = true; for (size_t i = 0; i < n; i += 4097) if (h[n + i] != (unsigned)i + 1) ok = false;
    printf("mode %d: alloc %.1f ms  host %p device %p (%d)  kernel over the host pointer: sync %d sum %s writes %s\n", mode, t1 - t0, p, dp, (int)e2, (int)e3, got == want ? "ok" : "WRONG", ok ? "ok" : "WRONG");
    if (mode == 0) hipHostFree(p); else { hipHostUnregister(p); munmap(p, N); }
  }
  return 0;
}

int main() { main_alloc(); return main_kernel(); }
