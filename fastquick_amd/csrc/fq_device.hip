// fq_device.hip -- gfx950 (MI355X) backend: __global__ wrappers, stream compaction / scan kernels
// with 64-wide wavefront ballots, memory + HIP-event timing.  Written for CDNA4 only.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <condition_variable>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "fq_backend.h"

namespace fqdev {

// Backend state belongs to an alignment context (fqdev::State, created by fq_ctx_create and freed by fq_ctx_destroy): the
// compute stream, a copy stream for input prefetch, the timing events and the scan / compaction temporaries.  A host thread that
// enters the library binds the context's state (fqdev::bind) and every call below works on the bound state, so several
// contexts can be driven concurrently (one thread each) and their kernels overlap; nothing device-side lives in thread-local
// storage or outlives its context.
struct Pending { int kid; hipEvent_t a, b; };
struct State {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t main_stream = nullptr, aux_stream = nullptr;   // `stream` is the one backend calls go to: main_stream, or aux_stream between stream_aux(1) and stream_aux(0)
  hipEvent_t fork_ev = nullptr, join_ev = nullptr;
  hipEvent_t marks[4] = {nullptr, nullptr, nullptr, nullptr};   // stream_mark / stream_wait_mark
  hipEvent_t prio_fork = nullptr;
  hipStream_t copy_stream = nullptr;              // one of the device's shared copy streams (not owned)
  hipEvent_t copy_done[2] = {nullptr, nullptr};
  hipEvent_t prep_done = nullptr;                 // completion of this context's most recent filter kernel (chained per device)
  hipEvent_t sync_ev = nullptr;                   // what sync() sleeps on (hipEventBlockingSync)
  std::vector<Pending> pending;
  std::vector<hipEvent_t> free_events;
  hipEvent_t open_begin[16] = {};
  struct SmallCopy { uint8_t *dst; const uint8_t *src; size_t bytes; };
  SmallCopy small[8];                              // copy_pinned calls waiting for the next launch on the stream (copy_flush)
  int n_small = 0;
  uint64_t *scan_tmp = nullptr; size_t scan_tmp_n = 0;
  uint32_t *cmp_cnt = nullptr; uint64_t *cmp_off = nullptr; size_t cmp_n = 0;
  Tune tune;
};
static thread_local State *g_cur = nullptr;
static thread_local std::string g_err;
#define g_stream (g_cur->stream)

#define FQ_HIP(call)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      g_err = std::string(#call) + ": " + hipGetErrorString(e_);                             \
      return -3;                                                                             \
    }                                                                                        \
  } while (0)

int copy_flush();
#define FQ_PRE() do { if (copy_flush()) return -3; } while (0)      // the queued small copies go first (copy_pinned)
void copy_discard() { if (g_cur) g_cur->n_small = 0; }
int copy_flush_now() { FQ_PRE(); return 0; }

const char *last_error() { return g_err.c_str(); }
bool is_real_gpu() { return true; }

static int set_kernel_attributes();
static std::mutex g_dev_mu;
static bool g_dev_ready[64];
// the filter kernels of all contexts of a device are chained (launch_prep): the most recent one's completion event, and the
// state that owns it (an event must not be destroyed while another stream may still have to wait on it: state_destroy
// synchronises its stream first and clears the slot)
static hipEvent_t g_prep_last[64];
static State *g_prep_owner[64];
// Input prefetch copies of all contexts of a device go through two shared copy streams, created at the first prefetch: a copy
// stream per context would double the number of streams, and streams beyond the hardware queues (GPU_MAX_HW_QUEUES) share
// queues -- a context's kernels then wait behind another context's long search kernel.
static hipStream_t g_copy_streams[64][2];
static unsigned g_copy_next[64];
// device_turn_begin / device_turn_end: at most `slots` contexts hold a turn on a device at a time (1: one after the other)
static std::mutex g_turn_mu[64];
static std::condition_variable g_turn_cv[64];
static int g_turn_held[64];
// Hardware queues: every context drives a stream of its own, and the HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware
// queues (4 unless the variable says otherwise, read when the runtime initialises).  Contexts that share a queue wait for each
// other's long kernels (16 streams on 16 queues: two share, and straggle by 25 %; 32 or more queues halve the throughput of 16
// streams).  The library changes nothing of the process on its own: a host program that wants more queues, or sleeping waits on
// the devices the library opens, says so with fq_runtime_configure() before its first HIP call (the command line and bench.py do);
// the library says once when more contexts are created on a device than the queues in effect can keep apart.
static int g_ctx_count[64];
static hipStream_t g_prio_stream[64];   // the filter kernels' stream of the highest priority (tuning key prep_priority), one per device
static bool g_queue_warned = false;
static int g_rt_blocking_waits = 0;
int runtime_configure(int hw_queues, int blocking_waits) {
  if (hw_queues > 0) {
    char b[16];
    snprintf(b, sizeof b, "%d", hw_queues);
    setenv("GPU_MAX_HW_QUEUES", b, 0);   // (a value the caller's environment already holds stands)
  }
  g_rt_blocking_waits = blocking_waits ? 1 : 0;
  return 0;
}

int device_count() { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; } return n < 0 ? 0 : n; }
void device_turn_begin(int slots) {
  const int d = g_cur->device;
  std::unique_lock<std::mutex> lk(g_turn_mu[d]);
  g_turn_cv[d].wait(lk, [&] { return g_turn_held[d] < (slots < 1 ? 1 : slots); });
  ++g_turn_held[d];
}
void device_turn_end() {
  const int d = g_cur->device;
  { std::lock_guard<std::mutex> lk(g_turn_mu[d]); --g_turn_held[d]; }
  g_turn_cv[d].notify_all();
}

State *state_create(int dev) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { g_err = "no HIP device visible"; return nullptr; }
  if (dev < 0 || dev >= n || dev >= 64) { g_err = "device ordinal out of range"; return nullptr; }
  if (hipSetDevice(dev) != hipSuccess) { g_err = "hipSetDevice failed"; return nullptr; }
  g_cur = nullptr;   // the thread's current device may have changed: the next bind() sets it again
  {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    if (!g_dev_ready[dev]) {
      // fq_runtime_configure(.., blocking_waits = 1): every wait of the runtime on this device sleeps instead of spinning (the library's own
      // waits sleep on blocking events whatever the flag: sync()); a process that has already fixed the device's flags keeps its choice
      if (g_rt_blocking_waits) { (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync); (void)hipGetLastError(); }
      if (set_kernel_attributes()) return nullptr;
      g_dev_ready[dev] = true;
    }
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    const int queues = q && atoi(q) > 0 ? atoi(q) : 4;
    if (++g_ctx_count[dev] + 2 > queues && !g_queue_warned) {
      g_queue_warned = true;
      fprintf(stderr, "fastquick_amd: %d alignment contexts on device %d with GPU_MAX_HW_QUEUES=%d: contexts will share hardware queues and wait for each other's kernels "
                      "(fq_runtime_configure(contexts + 4, ..) or GPU_MAX_HW_QUEUES before the process makes its first HIP call)\n", g_ctx_count[dev], dev, queues);
    }
  }
  State *s = new State;
  s->device = dev;
  bool ok = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&s->copy_done[0], hipEventDisableTiming | hipEventBlockingSync) == hipSuccess &&
            hipEventCreateWithFlags(&s->copy_done[1], hipEventDisableTiming | hipEventBlockingSync) == hipSuccess &&
            hipEventCreateWithFlags(&s->prep_done, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&s->sync_ev, hipEventDisableTiming | hipEventBlockingSync) == hipSuccess;
  if (!ok) { g_err = "stream / event creation failed"; state_destroy(s); return nullptr; }
  s->main_stream = s->stream;
  return s;
}
void state_destroy(State *s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  g_cur = nullptr;   // (as in state_create; also drops a pointer to the state being freed)
  s->stream = s->main_stream ? s->main_stream : s->stream;
  if (s->aux_stream) { (void)hipStreamSynchronize(s->aux_stream); (void)hipStreamDestroy(s->aux_stream); }
  if (s->sync_ev) (void)hipEventDestroy(s->sync_ev);
  if (s->fork_ev) (void)hipEventDestroy(s->fork_ev);
  if (s->join_ev) (void)hipEventDestroy(s->join_ev);
  for (hipEvent_t m : s->marks) if (m) (void)hipEventDestroy(m);
  if (s->prio_fork) (void)hipEventDestroy(s->prio_fork);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  if (s->copy_stream) (void)hipStreamSynchronize(s->copy_stream);   // (shared: also waits for other contexts' copies enqueued so far)
  {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    if (g_prep_owner[s->device] == s) { g_prep_owner[s->device] = nullptr; g_prep_last[s->device] = nullptr; }
    if (g_ctx_count[s->device] > 0) --g_ctx_count[s->device];
  }
  for (auto &p : s->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto e : s->free_events) (void)hipEventDestroy(e);
  for (auto e : s->copy_done) if (e) (void)hipEventDestroy(e);
  if (s->prep_done) (void)hipEventDestroy(s->prep_done);
  if (s->scan_tmp) (void)hipFree(s->scan_tmp);
  if (s->cmp_cnt) (void)hipFree(s->cmp_cnt);
  if (s->cmp_off) (void)hipFree(s->cmp_off);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}
Tune *tune(State *s) { return &s->tune; }
int bind(State *s) {
  if (!s) { g_err = "no device state"; return -3; }
  if (g_cur != s) { FQ_HIP(hipSetDevice(s->device)); g_cur = s; }
  return 0;
}

void *dmalloc(size_t bytes) {
  void *p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { g_err = "hipMalloc failed (" + std::to_string(bytes) + " B)"; return nullptr; }
  return p;
}
void dfree(void *p) { if (p) (void)hipFree(p); }
// Pinned host memory.  hipHostMalloc pins 4 KiB pages one by one: 0.15-0.4 ms per megabyte, and a context of a deep run wants 0.4 GB of staging in its first call.  Blocks of
// kRegisterMin bytes or more are anonymous mappings (huge pages asked for) made known to the runtime with hipHostRegister instead: four times faster to get, the same copy
// rate, the same address on the device (the staging kernels read and write them in place) -- measured on the box, tools/README.md (pin_bench).  FASTQUICK_PIN_REGISTER=0: hipHostMalloc for all.
static const size_t kRegisterMin = (size_t)4 << 20;
static std::mutex g_reg_mu;
static std::unordered_map<void *, size_t> g_registered;
// hipHostMalloc places its pages on the host node next to the device; a mapping is placed by whoever touches it first.  The mapping is bound (preferred, not strict) to
// the node the device's PCI function reports, so that the copy engines do not cross the sockets' link: -1 when the system does not say.
static int device_numa_node(int dev) {
  static std::mutex mu;
  static std::unordered_map<int, int> cache;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(dev);
  if (it != cache.end()) return it->second;
  int node = -1;
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, dev) == hipSuccess) {
    for (char *c = bus; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
    if (FILE *f = fopen(path.c_str(), "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
  } else (void)hipGetLastError();
  cache[dev] = node;
  return node;
}
static void prefer_node(void *m, size_t len, int node) {
  if (node < 0 || node >= 1024) return;
  unsigned long mask[16] = {0};
  mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
  (void)syscall(SYS_mbind, m, len, 1 /* MPOL_PREFERRED */, mask, (unsigned long)(8 * sizeof mask + 1), 0u);
}
void *hmalloc(size_t bytes) {
  static const bool use_register = [] { const char *e = getenv("FASTQUICK_PIN_REGISTER"); return !(e && *e == '0'); }();
  if (use_register && bytes >= kRegisterMin) {
    const size_t len = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    void *m = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m != MAP_FAILED) {
      (void)madvise(m, len, MADV_HUGEPAGE);
      int dev = 0;
      if (hipGetDevice(&dev) == hipSuccess) prefer_node(m, len, device_numa_node(dev)); else (void)hipGetLastError();
      if (hipHostRegister(m, len, hipHostRegisterDefault) == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        g_registered[m] = len;
        return m;
      }
      (void)hipGetLastError();
      munmap(m, len);      // (a limit on locked memory, a runtime that cannot: the ordinary way below)
    }
  }
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) { g_err = "hipHostMalloc failed"; return nullptr; }
  return p;
}
void hfree(void *p) {
  if (!p) return;
  size_t len = 0;
  {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    auto it = g_registered.find(p);
    if (it != g_registered.end()) { len = it->second; g_registered.erase(it); }
  }
  if (len) { (void)hipHostUnregister(p); munmap(p, len); }
  else (void)hipHostFree(p);
}
int h2d(void *dst, const void *src, size_t bytes) { FQ_PRE(); if (bytes) FQ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, g_stream)); return 0; }
int d2h(void *dst, const void *src, size_t bytes) { FQ_PRE(); if (bytes) FQ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, g_stream)); return 0; }
int d2d(void *dst, const void *src, size_t bytes) { FQ_PRE(); if (bytes) FQ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, g_stream)); return 0; }
int dzero(void *dst, size_t bytes) { FQ_PRE(); if (bytes) FQ_HIP(hipMemsetAsync(dst, 0, bytes, g_stream)); return 0; }
int dfill(void *dst, int byte, size_t bytes) { FQ_PRE(); if (bytes) FQ_HIP(hipMemsetAsync(dst, byte, bytes, g_stream)); return 0; }
// The calling thread SLEEPS until the stream has drained (an event with hipEventBlockingSync): hipStreamSynchronize spins, one core per
// waiting stream -- sixteen WGS streams, or the ranks of a node, then burn the host's cores (and, in a container with a CPU quota, its
// whole allowance) on waiting.  -DFQ_SPIN_SYNC / tuning key spin_sync=1: the spinning wait.
int sync() {
  FQ_PRE();
  if (g_cur->tune.spin_sync || !g_cur->sync_ev) { FQ_HIP(hipStreamSynchronize(g_stream)); return 0; }
  FQ_HIP(hipEventRecord(g_cur->sync_ev, g_stream));
  FQ_HIP(hipEventSynchronize(g_cur->sync_ev));
  return 0;
}
int stream_aux(int on) {
  FQ_PRE();
  if (on && !g_cur->aux_stream) {
    FQ_HIP(hipStreamCreateWithFlags(&g_cur->aux_stream, hipStreamNonBlocking));
    FQ_HIP(hipEventCreateWithFlags(&g_cur->fork_ev, hipEventDisableTiming));
    FQ_HIP(hipEventCreateWithFlags(&g_cur->join_ev, hipEventDisableTiming));
  }
  g_cur->stream = on ? g_cur->aux_stream : g_cur->main_stream;
  return 0;
}
int stream_fork() {
  FQ_PRE();
  if (!g_cur->aux_stream) return 0;
  FQ_HIP(hipEventRecord(g_cur->fork_ev, g_cur->main_stream));
  FQ_HIP(hipStreamWaitEvent(g_cur->aux_stream, g_cur->fork_ev, 0));
  return 0;
}
int stream_join() {
  FQ_PRE();
  if (!g_cur->aux_stream) return 0;
  FQ_HIP(hipEventRecord(g_cur->join_ev, g_cur->aux_stream));
  FQ_HIP(hipStreamWaitEvent(g_cur->main_stream, g_cur->join_ev, 0));
  return 0;
}
// A mark on the current stream (main or aux), and the current stream waiting for a mark: what stream_fork / stream_join do for "everything so
// far", per piece of work -- the front end's inflate launches run one chunk ahead of the kernels that read their text.
int stream_mark(int k) {
  FQ_PRE();
  hipEvent_t &m = g_cur->marks[k & 3];
  if (!m) FQ_HIP(hipEventCreateWithFlags(&m, hipEventDisableTiming));
  FQ_HIP(hipEventRecord(m, g_stream));
  return 0;
}
int stream_wait_mark(int k) {
  FQ_PRE();
  hipEvent_t m = g_cur->marks[k & 3];
  if (m) FQ_HIP(hipStreamWaitEvent(g_stream, m, 0));
  return 0;
}
// input prefetch: copies on the context's copy stream run under the compute stream's kernels; slot = which of the two input
// buffers the copies since the previous copy_record() filled
static int copy_stream_get() {
  if (g_cur->copy_stream) return 0;
  std::lock_guard<std::mutex> lk(g_dev_mu);
  const int dev = g_cur->device;
  for (int k = 0; k < 2; ++k)
    if (!g_copy_streams[dev][k]) FQ_HIP(hipStreamCreateWithFlags(&g_copy_streams[dev][k], hipStreamNonBlocking));
  g_cur->copy_stream = g_copy_streams[dev][g_copy_next[dev]++ & 1];
  return 0;
}
int h2d_copy(void *dst, const void *src, size_t bytes) {
  if (copy_stream_get()) return -3;
  if (bytes) FQ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, g_cur->copy_stream));
  return 0;
}
int copy_record(int slot) { if (copy_stream_get()) return -3; FQ_HIP(hipEventRecord(g_cur->copy_done[slot & 1], g_cur->copy_stream)); return 0; }
int copy_wait(int slot) { FQ_HIP(hipEventSynchronize(g_cur->copy_done[slot & 1])); return 0; }
int compute_wait_copy(int slot) { FQ_PRE(); FQ_HIP(hipStreamWaitEvent(g_stream, g_cur->copy_done[slot & 1], 0)); return 0; }

// Small copies between pinned host memory and device memory by a kernel on the compute stream (the pinned range is mapped into
// the device's address space): no DMA engine, no staging, one kernel launch of latency -- the per-stage lists of a call are a
// few KB to a few hundred KB, and under load a hipMemcpyAsync + synchronize round trip for them cost milliseconds.
// The copies queued since the last launch go out as ONE kernel (blockIdx.y = which copy): a call makes a few dozen of them, and a
// launch apiece was a dispatch slot each -- under sixteen streams they queued behind the other streams' long kernels.
struct FqCopySet { uint8_t *dst[8]; const uint8_t *src[8]; size_t bytes[8]; };
__global__ void __launch_bounds__(256) k_copy_bytes(FqCopySet cs) {
  uint8_t *dst = cs.dst[blockIdx.y];
  const uint8_t *src = cs.src[blockIdx.y];
  const size_t bytes = cs.bytes[blockIdx.y];
  const size_t n16 = ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) ? bytes / 16 : 0;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = t; i < n16; i += stride) ((uint4 *)dst)[i] = ((const uint4 *)src)[i];
  for (size_t i = n16 * 16 + t; i < bytes; i += stride) dst[i] = src[i];
}
static const size_t kSmallCopy = (size_t)8 << 20;
// every entry point that puts work on the stream calls this first: the queued copies keep their place in stream order
int copy_flush() {
  State *s = g_cur;
  if (!s || s->n_small == 0) return 0;
  FqCopySet cs;
  size_t most = 0;
  for (int k = 0; k < 8; ++k) {
    const bool on = k < s->n_small;
    cs.dst[k] = on ? s->small[k].dst : nullptr; cs.src[k] = on ? s->small[k].src : nullptr; cs.bytes[k] = on ? s->small[k].bytes : 0;
    if (on) most = std::max(most, s->small[k].bytes);
  }
  const int n = s->n_small;
  s->n_small = 0;
  const unsigned blocks = (unsigned)std::min<size_t>(256, (most / 16 + 255) / 256 + 1);
  hipLaunchKernelGGL(k_copy_bytes, dim3(blocks, (unsigned)n), dim3(256), 0, s->stream, cs);
  FQ_HIP(hipGetLastError());
  return 0;
}
int copy_pinned(void *dst, const void *src, size_t bytes, int to_device) {
  if (!bytes) return 0;
  if (bytes > kSmallCopy) { FQ_PRE(); FQ_HIP(hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, g_stream)); return 0; }
  if (g_cur->n_small == 8) FQ_PRE();
  g_cur->small[g_cur->n_small++] = {(uint8_t *)dst, (const uint8_t *)src, bytes};
  return 0;
}

// ---- timing ---------------------------------------------------------------------------------
static hipEvent_t get_event() {
  if (!g_cur->free_events.empty()) { hipEvent_t e = g_cur->free_events.back(); g_cur->free_events.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
void time_begin(int kid) { (void)copy_flush(); hipEvent_t e = get_event(); (void)hipEventRecord(e, g_stream); g_cur->open_begin[kid] = e; }
void time_end(int kid) { (void)copy_flush(); hipEvent_t e = get_event(); (void)hipEventRecord(e, g_stream); g_cur->pending.push_back({kid, g_cur->open_begin[kid], e}); }
// start/stop events attached to one kernel (hipExtLaunchKernelGGL): the kernel's own begin/end timestamps, i.e. what
// rocprofv3 --kernel-trace reports, unaffected by dispatch queueing when several streams share the GPU
static void kernel_events(int kid, hipEvent_t *a, hipEvent_t *b) { *a = get_event(); *b = get_event(); g_cur->pending.push_back({kid, *a, *b}); }
void time_collect(double ms[], uint64_t launches[], int n_ids) {
  std::vector<decltype(g_cur->pending)::value_type> later;
  for (auto &p : g_cur->pending) {
    if (hipEventQuery(p.b) == hipErrorNotReady) { later.push_back(p); continue; }     // (work on the other stream that is still running: next time)
    float t = 0.f;
    if (hipEventElapsedTime(&t, p.a, p.b) == hipSuccess && p.kid < n_ids) { ms[p.kid] += t; launches[p.kid] += 1; }
    g_cur->free_events.push_back(p.a);
    g_cur->free_events.push_back(p.b);
  }
  g_cur->pending.swap(later);
}

// ---- kernels --------------------------------------------------------------------------------
static inline unsigned nblk(uint64_t n, unsigned b) { return (unsigned)((n + b - 1) / b); }

__global__ void __launch_bounds__(256) k_prep(FqPrepArgs a) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < a.n_reads) fq_prep_thread(a, r);
}
extern __shared__ __align__(16) unsigned char fq_dyn_lds[];
// (LDS per block: 2 x seed_len x 256 bytes, 16 KB at the default seed length.  94 VGPRs = 5 wavefronts per SIMD; asking the compiler for 6 / 8
//  (13 / 30 registers spilled) gave 45.6 / 54.3 ms against 40.7 per on-target call of 8.4 M reads: more wavefronts do not help it)
__global__ void __launch_bounds__(256) k_width(FqWidthArgs a) {
  uint8_t *seed_bits = (uint8_t *)fq_dyn_lds;   // [strand][ii][thread]: lane-interleaved, conflict free
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < a.n_work) fq_width_read(a, w, seed_bits + threadIdx.x, 256);
}
// One thread per (read, strand); the workgroups of XCDs 0..3 take strand 0, those of XCDs 4..7 strand 1 (workgroups are dealt
// round-robin over the 8 XCDs: blockIdx % 8 names the XCD), so that an XCD's 4 MB L2 holds one strand's Occ table, not two halves.
__global__ void __launch_bounds__(256) k_width_strand(FqWidthArgs a) {
  uint8_t *seed_bits = (uint8_t *)fq_dyn_lds;   // [ii][thread]
  const int xcd = blockIdx.x & 7, strand = xcd >> 2;
  const int q = (int)(blockIdx.x >> 3) * 4 + (xcd & 3);      // this workgroup's number among its strand's
  const int w = q * 256 + threadIdx.x;
  if (w < a.n_work) fq_width_strand(a, w, strand, seed_bits + threadIdx.x, 256);
}
struct FqQueueFetch {
  uint32_t *cursor;
  int n_work;
  __device__ uint32_t operator()(uint32_t n) const { return atomicAdd(cursor, n); }
};
// The lane kernels' queue is two blocks (fq_order_key): a wavefront draws from the block of its XCD's half first -- workgroups are
// dealt round-robin over the 8 XCDs, so blockIdx % 8 tells which workgroups share an XCD (MI355X_MICROARCH.md, workgroup
// dispatch) -- and from the other block when its own is exhausted.
// -DFQ_GAP_WPE=n: ask for n wavefronts per SIMD in the lane search kernels (register budget 512 / n)
#if defined(FQ_GAP_WPE)
#define FQ_GAP_OCC __attribute__((amdgpu_waves_per_eu(FQ_GAP_WPE, FQ_GAP_WPE)))
#else
#define FQ_GAP_OCC
#endif
// -DFQ_NOGAP_WPE=n: the same for the round without gap children only (its lane state is smaller)
#if defined(FQ_NOGAP_WPE)
#define FQ_NOGAP_OCC __attribute__((amdgpu_waves_per_eu(FQ_NOGAP_WPE, FQ_NOGAP_WPE)))
#define FQ_NOGAP_WAVES_PER_CU (4 * FQ_NOGAP_WPE)
#else
#define FQ_NOGAP_OCC FQ_GAP_OCC
#define FQ_NOGAP_WAVES_PER_CU 16
#endif
struct FqQueueFetch2 {
  uint32_t *cursor;
  const uint32_t *split;
  uint32_t n_work;
  int pref, seg, n_seg;
  __device__ uint64_t operator()(uint32_t n) const {
    const uint32_t sp = split ? *split : n_work;
    const uint32_t blk_start[2] = {0u, sp}, blk_len[2] = {sp, n_work - sp};
    for (int t = 0; t < 2; ++t) {
      const int b = t == 0 ? pref : 1 - pref;
      uint32_t lo, hi;
      fq_seg_range(blk_len[b], seg, n_seg, &lo, &hi);      // this launch's share of the block
      const uint32_t len = hi - lo;
      if (len == 0) continue;
      if (__hip_atomic_load(&cursor[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= len) continue;
      const uint32_t base = atomicAdd(&cursor[b], n);
      if (base < len) return (uint64_t)(blk_start[b] + lo + base) | (uint64_t)(blk_start[b] + hi) << 32;
    }
    return 0;
  }
};
#define FQ_LANE_FETCH(a) FqQueueFetch2{(a).queue, (a).split, (uint32_t)(a).n_work, (int)((blockIdx.x & 7u) >> 2), (a).seg, (a).n_seg}
// persistent wavefronts: every lane pulls reads from the queue until it is empty.  One wavefront per block.
// LDS per lane: n_buckets 16-bit bucket heads, lane-interleaved.
__global__ void __launch_bounds__(64) FQ_GAP_OCC k_gap_persist_lds(FqGapArgs a) {
  uint16_t *heads = (uint16_t *)fq_dyn_lds;
  FqGapStoreLds st = {heads + threadIdx.x, 64, (uint32_t *)(heads + 64 * a.o.n_buckets) + threadIdx.x};
  fq_gap_lanes<false>(a, st, FQ_LANE_FETCH(a), (int)(blockIdx.x * 64 + threadIdx.x));
}
// first round of a device-filling launch: the search without its gap children (FqGapLane, NOGAP)
__global__ void __launch_bounds__(64) FQ_NOGAP_OCC k_gap_nogap_lds(FqGapArgs a) {
  uint16_t *heads = (uint16_t *)fq_dyn_lds;
  FqGapStoreLds st = {heads + threadIdx.x, 64, (uint32_t *)(heads + 64 * a.o.n_buckets) + threadIdx.x};
  fq_gap_lanes<true>(a, st, FQ_LANE_FETCH(a), (int)(blockIdx.x * 64 + threadIdx.x));
}
// the same two kernels for FASTQuick's own option block (FqOptsStock): options as constants
__global__ void __launch_bounds__(64) FQ_GAP_OCC k_gap_persist_stock(FqGapArgs a) {
  uint16_t *heads = (uint16_t *)fq_dyn_lds;
  FqGapStoreLds st = {heads + threadIdx.x, 64, (uint32_t *)(heads + 64 * a.o.n_buckets) + threadIdx.x};
  fq_gap_lanes<false, FqOptsStock>(a, st, FQ_LANE_FETCH(a), (int)(blockIdx.x * 64 + threadIdx.x));
}
__global__ void __launch_bounds__(64) FQ_NOGAP_OCC k_gap_nogap_stock(FqGapArgs a) {
  uint16_t *heads = (uint16_t *)fq_dyn_lds;
  FqGapStoreLds st = {heads + threadIdx.x, 64, (uint32_t *)(heads + 64 * a.o.n_buckets) + threadIdx.x};
  fq_gap_lanes<true, FqOptsStock>(a, st, FQ_LANE_FETCH(a), (int)(blockIdx.x * 64 + threadIdx.x));
}
__global__ void __launch_bounds__(64) k_gap_persist(FqGapArgs a) {   // any pool size: bucket heads in HBM
  FqGapStoreGlobal st = {nullptr, {}};
  fq_gap_lanes<false>(a, st, FQ_LANE_FETCH(a), (int)(blockIdx.x * 64 + threadIdx.x));
}
// one read per wavefront (long searches): run table of the read's score buckets in LDS
__global__ void __launch_bounds__(64) k_gap_coop(FqGapArgs a) {
  __shared__ uint32_t heads[2 * FQ_MAX_BUCKETS];
  fq_gap_coop_wave(a, heads, FqQueueFetch{a.queue, a.n_work}, (int)blockIdx.x);
}
__global__ void __launch_bounds__(256) k_sa(FqSaArgs a) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < a.n_rows) fq_sa_thread(a, q);
}
template <bool PACKED>
__device__ int fq_global_align_wave(const uint8_t *s1, int len1, const uint8_t *s2, int len2, int band, int gap_end,
                                    int *RM, int *RI, int *RD, uint8_t *trace, uint8_t *ops, int *n_ops, int *fi, int *fj);
// ---- mate-rescue Smith-Waterman: one 64-lane wavefront per task ---------------------------------------------
// Forward pass of aln_local_core as an anti-diagonal wavefront: lane l owns query row j0+l+1 of a 64-row
// stripe and walks the reference columns one step behind lane l-1; H/E of the row above arrive by a one-lane
// shuffle, the diagonal value and the row's (last_h, f) state live in registers, so the 150 x ~480 cell matrix
// costs ~3*(480+64) steps with no memory traffic except the stripe boundary kept in LDS.  The cell rule is the
// reference's (fq_sw_cell), the end cell is the first strict maximum in row-major order exactly as the
// sequential scan finds it.  Lane 0 then runs the (inherently serial, data-dependent band) reverse pass and the
// banded global fill out of LDS.
//
// aln_local_core's reverse pass (stdaln.c:626-679; fq_sw_reverse is the serial statement) by the whole wavefront, a row at a time.
// The band of a row depends on the rows before it (it follows the running maximum), so rows stay sequential; inside a row the only
// chain is the horizontal gap, f_g = max(f_{g-1} - r, h_{g-1} - q - r) applied when h_{g-1} > 0.  Every h is >= 0, so a non-positive f
// never changes a cell, and a positive one always stems from the current run of positive cells: the cells of a row are
// h_g = max(a_g, max_{k<g}(a_k - q - (g - k) r)) with a_g the cell without its horizontal gap -- a prefix maximum of a_k - q + k r over
// the lanes (carried from one group of 64 cells to the next).  The running maximum, "the first cell that reaches score_f" (where the
// pass stops) and the start cell come from a second prefix maximum in the order the serial loop visits the cells.  The arrays end up as
// the serial loop leaves them: H[x] = h(x) for the row's columns, H[start + 1] = 0, E[x + 1] = e(x), E[end + 1] = 0 -- a cell's inputs
// are all read before anything of its row is written (the next group's are fetched before this group's are stored).
__device__ __forceinline__ int fq_wave_incl_max(int x, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d, 64); if (lane >= d) x = x > y ? x : y; }
  return x;
}
__device__ void fq_sw_reverse_wave(const uint8_t *ref, const uint8_t *qry, int score_f, int end_i, int end_j, int *H, int *E, int *start_i_out, int *start_j_out, int *score_r_out) {
  const int lane = threadIdx.x;
  const int q = FQ_GAP_O, rr = FQ_GAP_E, qr = q + rr;
  const int kNeg = -(1 << 29);
  for (int i = lane; i <= end_i; i += 64) { H[i] = 0; E[i] = 0; }
  __syncthreads();
  int score_r = fq_sm_maq(ref[end_i - 1], qry[end_j - 1]);
  int start_i = end_i, start_j = end_j;
  if (lane == 0) { H[end_i] = qr + score_r; E[end_i] = 0; }
  __syncthreads();
  int start = end_i - 1, end = end_i - 3;
  if (end <= 0) end = 0;
  bool stop = false;
  for (int j = end_j - 1; j != 0; --j) {
    const int c2 = qry[j - 1];
    const int n = start - end;                       // the row's cells: columns start, start - 1, ..., end + 1 (traversal index g = 0 ..)
    int carry_b = kNeg, carry_m = score_r;
    int d_in = 0, s_in = 0, e_in = 0;
    if (lane < n) { const int i = start - lane; d_in = H[i + 1]; s_in = H[i]; e_in = E[i + 1]; }
    for (int g0 = 0; g0 < n; g0 += 64) {
      const int g = g0 + lane, i = start - g;
      const bool act = g < n;
      int d_nx = 0, s_nx = 0, e_nx = 0;
      if (g + 64 < n) { const int i2 = i - 64; d_nx = H[i2 + 1]; s_nx = H[i2]; e_nx = E[i2 + 1]; }
      int e = e_in - rr > s_in - qr ? e_in - rr : s_in - qr;
      if (e < 0) e = 0;
      int a0 = act ? d_in + fq_sm_maq(ref[i - 1], c2) : 0;
      if (a0 < 0) a0 = 0;
      if (a0 < e) a0 = e;
      const int b = act ? a0 - q + g * rr : kNeg;
      const int bi = fq_wave_incl_max(b, lane);
      int bx = __shfl_up(bi, 1, 64);
      if (lane == 0) bx = kNeg;
      if (bx < carry_b) bx = carry_b;
      const int fpos = bx - g * rr;
      const int h = a0 > fpos ? a0 : fpos;
      const int b_last = __shfl(bi, 63, 64);
      if (carry_b < b_last) carry_b = b_last;
      // the running maximum in visiting order
      const int mi = fq_wave_incl_max(act ? h : kNeg, lane);
      int mx = __shfl_up(mi, 1, 64);
      if (lane == 0) mx = kNeg;
      if (mx < carry_m) mx = carry_m;
      const bool improve = act && h > mx;
      const unsigned long long stops = __ballot(improve && h - qr == score_f);
      if (stops) {                                   // the pass ends at the first such cell; nothing of the arrays is read again
        const int ls = __ffsll((long long)stops) - 1;
        score_r = __shfl(h, ls, 64); start_i = start - (g0 + ls); start_j = j;
        stop = true;
        break;
      }
      const int m_last = __shfl(mi, 63, 64);
      if (m_last > carry_m) {
        const unsigned long long at = __ballot(act && h == m_last);
        start_i = start - (g0 + (__ffsll((long long)at) - 1)); start_j = j;
        carry_m = m_last;
      }
      // lane 0's fetch for the next group (H[start - g0 - 63], above) is the cell lane 63 stores below: the fetches of all lanes are
      // complete before any lane stores (shuffles and ballots order nothing in memory; the block is one wavefront, so the barrier is a wait)
      __syncthreads();
      if (act) { H[i] = h; E[i + 1] = e; }
      d_in = d_nx; s_in = s_nx; e_in = e_nx;
    }
    if (stop) break;
    score_r = carry_m;
    if (lane == 0) { H[start + 1] = 0; E[end + 1] = 0; }
    __syncthreads();
    if (H[start] <= qr) --start;
    if (start <= 0) start = 0;
    end = start_i - (start_j - j) - (score_r + (start_j - j) * 11) / rr - 1;
    if (end <= 0) end = 0;
    __syncthreads();
  }
  *start_i_out = start_i; *start_j_out = start_j; *score_r_out = score_r - qr;
}
__global__ void __launch_bounds__(64) k_sw_wave(FqSwArgs a) {
  const int t = blockIdx.x, lane = threadIdx.x;
  const FqSwTask T = a.task[t];
  const int RL = a.RL, QL = a.QL;
  // LDS carve: Hb, Eb (RL+2 ints each), rows M/I/D (RL+1 ints each), ref (RL bytes), qry (QL bytes)
  int *Hb = (int *)fq_dyn_lds;
  int *Eb = Hb + (RL + 2);
  int *rM = Eb + (RL + 2), *rI = rM + (RL + 1), *rD = rI + (RL + 1);
  uint8_t *ref = (uint8_t *)(rD + (RL + 1));
  uint8_t *qry = ref + ((RL + 16) & ~15);
  __shared__ int s_ok, s_len1;
  FqSwOut O;
  O.beg = T.beg; O.cnt = 0; O.n_cigar = 0;
  const int len = a.len_trim[T.read];
  const uint8_t *row = a.seq + (size_t)T.read * (size_t)a.stride;
  const int64_t l_pac = a.ix.l_pac;
  bool ok = !(T.reglen < 20 || l_pac - T.beg < len);
  int len1 = 0;
  if (ok) {
    int nn = 0;
    for (int k = lane; k < len; k += 64) {
      const int c = T.use_rc ? fq_comp(fq_nt4(row[len - 1 - k])) : fq_nt4(row[k]);
      qry[k] = (uint8_t)c;
      nn += c >= 4;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) nn += __shfl_xor(nn, d, 64);
    if ((float)nn / len >= 0.25f || len - nn < 20) ok = false;
    const int64_t avail = l_pac - T.beg;
    len1 = (int)(avail < (int64_t)T.reglen ? avail : (int64_t)T.reglen);
    if (len1 < 0) len1 = 0;
    for (int k = lane; k < len1; k += 64) ref[k] = (uint8_t)fq_pac_base(a.ix.pac, T.beg + k);
    for (int k = lane; k <= len1 + 1; k += 64) { Hb[k] = 0; Eb[k] = 0; }
  }
  __syncthreads();
  if (!ok) { if (lane == 0) a.out[t] = O; return; }
  const int len2 = len;
  int best_h = 0, best_i = 0, best_j = 0;
  for (int j0 = 0; j0 < len2; j0 += 64) {
    const int j = j0 + lane + 1;
    const bool active = j <= len2;
    const int c2 = active ? qry[j - 1] : 4;
    const int nrows = len2 - j0 < 64 ? len2 - j0 : 64;
    int last_h = 0, f = 0, diag = 0, out_h = 0, out_e = 0;
    const int steps = len1 + nrows - 1;
    for (int st = 1; st <= steps; ++st) {
      int up_h = __shfl_up(out_h, 1, 64), up_e = __shfl_up(out_e, 1, 64);
      const int i = st - lane;
      const bool in = active && i >= 1 && i <= len1;
      if (lane == 0 && in) { up_h = Hb[i]; up_e = Eb[i]; }
      if (in) {
        int e_out;
        const int h = fq_sw_cell(diag, up_h, up_e, fq_sm_maq(ref[i - 1], c2), last_h, f, e_out);
        diag = up_h;
        out_h = h; out_e = e_out;
        if (best_h < h) { best_h = h; best_i = i; best_j = j; }
        if (lane == nrows - 1) { Hb[i] = h; Eb[i] = e_out; }   // boundary for the next stripe (lane 0 read column i >= nrows-1 steps ago)
      }
    }
    __syncthreads();
  }
  // first strict maximum in row-major order == (max h, then min j); i is that row's first maximum
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const int oh = __shfl_xor(best_h, d, 64), oi = __shfl_xor(best_i, d, 64), oj = __shfl_xor(best_j, d, 64);
    if (oh > best_h || (oh == best_h && oh > 0 && oj < best_j)) { best_h = oh; best_i = oi; best_j = oj; }
  }
  // reverse pass on lane 0 (its band follows the running maximum), banded global fill of the sub-rectangle by the whole
  // wavefront (trace matrix in LDS when the launcher found room for it, else in the task's global scratch), the rest on lane 0
  if (best_h < 1) { if (lane == 0) a.out[t] = O; return; }
  FqDpScratch S = fq_dp_carve(a.scratch + (size_t)t * a.scratch_stride, RL, QL);
  uint8_t *ops = qry + ((QL + 16) & ~15);
  uint8_t *trace = a.trace_in_lds ? ops + ((RL + QL + 16) & ~15) : S.trace;
  int start_i = 0, start_j = 0, score_r = 0;
  if (a.serial_reverse) {                            // (test knob sw_serial_reverse: the serial statement on lane 0)
    if (lane == 0) fq_sw_reverse(ref, qry, best_h, best_i, best_j, Hb, Eb, &start_i, &start_j, &score_r);
    start_i = __shfl(start_i, 0); start_j = __shfl(start_j, 0); score_r = __shfl(score_r, 0);
  } else fq_sw_reverse_wave(ref, qry, best_h, best_i, best_j, Hb, Eb, &start_i, &start_j, &score_r);
  __syncthreads();
  int n_ops = 0, fi = 0, fj = 0, score_g;
  const int jmax = (best_i - start_i > best_j - start_j ? best_i - start_i : best_j - start_j) + 1;
  for (int b = FQ_BAND;; b <<= 1) {   // doubling band (stdaln.c:705-716)
    score_g = fq_global_align_wave<false>(ref + start_i - 1, best_i - start_i + 1, qry + start_j - 1, best_j - start_j + 1, b, -1, rM, rI, rD, trace, ops, &n_ops, &fi, &fj);
    if (score_g == score_r || best_h == score_g) break;
    if (b > jmax) break;
  }
  if (lane == 0) {
    if (!(score_r > score_g && best_h > score_g))   // else: the "Potential bug" arm of the reference, ret < 0
      fq_sw_post(T, ref, qry, len, best_j, start_i, start_j, fi, fj, ops, n_ops, a.cigar + (size_t)t * (size_t)a.cig_cap, a.cig_cap, O);
    a.out[t] = O;
  }
}

// ---- gapped refinement: one lane per task, the DP row lives in lane-interleaved LDS ---------------------------------
__global__ void __launch_bounds__(64) k_refine_lds(FqRefineArgs a) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  int *base = (int *)fq_dyn_lds;
  const int W = a.RL + 1;
  FqRowsPlanar R = {base + threadIdx.x, base + (size_t)W * 64 + threadIdx.x, base + (size_t)2 * W * 64 + threadIdx.x, 64};
  if (t < a.n_task) fq_refine_task(a, t, R);
}
// ---- banded global alignment (aln_global_core, stdaln.c:345-525) by one wavefront -----------------------------------------
// Same cells, same rules and the same trace bytes as fq_global_align (fq_kernels.h), filled along anti-diagonals: lane l owns
// row j0+l of a 64-row stripe and is one column behind lane l-1, whose cell of the column above arrives by a one-lane
// shuffle; the last row of a stripe is kept in LDS (RM/RI/RD) for the first lane of the next one -- the array the serial code
// updates in place.  A row's first position (column 0 in phases 1/5, column j-b2 otherwise) is the serial code's "left"
// initialisation.  Values of the row above outside its band are never consumed (the i == hi rules of each phase).  The
// traceback is serial (lane 0, out of LDS).  Block = one wavefront.
// PACKED: the trace matrix holds two cells per byte (a cell's trace is four bits; a row's cells are written by one lane, left to
// right, so the lane keeps the low half of a byte until its neighbour arrives): the refine kernel's LDS drops from 27 to 15 KB per
// wavefront, twice the wavefronts per CU -- the kernel is bound by the issue of dependent instructions, which more wavefronts per SIMD hide.
template <bool PACKED>
__device__ __forceinline__ uint8_t fq_trace_get(const uint8_t *trace, int W, int j, int i) {
  if (!PACKED) return trace[(size_t)j * W + i];
  return (uint8_t)((trace[(size_t)j * ((W + 1) >> 1) + (i >> 1)] >> (4 * (i & 1))) & 15);
}
template <bool PACKED>
__device__ int fq_global_align_wave(const uint8_t *s1, int len1, const uint8_t *s2, int len2, int band, int gap_end,
                                    int *RM, int *RI, int *RD, uint8_t *trace, uint8_t *ops, int *n_ops, int *fi, int *fj) {
  const int lane = threadIdx.x;
  if (len1 == 0 || len2 == 0) { *n_ops = 0; return 0; }
  int b1, b2;
  if (len1 > len2) { b1 = len1 - len2 + band; b2 = band; } else { b1 = band; b2 = len2 - len1 + band; }
  if (b1 > len1) b1 = len1;
  if (b2 > len2) b2 = len2;
  const int W = len1 + 1;
  const int end_ext = gap_end >= 0 ? gap_end : FQ_GAP_E;
  if (lane == 0) {   // row 0
    uint8_t fm;
    int lM = 0, lD = FQ_NEG_INF;
    RM[0] = 0; RI[0] = RD[0] = FQ_NEG_INF;
    for (int i = 1; i < b1; ++i) {
      const int d = fq_pick_gap(lM, lD, end_ext, fm);
      if (PACKED) { uint8_t *b = trace + (i >> 1); *b = (uint8_t)((i & 1) ? ((*b & 15) | ((fm ? 0 : 8) << 4)) : (fm ? 0 : 8)); }
      else trace[i] = (uint8_t)(fm ? 0 : 8);
      RM[i] = RI[i] = FQ_NEG_INF; RD[i] = d;
      lM = FQ_NEG_INF; lD = d;
    }
  }
  __syncthreads();
  const int p1_end = b2 < len2 ? b2 : len2 - 1;
  for (int j0 = 1; j0 <= len2; j0 += 64) {
    const int j = j0 + lane;
    const bool row_on = j <= len2;
    int phase;
    if (j <= p1_end) phase = 1;
    else if (j == p1_end + 1 && j == len2 && b2 != len2 - 1) phase = 5;
    else if (j <= len2 - b2 + 1) phase = 2;
    else if (j < len2) phase = 3;
    else phase = 4;
    const bool p15 = phase == 1 || phase == 5, endD = phase == 5 || phase == 4;
    const int c0 = p15 ? 0 : j - b2;
    const int hi = !row_on ? -1 : p15 ? ((j + b1 <= len1 + 1) ? j + b1 - 1 : len1) : (phase == 2 ? j + b1 - 1 : len1);
    const int c2 = row_on ? s2[j - 1] : 0;
    const bool keeps_row = row_on && (lane == 63 || j == len2);
    uint8_t *tr = trace + (size_t)j * (size_t)(PACKED ? (W + 1) >> 1 : W);
    uint32_t half = 0;      // PACKED: the trace of this row's last even column
    const int n_rows = len2 - j0 + 1 < 64 ? len2 - j0 + 1 : 64;
    const int t0 = __shfl(c0, 0), t1 = len1 + n_rows - 1;
    int oM = 0, oI = 0, oD = 0;
    FqCell diag, left;
    diag.M = diag.I = diag.D = 0; left = diag;
    for (int t = t0; t <= t1; ++t) {
      const int i = t - lane;
      FqCell up;
      up.M = __shfl_up(oM, 1); up.I = __shfl_up(oI, 1); up.D = __shfl_up(oD, 1);
      if (lane == 0) { const int ii = i < 0 ? 0 : (i > len1 ? len1 : i); up.M = RM[ii]; up.I = RI[ii]; up.D = RD[ii]; }
      if (row_on && i >= c0 && i <= hi) {
        FqCell c;
        uint8_t fm;
        if (i == c0) {
          c.M = c.I = c.D = FQ_NEG_INF;
          uint8_t tb0 = 0;
          if (p15) { c.I = fq_pick_gap(up.M, up.I, end_ext, fm); tb0 = (uint8_t)(fm ? 0 : 4); }
          if (!PACKED) { if (p15) tr[0] = tb0; }
          else if (i & 1) tr[i >> 1] = (uint8_t)(tb0 << 4);      // (a row that starts at an odd column: the low half belongs to no cell of it)
          else { half = tb0; tr[i >> 1] = tb0; }
        } else {
          uint8_t tM, tb;
          c.M = fq_pick_M(diag, fq_sm_maq(s1[i - 1], c2), tM);
          tb = tM;
          if (i != hi) { c.I = fq_pick_gap(up.M, up.I, FQ_GAP_E, fm); tb |= (uint8_t)(fm ? 0 : 4); }
          else if (p15) {
            if (j + b1 - 1 > len1) { c.I = fq_pick_gap(up.M, up.I, end_ext, fm); tb |= (uint8_t)(fm ? 0 : 4); }
            else c.I = FQ_NEG_INF;
          } else if (phase == 2) c.I = FQ_NEG_INF;
          else { c.I = fq_pick_gap(up.M, up.I, end_ext, fm); tb |= (uint8_t)(fm ? 0 : 4); }
          c.D = fq_pick_gap(left.M, left.D, endD ? end_ext : FQ_GAP_E, fm);
          tb |= (uint8_t)(fm ? 0 : 8);
          if (!PACKED) tr[i] = tb;
          else if (i & 1) tr[i >> 1] = (uint8_t)(half | (uint32_t)tb << 4);
          else { half = tb; tr[i >> 1] = tb; }
        }
        diag = up; left = c;
        oM = c.M; oI = c.I; oD = c.D;
        if (keeps_row) { RM[i] = c.M; RI[i] = c.I; RD[i] = c.D; }
      }
    }
    __syncthreads();
  }
  // traceback (stdaln.c:484-512), lane 0; every lane returns its results through LDS-resident ops / the broadcast below
  int mx = 0, n = 0, li = 0, lj = 0;
  if (lane == 0) {
    int i = len1, j = len2;
    mx = RM[len1];
    uint8_t tb = fq_trace_get<PACKED>(trace, W, j, i);
    int type = tb & 3, ctype = FQ_OP_M;
    if (RI[len1] > mx) { mx = RI[len1]; type = (tb & 4) ? FQ_OP_I : FQ_OP_M; ctype = FQ_OP_I; }
    if (RD[len1] > mx) { mx = RD[len1]; type = (tb & 8) ? FQ_OP_D : FQ_OP_M; ctype = FQ_OP_D; }
    li = i; lj = j;
    ops[n++] = (uint8_t)ctype;
    do {
      if (ctype == FQ_OP_M) { --i; --j; } else if (ctype == FQ_OP_I) --j; else --i;
      ctype = type;
      tb = fq_trace_get<PACKED>(trace, W, j, i);
      type = type == FQ_OP_M ? (tb & 3) : type == FQ_OP_I ? ((tb & 4) ? FQ_OP_I : FQ_OP_M) : ((tb & 8) ? FQ_OP_D : FQ_OP_M);
      if (i || j) { ops[n++] = (uint8_t)ctype; li = i; lj = j; }
    } while (i || j);
  }
  __syncthreads();
  *n_ops = __shfl(n, 0); *fi = __shfl(li, 0); *fj = __shfl(lj, 0);
  return __shfl(mx, 0);
}
// refine_gapped_core (libbwa/bwase.c:183-232), one task per wavefront; LDS: RM/RI/RD, ref, qry, ops, trace
__global__ void __launch_bounds__(64) k_refine_wave(FqRefineArgs a) {
  const int t = blockIdx.x, lane = threadIdx.x;
  const int RL = a.RL, QL = a.QL;
  int *RM = (int *)fq_dyn_lds, *RI = RM + (RL + 1), *RD = RI + (RL + 1);
  uint8_t *ref = (uint8_t *)(RD + (RL + 1));
  uint8_t *qry = ref + ((RL + 16) & ~15);
  uint8_t *ops = qry + ((QL + 16) & ~15);
  uint8_t *trace = ops + ((RL + QL + 16) & ~15);
  const FqRefTask T = a.task[t];
  const int len = a.len_trim[T.read];
  const uint8_t *row = a.seq + (size_t)T.read * (size_t)a.stride;
  const int64_t l_pac = a.ix.l_pac;
  for (int k = lane; k < len; k += 64) qry[k] = (uint8_t)(T.strand ? fq_comp(fq_nt4(row[len - 1 - k])) : fq_nt4(row[k]));
  int64_t pos = (int64_t)T.pos > l_pac ? (int64_t)(int32_t)T.pos : (int64_t)T.pos;
  const int aext = T.ext < 0 ? -T.ext : T.ext, ref_len = len + aext;
  int64_t k0, k1;
  if (T.ext > 0) { k0 = pos; k1 = pos + ref_len < l_pac ? pos + ref_len : l_pac; }
  else { const int64_t x = pos + len; k0 = x - ref_len > 0 ? x - ref_len : 0; k1 = x < l_pac ? x : l_pac; }
  const int l = k1 > k0 ? (int)(k1 - k0) : 0;
  for (int k = lane; k < l; k += 64) ref[k] = (uint8_t)fq_pac_base(a.ix.pac, k0 + k);
  __syncthreads();
  int n_ops = 0, fi, fj;
  fq_global_align_wave<true>(ref, l, qry, len, FQ_BAND, FQ_GAP_END, RM, RI, RD, trace, ops, &n_ops, &fi, &fj);
  if (lane != 0) return;
  uint16_t *cg = a.cigar + (size_t)t * (size_t)a.cig_cap;
  int n = fq_ops_to_cigar(ops, n_ops, cg, a.cig_cap);
  FqRefOut O;
  if (n <= 0) { O.pos = T.pos; O.n_cigar = 0; a.out[t] = O; return; }
  if (T.ext < 0) {
    int d = 0;
    for (int k = 0; k < n; ++k) { const int op = cg[k] >> 14, ln = cg[k] & 0x3fff; if (op == FQ_OP_D) d -= ln; else if (op == FQ_OP_I) d += ln; }
    pos += d;
  }
  if ((cg[0] >> 14) == FQ_OP_D) { pos += cg[0] & 0x3fff; for (int k = 0; k < n - 1; ++k) cg[k] = cg[k + 1]; --n; }
  if ((cg[n - 1] >> 14) == FQ_OP_D) --n;
  if ((cg[n - 1] >> 14) == FQ_OP_I) cg[n - 1] = (uint16_t)(FQ_OP_S << 14 | (cg[n - 1] & 0x3fff));
  if ((cg[0] >> 14) == FQ_OP_I) cg[0] = (uint16_t)(FQ_OP_S << 14 | (cg[0] & 0x3fff));
  O.pos = (uint32_t)pos; O.n_cigar = n;
  a.out[t] = O;
}
__global__ void __launch_bounds__(64) k_refine(FqRefineArgs a) {   // long reads: DP row in global memory
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < a.n_task) fq_refine_thread(a, t);
}
__global__ void __launch_bounds__(256) k_pack_aln(const FqAln *aln, const uint32_t *n_aln, const uint64_t *off, uint32_t cap, uint32_t n_work, FqAln *packed) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_work) return;
  const uint32_t n = n_aln[w];
  const uint64_t o = off[w];
  for (uint32_t j = 0; j < n; ++j) packed[o + j] = aln[(size_t)w * cap + j];
}

// ---- block-level exclusive scan built from 64-lane wavefront scans ------------------------------
// returns the exclusive prefix of v within the 256-thread block and the block total in *total.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *total) {
  __shared__ uint32_t wsum[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) wsum[wid] = x;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { if (k < wid) base += wsum[k]; tot += wsum[k]; }
  __syncthreads();
  *total = tot;
  return base + x - v;
}

// generic 3-phase scan: phase A block totals, phase B scan of totals (one block), phase C final
__global__ void __launch_bounds__(256) k_scan_a(const uint32_t *in, uint32_t n, uint64_t *blk_tot) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  uint32_t tot;
  (void)block_excl_scan(i < n ? in[i] : 0u, &tot);
  if (threadIdx.x == 0) blk_tot[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(256) k_scan_b(uint64_t *blk_tot, uint32_t nb) {
  // nb block totals -> exclusive prefix in place, total at blk_tot[nb]; processed in chunks of 256 by one block
  __shared__ uint64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nb; base += 256) {
    const uint32_t i = base + threadIdx.x;
    const uint64_t v = i < nb ? blk_tot[i] : 0;
    // 64-bit wave scan
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __shared__ uint64_t ws[4];
    uint64_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    if (lane == 63) ws[wid] = x;
    __syncthreads();
    uint64_t b = carry, tot = 0;
    for (int k = 0; k < 4; ++k) { if (k < wid) b += ws[k]; tot += ws[k]; }
    if (i < nb) blk_tot[i] = b + x - v;
    __syncthreads();
    if (threadIdx.x == 0) carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) blk_tot[nb] = carry;
}
__global__ void __launch_bounds__(256) k_scan_c(const uint32_t *in, uint32_t n, const uint64_t *blk_off, uint64_t *out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  uint32_t tot;
  const uint32_t e = block_excl_scan(i < n ? in[i] : 0u, &tot);
  if (i < n) out[i] = blk_off[blockIdx.x] + e;
  if (i == 0) out[n] = blk_off[gridDim.x];
}

#define g_scan_tmp (g_cur->scan_tmp)
#define g_scan_tmp_n (g_cur->scan_tmp_n)
int launch_scan(const uint32_t *in, uint64_t *out, uint32_t n) {
  FQ_PRE();
  if (n == 0) { uint64_t z = 0; return h2d(out, &z, 8) ? -3 : sync(); }
  const unsigned nb = nblk(n, 256);
  if (g_scan_tmp_n < nb + 1) {
    dfree(g_scan_tmp);
    g_scan_tmp_n = (size_t)nb * 2 + 64;
    g_scan_tmp = (uint64_t *)dmalloc(g_scan_tmp_n * 8);
    if (!g_scan_tmp) return -4;
  }
  hipLaunchKernelGGL(k_scan_a, dim3(nb), dim3(256), 0, g_stream, in, n, g_scan_tmp);
  hipLaunchKernelGGL(k_scan_b, dim3(1), dim3(256), 0, g_stream, g_scan_tmp, nb);
  hipLaunchKernelGGL(k_scan_c, dim3(nb), dim3(256), 0, g_stream, in, n, g_scan_tmp, out);
  FQ_HIP(hipGetLastError());
  return 0;
}

// ---- ordered compaction of survivors --------------------------------------------------------------
// per pair p: keep = !(f0 && f1); reads kept: (!f0) then (!f1).  Ballot + popcount give each lane its
// rank inside the wavefront; LDS combines the four wavefronts of a block; a scanned array of block
// totals gives the global offset.  Output order is ascending pair index (stable), which is the order the
// consumers (StatCollector / BAM writer) require.
__global__ void __launch_bounds__(256) k_compact_a(const uint8_t *filt, int n_pairs, uint32_t *pair_cnt, uint32_t *read_cnt) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  uint32_t keep = 0, nr = 0;
  if (p < n_pairs) {
    const uint32_t f0 = filt[p], f1 = filt[n_pairs + p];
    keep = !(f0 && f1);
    nr = (!f0) + (!f1);
  }
  const unsigned long long bal = __ballot(keep);
  __shared__ uint32_t s_pairs[4], s_reads[4];
  uint32_t x = nr;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
  if ((threadIdx.x & 63) == 0) { s_pairs[threadIdx.x >> 6] = (uint32_t)__popcll(bal); s_reads[threadIdx.x >> 6] = x; }
  __syncthreads();
  if (threadIdx.x == 0) {
    pair_cnt[blockIdx.x] = s_pairs[0] + s_pairs[1] + s_pairs[2] + s_pairs[3];
    read_cnt[blockIdx.x] = s_reads[0] + s_reads[1] + s_reads[2] + s_reads[3];
  }
}
__global__ void __launch_bounds__(256) k_compact_c(const uint8_t *filt, int n_pairs, const uint64_t *pair_off, const uint64_t *read_off,
                                                   int32_t *read_list, int32_t *sidx, int32_t *pair_list, int32_t *counts) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  uint32_t keep = 0, f0 = 1, f1 = 1;
  if (p < n_pairs) { f0 = filt[p]; f1 = filt[n_pairs + p]; keep = !(f0 && f1); }
  const uint32_t nr = (!f0) + (!f1);
  uint32_t tot_p, tot_r;
  const uint32_t rank_p = block_excl_scan(keep, &tot_p);
  const uint32_t rank_r = block_excl_scan(nr, &tot_r);
  if (p < n_pairs) {
    if (keep) pair_list[pair_off[blockIdx.x] + rank_p] = p;
    uint32_t s = (uint32_t)read_off[blockIdx.x] + rank_r;
    if (!f0) { read_list[s] = p; sidx[p] = (int32_t)s; ++s; } else sidx[p] = -1;
    if (!f1) { read_list[s] = n_pairs + p; sidx[n_pairs + p] = (int32_t)s; } else sidx[n_pairs + p] = -1;
  }
  if (p == 0) { counts[0] = (int32_t)read_off[gridDim.x]; counts[1] = (int32_t)pair_off[gridDim.x]; }
}
#define g_cmp_cnt (g_cur->cmp_cnt)
#define g_cmp_off (g_cur->cmp_off)
#define g_cmp_n (g_cur->cmp_n)
int launch_compact(const uint8_t *filtered, int n_pairs, int32_t *read_list, int32_t *sidx, int32_t *pair_list, int32_t *counts) {
  FQ_PRE();
  if (n_pairs <= 0) return dzero(counts, 8);
  const unsigned nb = nblk((uint64_t)n_pairs, 256);
  if (g_cmp_n < nb) {
    dfree(g_cmp_cnt); dfree(g_cmp_off);
    g_cmp_n = (size_t)nb * 2 + 64;
    g_cmp_cnt = (uint32_t *)dmalloc(g_cmp_n * 2 * 4);
    g_cmp_off = (uint64_t *)dmalloc((g_cmp_n + 2) * 2 * 8);
    if (!g_cmp_cnt || !g_cmp_off) return -4;
  }
  uint32_t *pc = g_cmp_cnt, *rc = g_cmp_cnt + g_cmp_n;
  uint64_t *po = g_cmp_off, *ro = g_cmp_off + g_cmp_n + 2;
  hipLaunchKernelGGL(k_compact_a, dim3(nb), dim3(256), 0, g_stream, filtered, n_pairs, pc, rc);
  int rc1 = launch_scan(pc, po, nb);
  if (rc1) return rc1;
  rc1 = launch_scan(rc, ro, nb);
  if (rc1) return rc1;
  hipLaunchKernelGGL(k_compact_c, dim3(nb), dim3(256), 0, g_stream, filtered, n_pairs, po, ro, read_list, sidx, pair_list, counts);
  FQ_HIP(hipGetLastError());
  return 0;
}

// ---- filter bitmap construction from a list of set bits (index load) ---------------------------------
__global__ void __launch_bounds__(256) k_bitmap_scatter(uint8_t *bitmap, const uint32_t *bits, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t x = bits[i];
  atomicOr((unsigned int *)(bitmap + ((x >> 5) << 2)), 1u << (x & 31));   // little endian: bit x&7 of byte x>>3
}
int launch_bitmap_scatter(uint8_t *bitmap, const uint32_t *bits, uint64_t n) {
  FQ_PRE();
  if (!n) return 0;
  hipLaunchKernelGGL(k_bitmap_scatter, dim3(nblk(n, 256)), dim3(256), 0, g_stream, bitmap, bits, n);
  FQ_HIP(hipGetLastError());
  return 0;
}

// ... and straight from the reference in HBM (fq_kernels.h: fq_bitmap_kmer_thread)
__global__ void __launch_bounds__(256) k_bitmap_kmers(FqBitmapArgs a) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= 2 * a.l_pac) return;
  fq_bitmap_kmer_thread(a, idx, [&](int t, uint32_t x) { atomicOr(&a.bitmap[t][x >> 5], 1u << (x & 31)); });
}
int launch_bitmap_kmers(const FqBitmapArgs &a) {
  FQ_PRE();
  if (a.l_pac <= 0 || a.n_rec <= 0) return 0;
  hipLaunchKernelGGL(k_bitmap_kmers, dim3(nblk((uint64_t)(2 * a.l_pac), 256)), dim3(256), 0, g_stream, a);
  FQ_HIP(hipGetLastError());
  return 0;
}

// ---- FASTQ front end (fq_frontend.h) ----------------------------------------------------------------------------------
static FqzCrcConst *g_crc_const[64];
const FqzCrcConst *crc_const() {
  std::lock_guard<std::mutex> lk(g_dev_mu);
  const int dev = g_cur->device;
  if (!g_crc_const[dev]) {
    FqzCrcConst *h = new FqzCrcConst;
    fqz_crc_const_make(h);
    void *d = nullptr;
    if (hipMalloc(&d, sizeof(FqzCrcConst)) != hipSuccess || hipMemcpy(d, h, sizeof(FqzCrcConst), hipMemcpyHostToDevice) != hipSuccess) { g_err = "crc_const: hipMalloc / hipMemcpy failed"; delete h; return nullptr; }
    delete h;
    g_crc_const[dev] = (FqzCrcConst *)d;
  }
  return g_crc_const[dev];
}
// One wavefront per BGZF member: the output ring and the tables in LDS (6.4 KB per wavefront: twenty-four wavefronts per CU).
#ifndef FQZ_WAVES_PER_EU
#define FQZ_WAVES_PER_EU 6
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FQZ_WAVES_PER_EU))) k_inflate_bgzf(FqInflateArgs a, FqInflateArgs b) {
  __shared__ __attribute__((aligned(FQZ_RING))) FqzLds lds;      // (the ring's LDS address has no bit of a ring offset set: fq_frontend.h)
  // two member tables in one launch (the two files of a pair): a launch's last round of wavefronts is as full as the member count allows
  const bool second = (int)blockIdx.x >= a.n_mem;
  const FqInflateArgs &A = second ? b : a;
  const int m = second ? (int)blockIdx.x - a.n_mem : (int)blockIdx.x;
  const uint32_t st = fqz_inflate_member(A, m, lds);
  if (threadIdx.x == 0) A.status[m] = st;
}
int launch_inflate2(const FqInflateArgs &a, const FqInflateArgs &b) {
  FQ_PRE();
  const int n = (a.n_mem > 0 ? a.n_mem : 0) + (b.n_mem > 0 ? b.n_mem : 0);
  if (n <= 0) return 0;
  FqInflateArgs x = a, y = b;
  if (x.n_mem < 0) x.n_mem = 0;
  if (y.n_mem < 0) y.n_mem = 0;
  hipLaunchKernelGGL(k_inflate_bgzf, dim3((unsigned)n), dim3(64), 0, g_stream, x, y);
  FQ_HIP(hipGetLastError());
  return 0;
}
int launch_inflate(const FqInflateArgs &a) { FqInflateArgs none{}; return launch_inflate2(a, none); }

// Line ends of a text: every thread looks at 16 bytes (one aligned load), a block at 4 KiB; the blocks' counts are scanned (launch_scan)
// and the positions written in order: ballot-free ranks from a prefix sum over the block's 256 counts in LDS.
__device__ __forceinline__ uint32_t fqt_nl_mask16(const uint8_t *text, uint32_t n, uint32_t at, uint32_t lo) {   // bit j: text[at + j] is a line end (at a multiple of 16)
  if (at >= n) return 0;
  const FqU4 v = *(const FqU4 *)(text + at);
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  uint32_t m = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) m |= (uint32_t)(((w[q] >> (8 * j)) & 0xffu) == 0x0au) << (4 * q + j);
  if (at + 16 > n) m &= (1u << (n - at)) - 1u;
  if (at < lo) m &= ~((1u << (lo - at)) - 1u);      // (lo < 16: bytes in front of the text's first, there for the alignment of the loads)
  return m;
}
__global__ void __launch_bounds__(256) k_nl_count(const uint8_t *text, uint32_t n, uint32_t lo, uint32_t *blk_cnt) {
  const uint32_t at = (blockIdx.x * 256u + threadIdx.x) * 16u;
  uint32_t c = (uint32_t)__popc(fqt_nl_mask16(text, n, at, lo));
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) c += (uint32_t)__shfl_xor((int)c, d, 64);
  __shared__ uint32_t s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) blk_cnt[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ void __launch_bounds__(256) k_nl_fill(const uint8_t *text, uint32_t n, uint32_t lo, const uint64_t *blk_off, uint32_t *nl, uint32_t cap) {
  const uint32_t at = (blockIdx.x * 256u + threadIdx.x) * 16u;
  uint32_t m = fqt_nl_mask16(text, n, at, lo);
  const uint32_t c = (uint32_t)__popc(m);
  // exclusive prefix of the counts inside the block: inside the wavefront by shuffles, across the four wavefronts through LDS
  uint32_t x = c;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)x, d, 64); if (lane >= d) x += y; }
  __shared__ uint32_t s[4];
  if (lane == 63) s[threadIdx.x >> 6] = x;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += s[w];
  uint64_t rank = blk_off[blockIdx.x] + base + (x - c);
  while (m) {
    const int j = __ffs((int)m) - 1;
    m &= m - 1;
    if (rank < cap) nl[rank] = at + (uint32_t)j;
    ++rank;
  }
}
__global__ void __launch_bounds__(64) k_nl_total(const uint64_t *blk_off, uint32_t nb, uint32_t *count) {
  if (threadIdx.x == 0) { const uint64_t t = blk_off[nb]; *count = t > 0xffffffffull ? 0xffffffffu : (uint32_t)t; }
}
int launch_nl_index(const uint8_t *text, uint32_t n, uint32_t lo, uint32_t *nl, uint32_t cap, uint32_t *count) {
  FQ_PRE();
  const unsigned nb = nblk((uint64_t)n, 4096);
  if (nb == 0) { FQ_HIP(hipMemsetAsync(count, 0, 4, g_stream)); return 0; }
  if (g_cmp_n < (size_t)nb + 1) {
    dfree(g_cmp_cnt); dfree(g_cmp_off);
    g_cmp_n = (size_t)nb * 2 + 64;
    g_cmp_cnt = (uint32_t *)dmalloc(g_cmp_n * 2 * 4);      // (launch_compact keeps two count arrays here)
    g_cmp_off = (uint64_t *)dmalloc((g_cmp_n + 2) * 2 * 8);
    if (!g_cmp_cnt || !g_cmp_off) { g_cmp_n = 0; return -4; }
  }
  hipLaunchKernelGGL(k_nl_count, dim3(nb), dim3(256), 0, g_stream, text, n, lo, g_cmp_cnt);
  FQ_HIP(hipGetLastError());
  if (launch_scan(g_cmp_cnt, g_cmp_off, nb)) return -3;
  hipLaunchKernelGGL(k_nl_fill, dim3(nb), dim3(256), 0, g_stream, text, n, lo, (const uint64_t *)g_cmp_off, nl, cap);
  hipLaunchKernelGGL(k_nl_total, dim3(1), dim3(64), 0, g_stream, (const uint64_t *)g_cmp_off, nb, count);
  FQ_HIP(hipGetLastError());
  return 0;
}
__global__ void __launch_bounds__(256) k_tok_rec(FqTokArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  FqTokStat t = i < a.n_rec ? fqt_rec_thread(a, i) : fqt_stat_none();
  // the wavefront's records folded into one set of statistics, then one atomic each by its first lane
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    t.first_bad = min(t.first_bad, (uint32_t)__shfl_xor((int)t.first_bad, d, 64));
    t.max_name = max(t.max_name, (uint32_t)__shfl_xor((int)t.max_name, d, 64));
    t.min_name = min(t.min_name, (uint32_t)__shfl_xor((int)t.min_name, d, 64));
    t.max_len = max(t.max_len, (uint32_t)__shfl_xor((int)t.max_len, d, 64));
    t.min_len = min(t.min_len, (uint32_t)__shfl_xor((int)t.min_len, d, 64));
  }
  if ((threadIdx.x & 63) == 0) fqt_stat_commit(a, t);
}
__global__ void __launch_bounds__(256) k_tok_pieces(FqTokArgs a, int64_t n) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g < n) fqt_piece_thread(a, g);
}
__global__ void __launch_bounds__(256) k_slot_bases(FqSlotArgs a) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s < a.n_slots) fqt_slot_bases_thread(a, s);
}
__global__ void __launch_bounds__(256) k_slot_names(FqSlotArgs a) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s < a.n_slots) fqt_slot_names_thread(a, s);
}
__global__ void __launch_bounds__(256) k_text_gather(FqTextGatherArgs a, int64_t n) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g < n) fqt_gather_piece(a, g);
}
__global__ void __launch_bounds__(256) k_text_trim_all(FqTextTrimArgs a) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r < a.n_rows) fqt_trim_all_thread(a, r);
}
int launch_tok_rec(const FqTokArgs &a) {
  FQ_PRE();
  if (a.n_rec <= 0) return 0;
  hipLaunchKernelGGL(k_tok_rec, dim3(nblk((uint64_t)a.n_rec, 256)), dim3(256), 0, g_stream, a);
  FQ_HIP(hipGetLastError());
  return 0;
}
int launch_tok_pieces(const FqTokArgs &a) {
  FQ_PRE();
  const int64_t n = (int64_t)a.n_rec * ((a.max_len + 31) >> 5);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_tok_pieces, dim3(nblk((uint64_t)n, 256)), dim3(256), 0, g_stream, a, n);
  FQ_HIP(hipGetLastError());
  return 0;
}
int launch_slot_bases(const FqSlotArgs &a) {
  FQ_PRE();
  if (a.n_rec <= 0) return 0;
  hipLaunchKernelGGL(k_slot_bases, dim3(nblk((uint64_t)a.n_slots, 256)), dim3(256), 0, g_stream, a);
  FQ_HIP(hipGetLastError());
  return 0;
}
__global__ void __launch_bounds__(256) k_names_plain(FqSlotArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < a.n_rec) fqt_names_plain_thread(a, i);
}
int launch_slot_names(const FqSlotArgs &a) {
  FQ_PRE();
  if (a.n_rec <= 0) return 0;
  if (a.plain_names) {
    hipLaunchKernelGGL(k_names_plain, dim3(nblk((uint64_t)a.n_rec, 256)), dim3(256), 0, g_stream, a);
    if (a.mode != 0) { FQ_HIP(hipGetLastError()); return 0; }
  }
  hipLaunchKernelGGL(k_slot_names, dim3(nblk((uint64_t)a.n_slots, 256)), dim3(256), 0, g_stream, a);
  FQ_HIP(hipGetLastError());
  return 0;
}
int launch_text_gather(const FqTextGatherArgs &a) {
  FQ_PRE();
  const int64_t n = (int64_t)a.n_out * (a.stride >> 4);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_text_gather, dim3(nblk((uint64_t)n, 256)), dim3(256), 0, g_stream, a, n);
  FQ_HIP(hipGetLastError());
  return 0;
}
int launch_text_trim_all(const FqTextTrimArgs &a) {
  FQ_PRE();
  if (a.n_rows <= 0) return 0;
  hipLaunchKernelGGL(k_text_trim_all, dim3(nblk((uint64_t)a.n_rows, 256)), dim3(256), 0, g_stream, a);
  FQ_HIP(hipGetLastError());
  return 0;
}
int dfill32(void *dst, uint32_t v, size_t n_words) { FQ_PRE(); if (n_words) FQ_HIP(hipMemsetD32Async((hipDeviceptr_t)dst, (int)v, n_words, g_stream)); return 0; }

// ---- launch wrappers ------------------------------------------------------------------------------
// The filter kernel fills the device and is bound by its random probes: two of them at once only slow each other down.  The
// launches of all streams (threads) of a device are therefore chained on the device: each waits for the previous one's
// completion event, so they run back to back without a host round trip in between, while everything else a stream does
// overlaps freely.  FQ_FILTER_NO_TURNS lets them overlap.
int launch_prep(const FqPrepArgs &a) {
  FQ_PRE();
  if (a.n_reads <= 0) return 0;
  hipEvent_t e0, e1;
  kernel_events(FQ_K_PREP_KERNEL, &e0, &e1);
  if (!g_cur->tune.filter_no_turns) {
    const int dev = g_cur->device;
    std::lock_guard<std::mutex> lk(g_dev_mu);
    if (g_prep_last[dev] && g_prep_owner[dev] != g_cur) FQ_HIP(hipStreamWaitEvent(g_stream, g_prep_last[dev], 0));
    hipExtLaunchKernelGGL(k_prep, dim3(nblk((uint64_t)a.n_reads, 256)), dim3(256), 0, g_stream, e0, e1, 0, a);
    FQ_HIP(hipEventRecord(g_cur->prep_done, g_stream));
    g_prep_last[dev] = g_cur->prep_done; g_prep_owner[dev] = g_cur;
  } else {
    hipExtLaunchKernelGGL(k_prep, dim3(nblk((uint64_t)a.n_reads, 256)), dim3(256), 0, g_stream, e0, e1, 0, a);
  }
  FQ_HIP(hipGetLastError());
  return 0;
}
__global__ void __launch_bounds__(256) k_surv_gather(const int32_t *pair_list, int n_surv, int n_pairs, const int32_t *len_trim, const uint8_t *filtered, const int32_t *sidx, FqSurvInfo *out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < 2 * n_surv) fq_surv_gather_thread(pair_list, n_pairs, len_trim, filtered, sidx, out, t);
}
int launch_surv_gather(const int32_t *pair_list, int n_surv, int n_pairs, const int32_t *len_trim, const uint8_t *filtered, const int32_t *sidx, FqSurvInfo *out) {
  FQ_PRE();
  if (n_surv <= 0) return 0;
  hipLaunchKernelGGL(k_surv_gather, dim3(nblk((uint64_t)n_surv * 2, 256)), dim3(256), 0, g_stream, pair_list, n_surv, n_pairs, len_trim, filtered, sidx, out);
  FQ_HIP(hipGetLastError());
  return 0;
}
__global__ void __launch_bounds__(256) k_prep_packed(FqPrepPackedArgs a) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < a.n_reads) fq_prep_packed_thread(a, r);
}
int launch_prep_packed(const FqPrepPackedArgs &a) {
  FQ_PRE();
  if (a.n_reads <= 0) return 0;
  hipEvent_t e0, e1;
  kernel_events(FQ_K_PREP_KERNEL, &e0, &e1);
  const dim3 grid(nblk((uint64_t)a.n_reads, 256));
  // The filter kernel is the call's one bandwidth-bound launch; beside other streams' search kernels (latency-bound, long, every CU's wave
  // slots taken) its workgroups wait for slots.  prep_priority: it goes to a stream of the highest priority, whose workgroups the dispatcher
  // places first; the call's stream waits for it.
  hipStream_t st = g_stream;
  if (g_cur->tune.prep_priority) {
    // (ONE such stream per device, shared by its contexts -- their filter launches wait for each other anyway; a stream per context would be
    //  sixteen more streams than there are hardware queues)
    {
      std::lock_guard<std::mutex> lk(g_dev_mu);
      if (!g_prio_stream[g_cur->device]) {
        int lo = 0, hi = 0;
        FQ_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        FQ_HIP(hipStreamCreateWithPriority(&g_prio_stream[g_cur->device], hipStreamNonBlocking, hi));
      }
    }
    if (!g_cur->prio_fork) FQ_HIP(hipEventCreateWithFlags(&g_cur->prio_fork, hipEventDisableTiming));
    FQ_HIP(hipEventRecord(g_cur->prio_fork, g_stream));
    FQ_HIP(hipStreamWaitEvent(g_prio_stream[g_cur->device], g_cur->prio_fork, 0));
    st = g_prio_stream[g_cur->device];
  }
  if (!g_cur->tune.filter_no_turns) {   // chained per device like launch_prep
    const int dev = g_cur->device;
    std::lock_guard<std::mutex> lk(g_dev_mu);
    if (g_prep_last[dev] && g_prep_owner[dev] != g_cur) FQ_HIP(hipStreamWaitEvent(st, g_prep_last[dev], 0));
    hipExtLaunchKernelGGL(k_prep_packed, grid, dim3(256), 0, st, e0, e1, 0, a);
    FQ_HIP(hipEventRecord(g_cur->prep_done, st));
    g_prep_last[dev] = g_cur->prep_done; g_prep_owner[dev] = g_cur;
    if (st != g_stream) FQ_HIP(hipStreamWaitEvent(g_stream, g_cur->prep_done, 0));
  } else {
    hipExtLaunchKernelGGL(k_prep_packed, grid, dim3(256), 0, st, e0, e1, 0, a);
    if (st != g_stream) { FQ_HIP(hipEventRecord(g_cur->prep_done, st)); FQ_HIP(hipStreamWaitEvent(g_stream, g_cur->prep_done, 0)); }
  }
  FQ_HIP(hipGetLastError());
  return 0;
}
__global__ void __launch_bounds__(256) k_surv_map(const int32_t *pair_list, int n_surv, int n_pairs, const uint8_t *filtered, const int32_t *sidx, FqSurvInfo *out,
                                                  int32_t *row_map, int32_t *read_list_c, int32_t *crow_of) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < 2 * n_surv) fq_surv_map_thread(pair_list, n_pairs, filtered, sidx, out, row_map, read_list_c, crow_of, t);
}
int launch_surv_map(const int32_t *pair_list, int n_surv, int n_pairs, const uint8_t *filtered, const int32_t *sidx, FqSurvInfo *out,
                    int32_t *row_map, int32_t *read_list_c, int32_t *crow_of) {
  FQ_PRE();
  if (n_surv <= 0) return 0;
  hipLaunchKernelGGL(k_surv_map, dim3(nblk((uint64_t)n_surv * 2, 256)), dim3(256), 0, g_stream, pair_list, n_surv, n_pairs, filtered, sidx, out, row_map, read_list_c, crow_of);
  FQ_HIP(hipGetLastError());
  return 0;
}
__global__ void __launch_bounds__(256) k_unpack(FqUnpackArgs a) {   // one 16-byte piece of the output per thread
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < (int64_t)a.n_rows * (a.stride >> 4)) fq_unpack_piece(a, g);
}
int launch_unpack(const FqUnpackArgs &a) {
  FQ_PRE();
  if (a.n_rows <= 0) return 0;
  if (a.stride & 15) { g_err = "launch_unpack: the compact row stride must be a multiple of 16"; return -1; }
  hipLaunchKernelGGL(k_unpack, dim3(nblk((uint64_t)a.n_rows * (uint64_t)(a.stride >> 4), 256)), dim3(256), 0, g_stream, a);
  FQ_HIP(hipGetLastError());
  return 0;
}
__global__ void __launch_bounds__(256) k_patch(FqPatchArgs a) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < a.n_exc) fq_patch_thread(a, q);
}
int launch_patch(const FqPatchArgs &a) {
  FQ_PRE();
  if (a.n_exc <= 0) return 0;
  hipLaunchKernelGGL(k_patch, dim3(nblk((uint64_t)a.n_exc, 256)), dim3(256), 0, g_stream, a);
  FQ_HIP(hipGetLastError());
  return 0;
}
__global__ void __launch_bounds__(256) k_trim(FqTrimArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < a.n_rows) fq_trim_thread(a, t);
}
int launch_trim(const FqTrimArgs &a) {
  FQ_PRE();
  if (a.n_rows <= 0) return 0;
  hipLaunchKernelGGL(k_trim, dim3(nblk((uint64_t)a.n_rows, 256)), dim3(256), 0, g_stream, a);
  FQ_HIP(hipGetLastError());
  return 0;
}
__global__ void __launch_bounds__(256) k_trim_all(FqTrimAllArgs a) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < a.n_reads) fq_trim_all_thread(a, r);
}
int launch_trim_all(const FqTrimAllArgs &a) {
  FQ_PRE();
  if (a.n_reads <= 0) return 0;
  hipLaunchKernelGGL(k_trim_all, dim3(nblk((uint64_t)a.n_reads, 256)), dim3(256), 0, g_stream, a);
  FQ_HIP(hipGetLastError());
  return 0;
}
int launch_width(const FqWidthArgs &a) {
  FQ_PRE();
  if (a.n_work <= 0) return 0;
  hipEvent_t e0, e1;
  kernel_events(FQ_K_WIDTH_KERNEL, &e0, &e1);
  if (!g_cur->tune.width_both_strands) {
    const unsigned per_strand = nblk((uint64_t)a.n_work, 256);
    const unsigned grid = ((per_strand + 3) / 4) * 8;      // groups of 8 workgroups: 4 per strand
    hipExtLaunchKernelGGL(k_width_strand, dim3(grid), dim3(256), (size_t)a.o.seed_len * 256, g_stream, e0, e1, 0, a);
  } else
  hipExtLaunchKernelGGL(k_width, dim3(nblk((uint64_t)a.n_work, 256)), dim3(256), (size_t)2 * (size_t)a.o.seed_len * 256, g_stream, e0, e1, 0, a);
  FQ_HIP(hipGetLastError());
  return 0;
}
// counting sort of the work items by scheduling key: block-local counts in LDS, one global atomic per key and block
__global__ void __launch_bounds__(256) k_order_count(const uint8_t *bid_end, int n, int n_hard, uint32_t *cnt) {
  __shared__ uint32_t h[FQ_ORDER_KEYS];
  if (threadIdx.x < FQ_ORDER_KEYS) h[threadIdx.x] = 0;
  __syncthreads();
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < n) atomicAdd(&h[fq_order_key(bid_end, w, n_hard)], 1u);
  __syncthreads();
  if (threadIdx.x < FQ_ORDER_KEYS && h[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], h[threadIdx.x]);
}
__global__ void __launch_bounds__(256) k_order_scatter(const uint8_t *bid_end, int n, int n_hard, uint32_t *cnt, int32_t *order, int asc) {
  __shared__ uint32_t h[FQ_ORDER_KEYS], base[FQ_ORDER_KEYS];
  if (threadIdx.x < FQ_ORDER_KEYS) h[threadIdx.x] = 0;
  __syncthreads();
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  int key = 0;
  uint32_t rank = 0;
  if (w < n) { key = fq_order_key(bid_end, w, n_hard); rank = atomicAdd(&h[key], 1u); }
  __syncthreads();
  if (threadIdx.x < FQ_ORDER_KEYS) {
    // start of key k in the output = number of items with a larger key (descending order); cursors live behind the counts
    uint32_t start = 0;
    if (asc) { for (int k = 0; k < (int)threadIdx.x; ++k) start += cnt[k]; }
    else { for (int k = FQ_ORDER_KEYS - 1; k > (int)threadIdx.x; --k) start += cnt[k]; }
    base[threadIdx.x] = h[threadIdx.x] ? start + atomicAdd(&cnt[FQ_ORDER_KEYS + threadIdx.x], h[threadIdx.x]) : 0u;
  }
  __syncthreads();
  if (w < n) order[base[key] + rank] = w;
  if (blockIdx.x == 0 && threadIdx.x == 0) {   // length of the first block of the queue (keys of the upper half; descending order)
    uint32_t sp = 0;
    if (asc) { for (int k = 0; k < FQ_ORDER_KEYS / 2; ++k) sp += cnt[k]; }
    else { for (int k = FQ_ORDER_KEYS / 2; k < FQ_ORDER_KEYS; ++k) sp += cnt[k]; }
    cnt[2 * FQ_ORDER_KEYS] = sp;
  }
}
// the unfinished work items of one queue segment (both blocks' shares of it): their search indices, appended in no particular order
__global__ void __launch_bounds__(256) k_collect(const int32_t *order, const uint32_t *split, int n_work, int seg, int n_seg, const uint32_t *status, const int32_t *work,
                                                  int32_t *out, uint32_t *count) {
  const uint32_t sp = split ? *split : (uint32_t)n_work;
  const uint32_t blk_start[2] = {0u, sp}, blk_len[2] = {sp, (uint32_t)n_work - sp};
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  for (int b = 0; b < 2; ++b) {
    uint32_t lo, hi;
    fq_seg_range(blk_len[b], seg, n_seg, &lo, &hi);
    for (uint32_t pos = blk_start[b] + lo + t; pos < blk_start[b] + hi; pos += stride) {
      const int w = order ? order[pos] : (int)pos;
      if (status[w]) out[atomicAdd(count, 1u)] = work[w];
    }
  }
}
int launch_collect(const int32_t *order, const uint32_t *split, int n_work, int seg, int n_seg, const uint32_t *status, const int32_t *work, int32_t *out, uint32_t *count) {
  FQ_PRE();
  if (n_work <= 0) return 0;
  hipLaunchKernelGGL(k_collect, dim3(1024), dim3(256), 0, g_stream, order, split, n_work, seg, n_seg, status, work, out, count);
  FQ_HIP(hipGetLastError());
  return 0;
}
int launch_order(const uint8_t *bid_end, int n, int n_hard, int32_t *order, uint32_t *cnt) {
  FQ_PRE();
  if (n <= 0) return 0;
  FQ_HIP(hipMemsetAsync(cnt, 0, (2 * FQ_ORDER_KEYS + 1) * 4, g_stream));
  hipLaunchKernelGGL(k_order_count, dim3(nblk((uint64_t)n, 256)), dim3(256), 0, g_stream, bid_end, n, n_hard, cnt);
  hipLaunchKernelGGL(k_order_scatter, dim3(nblk((uint64_t)n, 256)), dim3(256), 0, g_stream, bid_end, n, n_hard, cnt, order, g_cur->tune.gap_order_asc ? 1 : 0);   // (experiment knob: ascending)
  FQ_HIP(hipGetLastError());
  return 0;
}
int gap_lane_slots(const FqGapArgs &a) {
  if (a.n_work <= 0) return 0;
  if (a.tier.coop) {   // wavefronts, one pool each
    const unsigned cw = g_cur->tune.gap_coop_waves > 0 ? (unsigned)g_cur->tune.gap_coop_waves : 1024u;
    unsigned w = std::min<unsigned>((unsigned)a.n_work, a.tier.exact ? 64u : cw);
    if (a.max_waves > 0) w = std::min<unsigned>(w, (unsigned)a.max_waves);
    return (int)w;
  }
  const int env_waves = g_cur->tune.gap_waves_per_cu;
  const unsigned need = nblk((uint64_t)a.n_work, 64);
  unsigned per_cu = 8;
  if (a.tier.pool_cap <= 65535u) {
    const size_t lds = (size_t)64 * (size_t)a.o.n_buckets * 2 + (size_t)64 * FQ_COLD_N * 4;   // bucket heads + the lanes' cold words
    per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(a.tier.nogap ? FQ_NOGAP_WAVES_PER_CU : 16, (150 * 1024) / lds));
  }
  if (env_waves > 0) per_cu = std::min<unsigned>(per_cu, (unsigned)env_waves);
  unsigned waves = std::min(need, 256u * per_cu);
  if (a.max_waves > 0) waves = std::min<unsigned>(waves, (unsigned)a.max_waves);
  return (int)(waves * 64u);
}
int launch_gap(const FqGapArgs &a_in) {
  FQ_PRE();
  if (a_in.n_work <= 0) return 0;
  FqGapArgs a = a_in;
  const int env_refill = g_cur->tune.gap_refill_min;
  a.refill_min = env_refill > 0 ? env_refill : a_in.refill_min > 0 ? a_in.refill_min : FQ_REFILL_MIN;   // (tuning key, else the caller's choice for this round, else 64)
  FQ_HIP(hipMemsetAsync(a.queue, 0, 8, g_stream));
  hipEvent_t e0, e1;
  kernel_events(a.tier.nogap ? FQ_K_GAP_NOGAP : FQ_K_GAP_KERNEL, &e0, &e1);
  if (a.tier.coop) {
    hipExtLaunchKernelGGL(k_gap_coop, dim3((unsigned)gap_lane_slots(a)), dim3(64), 0, g_stream, e0, e1, 0, a);
  } else {
    // LDS-resident bucket heads when slot indices fit 16 bits
    const size_t lds = (size_t)64 * (size_t)a.o.n_buckets * 2 + (size_t)64 * FQ_COLD_N * 4;   // bucket heads + the lanes' cold words
    const unsigned grid = (unsigned)gap_lane_slots(a) / 64u;
    if (a.tier.exact) { g_err = "launch_gap: the exact tier is the wavefront kernel's"; return -1; }
    const bool stock = FqOptsStock::matches(a.o) && !g_cur->tune.gap_generic_opts;
    if (a.tier.pool_cap <= 65535u && a.tier.nogap) hipExtLaunchKernelGGL(stock ? k_gap_nogap_stock : k_gap_nogap_lds, dim3(grid), dim3(64), lds, g_stream, e0, e1, 0, a);
    else if (a.tier.pool_cap <= 65535u) hipExtLaunchKernelGGL(stock ? k_gap_persist_stock : k_gap_persist_lds, dim3(grid), dim3(64), lds, g_stream, e0, e1, 0, a);
    else hipExtLaunchKernelGGL(k_gap_persist, dim3(grid), dim3(64), 0, g_stream, e0, e1, 0, a);
  }
  FQ_HIP(hipGetLastError());
  return 0;
}
int launch_pack_aln(const FqAln *aln, const uint32_t *n_aln, const uint64_t *off, uint32_t cap, uint32_t n_work, FqAln *packed) {
  FQ_PRE();
  if (!n_work) return 0;
  hipLaunchKernelGGL(k_pack_aln, dim3(nblk(n_work, 256)), dim3(256), 0, g_stream, aln, n_aln, off, cap, n_work, packed);
  FQ_HIP(hipGetLastError());
  return 0;
}
int launch_sa(const FqSaArgs &a) {
  FQ_PRE();
  if (!a.n_rows) return 0;
  hipEvent_t e0, e1;
  kernel_events(FQ_K_SA_KERNEL, &e0, &e1);
  hipExtLaunchKernelGGL(k_sa, dim3(nblk(a.n_rows, 256)), dim3(256), 0, g_stream, e0, e1, 0, a);
  FQ_HIP(hipGetLastError());
  return 0;
}
static const size_t kLdsBudget = 150 * 1024;
int launch_sw(const FqSwArgs &a) {
  FQ_PRE();
  if (a.n_task <= 0) return 0;
  size_t lds = (size_t)(2 * (a.RL + 2) + 3 * (a.RL + 1)) * 4 + ((a.RL + 16) & ~15) + ((a.QL + 16) & ~15) + ((a.RL + a.QL + 16) & ~15);
  if (lds > kLdsBudget) { g_err = "SW window too large for LDS (" + std::to_string(a.RL) + " bases)"; return -5; }
  FqSwArgs b = a;
  const size_t trace_bytes = (size_t)(a.RL + 1) * (a.QL + 1) + 16;
  // The trace matrix in LDS makes a task faster but limits a CU to one block; with more tasks than CUs it is better to keep it in
  // the task's global scratch and have every task resident at once (each spends most of its time in the serial reverse pass).
  b.trace_in_lds = (lds + trace_bytes <= kLdsBudget && a.n_task <= 256) ? 1 : 0;
  b.serial_reverse = g_cur->tune.sw_serial_reverse;
  if (b.trace_in_lds) lds += trace_bytes;
  hipEvent_t e0, e1;
  kernel_events(FQ_K_SW_KERNEL, &e0, &e1);
  hipExtLaunchKernelGGL(k_sw_wave, dim3((unsigned)a.n_task), dim3(64), lds, g_stream, e0, e1, 0, b);
  FQ_HIP(hipGetLastError());
  return 0;
}
__global__ void __launch_bounds__(64) k_sw_thread(FqSwArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < a.n_task) fq_sw_thread(a, t);
}
int launch_sw_serial(const FqSwArgs &a) {
  FQ_PRE();
  if (a.n_task <= 0) return 0;
  hipEvent_t e0, e1;
  kernel_events(FQ_K_SW_KERNEL, &e0, &e1);
  hipExtLaunchKernelGGL(k_sw_thread, dim3(nblk((uint64_t)a.n_task, 64)), dim3(64), 0, g_stream, e0, e1, 0, a);
  FQ_HIP(hipGetLastError());
  return 0;
}
int launch_refine(const FqRefineArgs &a) {
  FQ_PRE();
  if (a.n_task <= 0) return 0;
  // one task per wavefront while row arrays + sequences + trace matrix fit in LDS with several blocks per CU; longer reads
  // fall back to one task per lane
  const size_t wave_lds = (size_t)3 * (a.RL + 1) * 4 + ((a.RL + 16) & ~15) + ((a.QL + 16) & ~15) + ((a.RL + a.QL + 16) & ~15) + (size_t)((a.RL + 2) >> 1) * (a.QL + 1) + 16;   // (trace: two cells per byte)
  const bool no_wave = g_cur->tune.refine_lanes != 0;   // test knob: force the lane-per-task kernels
  hipEvent_t e0, e1;
  kernel_events(FQ_K_REFINE_KERNEL, &e0, &e1);
  if (wave_lds <= 64 * 1024 && !no_wave) {
    hipExtLaunchKernelGGL(k_refine_wave, dim3((unsigned)a.n_task), dim3(64), wave_lds, g_stream, e0, e1, 0, a);
    FQ_HIP(hipGetLastError());
    return 0;
  }
  const size_t lds = (size_t)3 * (a.RL + 1) * 64 * 4;
  if (lds <= kLdsBudget) {
    hipExtLaunchKernelGGL(k_refine_lds, dim3(nblk((uint64_t)a.n_task, 64)), dim3(64), lds, g_stream, e0, e1, 0, a);
  } else {
    hipExtLaunchKernelGGL(k_refine, dim3(nblk((uint64_t)a.n_task, 64)), dim3(64), 0, g_stream, e0, e1, 0, a);
  }
  FQ_HIP(hipGetLastError());
  return 0;
}

// ---- the record stages (fq_records.h): one thread per read, pair or task ------------------------------------------------------
#define FQ_REC_KERNEL(name, body)                                                    \
  __global__ void __launch_bounds__(256) name(FqRecArgs a, int n) {                  \
    const int i = blockIdx.x * blockDim.x + threadIdx.x;                             \
    if (i < n) body(a, i);                                                           \
  }
FQ_REC_KERNEL(k_rec_init, fq_rec_init_thread)
FQ_REC_KERNEL(k_rec_nocc, fq_rec_nocc_thread)
FQ_REC_KERNEL(k_enum_plan, fq_enum_plan_thread)
FQ_REC_KERNEL(k_enum_fill, fq_enum_fill_thread)
FQ_REC_KERNEL(k_main_hit, fq_main_hit_thread)
FQ_REC_KERNEL(k_compact_idx, fq_compact_thread)
FQ_REC_KERNEL(k_pair_rec, fq_pair_rec_thread)
FQ_REC_KERNEL(k_pair_gather, fq_pair_gather_thread)
FQ_REC_KERNEL(k_pair_scatter, fq_pair_scatter_thread)
FQ_REC_KERNEL(k_xa_count, fq_xa_count_thread)
FQ_REC_KERNEL(k_xa_fill, fq_xa_fill_thread)
FQ_REC_KERNEL(k_sw_plan, fq_sw_plan_thread)
FQ_REC_KERNEL(k_sw_fill, fq_sw_fill_thread)
FQ_REC_KERNEL(k_rec_gather, fq_rec_gather_thread)
FQ_REC_KERNEL(k_rec_scatter, fq_rec_scatter_thread)
FQ_REC_KERNEL(k_ref_count, fq_ref_count_thread)
FQ_REC_KERNEL(k_ref_fill, fq_ref_fill_thread)
FQ_REC_KERNEL(k_ref_apply, fq_ref_apply_thread)
FQ_REC_KERNEL(k_md_rec, fq_md_rec_thread)
FQ_REC_KERNEL(k_md_mask, fq_md_mask_piece)
FQ_REC_KERNEL(k_flat_count, fq_flat_count_thread)
FQ_REC_KERNEL(k_flat_fill, fq_flat_fill_thread)
int launch_rec(int op, const FqRecArgs &a, int64_t n) {
  FQ_PRE();
  if (n <= 0) return 0;
  if (n > 0x7fffffff) { g_err = "record stage: more than 2^31 items"; return -5; }
  typedef void (*Kern)(FqRecArgs, int);
  static const Kern kerns[FQ_ROP_COUNT] = {k_rec_init, k_rec_nocc, k_enum_plan, k_enum_fill, k_main_hit, k_compact_idx, k_pair_rec, k_pair_gather, k_pair_scatter,
                                           k_xa_count, k_xa_fill, k_sw_plan, k_sw_fill, k_rec_gather, k_rec_scatter, k_ref_count, k_ref_fill, k_ref_apply,
                                           k_md_rec, k_md_mask, k_flat_count, k_flat_fill};
  if (op < 0 || op >= FQ_ROP_COUNT) { g_err = "record stage: unknown operation"; return -1; }
  hipEvent_t e0, e1;
  kernel_events(op == FQ_ROP_MD || op == FQ_ROP_MD_MASK ? FQ_K_MD_KERNEL : FQ_K_REC_KERNEL, &e0, &e1);
  hipExtLaunchKernelGGL(kerns[op], dim3(nblk((uint64_t)n, 256)), dim3(256), 0, g_stream, e0, e1, 0, a, (int)n);
  FQ_HIP(hipGetLastError());
  return 0;
}
// ---- the consumers on the device (fq_emit.h) ----
__global__ void __launch_bounds__(256) k_sam_len(FqSamArgs a, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) fq_sam_len_thread(a, i);
}
__global__ void __launch_bounds__(256) k_sam_fill(FqSamArgs a, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) fq_sam_fill_thread(a, i);
}
__global__ void __launch_bounds__(256) k_sam_body(FqSamArgs a, int n, int pieces) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int idx, c;
  if (t < 0x100000000ull) { idx = (int)((uint32_t)t / (uint32_t)pieces); c = (int)((uint32_t)t - (uint32_t)idx * (uint32_t)pieces); }      // (a 64-bit division is a subroutine)
  else { idx = (int)(t / (uint64_t)pieces); c = (int)(t % (uint64_t)pieces); }
  if (idx < n) fq_sam_body_piece(a, idx, c);
}
int launch_sam(int op, const FqSamArgs &a, int64_t n) {
  FQ_PRE();
  if (n <= 0) return 0;
  if (n > 0x7fffffff) { g_err = "SAM text: more than 2^31 records"; return -5; }
  hipEvent_t e0, e1;
  kernel_events(FQ_K_EMIT, &e0, &e1);
  if (op == FQ_EOP_SAM_LEN) hipExtLaunchKernelGGL(k_sam_len, dim3(nblk((uint64_t)n, 256)), dim3(256), 0, g_stream, e0, e1, 0, a, (int)n);
  else if (op == FQ_EOP_SAM_FILL) hipExtLaunchKernelGGL(k_sam_fill, dim3(nblk((uint64_t)n, 256)), dim3(256), 0, g_stream, e0, e1, 0, a, (int)n);
  else if (op == FQ_EOP_SAM_BODY) {      // a thread per sixteen bytes of a record's SEQ / tab / QUAL run (2 x row stride + 1 bytes at most)
    const int pieces = (2 * a.stride + 1 + FQ_SAM_PIECE - 1) / FQ_SAM_PIECE;
    hipExtLaunchKernelGGL(k_sam_body, dim3(nblk((uint64_t)n * (uint64_t)pieces, 256)), dim3(256), 0, g_stream, e0, e1, 0, a, (int)n, pieces);
  }
  else { g_err = "SAM text: unknown operation"; return -1; }
  FQ_HIP(hipGetLastError());
  return 0;
}
__global__ void __launch_bounds__(256) k_bam_len(FqBamArgs a, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) fq_bam_len_thread(a, i);
}
__global__ void __launch_bounds__(256) k_bam_fill(FqBamArgs a, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) fq_bam_fill_thread(a, i);
}
__global__ void __launch_bounds__(256) k_bam_body(FqBamArgs a, int n, int pieces) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int idx, c;
  if (t < 0x100000000ull) { idx = (int)((uint32_t)t / (uint32_t)pieces); c = (int)((uint32_t)t - (uint32_t)idx * (uint32_t)pieces); }
  else { idx = (int)(t / (uint64_t)pieces); c = (int)(t % (uint64_t)pieces); }
  if (idx < n) fq_bam_body_piece(a, idx, c);
}
int launch_bam(int op, const FqBamArgs &a, int64_t n) {
  FQ_PRE();
  if (n <= 0) return 0;
  if (n > 0x7fffffff) { g_err = "BAM records: more than 2^31 records"; return -5; }
  hipEvent_t e0, e1;
  kernel_events(FQ_K_EMIT, &e0, &e1);
  if (op == FQ_EOP_BAM_LEN) hipExtLaunchKernelGGL(k_bam_len, dim3(nblk((uint64_t)n, 256)), dim3(256), 0, g_stream, e0, e1, 0, a, (int)n);
  else if (op == FQ_EOP_BAM_FILL) hipExtLaunchKernelGGL(k_bam_fill, dim3(nblk((uint64_t)n, 256)), dim3(256), 0, g_stream, e0, e1, 0, a, (int)n);
  else if (op == FQ_EOP_BAM_BODY) {      // a thread per sixteen bytes of a record's packed bases and qualities (1.5 x row stride + 1 bytes at most)
    const int pieces = ((a.s.stride + 1) / 2 + a.s.stride + FQ_SAM_PIECE - 1) / FQ_SAM_PIECE;
    hipExtLaunchKernelGGL(k_bam_body, dim3(nblk((uint64_t)n * (uint64_t)pieces, 256)), dim3(256), 0, g_stream, e0, e1, 0, a, (int)n, pieces);
  }
  else { g_err = "BAM records: unknown operation"; return -1; }
  FQ_HIP(hipGetLastError());
  return 0;
}
// one wavefront per BGZF member (fq_deflate.h)
__global__ void __launch_bounds__(256) k_bgzf_deflate(FqDeflateArgs a) {
  __shared__ FqdLds lds[4];
  const uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (b >= a.n_blocks) return;
  const uint32_t bs = fqd_member(a, b, lds[threadIdx.x >> 6]);
  if ((threadIdx.x & 63) == 0) a.bsize[b] = bs;
}
__global__ void __launch_bounds__(256) k_bgzf_pack(FqDeflatePackArgs a, uint64_t n) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) fqd_pack_piece(a, t);
}
int launch_deflate(const FqDeflateArgs &a) {
  FQ_PRE();
  if (!a.n_blocks) return 0;
  hipEvent_t e0, e1;
  kernel_events(FQ_K_EMIT, &e0, &e1);
  hipExtLaunchKernelGGL(k_bgzf_deflate, dim3(nblk(a.n_blocks, 4)), dim3(256), 0, g_stream, e0, e1, 0, a);
  FQ_HIP(hipGetLastError());
  return 0;
}
int launch_deflate_pack(const FqDeflatePackArgs &a) {
  FQ_PRE();
  if (!a.n_blocks) return 0;
  const uint64_t n = (uint64_t)a.n_blocks * (FQD_SLOT / 16);
  hipLaunchKernelGGL(k_bgzf_pack, dim3(nblk(n, 256)), dim3(256), 0, g_stream, a, n);
  FQ_HIP(hipGetLastError());
  return 0;
}
__global__ void __launch_bounds__(256) k_qc_pair(FqQcArgs a, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) fq_qc_pair_thread(a, i);
}
__global__ void __launch_bounds__(256) k_qc_ist_fill(FqQcArgs a, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) fq_qc_ist_fill_thread(a, i);
}
__global__ void __launch_bounds__(256) k_qc_pile_fill(FqQcArgs a, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) fq_qc_pile_fill_thread(a, i);
}
// a wavefront per record, resident workgroups walking the records: 64 consecutive positions of a read are one atomic instruction on the depth
// tables; the quality / cycle histograms are kept per workgroup in LDS and reach the consumer's once, at the end
// (L lanes per record: what a record costs is the latency of its chain of dependent loads -- its contig, the contig's regions, the CIGAR, the bases --, and a wavefront
//  that walks 64 / L records side by side pays it once for all of them; the per-base work is a few instructions either way)
template <int L>
__global__ void __launch_bounds__(256) k_qc_base(FqQcArgs a, int n) {
  __shared__ uint32_t hist[4 * 256];
  for (int b = threadIdx.x; b < 4 * 256; b += 256) hist[b] = 0;
  __syncthreads();
  const int lane = threadIdx.x & (L - 1), slot = (blockIdx.x * 256 + threadIdx.x) / L, n_slots = (gridDim.x * 256) / L;
  for (int idx = slot; idx < n; idx += n_slots) fq_qc_base_record(a, idx, lane, L, hist);
  __syncthreads();
  for (int b = threadIdx.x; b < 4 * 256; b += 256) if (hist[b]) atomicAdd((unsigned long long *)&a.hist[b], (unsigned long long)hist[b]);
}
__global__ void __launch_bounds__(256) k_dup_rehash(const uint64_t *old, uint64_t n, uint64_t *tab, uint64_t mask) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) fq_dupset_rehash_thread(old, tab, mask, (int64_t)i);
}
int launch_qc(int op, const FqQcArgs &a, int64_t n) {
  FQ_PRE();
  if (n <= 0) return 0;
  if (n > 0x7fffffff) { g_err = "statistics: more than 2^31 records"; return -5; }
  hipEvent_t e0, e1;
  kernel_events(FQ_K_EMIT, &e0, &e1);
  if (op == FQ_QOP_PAIR) hipExtLaunchKernelGGL(k_qc_pair, dim3(nblk((uint64_t)n, 256)), dim3(256), 0, g_stream, e0, e1, 0, a, (int)n);
  else if (op == FQ_QOP_IST_FILL) hipExtLaunchKernelGGL(k_qc_ist_fill, dim3(nblk((uint64_t)n, 256)), dim3(256), 0, g_stream, e0, e1, 0, a, (int)n);
  else if (op == FQ_QOP_PILE_FILL) hipExtLaunchKernelGGL(k_qc_pile_fill, dim3(nblk((uint64_t)n, 256)), dim3(256), 0, g_stream, e0, e1, 0, a, (int)n);
  else if (op == FQ_QOP_BASE) {
    static const int lanes = [] { const char *e = getenv("FASTQUICK_QC_LANES"); const int v = e ? atoi(e) : 16; return v == 64 || v == 32 || v == 8 ? v : 16; }();      // (A/B)
    const dim3 grid(std::min<unsigned>(nblk((uint64_t)n * (uint64_t)lanes, 256), 256u * 8u));
    if (lanes == 64) hipExtLaunchKernelGGL(k_qc_base<64>, grid, dim3(256), 0, g_stream, e0, e1, 0, a, (int)n);
    else if (lanes == 32) hipExtLaunchKernelGGL(k_qc_base<32>, grid, dim3(256), 0, g_stream, e0, e1, 0, a, (int)n);
    else if (lanes == 8) hipExtLaunchKernelGGL(k_qc_base<8>, grid, dim3(256), 0, g_stream, e0, e1, 0, a, (int)n);
    else hipExtLaunchKernelGGL(k_qc_base<16>, grid, dim3(256), 0, g_stream, e0, e1, 0, a, (int)n);
  }
  else { g_err = "statistics: unknown operation"; return -1; }
  FQ_HIP(hipGetLastError());
  return 0;
}
int launch_dup_rehash(const uint64_t *old, uint64_t old_cap, uint64_t *tab, uint64_t mask) {
  FQ_PRE();
  if (!old_cap) return 0;
  hipLaunchKernelGGL(k_dup_rehash, dim3(nblk(old_cap, 256)), dim3(256), 0, g_stream, old, old_cap, tab, mask);
  FQ_HIP(hipGetLastError());
  return 0;
}
__global__ void __launch_bounds__(256) k_aln_index(const int32_t *work, const uint32_t *status, const uint64_t *off, const uint32_t *naln, uint64_t base, uint64_t *aoff, uint32_t *an, int n) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < n) fq_aln_index_thread(work, status, off, naln, base, aoff, an, w);
}
int launch_aln_index(const int32_t *work, const uint32_t *status, const uint64_t *off, const uint32_t *naln, uint64_t base, uint64_t *aoff, uint32_t *an, int n) {
  FQ_PRE();
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_aln_index, dim3(nblk((uint64_t)n, 256)), dim3(256), 0, g_stream, work, status, off, naln, base, aoff, an, n);
  FQ_HIP(hipGetLastError());
  return 0;
}

// dynamic-LDS limits of the kernels that ask for more than the default 64 KB: once per device (state_create)
static int set_kernel_attributes() {
  FQ_HIP(hipFuncSetAttribute((const void *)k_sw_wave, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget));
  FQ_HIP(hipFuncSetAttribute((const void *)k_refine_wave, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  FQ_HIP(hipFuncSetAttribute((const void *)k_refine_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget));
  return 0;
}

}  // namespace fqdev
