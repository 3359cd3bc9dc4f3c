// fq_backend.h -- the narrow device API the host pipeline (fq_align.cpp) drives.
// Implemented for gfx950 in fq_device.hip (the product).  tests/emu/ carries a host-loop
// implementation of the same interface for the CPU-only test tier; it is never part of the product.
#pragma once
#include "fq_kernels.h"
#include "fq_records.h"
#include "fq_frontend.h"
#include "fq_emit.h"
#include "fq_deflate.h"

namespace fqdev {

// Device state of one alignment context (or of an index load): compute stream, copy stream, timing events, scan / compaction
// temporaries.  Created and destroyed with its owner; a host thread binds it on entry to the library and every call below acts
// on the bound state.  Nothing device-side is kept per thread or per process.
struct State;
struct Tune {   // experiment / test knobs (fq_ctx_set_tuning); the defaults are what DESIGN.md measures
  int gap_coop_waves = 0;   // wavefronts of the wavefront-per-read search kernel (0: 1,024)
  int gap_waves_per_cu = 0, gap_refill_min = 0, gap_order_asc = 0, filter_no_turns = 0, refine_lanes = 0;
  int spin_sync = 0;        // 1: sync() spins (hipStreamSynchronize) instead of sleeping on a blocking event
  int gap_generic_opts = 0; // 1: never the kernels specialised for FASTQuick's own option block (FqOptsStock)
  int width_both_strands = 0; // 1: the width kernel with one thread per read walking both strands (round 3's); default: one thread per (read, strand), strands apart by XCD
  int prep_priority = 0;     // 1: the packed filter kernel on a stream of the highest priority
  int sw_serial_reverse = 0; // 1: the mate-rescue kernel's reverse pass as the serial statement on one lane (the wavefront form is the default)
};
int runtime_configure(int hw_queues, int blocking_waits);   // fq_runtime_configure: before the process's first HIP call
int device_count();                        // devices the process sees (0: none)
State *state_create(int device_ordinal);   // nullptr on failure (last_error())
void state_destroy(State *s);              // synchronises the state's streams, frees everything it owns
int bind(State *s);                        // 0 or FQ_ENODEV-style negative
Tune *tune(State *s);
const char *last_error();
bool is_real_gpu();                    // true for the HIP backend
// Device-filling stages of different contexts take turns: a per-device lock held by the context whose search stage owns the
// device.  (Two such launches side by side share wave slots and L2 and both last longer than they would one after the other; what
// the other contexts' host threads do meanwhile -- their host phases, their small kernels -- is not held up.)
void device_turn_begin(int slots = 1);   // at most `slots` contexts hold a turn at a time
void device_turn_end();

void *dmalloc(size_t bytes);           // device memory (nullptr on failure)
void dfree(void *p);
void *hmalloc(size_t bytes);           // pinned host memory
void hfree(void *p);
int h2d(void *dst, const void *src, size_t bytes);
int d2h(void *dst, const void *src, size_t bytes);
// host side pinned (hmalloc): small sizes are copied by a kernel on the compute stream, large ones by hipMemcpyAsync
int copy_pinned(void *dst, const void *src, size_t bytes, int to_device);
int d2d(void *dst, const void *src, size_t bytes);   // device to device, on the compute stream
int dzero(void *dst, size_t bytes);
int dfill(void *dst, int byte, size_t bytes);
int sync();
// input prefetch on the state's copy stream (runs under the compute stream's kernels): copies, then copy_record(slot);
// compute_wait_copy(slot) makes everything enqueued afterwards on the compute stream wait for that slot's copies
int h2d_copy(void *dst, const void *src, size_t bytes);
int copy_record(int slot);
int copy_wait(int slot);                 // host waits for the copies recorded for `slot`
void copy_discard();                     // forget copy_pinned copies that were queued and not yet launched (a call that ends early)
int compute_wait_copy(int slot);
int copy_flush_now();                    // launches the small copy_pinned copies that are still queued (a call that ends without a sync)

// HIP-event timing of everything enqueued between begin/end, accumulated per kernel id
void time_begin(int kid);
void time_end(int kid);
void time_collect(double ms[], uint64_t launches[], int n_ids);   // after sync(); resets the pending list

// launchers (asynchronous on the backend's stream unless stated)
int launch_prep(const FqPrepArgs &a);
// ordered stream compaction with wavefront ballot + prefix sums:
//   pair_list[0..n_surv)   indices of pairs with at least one unfiltered mate, ascending
//   read_list[0..n_search) unfiltered reads, ordered by (pair, end); sidx[r] = s or -1
//   counts[0] = n_search, counts[1] = n_surv   (device memory)
int launch_compact(const uint8_t *filtered, int n_pairs, int32_t *read_list, int32_t *sidx, int32_t *pair_list, int32_t *counts);
// out[2*sp+e] = {len_trim, filtered, sidx} of read e of surviving pair sp
int launch_surv_gather(const int32_t *pair_list, int n_surv, int n_pairs, const int32_t *len_trim, const uint8_t *filtered, const int32_t *sidx, FqSurvInfo *out);
// packed input (fq_packed_batch_t): filter over the uploaded k-mers; survivors' rows unpacked to ASCII afterwards
int launch_prep_packed(const FqPrepPackedArgs &a);
int launch_surv_map(const int32_t *pair_list, int n_surv, int n_pairs, const uint8_t *filtered, const int32_t *sidx, FqSurvInfo *out,
                    int32_t *row_map, int32_t *read_list_c, int32_t *crow_of);
int launch_unpack(const FqUnpackArgs &a);
int launch_patch(const FqPatchArgs &a);
int launch_trim(const FqTrimArgs &a);
int launch_trim_all(const FqTrimAllArgs &a);
int launch_width(const FqWidthArgs &a);
// order[0..n) = the work items sorted by descending fq_order_key (any order inside a key); cnt: FQ_ORDER_KEYS*2 + 1 words of scratch,
// cnt[2 * FQ_ORDER_KEYS] = length of the first block of the queue (the keys of the upper half) on return
int launch_order(const uint8_t *bid_end, int n, int n_hard, int32_t *order, uint32_t *cnt);   // n_hard: fq_order_key
// number of persistent lanes launch_gap() will start for these arguments (sizes a.pool / a.heads)
int gap_lane_slots(const FqGapArgs &a);
int launch_gap(const FqGapArgs &a);
// out[0..n] = exclusive prefix sums of in[0..n) (64-bit); out[n] = total
int launch_scan(const uint32_t *in, uint64_t *out, uint32_t n);
// packed[off[w] + j] = aln[w*cap + j]
int launch_pack_aln(const FqAln *aln, const uint32_t *n_aln, const uint64_t *off, uint32_t cap, uint32_t n_work, FqAln *packed);
int launch_sa(const FqSaArgs &a);
// A second stream of the context for work that runs beside the main stream's (the second search round of one part of a large call
// under the first round of the next part): stream_aux(1) sends the following backend calls to it, stream_aux(0) back to the main
// stream; stream_fork(): the aux stream waits for what the main stream holds so far; stream_join(): the reverse.
int stream_aux(int on);
int stream_fork();
int stream_join();
int stream_mark(int k);        // a mark (0..3) on the current stream (main, or aux under stream_aux(1)) behind what it holds so far
int stream_wait_mark(int k);   // the current stream waits for that mark's work
// the work items of one queue segment whose search did not complete (status != 0): search indices appended to out[*count ...]
int launch_collect(const int32_t *order, const uint32_t *split, int n_work, int seg, int n_seg, const uint32_t *status, const int32_t *work, int32_t *out, uint32_t *count);
// the stages over the device-resident records (fq_records.h): operation FQ_ROP_*, one thread per item
int launch_rec(int op, const FqRecArgs &a, int64_t n);
// by search index: aoff[work[w]] = base + off[w], an[work[w]] = naln[w] for the work items of a launch that completed (status 0)
int launch_aln_index(const int32_t *work, const uint32_t *status, const uint64_t *off, const uint32_t *naln, uint64_t base, uint64_t *aoff, uint32_t *an, int n);
// the consumers on the device (fq_emit.h): SAM text -- operation FQ_EOP_SAM_*, one thread per record
int launch_sam(int op, const FqSamArgs &a, int64_t n);
int launch_bam(int op, const FqBamArgs &a, int64_t n);
// BGZF members of a byte stream (fq_deflate.h): a wavefront per block of FQD_BLOCK bytes into its staging slot; then the members packed behind each other
int launch_deflate(const FqDeflateArgs &a);
int launch_deflate_pack(const FqDeflatePackArgs &a);     // BAM records -- FQ_EOP_BAM_*, one thread per record
// ... StatCollector's part -- operation FQ_QOP_*: a thread per pair (PAIR, IST_FILL) or per record (PILE_FILL); BASE: a wavefront per record,
// the quality / cycle histograms of a workgroup in LDS
int launch_qc(int op, const FqQcArgs &a, int64_t n);
int launch_dup_rehash(const uint64_t *old, uint64_t old_cap, uint64_t *tab, uint64_t mask);   // every key of `old` into `tab` (filled with FQ_QC_DUP_EMPTY)
int launch_sw(const FqSwArgs &a);          // one task per wavefront (window + query in LDS)
int launch_sw_serial(const FqSwArgs &a);   // one task per lane out of the task's global scratch: any window size
int launch_refine(const FqRefineArgs &a);

// ---- FASTQ front end (fq_frontend.h) ----
// the CRC tables and powers the member decoder reads, resident on the bound state's device (made once per device)
const FqzCrcConst *crc_const();
int launch_inflate(const FqInflateArgs &a);     // one wavefront per BGZF member
int launch_inflate2(const FqInflateArgs &a, const FqInflateArgs &b);   // two member tables (the two files of a pair) in one launch
// positions of the line ends of text[0, n): nl[0 .. min(count, cap)) ascending, *count (device memory) = how many there are
int launch_nl_index(const uint8_t *text, uint32_t n, uint32_t lo, uint32_t *nl, uint32_t cap, uint32_t *count);
int launch_tok_rec(const FqTokArgs &a);         // a thread per record
int launch_tok_pieces(const FqTokArgs &a);      // a thread per (record, 32 bases)
int launch_slot_bases(const FqSlotArgs &a);     // a thread per read slot
int launch_slot_names(const FqSlotArgs &a);     // (plain_names: a thread per record for the names, a thread per slot for what the slots keep)
int launch_text_gather(const FqTextGatherArgs &a);   // a thread per 16 bytes of a surviving read's row
int launch_text_trim_all(const FqTextTrimArgs &a);   // a thread per read
int dfill32(void *dst, uint32_t v, size_t n_words);  // (hipMemsetD32 on the state's stream)

}  // namespace fqdev
