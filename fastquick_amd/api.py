"""ctypes binding of the C ABI in include/fastquick_amd.h (plumbing for tests and bench.py).

The product library is fastquick_amd/libfastquick_amd.so (HIP, gfx950).  There is no CPU fallback:
if the library is missing or no HIP device is usable, loading / fq_index_load raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.environ.get("FQ_LIB_EXPERIMENT") or os.path.join(HERE, "libfastquick_amd.so")   # (FQ_LIB_EXPERIMENT: an instrumented build of the same library)

FQ_K_NAMES = ("prep", "width", "gap", "sa", "sw", "refine", "prep_kernel", "gap_kernel")


class Opts(C.Structure):
    _fields_ = [
        ("s_mm", C.c_int32), ("s_gapo", C.c_int32), ("s_gape", C.c_int32), ("mode", C.c_int32),
        ("indel_end_skip", C.c_int32), ("max_del_occ", C.c_int32), ("max_entries", C.c_int32),
        ("fnr", C.c_double), ("max_diff", C.c_int32), ("max_gapo", C.c_int32), ("max_gape", C.c_int32),
        ("max_seed_diff", C.c_int32), ("seed_len", C.c_int32), ("max_top2", C.c_int32), ("trim_qual", C.c_int32),
        ("filter_thresh", C.c_int32), ("max_isize", C.c_int32), ("force_isize", C.c_int32), ("max_occ", C.c_uint32),
        ("n_multi", C.c_int32), ("N_multi", C.c_int32), ("is_sw", C.c_int32), ("ap_prior", C.c_double),
        ("host_threads", C.c_int32), ("batch_pairs", C.c_int32), ("single_end", C.c_int32), ("pad_opts", C.c_int32),
    ]


class ReadBatch(C.Structure):
    _fields_ = [("n_pairs", C.c_int32), ("stride", C.c_int32), ("seq", C.c_void_p), ("qual", C.c_void_p),
                ("len", C.c_void_p), ("names", C.c_void_p), ("name_stride", C.c_int32), ("names_mate", C.c_void_p)]


class PackedBatch(C.Structure):
    _fields_ = [("n_pairs", C.c_int32), ("uniform_len", C.c_int32), ("head", C.c_void_p), ("body", C.c_void_p),
                ("body_stride", C.c_int32), ("qual_stride", C.c_int32), ("len", C.c_void_p), ("exc", C.c_void_p),
                ("n_exc", C.c_int64), ("qual", C.c_void_p), ("qual_last", C.c_void_p), ("names", C.c_void_p), ("name_stride", C.c_int32),
                ("names_mate", C.c_void_p), ("serial", C.c_uint64), ("single_end", C.c_int32), ("pad_packed", C.c_int32)]


class FastqRows(C.Structure):
    _fields_ = [("stride", C.c_int32), ("name_stride", C.c_int32), ("seq", C.c_void_p), ("qual", C.c_void_p), ("len", C.c_void_p), ("names", C.c_void_p)]


class QcOpts(C.Structure):
    _fields_ = [("flank_len", C.c_int32), ("flank_long_len", C.c_int32), ("read_len", C.c_int32), ("cal_dup", C.c_int32),
                ("genome_size", C.c_int64), ("genome_n_size", C.c_int64), ("mode", C.c_int32), ("pad", C.c_int32)]


class Multi(C.Structure):
    _fields_ = [("pos", C.c_uint32), ("cigar_off", C.c_uint32), ("n_cigar", C.c_uint16), ("gap", C.c_uint8),
                ("mm", C.c_uint8), ("strand", C.c_uint8), ("pad", C.c_uint8 * 3)]


class Result(C.Structure):
    _fields_ = [("pos", C.c_uint32), ("sa", C.c_uint32), ("c1", C.c_uint32), ("c2", C.c_uint32), ("score", C.c_int32),
                ("len", C.c_int32), ("full_len", C.c_int32), ("clip_len", C.c_int32),
                ("type", C.c_uint8), ("strand", C.c_uint8), ("filtered", C.c_uint8), ("extra_flag", C.c_uint8),
                ("n_mm", C.c_uint8), ("n_gapo", C.c_uint8), ("n_gape", C.c_uint8), ("mapQ", C.c_uint8),
                ("seQ", C.c_uint8), ("revived", C.c_uint8), ("nm", C.c_uint16), ("n_cigar", C.c_uint16),
                ("n_multi", C.c_uint16), ("cigar_off", C.c_uint32), ("md_off", C.c_uint32), ("multi_off", C.c_uint32)]


class Isize(C.Structure):
    _fields_ = [("avg", C.c_double), ("std", C.c_double), ("ap_prior", C.c_double), ("low", C.c_uint32),
                ("high", C.c_uint32), ("high_bayesian", C.c_uint32), ("pad", C.c_uint32)]


class ResultBatch(C.Structure):
    _fields_ = [("n_pairs", C.c_int32), ("n_survivors", C.c_int32), ("n_both_filtered", C.c_int32),
                ("n_both_unmapped", C.c_int32), ("pair_idx", C.POINTER(C.c_int32)), ("rec", C.POINTER(Result)),
                ("cigar", C.POINTER(C.c_uint16)), ("md", C.c_void_p), ("multi", C.POINTER(Multi)),
                ("isize", Isize), ("n_bases", C.c_int64), ("n_sub", C.c_int32), ("isize_sub", C.POINTER(Isize))]


class Stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double * 16), ("kernel_launches", C.c_uint64 * 16),
                ("occ_block_touches", C.c_uint64), ("gap_occ_touches", C.c_uint64), ("gap_nogap_touches", C.c_uint64), ("filter_probes", C.c_uint64),
                ("stack_pops", C.c_uint64), ("stack_pushes", C.c_uint64), ("sa_rows", C.c_uint64),
                ("reads_searched", C.c_uint64), ("pairs", C.c_uint64), ("sw_tasks", C.c_uint64),
                ("refine_tasks", C.c_uint64), ("tier_retries", C.c_uint64),
                ("max_pops_per_read", C.c_uint64), ("reads_over_4k_pops", C.c_uint64), ("max_wave_trips", C.c_uint64),
                ("host_ms_serial", C.c_double), ("host_ms_pair", C.c_double), ("host_ms_total", C.c_double),
                ("wall_ms_total", C.c_double), ("wave_trips", C.c_uint64), ("lane_trips", C.c_uint64),
                ("h2d_bytes", C.c_uint64), ("d2h_bytes", C.c_uint64), ("pairs_on_device", C.c_uint64), ("dbg", C.c_uint64 * 16),
                ("width_occ_touches", C.c_uint64), ("md_reads", C.c_uint64), ("host_pairs", C.c_uint64),
                ("device_wait_ms", C.c_double), ("host_cpu_ms", C.c_double)]


class FrontEndStats(C.Structure):
    _fields_ = [("ms_inflate", C.c_double), ("ms_tokenise", C.c_double), ("members", C.c_int64), ("refused", C.c_int64),
                ("text_bytes", C.c_int64), ("comp_bytes", C.c_int64), ("pairs", C.c_int64),
                ("ms_lines", C.c_double), ("ms_records", C.c_double), ("ms_slots", C.c_double), ("inflate_launches", C.c_int64), ("chunks", C.c_int64),
                ("ms_wait_reader", C.c_double), ("ms_wait_slot", C.c_double), ("ms_read", C.c_double), ("ms_upload", C.c_double)]


EXPORTS = ["fq_default_opts", "fq_index_build", "fq_index_load", "fq_index_destroy", "fq_index_l_pac",
           "fq_index_n_contigs", "fq_index_contig", "fq_ctx_create", "fq_ctx_destroy", "fq_ctx_last_error",
           "fq_ctx_set_debug", "fq_align_batch", "fq_batch_upload", "fq_align_resident", "fq_sam_header",
           "fq_sam_format_last", "fq_stage_dump_last", "fq_stats_get", "fq_stats_reset", "fq_version", "fq_host_cpus", "fq_runtime_configure", "fq_device_count", "fq_index_bitmap_fetch",
           "fq_pinned_alloc", "fq_pinned_free", "fq_pack_reads", "fq_packed_free", "fq_packed_create", "fq_pack_reads_into", "fq_pack_single_reads_into", "fq_packed_prefetch", "fq_packed_cancel", "fq_align_packed", "fq_stream_run",
           "fq_ctx_set_tuning", "fq_ctx_set_serial_hooks", "fq_ctx_mark_stream_broken", "fq_ctx_state_export", "fq_ctx_state_import", "fq_ctx_state_move", "fq_qc_default_opts", "fq_qc_create", "fq_qc_destroy", "fq_qc_last_error", "fq_qc_begin_file",
           "fq_qc_add_last", "fq_qc_end_file", "fq_qc_write", "fq_qc_state_reset", "fq_qc_state_export", "fq_qc_merge", "fq_bam_create", "fq_bam_add_last", "fq_bam_format_last", "fq_bam_write_records", "fq_bam_close",
           "fq_fastq_open", "fq_fastq_configure", "fq_fastq_set_sampling", "fq_fastq_read", "fq_fastq_last_error", "fq_fastq_dropped_record", "fq_fastq_unequal_lengths", "fq_fastq_is_bgzf", "fq_fastq_close", "fq_inflate_raw", "fq_crc32", "fq_inflate_device", "fq_bgzf_inflate_device",
           "fq_frontend_open", "fq_frontend_next", "fq_frontend_release", "fq_frontend_handover", "fq_frontend_unequal_lengths", "fq_frontend_stats", "fq_frontend_last_error", "fq_frontend_close",
           "fq_text_batch_pairs", "fq_text_batch_first_name", "fq_align_text", "fq_text_batch_fetch",
           "fq_ctx_set_emit", "fq_sam_device_last", "fq_sam_device_bytes", "fq_ctx_attach_qc", "fq_ctx_attach_bam", "fq_bgzf_deflate_device"]
SINK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)
EMIT_SAM = 1

SERIAL_HOOK = C.CFUNCTYPE(None, C.c_void_p)
STREAM_CALL = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p)      # (user, stream, call, fq_result_batch_t *)
_libs = {}


class FastquickError(RuntimeError):
    pass


def load_library(path: str | None = None):
    path = path or DEFAULT_LIB
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise FastquickError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(there is no CPU fallback)" % path)
    L = C.CDLL(path)
    L.fq_default_opts.argtypes = [C.POINTER(Opts)]
    L.fq_index_build.argtypes = [C.c_char_p, C.c_int]
    L.fq_index_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
    L.fq_index_destroy.argtypes = [C.c_void_p]
    L.fq_index_bitmap_fetch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.fq_index_l_pac.restype = C.c_int64
    L.fq_index_l_pac.argtypes = [C.c_void_p]
    L.fq_index_n_contigs.argtypes = [C.c_void_p]
    L.fq_ctx_create.argtypes = [C.c_void_p, C.POINTER(Opts), C.c_int32, C.POINTER(C.c_void_p)]
    L.fq_ctx_destroy.argtypes = [C.c_void_p]
    L.fq_ctx_last_error.restype = C.c_char_p
    L.fq_ctx_last_error.argtypes = [C.c_void_p]
    L.fq_ctx_set_debug.argtypes = [C.c_void_p, C.c_int]
    L.fq_align_batch.argtypes = [C.c_void_p, C.POINTER(ReadBatch), C.POINTER(ResultBatch)]
    L.fq_batch_upload.argtypes = [C.c_void_p, C.POINTER(ReadBatch)]
    L.fq_align_resident.argtypes = [C.c_void_p, C.POINTER(ResultBatch)]
    L.fq_sam_header.restype = C.c_int64
    L.fq_sam_header.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.fq_sam_format_last.restype = C.c_int64
    L.fq_sam_format_last.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.fq_ctx_set_emit.argtypes = [C.c_void_p, C.c_int32]
    L.fq_sam_device_last.restype = C.c_int64
    L.fq_sam_device_last.argtypes = [C.c_void_p, SINK_FN, C.c_void_p]
    L.fq_sam_device_bytes.restype = C.c_int64
    L.fq_sam_device_bytes.argtypes = [C.c_void_p]
    L.fq_ctx_attach_qc.argtypes = [C.c_void_p, C.c_void_p]
    L.fq_ctx_attach_bam.argtypes = [C.c_void_p, C.c_void_p]
    L.fq_stage_dump_last.restype = C.c_int64
    L.fq_stage_dump_last.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.fq_stats_get.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.fq_stats_reset.argtypes = [C.c_void_p]
    L.fq_version.restype = C.c_char_p
    L.fq_host_cpus.restype = C.c_int
    L.fq_pinned_alloc.restype = C.c_void_p
    L.fq_pinned_alloc.argtypes = [C.c_size_t]
    L.fq_pinned_free.argtypes = [C.c_void_p]
    L.fq_pack_reads.argtypes = [C.POINTER(ReadBatch), C.c_int, C.POINTER(C.POINTER(PackedBatch))]
    L.fq_packed_free.argtypes = [C.POINTER(PackedBatch)]
    L.fq_packed_create.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.POINTER(PackedBatch))]
    L.fq_pack_reads_into.argtypes = [C.POINTER(ReadBatch), C.c_int, C.POINTER(PackedBatch)]
    L.fq_pack_single_reads_into.argtypes = [C.POINTER(ReadBatch), C.c_int, C.POINTER(PackedBatch)]
    L.fq_packed_cancel.argtypes = [C.c_void_p, C.POINTER(PackedBatch)]
    L.fq_packed_prefetch.argtypes = [C.c_void_p, C.POINTER(PackedBatch)]
    L.fq_align_packed.argtypes = [C.c_void_p, C.POINTER(PackedBatch), C.POINTER(ResultBatch)]
    L.fq_stream_run.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.POINTER(C.POINTER(PackedBatch))), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.c_int32,
                                STREAM_CALL, C.c_void_p, C.POINTER(C.c_int64)]
    L.fq_fastq_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
    L.fq_fastq_configure.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64]
    L.fq_fastq_set_sampling.argtypes = [C.c_void_p, C.c_double]
    L.fq_fastq_read.restype = C.c_int64
    L.fq_fastq_read.argtypes = [C.c_void_p, C.c_int64, C.POINTER(FastqRows)]
    L.fq_fastq_last_error.restype = C.c_char_p
    L.fq_fastq_last_error.argtypes = [C.c_void_p]
    L.fq_fastq_dropped_record.restype = C.c_char_p
    L.fq_fastq_dropped_record.argtypes = [C.c_void_p]
    L.fq_fastq_is_bgzf.argtypes = [C.c_void_p]
    L.fq_inflate_raw.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.fq_inflate_raw.restype = C.c_int
    L.fq_crc32.argtypes = [C.c_void_p, C.c_size_t]
    L.fq_crc32.restype = C.c_uint32
    L.fq_inflate_device.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                    C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_double)]
    L.fq_frontend_open.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    L.fq_frontend_next.restype = C.c_int64
    L.fq_frontend_next.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.fq_frontend_release.restype = None
    L.fq_frontend_release.argtypes = [C.c_void_p, C.c_void_p]
    L.fq_frontend_handover.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.fq_frontend_unequal_lengths.argtypes = [C.c_void_p]
    L.fq_frontend_stats.restype = None
    L.fq_frontend_stats.argtypes = [C.c_void_p, C.POINTER(FrontEndStats)]
    L.fq_frontend_last_error.restype = C.c_char_p
    L.fq_frontend_last_error.argtypes = [C.c_void_p]
    L.fq_frontend_close.restype = None
    L.fq_frontend_close.argtypes = [C.c_void_p]
    L.fq_text_batch_pairs.argtypes = [C.c_void_p]
    L.fq_text_batch_first_name.restype = C.c_char_p
    L.fq_text_batch_first_name.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.fq_align_text.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(ResultBatch)]
    L.fq_text_batch_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.fq_bgzf_inflate_device.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_double)]
    L.fq_fastq_close.argtypes = [C.c_void_p]
    L.fq_ctx_set_tuning.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.fq_ctx_set_serial_hooks.argtypes = [C.c_void_p, SERIAL_HOOK, SERIAL_HOOK, C.c_void_p]
    L.fq_ctx_mark_stream_broken.argtypes = [C.c_void_p]
    L.fq_ctx_state_export.restype = C.c_int64
    L.fq_ctx_state_export.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.fq_ctx_state_import.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.fq_ctx_state_move.argtypes = [C.c_void_p, C.c_void_p]
    L.fq_qc_default_opts.argtypes = [C.POINTER(QcOpts)]
    L.fq_qc_create.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(QcOpts), C.POINTER(C.c_void_p)]
    L.fq_qc_destroy.argtypes = [C.c_void_p]
    L.fq_qc_last_error.restype = C.c_char_p
    L.fq_qc_last_error.argtypes = [C.c_void_p]
    L.fq_qc_begin_file.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.fq_qc_add_last.argtypes = [C.c_void_p, C.c_void_p]
    L.fq_qc_end_file.argtypes = [C.c_void_p]
    L.fq_qc_write.argtypes = [C.c_void_p]
    L.fq_bam_create.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(QcOpts), C.POINTER(C.c_void_p)]
    L.fq_bam_add_last.argtypes = [C.c_void_p, C.c_void_p]
    L.fq_bam_close.argtypes = [C.c_void_p]
    _libs[path] = L
    return L


def default_opts(lib=None, **kw) -> Opts:
    L = lib or load_library()
    o = Opts()
    L.fq_default_opts(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def build_index(fasta_path: str, write_rollhash: bool = False, lib=None) -> None:
    L = lib or load_library()
    rc = L.fq_index_build(fasta_path.encode(), int(write_rollhash))
    if rc:
        raise FastquickError("fq_index_build(%s) failed: %d" % (fasta_path, rc))


class Index:
    def __init__(self, prefix: str, device: int = 0, lib=None):
        self.L = lib or load_library()
        h = C.c_void_p()
        rc = self.L.fq_index_load(prefix.encode(), device, C.byref(h))
        if rc:
            raise FastquickError("fq_index_load(%s) failed: %d (no usable HIP device?)" % (prefix, rc))
        self.h = h
        self.prefix = prefix

    @property
    def l_pac(self) -> int:
        return int(self.L.fq_index_l_pac(self.h))

    def bitmap_bits(self, table: int) -> np.ndarray:
        """(tests) the set bits of filter table `table`, ascending"""
        buf = np.empty(1 << 29, dtype=np.uint8)
        rc = self.L.fq_index_bitmap_fetch(self.h, table, buf.ctypes.data)
        if rc:
            raise FastquickError("fq_index_bitmap_fetch failed: %d" % rc)
        nz = np.flatnonzero(buf)
        bits = np.unpackbits(buf[nz].reshape(-1, 1), axis=1, bitorder="little")
        r, c = np.nonzero(bits)
        return (nz[r].astype(np.uint64) * 8 + c.astype(np.uint64)).astype(np.uint32)

    def sam_header(self) -> bytes:
        n = self.L.fq_sam_header(self.h, None, 0)
        buf = C.create_string_buffer(n + 1)
        self.L.fq_sam_header(self.h, buf, n + 1)
        return buf.raw[:n]

    def close(self):
        if self.h:
            self.L.fq_index_destroy(self.h)
            self.h = None


class PairNames(list):
    """Read names of the first mates; `.mate` holds the second mates' names when they differ (else None)."""
    mate = None

    def __getitem__(self, k):
        out = list.__getitem__(self, k)
        if isinstance(k, slice):
            out = PairNames(out)
            out.mate = self.mate[k] if self.mate is not None else None
        return out


def pack_names(names, stride: int = 64) -> np.ndarray:
    buf = np.zeros((len(names), stride), dtype=np.uint8)
    for i, nm in enumerate(names):
        nm = nm[:stride - 1]
        buf[i, :len(nm)] = np.frombuffer(nm, dtype=np.uint8)
    return buf


class HostPacked:
    """A packed batch (fq_packed_batch_t) in pinned host memory, made by fq_pack_reads from ASCII rows.  Its qualities and
    names alias the numpy arrays it was packed from, which this object keeps alive."""

    def __init__(self, seq, qual, lens, names=None, threads: int = 0, lib=None):
        self.L = lib or load_library()
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        mate = getattr(names, "mate", None)
        nm = pack_names(names) if names is not None else None
        nm2 = pack_names(mate) if mate is not None else None
        b = ReadBatch(seq.shape[1], seq.shape[2], seq.ctypes.data, qual.ctypes.data, lens.ctypes.data,
                      nm.ctypes.data if nm is not None else None, 64, nm2.ctypes.data if nm2 is not None else None)
        self._keep = (seq, qual, lens, nm, nm2, b)
        self.single_end = seq.shape[0] == 1          # rows [1][n][stride]: the reads of one file (BwtMapper::SingleEndMapper)
        self.p = C.POINTER(PackedBatch)()
        if self.single_end:
            rc = self.L.fq_packed_create(0, 0, C.byref(self.p)) or self.L.fq_pack_single_reads_into(C.byref(b), threads, self.p)
        else:
            rc = self.L.fq_pack_reads(C.byref(b), threads, C.byref(self.p))
        if rc:
            raise FastquickError("fq_pack_reads failed: %d" % rc)
        self.n_pairs = seq.shape[1]

    def repack(self, threads: int = 0) -> None:
        """Packs the same rows again into the batch's storage (fq_pack_reads_into): what a front end does with every new chunk."""
        fn = self.L.fq_pack_single_reads_into if self.single_end else self.L.fq_pack_reads_into
        rc = fn(C.byref(self._keep[5]), threads, self.p)
        if rc:
            raise FastquickError("fq_pack_reads_into failed: %d" % rc)

    @property
    def h2d_head_bytes(self) -> int:
        return 48 * self.n_pairs

    def free(self):
        if self.p:
            self.L.fq_packed_free(self.p)
            self.p = None


class FastqFile:
    """One FASTQ file through the library's front end (fq_fastq_*): read(max_reads) -> (seq, qual, lens, names) numpy rows."""
    SLOTS_REUSED, SLOTS_CLEAN_NAMES, SLOTS_FRESH = 0, 1, 2

    def __init__(self, path: str, threads: int = 0, batch_pairs: int = 262144, slot_mode: int = 1, block_bytes: int = 0,
                 stride: int = 160, name_stride: int = 64, lib=None, frac: float = 1.0):
        self.L = lib or load_library()
        self.h = C.c_void_p()
        rc = self.L.fq_fastq_open(path.encode(), threads, C.byref(self.h))
        if rc:
            raise FastquickError("fq_fastq_open(%s) failed: %d" % (path, rc))
        rc = self.L.fq_fastq_configure(self.h, batch_pairs, slot_mode, block_bytes)
        if rc:
            raise FastquickError("fq_fastq_configure failed: %d" % rc)
        if frac != 1.0 and self.L.fq_fastq_set_sampling(self.h, frac):
            raise FastquickError("fq_fastq_set_sampling failed")
        self.stride, self.name_stride = stride, name_stride

    @property
    def is_bgzf(self) -> bool:
        return bool(self.L.fq_fastq_is_bgzf(self.h))

    def read_into(self, seq, qual, lens, names) -> int:
        """rows of preallocated arrays [cap][stride] / [cap] / [cap][name_stride]; returns the number of records"""
        rows = FastqRows(seq.shape[1], names.shape[1], seq.ctypes.data, qual.ctypes.data, lens.ctypes.data, names.ctypes.data)
        n = self.L.fq_fastq_read(self.h, seq.shape[0], C.byref(rows))
        if n < 0:
            raise FastquickError("fq_fastq_read failed: %d (%s)" % (n, self.L.fq_fastq_last_error(self.h).decode()))
        return int(n)

    def read(self, max_reads: int):
        seq = np.empty((max_reads, self.stride), dtype=np.uint8)
        qual = np.empty((max_reads, self.stride), dtype=np.uint8)
        lens = np.empty(max_reads, dtype=np.int32)
        names = np.empty((max_reads, self.name_stride), dtype=np.uint8)
        n = self.read_into(seq, qual, lens, names)
        return seq[:n], qual[:n], lens[:n], names[:n]

    def dropped_record(self):
        p = self.L.fq_fastq_dropped_record(self.h)
        return p.decode() if p else None

    def close(self):
        if self.h:
            self.L.fq_fastq_close(self.h)
            self.h = None


class Aligner:
    """One alignment context == one FASTQ pair stream of the reference (drand48 / last_ii / cache carry over)."""

    def __init__(self, index: Index, opts: Opts | None = None, max_pairs: int = 262144, debug: bool = False, tuning: dict | None = None, emit: int | None = None):
        """emit: FQ_EMIT_* flags -- the consumers on the device (None: FASTQUICK_API_EMIT, which the test suite sets, so that every sam_text() of
        every parity test also formats the text on the device and holds it to the host formatter's bytes)"""
        self.L = index.L
        self.index = index
        self.opts = opts or default_opts(self.L)
        h = C.c_void_p()
        rc = self.L.fq_ctx_create(index.h, C.byref(self.opts), max_pairs, C.byref(h))
        if rc:
            raise FastquickError("fq_ctx_create failed: %d" % rc)
        self.h = h
        if debug:
            self.L.fq_ctx_set_debug(self.h, 1)
        for k, v in (tuning or {}).items():
            if self.L.fq_ctx_set_tuning(self.h, k.encode(), int(v)):
                raise FastquickError("fq_ctx_set_tuning: unknown key %r" % k)
        self._keep = None
        self._keep_packed = None
        self.result = ResultBatch()
        self.emit = int(os.environ.get("FASTQUICK_API_EMIT", "0") or 0) if emit is None else int(emit)
        if self.emit and self.L.fq_ctx_set_emit(self.h, self.emit):
            raise FastquickError("fq_ctx_set_emit(%d) refused" % self.emit)

    def _batch(self, seq, qual, lens, names):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        mate = getattr(names, "mate", None)     # PairNames: the second mates' names where the two files disagree
        nm = pack_names(names) if names is not None else None
        nm2 = pack_names(mate) if mate is not None else None
        b = ReadBatch(seq.shape[1], seq.shape[2], seq.ctypes.data, qual.ctypes.data, lens.ctypes.data,
                      nm.ctypes.data if nm is not None else None, 64, nm2.ctypes.data if nm2 is not None else None)
        self._keep = (seq, qual, lens, nm, nm2, b)   # the library reads these until the next call
        return b

    def _check(self, rc, what):
        err = getattr(self, "_hook_error", None)
        if err is not None:
            self._hook_error = None
            raise FastquickError("%s: a serial hook raised %r" % (what, err)) from err
        if rc:
            raise FastquickError("%s failed: %d (%s)" % (what, rc, self.L.fq_ctx_last_error(self.h).decode()))

    def align(self, seq, qual, lens, names=None) -> ResultBatch:
        b = self._batch(seq, qual, lens, names)
        self._check(self.L.fq_align_batch(self.h, C.byref(b), C.byref(self.result)), "fq_align_batch")
        return self.result

    def upload(self, seq, qual, lens, names=None) -> None:
        b = self._batch(seq, qual, lens, names)
        self._check(self.L.fq_batch_upload(self.h, C.byref(b)), "fq_batch_upload")

    def export_state(self) -> bytes:
        n = self.L.fq_ctx_state_export(self.h, None, 0)
        buf = C.create_string_buffer(int(n))
        self.L.fq_ctx_state_export(self.h, buf, n)
        return buf.raw[:n]

    def import_state(self, blob: bytes) -> None:
        self._check(self.L.fq_ctx_state_import(self.h, blob, len(blob)), "fq_ctx_state_import")

    def take_state_of(self, other: "Aligner") -> None:
        """fq_ctx_state_move: this context goes on with the stream `other` has aligned so far (one process, nothing serialised)."""
        self._check(self.L.fq_ctx_state_move(self.h, other.h), "fq_ctx_state_move")

    def set_serial_hooks(self, before, after) -> None:
        """before() / after(): Python callables run around the order-dependent part of every call (None clears)."""
        # an exception inside a hook cannot cross the C frames above it: it is kept and raised when the call has returned
        self._hook_error = None

        def guard(fn):
            def run(_u):
                try:
                    fn()
                except BaseException as e:      # noqa: BLE001 -- re-raised by _check
                    if self._hook_error is None:
                        self._hook_error = e
                    # tell the library: the call must not go on from a state this hook failed to bring in, and whatever is
                    # exported from now on is marked broken for the ranks behind
                    self.L.fq_ctx_mark_stream_broken(self.h)
            return run
        self._hooks = (SERIAL_HOOK(guard(before)) if before else SERIAL_HOOK(0), SERIAL_HOOK(guard(after)) if after else SERIAL_HOOK(0))
        self._check(self.L.fq_ctx_set_serial_hooks(self.h, self._hooks[0], self._hooks[1], None), "fq_ctx_set_serial_hooks")

    def prefetch(self, packed: HostPacked) -> None:
        self._check(self.L.fq_packed_prefetch(self.h, packed.p), "fq_packed_prefetch")

    def align_packed(self, packed: HostPacked) -> ResultBatch:
        self._keep_packed = packed
        self._check(self.L.fq_align_packed(self.h, packed.p, C.byref(self.result)), "fq_align_packed")
        return self.result

    def align_text(self, batch) -> ResultBatch:
        """a batch of the device front end (DeviceFrontEnd.next): nothing of the input crosses PCIe"""
        self._check(self.L.fq_align_text(self.h, batch, C.byref(self.result)), "fq_align_text")
        return self.result

    def align_resident(self) -> ResultBatch:
        self._check(self.L.fq_align_resident(self.h, C.byref(self.result)), "fq_align_resident")
        return self.result

    def _text(self, fn) -> bytes:
        n = fn(self.h, None, 0)
        if n < 0:
            raise FastquickError("formatter failed: %d" % n)
        buf = C.create_string_buffer(n + 1)
        fn(self.h, buf, n + 1)
        return buf.raw[:n]

    def sam_text(self) -> bytes:
        host = self._text(self.L.fq_sam_format_last)
        if self.emit & EMIT_SAM:       # the same text formatted by the kernels of fq_emit.h: every caller checks both
            dev = self.sam_text_device()
            if dev != host:
                at = next((i for i, (a, b) in enumerate(zip(dev, host)) if a != b), min(len(dev), len(host)))
                raise FastquickError("SAM text formatted on the device differs from the host formatter's at byte %d of %d / %d: %r vs %r"
                                     % (at, len(dev), len(host), dev[max(0, at - 60):at + 60], host[max(0, at - 60):at + 60]))
        return host

    def sam_text_device(self) -> bytes:
        """the SAM text of the last call as the device formatted it (fq_ctx_set_emit(FQ_EMIT_SAM)), streamed off in slices"""
        parts = []

        def sink(_user, data, n):
            parts.append(C.string_at(data, n))
            return 0
        n = self.L.fq_sam_device_last(self.h, SINK_FN(sink), None)
        if n < 0:
            raise FastquickError("fq_sam_device_last failed: %d (%s)" % (n, self.L.fq_ctx_last_error(self.h).decode()))
        out = b"".join(parts)
        assert len(out) == n == self.L.fq_sam_device_bytes(self.h)
        return out

    def stage_text(self) -> bytes:
        return self._text(self.L.fq_stage_dump_last)

    def stats(self) -> dict:
        s = Stats()
        self.L.fq_stats_get(self.h, C.byref(s))
        d = {}
        for name, _ in Stats._fields_:
            v = getattr(s, name)
            d[name] = list(v) if hasattr(v, "__len__") else v
        return d

    def reset_stats(self):
        self.L.fq_stats_reset(self.h)

    def close(self):
        if self.h:
            self.L.fq_ctx_destroy(self.h)
            self.h = None


class QC:
    """The QC consumer (StatCollector's side): feed it every batch of a stream, then write()."""

    def __init__(self, index: Index, ref_prefix: str, out_prefix: str, **kw):
        self.L = index.L
        o = QcOpts()
        self.L.fq_qc_default_opts(C.byref(o))
        for k, v in kw.items():
            setattr(o, k, v)
        h = C.c_void_p()
        rc = self.L.fq_qc_create(index.h, ref_prefix.encode(), out_prefix.encode(), C.byref(o), C.byref(h))
        if rc:
            raise FastquickError("fq_qc_create failed: %d" % rc)
        self.h = h

    def _ck(self, rc, what):
        if rc:
            raise FastquickError("%s failed: %d (%s)" % (what, rc, self.L.fq_qc_last_error(self.h).decode()))

    def begin_file(self, fq1: str, fq2: str):
        self._ck(self.L.fq_qc_begin_file(self.h, fq1.encode(), fq2.encode()), "fq_qc_begin_file")

    def attach(self, aligner: "Aligner"):
        """StatCollector's part of every later call of `aligner` runs on the device, inside the call (fq_ctx_attach_qc); add() then only appends."""
        self._ck(self.L.fq_ctx_attach_qc(aligner.h, self.h), "fq_ctx_attach_qc")

    def add(self, aligner: "Aligner"):
        self._ck(self.L.fq_qc_add_last(self.h, aligner.h), "fq_qc_add_last")

    def end_file(self):
        self._ck(self.L.fq_qc_end_file(self.h), "fq_qc_end_file")

    def write(self):
        self._ck(self.L.fq_qc_write(self.h), "fq_qc_write")

    def state_reset(self):
        """Makes this a shard consumer and starts a segment (fq_qc_state_reset)."""
        self._ck(self.L.fq_qc_state_reset(self.h), "fq_qc_state_reset")

    def state_export(self) -> bytes:
        self.L.fq_qc_state_export.restype = C.c_int64
        self.L.fq_qc_state_export.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        n = self.L.fq_qc_state_export(self.h, None, 0)
        if n < 0:
            self._ck(int(n), "fq_qc_state_export")
        buf = C.create_string_buffer(int(n))
        n2 = self.L.fq_qc_state_export(self.h, buf, n)
        if n2 != n:
            raise FastquickError("fq_qc_state_export: size changed between calls")
        return buf.raw[:n]

    def merge(self, blob: bytes):
        self.L.fq_qc_merge.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        self._ck(self.L.fq_qc_merge(self.h, blob, len(blob)), "fq_qc_merge")

    def close(self):
        if self.h:
            self.L.fq_qc_destroy(self.h)
            self.h = None


class BamWriter:
    """The BAM consumer (SetSamRecord / SetSamFileHeader): genome-coordinate records of every batch of a stream."""

    def __init__(self, index: Index, fai_path: str, bam_path: str, rg: str = "@RG\\tID:foo\\tSM:bar", **kw):
        self.L = index.L
        o = QcOpts()
        self.L.fq_qc_default_opts(C.byref(o))
        for k, v in kw.items():
            setattr(o, k, v)
        h = C.c_void_p()
        rc = self.L.fq_bam_create(index.h, fai_path.encode(), bam_path.encode(), rg.encode(), C.byref(o), C.byref(h))
        if rc:
            raise FastquickError("fq_bam_create failed: %d" % rc)
        self.h = h

    def attach(self, aligner: "Aligner"):
        """the records of every later call of `aligner` are formatted by the kernels of fq_emit.h inside the call (fq_ctx_attach_bam); add() fetches the bytes"""
        if self.L.fq_ctx_attach_bam(aligner.h, self.h):
            raise FastquickError("fq_ctx_attach_bam failed")

    def add(self, aligner: "Aligner"):
        rc = self.L.fq_bam_add_last(self.h, aligner.h)
        if rc:
            raise FastquickError("fq_bam_add_last failed: %d" % rc)

    def close(self):
        if self.h:
            rc = self.L.fq_bam_close(self.h)
            self.h = None
            if rc:
                raise FastquickError("fq_bam_close failed: %d" % rc)


def align_stream(aligner: Aligner, names, seq, qual, lens, batch: int, stages_path=None, sam_path=None, header=True, packed: bool = False, qc: "QC | None" = None,
                 bam: "BamWriter | None" = None) -> int:
    """Feed n pairs in batches of `batch` (mirrors PairEndMapper's loop); returns pairs with SAM records.  packed=True goes
    through the packed-batch boundary (fq_pack_reads -> fq_packed_prefetch of the next chunk -> fq_align_packed)."""
    n = seq.shape[1]
    st = open(stages_path, "wb") if stages_path else None
    sm = open(sam_path, "wb") if sam_path else None
    if sm and header:
        sm.write(aligner.index.sam_header())
    total = 0
    nxt = None
    for b0 in range(0, n, batch):
        b1 = min(n, b0 + batch)
        if packed:
            cur = nxt or HostPacked(seq[:, b0:b1], qual[:, b0:b1], lens[:, b0:b1], names[b0:b1], lib=aligner.L)
            nxt = None
            if b1 < n:
                b2 = min(n, b1 + batch)
                nxt = HostPacked(seq[:, b1:b2], qual[:, b1:b2], lens[:, b1:b2], names[b1:b2], lib=aligner.L)
                aligner.prefetch(nxt)
            res = aligner.align_packed(cur)
        else:
            res = aligner.align(seq[:, b0:b1], qual[:, b0:b1], lens[:, b0:b1], names[b0:b1])
        total += res.n_survivors - res.n_both_unmapped
        if st:
            st.write(aligner.stage_text())
        if sm:
            sm.write(aligner.sam_text())
        if qc is not None:
            qc.add(aligner)
        if bam is not None:
            bam.add(aligner)
        if packed:
            aligner._keep_packed = None
            cur.free()
    if st:
        st.close()
    if sm:
        sm.close()
    return total


def inflate_device(streams, device: int = 0, lib=None, repeats: int = 1):
    """The device front end's member decoder on raw DEFLATE streams: streams = [(comp, out_len, crc32)], one wavefront each.
    Returns ([(status, bytes)], kernel_ms)."""
    L = lib or load_library()
    n = len(streams)
    keep = [C.create_string_buffer(bytes(c), max(1, len(c))) for c, _, _ in streams]
    outs = [C.create_string_buffer(max(1, ol)) for _, ol, _ in streams]
    src = (C.c_void_p * max(1, n))(*[C.cast(k, C.c_void_p) for k in keep])
    dst = (C.c_void_p * max(1, n))(*[C.cast(o, C.c_void_p) for o in outs])
    lens = (C.c_size_t * max(1, n))(*[len(c) for c, _, _ in streams])
    olen = (C.c_uint32 * max(1, n))(*[ol for _, ol, _ in streams])
    crc = (C.c_uint32 * max(1, n))(*[cr & 0xffffffff for _, _, cr in streams])
    st = (C.c_uint32 * max(1, n))()
    ms = C.c_double(0)
    rc = L.fq_inflate_device(device, n, src, lens, dst, olen, crc, st, repeats, C.byref(ms))
    if rc:
        raise FastquickError("fq_inflate_device failed: %d" % rc)
    return [(int(st[k]), outs[k].raw[:streams[k][1]]) for k in range(n)], ms.value


def bgzf_deflate_device(data: bytes, device: int = 0, lib=None):
    """data as BGZF members written by the device's compressor (fq_deflate.h): (members, kernel_ms)"""
    L = lib or load_library()
    n = len(data)
    cap = (n // 0xd000 + 2) * 65536
    out = np.empty(cap, dtype=np.uint8)
    src = np.frombuffer(data, dtype=np.uint8) if n else np.zeros(1, np.uint8)
    ol, ms = C.c_int64(0), C.c_double(0)
    L.fq_bgzf_deflate_device.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    rc = L.fq_bgzf_deflate_device(device, src.ctypes.data, n, out.ctypes.data, cap, C.byref(ol), C.byref(ms))
    if rc:
        raise FastquickError("fq_bgzf_deflate_device failed: %d" % rc)
    return out[:ol.value].tobytes(), ms.value


def bgzf_inflate_device(blob: bytes, text_cap: int, device: int = 0, lib=None, repeats: int = 1):
    """A run of whole BGZF members inflated by the device: (text, statuses, kernel_ms)."""
    L = lib or load_library()
    out = np.empty(max(1, text_cap), dtype=np.uint8)
    nm, tl, ms = C.c_int64(0), C.c_int64(0), C.c_double(0)
    cap = len(blob) // 28 + 8
    st = (C.c_uint32 * cap)()
    buf = np.frombuffer(blob, dtype=np.uint8)
    rc = L.fq_bgzf_inflate_device(device, buf.ctypes.data, len(blob), out.ctypes.data, out.size, C.byref(nm), C.byref(tl), st, cap, repeats, C.byref(ms))
    if rc:
        raise FastquickError("fq_bgzf_inflate_device failed: %d" % rc)
    return out[:tl.value], [int(st[k]) for k in range(nm.value)], ms.value


FQ_EFALLBACK = -6


def stream_run(aligners, batches, n_calls, first=None, on_call=None, prefetch_beyond=False, lib=None):
    """fq_stream_run: stream s = aligners[s] over batches[s] (lists of HostPacked), n_calls calls each, inside the library (no Python thread per
    stream).  on_call(stream, call): called after each call on the stream's thread (the aligner's sam_text() etc. are valid in it).  Returns the
    surviving pairs per stream."""
    L = lib or aligners[0].L
    n = len(aligners)
    ctxs = (C.c_void_p * n)(*[al.h for al in aligners])
    rows = [(C.POINTER(PackedBatch) * len(bs))(*[b.p for b in bs]) for bs in batches]
    tab = (C.POINTER(C.POINTER(PackedBatch)) * n)(*[C.cast(r, C.POINTER(C.POINTER(PackedBatch))) for r in rows])
    nb = (C.c_int32 * n)(*[len(bs) for bs in batches])
    fst = (C.c_int32 * n)(*(first or [0] * n))
    surv = (C.c_int64 * n)()
    errs = []

    def cb(_user, stream, call, _res):
        try:
            on_call(stream, call)
            return 0
        except Exception as e:      # noqa: BLE001
            errs.append(e)
            return -1
    fn = STREAM_CALL(cb) if on_call else STREAM_CALL(0)
    rc = L.fq_stream_run(ctxs, n, tab, nb, fst, n_calls, 1 if prefetch_beyond else 0, fn, None, surv)
    if errs:
        raise errs[0]
    if rc:
        raise FastquickError("fq_stream_run failed: %d (%s)" % (rc, "; ".join(x for x in (L.fq_ctx_last_error(al.h).decode(errors="replace") for al in aligners) if x)))
    return list(surv)


class DeviceFrontEnd:
    """One FASTQ pair (or one single-end file) of BGZF files through the front end on the device (fq_frontend_*): next() -> (n_pairs, batch)
    with the reads resident in HBM; 0 at the end; FQ_EFALLBACK when the rest of the stream is the host readers' (handover())."""

    def __init__(self, fq1: str, fq2: str | None, batch_pairs: int = 262144, chunk_pairs: int = 16 * 262144, slot_mode: int = 0, max_read_len: int = 160,
                 device: int = 0, lib=None):
        self.L = lib or load_library()
        self.h = C.c_void_p()
        rc = self.L.fq_frontend_open(device, fq1.encode(), fq2.encode() if fq2 else None, batch_pairs, chunk_pairs, slot_mode, max_read_len, C.byref(self.h))
        if rc:
            raise FastquickError("fq_frontend_open failed: %d" % rc)
        self.batch_pairs, self.slot_mode = batch_pairs, slot_mode

    def next(self):
        b = C.c_void_p()
        n = self.L.fq_frontend_next(self.h, C.byref(b))
        if n < 0 and n != FQ_EFALLBACK:
            raise FastquickError("fq_frontend_next failed: %d (%s)" % (n, self.L.fq_frontend_last_error(self.h).decode(errors="replace")))
        return int(n), b

    def release(self, b) -> None:
        self.L.fq_frontend_release(self.h, b)

    def first_name(self, b, sub_batch: int, end: int) -> bytes:
        return self.L.fq_text_batch_first_name(b, sub_batch, end)

    def fetch(self, b, n_pairs: int, single_end: bool = False):
        """(head[3][rows] uint64, len[rows] uint16, names[rows][stride] uint8) of a batch, copied to the host (tests)"""
        rows = n_pairs * (1 if single_end else 2)
        head = np.zeros((3, rows), dtype=np.uint64)
        lens = np.zeros(rows, dtype=np.uint16)
        names = np.zeros((rows, 304), dtype=np.uint8)
        ns = self.L.fq_text_batch_fetch(self.h, b, head.ctypes.data, lens.ctypes.data, names.ctypes.data, names.size)
        if ns < 0:
            raise FastquickError("fq_text_batch_fetch failed: %d" % ns)
        return head, lens, names.reshape(-1)[:rows * ns].reshape(rows, ns)

    def handover(self, threads: int = 2, stride: int = 160, name_stride: int = 304):
        """the host readers (FastqFile) standing where the device's part of the stream ended"""
        hs = (C.c_void_p * 2)()
        rc = self.L.fq_frontend_handover(self.h, threads, hs)
        if rc:
            raise FastquickError("fq_frontend_handover failed: %d" % rc)
        out = []
        for h in hs:
            if h:
                f = FastqFile.__new__(FastqFile)
                f.L, f.h, f.stride, f.name_stride = self.L, C.c_void_p(h), stride, name_stride
                out.append(f)
        return out

    def stats(self) -> dict:
        s = FrontEndStats()
        self.L.fq_frontend_stats(self.h, C.byref(s))
        return {k: getattr(s, k) for k, _ in FrontEndStats._fields_}

    def unequal_lengths(self) -> bool:
        return bool(self.L.fq_frontend_unequal_lengths(self.h))

    def close(self):
        if self.h:
            self.L.fq_frontend_close(self.h)
            self.h = None
