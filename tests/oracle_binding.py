"""ctypes binding of the CPU parity checker (oracle/libfq_oracle.so).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libfq_oracle.so")
REF_DRIVER = os.path.join(ORACLE_DIR, "_ref", "fq_ref_driver")


class Opts(C.Structure):
    _fields_ = [
        ("s_mm", C.c_int), ("s_gapo", C.c_int), ("s_gape", C.c_int), ("mode", C.c_int),
        ("indel_end_skip", C.c_int), ("max_del_occ", C.c_int), ("max_entries", C.c_int),
        ("fnr", C.c_double), ("max_diff", C.c_int), ("max_gapo", C.c_int), ("max_gape", C.c_int),
        ("max_seed_diff", C.c_int), ("seed_len", C.c_int), ("max_top2", C.c_int), ("trim_qual", C.c_int),
        ("filter_thresh", C.c_int), ("max_isize", C.c_int), ("force_isize", C.c_int), ("max_occ", C.c_uint32),
        ("n_multi", C.c_int), ("N_multi", C.c_int), ("is_sw", C.c_int), ("ap_prior", C.c_double),
    ]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("occ_block_touches", "filter_probes", "stack_pops", "sa_calls", "sa_steps", "reads_aligned", "pairs", "occ_gap_touches")]


_lib = None


def build_oracle() -> None:
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(ORACLE_DIR, "fq_oracle.c")):
            build_oracle()
        L = C.CDLL(LIB_PATH)
        L.fqo_index_load.restype = C.c_void_p
        L.fqo_index_load.argtypes = [C.c_char_p]
        L.fqo_index_free.argtypes = [C.c_void_p]
        L.fqo_index_lpac.restype = C.c_int64
        L.fqo_index_lpac.argtypes = [C.c_void_p]
        L.fqo_ctx_create.restype = C.c_void_p
        L.fqo_ctx_create.argtypes = [C.c_void_p, C.POINTER(Opts)]
        L.fqo_ctx_free.argtypes = [C.c_void_p]
        L.fqo_default_opts.argtypes = [C.POINTER(Opts)]
        L.fqo_align_batch.restype = C.c_int
        L.fqo_align_batch.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_void_p, C.c_void_p]
        L.fqo_print_sam_header.argtypes = [C.c_void_p, C.c_void_p]
        L.fqo_get_counters.argtypes = [C.c_void_p, C.POINTER(Counters)]
        L.fqo_drand48_selftest.restype = C.c_double
        L.fqo_drand48_selftest.argtypes = [C.c_int, C.POINTER(C.c_uint64)]
        L.fqo_cal_maxdiff.restype = C.c_int
        L.fqo_cal_maxdiff.argtypes = [C.c_int, C.c_double, C.c_double]
        L.fqo_occ.restype = C.c_uint32
        L.fqo_occ.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_int]
        L.fqo_sa.restype = C.c_uint32
        L.fqo_sa.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
        L.fqo_bitmap_popcount.restype = C.c_uint64
        L.fqo_bitmap_popcount.argtypes = [C.c_void_p, C.c_int]
        L.fqo_bitmap_fnv.restype = C.c_uint64
        L.fqo_bitmap_fnv.argtypes = [C.c_void_p, C.c_int]
        L.fqo_bitmaps_from_fasta.restype = C.c_int
        L.fqo_bitmaps_from_fasta.argtypes = [C.c_char_p, C.POINTER(C.c_uint64 * 6), C.POINTER(C.c_uint64 * 6)]
        L.fqo_global_align.restype = C.c_int
        L.fqo_global_align.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.POINTER(C.c_int)]
        _libc = C.CDLL(None)
        _libc.fopen.restype = C.c_void_p
        _libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
        _libc.fclose.argtypes = [C.c_void_p]
        L._libc = _libc
        _lib = L
    return _lib


def default_opts(**kw) -> Opts:
    o = Opts()
    lib().fqo_default_opts(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def read_fastq_pair(path1: str, path2: str):
    """Tiny FASTQ parser (4-line records) -> (names, seq[2,n,L], qual[2,n,L], lens[2,n])."""
    recs = []
    for path in (path1, path2):
        with open(path, "rb") as fh:
            lines = fh.read().split(b"\n")
        n = len(lines) // 4
        recs.append([(lines[4 * i][1:].split()[0], lines[4 * i + 1], lines[4 * i + 3]) for i in range(n)])
    n = len(recs[0])
    assert len(recs[1]) == n
    L = max(max(len(r[1]) for r in recs[0]), max(len(r[1]) for r in recs[1]))
    seq = np.zeros((2, n, L), dtype=np.uint8)
    qual = np.zeros((2, n, L), dtype=np.uint8)
    lens = np.zeros((2, n), dtype=np.int32)
    for e in range(2):
        for i, (_, s, q) in enumerate(recs[e]):
            seq[e, i, :len(s)] = np.frombuffer(s, dtype=np.uint8)
            qual[e, i, :len(q)] = np.frombuffer(q, dtype=np.uint8)
            lens[e, i] = len(s)
    from fastquick_amd.api import PairNames
    names = PairNames(r[0] for r in recs[0])
    mate = [r[0] for r in recs[1]]
    if mate != list(names):     # the reference prints every record under its own read's name
        names.mate = mate
    return names, seq, qual, lens


def apply_slot_history(seq, lens, batch: int) -> None:
    """What `FASTQuick_amd align` (fq_cli.cpp, ReadSlots) does to rows: the reference reuses two sets of `batch` read slots, and the
    filter sees, behind a read shorter than 96 bp, the bases earlier reads of its slot left.  In place."""
    n, width = seq.shape[1], min(96, seq.shape[2])
    for e in range(2):
        hist = [np.zeros((batch, width), dtype=np.uint8) for _ in range(2)]
        for g in range(n):
            h = hist[(g // batch) & 1][g % batch]
            ln = int(lens[e, g])
            seq[e, g, ln:] = 0
            if ln < width:
                seq[e, g, ln:width] = h[ln:width]
            h[:min(ln, width)] = seq[e, g, :min(ln, width)]


def pack_names(names, stride: int = 64) -> bytes:
    buf = bytearray(stride * len(names))
    for i, nm in enumerate(names):
        nm = nm[:stride - 1]
        buf[i * stride:i * stride + len(nm)] = nm
    return bytes(buf)


class OracleAligner:
    """Stateful oracle context (drand48 stream, last_ii, (k,l) cache persist across batches)."""

    def __init__(self, prefix: str, opts: Opts | None = None, share: "OracleAligner | None" = None):
        self.L = lib()
        self.owns_index = share is None
        self.ix = share.ix if share is not None else self.L.fqo_index_load(prefix.encode())   # the index is read-only: contexts may share it
        if not self.ix:
            raise RuntimeError("oracle: cannot load index %s" % prefix)
        self.opts = opts or default_opts()
        self.ctx = self.L.fqo_ctx_create(self.ix, C.byref(self.opts))

    def set_threads(self, n_threads: int) -> None:
        """--t of the reference: stage A on n_threads workers sliced as src/BwtMapper.cpp:1490-1513 (results unchanged)."""
        self.L.fqo_ctx_set_threads.argtypes = [C.c_void_p, C.c_int]
        self.L.fqo_ctx_set_threads.restype = None
        self.L.fqo_ctx_set_threads(self.ctx, int(n_threads))

    def set_rng(self, state: int) -> None:
        """The drand48 state of the stream, as fq_ctx_state_import sets it on the library's side."""
        self.L.fqo_ctx_set_rng.argtypes = [C.c_void_p, C.c_uint64]
        self.L.fqo_ctx_set_rng.restype = None
        self.L.fqo_ctx_set_rng(self.ctx, int(state))

    def close(self):
        if self.ctx:
            self.L.fqo_ctx_free(self.ctx)
            if self.owns_index:
                self.L.fqo_index_free(self.ix)
            self.ctx = None

    def counters(self) -> dict:
        c = Counters()
        self.L.fqo_get_counters(self.ctx, C.byref(c))
        return {n: int(getattr(c, n)) for n, _ in Counters._fields_}

    def align(self, names, seq, qual, lens, stages_path=None, sam_path=None, batch=None, header=True) -> int:
        """Run n pairs through the oracle in batches of `batch` pairs (default: one batch)."""
        n = seq.shape[1]
        batch = batch or n
        libc = self.L._libc
        st = libc.fopen(stages_path.encode(), b"w") if stages_path else None
        sm = libc.fopen(sam_path.encode(), b"w") if sam_path else None
        if sm and header:
            self.L.fqo_print_sam_header(self.ix, sm)
        total = 0
        stride = seq.shape[2]
        for b0 in range(0, n, batch):
            b1 = min(n, b0 + batch)
            s = np.ascontiguousarray(seq[:, b0:b1])
            q = np.ascontiguousarray(qual[:, b0:b1])
            ln = np.ascontiguousarray(lens[:, b0:b1])
            nm = pack_names(names[b0:b1])
            mate = getattr(names, "mate", None)
            nm2 = pack_names(mate[b0:b1]) if mate is not None else None
            rc = self.L.fqo_align_batch(self.ctx, b1 - b0, nm, nm2, 64, s.ctypes.data, q.ctypes.data, ln.ctypes.data, stride, st, sm)
            if rc < 0:
                raise RuntimeError("oracle align failed")
            total += rc
        if st:
            libc.fclose(st)
        if sm:
            libc.fclose(sm)
        return total


    def align_se(self, names, seq, qual, lens, stages_path=None, sam_path=None, batch=None, header=True) -> int:
        """Single-end reads (BwtMapper::SingleEndMapper): seq / qual [n][stride], lens [n]."""
        n = seq.shape[0]
        batch = batch or n
        libc = self.L._libc
        st = libc.fopen(stages_path.encode(), b"w") if stages_path else None
        sm = libc.fopen(sam_path.encode(), b"w") if sam_path else None
        if sm and header:
            self.L.fqo_print_sam_header(self.ix, sm)
        self.L.fqo_align_batch_se.restype = C.c_int
        self.L.fqo_align_batch_se.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        total = 0
        stride = seq.shape[1]
        for b0 in range(0, n, batch):
            b1 = min(n, b0 + batch)
            s = np.ascontiguousarray(seq[b0:b1])
            q = np.ascontiguousarray(qual[b0:b1])
            ln = np.ascontiguousarray(lens[b0:b1])
            rc = self.L.fqo_align_batch_se(self.ctx, b1 - b0, pack_names(names[b0:b1]), 64, s.ctypes.data, q.ctypes.data, ln.ctypes.data, stride, st, sm)
            if rc < 0:
                raise RuntimeError("oracle align failed")
            total += rc
        if st:
            libc.fclose(st)
        if sm:
            libc.fclose(sm)
        return total


def run_reference(prefix: str, fq1: str, fq2: str, out_prefix: str, *extra) -> None:
    """Run the real reference (only possible where oracle/_ref was built, i.e. the build container)."""
    subprocess.check_call([REF_DRIVER, "align", prefix, fq1, fq2, out_prefix, *map(str, extra)],
                          stderr=subprocess.DEVNULL)


def normalise_stage_line(line: str) -> str:
    """Drop fields that are undefined in the reference (uninitialised ap_prior of a failed isize inference,
    per-batch B counter which is process-global in the oracle)."""
    if line.startswith("I ") and "avg=bff0000000000000" in line:
        parts = line.split()
        parts = [p for p in parts if not p.startswith("ap=")]
        return " ".join(parts)
    if line.startswith("B "):
        return "B"
    return line


def diff_stage_files(a: str, b: str, max_report: int = 10):
    """Differences between two stage dumps, including missing / extra lines per tag (a truncated dump must not pass on its
    common prefix).  The only line one producer writes and the others do not is the reference driver's closing `E` summary."""
    out = []
    with open(a) as fa, open(b) as fb:
        la = [normalise_stage_line(x.rstrip("\n")) for x in fa if not x.startswith("E ")]
        lb = [normalise_stage_line(x.rstrip("\n")) for x in fb if not x.startswith("E ")]
    if len(la) != len(lb):
        import collections
        ca, cb = collections.Counter(x[:1] for x in la), collections.Counter(x[:1] for x in lb)
        out.append("lines per tag differ: %s vs %s" % (sorted(ca.items()), sorted(cb.items())))
    for i, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            out.append("line %d:\n  A: %s\n  B: %s" % (i + 1, x[:600], y[:600]))
            if len(out) >= max_report:
                break
    return out
