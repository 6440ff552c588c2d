"""ctypes binding of libfzphase.so (include/fzphase.h).  No torch, no CPU fallback.

Records cross the boundary as numpy structured arrays whose dtypes mirror the C structs.
"""
from __future__ import annotations

import ctypes as C
import weakref
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FZP_LIB") or os.path.join(_HERE, "libfzphase.so")   # FZP_LIB: another build of the same library (experiments)

FZP_OK = 0
FZP_EINVAL, FZP_EZERODIV, FZP_ENOMEM, FZP_EUNSORTED, FZP_EDEVICE, FZP_ENODEVICE, FZP_EIO = -1, -2, -3, -4, -5, -6, -7
STAGE_HET, STAGE_ASSOC, STAGE_BLOCKS, STAGE_READS, STAGE_ALL = 1, 2, 4, 8, 15

SITE = np.dtype([("pos", "<i4"), ("ref_base", "u1"), ("base", "u1", (4,)), ("pad_", "u1", (3,)), ("total", "<i4"),
                 ("count", "<i4", (4,)), ("row_off", "<i8")], align=True)
AROW = np.dtype([("site1", "<i4"), ("site2", "<i4"), ("n", "<i4", (4,))], align=True)
PVAR = np.dtype([("block", "<i4"), ("site", "<i4"), ("b1", "u1"), ("b2", "u1"), ("pad_", "u1", (2,)), ("lext", "<i4"),
                 ("rext", "<i4"), ("lscore", "<i4"), ("rscore", "<i4")], align=True)
PREAD = np.dtype([("q_id", "<i4"), ("block", "<i4"), ("phase", "<i4"), ("n0", "<i4"), ("n1", "<i4")], align=True)
R2P = np.dtype([("arid", "<i4"), ("ctg", "<i4"), ("block", "<i4"), ("phase", "<i4")], align=True)
ALN_SUMMARY = np.dtype([("aligned", "<i4"), ("strand", "<i4"), ("pos", "<i4"), ("ref_end", "<i4"), ("q_start", "<i4"),
                        ("q_end", "<i4"), ("score", "<i4"), ("n_cigar", "<i4"), ("cells", "<i8"), ("n_columns", "<i4"),
                        ("n_match", "<i4")], align=True)
assert SITE.itemsize == 40 and AROW.itemsize == 24 and PVAR.itemsize == 28 and PREAD.itemsize == 20
assert R2P.itemsize == 16 and ALN_SUMMARY.itemsize == 48


class FzpError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "libfzphase error %d: %s" % (code, msg))
        self.code = code


class AlnSetStruct(C.Structure):
    _fields_ = [("n_rec", C.c_int64), ("rec_qid", C.POINTER(C.c_int32)), ("rec_pos", C.POINTER(C.c_int32)),
                ("cig_off", C.POINTER(C.c_int64)), ("cigar", C.POINTER(C.c_uint32)), ("seq_off", C.POINTER(C.c_int64)),
                ("seq", C.POINTER(C.c_uint8)), ("n_qid", C.c_int32), ("qname_off", C.POINTER(C.c_int64)),
                ("qnames", C.POINTER(C.c_char)), ("last_pos", C.c_int32), ("max_ref_span", C.c_int32),
                ("n_columns", C.c_int64)]


class ResultStruct(C.Structure):
    _fields_ = [("n_sites", C.c_int64), ("sites", C.c_void_p), ("n_rows", C.c_int64), ("vmap_qid", C.c_void_p),
                ("n_arows", C.c_int64), ("arows", C.c_void_p), ("n_pvars", C.c_int64), ("pvars", C.c_void_p),
                ("n_preads", C.c_int64), ("preads", C.c_void_p)]


class ResultAllStruct(C.Structure):
    _fields_ = [("all", ResultStruct), ("site_begin", C.POINTER(C.c_int64)), ("row_begin", C.POINTER(C.c_int64)),
                ("arow_begin", C.POINTER(C.c_int64)), ("pvar_begin", C.POINTER(C.c_int64)), ("pread_begin", C.POINTER(C.c_int64))]


TIG = np.dtype([("ctg", "<i4"), ("block", "<i4"), ("phase", "<i4"), ("lo", "<i4"), ("hi", "<i4"), ("n_records", "<i4"),
                ("seq_off", "<i8"), ("seq_len", "<i8")], align=True)
assert TIG.itemsize == 40


class TigsStruct(C.Structure):
    _fields_ = [("n_tigs", C.c_int64), ("tigs", C.c_void_p), ("n_seq", C.c_int64), ("seq", C.c_void_p)]


class Tigs:
    """K6 results of a batch: one consensus sequence per (contig, block, phase) with a non-empty pile."""

    def __init__(self, ts: TigsStruct):
        self._ts = ts
        n = int(ts.n_tigs)
        self.tigs = _copy_in(ts.tigs, n, TIG) if n else np.zeros(0, TIG)
        self._seq = None      # (the bases stay in the library's block until somebody asks: a polished genome is hundreds of megabytes, copied into fresh pages at 4 GB/s)

    @property
    def seq(self):
        if self._seq is None:
            if self._ts is None:
                raise FzpError(-1, "Tigs: closed")
            self._seq = C.string_at(self._ts.seq, int(self._ts.n_seq)) if self._ts.n_seq else b""
        return self._seq

    def sequence(self, i):
        t = self.tigs[i]
        if self._seq is None and self._ts is not None:
            return C.string_at(self._ts.seq + int(t["seq_off"]), int(t["seq_len"])) if t["seq_len"] else b""
        return self.seq[int(t["seq_off"]):int(t["seq_off"] + t["seq_len"])]

    def fasta(self, ctg, ctg_id: str):
        return _fmt("fzp_format_tigs", C.byref(self._ts), C.c_int32(ctg), ctg_id.encode())

    def close(self):
        if self._ts is not None:
            load().fzp_tigs_free(C.byref(self._ts))
            self._ts = None

    __del__ = close


class OvlpParams(C.Structure):
    _fields_ = [("max_diff", C.c_int64), ("max_cov", C.c_int64), ("min_cov", C.c_int64), ("min_len", C.c_int64), ("bestn", C.c_int64)]


class AlignParams(C.Structure):
    _fields_ = [("kmer", C.c_int32), ("seed_stride", C.c_int32), ("match", C.c_int32), ("mismatch", C.c_int32),
                ("gap", C.c_int32), ("min_seed_hits", C.c_int32), ("min_pct_identity", C.c_int32), ("seed_anchored", C.c_int32), ("band", C.c_int32), ("reserved", C.c_int32 * 7)]


_lib = None


PIPE_CONSENSUS, PIPE_ASYNC_WRITES, PIPE_REBUILD_INDEX, PIPE_BAM, PIPE_SENTINELS = 1, 2, 4, 8, 16
TEXT_VARIANT_MAP, TEXT_ATABLE = 1, 2


class NamesStruct(C.Structure):
    _fields_ = [("n_ctg", C.c_int32), ("ctg_id", C.POINTER(C.c_char_p)), ("name_off", C.c_void_p), ("names", C.c_char_p)]


class PipeOpts(C.Structure):
    _fields_ = [("out_dir", C.c_char_p), ("rawread_ids", C.c_char_p), ("rr_len", C.c_size_t), ("pread_ids", C.c_char_p), ("pi_len", C.c_size_t),
                ("pread_to_contigs", C.c_char_p), ("pc_len", C.c_size_t), ("ctg_index", C.c_void_p), ("n_threads", C.c_int32), ("n_lanes", C.c_int32),
                ("group_bases", C.c_int64), ("flags", C.c_uint), ("align", AlignParams)]


class PipeOut(C.Structure):
    _fields_ = [("r2p", C.c_void_p), ("n_r2p", C.c_int64)] + [(k, C.c_int64) for k in ("n_reads", "n_aligned", "n_rec", "n_sites", "n_rows", "n_arows", "n_pvars",
                                                                                     "n_preads", "n_groups", "bytes_written")] + \
               [(k, C.c_double) for k in ("dp_cells", "ms_upload", "ms_k1", "ms_phase", "ms_results", "ms_text")]


def load():
    """Load the shared library; raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(make -C falcon_unzip_amd/csrc).  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    VP, I32, I64, SZ, PP = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t, C.POINTER(C.c_void_p)
    PI64, PSZ, CP = C.POINTER(C.c_int64), C.POINTER(C.c_size_t), C.c_char_p
    sigs = {
        "fzp_free": (None, [VP]),
        "fzp_ctx_create": (C.c_int, [C.c_int, C.c_uint, PP]),
        "fzp_ctx_destroy": (None, [VP]),
        "fzp_ctx_synchronize": (C.c_int, [VP]),
        "fzp_prof_enable": (C.c_int, [VP, C.c_int]),
        "fzp_prof_reset": (C.c_int, [VP]),
        "fzp_prof_get": (C.c_int, [VP, CP, C.POINTER(C.c_double), PI64]),
        "fzp_prof_names": (C.c_int, [VP, PP]),
        "fzp_parse_sam": (C.c_int, [CP, SZ, PP]),
        "fzp_alnset_free": (None, [VP]),
        "fzp_result_free": (None, [VP]),
        "fzp_het_call": (C.c_int, [VP, VP, CP, I64, PP, PI64, PP, PI64]),
        "fzp_assoc_table": (C.c_int, [VP, VP, I64, VP, I64, PP, PI64]),
        "fzp_phase_blocks": (C.c_int, [VP, VP, I64, VP, I64, PP, PI64]),
        "fzp_phase_reads": (C.c_int, [VP, VP, I64, VP, I64, VP, I64, I32, PP, PI64]),
        "fzp_batch_create": (C.c_int, [VP, I32, VP, VP, VP, PP]),
        "fzp_batch_run": (C.c_int, [VP, VP, C.c_uint]),
        "fzp_batch_result": (C.c_int, [VP, VP, I32, VP]),
        "fzp_batch_counts": (C.c_int, [VP, VP] + [PI64] * 8),
        "fzp_batch_result_all": (C.c_int, [VP, VP, VP]),
        "fzp_result_all_free": (None, [VP]),
        "fzp_batch_destroy": (None, [VP, VP]),
        "fzp_format_variant_pos": (C.c_int, [VP, I64, PP, PSZ]),
        "fzp_format_variant_map": (C.c_int, [VP, I64, VP, PP, PSZ]),
        "fzp_format_q_id_map": (C.c_int, [VP, PP, PSZ]),
        "fzp_format_atable": (C.c_int, [VP, VP, I64, PP, PSZ]),
        "fzp_format_phased_variants": (C.c_int, [VP, VP, I64, PP, PSZ]),
        "fzp_format_phased_reads": (C.c_int, [VP, I64, CP, VP, CP, I32, PP, PSZ]),
        "fzp_format_sam": (C.c_int, [VP, CP, VP, PP, PSZ]),
        "fzp_readmap": (C.c_int, [CP, SZ, CP, SZ, CP, SZ, CP, SZ, CP, I32, PP, PI64, PP, PSZ]),
        "fzp_align_params_default": (None, [VP]),
        "fzp_align_create": (C.c_int, [VP, I32, VP, VP, I64, VP, VP, VP, VP, PP]),
        "fzp_align_create_spans": (C.c_int, [VP, I32, VP, VP, I64, VP, VP, VP, VP, PP]),
        "fzp_align_run": (C.c_int, [VP, VP]),
        "fzp_align_invalidate_index": (C.c_int, [VP]),
        "fzp_align_summaries": (C.c_int, [VP, VP, VP]),
        "fzp_align_cigar_hashes": (C.c_int, [VP, VP, VP]),
        "fzp_align_n_second": (I64, [VP]),
        "fzp_batch_text": (C.c_int, [VP, VP, C.c_int, PP, PSZ, PP]),
        "fzp_pipe_opts_default": (None, [VP]),
        "fzp_job_phase_write": (C.c_int, [VP, VP, VP, VP, VP]),
        "fzp_phase_contigs": (C.c_int, [VP, I32, VP, VP, I64, VP, VP, VP, VP, VP, VP]),
        "fzp_phase_contigs_files": (C.c_int, [VP, CP, VP, VP, VP]),
        "fzp_pipe_out_free": (None, [VP]),
        "fzp_pipe_flush": (C.c_int, [VP]),
        "fzp_mem_info": (C.c_int, [VP, PSZ, PSZ]),
        "fzp_comm_unique_id": (C.c_int, [VP]),
        "fzp_comm_create": (C.c_int, [VP, C.c_int, C.c_int, VP, PP]),
        "fzp_comm_ranks": (C.c_int, [VP, VP, VP]),
        "fzp_comm_destroy": (None, [VP]),
        "fzp_allgather_rid_to_phase": (C.c_int, [VP, VP, I64, PP, PI64]),
        "fzp_format_rid_to_phase_all": (C.c_int, [VP, I64, VP, I32, PP, PSZ]),
        "fzp_align_alnset": (C.c_int, [VP, VP, I32, VP, CP, PP, PP]),
        "fzp_align_alnset_all": (C.c_int, [VP, VP, I32, VP, CP, PP, PP]),
        "fzp_align_to_batch": (C.c_int, [VP, VP, PP]),
        "fzp_align_destroy": (None, [VP, VP]),
        "fzp_batch_consensus": (C.c_int, [VP, VP, VP]),
        "fzp_batch_consensus_v": (C.c_int, [VP, VP, C.c_int, VP]),
        "fzp_polish_tigs": (C.c_int, [VP, I32, VP, VP, I64, VP, VP, CP, VP, VP]),
        "fzp_tigs_free": (None, [VP]),
        "fzp_format_tigs": (C.c_int, [VP, I32, CP, PP, PSZ]),
        "fzp_format_bam": (C.c_int, [VP, CP, I64, VP, PP, PSZ, PP, PSZ]),
        "fzp_bam_to_sam": (C.c_int, [CP, SZ, CP, PP, PSZ]),
        "fzp_bam_open": (C.c_int, [CP, SZ, PP]),
        "fzp_bam_view_free": (None, [VP]),
        "fzp_bam_write": (C.c_int, [CP, SZ, I32, CP, SZ, I32, VP, VP, PP, PSZ]),
        "fzp_bam_read_header": (C.c_int, [CP, PP, PSZ, VP, PP, PSZ]),
        "fzp_bam_route": (C.c_int, [I32, VP, C.c_int64, VP, CP, VP, I32, VP, CP, SZ, I32, CP, SZ, VP, VP]),
        "fzp_ovl_parse": (C.c_int, [VP, I32, VP, VP, CP, SZ, PP]),
        "fzp_ovlset_free": (None, [VP]),
        "fzp_ovl_n_lines": (I64, [VP]),
        "fzp_ovl_n_rows": (I64, [VP]),
        "fzp_ovl_filter": (C.c_int, [VP, VP, VP, PP, PI64, PP, PI64, PP, PI64]),
        "fzp_ovl_format": (C.c_int, [VP, VP, I64, PP, PSZ]),
        "fzp_ovl_id_name": (C.c_int, [VP, I32, PP, C.POINTER(I32)]),
        "fzp_track_reads": (C.c_int, [VP, I32, VP, VP, CP, SZ, CP, SZ, CP, SZ, I64, I64, PP, PSZ]),
    }
    lib.fzp_last_error.restype = C.c_char_p
    lib.fzp_version.restype = C.c_char_p
    for name, (res, args) in sigs.items():
        if hasattr(lib, name):
            f = getattr(lib, name)
            f.restype = res
            f.argtypes = args
    _lib = lib
    return lib


def _check(rc):
    if rc != FZP_OK:
        raise FzpError(rc, load().fzp_last_error().decode("utf-8", "replace"))


def _take(ptr, n, dtype):
    """Copy a library-allocated array into numpy and free it."""
    lib = load()
    n = int(n)
    if not ptr:
        return np.zeros(0, dtype=dtype)
    arr = _copy_in(ptr, n, dtype)
    lib.fzp_free(ptr)
    return arr


def _copy_in(ptr, n, dtype):
    """n records at `ptr` into a fresh numpy array with ONE memmove.  (np.frombuffer(...).copy() of a structured dtype goes field by field: 0.3 ms for 40 000 rid_to_phase
    records, 2 ms for the 320 000 an eight-rank gather returns -- per step.)"""
    arr = np.empty(int(n), dtype=dtype)
    if n:
        C.memmove(arr.ctypes.data, ptr, int(n) * arr.dtype.itemsize)
    return arr


def _take_text(ptr, n):
    lib = load()
    s = C.string_at(ptr, n.value)
    lib.fzp_free(ptr)
    return s


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class AlnSet:
    """Owning handle of a fzp_alnset (accepted alignment records of one contig + the q_id table)."""

    def __init__(self, ptr):
        self._p = ptr
        self.s = C.cast(ptr, C.POINTER(AlnSetStruct)).contents

    def __del__(self):
        if getattr(self, "_p", None):
            load().fzp_alnset_free(self._p)
            self._p = None

    n_rec = property(lambda self: int(self.s.n_rec))
    n_qid = property(lambda self: int(self.s.n_qid))
    last_pos = property(lambda self: int(self.s.last_pos))
    n_columns = property(lambda self: int(self.s.n_columns))

    def qname_table(self):
        off = np.ctypeslib.as_array(self.s.qname_off, shape=(self.n_qid + 1,)).copy()
        names = C.string_at(self.s.qnames, int(off[-1]))
        return off, names

    def qnames(self):
        off, names = self.qname_table()
        return [names[off[i]:off[i + 1]].decode() for i in range(self.n_qid)]

    def rec_pos(self):
        return np.ctypeslib.as_array(self.s.rec_pos, shape=(self.n_rec,)).copy() if self.n_rec else np.zeros(0, np.int32)

    def rec_qid(self):
        return np.ctypeslib.as_array(self.s.rec_qid, shape=(self.n_rec,)).copy() if self.n_rec else np.zeros(0, np.int32)

    def cigar_of(self, r):
        a, b = int(self.s.cig_off[r]), int(self.s.cig_off[r + 1])
        return [(int(self.s.cigar[k]) >> 4, int(self.s.cigar[k]) & 15) for k in range(a, b)]

    def seq_of(self, r):
        a, b = int(self.s.seq_off[r]), int(self.s.seq_off[r + 1])
        return C.string_at(C.addressof(self.s.seq.contents) + a, b - a) if b > a else b""


def parse_sam(sam: bytes) -> AlnSet:
    lib = load()
    p = C.c_void_p()
    _check(lib.fzp_parse_sam(sam, C.c_size_t(len(sam)), C.byref(p)))
    return AlnSet(p.value)


class Result:
    """Records of one contig (numpy copies; views of the library's pinned buffer when copy=False)."""

    def __init__(self, rs: ResultStruct, copy=True):
        def grab(ptr, n, dt):
            n = int(n)
            if not ptr or n == 0:
                return np.zeros(0, dtype=dt)
            nbytes = n * np.dtype(dt).itemsize
            raw = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(nbytes,))
            return raw.view(dt).copy() if copy else raw.view(dt)
        self.sites = grab(rs.sites, rs.n_sites, SITE)
        self.vmap_qid = grab(rs.vmap_qid, rs.n_rows, np.int32)
        self.arows = grab(rs.arows, rs.n_arows, AROW)
        self.pvars = grab(rs.pvars, rs.n_pvars, PVAR)
        self.preads = grab(rs.preads, rs.n_preads, PREAD)


class Batch:
    def __init__(self, eng, ptr, alnsets):
        self.eng, self._p, self._keep = eng, ptr, alnsets
        self.n_ctg = len(alnsets)

    def run(self, stages=STAGE_ALL):
        _check(load().fzp_batch_run(self.eng._p, self._p, C.c_uint(stages)))

    def result(self, ctg) -> Result:
        rs = ResultStruct()
        _check(load().fzp_batch_result(self.eng._p, self._p, C.c_int32(ctg), C.byref(rs)))
        r = Result(rs)
        load().fzp_result_free(C.byref(rs))
        return r

    def results(self, copy=True):
        """Every contig's records with ONE device-to-host copy per array -> list of Result (local indices).

        copy=False hands out views of the batch's pinned staging block: valid until the batch is run again,
        asked for results again or closed."""
        lib = load()
        ra = ResultAllStruct()
        _check(lib.fzp_batch_result_all(self.eng._p, self._p, C.byref(ra)))
        full = Result(ra.all, copy=copy)
        n = self.n_ctg

        def beg(p):
            return np.ctypeslib.as_array(p, shape=(n + 1,)).copy() if p else np.zeros(n + 1, np.int64)
        sb, rb, ab, pb, qb = beg(ra.site_begin), beg(ra.row_begin), beg(ra.arow_begin), beg(ra.pvar_begin), beg(ra.pread_begin)
        lib.fzp_result_all_free(C.byref(ra))
        # contig-local indices, vectorised over the whole batch (views are handed out below)
        if len(full.sites):
            full.sites["row_off"] -= np.repeat(rb[:-1], np.diff(sb))
        if len(full.arows):
            shift = np.repeat(sb[:-1], np.diff(ab)).astype(np.int32)
            full.arows["site1"] -= shift
            full.arows["site2"] -= shift
        if len(full.pvars):
            full.pvars["site"] -= np.repeat(sb[:-1], np.diff(pb)).astype(np.int32)
        self.last_full = (full, {"site": sb, "row": rb, "arow": ab, "pvar": pb, "pread": qb})   # whole-batch arrays + per-contig begins
        out = []
        for c in range(n):
            r = Result.__new__(Result)
            r.sites = full.sites[sb[c]:sb[c + 1]]
            r.vmap_qid = full.vmap_qid[rb[c]:rb[c + 1]]
            r.arows = full.arows[ab[c]:ab[c + 1]]
            r.pvars = full.pvars[pb[c]:pb[c + 1]]
            r.preads = full.preads[qb[c]:qb[c + 1]]
            out.append(r)
        return out

    def text(self, what):
        """`het_call/variant_map` (what=1) or `g_atable/atable` (what=2) of all contigs, serialised on the device -> (bytes, ctg_begin)"""
        p, n, cb = C.c_void_p(), C.c_size_t(), C.c_void_p()
        _check(load().fzp_batch_text(self.eng._p, self._p, what, C.byref(p), C.byref(n), C.byref(cb)))
        begin = _take(cb.value, self.n_ctg + 1, np.int64)
        return _take_text(p, n), begin

    def consensus(self, version=3) -> "Tigs":
        """K6: phased-pile consensus of every (block, phase) (fzcns v3: insertion length by the pile's median, then its bases; version=2: bases gated
        on prefix-linked majorities; version=1: one inserted base at most); run(STAGE_ALL) first."""
        ts = TigsStruct()
        _check(load().fzp_batch_consensus_v(self.eng._p, self._p, version, C.byref(ts)))
        return Tigs(ts)

    def counts(self):
        v = [C.c_int64() for _ in range(8)]
        _check(load().fzp_batch_counts(self.eng._p, self._p, *[C.byref(x) for x in v]))
        keys = ("n_rec", "n_columns", "n_positions", "n_sites", "n_rows", "n_arows", "n_pvars", "n_preads")
        return dict(zip(keys, (int(x.value) for x in v)))

    def close(self):
        if self._p:
            load().fzp_batch_destroy(self.eng._p, self._p)
            self._p = None

    __del__ = close


class Engine:
    """One fzp_ctx: a (process, device) pair.  Fails loudly when no gfx950 device is usable."""

    def __init__(self, device=0):
        lib = load()
        p = C.c_void_p()
        _check(lib.fzp_ctx_create(C.c_int(device), C.c_uint(0), C.byref(p)))
        self._p = p.value
        self.device = device

    def close(self):
        if getattr(self, "_p", None):
            load().fzp_ctx_destroy(self._p)
            self._p = None

    __del__ = close

    def mem_info(self):
        """(free, total) bytes of the engine's device"""
        fr, tot = C.c_size_t(), C.c_size_t()
        _check(load().fzp_mem_info(self._p, C.byref(fr), C.byref(tot)))
        return int(fr.value), int(tot.value)

    def pipe_flush(self):
        """wait for the background file writes of phase_write(async_writes=True) calls; raises on the first write error"""
        _check(load().fzp_pipe_flush(self._p))

    def synchronize(self):
        _check(load().fzp_ctx_synchronize(self._p))

    # ---- profiling
    def prof_enable(self, on=True):
        """True / 1: every kernel bracket; 2: the DP stage (k1_sw) only; False / 0: off"""
        _check(load().fzp_prof_enable(self._p, C.c_int(int(on) if not isinstance(on, bool) else (1 if on else 0))))

    def prof_reset(self):
        _check(load().fzp_prof_reset(self._p))

    def prof(self):
        lib = load()
        p = C.c_void_p()
        _check(lib.fzp_prof_names(self._p, C.byref(p)))
        names = C.string_at(p).decode().split("\n")
        lib.fzp_free(p)
        out = {}
        for nme in names:
            if not nme:
                continue
            ms, cnt = C.c_double(), C.c_int64()
            _check(lib.fzp_prof_get(self._p, nme.encode(), C.byref(ms), C.byref(cnt)))
            out[nme] = (ms.value, cnt.value)
        return out

    # ---- stages (single contig, host records)
    def het_call(self, aln: AlnSet, ref_seq: bytes):
        lib = load()
        sp, ns, qp, nr = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64()
        _check(lib.fzp_het_call(self._p, aln._p, ref_seq, len(ref_seq), C.byref(sp), C.byref(ns), C.byref(qp), C.byref(nr)))
        return _take(sp.value, ns.value, SITE), _take(qp.value, nr.value, np.int32)

    def assoc_table(self, sites, vmap_qid):
        lib = load()
        sites = np.ascontiguousarray(sites, dtype=SITE)
        vmap_qid = np.ascontiguousarray(vmap_qid, dtype=np.int32)
        ap, na = C.c_void_p(), C.c_int64()
        _check(lib.fzp_assoc_table(self._p, _ptr(sites), C.c_int64(len(sites)), _ptr(vmap_qid), C.c_int64(len(vmap_qid)), C.byref(ap), C.byref(na)))
        return _take(ap.value, na.value, AROW)

    def phase_blocks(self, sites, arows):
        lib = load()
        sites = np.ascontiguousarray(sites, dtype=SITE)
        arows = np.ascontiguousarray(arows, dtype=AROW)
        pp, npv = C.c_void_p(), C.c_int64()
        _check(lib.fzp_phase_blocks(self._p, _ptr(sites), C.c_int64(len(sites)), _ptr(arows), C.c_int64(len(arows)), C.byref(pp), C.byref(npv)))
        return _take(pp.value, npv.value, PVAR)

    def phase_reads(self, sites, vmap_qid, pvars, n_qid):
        lib = load()
        sites = np.ascontiguousarray(sites, dtype=SITE)
        vmap_qid = np.ascontiguousarray(vmap_qid, dtype=np.int32)
        pvars = np.ascontiguousarray(pvars, dtype=PVAR)
        rp, nr = C.c_void_p(), C.c_int64()
        _check(lib.fzp_phase_reads(self._p, _ptr(sites), C.c_int64(len(sites)), _ptr(vmap_qid), C.c_int64(len(vmap_qid)), _ptr(pvars),
                                   C.c_int64(len(pvars)), C.c_int32(n_qid), C.byref(rp), C.byref(nr)))
        return _take(rp.value, nr.value, PREAD)

    # ---- many contigs, HBM-resident
    def batch(self, alnsets, ref_seqs) -> Batch:
        lib = load()
        n = len(alnsets)
        aptr = (C.c_void_p * n)(*[a._p for a in alnsets])
        bufs = [C.create_string_buffer(r, len(r)) if len(r) else C.create_string_buffer(1) for r in ref_seqs]
        rptr = (C.c_void_p * n)(*[C.cast(b, C.c_void_p).value for b in bufs])
        lens = (C.c_int64 * n)(*[len(r) for r in ref_seqs])
        p = C.c_void_p()
        _check(lib.fzp_batch_create(self._p, C.c_int32(n), aptr, rptr, lens, C.byref(p)))
        return Batch(self, p.value, list(alnsets))


class AlignJob:
    """K1 job: reads of one or more contigs resident in HBM (fzp_align_*)."""

    def __init__(self, eng, ptr, n_reads, n_ctg):
        self.eng, self._p, self.n_reads, self.n_ctg = eng, ptr, n_reads, n_ctg
        self._batches = []      # weak references to the batches to_batch made: they read this job's packed records in place (include/fzphase.h, fzp_align_to_batch)

    def _open_batches(self):
        self._batches = [w for w in self._batches if w() is not None and w()._p]
        return len(self._batches)

    def run(self):
        if self._open_batches():
            raise FzpError(-1, "AlignJob.run: a Batch made by to_batch() is still open -- it reads this job's records in place; close it first")
        _check(load().fzp_align_run(self.eng._p, self._p))

    def n_second(self):
        """reads of the last run whose second candidate placement was extended as well"""
        return int(load().fzp_align_n_second(self._p))

    def summaries(self):
        out = np.zeros(self.n_reads, ALN_SUMMARY)
        _check(load().fzp_align_summaries(self.eng._p, self._p, _ptr(out)))
        return out

    def cigar_hashes(self):
        """64-bit fingerprint of every read's CIGAR (M / I / D / S runs as the device keeps them; 0 = unaligned): see cigar_hash_of_words for the same over another aligner's words"""
        out = np.zeros(self.n_reads, np.uint64)
        _check(load().fzp_align_cigar_hashes(self.eng._p, self._p, _ptr(out)))
        return out

    def alnset(self, ctg=0, names=None, all_records=False):
        """Alignment records of contig `ctg` as the phasing stages see them -> (AlnSet, read_index); all_records=True: also the
        reads make_het_call's filters drop (what the blasr task's BAM holds)."""
        lib = load()
        fn = lib.fzp_align_alnset_all if all_records else lib.fzp_align_alnset
        ap, ip = C.c_void_p(), C.c_void_p()
        if names is not None:
            if isinstance(names, tuple):                      # (name_off, blob) built once by the caller
                off, blob = names
            else:
                enc = [nm.encode() if isinstance(nm, str) else nm for nm in names]
                off = np.zeros(len(enc) + 1, np.int64)
                off[1:] = np.cumsum([len(e) for e in enc])
                blob = b"".join(enc)
            _check(fn(self.eng._p, self._p, ctg, _ptr(off), blob, C.byref(ap), C.byref(ip)))
        else:
            _check(fn(self.eng._p, self._p, ctg, None, None, C.byref(ap), C.byref(ip)))
        a = AlnSet(ap.value)
        idx = _take(ip.value, a.n_rec, np.int64)
        return a, idx

    def phase_write(self, ctg_ids, names=None, out_dir=None, read_maps=None, ctg_index=None, n_threads=0, consensus=False, async_writes=False, rebuild_index=False,
                    bam=False, sentinels=False):
        """fzp_job_phase_write: K1 -> K5 of every contig of the job, every file of every contig under out_dir, rid_to_phase records.
        names: (name_off int64 [n_reads+1], blob) or a list; read_maps: (rawread_ids, pread_ids, pread_to_contigs) bytes.  -> (stats dict, R2P records)"""
        nm, opts, keep = _pipe_args(ctg_ids, names, out_dir, read_maps, ctg_index, n_threads, 0, 0, None,
                                    (PIPE_CONSENSUS if consensus else 0) | (PIPE_ASYNC_WRITES if async_writes else 0) | (PIPE_REBUILD_INDEX if rebuild_index else 0)
                                    | (PIPE_BAM if bam else 0) | (PIPE_SENTINELS if sentinels else 0))
        out = PipeOut()
        _check(load().fzp_job_phase_write(self.eng._p, self._p, C.byref(nm), C.byref(opts), C.byref(out)))
        return _pipe_result(out)

    def to_batch(self) -> "Batch":
        p = C.c_void_p()
        _check(load().fzp_align_to_batch(self.eng._p, self._p, C.byref(p)))
        b = Batch(self.eng, p.value, [None] * self.n_ctg)
        b._job = self      # the batch reads this job's packed records (2-bit op streams, 2-bit reads) where they lie: the job stays alive as long as the batch does
        self._batches.append(weakref.ref(b))
        return b

    def close(self):
        if self._p and self._open_batches():
            raise FzpError(-1, "AlignJob.close: a Batch made by to_batch() is still open -- close it first")
        if self._p:
            load().fzp_align_destroy(self.eng._p, self._p)
            self._p = None

    __del__ = close


def _contig_ptrs(contigs):
    """-> (keep-alive list, void* array, int64 lengths) of the contigs' bytes where they are: bytes objects lend their buffer, anything
    else goes through numpy; no copies"""
    nc = len(contigs)
    bufs = [c if isinstance(c, bytes) else np.ascontiguousarray(np.frombuffer(c, dtype=np.uint8)) for c in contigs]
    empty = C.create_string_buffer(1)
    ptr = (C.c_void_p * nc)(*[(C.cast(C.c_char_p(b), C.c_void_p).value if isinstance(b, bytes) else b.ctypes.data) if len(b) else C.addressof(empty) for b in bufs])
    return (bufs, empty), ptr, (C.c_int64 * nc)(*[len(c) for c in contigs])


def align_job(eng, contigs, reads, read_ctg=None, params=None) -> AlignJob:
    """contigs: list of bytes; reads: list of bytes (as sequenced); read_ctg: contig index per read."""
    lib = load()
    nc, nr = len(contigs), len(reads)
    cbufs, cptr, clen = _contig_ptrs(contigs)
    rc = np.zeros(nr, np.int32) if read_ctg is None else np.ascontiguousarray(read_ctg, dtype=np.int32)
    off = np.zeros(nr + 1, np.int64)
    off[1:] = np.cumsum([len(r) for r in reads])
    blob = b"".join(reads)
    P = AlignParams()
    lib.fzp_align_params_default(C.byref(P))
    for k, v in (params or {}).items():
        setattr(P, k, v)
    p = C.c_void_p()
    _check(lib.fzp_align_create(eng._p, nc, cptr, clen, nr, _ptr(rc), _ptr(off), blob, C.byref(P), C.byref(p)))
    return AlignJob(eng, p.value, nr, nc)


def align_band() -> int:
    """cells of K1's adaptive band under the default parameters (fzalign v1.8: 32; FZP_ALIGN_BAND=64 brings v1.7's band back): a read's `cells` = its DP steps x this"""
    P = AlignParams()
    load().fzp_align_params_default(C.byref(P))
    return int(P.band)


def polish_tigs(eng, tigs, read_blob: bytes, read_off, read_tig, params=None) -> "Tigs":
    """fzp_polish_tigs: every read (read_blob[read_off[r]:read_off[r + 1]], as sequenced) aligned to ITS tig (tigs[read_tig[r]]) by K1, the whole tig called as one pile
    by K6 (fzcns v3) -- the consensus role of run_quiver.py:82-97.  -> Tigs with one entry per input tig, input order (n_records = 0: no read aligned, the tig as it came)."""
    lib = load()
    nt = len(tigs)
    read_off = np.ascontiguousarray(read_off, dtype=np.int64)
    nr = len(read_off) - 1
    cbufs, cptr, clen = _contig_ptrs(tigs)
    rt = np.ascontiguousarray(read_tig, dtype=np.int32)
    P = AlignParams()
    lib.fzp_align_params_default(C.byref(P))
    for k, v in (params or {}).items():
        setattr(P, k, v)
    ts = TigsStruct()
    _check(lib.fzp_polish_tigs(eng._p, nt, cptr, clen, nr, _ptr(rt), _ptr(read_off), read_blob, C.byref(P), C.byref(ts)))
    return Tigs(ts)


def align_job_raw(eng, contigs, read_blob: bytes, read_off, read_ctg, params=None) -> AlignJob:
    """Like align_job, with the reads already concatenated (read_off: int64 [n+1])."""
    lib = load()
    nc = len(contigs)
    read_off = np.ascontiguousarray(read_off, dtype=np.int64)
    nr = len(read_off) - 1
    cbufs, cptr, clen = _contig_ptrs(contigs)
    rc = np.ascontiguousarray(read_ctg, dtype=np.int32)
    P = AlignParams()
    lib.fzp_align_params_default(C.byref(P))
    for k, v in (params or {}).items():
        setattr(P, k, v)
    p = C.c_void_p()
    _check(lib.fzp_align_create(eng._p, nc, cptr, clen, nr, _ptr(rc), _ptr(read_off), read_blob, C.byref(P), C.byref(p)))
    return AlignJob(eng, p.value, nr, nc)


def align_job_spans(eng, contigs, buf: bytes, read_be, read_ctg, params=None) -> AlignJob:
    """fzp_align_create_spans: read r = buf[read_be[r, 0] : read_be[r, 1]] -- e.g. the bytes of a FASTA file as they are, with the spans of its sequence lines."""
    lib = load()
    nc = len(contigs)
    be = np.ascontiguousarray(read_be, dtype=np.int64).reshape(-1, 2)
    nr = len(be)
    cbufs, cptr, clen = _contig_ptrs(contigs)
    rc = np.ascontiguousarray(read_ctg, dtype=np.int32)
    P = AlignParams()
    lib.fzp_align_params_default(C.byref(P))
    for k, v in (params or {}).items():
        setattr(P, k, v)
    p = C.c_void_p()
    _check(lib.fzp_align_create_spans(eng._p, nc, cptr, clen, nr, _ptr(rc), _ptr(be), buf, C.byref(P), C.byref(p)))
    return AlignJob(eng, p.value, nr, nc)


def cigar_hash_of_words(cigars):
    """AlignJob.cigar_hashes() for CIGARs given as a list of uint32 word arrays (len << 4 | op) that may spell aligned columns as '=' / 'X' runs: those become M and adjacent M
    runs merge (the form the device keeps), then sum over the words of splitmix64(index << 32 | word) mod 2^64; an empty CIGAR hashes to 0."""
    n = len(cigars)
    out = np.zeros(n, np.uint64)
    lens = np.array([len(c) for c in cigars], np.int64)
    if lens.sum() == 0:
        return out
    w = np.concatenate([np.asarray(c, np.uint32) for c in cigars if len(c)]).astype(np.uint64)
    rid = np.repeat(np.arange(n), lens)
    op = w & np.uint64(15)
    op = np.where((op == 7) | (op == 8), np.uint64(0), op)
    ln = w >> np.uint64(4)
    start = np.ones(len(w), bool)
    start[1:] = (op[1:] != op[:-1]) | (rid[1:] != rid[:-1]) | (op[1:] != 0)      # only M runs merge
    idx = np.flatnonzero(start)
    mlen = np.add.reduceat(ln, idx)
    mw = (mlen << np.uint64(4)) | op[idx]
    mr = rid[idx]
    first = np.ones(len(idx), bool)
    first[1:] = mr[1:] != mr[:-1]
    fpos = np.flatnonzero(first)
    k = np.arange(len(idx), dtype=np.int64) - np.repeat(fpos, np.diff(np.append(fpos, len(idx))))
    with np.errstate(over="ignore"):
        x = (k.astype(np.uint64) << np.uint64(32)) | mw
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
        out[mr[fpos]] = np.add.reduceat(x, fpos)
    return out


def _pipe_args(ctg_ids, names, out_dir, read_maps, ctg_index, n_threads, n_lanes, group_bases, params, flags=0):
    lib = load()
    keep = []
    ids = [c.encode() if isinstance(c, str) else c for c in ctg_ids]
    arr = (C.c_char_p * len(ids))(*ids)
    nm = NamesStruct()
    nm.n_ctg = len(ids)
    nm.ctg_id = arr
    if names is not None:
        if isinstance(names, tuple):
            off, blob = names
            off = np.ascontiguousarray(off, dtype=np.int64)
        else:
            enc = [x.encode() if isinstance(x, str) else x for x in names]
            off = np.zeros(len(enc) + 1, np.int64)
            off[1:] = np.cumsum([len(e) for e in enc])
            blob = b"".join(enc)
        keep += [off, blob]
        nm.name_off = off.ctypes.data
        nm.names = blob
    opts = PipeOpts()
    lib.fzp_pipe_opts_default(C.byref(opts))
    if out_dir is not None:
        opts.out_dir = out_dir.encode() if isinstance(out_dir, str) else out_dir
    if read_maps is not None:
        rr, pi, pc = read_maps
        keep += [rr, pi, pc]
        opts.rawread_ids, opts.rr_len, opts.pread_ids, opts.pi_len, opts.pread_to_contigs, opts.pc_len = rr, len(rr), pi, len(pi), pc, len(pc)
    if ctg_index is not None:
        ci = np.ascontiguousarray(ctg_index, dtype=np.int32)
        keep.append(ci)
        opts.ctg_index = ci.ctypes.data
    opts.n_threads, opts.n_lanes, opts.group_bases, opts.flags = n_threads, n_lanes, group_bases, flags
    for k, v in (params or {}).items():
        setattr(opts.align, k, v)
    keep += [arr, ids]
    return nm, opts, keep


def _pipe_result(out):
    recs = _copy_in(out.r2p, out.n_r2p, R2P) if out.n_r2p else np.zeros(0, R2P)
    stats = {k: getattr(out, k) for k, _ in PipeOut._fields_ if k not in ("r2p",)}
    load().fzp_pipe_out_free(C.byref(out))
    return stats, recs


def phase_contigs(eng, contigs, read_blob: bytes, read_off, read_ctg, ctg_ids, names=None, out_dir=None, read_maps=None, ctg_index=None,
                  n_threads=0, n_lanes=0, group_bases=0, params=None, consensus=False, async_writes=False, bam=False, sentinels=False):
    """fzp_phase_contigs: inputs in host memory -> every file of every contig + rid_to_phase records; contig groups are streamed through the
    device on `n_lanes` lanes.  bam: also <ctg>/blasr/<ctg>_sorted.bam(.bai) from the same alignment pass; sentinels: the job_done files
    of the reference's two per-contig tasks.  -> (stats dict, R2P records)"""
    lib = load()
    nc = len(contigs)
    read_off = np.ascontiguousarray(read_off, dtype=np.int64)
    nr = len(read_off) - 1
    cbufs, cptr, clen = _contig_ptrs(contigs)
    rc = np.ascontiguousarray(read_ctg, dtype=np.int32)
    nm, opts, keep = _pipe_args(ctg_ids, names, out_dir, read_maps, ctg_index, n_threads, n_lanes, group_bases, params,
                                (PIPE_CONSENSUS if consensus else 0) | (PIPE_ASYNC_WRITES if async_writes else 0) | (PIPE_BAM if bam else 0) | (PIPE_SENTINELS if sentinels else 0))
    out = PipeOut()
    _check(lib.fzp_phase_contigs(eng._p, nc, cptr, clen, nr, _ptr(rc), _ptr(read_off), read_blob, C.byref(nm), C.byref(opts), C.byref(out)))
    return _pipe_result(out)


def phase_contigs_files(eng, reads_dir, ctg_ids, out_dir=None, read_maps=None, ctg_index=None, n_threads=0, n_lanes=0, group_bases=0, params=None, consensus=False,
                        async_writes=False, bam=False, sentinels=False):
    """fzp_phase_contigs_files: like phase_contigs, the inputs read by the library from <reads_dir>/<ctg>_ref.fa and <ctg>_reads.fa.  -> (stats dict, R2P records)"""
    nm, opts, keep = _pipe_args(ctg_ids, None, out_dir, read_maps, ctg_index, n_threads, n_lanes, group_bases, params,
                                (PIPE_CONSENSUS if consensus else 0) | (PIPE_ASYNC_WRITES if async_writes else 0) | (PIPE_BAM if bam else 0) | (PIPE_SENTINELS if sentinels else 0))
    out = PipeOut()
    _check(load().fzp_phase_contigs_files(eng._p, reads_dir.encode() if isinstance(reads_dir, str) else reads_dir, C.byref(nm), C.byref(opts), C.byref(out)))
    return _pipe_result(out)


def comm_unique_id() -> bytes:
    """RCCL unique id (rank 0 creates it and hands it to the other ranks)"""
    buf = C.create_string_buffer(128)
    _check(load().fzp_comm_unique_id(buf))
    return buf.raw


def sched_status():
    lib = load()
    lib.fzp_sched_status.restype = C.c_int
    return int(lib.fzp_sched_status())


def comm_library():
    """(path, version) of the RCCL the library bound in this process (fzp_comm_library); raises when RCCL cannot be loaded"""
    lib = load()
    lib.fzp_comm_library.restype = C.c_int
    lib.fzp_comm_library.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
    buf, ver = C.create_string_buffer(1024), C.c_int()
    _check(lib.fzp_comm_library(buf, 1024, C.byref(ver)))
    return buf.value.decode(), int(ver.value)


class Comm:
    """fzp_comm: this rank's RCCL communicator for the rid_to_phase all-gather (one process per GPU)."""

    def __init__(self, eng, rank, world, uid: bytes):
        self._p = None
        p = C.c_void_p()
        self._keep = C.create_string_buffer(uid, 128)
        _check(load().fzp_comm_create(eng._p, rank, world, self._keep, C.byref(p)))
        self._p, self.eng = p.value, eng

    def allgather_r2p(self, local):
        local = np.ascontiguousarray(local, dtype=R2P)
        p, n = C.c_void_p(), C.c_int64()
        _check(load().fzp_allgather_rid_to_phase(self._p, _ptr(local), len(local), C.byref(p), C.byref(n)))
        return _take(p.value, n.value, R2P)

    def ranks(self):
        """(rank, world) as the communicator itself reports them"""
        r, w = C.c_int(), C.c_int()
        _check(load().fzp_comm_ranks(self._p, C.cast(C.byref(r), C.c_void_p), C.cast(C.byref(w), C.c_void_p)))
        return r.value, w.value

    def close(self):
        if self._p:
            load().fzp_comm_destroy(self._p)
            self._p = None

    __del__ = close


def format_rid_to_phase_all(records, ctg_ids):
    """`rid_to_phase.all` (unzip.py:285) from gathered records in the order given: fzp_format_rid_to_phase_all"""
    recs = np.ascontiguousarray(records, dtype=R2P)
    enc = [c.encode() if isinstance(c, str) else c for c in ctg_ids]
    arr = (C.c_char_p * max(1, len(enc)))(*enc)
    return _fmt("fzp_format_rid_to_phase_all", _ptr(recs), C.c_int64(len(recs)), arr, C.c_int32(len(enc)))


def format_sam(aln: AlnSet, ctg_id: str, flags=None):
    fl = None if flags is None else np.ascontiguousarray(flags, dtype=np.int32)
    return _fmt("fzp_format_sam", C.c_void_p(aln._p), ctg_id.encode(), None if fl is None else _ptr(fl))


# ---------------------------------------------------------------------------- serializers
def _fmt(fname, *args):
    lib = load()
    tp, tn = C.c_void_p(), C.c_size_t()
    _check(getattr(lib, fname)(*args, C.byref(tp), C.byref(tn)))
    return _take_text(tp, tn)


def format_variant_pos(sites):
    sites = np.ascontiguousarray(sites, dtype=SITE)
    return _fmt("fzp_format_variant_pos", _ptr(sites), C.c_int64(len(sites)))


def format_variant_map(sites, vmap_qid):
    sites = np.ascontiguousarray(sites, dtype=SITE)
    vmap_qid = np.ascontiguousarray(vmap_qid, dtype=np.int32)
    return _fmt("fzp_format_variant_map", _ptr(sites), C.c_int64(len(sites)), _ptr(vmap_qid))


def format_q_id_map(aln: AlnSet):
    return _fmt("fzp_format_q_id_map", C.c_void_p(aln._p))


def format_atable(sites, arows):
    sites = np.ascontiguousarray(sites, dtype=SITE)
    arows = np.ascontiguousarray(arows, dtype=AROW)
    return _fmt("fzp_format_atable", _ptr(sites), _ptr(arows), C.c_int64(len(arows)))


def format_phased_variants(sites, pvars):
    sites = np.ascontiguousarray(sites, dtype=SITE)
    pvars = np.ascontiguousarray(pvars, dtype=PVAR)
    return _fmt("fzp_format_phased_variants", _ptr(sites), _ptr(pvars), C.c_int64(len(pvars)))


def format_phased_reads(preads, ctg_id: str, qname_off, qnames: bytes):
    preads = np.ascontiguousarray(preads, dtype=PREAD)
    qname_off = np.ascontiguousarray(qname_off, dtype=np.int64)
    return _fmt("fzp_format_phased_reads", _ptr(preads), C.c_int64(len(preads)), ctg_id.encode(), _ptr(qname_off), qnames,
                C.c_int32(len(qname_off) - 1))


def readmap(phased_reads: bytes, rawread_ids: bytes, pread_ids: bytes, pread_to_contigs: bytes, ctg_id: str, ctg_index=0):
    """-> (records ndarray[R2P], rid_to_phase text)"""
    lib = load()
    rp, nr, tp, tn = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_size_t()
    _check(lib.fzp_readmap(phased_reads, C.c_size_t(len(phased_reads)), rawread_ids, C.c_size_t(len(rawread_ids)), pread_ids,
                           C.c_size_t(len(pread_ids)), pread_to_contigs, C.c_size_t(len(pread_to_contigs)), ctg_id.encode(),
                           C.c_int32(ctg_index), C.byref(rp), C.byref(nr), C.byref(tp), C.byref(tn)))
    return _take(rp.value, nr.value, R2P), _take_text(tp, tn)


def format_bam(aln: AlnSet, ctg_id: str, ctg_len: int, flags=None, with_index=True):
    """Coordinate-sorted BAM of an alnset (+ its .bai) -> (bam bytes, bai bytes or None)."""
    lib = load()
    fl = None if flags is None else np.ascontiguousarray(flags, dtype=np.int32)
    b, nb, x, nx = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
    _check(lib.fzp_format_bam(C.c_void_p(aln._p), ctg_id.encode(), C.c_int64(int(ctg_len)), None if fl is None else _ptr(fl), C.byref(b), C.byref(nb),
                              C.byref(x) if with_index else None, C.byref(nx) if with_index else None))
    bam = _take_text(b, nb)
    return bam, (_take_text(x, nx) if with_index else None)


def bam_to_sam(bam: bytes, region=None):
    """`samtools view <bam> [region]` without samtools: SAM text lines (11 mandatory columns)."""
    return _fmt("fzp_bam_to_sam", bam, C.c_size_t(len(bam)), None if region is None else region.encode())


# ---------------------------------------------------------------------------- overlap filter (fzp_ovl_*)
class BamViewStruct(C.Structure):
    _fields_ = [("header_text", C.c_void_p), ("header_len", C.c_size_t), ("n_ref", C.c_int32), ("ref_block", C.c_void_p), ("ref_block_len", C.c_size_t),
                ("n_rec", C.c_int64), ("rec_off", C.c_void_p), ("records", C.c_void_p), ("records_len", C.c_size_t), ("name_off", C.c_void_p), ("names", C.c_void_p)]


class BamView:
    """Every record of a BAM file as raw bytes + read names (fzp_bam_open)."""

    def __init__(self, bam: bytes):
        p = C.c_void_p()
        _check(load().fzp_bam_open(bam, len(bam), C.byref(p)))
        v = C.cast(p, C.POINTER(BamViewStruct)).contents
        self.header = C.string_at(v.header_text, v.header_len)
        self.n_ref = int(v.n_ref)
        self.ref_block = C.string_at(v.ref_block, v.ref_block_len)
        n = int(v.n_rec)
        self.rec_off = np.frombuffer(C.string_at(v.rec_off, (n + 1) * 8), dtype=np.int64).copy()
        self.records = C.string_at(v.records, v.records_len)
        noff = np.frombuffer(C.string_at(v.name_off, (n + 1) * 8), dtype=np.int64)
        names = C.string_at(v.names, int(noff[-1]))
        self.names = [names[noff[i]:noff[i + 1]] for i in range(n)]
        load().fzp_bam_view_free(p)

    def record(self, i) -> bytes:
        return self.records[self.rec_off[i]:self.rec_off[i + 1]]


def bam_write(header: bytes, n_ref: int, ref_block: bytes, parts) -> bytes:
    """BAM file bytes: header + whole raw records (each part a run of records with their block_size prefixes)"""
    n = len(parts)
    arr = (C.c_char_p * max(1, n))(*parts)
    lens = (C.c_size_t * max(1, n))(*[len(p) for p in parts])
    out, ol = C.c_void_p(), C.c_size_t()
    _check(load().fzp_bam_write(header, len(header), n_ref, ref_block, len(ref_block), n, arr, lens, C.byref(out), C.byref(ol)))
    return _take_text(out, ol)


def bam_read_header(path: str):
    """(header text, n_ref, reference block) of a BAM file; only its first BGZF blocks are decoded (fzp_bam_read_header)"""
    tp, tl, rp, rl, nr = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t(), C.c_int32()
    _check(load().fzp_bam_read_header(os.fsencode(path), C.byref(tp), C.byref(tl), C.cast(C.byref(nr), C.c_void_p), C.byref(rp), C.byref(rl)))
    return _take_text(tp, tl), int(nr.value), _take_text(rp, rl)


def bam_route(in_paths, names, name_dest, dest_paths, header: bytes, n_ref: int, ref_block: bytes):
    """fzp_bam_route: stream the records of `in_paths` (in order) into `dest_paths[name_dest[i]]` by read name `names[i]` (bytes); bounded
    memory whatever the input sizes.  -> (records written per destination, destinations in order of first use)"""
    n_in, n_dest, n_names = len(in_paths), len(dest_paths), len(names)
    ins = (C.c_char_p * max(1, n_in))(*[os.fsencode(p) for p in in_paths])
    outs = (C.c_char_p * max(1, n_dest))(*[os.fsencode(p) for p in dest_paths])
    off = np.zeros(n_names + 1, np.int64)
    off[1:] = np.cumsum([len(x) for x in names])
    blob = b"".join(names)
    dest = np.ascontiguousarray(name_dest, dtype=np.int32)
    cnt = np.zeros(max(1, n_dest), np.int64)
    first = np.full(max(1, n_dest), -1, np.int32)
    _check(load().fzp_bam_route(n_in, ins, n_names, _ptr(off), blob, _ptr(dest) if n_names else None, n_dest, outs, header, len(header), n_ref, ref_block, len(ref_block),
                                _ptr(cnt), _ptr(first)))
    return cnt[:n_dest], [int(d) for d in first[:n_dest] if d >= 0]


class OvlSet:
    """Tokenised `LA4Falcon -mo` dumps + rid_to_phase.all (fzp_ovl_parse)."""

    def __init__(self, eng, files, rid_map: bytes):
        lib = load()
        n = len(files)
        # the library BORROWS the dumps (no copy on either side): they stay referenced here until the set is closed
        self._keep = [f if isinstance(f, bytes) else bytes(f) for f in files]
        texts = (C.c_char_p * max(1, n))(*self._keep)
        lens = (C.c_size_t * max(1, n))(*[len(f) for f in self._keep])
        p = C.c_void_p()
        _check(lib.fzp_ovl_parse(eng._p, n, texts, lens, rid_map, len(rid_map), C.byref(p)))
        self._p = p.value

    def close(self):
        if getattr(self, "_p", None):
            load().fzp_ovlset_free(self._p)
            self._p = None
        self._keep = None

    __del__ = close

    @property
    def n_lines(self):
        return int(load().fzp_ovl_n_lines(self._p))

    @property
    def n_rows(self):
        return int(load().fzp_ovl_n_rows(self._p))

    def id_name(self, idx):
        p, n = C.c_void_p(), C.c_int32()
        _check(load().fzp_ovl_id_name(self._p, int(idx), C.byref(p), C.byref(n)))
        return C.string_at(p, n.value)

    def format(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        return _fmt("fzp_ovl_format", C.c_void_p(self._p), _ptr(rows), C.c_int64(len(rows)))


def ovl_filter(eng, ovl: OvlSet, max_diff, max_cov, min_cov, min_len=2500, bestn=10):
    """filter_stage1..3 on the device -> (selected line numbers in print order, ignore ids, contained ids)"""
    lib = load()
    P = OvlpParams(int(max_diff), int(max_cov), int(min_cov), int(min_len), int(bestn))
    r, nr, ig, nig, ct, nct = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64()
    _check(lib.fzp_ovl_filter(eng._p, ovl._p, C.byref(P), C.byref(r), C.byref(nr), C.byref(ig), C.byref(nig), C.byref(ct), C.byref(nct)))
    return _take(r, nr.value, np.int64), _take(ig, nig.value, np.int32), _take(ct, nct.value, np.int32)


def track_reads(eng, files, phased_reads: bytes, read_to_contig_map: bytes, rawread_ids: bytes, min_len=2500, bestn=40):
    """rr_hctg_track.py's run_track_reads on the device -> the rawread_to_contigs text (canonical line order)."""
    n = len(files)
    keep = [f if isinstance(f, bytes) else bytes(f) for f in files]          # lent for the duration of the call
    texts = (C.c_char_p * max(1, n))(*keep)
    lens = (C.c_size_t * max(1, n))(*[len(f) for f in keep])
    return _fmt("fzp_track_reads", eng._p, C.c_int32(n), texts, lens, phased_reads, C.c_size_t(len(phased_reads)), read_to_contig_map,
                C.c_size_t(len(read_to_contig_map)), rawread_ids, C.c_size_t(len(rawread_ids)), C.c_int64(int(min_len)), C.c_int64(int(bestn)))
