"""ctypes binding of oracle/libfzp_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package (falcon_unzip_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
SO = os.path.join(ORACLE_DIR, "libfzp_oracle.so")


class OracleError(RuntimeError):
    pass


def build(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith(".c")]
    if force or not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libfzp_oracle.so"], stdout=subprocess.DEVNULL)
    return SO


class Oracle:
    def __init__(self, path):
        self.lib = C.CDLL(path)
        self.lib.orc_free.argtypes = [C.c_void_p]

    def _call(self, fname, ins, n_out, extra=()):
        """ins: list of bytes; returns n_out bytes objects."""
        f = getattr(self.lib, fname)
        args = []
        keep = []
        for b in ins:
            buf = C.create_string_buffer(b, len(b)) if len(b) else C.create_string_buffer(1)
            keep.append(buf)
            args += [buf, C.c_size_t(len(b))]
        for e in extra:
            args.append(C.c_char_p(e))
        outs = []
        for _ in range(n_out):
            p, n = C.c_void_p(), C.c_size_t()
            outs.append((p, n))
            args += [C.byref(p), C.byref(n)]
        f.restype = C.c_int
        rc = f(*args)
        if rc != 0:
            raise OracleError("%s failed: rc=%d" % (fname, rc))
        res = []
        for p, n in outs:
            res.append(C.string_at(p, n.value))
            self.lib.orc_free(p)
        return res

    def make_het_call(self, sam: bytes, ref_seq: bytes):
        """-> (variant_pos, variant_map, q_id_map) text"""
        return self._call("orc_make_het_call", [sam, ref_seq], 3)

    def generate_association_table(self, vmap: bytes):
        return self._call("orc_generate_association_table", [vmap], 1)[0]

    def get_phased_blocks(self, vmap: bytes, atable: bytes):
        return self._call("orc_get_phased_blocks", [vmap, atable], 1)[0]

    def get_phased_reads(self, vmap: bytes, qmap: bytes, pv: bytes, ctg_id: str):
        # signature: (vmap, len, qmap, len, pv, len, ctg_id, out, out_len)
        f = self.lib.orc_get_phased_reads
        f.restype = C.c_int
        p, n = C.c_void_p(), C.c_size_t()
        rc = f(vmap, C.c_size_t(len(vmap)), qmap, C.c_size_t(len(qmap)), pv, C.c_size_t(len(pv)),
               ctg_id.encode(), C.byref(p), C.byref(n))
        if rc:
            raise OracleError("orc_get_phased_reads rc=%d" % rc)
        out = C.string_at(p, n.value)
        self.lib.orc_free(p)
        return out

    def phasing_readmap(self, phased_reads: bytes, rawread_ids: bytes, pread_ids: bytes, p2c: bytes, ctg_id: str):
        f = self.lib.orc_phasing_readmap
        f.restype = C.c_int
        p, n = C.c_void_p(), C.c_size_t()
        rc = f(phased_reads, C.c_size_t(len(phased_reads)), rawread_ids, C.c_size_t(len(rawread_ids)),
               pread_ids, C.c_size_t(len(pread_ids)), p2c, C.c_size_t(len(p2c)), ctg_id.encode(),
               C.byref(p), C.byref(n))
        if rc:
            raise OracleError("orc_phasing_readmap rc=%d" % rc)
        out = C.string_at(p, n.value)
        self.lib.orc_free(p)
        return out

    def phase_all(self, sam: bytes, ref_seq: bytes, ctg_id: str):
        """The whole phasing.py chain -> dict of the six texts."""
        vpos, vmap, qmap = self.make_het_call(sam, ref_seq)
        atable = self.generate_association_table(vmap)
        pv = self.get_phased_blocks(vmap, atable)
        pr = self.get_phased_reads(vmap, qmap, pv, ctg_id)
        return {"variant_pos": vpos, "variant_map": vmap, "q_id_map": qmap, "atable": atable,
                "phased_variants": pv, "phased_reads": pr}


ALN_SUMMARY = None


def _aln_dtype():
    global ALN_SUMMARY
    if ALN_SUMMARY is None:
        import numpy as np
        ALN_SUMMARY = np.dtype([("aligned", "<i4"), ("strand", "<i4"), ("pos", "<i4"), ("ref_end", "<i4"), ("q_start", "<i4"),
                                ("q_end", "<i4"), ("score", "<i4"), ("n_cigar", "<i4"), ("cells", "<i8"), ("n_columns", "<i4"),
                                ("n_match", "<i4")], align=True)
        assert ALN_SUMMARY.itemsize == 48
    return ALN_SUMMARY


class AlignParams(C.Structure):
    _fields_ = [("kmer", C.c_int32), ("seed_stride", C.c_int32), ("match", C.c_int32), ("mismatch", C.c_int32),
                ("gap", C.c_int32), ("min_seed_hits", C.c_int32), ("min_pct_identity", C.c_int32), ("seed_anchored", C.c_int32), ("band", C.c_int32), ("reserved", C.c_int32 * 7)]


def align_reads(orc, ctg: bytes, reads, params=None, n_threads=1, seconds=None):
    """CPU twin of K1 (oracle/align_oracle.c): -> (summaries ndarray, list of cigar word arrays).  n_threads > 1: the same
    results from orc_align_reads_mt (reads dealt to host threads).  seconds: a list that receives [index build s, seeding + DP s]."""
    import numpy as np
    lib = orc.lib
    P = AlignParams()
    lib.orc_align_params_default(C.byref(P))
    for k, v in (params or {}).items():
        setattr(P, k, v)
    n = len(reads)
    off = np.zeros(n + 1, np.int64)
    off[1:] = np.cumsum([len(r) for r in reads])
    blob = b"".join(reads)
    out = np.zeros(n, _aln_dtype())
    cig_off = np.zeros(n + 1, np.int64)
    cp = C.c_void_p()
    if n_threads > 1 or seconds is not None:
        f = lib.orc_align_reads_mt_timed
        f.restype = C.c_int
        f.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_void_p]
        sec = (C.c_double * 2)()
        rc = f(ctg, len(ctg), n, off.ctypes.data, blob, C.addressof(P), out.ctypes.data, C.byref(cp), cig_off.ctypes.data, n_threads, C.addressof(sec))
        if seconds is not None:
            seconds[:] = [sec[0], sec[1]]
    else:
        f = lib.orc_align_reads
        f.restype = C.c_int
        f.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]
        rc = f(ctg, len(ctg), n, off.ctypes.data, blob, C.addressof(P), out.ctypes.data, C.byref(cp), cig_off.ctypes.data)
    if rc:
        raise OracleError("orc_align_reads rc=%d" % rc)
    total = int(cig_off[-1])
    words = np.frombuffer(C.string_at(cp.value, total * 4), dtype=np.uint32).copy() if total else np.zeros(0, np.uint32)
    lib.orc_free(cp)
    return out, [words[cig_off[i]:cig_off[i + 1]] for i in range(n)]


def align_origins(orc, ctg: bytes, read: bytes, params=None):
    """candidate origins of one read as the twin's seeding finds them -> list of (strand, i_a, c_a, forward terminal score, backward terminal
    score or -2^26) (orc_align_origins, a test hook)"""
    import numpy as np
    P = AlignParams()
    orc.lib.orc_align_params_default(C.byref(P))
    for k, v in (params or {}).items():
        setattr(P, k, v)
    out = np.zeros(10, np.int64)
    f = orc.lib.orc_align_origins
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p]
    n = f(ctg, len(ctg), read, len(read), C.addressof(P), out.ctypes.data)
    if n < 0:
        raise OracleError("orc_align_origins rc=%d" % n)
    return [tuple(int(x) for x in out[5 * c:5 * c + 5]) for c in range(n)]


def load():
    return Oracle(build())


class OvlpParams(C.Structure):
    _fields_ = [("max_diff", C.c_longlong), ("max_cov", C.c_longlong), ("min_cov", C.c_longlong), ("min_len", C.c_longlong),
                ("bestn", C.c_longlong)]


def ovlp_filter(orc, files, rid_map: bytes, params: dict):
    """oracle/ovlp_oracle.c: orc_ovlp_filter -> (stdout text, ignore ids, contained ids)"""
    lib = orc.lib
    n = len(files)
    bufs = [C.create_string_buffer(f, len(f)) if len(f) else C.create_string_buffer(1) for f in files]
    texts = (C.c_char_p * max(1, n))(*[C.cast(b, C.c_char_p) for b in bufs])
    lens = (C.c_size_t * max(1, n))(*[len(f) for f in files])
    mp = C.create_string_buffer(rid_map, len(rid_map)) if rid_map else C.create_string_buffer(1)
    P = OvlpParams(params["max_diff"], params["max_cov"], params["min_cov"], params["min_len"], params["bestn"])
    outs = [(C.c_void_p(), C.c_size_t()) for _ in range(3)]
    f = lib.orc_ovlp_filter
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t, C.POINTER(OvlpParams)] + [C.c_void_p] * 6
    rc = f(n, texts, lens, C.cast(mp, C.c_char_p), len(rid_map), C.byref(P), *[x for p, q in outs for x in (C.byref(p), C.byref(q))])
    if rc != 0:
        raise OracleError("orc_ovlp_filter failed: rc=%d" % rc)
    res = []
    for p, q in outs:
        res.append(C.string_at(p, q.value))
        lib.orc_free(p)
    return res[0], res[1].decode().split(), res[2].decode().split()


def consensus(orc, sam: bytes, ref_seq: bytes, phased_reads: bytes, phased_variants: bytes, ctg_id: str, version=3):
    """oracle/cns_oracle.c: orc_consensus (fzcns v3, the default) / orc_consensus_v2 / orc_consensus_v1 -> FASTA text of the (block, phase) consensus sequences"""
    return orc._call({1: "orc_consensus_v1", 2: "orc_consensus_v2", 3: "orc_consensus"}[version], [sam, ref_seq, phased_reads, phased_variants], 1, extra=[ctg_id.encode()])[0]


def polish(orc, sam: bytes, tig: bytes):
    """oracle/cns_oracle.c: orc_polish -> the tig called over the pile of the SAM text's accepted records (fzcns v3, the whole tig one block)"""
    return orc._call("orc_polish", [sam, tig], 1)[0]


def track_reads(orc, files, phased_reads: bytes, read_to_contig_map: bytes, rawread_ids: bytes, min_len: int, bestn: int):
    """oracle/track_oracle.c: orc_track_reads -> rawread_to_contigs text in canonical order"""
    lib = orc.lib
    n = len(files)
    bufs = [C.create_string_buffer(f, len(f)) if len(f) else C.create_string_buffer(1) for f in files]
    texts = (C.c_char_p * max(1, n))(*[C.cast(b, C.c_char_p) for b in bufs])
    lens = (C.c_size_t * max(1, n))(*[len(f) for f in files])
    p, q = C.c_void_p(), C.c_size_t()
    f = lib.orc_track_reads
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                  C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p]
    rc = f(n, texts, lens, phased_reads, len(phased_reads), read_to_contig_map, len(read_to_contig_map), rawread_ids, len(rawread_ids), min_len, bestn,
           C.byref(p), C.byref(q))
    if rc != 0:
        raise OracleError("orc_track_reads failed: rc=%d" % rc)
    out = C.string_at(p, q.value)
    lib.orc_free(p)
    return out


def debug_index(orc, ctg: bytes, params=None):
    """oracle/align_oracle.c: orc_debug_index -> the contig's k-mer index as sorted uint64 entries (key << 32 | position << 1 | strand bit), the device table's entry format"""
    import numpy as np
    P = AlignParams()
    orc.lib.orc_align_params_default(C.byref(P))
    for k, v in (params or {}).items():
        setattr(P, k, v)
    p, n = C.c_void_p(), C.c_int64()
    f = orc.lib.orc_debug_index
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    if f(ctg, len(ctg), C.byref(P), C.byref(p), C.byref(n)) != 0:
        raise OracleError("orc_debug_index failed")
    out = np.frombuffer(C.string_at(p, 8 * n.value), np.uint64).copy()
    orc.lib.orc_free(p)
    return out


def debug_hits(orc, ctg: bytes, read: bytes, params=None):
    """oracle/align_oracle.c: orc_debug_hits -> the read's hit list in spec order, uint32 [n, 2]: (strand << 31 | oriented read offset, contig position)"""
    import numpy as np
    P = AlignParams()
    orc.lib.orc_align_params_default(C.byref(P))
    for k, v in (params or {}).items():
        setattr(P, k, v)
    out = np.zeros((4096, 2), np.uint32)
    n = C.c_int64()
    f = orc.lib.orc_debug_hits
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    if f(ctg, len(ctg), read, len(read), C.byref(P), out.ctypes.data_as(C.c_void_p), C.byref(n)) != 0:
        raise OracleError("orc_debug_hits failed")
    return out[:n.value].copy()
