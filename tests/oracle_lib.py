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


def load():
    return Oracle(build())
