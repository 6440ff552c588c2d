"""Loading the committed golden cases (tests/golden/<case>/)."""
from __future__ import annotations

import gzip
import hashlib
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def cases():
    return sorted(d for d in os.listdir(GOLDEN) if os.path.isfile(os.path.join(GOLDEN, d, "manifest.json")))


class Case:
    def __init__(self, name):
        self.name = name
        self.dir = os.path.join(GOLDEN, name)
        with open(os.path.join(self.dir, "manifest.json")) as f:
            self.manifest = json.load(f)
        self.ctg_id = self.manifest["ctg_id"]

    def _gz(self, fn):
        with gzip.open(os.path.join(self.dir, fn), "rb") as f:
            return f.read()

    @property
    def sam(self) -> bytes:
        return self._gz("input.sam.gz")

    @property
    def fasta(self) -> bytes:
        return self._gz("ref.fa.gz")

    @property
    def ref_seq(self) -> bytes:
        """What phasing.py:489-494 extracts: record whose first name word == ctg_id, upper-cased."""
        name, chunks, seq = None, [], b""
        for line in self.fasta.split(b"\n"):
            if line.startswith(b">"):
                if name == self.ctg_id.encode():
                    seq = b"".join(chunks).upper()
                name, chunks = (line[1:].split() or [b""])[0], []
            else:
                chunks.append(line)
        if name == self.ctg_id.encode():
            seq = b"".join(chunks).upper()
        return seq

    def has(self, key):
        return key in self.manifest["outputs"]

    def readmap_inputs(self):
        return {k: self._gz(k + ".gz") for k in ("rawread_ids", "pread_ids", "pread_to_contigs")}

    def expected(self, key):
        """-> bytes, or None when the output is pinned by sha256 only."""
        p = os.path.join(self.dir, key)
        if os.path.exists(p):
            with open(p, "rb") as f:
                return f.read()
        return None

    def check(self, key, got: bytes):
        meta = self.manifest["outputs"][key]
        exp = self.expected(key)
        if exp is not None and exp != got:
            el, gl = exp.split(b"\n"), got.split(b"\n")
            for i, (a, b) in enumerate(zip(el, gl)):
                if a != b:
                    raise AssertionError("%s/%s differs at line %d:\n  expected %r\n  got      %r" % (self.name, key, i + 1, a, b))
            raise AssertionError("%s/%s: %d lines expected, %d got" % (self.name, key, len(el) - 1, len(gl) - 1))
        assert len(got) == meta["bytes"], "%s/%s: size %d != %d" % (self.name, key, len(got), meta["bytes"])
        assert hashlib.sha256(got).hexdigest() == meta["sha256"], "%s/%s: sha256 mismatch" % (self.name, key)
