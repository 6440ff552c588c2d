#!/usr/bin/env python3
"""Golden vectors for select_reads_from_bam: RUN the reference's falcon_unzip/select_reads_from_bam.py (from /root/reference, this
container only; lib2to3 in memory) with a pysam stand-in -- a small pure-Python BAM reader / writer, independent of this repo's
library -- on synthetic subread BAMs, and store inputs + what every output BAM holds (header text, record names, sha256 of the raw
record bytes) as data under tests/golden_select/<case>/.  Header serialisation of the stand-in: record types in SAM order
(@HD, @SQ, @RG, @PG, @CO), fields in stored order.  usage: make_golden_select.py"""
import gzip
import hashlib
import io
import json
import os
import shutil
import struct
import sys
import tempfile
import types
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/falcon_unzip/select_reads_from_bam.py"


# ---------------------------------------------------------------- minimal BAM i/o (stand-in for pysam)
def bgzf_write(raw: bytes) -> bytes:
    out = io.BytesIO()
    for i in range(0, max(len(raw), 1), 0xff00):
        chunk = raw[i:i + 0xff00]
        if not chunk:
            break
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(chunk) + c.flush()
        out.write(struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 66, 67, 2, len(comp) + 25))
        out.write(comp)
        out.write(struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
    out.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return out.getvalue()


def bgzf_read(data: bytes) -> bytes:
    out, p = [], 0
    while p < len(data):
        xlen = struct.unpack_from("<H", data, p + 10)[0]
        bsize = struct.unpack_from("<H", data, p + 16)[0] + 1
        out.append(zlib.decompress(data[p + 12 + xlen:p + bsize - 8], -15))
        p += bsize
    return b"".join(out)


def parse_header(text):
    h = {}
    for line in text.split("\n"):
        if not line:
            continue
        typ, fields = line[1:3], line.split("\t")[1:]
        if typ == "CO":
            h.setdefault("CO", []).append("\t".join(fields))
            continue
        rec = dict(f.split(":", 1) for f in fields)
        if typ == "HD":
            h["HD"] = rec
        else:
            h.setdefault(typ, []).append(rec)
    return h


def format_header(h):
    lines = []
    for typ in ("HD", "SQ", "RG", "PG", "CO"):
        if typ not in h:
            continue
        if typ == "HD":
            lines.append("@HD\t" + "\t".join("%s:%s" % kv for kv in h["HD"].items()))
        elif typ == "CO":
            lines += ["@CO\t" + c for c in h["CO"]]
        else:
            lines += ["@%s\t" % typ + "\t".join("%s:%s" % kv for kv in r.items()) for r in h[typ]]
    return "".join(l + "\n" for l in lines)


class Rec:
    def __init__(self, raw):
        self.raw = raw
        l_name = raw[4 + 8]
        self.query_name = raw[4 + 32:4 + 32 + l_name - 1].decode()


class AlignmentFile:
    def __init__(self, fn, mode, check_sq=True, header=None):
        self.fn, self.mode = fn, mode
        if mode == "rb":
            d = bgzf_read(open(fn, "rb").read())
            assert d[:4] == b"BAM\1"
            l_text = struct.unpack_from("<i", d, 4)[0]
            self.header = parse_header(d[8:8 + l_text].rstrip(b"\0").decode())
            o = 8 + l_text
            n_ref = struct.unpack_from("<i", d, o)[0]
            o += 4
            r0 = o
            for _ in range(n_ref):
                ln = struct.unpack_from("<i", d, o)[0]
                o += 4 + ln + 4
            self.refs = (n_ref, d[r0:o])
            self.recs = []
            while o < len(d):
                bs = struct.unpack_from("<i", d, o)[0]
                self.recs.append(Rec(d[o:o + 4 + bs]))
                o += 4 + bs
        else:
            self.header, self.out = header, []

    def fetch(self, until_eof=False):
        return iter(self.recs)

    def write(self, r):
        self.out.append(r.raw)

    def close(self):
        if self.mode == "wb":
            text = format_header(self.header).encode()
            raw = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 0) + b"".join(self.out)
            with open(self.fn, "wb") as f:
                f.write(bgzf_write(raw))


def make_record(name, seq, rg):
    nm = name.encode() + b"\0"
    packed = bytearray((len(seq) + 1) // 2)
    for i, c in enumerate(seq):
        packed[i // 2] |= "=ACMGRSVTWYHKDBN".index(c) << (4 if i % 2 == 0 else 0)
    tags = b"RGZ" + rg.encode() + b"\0" + b"zmi" + struct.pack("<i", int(name.split("/")[1]))
    body = struct.pack("<iiBBHHHiiii", -1, -1, len(nm), 255, 4680, 0, 4, len(seq), -1, -1, 0) + nm + bytes(packed) + b"\xff" * len(seq) + tags
    return struct.pack("<i", len(body)) + body


def make_inputs(work, seed):
    import random
    rnd = random.Random(seed)
    names = ["m%d_c100/%d/0_%d" % (i % 2, 1000 + i, 400 + 10 * (i % 17)) for i in range(130)]
    rnd.shuffle(names)
    with open(os.path.join(work, "rawread_ids"), "w") as f:
        f.write("\n".join(names) + "\n")
    plan = [("000000F", range(0, 45)), ("000001F", range(45, 75)), ("000002F", range(75, 90)), ("000000F_001", range(90, 115))]
    rows = []
    for ctg, rng_ in plan:
        for r in rng_:
            rows.append("%09d %s 5 0 %d 1" % (r, ctg, -900 + 7 * (r % 13)))
            if r % 9 == 0:
                rows.append("%09d %s 3 1 -100 1" % (r, "000001F"))                # second-best hit: ignored
            if r % 11 == 0:
                rows.append("%09d %s 5 0 %d 1" % (r, "000001F", -905 + (r % 5)))   # a second rank-0 row: lower score wins, then the name
    rows += ["%09d NA 0 0 0 0" % r for r in range(115, 120)]
    rnd.shuffle(rows)
    with open(os.path.join(work, "rawread_to_contigs"), "w") as f:
        f.write("\n".join(rows) + "\n")
    bams = []
    for b in range(2):
        hdr = "@HD\tVN:1.5\tSO:unknown\tpb:3.0.1\n@RG\tID:rg%d\tPL:PACBIO\tDS:READTYPE=SUBREAD\tPU:m%d_c100\n@PG\tID:baz2bam\tPN:baz2bam\tVN:3.0\n@CO\tmovie %d\n" % (b, b, b)
        recs = []
        for i in rnd.sample(range(130), 130):
            if i % 2 == b and i < 126:                                        # reads 126..129 are in no BAM
                recs.append(make_record(names[i], "".join(rnd.choice("ACGT") for _ in range(40 + i % 23)), "rg%d" % b))
        recs.append(make_record("m%d_c100/9999/0_50" % b, "ACGT" * 5, "rg%d" % b))       # a read the map does not know
        text = hdr.encode()
        raw = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 0) + b"".join(recs)
        fn = os.path.join(work, "movie%d.subreads.bam" % b)
        with open(fn, "wb") as f:
            f.write(bgzf_write(raw))
        bams.append(os.path.basename(fn))
    with open(os.path.join(work, "input_bam.fofn"), "w") as f:
        f.write("\n".join(bams) + "\n")


def load_reference():
    from lib2to3 import refactor
    src = open(REF).read()
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
    py3 = str(tool.refactor_string(src + "\n", REF))
    pysam = types.ModuleType("pysam")
    pysam.AlignmentFile = AlignmentFile
    saved = sys.modules.get("pysam")
    sys.modules["pysam"] = pysam
    mod = types.ModuleType("ref_select_reads_from_bam")
    try:
        exec(compile(py3, REF, "exec"), mod.__dict__)
    finally:
        if saved is None:
            sys.modules.pop("pysam", None)
        else:
            sys.modules["pysam"] = saved
    return mod


if __name__ == "__main__":
    mod = load_reference()
    for case, seed in (("s1", 41), ("s2", 42)):
        work = tempfile.mkdtemp(prefix="select_")
        make_inputs(work, seed)
        os.makedirs(os.path.join(work, "out"))
        mod.select_reads_from_bam(os.path.join(work, "input_bam.fofn"), os.path.join(work, "rawread_to_contigs"), os.path.join(work, "rawread_ids"), os.path.join(work, "out"))
        out = os.path.join(HERE, case)
        shutil.rmtree(out, ignore_errors=True)
        os.makedirs(out)
        for fn in ("input_bam.fofn", "rawread_to_contigs", "rawread_ids", "movie0.subreads.bam", "movie1.subreads.bam"):
            shutil.copy(os.path.join(work, fn), os.path.join(out, fn))
        exp = {}
        for fn in sorted(os.listdir(os.path.join(work, "out"))):
            a = AlignmentFile(os.path.join(work, "out", fn), "rb", check_sq=False)
            exp[fn] = {"header": format_header(a.header), "names": [r.query_name for r in a.recs], "records_sha256": hashlib.sha256(b"".join(r.raw for r in a.recs)).hexdigest()}
        with open(os.path.join(out, "expected.json"), "w") as f:
            json.dump(exp, f, indent=1, sort_keys=True)
        print(case, {k: len(v["names"]) for k, v in exp.items()})
        shutil.rmtree(work, ignore_errors=True)
