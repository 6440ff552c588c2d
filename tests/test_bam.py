"""BAM / BGZF / BAI emitter and reader (fzp_format_bam, fzp_bam_to_sam): host code, no GPU needed.
The files are checked with an independent reader written here from the SAM/BAM specification (Python gzip + struct)."""
import gzip
import struct

import numpy as np
import pytest

from tests.golden_util import Case


def _lib():
    from falcon_unzip_amd import _lib
    return _lib


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def reg2bins(beg, end):
    end -= 1
    out = [0]
    for shift, off in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
        out.extend(range(off + (beg >> shift), off + (end >> shift) + 1))
    return out


def bgzf_blocks(b):
    """-> list of (file offset, payload bytes)"""
    out, p = [], 0
    while p < len(b):
        assert b[p:p + 4] == b"\x1f\x8b\x08\x04"
        xlen = struct.unpack_from("<H", b, p + 10)[0]
        assert b[p + 12:p + 14] == b"BC" and struct.unpack_from("<H", b, p + 14)[0] == 2
        bsize = struct.unpack_from("<H", b, p + 16)[0] + 1
        data = gzip.decompress(b[p:p + bsize])
        assert len(data) <= 65536 and struct.unpack_from("<I", b, p + bsize - 4)[0] == len(data)
        out.append((p, data))
        p += bsize
    return out


def parse_bam(stream):
    assert stream[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", stream, 4)[0]
    text = stream[8:8 + l_text]
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", stream, o)[0]; o += 4
    refs = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", stream, o)[0]; o += 4
        name = stream[o:o + ln - 1]; o += ln
        refs.append((name, struct.unpack_from("<i", stream, o)[0])); o += 4
    recs = []
    while o < len(stream):
        start = o
        bs = struct.unpack_from("<i", stream, o)[0]; o += 4
        ref, pos, l_name, mapq, bin_, n_cig, flag, l_seq, nref, npos, tlen = struct.unpack_from("<iiBBHHHiiii", stream, o)
        p = o + 32
        name = stream[p:p + l_name - 1]; p += l_name
        cig = struct.unpack_from("<%dI" % n_cig, stream, p); p += 4 * n_cig
        seq = "".join("=ACMGRSVTWYHKDBN"[(stream[p + k // 2] >> (0 if k & 1 else 4)) & 15] for k in range(l_seq)); p += (l_seq + 1) // 2
        qual = stream[p:p + l_seq]; p += l_seq
        assert p == o + bs
        recs.append(dict(start=start, end=o + bs, ref=ref, pos=pos, mapq=mapq, bin=bin_, flag=flag, name=name, cig=cig, seq=seq, qual=qual))
        o += bs
    return text, refs, recs


@pytest.fixture(scope="module")
def g2():
    c = Case("g2_cfg1_clr")
    L = _lib()
    aln = L.parse_sam(c.sam)
    return c, aln


def test_bam_is_valid_and_round_trips(g2):
    c, aln = g2
    L = _lib()
    flags = (np.arange(aln.n_rec) % 2 * 16).astype(np.int32)
    ctg_len = len(c.ref_seq)
    bam, bai = L.format_bam(aln, c.ctg_id, ctg_len, flags)
    assert bam.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))     # BGZF EOF marker
    blocks = bgzf_blocks(bam)
    stream = b"".join(d for _, d in blocks)
    assert gzip.decompress(bam) == stream                                   # every block is a gzip member
    text, refs, recs = parse_bam(stream)
    assert b"SO:coordinate" in text and refs == [(c.ctg_id.encode(), ctg_len)]
    assert len(recs) == aln.n_rec and all(r["ref"] == 0 and r["mapq"] == 254 for r in recs)
    sam_lines = L.format_sam(aln, c.ctg_id, flags).decode().splitlines()
    for r, line in zip(recs, sam_lines):
        f = line.split("\t")
        assert r["name"].decode() == f[0] and r["flag"] == int(f[1]) and r["pos"] + 1 == int(f[3]) and r["seq"] == f[9]
        cig = "".join("%d%s" % (w >> 4, "MIDNSHP=X"[w & 15]) for w in r["cig"])
        assert cig == f[5]
        rlen = sum(w >> 4 for w in r["cig"] if (w & 15) in (0, 2, 3, 7, 8))
        assert r["bin"] == reg2bin(r["pos"], r["pos"] + rlen)
        assert set(r["qual"]) <= {0xff}
    assert [r["pos"] for r in recs] == sorted(r["pos"] for r in recs)
    # the library's own reader gives back exactly the SAM text (the `samtools view` role), with and without a region
    assert L.bam_to_sam(bam) == L.format_sam(aln, c.ctg_id, flags)
    assert L.bam_to_sam(bam, c.ctg_id) == L.format_sam(aln, c.ctg_id, flags)
    assert L.bam_to_sam(bam, "no_such_contig") == b""
    # and the phasing parser sees the same records
    again = L.parse_sam(L.bam_to_sam(bam, c.ctg_id))
    assert again.n_rec == aln.n_rec and again.n_qid == aln.n_qid


def test_bai_finds_every_overlapping_record(g2):
    c, aln = g2
    L = _lib()
    bam, bai = L.format_bam(aln, c.ctg_id, len(c.ref_seq))
    blocks = bgzf_blocks(bam)
    starts = {}
    acc = 0
    stream = b""
    for off, d in blocks:
        starts[off] = len(stream)
        stream += d
    _, _, recs = parse_bam(stream)

    def upos(v):            # virtual offset -> position in the uncompressed stream
        return starts[v >> 16] + (v & 0xffff) if (v >> 16) in starts else len(stream)
    assert bai[:4] == b"BAI\x01" and struct.unpack_from("<i", bai, 4)[0] == 1
    o = 8
    n_bin = struct.unpack_from("<i", bai, o)[0]; o += 4
    bins = {}
    for _ in range(n_bin):
        b, n_chunk = struct.unpack_from("<Ii", bai, o); o += 8
        bins[b] = [struct.unpack_from("<QQ", bai, o + 16 * k) for k in range(n_chunk)]; o += 16 * n_chunk
    n_intv = struct.unpack_from("<i", bai, o)[0]; o += 4
    linear = struct.unpack_from("<%dQ" % n_intv, bai, o); o += 8 * n_intv
    assert o + 8 == len(bai)
    assert bins[37450][1] == (len(recs), 0)                       # samtools' metadata pseudo-bin: mapped, unmapped
    span = lambda r: (r["pos"], r["pos"] + max(1, sum(w >> 4 for w in r["cig"] if (w & 15) in (0, 2, 3, 7, 8))))
    rng = np.random.default_rng(5)
    L_ctg = len(c.ref_seq)
    for _ in range(60):
        beg = int(rng.integers(0, L_ctg - 1)); end = int(min(L_ctg, beg + rng.integers(1, 20000)))
        want = [r["start"] for r in recs if span(r)[0] < end and span(r)[1] > beg]
        min_off = linear[beg >> 14] if (beg >> 14) < len(linear) else 0
        got = set()
        for b in reg2bins(beg, end):
            for cb, ce in bins.get(b, []) if b != 37450 else []:
                if ce <= min_off:
                    continue
                lo, hi = upos(cb), upos(ce)
                for r in recs:
                    if lo <= r["start"] < hi and span(r)[0] < end and span(r)[1] > beg:
                        got.add(r["start"])
        assert sorted(got) == want
    for r in recs:                                                 # every record sits in a chunk of its own bin
        assert any(upos(cb) <= r["start"] < upos(ce) for cb, ce in bins[r["bin"]])


def test_reader_rejects_garbage():
    L = _lib()
    with pytest.raises(L.FzpError):
        L.bam_to_sam(b"not a bam file at all, really")
    with pytest.raises(L.FzpError):
        L.bam_to_sam(gzip.compress(b"BAM\x01"))                    # gzip, but not BGZF


def test_more_than_65535_cigar_ops_round_trip():
    """A read of > 100 kb can have more CIGAR ops than the 16-bit field holds: the writer moves the CIGAR into a CG:B,I tag (SAMv1 4.2.2) and the
    reader puts it back (this used to be EINVAL and aborted the pipeline)."""
    L = _lib()
    n_ops = 70001
    cigar = "".join("1=" if k % 2 == 0 else "1X" for k in range(n_ops))
    seq = "ACGT" * (n_ops // 4) + "A" * (n_ops % 4)
    sam = ("big/1/0_%d\t0\tctg\t5\t254\t%s\t*\t0\t0\t%s\t*\n" % (n_ops, cigar, seq)).encode()
    aln = L.parse_sam(sam)
    bam, bai = L.format_bam(aln, "ctg", n_ops + 100)
    back = L.bam_to_sam(bam, "ctg")
    f = back.split(b"\t")
    assert f[5].decode() == cigar and f[9].decode() == seq and f[3] == b"5"
    # the record itself, read independently: a 2-op placeholder CIGAR (<l_seq>S<ref span>N) and the real ops in the CG tag
    raw = b"".join(d for _, d in bgzf_blocks(bam))
    o = 8 + struct.unpack_from("<i", raw, 4)[0]
    n_ref = struct.unpack_from("<i", raw, o)[0]; o += 4
    for _ in range(n_ref):
        o += 4 + struct.unpack_from("<i", raw, o)[0] + 4
    bs = struct.unpack_from("<i", raw, o)[0]; o += 4
    ref, pos, l_name, mapq, bin_, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", raw, o)
    assert n_cig == 2 and l_seq == n_ops
    c0 = o + 32 + l_name
    w0, w1 = struct.unpack_from("<II", raw, c0)
    assert (w0 & 15, w0 >> 4) == (4, n_ops) and (w1 & 15, w1 >> 4) == (3, n_ops)
    t0 = c0 + 8 + (l_seq + 1) // 2 + l_seq
    assert raw[t0:t0 + 4] == b"CGBI" and struct.unpack_from("<i", raw, t0 + 4)[0] == n_ops and t0 + 8 + 4 * n_ops == o + bs
