"""Polishing a tig with the reads routed to it: the consensus ROLE of the reference's quiver task (falcon_unzip/run_quiver.py:82-97 -- `pbalign` of <ctg>.bam to
<ctg>_ref.fa, then `variantCaller --algorithm=arrow` -> cns-<ctg>.fasta.gz), on the MI355X engine's own aligner and pile vote (fzp_polish_tigs: K1 + K6 with the tig as
the template).  NOT the quiver WORKFLOW (run_quiver.py:150-384: the pypeflow graph over all tigs, SURVEY row 12 -- out of scope): one call, one or many tigs.

Inputs as the reference's task takes them: `--ref_fasta` = the tig(s) (p_ctg.<ctg>.fa / h_ctg_all.<ctg>.fa as graphs_to_h_tigs.py:406-410,558-562 writes them, or the
per-tig <ctg>_ref.fa of run_quiver.py:208-217), `--read_bam` = the tig's reads (select_reads_from_bam.py's <ctg>.bam; unaligned subreads -- only names and SEQ are
read) or `--read_fasta`.  With several tigs in `--ref_fasta`, `--read_to_tig` (rows `read_name tig_name`) says which read belongs where; with one tig every read is its.
Output: `--cns_fasta` (gzip when the name ends in .gz, as cns-<ctg>.fasta.gz), one record per tig in input order, header `>{tig}|fzcns {n_records}`."""
from __future__ import annotations

import argparse
import gzip
import sys

import numpy as np

from . import _lib


def read_fasta(path):
    """-> [(name, sequence bytes)]: header's first word, lines joined, white space dropped (falcon_kit's FastaReader, as phasing.py:490-494 uses it)"""
    out, name, parts = [], None, []
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        for line in f:
            line = line.strip()
            if line.startswith(b">"):
                if name is not None:
                    out.append((name, b"".join(parts)))
                w = line[1:].split()
                name, parts = (w[0].decode() if w else ""), []
            elif name is not None and line:
                parts.append(line)
    if name is not None:
        out.append((name, b"".join(parts)))
    return out


def reads_of_bam(path):
    """-> [(name, SEQ)] of every record of a BAM file (the library's reader; no samtools)"""
    with open(path, "rb") as f:
        sam = _lib.bam_to_sam(f.read())
    out = []
    for line in sam.split(b"\n"):
        if not line or line.startswith(b"@"):
            continue
        c = line.split(b"\t")
        out.append((c[0].decode(), c[9]))
    return out


def polish(eng, tigs, reads, read_tig=None, params=None):
    """tigs: [(name, seq)]; reads: [(name, seq)]; read_tig: tig index per read (default: all 0).  -> ([(name, polished seq, n_records)], Tigs table)"""
    n = len(reads)
    off = np.zeros(n + 1, np.int64)
    off[1:] = np.cumsum([len(s) for _, s in reads])
    blob = b"".join(s for _, s in reads)
    rt = np.zeros(n, np.int32) if read_tig is None else np.asarray(read_tig, np.int32)
    t = _lib.polish_tigs(eng, [s for _, s in tigs], blob, off, rt, params=params)
    res = [(tigs[i][0], t.sequence(i), int(t.tigs[i]["n_records"])) for i in range(len(tigs))]
    return res, t


def write_fasta(path, records):
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "wb") as f:
        for name, seq, n_rec in records:
            f.write((">%s|fzcns %d\n" % (name, n_rec)).encode())
            f.write(seq)
            f.write(b"\n")


def parse_args(argv):
    ap = argparse.ArgumentParser(description="polish tigs with the reads routed to them (the consensus role of run_quiver.py:82-97 on the MI355X engine)")
    ap.add_argument("--ref_fasta", required=True, help="the tig(s): p_ctg.<ctg>.fa / h_ctg_all.<ctg>.fa / <ctg>_ref.fa")
    ap.add_argument("--read_bam", default=None, help="the tig's reads as select_reads_from_bam.py leaves them (<ctg>.bam)")
    ap.add_argument("--read_fasta", default=None, help="... or as FASTA")
    ap.add_argument("--read_to_tig", default=None, help="rows `read_name tig_name` when --ref_fasta holds several tigs")
    ap.add_argument("--cns_fasta", required=True, help="output (cns-<ctg>.fasta[.gz])")
    ap.add_argument("--device", type=int, default=0)
    return ap.parse_args(argv[1:])


def main(argv=sys.argv):
    args = parse_args(argv)
    if (args.read_bam is None) == (args.read_fasta is None):
        raise SystemExit("give exactly one of --read_bam / --read_fasta")
    tigs = read_fasta(args.ref_fasta)
    if not tigs:
        raise SystemExit("%s: no FASTA record" % args.ref_fasta)
    reads = reads_of_bam(args.read_bam) if args.read_bam else read_fasta(args.read_fasta)
    read_tig = None
    if len(tigs) > 1:
        if not args.read_to_tig:
            raise SystemExit("--ref_fasta holds %d tigs: --read_to_tig has to say which read belongs to which" % len(tigs))
        idx = {name: i for i, (name, _) in enumerate(tigs)}
        home = {}
        with open(args.read_to_tig) as f:
            for line in f:
                w = line.split()
                if len(w) >= 2 and w[1] in idx:
                    home[w[0]] = idx[w[1]]
        keep = [(nm, s) for nm, s in reads if nm in home]
        read_tig = [home[nm] for nm, _ in keep]
        reads = keep
    eng = _lib.Engine(args.device)
    try:
        res, t = polish(eng, tigs, reads, read_tig)
        t.close()
    finally:
        eng.close()
    write_fasta(args.cns_fasta, res)
    for name, seq, n_rec in res:
        print("%s %d -> %d bases, %d reads in the pile" % (name, len(dict(tigs)[name]), len(seq), n_rec))


if __name__ == "__main__":
    main(sys.argv)
