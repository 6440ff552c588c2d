"""Route PacBio subreads into one `<ctg>.bam` per contig for polishing -- the job of `falcon_unzip/select_reads_from_bam.py` (SURVEY 8f
row n4, second half), same command line and same files, on the library's streaming BAM code instead of pysam.

What the reference's function decides (line numbers of falcon_unzip/select_reads_from_bam.py), restated as three tables:

  votes      contig -> distinct read names that have a rank-0 row for it in rawread_to_contigs; 'NA' rows and rows of rank >= 1 do not
             count (19-24).  A contig gets a file only if it has MORE than 20 such reads (62-67).
  home       read name -> the contig of its smallest (score, contig name) pair among its rank-0 rows (the sorted list's head, 77-79).
  route      a record goes to home[name] if that contig passed the vote, else nowhere (73-87).

Records keep their input order: files in FOFN order (paths relative to the FOFN's directory, 36-42), records in file order.  Every
output carries the first input's header without its @PG lines, with the later inputs' @RG lines added (44-55).

The routing itself is `fzp_bam_route` (csrc/fzp_bam.hip): inputs are read one BGZF block at a time and records are appended to their
destination as its blocks fill, so the memory needed does not grow with the size of the subreads BAMs (the reference streams through pysam;
real inputs are tens of GB)."""
from __future__ import annotations

import argparse
import os
import sys

from . import _lib

MIN_READS_EXCLUSIVE = 20      # a contig needs more than this many distinct reads (62-67)


def _say(label, *values, to=None):
    # the reference's progress lines, value for value (they are its only stdout / stderr output)
    print(label, *values, file=to or sys.stdout)


def _tables(map_path, ids_path):
    """-> (votes, home) from rawread_to_contigs + rawread_ids; a row is `rid contig <n> rank score ...`, rid indexes the lines of rawread_ids"""
    with open(ids_path) as fh:
        name_of = fh.read().split("\n")
    votes, home = {}, {}
    with open(map_path) as fh:
        for line in fh:
            col = line.split()
            if int(col[3]) > 0:                 # only a read's best hit counts
                continue
            contig = col[1]
            if contig == "NA":
                continue
            name = name_of[int(col[0])]
            votes.setdefault(contig, set()).add(name)
            cand = (int(col[4]), contig)
            if name not in home or cand < home[name]:
                home[name] = cand
    return votes, home


def _fofn_entries(fofn_path):
    """the BAM paths of a FOFN; relative ones are taken from the FOFN's own directory, not from cwd"""
    root = os.path.normpath(os.path.dirname(fofn_path))
    with open(fofn_path) as fh:
        lines = [l.strip() for l in fh]
    return [l if os.path.isabs(l) else os.path.join(root, l) for l in lines]


def _merged_header(headers):
    """first header minus its @PG lines; the other files' @RG lines go right behind its own @RG lines (at the end if it has none)"""
    if not headers:
        return b""
    own = [l for l in headers[0].decode("latin-1").split("\n") if l and not l.startswith("@PG")]
    borrowed = [l for h in headers[1:] for l in h.decode("latin-1").split("\n") if l.startswith("@RG")]
    cut = 1 + max((i for i, l in enumerate(own) if l.startswith("@RG")), default=len(own) - 1)
    lines = own[:cut] + borrowed + own[cut:]
    return "".join(l + "\n" for l in lines).encode("latin-1")


def select_reads_from_bam(input_bam_fofn_fn, rawread_to_contigs_fn, rawread_ids_fn, sam_dir):
    """<sam_dir>/<ctg>.bam for every contig with enough reads; -> sorted list of the contigs written"""
    _say("rawread_ids_fn:", repr(rawread_ids_fn))
    _say("rawread_to_contigs_fn:", repr(rawread_to_contigs_fn))
    votes, home = _tables(rawread_to_contigs_fn, rawread_ids_fn)
    _say("num read_partitions:", len(votes))
    _say("num read_to_ctgs:", len(home))

    bams = _fofn_entries(input_bam_fofn_fn)
    heads = [_lib.bam_read_header(p) for p in bams]            # first blocks only
    header = _merged_header([h[0] for h in heads])

    passed = []
    for contig in sorted(votes):
        _say("ctg, len:", contig, len(votes[contig]))
        if len(votes[contig]) > MIN_READS_EXCLUSIVE:
            passed.append(contig)
    slot = {contig: k for k, contig in enumerate(passed)}
    routed = [(name.encode("latin-1"), slot[pair[1]]) for name, pair in home.items() if pair[1] in slot]
    targets = [os.path.join(sam_dir, "%s.bam" % contig) for contig in passed]
    if not bams:
        return []
    counts, order = _lib.bam_route(bams, [r[0] for r in routed], [r[1] for r in routed], targets, header, heads[0][1], heads[0][2])
    for k in order:
        _say("samfile_fn:{!r}".format(targets[k]), to=sys.stderr)
    return sorted(passed[k] for k in order)


def parse_args(argv):
    ap = argparse.ArgumentParser(description="Split BAM subreads into one BAM per contig (streaming; bounded memory).",
                                 formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    ap.add_argument("--rawread-to-contigs", type=str, default="./2-asm-falcon/read_maps/dump_rawread_ids/rawread_to_contigs",
                    help="read -> contig table written by the raw-read tracker (fc_rr_hctg_track2)")
    ap.add_argument("--rawread-ids", type=str, default="./2-asm-falcon/read_maps/dump_rawread_ids/rawread_ids", help="read names, one per line, line number = read id")
    ap.add_argument("--sam-dir", type=str, default="./4-quiver/reads", help="where the <ctg>.bam files go")
    ap.add_argument("input_bam_fofn", type=str, help="file listing the input BAMs; relative entries are resolved against this file's directory")
    return ap.parse_args(argv[1:])


def main(argv=sys.argv):
    a = parse_args(argv)
    select_reads_from_bam(a.input_bam_fofn, a.rawread_to_contigs, a.rawread_ids, a.sam_dir)
