"""Mirror of `falcon_unzip/select_reads_from_bam.py` (second half): route the subreads of the input BAMs into one
`<ctg>.bam` per contig for polishing.  Same function, same flags, same prints; BAM reading / writing goes through the library's
BAM code (csrc/fzp_bam.hip: fzp_bam_open / fzp_bam_write) instead of pysam.

Rules kept from the reference (line numbers of select_reads_from_bam.py): only rank-0 rows of rawread_to_contigs count and 'NA'
contigs are skipped (19-24); a contig is written only if more than 20 distinct reads were assigned to it (62-67); a read that maps to
several contigs goes to the one that sorts first by (score, contig name) (77-79); records keep their input order, files are visited in
FOFN order, paths in the FOFN are relative to the FOFN (37-42); the output header is the first file's header with the other
files' @RG lines appended and every @PG line removed (44-55)."""
from __future__ import annotations

import argparse
import os
import sys

from . import _lib


def _merged_header(headers):
    """first header without @PG lines (51-54), the later files' @RG lines appended after its own @RG lines (48-49)"""
    first = [l for l in headers[0].decode("latin-1").split("\n") if l]
    extra = [l for h in headers[1:] for l in h.decode("latin-1").split("\n") if l.startswith("@RG")]
    out, placed = [], False
    last_rg = max([i for i, l in enumerate(first) if l.startswith("@RG")], default=-1)
    for i, l in enumerate(first):
        if l.startswith("@PG"):
            continue
        out.append(l)
        if i == last_rg:
            out.extend(extra)
            placed = True
    if not placed:
        out.extend(extra)
    return ("\n".join(out) + "\n").encode("latin-1") if out else b""


def select_reads_from_bam(input_bam_fofn_fn, rawread_to_contigs_fn, rawread_ids_fn, sam_dir):
    """Write <ctg>.bam files into sam_dir, for each 'ctg' read in input BAMs."""
    read_partition = {}
    read_to_ctgs = {}
    print("rawread_ids_fn:", repr(rawread_ids_fn))
    print("rawread_to_contigs_fn:", repr(rawread_to_contigs_fn))
    with open(rawread_ids_fn) as f:
        rid_to_oid = f.read().split('\n')
    with open(rawread_to_contigs_fn) as f:
        for row in f:
            row = row.strip().split()
            if int(row[3]) >= 1:          # keep top one hits
                continue
            ctg_id = row[1]
            if ctg_id == 'NA':
                continue
            o_id = rid_to_oid[int(row[0])]
            read_partition.setdefault(ctg_id, set()).add(o_id)
            read_to_ctgs.setdefault(o_id, []).append((int(row[4]), ctg_id))
    print("num read_partitions:", len(read_partition))
    print("num read_to_ctgs:", len(read_to_ctgs))
    fofn_basedir = os.path.normpath(os.path.dirname(input_bam_fofn_fn))

    def abs_fn(maybe_rel_fn):
        return maybe_rel_fn if os.path.isabs(maybe_rel_fn) else os.path.join(fofn_basedir, maybe_rel_fn)
    with open(input_bam_fofn_fn) as f:
        fns = [abs_fn(row.strip()) for row in f]
    views = []
    for fn in fns:
        with open(fn, "rb") as f:
            views.append(_lib.BamView(f.read()))
    header = _merged_header([v.header for v in views]) if views else b""
    selected_ctgs = set()
    for ctg in sorted(read_partition):
        picked_reads = read_partition[ctg]
        print("ctg, len:", ctg, len(picked_reads))
        if len(picked_reads) > 20:
            selected_ctgs.add(ctg)
    parts = {}                                 # ctg -> list of raw records, input order; dict order = first use, like the reference's outfile dict
    for v in views:
        for i, name in enumerate(v.names):
            ctg_list = read_to_ctgs.get(name.decode("latin-1"))
            if ctg_list is None:
                continue
            ctg_list.sort()
            score, ctg = ctg_list[0]
            if ctg not in selected_ctgs:
                continue
            if ctg not in parts:
                print('samfile_fn:{!r}'.format(os.path.join(sam_dir, '%s.bam' % ctg)), file=sys.stderr)
                parts[ctg] = []
            parts[ctg].append(v.record(i))
    for ctg, recs in parts.items():
        data = _lib.bam_write(header, views[0].n_ref, views[0].ref_block, [b"".join(recs)])
        with open(os.path.join(sam_dir, '%s.bam' % ctg), "wb") as f:
            f.write(data)
    return sorted(parts)


def parse_args(argv):
    parser = argparse.ArgumentParser(description='Write ctg.sam files, based on BAM subreads.', formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument('--rawread-to-contigs', type=str, default='./2-asm-falcon/read_maps/dump_rawread_ids/rawread_to_contigs', help='rawread_to_contigs file (from where?)')
    parser.add_argument('--rawread-ids', type=str, default='./2-asm-falcon/read_maps/dump_rawread_ids/rawread_ids', help='rawread_ids file (from where?)')
    parser.add_argument('--sam-dir', type=str, default='./4-quiver/reads', help='Output directory for ctg.sam files')
    parser.add_argument('input_bam_fofn', type=str, help='File of BAM filenames. Paths are relative to dir of FOFN, not CWD.')
    return parser.parse_args(argv[1:])


def main(argv=sys.argv):
    args = parse_args(argv)
    select_reads_from_bam(args.input_bam_fofn, args.rawread_to_contigs, args.rawread_ids, args.sam_dir)
