"""Drop-in for falcon_unzip/phasing_readmap.py (reference lines 8-73): same CLI, same output file
`rid_to_phase.<ctg>`.  Pure host bookkeeping (three id tables and a dict), done in libfzphase's
host code (`fzp_readmap`); rows are written in ascending pread-id order (the reference's order is
Python-2 dict order, i.e. unspecified; every consumer keys by id)."""
from __future__ import annotations

import argparse
import logging
import os
import sys

from . import _lib


def readmap_records(phased_reads, read_map_dir, the_ctg_id, ctg_index=0):
    """-> (records ndarray[_lib.R2P], text) for one contig (phasing_readmap.py:15-46)."""
    def slurp(p):
        with open(p, "rb") as f:
            return f.read()
    rawread_ids = slurp(os.path.join(read_map_dir, 'dump_rawread_ids', 'rawread_ids'))
    pread_ids = slurp(os.path.join(read_map_dir, 'dump_pread_ids', 'pread_ids'))
    p2c = slurp(os.path.join(read_map_dir, 'pread_to_contigs'))
    return _lib.readmap(slurp(phased_reads), rawread_ids, pread_ids, p2c, the_ctg_id, ctg_index)


def get_phasing_readmap(args):
    recs, text = readmap_records(args.phased_reads, args.read_map_dir, args.ctg_id)
    with open(os.path.join(args.base_dir, 'rid_to_phase.%s' % args.ctg_id), 'wb') as f:
        f.write(text)
    return recs


def parse_args(argv):
    parser = argparse.ArgumentParser(description='mapping internal daligner read id to phase block and phase',
                                     formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument('--phased_reads', type=str, help='path to read vs. phase map', required=True)
    parser.add_argument('--read_map_dir', type=str, help='path to the read map directory', required=True)
    parser.add_argument('--ctg_id', type=str, help='contig identifier in the bam file', required=True)
    parser.add_argument('--base_dir', type=str, default="./", help='the output base_dir, default to current working directory')
    args = parser.parse_args(argv[1:])
    return args


def main(argv=sys.argv):
    logging.basicConfig()
    args = parse_args(argv)
    get_phasing_readmap(args)
