"""Drop-in for falcon_unzip/phasing.py: same task functions, same CLI, same files -- the bodies run
on the MI355X through libfzphase.so (HIP kernels, include/fzphase.h).  There is no CPU path here.

Mirrors (reference file:line):
  make_het_call                falcon_unzip/phasing.py:14-135
  generate_association_table   falcon_unzip/phasing.py:137-206
  get_phased_blocks            falcon_unzip/phasing.py:216-421
  get_phased_reads             falcon_unzip/phasing.py:423-480
  phasing / parse_args / main  falcon_unzip/phasing.py:482-575

Each task reads and writes exactly the files its reference twin does; when the four tasks run in one
process (the normal case, phasing.py:496-553) the records produced by one task are also handed to
the next in memory, so the text files are written once and never re-parsed.
"""
from __future__ import annotations

import argparse
import logging
import os
import shlex
import subprocess
import sys

try:  # the reference's workflow engine, when it is installed
    from pypeflow.simple_pwatcher_bridge import (PypeProcWatcherWorkflow, MyFakePypeThreadTaskBase,  # noqa: F401
                                                 makePypeLocalFile, fn, PypeTask)
except ImportError:  # in-process stand-in with the same call surface
    from ._miniflow import (PypeProcWatcherWorkflow, MyFakePypeThreadTaskBase,  # noqa: F401
                            makePypeLocalFile, fn, PypeTask)

from . import _lib, textio

_ENGINE = None
_HANDOFF = {}   # path -> records produced in this process (path is still written to disk)


def engine():
    """The process-wide fzp_ctx; device from FZP_DEVICE, else LOCAL_RANK, else 0."""
    global _ENGINE
    if _ENGINE is None:
        dev = int(os.environ.get("FZP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _ENGINE = _lib.Engine(dev)
    return _ENGINE


def _write(path, data: bytes):
    with open(path, "wb") as f:
        f.write(data)


def _read(path) -> bytes:
    with open(path, "rb") as f:
        return f.read()


def _key(path):
    return os.path.abspath(path)


def make_het_call(self):
    bam_fn = fn(self.bam_file)
    ctg_id = self.parameters["ctg_id"]
    ref_seq = self.parameters["ref_seq"]
    base_dir = self.parameters["base_dir"]
    samtools = self.parameters["samtools"]
    vmap_fn = fn(self.vmap_file)
    vpos_fn = fn(self.vpos_file)
    q_id_map_fn = fn(self.q_id_map_file)

    if samtools == "builtin":
        # no samtools on the node: the library's own BGZF/BAM reader plays `samtools view <bam> <ctg>`
        with open(bam_fn, "rb") as f:
            sam = _lib.bam_to_sam(f.read(), ctg_id)
    else:
        p = subprocess.Popen(shlex.split("%s view %s %s" % (samtools, bam_fn, ctg_id)), stdout=subprocess.PIPE)
        sam, _ = p.communicate()

    try:
        os.makedirs("%s/%s" % (base_dir, ctg_id))
    except OSError:
        pass

    aln = _lib.parse_sam(sam)
    ref = ref_seq.encode() if isinstance(ref_seq, str) else ref_seq
    sites, vmap_qid = engine().het_call(aln, ref)
    _write(vpos_fn, _lib.format_variant_pos(sites))
    _write(vmap_fn, _lib.format_variant_map(sites, vmap_qid))
    _write(q_id_map_fn, _lib.format_q_id_map(aln))
    _HANDOFF[_key(vmap_fn)] = (sites, vmap_qid)
    _HANDOFF[_key(q_id_map_fn)] = aln.qname_table()


def _load_vmap(vmap_fn):
    got = _HANDOFF.get(_key(vmap_fn))
    if got is None:
        got = textio.parse_variant_map(_read(vmap_fn))
        _HANDOFF[_key(vmap_fn)] = got
    return got


def generate_association_table(self):
    vmap_fn = fn(self.vmap_file)
    atable_fn = fn(self.atable_file)
    sites, vmap_qid = _load_vmap(vmap_fn)
    arows = engine().assoc_table(sites, vmap_qid)
    _write(atable_fn, _lib.format_atable(sites, arows))
    _HANDOFF[_key(atable_fn)] = arows


def get_phased_blocks(self):
    vmap_fn = fn(self.vmap_file)
    atable_fn = fn(self.atable_file)
    p_variant_fn = fn(self.phased_variant_file)
    sites, _ = _load_vmap(vmap_fn)
    arows = _HANDOFF.get(_key(atable_fn))
    if arows is None:
        arows = textio.parse_atable(_read(atable_fn), sites)
    pvars = engine().phase_blocks(sites, arows)
    _write(p_variant_fn, _lib.format_phased_variants(sites, pvars))
    _HANDOFF[_key(p_variant_fn)] = pvars


def get_phased_reads(self):
    q_id_map_fn = fn(self.q_id_map_file)
    vmap_fn = fn(self.vmap_file)
    p_variant_fn = fn(self.phased_variant_file)
    ctg_id = self.parameters["ctg_id"]
    phased_read_fn = fn(self.phased_read_file)
    sites, vmap_qid = _load_vmap(vmap_fn)
    names = _HANDOFF.get(_key(q_id_map_fn))
    if names is None:
        names = textio.parse_q_id_map(_read(q_id_map_fn))
    qname_off, qnames = names
    pvars = _HANDOFF.get(_key(p_variant_fn))
    if pvars is None:
        pvars = textio.parse_phased_variants(_read(p_variant_fn), sites)
    preads = engine().phase_reads(sites, vmap_qid, pvars, len(qname_off) - 1)
    _write(phased_read_fn, _lib.format_phased_reads(preads, ctg_id, qname_off, qnames))


def read_contig(fasta_fn, ctg_id):
    """phasing.py:489-494: the record whose first header word is ctg_id, upper-cased ('' if absent)."""
    ref_seq = ""
    name, chunks = None, []

    def flush(seq):
        if name is not None and (name.split() or [""])[0] == ctg_id:
            return "".join(chunks).upper()
        return seq

    with open(fasta_fn) as f:
        for line in f:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                ref_seq = flush(ref_seq)
                name, chunks = line[1:], []
            else:
                chunks.append(line.strip())
    return flush(ref_seq)


def phasing(args):
    bam_fn = args.bam
    fasta_fn = args.fasta
    ctg_id = args.ctg_id
    base_dir = args.base_dir
    samtools = args.samtools

    ref_seq = read_contig(fasta_fn, ctg_id)

    wf = PypeProcWatcherWorkflow(max_jobs=1)

    bam_file = makePypeLocalFile(bam_fn)
    vmap_file = makePypeLocalFile(os.path.join(base_dir, ctg_id, 'het_call', "variant_map"))
    vpos_file = makePypeLocalFile(os.path.join(base_dir, ctg_id, 'het_call', "variant_pos"))
    q_id_map_file = makePypeLocalFile(os.path.join(base_dir, ctg_id, 'het_call', "q_id_map"))
    parameters = {"ctg_id": ctg_id, "ref_seq": ref_seq, "base_dir": base_dir, "samtools": samtools}
    make_het_call_task = PypeTask(inputs={"bam_file": bam_file},
                                  outputs={"vmap_file": vmap_file, "vpos_file": vpos_file, "q_id_map_file": q_id_map_file},
                                  parameters=parameters)(make_het_call)
    wf.addTasks([make_het_call_task])

    atable_file = makePypeLocalFile(os.path.join(base_dir, ctg_id, 'g_atable', "atable"))
    generate_association_table_task = PypeTask(inputs={"vmap_file": vmap_file}, outputs={"atable_file": atable_file},
                                               parameters={"ctg_id": ctg_id, "base_dir": base_dir})(generate_association_table)
    wf.addTasks([generate_association_table_task])

    phased_variant_file = makePypeLocalFile(os.path.join(base_dir, ctg_id, 'get_phased_blocks', "phased_variants"))
    get_phased_blocks_task = PypeTask(inputs={"vmap_file": vmap_file, "atable_file": atable_file},
                                      outputs={"phased_variant_file": phased_variant_file})(get_phased_blocks)
    wf.addTasks([get_phased_blocks_task])

    phased_read_file = makePypeLocalFile(os.path.join(base_dir, ctg_id, "phased_reads"))
    get_phased_reads_task = PypeTask(inputs={"vmap_file": vmap_file, "q_id_map_file": q_id_map_file,
                                             "phased_variant_file": phased_variant_file},
                                     outputs={"phased_read_file": phased_read_file},
                                     parameters={"ctg_id": ctg_id})(get_phased_reads)
    wf.addTasks([get_phased_reads_task])

    wf.refreshTargets()
    _HANDOFF.clear()


def parse_args(argv):
    parser = argparse.ArgumentParser(description='phasing variants and reads from a bam file')
    parser.add_argument('--bam', type=str, help='path to sorted bam file', required=True)
    parser.add_argument('--fasta', type=str, help='path to the fasta file of contain the contig', required=True)
    parser.add_argument('--ctg_id', type=str, help='contig identifier in the bam file', required=True)
    parser.add_argument('--base_dir', type=str, default="./", help='the output base_dir, default to current working directory')
    parser.add_argument('--samtools', type=str, default="samtools", help='path to samtools ("builtin": read the BAM with the library instead)')
    args = parser.parse_args(argv[1:])
    return args


def main(argv=sys.argv):
    logging.basicConfig()
    args = parse_args(argv)
    phasing(args)
