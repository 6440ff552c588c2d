"""Text -> record parsers for the reference's on-disk intermediates (host logic, numpy).

The reference's tasks hand data to each other through text files (variant_map, atable,
phased_variants, q_id_map; formats at falcon_unzip/phasing.py:124-134,199,418-421).  The drop-in
task functions read those files exactly where the reference does; these helpers turn them into the
fixed-width records of include/fzphase.h.
"""
from __future__ import annotations

import numpy as np

from ._lib import AROW, PVAR, SITE

_PY2_ORDER = {ord("A"): 0, ord("C"): 1, ord("T"): 2, ord("G"): 3}


class FormatError(ValueError):
    pass


def _columns(text: bytes, ncol: int, what: str):
    tok = text.split()
    if len(tok) % ncol:
        raise FormatError("%s: %d tokens is not a multiple of %d columns" % (what, len(tok), ncol))
    return [tok[k::ncol] for k in range(ncol)]


def _ints(col, dtype=np.int64):
    return np.array([int(x) for x in col], dtype=dtype) if col else np.zeros(0, dtype)


def _chars(col):
    return np.array([x[0] for x in col], dtype=np.uint8) if col else np.zeros(0, np.uint8)


def parse_variant_map(text: bytes):
    """variant_map rows `pos ref base q_id` -> (sites[SITE], vmap_qid[int32]).

    Rows of one site are contiguous, major-allele rows first (phasing.py:125-128); `total`,
    `count[2:]`, `base[2:]` are not recoverable from this file and stay 0."""
    c = _columns(text, 4, "variant_map")
    qid = _ints(c[3], np.int32)
    n = len(qid)
    if n == 0:
        return np.zeros(0, SITE), qid
    pos, refb, base = _ints(c[0]), _chars(c[1]), _chars(c[2])
    new_site = np.ones(n, bool)
    new_site[1:] = (pos[1:] != pos[:-1]) | (refb[1:] != refb[:-1])
    starts = np.flatnonzero(new_site)
    ends = np.append(starts[1:], n)
    # allele change points inside each site
    chg = np.zeros(n, bool)
    chg[1:] = (base[1:] != base[:-1]) & ~new_site[1:]
    nchg = np.add.reduceat(chg.astype(np.int64), starts)
    if np.any(nchg != 1):
        bad = int(pos[starts[np.flatnonzero(nchg != 1)[0]]])
        raise FormatError("variant_map: site at %d does not hold exactly two contiguous alleles" % bad)
    first_chg = np.flatnonzero(chg)          # exactly one per site, in site order
    sites = np.zeros(len(starts), SITE)
    sites["pos"] = pos[starts] - 1
    sites["ref_base"] = refb[starts]
    sites["base"][:, 0] = base[starts]
    sites["base"][:, 1] = base[first_chg]
    sites["count"][:, 0] = first_chg - starts
    sites["count"][:, 1] = ends - first_chg
    sites["row_off"] = starts
    if np.any(np.diff(sites["pos"]) <= 0):
        raise FormatError("variant_map: positions are not strictly ascending")
    return sites, qid


def _site_index(sites, pos1, what):
    if len(pos1) == 0:
        return np.zeros(0, np.int32)
    if len(sites) == 0:
        raise FormatError("%s: position %d is not a site of variant_map (reference: KeyError)" % (what, int(pos1[0])))
    idx = np.searchsorted(sites["pos"], pos1 - 1)
    bad = (idx >= len(sites)) | (sites["pos"][np.minimum(idx, len(sites) - 1)] != pos1 - 1)
    if np.any(bad):
        raise FormatError("%s: position %d is not a site of variant_map (reference: KeyError)" % (what, int(pos1[np.flatnonzero(bad)[0]])))
    return idx.astype(np.int32)


def py2_pair(site):
    """The site's two alleles in CPython-2.7 dict order A<C<T<G (phasing.py:175,181)."""
    a, b = int(site["base"][0]), int(site["base"][1])
    return (a, b) if _PY2_ORDER.get(a, 9) < _PY2_ORDER.get(b, 9) else (b, a)


def parse_atable(text: bytes, sites):
    """atable rows (phasing.py:199) -> AROW records; allele columns must be the sites' alleles in A<C<T<G order."""
    c = _columns(text, 10, "atable")
    n = len(c[0])
    rows = np.zeros(n, AROW)
    if n == 0:
        return rows
    rows["site1"] = _site_index(sites, _ints(c[0]), "atable")
    rows["site2"] = _site_index(sites, _ints(c[3]), "atable")
    for k in range(4):
        rows["n"][:, k] = _ints(c[6 + k], np.int32)
    # the allele columns are implied by the sites; verify them (vectorised)
    b = sites["base"][:, :2].astype(np.int64)
    rank = np.vectorize(lambda v: _PY2_ORDER.get(int(v), 9))(b) if len(b) else b
    first = np.where(rank[:, 0] < rank[:, 1], b[:, 0], b[:, 1])
    second = np.where(rank[:, 0] < rank[:, 1], b[:, 1], b[:, 0])
    ok = (first[rows["site1"]] == _chars(c[1])) & (second[rows["site1"]] == _chars(c[2])) & \
         (first[rows["site2"]] == _chars(c[4])) & (second[rows["site2"]] == _chars(c[5]))
    if not np.all(ok):
        raise FormatError("atable row %d: allele columns do not match variant_map" % int(np.flatnonzero(~ok)[0]))
    return rows


def parse_phased_variants(text: bytes, sites):
    """'V' lines of phased_variants (phasing.py:421) -> PVAR records ('P' lines are derived data)."""
    recs = []
    for line in text.split(b"\n"):
        l = line.split()
        if not l or l[0] != b"V":
            continue
        recs.append(l)
    out = np.zeros(len(recs), PVAR)
    if not recs:
        return out
    pos1 = np.array([int(l[2]) for l in recs], dtype=np.int64)
    out["site"] = _site_index(sites, pos1, "phased_variants")
    out["block"] = [int(l[1]) for l in recs]
    out["b1"] = [l[3][-1] for l in recs]
    out["b2"] = [l[4][-1] for l in recs]
    if all(len(l) >= 9 for l in recs):
        out["lext"] = [int(l[5]) for l in recs]
        out["rext"] = [int(l[6]) for l in recs]
        out["lscore"] = [int(l[7]) for l in recs]
        out["rscore"] = [int(l[8]) for l in recs]
    return out


def parse_q_id_map(text: bytes):
    """q_id_map rows `q_id QNAME` -> (qname_off[int64], qnames bytes), dense ids (phasing.py:132-134)."""
    names = {}
    for line in text.split(b"\n"):
        l = line.split()
        if not l:
            continue
        names[int(l[0])] = l[1]
    n = (max(names) + 1) if names else 0
    off = np.zeros(n + 1, np.int64)
    parts = []
    for i in range(n):
        nm = names.get(i, b"")
        parts.append(nm)
        off[i + 1] = off[i] + len(nm)
    return off, b"".join(parts)
