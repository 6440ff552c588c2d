"""The bit-sliced cell function of the K1 DP (falcon_unzip_amd/csrc/fzp_swb_core.h, what k_swb runs per lane and k_swb2 per pair of lanes) on the host against the scalar twin's
extension (oracle/align_oracle.c dp_extend): same moves at every step, same trace-back masks on every cell a walk can visit, same terminal."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests import oracle_lib

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def swb():
    so = os.path.join(HERE, "_swb_host.so")
    src = os.path.join(HERE, "swb_host.cpp")
    hdr = os.path.join(HERE, "..", "falcon_unzip_amd", "csrc", "fzp_swb_core.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
    return C.CDLL(so)


def _mutate(rng, seq, sub, ins, dele):
    out = []
    for b in seq:
        r = rng.random()
        if r < dele:
            continue
        if r < dele + sub:
            out.append((b + rng.integers(1, 4)) & 3)
        else:
            out.append(b)
        while rng.random() < ins:
            out.append(rng.integers(0, 4))
    return np.array(out, np.uint8)


def _both(oracle, swb, q, t, inner=0, band=64):
    """-> [twin, host loop of the one-lane form, host loop of the pair form] for the 64-cell band; [twin, host loop of swb_step32's form] for the 32-cell one (fzalign v1.8)"""
    nq, nt = len(q), len(t)
    cap = nq + nt + 8
    P = oracle_lib.AlignParams()
    oracle.lib.orc_align_params_default(C.byref(P))
    P.band = band
    res = []
    forms = ((oracle.lib.orc_dp_extend_raw, True), (swb.swb_extend_host, False), (swb.swb_extend_pair_host, False)) if band == 64 else ((oracle.lib.orc_dp_extend_raw, True), (swb.swb_extend_host32, False))
    for fn, extra in forms:
        D = np.zeros(cap, np.uint64); G = np.zeros(cap, np.uint64); mv = np.zeros(cap, np.uint8); out = np.zeros(4, np.int64)
        fn.restype = C.c_int
        args = [q.ctypes.data_as(C.c_void_p), C.c_int64(nq), t.ctypes.data_as(C.c_void_p), C.c_int64(nt)]
        if extra:
            args.append(C.byref(P))
        args += [D.ctypes.data_as(C.c_void_p), G.ctypes.data_as(C.c_void_p), mv.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_int(inner)]
        assert fn(*args) == 0
        res.append((D, G, mv, out))
    return res


def _compare(a, b, nq, nt, band=64):
    (D0, G0, m0, o0), (D1, G1, m1, o1) = a, b
    assert tuple(o0) == tuple(o1), (tuple(o0), tuple(o1))
    steps = int(o0[0])
    assert np.array_equal(m0[:steps], m1[:steps])
    i0 = -(band // 2 + 1) + np.cumsum(m0[:steps].astype(np.int64))
    tt = np.arange(steps, dtype=np.int64)
    k = np.arange(band, dtype=np.int64)
    i = i0[:, None] + k[None, :]
    j = tt[:, None] - i
    real = (i >= 0) & (i < nq) & (j >= 0) & (j < nt)                 # the cells a walk can visit
    bits = np.uint64(1) << k.astype(np.uint64)
    d0 = (D0[:steps, None] & bits[None, :]) != 0
    d1 = (D1[:steps, None] & bits[None, :]) != 0
    assert np.array_equal(d0[real], d1[real]), np.argwhere(real & (d0 != d1))[:5]
    g0 = (G0[:steps, None] & bits[None, :]) != 0
    g1 = (G1[:steps, None] & bits[None, :]) != 0
    look = real & ~d0                                               # the walk reads G only where the diagonal lost
    assert np.array_equal(g0[look], g1[look]), np.argwhere(look & (g0 != g1))[:5]


@pytest.mark.parametrize("seed,L,sub,ins,dele", [(1, 300, 0.01, 0.08, 0.04), (2, 1500, 0.01, 0.08, 0.04), (3, 4000, 0.02, 0.10, 0.05), (4, 2500, 0.0, 0.0, 0.0),
                                                 (5, 2000, 0.05, 0.02, 0.12), (6, 2000, 0.05, 0.14, 0.02), (7, 900, 0.25, 0.2, 0.2)])
def test_masks_moves_terminal_equal_the_twin(oracle, swb, seed, L, sub, ins, dele):
    rng = np.random.Generator(np.random.PCG64(seed))
    for rep in range(6):
        ref = rng.integers(0, 4, size=L + L // 3 + 200, dtype=np.uint8)
        q = _mutate(rng, ref[:L], sub, ins, dele)
        nq = len(q)
        nt = min(len(ref), nq + nq // 4 + 64) if rep % 3 else min(len(ref), max(64, int(nq * (0.7 + 0.1 * rep))))     # also windows that end before the read does
        t = np.ascontiguousarray(ref[:nt])
        a, b, c = _both(oracle, swb, q, t)
        _compare(a, b, nq, nt)
        _compare(a, c, nq, nt)
        # the same sub-matrix as an INNER piece (fzalign v1.6): the terminal is the border cell the global alignment to the corner passes through
        a, b, c = _both(oracle, swb, q, t, inner=1)
        _compare(a, b, nq, nt)
        _compare(a, c, nq, nt)
        # fzalign v1.8: the 32-cell band (one register per plane: swb_step32's form) against the twin with band = 32, free and inner
        for inner in (0, 1):
            a, b = _both(oracle, swb, q, t, inner=inner, band=32)
            _compare(a, b, nq, nt, band=32)


def test_unrelated_sequences_and_low_complexity(oracle, swb):
    rng = np.random.Generator(np.random.PCG64(11))
    for rep in range(6):
        q = rng.integers(0, 4, size=700 + 50 * rep, dtype=np.uint8)
        t = rng.integers(0, 4, size=900, dtype=np.uint8)
        if rep >= 3:                       # homopolymers and dinucleotide repeats: ties everywhere
            q[:] = np.tile(np.array([0, 1], np.uint8), len(q))[:len(q)] if rep == 3 else 2
            t[:] = np.tile(np.array([0, 1], np.uint8), len(t))[:len(t)] if rep != 5 else 2
        a, b, c = _both(oracle, swb, q, t)
        _compare(a, b, len(q), len(t))
        _compare(a, c, len(q), len(t))
        a, b = _both(oracle, swb, q, t, band=32)
        _compare(a, b, len(q), len(t), band=32)


def test_interior_block_never_touches_a_border(swb):
    """k_sw runs `sw_interior_safe(t, i0, nq, nt)` steps without looking at borders (fzp_swb_core.h; the kernel calls the same function): in none of them -- whatever the
    moves -- may a cell of the band lie on the matrix's last row or last column, because those cells are the terminal's candidates and only the checked steps see them.
    Exhaustive over band positions on small matrices; the r3 form of the count (one step more: ADVICE r3) is the negative control and must be caught."""
    f = swb.sw_safe_probe
    f.restype = C.c_int
    safe = C.c_int32()
    seen_ok = caught = tight = 0
    for nq in (70, 97, 128, 200):
        for nt in (66, 90, 131, 260):
            for t in range(64, nq + nt):
                for i0 in range(0, t - 63):
                    if i0 + 63 > nq - 1 or (t - 1) - i0 > nt - 1:
                        continue                                     # the band already past a border: the kernel is in its checked steps
                    r = f(t, i0, nq, nt, 0, C.byref(safe))
                    assert r in (-1, 0), (nq, nt, t, i0, safe.value)
                    seen_ok += r == 0
                    s0 = safe.value
                    r1 = f(t, i0, nq, nt, 1, C.byref(safe))
                    if safe.value > 0:
                        assert r1 == 1, (nq, nt, t, i0)              # one step more always CAN reach a border: the count is tight, and the r3 form is wrong
                        caught += 1
                        tight += safe.value == max(s0, 0) + 1
    assert seen_ok > 10000 and caught > 10000 and tight == caught
