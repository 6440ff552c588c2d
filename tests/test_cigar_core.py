"""k_tb_cigar's best-sub-path pass (fzalign v1.5's rule, r6's one-sweep form) on the host: the kernel's own word-level functions (falcon_unzip_amd/csrc/fzp_cigar_core.h) inside
a loop that plays the wave (tests/cigar_host.cpp) against the serial rule of the twin (oracle/align_oracle.c finish_path) -- same score, same e, same s, under scores that
make prefixes tie at every other op (1 / 1 / 1, 1 / 0 / 1, 3 / 0 / 2), on streams from one op to several chunks of 64 words, with random and with mostly-matching bases."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def cig():
    so = os.path.join(HERE, "_cigar_host.so")
    src = os.path.join(HERE, "cigar_host.cpp")
    hdr = os.path.join(HERE, "..", "falcon_unzip_amd", "csrc", "fzp_cigar_core.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
    lib = C.CDLL(so)
    for f in (lib.cig_serial, lib.cig_wave):
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
    return lib


def _pack2(codes, pad_words=4):
    n = len(codes)
    w = np.zeros((n + 15) // 16 + pad_words, np.uint32)
    idx = np.arange(n)
    np.bitwise_or.at(w, idx >> 4, (codes.astype(np.uint32) & 3) << (2 * (idx & 15)).astype(np.uint32))
    return w


def _case(rng, L, p_ins, p_del, matching):
    """an op stream (END first, as the walk leaves it) with the bases it runs over: read / contig packed 16 to a word, the cell its first op leaves"""
    u = rng.random(L)
    ops = np.where(u < p_ins, 1, np.where(u < p_ins + p_del, 2, 0)).astype(np.uint8)
    ni, nj = int((ops != 2).sum()), int((ops != 1).sum())
    i_end, j_end = ni - 1 + int(rng.integers(0, 40)), nj - 1 + int(rng.integers(0, 40))
    read = rng.integers(0, 4, i_end + 1 + int(rng.integers(0, 20))).astype(np.uint8)
    ctg = rng.integers(0, 4, j_end + 1 + int(rng.integers(0, 20))).astype(np.uint8)
    if matching:      # the aligned columns agree (but for a tenth of them)
        i, j = i_end, j_end
        for op in ops:
            if op == 0:
                if rng.random() < 0.9:
                    read[i] = ctg[j]
                i -= 1; j -= 1
            elif op == 1:
                i -= 1
            else:
                j -= 1
    return _pack2(ops), _pack2(read), _pack2(ctg), i_end, j_end


@pytest.mark.parametrize("scores", [(2, 4, 3), (1, 1, 1), (1, 0, 1), (3, 0, 2), (7, 5, 1), (4096, 4096, 4096)])
def test_one_sweep_form_equals_the_serial_rule(cig, scores):
    rng = np.random.default_rng(sum(scores))
    lengths = [1, 2, 3, 15, 16, 17, 31, 32, 33, 63, 64, 65, 1023, 1024, 1025, 1040] + [int(x) for x in rng.integers(1, 3000, 120)] + [int(x) for x in rng.integers(3000, 20000, 12)]
    n_pos = 0
    for k, L in enumerate(lengths):
        for matching in (False, True):
            p_ins, p_del = ((0.08, 0.04), (0.3, 0.3), (0.0, 0.0), (0.02, 0.5))[k % 4]
            ops, q, t, i_end, j_end = _case(rng, L, p_ins, p_del, matching)
            a, b = np.zeros(3, np.int64), np.zeros(3, np.int64)
            args = (ops.ctypes.data, L, q.ctypes.data, t.ctypes.data, i_end, j_end) + scores
            cig.cig_serial(*args, a.ctypes.data)
            cig.cig_wave(*args, b.ctypes.data)
            assert list(a) == list(b), (scores, L, matching, p_ins, p_del, list(a), list(b))
            n_pos += int(a[0] > 0)
    assert n_pos > len(lengths)      # (most streams do hold a matching column)
