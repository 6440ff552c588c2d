"""Helpers for the consensus tests: a banded edit distance (numpy) and a small simulated diploid contig."""
import numpy as np


def banded_edit_distance(a: bytes, b: bytes, band=200):
    """Levenshtein distance restricted to |i - j| <= band (exact when the true distance is below the band)."""
    x = np.frombuffer(a, np.uint8)
    y = np.frombuffer(b, np.uint8)
    n, m = len(x), len(y)
    INF = 1 << 30
    w = 2 * band + 1
    prev = np.full(w, INF, np.int64)          # prev[k] = D[i-1][i-1 + k - band]
    for k in range(w):
        j = k - band
        if 0 <= j <= m:
            prev[k] = j                        # row 0
    for i in range(1, n + 1):
        j = np.arange(i - band, i + band + 1)
        valid = (j >= 0) & (j <= m)
        cur = np.full(w, INF, np.int64)
        # substitution / match: D[i-1][j-1] -> same k in prev
        yj = np.where((j >= 1) & (j <= m), y[np.clip(j - 1, 0, m - 1)], 255)
        sub = prev + (yj != x[i - 1])
        sub[j < 1] = INF
        # deletion of x[i-1]: D[i-1][j] -> prev[k+1]
        dele = np.full(w, INF, np.int64)
        dele[:-1] = prev[1:] + 1
        cur = np.minimum(sub, dele)
        cur[~valid] = INF
        if i <= band:
            cur[band - i] = i                  # column 0
        # insertion: D[i][j-1] -> cur[k-1] + 1, a running minimum along the row
        idx = np.arange(w)
        cur = np.minimum.accumulate(np.where(cur < INF, cur - idx, INF)) + idx
        cur[~valid] = INF
        prev = cur
    k = m - n + band
    return int(prev[k]) if 0 <= k < w else INF


def diploid_case(seed, L=60000, n_reads=360, R=9000, het_rate=1.0 / 400):
    from falcon_unzip_amd import sim
    rng = np.random.Generator(np.random.PCG64(seed))
    hap0, hap1, het = sim.make_diploid(L, rng, het_rate=het_rate)
    reads = sim.simulate_reads(hap0, hap1, n_reads, R, rng)
    return hap0, hap1, het, reads
