/*
 * align_oracle.c -- scalar CPU twin of the K1 read->contig aligner.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED vs the reference: FALCON_unzip shells out to `blasr` (falcon_unzip/unzip.py:86-88),
 * a third-party C++ aligner that is not vendored under /root/reference and whose results
 * (placement, clipping, `--hitPolicy randombest --randomSeed 42` tie-breaks) cannot be reproduced
 * here.  This file therefore DEFINES the aligner ("fzalign v1.8"); the HIP kernels in
 * falcon_unzip_amd/csrc/fzp_align.hip must match it bit-for-bit (summaries, CIGARs, DP cell counts),
 * and its quality is judged against the simulator's true alignments.
 *
 * fzalign v1.8  (v1.1: one index position per k-mer, one candidate placement per read, no identity gate; v1.2: multi-position index, two
 *               candidates, chains; v1.3: best-start soft clip; v1.4: anchor = the chain's first hit, extension forward AND backward from it;
 *               v1.5: an extension runs to the matrix BORDER, the alignment is the best-scoring stretch of the joined path;
 *               v1.6: the chain's hits every >= PIECE read bases are WAYPOINTS, the forward extension is a sequence of independent banded DPs from one
 *               waypoint to the next (what blasr does between the anchors of its chain, unzip.py:86-88), free ends are limited;
 *               v1.7: index and look-ups use the ANCHORED k-mers -- those that start with AC or end with GT -- instead of fixed strides;
 *               v1.8: the DP band is a parameter, 32 cells by default (orc_align_params.band, FZP_ALIGN_BAND; 64 = v1.7's): same scores on every compared read,
 *               profiles/r6_band32_go_nogo.txt)
 *   bases     A/a C/c G/g T/t -> 0..3, anything else -> 0
 *   selected  (v1.7) a k-mer position p is SELECTED iff bases p, p+1 are A, C or bases p+k-2, p+k-1 are G, T: about an eighth of the positions, decided by the
 *             k-mer alone -- the contig and a read select the same k-mers wherever they agree, on either strand (a k-mer ends with GT exactly when its reverse
 *             complement starts with AC) -- and found sixteen positions at a time with a few bit operations on the packed words.  (seed_anchored = 0 keeps
 *             v1.6: "selected" contig positions = every 2nd one, the samples of a read = every `stride`-th / 3 x `stride`-th k-mer.)
 *   index     canonical k-mers (k<=16, 2 bits/base, base m of a k-mer at bits 2m; canonical = the smaller of
 *             the k-mer and its reverse complement) of every SELECTED contig position -> EVERY such position
 *             (+ whether the canonical form was the reverse complement).  A k-mer with more than MAX_OCC = 8
 *             index entries is repetitive and never produces a hit.
 *   hits      the samples of a read = its selected FORWARD k-mers in read order -- all of them for a read below 8 192 bases, every ms-th
 *             (ms = 3 x ceil(n / 131 072): a long read has seeds to spare, a short one needs all of them) otherwise, 8 192 samples at most;
 *             every sample is looked up once; each index
 *             entry of it, by increasing contig position, is a hit: equal orientation bits -> the read matches as
 *             sequenced (strand 0, oriented offset i = pf), different -> its reverse complement does
 *             (i = n-k-pf); diagonal value dv = cpos - i + n.  Only the first HIT_CAP = 4096 hits of a read
 *             in that order exist.  A hit votes for bin = dv >> shift, shift = smallest s>=10 with
 *             ((Lc+n)>>s)+2 <= 8192.
 *   windows   a window (strand, b) scores votes[b]+votes[b+1].  W1 = the best window (ties: forward strand,
 *             lower bin); fewer than min_seed_hits votes -> unaligned.  W2 = the best window on the other
 *             strand or at least 3 bins away from W1 (same ties); it counts only if it has >= min_seed_hits
 *             votes and 4*votes(W2) >= votes(W1)  (the second placement of blasr's --bestn selection,
 *             unzip.py:86-88).
 *   chains    per window: its hits = those of its strand in bins b-1-ext .. b+2+ext, ext = n >> 14 (v1.6: a long read drifts across bins; W2 likewise lies
 *             at least 3 + ext bins from W1), by increasing oriented offset
 *             (hit order on the forward strand, reversed hit order on the other).  Chain length f(h) = 1 +
 *             max f(p) over the at most 64 preceding window hits p with 1 <= i_h - i_p <= 2048, cpos_p < cpos_h
 *             and |dv_h - dv_p| <= 16 + (i_h - i_p)/16 (ties: the closest p), else 1; start(h) = start(p) or
 *             h itself.  The longest chain (ties: the earliest end) gives the anchor = its first hit (i_h, c_h), a cell
 *             of the true path.  Candidates in order W1, W2.
 *   waypoints (v1.6) piece = max(3072, ceil(n / 30)).  Along a chain, its first hit is waypoint 0; a later hit of the chain is the next
 *             waypoint when its oriented offset lies at least `piece` bases after the previous waypoint's (so a read has at most 31).
 *             The longest chain's waypoints w_0 (the anchor) .. w_m cut the forward extension into pieces.
 *   pieces    piece k < m is INNER: the extension DP (below) from w_k on the sub-matrix of the nq = i(w_k+1) - i(w_k) read bases and the
 *             nt = c(w_k+1) - c(w_k) contig bases up to the next waypoint; its terminal is the valid border cell with the largest
 *             H - gap * (distance to the sub-matrix's corner (nq-1, nt-1) along the border) -- the global alignment through the band --
 *             and the path of the piece is the gap moves from the corner to the terminal, the walk, the gap moves its exit implies.
 *             Piece m is FREE: from w_m over min(n - i(w_m), 2 * piece) read bases and min(Lc - c(w_m), nq + nq/4 + 64) contig bases,
 *             terminal = the best valid border cell (v1.5).  BACKWARD from the anchor: the same free DP on the reversed read prefix of
 *             min(i_h, piece) bases and the reversed contig window of min(c_h, nq + nq/4 + 64) bases before c_h; its path (from ITS terminal)
 *             is joined to the forward one at the anchor's corner, together with the gap moves either walk's exit
 *             through row / column -1 implies.  Read bases no piece reaches are soft clip.
 *   selection every candidate is extended; the one whose FORWARD pieces' terminal scores sum highest wins (ties: the earlier
 *             candidate) -- blasr's --bestn 1.  `cells` counts every DP of every candidate.
 *   best sub-path (v1.5; v1.3's "best start" at both ends)  P(k) = score of the joined path's first k ops counted from the
 *             forward terminal; the alignment is ops e..s with the largest P(s+1) - P(e) (ties: the smallest s, then
 *             the largest e) -- it begins and ends with a match column; what the path holds outside is soft clip.
 *             `score` = that stretch's score.  A path without a match column gives no alignment.
 *   identity  n_match = (score + mismatch*columns + gap*(insertions + deletions)) / (match + mismatch)
 *             over the alignment (exact); the alignment is dropped (unaligned) when
 *             100*n_match < 70*(columns + inserted + deleted bases of the trimmed alignment)
 *             (blasr --minPctIdentity 70.0, unzip.py:87).
 *   extension adaptive anti-diagonal band of B cells (B = 32 since v1.8, 64 before; Suzuki-Kasahara style) from the cell before the anchor
 *             (origin (-1, -1) of the anchor-relative matrix): linear gaps, H = max(diag + (match | -mismatch), up - gap, left - gap), no zero
 *             floor; the first B steps alternate down/right, afterwards the band moves RIGHT when
 *             H[lane 0] > H[lane B-1], else DOWN.  The diagonal operand is H of two steps ago in that
 *             step's own lane layout: the predecessor of lane k sits in lane k - 1 + (number of DOWN
 *             moves among the last two); a lane outside 0..B-1 reads as minus infinity.  The extension's TERMINAL is the best-scoring valid cell
 *             of the matrix border -- the read's last row or the window's last column -- (first in step order, then lowest lane): the
 *             DP needs no score of any other cell, only which of the three moves won (v1.5).  Trace-back priority: diagonal, then the gap whose source is the
 *             same lane of the previous step (the cell above after a DOWN move, the cell to the left
 *             after a RIGHT move), then the other gap.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define W 64             /* room for the widest band; the band itself is a parameter since v1.8 (orc_align_params.band: 64 or 32, 0 = FZP_ALIGN_BAND or the default) */
#define ORC_DEFAULT_BAND 32
#define NEG (-(1 << 26))

typedef struct {
    int32_t kmer, seed_stride, match, mismatch, gap, min_seed_hits;
    int32_t min_pct_identity;            /* 70 = blasr --minPctIdentity 70.0 (unzip.py:87); 0 disables the gate */
    int32_t seed_anchored;               /* v1.7 (default 1): index and samples are the ANCHORED k-mers (start with AC or end with GT); 0 = v1.6's fixed strides (every 2nd contig position, every stride-th read k-mer) */
    int32_t band;                        /* v1.8: cells of the adaptive band, 64 or 32 (tools/runs/band32_go_nogo.py: on every test set the 32-cell band finds the same scores) */
    int32_t reserved[7];
} orc_align_params;

typedef struct {
    int32_t aligned, strand, pos, ref_end, q_start, q_end, score, n_cigar;
    int64_t cells;
    int32_t n_columns, n_match;
} orc_aln_summary;

#define MAX_OCC 8
#define HIT_CAP 4096
#define CHAIN_LOOKBACK 64
#define CHAIN_MAX_GAP 2048
#define BRIDGE_MAX_GAP 4096 /* v1.6: a chain may cross a seedless stretch of up to this many read bases ... */
#define BRIDGE_COST 4       /* ... for the price of this many hits */
#define LONG_READ 8192      /* v1.6: reads of at least this many bases are sampled at LONG_STRIDE times the stride */
#define LONG_STRIDE 3
#define LONG_MS 3           /* v1.7: ... and of their selected k-mers every LONG_MS-th (x 2, x 3 ... per further 131 072 bases) is looked up */
#define SAMPLE_CAP 8192     /* v1.7: k-mers looked up per read at most (the first ones in read order) */
#define PIECE_LEN 3072      /* v1.6: read bases between waypoints (at least) */
#define MAX_WP 31           /* waypoints per candidate (the device joins a read's pieces one per lane: 2 x 32 slots) */

static inline int code_of(uint8_t c) {
    switch (c) {
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 0;
    }
}

typedef struct { uint32_t key; int32_t pos; } kp_t;
static int cmp_kp(const void *a, const void *b) {
    const kp_t *x = (const kp_t *)a, *y = (const kp_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return (x->pos > y->pos) - (x->pos < y->pos);
}

typedef struct {
    kp_t *kp; int64_t n;     /* sorted (key, pos) */
    const uint8_t *codes; int64_t len;
} ctg_index;

static uint32_t kmer_at(const uint8_t *codes, int64_t p, int k) {
    uint32_t key = 0;
    for (int m = 0; m < k; m++) key |= (uint32_t)codes[p + m] << (2 * m);
    return key;
}

static uint32_t rc_of(uint32_t key, int k) {
    uint32_t r = 0;
    for (int m = 0; m < k; m++) r |= (3u - ((key >> (2 * m)) & 3u)) << (2 * (k - 1 - m));
    return r;
}

/* ---- v1.7: anchored k-mers.  A k-mer is SELECTED iff it starts with AC or ends with GT (codes 0 1 / 2 3): about an eighth of all positions, decided by the k-mer alone --
 * so the contig and a read pick the SAME k-mers wherever they agree, on either strand (a k-mer ends with GT exactly when its reverse complement starts with AC) -- and by
 * two bases of it: both sides find their selected positions with a few bit operations per 16 bases.  The index holds every selected contig position (v1.6: every 2nd
 * position, four times as many entries); a read looks up its selected k-mers (v1.6: every 4th / 12th k-mer, of which only those on indexed positions could hit). */
static inline int kmer_selected(const uint8_t *codes, int64_t p, int k) {
    return (codes[p] == 0 && codes[p + 1] == 1) || (codes[p + k - 2] == 2 && codes[p + k - 1] == 3);
}
/* out[0 .. return value) = the selected positions of codes[0 .. len), increasing */
static int64_t selected_positions(const uint8_t *codes, int64_t len, int k, int64_t *out) {
    int64_t n = 0;
    for (int64_t p = 0; p + k <= len; p++) if (kmer_selected(codes, p, k)) out[n++] = p;
    return n;
}

/* index entries of the canonical key: [*lo, *lo + return value) */
static int64_t index_range(const ctg_index *ix, uint32_t key, int64_t *first) {
    int64_t lo = 0, hi = ix->n;
    while (lo < hi) { int64_t m = (lo + hi) >> 1; if (ix->kp[m].key < key) lo = m + 1; else hi = m; }
    int64_t e = lo;
    while (e < ix->n && ix->kp[e].key == key) e++;
    *first = lo;
    return e - lo;
}

typedef struct { uint32_t *v; int64_t n, cap; } u32vec;
static void push(u32vec *v, uint32_t x) {
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1 << 16; v->v = (uint32_t *)realloc(v->v, (size_t)v->cap * 4); }
    v->v[v->n++] = x;
}

typedef struct { int32_t s; int64_t i, cp, dv; } hit_t;
static hit_t *g_dbg_hits;      /* seed_candidates copies its hit list here when set (orc_debug_hits, a single-threaded test hook) */
static int64_t g_dbg_nh;
typedef struct { int strand; int64_t i_a, c_a; int n_wp; int64_t wi[MAX_WP], wc[MAX_WP]; } anchor_t;      /* (wi[0], wc[0]) = the anchor */
static inline int64_t piece_len(int64_t n) { const int64_t p = (n + MAX_WP - 2) / (MAX_WP - 1); return p > PIECE_LEN ? p : PIECE_LEN; }

/* ---- seeding: hits, coarse windows, one chain per window -> up to 2 anchors (spec in the header) */
/* per-thread grow-only scratch for the per-read work arrays: hundreds of KB each, i.e. above malloc's mmap threshold -- 256 threads that
 * mmap / page-fault / munmap them for every read serialise on the process's address-space lock (measured on the 256-thread bench host:
 * 7.8 Mcell/s per thread against 97 single-threaded) */
static __thread struct { void *p; size_t cap; } scratch_[24];
static void *scratch_get(int k, size_t bytes) {
    if (scratch_[k].cap < bytes) {
        free(scratch_[k].p);
        scratch_[k].cap = bytes + bytes / 4 + 64;
        scratch_[k].p = malloc(scratch_[k].cap);
    }
    return scratch_[k].p;
}
static void scratch_release(void) {
    for (int k = 0; k < 24; k++) { free(scratch_[k].p); scratch_[k].p = NULL; scratch_[k].cap = 0; }
}

static int seed_candidates(const ctg_index *ix, const uint8_t *fwd, int64_t n, const orc_align_params *P, anchor_t *cand) {
    const int k = P->kmer, stride = P->seed_stride * (n >= LONG_READ ? LONG_STRIDE : 1);      /* v1.6: a long read has seeds to spare */
    const int64_t Lc = ix->len;
    int shift = 10;
    while ((((Lc + n) >> shift) + 2) > 8192) shift++;
    const int64_t NB = ((Lc + n) >> shift) + 2;
    const int64_t ext = n >> 14;      /* v1.6: a long read's diagonal drifts (CLR reads carry more inserted than deleted bases): its window widens by a bin per 16 384 bases */
    uint32_t *votes = (uint32_t *)scratch_get(6, (size_t)(2 * NB) * 4);
    memset(votes, 0, (size_t)(2 * NB) * 4);
    hit_t *hits = (hit_t *)scratch_get(7, (size_t)HIT_CAP * sizeof(hit_t));
    int64_t nh = 0;
    /* the samples: v1.6 every stride-th forward k-mer; v1.7 the read's selected k-mers -- all of a short read's, every ms-th of a long one's, SAMPLE_CAP at most */
    int64_t *samp = NULL, n_samp = 0;
    if (P->seed_anchored) {
        int64_t *mp = (int64_t *)scratch_get(8, (size_t)(n > 0 ? n : 1) * 8);
        const int64_t nm = selected_positions(fwd, n, k, mp);
        const int64_t ms = n >= LONG_READ ? (int64_t)LONG_MS * ((n + 131071) / 131072) : 1;
        samp = mp;
        for (int64_t o = 0; o < nm && n_samp < SAMPLE_CAP; o += ms) samp[n_samp++] = mp[o];      /* (in place: o >= n_samp) */
    }
    for (int64_t sx = 0, pf0 = 0; nh < HIT_CAP; sx++, pf0 += stride) {
        int64_t pf;
        if (samp) { if (sx >= n_samp) break; pf = samp[sx]; }
        else { if (pf0 + k > n) break; pf = pf0; }
        uint32_t kf = kmer_at(fwd, pf, k), kr = rc_of(kf, k);
        uint32_t orr = kr < kf ? 1u : 0u;
        int64_t first, cnt = index_range(ix, kr < kf ? kr : kf, &first);
        if (cnt == 0 || cnt > MAX_OCC) continue;
        for (int64_t e = first; e < first + cnt && nh < HIT_CAP; e++) {      /* entries of a key are sorted by position */
            uint32_t hit = (uint32_t)ix->kp[e].pos;
            int s = (int)((hit & 1u) ^ orr);
            int64_t cp = hit >> 1, i = s ? n - k - pf : pf;
            hits[nh].s = s; hits[nh].i = i; hits[nh].cp = cp; hits[nh].dv = cp - i + n; nh++;
            votes[s * NB + ((cp - i + n) >> shift)]++;
        }
    }
    if (g_dbg_hits) { memcpy(g_dbg_hits, hits, (size_t)nh * sizeof(hit_t)); g_dbg_nh = nh; }
    int n_cand = 0;
    uint32_t w1 = 0; int s1 = 0; int64_t b1 = 0;
    for (int s = 0; s < 2; s++)
        for (int64_t b = 0; b + 1 < NB; b++) {
            uint32_t sc = votes[s * NB + b] + votes[s * NB + b + 1];
            if (sc > w1) { w1 = sc; s1 = s; b1 = b; }
        }
    if ((int32_t)w1 < P->min_seed_hits || w1 == 0) return 0;
    uint32_t w2 = 0; int s2 = 0; int64_t b2 = 0;
    for (int s = 0; s < 2; s++)
        for (int64_t b = 0; b + 1 < NB; b++) {
            if (s == s1 && b - b1 < 3 + ext && b1 - b < 3 + ext) continue;
            uint32_t sc = votes[s * NB + b] + votes[s * NB + b + 1];
            if (sc > w2) { w2 = sc; s2 = s; b2 = b; }
        }
    const int n_win = ((int32_t)w2 >= P->min_seed_hits && w2 > 0 && 4ull * w2 >= w1) ? 2 : 1;
    int32_t *wh = (int32_t *)malloc((size_t)(nh ? nh : 1) * 4), *f = (int32_t *)malloc((size_t)(nh ? nh : 1) * 4), *st = (int32_t *)malloc((size_t)(nh ? nh : 1) * 4);
    int32_t *wl = (int32_t *)malloc((size_t)(nh ? nh : 1) * 4), *wprev = (int32_t *)malloc((size_t)(nh ? nh : 1) * 4);      /* v1.6: last waypoint of the chain ending in e; the waypoint before waypoint e */
    const int64_t piece = piece_len(n);
    for (int w = 0; w < n_win; w++) {
        const int ws = w ? s2 : s1;
        const int64_t wb = w ? b2 : b1;
        /* the window's hits (bins wb-1 .. wb+2) by increasing oriented offset: list order on the forward strand, reversed list order on the other */
        int64_t m = 0;
        for (int64_t x = 0; x < nh; x++) {
            const int64_t h = ws ? nh - 1 - x : x;
            if (hits[h].s != ws) continue;
            const int64_t b = hits[h].dv >> shift;
            if (b >= wb - 1 - ext && b <= wb + 2 + ext) wh[m++] = (int32_t)h;
        }
        int32_t best_f = 0; int64_t best_e = -1;
        for (int64_t e = 0; e < m; e++) {
            const hit_t *H = &hits[wh[e]];
            int32_t bf = 1, bst = (int32_t)e; int64_t bp = -1;
            for (int64_t back = 1; back <= CHAIN_LOOKBACK && e - back >= 0; back++) {        /* closest predecessor first */
                const hit_t *Q = &hits[wh[e - back]];
                const int64_t di = H->i - Q->i;
                if (di < 1 || di > BRIDGE_MAX_GAP || H->cp <= Q->cp) continue;
                int64_t dd = H->dv - Q->dv; if (dd < 0) dd = -dd;
                int32_t v;
                if (di <= CHAIN_MAX_GAP && dd <= 16 + di / 16) v = f[e - back] + 1;
                else if (dd <= 64 + di / 8) v = f[e - back] + 1 - BRIDGE_COST;       /* v1.6: a bridge over a stretch too noisy for seeds (the diagonal may drift further there) */
                else continue;
                if (v > bf) { bf = v; bst = st[e - back]; bp = e - back; }
            }
            f[e] = bf; st[e] = bst;
            if (bp < 0) { wl[e] = (int32_t)e; wprev[e] = -1; }
            else if (H->i - hits[wh[wl[bp]]].i >= piece) { wl[e] = (int32_t)e; wprev[e] = wl[bp]; }
            else wl[e] = wl[bp];
            if (bf > best_f) { best_f = bf; best_e = e; }
        }
        if (best_e < 0) continue;                      /* cannot happen: the window has votes */
        const hit_t *A = &hits[wh[st[best_e]]];
        if (getenv("ORC_ALIGN_DEBUG")) fprintf(stderr, "window %d: strand %d bin %lld votes %u/%u hits %lld chain %d anchor (%lld, %lld)\n", w, ws, (long long)wb, w ? w2 : w1, w1, (long long)m, best_f, (long long)A->i, (long long)A->cp);
        /* v1.4: the chain's first hit IS the anchor -- a cell of the true path; the extension runs forward from it and backward from it */
        cand[n_cand].strand = ws;
        cand[n_cand].i_a = A->i;
        cand[n_cand].c_a = A->cp;
        {   /* v1.6: the longest chain's waypoints, first to last */
            int nw = 0;
            for (int32_t x = wl[best_e]; x >= 0; x = wprev[x]) nw++;
            if (nw > MAX_WP) { fprintf(stderr, "align_oracle: %d waypoints\n", nw); abort(); }      /* cannot happen: they lie `piece` bases apart */
            cand[n_cand].n_wp = nw;
            int at = nw;
            for (int32_t x = wl[best_e]; x >= 0; x = wprev[x]) { at--; cand[n_cand].wi[at] = hits[wh[x]].i; cand[n_cand].wc[at] = hits[wh[x]].cp; }
            if (cand[n_cand].wi[0] != A->i || cand[n_cand].wc[0] != A->cp) { fprintf(stderr, "align_oracle: waypoint 0 is not the anchor\n"); abort(); }
        }
        n_cand++;
    }
    free(wh); free(f); free(st); free(wl); free(wprev);
    return n_cand;
}

/* ---- the banded extension DP from the origin cell (-1, -1) of q[0..nq) x t[0..nt): fills the masks / moves of its scratch set `sb`
 * (arrays sb, sb+1, sb+2), finds the best valid cell.  Used forward from the anchor and, on reversed sequences, backward from it. */
typedef struct { int64_t steps; int32_t score; int64_t ts, lane; uint64_t *tbD, *tbU; uint8_t *mv; int band; } dp_t;
static inline int band_of(const orc_align_params *P) { return P->band == 32 ? 32 : 64; }
/* inner (v1.6): the extension has to arrive at the sub-matrix's corner (nq-1, nt-1): a border cell is valued H - gap * (its distance to the corner) */
static dp_t dp_extend(const uint8_t *q, int64_t nq, const uint8_t *t, int64_t nt, const orc_align_params *P, int sb, int inner) {
    dp_t R;
    memset(&R, 0, sizeof R);
    const int64_t max_steps = nq + nt + 2;
    uint64_t *tbD = (uint64_t *)scratch_get(sb, (size_t)max_steps * 8), *tbU = (uint64_t *)scratch_get(sb + 1, (size_t)max_steps * 8);
    uint8_t *mv = (uint8_t *)scratch_get(sb + 2, (size_t)max_steps);
#define QC(i) (((i) >= 0 && (i) < nq) ? q[i] : 4)
#define TC(j) (((j) >= 0 && (j) < nt) ? t[j] : 5)
    int32_t Hp[W], X[W], H[W], bsc[W]; int64_t bt[W];
    int qc[W], tc[W];
    const int Wb = band_of(P), HB = Wb / 2;      /* lanes 0 .. Wb - 1; the origin's neighbours sit in lanes HB and HB + 1 */
    R.band = Wb;
    int64_t i0 = -(HB + 1);
    for (int kk = 0; kk < Wb; kk++) {
        X[kk] = kk == HB ? 0 : NEG;      /* H(-2), in the lane layout before the virtual RIGHT move of step -1 */
        Hp[kk] = (kk == HB || kk == HB + 1) ? -P->gap : NEG;
        int64_t i = kk - (HB + 1), j = HB - kk;
        qc[kk] = QC(i); tc[kk] = TC(j);
        bsc[kk] = NEG; bt[kk] = -1;
    }
    int steer = 1;
    int prev_down = 0;                   /* move of step -1 (virtual RIGHT) */
    int64_t tt = 0;
    for (;;) {
        int down = tt < Wb ? ((tt & 1) == 0) : steer;
        int32_t A[W], B[W], dg[W], Xn[W];   /* A: previous step, same lane; B: previous step, neighbour lane (both minus gap) */
        if (down) {
            i0++;
            for (int kk = 0; kk < Wb - 1; kk++) qc[kk] = qc[kk + 1];
            qc[Wb - 1] = QC(i0 + Wb - 1);
            for (int kk = 0; kk < Wb; kk++) { A[kk] = Hp[kk] - P->gap; B[kk] = (kk < Wb - 1 ? Hp[kk + 1] : NEG) - P->gap; }
            /* X = H(t-2) in ITS lane layout; the two moves since shift the diagonal predecessor of lane kk to
             * lane kk - 1 + (#DOWN among them): DOWN,DOWN -> kk+1; one of each -> kk; RIGHT,RIGHT -> kk-1 */
            for (int kk = 0; kk < Wb; kk++) { int s_ = kk + prev_down; dg[kk] = s_ < Wb ? X[s_] : NEG; Xn[kk] = Hp[kk]; }
        } else {
            for (int kk = Wb - 1; kk > 0; kk--) tc[kk] = tc[kk - 1];
            tc[0] = TC(tt - i0);
            for (int kk = 0; kk < Wb; kk++) { A[kk] = Hp[kk] - P->gap; B[kk] = (kk > 0 ? Hp[kk - 1] : NEG) - P->gap; }
            for (int kk = 0; kk < Wb; kk++) { int s_ = kk - 1 + prev_down; dg[kk] = s_ >= 0 ? X[s_] : NEG; Xn[kk] = Hp[kk]; }
        }
        uint64_t D = 0, U = 0;
        for (int kk = 0; kk < Wb; kk++) {
            int32_t s = qc[kk] == tc[kk] ? P->match : -P->mismatch;
            int32_t hd = dg[kk] + s;
            int32_t h = hd > A[kk] ? hd : A[kk]; if (B[kk] > h) h = B[kk];
            H[kk] = h;
            if (h == hd) D |= 1ull << kk;
            if (A[kk] >= B[kk]) U |= 1ull << kk;       /* "the gap comes from the same lane" */
            int64_t i = i0 + kk, j = tt - i;
            /* v1.5: the extension runs to a BORDER of the matrix -- the read's last row or the window's last column; its terminal is the best valid cell there */
            if (i >= 0 && i < nq && j >= 0 && j < nt && (i == nq - 1 || j == nt - 1)) {
                const int32_t v = inner ? h - P->gap * (int32_t)((nq - 1 - i) + (nt - 1 - j)) : h;
                if (v > bsc[kk]) { bsc[kk] = v; bt[kk] = tt; }
            }
        }
        tbD[tt] = D; tbU[tt] = U; mv[tt] = (uint8_t)down;
        steer = !(H[0] > H[Wb - 1]);
        prev_down = down;
        memcpy(X, Xn, sizeof X); memcpy(Hp, H, sizeof H);
        tt++;
        if (i0 > nq - 1) break;
        if ((tt - 1) - (i0 + Wb - 1) > nt - 1) break;
        if (tt >= max_steps) break;
    }
#undef QC
#undef TC
    R.steps = tt; R.tbD = tbD; R.tbU = tbU; R.mv = mv;
    /* terminal: max score among the valid border cells, then earliest step, then lowest lane */
    int bk = -1;
    for (int kk = 0; kk < Wb; kk++) {
        if (bt[kk] < 0) continue;
        if (bk < 0 || bsc[kk] > bsc[bk] || (bsc[kk] == bsc[bk] && bt[kk] < bt[bk])) bk = kk;
    }
    R.score = bk >= 0 ? bsc[bk] : NEG;
    R.ts = bk >= 0 ? bt[bk] : -1; R.lane = bk;
    return R;
}
/* walk the masks back from the DP's best cell: one byte per op, END first -- 0 column, 1 I (read base), 2 D (contig base).  *i_end, *j_end =
 * the best cell; *i_stop, *j_stop = where the walk left the matrix (one of them is -1).  Scratch array `sb` holds i0 per step. */
static int64_t dp_walk(const dp_t *R, int sb, uint8_t *ops, int64_t *i_end, int64_t *j_end, int64_t *i_stop, int64_t *j_stop) {
    int64_t *i0s = (int64_t *)scratch_get(sb, (size_t)R->steps * 8);
    int64_t cur = -(R->band / 2 + 1);
    for (int64_t s2 = 0; s2 < R->steps; s2++) { cur += R->mv[s2]; i0s[s2] = cur; }
    int64_t ts = R->ts;
    int64_t i = i0s[ts] + R->lane, j = ts - i, n = 0;
    *i_end = i; *j_end = j;
    while (i >= 0 && j >= 0) {
        int kk = (int)(i - i0s[ts]);
        if ((R->tbD[ts] >> kk) & 1) { ops[n++] = 0; i--; j--; ts -= 2; }
        else if ((((R->tbU[ts] >> kk) & 1) != 0) == (R->mv[ts] != 0)) { ops[n++] = 1; i--; ts -= 1; }   /* the cell above */
        else { ops[n++] = 2; j--; ts -= 1; }
    }
    *i_stop = i; *j_stop = j;
    return n;
}

/* ---- the path of one candidate (v1.6): the forward pieces from the last waypoint's free extension down to the anchor, then the backward extension
 * from the anchor out -- each piece its own DP and walk, joined by the gap moves their ends imply.  r = the oriented read codes.  ops (END first:
 * 0 column, 1 I, 2 D) go to scratch array `ob`; (*i_end, *j_end) = the path's last cell in read / contig coordinates. */
typedef struct { uint8_t *ops; int64_t L, i_end, j_end, cells; int32_t fwd_score, back_score; int ok; } path_t;
static path_t build_path(const ctg_index *ix, const uint8_t *r, int64_t n, const anchor_t *an, const orc_align_params *P, int ob) {
    path_t R;
    memset(&R, 0, sizeof R);
    R.back_score = NEG;
    const int64_t Lc = ix->len, piece = piece_len(n);
    const int64_t i_a = an->i_a, c_a = an->c_a;
    int64_t cap = 1024 + 2 * n + 2 * (n + n / 4 + 64) + 2 * (piece + piece / 4 + 64);      /* every piece: at most nq + nt ops and border moves */
    if (an->n_wp > 1) cap += 2 * (an->wc[an->n_wp - 1] - c_a);
    uint8_t *ops = (uint8_t *)scratch_get(ob, (size_t)cap);
    uint8_t *po = NULL;
    int64_t L = 0;
    R.ops = ops; R.ok = 1;
    for (int k = an->n_wp - 1; k >= 0; k--) {
        const int inner = k < an->n_wp - 1;
        const int64_t oi = an->wi[k], oc = an->wc[k];
        int64_t nq, nt;
        if (inner) { nq = an->wi[k + 1] - oi; nt = an->wc[k + 1] - oc; }
        else {
            nq = n - oi; if (nq > 2 * piece) nq = 2 * piece;
            nt = Lc - oc; if (nt > nq + nq / 4 + 64) nt = nq + nq / 4 + 64;
        }
        const dp_t F = dp_extend(r + oi, nq, ix->codes + oc, nt, P, 0, inner);
        R.cells += F.steps * band_of(P);
        if (F.lane < 0) { R.ok = 0; R.fwd_score = NEG; break; }          /* (no valid border cell: cannot happen for nq, nt >= 1) */
        R.fwd_score += F.score;
        po = (uint8_t *)scratch_get(15, (size_t)(nq + nt + 64));
        int64_t ie, je, is, js;
        const int64_t np = dp_walk(&F, 3, po, &ie, &je, &is, &js);
        if (!inner) { R.i_end = oi + ie; R.j_end = oc + je; }
        else {      /* from the sub-matrix's corner to the terminal, along the border */
            for (int64_t x = je; x < nt - 1; x++) ops[L++] = 2;
            for (int64_t x = ie; x < nq - 1; x++) ops[L++] = 1;
        }
        memcpy(ops + L, po, (size_t)np); L += np;
        for (int64_t x = 0; x <= js && is < 0; x++) ops[L++] = 2;     /* left through row -1: the contig bases 0..js were skipped */
        for (int64_t x = 0; x <= is && js < 0; x++) ops[L++] = 1;     /* left through column -1 */
    }
    if (i_a > 0 && c_a > 0) {
        const int64_t nqb = i_a < piece ? i_a : piece;
        int64_t ntb = c_a;
        if (ntb > nqb + nqb / 4 + 64) ntb = nqb + nqb / 4 + 64;
        uint8_t *qb = (uint8_t *)scratch_get(13, (size_t)nqb), *tb = (uint8_t *)scratch_get(14, (size_t)ntb);
        for (int64_t x = 0; x < nqb; x++) qb[x] = r[i_a - 1 - x];
        for (int64_t x = 0; x < ntb; x++) tb[x] = ix->codes[c_a - 1 - x];
        const dp_t B = dp_extend(qb, nqb, tb, ntb, P, 8, 0);
        R.cells += B.steps * band_of(P);
        if (B.lane >= 0 && R.ok) {
            R.back_score = B.score;
            po = (uint8_t *)scratch_get(15, (size_t)(nqb + ntb + 64));
            int64_t bie, bje, bis, bjs;
            const int64_t nb = dp_walk(&B, 11, po, &bie, &bje, &bis, &bjs);
            for (int64_t x = 0; x <= bjs && bis < 0; x++) ops[L++] = 2;      /* next to the anchor, as above */
            for (int64_t x = 0; x <= bis && bjs < 0; x++) ops[L++] = 1;
            for (int64_t x = nb - 1; x >= 0; x--) ops[L++] = po[x];         /* the walk came from the far end towards the anchor: turned round */
        }
    }
    R.L = L;
    return R;
}

/* ---- from the winner's path to its alignment: best sub-path, CIGAR, summary */
static void finish_path(const ctg_index *ix, const uint8_t *r, int64_t n, int strand, const path_t *Pt, const orc_align_params *P, orc_aln_summary *out, u32vec *cig) {
    const uint8_t *ops = Pt->ops;
    const int64_t L = Pt->L;
    int64_t i_end = Pt->i_end, j_end = Pt->j_end;
    out->score = 0;
    if (!Pt->ok) return;
    {
        /* v1.5 "best sub-path": the joined path runs from the forward terminal (op 0) to the backward one; with P(k) = score of ops 0..k-1, the
         * alignment is ops e..s with the largest P(s+1) - P(e)  (ties: the smallest s, then the largest e): both of its ends are match columns, what
         * the path holds outside it -- a tail dragged to the border through noise, a head likewise -- is soft clip. */
        int64_t e_best = -1, s_best = -1, bestS = 0;
        int64_t ci_e = 0, cj_e = 0;                 /* read / contig bases the ops before e_best consume */
        {
            int64_t i = i_end, j = j_end, Pk = 0, minP = 0, e_min = 0, ci = 0, cj = 0, ci_min = 0, cj_min = 0;
            for (int64_t x = 0; x < L; x++) {
                if (Pk <= minP) { minP = Pk; e_min = x; ci_min = ci; cj_min = cj; }          /* '<=': the largest e among equal prefixes */
                if (ops[x] == 0) { Pk += r[i] == ix->codes[j] ? P->match : -P->mismatch; i--; j--; ci++; cj++; }
                else if (ops[x] == 1) { Pk -= P->gap; i--; ci++; }
                else { Pk -= P->gap; j--; cj++; }
                if (Pk - minP > bestS) { bestS = Pk - minP; s_best = x; e_best = e_min; ci_e = ci_min; cj_e = cj_min; }
            }
        }
        if (s_best < 0) return;                     /* not one match column on the path */
        out->score = (int32_t)bestS;
        i_end -= ci_e; j_end -= cj_e;
        /* op by op from the alignment's END: cell (i, j) in read / contig coordinates, run-length encoding */
        int64_t i = i_end, j = j_end;
        u32vec rev = {0};
        int cur_op = -1; uint32_t cur_len = 0; int32_t ncol = 0, n_eq = 0;
        for (int64_t x = e_best; x <= s_best; x++) {
            int op;
            if (ops[x] == 0) { op = r[i] == ix->codes[j] ? 7 : 8; n_eq += op == 7; i--; j--; ncol++; }
            else if (ops[x] == 1) { op = 1; i--; }
            else { op = 2; j--; }
            if (op == cur_op) cur_len++;
            else { if (cur_len) push(&rev, (cur_len << 4) | (uint32_t)cur_op); cur_op = op; cur_len = 1; }
        }
        if (cur_len) push(&rev, (cur_len << 4) | (uint32_t)cur_op);
        {   /* the device derives the match count from the score of the kept ops (it only sees the bases while scoring them): both must agree */
            const int64_t num = bestS + (int64_t)P->mismatch * ncol + (int64_t)P->gap * ((i_end - i) + (j_end - j) - 2 * (int64_t)ncol);
            if (num % (P->match + P->mismatch) != 0 || num / (P->match + P->mismatch) != n_eq) {
                fprintf(stderr, "align_oracle: match-count identity violated (%lld vs %d)\n", (long long)num, n_eq);
                abort();
            }
        }
        int64_t q_lead = i + 1, r_lead = j + 1;        /* the first read / contig base of the path's kept stretch */
        /* forward order; strip leading / trailing non-match ops */
        int64_t a = rev.n - 1, b = 0;                   /* forward index f = rev[a - f] */
        while (a >= b && ((rev.v[a] & 15) == 1 || (rev.v[a] & 15) == 2)) {
            if ((rev.v[a] & 15) == 1) q_lead += rev.v[a] >> 4; else r_lead += rev.v[a] >> 4;
            a--;
        }
        int64_t q_trail = 0, r_trail = 0;
        while (b <= a && ((rev.v[b] & 15) == 1 || (rev.v[b] & 15) == 2)) {
            if ((rev.v[b] & 15) == 1) q_trail += rev.v[b] >> 4; else r_trail += rev.v[b] >> 4;
            b++;
        }
        /* the device emits diagonal runs as 'M' (the '=' / 'X' split is host work); the overflow rule and
         * n_cigar are defined on those merged runs */
        int64_t n_mraw = 0; int prev_m = -1;
        for (int64_t f = 0; f < rev.n; f++) {
            int o = (int)(rev.v[f] & 15); int mo = (o == 7 || o == 8) ? 0 : o;
            if (mo != prev_m) { n_mraw++; prev_m = mo; }
        }
        if (a >= b && ncol > 0 && (n_mraw <= n + 16)) {
            const int64_t pos = r_lead, ref_end = j_end + 1 - r_trail;
            const int64_t q_start = q_lead, q_end = i_end + 1 - q_trail;
            const int64_t aln_len = (q_end - q_start) + (ref_end - pos) - ncol;     /* columns + inserted + deleted bases */
            if (P->min_pct_identity <= 0 || 100 * (int64_t)n_eq >= (int64_t)P->min_pct_identity * aln_len) {
                out->aligned = 1;
                out->strand = strand;
                out->pos = (int32_t)pos;
                out->ref_end = (int32_t)ref_end;
                out->q_start = (int32_t)q_start;
                out->q_end = (int32_t)q_end;
                out->n_columns = ncol;
                out->n_match = n_eq;
                int32_t nc = 0; prev_m = -1;
                if (out->q_start > 0) { push(cig, ((uint32_t)out->q_start << 4) | 4u); nc++; }
                for (int64_t f = a; f >= b; f--) {
                    push(cig, rev.v[f]);
                    int o = (int)(rev.v[f] & 15); int mo = (o == 7 || o == 8) ? 0 : o;
                    if (mo != prev_m) { nc++; prev_m = mo; }
                }
                if (n - out->q_end > 0) { push(cig, ((uint32_t)(n - out->q_end) << 4) | 4u); nc++; }
                out->n_cigar = nc;   /* words of the device CIGAR (M runs); the =/X CIGAR pushed above has more */
            }
        }
        free(rev.v);
    }
}

/* One read: candidates, the path of each, the best forward score wins. */
static void align_one(const ctg_index *ix, const uint8_t *fwd, int64_t n, const orc_align_params *P,
                      orc_aln_summary *out, u32vec *cig) {
    memset(out, 0, sizeof *out);
    const int k = P->kmer;
    if (n < k || ix->len < k) return;
    anchor_t cand[2];
    const int nc = seed_candidates(ix, fwd, n, P, cand);
    if (nc == 0) return;
    uint8_t *ori[2];
    ori[0] = (uint8_t *)scratch_get(4, (size_t)n); ori[1] = (uint8_t *)scratch_get(5, (size_t)n);
    for (int64_t i = 0; i < n; i++) { ori[0][i] = fwd[i]; ori[1][i] = (uint8_t)(3 - fwd[n - 1 - i]); }
    /* every candidate is extended piece by piece; the FORWARD pieces' scores decide the selection (blasr --bestn 1) */
    int win = 0;
    int64_t cells = 0;
    path_t Pc[2];
    static int dbg = -1;
    if (dbg < 0) dbg = getenv("ORC_ALIGN_DEBUG") != NULL;
    for (int c = 0; c < nc; c++) {
        Pc[c] = build_path(ix, ori[cand[c].strand], n, &cand[c], P, c ? 17 : 12);       /* own ops array per candidate */
        cells += Pc[c].cells;
        if (dbg) fprintf(stderr, "cand %d: strand %d anchor (%lld, %lld), %d waypoints -> forward score %d\n", c, cand[c].strand, (long long)cand[c].i_a, (long long)cand[c].c_a, cand[c].n_wp, Pc[c].fwd_score);
        if (c > 0 && Pc[c].fwd_score > Pc[win].fwd_score) win = c;
    }
    u32vec wcig = {0};
    finish_path(ix, ori[cand[win].strand], n, cand[win].strand, &Pc[win], P, out, &wcig);
    out->cells = cells;
    if (!out->aligned) { out->score = 0; out->strand = 0; }
    else for (int64_t x = 0; x < wcig.n; x++) push(cig, wcig.v[x]);
    free(wcig.v);
}

void orc_align_params_default(orc_align_params *p) {
    memset(p, 0, sizeof *p);
    p->kmer = 16; p->seed_stride = 4; p->match = 2; p->mismatch = 4; p->gap = 3; p->min_seed_hits = 8;
    p->min_pct_identity = 70;
    p->seed_anchored = 1;
    { const char *e = getenv("FZP_SEED_ANCHORED"); if (e) p->seed_anchored = atoi(e) != 0; }      /* (A/B runs: the same switch as the library's) */
    p->band = ORC_DEFAULT_BAND;
    { const char *e = getenv("FZP_ALIGN_BAND"); if (e && (atoi(e) == 32 || atoi(e) == 64)) p->band = atoi(e); }
}

/* the contig's index entries, sorted by (key, position): v1.6 every 2nd position, v1.7 the contig's selected (anchored) k-mers */
static void build_index_entries(ctg_index *ix, const uint8_t *codes, int64_t ctg_len, const orc_align_params *P) {
    const int64_t nk = ctg_len >= P->kmer ? ctg_len - P->kmer + 1 : 0;
    int64_t *pos = (int64_t *)malloc((size_t)(nk ? nk : 1) * 8);
    int64_t np = 0;
    if (P->seed_anchored) np = selected_positions(codes, ctg_len, P->kmer, pos);
    else for (int64_t p = 0; p < nk; p += 2) pos[np++] = p;
    ix->n = np;
    ix->kp = (kp_t *)malloc((size_t)(np ? np : 1) * sizeof(kp_t));
    for (int64_t q = 0; q < np; q++) {
        const int64_t p = pos[q];
        uint32_t kf = kmer_at(codes, p, P->kmer), kr = rc_of(kf, P->kmer);
        ix->kp[q].key = kr < kf ? kr : kf;
        ix->kp[q].pos = (int32_t)((p << 1) | (kr < kf ? 1 : 0));
    }
    free(pos);
    qsort(ix->kp, (size_t)ix->n, sizeof(kp_t), cmp_kp);
}

/* All reads against ONE contig.  cigar_out: concatenated BAM-style words; cig_off[n_reads+1]. */
int orc_align_reads(const uint8_t *ctg_ascii, int64_t ctg_len, int64_t n_reads, const int64_t *read_off,
                    const uint8_t *read_ascii, const orc_align_params *P, orc_aln_summary *out,
                    uint32_t **cigar_out, int64_t *cig_off) {
    if (P->kmer < 8 || P->kmer > 16 || P->seed_stride < 1) return -1;
    ctg_index ix;
    uint8_t *codes = (uint8_t *)malloc((size_t)(ctg_len ? ctg_len : 1));
    for (int64_t i = 0; i < ctg_len; i++) codes[i] = (uint8_t)code_of(ctg_ascii[i]);
    ix.codes = codes; ix.len = ctg_len;
    build_index_entries(&ix, codes, ctg_len, P);
    u32vec cig = {0};
    cig_off[0] = 0;
    for (int64_t r = 0; r < n_reads; r++) {
        int64_t n = read_off[r + 1] - read_off[r];
        uint8_t *fwd = (uint8_t *)malloc((size_t)(n ? n : 1));
        for (int64_t i = 0; i < n; i++) fwd[i] = (uint8_t)code_of(read_ascii[read_off[r] + i]);
        align_one(&ix, fwd, n, P, &out[r], &cig);
        cig_off[r + 1] = cig.n;
        free(fwd);
    }
    scratch_release();
    free(ix.kp); free(codes);
    if (!cig.v) cig.v = (uint32_t *)malloc(4);
    *cigar_out = cig.v;
    return 0;
}

/* ---- the same over several host threads (the CPU baseline of bench.py uses every core): reads are dealt round-robin,
 * results are identical to orc_align_reads. */
typedef struct {
    const ctg_index *ix; const orc_align_params *P; int64_t n_reads; const int64_t *read_off; const uint8_t *read_ascii;
    orc_aln_summary *out; u32vec *cigs; int t, T;
} mt_arg;
static void *mt_worker(void *vp) {
    mt_arg *a = (mt_arg *)vp;
    for (int64_t r = a->t; r < a->n_reads; r += a->T) {
        int64_t n = a->read_off[r + 1] - a->read_off[r];
        uint8_t *fwd = (uint8_t *)malloc((size_t)(n ? n : 1));
        for (int64_t i = 0; i < n; i++) fwd[i] = (uint8_t)code_of(a->read_ascii[a->read_off[r] + i]);
        align_one(a->ix, fwd, n, a->P, &a->out[r], &a->cigs[r]);
        free(fwd);
    }
    scratch_release();
    return NULL;
}
/* test hook (tests/: the spec-independent full-matrix check needs the origin an extension started from; since v1.3 the reported alignment
 * no longer reveals it): the candidate origins of one read as seed_candidates finds them, out[c] = {strand, i_a, c_a, forward terminal score, backward terminal score (or -2^26)}; returns their number */
/* test hooks (tests/test_gpu_align.py: K1's intermediates on the device against these): the index as sorted entries key << 32 | (position << 1 | strand bit) -- the device
 * table's own entry format --, and a read's hit list in spec order as pairs (strand << 31 | oriented offset, contig position) */
int orc_debug_index(const uint8_t *ctg_ascii, int64_t ctg_len, const orc_align_params *P, uint64_t **entries, int64_t *n) {
    ctg_index ix;
    uint8_t *codes = (uint8_t *)malloc((size_t)(ctg_len ? ctg_len : 1));
    for (int64_t i = 0; i < ctg_len; i++) codes[i] = (uint8_t)code_of(ctg_ascii[i]);
    ix.codes = codes; ix.len = ctg_len;
    build_index_entries(&ix, codes, ctg_len, P);
    uint64_t *o = (uint64_t *)malloc((size_t)(ix.n ? ix.n : 1) * 8);
    for (int64_t q = 0; q < ix.n; q++) o[q] = ((uint64_t)ix.kp[q].key << 32) | (uint32_t)ix.kp[q].pos;
    *entries = o; *n = ix.n;
    free(ix.kp); free(codes);
    return 0;
}
int orc_debug_hits(const uint8_t *ctg_ascii, int64_t ctg_len, const uint8_t *read_ascii, int64_t n, const orc_align_params *P, uint32_t *out, int64_t *n_hits) {
    ctg_index ix;
    uint8_t *codes = (uint8_t *)malloc((size_t)(ctg_len ? ctg_len : 1));
    for (int64_t i = 0; i < ctg_len; i++) codes[i] = (uint8_t)code_of(ctg_ascii[i]);
    ix.codes = codes; ix.len = ctg_len;
    build_index_entries(&ix, codes, ctg_len, P);
    uint8_t *fwd = (uint8_t *)malloc((size_t)(n ? n : 1));
    for (int64_t i = 0; i < n; i++) fwd[i] = (uint8_t)code_of(read_ascii[i]);
    anchor_t cand[2];
    g_dbg_hits = (hit_t *)malloc((size_t)HIT_CAP * sizeof(hit_t)); g_dbg_nh = 0;
    if (n >= P->kmer && ix.len >= P->kmer) (void)seed_candidates(&ix, fwd, n, P, cand);
    for (int64_t h = 0; h < g_dbg_nh; h++) { out[2 * h] = ((uint32_t)g_dbg_hits[h].s << 31) | (uint32_t)g_dbg_hits[h].i; out[2 * h + 1] = (uint32_t)g_dbg_hits[h].cp; }
    *n_hits = g_dbg_nh;
    free(g_dbg_hits); g_dbg_hits = NULL;
    scratch_release();
    free(fwd); free(ix.kp); free(codes);
    return 0;
}

int orc_align_origins(const uint8_t *ctg_ascii, int64_t ctg_len, const uint8_t *read_ascii, int64_t n, const orc_align_params *P, int64_t *out) {
    if (P->kmer < 8 || P->kmer > 16 || P->seed_stride < 1) return -1;
    ctg_index ix;
    uint8_t *codes = (uint8_t *)malloc((size_t)(ctg_len ? ctg_len : 1));
    for (int64_t i = 0; i < ctg_len; i++) codes[i] = (uint8_t)code_of(ctg_ascii[i]);
    ix.codes = codes; ix.len = ctg_len;
    build_index_entries(&ix, codes, ctg_len, P);
    uint8_t *fwd = (uint8_t *)malloc((size_t)(n ? n : 1));
    for (int64_t i = 0; i < n; i++) fwd[i] = (uint8_t)code_of(read_ascii[i]);
    anchor_t cand[2];
    int nc = 0;
    if (n >= P->kmer && ix.len >= P->kmer) nc = seed_candidates(&ix, fwd, n, P, cand);
    for (int c = 0; c < nc; c++) {
        out[5 * c] = cand[c].strand; out[5 * c + 1] = cand[c].i_a; out[5 * c + 2] = cand[c].c_a;
        /* the extensions' scores at their terminals (forward: summed over the pieces), as align_one runs them */
        uint8_t *ori = (uint8_t *)malloc((size_t)(n ? n : 1));
        for (int64_t i = 0; i < n; i++) ori[i] = cand[c].strand ? (uint8_t)(3 - fwd[n - 1 - i]) : fwd[i];
        const path_t Pt = build_path(&ix, ori, n, &cand[c], P, 12);
        out[5 * c + 3] = Pt.ok ? Pt.fwd_score : NEG;
        out[5 * c + 4] = Pt.back_score;
        free(ori);
    }
    scratch_release();
    free(fwd); free(ix.kp); free(codes);
    return nc;
}

/* test hook (tests/test_swb_core.py: the bit-sliced cell function of falcon_unzip_amd/csrc/fzp_swb_core.h against this scalar DP): one extension of
 * q[0..nq) x t[0..nt) (codes 0..3) -- per step the two masks and the move, out = {steps, terminal score, terminal step, terminal lane} */
int orc_dp_extend_raw(const uint8_t *q, int64_t nq, const uint8_t *t, int64_t nt, const orc_align_params *P, uint64_t *tbD, uint64_t *tbU, uint8_t *mv, int64_t *out, int inner) {
    const dp_t R = dp_extend(q, nq, t, nt, P, 0, inner);
    memcpy(tbD, R.tbD, (size_t)R.steps * 8); memcpy(tbU, R.tbU, (size_t)R.steps * 8); memcpy(mv, R.mv, (size_t)R.steps);
    out[0] = R.steps; out[1] = R.score; out[2] = R.ts; out[3] = R.lane;
    scratch_release();
    return 0;
}

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
/* seconds[0] = index build (serial, one contig), seconds[1] = seeding + DP + trace-back of all reads (n_threads threads); may be NULL */
int orc_align_reads_mt_timed(const uint8_t *ctg_ascii, int64_t ctg_len, int64_t n_reads, const int64_t *read_off,
                             const uint8_t *read_ascii, const orc_align_params *P, orc_aln_summary *out,
                             uint32_t **cigar_out, int64_t *cig_off, int n_threads, double *seconds) {
    if (P->kmer < 8 || P->kmer > 16 || P->seed_stride < 1) return -1;
    const double t_begin = now_s();
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    ctg_index ix;
    uint8_t *codes = (uint8_t *)malloc((size_t)(ctg_len ? ctg_len : 1));
    for (int64_t i = 0; i < ctg_len; i++) codes[i] = (uint8_t)code_of(ctg_ascii[i]);
    ix.codes = codes; ix.len = ctg_len;
    build_index_entries(&ix, codes, ctg_len, P);
    const double t_indexed = now_s();
    u32vec *cigs = (u32vec *)calloc((size_t)(n_reads ? n_reads : 1), sizeof(u32vec));
    pthread_t th[256];
    mt_arg args[256];
    for (int t = 0; t < n_threads; t++) {
        mt_arg a = {&ix, P, n_reads, read_off, read_ascii, out, cigs, t, n_threads};
        args[t] = a;
        pthread_create(&th[t], NULL, mt_worker, &args[t]);
    }
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    if (seconds) { seconds[0] = t_indexed - t_begin; seconds[1] = now_s() - t_indexed; }
    int64_t total = 0;
    cig_off[0] = 0;
    for (int64_t r = 0; r < n_reads; r++) { total += cigs[r].n; cig_off[r + 1] = total; }
    uint32_t *all = (uint32_t *)malloc((size_t)(total ? total : 1) * 4);
    for (int64_t r = 0; r < n_reads; r++) { if (cigs[r].n) memcpy(all + cig_off[r], cigs[r].v, (size_t)cigs[r].n * 4); free(cigs[r].v); }
    free(cigs); free(ix.kp); free(codes);
    *cigar_out = all;
    return 0;
}
int orc_align_reads_mt(const uint8_t *ctg_ascii, int64_t ctg_len, int64_t n_reads, const int64_t *read_off,
                       const uint8_t *read_ascii, const orc_align_params *P, orc_aln_summary *out,
                       uint32_t **cigar_out, int64_t *cig_off, int n_threads) {
    return orc_align_reads_mt_timed(ctg_ascii, ctg_len, n_reads, read_off, read_ascii, P, out, cigar_out, cig_off, n_threads, NULL);
}
