/*
 * align_oracle.c -- scalar CPU twin of the K1 read->contig aligner.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED vs the reference: FALCON_unzip shells out to `blasr` (falcon_unzip/unzip.py:86-88),
 * a third-party C++ aligner that is not vendored under /root/reference and whose results
 * (placement, clipping, `--hitPolicy randombest --randomSeed 42` tie-breaks) cannot be reproduced
 * here.  This file therefore DEFINES the aligner ("fzalign v1.1"); the HIP kernels in
 * falcon_unzip_amd/csrc/fzp_align.hip must match it bit-for-bit (summaries, CIGARs, DP cell counts),
 * and its quality is judged against the simulator's true alignments.
 *
 * fzalign v1.1  (v1 dropped the diagonal predecessor of lane 63 after a RIGHT move followed by a DOWN move)
 *   bases     A/a C/c G/g T/t -> 0..3, anything else -> 0
 *   seeding   canonical k-mers (k<=16, 2 bits/base, base m of a k-mer at bits 2m; canonical = the smaller of
 *             the k-mer and its reverse complement) of every 2nd contig position -> smallest start position
 *             (+ whether the canonical form was the reverse complement); every `stride`-th FORWARD read
 *             k-mer is looked up once: equal orientation bits -> the read matches as sequenced (strand 0,
 *             oriented offset i = pf), different -> its reverse complement does (i = n-k-pf); it votes for
 *             bin = (cpos - i + n) >> shift, shift = smallest s>=10 with ((Lc+n)>>s)+2 <= 8192;
 *             best (strand, bin) maximises votes[bin]+votes[bin+1] (ties: forward strand, lower bin);
 *             fewer than min_seed_hits votes -> unaligned.  The hit with the smallest read offset inside
 *             the two winning bins fixes the diagonal d = cpos - i; the extension starts at the read's
 *             first base on that diagonal: origin (max(0,-d), max(0,d)).
 *   extension adaptive anti-diagonal band of 64 cells (Suzuki-Kasahara style), forward from that
 *             origin: linear gaps, H = max(diag + (match | -mismatch), up - gap, left - gap), no zero
 *             floor; the first 64 steps alternate down/right, afterwards the band moves RIGHT when
 *             H[lane 0] > H[lane 63], else DOWN.  The diagonal operand is H of two steps ago in that
 *             step's own lane layout: the predecessor of lane k sits in lane k - 1 + (number of DOWN
 *             moves among the last two); a lane outside 0..63 reads as minus infinity.  The alignment ends at the best-scoring valid cell
 *             (first in step order, then lowest lane); read bases before the anchor and after the
 *             end are soft-clipped.  Trace-back priority: diagonal, then the gap whose source is the
 *             same lane of the previous step (the cell above after a DOWN move, the cell to the left
 *             after a RIGHT move), then the other gap.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define W 64
#define NEG (-(1 << 26))

typedef struct {
    int32_t kmer, seed_stride, match, mismatch, gap, min_seed_hits;
    int32_t reserved[10];
} orc_align_params;

typedef struct {
    int32_t aligned, strand, pos, ref_end, q_start, q_end, score, n_cigar;
    int64_t cells;
    int32_t n_columns, pad_;
} orc_aln_summary;

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

/* -> (position << 1 | orientation bit) of the smallest position holding the canonical key, or -1 */
static int32_t index_lookup(const ctg_index *ix, uint32_t key) {
    int64_t lo = 0, hi = ix->n;
    while (lo < hi) { int64_t m = (lo + hi) >> 1; if (ix->kp[m].key < key) lo = m + 1; else hi = m; }
    return (lo < ix->n && ix->kp[lo].key == key) ? ix->kp[lo].pos : -1;
}

typedef struct { uint32_t *v; int64_t n, cap; } u32vec;
static void push(u32vec *v, uint32_t x) {
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1 << 16; v->v = (uint32_t *)realloc(v->v, (size_t)v->cap * 4); }
    v->v[v->n++] = x;
}

/* One read.  r = oriented read codes are derived on the fly from fwd codes. */
static void align_one(const ctg_index *ix, const uint8_t *fwd, int64_t n, const orc_align_params *P,
                      orc_aln_summary *out, u32vec *cig) {
    memset(out, 0, sizeof *out);
    const int k = P->kmer, stride = P->seed_stride;
    const int64_t Lc = ix->len;
    if (n < k || Lc < k) return;
    uint8_t *ori[2];
    ori[0] = (uint8_t *)malloc((size_t)n); ori[1] = (uint8_t *)malloc((size_t)n);
    for (int64_t i = 0; i < n; i++) { ori[0][i] = fwd[i]; ori[1][i] = (uint8_t)(3 - fwd[n - 1 - i]); }
    int shift = 10;
    while ((((Lc + n) >> shift) + 2) > 8192) shift++;
    const int64_t NB = ((Lc + n) >> shift) + 2;
    uint32_t *votes = (uint32_t *)calloc((size_t)(2 * NB), 4);
    for (int64_t pf = 0; pf + k <= n; pf += stride) {
        uint32_t kf = kmer_at(ori[0], pf, k), kr = rc_of(kf, k);
        uint32_t orr = kr < kf ? 1u : 0u;
        int32_t hit = index_lookup(ix, kr < kf ? kr : kf);
        if (hit < 0) continue;
        int s = (int)(((uint32_t)hit & 1u) ^ orr);
        int64_t cp = (uint32_t)hit >> 1, i = s ? n - k - pf : pf;
        votes[s * NB + ((cp - i + n) >> shift)]++;
    }
    uint32_t best = 0; int bs_ = 0; int64_t bb = 0;
    for (int s = 0; s < 2; s++)
        for (int64_t b = 0; b + 1 < NB; b++) {
            uint32_t sc = votes[s * NB + b] + votes[s * NB + b + 1];
            if (sc > best) { best = sc; bs_ = s; bb = b; }
        }
    free(votes);
    if ((int32_t)best < P->min_seed_hits || best == 0) { free(ori[0]); free(ori[1]); return; }
    const uint8_t *r = ori[bs_];
    int64_t i_a = -1, c_a = -1;
    for (int64_t pf = 0; pf + k <= n; pf += stride) {      /* smallest oriented offset, then smallest position */
        uint32_t kf = kmer_at(ori[0], pf, k), kr = rc_of(kf, k);
        uint32_t orr = kr < kf ? 1u : 0u;
        int32_t hit = index_lookup(ix, kr < kf ? kr : kf);
        if (hit < 0) continue;
        int s = (int)(((uint32_t)hit & 1u) ^ orr);
        if (s != bs_) continue;
        int64_t cp = (uint32_t)hit >> 1, i = s ? n - k - pf : pf;
        int64_t b = (cp - i + n) >> shift;
        if ((b == bb || b == bb + 1) && (i_a < 0 || i < i_a || (i == i_a && cp < c_a))) { i_a = i; c_a = cp; }
    }
    if (i_a < 0) { free(ori[0]); free(ori[1]); return; }
    {   /* the seed fixes the diagonal; the extension starts at the read's first base on that diagonal
         * (or at the contig's first base when the read overhangs it) */
        int64_t d = c_a - i_a;
        i_a = d < 0 ? -d : 0;
        c_a = d < 0 ? 0 : d;
    }

    /* ---- adaptive banded extension from the anchor */
    const uint8_t *q = r + i_a;
    const int64_t nq = n - i_a;
    int64_t nt = Lc - c_a;
    if (nt > nq + nq / 4 + 64) nt = nq + nq / 4 + 64;
    const uint8_t *t = ix->codes + c_a;
    const int64_t max_steps = nq + nt + 2;
    uint64_t *tbD = (uint64_t *)malloc((size_t)max_steps * 8), *tbU = (uint64_t *)malloc((size_t)max_steps * 8);
    uint8_t *mv = (uint8_t *)malloc((size_t)max_steps);
#define QC(i) (((i) >= 0 && (i) < nq) ? q[i] : 4)
#define TC(j) (((j) >= 0 && (j) < nt) ? t[j] : 5)
    int32_t Hp[W], X[W], H[W], bsc[W]; int64_t bt[W];
    int qc[W], tc[W];
    int64_t i0 = -33;
    for (int kk = 0; kk < W; kk++) {
        X[kk] = kk == 32 ? 0 : NEG;      /* H(-2), in the lane layout before the virtual RIGHT move of step -1 */
        Hp[kk] = (kk == 32 || kk == 33) ? -P->gap : NEG;
        int64_t i = kk - 33, j = 32 - kk;
        qc[kk] = QC(i); tc[kk] = TC(j);
        bsc[kk] = NEG; bt[kk] = -1;
    }
    int steer = 1;
    int prev_down = 0;                   /* move of step -1 (virtual RIGHT) */
    int64_t tt = 0;
    for (;;) {
        int down = tt < 64 ? ((tt & 1) == 0) : steer;
        int32_t A[W], B[W], dg[W], Xn[W];   /* A: previous step, same lane; B: previous step, neighbour lane (both minus gap) */
        if (down) {
            i0++;
            for (int kk = 0; kk < W - 1; kk++) qc[kk] = qc[kk + 1];
            qc[W - 1] = QC(i0 + 63);
            for (int kk = 0; kk < W; kk++) { A[kk] = Hp[kk] - P->gap; B[kk] = (kk < W - 1 ? Hp[kk + 1] : NEG) - P->gap; }
            /* X = H(t-2) in ITS lane layout; the two moves since shift the diagonal predecessor of lane kk to
             * lane kk - 1 + (#DOWN among them): DOWN,DOWN -> kk+1; one of each -> kk; RIGHT,RIGHT -> kk-1 */
            for (int kk = 0; kk < W; kk++) { int s_ = kk + prev_down; dg[kk] = s_ < W ? X[s_] : NEG; Xn[kk] = Hp[kk]; }
        } else {
            for (int kk = W - 1; kk > 0; kk--) tc[kk] = tc[kk - 1];
            tc[0] = TC(tt - i0);
            for (int kk = 0; kk < W; kk++) { A[kk] = Hp[kk] - P->gap; B[kk] = (kk > 0 ? Hp[kk - 1] : NEG) - P->gap; }
            for (int kk = 0; kk < W; kk++) { int s_ = kk - 1 + prev_down; dg[kk] = s_ >= 0 ? X[s_] : NEG; Xn[kk] = Hp[kk]; }
        }
        uint64_t D = 0, U = 0;
        for (int kk = 0; kk < W; kk++) {
            int32_t s = qc[kk] == tc[kk] ? P->match : -P->mismatch;
            int32_t hd = dg[kk] + s;
            int32_t h = hd > A[kk] ? hd : A[kk]; if (B[kk] > h) h = B[kk];
            H[kk] = h;
            if (h == hd) D |= 1ull << kk;
            if (A[kk] >= B[kk]) U |= 1ull << kk;       /* "the gap comes from the same lane" */
            int64_t i = i0 + kk, j = tt - i;
            if (i >= 0 && i < nq && j >= 0 && j < nt && h > bsc[kk]) { bsc[kk] = h; bt[kk] = tt; }
        }
        tbD[tt] = D; tbU[tt] = U; mv[tt] = (uint8_t)down;
        steer = !(H[0] > H[W - 1]);
        prev_down = down;
        memcpy(X, Xn, sizeof X); memcpy(Hp, H, sizeof H);
        tt++;
        if (i0 > nq - 1) break;
        if ((tt - 1) - (i0 + 63) > nt - 1) break;
        if (tt >= max_steps) break;
    }
    const int64_t steps = tt;
    out->cells = steps * W;
    /* best cell: max score, then earliest step, then lowest lane */
    int bk = -1;
    for (int kk = 0; kk < W; kk++) {
        if (bt[kk] < 0) continue;
        if (bk < 0 || bsc[kk] > bsc[bk] || (bsc[kk] == bsc[bk] && bt[kk] < bt[bk])) bk = kk;
    }
    if (bk < 0 || bsc[bk] <= 0) goto done;
    {
        /* i0 at every step: replay the moves */
        int64_t *i0s = (int64_t *)malloc((size_t)steps * 8);
        int64_t cur = -33;
        for (int64_t s2 = 0; s2 < steps; s2++) { cur += mv[s2]; i0s[s2] = cur; }
        int64_t ts = bt[bk];
        int64_t i = i0s[ts] + bk, j = ts - i;
        const int64_t i_end = i, j_end = j;
        /* reversed raw op stream, run-length encoded on the fly: op codes 7 '=', 8 'X', 1 'I', 2 'D' */
        u32vec rev = {0};
        int cur_op = -1; uint32_t cur_len = 0; int32_t ncol = 0;
        while (i >= 0 && j >= 0) {
            int kk = (int)(i - i0s[ts]);
            int op;
            if ((tbD[ts] >> kk) & 1) { op = q[i] == t[j] ? 7 : 8; i--; j--; ts -= 2; ncol++; }
            else if ((((tbU[ts] >> kk) & 1) != 0) == (mv[ts] != 0)) { op = 1; i--; ts -= 1; }   /* the cell above */
            else { op = 2; j--; ts -= 1; }
            if (op == cur_op) cur_len++;
            else { if (cur_len) push(&rev, (cur_len << 4) | (uint32_t)cur_op); cur_op = op; cur_len = 1; }
        }
        if (cur_len) push(&rev, (cur_len << 4) | (uint32_t)cur_op);
        free(i0s);
        int64_t q_lead = i + 1, r_lead = j + 1;        /* bases before the first path op */
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
            out->aligned = 1;
            out->strand = bs_;
            out->pos = (int32_t)(c_a + r_lead);
            out->ref_end = (int32_t)(c_a + j_end + 1 - r_trail);
            out->q_start = (int32_t)(i_a + q_lead);
            out->q_end = (int32_t)(i_a + i_end + 1 - q_trail);
            out->score = bsc[bk];
            out->n_columns = ncol;
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
        free(rev.v);
    }
done:
    free(tbD); free(tbU); free(mv); free(ori[0]); free(ori[1]);
#undef QC
#undef TC
}

void orc_align_params_default(orc_align_params *p) {
    memset(p, 0, sizeof *p);
    p->kmer = 16; p->seed_stride = 4; p->match = 2; p->mismatch = 4; p->gap = 3; p->min_seed_hits = 8;
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
    {
        int64_t nk = ctg_len >= P->kmer ? ctg_len - P->kmer + 1 : 0;
        ix.n = (nk + 1) / 2;                                  /* every 2nd position */
        ix.kp = (kp_t *)malloc((size_t)(ix.n ? ix.n : 1) * sizeof(kp_t));
        for (int64_t q = 0; q < ix.n; q++) {
            int64_t p = 2 * q;
            uint32_t kf = kmer_at(codes, p, P->kmer), kr = rc_of(kf, P->kmer);
            ix.kp[q].key = kr < kf ? kr : kf;
            ix.kp[q].pos = (int32_t)((p << 1) | (kr < kf ? 1 : 0));
        }
    }
    qsort(ix.kp, (size_t)ix.n, sizeof(kp_t), cmp_kp);
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
    free(ix.kp); free(codes);
    if (!cig.v) cig.v = (uint32_t *)malloc(4);
    *cigar_out = cig.v;
    return 0;
}
