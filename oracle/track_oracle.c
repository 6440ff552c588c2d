/* track_oracle.c -- CPU restatement of falcon_unzip/rr_hctg_track.py (run_track_reads, :68-139; tr_stage1, :31-66):
 * for every raw read that appears as the B-read of an overlap keep its `bestn` best A-reads (by overlap length, then
 * A-read id string -- the heap of (overlap_len, q_id) tuples, :60-64, :99-106), then score the contigs those A-reads map
 * to.  TEST INFRASTRUCTURE (tests/, smoke, bench cpu_baseline only).  Pinned against tests/golden_ovlp/t*_ (outputs of
 * the reference itself, make_golden_track.py).
 *
 * The reference's line order is a dict order (:111) and contigs of equal score keep dict order (:125-126): both are
 * unspecified, so this restatement -- like the fixtures -- emits the CANONICAL order: lines sorted by (read id string,
 * score, contig string), ranks assigned in that order.
 */
#include <ctype.h>
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { const char *p; int n; } ttok;
static int t_space(int c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f' || c == '\n'; }
static int t_split(const char *s, const char *e, ttok *out, int max) {
    int n = 0;
    while (s < e) {
        while (s < e && t_space((unsigned char)*s)) s++;
        if (s >= e) break;
        const char *b = s;
        while (s < e && !t_space((unsigned char)*s)) s++;
        if (n < max) { out[n].p = b; out[n].n = (int)(s - b); }
        n++;
    }
    return n;
}
static int t_eq(ttok a, ttok b) { return a.n == b.n && memcmp(a.p, b.p, (size_t)a.n) == 0; }
static int t_cmp(ttok a, ttok b) { int m = a.n < b.n ? a.n : b.n; int c = memcmp(a.p, b.p, (size_t)m); return c ? c : a.n - b.n; }
static uint64_t t_hash(ttok t) { uint64_t h = 1469598103934665603ull; for (int i = 0; i < t.n; i++) { h ^= (unsigned char)t.p[i]; h *= 1099511628211ull; } return h; }
static int t_int(ttok t, long long *out) {
    if (t.n <= 0 || t.n > 30) return -1;
    char b[32]; memcpy(b, t.p, (size_t)t.n); b[t.n] = 0;
    int i = (b[0] == '+' || b[0] == '-') ? 1 : 0;
    if (!b[i]) return -1;
    for (int k = i; b[k]; k++) if (!isdigit((unsigned char)b[k])) return -1;
    errno = 0; *out = strtoll(b, NULL, 10);
    return errno ? -1 : 0;
}
static int t_float_ok(ttok t) {
    if (t.n <= 0 || t.n > 62) return 0;
    char b[64]; memcpy(b, t.p, (size_t)t.n); b[t.n] = 0;
    for (int k = 0; b[k]; k++) if (b[k] == 'x' || b[k] == 'X' || b[k] == 'p' || b[k] == 'P') return 0;
    char *e; (void)strtod(b, &e);
    return !(*e || e == b);
}

/* string-keyed table: key -> dense index (first appearance) */
typedef struct { ttok *key; int64_t n, cap; int64_t *slot; int64_t n_slot; } stab;
static void stab_init(stab *t, int64_t cap) {
    t->cap = cap > 16 ? cap : 16; t->n = 0; t->key = (ttok *)malloc((size_t)t->cap * sizeof(ttok));
    t->n_slot = 32; while (t->n_slot < 2 * t->cap) t->n_slot <<= 1;
    t->slot = (int64_t *)malloc((size_t)t->n_slot * sizeof(int64_t));
    for (int64_t i = 0; i < t->n_slot; i++) t->slot[i] = -1;
}
static int64_t stab_find(const stab *t, ttok k) {
    uint64_t h = t_hash(k) & (uint64_t)(t->n_slot - 1);
    for (;;) { int64_t e = t->slot[h]; if (e < 0) return -1; if (t_eq(t->key[e], k)) return e; h = (h + 1) & (uint64_t)(t->n_slot - 1); }
}
static int64_t stab_add(stab *t, ttok k) {          /* capacity is sized by the caller */
    int64_t e = stab_find(t, k);
    if (e >= 0) return e;
    e = t->n++; t->key[e] = k;
    uint64_t h = t_hash(k) & (uint64_t)(t->n_slot - 1);
    while (t->slot[h] >= 0) h = (h + 1) & (uint64_t)(t->n_slot - 1);
    t->slot[h] = e;
    return e;
}
static int64_t count_lines(const char *s, size_t n) { int64_t c = 1; for (size_t i = 0; i < n; i++) c += s[i] == '\n'; return c; }

typedef struct { int has; int64_t ctg; long long block, phase; } phase_t;
typedef struct { long long ovl; ttok q; int64_t t; } hit_t;
static int hit_cmp(const void *a_, const void *b_) {         /* by B-read, then best first */
    const hit_t *a = (const hit_t *)a_, *b = (const hit_t *)b_;
    if (a->t != b->t) return a->t < b->t ? -1 : 1;
    if (a->ovl != b->ovl) return a->ovl > b->ovl ? -1 : 1;
    return -t_cmp(a->q, b->q);
}
typedef struct { ttok bread, ctg; long long count, score; int in_ctg; } orow;
static int orow_cmp(const void *a_, const void *b_) {
    const orow *a = (const orow *)a_, *b = (const orow *)b_;
    int c = t_cmp(a->bread, b->bread);
    if (c) return c;
    if (a->score != b->score) return a->score < b->score ? -1 : 1;
    return t_cmp(a->ctg, b->ctg);
}

int orc_track_reads(int n_files, const char *const *texts, const size_t *lens, const char *phased_reads, size_t pr_len, const char *r2c, size_t r2c_len,
                    const char *rawread_ids, size_t ri_len, long long min_len, long long bestn, char **out_txt, size_t *out_len) {
    /* ---- rid -> set of contigs (:14-23): pairs (rid, ctg), distinct */
    stab rids, ctgs;
    stab_init(&rids, count_lines(r2c, r2c_len)); stab_init(&ctgs, count_lines(r2c, r2c_len) + count_lines(phased_reads, pr_len));
    int64_t n_pair = 0, pcap = count_lines(r2c, r2c_len);
    int64_t *pair_r = (int64_t *)malloc((size_t)pcap * sizeof(int64_t)), *pair_c = (int64_t *)malloc((size_t)pcap * sizeof(int64_t));
    for (const char *s = r2c, *end = r2c + r2c_len; s < end;) {
        const char *nl = memchr(s, '\n', (size_t)(end - s)); const char *e = nl ? nl : end;
        ttok t[6]; int nt = t_split(s, e, t, 6);
        s = nl ? nl + 1 : end;
        if (nt != 4) return -1;                                        /* `pid, rid, oid, ctg = row` */
        int64_t r = stab_add(&rids, t[1]), c = stab_add(&ctgs, t[3]);
        int dup = 0;
        for (int64_t k = 0; k < n_pair && !dup; k++) dup = pair_r[k] == r && pair_c[k] == c;    /* a set */
        if (!dup) { pair_r[n_pair] = r; pair_c[n_pair] = c; n_pair++; }
    }
    /* ---- oid -> phase (:73-81), rid -> oid -> phase (:82-86) */
    stab oids; stab_init(&oids, count_lines(phased_reads, pr_len));
    phase_t *oph = (phase_t *)calloc((size_t)count_lines(phased_reads, pr_len) + 1, sizeof(phase_t));
    for (const char *s = phased_reads, *end = phased_reads + pr_len; s < end;) {
        const char *nl = memchr(s, '\n', (size_t)(end - s)); const char *e = nl ? nl : end;
        ttok t[8]; int nt = t_split(s, e, t, 8);
        s = nl ? nl + 1 : end;
        if (nt < 7) return -1;
        long long b, p;
        if (t_int(t[2], &b) || t_int(t[3], &p)) return -1;
        int64_t o = stab_add(&oids, t[6]);
        oph[o].has = 1; oph[o].ctg = stab_add(&ctgs, t[1]); oph[o].block = b; oph[o].phase = p;
    }
    int64_t n_rid = 1;
    for (size_t i = 0; i < ri_len; i++) n_rid += rawread_ids[i] == '\n';       /* str.split('\n') */
    phase_t *rph = (phase_t *)calloc((size_t)n_rid, sizeof(phase_t));
    {
        int64_t r = 0; size_t b = 0;
        for (size_t i = 0; i <= ri_len; i++)
            if (i == ri_len || rawread_ids[i] == '\n') {
                ttok o = {rawread_ids + b, (int)(i - b)};
                int64_t k = stab_find(&oids, o);
                if (k >= 0) rph[r] = oph[k];
                r++; b = i + 1;
            }
    }
    /* ---- tr_stage1 over every file (:31-66): hits that pass, grouped by B-read string */
    int64_t n_line = 0;
    for (int k = 0; k < n_files; k++) n_line += count_lines(texts[k], lens[k]);
    stab breads; stab_init(&breads, n_line);
    hit_t *hit = (hit_t *)malloc((size_t)(n_line ? n_line : 1) * sizeof(hit_t));
    int64_t n_hit = 0;
    for (int k = 0; k < n_files; k++)
        for (const char *s = texts[k], *end = texts[k] + lens[k]; s < end;) {
            const char *nl = memchr(s, '\n', (size_t)(end - s)); const char *e = nl ? nl : end;
            ttok t[16]; int nt = t_split(s, e, t, 16);
            s = nl ? nl + 1 : end;
            if (nt < 12) return -1;
            long long v[7]; static const int col[7] = {2, 5, 6, 7, 9, 10, 11};
            for (int c = 0; c < 7; c++) if (t_int(t[col[c]], &v[c])) return -1;
            if (!t_float_ok(t[3])) return -1;
            if (v[6] < min_len) continue;                              /* t_l */
            if (stab_find(&rids, t[0]) < 0) continue;
            long long ti, qi;
            if (t_int(t[1], &ti)) return -1;
            if (ti < 0) ti += n_rid;
            if (ti < 0 || ti >= n_rid) return -1;                      /* IndexError */
            const phase_t tp = rph[ti];
            if (tp.has && tp.block != -1) {
                if (t_int(t[0], &qi)) return -1;
                if (qi < 0) qi += n_rid;
                if (qi < 0 || qi >= n_rid) return -1;
                const phase_t qp = rph[qi];
                if (qp.has && qp.ctg == tp.ctg && qp.block == tp.block && qp.phase != tp.phase) continue;
            }
            hit[n_hit].ovl = -v[0]; hit[n_hit].q = t[0]; hit[n_hit].t = stab_add(&breads, t[1]); n_hit++;
        }
    qsort(hit, (size_t)n_hit, sizeof(hit_t), hit_cmp);
    /* ---- per B-read: the bestn best hits, contig scores (:108-135) */
    orow *rows = (orow *)malloc((size_t)(n_hit * 2 + n_pair + 1) * sizeof(orow));
    int64_t n_rows = 0;
    for (int64_t a = 0; a < n_hit;) {
        int64_t b = a;
        while (b < n_hit && hit[b].t == hit[a].t) b++;
        const int64_t keep = b - a < bestn ? b - a : bestn;
        const int64_t first_row = n_rows;
        const ttok bread = breads.key[hit[a].t];
        const int64_t brid = stab_find(&rids, bread);
        for (int64_t h = a; h < a + keep; h++) {
            const int64_t r = stab_find(&rids, hit[h].q);
            for (int64_t k = 0; k < n_pair; k++) {
                if (pair_r[k] != r) continue;
                int64_t x = first_row;
                while (x < n_rows && !t_eq(rows[x].ctg, ctgs.key[pair_c[k]])) x++;
                if (x == n_rows) {
                    rows[x].bread = bread; rows[x].ctg = ctgs.key[pair_c[k]]; rows[x].count = 0; rows[x].score = 0; rows[x].in_ctg = 0;
                    for (int64_t z = 0; z < n_pair; z++) if (pair_r[z] == brid && pair_c[z] == pair_c[k]) rows[x].in_ctg = 1;
                    n_rows++;
                }
                rows[x].score += -hit[h].ovl; rows[x].count += 1;
            }
        }
        a = b;
    }
    qsort(rows, (size_t)n_rows, sizeof(orow), orow_cmp);
    size_t cap = 1 << 16, n_out = 0;
    char *out = (char *)malloc(cap);
    long long rank = 0;
    for (int64_t i = 0; i < n_rows; i++) {
        rank = (i > 0 && t_eq(rows[i].bread, rows[i - 1].bread)) ? rank + 1 : 0;
        char num[96];
        int nl_ = snprintf(num, sizeof num, " %lld %lld %lld %d\n", rows[i].count, rank, rows[i].score, rows[i].in_ctg);
        size_t need = (size_t)rows[i].bread.n + 1 + (size_t)rows[i].ctg.n + (size_t)nl_ + 1;
        if (n_out + need > cap) { while (n_out + need > cap) cap *= 2; out = (char *)realloc(out, cap); }
        memcpy(out + n_out, rows[i].bread.p, (size_t)rows[i].bread.n); n_out += (size_t)rows[i].bread.n;
        out[n_out++] = ' ';
        memcpy(out + n_out, rows[i].ctg.p, (size_t)rows[i].ctg.n); n_out += (size_t)rows[i].ctg.n;
        memcpy(out + n_out, num, (size_t)nl_); n_out += (size_t)nl_;
    }
    out[n_out] = 0;
    free(rows); free(hit); free(rph); free(oph); free(pair_r); free(pair_c);
    free(rids.key); free(rids.slot); free(ctgs.key); free(ctgs.slot); free(oids.key); free(oids.slot); free(breads.key); free(breads.slot);
    *out_txt = out; *out_len = n_out;
    return 0;
}
