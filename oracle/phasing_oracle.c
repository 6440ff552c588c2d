/*
 * phasing_oracle.c -- CPU restatement of FALCON_unzip's phasing path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP library: a plain-C, text-in/text-out restatement
 * of the four chained tasks of /root/reference/falcon_unzip/phasing.py and of
 * /root/reference/falcon_unzip/phasing_readmap.py.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product path (falcon_unzip_amd/) never does.
 *
 * Pinning: every function here is checked byte-for-byte against tests/golden/<case>/, which were
 * produced by RUNNING the reference (tests/golden/make_golden.py: in-memory lib2to3 translation
 * with the three Python-2 patches of SURVEY.md section 8c).  Parity is therefore pinned for
 * orc_make_het_call .. orc_phasing_readmap.  Python-2 semantics restated here on purpose:
 *   - allele keys of a site iterate in CPython-2.7 dict order A < C < T < G   (phasing.py:175,181)
 *   - a float is printed as '%.12g' (+ ".0" when it looks integral)            (phasing.py:418)
 *   - phased_reads / rid_to_phase rows are emitted in canonical sorted order   (phasing.py:466,
 *     phasing_readmap.py:50: py2 dict order is unspecified, consumers are order-insensitive)
 *
 * Each function cites the reference lines it follows.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_EINVAL -1      /* malformed input the reference would raise on (IndexError/KeyError/ValueError) */
#define ORC_EZERODIV -2    /* CIGAR with no ops: ZeroDivisionError at phasing.py:72 */
#define ORC_ENOMEM -3

/* ------------------------------------------------------------------ small containers */
typedef struct { char *p; size_t n, cap; } sbuf;

static int sb_reserve(sbuf *b, size_t extra) {
    if (b->n + extra + 1 <= b->cap) return 0;
    size_t nc = b->cap ? b->cap * 2 : 4096;
    while (nc < b->n + extra + 1) nc *= 2;
    char *q = (char *)realloc(b->p, nc);
    if (!q) return -1;
    b->p = q; b->cap = nc;
    return 0;
}
static void sb_put(sbuf *b, const char *s, size_t n) {
    if (sb_reserve(b, n)) return;
    memcpy(b->p + b->n, s, n); b->n += n; b->p[b->n] = 0;
}
static void sb_putc(sbuf *b, char c) { sb_put(b, &c, 1); }
static void sb_puti(sbuf *b, long long v) {
    char t[32]; int n = snprintf(t, sizeof t, "%lld", v); sb_put(b, t, (size_t)n);
}
static void sb_finish(sbuf *b, char **out, size_t *out_len) {
    if (!b->p) { b->p = (char *)malloc(1); b->p[0] = 0; }
    *out = b->p; *out_len = b->n;
}

/* whitespace tokenizer == Python's  line.strip().split()  */
typedef struct { const char *s; size_t n; } tok;
static int split_ws(const char *l, size_t n, tok *t, int maxt) {
    int k = 0; size_t i = 0;
    while (i < n) {
        while (i < n && isspace((unsigned char)l[i])) i++;
        if (i >= n) break;
        size_t j = i;
        while (j < n && !isspace((unsigned char)l[j])) j++;
        if (k < maxt) { t[k].s = l + i; t[k].n = j - i; }
        k++; i = j;
    }
    return k;
}
static int tok_int(tok t, long long *v) {
    if (t.n == 0 || t.n > 20) return -1;
    char b[24]; memcpy(b, t.s, t.n); b[t.n] = 0;
    char *e; *v = strtoll(b, &e, 10);
    return (*e == 0) ? 0 : -1;
}
/* line iterator over a text buffer (Python's `for l in f`) */
static int next_line(const char *buf, size_t len, size_t *off, const char **l, size_t *n) {
    if (*off >= len) return 0;
    const char *s = buf + *off;
    const char *e = (const char *)memchr(s, '\n', len - *off);
    size_t ln = e ? (size_t)(e - s) : len - *off;
    *l = s; *n = ln; *off += ln + (e ? 1 : 0);
    return 1;
}

/* string -> int map (q_name_to_id, phasing.py:32,48-53) */
typedef struct { const char **key; size_t *klen; int *val; size_t cap, n; } smap;
static uint64_t fnv(const char *s, size_t n) {
    uint64_t h = 1469598103934665603ULL;
    for (size_t i = 0; i < n; i++) { h ^= (unsigned char)s[i]; h *= 1099511628211ULL; }
    return h;
}
static int smap_init(smap *m, size_t cap) {
    m->cap = 64; while (m->cap < cap * 2) m->cap *= 2;
    m->key = (const char **)calloc(m->cap, sizeof *m->key);
    m->klen = (size_t *)calloc(m->cap, sizeof *m->klen);
    m->val = (int *)calloc(m->cap, sizeof *m->val);
    m->n = 0;
    return (m->key && m->klen && m->val) ? 0 : -1;
}
static void smap_free(smap *m) { free(m->key); free(m->klen); free(m->val); }
static int smap_grow(smap *m);
static int *smap_slot(smap *m, const char *s, size_t n, int *found) {
    size_t i = fnv(s, n) & (m->cap - 1);
    while (m->key[i]) {
        if (m->klen[i] == n && memcmp(m->key[i], s, n) == 0) { *found = 1; return &m->val[i]; }
        i = (i + 1) & (m->cap - 1);
    }
    *found = 0;
    if ((m->n + 1) * 2 > m->cap) { if (smap_grow(m)) return NULL; return smap_slot(m, s, n, found); }
    m->key[i] = s; m->klen[i] = n; m->n++;
    return &m->val[i];
}
static const int *smap_find(const smap *m, const char *s, size_t n) {
    size_t i = fnv(s, n) & (m->cap - 1);
    while (m->key[i]) {
        if (m->klen[i] == n && memcmp(m->key[i], s, n) == 0) return &m->val[i];
        i = (i + 1) & (m->cap - 1);
    }
    return NULL;
}
static int smap_grow(smap *m) {
    smap o = *m;
    if (smap_init(m, o.cap)) return -1;
    for (size_t i = 0; i < o.cap; i++) if (o.key[i]) {
        int f; int *v = smap_slot(m, o.key[i], o.klen[i], &f); *v = o.val[i];
    }
    smap_free(&o);
    return 0;
}

void orc_free(void *p) { free(p); }

/* ------------------------------------------------------------------ T1 make_het_call
 * reference: falcon_unzip/phasing.py:14-135.  Streaming restatement: the pileup is a
 * per-position list of (symbol, q_id) in arrival order (phasing.py:87-90); after every
 * ACCEPTED read all live positions < POS are evaluated in ascending order and deleted
 * (phasing.py:98-129); positions >= POS of the last accepted read are never evaluated. */
typedef struct { uint32_t n, cap; uint64_t *e; } pcol;      /* e = sym<<32 | q_id */

typedef struct {
    pcol *col; size_t ncol;
} pileup_t;

static int pile_reserve(pileup_t *p, size_t pos) {
    if (pos < p->ncol) return 0;
    size_t nc = p->ncol ? p->ncol : 1 << 16;
    while (nc <= pos) nc *= 2;
    pcol *q = (pcol *)realloc(p->col, nc * sizeof(pcol));
    if (!q) return -1;
    memset(q + p->ncol, 0, (nc - p->ncol) * sizeof(pcol));
    p->col = q; p->ncol = nc;
    return 0;
}
static int pile_add(pileup_t *p, size_t pos, unsigned char sym, int qid) {
    if (pile_reserve(p, pos)) return -1;
    pcol *c = &p->col[pos];
    if (c->n == c->cap) {
        uint32_t nc = c->cap ? c->cap * 2 : 16;
        uint64_t *q = (uint64_t *)realloc(c->e, nc * sizeof(uint64_t));
        if (!q) return -1;
        c->e = q; c->cap = nc;
    }
    c->e[c->n++] = ((uint64_t)sym << 32) | (uint32_t)qid;
    return 0;
}

/* the cigar regex r"(\d+)([MIDNSHP=X])" applied with finditer (phasing.py:12,65,77) */
static int cigar_next(const char *c, size_t n, size_t *i, long long *adv, char *op) {
    size_t k = *i;
    while (k < n) {
        if (!isdigit((unsigned char)c[k])) { k++; continue; }
        size_t d0 = k; long long v = 0;
        while (k < n && isdigit((unsigned char)c[k])) { v = v * 10 + (c[k] - '0'); k++; }
        if (k < n && strchr("MIDNSHP=X", c[k]) && c[k] != 0) {
            *adv = v; *op = c[k]; *i = k + 1; (void)d0; return 1;
        }
        /* digits not followed by an op char: regex restarts after them */
    }
    *i = n;
    return 0;
}

static int evaluate_pos(pileup_t *pl, size_t pos, const char *ref_seq, size_t ref_len, sbuf *vpos, sbuf *vmap) {
    pcol *c = &pl->col[pos];
    if (c->n == 0) return 0;                                  /* not live */
    /* len(pileup[pos]) < 2 : number of distinct symbols (phasing.py:103) */
    int seen[256]; memset(seen, 0, sizeof seen);
    int distinct = 0;
    for (uint32_t i = 0; i < c->n; i++) { unsigned s = (unsigned)(c->e[i] >> 32); if (!seen[s]) { seen[s] = 1; distinct++; } }
    int rc = 0;
    if (distinct >= 2) {
        /* phasing.py:106-111 */
        const char B[4] = {'A', 'C', 'G', 'T'};
        long long cnt[4] = {0, 0, 0, 0}, total = 0;
        for (uint32_t i = 0; i < c->n; i++) {
            unsigned s = (unsigned)(c->e[i] >> 32);
            for (int b = 0; b < 4; b++) if (s == (unsigned char)B[b]) cnt[b]++;
        }
        for (int b = 0; b < 4; b++) total += cnt[b];
        if (total >= 10) {                                     /* phasing.py:112 */
            /* sort (count, base) ascending then reverse (phasing.py:116-117) */
            int ord[4] = {0, 1, 2, 3};
            for (int a = 0; a < 4; a++) for (int b = a + 1; b < 4; b++) {
                int x = ord[a], y = ord[b];
                /* descending by (count, base) */
                if (cnt[y] > cnt[x] || (cnt[y] == cnt[x] && B[y] > B[x])) { ord[a] = y; ord[b] = x; }
            }
            double th = 0.25;
            double p0 = 1.0 * (double)cnt[ord[0]] / (double)total;
            double p1 = 1.0 * (double)cnt[ord[1]] / (double)total;
            if (p0 < 1.0 - th && p1 > th) {                   /* phasing.py:118-120 */
                if (pos >= ref_len) rc = ORC_EINVAL;          /* ref_seq[pos] IndexError */
                else {
                    char rb = ref_seq[pos];
                    sb_puti(vpos, (long long)pos + 1); sb_putc(vpos, ' '); sb_putc(vpos, rb); sb_putc(vpos, ' ');
                    sb_puti(vpos, total);
                    for (int k = 0; k < 4; k++) { sb_putc(vpos, ' '); sb_putc(vpos, B[ord[k]]); sb_putc(vpos, ' '); sb_puti(vpos, cnt[ord[k]]); }
                    sb_putc(vpos, '\n');
                    for (int k = 0; k < 2; k++) {               /* phasing.py:125-128 */
                        unsigned char bb = (unsigned char)B[ord[k]];
                        for (uint32_t i = 0; i < c->n; i++) if ((unsigned)(c->e[i] >> 32) == bb) {
                            sb_puti(vmap, (long long)pos + 1); sb_putc(vmap, ' '); sb_putc(vmap, rb); sb_putc(vmap, ' ');
                            sb_putc(vmap, (char)bb); sb_putc(vmap, ' '); sb_puti(vmap, (long long)(uint32_t)c->e[i]); sb_putc(vmap, '\n');
                        }
                    }
                }
            }
        }
    }
    free(c->e); c->e = NULL; c->n = c->cap = 0;               /* del pileup[pos] */
    return rc;
}

int orc_make_het_call(const char *sam, size_t sam_len, const char *ref_seq, size_t ref_len,
                      char **vpos_out, size_t *vpos_len, char **vmap_out, size_t *vmap_len,
                      char **qmap_out, size_t *qmap_len) {
    sbuf vpos = {0}, vmap = {0}, qmap = {0};
    pileup_t pl = {0};
    smap names; if (smap_init(&names, 1024)) return ORC_ENOMEM;
    const char **qname = NULL; size_t *qnlen = NULL; size_t qcap = 0; int q_max_id = 0;
    size_t off = 0, ln; const char *l;
    long long low_live = -1;                                    /* lower bound of live positions */
    int rc = ORC_OK;
    while (rc == ORC_OK && next_line(sam, sam_len, &off, &l, &ln)) {
        tok t[12];
        int nt = split_ws(l, ln, t, 12);
        if (nt == 0) { rc = ORC_EINVAL; break; }               /* l[0] IndexError */
        if (t[0].s[0] == '@') continue;                        /* phasing.py:44-45 */
        if (nt < 10) { rc = ORC_EINVAL; break; }
        /* phasing.py:47-54 : q_id by first appearance, before any filter */
        int found; int *slot = smap_slot(&names, t[0].s, t[0].n, &found);
        if (!slot) { rc = ORC_ENOMEM; break; }
        if (!found) {
            *slot = q_max_id;
            if ((size_t)q_max_id == qcap) {
                qcap = qcap ? qcap * 2 : 1024;
                qname = (const char **)realloc(qname, qcap * sizeof *qname);
                qnlen = (size_t *)realloc(qnlen, qcap * sizeof *qnlen);
            }
            qname[q_max_id] = t[0].s; qnlen[q_max_id] = t[0].n;
            q_max_id++;
        }
        int q_id = *slot;
        long long flag, pos1;
        if (tok_int(t[1], &flag) || tok_int(t[3], &pos1)) { rc = ORC_EINVAL; break; }
        long long POS = pos1 - 1;                              /* phasing.py:57 */
        const char *cig = t[5].s; size_t cn = t[5].n;
        const char *SEQ = t[9].s; size_t sn = t[9].n;
        /* phasing.py:63-75 */
        long long skip_base = 0, total_aln_pos = 0, adv; char op; size_t ci = 0;
        while (cigar_next(cig, cn, &ci, &adv, &op)) { total_aln_pos += adv; if (op == 'S') skip_base += adv; }
        if (total_aln_pos == 0) { rc = ORC_EZERODIV; break; }
        if (1.0 - 1.0 * (double)skip_base / (double)total_aln_pos < 0.1) continue;
        if (total_aln_pos < 2000) continue;
        if (POS < 0) { rc = ORC_EINVAL; break; }               /* negative keys: not restated */
        /* phasing.py:77-96 */
        long long rp = POS, qp = 0; ci = 0;
        while (rc == ORC_OK && cigar_next(cig, cn, &ci, &adv, &op)) {
            if (op == 'S') qp += adv;
            if (op == 'M' || op == '=' || op == 'X') {
                for (long long i = 0; i < adv; i++) {
                    if ((size_t)qp >= sn) { rc = ORC_EINVAL; break; }   /* SEQ[qp] IndexError */
                    if (pile_add(&pl, (size_t)rp, (unsigned char)SEQ[qp], q_id)) { rc = ORC_ENOMEM; break; }
                    rp++; qp++;
                }
            } else if (op == 'I') qp += adv;
            else if (op == 'D') rp += adv;
        }
        if (rc) break;
        if (low_live < 0 || POS < low_live) low_live = POS;
        /* phasing.py:98-129 : evaluate every live position < POS, ascending */
        long long hi = POS < (long long)pl.ncol ? POS : (long long)pl.ncol;
        for (long long p = low_live; p < hi && rc == ORC_OK; p++) rc = evaluate_pos(&pl, (size_t)p, ref_seq, ref_len, &vpos, &vmap);
        if (POS > low_live) low_live = POS;
    }
    /* phasing.py:132-134 (ascending q_id) */
    for (int q = 0; q < q_max_id && rc == ORC_OK; q++) {
        sb_puti(&qmap, q); sb_putc(&qmap, ' '); sb_put(&qmap, qname[q], qnlen[q]); sb_putc(&qmap, '\n');
    }
    for (size_t i = 0; i < pl.ncol; i++) free(pl.col[i].e);
    free(pl.col); free(qname); free(qnlen); smap_free(&names);
    if (rc) { free(vpos.p); free(vmap.p); free(qmap.p); return rc; }
    sb_finish(&vpos, vpos_out, vpos_len); sb_finish(&vmap, vmap_out, vmap_len); sb_finish(&qmap, qmap_out, qmap_len);
    return ORC_OK;
}

/* ------------------------------------------------------------------ shared: parse variant_map */
typedef struct {
    long long pos; char ref_b;
    int nallele; char allele[4];            /* in first-appearance order */
    int *q[4]; int nq[4], capq[4];          /* q_id lists in file order */
} vsite;
typedef struct { vsite *s; size_t n, cap; } vsites;

static void vsites_free(vsites *v) {
    for (size_t i = 0; i < v->n; i++) for (int a = 0; a < 4; a++) free(v->s[i].q[a]);
    free(v->s);
}
/* vmap[(pos, ref_b)][v_b].append(q_id); v_positions in first-appearance order (phasing.py:147-158) */
static int parse_vmap(const char *vmap, size_t len, vsites *out) {
    size_t off = 0, ln; const char *l;
    /* key lookup: linear probe hash on (pos, ref_b) */
    size_t hcap = 1024; size_t *ht = (size_t *)malloc(hcap * sizeof(size_t));
    if (!ht) return ORC_ENOMEM;
    memset(ht, 0xff, hcap * sizeof(size_t));
    while (next_line(vmap, len, &off, &l, &ln)) {
        tok t[4]; if (split_ws(l, ln, t, 4) < 4) { free(ht); return ORC_EINVAL; }
        long long pos, qid;
        if (tok_int(t[0], &pos) || tok_int(t[3], &qid) || t[1].n != 1 || t[2].n != 1) { free(ht); return ORC_EINVAL; }
        char rb = t[1].s[0], vb = t[2].s[0];
        uint64_t h = ((uint64_t)pos * 1000003ULL) ^ (unsigned char)rb;
        size_t i = (size_t)(h * 11400714819323198485ULL >> 20) & (hcap - 1);
        vsite *s = NULL;
        while (ht[i] != (size_t)-1) {
            vsite *c = &out->s[ht[i]];
            if (c->pos == pos && c->ref_b == rb) { s = c; break; }
            i = (i + 1) & (hcap - 1);
        }
        if (!s) {
            if (out->n == out->cap) {
                out->cap = out->cap ? out->cap * 2 : 1024;
                out->s = (vsite *)realloc(out->s, out->cap * sizeof(vsite));
                if (!out->s) { free(ht); return ORC_ENOMEM; }
            }
            s = &out->s[out->n]; memset(s, 0, sizeof *s); s->pos = pos; s->ref_b = rb;
            ht[i] = out->n++;
            if (out->n * 2 > hcap) {              /* rehash */
                hcap *= 2; ht = (size_t *)realloc(ht, hcap * sizeof(size_t)); memset(ht, 0xff, hcap * sizeof(size_t));
                for (size_t k = 0; k < out->n; k++) {
                    uint64_t hh = ((uint64_t)out->s[k].pos * 1000003ULL) ^ (unsigned char)out->s[k].ref_b;
                    size_t j = (size_t)(hh * 11400714819323198485ULL >> 20) & (hcap - 1);
                    while (ht[j] != (size_t)-1) j = (j + 1) & (hcap - 1);
                    ht[j] = k;
                }
                s = &out->s[out->n - 1];
            }
        }
        int a = -1;
        for (int k = 0; k < s->nallele; k++) if (s->allele[k] == vb) a = k;
        if (a < 0) { if (s->nallele == 4) { free(ht); return ORC_EINVAL; } a = s->nallele++; s->allele[a] = vb; }
        if (s->nq[a] == s->capq[a]) { s->capq[a] = s->capq[a] ? s->capq[a] * 2 : 32; s->q[a] = (int *)realloc(s->q[a], (size_t)s->capq[a] * sizeof(int)); }
        s->q[a][s->nq[a]++] = (int)qid;
    }
    free(ht);
    return ORC_OK;
}

static int cmp_int(const void *a, const void *b) { int x = *(const int *)a, y = *(const int *)b; return (x > y) - (x < y); }
/* CPython 2.7 iteration order of a small dict keyed by one-character strings: slot = hash & 7,
 * 'A'->0 'C'->2 'T'->5 'G'->6   */
static int py2_allele_rank(char b) { const char *o = "ACTG"; const char *p = strchr(o, b); return p ? (int)(p - o) : 4 + (unsigned char)b; }

/* ------------------------------------------------------------------ T2 generate_association_table
 * reference: falcon_unzip/phasing.py:137-206 */
int orc_generate_association_table(const char *vmap, size_t vmap_len, char **atable_out, size_t *atable_len) {
    vsites vs = {0};
    int rc = parse_vmap(vmap, vmap_len, &vs);
    if (rc) { vsites_free(&vs); return rc; }
    sbuf out = {0};
    /* per site: allele order as py2 .items() yields it; sets of q_ids (sorted unique) */
    size_t n = vs.n;
    int (*ord)[4] = (int (*)[4])malloc(n * sizeof *ord);
    int **uq = (int **)calloc(n * 4, sizeof(int *)); int *nu = (int *)calloc(n * 4, sizeof(int));
    for (size_t i = 0; i < n; i++) {
        vsite *s = &vs.s[i];
        for (int a = 0; a < s->nallele; a++) ord[i][a] = a;
        for (int a = 0; a < s->nallele; a++) for (int b = a + 1; b < s->nallele; b++)
            if (py2_allele_rank(s->allele[ord[i][b]]) < py2_allele_rank(s->allele[ord[i][a]])) { int t = ord[i][a]; ord[i][a] = ord[i][b]; ord[i][b] = t; }
        for (int a = 0; a < s->nallele; a++) {
            int *u = (int *)malloc((size_t)s->nq[a] * sizeof(int) + 4);
            memcpy(u, s->q[a], (size_t)s->nq[a] * sizeof(int));
            qsort(u, (size_t)s->nq[a], sizeof(int), cmp_int);
            int m = 0; for (int k = 0; k < s->nq[a]; k++) if (m == 0 || u[m - 1] != u[k]) u[m++] = u[k];
            uq[i * 4 + a] = u; nu[i * 4 + a] = m;
        }
    }
    for (size_t i1 = 0; i1 < n && rc == ORC_OK; i1++) {
        int link_count = 0;
        for (size_t i2 = i1 + 1; i2 < n; i2++) {
            vsite *s1 = &vs.s[i1], *s2 = &vs.s[i2];
            if (s2->pos - s1->pos > (1 << 16)) continue;                      /* phasing.py:169 */
            long long ct[4][4]; long long total_s = 0;
            for (int a = 0; a < s1->nallele; a++) for (int b = 0; b < s2->nallele; b++) {
                const int *x = uq[i1 * 4 + ord[i1][a]], *y = uq[i2 * 4 + ord[i2][b]];
                int nx = nu[i1 * 4 + ord[i1][a]], ny = nu[i2 * 4 + ord[i2][b]], p = 0, q = 0; long long c = 0;
                while (p < nx && q < ny) { if (x[p] < y[q]) p++; else if (x[p] > y[q]) q++; else { c++; p++; q++; } }
                ct[a][b] = c; total_s += c;                                   /* phasing.py:186-191 */
            }
            if (total_s < 6) continue;                                        /* phasing.py:192 */
            if (s1->nallele < 2 || s2->nallele < 2) { rc = ORC_EINVAL; break; } /* p1table[1] IndexError */
            sb_puti(&out, s1->pos); sb_putc(&out, ' '); sb_putc(&out, s1->allele[ord[i1][0]]); sb_putc(&out, ' '); sb_putc(&out, s1->allele[ord[i1][1]]); sb_putc(&out, ' ');
            sb_puti(&out, s2->pos); sb_putc(&out, ' '); sb_putc(&out, s2->allele[ord[i2][0]]); sb_putc(&out, ' '); sb_putc(&out, s2->allele[ord[i2][1]]); sb_putc(&out, ' ');
            sb_puti(&out, ct[0][0]); sb_putc(&out, ' '); sb_puti(&out, ct[0][1]); sb_putc(&out, ' '); sb_puti(&out, ct[1][0]); sb_putc(&out, ' '); sb_puti(&out, ct[1][1]); sb_putc(&out, '\n');
            link_count++;
            if (link_count > 500) break;                                      /* phasing.py:204-206 */
        }
    }
    for (size_t i = 0; i < n * 4; i++) free(uq[i]);
    free(uq); free(nu); free(ord); vsites_free(&vs);
    if (rc) { free(out.p); return rc; }
    sb_finish(&out, atable_out, atable_len);
    return ORC_OK;
}

/* ------------------------------------------------------------------ T3 get_phased_blocks
 * reference: falcon_unzip/phasing.py:208-421 */
typedef struct { long long pos1, pos2; char b11, b12, b21, b22; long long cis, trans; } link_t;
typedef struct { int *v; int n, cap; } ivec;
static void iv_push(ivec *v, int x) { if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 8; v->v = (int *)realloc(v->v, (size_t)v->cap * sizeof(int)); } v->v[v->n++] = x; }

typedef struct { char a, b; } state_t;
/* get_score (phasing.py:208-214) + the c_score dict built at phasing.py:255-256 */
static int get_score(const link_t *lk, state_t s1, state_t s2, long long *out) {
    /* caller passes states already ordered (pos1's state first) */
    char k0[2] = {s1.a, s2.a}, k1[2] = {s1.b, s2.b};
    const char keys[4][4] = {
        {lk->b11, lk->b21, lk->b12, lk->b22}, {lk->b12, lk->b22, lk->b11, lk->b21},
        {lk->b12, lk->b21, lk->b11, lk->b22}, {lk->b11, lk->b22, lk->b12, lk->b21}};
    const long long val[4] = {lk->cis, lk->cis, lk->trans, lk->trans};
    int hit = -1;
    for (int k = 0; k < 4; k++)     /* later dict entries overwrite earlier equal keys */
        if (keys[k][0] == k0[0] && keys[k][1] == k0[1] && keys[k][2] == k1[0] && keys[k][3] == k1[1]) hit = k;
    if (hit < 0) return -1;         /* KeyError */
    *out = val[hit];
    return 0;
}

static int cmp_ll(const void *a, const void *b) { long long x = *(const long long *)a, y = *(const long long *)b; return (x > y) - (x < y); }

static void py2_float(sbuf *b, double x) {
    char t[64]; snprintf(t, sizeof t, "%.12g", x);
    if (!strpbrk(t, ".enN")) strcat(t, ".0");
    sb_put(b, t, strlen(t));
}

int orc_get_phased_blocks(const char *vmap, size_t vmap_len, const char *atable, size_t atable_len,
                          char **pv_out, size_t *pv_len) {
    int rc = ORC_OK;
    /* ref_base[pos] = ref_b, last one wins (phasing.py:230-238) */
    size_t off = 0, ln; const char *l;
    size_t nrb = 0, crb = 0; long long *rb_pos = NULL; char *rb_b = NULL;
    while (next_line(vmap, vmap_len, &off, &l, &ln)) {
        tok t[4]; if (split_ws(l, ln, t, 4) < 4) { free(rb_pos); free(rb_b); return ORC_EINVAL; }
        long long pos; if (tok_int(t[0], &pos)) { free(rb_pos); free(rb_b); return ORC_EINVAL; }
        if (nrb && rb_pos[nrb - 1] == pos) { rb_b[nrb - 1] = t[1].s[0]; continue; }
        if (nrb == crb) { crb = crb ? crb * 2 : 1024; rb_pos = (long long *)realloc(rb_pos, crb * sizeof(long long)); rb_b = (char *)realloc(rb_b, crb); }
        rb_pos[nrb] = pos; rb_b[nrb] = t[1].s[0]; nrb++;
    }
    /* first pass over atable: collect kept rows (phasing.py:240-256) */
    size_t nl = 0, cl = 0; link_t *lk = NULL;
    off = 0;
    while (next_line(atable, atable_len, &off, &l, &ln)) {
        tok t[10]; if (split_ws(l, ln, t, 10) != 10) { rc = ORC_EINVAL; break; }
        long long p1, p2, s11, s12, s21, s22;
        if (tok_int(t[0], &p1) || tok_int(t[3], &p2) || tok_int(t[6], &s11) || tok_int(t[7], &s12) || tok_int(t[8], &s21) || tok_int(t[9], &s22)) { rc = ORC_EINVAL; break; }
        if (llabs(s11 + s22 - s12 - s21) < 6) continue;                        /* phasing.py:245 */
        if (nl == cl) { cl = cl ? cl * 2 : 4096; lk = (link_t *)realloc(lk, cl * sizeof(link_t)); }
        link_t *k = &lk[nl++];
        k->pos1 = p1; k->pos2 = p2; k->b11 = t[1].s[0]; k->b12 = t[2].s[0]; k->b21 = t[4].s[0]; k->b22 = t[5].s[0];
        k->cis = s11 + s22; k->trans = s12 + s21;
    }
    if (rc) { free(lk); free(rb_pos); free(rb_b); return rc; }
    /* positions = sorted set (phasing.py:249-250,311-312) */
    long long *positions = (long long *)malloc((2 * nl + 1) * sizeof(long long)); size_t np = 0;
    for (size_t i = 0; i < nl; i++) { positions[np++] = lk[i].pos1; positions[np++] = lk[i].pos2; }
    qsort(positions, np, sizeof(long long), cmp_ll);
    { size_t m = 0; for (size_t i = 0; i < np; i++) if (m == 0 || positions[m - 1] != positions[i]) positions[m++] = positions[i]; np = m; }
#define PIDX(P) ((int)((long long *)bsearch(&(P), positions, np, sizeof(long long), cmp_ll) - positions))
    ivec *left = (ivec *)calloc(np + 1, sizeof(ivec)), *right = (ivec *)calloc(np + 1, sizeof(ivec));  /* hold link indices */
    state_t *states = (state_t *)calloc(np + 1, sizeof(state_t)); char *has = (char *)calloc(np + 1, 1);
    /* streaming greedy initialisation, interleaved with reading (phasing.py:251-309) */
    for (size_t i = 0; i < nl && rc == ORC_OK; i++) {
        int i1 = PIDX(lk[i].pos1), i2 = PIDX(lk[i].pos2);
        iv_push(&right[i1], (int)i); iv_push(&left[i2], (int)i);
        for (int side = 0; side < 2 && rc == ORC_OK; side++) {
            int me = side == 0 ? i1 : i2;
            if (has[me]) continue;
            state_t st1 = side == 0 ? (state_t){lk[i].b11, lk[i].b12} : (state_t){lk[i].b21, lk[i].b22};
            state_t st2 = {st1.b, st1.a};
            long long score1 = 0, score2 = 0, v;
            for (int k = 0; k < left[me].n; k++) {               /* pp -> me, pp is pos1 of that link */
                const link_t *e = &lk[left[me].v[k]]; int pp = PIDX(e->pos1);
                if (!has[pp]) continue;
                if (get_score(e, states[pp], st1, &v)) { rc = ORC_EINVAL; break; } score1 += v;
                if (get_score(e, states[pp], st2, &v)) { rc = ORC_EINVAL; break; } score2 += v;
            }
            for (int k = 0; k < right[me].n && rc == ORC_OK; k++) { /* me -> pp */
                const link_t *e = &lk[right[me].v[k]]; int pp = PIDX(e->pos2);
                if (!has[pp]) continue;
                if (get_score(e, st1, states[pp], &v)) { rc = ORC_EINVAL; break; } score1 += v;
                if (get_score(e, st2, states[pp], &v)) { rc = ORC_EINVAL; break; } score2 += v;
            }
            states[me] = score1 >= score2 ? st1 : st2; has[me] = 1;
        }
    }
    /* iterative refinement, left links only, in place (phasing.py:315-344) */
    int iter_count = 0;
    while (rc == ORC_OK) {
        iter_count++;
        if (iter_count > 10) break;
        int update_count = 0;
        for (size_t p = 0; p < np && rc == ORC_OK; p++) {
            state_t st1 = states[p], st2 = {st1.b, st1.a};
            long long score1 = 0, score2 = 0, v;
            for (int k = 0; k < left[p].n; k++) {
                const link_t *e = &lk[left[p].v[k]]; int pp = PIDX(e->pos1);
                if (get_score(e, states[pp], st1, &v)) { rc = ORC_EINVAL; break; } score1 += v;
                if (get_score(e, states[pp], st2, &v)) { rc = ORC_EINVAL; break; } score2 += v;
            }
            if (score1 >= score2) states[p] = st1; else { states[p] = st2; update_count++; }
        }
        if (update_count == 0) break;
    }
    /* extents and scores (phasing.py:347-383) */
    long long *lext = (long long *)calloc(np + 1, sizeof(long long)), *rext = (long long *)calloc(np + 1, sizeof(long long));
    long long *lsc = (long long *)calloc(np + 1, sizeof(long long)), *rsc = (long long *)calloc(np + 1, sizeof(long long));
    for (size_t p = 0; p < np && rc == ORC_OK; p++) {
        long long P = positions[p];
        lext[p] = P; lsc[p] = 0;
        state_t st0 = states[p], st0_ = {st0.b, st0.a};
        long long s, s_;
        long long lft = P;
        for (int k = 0; k < left[p].n; k++) {
            const link_t *e = &lk[left[p].v[k]]; int pp = PIDX(e->pos1);
            if (get_score(e, states[pp], st0, &s) || get_score(e, states[pp], st0_, &s_)) { rc = ORC_EINVAL; break; }
            lsc[p] += s - s_;
            if (s - s_ > 0 && e->pos1 < lft) lft = e->pos1;
        }
        lext[p] = lft;
        rext[p] = P; rsc[p] = 0;
        long long rgt = P;
        for (int k = 0; k < right[p].n && rc == ORC_OK; k++) {
            const link_t *e = &lk[right[p].v[k]]; int pp = PIDX(e->pos2);
            if (get_score(e, st0, states[pp], &s) || get_score(e, st0_, states[pp], &s_)) { rc = ORC_EINVAL; break; }
            rsc[p] += s - s_;
            if (s - s_ > 0 && e->pos2 > rgt) rgt = e->pos2;
        }
        rext[p] = rgt;
    }
    /* block segmentation (phasing.py:388-408) */
    sbuf out = {0};
    if (rc == ORC_OK) {
        int *blk = (int *)calloc(np + 1, sizeof(int));       /* block id per kept site, 0 = none */
        int phase_block_id = 1; long long max_right_ext = 0;
        size_t *pb = (size_t *)malloc((np + 1) * sizeof(size_t)); size_t npb = 0;
        for (size_t p = 0; p < np; p++) {
            if (rsc[p] < 10 || lsc[p] < 10) continue;
            if (max_right_ext < lext[p]) {
                if (npb > 3) { for (size_t k = 0; k < npb; k++) blk[pb[k]] = phase_block_id; phase_block_id++; }
                npb = 0;
            }
            pb[npb++] = p;
            if (rext[p] > max_right_ext) max_right_ext = rext[p];
        }
        if (npb > 3) { for (size_t k = 0; k < npb; k++) blk[pb[k]] = phase_block_id; } else phase_block_id--;
        /* output (phasing.py:411-421) */
        for (int pid = 1; pid <= phase_block_id; pid++) {
            long long mn = 0, mx = 0, cnt = 0;
            for (size_t p = 0; p < np; p++) if (blk[p] == pid) { if (!cnt || positions[p] < mn) mn = positions[p]; if (!cnt || positions[p] > mx) mx = positions[p]; cnt++; }
            if (cnt == 0) continue;
            sb_put(&out, "P ", 2); sb_puti(&out, pid); sb_putc(&out, ' '); sb_puti(&out, mn); sb_putc(&out, ' '); sb_puti(&out, mx); sb_putc(&out, ' ');
            sb_puti(&out, mx - mn); sb_putc(&out, ' '); sb_puti(&out, cnt); sb_putc(&out, ' '); py2_float(&out, 1.0 * (double)(mx - mn) / (double)cnt); sb_putc(&out, '\n');
            for (size_t p = 0; p < np; p++) if (blk[p] == pid) {
                long long P = positions[p];
                long long *f = (long long *)bsearch(&P, rb_pos, nrb, sizeof(long long), cmp_ll);
                if (!f) { rc = ORC_EINVAL; break; }           /* ref_base[p] KeyError */
                char rb = rb_b[f - rb_pos];
                sb_put(&out, "V ", 2); sb_puti(&out, pid); sb_putc(&out, ' '); sb_puti(&out, P); sb_putc(&out, ' ');
                sb_puti(&out, P); sb_putc(&out, '_'); sb_putc(&out, rb); sb_putc(&out, '_'); sb_putc(&out, states[p].a); sb_putc(&out, ' ');
                sb_puti(&out, P); sb_putc(&out, '_'); sb_putc(&out, rb); sb_putc(&out, '_'); sb_putc(&out, states[p].b); sb_putc(&out, ' ');
                sb_puti(&out, lext[p]); sb_putc(&out, ' '); sb_puti(&out, rext[p]); sb_putc(&out, ' '); sb_puti(&out, lsc[p]); sb_putc(&out, ' '); sb_puti(&out, rsc[p]); sb_putc(&out, '\n');
            }
        }
        free(blk); free(pb);
    }
    for (size_t p = 0; p <= np; p++) { free(left[p].v); free(right[p].v); }
    free(left); free(right); free(states); free(has); free(lext); free(rext); free(lsc); free(rsc);
    free(positions); free(lk); free(rb_pos); free(rb_b);
    if (rc) { free(out.p); return rc; }
    sb_finish(&out, pv_out, pv_len);
    return ORC_OK;
}

/* ------------------------------------------------------------------ T4 get_phased_reads
 * reference: falcon_unzip/phasing.py:423-480.  Rows are emitted by ascending q_id (canonical order). */
typedef struct { long long pos; char rb, vb; int qid; } vrow;
typedef struct { long long pos; char rb, vb; int blk, phase; } vphase;
static int cmp_vrow(const void *a, const void *b) {
    const vrow *x = (const vrow *)a, *y = (const vrow *)b;
    if (x->qid != y->qid) return (x->qid > y->qid) - (x->qid < y->qid);
    if (x->pos != y->pos) return (x->pos > y->pos) - (x->pos < y->pos);
    if (x->rb != y->rb) return (x->rb > y->rb) - (x->rb < y->rb);
    return (x->vb > y->vb) - (x->vb < y->vb);
}
static int cmp_vphase(const void *a, const void *b) {
    const vphase *x = (const vphase *)a, *y = (const vphase *)b;
    if (x->pos != y->pos) return (x->pos > y->pos) - (x->pos < y->pos);
    if (x->rb != y->rb) return (x->rb > y->rb) - (x->rb < y->rb);
    return (x->vb > y->vb) - (x->vb < y->vb);
}
/* "pos_R_B" -> (pos, R, B) */
static int parse_variant(tok t, long long *pos, char *rb, char *vb) {
    if (t.n < 5 || t.s[t.n - 2] != '_' || t.s[t.n - 4] != '_') return -1;
    tok pt = {t.s, t.n - 4};
    if (tok_int(pt, pos)) return -1;
    *rb = t.s[t.n - 3]; *vb = t.s[t.n - 1];
    return 0;
}

int orc_get_phased_reads(const char *vmap, size_t vmap_len, const char *qmap, size_t qmap_len,
                         const char *pv, size_t pv_len, const char *ctg_id,
                         char **out_p, size_t *out_len) {
    size_t off = 0, ln; const char *l; int rc = ORC_OK;
    /* rid_map (phasing.py:434-438) */
    size_t nq = 0, cq = 0; tok *qn = NULL;
    while (next_line(qmap, qmap_len, &off, &l, &ln)) {
        tok t[2]; if (split_ws(l, ln, t, 2) < 2) { free(qn); return ORC_EINVAL; }
        long long q; if (tok_int(t[0], &q) || q < 0) { free(qn); return ORC_EINVAL; }
        while ((size_t)q >= cq) { size_t nc = cq ? cq * 2 : 1024; qn = (tok *)realloc(qn, nc * sizeof(tok)); memset(qn + cq, 0, (nc - cq) * sizeof(tok)); cq = nc; }
        qn[q] = t[1]; if ((size_t)q + 1 > nq) nq = (size_t)q + 1;
    }
    /* read_to_variants: set of variants per read (phasing.py:441-451) */
    size_t nr = 0, cr = 0; vrow *rows = NULL; off = 0;
    while (next_line(vmap, vmap_len, &off, &l, &ln)) {
        tok t[4]; if (split_ws(l, ln, t, 4) < 4) { rc = ORC_EINVAL; break; }
        long long pos, q; if (tok_int(t[0], &pos) || tok_int(t[3], &q)) { rc = ORC_EINVAL; break; }
        if (nr == cr) { cr = cr ? cr * 2 : 4096; rows = (vrow *)realloc(rows, cr * sizeof(vrow)); }
        rows[nr].pos = pos; rows[nr].rb = t[1].s[0]; rows[nr].vb = t[2].s[0]; rows[nr].qid = (int)q; nr++;
    }
    /* variant_to_phase (phasing.py:454-463) */
    size_t nv = 0, cv = 0; vphase *vp = NULL; off = 0;
    while (rc == ORC_OK && next_line(pv, pv_len, &off, &l, &ln)) {
        tok t[9]; int nt = split_ws(l, ln, t, 9);
        if (nt == 0) { rc = ORC_EINVAL; break; }
        if (!(t[0].n == 1 && t[0].s[0] == 'V')) continue;
        if (nt < 5) { rc = ORC_EINVAL; break; }
        long long pb; if (tok_int(t[1], &pb)) { rc = ORC_EINVAL; break; }
        for (int ph = 0; ph < 2; ph++) {
            if (nv == cv) { cv = cv ? cv * 2 : 1024; vp = (vphase *)realloc(vp, cv * sizeof(vphase)); }
            if (parse_variant(t[3 + ph], &vp[nv].pos, &vp[nv].rb, &vp[nv].vb)) { rc = ORC_EINVAL; break; }
            vp[nv].blk = (int)pb; vp[nv].phase = ph; nv++;
        }
    }
    sbuf out = {0};
    if (rc == ORC_OK) {
        qsort(rows, nr, sizeof(vrow), cmp_vrow);
        /* stable w.r.t. "later assignment wins": mergesort-free trick: tag order, then sort */
        qsort(vp, nv, sizeof(vphase), cmp_vphase);
        size_t i = 0;
        size_t ctg_n = strlen(ctg_id);
        while (i < nr && rc == ORC_OK) {
            size_t j = i; int r = rows[i].qid;
            /* per read: vl[(block, phase)] over DISTINCT variants (phasing.py:466-473) */
            size_t ne = 0; int eb[4096]; long long e0[4096], e1[4096];
            int *ebp = eb; long long *e0p = e0, *e1p = e1; size_t ecap = 4096; int heap = 0;
            while (j < nr && rows[j].qid == r) {
                if (j > i && cmp_vrow(&rows[j], &rows[j - 1]) == 0) { j++; continue; }   /* set semantics */
                vphase key = {rows[j].pos, rows[j].rb, rows[j].vb, 0, 0};
                vphase *f = (vphase *)bsearch(&key, vp, nv, sizeof(vphase), cmp_vphase);
                if (f) {
                    while (f + 1 < vp + nv && cmp_vphase(f + 1, &key) == 0) f++;          /* last assignment wins */
                    size_t k = 0; while (k < ne && ebp[k] != f->blk) k++;
                    if (k == ne) {
                        if (ne == ecap) {
                            size_t nc = ecap * 2;
                            int *nb = (int *)malloc(nc * sizeof(int)); long long *n0 = (long long *)malloc(nc * sizeof(long long)), *n1 = (long long *)malloc(nc * sizeof(long long));
                            memcpy(nb, ebp, ne * sizeof(int)); memcpy(n0, e0p, ne * sizeof(long long)); memcpy(n1, e1p, ne * sizeof(long long));
                            if (heap) { free(ebp); free(e0p); free(e1p); }
                            ebp = nb; e0p = n0; e1p = n1; ecap = nc; heap = 1;
                        }
                        ebp[ne] = f->blk; e0p[ne] = 0; e1p[ne] = 0; ne++;
                    }
                    if (f->phase == 0) e0p[k]++; else e1p[k]++;
                }
                j++;
            }
            /* pl.sort(); emit (phasing.py:474-480) */
            for (size_t a = 0; a < ne; a++) for (size_t b = a + 1; b < ne; b++) if (ebp[b] < ebp[a]) {
                int tb = ebp[a]; ebp[a] = ebp[b]; ebp[b] = tb; long long t0 = e0p[a]; e0p[a] = e0p[b]; e0p[b] = t0; long long t1 = e1p[a]; e1p[a] = e1p[b]; e1p[b] = t1; }
            for (size_t a = 0; a < ne; a++) {
                int ph = -1;
                if (e0p[a] - e1p[a] > 1) ph = 0; else if (e1p[a] - e0p[a] > 1) ph = 1;
                if (ph < 0) continue;
                if (r < 0 || (size_t)r >= nq || qn[r].s == NULL) { rc = ORC_EINVAL; break; }   /* rid_map[r] KeyError */
                sb_puti(&out, r); sb_putc(&out, ' '); sb_put(&out, ctg_id, ctg_n); sb_putc(&out, ' '); sb_puti(&out, ebp[a]); sb_putc(&out, ' ');
                sb_puti(&out, ph); sb_putc(&out, ' '); sb_puti(&out, e0p[a]); sb_putc(&out, ' '); sb_puti(&out, e1p[a]); sb_putc(&out, ' ');
                sb_put(&out, qn[r].s, qn[r].n); sb_putc(&out, '\n');
            }
            if (heap) { free(ebp); free(e0p); free(e1p); }
            i = j;
        }
    }
    free(rows); free(vp); free(qn);
    if (rc) { free(out.p); return rc; }
    sb_finish(&out, out_p, out_len);
    return ORC_OK;
}

/* ------------------------------------------------------------------ get_phasing_readmap
 * reference: falcon_unzip/phasing_readmap.py:8-51.  Rows sorted by the '%09d' id string. */
typedef struct { tok *v; size_t n, cap; } tokvec;
static void tv_push(tokvec *v, tok t) { if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->v = (tok *)realloc(v->v, v->cap * sizeof(tok)); } v->v[v->n++] = t; }
/* text.split('\n') */
static void split_nl(const char *s, size_t n, tokvec *out) {
    size_t i = 0;
    for (;;) {
        const char *e = (const char *)memchr(s + i, '\n', n - i);
        size_t ln = e ? (size_t)(e - (s + i)) : n - i;
        tok t = {s + i, ln}; tv_push(out, t);
        if (!e) break;
        i += ln + 1;
    }
}
typedef struct { char arid[24]; size_t seq; int blk, ph; } arow;
static int cmp_arow(const void *a, const void *b) {
    const arow *x = (const arow *)a, *y = (const arow *)b;
    int c = strcmp(x->arid, y->arid);
    return c ? c : (x->seq > y->seq) - (x->seq < y->seq);
}

int orc_phasing_readmap(const char *phased_reads, size_t pr_len, const char *rawread_ids, size_t rr_len,
                        const char *pread_ids, size_t pi_len, const char *pread_to_contigs, size_t pc_len,
                        const char *the_ctg_id, char **out_p, size_t *out_len) {
    int rc = ORC_OK;
    tokvec rid_to_oid = {0}, pid_to_fid = {0};
    split_nl(rawread_ids, rr_len, &rid_to_oid);                 /* phasing_readmap.py:17 */
    split_nl(pread_ids, pi_len, &pid_to_fid);                   /* phasing_readmap.py:18 */
    /* rid_to_phase[row[6]] = (block, phase), last line wins (phasing_readmap.py:29-33) */
    smap names; smap_init(&names, 1024);
    size_t np = 0, cp = 0; int (*ph)[2] = NULL;
    size_t off = 0, ln; const char *l;
    while (next_line(phased_reads, pr_len, &off, &l, &ln)) {
        tok t[7]; if (split_ws(l, ln, t, 7) < 7) { rc = ORC_EINVAL; break; }
        long long b, p; if (tok_int(t[2], &b) || tok_int(t[3], &p)) { rc = ORC_EINVAL; break; }
        int f; int *slot = smap_slot(&names, t[6].s, t[6].n, &f);
        if (!f) { if (np == cp) { cp = cp ? cp * 2 : 1024; ph = (int (*)[2])realloc(ph, cp * sizeof *ph); } *slot = (int)np++; }
        ph[*slot][0] = (int)b; ph[*slot][1] = (int)p;
    }
    size_t na = 0, ca = 0; arow *ar = NULL; off = 0;
    size_t ctg_n = strlen(the_ctg_id);
    while (rc == ORC_OK && next_line(pread_to_contigs, pc_len, &off, &l, &ln)) {
        tok t[4]; int nt = split_ws(l, ln, t, 4);
        if (nt < 2) { rc = ORC_EINVAL; break; }
        if (!(t[1].n >= ctg_n && memcmp(t[1].s, the_ctg_id, ctg_n) == 0)) continue;   /* startswith, line 41 */
        if (nt < 4) { rc = ORC_EINVAL; break; }
        long long rank, pid; if (tok_int(t[3], &rank)) { rc = ORC_EINVAL; break; }
        if (rank != 0) continue;                                                      /* line 43 */
        if (tok_int(t[0], &pid) || pid < 0 || (size_t)pid >= pid_to_fid.n) { rc = ORC_EINVAL; break; }
        /* pid_to_oid (lines 20-23): fid.split('/')[1], integer-divided by 10 */
        tok fid = pid_to_fid.v[pid];
        const char *s1 = (const char *)memchr(fid.s, '/', fid.n);
        if (!s1) { rc = ORC_EINVAL; break; }
        s1++;
        const char *s2 = (const char *)memchr(s1, '/', fid.n - (size_t)(s1 - fid.s));
        tok mid = {s1, s2 ? (size_t)(s2 - s1) : fid.n - (size_t)(s1 - fid.s)};
        long long raw; if (tok_int(mid, &raw) || raw < 0) { rc = ORC_EINVAL; break; }
        raw /= 10;
        if ((size_t)raw >= rid_to_oid.n) { rc = ORC_EINVAL; break; }
        tok oid = rid_to_oid.v[raw];
        int blk = -1, p = 0;                                                          /* line 46 default */
        const int *slot = smap_find(&names, oid.s, oid.n);
        if (slot) { blk = ph[*slot][0]; p = ph[*slot][1]; }
        if (na == ca) { ca = ca ? ca * 2 : 1024; ar = (arow *)realloc(ar, ca * sizeof(arow)); }
        snprintf(ar[na].arid, sizeof ar[na].arid, "%09lld", pid);
        ar[na].seq = na; ar[na].blk = blk; ar[na].ph = p; na++;
    }
    sbuf out = {0};
    if (rc == ORC_OK) {
        qsort(ar, na, sizeof(arow), cmp_arow);
        for (size_t k = 0; k < na; k++) {
            if (k + 1 < na && strcmp(ar[k].arid, ar[k + 1].arid) == 0) continue;     /* dict: last assignment wins */
            sb_put(&out, ar[k].arid, strlen(ar[k].arid)); sb_putc(&out, ' '); sb_put(&out, the_ctg_id, ctg_n); sb_putc(&out, ' ');
            sb_puti(&out, ar[k].blk); sb_putc(&out, ' '); sb_puti(&out, ar[k].ph); sb_putc(&out, '\n');
        }
    }
    free(ar); free(ph); smap_free(&names); free(rid_to_oid.v); free(pid_to_fid.v);
    if (rc) { free(out.p); return rc; }
    sb_finish(&out, out_p, out_len);
    return ORC_OK;
}
