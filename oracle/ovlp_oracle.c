/* ovlp_oracle.c -- CPU restatement of falcon_unzip/ovlp_filter_with_phase.py (the overlap filter that consumes
 * rid_to_phase.all).  TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench cpu_baseline legs may
 * use it; the product path is the HIP library.  Pinned against tests/golden_ovlp/ (outputs of the reference itself,
 * tests/golden_ovlp/make_golden_ovlp.py).
 *
 * Text in, text out, line by line as the reference does it:
 *   main                 ovlp_filter_with_phase.py:296-354   (rid map -> arid2phase; three passes over every file)
 *   filter_stage1        :49-143   per query: 5'/3' overlap counts -> ignore list
 *   filter_stage2        :145-186  containment -> contained set
 *   filter_stage3        :188-277  best-n overlaps per end, in-phase first
 * Python semantics kept: str.split() tokenisation, ids compared as strings, `float(l[3]) < 90`, tuple sort with the
 * token list as the last key (list-of-str comparison), a stable sort, groups = runs of equal q among the lines that
 * pass the four phase checks (the group test at :77 / :215 sits after those `continue`s).
 */
#include <ctype.h>
#include <errno.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { const char *p; int n; } tok;
typedef struct { tok key, ctg, blk, ph; } arid_ent;

typedef struct {
    arid_ent *ent; int64_t n_ent, cap_ent;
    int64_t *slot; int64_t n_slot;      /* open addressing over ent indices; later rows overwrite (dict assignment) */
} arid_map;

static uint64_t hash_tok(tok t) {
    uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < t.n; i++) { h ^= (unsigned char)t.p[i]; h *= 1099511628211ull; }
    return h;
}
static int tok_eq(tok a, tok b) { return a.n == b.n && memcmp(a.p, b.p, (size_t)a.n) == 0; }
static int tok_is(tok a, const char *s) { return a.n == (int)strlen(s) && memcmp(a.p, s, (size_t)a.n) == 0; }
static int tok_cmp(tok a, tok b) {      /* Python str comparison (bytes / code points) */
    int m = a.n < b.n ? a.n : b.n;
    int c = memcmp(a.p, b.p, (size_t)m);
    if (c) return c;
    return a.n - b.n;
}
static int is_space(int c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f' || c == '\n'; }

/* str.split(): tokens of one line; returns count (up to max) */
static int split_line(const char *s, const char *e, tok *out, int max) {
    int n = 0;
    while (s < e) {
        while (s < e && is_space((unsigned char)*s)) s++;
        if (s >= e) break;
        const char *b = s;
        while (s < e && !is_space((unsigned char)*s)) s++;
        if (n < max) { out[n].p = b; out[n].n = (int)(s - b); }
        n++;
    }
    return n;
}

static int64_t map_find(const arid_map *m, tok k) {
    if (!m->n_slot) return -1;
    uint64_t h = hash_tok(k) & (uint64_t)(m->n_slot - 1);
    for (;;) {
        int64_t e = m->slot[h];
        if (e < 0) return -1;
        if (tok_eq(m->ent[e].key, k)) return e;
        h = (h + 1) & (uint64_t)(m->n_slot - 1);
    }
}

static int map_build(arid_map *m, const char *txt, size_t len) {     /* :306-309 */
    memset(m, 0, sizeof *m);
    int64_t n_lines = 0;
    for (size_t i = 0; i < len; i++) n_lines += txt[i] == '\n';
    n_lines += 1;
    m->cap_ent = n_lines;
    m->ent = (arid_ent *)malloc((size_t)n_lines * sizeof(arid_ent));
    m->n_slot = 16;
    while (m->n_slot < 2 * n_lines) m->n_slot <<= 1;
    m->slot = (int64_t *)malloc((size_t)m->n_slot * sizeof(int64_t));
    if (!m->ent || !m->slot) return -2;
    for (int64_t i = 0; i < m->n_slot; i++) m->slot[i] = -1;
    const char *s = txt, *end = txt + len;
    while (s < end) {
        const char *nl = memchr(s, '\n', (size_t)(end - s));
        const char *e = nl ? nl : end;
        tok t[5];
        int nt = split_line(s, e, t, 5);
        if (nt < 4) return -1;                                        /* IndexError at :309 */
        int64_t at = map_find(m, t[0]);
        if (at < 0) {
            at = m->n_ent++;
            m->ent[at].key = t[0];
            uint64_t h = hash_tok(t[0]) & (uint64_t)(m->n_slot - 1);
            while (m->slot[h] >= 0) h = (h + 1) & (uint64_t)(m->n_slot - 1);
            m->slot[h] = at;
        }
        m->ent[at].ctg = t[1]; m->ent[at].blk = t[2]; m->ent[at].ph = t[3];
        s = nl ? nl + 1 : end;
    }
    return 0;
}

static int parse_int(tok t, long long *out) {     /* int(str) */
    if (t.n <= 0 || t.n > 30) return -1;
    char buf[32];
    memcpy(buf, t.p, (size_t)t.n); buf[t.n] = 0;
    int i = 0;
    if (buf[i] == '+' || buf[i] == '-') i++;
    if (!buf[i]) return -1;
    for (int k = i; buf[k]; k++) if (!isdigit((unsigned char)buf[k])) return -1;
    errno = 0;
    *out = strtoll(buf, NULL, 10);
    return errno ? -1 : 0;
}
static int parse_float(tok t, double *out) {      /* float(str) */
    if (t.n <= 0 || t.n > 62) return -1;
    char buf[64];
    memcpy(buf, t.p, (size_t)t.n); buf[t.n] = 0;
    for (int k = 0; buf[k]; k++) if (buf[k] == 'x' || buf[k] == 'X' || buf[k] == 'p' || buf[k] == 'P') return -1;
    char *endp;
    *out = strtod(buf, &endp);
    return (*endp || endp == buf) ? -1 : 0;
}

#define MAXTOK 64
typedef struct {
    tok t[MAXTOK]; int nt;
    int64_t q, tt;            /* arid entries */
} line_t;

/* the four checks every stage starts with (:64-73, :153-163, :200-210); 1 = keep, 0 = skip, <0 = error */
static int phase_checks(const arid_map *m, const char *s, const char *e, line_t *L) {
    L->nt = split_line(s, e, L->t, MAXTOK);
    if (L->nt < 2) return -1;                                         /* ValueError: need two values to unpack */
    if (L->nt > MAXTOK) return -1;
    L->q = map_find(m, L->t[0]);
    if (L->q < 0) return 0;
    L->tt = map_find(m, L->t[1]);
    if (L->tt < 0) return 0;
    const arid_ent *a = &m->ent[L->q], *b = &m->ent[L->tt];
    if (!tok_eq(a->ctg, b->ctg)) return 0;
    if (tok_eq(a->blk, b->blk) && !tok_eq(a->ph, b->ph)) return 0;
    return 1;
}

typedef struct { long long ovl, q_s, q_e, q_l, t_s, t_e, t_l; double idt; } fields_t;
static int parse_fields(const line_t *L, fields_t *f, int need_ovl) {
    if (L->nt < 12) return -1;                                        /* IndexError */
    if (need_ovl) { long long v; if (parse_int(L->t[2], &v)) return -1; f->ovl = -v; }
    if (parse_float(L->t[3], &f->idt)) return -1;
    if (parse_int(L->t[5], &f->q_s) || parse_int(L->t[6], &f->q_e) || parse_int(L->t[7], &f->q_l)) return -1;
    if (parse_int(L->t[9], &f->t_s) || parse_int(L->t[10], &f->t_e) || parse_int(L->t[11], &f->t_l)) return -1;
    return 0;
}

typedef struct { long long max_diff, max_cov, min_cov, min_len, bestn; } orc_ovlp_params;

static void stage1_close(long long left, long long right, const orc_ovlp_params *P, int64_t cur, uint8_t *ignore) {   /* :79-87, :123-130 */
    int ig = 0;
    if (llabs(left - right) > P->max_diff) ig = 1;
    else if (left > P->max_cov || right > P->max_cov) ig = 1;
    else if (left < P->min_cov || right < P->min_cov) ig = 1;
    if (ig && cur >= 0) ignore[cur] = 1;     /* cur < 0 is the reference appending None: never matches an id */
    /* :89-94 "sandwiched" test is dead code: every current_q_id passed `q_id in arid2phase` */
}

static int stage1(const arid_map *m, const char *txt, size_t len, const orc_ovlp_params *P, uint8_t *ignore) {
    const char *s = txt, *end = txt + len;
    int64_t cur = -1; int have_q = 0;
    long long left = 0, right = 0;
    while (s < end) {
        const char *nl = memchr(s, '\n', (size_t)(end - s));
        const char *e = nl ? nl : end;
        line_t L;
        int k = phase_checks(m, s, e, &L);
        s = nl ? nl + 1 : end;
        if (k < 0) return -1;
        have_q = 1;                              /* q_id is bound by every line, kept or not (:62, :122) */
        if (!k) continue;
        if (L.q != cur) {                        /* :77 (q_id != None always holds here) */
            stage1_close(left, right, P, cur, ignore);
            left = right = 0;
            cur = L.q;
        }
        fields_t f;
        if (parse_fields(&L, &f, 1)) return -1;
        if (f.idt < 90) continue;
        if (f.q_l < P->min_len || f.t_l < P->min_len) continue;
        if (f.q_s == 0) left++;
        if (f.q_e == f.q_l) right++;
    }
    if (have_q) stage1_close(left, right, P, cur, ignore);
    return 0;
}

static int stage2(const arid_map *m, const char *txt, size_t len, const orc_ovlp_params *P, const uint8_t *ignore, uint8_t *contained) {
    const char *s = txt, *end = txt + len;
    while (s < end) {
        const char *nl = memchr(s, '\n', (size_t)(end - s));
        const char *e = nl ? nl : end;
        line_t L;
        int k = phase_checks(m, s, e, &L);
        s = nl ? nl + 1 : end;
        if (k < 0) return -1;
        if (!k) continue;
        fields_t f;
        if (parse_fields(&L, &f, 0)) return -1;
        if (f.idt < 90) continue;
        if (f.q_l < P->min_len || f.t_l < P->min_len) continue;
        if (ignore[L.q] || ignore[L.tt]) continue;
        tok last = L.t[L.nt - 1];
        if (tok_is(last, "contained")) contained[L.q] = 1;
        if (tok_is(last, "contains")) contained[L.tt] = 1;
    }
    return 0;
}

typedef struct { int ninph; long long negovl, m_range; line_t L; int64_t seq; } cand_t;
static int cand_cmp(const void *a_, const void *b_) {
    const cand_t *a = (const cand_t *)a_, *b = (const cand_t *)b_;
    if (a->ninph != b->ninph) return a->ninph < b->ninph ? -1 : 1;
    if (a->negovl != b->negovl) return a->negovl < b->negovl ? -1 : 1;
    if (a->m_range != b->m_range) return a->m_range < b->m_range ? -1 : 1;
    int n = a->L.nt < b->L.nt ? a->L.nt : b->L.nt;
    for (int i = 0; i < n; i++) { int c = tok_cmp(a->L.t[i], b->L.t[i]); if (c) return c; }
    if (a->L.nt != b->L.nt) return a->L.nt - b->L.nt;
    /* the two appended phase strings are functions of (q, t): equal here */
    return a->seq < b->seq ? -1 : (a->seq > b->seq ? 1 : 0);
}

typedef struct { char *p; size_t n, cap; } sbuf;
static int sb_put(sbuf *b, const char *s, size_t n) {
    if (b->n + n + 1 > b->cap) {
        size_t c = b->cap ? b->cap * 2 : 4096;
        while (c < b->n + n + 1) c *= 2;
        char *q = (char *)realloc(b->p, c);
        if (!q) return -2;
        b->p = q; b->cap = c;
    }
    memcpy(b->p + b->n, s, n); b->n += n; b->p[b->n] = 0;
    return 0;
}

static int emit(const arid_map *m, cand_t *v, int64_t n, long long bestn, sbuf *out) {   /* :221-235 */
    qsort(v, (size_t)n, sizeof(cand_t), cand_cmp);
    for (int64_t i = 0; i < n; i++) {
        const line_t *L = &v[i].L;
        for (int k = 0; k < L->nt; k++) { if (k && sb_put(out, " ", 1)) return -2; if (sb_put(out, L->t[k].p, (size_t)L->t[k].n)) return -2; }
        const arid_ent *pq = &m->ent[L->q], *pt = &m->ent[L->tt];
        const arid_ent *pp[2] = {pq, pt};
        for (int z = 0; z < 2; z++) {
            if (sb_put(out, " ", 1) || sb_put(out, pp[z]->ctg.p, (size_t)pp[z]->ctg.n) || sb_put(out, ".", 1) || sb_put(out, pp[z]->blk.p, (size_t)pp[z]->blk.n) ||
                sb_put(out, ".", 1) || sb_put(out, pp[z]->ph.p, (size_t)pp[z]->ph.n)) return -2;
        }
        if (sb_put(out, "\n", 1)) return -2;
        if (i >= bestn && v[i].m_range > 1000) break;
    }
    return 0;
}

static int stage3(const arid_map *m, const char *txt, size_t len, const orc_ovlp_params *P, const uint8_t *ignore, const uint8_t *contained, sbuf *out) {
    const char *s = txt, *end = txt + len;
    int64_t cur = -1;
    cand_t *lf = NULL, *rt = NULL; int64_t nl_ = 0, nr_ = 0, cl = 0, cr = 0, seq = 0;
    int rc = 0;
    while (s < end) {
        const char *nl = memchr(s, '\n', (size_t)(end - s));
        const char *e = nl ? nl : end;
        line_t L;
        int k = phase_checks(m, s, e, &L);
        s = nl ? nl + 1 : end;
        if (k < 0) { rc = -1; break; }
        if (!k) continue;
        if (cur < 0) cur = L.q;                                        /* :212-214 */
        else if (L.q != cur) {                                         /* :216-238 */
            if ((rc = emit(m, lf, nl_, P->bestn, out)) || (rc = emit(m, rt, nr_, P->bestn, out))) break;
            nl_ = nr_ = 0;
            cur = L.q;
        }
        if (contained[L.q] || contained[L.tt] || ignore[L.q] || ignore[L.tt]) continue;
        fields_t f;
        if (parse_fields(&L, &f, 1)) { rc = -1; break; }
        if (f.idt < 90) continue;
        if (f.q_l < P->min_len || f.t_l < P->min_len) continue;
        const arid_ent *a = &m->ent[cur], *b = &m->ent[L.tt];
        int inphase = tok_eq(a->ctg, b->ctg) && tok_eq(a->blk, b->blk) && tok_eq(a->ph, b->ph);
        cand_t c;
        c.ninph = -inphase; c.negovl = -f.ovl; c.m_range = f.t_l - (f.t_e - f.t_s); c.L = L; c.seq = seq++;
        if (f.q_s == 0) {
            if (nl_ == cl) { cl = cl ? cl * 2 : 64; lf = (cand_t *)realloc(lf, (size_t)cl * sizeof(cand_t)); if (!lf) { rc = -2; break; } }
            lf[nl_++] = c;
        } else if (f.q_e == f.q_l) {
            if (nr_ == cr) { cr = cr ? cr * 2 : 64; rt = (cand_t *)realloc(rt, (size_t)cr * sizeof(cand_t)); if (!rt) { rc = -2; break; } }
            rt[nr_++] = c;
        }
    }
    if (!rc) rc = emit(m, lf, nl_, P->bestn, out);                      /* :262-276 */
    if (!rc) rc = emit(m, rt, nr_, P->bestn, out);
    free(lf); free(rt);
    return rc;
}

static int list_ids(const arid_map *m, const uint8_t *flag, sbuf *out) {
    for (int64_t i = 0; i < m->n_ent; i++)
        if (flag[i]) { if (sb_put(out, m->ent[i].key.p, (size_t)m->ent[i].key.n) || sb_put(out, "\n", 1)) return -2; }
    return 0;
}

/* main (:296-354).  Outputs are malloc'ed (release with orc_free from phasing_oracle.c / free()); ignore_txt and
 * contained_txt list the ids of the two sets, one per line, in rid-map order.  Returns 0, -1 (the reference would
 * have raised) or -2 (out of memory). */
int orc_ovlp_filter(int n_files, const char *const *texts, const size_t *lens, const char *rid_map, size_t map_len, const orc_ovlp_params *P,
                    char **out_txt, size_t *out_len, char **ignore_txt, size_t *ignore_len, char **contained_txt, size_t *contained_len) {
    arid_map m;
    int rc = map_build(&m, rid_map, map_len);
    uint8_t *ignore = NULL, *contained = NULL;
    sbuf out = {0, 0, 0}, ig = {0, 0, 0}, ct = {0, 0, 0};
    if (!rc) {
        ignore = (uint8_t *)calloc((size_t)m.n_ent + 1, 1);
        contained = (uint8_t *)calloc((size_t)m.n_ent + 1, 1);
        if (!ignore || !contained) rc = -2;
    }
    for (int k = 0; k < n_files && !rc; k++) rc = stage1(&m, texts[k], lens[k], P, ignore);
    for (int k = 0; k < n_files && !rc; k++) rc = stage2(&m, texts[k], lens[k], P, ignore, contained);
    for (int k = 0; k < n_files && !rc; k++) rc = stage3(&m, texts[k], lens[k], P, ignore, contained, &out);
    if (!rc) rc = sb_put(&out, "", 0);
    if (!rc) rc = sb_put(&ig, "", 0);
    if (!rc) rc = sb_put(&ct, "", 0);
    if (!rc) rc = list_ids(&m, ignore, &ig);
    if (!rc) rc = list_ids(&m, contained, &ct);
    free(ignore); free(contained); free(m.ent); free(m.slot);
    if (rc) { free(out.p); free(ig.p); free(ct.p); return rc; }
    *out_txt = out.p; *out_len = out.n;
    if (ignore_txt) { *ignore_txt = ig.p; *ignore_len = ig.n; } else free(ig.p);
    if (contained_txt) { *contained_txt = ct.p; *contained_len = ct.n; } else free(ct.p);
    return 0;
}
