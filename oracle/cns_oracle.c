/* cns_oracle.c -- scalar CPU twin of K6, the phased-pile consensus ("fzcns v2" and its one-base predecessor v1).
 * TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench cpu_baseline legs may use it.
 *
 * PARITY UNPINNED against the reference: FALCON_unzip has no consensus code of its own -- haplotig consensus is
 * falcon_kit's falcon_sense / Arrow via `variantCaller` (run_quiver.py:82-97), external and absent.  This file
 * DEFINES the consensus the HIP kernels implement; HIP == twin byte for byte, accuracy is asserted against the
 * simulator's true haplotypes.
 *
 * Inputs are the texts the phasing chain already has: the SAM lines (records accepted exactly as make_het_call
 * accepts them, phasing.py:47-75), `phased_reads` (q_id ctg block phase n0 n1 QNAME, phasing.py:478-480) and
 * `phased_variants` (P lines: id min max ..., phasing.py:418, 1-based positions).
 * For block b and phase p the pile is every accepted record whose q_id has a (b, p) row; over the block's span
 * [min, max] each reference position tallies A/C/G/T and deletions from the records' CIGAR walk (S, I advance the query;
 * M,=,X one column per base; D advances the reference; N,H,P nothing -- the walk of phasing.py:77-96), and every I op
 * that follows a consumed reference position tallies one insertion (and its first base) on that position.
 * Call per position, cov = A+C+G+T+del:
 *     cov == 0            -> the contig's base
 *     2*del > cov         -> nothing
 *     else                -> the most frequent base (ties: the contig's base if it is among them, else A<C<G<T)
 *     v1: then 2*ins > cov -> the most frequent first inserted base (ties A<C<G<T)            [ins = I ops following the position]
 *     v2: inserted bases are chosen one level at a time, the way falcon_sense's tag graph links (t_pos, delta, base) nodes to their
 *         predecessors: level d = 1..8 looks at the I ops of the pile at this position that are at least d long and whose first
 *         d-1 bases are the ones already chosen; the most frequent base at level d (ties A<C<G<T) is emitted if 2*count > cov,
 *         otherwise the insertion ends.  (A tag's weight in falcon_sense is its link count minus half the coverage: the same gate.)
 *     v3 (the default, "fzcns v3"): noisy reads spell a 3-base inserted het as 2, 3, 4 or 5 bases -- no single spelling reaches half of the
 *         coverage, and v2 loses the insertion.  v3 decides the LENGTH first: L = the largest l <= 8 such that more than half of the
 *         coverage carries an I op of at least l bases at this position (the median inserted length over the pile's reads, 0 for a read
 *         without one); a one-base insertion is what single-molecule reads produce by themselves (homopolymers above all), so L = 1 is kept
 *         only with v2's gate, more than half of the coverage inserting the SAME base; then the bases, level d = 1..L: the most
 *         frequent base at level d among the position's I ops that are at least d long (ties A<C<G<T).
 * Output: one FASTA record per (block, phase) with at least one record in its pile, blocks ascending, phase 0 then 1:
 *     >{ctg}_{block:03d}_{phase} {min} {max} {n_records}\n{sequence}\n
 */
#include <ctype.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { const char *s; size_t n; } ctok;
static int c_split(const char *l, size_t n, ctok *t, int maxt, int tabs_only) {
    int k = 0;
    size_t i = 0;
    while (i < n && k < maxt) {
        if (!tabs_only) while (i < n && isspace((unsigned char)l[i])) i++;
        if (i >= n && !tabs_only) break;
        size_t b = i;
        if (tabs_only) while (i < n && l[i] != '\t') i++; else while (i < n && !isspace((unsigned char)l[i])) i++;
        t[k].s = l + b; t[k].n = i - b; k++;
        if (tabs_only) { if (i < n) i++; else break; }
    }
    return k;
}
static long long c_int(ctok t) { char b[32]; size_t n = t.n < 31 ? t.n : 31; memcpy(b, t.s, n); b[n] = 0; return atoll(b); }

typedef struct { char *name; size_t nlen; } qname_t;
typedef struct { int qid; long long pos; const char *cig; size_t cn; const char *seq; size_t sn; } crec;
typedef struct { int qid, block, phase; } prow;
typedef struct { int id; long long lo, hi; } blk_t;

static int code_of(unsigned char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; }

typedef struct { size_t x; const char *s; long long n; } ins_t;      /* an I op: position index in the block, its bases */

int orc_consensus_v(int version, const char *sam, size_t sam_len, const char *ref_seq, size_t ref_len, const char *phased_reads, size_t pr_len,
                    const char *phased_variants, size_t pv_len, const char *ctg_id, char **out_txt, size_t *out_len) {
    /* ---- records (phasing.py:42-75) */
    size_t n_lines = 1;
    for (size_t i = 0; i < sam_len; i++) n_lines += sam[i] == '\n';
    crec *rec = (crec *)malloc(n_lines * sizeof(crec));
    qname_t *qn = (qname_t *)malloc(n_lines * sizeof(qname_t));
    size_t n_rec = 0, n_q = 0;
    size_t off = 0;
    while (off < sam_len) {
        const char *l = sam + off;
        const char *e = memchr(l, '\n', sam_len - off);
        size_t ln = e ? (size_t)(e - l) : sam_len - off;
        off += ln + (e ? 1 : 0);
        while (ln && (l[ln - 1] == '\r' || l[ln - 1] == ' ' || l[ln - 1] == '\t')) ln--;
        if (!ln || l[0] == '@') continue;
        ctok t[12];
        int nt = c_split(l, ln, t, 12, 0);
        if (nt < 10) { free(rec); free(qn); return -1; }
        int qid = -1;
        for (size_t k = 0; k < n_q; k++) if (qn[k].nlen == t[0].n && memcmp(qn[k].name, t[0].s, t[0].n) == 0) { qid = (int)k; break; }
        if (qid < 0) { qn[n_q].name = (char *)t[0].s; qn[n_q].nlen = t[0].n; qid = (int)n_q++; }
        long long total = 0, skip = 0;
        for (size_t i = 0; i < t[5].n;) {
            long long v = 0; size_t j = i;
            while (j < t[5].n && isdigit((unsigned char)t[5].s[j])) v = v * 10 + (t[5].s[j++] - '0');
            if (j == i || j >= t[5].n) break;
            if (strchr("MIDNSHP=X", t[5].s[j])) { total += v; if (t[5].s[j] == 'S') skip += v; }
            i = j + 1;
        }
        if (total == 0) { free(rec); free(qn); return -1; }
        if (1.0 - 1.0 * (double)skip / (double)total < 0.1) continue;
        if (total < 2000) continue;
        rec[n_rec].qid = qid; rec[n_rec].pos = c_int(t[3]) - 1; rec[n_rec].cig = t[5].s; rec[n_rec].cn = t[5].n; rec[n_rec].seq = t[9].s; rec[n_rec].sn = t[9].n;
        n_rec++;
    }
    /* ---- phased_reads rows and blocks */
    size_t pcap = 1;
    for (size_t i = 0; i < pr_len; i++) pcap += phased_reads[i] == '\n';
    prow *pr = (prow *)malloc(pcap * sizeof(prow));
    size_t n_pr = 0;
    off = 0;
    while (off < pr_len) {
        const char *l = phased_reads + off;
        const char *e = memchr(l, '\n', pr_len - off);
        size_t ln = e ? (size_t)(e - l) : pr_len - off;
        off += ln + (e ? 1 : 0);
        ctok t[8];
        if (c_split(l, ln, t, 8, 0) < 4) continue;
        pr[n_pr].qid = (int)c_int(t[0]); pr[n_pr].block = (int)c_int(t[2]); pr[n_pr].phase = (int)c_int(t[3]); n_pr++;
    }
    size_t bcap = 1;
    for (size_t i = 0; i < pv_len; i++) bcap += phased_variants[i] == '\n';
    blk_t *blk = (blk_t *)malloc(bcap * sizeof(blk_t));
    size_t n_blk = 0;
    off = 0;
    while (off < pv_len) {
        const char *l = phased_variants + off;
        const char *e = memchr(l, '\n', pv_len - off);
        size_t ln = e ? (size_t)(e - l) : pv_len - off;
        off += ln + (e ? 1 : 0);
        ctok t[8];
        if (c_split(l, ln, t, 8, 0) < 4 || t[0].n != 1 || t[0].s[0] != 'P') continue;
        blk[n_blk].id = (int)c_int(t[1]); blk[n_blk].lo = c_int(t[2]) - 1; blk[n_blk].hi = c_int(t[3]) - 1; n_blk++;
    }
    /* ---- tally and call */
    size_t cap = 1 << 16, n_out = 0;
    char *out = (char *)malloc(cap);
#define PUT(ptr, len_)                                                              \
    do {                                                                            \
        size_t l__ = (len_);                                                        \
        if (n_out + l__ + 1 > cap) { while (n_out + l__ + 1 > cap) cap *= 2; out = (char *)realloc(out, cap); } \
        memcpy(out + n_out, (ptr), l__); n_out += l__;                              \
    } while (0)
    for (size_t b = 0; b < n_blk; b++) {
        const long long lo = blk[b].lo, hi = blk[b].hi;
        if (hi < lo || lo < 0 || (size_t)hi >= ref_len) continue;
        const size_t L = (size_t)(hi - lo + 1);
        for (int ph = 0; ph < 2; ph++) {
            uint32_t *cnt = (uint32_t *)calloc(L * 10, sizeof(uint32_t));
            long long n_used = 0;
            ins_t *ins = NULL; size_t n_ins = 0, ins_cap = 0;
            for (size_t r = 0; r < n_rec; r++) {
                int use = 0;
                for (size_t k = 0; k < n_pr; k++) if (pr[k].qid == rec[r].qid && pr[k].block == blk[b].id && pr[k].phase == ph) { use = 1; break; }
                if (!use) continue;
                /* reference span, to count a record only if it reaches the block */
                long long rp = rec[r].pos, qp = 0, span = 0;
                for (size_t i = 0; i < rec[r].cn;) {
                    long long v = 0; size_t j = i;
                    while (j < rec[r].cn && isdigit((unsigned char)rec[r].cig[j])) v = v * 10 + (rec[r].cig[j++] - '0');
                    if (j == i || j >= rec[r].cn) break;
                    char op = rec[r].cig[j];
                    if (op == 'M' || op == '=' || op == 'X' || op == 'D') span += v;
                    i = j + 1;
                }
                if (rec[r].pos > hi || rec[r].pos + span <= lo) continue;
                n_used++;
                for (size_t i = 0; i < rec[r].cn;) {
                    long long v = 0; size_t j = i;
                    while (j < rec[r].cn && isdigit((unsigned char)rec[r].cig[j])) v = v * 10 + (rec[r].cig[j++] - '0');
                    if (j == i || j >= rec[r].cn) break;
                    char op = rec[r].cig[j];
                    i = j + 1;
                    if (op == 'S') qp += v;
                    else if (op == 'I') {
                        long long pp = rp - 1;
                        long long qi = qp;                      /* first inserted base */
                        if (pp >= rec[r].pos && pp >= lo && pp <= hi && v > 0 && (size_t)qi < rec[r].sn) {
                            cnt[(size_t)(pp - lo) * 10 + 5]++;
                            int c = code_of((unsigned char)rec[r].seq[qi]);
                            if (c < 4) cnt[(size_t)(pp - lo) * 10 + 6 + c]++;
                            if (v >= 2) {
                                if (n_ins == ins_cap) { ins_cap = ins_cap ? ins_cap * 2 : 1024; ins = (ins_t *)realloc(ins, ins_cap * sizeof(ins_t)); }
                                long long nn = v;
                                if ((size_t)(qi + nn) > rec[r].sn) nn = (long long)rec[r].sn - qi;
                                ins[n_ins].x = (size_t)(pp - lo); ins[n_ins].s = rec[r].seq + qi; ins[n_ins].n = nn; n_ins++;
                            }
                        }
                        qp += v;
                    } else if (op == 'M' || op == '=' || op == 'X') {
                        for (long long d = 0; d < v; d++, rp++, qp++) {
                            if (rp < lo || rp > hi || (size_t)qp >= rec[r].sn) continue;
                            int c = code_of((unsigned char)rec[r].seq[qp]);
                            if (c < 4) cnt[(size_t)(rp - lo) * 10 + c]++;
                        }
                    } else if (op == 'D') {
                        for (long long d = 0; d < v; d++, rp++) if (rp >= lo && rp <= hi) cnt[(size_t)(rp - lo) * 10 + 4]++;
                    }
                }
            }
            if (n_used > 0) {
                char hdr[256];
                int hl = snprintf(hdr, sizeof hdr, ">%s_%03d_%d %lld %lld %lld\n", ctg_id, blk[b].id, ph, lo + 1, hi + 1, n_used);
                PUT(hdr, (size_t)hl);
                for (size_t x = 0; x < L; x++) {
                    const uint32_t *c = cnt + x * 10;
                    uint32_t cov = c[0] + c[1] + c[2] + c[3] + c[4];
                    char refb = (char)toupper((unsigned char)ref_seq[lo + (long long)x]);
                    if (cov == 0) { PUT(&refb, 1); continue; }
                    if (2 * c[4] <= cov) {
                        uint32_t mx = c[0];
                        for (int k = 1; k < 4; k++) if (c[k] > mx) mx = c[k];
                        int rc = code_of((unsigned char)refb), pick = -1;
                        if (rc < 4 && c[rc] == mx) pick = rc;
                        for (int k = 0; k < 4 && pick < 0; k++) if (c[k] == mx) pick = k;
                        char ch = "ACGT"[pick];
                        PUT(&ch, 1);
                    }
                    if (version == 1) {
                        if (2 * c[5] > cov) {
                            uint32_t mx = c[6];
                            int pick = 0;
                            for (int k = 1; k < 4; k++) if (c[6 + k] > mx) { mx = c[6 + k]; pick = k; }
                            if (mx > 0) { char ch = "ACGT"[pick]; PUT(&ch, 1); }
                        }
                    } else if (version >= 3) {
                        /* v3: WHETHER there is an insertion is decided on all I ops of the (left-normalised) column -- more than half of the
                         * coverage, as v1; HOW LONG it is, by their lower median length (noisy reads spell a 3-base insertion as 2, 3, 4 or 5
                         * bases: no single spelling has a majority, the length does); WHICH bases, level by level, by the most frequent base at
                         * that level among the ops long enough to have one (ties A<C<G<T).  Lengths above 8 count as 8. */
                        if (2 * c[5] > cov) {
                            uint32_t hist[9] = {0};
                            uint32_t lvl[8][4];
                            memset(lvl, 0, sizeof lvl);
                            uint32_t n_long = 0;
                            for (size_t e = 0; e < n_ins; e++) {
                                if (ins[e].x != x) continue;
                                long long ln = ins[e].n > 8 ? 8 : ins[e].n;
                                hist[ln]++; n_long++;
                                for (long long q = 1; q < ln; q++) { int cc = code_of((unsigned char)ins[e].s[q]); if (cc < 4) lvl[q][cc]++; }
                            }
                            hist[1] = c[5] - n_long;
                            for (int k = 0; k < 4; k++) lvl[0][k] = c[6 + k];
                            uint32_t ge = c[5]; int Ls = 0;                        /* ge = I ops at least ln long */
                            for (int ln = 1; ln <= 8; ln++) { if (2 * ge > cov) Ls = ln; else break; ge -= hist[ln]; }   /* the median inserted length over the pile's reads (0 for a read without one) */
                            /* a one-base insertion is what single-molecule reads produce by themselves, above all inside homopolymers (a run of
                             * five sees one in most reads): it is believed only with v2's gate, more than half of the coverage inserting the SAME base */
                            if (Ls == 1) { uint32_t mx1 = c[6]; for (int k = 1; k < 4; k++) if (c[6 + k] > mx1) mx1 = c[6 + k]; if (!(2 * mx1 > cov)) Ls = 0; }
                            for (int q = 0; q < Ls; q++) {
                                uint32_t mx = lvl[q][0]; int pick = 0;
                                for (int k = 1; k < 4; k++) if (lvl[q][k] > mx) { mx = lvl[q][k]; pick = k; }
                                if (mx == 0) break;
                                char ch = "ACGT"[pick];
                                PUT(&ch, 1);
                            }
                        }
                    } else {
                        int chosen[8];
                        int nd = 0;
                        uint32_t lv[4] = {c[6], c[7], c[8], c[9]};
                        for (;;) {
                            uint32_t mx = lv[0];
                            int pick = 0;
                            for (int k = 1; k < 4; k++) if (lv[k] > mx) { mx = lv[k]; pick = k; }
                            if (!(2 * mx > cov)) break;
                            chosen[nd++] = pick;
                            char ch = "ACGT"[pick];
                            PUT(&ch, 1);
                            if (nd == 8) break;
                            lv[0] = lv[1] = lv[2] = lv[3] = 0;       /* next level: I ops here that are longer and spell the chosen bases so far */
                            for (size_t e = 0; e < n_ins; e++) {
                                if (ins[e].x != x || ins[e].n <= nd) continue;
                                int okp = 1;
                                for (int q = 0; q < nd && okp; q++) okp = code_of((unsigned char)ins[e].s[q]) == chosen[q];
                                if (!okp) continue;
                                int cc = code_of((unsigned char)ins[e].s[nd]);
                                if (cc < 4) lv[cc]++;
                            }
                        }
                    }
                }
                PUT("\n", 1);
            }
            free(cnt);
            free(ins);
        }
    }
    out[n_out] = 0;
    free(rec); free(qn); free(pr); free(blk);
    *out_txt = out; *out_len = n_out;
    return 0;
}

int orc_consensus(const char *sam, size_t sam_len, const char *ref_seq, size_t ref_len, const char *phased_reads, size_t pr_len,
                  const char *phased_variants, size_t pv_len, const char *ctg_id, char **out_txt, size_t *out_len) {
    return orc_consensus_v(3, sam, sam_len, ref_seq, ref_len, phased_reads, pr_len, phased_variants, pv_len, ctg_id, out_txt, out_len);
}
int orc_consensus_v2(const char *sam, size_t sam_len, const char *ref_seq, size_t ref_len, const char *phased_reads, size_t pr_len,
                     const char *phased_variants, size_t pv_len, const char *ctg_id, char **out_txt, size_t *out_len) {
    return orc_consensus_v(2, sam, sam_len, ref_seq, ref_len, phased_reads, pr_len, phased_variants, pv_len, ctg_id, out_txt, out_len);
}
int orc_consensus_v1(const char *sam, size_t sam_len, const char *ref_seq, size_t ref_len, const char *phased_reads, size_t pr_len,
                     const char *phased_variants, size_t pv_len, const char *ctg_id, char **out_txt, size_t *out_len) {
    return orc_consensus_v(1, sam, sam_len, ref_seq, ref_len, phased_reads, pr_len, phased_variants, pv_len, ctg_id, out_txt, out_len);
}

/* ---- the twin with a TEMPLATE argument (fzp_polish_tigs: the consensus role of run_quiver.py:82-97 with a tig as the template).  The pile is every accepted record
 * of the SAM text (the tig's reads aligned to the tig), the span the whole tig: one block [1, tig_len], every q_id a member of (block 1, phase 0) -- the same tally,
 * the same fzcns v3 call.  Output: the polished sequence alone (no header); a tig without an accepted record comes back as it is. */
int orc_polish(const char *sam, size_t sam_len, const char *tig, size_t tig_len, char **out_txt, size_t *out_len) {
    size_t n_lines = 1;
    for (size_t i = 0; i < sam_len; i++) n_lines += sam[i] == '\n';
    char *pr = (char *)malloc(n_lines * 40 + 64), pv[96];
    size_t n_pr = 0;
    for (size_t q = 0; q < n_lines; q++) n_pr += (size_t)sprintf(pr + n_pr, "%zu tig 1 0 0 0 r\n", q);      /* (more q_ids than the text has names: the extra rows match nothing) */
    const int n_pv = sprintf(pv, "P 1 1 %zu %zu 1 1.0\n", tig_len, tig_len);
    char *fa = NULL;
    size_t n_fa = 0;
    const int rc = orc_consensus_v(3, sam, sam_len, tig, tig_len, pr, n_pr, pv, (size_t)n_pv, "tig", &fa, &n_fa);
    free(pr);
    if (rc) return rc;
    char *out = (char *)malloc(tig_len + n_fa + 1);
    size_t n_out = 0;
    const char *nl = n_fa ? (const char *)memchr(fa, '\n', n_fa) : NULL;
    if (nl) {      /* >tig_001_0 1 L n \n sequence \n */
        const char *s = nl + 1, *e = (const char *)memchr(s, '\n', n_fa - (size_t)(s - fa));
        n_out = e ? (size_t)(e - s) : n_fa - (size_t)(s - fa);
        memcpy(out, s, n_out);
    } else {       /* no pile: the template */
        for (size_t i = 0; i < tig_len; i++) out[n_out++] = (char)toupper((unsigned char)tig[i]);
    }
    free(fa);
    *out_txt = out; *out_len = n_out;
    return 0;
}
