// CPU-only sanitizer harness for the host side of libfzphase (parser, serializers, readmap).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "fzphase.h"
int main() {
    std::mt19937 rng(1234);
    const char *frag[] = {"r1", "r2\t", "\t", "0", "16", "ctg", "1", "100", "254", "2500M", "9000S1000=", "3M2I", "*", "10H5P3N", "12Z3M", "=", "X",
                          "ACGTACGTAC", "NNNN", "\n", " ", "@HD\tVN:1", "-5", "99999999999999999999", "4294967296M", "0M", "1999=", "2000=", ""};
    const int nf = sizeof frag / sizeof frag[0];
    long ok = 0, err = 0;
    for (int it = 0; it < 20000; it++) {
        std::string s;
        int nl = rng() % 6;
        for (int l = 0; l < nl; l++) {
            int nt = rng() % 14;
            for (int t = 0; t < nt; t++) { s += frag[rng() % nf]; s += (rng() % 5 == 0) ? " " : "\t"; }
            if (rng() % 3 == 0) { s += std::string(rng() % 3000, "ACGTN"[rng() % 5]); }
            s += "\n";
        }
        fzp_alnset *a = nullptr;
        int rc = fzp_parse_sam(s.data(), s.size(), &a);
        if (rc == 0) {
            ok++;
            char *t; size_t n;
            if (fzp_format_q_id_map(a, &t, &n) == 0) fzp_free(t);
            if (fzp_format_sam(a, "c", nullptr, &t, &n) == 0) fzp_free(t);
            fzp_alnset_free(a);
        } else err++;
    }
    // well-formed record with long CIGAR
    {
        std::string seq(3000, 'A'), s = "q\t0\tc\t5\t254\t1000M10I990M5D1000M\t*\t0\t0\t" + seq + "\t*\n";
        fzp_alnset *a = nullptr;
        int rc = fzp_parse_sam(s.data(), s.size(), &a);
        printf("wellformed rc=%d n_rec=%lld cols=%lld\n", rc, rc ? 0LL : (long long)a->n_rec, rc ? 0LL : (long long)a->n_columns);
        if (!rc) fzp_alnset_free(a);
    }
    // readmap with hostile tables
    const char *pr = "0 c 1 0 5 2 name/1\n1 c 2 1 3 9 name/2\nshort row\n";
    const char *rr = "name/0\nname/1\nname/2\n", *pi = "x/0/0_1\nx/10/0_1\nbad\nx/999999/0\n", *pc = "0 c 5 0 1\n1 c 5 0 1\n2 c 5 0 1\n3 c 5 0 1\n7 c 5 0 1\n";
    fzp_r2p *rec; int64_t nrec; char *t; size_t n;
    int rc = fzp_readmap(pr, strlen(pr), rr, strlen(rr), pi, strlen(pi), pc, strlen(pc), "c", 0, &rec, &nrec, &t, &n);
    printf("readmap hostile rc=%d (%s)\n", rc, fzp_last_error());
    const char *pr2 = "0 c 1 0 5 2 name/1\n";
    rc = fzp_readmap(pr2, strlen(pr2), rr, strlen(rr), "x/0/0_1\nx/10/0_1\n", 17, "0 c 5 0 1\n1 c 5 0 1\n", 20, "c", 3, &rec, &nrec, &t, &n);
    printf("readmap ok rc=%d nrec=%lld text=%.*s", rc, (long long)nrec, (int)n, t);
    if (!rc) { fzp_free(rec); fzp_free(t); }
    // serializers on empty inputs
    if (fzp_format_variant_pos(nullptr, 0, &t, &n) == 0) fzp_free(t);
    if (fzp_format_atable(nullptr, nullptr, 0, &t, &n) == 0) fzp_free(t);
    if (fzp_format_phased_variants(nullptr, nullptr, 0, &t, &n) == 0) fzp_free(t);
    printf("parse ok=%ld err=%ld\n", ok, err);
    return 0;
}
