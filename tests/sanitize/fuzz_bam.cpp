// CPU-only sanitizer harness for the BAM writer / reader (fzp_format_bam, fzp_bam_to_sam): round trips of random
// alnsets, then the reader on truncated and bit-flipped files.  Must finish without a sanitizer report.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "fzphase.h"
int main() {
    std::mt19937 rng(99);
    long ok = 0, rejected = 0, accepted = 0;
    for (int it = 0; it < 300; it++) {
        // a coordinate-sorted SAM with random records
        std::string sam;
        int pos = 1;
        int nrec = rng() % 40;
        for (int r = 0; r < nrec; r++) {
            pos += rng() % 500;
            int m1 = 1000 + rng() % 3000, ins = rng() % 20, del = rng() % 20, m2 = 1000 + rng() % 2000, clip = rng() % 50;
            std::string cig = std::to_string(clip) + "S" + std::to_string(m1) + "=" + (ins ? std::to_string(ins) + "I" : "") + (del ? std::to_string(del) + "D" : "") + std::to_string(m2) + "X";
            std::string seq(clip + m1 + ins + m2, 'A');
            for (auto &c : seq) c = "ACGTN"[rng() % 5];
            sam += "read/" + std::to_string(rng() % 30) + "\t" + std::to_string((rng() % 2) * 16) + "\tctg\t" + std::to_string(pos) + "\t254\t" + cig + "\t*\t0\t0\t" + seq + "\t*\n";
        }
        fzp_alnset *a = nullptr;
        if (fzp_parse_sam(sam.data(), sam.size(), &a) != 0) continue;
        uint8_t *bam = nullptr, *bai = nullptr;
        size_t nb = 0, ni = 0;
        if (fzp_format_bam(a, "ctg", 10000000, nullptr, &bam, &nb, &bai, &ni) != 0) { fzp_alnset_free(a); continue; }
        char *t = nullptr; size_t n = 0;
        if (fzp_bam_to_sam(bam, nb, "ctg", &t, &n) == 0) { ok++; fzp_free(t); }
        // hostile variants of the same file
        for (int k = 0; k < 40; k++) {
            std::vector<uint8_t> h(bam, bam + nb);
            int mode = rng() % 3;
            if (mode == 0 && !h.empty()) h.resize(rng() % h.size());
            else if (mode == 1) for (int f = 0; f < 1 + (int)(rng() % 8); f++) h[rng() % h.size()] ^= (uint8_t)(1u << (rng() % 8));
            else { size_t at = rng() % h.size(); for (size_t z = at; z < at + 16 && z < h.size(); z++) h[z] = (uint8_t)rng(); }
            if (fzp_bam_to_sam(h.data(), h.size(), k & 1 ? "ctg" : nullptr, &t, &n) == 0) { accepted++; fzp_free(t); } else rejected++;
        }
        fzp_free(bam); fzp_free(bai);
        fzp_alnset_free(a);
    }
    printf("round trips ok=%ld, hostile files: %ld rejected, %ld still parse\n", ok, rejected, accepted);
    return 0;
}
