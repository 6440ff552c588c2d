// fzp_cigar_core.h -- what a LANE of k_tb_cigar's best-sub-path pass does with its word of sixteen ops (fzalign v1.5's rule, oracle/align_oracle.c finish_path: with
// P(k) = score of the path's first k ops, the alignment is ops e..s with the largest P(s+1) - P(e); ties: the smallest s, then the largest e).  Plain C++: the kernel
// (fzp_align.hip) includes it for the device, tests/cigar_host.cpp for the host, where a loop over "lanes" plays the wave and the whole pass is held against the serial
// rule on random op streams -- with scores such as 1 / 1 / 1 and 1 / 0 / 1, under which every other prefix ties (tests/test_cigar_core.py).
//
// The word's ops are spelled per op as 0 a mismatching column, 1 a matching column, 2 (or 3) a gap.  Everything a sweep over them has to remember lives in scaled keys --
// prefix x 256 with a position in the low byte -- so that "lowest prefix, latest position", "highest prefix, earliest op" and "largest gain, earliest end, its start" are one
// min / max each.  Scores up to 4 096 fit (fzp_align_create checks).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define FZP_CIG_HD __host__ __device__ inline
#else
#define FZP_CIG_HD inline
#endif

namespace cigc {
constexpr uint32_t EVEN = 0x55555555u;

struct alignas(16) Ent { int32_t x, y, z, w; };
struct Scores { int32_t cE, cG, cX; };      // 256 x (match + mismatch), 256 x (mismatch - gap), 256 x mismatch
FZP_CIG_HD Scores scores_of(int match, int mismatch, int gap) { return Scores{256 * (match + mismatch), 256 * (mismatch - gap), 256 * mismatch}; }
FZP_CIG_HD int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }
FZP_CIG_HD int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }

// entry c of the table of four-op groups: the sweep over ops c & 3, (c >> 2) & 3, .. from prefix 0, positions 0..3:
//   x = their score x 256
//   y = lowest (prefix x 256 | 3 - o) BEFORE an op o           (equal prefixes: the later op)
//   z = highest (prefix x 256 | 16 x (3 - o) + 15) AFTER an op  (equal prefixes: the earlier op)
//   w = best z-key minus y-key over e <= s = 256 x gain + 16 x (3 - s) + e + 12
FZP_CIG_HD Ent lut_entry(int c, const Scores &S) {
    int32_t pp = 0, km = 0x7fffffff, xB = (int32_t)0x80000000, xA = (int32_t)0x80000000;
    for (int o = 0; o < 4; o++) {
        km = imin(km, pp | (3 - o));
        const int32_t code = (c >> (2 * o)) & 3;
        pp += code == 1 ? S.cE - S.cX : (code == 0 ? -S.cX : S.cG - S.cX);
        const int32_t tB = pp | ((3 - o) * 16 + 15);
        xB = imax(xB, tB);
        xA = imax(xA, tB - km);
    }
    return Ent{pp, km, xB, xA};
}

// a word's running values, in word-level keys: pk = score so far x 256; kmin = lowest (prefix x 256 | 15 - o) before an op; bB = highest (prefix x 256 | 16 x (15 - o) + 15)
// after an op; bA = best 256 x gain + 16 x (15 - s) + e over e <= s
struct Word { int32_t pk, kmin, bA, bB; };
FZP_CIG_HD Word word_begin() { return Word{0, 15, (int32_t)0x80000000, (int32_t)0x80000000}; }      // (15: the prefix before op 0)
FZP_CIG_HD void word_join(Word &W, const Ent &E, int g) {      // group g (ops 4 g .. 4 g + 3) behind what the word holds
    const int32_t t1 = W.pk + E.z + ((12 - 4 * g) * 16);                                  // the group's highest prefix-after-an-op
    W.bA = imax(W.bA, imax(t1 - W.kmin, E.w + ((12 - 4 * g) * 16 + 4 * g - 12)));         // ... against the lowest prefix BEFORE the group; the group's own best pair
    W.bB = imax(W.bB, t1);
    W.kmin = imin(W.kmin, W.pk + E.y + (12 - 4 * g));
    W.pk += E.x;
}
struct WordOut { int32_t tot, lmin, lpos, Av, sA, eA, Bv, sB; };
FZP_CIG_HD WordOut word_end(const Word &W) {
    return WordOut{W.pk >> 8, W.kmin >> 8, 15 - (W.kmin & 15), W.bA >> 8, 15 - ((W.bA >> 4) & 15), W.bA & 15, W.bB >> 8, 15 - ((W.bB >> 4) & 15)};
}

// The word against the lowest prefix before it (gm at stream position gp; start = the prefix at the word's first op): the serial rule's "P(s+1) - min(gm, lowest prefix
// inside the word up to s)" has the value max(A, B - G); a prefix inside the word wins a tie against gm (it lies later), the smaller s wins among equal values.
struct Pick { int32_t V, s, e; };
FZP_CIG_HD Pick word_pick(const WordOut &O, int32_t start, int32_t gm, int32_t gp, int32_t wi) {
    const int32_t Bg = O.Bv + start - gm;
    const bool takeA = O.Av > Bg || (O.Av == Bg && O.sA <= O.sB);
    return Pick{takeA ? O.Av : Bg, 16 * wi + (takeA ? O.sA : O.sB), takeA ? 16 * wi + O.eA : gp};
}

// one turn of the hole shifts: the lowest D op still in `rem` leaves its read base to the ops above it -- every field above takes the field below (a word without one stays)
FZP_CIG_HD void hole_turn(uint32_t &bases, uint32_t &rem) {
    const uint32_t low = rem & (0u - rem), below = low - 1u;
    bases = (bases & below) | ((bases << 2) & ~below);
    rem ^= low;
}

// 16 x 2-bit fields turned round (field k <-> field 15 - k) given the word's bit reversal
FZP_CIG_HD uint32_t fields_of_reversed_bits(uint32_t rv) { return ((rv >> 1) & EVEN) | ((rv & EVEN) << 1); }
}  // namespace cigc
