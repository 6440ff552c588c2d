// Test infrastructure: k_tb_cigar's best-sub-path pass on the host.  The word-level functions are the kernel's own (falcon_unzip_amd/csrc/fzp_cigar_core.h); the wave around
// them -- 64 "lanes" a chunk, the scans over the lanes, the lowest prefix carried from chunk to chunk, the argmax at the end -- is restated here as plain loops, one statement
// per step of the kernel (fzp_align.hip, "pass 0").  cig_serial is the rule itself (oracle/align_oracle.c finish_path:489-505).  tests/test_cigar_core.py holds one against
// the other on random streams under scores that make prefixes tie all the time.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../falcon_unzip_amd/csrc/fzp_cigar_core.h"

namespace {
inline uint32_t op_at(const uint32_t *ops, int64_t k) { return (ops[k >> 4] >> (2 * (k & 15))) & 3u; }
inline uint32_t base_at(const uint32_t *pk, int64_t i) { return (pk[i >> 4] >> (2 * (i & 15))) & 3u; }
inline uint32_t bitrev32(uint32_t v) { uint32_t r = 0; for (int b = 0; b < 32; b++) if (v & (1u << b)) r |= 1u << (31 - b); return r; }
// the kernel's desc16: bases hi, hi - 1, .., hi - 15 in fields 0..15 (a base below 0: 0); the funnel shift spelled with 64 bits
uint32_t desc16(const uint32_t *pk, int32_t hi) {
    uint32_t W = 0;
    const int32_t p0 = hi - 15;
    if (hi >= 0) {
        if (p0 >= 0) { const int32_t wl = p0 >> 4; const uint32_t sh = 2u * (uint32_t)(p0 & 15); const uint64_t two = ((uint64_t)(sh ? pk[wl + 1] : 0u) << 32) | pk[wl]; W = (uint32_t)(two >> sh); }
        else W = pk[0] << (2u * (uint32_t)(-p0));
    }
    return cigc::fields_of_reversed_bits(bitrev32(W));
}
}  // namespace

// the rule: out = {best score, e, s} (score 0: not one matching column, e = s = -1)
extern "C" void cig_serial(const uint32_t *ops, int64_t L, const uint32_t *qpk, const uint32_t *tpk, int64_t i_end, int64_t j_end, int match, int mismatch, int gap, int64_t *out) {
    int64_t i = i_end, j = j_end, Pk = 0, minP = 0, e_min = 0, bestS = 0, s_best = -1, e_best = -1;
    for (int64_t x = 0; x < L; x++) {
        if (Pk <= minP) { minP = Pk; e_min = x; }
        const uint32_t op = op_at(ops, x);
        if (op == 0) { Pk += base_at(qpk, i) == base_at(tpk, j) ? match : -mismatch; i--; j--; }
        else if (op == 1) { Pk -= gap; i--; }
        else { Pk -= gap; j--; }
        if (Pk - minP > bestS) { bestS = Pk - minP; s_best = x; e_best = e_min; }
    }
    out[0] = bestS; out[1] = e_best; out[2] = s_best;
}

// the kernel's pass, lane by lane
extern "C" void cig_wave(const uint32_t *ops, int64_t L64, const uint32_t *qpk, const uint32_t *tpk, int64_t i_end, int64_t j_end, int match, int mismatch, int gap, int64_t *out) {
    using namespace cigc;
    const int32_t L = (int32_t)L64, nW = (L + 15) >> 4;
    auto valid_mask = [&](int32_t wi) -> uint32_t { const int32_t nv = L - 16 * wi < 16 ? L - 16 * wi : 16; return nv >= 16 ? EVEN : (nv <= 0 ? 0u : (((1u << (2 * nv)) - 1u) & EVEN)); };
    Ent lut[256];
    const Scores sc = scores_of(match, mismatch, gap);
    for (int c = 0; c < 256; c++) lut[c] = lut_entry(c, sc);
    int32_t base_S = 0, base_i = 0, base_j = 0, base_min = 0, base_min_pos = 0;
    int32_t bestS[64], bestP[64], bestE[64];
    for (int l = 0; l < 64; l++) { bestS[l] = 0; bestP[l] = -1; bestE[l] = 0; }
    for (int32_t wb = 0; wb < nW; wb += 64) {
        WordOut O[64];
        int32_t ci[64], cj[64];
        uint32_t fMv[64], fIv[64], fDv[64], vmv[64];
        for (int l = 0; l < 64; l++) {
            const int32_t wi = wb + l;
            const uint32_t x = wi < nW ? ops[wi] : 0u, vm = wi < nW ? valid_mask(wi) : 0u;
            fMv[l] = ~(x | (x >> 1)) & vm; fIv[l] = (x & ~(x >> 1)) & vm; fDv[l] = (~x & (x >> 1)) & vm; vmv[l] = vm;
            ci[l] = __builtin_popcount(fMv[l] | fIv[l]); cj[l] = __builtin_popcount(fMv[l] | fDv[l]);
        }
        int32_t si = 0, sj = 0;      // inclusive scans, lane by lane
        for (int l = 0; l < 64; l++) {
            si += ci[l]; sj += cj[l];
            const int32_t i = (int32_t)i_end - (base_i + si - ci[l]), j = (int32_t)j_end - (base_j + sj - cj[l]);
            uint32_t Qs = ci[l] ? desc16(qpk, i) : 0u, Ts = cj[l] ? desc16(tpk, j) : 0u;
            for (uint32_t rd = fDv[l], ri = fIv[l]; (rd | ri) != 0u;) { hole_turn(Qs, rd); hole_turn(Ts, ri); }      // (the wave goes on while ANY lane has one left: a lane that is through changes nothing)
            for (int extra = 0; extra < 3; extra++) { uint32_t z0 = 0, z1 = 0; hole_turn(Qs, z0); hole_turn(Ts, z1); }        // ... shown here: three turns past the end
            const uint32_t E = Qs ^ Ts;
            const uint32_t eqw = fMv[l] & ~(E | (E >> 1)), gapw = fIv[l] | fDv[l] | (EVEN & ~vmv[l]);
            const uint32_t cw = eqw | (gapw << 1);
            Word W = word_begin();
            for (int g = 0; g < 4; g++) word_join(W, lut[(cw >> (8 * g)) & 255u], g);
            O[l] = word_end(W);
        }
        int32_t ss = 0, mv = 0x3fffffff, mp = 0;      // the inclusive (min, position) scan: a later lane keeps its own on a tie
        int32_t mv_prev = 0x3fffffff, mp_prev = 0;
        for (int l = 0; l < 64; l++) {
            const int32_t wi = wb + l;
            ss += O[l].tot;
            const int32_t start = base_S + ss - O[l].tot;
            const int32_t own_v = start + O[l].lmin, own_p = 16 * wi + O[l].lpos;
            mv_prev = mv; mp_prev = mp;
            if (l == 0 || !(mv < own_v)) { mv = own_v; mp = own_p; }
            int32_t gm = mv_prev, gp = mp_prev;      // ... before this word
            if (l == 0 || base_min < gm) { gm = base_min; gp = base_min_pos; }
            const Pick pick = word_pick(O[l], start, gm, gp, wi);
            if (wi < nW && pick.V > bestS[l]) { bestS[l] = pick.V; bestP[l] = pick.s; bestE[l] = pick.e; }
        }
        if (mv <= base_min) { base_min = mv; base_min_pos = mp; }
        base_S += ss; base_i += si; base_j += sj;
    }
    // wave argmax: largest score, then smallest s
    int32_t vS = 0, vP = 0x7fffffff, vE = 0;
    for (int l = 0; l < 64; l++) {
        const int32_t oP = bestP[l] < 0 ? 0x7fffffff : bestP[l];
        if (bestS[l] > vS || (bestS[l] == vS && oP < vP)) { vS = bestS[l]; vP = oP; vE = bestE[l]; }
    }
    if (vP == 0x7fffffff || vS <= 0) { out[0] = 0; out[1] = -1; out[2] = -1; return; }
    out[0] = vS; out[1] = vE; out[2] = vP;
}
