// Host build of the bit-sliced DP (falcon_unzip_amd/csrc/fzp_swb_core.h) with the loop around the cell function written the way k_swb runs it per
// lane: planes, windows, forbid bits, steering by two tracked edge scores, terminal by accumulated differences along the border.  Test infrastructure
// (tests/test_swb_core.py compares it with the scalar twin's orc_dp_extend_raw); built by the test with g++.
#include <cstdint>
#include <cstring>
#include "../falcon_unzip_amd/csrc/fzp_swb_core.h"

// BAND: the band's cells (fzalign v1.8: 64 or 32), BW: a word that wide.  k_swb runs the 64-cell form as written in swb_step, the 32-cell form as in swb_step32.
template <class BW, int BAND>
static int swb_extend_host_t(const uint8_t *q, int64_t nq, const uint8_t *t, int64_t nt, uint64_t *tbD, uint64_t *tbG, uint8_t *mv, int64_t *out, int inner) {
    using namespace swb;
    constexpr int HB = BAND / 2, LAST = BAND - 1;
    if (nq < BAND || nt < BAND) return -1;                    // (the kernel leaves such extensions to k_sw)
    const int64_t max_steps = nq + nt + 2;
    // step -1: the anti-diagonal i + j = -1 of the virtual border, lane k = cell (k - HB - 1, HB - k):  Pv = 0 where j >= 0 (k <= HB) else 4, Qv = 0 where i >= 0 (k > HB) else 4
    PlanesT<BW> P = {0, 0, (BW)(~(BW)0 << (HB + 1))}, Q = {0, 0, (BW)(((BW)1 << (HB + 1)) - 1)};
    BW R0 = 0, R1 = 0, C0 = 0, C1 = 0;                        // windows: bit k = read base i0 + k / contig base t - i0 - k (bit planes of the 2-bit codes)
    for (int k = 0; k < BAND; k++) {
        const int64_t i = k - (HB + 1), j = HB - k;
        if (i >= 0 && i < nq) { R0 |= (BW)(q[i] & 1) << k; R1 |= (BW)(q[i] >> 1) << k; }
        if (j >= 0 && j < nt) { C0 |= (BW)(t[j] & 1) << k; C1 |= (BW)(t[j] >> 1) << k; }
    }
    int64_t i0 = -(HB + 1), tt = 0;
    int32_t S0 = -3 - 8 * HB, E2 = 8;                         // score of lane 0's cell (-259 / -131); (score of the last lane's - S0) / 2 -- at step -1 from the border's closed form
    bool down = true, pdown = false;
    bool row_on = false, col_on = false;
    int32_t Hrow = 0, Hcol = 0, best = -(1 << 26);
    int64_t bt = -1, bl = -1;
    for (;;) {
        PlanesT<BW> p, qq;
        if (down) {
            i0++;
            R0 >>= 1; R1 >>= 1;
            const int64_t i = i0 + LAST;
            if (i >= 0 && i < nq) { R0 |= (BW)(q[i] & 1) << LAST; R1 |= (BW)(q[i] >> 1) << LAST; }
            p = P; qq = {(BW)(Q.v0 >> 1), (BW)(Q.v1 >> 1), (BW)(Q.v2 >> 1)};
        } else {
            C0 <<= 1; C1 <<= 1;
            const int64_t j = tt - i0;
            if (j >= 0 && j < nt) { C0 |= (BW)(t[j] & 1); C1 |= (BW)(t[j] >> 1); }
            p = {(BW)(P.v0 << 1), (BW)(P.v1 << 1), (BW)(P.v2 << 1)}; qq = Q;
        }
        BW xm = (R0 ^ C0) | (R1 ^ C1);
        {   // bases past the read's / the window's end never match (tail of the extension only)
            const int64_t nv = nq - i0;                                                 // lanes k < nv hold read bases
            if (nv < BAND) xm |= nv <= 0 ? ~(BW)0 : (BW)~(((BW)1 << nv) - 1);
            const int64_t kmin = tt - i0 - nt + 1;                                      // lanes k >= kmin hold contig bases
            if (kmin > 0) xm |= kmin >= BAND ? ~(BW)0 : (BW)(((BW)1 << kmin) - 1);
        }
        const BW f = (down && pdown) ? (BW)1 << LAST : ((!down && !pdown) ? (BW)1 : (BW)0);
        BW D, G;
        cells<BW>(xm, f, down ? ~(BW)0 : (BW)0, p, qq, &P, &Q, &D, &G);
        tbD[tt] = (uint64_t)D; tbG[tt] = (uint64_t)G; mv[tt] = down ? 1 : 0;
        // edge scores: every lane's cell moved down (vertical difference) or right (horizontal)
        const PlanesT<BW> &X = down ? Q : P;
        const int32_t v0 = value_at(X, 0), vl = value_at(X, LAST);
        S0 += 2 * v0 - 3;
        E2 += vl - v0;
        // terminal: best valid cell of the last row / the last column, by differences along them
        {
            const int64_t kc = tt - (nt - 1) - i0, kr = nq - 1 - i0;
            if (!col_on && !down && kc == 0) { col_on = true; Hcol = S0; }
            else if (col_on && kc >= 0 && kc <= LAST) Hcol += 2 * value_at(Q, (int)kc) - 3;
            if (!row_on && down && kr == LAST) { row_on = true; Hrow = S0 + 2 * E2; }
            else if (row_on && kr >= 0 && kr <= LAST) Hrow += 2 * value_at(P, (int)kr) - 3;
            // (inner pieces, fzalign v1.6: a border cell is valued by the global alignment through it -- minus the gap moves from it to the corner)
            if (col_on && kc >= 0 && kc <= LAST) { const int64_t i = i0 + kc; const int32_t vc = inner ? Hcol - 3 * (int32_t)(nq - 1 - i) : Hcol; if (i >= 0 && i < nq && vc > best) { best = vc; bt = tt; bl = kc; } }
            if (row_on && kr >= 0 && kr <= LAST) { const int64_t j = tt - (nq - 1); const int32_t vr = inner ? Hrow - 3 * (int32_t)(nt - 1 - j) : Hrow; if (j >= 0 && j < nt && vr > best) { best = vr; bt = tt; bl = kr; } }
        }
        pdown = down;
        tt++;
        down = tt < BAND ? ((tt & 1) == 0) : (E2 >= 0);
        if (i0 > nq - 1) break;
        if ((tt - 1) - (i0 + LAST) > nt - 1) break;
        if (tt >= max_steps) break;
    }
    out[0] = tt; out[1] = bl >= 0 ? best : -(1 << 26); out[2] = bt; out[3] = bl;
    return 0;
}
extern "C" int swb_extend_host(const uint8_t *q, int64_t nq, const uint8_t *t, int64_t nt, uint64_t *tbD, uint64_t *tbG, uint8_t *mv, int64_t *out, int inner) {
    return swb_extend_host_t<uint64_t, 64>(q, nq, t, nt, tbD, tbG, mv, out, inner);
}
extern "C" int swb_extend_host32(const uint8_t *q, int64_t nq, const uint8_t *t, int64_t nt, uint64_t *tbD, uint64_t *tbG, uint8_t *mv, int64_t *out, int inner) {
    return swb_extend_host_t<uint32_t, 32>(q, nq, t, nt, tbD, tbG, mv, out, inner);
}


// ---- the same extension with the band split over a pair of lanes (k_swb2's form): two Half states, explicit exchange where the kernel uses a DPP swap
static uint32_t rev32(uint32_t v) { uint32_t r = 0; for (int b = 0; b < 32; b++) r |= ((v >> b) & 1u) << (31 - b); return r; }

extern "C" int swb_extend_pair_host(const uint8_t *q, int64_t nq, const uint8_t *t, int64_t nt, uint64_t *tbD, uint64_t *tbG, uint8_t *mv, int64_t *out, int inner) {
    using namespace swb;
    if (nq < 64 || nt < 64) return -1;
    const int64_t max_steps = nq + nt + 2;
    Half h[2];
    // step -1 (see swb_extend_host): P value 4 on cells >= 33, Q value 4 on cells <= 32.  Low: A = P, B = Q, bit = cell; high: A = Q, B = P, bit = 63 - cell
    h[0].A = {0, 0, 0}; h[0].B = {0, 0, ~0u};                   // cells 0..31: P = 0, Q = 4
    h[1].A = {0, 0, 1u << 31}; h[1].B = {0, 0, ~0u >> 1};       // Q = 4 on cell 32 only (bit 31); P = 4 on cells 33..63 (bits 30..0)
    for (int l = 0; l < 2; l++) h[l].Wm0 = h[l].Wm1 = h[l].Wo0 = h[l].Wo1 = 0;
    for (int k = 0; k < 64; k++) {
        const int64_t i = k - 33, j = 32 - k;
        const int l = k >> 5, b = l ? 63 - k : k;
        if (i >= 0 && i < nq) { uint32_t &w0 = l ? h[1].Wm0 : h[0].Wo0, &w1 = l ? h[1].Wm1 : h[0].Wo1; w0 |= (uint32_t)(q[i] & 1) << b; w1 |= (uint32_t)(q[i] >> 1) << b; }   // read: low's other window, high's own
        if (j >= 0 && j < nt) { uint32_t &w0 = l ? h[1].Wo0 : h[0].Wm0, &w1 = l ? h[1].Wo1 : h[0].Wm1; w0 |= (uint32_t)(t[j] & 1) << b; w1 |= (uint32_t)(t[j] >> 1) << b; }   // contig: low's own, high's other
    }
    int64_t i0 = -33, tt = 0;
    int32_t sv0 = 0, E2 = 8;
    bool down = true, pdown = false;
    bool row_on = false, col_on = false;
    int32_t Hrow = 0, Hcol = 0, best = -(1 << 26);
    int64_t bt = -1, bl = -1;
    for (;;) {
        const uint32_t sd = down ? 1u : 0u, sr = 1u - sd;
        if (down) i0++;
        const int64_t kr = nq - 1 - i0, kc = tt - (nt - 1) - i0, nv = kr + 1;
        const HalfOut o0 = half_out(h[0]), o1 = half_out(h[1]);
        uint32_t base0 = 0, base1 = 0;
        if (!down) { const int64_t j = tt - i0; if (j >= 0 && j < nt) base0 = t[j]; }
        else { const int64_t i = i0 + 63; if (i >= 0 && i < nq) base1 = q[i]; }
        // cells past the ends: low lane bit c = cell c, high lane bit b = cell 63 - b
        uint32_t bad0 = 0, bad1 = 0;
        if (nv < 32) bad0 |= nv <= 0 ? ~0u : ~0u << nv;
        if (nv < 64) bad1 |= nv <= 32 ? ~0u : (1u << (64 - nv)) - 1u;
        if (kc > 0) bad0 |= kc >= 32 ? ~0u : (1u << kc) - 1u;
        if (kc > 32) bad1 |= kc >= 64 ? ~0u : ~0u << (64 - kc);
        uint32_t D0, G0, D1, G1;
        half_step(h[0], sr, base0, o1, sr & (pdown ? 0u : 1u), bad0, &D0, &G0);
        half_step(h[1], sd, base1, o0, sd & (pdown ? 1u : 0u), bad1, &D1, &G1);
        tbD[tt] = (uint64_t)D0 | ((uint64_t)rev32(D1) << 32); tbG[tt] = (uint64_t)G0 | ((uint64_t)rev32(G1) << 32); mv[tt] = down ? 1 : 0;
        const int32_t v0 = half_edge(h[0], sr), v63 = half_edge(h[1], sd);
        sv0 += v0;
        E2 += v63 - v0;
        {   // terminal: P at cell kr is the low lane's A / the high lane's B; Q at cell kc the low lane's B / the high lane's A
            const int32_t S0 = -259 + 2 * sv0 - 3 * (int32_t)(tt + 1);
            auto Pval = [&](int64_t c) { return c < 32 ? value_at(h[0].A, (int)c) : value_at(h[1].B, (int)(63 - c)); };
            auto Qval = [&](int64_t c) { return c < 32 ? value_at(h[0].B, (int)c) : value_at(h[1].A, (int)(63 - c)); };
            if (!col_on && !down && kc == 0) { col_on = true; Hcol = S0; }
            else if (col_on && kc >= 0 && kc <= 63) Hcol += 2 * Qval(kc) - 3;
            if (!row_on && down && kr == 63) { row_on = true; Hrow = S0 + 2 * E2; }
            else if (row_on && kr >= 0 && kr <= 63) Hrow += 2 * Pval(kr) - 3;
            // (inner pieces, fzalign v1.6: a border cell is valued by the global alignment through it -- minus the gap moves from it to the corner)
            if (col_on && kc >= 0 && kc <= 63) { const int64_t i = i0 + kc; const int32_t vc = inner ? Hcol - 3 * (int32_t)(nq - 1 - i) : Hcol; if (i >= 0 && i < nq && vc > best) { best = vc; bt = tt; bl = kc; } }
            if (row_on && kr >= 0 && kr <= 63) { const int64_t j = tt - (nq - 1); const int32_t vr = inner ? Hrow - 3 * (int32_t)(nt - 1 - j) : Hrow; if (j >= 0 && j < nt && vr > best) { best = vr; bt = tt; bl = kr; } }
        }
        pdown = down;
        tt++;
        down = tt < 64 ? ((tt & 1) == 0) : (E2 >= 0);
        if (i0 > nq - 1) break;
        if ((tt - 1) - (i0 + 63) > nt - 1) break;
        if (tt >= max_steps) break;
    }
    out[0] = tt; out[1] = bl >= 0 ? best : -(1 << 26); out[2] = bt; out[3] = bl;
    return 0;
}

// k_sw's check-free stretch (fzp_swb_core.h: sw_interior_safe), pure geometry: from the band position (t, i0) on an nq x nt matrix, walk EVERY sequence of `steps` moves
// (the band's cells only depend on how many of them were DOWN) and report whether any cell of any step lies on the last row or the last column -- or outside the matrix.
// which = 0: the formula the kernel uses; 1: the r3 form without the "- 1" (the test's negative control).  Returns -1 when the band is not inside yet.
extern "C" int sw_safe_probe(int32_t t, int32_t i0, int32_t nq, int32_t nt, int which, int32_t *safe_out) {
    int32_t safe = swb::sw_interior_safe(t, i0, nq, nt);
    if (which == 1 && safe > 0) safe += 1;
    if (which == 1 && safe == 0 && t >= 64 && i0 >= 0 && (t - 1) - (i0 + 63) >= 0) {
        const int32_t rl = nq - 1 - (i0 + 63), cl = nt - 1 - ((t - 1) - i0);
        safe = rl < cl ? rl : cl;
    }
    *safe_out = safe;
    if (safe <= 0) return -1;
    for (int32_t s = 1; s <= safe; s++)            // after s steps ...
        for (int32_t d = 0; d <= s; d++) {         // ... d of them DOWN: lane k holds cell (i0 + d + k, (t - 1 + s) - (i0 + d + k))
            const int32_t row63 = i0 + d + 63, col0 = (t - 1 + s) - (i0 + d);
            if (row63 >= nq - 1 || col0 >= nt - 1) return 1;      // lane 63 on (or past) the last row, or lane 0 on (or past) the last column
        }
    return 0;
}
