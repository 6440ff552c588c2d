// Host build of the bit-sliced DP (falcon_unzip_amd/csrc/fzp_swb_core.h) with the loop around the cell function written the way k_swb runs it per
// lane: planes, windows, forbid bits, steering by two tracked edge scores, terminal by accumulated differences along the border.  Test infrastructure
// (tests/test_swb_core.py compares it with the scalar twin's orc_dp_extend_raw); built by the test with g++.
#include <cstdint>
#include <cstring>
#include "../falcon_unzip_amd/csrc/fzp_swb_core.h"

extern "C" int swb_extend_host(const uint8_t *q, int64_t nq, const uint8_t *t, int64_t nt, uint64_t *tbD, uint64_t *tbG, uint8_t *mv, int64_t *out) {
    using namespace swb;
    if (nq < 64 || nt < 64) return -1;                        // (the kernel leaves such extensions to k_sw)
    const int64_t max_steps = nq + nt + 2;
    // step -1: the anti-diagonal i + j = -1 of the virtual border, lane k = cell (k - 33, 32 - k):  Pv = 0 where j >= 0 (k <= 32) else 4, Qv = 0 where i >= 0 (k >= 33) else 4
    Planes P = {0, 0, ~0ull << 33}, Q = {0, 0, (1ull << 33) - 1};
    uint64_t R0 = 0, R1 = 0, C0 = 0, C1 = 0;                   // windows: bit k = read base i0 + k / contig base t - i0 - k (bit planes of the 2-bit codes)
    for (int k = 0; k < 64; k++) {
        const int64_t i = k - 33, j = 32 - k;
        if (i >= 0 && i < nq) { R0 |= (uint64_t)(q[i] & 1) << k; R1 |= (uint64_t)(q[i] >> 1) << k; }
        if (j >= 0 && j < nt) { C0 |= (uint64_t)(t[j] & 1) << k; C1 |= (uint64_t)(t[j] >> 1) << k; }
    }
    int64_t i0 = -33, tt = 0;
    int32_t S0 = -259, E2 = 8;                                 // score of lane 0's cell; (score of lane 63's - S0) / 2 -- at step -1 from the border's closed form
    bool down = true, pdown = false;
    bool row_on = false, col_on = false;
    int32_t Hrow = 0, Hcol = 0, best = -(1 << 26);
    int64_t bt = -1, bl = -1;
    for (;;) {
        Planes p, qq;
        if (down) {
            i0++;
            R0 >>= 1; R1 >>= 1;
            const int64_t i = i0 + 63;
            if (i >= 0 && i < nq) { R0 |= (uint64_t)(q[i] & 1) << 63; R1 |= (uint64_t)(q[i] >> 1) << 63; }
            p = P; qq = {Q.v0 >> 1, Q.v1 >> 1, Q.v2 >> 1};
        } else {
            C0 <<= 1; C1 <<= 1;
            const int64_t j = tt - i0;
            if (j >= 0 && j < nt) { C0 |= (uint64_t)(t[j] & 1); C1 |= (uint64_t)(t[j] >> 1); }
            p = {P.v0 << 1, P.v1 << 1, P.v2 << 1}; qq = Q;
        }
        uint64_t xm = (R0 ^ C0) | (R1 ^ C1);
        {   // bases past the read's / the window's end never match (tail of the extension only)
            const int64_t nv = nq - i0;                                                 // lanes k < nv hold read bases
            if (nv < 64) xm |= nv <= 0 ? ~0ull : ~((1ull << nv) - 1);
            const int64_t kmin = tt - i0 - nt + 1;                                      // lanes k >= kmin hold contig bases
            if (kmin > 0) xm |= kmin >= 64 ? ~0ull : (1ull << kmin) - 1;
        }
        const uint64_t f = (down && pdown) ? 1ull << 63 : ((!down && !pdown) ? 1ull : 0ull);
        uint64_t D, G;
        cells(xm, f, down ? ~0ull : 0ull, p, qq, &P, &Q, &D, &G);
        tbD[tt] = D; tbG[tt] = G; mv[tt] = down ? 1 : 0;
        // edge scores: every lane's cell moved down (vertical difference) or right (horizontal)
        const Planes &X = down ? Q : P;
        const int32_t v0 = value_at(X, 0), v63 = value_at(X, 63);
        S0 += 2 * v0 - 3;
        E2 += v63 - v0;
        // terminal: best valid cell of the last row / the last column, by differences along them
        {
            const int64_t kc = tt - (nt - 1) - i0, kr = nq - 1 - i0;
            if (!col_on && !down && kc == 0) { col_on = true; Hcol = S0; }
            else if (col_on && kc >= 0 && kc <= 63) Hcol += 2 * value_at(Q, (int)kc) - 3;
            if (!row_on && down && kr == 63) { row_on = true; Hrow = S0 + 2 * E2; }
            else if (row_on && kr >= 0 && kr <= 63) Hrow += 2 * value_at(P, (int)kr) - 3;
            if (col_on && kc >= 0 && kc <= 63) { const int64_t i = i0 + kc; if (i >= 0 && i < nq && Hcol > best) { best = Hcol; bt = tt; bl = kc; } }
            if (row_on && kr >= 0 && kr <= 63) { const int64_t j = tt - (nq - 1); if (j >= 0 && j < nt && Hrow > best) { best = Hrow; bt = tt; bl = kr; } }
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
