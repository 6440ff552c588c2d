// fzp_swb_core.h -- the cell function of the bit-sliced banded DP (k_swb in fzp_align.hip): one READ per lane, the 64 cells of its band's
// anti-diagonal in the 64 bits of a word, the DP values as DIFFERENCES in three bit planes.  Plain integer code: the kernel runs it per lane,
// tests/swb_core_check.cpp runs it on the host against the scalar twin (same masks, same moves, same terminal).
//
// Cost form of the recurrence.  With match +2, mismatch -4, gap -3 a path to (i, j) scores S = (i + j + 2) - 2 D, D = 3 x + 2 g (x mismatches,
// g gap bases): maximising S is minimising D, a cell adds c = 0 (match) or 3 to its diagonal predecessor or 2 to the cell above / to the left, and
// neighbouring cells differ by -2..2.  A cell keeps
//      Pv = 2 - (D(i,j) - D(i,j-1))   (its horizontal difference)        Qv = 2 - (D(i,j) - D(i-1,j))   (its vertical difference)       both in 0..4
// and takes p = Pv of the cell ABOVE it and q = Qv of the cell to its LEFT (both on the previous anti-diagonal):
//      M = max(p, q, e)   e = 4 on a match, 1 on a mismatch, 0 where the diagonal predecessor lies outside the band ("forbid")
//      Pv = M - q,  Qv = M - p
//      D bit (diagonal chosen, ties included) = match | (mismatch & p <= 1 & q <= 1), never under forbid
//      p >= q  <=>  the cell above is at least as good as the one to the left
// A neighbour outside the band enters as p = 0 / q = 0 ("two worse than the diagonal": never chosen).  The function is closed on 0..4, so the planes never
// overflow whatever the band edges do.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define FZP_HD __host__ __device__ __forceinline__
#else
#define FZP_HD inline
#endif

namespace swb {

// three-input boolean function by truth table: bit (a << 2 | b << 1 | c) of TT (the v_bitop3_b32 convention: evaluate the formula on TA, TB, TC)
constexpr uint8_t TA = 0xF0, TB = 0xCC, TC = 0xAA;
template <uint8_t TT>
FZP_HD uint32_t lut3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
#else
    uint32_t r = 0;
    for (int m = 0; m < 8; m++)
        if ((TT >> m) & 1) r |= ((m & 4) ? a : ~a) & ((m & 2) ? b : ~b) & ((m & 1) ? c : ~c);
    return r;
#endif
}
template <uint8_t TT>
FZP_HD uint64_t lut3(uint64_t a, uint64_t b, uint64_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, TT);
    const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), TT);
    return ((uint64_t)hi << 32) | lo;
#else
    uint64_t r = 0;
    for (int m = 0; m < 8; m++)
        if ((TT >> m) & 1) r |= ((m & 4) ? a : ~a) & ((m & 2) ? b : ~b) & ((m & 1) ? c : ~c);
    return r;
#endif
}

template <class W> struct PlanesT { W v0, v1, v2; };      // a value 0..4 per bit position: v0 + 2 v1 + 4 v2
using Planes = PlanesT<uint64_t>;                        // 64 cells of a band in one lane (k_swb)
using Planes32 = PlanesT<uint32_t>;                      // half a band per lane (k_swb2)

// a - b on planes for a >= b (no borrow out of the top)
template <class W>
FZP_HD PlanesT<W> minus(const PlanesT<W> a, const PlanesT<W> b) {
    constexpr uint8_t BORROW = (uint8_t)((~TA & TB) | (~(TA ^ TB) & TC));       // borrow out of a - b - c
    constexpr uint8_t XOR3 = (uint8_t)(TA ^ TB ^ TC);
    const W br0 = lut3<(uint8_t)(~TA & TB)>(a.v0, b.v0, (W)0);
    const W br1 = lut3<BORROW>(a.v1, b.v1, br0);
    PlanesT<W> r;
    r.v0 = a.v0 ^ b.v0;
    r.v1 = lut3<XOR3>(a.v1, b.v1, br0);
    r.v2 = lut3<XOR3>(a.v2, b.v2, br1);
    return r;
}

// One anti-diagonal of cells.  xm: mismatch bits; f: forbid bits (at most the band-edge lane); dn: all ones when the step moved DOWN, else zero.
// p, q: the neighbours' planes lined up with the cells (already shifted).  Returns the cells' planes and the two trace-back masks of the step:
// D = diagonal chosen, G = "the gap comes from the same lane of the previous step" (the cell above after DOWN, the one to the left after RIGHT;
// defined where D = 0, the only place the walk looks at it).  26 functions of up to three inputs: max(p, q) by a borrow chain and three selects, the step's e folded
// into its planes, two borrow-free subtractions; D needs only "max(p, q) <= 1"; and where the diagonal lost, q >= p  <=>  Pv = 0, p >= q  <=>  Qv = 0.
template <class W>
FZP_HD void cells(const W xm, const W f, const W dn, const PlanesT<W> p, const PlanesT<W> q, PlanesT<W> *Pv, PlanesT<W> *Qv, W *D, W *G) {
    constexpr uint8_t BORROW = (uint8_t)((~TA & TB) | (~(TA ^ TB) & TC));
    constexpr uint8_t SEL = (uint8_t)((TA & TB) | (~TA & TC));                 // a ? b : c
    const W b0 = lut3<(uint8_t)(~TA & TB)>(p.v0, q.v0, (W)0);
    const W b1 = lut3<BORROW>(p.v1, q.v1, b0);
    const W lt = lut3<BORROW>(p.v2, q.v2, b1);       // p < q
    const W m0 = lut3<SEL>(lt, q.v0, p.v0), m1 = lut3<SEL>(lt, q.v1, p.v1), m2 = lut3<SEL>(lt, q.v2, p.v2);      // max(p, q)
    const W t = m1 | m2;                             // max(p, q) >= 2
    PlanesT<W> M;                                    // max(p, q, e):  match -> 4;  mismatch -> at least 1;  forbid -> max(p, q).  "Not a match" for the value logic: xm | f
    M.v2 = lut3<(uint8_t)(TA | ~(TB | TC))>(m2, xm, f);
    M.v1 = lut3<(uint8_t)(TA & (TB | TC))>(m1, xm, f);
    M.v0 = lut3<(uint8_t)(TA & (TB | TC))>(m0, xm, f) | lut3<(uint8_t)(TA & ~TB & ~TC)>(xm, f, t);      // ... | a plain mismatch (e = 1) where max(p, q) < 2
    const PlanesT<W> P = minus(M, q), Q = minus(M, p);
    *Pv = P; *Qv = Q;
    *D = lut3<(uint8_t)(~TA & (~TB | ~TC))>(f, xm, t);                          // ~f & (match | max(p, q) <= 1)
    const W nzP = lut3<(uint8_t)(TA | TB | TC)>(P.v0, P.v1, P.v2), nzQ = lut3<(uint8_t)(TA | TB | TC)>(Q.v0, Q.v1, Q.v2);
    *G = lut3<(uint8_t)((TA & ~TB) | (~TA & ~TC))>(dn, nzQ, nzP);               // DOWN: p >= q (above is the same lane);  RIGHT: q >= p
}

// value 0..4 at bit position k of planes
template <class W>
FZP_HD int32_t value_at(const PlanesT<W> v, int k) { return (int32_t)(((v.v0 >> k) & 1) | (((v.v1 >> k) & 1) << 1) | (((v.v2 >> k) & 1) << 2)); }

// ---- half a band per lane (k_swb2): lanes 2r and 2r+1 share a read.  The LOW lane holds cells 0..31 with cell c at bit c, the HIGH lane cells 32..63 MIRRORED
// (cell c at bit 63 - c) and TRANSPOSED (P and Q swap roles, DOWN and RIGHT swap roles): the recurrence is symmetric under that, so both lanes run the same
// code on "my move" (RIGHT for the low lane: a contig base enters at cell 0; DOWN for the high lane: a read base enters at cell 63 -- bit 0 in both) and the
// "other move".  Planes: A shifts on my move (low: P, high: Q), B on the other one (low: Q, high: P); windows: Wm = the sequence that enters in this lane, Wo = the
// other.  What crosses the middle of the band -- the bit a left shift pushes out of A / Wm at bit 31 -- enters the partner's B / Wo at ITS bit 31.
struct Half {
    Planes32 A, B;
    uint32_t Wm0, Wm1, Wo0, Wo1;
};
struct HalfOut { uint32_t a0, a1, a2, w0, w1; };      // bit 31 of A's planes and of Wm's, as 0 / 1: what the partner takes in on its other move

FZP_HD HalfOut half_out(const Half &h) { return HalfOut{h.A.v0 >> 31, h.A.v1 >> 31, h.A.v2 >> 31, h.Wm0 >> 31, h.Wm1 >> 31}; }

// my = 1: this step is my move (the partner's other move); base = the 2-bit code entering (my move only); in = the partner's half_out before the step;
// f = 1: two moves of my kind in a row (the edge cell's diagonal predecessor is outside the band); bad = cells whose bases lie past an end (tail of the extension).
// Returns the step's masks for this half (bit = this lane's bit order) and leaves the new planes in h.
FZP_HD void half_step(Half &h, const uint32_t my, const uint32_t base, const HalfOut in, const uint32_t f, const uint32_t bad, uint32_t *D, uint32_t *G) {
    const uint32_t ot = 1u - my;
    h.A.v0 <<= my; h.A.v1 <<= my; h.A.v2 <<= my;
    h.Wm0 = (h.Wm0 << my) | (base & 1u); h.Wm1 = (h.Wm1 << my) | (base >> 1);
    h.B.v0 = (h.B.v0 >> ot) | ((in.a0 & ot) << 31); h.B.v1 = (h.B.v1 >> ot) | ((in.a1 & ot) << 31); h.B.v2 = (h.B.v2 >> ot) | ((in.a2 & ot) << 31);
    h.Wo0 = (h.Wo0 >> ot) | ((in.w0 & ot) << 31); h.Wo1 = (h.Wo1 >> ot) | ((in.w1 & ot) << 31);
    const uint32_t xm = lut3<(uint8_t)((TA ^ TB) | TC)>(h.Wm0, h.Wo0, h.Wm1 ^ h.Wo1) | bad;
    Planes32 nA, nB;
    cells<uint32_t>(xm, f, 0u - ot, h.A, h.B, &nA, &nB, D, G);
    h.A = nA; h.B = nB;
}
// the band-edge cell of this half (cell 0 / cell 63: bit 0 either way) moved along my plane on my move, along the other one else: its difference code
FZP_HD int32_t half_edge(const Half &h, const uint32_t my) {
    const uint32_t m = 0u - my;
    const uint32_t x0 = (h.A.v0 & m) | (h.B.v0 & ~m), x1 = (h.A.v1 & m) | (h.B.v1 & ~m), x2 = (h.A.v2 & m) | (h.B.v2 & ~m);
    return (int32_t)((x0 & 1u) | ((x1 & 1u) << 1) | ((x2 & 1u) << 2));
}

// ---- the wave-per-piece kernel's check-free stretch (k_sw's interior blocks; on the host: tests/test_swb_core.py::test_interior_block_never_touches_a_border).
// Before step t the band's lane k holds cell (i0 + k, (t - 1) - (i0 + k)); a step moves the band DOWN (i0 + 1) or RIGHT.  How many steps can run with NO cell of the band
// on the matrix's last row or last column -- the terminal candidates, which only the checked steps look at -- whatever the moves?  A DOWN brings lane 63 one row nearer
// the last row, a RIGHT lane 0 one column nearer the last column: one step less than the smaller of the two distances.  (Without the "- 1" -- r3 -- the block's last step
// could put lane 63 ON the last row unseen: ADVICE r3; pinned by the host test, which that form fails.)  0: the band is not yet inside the matrix.
FZP_HD int32_t sw_interior_safe(int32_t t, int32_t i0, int32_t nq, int32_t nt, int32_t band = 64) {      // band: the band's cells (v1.8: 64 or 32)
    if (!(t >= band && i0 >= 0 && (t - 1) - (i0 + band - 1) >= 0)) return 0;
    const int32_t rows_left = nq - 1 - (i0 + band - 1), cols_left = nt - 1 - ((t - 1) - i0);
    return (rows_left < cols_left ? rows_left : cols_left) - 1;
}
}   // namespace swb
