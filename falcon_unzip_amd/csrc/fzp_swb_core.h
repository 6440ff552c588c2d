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
// overflow whatever the band edges do.  Equivalent forms used below: on a mismatch (Pv, Qv) = (p -. q, q -. p) (+1 each when p = q = 0), on a match
// (4 - q, 4 - p) -- both are a -. b ("monus") with a = (match ? 4 : p).
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

struct Planes { uint64_t v0, v1, v2; };      // a value 0..4 per bit position: v0 + 2 v1 + 4 v2

// a -. b = max(a - b, 0) on planes (a, b in 0..4); *borrow = (a < b)
FZP_HD Planes monus(const Planes a, const Planes b, uint64_t *borrow) {
    constexpr uint8_t BORROW = (uint8_t)((~TA & TB) | (~(TA ^ TB) & TC));       // borrow out of a - b - c
    constexpr uint8_t XOR3 = (uint8_t)(TA ^ TB ^ TC);
    const uint64_t br0 = ~a.v0 & b.v0;
    const uint64_t br1 = lut3<BORROW>(a.v1, b.v1, br0);
    const uint64_t B = lut3<BORROW>(a.v2, b.v2, br1);
    Planes r;
    r.v0 = lut3<(uint8_t)((TA ^ TB) & ~TC)>(a.v0, b.v0, B);
    r.v1 = lut3<XOR3>(a.v1, b.v1, br0) & ~B;
    r.v2 = lut3<XOR3>(a.v2, b.v2, br1) & ~B;
    *borrow = B;
    return r;
}

// One anti-diagonal of cells.  xm: mismatch bits; f: forbid bits (at most the band-edge lane); dn: all ones when the step moved DOWN, else zero.
// p, q: the neighbours' planes lined up with the cells (already shifted).  Returns the cells' planes and the two trace-back masks of the step:
// D = diagonal chosen, G = "the gap comes from the same lane of the previous step" (the cell above after DOWN, the one to the left after RIGHT;
// defined where D = 0, the only place the walk looks at it).
FZP_HD void cells(const uint64_t xm, const uint64_t f, const uint64_t dn, const Planes p, const Planes q, Planes *Pv, Planes *Qv, uint64_t *D, uint64_t *G) {
    const uint64_t x = xm | f;                       // "not a match" for the value logic
    Planes ph, qh;                                   // match ? 4 : p
    ph.v0 = p.v0 & x; ph.v1 = p.v1 & x; ph.v2 = p.v2 | ~x;
    qh.v0 = q.v0 & x; qh.v1 = q.v1 & x; qh.v2 = q.v2 | ~x;
    uint64_t b_pq, b_qp;
    Planes P = monus(ph, q, &b_pq), Q = monus(qh, p, &b_qp);
    const uint64_t tp = p.v1 | p.v2, tq = q.v1 | q.v2;
    const uint64_t le1 = ~(tp | tq);                                                 // p <= 1 and q <= 1
    const uint64_t z = lut3<(uint8_t)(TA & ~TB & ~TC)>(le1, p.v0, q.v0) & xm & ~f;   // mismatch with p = q = 0: M = 1
    P.v0 |= z; Q.v0 |= z;
    *Pv = P; *Qv = Q;
    *D = lut3<(uint8_t)(~TA & (~TB | TC))>(f, xm, le1);                              // ~f & (match | le1)
    *G = lut3<(uint8_t)((TA & ~TB) | (~TA & ~TC))>(dn, b_pq, b_qp);                  // DOWN: p >= q (above is the same lane);  RIGHT: q >= p
}

// value 0..4 at bit position k of planes
FZP_HD int32_t value_at(const Planes v, int k) { return (int32_t)(((v.v0 >> k) & 1) | (((v.v1 >> k) & 1) << 1) | (((v.v2 >> k) & 1) << 2)); }

}   // namespace swb
