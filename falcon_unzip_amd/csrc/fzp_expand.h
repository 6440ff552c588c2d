// fzp_expand.h -- wave-cooperative CIGAR expansion shared by K2 (pileup, variant_map) and K6 (consensus tally).
#pragma once
#include "fzp_batch.h"

namespace {
struct RecView {
    const int32_t *rec_pos, *rec_qid, *rec_ctg;
    const int64_t *cig_off, *seq_off;
    const uint32_t *cigar;
    const uint8_t *seq;
    const int64_t *ctg_goff;
    const int32_t *ctg_limit;
    int64_t n_rec;
};

// Wave-cooperative walk of one record's CIGAR (phasing.py:77-96): S,I advance the query; M,=,X emit
// one column per base and advance both; D advances the reference; N,H,P do nothing.
// 64 ops are loaded per step, their (ref, query, column) advances prefix-summed across the wave,
// then the columns of those ops are dealt 64 at a time: lane t finds its op by a 6-step search.
// visit(pos, sym) is called by every lane of a group; lanes that hold no column get pos = INT_MIN.
struct NoGapOps { __device__ __forceinline__ void operator()(uint32_t, int32_t, uint32_t, int64_t) const {} };
// gap_op(op, first reference position of the op (for I: the position AFTER the insertion point), length, index of the
// op's first query base in v.seq) is called by the lane that holds a D or I op.
template <class Visit, class GapOp = NoGapOps>
__device__ __forceinline__ void expand_record(const RecView &v, int64_t r, Visit &&visit, int64_t first_chunk = 0, int32_t rp0 = 0, int32_t qp0 = 0,
                                              int32_t stop_pos = 0x7fffffff, GapOp &&gap_op = GapOp()) {
    // first_chunk / rp0 / qp0: resume at a 64-op checkpoint (offsets relative to the record's POS / SEQ start);
    // stop_pos: no column at or beyond it is wanted (wave-uniform early exit)
    const int lane = lane_id();
    const int64_t c0 = v.cig_off[r] + first_chunk * 64, c1 = v.cig_off[r + 1];
    const int64_t sbase = v.seq_off[r];
    int32_t rp = v.rec_pos[r] + rp0;
    int64_t qp = qp0;
    uint32_t w = (c0 + lane < c1) ? v.cigar[c0 + lane] : 0u;
    for (int64_t cb = c0; cb < c1 && rp < stop_pos; cb += 64) {
        const uint32_t w_next = (cb + 64 + lane < c1) ? v.cigar[cb + 64 + lane] : 0u;   // in flight while this chunk is dealt
        uint32_t len = w >> 4, t = w & 15u;
        bool isM = (t == FZP_OP_M) | (t == FZP_OP_EQ) | (t == FZP_OP_X);
        uint32_t radv = (isM | (t == FZP_OP_D)) ? len : 0u;
        uint32_t qadv = (isM | (t == FZP_OP_I) | (t == FZP_OP_S)) ? len : 0u;
        uint32_t cadv = isM ? len : 0u;
        uint32_t rs = wave_incl_scan_u32(radv), qs = wave_incl_scan_u32(qadv), cs = wave_incl_scan_u32(cadv);
        uint32_t ctot = bcast_u32(cs, 63);
        uint32_t rex = rs - radv, qex = qs - qadv, cex = cs - cadv;
        // four groups of 64 columns per round so that four symbol loads are in flight per wave
        for (uint32_t base = 0; base < ctot; base += 256) {
            int32_t cpos[4];
            int64_t coff[4];
            bool cval[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                cval[u] = false; cpos[u] = 0; coff[u] = 0;
                if (base + u * 64 >= ctot) continue;   // wave-uniform
                uint32_t tc = base + u * 64 + lane;
                int j = 0;   // smallest j with cs[j] > tc
#pragma unroll
                for (int s = 32; s >= 1; s >>= 1) {
                    uint32_t x = bcast_u32(cs, j + s - 1);
                    if (x <= tc) j += s;
                }
                j = min(j, 63);
                uint32_t jc = bcast_u32(cex, j), jr = bcast_u32(rex, j), jq = bcast_u32(qex, j);
                uint32_t d = tc - jc;
                cval[u] = tc < ctot;
                cpos[u] = (int32_t)(rp + jr + d);
                coff[u] = sbase + qp + jq + d;
            }
            uint8_t csym[4];
#pragma unroll
            for (int u = 0; u < 4; u++) csym[u] = cval[u] ? v.seq[coff[u]] : (uint8_t)0;
#pragma unroll
            for (int u = 0; u < 4; u++) visit(cval[u] ? cpos[u] : (int32_t)0x80000000, csym[u]);   // lanes without a column see a position no tile holds
        }
        // deletions and insertions of this chunk, one op per lane (short runs; K6 tallies them, K2 ignores them)
        if ((t == FZP_OP_D || t == FZP_OP_I) && len > 0) gap_op(t, (int32_t)(rp + rex), len, sbase + qp + qex);
        rp += (int32_t)bcast_u32(rs, 63);
        qp += bcast_u32(qs, 63);
        w = w_next;
    }
}

}  // namespace
