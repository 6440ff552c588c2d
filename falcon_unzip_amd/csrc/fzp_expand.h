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
// 64 ops are loaded per step (lane = op) and their column / deleted / inserted lengths prefix-summed across the wave on the DPP
// network.  The columns of those ops are then dealt 64 at a time (lane = column).  What a column needs from its op is two
// numbers: how many deleted and how many inserted / clipped bases precede it inside the chunk -- its reference position is
// rp + column + deleted, its base sits at qp + column + inserted.  Both only grow from op to op, so every op drops the pair
// (16 bits each) into the slot of its FIRST column in a per-wave LDS window of EXP_WIN columns, and a running maximum over
// the window (DPP again) hands every column the pair of the op that covers it: one LDS read and six DPP steps per 64
// columns instead of a six-step search through ds_bpermute.  Chunks whose deleted or inserted total reaches 65 536 (a long
// soft clip, a long gap) take the search.
// visit(pos, sym) is called by every lane of a group; lanes that hold no column get pos = INT_MIN.
constexpr int EXP_WIN = 256;                  // columns per window: four groups of 64, four symbol loads in flight per wave
struct NoGapOps { __device__ __forceinline__ void operator()(uint32_t, int32_t, uint32_t, int64_t) const {} };
// gap_op(op, first reference position of the op (for I: the position AFTER the insertion point), length, index of the
// op's first query base in v.seq) is called by the lane that holds a D or I op.
// win: EXP_WIN words of LDS that belong to the calling wave (16-byte aligned).
template <class Visit, class GapOp = NoGapOps>
__device__ __forceinline__ void expand_record(const RecView &v, int64_t r, uint32_t *win, Visit &&visit, int64_t first_chunk = 0, int32_t rp0 = 0, int32_t qp0 = 0,
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
        const uint32_t len = w >> 4, t = w & 15u;
        const bool isM = (t == FZP_OP_M) | (t == FZP_OP_EQ) | (t == FZP_OP_X);
        const uint32_t cadv = isM ? len : 0u;
        const uint32_t dadv = (t == FZP_OP_D) ? len : 0u;
        const uint32_t iadv = ((t == FZP_OP_I) | (t == FZP_OP_S)) ? len : 0u;
        const uint32_t cs = wave_incl_scan_u32_dpp(cadv), ds = wave_incl_scan_u32_dpp(dadv), is = wave_incl_scan_u32_dpp(iadv);
        const uint32_t ctot = (uint32_t)__builtin_amdgcn_readlane((int)cs, 63), dtot = (uint32_t)__builtin_amdgcn_readlane((int)ds, 63),
                       itot = (uint32_t)__builtin_amdgcn_readlane((int)is, 63);
        const uint32_t cex = cs - cadv, dex = ds - dadv, iex = is - iadv;
        const bool packed = (dtot | itot) < 65536u;                                      // wave-uniform
        const uint32_t ab = (dex << 16) | (iex & 0xffffu);
        uint32_t carry = 0;
        for (uint32_t base = 0; base < ctot; base += EXP_WIN) {
            int32_t cpos[4];
            int64_t coff[4];
            bool cval[4];
            if (packed) {
                ((uint4 *)win)[lane] = make_uint4(0u, 0u, 0u, 0u);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                if (cadv != 0u && cex - base < (uint32_t)EXP_WIN) win[cex - base] = ab;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                cval[u] = false; cpos[u] = 0; coff[u] = 0;
                if (base + u * 64 >= ctot) continue;   // wave-uniform
                const uint32_t tc = base + u * 64 + lane;
                uint32_t dd, ii;
                if (packed) {
                    uint32_t x = wave_incl_maxscan_u32_dpp(win[u * 64 + lane]);
                    x = max(x, carry);
                    carry = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
                    dd = x >> 16; ii = x & 0xffffu;
                } else {
                    int j = 0;   // smallest j with cs[j] > tc
#pragma unroll
                    for (int s = 32; s >= 1; s >>= 1) {
                        const uint32_t x = bcast_u32(cs, j + s - 1);
                        if (x <= tc) j += s;
                    }
                    j = min(j, 63);
                    dd = bcast_u32(dex, j); ii = bcast_u32(iex, j);
                }
                cval[u] = tc < ctot;
                cpos[u] = (int32_t)(rp + tc + dd);
                coff[u] = sbase + qp + tc + ii;
            }
            uint8_t csym[4];
#pragma unroll
            for (int u = 0; u < 4; u++) csym[u] = cval[u] ? v.seq[coff[u]] : (uint8_t)0;
#pragma unroll
            for (int u = 0; u < 4; u++) visit(cval[u] ? cpos[u] : (int32_t)0x80000000, csym[u]);   // lanes without a column see a position no tile holds
        }
        // deletions and insertions of this chunk, one op per lane (short runs; K6 tallies them, K2 ignores them)
        if ((t == FZP_OP_D || t == FZP_OP_I) && len > 0) gap_op(t, (int32_t)(rp + cex + dex), len, sbase + qp + cex + iex);
        rp += (int32_t)(ctot + dtot);
        qp += ctot + itot;
        w = w_next;
    }
}

}  // namespace
