// fzp_pk.h -- reading K1's packed records (fzp_batch.h: PkSrc): the view and the small helpers K2's packed kernels (fzp_hetcall.hip) and K6's packed tally (fzp_cns.hip) share
#pragma once
#include "fzp_batch.h"

struct PkView {
    const int32_t *rec_pos, *rec_qid, *rec_ctg;
    const int64_t *rec_read;
    PkSrc s;
    const int64_t *ctg_goff;
    const int32_t *ctg_limit;
    int64_t n_rec;
};
constexpr uint32_t PK_EVEN = 0x55555555u;
__device__ __forceinline__ uint32_t pk_valid(int32_t wi, int32_t L) {       // one (even) bit per op of word wi that belongs to the stream
    const int32_t nv = L - 16 * wi;
    return nv >= 16 ? PK_EVEN : (nv <= 0 ? 0u : (((1u << (2 * nv)) - 1u) & PK_EVEN));
}
// the 16 read bases ending at base i (i >= 0), base i in bits 30..31
__device__ __forceinline__ uint32_t pk_bases16(const uint32_t *__restrict__ pk, int32_t i) {
    const int32_t w1 = i >> 4;
    const uint64_t two = ((uint64_t)pk[w1] << 32) | (w1 > 0 ? pk[w1 - 1] : 0u);
    return (uint32_t)(two >> (2 * (i & 15) + 2));
}
// largest checkpoint k in [0, nck) whose contig consumption is <= want (checkpoint 0 consumes nothing)
__device__ __forceinline__ int32_t pk_ck_search(const int2 *__restrict__ ck, int32_t nck, int32_t want) {
    int32_t a = 0, b = nck;
    while (b - a > 1) { const int32_t m = (a + b) >> 1; if (ck[m].y <= want) a = m; else b = m; }
    return a;
}

inline PkView pk_view(const fzp_batch *b) {
    PkView v;
    v.rec_pos = b->rec_pos.p; v.rec_qid = b->rec_qid.p; v.rec_ctg = b->rec_ctg.p; v.rec_read = b->rec_read.p;
    v.s = b->pk;
    v.ctg_goff = b->ctg_goff.p; v.ctg_limit = b->ctg_limit.p; v.n_rec = b->n_rec;
    return v;
}
