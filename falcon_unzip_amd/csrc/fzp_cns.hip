// fzp_cns.hip -- K6: phased-pile consensus of every (block, phase) of a batch ("fzcns v1"; twin: oracle/cns_oracle.c).
//
// BASELINE north_star config 4 asks for a haplotig consensus kernel on the phased read piles.  The reference has no
// consensus code of its own (falcon_sense lives in falcon_kit, Arrow in `variantCaller`, run_quiver.py:82-97), so this
// is this repo's own definition; the records, phases and blocks it consumes are the K1..K5 results already in HBM.
//   k_cns_extent   block spans [min site, max site] from K4's per-site block ids
//   k_cns_nrec     records per (block, phase) pile
//   k_cns_tiles    a workgroup owns 512 consecutive positions of one block: both phases' counters (10 x u32 per
//                  position and phase) live in LDS, its waves walk the records that overlap the tile -- phase looked
//                  up in K5's rows, CIGAR resumed at K2's 64-op checkpoint before the tile, columns dealt 64 at a time
//                  by the shared expander, D / I ops one per lane -- and the tile is written once, coalesced
//   k_cns_call     per position: 0..2 output bases (deletion / majority base / majority inserted base)
//   scan + k_cns_emit   sequences laid out per (block, phase), order fixed by the scan
// HBM-bound integer work: 1 B symbol + 4 B/op in, 40 B of counters per (position, phase) touched by atomics, 1 B out.
#include <algorithm>
#include "fzp_expand.h"

namespace {
constexpr int CN = 10;   // counters per (position, phase): A C G T del ins insA insC insG insT

__global__ void k_cns_nblk(int n_ctg, const int64_t *__restrict__ pvar_begin, const fzp_pvar *__restrict__ pvars, int32_t *__restrict__ nblk) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_ctg) return;
    nblk[c] = pvar_begin[c + 1] > pvar_begin[c] ? pvars[pvar_begin[c + 1] - 1].block : 0;
}
__global__ void __launch_bounds__(256) k_cns_extent(int64_t n_pvars, const fzp_pvar *__restrict__ pvars, const fzp_site *__restrict__ sites, const int32_t *__restrict__ site_ctg,
                                                    const int32_t *__restrict__ blk_base, int32_t *__restrict__ lo, int32_t *__restrict__ hi) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pvars) return;
    const fzp_pvar v = pvars[i];
    const int32_t g = blk_base[site_ctg[v.site]] + v.block - 1;
    const int32_t pos = sites[v.site].pos;
    atomicMin(&lo[g], pos);
    atomicMax(&hi[g], pos);
}

struct CnsView {
    const int32_t *rec_pos, *rec_qid, *rec_ctg, *rec_span;
    const int64_t *cig_off, *seq_off, *ck_off;
    const int32_t *ck_ref, *ck_q;
    const uint32_t *cigar;
    const uint8_t *seq;
    const fzp_pread *preads;
    const int64_t *pread_begin;
    const int32_t *blk_base, *lo, *hi;
    const int64_t *cnt_off;          // per block: first counter position (both phases: 2 * len slots)
    int64_t n_rec;
};

// records per pile (the header's n_records, and "is there a pile at all")
__global__ void __launch_bounds__(256) k_cns_nrec(CnsView v, uint32_t *__restrict__ n_records) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= v.n_rec) return;
    const int c = v.rec_ctg[r];
    const int32_t q = v.rec_qid[r], pos0 = v.rec_pos[r], span = v.rec_span[r];
    int64_t a = v.pread_begin[c], b = v.pread_begin[c + 1];
    const int64_t pe = b;
    while (a < b) { const int64_t m = (a + b) >> 1; if (v.preads[m].q_id < q) a = m + 1; else b = m; }
    for (int64_t e = a; e < pe && v.preads[e].q_id == q; e++) {
        const int32_t g = v.blk_base[c] + v.preads[e].block - 1;
        if (pos0 > v.hi[g] || pos0 + span <= v.lo[g]) continue;
        atomicAdd(&n_records[2 * g + v.preads[e].phase], 1u);
    }
}

constexpr int CNS_TILE = 512, CNS_THREADS = 512;
__global__ void __launch_bounds__(CNS_THREADS) k_cns_tiles(RecView rv, CnsView v, const int32_t *__restrict__ tile_blk, const int32_t *__restrict__ tile_start,
                                                           const int32_t *__restrict__ blk_ctg, const int64_t *__restrict__ ctg_rec_begin,
                                                           const int32_t *__restrict__ ctg_maxspan, uint32_t *__restrict__ cnt) {
    __shared__ uint32_t l_cnt[2 * CN * CNS_TILE];      // [phase][counter][position]: consecutive lanes -> distinct banks
    const int32_t g = tile_blk[blockIdx.x], ts = tile_start[blockIdx.x];
    const int c = blk_ctg[g];
    const int32_t lo = v.lo[g], hi = v.hi[g];
    const int32_t te = min(ts + CNS_TILE, hi + 1);      // exclusive
    for (int i = threadIdx.x; i < 2 * CN * CNS_TILE; i += CNS_THREADS) l_cnt[i] = 0;
    __syncthreads();
    // records of this contig that can overlap [ts, te): POS < te and POS > ts - max_span
    const int64_t rb = ctg_rec_begin[c], re = ctg_rec_begin[c + 1];
    const int32_t ms = ctg_maxspan[c];
    int64_t first, last;
    {
        int64_t a = rb, b = re;
        while (a < b) { const int64_t m = (a + b) >> 1; if (v.rec_pos[m] <= ts - ms) a = m + 1; else b = m; }
        first = a;
        b = re;
        while (a < b) { const int64_t m = (a + b) >> 1; if (v.rec_pos[m] < te) a = m + 1; else b = m; }
        last = a;
    }
    const int wave = threadIdx.x >> 6, lane = lane_id();
    constexpr int NW = CNS_THREADS / 64;
    const int32_t blk_id = g - v.blk_base[c] + 1;
    for (int64_t i0 = 0; first + i0 * NW + wave < last; i0 += 64) {
        // lane i prepares candidate first + (i0 + i) * NW + wave: overlap test, phase look-up, checkpoint search
        const int64_t r = first + (i0 + lane) * NW + wave;
        bool ok = r < last;
        int32_t ca = 0, cr = 0, cq = 0, ph = 0;
        if (ok) {
            const int32_t pos0 = v.rec_pos[r];
            ok = pos0 + v.rec_span[r] > ts;
            if (ok) {
                const int32_t q = v.rec_qid[r];
                int64_t a = v.pread_begin[c], b = v.pread_begin[c + 1];
                while (a < b) {      // first row with (q_id, block) >= (q, blk_id)
                    const int64_t m = (a + b) >> 1;
                    const fzp_pread pr = v.preads[m];
                    if (pr.q_id < q || (pr.q_id == q && pr.block < blk_id)) a = m + 1; else b = m;
                }
                ok = a < v.pread_begin[c + 1] && v.preads[a].q_id == q && v.preads[a].block == blk_id;
                if (ok) {
                    ph = v.preads[a].phase;
                    const int64_t k0 = v.ck_off[r];
                    int32_t cb = (int32_t)(v.ck_off[r + 1] - k0);
                    const int32_t want = ts - pos0;
                    while (cb - ca > 1) { const int32_t m = (ca + cb) >> 1; if (v.ck_ref[k0 + m] <= want) ca = m; else cb = m; }
                    cr = v.ck_ref[k0 + ca]; cq = v.ck_q[k0 + ca];
                }
            }
        }
        const int32_t rel = (int32_t)(r - first);
        for (uint64_t todo = __ballot(ok); todo; todo &= todo - 1) {
            const int l = __builtin_ctzll(todo);
            const int64_t ru = first + __builtin_amdgcn_readlane(rel, l);
            uint32_t *lc = l_cnt + __builtin_amdgcn_readlane(ph, l) * (CN * CNS_TILE);
            const int32_t pos0 = rv.rec_pos[ru];
            expand_record(rv, ru,
                [&](int32_t pos, uint8_t sym) {
                    const uint32_t p = (uint32_t)(pos - ts);
                    const int code = sym_code(sym);
                    if ((p < (uint32_t)(te - ts)) & (code < 4)) atomicAdd(&lc[code * CNS_TILE + p], 1u);
                },
                __builtin_amdgcn_readlane(ca, l), __builtin_amdgcn_readlane(cr, l), __builtin_amdgcn_readlane(cq, l), te + 1,
                [&](uint32_t op, int32_t r0, uint32_t n, int64_t qidx) {
                    if (op == FZP_OP_D) {
                        for (uint32_t d = 0; d < n; d++) {
                            const int32_t p = r0 + (int32_t)d;
                            if (p >= ts && p < te) atomicAdd(&lc[4 * CNS_TILE + (p - ts)], 1u);
                        }
                    } else {
                        const int32_t p = r0 - 1;                    // the insertion follows position p
                        if (p >= pos0 && p >= ts && p < te) {
                            atomicAdd(&lc[5 * CNS_TILE + (p - ts)], 1u);
                            const int code = sym_code(rv.seq[qidx]);
                            if (code < 4) atomicAdd(&lc[(6 + code) * CNS_TILE + (p - ts)], 1u);
                        }
                    }
                });
        }
    }
    __syncthreads();
    const int64_t len = (int64_t)hi - lo + 1;
    const int np = te - ts;
    for (int ph = 0; ph < 2; ph++) {
        uint32_t *dst = cnt + (2 * v.cnt_off[g] + (int64_t)ph * len + (ts - lo)) * CN;
        for (int i = threadIdx.x; i < np * CN; i += CNS_THREADS) dst[i] = l_cnt[ph * (CN * CNS_TILE) + (i % CN) * CNS_TILE + (i / CN)];
    }
}

// one thread per (block, phase, position): how many bases come out (0..2) and which
__global__ void __launch_bounds__(256) k_cns_call(int64_t n_slots, int n_blk, const int64_t *__restrict__ cnt_off, const int32_t *__restrict__ lo, const int32_t *__restrict__ hi,
                                                  const int32_t *__restrict__ blk_ctg, const int64_t *__restrict__ ctg_goff, const uint8_t *__restrict__ ref,
                                                  const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ n_records, uint32_t *__restrict__ n_out, uint8_t *__restrict__ sym2) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_slots) return;
    int a = 0, b = n_blk;                                   // block g with 2*cnt_off[g] <= i < 2*cnt_off[g+1]
    while (b - a > 1) { const int m = (a + b) >> 1; if (2 * cnt_off[m] <= i) a = m; else b = m; }
    const int g = a;
    const int64_t len = (int64_t)hi[g] - lo[g] + 1;
    const int64_t local = i - 2 * cnt_off[g];
    const int ph = local >= len ? 1 : 0;
    const int64_t x = local - (int64_t)ph * len;
    uint32_t no = 0;
    uint8_t s0 = 0, s1 = 0;
    if (n_records[2 * g + ph] > 0) {
        const uint32_t *c = cnt + i * CN;
        const uint32_t cov = c[0] + c[1] + c[2] + c[3] + c[4];
        uint8_t refb = ref[ctg_goff[blk_ctg[g]] + lo[g] + x];
        if (refb >= 'a' && refb <= 'z') refb -= 32;
        if (cov == 0) { s0 = refb; no = 1; }
        else {
            if (2 * c[4] <= cov) {
                uint32_t mx = max(max(c[0], c[1]), max(c[2], c[3]));
                const int rc = sym_code(refb);
                int pick = -1;
                if (rc < 4 && c[rc] == mx) pick = rc;
                for (int k = 0; k < 4 && pick < 0; k++) if (c[k] == mx) pick = k;
                s0 = code_sym(pick); no = 1;
            }
            if (2 * c[5] > cov) {
                uint32_t mx = c[6];
                int pick = 0;
                for (int k = 1; k < 4; k++) if (c[6 + k] > mx) { mx = c[6 + k]; pick = k; }
                if (mx > 0) { if (no) s1 = code_sym(pick); else s0 = code_sym(pick); no++; }
            }
        }
    }
    n_out[i] = no;
    sym2[2 * i] = s0; sym2[2 * i + 1] = s1;
}
__global__ void __launch_bounds__(256) k_cns_emit(int64_t n_slots, const uint32_t *__restrict__ n_out, const uint32_t *__restrict__ off, const uint8_t *__restrict__ sym2,
                                                  uint8_t *__restrict__ seq) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_slots) return;
    const uint32_t n = n_out[i], o = off[i];
    if (n > 0) seq[o] = sym2[2 * i];
    if (n > 1) seq[o + 1] = sym2[2 * i + 1];
}
// first output offset of every (block, phase): one gathered array instead of 2 * blocks tiny copies
__global__ void k_cns_first(int n_blk, const int64_t *__restrict__ cnt_off, const uint32_t *__restrict__ off, uint32_t total, uint32_t *__restrict__ first) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * n_blk) return;
    const int g = i >> 1, ph = i & 1;
    const int64_t len = cnt_off[g + 1] - cnt_off[g];
    first[i] = len > 0 ? off[2 * cnt_off[g] + (int64_t)ph * len] : total;
}
__global__ void k_fill32(int32_t *p, int64_t n, int32_t v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
inline unsigned nblocks(int64_t n, int per) { return (unsigned)std::max<int64_t>(1, (n + per - 1) / per); }
}  // namespace

extern "C" void fzp_tigs_free(fzp_tigs *t) {
    if (!t) return;
    free(t->tigs); free(t->seq);
    t->tigs = nullptr; t->seq = nullptr; t->n_tigs = 0; t->n_seq = 0;
}

extern "C" int fzp_batch_consensus(fzp_ctx *ctx, fzp_batch *b, fzp_tigs *out) {
    if (!ctx || !b || !out) { fzp_set_error("fzp_batch_consensus: bad arguments"); return FZP_EINVAL; }
    memset(out, 0, sizeof *out);
    if (!b->have_aln || !b->have_blocks || !b->have_preads || !b->have_sites) { fzp_set_error("fzp_batch_consensus: run FZP_STAGE_ALL on a batch with alignment records first"); return FZP_EINVAL; }
    FZP_TRY(fzp_bind(ctx));
    hipStream_t st = ctx->stream;
    const int nc = b->n_ctg;
    // ---- blocks per contig and their spans
    DevBuf<int32_t> d_nblk, d_base, d_lo, d_hi, d_bctg;
    FZP_TRY(d_nblk.alloc((size_t)nc));
    hipLaunchKernelGGL(k_cns_nblk, dim3(nblocks(nc, 64)), dim3(64), 0, st, nc, b->pvar_begin.p, b->pvars.p, d_nblk.p);
    std::vector<int32_t> nblk((size_t)nc), base((size_t)nc + 1, 0);
    FZP_TRY(d_nblk.download(nblk.data(), (size_t)nc, st));
    FZP_HIP(hipStreamSynchronize(st));
    for (int c = 0; c < nc; c++) base[(size_t)c + 1] = base[(size_t)c] + nblk[(size_t)c];
    const int NB = base[(size_t)nc];
    if (NB == 0 || b->n_pvars == 0) return FZP_OK;
    std::vector<int32_t> bctg((size_t)NB);
    for (int c = 0; c < nc; c++) for (int k = base[(size_t)c]; k < base[(size_t)c + 1]; k++) bctg[(size_t)k] = c;
    FZP_TRY(d_base.upload(base.data(), (size_t)nc + 1, st));
    FZP_TRY(d_bctg.upload(bctg.data(), (size_t)NB, st));
    FZP_TRY(d_lo.alloc((size_t)NB)); FZP_TRY(d_hi.alloc((size_t)NB));
    hipLaunchKernelGGL(k_fill32, dim3(nblocks(NB, 256)), dim3(256), 0, st, d_lo.p, (int64_t)NB, 0x7fffffff);
    hipLaunchKernelGGL(k_fill32, dim3(nblocks(NB, 256)), dim3(256), 0, st, d_hi.p, (int64_t)NB, -1);
    hipLaunchKernelGGL(k_cns_extent, dim3(nblocks(b->n_pvars, 256)), dim3(256), 0, st, b->n_pvars, b->pvars.p, b->sites.p, b->site_ctg.p, d_base.p, d_lo.p, d_hi.p);
    std::vector<int32_t> lo((size_t)NB), hi((size_t)NB);
    FZP_TRY(d_lo.download(lo.data(), (size_t)NB, st)); FZP_TRY(d_hi.download(hi.data(), (size_t)NB, st));
    FZP_HIP(hipStreamSynchronize(st));
    std::vector<int64_t> cnt_off((size_t)NB + 1, 0);
    for (int g = 0; g < NB; g++) cnt_off[(size_t)g + 1] = cnt_off[(size_t)g] + (hi[(size_t)g] >= lo[(size_t)g] ? (int64_t)hi[(size_t)g] - lo[(size_t)g] + 1 : 0);
    const int64_t n_slots = 2 * cnt_off[(size_t)NB];
    if (n_slots >= (1ll << 31)) { fzp_set_error("fzp_batch_consensus: %lld block positions (limit 2^30 per batch)", (long long)n_slots / 2); return FZP_EINVAL; }
    // ---- tally
    DevBuf<int64_t> d_cnt_off;
    DevBuf<uint32_t> cnt, n_records, n_out, off;
    DevBuf<uint8_t> sym2, seq;
    DevBuf<uint64_t> total;
    FZP_TRY(d_cnt_off.upload(cnt_off.data(), (size_t)NB + 1, st));
    FZP_TRY(cnt.alloc((size_t)n_slots * CN));
    FZP_TRY(n_records.alloc((size_t)NB * 2)); FZP_TRY(n_records.zero((size_t)NB * 2, st));
    CnsView v = {b->rec_pos.p, b->rec_qid.p, b->rec_ctg.p, b->rec_span.p, b->cig_off.p, b->seq_off.p, b->ck_off.p, b->ck_ref.p, b->ck_q.p, b->cigar.p, b->seq.p,
                 b->preads.p, b->pread_begin.p, d_base.p, d_lo.p, d_hi.p, d_cnt_off.p, b->n_rec};
    RecView rv = {b->rec_pos.p, b->rec_qid.p, b->rec_ctg.p, b->cig_off.p, b->seq_off.p, b->cigar.p, b->seq.p, b->ctg_goff.p, b->ctg_limit.p, b->n_rec};
    std::vector<int32_t> tblk, tstart;                    // tiles never span blocks
    for (int g = 0; g < NB; g++)
        for (int64_t t0 = lo[(size_t)g]; t0 <= hi[(size_t)g]; t0 += CNS_TILE) { tblk.push_back(g); tstart.push_back((int32_t)t0); }
    DevBuf<int32_t> d_tblk, d_tstart;
    FZP_TRY(d_tblk.upload(tblk.data(), tblk.size(), st)); FZP_TRY(d_tstart.upload(tstart.data(), tstart.size(), st));
    if (b->n_rec > 0) {
        ProfScope ps(ctx, "k6_tally");
        hipLaunchKernelGGL(k_cns_nrec, dim3(nblocks(b->n_rec, 256)), dim3(256), 0, st, v, n_records.p);
        if (!tblk.empty())
            hipLaunchKernelGGL(k_cns_tiles, dim3((unsigned)tblk.size()), dim3(CNS_THREADS), 0, st, rv, v, d_tblk.p, d_tstart.p, d_bctg.p, b->ctg_rec_begin.p, b->ctg_maxspan.p, cnt.p);
    } else {
        FZP_TRY(cnt.zero((size_t)n_slots * CN, st));
    }
    FZP_HIP(hipStreamSynchronize(st));                       // the tile vectors go out of scope
    // ---- call + layout
    FZP_TRY(n_out.alloc((size_t)n_slots)); FZP_TRY(off.alloc((size_t)n_slots)); FZP_TRY(sym2.alloc((size_t)n_slots * 2)); FZP_TRY(total.alloc(1));
    {
        ProfScope ps(ctx, "k6_call");
        hipLaunchKernelGGL(k_cns_call, dim3(nblocks(n_slots, 256)), dim3(256), 0, st, n_slots, NB, d_cnt_off.p, d_lo.p, d_hi.p, d_bctg.p, b->ctg_goff.p, b->ref.p, cnt.p, n_records.p,
                           n_out.p, sym2.p);
    }
    FZP_TRY(fzp_exclusive_scan_u32(ctx, n_out.p, off.p, (size_t)n_slots, total.p));
    uint64_t tot = 0;
    FZP_HIP(hipMemcpyAsync(&tot, total.p, 8, hipMemcpyDeviceToHost, st));
    FZP_HIP(hipStreamSynchronize(st));
    FZP_TRY(seq.alloc((size_t)tot));
    {
        ProfScope ps(ctx, "k6_emit");
        hipLaunchKernelGGL(k_cns_emit, dim3(nblocks(n_slots, 256)), dim3(256), 0, st, n_slots, n_out.p, off.p, sym2.p, seq.p);
    }
    // ---- results: sequence bytes, and per (block, phase) its offset = off[] at its first slot
    std::vector<uint32_t> h_nrec((size_t)NB * 2), h_first((size_t)NB * 2 + 1, (uint32_t)tot);
    FZP_TRY(n_records.download(h_nrec.data(), (size_t)NB * 2, st));
    DevBuf<uint32_t> d_first;
    FZP_TRY(d_first.alloc((size_t)NB * 2));
    hipLaunchKernelGGL(k_cns_first, dim3(nblocks(2 * NB, 256)), dim3(256), 0, st, NB, d_cnt_off.p, off.p, (uint32_t)tot, d_first.p);
    FZP_TRY(d_first.download(h_first.data(), (size_t)NB * 2, st));
    uint8_t *hseq = (uint8_t *)malloc((size_t)(tot ? tot : 1));
    if (!hseq) return FZP_ENOMEM;
    if (tot && hipMemcpyAsync(hseq, seq.p, (size_t)tot, hipMemcpyDeviceToHost, st) != hipSuccess) { free(hseq); fzp_set_error("consensus download failed"); return FZP_EDEVICE; }
    if (hipStreamSynchronize(st) != hipSuccess) { free(hseq); fzp_set_error("consensus download failed"); return FZP_EDEVICE; }
    // slot order == output order, so a tig ends where the next non-empty one starts
    std::vector<fzp_tig> tigs;
    std::vector<int64_t> starts;
    for (int g = 0; g < NB; g++)
        for (int ph = 0; ph < 2; ph++) {
            if (cnt_off[(size_t)g + 1] == cnt_off[(size_t)g]) continue;
            starts.push_back(h_first[(size_t)(2 * g + ph)]);
            if (!h_nrec[(size_t)(2 * g + ph)]) continue;
            fzp_tig t;
            memset(&t, 0, sizeof t);
            t.ctg = bctg[(size_t)g]; t.block = g - base[(size_t)t.ctg] + 1; t.phase = ph; t.lo = lo[(size_t)g]; t.hi = hi[(size_t)g];
            t.n_records = (int32_t)h_nrec[(size_t)(2 * g + ph)];
            t.seq_off = h_first[(size_t)(2 * g + ph)];
            t.seq_len = -(int64_t)starts.size();          // patched below: index of this tig's start in `starts`
            tigs.push_back(t);
        }
    starts.push_back((int64_t)tot);
    for (auto &t : tigs) { const size_t k = (size_t)(-t.seq_len) - 1; t.seq_len = starts[k + 1] - starts[k]; }
    out->n_tigs = (int64_t)tigs.size();
    out->tigs = (fzp_tig *)malloc((tigs.size() ? tigs.size() : 1) * sizeof(fzp_tig));
    if (!out->tigs) { free(hseq); return FZP_ENOMEM; }
    if (!tigs.empty()) memcpy(out->tigs, tigs.data(), tigs.size() * sizeof(fzp_tig));
    out->seq = hseq; out->n_seq = (int64_t)tot;
    FZP_HIP(hipGetLastError());
    return FZP_OK;
}

extern "C" int fzp_format_tigs(const fzp_tigs *t, int32_t ctg, const char *ctg_id, char **text, size_t *len) {
    if (!t || !ctg_id || !text || !len) { fzp_set_error("fzp_format_tigs: bad arguments"); return FZP_EINVAL; }
    std::string out;
    char hdr[320];
    for (int64_t i = 0; i < t->n_tigs; i++) {
        const fzp_tig &g = t->tigs[i];
        if (g.ctg != ctg) continue;
        const int hl = snprintf(hdr, sizeof hdr, ">%s_%03d_%d %d %d %d\n", ctg_id, g.block, g.phase, g.lo + 1, g.hi + 1, g.n_records);
        if (hl <= 0 || hl >= (int)sizeof hdr) { fzp_set_error("contig id too long"); return FZP_EINVAL; }
        out.append(hdr, (size_t)hl);
        out.append((const char *)t->seq + g.seq_off, (size_t)g.seq_len);
        out.push_back('\n');
    }
    char *p = (char *)malloc(out.size() + 1);
    if (!p) return FZP_ENOMEM;
    memcpy(p, out.data(), out.size());
    p[out.size()] = 0;
    *text = p; *len = out.size();
    return FZP_OK;
}
