// fzp_cns.hip -- K6: phased-pile consensus of every (block, phase) of a batch ("fzcns v2", v1 kept as a flag; twin: oracle/cns_oracle.c).
//
// BASELINE north_star config 4 asks for a haplotig consensus kernel on the phased read piles.  The reference has no
// consensus code of its own (falcon_sense lives in falcon_kit, Arrow in `variantCaller`, run_quiver.py:82-97), so this
// is this repo's own definition; the records, phases and blocks it consumes are the K1..K5 results already in HBM.
//   k_cns_extent   block spans [min site, max site] from K4's per-site block ids
//   k_cns_nrec     records per (block, phase) pile
//   k_cns_tiles    a workgroup owns 448 consecutive positions of one block: both phases' counters (10 x u32 per
//                  position and phase) live in LDS, its waves walk the records that overlap the tile -- phase looked
//                  up in K5's rows, CIGAR resumed at K2's 64-op checkpoint before the tile, columns dealt 64 at a time
//                  by the shared expander, D / I ops one per lane -- and the tile is written once, coalesced
//   k_cns_call     per position: deletion / majority base, and the first inserted base
//   k_ins_*        v2: insertions longer than one base.  The tally also lists every I op of >= 2 bases (slot, first base, length);
//                  level d = 2..8: the listed ops that still spell the chosen bases vote for their d-th base (global atomics on 4
//                  counters per slot), every voter reads the verdict (2 * count > cov, the weight falcon_sense gives a tag link),
//                  survivors go on to the next level.  A level costs three small launches over a quickly shrinking list.
//   scan + k_cns_emit   sequences laid out per (block, phase), order fixed by the scan
// HBM-bound integer work: 1 B symbol + 4 B/op in, 40 B of counters per (position, phase) touched by atomics, 1 B out.
#include <algorithm>
#include "fzp_expand.h"
#include <chrono>
#include "fzp_pk.h"

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
    const bool have = i < n_pvars;
    int32_t g = -1, pos = 0;
    if (have) {
        const fzp_pvar v = pvars[i];
        g = blk_base[site_ctg[v.site]] + v.block - 1;
        pos = sites[v.site].pos;
    }
    // a wave's records nearly always belong to one block: one pair of atomics per wave instead of sixty-four on the same two words (r5: 236 000 records on a few hundred
    // addresses took this kernel 0.3 ms per call at genome scale)
    const uint64_t m = __ballot(have);
    if (!m) return;
    const int32_t g0 = __builtin_amdgcn_readlane(g, __builtin_ctzll(m));
    if (__all(!have || g == g0)) {
        int32_t mn = have ? pos : 0x7fffffff, mx = have ? pos : -1;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { mn = min(mn, __shfl_xor(mn, d, 64)); mx = max(mx, __shfl_xor(mx, d, 64)); }
        if (lane_id() == 0) { atomicMin(&lo[g0], mn); atomicMax(&hi[g0], mx); }
    } else if (have) {
        atomicMin(&lo[g], pos);
        atomicMax(&hi[g], pos);
    }
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
        // (neighbouring records mostly add to the same pile: the lanes that do are counted by one of them)
        const uint32_t key = (uint32_t)(2 * g + v.preads[e].phase);
        for (uint64_t todo = __ballot(true); todo;) {
            const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, __builtin_ctzll(todo));
            const uint64_t same = __ballot(key == k0) & todo;
            if (key == k0) { if (lane_id() == __builtin_ctzll(same)) atomicAdd(&n_records[k0], (uint32_t)__popcll(same)); break; }
            todo &= ~same;
        }
    }
}

struct LongIns { int64_t slot, qidx; uint32_t n, alive; };      // an I op of >= 2 bases: counter slot of the position it follows, its first base in seq
constexpr int INS_MAX = 8;                                       // spec: at most 8 inserted bases per position

// 448 positions per tile: 2 x 10 x 448 counters (35 KB) + the expander's windows (8 KB) + the staged list entries (4.5 KB) = 47.5 KB, three
// workgroups per CU
constexpr int CNS_TILE = 448, CNS_THREADS = 512, CNS_STAGE = 192;
__global__ void __launch_bounds__(CNS_THREADS) k_cns_tiles(RecView rv, CnsView v, const int32_t *__restrict__ tile_blk, const int32_t *__restrict__ tile_start,
                                                           const int32_t *__restrict__ blk_ctg, const int64_t *__restrict__ ctg_rec_begin,
                                                           const int32_t *__restrict__ ctg_maxspan, uint32_t *__restrict__ cnt, LongIns *__restrict__ lins,
                                                           unsigned long long *__restrict__ n_lins, unsigned long long lins_cap) {
    __shared__ uint32_t l_cnt[2 * CN * CNS_TILE];      // [phase][counter][position]: consecutive lanes -> distinct banks
    __shared__ __attribute__((aligned(16))) uint32_t l_win[(CNS_THREADS / 64) * EXP_WIN];   // expand_record's window, one per wave
    // long I ops of this tile are staged here and appended to the global list with ONE atomic per workgroup: an atomic per op on the
    // list's counter -- 3.5 M of them on one address at cfg2 -- took 6 of this kernel's 8.6 ms
    __shared__ LongIns l_stage[CNS_STAGE];
    __shared__ uint32_t l_nstage, l_base_lo, l_base_hi;
    if (threadIdx.x == 0) l_nstage = 0;
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
            const int phu = __builtin_amdgcn_readlane(ph, l);
            uint32_t *lc = l_cnt + phu * (CN * CNS_TILE);
            const int32_t pos0 = rv.rec_pos[ru];
            const int64_t seq_end = rv.seq_off[ru + 1];
            expand_record(rv, ru, l_win + wave * EXP_WIN,
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
                            if (n >= 2 && lins) {
                                const int64_t len_ = (int64_t)v.hi[g] - v.lo[g] + 1;
                                const int64_t room = seq_end - qidx;
                                const LongIns e = LongIns{2 * v.cnt_off[g] + (int64_t)phu * len_ + (p - v.lo[g]), qidx, (uint32_t)(room < (int64_t)n ? room : (int64_t)n), 0u};
                                const uint32_t k = atomicAdd(&l_nstage, 1u);
                                if (k < (uint32_t)CNS_STAGE) l_stage[k] = e;
                                else {                                       // a tile with more than CNS_STAGE of them: straight to the list
                                    const unsigned long long at = atomicAdd(n_lins, 1ull);
                                    if (at < lins_cap) lins[at] = e;
                                }
                            }
                        }
                    }
                });
        }
    }
    __syncthreads();
    if (lins) {
        const uint32_t ns = min(l_nstage, (uint32_t)CNS_STAGE);
        if (threadIdx.x == 0 && ns) { const unsigned long long at = atomicAdd(n_lins, (unsigned long long)ns); l_base_lo = (uint32_t)at; l_base_hi = (uint32_t)(at >> 32); }
        __syncthreads();
        if (threadIdx.x < ns) {
            const unsigned long long at = (((unsigned long long)l_base_hi << 32) | l_base_lo) + threadIdx.x;
            if (at < lins_cap) lins[at] = l_stage[threadIdx.x];
        }
    }
    const int64_t len = (int64_t)hi - lo + 1;
    const int np = te - ts;
    for (int ph = 0; ph < 2; ph++) {
        uint32_t *dst = cnt + (2 * v.cnt_off[g] + (int64_t)ph * len + (ts - lo)) * CN;
        for (int i = threadIdx.x; i < np * CN; i += CNS_THREADS) dst[i] = l_cnt[ph * (CN * CNS_TILE) + (i % CN) * CNS_TILE + (i / CN)];
    }
}

// ---- the same tally from K1's packed records (r5): the alignment's 2-bit op stream (END first) and the 2-bit oriented read, as K2's k_pileup_pk reads them -- no byte
// SEQ, no run-length CIGAR, no checkpoint pass in front of K6 (k_gather16 + k_cig_ckpt + this kernel's run-length form were a quarter of a genome-scale run's kernel time).
// A lane owns a word of 16 ops.  An aligned column counts its base, a D op its position; an inserted base counts when it is the FIRST of its run in forward order -- the op
// whose next stream op is not an I -- at the position it follows (the cell's contig coordinate), with the run's length read off the I bits below it (this word's and the
// word's before) and, for runs of two and more, the run's first eight bases taken from the read and put INTO the list entry (qidx < 0: 2 bits per base from bit 0), so that
// what consumes the list needs no sequence.
__device__ __forceinline__ uint32_t even_bits16(uint32_t x) {      // the 16 even bits of x, packed
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u; x = (x | (x >> 2)) & 0x0f0f0f0fu; x = (x | (x >> 4)) & 0x00ff00ffu; x = (x | (x >> 8)) & 0xffffu;
    return x;
}
__global__ void __launch_bounds__(CNS_THREADS) k_cns_tiles_pk(PkView pv, CnsView v, const int32_t *__restrict__ tile_blk, const int32_t *__restrict__ tile_start,
                                                              const int32_t *__restrict__ blk_ctg, const int64_t *__restrict__ ctg_rec_begin,
                                                              const int32_t *__restrict__ ctg_maxspan, uint32_t *__restrict__ cnt, LongIns *__restrict__ lins,
                                                              unsigned long long *__restrict__ n_lins, unsigned long long lins_cap) {
    __shared__ uint32_t l_cnt[2 * CN * CNS_TILE];      // [phase][counter][position]
    __shared__ LongIns l_stage[CNS_STAGE];
    __shared__ uint32_t l_nstage, l_base_lo, l_base_hi;
    if (threadIdx.x == 0) l_nstage = 0;
    const int32_t g = tile_blk[blockIdx.x], ts = tile_start[blockIdx.x];
    const int c = blk_ctg[g];
    const int32_t lo = v.lo[g], hi = v.hi[g];
    const int32_t te = min(ts + CNS_TILE, hi + 1);      // exclusive
    for (int i = threadIdx.x; i < 2 * CN * CNS_TILE; i += CNS_THREADS) l_cnt[i] = 0;
    __syncthreads();
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
    const uint32_t span = (uint32_t)(te - ts);
    const int64_t len_g = (int64_t)hi - lo + 1;
    for (int64_t i0 = 0; first + i0 * NW + wave < last; i0 += 64) {
        // lane i prepares candidate first + (i0 + i) * NW + wave: overlap test, phase look-up, checkpoint search
        const int64_t r = first + (i0 + lane) * NW + wave;
        bool ok = r < last;
        int32_t ka = 0, ph = 0;
        // everything the walk of a record needs, fetched here by the record's lane and handed to the wave by lane reads: fetched in the walk, a record's five dependent loads
        // (record -> read -> stream offset -> checkpoint -> ...) were most of its time -- a record is one or two steps of 64 words in a tile of 448 positions
        int32_t L_iend = 0, L_jend = 0, L_nops = 0, L_strand = 0, L_cx = 0, L_cy = 0, L_pos0 = 0;
        uint32_t L_rq = 0, L_wlo = 0, L_whi = 0;
        if (ok) {
            const int32_t pos0 = v.rec_pos[r];
            L_pos0 = pos0;
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
                    const int64_t rd = pv.rec_read[r];
                    const PkRec p = pv.s.prec[rd];
                    const int32_t nck = (((p.n_ops + 15) >> 4) + 15) >> 4;
                    const uint32_t rq = pv.s.rcapq_scan[rd];
                    const int2 *ckr = pv.s.ck + ((size_t)(rq >> 2) + (size_t)rd);
                    // start one position ABOVE the tile's last: the inserted bases that follow position te - 1 stand in the stream in front of its column
                    ka = pk_ck_search(ckr, nck, p.j_end - te);
                    const int2 c0 = ckr[ka];
                    const int64_t wo = pv.s.read_woff[rd];
                    L_iend = p.i_end; L_jend = p.j_end; L_nops = p.n_ops; L_strand = p.strand; L_cx = c0.x; L_cy = c0.y; L_rq = rq; L_wlo = (uint32_t)wo; L_whi = (uint32_t)((uint64_t)wo >> 32);
                }
            }
        }
        const int32_t rel = (int32_t)(r - first);
        for (uint64_t todo = __ballot(ok); todo; todo &= todo - 1) {
            const int l = __builtin_ctzll(todo);
            const int64_t ru = first + __builtin_amdgcn_readlane(rel, l);
            const int32_t k0 = __builtin_amdgcn_readlane(ka, l);
            const int phu = __builtin_amdgcn_readlane(ph, l);
            uint32_t *lc = l_cnt + phu * (CN * CNS_TILE);
            (void)ru;
            const int32_t pos0 = __builtin_amdgcn_readlane(L_pos0, l);
            PkRec p;
            p.i_end = __builtin_amdgcn_readlane(L_iend, l); p.j_end = __builtin_amdgcn_readlane(L_jend, l); p.n_ops = __builtin_amdgcn_readlane(L_nops, l); p.strand = __builtin_amdgcn_readlane(L_strand, l);
            const uint32_t *__restrict__ ops = pv.s.ops + 4 * (size_t)(uint32_t)__builtin_amdgcn_readlane((int)L_rq, l);
            const int2 c0 = make_int2(__builtin_amdgcn_readlane(L_cx, l), __builtin_amdgcn_readlane(L_cy, l));
            const int64_t wo = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)L_whi, l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)L_wlo, l));
            const uint32_t *__restrict__ pk = (p.strand ? pv.s.read_rc : pv.s.read_pk) + wo;
            const int32_t nW = (p.n_ops + 15) >> 4;
            int32_t ib = p.i_end - c0.x, jb = p.j_end - c0.y;              // the cell the chunk's first op leaves
            for (int32_t w0 = 16 * k0; w0 < nW && jb >= ts; w0 += 64) {
                const int32_t wi = w0 + lane;
                const uint32_t x = wi < nW ? ops[wi] : 0u, vm = pk_valid(wi, p.n_ops);
                const uint32_t fM = ~(x | (x >> 1)) & vm, fI = (x & ~(x >> 1)) & vm, fD = (~x & (x >> 1)) & vm;
                const uint32_t cI = fM | fI, cJ = fM | fD;
                const uint32_t ci = (uint32_t)__popc(cI), cj = (uint32_t)__popc(cJ);
                const uint32_t si = wave_incl_scan_u32_dpp(ci), sj = wave_incl_scan_u32_dpp(cj);
                const int32_t i = ib - (int32_t)(si - ci), j = jb - (int32_t)(sj - cj);
                // this word's ops stand at contig positions [j - cj, j]
                if (vm != 0u && j >= ts && j - (int32_t)cj < te && i >= 0) {
                    const uint32_t q16 = pk_bases16(pk, i);
                    uint32_t bsh = 30u;                                        // bit offset of the base at the current read index inside q16
                    uint32_t pj = (uint32_t)(j - ts);                          // position inside the tile (wraps when outside)
                    int32_t ii = i;                                            // the current read index
                    // run starts: an I whose next stream op (the op above it: op o + 1, or the next word's op 0) is not an I
                    uint32_t above = fI >> 2;
                    if (wi + 1 < nW) { const uint32_t xn = ops[wi + 1]; if ((xn & 3u) == 1u && (pk_valid(wi + 1, p.n_ops) & 1u)) above |= 1u << 30; }
                    const uint32_t sI = fI & ~above;
                    uint32_t runs = 0;                                         // I bits of the word before this one (below) and of this word, one per op: bit 16 + o = op o
                    if (sI) runs = (even_bits16(fI) << 16) | (wi > 0 ? even_bits16(ops[wi - 1] & ~(ops[wi - 1] >> 1)) : 0u);
                    const uint32_t cI2 = cI << 1;
#pragma unroll
                    for (int o = 0; o < 16; o++) {
                        const uint32_t bit = 1u << (2 * o);
                        if (fM & bit) {
                            if (pj < span) atomicAdd(&lc[((q16 >> bsh) & 3u) * CNS_TILE + pj], 1u);
                        } else if (fD & bit) {
                            if (pj < span) atomicAdd(&lc[4 * CNS_TILE + pj], 1u);
                        } else if (sI & bit) {
                            if (pj < span && (int32_t)pj + ts >= pos0) {
                                atomicAdd(&lc[5 * CNS_TILE + pj], 1u);
                                atomicAdd(&lc[(6 + ((q16 >> bsh) & 3u)) * CNS_TILE + pj], 1u);
                                const uint32_t n_run = (uint32_t)__clz((int)~(runs << (15 - o)));      // consecutive I bits from op o down (this word, then the word below)
                                if (n_run >= 2u && lins) {
                                    const uint64_t two = ((uint64_t)pk[(ii >> 4) + 1] << 32) | pk[ii >> 4];      // the read from base ii on: the run's bases in forward order
                                    const uint32_t b16 = (uint32_t)(two >> (2 * (ii & 15))) & 0xffffu;
                                    const LongIns e = LongIns{2 * v.cnt_off[g] + (int64_t)phu * len_g + ((int64_t)pj + ts - lo), (int64_t)(0x8000000000000000ull | b16), n_run, 0u};
                                    const uint32_t k = atomicAdd(&l_nstage, 1u);
                                    if (k < (uint32_t)CNS_STAGE) l_stage[k] = e;
                                    else { const unsigned long long at = atomicAdd(n_lins, 1ull); if (at < lins_cap) lins[at] = e; }
                                }
                            }
                        }
                        const uint32_t took_i = (cI2 >> (2 * o)) & 2u;
                        bsh -= took_i;
                        ii -= (int32_t)(took_i >> 1);
                        pj -= (cJ >> (2 * o)) & 1u;
                    }
                }
                ib -= __builtin_amdgcn_readlane((int32_t)si, 63);
                jb -= __builtin_amdgcn_readlane((int32_t)sj, 63);
            }
        }
    }
    __syncthreads();
    if (lins) {
        const uint32_t ns = min(l_nstage, (uint32_t)CNS_STAGE);
        if (threadIdx.x == 0 && ns) { const unsigned long long at = atomicAdd(n_lins, (unsigned long long)ns); l_base_lo = (uint32_t)at; l_base_hi = (uint32_t)(at >> 32); }
        __syncthreads();
        if (threadIdx.x < ns) {
            const unsigned long long at = (((unsigned long long)l_base_hi << 32) | l_base_lo) + threadIdx.x;
            if (at < lins_cap) lins[at] = l_stage[threadIdx.x];
        }
    }
    const int np = te - ts;
    for (int ph = 0; ph < 2; ph++) {
        uint32_t *dst = cnt + (2 * v.cnt_off[g] + (int64_t)ph * len_g + (ts - lo)) * CN;
        for (int i = threadIdx.x; i < np * CN; i += CNS_THREADS) dst[i] = l_cnt[ph * (CN * CNS_TILE) + (i % CN) * CNS_TILE + (i / CN)];
    }
}

// one thread per (block, phase, position): the delta-0 call (0 or 1 base) and the first inserted base
// outputs: base0[i] (0 = nothing), ins_len[i] (0 / 1 after this kernel), ins_code[i] (2 bits per inserted base)
__global__ void __launch_bounds__(256) k_cns_call(int64_t n_slots, int n_blk, const int64_t *__restrict__ cnt_off, const int32_t *__restrict__ lo, const int32_t *__restrict__ hi,
                                                  const int32_t *__restrict__ blk_ctg, const int64_t *__restrict__ ctg_goff, const uint8_t *__restrict__ ref,
                                                  const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ n_records, int version, uint8_t *__restrict__ base0,
                                                  uint8_t *__restrict__ ins_len, uint32_t *__restrict__ ins_code, uint32_t *__restrict__ cand) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_slots) return;
    int a = 0, b = n_blk;                                   // block g with 2*cnt_off[g] <= i < 2*cnt_off[g+1]
    while (b - a > 1) { const int m = (a + b) >> 1; if (2 * cnt_off[m] <= i) a = m; else b = m; }
    const int g = a;
    const int64_t len = (int64_t)hi[g] - lo[g] + 1;
    const int64_t local = i - 2 * cnt_off[g];
    const int ph = local >= len ? 1 : 0;
    const int64_t x = local - (int64_t)ph * len;
    uint8_t s0 = 0, il = 0;
    uint32_t ic = 0, cd = 0;
    if (n_records[2 * g + ph] > 0) {
        const uint32_t *c = cnt + i * CN;
        const uint32_t cov = c[0] + c[1] + c[2] + c[3] + c[4];
        uint8_t refb = ref[ctg_goff[blk_ctg[g]] + lo[g] + x];
        if (refb >= 'a' && refb <= 'z') refb -= 32;
        if (cov == 0) s0 = refb;
        else {
            if (2 * c[4] <= cov) {
                uint32_t mx = max(max(c[0], c[1]), max(c[2], c[3]));
                const int rc = sym_code(refb);
                int pick = -1;
                if (rc < 4 && c[rc] == mx) pick = rc;
                for (int k = 0; k < 4 && pick < 0; k++) if (c[k] == mx) pick = k;
                s0 = code_sym(pick);
            }
            uint32_t mx = c[6];
            int pick = 0;
            for (int k = 1; k < 4; k++) if (c[6 + k] > mx) { mx = c[6 + k]; pick = k; }
            if (version >= 3) cd = 2 * c[5] > cov ? 1u : 0u;       // v3: more than half of the coverage carries an I op here -- length and bases come from k_ins_tally3 / k_ins_decide3
            else {
                const bool ins = version == 1 ? (2 * c[5] > cov && mx > 0) : (2 * mx > cov);
                if (ins) { il = 1; ic = (uint32_t)pick; }
            }
        }
    }
    base0[i] = s0; ins_len[i] = il; ins_code[i] = ic;
    if (cand) cand[i] = cd;
}
// ---- v3 ("fzcns v3", oracle/cns_oracle.c): per candidate position 8 length counters (I ops of exactly l bases, l = 2..8 counted here, l = 1 is the
// rest of the position's I ops) and 7 x 4 base counters (levels 2..8; level 1 is the tally's first-base counters) -- one pass over the list of long I ops
constexpr int V3_STRIDE = 40;      // [0..7] length l at [l - 1]; [8 + 4 q + base] level q + 1, q = 1..7
__global__ void __launch_bounds__(256) k_ins_tally3(int64_t n, const LongIns *__restrict__ e, const uint8_t *__restrict__ seq, const uint32_t *__restrict__ cand,
                                                    const uint32_t *__restrict__ cand_idx, uint32_t *__restrict__ ih) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const LongIns x = e[i];
    if (!cand[x.slot] || x.n < 2u) return;
    uint32_t *h = ih + (int64_t)cand_idx[x.slot] * V3_STRIDE;
    const uint32_t ln = min(x.n, (uint32_t)INS_MAX);
    atomicAdd(&h[ln - 1], 1u);
    for (uint32_t q = 1; q < ln; q++) {
        const int code = x.qidx < 0 ? (int)(((uint64_t)x.qidx >> (2 * q)) & 3u) : sym_code(seq[x.qidx + q]);      // (the packed tally's entries carry the run's first eight bases)
        if (code < 4) atomicAdd(&h[8 + 4 * q + code], 1u);
    }
}
__global__ void __launch_bounds__(256) k_ins_decide3(int64_t n_slots, const uint32_t *__restrict__ cand, const uint32_t *__restrict__ cand_idx, const uint32_t *__restrict__ ih,
                                                     const uint32_t *__restrict__ cnt, uint8_t *__restrict__ ins_len, uint32_t *__restrict__ ins_code) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_slots || !cand[i]) return;
    const uint32_t *c = cnt + i * CN;
    const uint32_t *h = ih + (int64_t)cand_idx[i] * V3_STRIDE;
    const uint32_t cov = c[0] + c[1] + c[2] + c[3] + c[4];
    uint32_t hist[9];
    uint32_t n_long = 0;
#pragma unroll
    for (int l = 2; l <= 8; l++) { hist[l] = h[l - 1]; n_long += hist[l]; }
    hist[1] = c[5] - n_long;
    uint32_t ge = c[5];
    int L = 0;
    for (int l = 1; l <= 8; l++) { if (2 * ge > cov) L = l; else break; ge -= hist[l]; }     // the median inserted length over the pile's reads
    if (L == 1) { const uint32_t mx1 = max(max(c[6], c[7]), max(c[8], c[9])); if (!(2 * mx1 > cov)) L = 0; }   // a single base: only with a majority for the same base
    uint32_t code = 0;
    int n = 0;
    for (int q = 0; q < L; q++) {
        const uint32_t *lv = q == 0 ? c + 6 : h + 8 + 4 * q;
        uint32_t mx = lv[0];
        int pick = 0;
        for (int k = 1; k < 4; k++) if (lv[k] > mx) { mx = lv[k]; pick = k; }
        if (mx == 0) break;
        code |= (uint32_t)pick << (2 * q);
        n++;
    }
    ins_len[i] = (uint8_t)n; ins_code[i] = code;
}
// ---- v2 insertion levels over the listed I ops
__global__ void __launch_bounds__(256) k_ins_init(int64_t n, LongIns *__restrict__ e, const uint8_t *__restrict__ seq, const uint8_t *__restrict__ ins_len, const uint32_t *__restrict__ ins_code,
                                                  unsigned long long *__restrict__ n_alive) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool al = false;
    if (i < n) {
        const LongIns x = e[i];
        al = ins_len[x.slot] == 1 && sym_code(seq[x.qidx]) == (int)(ins_code[x.slot] & 3u) && x.n >= 2;
        e[i].alive = al ? 1u : 0u;
    }
    const uint64_t m = __ballot(al);
    if (lane_id() == 0 && m) atomicAdd(n_alive, (unsigned long long)__popcll(m));
}
__global__ void __launch_bounds__(256) k_ins_vote(int64_t n, const LongIns *__restrict__ e, const uint8_t *__restrict__ seq, int d, uint32_t *__restrict__ lv) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const LongIns x = e[i];
    if (!x.alive || x.n < (uint32_t)d) return;
    const int code = sym_code(seq[x.qidx + d - 1]);
    if (code < 4) atomicAdd(&lv[x.slot * 4 + code], 1u);
}
__global__ void __launch_bounds__(256) k_ins_decide(int64_t n, LongIns *__restrict__ e, const uint8_t *__restrict__ seq, int d, const uint32_t *__restrict__ lv, const uint32_t *__restrict__ cnt,
                                                    uint8_t *__restrict__ ins_len, uint32_t *__restrict__ ins_code, unsigned long long *__restrict__ n_alive) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool al = false;
    if (i < n) {
        const LongIns x = e[i];
        if (x.alive && x.n >= (uint32_t)d) {
            const uint32_t *l = lv + x.slot * 4;
            uint32_t mx = l[0];
            int pick = 0;
            for (int k = 1; k < 4; k++) if (l[k] > mx) { mx = l[k]; pick = k; }
            const uint32_t *c = cnt + x.slot * CN;
            const uint32_t cov = c[0] + c[1] + c[2] + c[3] + c[4];
            if (2 * mx > cov) {
                // every voter of the slot writes the same values (the verdict is a function of the slot's counters)
                ins_len[x.slot] = (uint8_t)d;
                ins_code[x.slot] = (ins_code[x.slot] & ((1u << (2 * (d - 1))) - 1u)) | ((uint32_t)pick << (2 * (d - 1)));
                al = sym_code(seq[x.qidx + d - 1]) == pick && x.n > (uint32_t)d && d < INS_MAX;
            }
        }
        if (x.alive) e[i].alive = al ? 1u : 0u;
    }
    const uint64_t m = __ballot(al);
    if (lane_id() == 0 && m) atomicAdd(n_alive, (unsigned long long)__popcll(m));
}
__global__ void __launch_bounds__(256) k_ins_clear(int64_t n, const LongIns *__restrict__ e, uint32_t *__restrict__ lv) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint4 *p = (uint4 *)(lv + e[i].slot * 4);
    *p = make_uint4(0, 0, 0, 0);
}
__global__ void __launch_bounds__(256) k_cns_nout(int64_t n_slots, const uint8_t *__restrict__ base0, const uint8_t *__restrict__ ins_len, uint32_t *__restrict__ n_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_slots) n_out[i] = (base0[i] ? 1u : 0u) + ins_len[i];
}
__global__ void __launch_bounds__(256) k_cns_emit(int64_t n_slots, const uint32_t *__restrict__ off, const uint8_t *__restrict__ base0, const uint8_t *__restrict__ ins_len,
                                                  const uint32_t *__restrict__ ins_code, uint8_t *__restrict__ seq) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_slots) return;
    uint32_t o = off[i];
    if (base0[i]) seq[o++] = base0[i];
    const uint32_t n = ins_len[i], c = ins_code[i];
    for (uint32_t k = 0; k < n; k++) seq[o++] = code_sym((int)((c >> (2 * k)) & 3u));
}
// first output offset of every (block, phase): one gathered array instead of 2 * blocks tiny copies
__global__ void k_cns_first(int n_blk, const int64_t *__restrict__ cnt_off, const uint32_t *__restrict__ off, uint32_t total, uint32_t *__restrict__ first) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * n_blk) return;
    const int g = i >> 1, ph = i & 1;
    const int64_t len = cnt_off[g + 1] - cnt_off[g];
    first[i] = len > 0 ? off[2 * cnt_off[g] + (int64_t)ph * len] : total;
}
// polishing (fzp_polish_tigs): every aligned read of a tig is a member of the tig's one pile -- a phased_reads row (q_id, block 1, phase 0) per q_id, so that the
// tally's look-up finds what K5 would have written
__global__ void __launch_bounds__(256) k_polish_preads(int n_ctg, const int64_t *__restrict__ qoff, fzp_pread *__restrict__ preads) {
    const int c = blockIdx.y;
    if (c >= n_ctg) return;
    const int64_t a = qoff[c], n = qoff[c + 1] - a;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (int64_t)gridDim.x * 256) preads[a + q] = fzp_pread{(int32_t)q, 1, 0, 0, 0};
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

extern "C" int fzp_batch_consensus(fzp_ctx *ctx, fzp_batch *b, fzp_tigs *out) { return fzp_batch_consensus_v(ctx, b, 3, out); }

// The consensus with its sequence bytes left ON THE DEVICE (r5; fzp_pipe.hip copies them into the pinned block its write tasks own, under whatever runs next, and the
// tasks write every tig from where it lies -- the C-ABI entry below downloads them as before): the tig table (seq_off / seq_len into `seq`) comes back on the host.
// polish != nullptr (fzp_polish_tigs): the TEMPLATE is the pile's span -- one block per contig that has records, [0, its length), every record a member, the
// position's own base read from the whole template (the batch's ref holds only the prefix K2 evaluates).
int fzp_batch_consensus_dev(fzp_ctx *ctx, fzp_batch *b, int version, std::vector<fzp_tig> &tigs, DevBuf<uint8_t> &seq, uint64_t *n_seq, const fzp_cns_polish *polish) {
    if (!ctx || !b || version < 1 || version > 3) { fzp_set_error("fzp_batch_consensus: bad arguments"); return FZP_EINVAL; }
    tigs.clear(); *n_seq = 0;
    FZP_TRY(fzp_batch_source_ok(b));
    if (!b->have_aln || (!polish && (!b->have_blocks || !b->have_preads || !b->have_sites))) { fzp_set_error("fzp_batch_consensus: run FZP_STAGE_ALL on a batch with alignment records first"); return FZP_EINVAL; }
    FZP_TRY(fzp_bind(ctx));
    // a packed batch (fzp_align_to_batch) is tallied as it is by fzcns v3 (k_cns_tiles_pk); versions 1 and 2 walk the run-length records, which are made now (FZP_K6_BYTES: v3 too)
    const bool packed = b->packed && !b->have_bytes && version >= 3 && getenv("FZP_K6_BYTES") == nullptr;
    if (!packed) FZP_TRY(fzp_batch_need_bytes(ctx, b));
    hipStream_t st = ctx->stream;
    const int nc = b->n_ctg;
    // ---- blocks per contig and their spans
    DevBuf<int32_t> d_nblk, d_base, d_lo, d_hi, d_bctg;
    DevBuf<fzp_pread> p_preads;
    std::vector<int32_t> nblk((size_t)nc), base((size_t)nc + 1, 0);
    if (polish) {
        for (int c = 0; c < nc; c++) nblk[(size_t)c] = b->h_rec_begin[(size_t)c + 1] > b->h_rec_begin[(size_t)c] ? 1 : 0;
    } else {
        FZP_TRY(d_nblk.alloc((size_t)nc));
        hipLaunchKernelGGL(k_cns_nblk, dim3(nblocks(nc, 64)), dim3(64), 0, st, nc, b->pvar_begin.p, b->pvars.p, d_nblk.p);
        FZP_TRY(d_nblk.download(nblk.data(), (size_t)nc, st));
        FZP_HIP(hipStreamSynchronize(st));
    }
    for (int c = 0; c < nc; c++) base[(size_t)c + 1] = base[(size_t)c] + nblk[(size_t)c];
    const int NB = base[(size_t)nc];
    if (NB == 0 || (!polish && b->n_pvars == 0)) return FZP_OK;
    std::vector<int32_t> bctg((size_t)NB);
    for (int c = 0; c < nc; c++) for (int k = base[(size_t)c]; k < base[(size_t)c + 1]; k++) bctg[(size_t)k] = c;
    FZP_TRY(d_base.upload(base.data(), (size_t)nc + 1, st));
    FZP_TRY(d_bctg.upload(bctg.data(), (size_t)NB, st));
    FZP_TRY(d_lo.alloc((size_t)NB)); FZP_TRY(d_hi.alloc((size_t)NB));
    std::vector<int32_t> lo((size_t)NB), hi((size_t)NB);
    if (polish) {
        for (int g = 0; g < NB; g++) { lo[(size_t)g] = 0; hi[(size_t)g] = (int32_t)(polish->len[bctg[(size_t)g]] - 1); }
        FZP_TRY(d_lo.upload(lo.data(), (size_t)NB, st)); FZP_TRY(d_hi.upload(hi.data(), (size_t)NB, st));
        FZP_TRY(p_preads.alloc((size_t)std::max<int64_t>(b->n_qid, 1)));
        hipLaunchKernelGGL(k_polish_preads, dim3(64, (unsigned)nc), dim3(256), 0, st, nc, (const int64_t *)b->ctg_qoff.p, p_preads.p);
        FZP_HIP(hipStreamSynchronize(st));      // (lo / hi go out of this frame's vectors into the device arrays)
    } else {
        hipLaunchKernelGGL(k_fill32, dim3(nblocks(NB, 256)), dim3(256), 0, st, d_lo.p, (int64_t)NB, 0x7fffffff);
        hipLaunchKernelGGL(k_fill32, dim3(nblocks(NB, 256)), dim3(256), 0, st, d_hi.p, (int64_t)NB, -1);
        hipLaunchKernelGGL(k_cns_extent, dim3(nblocks(b->n_pvars, 256)), dim3(256), 0, st, b->n_pvars, b->pvars.p, b->sites.p, b->site_ctg.p, d_base.p, d_lo.p, d_hi.p);
        FZP_TRY(d_lo.download(lo.data(), (size_t)NB, st)); FZP_TRY(d_hi.download(hi.data(), (size_t)NB, st));
        FZP_HIP(hipStreamSynchronize(st));
    }
    std::vector<int64_t> cnt_off((size_t)NB + 1, 0);
    for (int g = 0; g < NB; g++) cnt_off[(size_t)g + 1] = cnt_off[(size_t)g] + (hi[(size_t)g] >= lo[(size_t)g] ? (int64_t)hi[(size_t)g] - lo[(size_t)g] + 1 : 0);
    const int64_t n_slots = 2 * cnt_off[(size_t)NB];
    if (n_slots >= (1ll << 31)) { fzp_set_error("fzp_batch_consensus: %lld block positions (limit 2^30 per batch)", (long long)n_slots / 2); return FZP_EINVAL; }
    // ---- tally
    DevBuf<int64_t> d_cnt_off;
    DevBuf<uint32_t> cnt, n_records, n_out, off, ins_code, lv;
    DevBuf<uint8_t> base0, ins_len;
    DevBuf<uint64_t> total;
    DevBuf<LongIns> lins;
    DevBuf<unsigned long long> n_lins;
    unsigned long long lins_cap = version >= 2 ? (unsigned long long)std::max<int64_t>(b->n_cig / 8, 4096) : 0;
    FZP_TRY(n_lins.alloc(2));
    FZP_TRY(d_cnt_off.upload(cnt_off.data(), (size_t)NB + 1, st));
    FZP_TRY(cnt.alloc((size_t)n_slots * CN));
    FZP_TRY(n_records.alloc((size_t)NB * 2)); FZP_TRY(n_records.zero((size_t)NB * 2, st));
    CnsView v = {b->rec_pos.p, b->rec_qid.p, b->rec_ctg.p, b->rec_span.p, b->cig_off.p, b->seq_off.p, b->ck_off.p, b->ck_ref.p, b->ck_q.p, b->cigar.p, b->seq.p,
                 polish ? p_preads.p : b->preads.p, polish ? b->ctg_qoff.p : b->pread_begin.p, d_base.p, d_lo.p, d_hi.p, d_cnt_off.p, b->n_rec};
    RecView rv = {b->rec_pos.p, b->rec_qid.p, b->rec_ctg.p, b->cig_off.p, b->seq_off.p, b->cigar.p, b->seq.p, b->ctg_goff.p, b->ctg_limit.p, b->n_rec};
    std::vector<int32_t> tblk, tstart;                    // tiles never span blocks
    for (int g = 0; g < NB; g++)
        for (int64_t t0 = lo[(size_t)g]; t0 <= hi[(size_t)g]; t0 += CNS_TILE) { tblk.push_back(g); tstart.push_back((int32_t)t0); }
    DevBuf<int32_t> d_tblk, d_tstart;
    FZP_TRY(d_tblk.upload(tblk.data(), tblk.size(), st)); FZP_TRY(d_tstart.upload(tstart.data(), tstart.size(), st));
    unsigned long long h_lins = 0;
    for (int attempt = 0; attempt < 2; attempt++) {         // second attempt only if the list of long I ops overflowed its first capacity
        if (version >= 2) FZP_TRY(lins.alloc((size_t)lins_cap));
        FZP_HIP(hipMemsetAsync(n_lins.p, 0, 16, st));
        if (b->n_rec > 0) {
            ProfScope ps(ctx, "k6_tally");
            if (attempt == 0) hipLaunchKernelGGL(k_cns_nrec, dim3(nblocks(b->n_rec, 256)), dim3(256), 0, st, v, n_records.p);
            if (!tblk.empty() && packed)
                hipLaunchKernelGGL(k_cns_tiles_pk, dim3((unsigned)tblk.size()), dim3(CNS_THREADS), 0, st, pk_view(b), v, d_tblk.p, d_tstart.p, d_bctg.p, b->ctg_rec_begin.p, b->ctg_maxspan.p,
                                   cnt.p, lins.p, n_lins.p, lins_cap);
            else if (!tblk.empty())
                hipLaunchKernelGGL(k_cns_tiles, dim3((unsigned)tblk.size()), dim3(CNS_THREADS), 0, st, rv, v, d_tblk.p, d_tstart.p, d_bctg.p, b->ctg_rec_begin.p, b->ctg_maxspan.p, cnt.p,
                                   version >= 2 ? lins.p : (LongIns *)nullptr, n_lins.p, lins_cap);
        } else {
            FZP_TRY(cnt.zero((size_t)n_slots * CN, st));
        }
        FZP_HIP(hipMemcpyAsync(&h_lins, n_lins.p, 8, hipMemcpyDeviceToHost, st));
        FZP_HIP(hipStreamSynchronize(st));                   // (the tile vectors also go out of scope below)
        if (h_lins <= lins_cap) break;
        lins_cap = h_lins;
    }
    // ---- call + layout
    FZP_TRY(n_out.alloc((size_t)n_slots)); FZP_TRY(off.alloc((size_t)n_slots)); FZP_TRY(base0.alloc((size_t)n_slots)); FZP_TRY(ins_len.alloc((size_t)n_slots));
    FZP_TRY(ins_code.alloc((size_t)n_slots)); FZP_TRY(total.alloc(1));
    {
        ProfScope ps(ctx, "k6_call");
        hipLaunchKernelGGL(k_cns_call, dim3(nblocks(n_slots, 256)), dim3(256), 0, st, n_slots, NB, d_cnt_off.p, d_lo.p, d_hi.p, d_bctg.p, polish ? polish->ref_off : b->ctg_goff.p, polish ? polish->ref : b->ref.p, cnt.p, n_records.p,
                           version, base0.p, ins_len.p, ins_code.p, version >= 3 ? n_out.p : (uint32_t *)nullptr);
    }
    if (version >= 3) {
        // candidate positions (more than half of the coverage inserts something) -> dense indices -> length and level counters -> the call
        ProfScope ps(ctx, "k6_ins_levels");
        FZP_TRY(fzp_exclusive_scan_u32(ctx, n_out.p, off.p, (size_t)n_slots, total.p));
        uint64_t n_cand = 0;
        FZP_HIP(hipMemcpyAsync(&n_cand, total.p, 8, hipMemcpyDeviceToHost, st));
        FZP_HIP(hipStreamSynchronize(st));
        if (n_cand > 0) {
            FZP_TRY(lv.alloc((size_t)n_cand * V3_STRIDE)); FZP_TRY(lv.zero((size_t)n_cand * V3_STRIDE, st));
            if (h_lins > 0) hipLaunchKernelGGL(k_ins_tally3, dim3(nblocks((int64_t)h_lins, 256)), dim3(256), 0, st, (int64_t)h_lins, lins.p, b->seq.p, n_out.p, off.p, lv.p);
            hipLaunchKernelGGL(k_ins_decide3, dim3(nblocks(n_slots, 256)), dim3(256), 0, st, n_slots, n_out.p, off.p, lv.p, cnt.p, ins_len.p, ins_code.p);
        }
    }
    if (version == 2 && h_lins > 0) {
        ProfScope ps(ctx, "k6_ins_levels");
        const int64_t ne = (int64_t)h_lins;
        FZP_TRY(lv.alloc((size_t)n_slots * 4)); FZP_TRY(lv.zero((size_t)n_slots * 4, st));
        unsigned long long alive = 0;
        FZP_HIP(hipMemsetAsync(n_lins.p + 1, 0, 8, st));
        hipLaunchKernelGGL(k_ins_init, dim3(nblocks(ne, 256)), dim3(256), 0, st, ne, lins.p, b->seq.p, ins_len.p, ins_code.p, n_lins.p + 1);
        FZP_HIP(hipMemcpyAsync(&alive, n_lins.p + 1, 8, hipMemcpyDeviceToHost, st));
        FZP_HIP(hipStreamSynchronize(st));
        for (int d = 2; d <= INS_MAX && alive > 0; d++) {
            hipLaunchKernelGGL(k_ins_vote, dim3(nblocks(ne, 256)), dim3(256), 0, st, ne, lins.p, b->seq.p, d, lv.p);
            FZP_HIP(hipMemsetAsync(n_lins.p + 1, 0, 8, st));
            hipLaunchKernelGGL(k_ins_decide, dim3(nblocks(ne, 256)), dim3(256), 0, st, ne, lins.p, b->seq.p, d, lv.p, cnt.p, ins_len.p, ins_code.p, n_lins.p + 1);
            hipLaunchKernelGGL(k_ins_clear, dim3(nblocks(ne, 256)), dim3(256), 0, st, ne, lins.p, lv.p);
            FZP_HIP(hipMemcpyAsync(&alive, n_lins.p + 1, 8, hipMemcpyDeviceToHost, st));
            FZP_HIP(hipStreamSynchronize(st));
        }
    }
    hipLaunchKernelGGL(k_cns_nout, dim3(nblocks(n_slots, 256)), dim3(256), 0, st, n_slots, base0.p, ins_len.p, n_out.p);
    FZP_TRY(fzp_exclusive_scan_u32(ctx, n_out.p, off.p, (size_t)n_slots, total.p));
    uint64_t tot = 0;
    FZP_HIP(hipMemcpyAsync(&tot, total.p, 8, hipMemcpyDeviceToHost, st));
    FZP_HIP(hipStreamSynchronize(st));
    FZP_TRY(seq.alloc((size_t)tot));
    {
        ProfScope ps(ctx, "k6_emit");
        hipLaunchKernelGGL(k_cns_emit, dim3(nblocks(n_slots, 256)), dim3(256), 0, st, n_slots, off.p, base0.p, ins_len.p, ins_code.p, seq.p);
    }
    // ---- results: sequence bytes, and per (block, phase) its offset = off[] at its first slot
    std::vector<uint32_t> h_nrec((size_t)NB * 2), h_first((size_t)NB * 2 + 1, (uint32_t)tot);
    FZP_TRY(n_records.download(h_nrec.data(), (size_t)NB * 2, st));
    DevBuf<uint32_t> d_first;
    FZP_TRY(d_first.alloc((size_t)NB * 2));
    hipLaunchKernelGGL(k_cns_first, dim3(nblocks(2 * NB, 256)), dim3(256), 0, st, NB, d_cnt_off.p, off.p, (uint32_t)tot, d_first.p);
    FZP_TRY(d_first.download(h_first.data(), (size_t)NB * 2, st));
    if (hipStreamSynchronize(st) != hipSuccess) { fzp_set_error("consensus tables: download failed"); return FZP_EDEVICE; }
    // slot order == output order, so a tig ends where the next non-empty one starts
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
    *n_seq = tot;
    FZP_HIP(hipGetLastError());
    return FZP_OK;
}

extern "C" int fzp_batch_consensus_v(fzp_ctx *ctx, fzp_batch *b, int version, fzp_tigs *out) {
    if (!ctx || !b || !out || version < 1 || version > 3) { fzp_set_error("fzp_batch_consensus: bad arguments"); return FZP_EINVAL; }
    memset(out, 0, sizeof *out);
    std::vector<fzp_tig> tigs;
    DevBuf<uint8_t> seq;
    uint64_t tot = 0;
    FZP_TRY(fzp_batch_consensus_dev(ctx, b, version, tigs, seq, &tot, nullptr));
    uint8_t *hseq = (uint8_t *)malloc((size_t)(tot ? tot : 1));
    if (!hseq) return FZP_ENOMEM;
    if (tot && (hipMemcpyAsync(hseq, seq.p, (size_t)tot, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)) {
        (void)hipGetLastError(); free(hseq); fzp_set_error("consensus download failed"); return FZP_EDEVICE;
    }
    out->n_tigs = (int64_t)tigs.size();
    out->tigs = (fzp_tig *)malloc((tigs.size() ? tigs.size() : 1) * sizeof(fzp_tig));
    if (!out->tigs) { free(hseq); return FZP_ENOMEM; }
    if (!tigs.empty()) memcpy(out->tigs, tigs.data(), tigs.size() * sizeof(fzp_tig));
    out->seq = hseq; out->n_seq = (int64_t)tot;
    return FZP_OK;
}

// ---- polishing: a tig as the template (the consensus role of run_quiver.py:82-97 -- `pbalign` of a tig's routed reads to the tig, then `variantCaller`'s per-tig call --
// with this repo's own aligner and pile vote: K1 aligns every read to ITS tig, K6's packed tally (fzcns v3, the insertion rule unchanged) runs over the whole tig as one
// pile).  The tigs are the layout's (graphs_to_h_tigs.py:406-410,558-562: p_ctg.<ctg>.fa / h_ctg_all.<ctg>.fa), the reads what fzp_track_reads / fzp_bam_route assign to them.
// Every input tig comes back, in input order: block 1, phase 0, [0, len); a tig no read aligned to comes back as it went in (upper-cased), n_records = 0.
extern "C" int fzp_polish_tigs(fzp_ctx *ctx, int32_t n_tigs, const uint8_t *const *tig_seq, const int64_t *tig_len, int64_t n_reads, const int32_t *read_tig,
                               const int64_t *read_off, const uint8_t *read_seq, const fzp_align_params *params, fzp_tigs *out) {
    if (!ctx || n_tigs <= 0 || !tig_seq || !tig_len || n_reads < 0 || (n_reads && (!read_tig || !read_off || !read_seq)) || !out) { fzp_set_error("fzp_polish_tigs: bad arguments"); return FZP_EINVAL; }
    memset(out, 0, sizeof *out);
    fzp_alnjob *job = nullptr;
    fzp_batch *b = nullptr;
    std::vector<fzp_tig> tigs;
    DevBuf<uint8_t> seq;
    uint64_t tot = 0;
    static const bool timing = getenv("FZP_PIPE_TIMING") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (timing) fprintf(stderr, "[fzp_polish_tigs] +%.2f ms: %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_0).count(), what); };
    int rc = fzp_align_create(ctx, n_tigs, tig_seq, tig_len, n_reads, read_tig, read_off, read_seq, params, &job);
    lap("alignment job created (upload, pack, index)");
    if (rc == FZP_OK) rc = fzp_align_run(ctx, job);
    if (rc == FZP_OK) rc = fzp_align_to_batch(ctx, job, &b);
    lap("reads aligned");
    if (rc == FZP_OK) {
        fzp_cns_polish P;
        P.len = tig_len;
        fzp_align_templates(job, &P.ref, &P.ref_off);
        rc = fzp_batch_consensus_dev(ctx, b, 3, tigs, seq, &tot, &P);
        lap("piles called");
    }
    // every tig once, input order; the ones without a pile as they came.  (r6: a tig's bases come straight from the device to their place in the caller's block -- through
    // a zero-filled vector and a second copy the 100 Mb of the bench's twenty tigs cost 50 ms of a 100 ms call)
    int64_t total = 0;
    std::vector<const fzp_tig *> of((size_t)n_tigs, nullptr);
    if (rc == FZP_OK) {
        for (const auto &t : tigs) of[(size_t)t.ctg] = &t;
        for (int c = 0; c < n_tigs; c++) total += of[(size_t)c] ? of[(size_t)c]->seq_len : tig_len[c];
        out->tigs = (fzp_tig *)malloc((size_t)n_tigs * sizeof(fzp_tig));
        out->seq = (uint8_t *)malloc((size_t)(total ? total : 1));
        if (!out->tigs || !out->seq) rc = FZP_ENOMEM;
    }
    if (rc == FZP_OK) {
        int64_t at = 0;
        for (int c = 0; c < n_tigs && rc == FZP_OK; c++) {
            fzp_tig t;
            memset(&t, 0, sizeof t);
            t.ctg = c; t.block = 1; t.phase = 0; t.lo = 0; t.hi = (int32_t)(tig_len[c] - 1); t.seq_off = at;
            if (of[(size_t)c]) {
                t.n_records = of[(size_t)c]->n_records; t.seq_len = of[(size_t)c]->seq_len;
                if (t.seq_len && hipMemcpyAsync(out->seq + at, seq.p + of[(size_t)c]->seq_off, (size_t)t.seq_len, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) {
                    (void)hipGetLastError(); fzp_set_error("fzp_polish_tigs: consensus download failed"); rc = FZP_EDEVICE;
                }
            } else {
                t.seq_len = tig_len[c];
                for (int64_t i = 0; i < tig_len[c]; i++) { const uint8_t ch = tig_seq[c][i]; out->seq[at + i] = (ch >= 'a' && ch <= 'z') ? (uint8_t)(ch - 32) : ch; }
            }
            at += t.seq_len;
            out->tigs[c] = t;
        }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == FZP_OK) { (void)hipGetLastError(); fzp_set_error("fzp_polish_tigs: consensus download failed"); rc = FZP_EDEVICE; }
        lap("sequences on the host");
    }
    if (b) fzp_batch_destroy(ctx, b);
    if (job) fzp_align_destroy(ctx, job);
    lap("job destroyed");
    if (rc != FZP_OK) { free(out->tigs); free(out->seq); memset(out, 0, sizeof *out); return rc; }
    out->n_tigs = n_tigs; out->n_seq = total;
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
