// fzp_hetcall.hip -- K2: pileup + het-SNP call + variant_map (make_het_call, phasing.py:14-135)
//                    K3: per-site allele sets + association table (generate_association_table, phasing.py:137-206)
//
// Batch form of the reference's streaming sweep (valid because accepted records are POS-sorted,
// enforced by fzp_parse_sam): count A/C/G/T (+ a non-ACGT symbol tracker) per evaluated position
// over ALL accepted records, then call every position p < POS(last accepted record).
//
// Roofline: HBM-bound integer scans.  Algorithmic bytes per aligned column: 1 B symbol + the CIGAR
// word stream (4 B/op); per evaluated position: 16 B counters + 4 B symbol tracker written once
// from LDS tiles and read back once, + 9 B of flag/scan state.  No MFMA: nothing here is GEMM-shaped.
#include "fzp_expand.h"
#include "fzp_pk.h"

namespace {

// ---- CIGAR checkpoints: (reference, query) offsets at the start of every 64-op chunk of every record
__global__ void __launch_bounds__(256) k_cig_ckpt(RecView v, const int64_t *__restrict__ ck_off, int32_t *__restrict__ ck_ref, int32_t *__restrict__ ck_q,
                                                  int32_t *__restrict__ rec_span, int32_t *__restrict__ ctg_maxspan) {
    const int lane = lane_id();
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < v.n_rec; r += (int64_t)gridDim.x * 4) {
        const int64_t c0 = v.cig_off[r], c1 = v.cig_off[r + 1];
        int64_t ck = ck_off[r];
        int32_t rp = 0, qp = 0;
        for (int64_t cb = c0; cb < c1; cb += 256) {         // four 64-op chunks per round: their loads are in flight together
            uint32_t w[4];
#pragma unroll
            for (int u = 0; u < 4; u++) w[u] = (cb + u * 64 + lane < c1) ? v.cigar[cb + u * 64 + lane] : 0u;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (cb + u * 64 >= c1) break;
                if (lane == 0) { ck_ref[ck] = rp; ck_q[ck] = qp; }
                ck++;
                const uint32_t len = w[u] >> 4, t = w[u] & 15u;
                const bool isM = (t == FZP_OP_M) | (t == FZP_OP_EQ) | (t == FZP_OP_X);
                rp += wave_sum_i32_dpp((int32_t)((isM | (t == FZP_OP_D)) ? len : 0u));
                qp += wave_sum_i32_dpp((int32_t)((isM | (t == FZP_OP_I) | (t == FZP_OP_S)) ? len : 0u));
            }
        }
        if (lane == 0) { rec_span[r] = rp; atomicMax(&ctg_maxspan[v.rec_ctg[r]], rp); }
    }
}

// ---- K2a: column counts by position tile.  A workgroup owns TILE consecutive positions of one contig, keeps
// their A/C/G/T counters in LDS (the tracker of symbols other than ACGT -- rare -- lives in HBM, pre-zeroed, and takes global atomics), lets its waves walk the records that overlap the tile
// -- each from the CIGAR checkpoint just before the tile -- and writes the counters out once, coalesced.  No global
// atomics: HBM sees the symbols (1 B/column), the CIGAR words and 20 B per position, about the algorithmic minimum.
constexpr int PILE_TILE = FZP_POS_TILE, PILE_THREADS = 512;
__global__ void __launch_bounds__(PILE_THREADS) k_pileup_tiles(RecView v, const int32_t *__restrict__ tile_ctg, const int32_t *__restrict__ tile_start,
                                                      const int64_t *__restrict__ ctg_rec_begin, const int32_t *__restrict__ ctg_maxspan,
                                                      const int32_t *__restrict__ rec_span, const int64_t *__restrict__ ck_off, const int32_t *__restrict__ ck_ref,
                                                      const int32_t *__restrict__ ck_q, uint32_t *__restrict__ cnt, uint32_t *__restrict__ oth,
                                                      unsigned long long *__restrict__ blk_live) {
    __shared__ uint32_t l_cnt[4 * PILE_TILE];   // [code][position]: consecutive lanes = consecutive positions = distinct banks
    __shared__ __attribute__((aligned(16))) uint32_t l_win[(PILE_THREADS / 64) * EXP_WIN];   // expand_record's window, one per wave
    const int c = tile_ctg[blockIdx.x];
    const int32_t ts = tile_start[blockIdx.x];
    const int32_t lim = v.ctg_limit[c];
    const int32_t te = min(ts + PILE_TILE, lim);
    // records of this contig that can overlap [ts, te): POS < te and POS > ts - max_span
    const int64_t rb = ctg_rec_begin[c], re = ctg_rec_begin[c + 1];
    const int32_t ms = ctg_maxspan[c];
    int64_t lo = rb, hi = re;
    {   // first record with POS > ts - ms
        int64_t a = rb, b = re;
        while (a < b) { int64_t m = (a + b) >> 1; if (v.rec_pos[m] <= ts - ms) a = m + 1; else b = m; }
        lo = a;
        a = lo; b = re;   // first record with POS >= te
        while (a < b) { int64_t m = (a + b) >> 1; if (v.rec_pos[m] < te) a = m + 1; else b = m; }
        hi = a;
    }
    // a tile no record can reach (the stretch before a contig's first read, a coverage hole) is left alone: its eight 256-position blocks
    // stay marked dead (blk_live was zeroed) and neither k_site_flag nor k_site_emit nor anybody else touches its counters
    if (lo >= hi) return;
    const int64_t g0 = v.ctg_goff[c] + ts;           // multiple of PILE_TILE: the tile is blocks (g0 >> 8) .. + 7
    if (threadIdx.x == 0) blk_live[g0 >> 11] = 0x0101010101010101ull;
    uint32_t *oth_t = oth + g0;                      // the tile's tracker words in HBM: zeroed here, by the only workgroup that uses them
    for (int i = threadIdx.x; i < PILE_TILE; i += PILE_THREADS) oth_t[i] = 0u;
    for (int i = threadIdx.x; i < PILE_TILE * 4; i += PILE_THREADS) l_cnt[i] = 0;
    __syncthreads();
    // candidate lo + i*NW + wave belongs to lane i of this wave: the per-record look-ups (span test, checkpoint
    // search) run lane-parallel, then the wave walks its records one at a time
    const int wave = threadIdx.x >> 6, lane = lane_id();
    constexpr int NW = PILE_THREADS / 64;
    for (int64_t i0 = 0; lo + i0 * NW + wave < hi; i0 += 64) {
        const int64_t r = lo + (i0 + lane) * NW + wave;
        bool ok = r < hi;
        int32_t a = 0, cr = 0, cq = 0;
        if (ok) {
            const int32_t pos0 = v.rec_pos[r];
            ok = pos0 + rec_span[r] > ts;
            if (ok) {
                // last checkpoint whose reference offset is <= ts - pos0 (chunk 0 if the record starts inside the tile)
                const int64_t k0 = ck_off[r];
                int32_t b = (int32_t)(ck_off[r + 1] - k0);
                const int32_t want = ts - pos0;
                while (b - a > 1) { int32_t m = (a + b) >> 1; if (ck_ref[k0 + m] <= want) a = m; else b = m; }
                cr = ck_ref[k0 + a]; cq = ck_q[k0 + a];
            }
        }
        const int32_t rel = (int32_t)(r - lo);
        for (uint64_t todo = __ballot(ok); todo; todo &= todo - 1) {
            const int l = __builtin_ctzll(todo);
            const int64_t ru = lo + __builtin_amdgcn_readlane(rel, l);
            expand_record(v, ru, l_win + wave * EXP_WIN, [&](int32_t pos, uint8_t sym) {
                const uint32_t p = (uint32_t)(pos - ts);
                const bool in = p < (uint32_t)(te - ts);
                const int code = sym_code(sym);
                if (in & (code < 4)) atomicAdd(&l_cnt[code * PILE_TILE + p], 1u);
                if (__any(in & (code == 4))) {          // symbols other than ACGT: rare, kept off the common path
                    if (in & (code == 4)) {
                        uint32_t old = atomicCAS(&oth_t[p], 0u, (uint32_t)sym);
                        if (old != 0u && (old & 0xffu) != (uint32_t)sym) atomicOr(&oth_t[p], 0x100u);
                    }
                }
            }, __builtin_amdgcn_readlane(a, l), __builtin_amdgcn_readlane(cr, l), __builtin_amdgcn_readlane(cq, l), te);
        }
    }
    __syncthreads();
    // the whole tile goes out (positions at or beyond the contig's limit hold zeros: the blocks the call kernels read are complete)
    for (int i = threadIdx.x; i < PILE_TILE * 4; i += PILE_THREADS) cnt[g0 * 4 + i] = l_cnt[(i & 3) * PILE_TILE + (i >> 2)];
}


// ================================================================================ K2 on K1's packed records (r5)
// The same pileup and the same variant_map rows, read from what K1 already holds: the alignment's 2-bit op stream (END first) and the 2-bit oriented read -- no byte SEQ,
// no run-length CIGAR, no 64-op checkpoint pass (PkRec / the 256-op checkpoints come out of k_tb_cigar's last sweep: fzp_batch.h).  A lane owns a WORD of 16 ops: two
// popcounts say what the word consumes, a wave scan gives every word the cell its first op leaves, and the lane walks its 16 ops -- an aligned column is an op 0 at cell
// (i, j): reference position j, symbol = base i of the read (phasing.py:77-96; the stream's I / D ops are the walk's `qp += n` / `rp += n`).  ~12 instructions per op and
// lane against ~55 per 64 columns and wave for the run-length form's expansion, i.e. about a third of the issue slots per column.
__global__ void __launch_bounds__(PILE_THREADS) k_pileup_pk(PkView v, const int32_t *__restrict__ tile_ctg, const int32_t *__restrict__ tile_start,
                                                            const int64_t *__restrict__ ctg_rec_begin, const int32_t *__restrict__ ctg_maxspan,
                                                            const int32_t *__restrict__ rec_span, uint32_t *__restrict__ cnt, uint32_t *__restrict__ oth,
                                                            unsigned long long *__restrict__ blk_live) {
    __shared__ uint32_t l_cnt[4 * PILE_TILE];   // [code][position]
    const int c = tile_ctg[blockIdx.x];
    const int32_t ts = tile_start[blockIdx.x];
    const int32_t lim = v.ctg_limit[c];
    const int32_t te = min(ts + PILE_TILE, lim);
    const int64_t rb = ctg_rec_begin[c], re = ctg_rec_begin[c + 1];
    const int32_t ms = ctg_maxspan[c];
    int64_t lo, hi;
    {
        int64_t a = rb, b = re;
        while (a < b) { int64_t m = (a + b) >> 1; if (v.rec_pos[m] <= ts - ms) a = m + 1; else b = m; }
        lo = a;
        b = re;
        while (a < b) { int64_t m = (a + b) >> 1; if (v.rec_pos[m] < te) a = m + 1; else b = m; }
        hi = a;
    }
    if (lo >= hi) return;                            // (a tile no record reaches stays dead: k_pileup_tiles)
    const int64_t g0 = v.ctg_goff[c] + ts;
    if (threadIdx.x == 0) blk_live[g0 >> 11] = 0x0101010101010101ull;
    uint32_t *oth_t = oth + g0;                      // packed reads hold A, C, G, T only: the tracker of other symbols stays zero
    for (int i = threadIdx.x; i < PILE_TILE; i += PILE_THREADS) oth_t[i] = 0u;
    for (int i = threadIdx.x; i < PILE_TILE * 4; i += PILE_THREADS) l_cnt[i] = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = lane_id();
    constexpr int NW = PILE_THREADS / 64;
    const uint32_t span = (uint32_t)(te - ts);
    for (int64_t i0 = 0; lo + i0 * NW + wave < hi; i0 += 64) {
        const int64_t r = lo + (i0 + lane) * NW + wave;
        bool ok = r < hi;
        int32_t ka = 0;
        // (what a record's walk needs is fetched here, by the record's lane, and handed to the wave by lane reads: fetched in the walk, the record's dependent loads -- record ->
        //  read -> stream offset -> checkpoint -- stood in front of every record's two or three steps of 64 words)
        int32_t L_iend = 0, L_jend = 0, L_nops = 0, L_strand = 0, L_cx = 0, L_cy = 0;
        uint32_t L_rq = 0, L_wlo = 0, L_whi = 0;
        if (ok) {
            ok = v.rec_pos[r] + rec_span[r] > ts;
            if (ok) {
                const int64_t rd = v.rec_read[r];
                const PkRec p = v.s.prec[rd];
                const int32_t nck = (((p.n_ops + 15) >> 4) + 15) >> 4;
                const uint32_t rq = v.s.rcapq_scan[rd];
                const int2 *ckr = v.s.ck + ((size_t)(rq >> 2) + (size_t)rd);
                // the stream runs from the alignment's end down: start at the last checkpoint that has not yet passed the tile's last position
                ka = pk_ck_search(ckr, nck, p.j_end - (te - 1));
                const int2 c0 = ckr[ka];
                const int64_t wo = v.s.read_woff[rd];
                L_iend = p.i_end; L_jend = p.j_end; L_nops = p.n_ops; L_strand = p.strand; L_cx = c0.x; L_cy = c0.y; L_rq = rq; L_wlo = (uint32_t)wo; L_whi = (uint32_t)((uint64_t)wo >> 32);
            }
        }
        for (uint64_t todo = __ballot(ok); todo; todo &= todo - 1) {
            const int l = __builtin_ctzll(todo);
            const int32_t k0 = __builtin_amdgcn_readlane(ka, l);
            PkRec p;
            p.i_end = __builtin_amdgcn_readlane(L_iend, l); p.j_end = __builtin_amdgcn_readlane(L_jend, l); p.n_ops = __builtin_amdgcn_readlane(L_nops, l); p.strand = __builtin_amdgcn_readlane(L_strand, l);
            const uint32_t *__restrict__ ops = v.s.ops + 4 * (size_t)(uint32_t)__builtin_amdgcn_readlane((int)L_rq, l);
            const int2 c0 = make_int2(__builtin_amdgcn_readlane(L_cx, l), __builtin_amdgcn_readlane(L_cy, l));
            const int64_t wo = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)L_whi, l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)L_wlo, l));
            const uint32_t *__restrict__ pk = (p.strand ? v.s.read_rc : v.s.read_pk) + wo;
            const int32_t nW = (p.n_ops + 15) >> 4;
            int32_t ib = p.i_end - c0.x, jb = p.j_end - c0.y;              // the cell the chunk's first op leaves
            for (int32_t w0 = 16 * k0; w0 < nW && jb >= ts; w0 += 64) {
                const int32_t wi = w0 + lane;
                const uint32_t x = wi < nW ? ops[wi] : 0u, vm = pk_valid(wi, p.n_ops);
                const uint32_t fM = ~(x | (x >> 1)) & vm, cI = fM | ((x & ~(x >> 1)) & vm), cJ = fM | ((~x & (x >> 1)) & vm);
                const uint32_t ci = (uint32_t)__popc(cI), cj = (uint32_t)__popc(cJ);
                const uint32_t si = wave_incl_scan_u32_dpp(ci), sj = wave_incl_scan_u32_dpp(cj);
                int32_t i = ib - (int32_t)(si - ci), j = jb - (int32_t)(sj - cj);
                // this word's columns lie in [j - cj + 1, j]
                if (fM != 0u && j >= ts && j - (int32_t)cj + 1 < te && i >= 0) {
                    const uint32_t q16 = pk_bases16(pk, i);
                    uint32_t bsh = 30u;                                        // bit offset of the base at the current i inside q16
                    uint32_t pj = (uint32_t)(j - ts);                          // position inside the tile (wraps when outside)
                    const uint32_t cI2 = cI << 1;
#pragma unroll
                    for (int o = 0; o < 16; o++) {
                        if ((fM >> (2 * o)) & 1u) {
                            if (pj < span) atomicAdd(&l_cnt[((q16 >> bsh) & 3u) * PILE_TILE + pj], 1u);
                        }
                        bsh -= (cI2 >> (2 * o)) & 2u;
                        pj -= (cJ >> (2 * o)) & 1u;
                    }
                }
                ib -= __builtin_amdgcn_readlane((int32_t)si, 63);
                jb -= __builtin_amdgcn_readlane((int32_t)sj, 63);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PILE_TILE * 4; i += PILE_THREADS) cnt[g0 * 4 + i] = l_cnt[(i & 3) * PILE_TILE + (i >> 2)];
}

// variant_map rows from the packed records: k_vmap_sites with the symbol looked up in the op stream -- the 256-op checkpoint, at most 16 words skipped by their popcounts,
// then the ops of one word
__global__ void __launch_bounds__(256) k_vmap_pk(PkView v, int64_t n_sites, const fzp_site *__restrict__ sites, const int32_t *__restrict__ site_ctg,
                                                 const int64_t *__restrict__ ctg_rec_begin, const int32_t *__restrict__ ctg_maxspan,
                                                 const int32_t *__restrict__ rec_span, int32_t *__restrict__ vmap_qid) {
    const int lane = lane_id();
    for (int64_t si = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); si < n_sites; si += (int64_t)gridDim.x * 4) {
        const fzp_site s = sites[si];
        const int c = site_ctg[si];
        const int32_t pos = s.pos;
        const int64_t rb = ctg_rec_begin[c], re = ctg_rec_begin[c + 1];
        int64_t lo, hi;
        {
            const int32_t ms = ctg_maxspan[c];
            int64_t a = rb, b = re;
            while (a < b) { const int64_t m = (a + b) >> 1; if (v.rec_pos[m] <= pos - ms) a = m + 1; else b = m; }
            lo = a;
            b = re;
            while (a < b) { const int64_t m = (a + b) >> 1; if (v.rec_pos[m] <= pos) a = m + 1; else b = m; }
            hi = a;
        }
        int32_t base0 = 0, base1 = 0;
        for (int64_t r0 = lo; r0 < hi; r0 += 64) {
            const int64_t r = r0 + lane;
            int al = -1;
            if (r < hi && pos - v.rec_pos[r] < rec_span[r]) {
                const int64_t rd = v.rec_read[r];
                const PkRec p = v.s.prec[rd];
                const int32_t nW = (p.n_ops + 15) >> 4, nck = (nW + 15) >> 4;
                const int2 *ckp = v.s.ck + ((size_t)(v.s.rcapq_scan[rd] >> 2) + (size_t)rd);
                const int32_t k0 = pk_ck_search(ckp, nck, p.j_end - pos);
                const int2 c0 = ckp[k0];
                const uint32_t *__restrict__ ops = v.s.ops + 4 * (size_t)v.s.rcapq_scan[rd];
                int32_t i = p.i_end - c0.x, j = p.j_end - c0.y;
                for (int32_t wi = 16 * k0; wi < nW && j >= pos; wi++) {
                    const uint32_t x = ops[wi], vm = pk_valid(wi, p.n_ops);
                    const uint32_t fM = ~(x | (x >> 1)) & vm, cI = fM | ((x & ~(x >> 1)) & vm), cJ = fM | ((~x & (x >> 1)) & vm);
                    const int32_t cj = __popc(cJ);
                    if (j - cj >= pos) { i -= __popc(cI); j -= cj; continue; }      // the op that consumes `pos` is not in this word
                    // it is the (j - pos + 1)-th contig-consuming op of the word
                    uint32_t m = cJ;
                    for (int32_t k = j - pos; k > 0; k--) m &= m - 1;
                    const int o2 = __builtin_ctz(m);                                   // its (even) bit
                    if ((fM >> o2) & 1u) {
                        const int32_t ii = i - __popc(cI & ((1u << o2) - 1u));
                        const uint32_t *__restrict__ pk = (p.strand ? v.s.read_rc : v.s.read_pk) + v.s.read_woff[rd];
                        const uint8_t sym = code_sym((int)((pk[ii >> 4] >> (2 * (ii & 15))) & 3u));
                        al = sym == s.base[0] ? 0 : (sym == s.base[1] ? 1 : -1);
                    }
                    break;
                }
            }
            const uint64_t m0 = __ballot(al == 0), m1 = __ballot(al == 1);
            const uint64_t below = (1ull << lane) - 1ull;
            if (al == 0) vmap_qid[s.row_off + base0 + __popcll(m0 & below)] = v.rec_qid[r];
            if (al == 1) vmap_qid[s.row_off + s.count[0] + base1 + __popcll(m1 & below)] = v.rec_qid[r];
            base0 += __popcll(m0); base1 += __popcll(m1);
        }
    }
}

struct CallInfo {
    bool called;
    uint8_t ord[4];
    uint32_t c[4];
    uint32_t total;
};

// phasing.py:103-120 for one position
__device__ __forceinline__ CallInfo evaluate_position(uint4 k, uint32_t oth) {
    CallInfo ci;
    uint32_t c[4] = {k.x, k.y, k.z, k.w};
    int distinct = (c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0) + (oth == 0u ? 0 : ((oth & 0x100u) ? 2 : 1));
    ci.total = c[0] + c[1] + c[2] + c[3];
    ci.called = false;
    // rank: count descending, ties larger letter first (sort ascending then reverse, phasing.py:116-117)
    uint64_t key[4];
#pragma unroll
    for (int i = 0; i < 4; i++) key[i] = ((uint64_t)c[i] << 2) | (uint64_t)i;
#define CSWAP(a, b) { uint64_t lo = key[a] < key[b] ? key[a] : key[b], hi = key[a] < key[b] ? key[b] : key[a]; key[a] = hi; key[b] = lo; }
    CSWAP(0, 1) CSWAP(2, 3) CSWAP(0, 2) CSWAP(1, 3) CSWAP(1, 2)
#undef CSWAP
#pragma unroll
    for (int i = 0; i < 4; i++) { ci.ord[i] = (uint8_t)(key[i] & 3); ci.c[i] = (uint32_t)(key[i] >> 2); }
    if (distinct < 2 || ci.total < 10) return ci;
    // p0 < 0.75 and p1 > 0.25 in IEEE double  <=>  4*c0 < 3*total and 4*c1 > total  (exact for total < 2^50)
    ci.called = (4ull * ci.c[0] < 3ull * ci.total) && (4ull * ci.c[1] > (uint64_t)ci.total);
    return ci;
}

// ---- K2b: per-position call flags + variant_map row counts ----------------------------------
__device__ __forceinline__ int find_ctg(const int64_t *goff, int n_ctg, int64_t g) {
    int lo = 0, hi = n_ctg;   // largest c with goff[c] <= g
    while (hi - lo > 1) {
        int m = (lo + hi) >> 1;
        if (goff[m] <= g) lo = m; else hi = m;
    }
    return lo;
}

// Called sites are sparse (one per few thousand positions), so the ordered compaction does not scan per-position
// arrays: k_site_flag leaves one flag byte per position and, per 256-position block, the number of sites and of
// variant_map rows; the two short block arrays are scanned; k_site_emit ranks the sites inside their block.
__global__ void __launch_bounds__(256) k_site_flag(const uint4 *__restrict__ cnt, const uint32_t *__restrict__ oth, int64_t n_pos, const uint8_t *__restrict__ blk_live,
                                                   uint8_t *__restrict__ flag8, uint32_t *__restrict__ blk_sites, uint32_t *__restrict__ blk_rows) {
    __shared__ uint32_t ws[4], wr[4];
    if (!blk_live[blockIdx.x]) {                      // no record reaches this block's tile: nothing to read, nothing called
        if (threadIdx.x == 0) { blk_sites[blockIdx.x] = 0u; blk_rows[blockIdx.x] = 0u; }
        return;
    }
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool called = false;
    uint32_t rows = 0;
    if (g < n_pos) {
        const CallInfo ci = evaluate_position(cnt[g], oth[g]);
        called = ci.called;
        rows = ci.called ? ci.c[0] + ci.c[1] : 0u;
        flag8[g] = called ? 1 : 0;
    }
    const uint32_t ns = (uint32_t)__popcll(__ballot(called));
    const uint32_t nr = (uint32_t)wave_sum_i32_dpp((int32_t)rows);
    if (lane_id() == 0) { ws[threadIdx.x >> 6] = ns; wr[threadIdx.x >> 6] = nr; }
    __syncthreads();
    if (threadIdx.x == 0) { blk_sites[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3]; blk_rows[blockIdx.x] = wr[0] + wr[1] + wr[2] + wr[3]; }
}
__global__ void __launch_bounds__(256) k_site_emit(const uint4 *__restrict__ cnt, const uint32_t *__restrict__ oth, const uint8_t *__restrict__ flag8,
                                                   const uint32_t *__restrict__ blk_sites_off, const uint32_t *__restrict__ blk_rows_off, const uint8_t *__restrict__ ref,
                                                   const int64_t *__restrict__ goff, int n_ctg, int64_t n_pos, fzp_site *__restrict__ sites,
                                                   int64_t *__restrict__ site_g, int32_t *__restrict__ site_ctg, const uint8_t *__restrict__ blk_live) {
    __shared__ uint32_t ws[4], wr[4];
    if (!blk_live[blockIdx.x]) return;                // its flags were never written
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const bool called = g < n_pos && flag8[g];
    const uint64_t bal = __ballot(called);
    if (__syncthreads_or(called ? 1 : 0) == 0) return;                 // nothing called in this block (the usual case)
    CallInfo ci;
    uint32_t rows = 0;
    if (called) { ci = evaluate_position(cnt[g], oth[g]); rows = ci.c[0] + ci.c[1]; }
    const uint32_t incl = wave_incl_scan_u32(rows);
    if (lane == 63) wr[wv] = incl;
    if (lane == 0) ws[wv] = (uint32_t)__popcll(bal);
    __syncthreads();
    if (!called) return;
    uint32_t si = blk_sites_off[blockIdx.x] + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull)), ro = blk_rows_off[blockIdx.x] + incl - rows;
    for (int w = 0; w < wv; w++) { si += ws[w]; ro += wr[w]; }
    const int c = find_ctg(goff, n_ctg, g);
    fzp_site s;
    s.pos = (int32_t)(g - goff[c]);
    s.ref_base = ref[g];
#pragma unroll
    for (int i = 0; i < 4; i++) { s.base[i] = code_sym(ci.ord[i]); s.count[i] = (int32_t)ci.c[i]; }
    s.pad_[0] = s.pad_[1] = s.pad_[2] = 0;
    s.total = (int32_t)ci.total;
    s.row_off = (int64_t)ro;
    sites[si] = s;
    site_g[si] = g;
    site_ctg[si] = c;
}

// the same from the per-block site counts' exclusive scan: a contig starts on a block boundary, so the sites before it are the sites before its first block (r5: known before
// the sites themselves are written, i.e. in time for the fetch that brings the site count over)
__global__ void k_site_begin_blk(const uint32_t *__restrict__ site_idx_scan, int64_t n_blk, const int64_t *__restrict__ goff, int n_ctg, const uint64_t *__restrict__ n_sites_dev,
                                 int64_t *__restrict__ site_begin) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_ctg) return;
    const int64_t blk = goff[c] >> 8;
    site_begin[c] = blk < n_blk ? (int64_t)site_idx_scan[blk] : (int64_t)*n_sites_dev;
}

// ---- K2d: scatter (record index, q_id) of every column that carries a called site's major/minor allele
// variant_map rows (phasing.py:125-128), site-centric: one wave per called site.  The records covering the site are
// a contiguous candidate range (POS-sorted, bounded by the contig's longest reference span); lane = candidate: it finds
// the record's symbol at the site through the 64-op CIGAR checkpoints (one binary search + a walk of at most one
// chunk), and the rows of each allele come out in record order by ballot ranks -- no atomics, no sort, and the
// 581 M columns of the batch are not expanded a second time for ~1 M rows.
__global__ void __launch_bounds__(256) k_vmap_sites(RecView v, int64_t n_sites, const fzp_site *__restrict__ sites, const int32_t *__restrict__ site_ctg,
                                                    const int64_t *__restrict__ ctg_rec_begin, const int32_t *__restrict__ ctg_maxspan,
                                                    const int32_t *__restrict__ rec_span, const int64_t *__restrict__ ck_off, const int32_t *__restrict__ ck_ref,
                                                    const int32_t *__restrict__ ck_q, int32_t *__restrict__ vmap_qid) {
    const int lane = lane_id();
    for (int64_t si = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); si < n_sites; si += (int64_t)gridDim.x * 4) {
        const fzp_site s = sites[si];
        const int c = site_ctg[si];
        const int32_t pos = s.pos;
        const int64_t rb = ctg_rec_begin[c], re = ctg_rec_begin[c + 1];
        int64_t lo, hi;
        {
            const int32_t ms = ctg_maxspan[c];
            int64_t a = rb, b = re;
            while (a < b) { const int64_t m = (a + b) >> 1; if (v.rec_pos[m] <= pos - ms) a = m + 1; else b = m; }
            lo = a;
            b = re;
            while (a < b) { const int64_t m = (a + b) >> 1; if (v.rec_pos[m] <= pos) a = m + 1; else b = m; }
            hi = a;
        }
        int32_t base0 = 0, base1 = 0;                      // rows of each allele written so far
        for (int64_t r0 = lo; r0 < hi; r0 += 64) {
            const int64_t r = r0 + lane;
            int al = -1;
            if (r < hi) {
                const int32_t rel = pos - v.rec_pos[r];
                if (rel < rec_span[r]) {
                    const int64_t k0 = ck_off[r];
                    int32_t ca = 0, cb = (int32_t)(ck_off[r + 1] - k0);
                    while (cb - ca > 1) { const int32_t m = (ca + cb) >> 1; if (ck_ref[k0 + m] <= rel) ca = m; else cb = m; }
                    int32_t rp = ck_ref[k0 + ca], qp = ck_q[k0 + ca];
                    const int64_t c1 = v.cig_off[r + 1];
                    for (int64_t k = v.cig_off[r] + (int64_t)ca * 64; k < c1; k++) {
                        const uint32_t w = v.cigar[k], len = w >> 4, t = w & 15u;
                        const bool isM = (t == FZP_OP_M) | (t == FZP_OP_EQ) | (t == FZP_OP_X);
                        if (isM | (t == FZP_OP_D)) {
                            if (rel < rp + (int32_t)len) {
                                if (isM) {
                                    const uint8_t sym = v.seq[v.seq_off[r] + qp + (rel - rp)];
                                    al = sym == s.base[0] ? 0 : (sym == s.base[1] ? 1 : -1);
                                }
                                break;
                            }
                            rp += (int32_t)len;
                        }
                        if (isM | (t == FZP_OP_I) | (t == FZP_OP_S)) qp += (int32_t)len;
                    }
                }
            }
            const uint64_t m0 = __ballot(al == 0), m1 = __ballot(al == 1);
            const uint64_t below = (1ull << lane) - 1ull;
            if (al == 0) vmap_qid[s.row_off + base0 + __popcll(m0 & below)] = v.rec_qid[r];
            if (al == 1) vmap_qid[s.row_off + s.count[0] + base1 + __popcll(m1 & below)] = v.rec_qid[r];
            base0 += __popcll(m0); base1 += __popcll(m1);
        }
    }
}

// ---- K3a: per site, the two alleles' DISTINCT q_ids (set semantics of phasing.py:189), ascending,
//           alleles in CPython-2.7 dict order A < C < T < G (phasing.py:175,181)
__global__ void __launch_bounds__(256) k_site_sets(const fzp_site *__restrict__ sites, int64_t n_sites, const int32_t *__restrict__ vq,
                                                   int32_t *__restrict__ setq, uint32_t *__restrict__ set_n, uint32_t *__restrict__ set_off) {
    const int lane = lane_id();
    int64_t w = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (w >= n_sites * 2) return;
    const int64_t si = w >> 1;
    const fzp_site &s = sites[si];
    const int x = (int)(w & 1);
    const int a0 = py2_rank(s.base[0]) < py2_rank(s.base[1]) ? 0 : 1;   // ranked allele that comes first in dict order
    const int a = x == 0 ? a0 : 1 - a0;
    const int32_t *src = vq + s.row_off + (a ? s.count[0] : 0);
    const int n = s.count[a];
    const int64_t doff = s.row_off + (x == 0 ? 0 : s.count[a0]);
    int32_t *dst = setq + doff;
    // 1) rank sort by (q_id, index)
    for (int base = 0; base < n; base += 64) {
        int idx = base + lane;
        int32_t my = idx < n ? src[idx] : 0x7fffffff;
        int rank = 0;
        for (int j0 = 0; j0 < n; j0 += 64) {
            int32_t other = (j0 + lane < n) ? src[j0 + lane] : 0x7fffffff;
            int m = min(64, n - j0);
            for (int k = 0; k < m; k++) {
                int32_t o = __shfl(other, k, 64);
                rank += (o < my || (o == my && j0 + k < idx)) ? 1 : 0;
            }
        }
        if (idx < n) dst[rank] = my;
    }
    __threadfence_block();
    // 2) in-place unique compaction, 64 at a time (writes never pass unread data)
    int out = 0;
    int32_t prev = -1;   // q_ids are >= 0
    for (int base = 0; base < n; base += 64) {
        int idx = base + lane;
        int32_t q = idx < n ? dst[idx] : 0x7fffffff;
        int32_t left = __shfl_up(q, 1, 64);
        if (lane == 0) left = prev;
        bool keep = idx < n && (q != left);
        uint64_t mask = __ballot(keep);
        int before = __popcll(mask & ((1ull << lane) - 1ull));
        prev = __shfl(q, 63, 64);
        __threadfence_block();
        if (keep) dst[out + before] = q;
        out += __popcll(mask);
    }
    if (lane == 0) { set_n[w] = (uint32_t)out; set_off[w] = (uint32_t)doff; }
}

// ---- K3b: candidate partners within the 65 536 bp window (phasing.py:169) ----------------------
__global__ void __launch_bounds__(256) k_cand(const int64_t *__restrict__ site_g, const int32_t *__restrict__ site_ctg, const int64_t *__restrict__ site_begin,
                                              int64_t n_sites, uint32_t *__restrict__ cand_n, uint32_t *__restrict__ cap) {
    int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n_sites) return;
    int64_t end = site_begin[site_ctg[s] + 1];
    int64_t key = site_g[s] + 65536;   // partners with g2 <= key
    int64_t lo = s + 1, hi = end;      // first index with site_g > key
    while (lo < hi) {
        int64_t m = (lo + hi) >> 1;
        if (site_g[m] <= key) lo = m + 1; else hi = m;
    }
    uint32_t n = (uint32_t)(lo - (s + 1));
    cand_n[s] = n;
    cap[s] = n < 501u ? n : 501u;
}

__device__ __forceinline__ bool set_contains(const int32_t *__restrict__ a, int n, int32_t q) {
    int lo = 0, hi = n;
    while (lo < hi) {
        int m = (lo + hi) >> 1;
        int32_t v = a[m];
        if (v < q) lo = m + 1; else hi = m;
    }
    return lo < n && a[lo] == q;
}

// One wave per site i1; partners are visited in ascending order because the 501-row cap
// (phasing.py:204-206) depends on how many earlier partners were kept.
__global__ void __launch_bounds__(256) k_assoc(int64_t n_sites, const uint32_t *__restrict__ cand_n, const uint32_t *__restrict__ cap_off,
                                               const int32_t *__restrict__ setq, const uint32_t *__restrict__ set_n, const uint32_t *__restrict__ set_off,
                                               fzp_arow *__restrict__ tmp, uint32_t *__restrict__ nkept) {
    const int lane = lane_id();
    for (int64_t i1 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6; i1 < n_sites; i1 += ((int64_t)gridDim.x * 256) >> 6) {
        const uint32_t nc = cand_n[i1];
        const int32_t *s1a = setq + set_off[2 * i1], *s1b = setq + set_off[2 * i1 + 1];
        const int n1a = (int)set_n[2 * i1], n1b = (int)set_n[2 * i1 + 1];
        fzp_arow *out = tmp + cap_off[i1];
        uint32_t kept = 0;
        // q_id range of site i1's reads (the sets are sorted): a partner whose range does not meet it shares no read with i1, all four counts
        // are 0 and the row is never kept (phasing.py:192) -- 64 partners are tested at once, lane per partner, and only the others are
        // intersected.  (q_ids follow POS, so at 15 kb reads three quarters of the partners inside the 65 536 bp window go this way.)
        const int32_t lo1 = min(n1a ? s1a[0] : 0x7fffffff, n1b ? s1b[0] : 0x7fffffff), hi1 = max(n1a ? s1a[n1a - 1] : -1, n1b ? s1b[n1b - 1] : -1);
        uint64_t todo = 0;
        for (uint32_t k = 0; k < nc && kept < 501u; k++) {
            if ((k & 63u) == 0) {
                const uint32_t kk = k + (uint32_t)lane;
                bool meet = false;
                if (kk < nc) {
                    const int64_t j = i1 + 1 + kk;
                    const int na = (int)set_n[2 * j], nb = (int)set_n[2 * j + 1];
                    const int32_t *pa = setq + set_off[2 * j], *pb = setq + set_off[2 * j + 1];
                    const int32_t lo2 = min(na ? pa[0] : 0x7fffffff, nb ? pb[0] : 0x7fffffff), hi2 = max(na ? pa[na - 1] : -1, nb ? pb[nb - 1] : -1);
                    meet = lo2 <= hi1 && lo1 <= hi2;
                }
                todo = __ballot(meet);
            }
            if (!((todo >> (k & 63u)) & 1ull)) continue;
            const int64_t i2 = i1 + 1 + k;
            const int32_t *s2a = setq + set_off[2 * i2], *s2b = setq + set_off[2 * i2 + 1];
            const int n2a = (int)set_n[2 * i2], n2b = (int)set_n[2 * i2 + 1];
            const int m = n2a + n2b;
            int n11 = 0, n12 = 0, n21 = 0, n22 = 0;
            for (int e0 = 0; e0 < m; e0 += 64) {
                int e = e0 + lane;
                bool act = e < m;
                bool y = e >= n2a;
                int32_t q = act ? (y ? s2b[e - n2a] : s2a[e]) : -1;
                bool ia = act && set_contains(s1a, n1a, q);
                bool ib = act && set_contains(s1b, n1b, q);
                n11 += __popcll(__ballot(ia && !y));
                n12 += __popcll(__ballot(ia && y));
                n21 += __popcll(__ballot(ib && !y));
                n22 += __popcll(__ballot(ib && y));
            }
            if (n11 + n12 + n21 + n22 >= 6) {   // phasing.py:192
                if (lane == 0) {
                    fzp_arow r;
                    r.site1 = (int32_t)i1; r.site2 = (int32_t)i2;
                    r.n[0] = n11; r.n[1] = n12; r.n[2] = n21; r.n[3] = n22;
                    out[kept] = r;
                }
                kept++;
            }
        }
        if (lane == 0) nkept[i1] = kept;
    }
}

__global__ void __launch_bounds__(256) k_assoc_compact(int64_t n_sites, const uint32_t *__restrict__ cap_off, const uint32_t *__restrict__ nkept,
                                                       const uint32_t *__restrict__ kept_off, const fzp_arow *__restrict__ tmp, fzp_arow *__restrict__ out) {
    const int lane = lane_id();
    int64_t i1 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (i1 >= n_sites) return;
    uint32_t n = nkept[i1];
    const fzp_arow *src = tmp + cap_off[i1];
    fzp_arow *dst = out + kept_off[i1];
    for (uint32_t k = lane; k < n; k += 64) dst[k] = src[k];
}

__global__ void k_arow_begin(const uint32_t *__restrict__ kept_off, const int64_t *__restrict__ site_begin, int n_ctg, int64_t n_sites, const uint64_t *__restrict__ n_arows_dev,
                             int64_t *__restrict__ arow_begin) {      // (the total from where the scan left it: the kernel runs before the host knows it)
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_ctg) return;
    int64_t s = site_begin[c];
    arow_begin[c] = s >= n_sites ? (n_arows_dev ? (int64_t)*n_arows_dev : 0) : (int64_t)kept_off[s];
}

inline unsigned grid_for(int64_t items, int per_block, int64_t cap = 1 << 20) {
    int64_t g = (items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

RecView rec_view(const fzp_batch *b) {
    RecView v;
    v.rec_pos = b->rec_pos.p; v.rec_qid = b->rec_qid.p; v.rec_ctg = b->rec_ctg.p;
    v.cig_off = b->cig_off.p; v.seq_off = b->seq_off.p; v.cigar = b->cigar.p; v.seq = b->seq.p;
    v.ctg_goff = b->ctg_goff.p; v.ctg_limit = b->ctg_limit.p; v.n_rec = b->n_rec;
    return v;
}


}  // namespace

// the 64-op checkpoints of the run-length records (+ every record's reference span and the contigs' longest one)
static int k2_checkpoints(fzp_ctx *ctx, fzp_batch *b) {
    hipStream_t st = ctx->stream;
    FZP_TRY(b->ck_ref.alloc((size_t)b->n_ck)); FZP_TRY(b->ck_q.alloc((size_t)b->n_ck));
    FZP_TRY(b->rec_span.alloc((size_t)b->n_rec)); FZP_TRY(b->ctg_maxspan.alloc((size_t)b->n_ctg));
    FZP_TRY(b->ctg_maxspan.zero((size_t)b->n_ctg, st));
    ProfScope ps(ctx, "k2_cig_ckpt");
    hipLaunchKernelGGL(k_cig_ckpt, dim3(grid_for(b->n_rec, 4, 1 << 16)), dim3(256), 0, st, rec_view(b), b->ck_off.p, b->ck_ref.p, b->ck_q.p, b->rec_span.p, b->ctg_maxspan.p);
    return FZP_OK;
}
// A batch that came from fzp_align_to_batch holds K1's packed records; whoever needs the run-length CIGAR words, the byte SEQ and their checkpoints (K6: its tally walks
// the D / I ops) asks here, once.
int fzp_batch_need_bytes(fzp_ctx *ctx, fzp_batch *b) {
    if (!b->packed || b->have_bytes) return FZP_OK;
    FZP_TRY(fzp_batch_source_ok(b));
    if (!b->make_bytes) { fzp_set_error("packed batch without a source job"); return FZP_EINVAL; }
    FZP_TRY(b->make_bytes(ctx, b));
    if (b->n_rec > 0 && b->n_pos > 0) FZP_TRY(k2_checkpoints(ctx, b));
    b->have_bytes = true;
    return FZP_OK;
}

namespace {
__global__ void __launch_bounds__(256) k_tile_tables(const int64_t *__restrict__ ctg_goff, int n_ctg, int64_t n_tiles, int32_t *__restrict__ tile_ctg, int32_t *__restrict__ tile_start) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_tiles) return;
    const int64_t g = t * PILE_TILE;
    int lo = 0, hi = n_ctg;                       // the last contig whose first padded position is <= g (empty contigs in front of it share that position and lose)
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ctg_goff[mid] <= g) lo = mid; else hi = mid; }
    tile_ctg[t] = lo;
    tile_start[t] = (int32_t)(g - ctg_goff[lo]);
}
}  // namespace

// ================================================================================ K2 driver
int fzp_k2_het_call(fzp_ctx *ctx, fzp_batch *b) {
    hipStream_t st = ctx->stream;
    const int64_t np = b->n_pos;
    FZP_TRY(b->totals.alloc(8));
    FZP_TRY(b->cnt.alloc((size_t)np * 4));
    FZP_TRY(b->oth.alloc((size_t)np));
    FZP_TRY(b->flag8.alloc((size_t)np));
    const size_t nblk = (size_t)((np + 255) / 256);      // np is a multiple of FZP_POS_TILE: nblk a multiple of 8
    FZP_TRY(b->blk_live.alloc(nblk + 8));
    FZP_TRY(b->blk_live.zero(nblk + 8, st));
    FZP_TRY(b->site_idx.alloc(nblk));      // per 256-position block: sites, then their exclusive scan
    FZP_TRY(b->row_off32.alloc(nblk));     // per block: variant_map rows, then their exclusive scan
    const bool packed = b->packed && !b->have_bytes && getenv("FZP_K2_BYTES") == nullptr;      // (FZP_K2_BYTES: the run-length path on a packed batch, for A/B runs and the parity test)
    if (b->packed && !packed) FZP_TRY(fzp_batch_need_bytes(ctx, b));
    RecView v = rec_view(b);
    if (b->n_rec > 0 && np > 0) {
        if (!b->packed) FZP_TRY(k2_checkpoints(ctx, b));        // (packed batches: spans from K1's summaries, checkpoints made by fzp_batch_need_bytes when the bytes are)
        // tiles never span contigs: a contig's padded length is whole tiles, so tile T is padded position T * PILE_TILE, and its contig the one whose range holds it.  Made on
        // the device (r5: the host loop over 49 000 tiles and the two uploads from pageable memory were 0.15 ms in which the GPU had nothing to do)
        const int64_t n_tiles = np / PILE_TILE;
        FZP_TRY(b->tile_ctg.alloc((size_t)n_tiles)); FZP_TRY(b->tile_start.alloc((size_t)n_tiles));
        hipLaunchKernelGGL(k_tile_tables, dim3(grid_for(n_tiles, 256, 1 << 30)), dim3(256), 0, st, b->ctg_goff.p, b->n_ctg, n_tiles, b->tile_ctg.p, b->tile_start.p);
        {
            ProfScope ps(ctx, "k2_pileup_count");
            if (packed)
                hipLaunchKernelGGL(k_pileup_pk, dim3((unsigned)n_tiles), dim3(PILE_THREADS), 0, st, pk_view(b), b->tile_ctg.p, b->tile_start.p, b->ctg_rec_begin.p, b->ctg_maxspan.p,
                                   b->rec_span.p, b->cnt.p, b->oth.p, (unsigned long long *)b->blk_live.p);
            else
                hipLaunchKernelGGL(k_pileup_tiles, dim3((unsigned)n_tiles), dim3(PILE_THREADS), 0, st, v, b->tile_ctg.p, b->tile_start.p, b->ctg_rec_begin.p, b->ctg_maxspan.p,
                                   b->rec_span.p, b->ck_off.p, b->ck_ref.p, b->ck_q.p, b->cnt.p, b->oth.p, (unsigned long long *)b->blk_live.p);
        }
    }                                                   // (no records: every block stays dead)
    if (np > 0) {
        ProfScope ps(ctx, "k2_site_flag");
        hipLaunchKernelGGL(k_site_flag, dim3(grid_for(np, 256, 1 << 30)), dim3(256), 0, st, (const uint4 *)b->cnt.p, b->oth.p, np, b->blk_live.p,
                           b->flag8.p, b->site_idx.p, b->row_off32.p);
    }
    FZP_TRY(fzp_exclusive_scan_u32(ctx, b->site_idx.p, b->site_idx.p, nblk, b->totals.p + 0));
    FZP_TRY(fzp_exclusive_scan_u32(ctx, b->row_off32.p, b->row_off32.p, nblk, b->totals.p + 1));
    uint64_t tot[2];
    FZP_TRY(b->site_begin.alloc((size_t)b->n_ctg + 1));
    b->h_site_begin.resize((size_t)b->n_ctg + 1);
    hipLaunchKernelGGL(k_site_begin_blk, dim3((b->n_ctg + 1 + 63) / 64), dim3(64), 0, st, b->site_idx.p, (int64_t)nblk, b->ctg_goff.p, b->n_ctg, b->totals.p, b->site_begin.p);
    FZP_TRY(fzp_fetch_with_begins(ctx, st, tot, b->totals.p, 2, b->h_site_begin.data(), b->site_begin.p, b->n_ctg));
    if (tot[1] >= (1ull << 31)) { fzp_set_error("variant_map has %llu rows (> 2^31)", (unsigned long long)tot[1]); return FZP_EINVAL; }
    b->n_sites = (int64_t)tot[0];
    b->n_rows = (int64_t)tot[1];
    FZP_TRY(b->sites.alloc((size_t)b->n_sites));
    FZP_TRY(b->site_g.alloc((size_t)b->n_sites));
    FZP_TRY(b->site_ctg.alloc((size_t)b->n_sites));
    FZP_TRY(b->vmap_qid.alloc((size_t)b->n_rows));
    if (b->n_sites > 0) {
        {
            ProfScope ps(ctx, "k2_site_emit");
            hipLaunchKernelGGL(k_site_emit, dim3(grid_for(np, 256, 1 << 30)), dim3(256), 0, st, (const uint4 *)b->cnt.p, b->oth.p, b->flag8.p,
                               b->site_idx.p, b->row_off32.p, b->ref.p, b->ctg_goff.p, b->n_ctg, np, b->sites.p, b->site_g.p, b->site_ctg.p, b->blk_live.p);
        }
        {
            ProfScope ps(ctx, "k2_vmap_scatter");
            if (packed)
                hipLaunchKernelGGL(k_vmap_pk, dim3(grid_for(b->n_sites, 4, 1 << 16)), dim3(256), 0, st, pk_view(b), b->n_sites, b->sites.p, b->site_ctg.p, b->ctg_rec_begin.p,
                                   b->ctg_maxspan.p, b->rec_span.p, b->vmap_qid.p);
            else
                hipLaunchKernelGGL(k_vmap_sites, dim3(grid_for(b->n_sites, 4, 1 << 16)), dim3(256), 0, st, v, b->n_sites, b->sites.p, b->site_ctg.p, b->ctg_rec_begin.p,
                                   b->ctg_maxspan.p, b->rec_span.p, b->ck_off.p, b->ck_ref.p, b->ck_q.p, b->vmap_qid.p);
        }
    }
    FZP_HIP(hipGetLastError());
    for (int c = 0; c < b->n_ctg; c++) {     // a site called beyond the contig's end: the reference dies on ref_seq[pos] (phasing.py:124)
        if ((int64_t)b->h_limit[(size_t)c] <= b->h_ref_len[(size_t)c] || b->h_site_begin[(size_t)c + 1] == b->h_site_begin[(size_t)c]) continue;
        int64_t g = 0;
        FZP_TRY(fzp_fetch(ctx, st, &g, b->site_g.p + b->h_site_begin[(size_t)c + 1] - 1, 8));
        if (g - b->h_goff[(size_t)c] >= b->h_ref_len[(size_t)c]) {
            fzp_set_error("contig %d: het site called at position %lld, beyond the contig end %lld (reference: IndexError on ref_seq[pos])", c,
                          (long long)(g - b->h_goff[(size_t)c] + 1), (long long)b->h_ref_len[(size_t)c]);
            return FZP_EINVAL;
        }
    }
    b->have_sites = true;
    b->have_sets = false;
    return FZP_OK;
}

// ================================================================================ K3 drivers
int fzp_k3_sets(fzp_ctx *ctx, fzp_batch *b) {
    hipStream_t st = ctx->stream;
    if (!b->have_sites) { fzp_set_error("association table needs het-call sites"); return FZP_EINVAL; }
    FZP_TRY(b->setq.alloc((size_t)b->n_rows));
    FZP_TRY(b->set_n.alloc((size_t)b->n_sites * 4));   // [0,2n): set_n, [2n,4n): set_off
    if (b->n_sites > 0) {
        ProfScope ps(ctx, "k3_site_sets");
        hipLaunchKernelGGL(k_site_sets, dim3(grid_for(b->n_sites * 2, 4, 1 << 30)), dim3(256), 0, st, b->sites.p, b->n_sites, b->vmap_qid.p,
                           b->setq.p, b->set_n.p, b->set_n.p + 2 * b->n_sites);
    }
    FZP_HIP(hipGetLastError());
    b->have_sets = true;
    return FZP_OK;
}

int fzp_k3_assoc(fzp_ctx *ctx, fzp_batch *b) {
    hipStream_t st = ctx->stream;
    if (!b->have_sets) FZP_TRY(fzp_k3_sets(ctx, b));
    const int64_t ns = b->n_sites;
    FZP_TRY(b->totals.alloc(8));
    FZP_TRY(b->cand_n.alloc((size_t)ns));
    FZP_TRY(b->cap_off.alloc((size_t)ns));
    FZP_TRY(b->nkept.alloc((size_t)ns));
    FZP_TRY(b->kept_off.alloc((size_t)ns));
    FZP_TRY(b->arow_begin.alloc((size_t)b->n_ctg + 1));
    uint64_t tot[1] = {0};
    if (ns > 0) {
        hipLaunchKernelGGL(k_cand, dim3(grid_for(ns, 256, 1 << 30)), dim3(256), 0, st, b->site_g.p, b->site_ctg.p, b->site_begin.p, ns, b->cand_n.p, b->cap_off.p);
        FZP_TRY(fzp_exclusive_scan_u32(ctx, b->cap_off.p, b->cap_off.p, (size_t)ns, b->totals.p + 2));
        // room for the candidate rows: a site has at most 501 of them (k_cand), so 501 per site is room enough and the host need not ask how many there are (r5: one
        // read-back less) -- while that bound is a small block: 8 M rows = 192 MB (the bench step: 16 000 sites).  Beyond that it asks (the bound is 1.5 GB per
        // context at 64 M rows whatever the real count, kept in the context's pool, and several lanes of a genome-scale job each hold one)
        const uint64_t bound = (uint64_t)ns * 501ull;
        if (bound <= (8ull << 20)) tot[0] = bound;
        else {
            FZP_TRY(fzp_fetch(ctx, st, tot, b->totals.p + 2, sizeof(uint64_t)));
            if (tot[0] >= (1ull << 31)) { fzp_set_error("association table bound %llu rows (> 2^31)", (unsigned long long)tot[0]); return FZP_EINVAL; }
        }
        FZP_TRY(b->arows_tmp.alloc((size_t)tot[0]));
        {
            ProfScope ps(ctx, "k3_assoc");
            hipLaunchKernelGGL(k_assoc, dim3(grid_for(ns, 4, 1 << 16)), dim3(256), 0, st, ns, b->cand_n.p, b->cap_off.p, b->setq.p, b->set_n.p,
                               b->set_n.p + 2 * ns, b->arows_tmp.p, b->nkept.p);
        }
        FZP_TRY(fzp_exclusive_scan_u32(ctx, b->nkept.p, b->kept_off.p, (size_t)ns, b->totals.p + 3));
        b->h_arow_begin.resize((size_t)b->n_ctg + 1);
        hipLaunchKernelGGL(k_arow_begin, dim3((b->n_ctg + 1 + 63) / 64), dim3(64), 0, st, b->kept_off.p, b->site_begin.p, b->n_ctg, ns, b->totals.p + 3, b->arow_begin.p);
        FZP_TRY(fzp_fetch_with_begins(ctx, st, tot, b->totals.p + 3, 1, b->h_arow_begin.data(), b->arow_begin.p, b->n_ctg));
        b->n_arows = (int64_t)tot[0];
        FZP_TRY(b->arows.alloc((size_t)b->n_arows));
        {
            ProfScope ps(ctx, "k3_assoc_compact");
            hipLaunchKernelGGL(k_assoc_compact, dim3(grid_for(ns, 4, 1 << 30)), dim3(256), 0, st, ns, b->cap_off.p, b->nkept.p, b->kept_off.p,
                               b->arows_tmp.p, b->arows.p);
        }
    } else {
        b->n_arows = 0;
        FZP_TRY(b->arows.alloc(0));
        b->h_arow_begin.assign((size_t)b->n_ctg + 1, 0);
        FZP_TRY(b->arow_begin.zero((size_t)b->n_ctg + 1, st));
    }
    FZP_HIP(hipGetLastError());
    b->have_arows = true;
    return FZP_OK;
}
