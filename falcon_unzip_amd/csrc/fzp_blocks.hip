// fzp_blocks.hip -- K4: phase-block assignment (get_score + get_phased_blocks, phasing.py:208-421)
//                   K5: read phasing vote (get_phased_reads, phasing.py:423-480)
//
// K4 restated for the GPU (derivations in DESIGN.md section 5):
//  * a site's state is one orientation bit o[s] relative to the atable's A<C<T<G allele order;
//    get_score(link, s1, s2) == cis if o[s1]==o[s2] else trans.
//  * the streaming greedy initialisation (phasing.py:259-309) decides every site from exactly ONE
//    link -- the link of its first appearance -- so it is a forest: o[s] = o[parent] ^ (cis<trans),
//    roots 0.  Resolved by pointer jumping (order-free, deterministic).
//  * the <=10 Gauss-Seidel sweeps (phasing.py:315-344) are order-dependent: sequential over the
//    sites of a contig, wave-parallel over a site's left links, contigs in parallel.
//  * extents are per-site reductions; block segmentation (phasing.py:388-408) is a prefix-max scan
//    because max_right_ext is never reset.
// All integer; outputs are bit-exact by construction, no atomics decide any order.
#include "fzp_batch.h"

namespace {

inline unsigned grid_for(int64_t items, int per_block, int64_t cap = 1 << 20) {
    int64_t g = (items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

constexpr uint32_t NOLINK = 0xffffffffu;

// ---- links: atable rows with |cis - trans| >= 6 (phasing.py:245)
__global__ void __launch_bounds__(256) k_link_flag(const fzp_arow *__restrict__ rows, int64_t n, uint32_t *__restrict__ flag) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int cis = rows[i].n[0] + rows[i].n[3], trans = rows[i].n[1] + rows[i].n[2];
    int d = cis - trans;
    flag[i] = (d >= 6 || d <= -6) ? 1u : 0u;
}

__global__ void __launch_bounds__(256) k_link_emit(const fzp_arow *__restrict__ rows, int64_t n, const uint32_t *__restrict__ idx, const uint64_t *__restrict__ n_links_dev,
                                                   int32_t *__restrict__ lk_i1, int32_t *__restrict__ lk_i2, int32_t *__restrict__ lk_cis, int32_t *__restrict__ lk_trans,
                                                   uint32_t *__restrict__ left_n, uint32_t *__restrict__ right_n, uint32_t *__restrict__ fr2) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t l = idx[i];
    uint32_t nxt = (i + 1 < n) ? idx[i + 1] : (uint32_t)*n_links_dev;      // (the number of links from where the scan left it: the host never asks -- they are at most the rows)
    if (nxt == l) return;   // not kept
    fzp_arow r = rows[i];
    lk_i1[l] = r.site1; lk_i2[l] = r.site2;
    lk_cis[l] = r.n[0] + r.n[3]; lk_trans[l] = r.n[1] + r.n[2];
    atomicAdd(&left_n[r.site2], 1u);
    atomicAdd(&right_n[r.site1], 1u);
    atomicMin(&fr2[r.site2], l);
}

__global__ void __launch_bounds__(256) k_left_fill(const uint64_t *__restrict__ n_links_dev, const int32_t *__restrict__ lk_i1, const int32_t *__restrict__ lk_i2,
                                                   const int32_t *__restrict__ lk_cis, const int32_t *__restrict__ lk_trans, const uint32_t *__restrict__ left_off,
                                                   uint32_t *__restrict__ left_fill, int32_t *__restrict__ left_lk, int4 *__restrict__ left_pk) {
    int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= (int64_t)*n_links_dev) return;
    int32_t s = lk_i2[l];
    uint32_t slot = atomicAdd(&left_fill[s], 1u);
    left_lk[left_off[s] + slot] = (int32_t)l;
    left_pk[left_off[s] + slot] = make_int4(lk_i1[l], lk_cis[l], lk_trans[l], s);   // the sweep reads the links of 64 sites as one run: (left site, cis, trans, the site itself)
}

// ---- greedy initialisation as a forest (phasing.py:259-309)
__global__ void __launch_bounds__(256) k_pj_init(int64_t n_sites, const uint32_t *__restrict__ right_n, const uint32_t *__restrict__ right_off,
                                                 const uint32_t *__restrict__ fr2, const int32_t *__restrict__ lk_i1, const int32_t *__restrict__ lk_i2,
                                                 const int32_t *__restrict__ lk_cis, const int32_t *__restrict__ lk_trans, uint32_t *__restrict__ pj) {
    int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n_sites) return;
    uint32_t f1 = right_n[s] ? right_off[s] : NOLINK;   // first link with this site as pos1
    uint32_t f2 = fr2[s];                                // first link with this site as pos2
    uint32_t parent = (uint32_t)s, flip = 0;
    if (f2 < f1) {                       // first seen as pos2: its only neighbour so far is that link's pos1
        parent = (uint32_t)lk_i1[f2];
        flip = lk_cis[f2] < lk_trans[f2] ? 1u : 0u;
    } else if (f1 != NOLINK) {           // first seen as pos1: the partner counts only if it already has a state
        uint32_t i2 = (uint32_t)lk_i2[f1];
        if (fr2[i2] < f1) {
            parent = i2;
            flip = lk_cis[f1] < lk_trans[f1] ? 1u : 0u;
        }
    }
    pj[s] = (parent << 1) | flip;
}

__global__ void __launch_bounds__(256) k_pj_resolve(int64_t n_sites, uint32_t *__restrict__ pj, uint8_t *__restrict__ orient) {
    int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n_sites) return;
    // invariant at all times: o[s] = o[parent(s)] ^ flip(s); every stored parent is an ancestor
    uint32_t w = __hip_atomic_load(&pj[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t a = w >> 1, f = w & 1u;
    for (;;) {
        if (a == (uint32_t)s) break;
        uint32_t wa = __hip_atomic_load(&pj[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t pa = wa >> 1;
        if (pa == a) break;              // a is a root (o = 0)
        f ^= (wa & 1u);
        a = pa;
        __hip_atomic_store(&pj[s], (a << 1) | f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    orient[s] = (uint8_t)f;
}

// ---- refinement sweeps (phasing.py:315-344): one wave per contig, 64 sites at a time
// The reference visits a contig's sites in order and flips a site when the links to its LEFT neighbours -- in their states of that moment, flips of this very sweep
// included -- score higher the other way: a serial chain over the sites (up to ten sweeps).  One site per step with the wave spread over its links (r2..r5) costs ~0.5 us a
// site in dependent accesses; a genome-scale group spends more time here than in the DP (configs[4], r5: 76 of 285 ms of kernels, ONE wave per contig).  This form takes 64
// consecutive sites per step and stays exact:
//   1. the links of the 64 sites are ONE run of the link array (CSR by site): the lanes read it coalesced, every link adds its score difference for its own site (the
//      link's fourth word) into that site's cell of a 64-entry LDS table -- d[p] = sum over p's left links of (same state ? cis - trans : trans - cis), with the states as
//      they are BEFORE the block;
//   2. the lowest site of the block with d < 0 is the first the reference would flip (every site before it kept its state, so what it saw was right).  It flips; the
//      links from it to LATER sites of the block (its "right" links, a run of the link arrays) change those sites' d by -2 x their contribution; then the next lowest
//      site with d < 0 behind it, and so on.  Sites behind the block see the new states when their block is summed.
// A block without flips -- nearly all of them after the first sweep -- costs its links once.
template <bool USE_LDS>
__global__ void __launch_bounds__(256) k_sweep(const int64_t *__restrict__ site_begin, const uint32_t *__restrict__ left_n, const uint32_t *__restrict__ left_off,
                                               const int4 *__restrict__ left_pk, const uint32_t *__restrict__ right_n, const uint32_t *__restrict__ right_off,
                                               const int32_t *__restrict__ lk_i2, const int32_t *__restrict__ lk_cis, const int32_t *__restrict__ lk_trans,
                                               uint8_t *__restrict__ orient) {
    // four waves per contig: the links of a block are summed by all 256 threads (that sum is nearly all of the work); every wave then reads the same 64 cells and takes the
    // same decisions, the flips' deltas are spread over the threads again
    extern __shared__ uint8_t sweep_lds[];
    int32_t *d = (int32_t *)sweep_lds;                    // 64 sites' score differences
    uint8_t *o_lds = sweep_lds + 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int c = blockIdx.x;
    const int64_t sb = site_begin[c], se = site_begin[c + 1];
    const int64_t n = se - sb;
    if (n <= 0) return;
    uint8_t *o = USE_LDS ? o_lds : (orient + sb);
    if (USE_LDS) {
        for (int64_t i = tid; i < n; i += 256) o_lds[i] = orient[sb + i];
        __syncthreads();
    }
    for (int iter = 1; iter <= 10; iter++) {
        int updates = 0;
        for (int64_t p0 = 0; p0 < n; p0 += 64) {
            const int nb = (int)min((int64_t)64, n - p0);
            const int64_t g0 = sb + p0;                   // the block's first site
            // 1. the block's links: [L0, L1) of the link array
            const uint32_t L0 = left_off[g0], L1 = left_off[g0 + nb - 1] + left_n[g0 + nb - 1];
            if (tid < 64) d[tid] = 0;
            __syncthreads();
            for (uint32_t j = L0 + (uint32_t)tid; j < L1; j += 1024) {      // four loads of a thread in flight
                int4 e[4];
#pragma unroll
                for (int u = 0; u < 4; u++) e[u] = j + 256u * u < L1 ? left_pk[j + 256u * u] : make_int4((int)g0, 0, 0, (int)g0);      // (left site, cis, trans, this site); beyond the run: adds 0
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const bool same = o[e[u].x - sb] == o[e[u].w - sb];
                    const int32_t cdiff = same ? e[u].y - e[u].z : e[u].z - e[u].y;
                    if (cdiff) atomicAdd(&d[e[u].w - (int32_t)g0], cdiff);
                }
            }
            __syncthreads();
            if (L1 == L0) continue;
            // 2. flips, lowest site first
            int start = 0;
            for (;;) {
                const int32_t dv = lane < nb ? d[lane] : 0;
                const uint64_t m = __ballot(lane >= start && dv < 0);      // (the same in every wave)
                if (!m) break;                            // score1 >= score2 keeps the state (phasing.py:338-342)
                const int F = __builtin_ctzll(m);
                const int64_t gF = g0 + F;
                const uint8_t oF = o[p0 + F];
                const uint32_t r0 = right_off[gF], rn = right_n[gF];
                __syncthreads();                          // every wave has read the cells before any of them changes
                for (uint32_t k = (uint32_t)tid; k < rn; k += 256) {      // what F's flip does to the later sites of this block
                    const int32_t t = lk_i2[r0 + k];
                    if (t < g0 + nb) {
                        const bool same = o[t - sb] == oF;
                        const int32_t cdiff = lk_cis[r0 + k] - lk_trans[r0 + k];
                        atomicAdd(&d[t - (int32_t)g0], same ? -2 * cdiff : 2 * cdiff);
                    }
                }
                __syncthreads();
                if (tid == 0) o[p0 + F] = oF ^ 1;
                updates++;
                if (!USE_LDS) __threadfence_block();
                __syncthreads();
                start = F + 1;
            }
        }
        if (updates == 0) break;
    }
    if (USE_LDS) {
        __syncthreads();
        for (int64_t i = tid; i < n; i += 256) orient[sb + i] = o_lds[i];
    }
}

// ---- per-site scores and extents (phasing.py:353-383): one wave per site
__global__ void __launch_bounds__(256) k_extents(int64_t n_sites, const fzp_site *__restrict__ sites, const uint32_t *__restrict__ left_n, const uint32_t *__restrict__ left_off,
                                                 const int32_t *__restrict__ left_lk, const uint32_t *__restrict__ right_n, const uint32_t *__restrict__ right_off,
                                                 const int32_t *__restrict__ lk_i1, const int32_t *__restrict__ lk_i2, const int32_t *__restrict__ lk_cis,
                                                 const int32_t *__restrict__ lk_trans, const uint8_t *__restrict__ orient, int32_t *__restrict__ lext,
                                                 int32_t *__restrict__ rext, int32_t *__restrict__ lscore, int32_t *__restrict__ rscore) {
    const int lane = lane_id();
    int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (p >= n_sites) return;
    const uint8_t op = orient[p];
    int sc = 0;
    int32_t ext = (int32_t)p;
    for (uint32_t k = lane; k < left_n[p]; k += 64) {
        int32_t l = left_lk[left_off[p] + k];
        int32_t pp = lk_i1[l];
        bool same = orient[pp] == op;
        int d = same ? lk_cis[l] - lk_trans[l] : lk_trans[l] - lk_cis[l];   // s - s_
        sc += d;
        if (d > 0 && pp < ext) ext = pp;
    }
    sc = wave_sum_i32(sc);
    ext = wave_min_i32(ext);
    if (lane == 0) { lscore[p] = sc; lext[p] = sites[ext].pos + 1; }
    sc = 0;
    ext = (int32_t)p;
    for (uint32_t k = lane; k < right_n[p]; k += 64) {
        int32_t l = (int32_t)(right_off[p] + k);
        int32_t pp = lk_i2[l];
        bool same = orient[pp] == op;
        int d = same ? lk_cis[l] - lk_trans[l] : lk_trans[l] - lk_cis[l];
        sc += d;
        if (d > 0 && pp > ext) ext = pp;
    }
    sc = wave_sum_i32(sc);
    ext = wave_max_i32(ext);
    if (lane == 0) { rscore[p] = sc; rext[p] = sites[ext].pos + 1; }
}

// ---- block segmentation (phasing.py:388-408) and 'V' records: one wave per contig
__global__ void __launch_bounds__(256) k_segment(const int64_t *__restrict__ site_begin, const fzp_site *__restrict__ sites, const uint8_t *__restrict__ orient,
                                                 const int32_t *__restrict__ lext, const int32_t *__restrict__ rext, const int32_t *__restrict__ lscore,
                                                 const int32_t *__restrict__ rscore, int32_t *__restrict__ rawblk, int32_t *__restrict__ blkcnt,
                                                 int32_t *__restrict__ blknew, fzp_pvar *__restrict__ pv_tmp, uint32_t *__restrict__ pv_n,
                                                 int32_t *__restrict__ site_blk, uint8_t *__restrict__ site_b1) {
    // (r5: 256 sites per step -- four waves whose running maxima and counts meet in LDS -- where one wave took 64; a genome-scale group spent 0.8 ms per call here on seven waves)
    __shared__ int32_t w_max[4], w_cnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x;
    const int64_t sb = site_begin[c], se = site_begin[c + 1];
    const int64_t n = se - sb;
    if (n <= 0) { if (tid == 0) pv_n[c] = 0; return; }
    // pass A: raw block ids.  max_right_ext is a running max over qualifying sites, never reset.
    int32_t carryM = 0, carryB = 0;
    for (int64_t base = 0; base < n; base += 256) {
        const int64_t s = sb + base + tid;
        const bool in = base + tid < n;
        const bool q = in && !(rscore[s] < 10 || lscore[s] < 10);
        const int32_t r = q ? rext[s] : 0;
        int32_t incl = r;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int32_t t = __shfl_up(incl, d, 64);
            if (lane >= d) incl = max(incl, t);
        }
        if (lane == 63) w_max[wave] = incl;
        __syncthreads();
        int32_t before = carryM, all = carryM;              // the running maximum before this wave's first site / behind the step's last
#pragma unroll
        for (int w = 0; w < 4; w++) { const int32_t v = w_max[w]; if (w < wave) before = max(before, v); all = max(all, v); }
        int32_t excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 0;
        excl = max(excl, before);
        const bool nb = q && (excl < lext[s]);
        const uint64_t m = __ballot(nb);
        if (lane == 0) w_cnt[wave] = __popcll(m);
        __syncthreads();
        int32_t cb = carryB, call = carryB;
#pragma unroll
        for (int w = 0; w < 4; w++) { const int32_t v = w_cnt[w]; if (w < wave) cb += v; call += v; }
        const int32_t id = cb + __popcll(m & ((2ull << lane) - 1ull));   // inclusive count
        if (in) rawblk[s] = q ? id : 0;
        if (q) atomicAdd(&blkcnt[sb + id - 1], 1);
        carryM = all;
        carryB = call;
        __syncthreads();                                     // (w_max / w_cnt are written again in the next step)
    }
    __threadfence_block();
    __syncthreads();
    // pass B: blocks with more than 3 variants get dense ids from 1 (phasing.py:398-408)
    int32_t carryI = 0;
    for (int32_t base = 0; base < carryB; base += 256) {
        const int32_t j = base + tid;
        const bool keep = j < carryB && blkcnt[sb + j] > 3;
        const uint64_t m = __ballot(keep);
        if (lane == 0) w_cnt[wave] = __popcll(m);
        __syncthreads();
        int32_t ci = carryI, call = carryI;
#pragma unroll
        for (int w = 0; w < 4; w++) { const int32_t v = w_cnt[w]; if (w < wave) ci += v; call += v; }
        if (j < carryB) blknew[sb + j] = keep ? ci + 1 + __popcll(m & ((1ull << lane) - 1ull)) : 0;
        carryI = call;
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    // pass C: 'V' records in site order
    uint32_t out = 0;
    for (int64_t base = 0; base < n; base += 256) {
        const int64_t s = sb + base + tid;
        const bool in = base + tid < n;
        const int32_t rb = in ? rawblk[s] : 0;
        const int32_t nid = rb > 0 ? blknew[sb + rb - 1] : 0;
        const bool has = nid > 0;
        const uint64_t m = __ballot(has);
        if (lane == 0) w_cnt[wave] = __popcll(m);
        __syncthreads();
        uint32_t at = out, call = out;
#pragma unroll
        for (int w = 0; w < 4; w++) { const uint32_t v = (uint32_t)w_cnt[w]; if (w < wave) at += v; call += v; }
        uint8_t b1 = 0, b2 = 0;
        if (in) {
            const fzp_site &st = sites[s];
            uint8_t x = st.base[0], y = st.base[1];
            if (py2_rank(y) < py2_rank(x)) { uint8_t t = x; x = y; y = t; }   // atable allele order
            if (orient[s]) { b1 = y; b2 = x; } else { b1 = x; b2 = y; }
            site_blk[s] = nid;
            site_b1[s] = b1;
        }
        if (has) {
            fzp_pvar v;
            v.block = nid; v.site = (int32_t)s; v.b1 = b1; v.b2 = b2; v.pad_[0] = v.pad_[1] = 0;
            v.lext = lext[s]; v.rext = rext[s]; v.lscore = lscore[s]; v.rscore = rscore[s];
            pv_tmp[sb + at + __popcll(m & ((1ull << lane) - 1ull))] = v;
        }
        out = call;
        __syncthreads();
    }
    if (tid == 0) pv_n[c] = out;
}

__global__ void __launch_bounds__(256) k_pv_compact(const int64_t *__restrict__ site_begin, const uint32_t *__restrict__ pv_n, const uint32_t *__restrict__ pv_off,
                                                    const fzp_pvar *__restrict__ tmp, fzp_pvar *__restrict__ out) {
    const int c = blockIdx.x;
    const fzp_pvar *src = tmp + site_begin[c];
    fzp_pvar *dst = out + pv_off[c];
    for (uint32_t k = threadIdx.x; k < pv_n[c]; k += 256) dst[k] = src[k];
}

__global__ void k_u32_to_i64_begin(const uint32_t *__restrict__ off, int n, const uint64_t *__restrict__ total_dev, int64_t *__restrict__ begin) {      // (the total from where the scan left it)
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n) return;
    begin[c] = c < n ? (int64_t)off[c] : (int64_t)*total_dev;
}

// ================================================================================ K5
__device__ __forceinline__ int find_ctg_q(const int64_t *qoff, int n_ctg, int64_t Q) {
    int lo = 0, hi = n_ctg;
    while (hi - lo > 1) {
        int m = (lo + hi) >> 1;
        if (qoff[m] <= Q) lo = m; else hi = m;
    }
    return lo;
}

// Every DISTINCT (read, site, allele) of a phased site votes (set semantics, phasing.py:448-449,469-473).
// MODE 0: per-read block range; MODE 1: per (read, block) phase counts.
template <int MODE>
__global__ void __launch_bounds__(256) k_read_votes(int64_t n_sites, const fzp_site *__restrict__ sites, const int32_t *__restrict__ site_ctg, const int64_t *__restrict__ ctg_qoff,
                                                    const int32_t *__restrict__ setq, const uint32_t *__restrict__ set_n, const uint32_t *__restrict__ set_off,
                                                    const int32_t *__restrict__ site_blk, const uint8_t *__restrict__ site_b1, int32_t *__restrict__ bmin,
                                                    int32_t *__restrict__ bmax, const uint32_t *__restrict__ rng_off, uint32_t *__restrict__ c0, uint32_t *__restrict__ c1) {
    const int lane = lane_id();
    int64_t w = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (w >= n_sites * 2) return;
    const int64_t s = w >> 1;
    const int32_t blk = site_blk[s];
    if (blk <= 0) return;
    const fzp_site &st = sites[s];
    uint8_t x = st.base[0], y = st.base[1];
    if (py2_rank(y) < py2_rank(x)) { uint8_t t = x; x = y; y = t; }
    const uint8_t allele = (w & 1) ? y : x;
    const int phase = allele == site_b1[s] ? 0 : 1;
    const int64_t qbase = ctg_qoff[site_ctg[s]];
    const int32_t *q = setq + set_off[w];
    for (uint32_t k = lane; k < set_n[w]; k += 64) {
        int64_t Q = qbase + q[k];
        if (MODE == 0) {
            atomicMin(&bmin[Q], blk);
            atomicMax(&bmax[Q], blk);
        } else {
            uint32_t slot = rng_off[Q] + (uint32_t)(blk - bmin[Q]);
            atomicAdd(phase == 0 ? &c0[slot] : &c1[slot], 1u);
        }
    }
}

__global__ void __launch_bounds__(256) k_range_n(int64_t nq, const int32_t *__restrict__ bmin, const int32_t *__restrict__ bmax, uint32_t *__restrict__ rng_n) {
    int64_t Q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (Q >= nq) return;
    rng_n[Q] = bmax[Q] > 0 ? (uint32_t)(bmax[Q] - bmin[Q] + 1) : 0u;
}

__global__ void __launch_bounds__(256) k_read_flag(int64_t n_slots, const uint32_t *__restrict__ c0, const uint32_t *__restrict__ c1, uint32_t *__restrict__ flag) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_slots) return;
    int d = (int)c0[i] - (int)c1[i];
    flag[i] = (d > 1 || d < -1) ? 1u : 0u;   // phasing.py:477-480
}

__global__ void __launch_bounds__(256) k_read_emit(int64_t n_slots, int64_t nq, int64_t n_out, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ rng_off,
                                                   const int32_t *__restrict__ bmin, const uint32_t *__restrict__ c0, const uint32_t *__restrict__ c1,
                                                   const int64_t *__restrict__ ctg_qoff, int n_ctg, fzp_pread *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_slots) return;
    uint32_t o = idx[i];
    uint32_t nxt = (i + 1 < n_slots) ? idx[i + 1] : (uint32_t)n_out;
    if (nxt == o) return;
    int64_t lo = 0, hi = nq;   // first Q with rng_off[Q] > i ; the owner is that - 1
    while (lo < hi) {
        int64_t m = (lo + hi) >> 1;
        if (rng_off[m] <= (uint32_t)i) lo = m + 1; else hi = m;
    }
    int64_t Q = lo - 1;
    int c = find_ctg_q(ctg_qoff, n_ctg, Q);
    fzp_pread r;
    r.q_id = (int32_t)(Q - ctg_qoff[c]);
    r.block = bmin[Q] + (int32_t)((uint32_t)i - rng_off[Q]);
    r.n0 = (int32_t)c0[i]; r.n1 = (int32_t)c1[i];
    r.phase = r.n0 > r.n1 ? 0 : 1;
    out[o] = r;
}

__global__ void k_pread_begin(const int64_t *__restrict__ ctg_qoff, int n_ctg, int64_t nq, int64_t n_slots, const uint64_t *__restrict__ n_out_dev, const uint32_t *__restrict__ rng_off,
                              const uint32_t *__restrict__ idx, int64_t *__restrict__ begin) {      // (n_out from where the scan left it; null: no slots, no records)
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_ctg) return;
    int64_t Q0 = ctg_qoff[c];
    int64_t v = n_out_dev ? (int64_t)*n_out_dev : 0;
    if (Q0 < nq) {
        uint32_t slot = rng_off[Q0];
        if ((int64_t)slot < n_slots) v = idx[slot];
    }
    begin[c] = v;
}

}  // namespace

// ================================================================================ K4 driver
int fzp_k4_blocks(fzp_ctx *ctx, fzp_batch *b) {
    hipStream_t st = ctx->stream;
    if (!b->have_sites || !b->have_arows) { fzp_set_error("phase blocks need sites and an association table"); return FZP_EINVAL; }
    const int64_t ns = b->n_sites, na = b->n_arows;
    FZP_TRY(b->totals.alloc(8));
    FZP_TRY(b->lk_flag.alloc((size_t)na));
    FZP_TRY(b->left_n.alloc((size_t)ns)); FZP_TRY(b->left_off.alloc((size_t)ns)); FZP_TRY(b->left_fill.alloc((size_t)ns));
    FZP_TRY(b->right_n.alloc((size_t)ns)); FZP_TRY(b->right_off.alloc((size_t)ns)); FZP_TRY(b->fr2.alloc((size_t)ns));
    FZP_TRY(b->pj.alloc((size_t)ns)); FZP_TRY(b->orient.alloc((size_t)ns));
    FZP_TRY(b->lext.alloc((size_t)ns)); FZP_TRY(b->rext.alloc((size_t)ns)); FZP_TRY(b->lscore.alloc((size_t)ns)); FZP_TRY(b->rscore.alloc((size_t)ns));
    FZP_TRY(b->rawblk.alloc((size_t)ns)); FZP_TRY(b->blkcnt.alloc((size_t)ns)); FZP_TRY(b->blknew.alloc((size_t)ns));
    FZP_TRY(b->pvars_tmp.alloc((size_t)ns)); FZP_TRY(b->site_blk.alloc((size_t)ns)); FZP_TRY(b->site_b1.alloc((size_t)ns));
    FZP_TRY(b->pv_n.alloc((size_t)b->n_ctg + 1)); FZP_TRY(b->pv_off.alloc((size_t)b->n_ctg + 1)); FZP_TRY(b->pvar_begin.alloc((size_t)b->n_ctg + 1));
    {
        const fzp_fill_piece fl[5] = {fzp_zeroes(b->left_n, (size_t)ns), fzp_zeroes(b->left_fill, (size_t)ns), fzp_zeroes(b->right_n, (size_t)ns), fzp_zeroes(b->blkcnt, (size_t)ns),
                                      fzp_ones(b->fr2, (size_t)ns)};
        FZP_TRY(fzp_fill(ctx, st, fl, 5));      // one launch (the runtime made twelve fills of these five)
    }
    // the links: at most the rows, so everything that holds them is sized by the rows and the kernels read their number from where the scan left it (r5: one read-back
    // less) -- while the rows are few (8 M: 36 B of link arrays per row = 288 MB; the bench step has 0.66 M).  A genome-scale group asks for the real number (ADVICE r5:
    // six arrays sized by the rows in every lane's pool where the exact sizing fitted)
    int64_t n_link_room = na;
    if (na > 0) {
        hipLaunchKernelGGL(k_link_flag, dim3(grid_for(na, 256, 1 << 30)), dim3(256), 0, st, b->arows.p, na, b->lk_flag.p);
        FZP_TRY(fzp_exclusive_scan_u32(ctx, b->lk_flag.p, b->lk_flag.p, (size_t)na, b->totals.p + 4));
        if (na > (8ll << 20)) {
            uint64_t nl = 0;
            FZP_TRY(fzp_fetch(ctx, st, &nl, b->totals.p + 4, sizeof(uint64_t)));
            n_link_room = (int64_t)std::min<uint64_t>(nl, (uint64_t)na);
        }
    }
    FZP_TRY(b->lk_i1.alloc((size_t)n_link_room)); FZP_TRY(b->lk_i2.alloc((size_t)n_link_room));
    FZP_TRY(b->lk_cis.alloc((size_t)n_link_room)); FZP_TRY(b->lk_trans.alloc((size_t)n_link_room)); FZP_TRY(b->left_lk.alloc((size_t)n_link_room));
    FZP_TRY(b->left_pk.alloc((size_t)n_link_room));
    if (na > 0) {
        ProfScope ps(ctx, "k4_links");
        hipLaunchKernelGGL(k_link_emit, dim3(grid_for(na, 256, 1 << 30)), dim3(256), 0, st, b->arows.p, na, b->lk_flag.p, b->totals.p + 4, b->lk_i1.p, b->lk_i2.p,
                           b->lk_cis.p, b->lk_trans.p, b->left_n.p, b->right_n.p, b->fr2.p);
    }
    if (ns > 0) {
        FZP_TRY(fzp_exclusive_scan_u32(ctx, b->left_n.p, b->left_off.p, (size_t)ns, nullptr));
        FZP_TRY(fzp_exclusive_scan_u32(ctx, b->right_n.p, b->right_off.p, (size_t)ns, nullptr));
        if (na > 0)
            hipLaunchKernelGGL(k_left_fill, dim3(grid_for(na, 256, 1 << 30)), dim3(256), 0, st, b->totals.p + 4, b->lk_i1.p, b->lk_i2.p, b->lk_cis.p, b->lk_trans.p,
                               b->left_off.p, b->left_fill.p, b->left_lk.p, b->left_pk.p);
        {
            ProfScope ps(ctx, "k4_greedy");
            hipLaunchKernelGGL(k_pj_init, dim3(grid_for(ns, 256, 1 << 30)), dim3(256), 0, st, ns, b->right_n.p, b->right_off.p, b->fr2.p, b->lk_i1.p, b->lk_i2.p,
                               b->lk_cis.p, b->lk_trans.p, b->pj.p);
            hipLaunchKernelGGL(k_pj_resolve, dim3(grid_for(ns, 256, 1 << 30)), dim3(256), 0, st, ns, b->pj.p, b->orient.p);
        }
        int64_t max_sites = 0;
        for (int c = 0; c < b->n_ctg; c++) max_sites = std::max<int64_t>(max_sites, b->h_site_begin[c + 1] - b->h_site_begin[c]);
        {
            ProfScope ps(ctx, "k4_sweep");
            if (max_sites <= 60 * 1024 && getenv("FZP_K4_SWEEP_GLOBAL") == nullptr)      // (FZP_K4_SWEEP_GLOBAL: the form for contigs whose states do not fit LDS, for the parity test)
                hipLaunchKernelGGL(k_sweep<true>, dim3(b->n_ctg), dim3(256), (size_t)(256 + ((max_sites + 15) & ~15LL)), st, b->site_begin.p, b->left_n.p, b->left_off.p,
                                   b->left_pk.p, b->right_n.p, b->right_off.p, b->lk_i2.p, b->lk_cis.p, b->lk_trans.p, b->orient.p);
            else
                hipLaunchKernelGGL(k_sweep<false>, dim3(b->n_ctg), dim3(256), 256, st, b->site_begin.p, b->left_n.p, b->left_off.p, b->left_pk.p, b->right_n.p, b->right_off.p,
                                   b->lk_i2.p, b->lk_cis.p, b->lk_trans.p, b->orient.p);
        }
        {
            ProfScope ps(ctx, "k4_extents");
            hipLaunchKernelGGL(k_extents, dim3(grid_for(ns, 4, 1 << 30)), dim3(256), 0, st, ns, b->sites.p, b->left_n.p, b->left_off.p, b->left_lk.p, b->right_n.p,
                               b->right_off.p, b->lk_i1.p, b->lk_i2.p, b->lk_cis.p, b->lk_trans.p, b->orient.p, b->lext.p, b->rext.p, b->lscore.p, b->rscore.p);
        }
    }
    {
        ProfScope ps(ctx, "k4_segment");
        hipLaunchKernelGGL(k_segment, dim3(b->n_ctg), dim3(256), 0, st, b->site_begin.p, b->sites.p, b->orient.p, b->lext.p, b->rext.p, b->lscore.p, b->rscore.p,
                           b->rawblk.p, b->blkcnt.p, b->blknew.p, b->pvars_tmp.p, b->pv_n.p, b->site_blk.p, b->site_b1.p);
    }
    FZP_TRY(fzp_exclusive_scan_u32(ctx, b->pv_n.p, b->pv_off.p, (size_t)b->n_ctg, b->totals.p + 5));
    uint64_t t = 0;
    b->h_pvar_begin.resize((size_t)b->n_ctg + 1);
    hipLaunchKernelGGL(k_u32_to_i64_begin, dim3((b->n_ctg + 1 + 63) / 64), dim3(64), 0, st, b->pv_off.p, b->n_ctg, b->totals.p + 5, b->pvar_begin.p);
    FZP_TRY(fzp_fetch_with_begins(ctx, st, &t, b->totals.p + 5, 1, b->h_pvar_begin.data(), b->pvar_begin.p, b->n_ctg));      // the count and the contigs' begins in one
    b->n_pvars = (int64_t)t;
    FZP_TRY(b->pvars.alloc((size_t)b->n_pvars));
    if (b->n_pvars > 0)
        hipLaunchKernelGGL(k_pv_compact, dim3(b->n_ctg), dim3(256), 0, st, b->site_begin.p, b->pv_n.p, b->pv_off.p, b->pvars_tmp.p, b->pvars.p);
    FZP_HIP(hipGetLastError());
    b->have_blocks = true;
    return FZP_OK;
}

// ================================================================================ K5 driver
int fzp_k5_reads(fzp_ctx *ctx, fzp_batch *b) {
    hipStream_t st = ctx->stream;
    if (!b->have_sites || !b->have_blocks) { fzp_set_error("read phasing needs sites and phase blocks"); return FZP_EINVAL; }
    if (!b->have_sets) FZP_TRY(fzp_k3_sets(ctx, b));
    const int64_t ns = b->n_sites, nq = b->n_qid;
    FZP_TRY(b->totals.alloc(8));
    FZP_TRY(b->bmin.alloc((size_t)nq)); FZP_TRY(b->bmax.alloc((size_t)nq)); FZP_TRY(b->rng_n.alloc((size_t)nq)); FZP_TRY(b->rng_off.alloc((size_t)nq));
    FZP_TRY(b->pread_begin.alloc((size_t)b->n_ctg + 1));
    int64_t n_slots = 0;
    b->n_preads = 0;
    bool begins_there = false;
    if (nq > 0) {
        { const fzp_fill_piece fl[2] = {{b->bmin.p, (size_t)nq * 4, 0x7fffffffu}, fzp_zeroes(b->bmax, (size_t)nq)}; FZP_TRY(fzp_fill(ctx, st, fl, 2)); }
        const uint32_t *set_n = b->set_n.p, *set_off = b->set_n.p + 2 * ns;
        if (ns > 0) {
            ProfScope ps(ctx, "k5_read_range");
            hipLaunchKernelGGL(k_read_votes<0>, dim3(grid_for(ns * 2, 4, 1 << 30)), dim3(256), 0, st, ns, b->sites.p, b->site_ctg.p, b->ctg_qoff.p, b->setq.p, set_n,
                               set_off, b->site_blk.p, b->site_b1.p, b->bmin.p, b->bmax.p, (const uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr);
        }
        hipLaunchKernelGGL(k_range_n, dim3(grid_for(nq, 256, 1 << 30)), dim3(256), 0, st, nq, b->bmin.p, b->bmax.p, b->rng_n.p);
        FZP_TRY(fzp_exclusive_scan_u32(ctx, b->rng_n.p, b->rng_off.p, (size_t)nq, b->totals.p + 6));
        uint64_t t = 0;
        FZP_TRY(fzp_fetch(ctx, st, &t, b->totals.p + 6, sizeof t));
        n_slots = (int64_t)t;
        FZP_TRY(b->c0.alloc((size_t)n_slots)); FZP_TRY(b->c1.alloc((size_t)n_slots)); FZP_TRY(b->pr_flag.alloc((size_t)n_slots));
        { const fzp_fill_piece fl[2] = {fzp_zeroes(b->c0, (size_t)n_slots), fzp_zeroes(b->c1, (size_t)n_slots)}; FZP_TRY(fzp_fill(ctx, st, fl, 2)); }
        if (n_slots > 0) {
            {
                ProfScope ps(ctx, "k5_read_count");
                hipLaunchKernelGGL(k_read_votes<1>, dim3(grid_for(ns * 2, 4, 1 << 30)), dim3(256), 0, st, ns, b->sites.p, b->site_ctg.p, b->ctg_qoff.p, b->setq.p, set_n,
                                   set_off, b->site_blk.p, b->site_b1.p, b->bmin.p, b->bmax.p, b->rng_off.p, b->c0.p, b->c1.p);
            }
            hipLaunchKernelGGL(k_read_flag, dim3(grid_for(n_slots, 256, 1 << 30)), dim3(256), 0, st, n_slots, b->c0.p, b->c1.p, b->pr_flag.p);
            FZP_TRY(fzp_exclusive_scan_u32(ctx, b->pr_flag.p, b->pr_flag.p, (size_t)n_slots, b->totals.p + 7));
            b->h_pread_begin.resize((size_t)b->n_ctg + 1);
            hipLaunchKernelGGL(k_pread_begin, dim3((b->n_ctg + 1 + 63) / 64), dim3(64), 0, st, b->ctg_qoff.p, b->n_ctg, nq, n_slots, b->totals.p + 7, b->rng_off.p, b->pr_flag.p,
                               b->pread_begin.p);
            FZP_TRY(fzp_fetch_with_begins(ctx, st, &t, b->totals.p + 7, 1, b->h_pread_begin.data(), b->pread_begin.p, b->n_ctg));      // the count and the contigs' begins in one
            begins_there = true;
            b->n_preads = (int64_t)t;
            FZP_TRY(b->preads.alloc((size_t)b->n_preads));
            if (b->n_preads > 0) {
                ProfScope ps(ctx, "k5_read_emit");
                hipLaunchKernelGGL(k_read_emit, dim3(grid_for(n_slots, 256, 1 << 30)), dim3(256), 0, st, n_slots, nq, b->n_preads, b->pr_flag.p, b->rng_off.p, b->bmin.p,
                                   b->c0.p, b->c1.p, b->ctg_qoff.p, b->n_ctg, b->preads.p);
            }
        }
    }
    FZP_TRY(b->preads.alloc((size_t)b->n_preads));
    if (!begins_there) {      // no reads or no slots: every contig begins (and ends) at 0
        b->h_pread_begin.assign((size_t)b->n_ctg + 1, 0);
        FZP_TRY(b->pread_begin.zero((size_t)b->n_ctg + 1, st));
    }
    FZP_HIP(hipGetLastError());
    b->have_preads = true;
    return FZP_OK;
}
