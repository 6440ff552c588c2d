// fzp_align.hip -- K1: read -> contig alignment on gfx950 (the role of blasr + samtools sort,
// falcon_unzip/unzip.py:86-91).  Spec "fzalign v1.1": oracle/align_oracle.c is its scalar twin and the
// kernels here match it bit-for-bit (summaries, CIGAR words, DP cell counts).  Parity vs blasr itself
// is UNPINNED (third-party binary, not vendored).
//
//   k_pack        ASCII -> 2 bit/base words (16 bases per u32, base m at bits 2m)
//   k_index_*     contig k-mers -> bucketed table of (key<<32 | position<<1 | strand bit), every sampled position, built partition by partition in LDS
//   k_seed        per read: diagonal-bin votes of sampled k-mers (both strands) in LDS, argmax, anchor
//   k_orient      per read: oriented (forward / reverse-complement) packed copy
//   k_sw          per read, ONE WAVE: adaptive anti-diagonal band, 64 cells = 64 lanes; neighbours
//                 arrive by DPP wave shifts; per step two 64-bit trace-back masks (v_cmp -> SGPR pair)
//                 go to HBM; steering compares lanes 0 and 63.  Integer VALU-bound, no MFMA.
//   k_tb_walk     per read, one lane: walks the masks back from the best cell, emits a 2-bit op stream
//   k_tb_cigar    per read, one wave: op stream -> forward run-length CIGAR, clips, summary
//   k_plan_*      per contig: aligned reads ordered by (POS, read), record filters (phasing.py:72-75), record offsets
//   k_gather(16)  accepted records -> contiguous CIGAR + ASCII SEQ arrays for an alnset / the phasing batch
#include <algorithm>
#include <type_traits>
#include <map>
#include <mutex>

#include "fzp_batch.h"
#include "fzp_swb_core.h"

namespace {
constexpr int32_t NEGV = -(1 << 26);
constexpr uint64_t EMPTY = ~0ull;
constexpr int MAX_BINS = 8192;

__host__ __device__ __forceinline__ int code_of(uint8_t c) {
    return (c == 'C' || c == 'c') ? 1 : (c == 'G' || c == 'g') ? 2 : (c == 'T' || c == 't') ? 3 : 0;
}

// ---- contigs arrive as the caller spells them; the phasing stages read the upper-cased text (phasing.py:494 `.upper()`)
__global__ void __launch_bounds__(256) k_upper(uint8_t *__restrict__ a, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i + 16 <= n) {
        uint4 v = *(const uint4 *)(a + i);
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t x = w[q];
            // bytes in 'a'..'z': (x + 0x1f) has bit 7 set for >= 'a' (0x61 + 0x1f = 0x80), (x + 0x05) has it set for > 'z' (0x7b + 0x05 = 0x80); 7-bit input assumed per byte check below
            const uint32_t hi = x & 0x80808080u, lo7 = x & 0x7f7f7f7fu;
            const uint32_t ge_a = (lo7 + 0x1f1f1f1fu) & 0x80808080u, gt_z = (lo7 + 0x05050505u) & 0x80808080u;
            const uint32_t is_lower = ge_a & ~gt_z & ~hi;
            w[q] = x - (is_lower >> 2);                       // 0x80 >> 2 = 0x20
        }
        *(uint4 *)(a + i) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        for (int64_t k = i; k < n; k++) { const uint8_t ch = a[k]; a[k] = (ch >= 'a' && ch <= 'z') ? (uint8_t)(ch - 32) : ch; }
    }
}

// ---- packing: one workgroup column per sequence, blockIdx.y strides over its words
__global__ void __launch_bounds__(256) k_pack(const uint8_t *__restrict__ ascii, const int64_t *__restrict__ seq_off, const int64_t *__restrict__ woff,
                                              uint32_t *__restrict__ out) {
    const int64_t s = blockIdx.x;
    const int64_t b0 = seq_off[s], n = seq_off[s + 1] - b0;
    const int64_t nw = ((n + 15) / 16 + 8 + 1) & ~1LL;   // zero pad words: 64-bit base windows may run past the end
    uint32_t *dst = out + woff[s];
    for (int64_t w = (int64_t)blockIdx.y * 256 + threadIdx.x; w < nw; w += (int64_t)gridDim.y * 256) {
        uint32_t v = 0;
        int64_t base = w * 16;
        for (int m = 0; m < 16; m++)
            if (base + m < n) v |= (uint32_t)code_of(ascii[b0 + base + m]) << (2 * m);
        dst[w] = v;
    }
}

__global__ void __launch_bounds__(256) k_pack2(const uint8_t *__restrict__ ascii, const int64_t *__restrict__ seq_be, const int64_t *__restrict__ woff,
                                               uint32_t *__restrict__ out) {      // like k_pack, sequence s = [seq_be[2s], seq_be[2s+1])
    const int64_t s = blockIdx.x;
    const int64_t b0 = seq_be[2 * s], n = seq_be[2 * s + 1] - b0;
    const int64_t nw = ((n + 15) / 16 + 8 + 1) & ~1LL;
    uint32_t *dst = out + woff[s];
    for (int64_t w = (int64_t)blockIdx.y * 256 + threadIdx.x; w < nw; w += (int64_t)gridDim.y * 256) {
        uint32_t v = 0;
        int64_t base = w * 16;
        for (int m = 0; m < 16; m++)
            if (base + m < n) v |= (uint32_t)code_of(ascii[b0 + base + m]) << (2 * m);
        dst[w] = v;
    }
}

__device__ __forceinline__ uint32_t base_at(const uint32_t *__restrict__ pk, int64_t i) { return (pk[i >> 4] >> ((i & 15) * 2)) & 3u; }

__device__ __forceinline__ uint32_t kmer_at(const uint32_t *__restrict__ pk, int64_t p, int k) {
    uint64_t w = (uint64_t)pk[p >> 4] | ((uint64_t)pk[(p >> 4) + 1] << 32);
    uint32_t key = (uint32_t)(w >> ((p & 15) * 2));
    return k < 16 ? (key & ((1u << (2 * k)) - 1u)) : key;
}
// reverse-complement of a k-mer key (base m at bits 2m)
__device__ __forceinline__ uint32_t rc_key(uint32_t key, int k) {
    uint32_t x = ~key;
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = __builtin_bswap32(x);
    return k < 16 ? (x >> (32 - 2 * k)) : x;
}
__device__ __forceinline__ uint32_t hash_slot(uint32_t key, int bits) { return (key * 0x9E3779B1u) >> (32 - bits); }

// ---- contig k-mer index: canonical k-mers (min of the k-mer and its reverse complement) of every
// CTG_STRIDE-th contig position -> EVERY such (position << 1 | "the canonical form is the reverse complement").
// Table: buckets of 4 entries (32 B, one sector per probe), entry = key << 32 | value; a key's entries fill the
// first free slots along its bucket chain (linear probing over the buckets of its partition), so a look-up may stop at the
// first bucket that still has a free slot: it has then seen every entry of the key (or more than MAX_OCC of them).
// Insertion order (a race) only decides which slot an entry lands in, never which entries a look-up finds.
constexpr int CTG_STRIDE = 2;
constexpr int MAX_OCC = 8;          // spec: k-mers with more index entries never produce a hit
constexpr int HIT_CAP = 4096;       // spec: hits of a read beyond the first HIT_CAP (sample order, then position) do not exist
constexpr int CHAIN_MAX_GAP = 2048;
__device__ __forceinline__ uint32_t canonical(uint32_t key, int k, uint32_t *is_rc) {
    uint32_t r = rc_key(key, k);
    *is_rc = r < key ? 1u : 0u;
    return r < key ? r : key;
}
// Build, two launches and no atomics on HBM beyond one per (workgroup, partition):
//   the table is cut into partitions of 2^PART_BITS buckets (64 KB = one LDS image); a key's bucket chain wraps inside its partition;
//   k_index_stage  a workgroup takes 65 536 consecutive sampled positions of a contig: LDS histogram over partitions -> one global
//                  atomicAdd per touched partition reserves a run in that partition's own 64 KB (used as the staging area) -> entries
//                  written there, unordered;
//   k_index_build  a workgroup per partition: staged entries into registers, the partition's table built in LDS (ds 64-bit CAS),
//                  written back as one coalesced 64 KB image (which also initialises every free slot: no memset).
// A key keeps at most MAX_OCC + 1 entries (more are never looked at: such k-mers do not produce hits), so homopolymer / satellite
// k-mers cannot flood a partition.
constexpr int PART_BITS = 11;                       // buckets per partition: 2048 x 4 slots x 8 B = 64 KB
constexpr int STAGE_KMERS = 65536;                  // sampled positions per k_index_stage workgroup
__device__ __forceinline__ uint32_t next_bucket(uint32_t bkt, uint32_t pmask) { return (bkt & ~pmask) | ((bkt + 1) & pmask); }

__global__ void __launch_bounds__(256) k_index_stage(const uint32_t *__restrict__ ctg_pk, const int64_t *__restrict__ ctg_woff, const int64_t *__restrict__ ctg_len,
                                                     const int64_t *__restrict__ idx_off, const int32_t *__restrict__ idx_bits, const int64_t *__restrict__ part_off, int k,
                                                     uint64_t *__restrict__ table, uint32_t *__restrict__ cursor, int32_t *__restrict__ overflow) {
    __shared__ uint32_t hist[1 << 12], base[1 << 12];          // partitions of one contig (<= 4096: contigs up to 2^31 bases / 2 per 8192-slot partition... see host check)
    const int c = blockIdx.y;
    const int64_t nk = ctg_len[c] - k + 1;
    const int64_t q0 = (int64_t)blockIdx.x * STAGE_KMERS;      // sampled index: position = 2 q
    if (q0 * CTG_STRIDE >= nk) return;
    const uint32_t *pk = ctg_pk + ctg_woff[c];
    const int bbits = idx_bits[c] - 2;
    const int pbits = bbits < PART_BITS ? bbits : PART_BITS;
    const int n_part = 1 << (bbits - pbits);
    uint64_t *tab = table + idx_off[c];
    uint32_t *cur = cursor + part_off[c];
    for (int i = threadIdx.x; i < n_part; i += 256) hist[i] = 0;
    __syncthreads();
    for (int pass = 0; pass < 2; pass++) {
        for (int64_t q = q0 + threadIdx.x; q < q0 + STAGE_KMERS; q += 256) {
            const int64_t pos = q * CTG_STRIDE;
            if (pos >= nk) break;
            uint32_t orc;
            const uint32_t key = canonical(kmer_at(pk, pos, k), k, &orc);
            const uint32_t part = hash_slot(key, bbits) >> pbits;
            if (pass == 0) atomicAdd(&hist[part], 1u);
            else {
                const uint32_t at = base[part] + atomicAdd(&hist[part], 1u);
                if (at < (4u << pbits)) tab[((size_t)part << (pbits + 2)) + at] = ((uint64_t)key << 32) | (uint64_t)(((uint32_t)pos << 1) | orc);
                else *overflow = 1;
            }
        }
        __syncthreads();
        if (pass == 0) {
            for (int i = threadIdx.x; i < n_part; i += 256) { const uint32_t h = hist[i]; base[i] = h ? atomicAdd(&cur[i], h) : 0u; hist[i] = 0; }
            __syncthreads();
        }
    }
}
__global__ void __launch_bounds__(256) k_index_build(const int32_t *__restrict__ part_ctg, const int64_t *__restrict__ part_off, const int64_t *__restrict__ idx_off,
                                                     const int32_t *__restrict__ idx_bits, uint64_t *__restrict__ table, const uint32_t *__restrict__ cursor,
                                                     int32_t *__restrict__ overflow) {
    extern __shared__ unsigned long long ltab[];               // 4 << pbits slots
    const int64_t gp = blockIdx.x;
    const int c = part_ctg[gp];
    const uint32_t part = (uint32_t)(gp - part_off[c]);
    const int bbits = idx_bits[c] - 2;
    const int pbits = bbits < PART_BITS ? bbits : PART_BITS;
    const uint32_t n_slots = 4u << pbits, pmask = (1u << pbits) - 1u;
    uint64_t *img = table + idx_off[c] + ((size_t)part << (pbits + 2));
    uint32_t cnt = cursor[gp];
    if (cnt > n_slots) cnt = n_slots;
    uint64_t mine[32];                                         // n_slots / 256 staged entries at most
    int nm = 0;
    for (uint32_t i = threadIdx.x; i < cnt; i += 256) mine[nm++] = img[i];
    for (uint32_t i = threadIdx.x; i < n_slots; i += 256) ltab[i] = (unsigned long long)EMPTY;
    __syncthreads();
    for (int m = 0; m < nm; m++) {
        const uint64_t word = mine[m];
        const uint32_t key = (uint32_t)(word >> 32);
        uint32_t bkt = hash_slot(key, bbits) & pmask;          // local bucket
        uint32_t s0 = (key >> 3) & 3u, copies = 0, walked = 0;
        bool done = false;
        while (!done) {
            for (int q = 0; q < 4 && !done; q++) {
                const uint32_t sl = bkt * 4 + ((s0 + q) & 3u);
                unsigned long long cur = ltab[sl];
                if (cur == (unsigned long long)EMPTY) {
                    if (copies > (uint32_t)MAX_OCC) { done = true; break; }      // the key already has MAX_OCC + 1 entries: enough to be ignored
                    cur = atomicCAS(&ltab[sl], (unsigned long long)EMPTY, (unsigned long long)word);
                    if (cur == (unsigned long long)EMPTY) { done = true; break; }
                }
                if ((uint32_t)(cur >> 32) == key) copies++;
            }
            if (done) break;
            bkt = (bkt + 1) & pmask;
            s0 = 0;
            if (++walked > pmask) { *overflow = 1; break; }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_slots; i += 256) img[i] = ltab[i];
}
// every index entry of `key` (values, unsorted) into e[0..MAX_OCC); returns the count, MAX_OCC + 1 if there are more.
// b0 = the key's first bucket, already loaded by the caller (several probes are kept in flight).
__device__ __forceinline__ int index_collect(const uint64_t *__restrict__ tab, int bbits, uint32_t key, uint32_t bkt, uint4 lo, uint4 hi, uint32_t *e) {
    const uint32_t pmask = (1u << (bbits < PART_BITS ? bbits : PART_BITS)) - 1u;
    int cnt = 0;
    for (;;) {
        const uint32_t kk[4] = {lo.y, lo.w, hi.y, hi.w}, vv[4] = {lo.x, lo.z, hi.x, hi.z};
        bool open = false;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (kk[q] == 0xffffffffu && vv[q] == 0xffffffffu) open = true;
            else if (kk[q] == key) { if (cnt < MAX_OCC) e[cnt] = vv[q]; cnt++; }
        }
        if (open || cnt > MAX_OCC) break;
        bkt = next_bucket(bkt, pmask);
        const uint4 *bp = (const uint4 *)(tab + (size_t)bkt * 4);
        lo = bp[0]; hi = bp[1];
    }
    return cnt > MAX_OCC ? MAX_OCC + 1 : cnt;
}

// the same walk along the key's bucket chain, counting only: returns the number of entries (MAX_OCC + 1 if there are more) and the first
// one's value.  Nearly every k-mer of a read has no entry or one; k_seed runs index_collect (arrays, ordering) only in waves where a
// k-mer has several.
__device__ __forceinline__ int index_count(const uint64_t *__restrict__ tab, int bbits, uint32_t key, uint32_t bkt, uint4 lo, uint4 hi, uint32_t *first) {
    const uint32_t pmask = (1u << (bbits < PART_BITS ? bbits : PART_BITS)) - 1u;
    int cnt = 0;
    for (;;) {
        const uint32_t kk[4] = {lo.y, lo.w, hi.y, hi.w}, vv[4] = {lo.x, lo.z, hi.x, hi.z};
        bool open = false;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (kk[q] == 0xffffffffu && vv[q] == 0xffffffffu) open = true;
            else if (kk[q] == key) { if (cnt == 0) *first = vv[q]; cnt++; }
        }
        if (open || cnt > MAX_OCC) break;
        bkt = next_bucket(bkt, pmask);
        const uint4 *bp = (const uint4 *)(tab + (size_t)bkt * 4);
        lo = bp[0]; hi = bp[1];
    }
    return cnt > MAX_OCC ? MAX_OCC + 1 : cnt;
}

struct Anchor { int32_t aligned, strand, i_a, c_a; };

__device__ __forceinline__ uint64_t block_max_u64(uint64_t v, uint64_t *sh) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { uint64_t o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
    if (lane_id() == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    uint64_t r = sh[0];
    for (int i = 1; i < 4; i++) r = sh[i] > r ? sh[i] : r;
    __syncthreads();
    return r;
}

// ---- seeding, part 1: one workgroup per read (spec: oracle/align_oracle.c header, "hits", "windows")
//   A. rounds of 1024 samples, four consecutive samples per thread (their first bucket loads in flight together);
//      a block scan of the per-thread hit counts puts the hits into the read's hit list (HBM, HIT_CAP slots per read of
//      the launch) in (sample, position) order -- the order the HIT_CAP of the spec is defined on -- and every stored
//      hit votes for its (strand, diagonal bin) in LDS;
//   B. W1 / W2 by block-wide argmax over the vote bins -> SeedWin.
struct SeedWin { int32_t n_hits, shift, s1, b1, have2, s2, b2, pad_; };
__global__ void __launch_bounds__(256) k_seed(int64_t first, const uint32_t *__restrict__ read_pk, const int64_t *__restrict__ read_woff, const int32_t *__restrict__ read_len,
                                              const int32_t *__restrict__ read_ctg, const int64_t *__restrict__ ctg_len, const int64_t *__restrict__ idx_off,
                                              const int32_t *__restrict__ idx_bits, const uint64_t *__restrict__ table, int k, int stride, int min_hits,
                                              uint2 *__restrict__ hits_g, SeedWin *__restrict__ win) {
    extern __shared__ uint32_t votes[];   // [2 * NB] vote bins
    __shared__ uint64_t red[4];
    __shared__ uint32_t wsum[4];
    const int64_t r = first + blockIdx.x;
    const int64_t n = read_len[r];
    const int c = read_ctg[r];
    const int64_t Lc = ctg_len[c];
    SeedWin sw = {0, 10, 0, 0, 0, 0, 0, 0};
    if (n < k || Lc < k) { if (threadIdx.x == 0) win[blockIdx.x] = sw; return; }
    int shift = 10;
    while ((((Lc + n) >> shift) + 2) > MAX_BINS) shift++;
    const int NB = (int)(((Lc + n) >> shift) + 2);
    const uint32_t *pk = read_pk + read_woff[r];
    const uint64_t *tab = table + idx_off[c];
    const int bbits = idx_bits[c] - 2;
    uint2 *hits = hits_g + (size_t)blockIdx.x * HIT_CAP;   // (strand << 31 | oriented offset, contig position), spec order
    for (int i = threadIdx.x; i < 2 * NB; i += 256) votes[i] = 0;
    __syncthreads();
    const int64_t ns = (n - k) / stride + 1;   // sampled FORWARD read offsets 0, stride, ...
    uint32_t n_hits = 0;                       // block-uniform
    const int lane = lane_id(), wid = threadIdx.x >> 6;
    for (int64_t base = 0; base < ns && n_hits < (uint32_t)HIT_CAP; base += 1024) {
        uint32_t key[4], orr[4], bkt[4];
        uint4 lo[4], hi[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t m = base + 4 * threadIdx.x + u;
            key[u] = 0; orr[u] = 0; bkt[u] = 0;
            lo[u] = hi[u] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
            if (m < ns) {
                key[u] = canonical(kmer_at(pk, m * stride, k), k, &orr[u]);
                bkt[u] = hash_slot(key[u], bbits);
                const uint4 *bp = (const uint4 *)(tab + (size_t)bkt[u] * 4);
                lo[u] = bp[0]; hi[u] = bp[1];
            }
        }
        uint32_t ent[4][MAX_OCC], e0[4];
        int cnt[4];
        uint32_t mine = 0;
        bool several = false;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t m = base + 4 * threadIdx.x + u;
            cnt[u] = 0; e0[u] = 0;
            if (m < ns) {
                cnt[u] = index_count(tab, bbits, key[u], bkt[u], lo[u], hi[u], &e0[u]);
                if (cnt[u] > MAX_OCC) cnt[u] = 0;
                several = several || cnt[u] > 1;
            }
            mine += (uint32_t)cnt[u];
        }
        if (__any(several)) {                                    // rare: a k-mer with 2..MAX_OCC entries -- all of them, by position
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (cnt[u] > 1) {
                    (void)index_collect(tab, bbits, key[u], bkt[u], lo[u], hi[u], ent[u]);
                    for (int a = 1; a < cnt[u]; a++) {           // insertion sort
                        const uint32_t v = ent[u][a];
                        int b = a - 1;
                        while (b >= 0 && ent[u][b] > v) { ent[u][b + 1] = ent[u][b]; b--; }
                        ent[u][b + 1] = v;
                    }
                }
            }
        }
        // slots in (sample, position) order: exclusive scan of the per-thread counts
        const uint32_t incl = wave_incl_scan_u32(mine);
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        uint32_t off = n_hits + incl - mine, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { if (w < wid) off += wsum[w]; tot += wsum[w]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t pf = (base + 4 * threadIdx.x + u) * stride;
            auto emit = [&](uint32_t hit) {
                const int s_ = (int)((hit & 1u) ^ orr[u]);
                const int64_t cp = hit >> 1, i = s_ ? n - k - pf : pf;
                hits[off] = make_uint2(((uint32_t)s_ << 31) | (uint32_t)i, (uint32_t)cp);
                atomicAdd(&votes[s_ * NB + (int)((cp - i + n) >> shift)], 1u);
            };
            if (cnt[u] == 1) { if (off < (uint32_t)HIT_CAP) emit(e0[u]); off++; }
            else if (cnt[u] > 1)
                for (int a = 0; a < cnt[u]; a++, off++) {
                    if (off >= (uint32_t)HIT_CAP) break;
                    emit(ent[u][a]);
                }
        }
        n_hits = min(n_hits + tot, (uint32_t)HIT_CAP);
        __syncthreads();
    }
    // W1: max votes[b]+votes[b+1]; ties -> forward strand, lower bin
    uint64_t best = 0;
    for (int x = threadIdx.x; x < 2 * NB; x += 256) {
        int b = x >= NB ? x - NB : x;
        if (b + 1 >= NB) continue;
        uint64_t sc = (uint64_t)votes[x] + votes[x + 1];
        uint64_t key = (sc << 32) | (uint64_t)(0xffffffffu - (uint32_t)x);
        best = key > best ? key : best;
    }
    best = block_max_u64(best, red);
    const uint32_t w1 = (uint32_t)(best >> 32);
    if ((int32_t)w1 < min_hits || w1 == 0) { if (threadIdx.x == 0) win[blockIdx.x] = sw; return; }
    const int x1 = (int)(0xffffffffu - (uint32_t)best);
    const int s1 = x1 >= NB, b1 = s1 ? x1 - NB : x1;
    // W2: the best window on the other strand or at least 3 bins away
    uint64_t best2 = 0;
    for (int x = threadIdx.x; x < 2 * NB; x += 256) {
        const int sx = x >= NB, b = sx ? x - NB : x;
        if (b + 1 >= NB) continue;
        if (sx == s1 && b - b1 < 3 && b1 - b < 3) continue;
        uint64_t sc = (uint64_t)votes[x] + votes[x + 1];
        uint64_t key = (sc << 32) | (uint64_t)(0xffffffffu - (uint32_t)x);
        best2 = key > best2 ? key : best2;
    }
    best2 = block_max_u64(best2, red);
    if (threadIdx.x == 0) {
        const uint32_t w2 = (uint32_t)(best2 >> 32);
        const int x2 = (int)(0xffffffffu - (uint32_t)best2);
        sw.n_hits = (int32_t)n_hits; sw.shift = shift; sw.s1 = s1; sw.b1 = b1;
        sw.have2 = ((int32_t)w2 >= min_hits && w2 > 0 && 4ull * w2 >= (uint64_t)w1) ? 1 : 0;
        sw.s2 = x2 >= NB; sw.b2 = sw.s2 ? x2 - NB : x2;
        win[blockIdx.x] = sw;
    }
}

// wave-uniform max of non-negative keys without LDS traffic: 4 DPP steps inside each row of 16 lanes, the 4 rows via SGPRs
__device__ __forceinline__ int32_t wave_max_nonneg_dpp(int32_t v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true));   // row_half_mirror
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true));   // row_mirror
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// ---- seeding, part 2 (spec "chains"): one wave per (read, window).  The window's hits are walked in list order (reversed on
// the reverse strand), 64 list entries per load; the last 64 window hits sit one per lane (lane = running index mod 64), the
// new hit is tested against all of them at once and the best predecessor comes out of one wave-wide max.  The chain is a
// serial dependence per read: thousands of waves in flight hide it.
__global__ void __launch_bounds__(64) k_chain(int64_t first, int64_t count, const int32_t *__restrict__ read_len, const uint2 *__restrict__ hits_g,
                                              const SeedWin *__restrict__ win, Anchor *__restrict__ anc, Anchor *__restrict__ ancB) {
    const int64_t wv = blockIdx.x;
    if (wv >= 2 * count) return;
    const int64_t slot = wv >> 1;
    const int which = (int)(wv & 1);
    const int64_t r = first + slot;
    const int lane = lane_id();
    const SeedWin sw = win[slot];
    Anchor *out = which ? ancB : anc;
    const Anchor none = {0, 0, 0, 0};
    if (sw.n_hits == 0 || (which && !sw.have2)) { if (lane == 0) out[r] = none; return; }
    const int32_t n = read_len[r];
    const int ws = which ? sw.s2 : sw.s1, wb = which ? sw.b2 : sw.b1, shift = sw.shift;
    const uint2 *hits = hits_g + (size_t)slot * HIT_CAP;
    const int32_t nh = sw.n_hits;
    int32_t ri = 0, rcp = 0, rd = 0, rf = 0, rst = 0;      // ring: lane L holds window hit number e with e % 64 == L
    int32_t e = 0, best_f = 0, best_st = -1;
    for (int32_t x0 = 0; x0 < nh; x0 += 64) {
        const int32_t xl = x0 + lane;
        const int32_t hl = ws ? nh - 1 - xl : xl;
        uint2 hv = make_uint2(0, 0);
        if (xl < nh) hv = hits[hl];
        // which of the 64 loaded hits belong to this window (strand, bins wb-1 .. wb+2): tested lane-parallel, the serial part below only
        // sees those (about a third of the list at cfg2; the per-hit scalar tests were most of this kernel's scalar instructions)
        bool inw = false;
        if (xl < nh && (int)(hv.x >> 31) == ws) {
            const int b = (int)(((int64_t)(int32_t)hv.y - (int32_t)(hv.x & 0x7fffffffu) + n) >> shift);
            inw = b >= wb - 1 && b <= wb + 2;
        }
        for (uint64_t todo = __ballot(inw); todo; todo &= todo - 1) {
            const int jx = __builtin_ctzll(todo);
            const uint32_t hx = (uint32_t)__builtin_amdgcn_readlane((int32_t)hv.x, jx), hy = (uint32_t)__builtin_amdgcn_readlane((int32_t)hv.y, jx);
            const int32_t i = (int32_t)(hx & 0x7fffffffu), cp = (int32_t)hy;
            const int32_t h = ws ? nh - 1 - (x0 + jx) : x0 + jx;
            const int32_t d = cp - i;
            const int32_t dist = (e - 1 - lane) & 63;           // this lane's hit is `dist + 1` window hits back
            const int32_t di = i - ri;
            int32_t dd = d - rd; dd = dd < 0 ? -dd : dd;
            const bool ok = dist < e && di >= 1 && di <= CHAIN_MAX_GAP && cp > rcp && dd <= 16 + (di >> 4);
            const int32_t key = ok ? ((rf << 6) | (63 - dist)) : 0;
            const int32_t K = wave_max_nonneg_dpp(key);
            int32_t f = 1, st = h;
            if (K > 0) {
                f = (K >> 6) + 1;
                const int32_t src = (e - 1 - (63 - (K & 63))) & 63;
                st = __builtin_amdgcn_readlane(rst, src);
            }
            if (lane == (e & 63)) { ri = i; rcp = cp; rd = d; rf = f; rst = st; }
            if (f > best_f) { best_f = f; best_st = st; }
            e++;
        }
    }
    if (lane == 0) {
        Anchor a = none;
        if (best_st >= 0) {
            // v1.4: the chain's first hit IS the anchor (a cell of the true path): the extension runs forward from it (k_sw) and, on the reversed
            // read prefix and contig window, backward from it (k_back_prep -> k_sw again)
            const uint2 hv = hits[best_st];
            a.aligned = 1; a.strand = ws; a.i_a = (int32_t)(hv.x & 0x7fffffffu); a.c_a = (int32_t)hv.y;
        }
        out[r] = a;
    }
}

// ---- oriented packed copy of each read (slot = read for the first candidates; the second candidates of the few reads
// that have one are compacted: slot w -> read ridx[w], own word offsets)
__global__ void __launch_bounds__(256) k_orient(int64_t first, const uint32_t *__restrict__ read_pk, const int64_t *__restrict__ read_woff, const int32_t *__restrict__ read_len,
                                                const Anchor *__restrict__ anc, const int32_t *__restrict__ ridx, const int64_t *__restrict__ out_woff, uint32_t *__restrict__ out) {
    const int64_t sl = first + blockIdx.x;
    const int64_t r = ridx ? ridx[sl] : sl;
    const int64_t n = read_len[r];
    const int64_t nw = ((n + 15) / 16 + 8 + 1) & ~1LL;
    const uint32_t *src = read_pk + read_woff[r];
    uint32_t *dst = out + out_woff[sl];
    const bool rc = anc[sl].strand != 0;
    for (int64_t w = (int64_t)blockIdx.y * 256 + threadIdx.x; w < nw; w += (int64_t)gridDim.y * 256) {
        uint32_t v = src[w];
        if (rc) {
            v = 0;
            for (int m = 0; m < 16; m++) {
                int64_t x = w * 16 + m;
                if (x < n) v |= (3u - base_at(src, n - 1 - x)) << (2 * m);
            }
        }
        dst[w] = v;
    }
}

struct DpInfo { int32_t steps, best_t, best_lane, best_score; };
// segmented trace-back (k_tb_walk<true>, below): segment length in DP steps, overlap, ops buffer per walker, pieces per read
constexpr int TBS_SEG = 4096, TBS_SEG_SHIFT = 12, TBS_OV = 512, TBS_RAW_WORDS = (TBS_SEG + TBS_OV) / 16 + 2, TBS_MAX_PIECES = 256;
constexpr int64_t TBS_SINGLE_STEPS = 40960;     // a read with at most this many DP steps of capacity is one walker's work (its serial walk is no longer than the launch anyway)

// wave-wide shifts by one lane (gfx9 DPP wave_shr / wave_shl); vacated lane takes `fill`
__device__ __forceinline__ int32_t wave_shr1(int32_t v, int32_t fill) {   // lane k <- lane k-1
    return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ int32_t wave_shl1(int32_t v, int32_t fill) {   // lane k <- lane k+1
    return __builtin_amdgcn_update_dpp(fill, v, 0x130, 0xf, 0xf, false);
}

// ---- K1 hot kernel: adaptive banded DP, one wave per read
//
// Lane k of the wave owns cell (i = i0 + k, j = t - i) of anti-diagonal t.  Per step the band moves
// DOWN (i0++) or RIGHT.  With A = the previous step's value in the same lane and B = the neighbour
// lane's (lane k+1 after DOWN, k-1 after RIGHT):
//      H = max(diag + s, max(A, B) - gap)
// The diagonal operand is kept pre-shifted: X = H(t-2) moved by (previous move) so that this step needs
// one more wave_shl only when it moves DOWN (after RIGHT->DOWN lane 63 sees the band edge).
// Scores are stored biased by 2^24 so that the DPP zero fill (bound_ctrl) of a missing neighbour IS the
// "minus infinity" of the spec; that lets the neighbour shifts ride on the add / max / compare
// themselves (v_add_u32_dpp, v_max_i32_dpp; G = (A >= B) is read off as max(A,B) == A) instead of separate moves.
// Trace-back masks per step: D = (H == diag + s), G = (A >= B) ("the gap comes from the same lane"), each a
// 64-bit lane mask; a step's pair is 16 B, stored step-major through the scalar cache.  Upcoming read / contig
// bases sit in 64-bit SGPR windows, so no step waits on memory.  The interior of the matrix runs a
// counted loop with no range checks; the first ~130 and last ~64 steps run the checked variant.
constexpr int32_t SW_BIAS = 1 << 24;   // stands in for the spec's -2^26 (any value far below every real score gives the same masks on
                                       // reachable cells); small enough that (H << 6) + 6 bits stays below 2^31 for scores < 2^24

struct BaseStream {          // wave-uniform: lives in SGPRs
    const uint64_t *pk;      // 32 bases per word
    int64_t w;               // index of the word in `cur`
    uint64_t cur, nxt;
    int cnt;                 // bases left in cur
    __device__ __forceinline__ void init(const uint32_t *pk32, int64_t abs_idx) {
        pk = (const uint64_t *)pk32;
        w = abs_idx >> 5;
        cur = pk[w] >> ((abs_idx & 31) * 2);
        cnt = 32 - (int)(abs_idx & 31);
        nxt = pk[w + 1];
    }
    __device__ __forceinline__ int32_t pop() {
        int32_t c = (int32_t)(cur & 3ull);
        cur >>= 2;
        if (--cnt == 0) { cur = nxt; cnt = 32; w++; nxt = pk[w + 1]; }
        return c;
    }
};

// wave-uniform 16-byte store through the scalar data cache (s_store_dwordx4): the per-step trace-back masks
// leave the DP kernel this way, off the vector path.  The data SGPRs are read at issue, so they may be reused at
// once; the kernel ends with s_dcache_wb.
__device__ __forceinline__ void scalar_store16(void *base, uint32_t byte_off, uint64_t lo, uint64_t hi) {
    const __uint128_t v = ((__uint128_t)hi << 64) | lo;
    asm volatile("s_store_dwordx4 %[v], %[p], %[o]" ::[v] "s"(v), [p] "s"(base), [o] "s"(byte_off) : "memory");
}

// one DP step of the checked variant (first ~130 and last ~64 steps of a read): plain HIP, sentinels and validity
// handled explicitly.  The masks of the step are wave ballots.
#define SW_STEP()                                                                                          \
    {                                                                                                      \
        int32_t hd, m, Hn;                                                                                 \
        if (down) {                                                                                        \
            int32_t c = qs.pop();                                                                          \
            c = qpos < nq ? c : 4;                                                                         \
            qpos++;                                                                                        \
            i0++;                                                                                          \
            qc = wave_shl1(qc, c);                                                                         \
            const int32_t sc = qc == tc ? match : -mismatch;                                               \
            hd = (pdown ? wave_shl1(X, 0) : X) + sc;                                                       \
            m = max(H, wave_shl1(H, 0));                                                                   \
        } else {                                                                                           \
            int32_t c = ts.pop();                                                                          \
            c = tpos < nt ? c : 5;                                                                         \
            tpos++;                                                                                        \
            tc = wave_shr1(tc, c);                                                                         \
            const int32_t sc = qc == tc ? match : -mismatch;                                               \
            hd = (pdown ? X : wave_shr1(X, 0)) + sc;                                                       \
            m = max(H, wave_shr1(H, 0));                                                                   \
        }                                                                                                  \
        const uint64_t gmask = __ballot(m == H);                                                           \
        Hn = max(hd, m - gap);                                                                             \
        const uint64_t dmask = __ballot(Hn == hd);                                                         \
        if (STORE) { const uint32_t t_ = (uint32_t)__builtin_amdgcn_readfirstlane(t); scalar_store16(tbr, ((t_ >> 6) * (uint32_t)stride + (t_ & 63u)) * 16u, dmask, gmask); } \
        bool upd = Hn > bs;                                                                                \
        {                                                                                                  \
            const int32_t ci = i0 + lane, cj = t - ci;                                                     \
            upd = upd && ci >= 0 && ci < nq && cj >= 0 && cj < nt && (ci == nq - 1 || cj == nt - 1);       \
        }                                                                                                  \
        bs = upd ? Hn : bs;                                                                                \
        bt = upd ? t : bt;                                                                                 \
        mvacc |= (uint64_t)(down ? 1 : 0) << (t & 63);                                                     \
        const int32_t top = __builtin_amdgcn_readlane(Hn, 0), bot = __builtin_amdgcn_readlane(Hn, 63);     \
        X = H;            /* H(t-2) stays in its own lane layout: the next step shifts it as its two moves say */ \
        H = Hn;                                                                                            \
        pdown = down;                                                                                      \
        t++;                                                                                               \
        down = t < 64 ? ((t & 1) == 0) : !(top > bot);                                                     \
    }

// after step t-1 completed a 64-step chunk (or at the very end): the chunk's move record
#define SW_FLUSH(PARTIAL)                                                                                  \
    {                                                                                                      \
        if ((PARTIAL) || ((t - 1) & 63) == 63) {                                                           \
            uint32_t glane = 32u;                                                                          \
            if ((t & (TBS_SEG - 1)) == 0) {   /* step t-1 tops a trace-back segment: the lane of its best H is where that segment's walker starts */ \
                asm volatile("" ::: "memory");   /* keeps the compiler from running this reduction on every flush and selecting afterwards (it did: +0.7 VALU per step) */ \
                int32_t hv = H, hl = lane;                                                                 \
                _Pragma("unroll") for (int d_ = 32; d_ >= 1; d_ >>= 1) {                                   \
                    const int32_t ov_ = __shfl_xor(hv, d_, 64), ol_ = __shfl_xor(hl, d_, 64);              \
                    if (ov_ > hv || (ov_ == hv && ol_ < hl)) { hv = ov_; hl = ol_; }                       \
                }                                                                                          \
                glane = (uint32_t)hl;                                                                      \
            }                                                                                              \
            if (lane == 0) mvr[(t - 1) >> 6] = make_ulonglong2(mvacc, (uint64_t)(uint32_t)(i0 - __popcll(mvacc)) | ((uint64_t)glane << 32)); \
            mvacc = 0;                                                                                     \
        }                                                                                                  \
    }

// ---- interior block: up to 32 steps with every lane strictly inside the matrix, hand-scheduled.
// The block is VALU-issue bound (a SIMD issues one wave64 VALU op per 4 cycles), so the point is the VALU count
// per step: 12 (v1.5: an interior block holds no border cell, so nothing of it can be the extension's terminal and it keeps no best-cell key), with 9 SALU and one scalar-memory op riding along on their own ports.
//   * H of two steps ago is never copied or shifted: the two registers swap roles every step (hence two code
//     parities) and the diagonal's lane shift rides on the add (hence a variant per pair of moves): 8 step variants;
//   * the two trace-back masks of a step are the 64-bit results of v_cmp_e64 landing in an SGPR quad that
//     goes out with one s_store_dwordx4 (16 B per step, step-major) -- no per-lane bit accumulators;
//   * bases entering the band come from two 64-bit SGPR windows (32 bases each, enough for a whole block):
//     s_bfe_u64 picks the next one, v_writelane drops it into lane 63 (DOWN) / lane 0 (RIGHT);
//   * moves are collected in a 32-bit shift register (first step of the block ends up in the highest used bit); its bit 0 is the block's last
//     move, so the steps do not keep a "previous move" register up to date (one SALU less per step: the scalar port has no slack).
// The store offset is the step counter: it enters 16 * steps below 2^31 and the signed overflow of its increment ends the block.
// No DPP source is written fewer than two instructions before it is read (gfx9 DPP hazard).
#define SWB_DPP_SHL " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define SWB_DPP_SHR " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
// one DOWN step.  HC: operand holding H of the previous step; XC: operand holding H of two steps ago, receives the new H;
// HDADD: the add that forms the diagonal operand (shifted by what the last two moves say); NP: parity after this step;
// the step ends by jumping to the variant (NP, previous = DOWN, next move) or to the exit of parity NP.
#define SWB_DOWN(LBL, HC, XC, HDADD, NP, KB, ST)                                                       \
    "Lsw%=_" LBL ":\n\t"                                                                          \
    "s_bfe_u64 s[56:57], %[qb], %[qsel]\n\t"                                                      \
    "v_mov_b32_dpp %[qc], %[qc] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"                        \
    "s_add_u32 %[qsel], %[qsel], 2\n\t"                                                           \
    "v_max_i32_dpp %[mm], " HC ", " HC SWB_DPP_SHL "\n\t"                                         \
    "v_writelane_b32 %[qc], s56, 63\n\t"                                                          \
    "v_cmp_eq_i32_e64 s[62:63], %[mm], " HC "\n\t"                                                \
    "v_cmp_eq_u32_e32 vcc, %[qc], %[tc]\n\t"                                                      \
    "v_cndmask_b32_e32 %[sc], %[vmis], %[vmat], vcc\n\t"                                          \
    HDADD "\n\t"                                                                                  \
    "v_subrev_u32_e32 %[mm], %[gap], %[mm]\n\t"                                                   \
    "v_max_i32_e32 " XC ", %[hd], %[mm]\n\t"                                                      \
    "v_cmp_eq_i32_e64 s[60:61], " XC ", %[hd]\n\t"                                                \
    "s_lshl1_add_u32 %[mv], %[mv], 1\n\t"                                                         \
    KB                                                      \
    "v_readlane_b32 %[top], " XC ", 0\n\t"                                                        \
    ST                                                                                            \
    "v_readlane_b32 %[bot], " XC ", 63\n\t"                                                       \
    "s_addk_i32 %[soff], 16\n\t"                                                                  \
    "s_cbranch_scc1 Lsw%=_end" NP "\n\t"                                                          \
    "s_cmp_gt_i32 %[top], %[bot]\n\t"                                                             \
    "s_cbranch_scc1 Lsw%=_p" NP "DR\n\t"                                                          \
    "s_branch Lsw%=_p" NP "DD\n"
#define SWB_RIGHT(LBL, HC, XC, HDADD, NP, KB, ST)                                                       \
    "Lsw%=_" LBL ":\n\t"                                                                          \
    "s_bfe_u64 s[56:57], %[tb], %[tsel]\n\t"                                                      \
    "v_mov_b32_dpp %[tc], %[tc] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"                        \
    "s_add_u32 %[tsel], %[tsel], 2\n\t"                                                           \
    "v_max_i32_dpp %[mm], " HC ", " HC SWB_DPP_SHR "\n\t"                                         \
    "v_writelane_b32 %[tc], s56, 0\n\t"                                                           \
    "v_cmp_eq_i32_e64 s[62:63], %[mm], " HC "\n\t"                                                \
    "v_cmp_eq_u32_e32 vcc, %[qc], %[tc]\n\t"                                                      \
    "v_cndmask_b32_e32 %[sc], %[vmis], %[vmat], vcc\n\t"                                          \
    HDADD "\n\t"                                                                                  \
    "v_subrev_u32_e32 %[mm], %[gap], %[mm]\n\t"                                                   \
    "v_max_i32_e32 " XC ", %[hd], %[mm]\n\t"                                                      \
    "v_cmp_eq_i32_e64 s[60:61], " XC ", %[hd]\n\t"                                                \
    "s_lshl_b32 %[mv], %[mv], 1\n\t"                                                              \
    KB                                                      \
    "v_readlane_b32 %[top], " XC ", 0\n\t"                                                        \
    ST                                                                                            \
    "v_readlane_b32 %[bot], " XC ", 63\n\t"                                                       \
    "s_addk_i32 %[soff], 16\n\t"                                                                  \
    "s_cbranch_scc1 Lsw%=_end" NP "\n\t"                                                          \
    "s_cmp_gt_i32 %[top], %[bot]\n\t"                                                             \
    "s_cbranch_scc1 Lsw%=_p" NP "RR\n\t"                                                          \
    "s_branch Lsw%=_p" NP "RD\n"
// label "pPab": parity P (0: H in %[H], X in %[X]; 1: swapped), a = previous move, b = this move
// the diagonal predecessor of lane k is lane k - 1 + (DOWN moves among the last two) of H(t-2)
#define SW_BLOCK_ASM(ST)                                                                            \
    asm volatile(                                                                                                                                                                   \
        "s_nop 1\n\t"                                                                                                                                                               \
        "s_cmp_eq_u32 %[dn], 0\n\t"                                                                                                                                                 \
        "s_cbranch_scc1 Lsw%=_enterR\n\t"                                                                                                                                           \
        "s_cmp_eq_u32 %[pm], 0\n\t"                                                                                                                                                 \
        "s_cbranch_scc1 Lsw%=_p0RD\n\t"                                                                                                                                             \
        "s_branch Lsw%=_p0DD\n"                                                                                                                                                     \
        "Lsw%=_enterR:\n\t"                                                                                                                                                         \
        "s_cmp_eq_u32 %[pm], 0\n\t"                                                                                                                                                 \
        "s_cbranch_scc1 Lsw%=_p0RR\n\t"                                                                                                                                             \
        "s_branch Lsw%=_p0DR\n"                                                                                                                                                     \
                                                                                                                                                                                    \
        SWB_DOWN("p0DD", "%[H]", "%[X]", "v_add_u32_dpp %[hd], %[X], %[sc]" SWB_DPP_SHL, "1", "", ST)                                                                               \
        SWB_DOWN("p0RD", "%[H]", "%[X]", "v_add_u32_e32 %[hd], %[X], %[sc]", "1", "", ST)                                                                                           \
        SWB_RIGHT("p0DR", "%[H]", "%[X]", "v_add_u32_e32 %[hd], %[X], %[sc]", "1", "", ST)                                                                                          \
        SWB_RIGHT("p0RR", "%[H]", "%[X]", "v_add_u32_dpp %[hd], %[X], %[sc]" SWB_DPP_SHR, "1", "", ST)                                                                              \
        SWB_DOWN("p1DD", "%[X]", "%[H]", "v_add_u32_dpp %[hd], %[H], %[sc]" SWB_DPP_SHL, "0", "", ST)                                        \
        SWB_DOWN("p1RD", "%[X]", "%[H]", "v_add_u32_e32 %[hd], %[H], %[sc]", "0", "", ST)                                                    \
        SWB_RIGHT("p1DR", "%[X]", "%[H]", "v_add_u32_e32 %[hd], %[H], %[sc]", "0", "", ST)                                                   \
        SWB_RIGHT("p1RR", "%[X]", "%[H]", "v_add_u32_dpp %[hd], %[H], %[sc]" SWB_DPP_SHR, "0", "", ST)                                       \
        "Lsw%=_end1:\n\t"                                                                                                                                                           \
        "v_swap_b32 %[H], %[X]\n"                                                                                                                                                   \
        "Lsw%=_end0:\n\t"                                                                                                                                                           \
        "s_cmp_gt_i32 %[top], %[bot]\n\t"                                                                                                                                           \
        "s_cselect_b32 %[dn], 0, 1"                                                                                                                                                 \
        : [H] "+v"(H), [X] "+v"(X), [qc] "+v"(qc), [tc] "+v"(tc), [kb] "+v"(kb), [mv] "+s"(mv), [soff] "+s"(soff),                                                                  \
          [dn] "+s"(dn), [pm] "+s"(pm), [qsel] "+s"(qsel), [tsel] "+s"(tsel), [hd] "=&v"(hd), [mm] "=&v"(mm), [sc] "=&v"(sc), [top] "=&s"(top), [bot] "=&s"(bot)                    \
        : [gap] "s"(gapS), [vmat] "v"(vmatS), [vmis] "v"(vmisS), [qb] "s"(qbits), [tb] "s"(tbits), [tbp] "s"(tbp)                                                                   \
        : "vcc", "scc", "s56", "s57", "s60", "s61", "s62", "s63", "memory");
#define SWB_STORE "s_store_dwordx4 s[60:63], %[tbp], %[soff]\n\t"
template <bool STORE>
__device__ __forceinline__ void sw_block(int32_t &H, int32_t &X, int32_t &qc, int32_t &tc, const uint64_t qbits, const uint64_t tbits, int32_t &kb,
                                         void *tbp, uint32_t &soff, uint32_t &mv, int32_t &dn, int32_t &pm, const int32_t gapS,
                                         const int32_t vmatS, const int32_t vmisS) {
    int32_t hd, mm, sc, top, bot;
    uint32_t qsel = 2u << 16, tsel = 2u << 16;   // s_bfe_u64 operand: width 2, offset 0
    if constexpr (STORE) { SW_BLOCK_ASM(SWB_STORE) } else { SW_BLOCK_ASM("") }     // the variant without mask stores exists for one measurement (FZP_SW_NO_MASKS)
}
#undef SW_BLOCK_ASM
#undef SWB_DOWN
#undef SWB_RIGHT
#undef SWB_DPP_SHL
#undef SWB_DPP_SHR

// 32 bases starting at packed index idx, as a wave-uniform 64-bit window (three scalar dword loads)
__device__ __forceinline__ uint64_t base_window(const uint32_t *__restrict__ pk, int64_t idx) {
    const int64_t w = idx >> 4;
    const uint32_t sh = (uint32_t)(idx & 15) * 2u;
    const uint64_t lo = (uint64_t)pk[w] | ((uint64_t)pk[w + 1] << 32);
    const uint64_t hi = pk[w + 2];
    const uint64_t v = sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)v);
}

template <bool STORE>
__global__ void __launch_bounds__(64) k_sw(int64_t first, int64_t count, const int32_t *__restrict__ ridx, const uint32_t *__restrict__ read_ori,
                                            const int64_t *__restrict__ ori_woff, const int32_t *__restrict__ read_len, const int32_t *__restrict__ read_ctg,
                                            const uint32_t *__restrict__ ctg_pk, const int64_t *__restrict__ ctg_woff, const int64_t *__restrict__ ctg_len,
                                            const Anchor *__restrict__ anc, const int64_t *__restrict__ tb_off, uint2 *__restrict__ tb, ulonglong2 *__restrict__ mvw,
                                            int match, int mismatch, int gap, DpInfo *__restrict__ info, int64_t *__restrict__ tbo, int64_t *__restrict__ mvo,
                                            const int32_t *__restrict__ order, int32_t prio_len, const int64_t *__restrict__ m_off, const int64_t *__restrict__ mv_off,
                                            const uint8_t *__restrict__ skip, int32_t m_stride, int32_t *__restrict__ tbs) {
    const int lane = lane_id();
    // wave-uniform on purpose: everything indexed by the read then lives in SGPRs / scalar loads
    const int64_t wq = (int64_t)blockIdx.x;   // one wave per workgroup: a finished read frees its slot at once
    if (wq >= count) return;
    // workgroups start in blockIdx order, so `order` (slots by decreasing read length) makes the launch longest-first: the last
    // waves to start are the shortest, and the tail of the grid is short however uneven the read lengths are
    const int64_t wv = order ? (int64_t)order[wq] : wq;
    // slot = candidate: the read itself for first candidates, an entry of the compacted list for second ones
    const int64_t sl = first + wv;
    if (skip && skip[sl]) return;              // the bit-sliced kernel ran this slot's extension
    const int64_t r = ridx ? ridx[sl] : sl;
    const Anchor a = anc[sl];
    // where the slot's masks and move words go in the chunk's buffers: in slot order, or (m_off) where the job planned them -- in launch order, so that the
    // streams of the reads that the bit-sliced kernel runs side by side in a wave lie side by side
    const int64_t toff = m_off ? m_off[sl] : tb_off[sl] - tb_off[first], moff = m_off ? mv_off[sl] : (toff >> 6) + wv;
    ulonglong2 *tbr = (ulonglong2 *)tb + toff;   // per step: {D mask, G mask} over the 64 band lanes
    ulonglong2 *mvr = mvw + moff;                // per 64 steps: {move bits, i0 before the chunk}
    // the records of steps 64 b .. 64 b + 63 of a slot start at record b * stride of its stream: 64 for a stream of its own; planned streams (m_off) are interleaved
    // block by block with the 63 others of their launch group (stride 64 x 64), so that what a wave of the bit-sliced kernel writes at a time lies within 64 KB
    const int32_t stride = m_off ? m_stride : 64;
    if (tbo && lane == 0) { tbo[sl] = toff; mvo[sl] = moff; if (tbs) tbs[sl] = stride; }   // element offsets into the chunk's buffers
    if (!a.aligned) { if (lane == 0) info[sl] = DpInfo{0, -1, 0, NEGV}; return; }
    const int c_idx = read_ctg[r];
    const int64_t n = read_len[r];
    // a read several times the usual length is a serial chain several times as long: its wave takes the SIMD's issue slots ahead of the
    // seven waves it shares them with (they lose little, it finishes up to 8 x sooner), so the launch does not end on one long read
    if (prio_len > 0) {
        if (n >= 4 * (int64_t)prio_len) __builtin_amdgcn_s_setprio(3);
        else if (n >= 3 * (int64_t)prio_len) __builtin_amdgcn_s_setprio(2);
        else if (n >= 2 * (int64_t)prio_len) __builtin_amdgcn_s_setprio(1);
    }
    const int32_t nq = (int32_t)(n - a.i_a);
    int64_t ntl = ctg_len[c_idx] - a.c_a;
    if (ntl > (int64_t)nq + nq / 4 + 64) ntl = (int64_t)nq + nq / 4 + 64;
    const int32_t nt = (int32_t)ntl;
    const uint32_t *qpk = read_ori + ori_woff[sl];
    const uint32_t *tpk = ctg_pk + ctg_woff[c_idx];
    const int64_t qb = a.i_a, tbase = a.c_a;
    const int32_t max_steps = nq + nt + 2;

    // state before step 0 (biased): H(-1), and X = H(-2) in the lane layout it was computed in
    int32_t H = (lane == 32 || lane == 33) ? SW_BIAS - gap : 0;
    int32_t X = lane == 32 ? SW_BIAS : 0;
    bool pdown = false;                                // move of the (virtual) step -1: RIGHT
    int32_t qc, tc;
    {
        int32_t i = lane - 33, j = 32 - lane;
        qc = (i >= 0 && i < nq) ? (int32_t)base_at(qpk, qb + i) : 4;
        tc = (j >= 0 && j < nt) ? (int32_t)base_at(tpk, tbase + j) : 5;
    }
    int32_t bs = 0, bt = -1;
    int32_t i0 = -33, t = 0, qpos = 31, tpos = 33;
    uint64_t mvacc = 0;
    bool down = true;
    BaseStream qs, ts;
    qs.init(qpk, qb + 31);
    ts.init(tpk, tbase + 33);
    bool done = false;
    while (!done) {
        t = __builtin_amdgcn_readfirstlane(t); i0 = __builtin_amdgcn_readfirstlane(i0);
        // how many steps can run with every lane strictly inside the matrix?  Each step advances i0 or
        // lane 0's column by one, so min(rows left, columns left) steps are safe once the band is inside.
        int32_t safe = 0;
        if (t >= 64 && i0 >= 0 && (t - 1) - (i0 + 63) >= 0) {
            const int32_t rows_left = nq - 1 - (i0 + 63), cols_left = nt - 1 - ((t - 1) - i0);
            safe = min(rows_left, cols_left) - 1;      // (- 1: after `rows_left` DOWN moves lane 63 sits ON the last row -- a terminal candidate, which only the checked steps look at)
        }
        if (safe > 0) {
            // ---- interior: asm blocks of <= 32 steps
            int32_t qpos_i = i0 + 64, tpos_i = t - i0;                 // next bases to enter at lane 63 / lane 0
            const int32_t vmatS = match, vmisS = -mismatch;
            const int32_t gapS = __builtin_amdgcn_readfirstlane(gap);
            int32_t dn = down ? 1 : 0;
            int32_t pm = pdown ? 1 : 0;
            while (safe > 0) {
                // (readfirstlane: these are wave-uniform, but hipcc's divergence analysis cannot always prove it)
                const uint64_t qbits = base_window(qpk, __builtin_amdgcn_readfirstlane((int32_t)qb + qpos_i));
                const uint64_t tbits = base_window(tpk, tbase + __builtin_amdgcn_readfirstlane(tpos_i));
                const int32_t n_steps = __builtin_amdgcn_readfirstlane(min(safe, 32 - (t & 31)));
                safe -= n_steps;
                uint32_t mv = 0;
                int32_t kb = 0;
                dn = __builtin_amdgcn_readfirstlane(dn);
                pm = __builtin_amdgcn_readfirstlane(pm);
                // the store offset doubles as the block's step counter: it starts 16 * n_steps below 2^31 (the base pointer makes up for
                // it), and the add that would carry it past 2^31 -- the signed overflow of s_addk_i32 -- ends the block
                uint32_t soff = 0x80000000u - 16u * (uint32_t)n_steps;
                const int32_t t_ = __builtin_amdgcn_readfirstlane(t);      // (a block never crosses a multiple of 32 steps: its records are contiguous)
                void *tbp = (void *)((char *)tbr + (((int64_t)(t_ >> 6) * stride + (t_ & 63)) * 16 - (int64_t)soff));
                sw_block<STORE>(H, X, qc, tc, qbits, tbits, kb, tbp, soff, mv, dn, pm, gapS, vmatS, vmisS);
                // (v1.5: no cell of an interior block is a border cell, so the block has no terminal candidates to report)
                pm = (int32_t)(mv & 1u);                                 // the block's last move (the steps no longer keep it up to date)
                const int32_t nd = __popc(mv);                          // DOWN moves of the block (mv holds exactly n_steps bits)
                i0 = __builtin_amdgcn_readfirstlane(i0 + nd);
                qpos_i += nd; tpos_i += n_steps - nd;
                mvacc |= (uint64_t)(__brev(mv) >> (32 - n_steps)) << (t & 63);   // step s of the block -> bit (t + s) & 63
                t = __builtin_amdgcn_readfirstlane(t + n_steps);         // (uniform; said aloud so that the counter and everything hanging off it stay scalar)
                if ((t & 63) == 0) SW_FLUSH(false)
            }
            down = dn != 0;
            pdown = pm != 0;
            // back to the checked variant: scalar base streams resume at the current positions
            qpos = i0 + 64; tpos = t - i0;
            qs.init(qpk, qb + qpos);
            ts.init(tpk, tbase + tpos);
        } else {
            SW_STEP()
            if ((t & 63) == 0) SW_FLUSH(false)
            if (i0 > nq - 1 || (t - 1) - (i0 + 63) > nt - 1 || t >= max_steps) done = true;
        }
    }
    if ((t & 63) != 0) SW_FLUSH(true)
    asm volatile("s_dcache_wb" ::: "memory");   // the masks went through the scalar cache
    // best cell: max score, then earliest step, then lowest lane
    int32_t s_b = bs, t_b = bt < 0 ? 0x7fffffff : bt, l_b = lane;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        int32_t so = __shfl_xor(s_b, d, 64), to = __shfl_xor(t_b, d, 64), lo = __shfl_xor(l_b, d, 64);
        bool take = so > s_b || (so == s_b && (to < t_b || (to == t_b && lo < l_b)));
        if (take) { s_b = so; t_b = to; l_b = lo; }
    }
    if (lane == 0) info[sl] = DpInfo{t, t_b == 0x7fffffff ? -1 : t_b, l_b, s_b - SW_BIAS};
}
#undef SW_STEP
#undef SW_FLUSH

// ---- K1 hot kernel, bit-sliced form: one READ per lane (64 extensions per wave), the 64 cells of a read's anti-diagonal in the 64 bits of a register
// pair, the DP values as differences in three bit planes (fzp_swb_core.h: the cell function, checked on the host against the scalar twin).  Same spec,
// same outputs as k_sw -- per step the masks {D, G}, per 64 steps the move record, per read the terminal -- from a sixth of the instructions: k_sw spends
// ~14 wave instructions on the 64 cells of one read's step, this kernel ~130 on the 4 096 cells of 64 reads' steps.  What the scores were needed for is
// tracked apart: the steering compares the two edge cells of the band, whose scores advance by the difference the step just computed for them; the
// terminal's candidates lie on the last row / last column, whose scores likewise accumulate while the band sweeps along them (v1.5 made that enough).
// A wave's reads advance in lockstep (the step counter is wave-uniform); a read that is done idles its lane, so the launch lists reads of similar
// length together.  A step costs a read's lane ~10x the latency it costs a k_sw wave: extensions longer than the caller's limit stay with k_sw, as do
// those shorter than the band (nq or nt < 64).
struct LaneStream {                // upcoming bases of one sequence, per lane: cur holds `have` (>= 1) of them, pend the 16 after those
    const uint32_t *pk;
    uint64_t cur;
    uint32_t pend, w, lim;         // w: the word `pend` holds; lim: words the sequence has (padding included) -- a lane whose extension is over, or that idles
    int32_t have;                  // beside longer ones, keeps refilling and must not walk off its sequence (it reads its last word again)
    __device__ __forceinline__ void init(const uint32_t *pk_, int64_t idx, uint32_t lim_) {
        const uint32_t w0 = (uint32_t)(idx >> 4);
        const uint32_t sh = (uint32_t)(idx & 15) * 2u;
        pk = pk_; lim = lim_;
        cur = ((uint64_t)pk[w0] | ((uint64_t)pk[w0 + 1] << 32)) >> sh;
        have = 32 - (int32_t)(idx & 15);
        w = w0 + 2;
        pend = pk[w < lim ? w : lim - 1u];
    }
    __device__ __forceinline__ void refill() {              // every 16 steps (a step takes at most one base); branch-free: a lane that needs nothing loads its word again
        const bool m = have <= 16;
        cur |= m ? (uint64_t)pend << (2 * have) : 0ull;
        have += m ? 16 : 0;
        w += m ? 1u : 0u;
        pend = pk[w < lim ? w : lim - 1u];
    }
    __device__ __forceinline__ uint32_t pop(uint32_t en) {  // en = 1: take a base; 0: leave the stream as it is (returns 0)
        const uint32_t c = (uint32_t)cur & (0u - en) & 3u;
        cur >>= 2u * en;
        have -= (int32_t)en;
        return c;
    }
};

struct LaneStreamL {
    const uint32_t *pk;            // the sequence's words
    uint32_t *ring;                // this lane's column of its stream's ring
    uint64_t cur;
    uint32_t pend, rd, wr, gw, lim;
    uint32_t hold[16], hold_base;  // sixteen words on their way from HBM to the ring (bulk)
    bool held;
    int32_t have;
    __device__ __forceinline__ void init(const uint32_t *pk_, int64_t idx, uint32_t lim_, uint32_t *ring_) {
        pk = pk_; ring = ring_; lim = lim_; held = false; hold_base = 0;
#pragma unroll
        for (int q = 0; q < 16; q++) hold[q] = 0u;
        const uint32_t w = (uint32_t)(idx >> 4);
        const uint32_t sh = (uint32_t)(idx & 15) * 2u;
        cur = ((uint64_t)pk[w] | ((uint64_t)pk[w + 1] << 32)) >> sh;
        have = 32 - (int32_t)(idx & 15);
        gw = w + 2;
        for (int q0 = 0; q0 < 64; q0 += 16) {
            uint32_t v[16];
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = 0u;
            if (gw + q0 < lim) {      // (a short sequence -- the backward extensions' -- ends within the first words: nothing to fetch for the rest of the ring)
#pragma unroll
                for (int q = 0; q < 16; q++) { const uint32_t ix = gw + q0 + q; v[q] = pk[ix < lim ? ix : lim - 1u]; }      // (all sixteen in flight together)
            }
#pragma unroll
            for (int q = 0; q < 16; q++) ring[(q0 + q) * 64] = gw + q0 + q < lim ? v[q] : 0u;
        }
        gw += 64; wr = 64; rd = 1;
        pend = ring[0];
    }
    __device__ __forceinline__ void refill() {
        const bool m = have <= 16;
        cur |= m ? (uint64_t)pend << (2 * have) : 0ull;
        have += m ? 16 : 0;
        const uint32_t nx = ring[(rd & 63u) * 64];
        pend = m ? nx : pend;
        rd += m ? 1u : 0u;
    }
    // every 256 steps (at most 16 words leave the ring in between).  What a call loads goes into the ring at the NEXT call: by then more than 63 vector-memory operations
    // have been issued behind the loads, which is more than can be outstanding -- no wait is needed to use them, and none is spent behind the mask stores.
    __device__ __forceinline__ void bulk() {
        if (held) {
#pragma unroll
            for (int q = 0; q < 16; q++) ring[((wr + q) & 63u) * 64] = hold_base + q < lim ? hold[q] : 0u;
            wr += 16u;
        }
        held = wr - rd <= 32u;
        if (held) {
            hold_base = gw;
#pragma unroll
            for (int q = 0; q < 16; q++) { const uint32_t ix = gw + q; hold[q] = pk[ix < lim ? ix : lim - 1u]; }
            gw += 16u;
        }
    }
    __device__ __forceinline__ uint32_t pop(uint32_t en) {
        const uint32_t c = (uint32_t)cur & (0u - en) & 3u;
        cur >>= 2u * en;
        have -= (int32_t)en;
        return c;
    }
};

template <class STREAM>
struct SwbLaneT {                  // one extension's state (a lane's registers)
    swb::Planes P, Q;              // difference planes of the previous anti-diagonal
    uint64_t R0, R1, C0, C1;       // base windows as bit planes: bit k = read base i0 + k / contig base t - i0 - k
    STREAM qs, ts;
    int32_t i0, E2, sv0;           // E2 = (score of lane 63's cell - score of lane 0's) / 2;  sv0 = sum of lane 0's difference codes: its score is -259 + 2 sv0 - 3 (t + 1)
    uint32_t down, pdown;
    uint64_t mvacc;
};

// one DP step of every lane's extension.  CHECKED = false: the 64 steps of an interior block -- no lane of the wave can reach a border of its matrix in them, so
// there is nothing to validate, no terminal candidate and no end; lanes whose extension is over run along on their stale state (nothing of theirs is stored).
// CHECKED = true: validity of the bases near the ends, terminal candidates, the end of the extension.
template <bool CHECKED, class LANE>
__device__ __forceinline__ void swb_step(LANE &L, const int32_t t, ulonglong2 &rec, const int32_t nq, const int32_t nt, const int32_t max_steps, bool &active,
                                         bool &row_on, bool &col_on, int32_t &Hrow, int32_t &Hcol, int32_t &best, int32_t &bt, int32_t &bl, int32_t &steps) {
    using namespace swb;
    const uint32_t sd = L.down, sr = 1u - sd;
    L.i0 += (int32_t)sd;
    {   // the windows slide: a new read base enters at lane 63 (DOWN), a new contig base at lane 0 (RIGHT)
        const uint32_t cq = L.qs.pop(sd), ct = L.ts.pop(sr);
        L.R0 = (L.R0 >> sd) | ((uint64_t)(cq << 31) << 32); L.R1 = (L.R1 >> sd) | ((uint64_t)((cq << 30) & 0x80000000u) << 32);
        L.C0 = (L.C0 << sr) | (uint64_t)(ct & 1u); L.C1 = (L.C1 << sr) | (uint64_t)(ct >> 1);
    }
    const Planes p = {L.P.v0 << sr, L.P.v1 << sr, L.P.v2 << sr}, q = {L.Q.v0 >> sd, L.Q.v1 >> sd, L.Q.v2 >> sd};
    uint64_t xm = (L.R0 ^ L.C0) | (L.R1 ^ L.C1);
    const int32_t kr = nq - 1 - L.i0, kc = t - (nt - 1) - L.i0;      // lanes of the last row / the last column
    if (CHECKED) {   // bases past the read's / the window's end never match
        const int32_t nv = kr + 1;                                    // lanes k < nv hold read bases
        const uint64_t bad_r = nv >= 64 ? 0ull : (nv <= 0 ? ~0ull : ~0ull << nv);
        const uint64_t bad_c = kc <= 0 ? 0ull : (kc >= 64 ? ~0ull : ~(~0ull << kc));   // lanes k >= kc hold contig bases
        xm |= bad_r | bad_c;
    }
    const uint64_t f = ((uint64_t)((sd & L.pdown) << 31) << 32) | (uint64_t)(sr & (1u - L.pdown));      // two moves the same way: the edge lane's diagonal predecessor is outside the band
    uint64_t D, G;
    cells<uint64_t>(xm, f, (uint64_t)0 - (uint64_t)sd, p, q, &L.P, &L.Q, &D, &G);
    rec = make_ulonglong2(D, G);
    L.mvacc |= (uint64_t)sd << (t & 63);
    // the edge cells' scores: every lane's cell moved down (its vertical difference) or right (its horizontal one)
    const uint32_t dm = 0u - sd;
    const uint32_t xl0 = ((uint32_t)L.Q.v0 & dm) | ((uint32_t)L.P.v0 & ~dm), xl1 = ((uint32_t)L.Q.v1 & dm) | ((uint32_t)L.P.v1 & ~dm), xl2 = ((uint32_t)L.Q.v2 & dm) | ((uint32_t)L.P.v2 & ~dm);
    const uint32_t xh0 = ((uint32_t)(L.Q.v0 >> 32) & dm) | ((uint32_t)(L.P.v0 >> 32) & ~dm), xh1 = ((uint32_t)(L.Q.v1 >> 32) & dm) | ((uint32_t)(L.P.v1 >> 32) & ~dm),
                   xh2 = ((uint32_t)(L.Q.v2 >> 32) & dm) | ((uint32_t)(L.P.v2 >> 32) & ~dm);
    const int32_t v0 = (int32_t)((xl0 & 1u) | ((xl1 & 1u) << 1) | ((xl2 & 1u) << 2)), v63 = (int32_t)((xh0 >> 31) | ((xh1 >> 31) << 1) | ((xh2 >> 31) << 2));
    L.sv0 += v0;
    L.E2 += v63 - v0;
    if (CHECKED) {   // terminal: the best valid cell of the last row / last column, their scores by differences along them
        if (active) {
            const int32_t S0 = -259 + 2 * L.sv0 - 3 * (t + 1);
            const bool kc_in = kc >= 0 && kc <= 63, kr_in = kr >= 0 && kr <= 63;
            const bool col_start = !col_on && sr && kc == 0, row_start = !row_on && sd && kr == 63;
            if (col_start) Hcol = S0; else if (col_on && kc_in) Hcol += 2 * value_at(L.Q, kc & 63) - 3;
            if (row_start) Hrow = S0 + 2 * L.E2; else if (row_on && kr_in) Hrow += 2 * value_at(L.P, kr & 63) - 3;
            col_on = col_on || col_start; row_on = row_on || row_start;
            { const int32_t i = L.i0 + kc; if (col_on && kc_in && i >= 0 && i < nq && Hcol > best) { best = Hcol; bt = t; bl = kc; } }
            { const int32_t jj = t - (nq - 1); if (row_on && kr_in && jj >= 0 && jj < nt && Hrow > best) { best = Hrow; bt = t; bl = kr; } }
            if (L.i0 > nq - 1 || t - (L.i0 + 63) > nt - 1 || t + 1 >= max_steps) { active = false; steps = t + 1; }
        }
    }
    L.pdown = sd;
    L.down = (CHECKED && (t + 1) < 64) ? (uint32_t)(((t + 1) & 1) == 0) : (uint32_t)(L.E2 >= 0);
}

constexpr int SWB_WPG = 1;        // waves per workgroup of k_swb (four measured: 10.1 against 9.7 ms)
constexpr int SWB_GROUP = 8;      // steps whose mask records leave together (64 B per lane)
// RING: the base streams through rings in LDS (LaneStreamL) or straight from HBM (LaneStream; FZP_SWB_NO_RING, for comparisons)
template <bool RING>
__global__ void __launch_bounds__(256) k_swb(int64_t first, int64_t count, const int32_t *__restrict__ list, const int32_t *__restrict__ ridx, const uint32_t *__restrict__ read_ori,
                                             const int64_t *__restrict__ ori_woff, const int32_t *__restrict__ read_len, const int32_t *__restrict__ read_ctg,
                                             const uint32_t *__restrict__ ctg_pk, const int64_t *__restrict__ ctg_woff, const int64_t *__restrict__ ctg_len,
                                             const Anchor *__restrict__ anc, const int64_t *__restrict__ tb_off, uint2 *__restrict__ tb, ulonglong2 *__restrict__ mvw,
                                             DpInfo *__restrict__ info, int64_t *__restrict__ tbo, int64_t *__restrict__ mvo, const int64_t *__restrict__ m_off, const int64_t *__restrict__ mv_off, uint8_t *__restrict__ handled, int32_t steps_limit, int dbg,
                                             int32_t m_stride, int32_t *__restrict__ tbs, uint32_t *__restrict__ start_flag, uint32_t start_val) {
    if (start_flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(start_flag, start_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);      // "on the chip" (k_wait_started)
    using namespace swb;
    // dbg: MEASUREMENT switches (FZP_SWB_DBG, tools/runs/swb_probe.py; the results of such a run are not used): bit 0 = no mask stores, bit 1 = no stream refills
    // (workgroups of one wave: four-wave workgroups, which suit k_swb2, put 256 mask streams on a CU and cost this kernel 10 % -- address translation again)
    const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t wv = li < count ? list[li] : -1;       // the lane's slot in the chunk, -1: none
    bool active = wv >= 0;
    const int64_t sl = first + (active ? wv : 0);
    const int64_t r = ridx ? ridx[sl] : sl;
    const Anchor a = anc[sl];
    // the lanes of a wave stand side by side in the launch list, and so do their mask streams (m_off: planned in launch order): 64 streams scattered over the
    // buffer cost twice the time in address translation alone
    const int64_t toff = m_off ? m_off[sl] : tb_off[sl] - tb_off[first], moff = m_off ? mv_off[sl] : (toff >> 6) + (active ? wv : 0);
    ulonglong2 *tbr = (ulonglong2 *)tb + toff;
    ulonglong2 *mvr = mvw + moff;
    const int c_idx = read_ctg[r];
    const int64_t n = read_len[r];
    const int32_t nq = (int32_t)(n - a.i_a);
    int64_t ntl = ctg_len[c_idx] - a.c_a;
    if (ntl > (int64_t)nq + nq / 4 + 64) ntl = (int64_t)nq + nq / 4 + 64;
    const int32_t nt = (int32_t)ntl;
    if (handled) {   // the caller did not sort the slots by kernel: this one takes what fits it and says so, k_sw runs the rest
        const bool mine = active && (!a.aligned || (nq >= 64 && nt >= 64 && nq + nt + 2 <= steps_limit));
        if (active) handled[sl] = mine ? 1 : 0;
        active = mine;
    }
    const int32_t stride = m_off ? m_stride : 64;        // records from one 64-step block of this stream to the next (k_sw)
    if (active && tbo) { tbo[sl] = toff; mvo[sl] = moff; if (tbs) tbs[sl] = stride; }
    if (active && !a.aligned) { info[sl] = DpInfo{0, -1, 0, NEGV}; active = false; }
    const uint32_t *qpk = read_ori + ori_woff[sl];
    const uint32_t *tpk = ctg_pk + ctg_woff[c_idx];
    const int64_t qb = a.i_a, tbase = a.c_a;
    const int32_t max_steps = nq + nt + 2;
    SwbLaneT<typename std::conditional<RING, LaneStreamL, LaneStream>::type> L;
    // step -1: the anti-diagonal i + j = -1 of the virtual border, lane k = cell (k - 33, 32 - k): Pv = 0 where j >= 0 (k <= 32) else 4, Qv = 0 where i >= 0 (k >= 33) else 4
    L.P = {0, 0, ~0ull << 33}; L.Q = {0, 0, (1ull << 33) - 1};
    L.R0 = L.R1 = L.C0 = L.C1 = 0;
    for (int k = 33; k < 64; k++) { const uint32_t c = base_at(qpk, qb + (k - 33)); L.R0 |= (uint64_t)(c & 1u) << k; L.R1 |= (uint64_t)(c >> 1) << k; }
    for (int k = 0; k <= 32; k++) { const uint32_t c = base_at(tpk, tbase + (32 - k)); L.C0 |= (uint64_t)(c & 1u) << k; L.C1 |= (uint64_t)(c >> 1) << k; }
    __shared__ uint32_t srng[RING ? 2 * 64 * 64 : 1];                      // the two streams' rings (32 KB per wave)
    if constexpr (RING) {
        L.qs.init(qpk, qb + 31, (uint32_t)((qb + nq + 15) >> 4) + 1u, srng + threadIdx.x);
        L.ts.init(tpk, tbase + 33, (uint32_t)((tbase + nt + 15) >> 4) + 1u, srng + 64 * 64 + threadIdx.x);
    } else {
        L.qs.init(qpk, qb + 31, (uint32_t)((qb + nq + 15) >> 4) + 1u);
        L.ts.init(tpk, tbase + 33, (uint32_t)((tbase + nt + 15) >> 4) + 1u);
    }
    L.i0 = -33; L.E2 = 8; L.sv0 = 0;                            // at step -1 from the border's closed form: lane 0's cell scores -259, lane 63's -243
    L.down = 1; L.pdown = 0;
    bool row_on = false, col_on = false;
    int32_t Hrow = 0, Hcol = 0, best = NEGV, bt = -1, bl = 0, steps = 0;
    int32_t t = 0;                                            // wave-uniform
    while (__ballot(active)) {
        const bool blk_active = active;
        const int32_t i0_blk = L.i0;
        L.mvacc = 0;
        // an interior block?  every running lane more than 64 steps away from its last row and its last column (a step brings either one closer by at most one)
        const bool far = !active || (nq - 1 - (L.i0 + 63) > 64 && nt - 1 - (t - L.i0) > 64);
        const bool interior = t >= 64 && __ballot(!far) == 0ull;
        if constexpr (RING) { if ((t & 255) == 0 && t > 0) { L.qs.bulk(); L.ts.bulk(); } }
        for (int g8 = 0; g8 < 64 / SWB_GROUP; g8++) {
            const bool grp_active = active;
            ulonglong2 rec[SWB_GROUP];
            if (interior) {
#pragma unroll
                for (int s8 = 0; s8 < SWB_GROUP; s8++) { swb_step<false>(L, t, rec[s8], nq, nt, max_steps, active, row_on, col_on, Hrow, Hcol, best, bt, bl, steps); t++; }
            } else {
#pragma unroll
                for (int s8 = 0; s8 < SWB_GROUP; s8++) { swb_step<true>(L, t, rec[s8], nq, nt, max_steps, active, row_on, col_on, Hrow, Hcol, best, bt, bl, steps); t++; }
            }
            // the streams top up every 16 steps, and they do it HERE, ahead of a group's stores: taking the word loaded 16 steps ago means waiting on the vector-memory
            // counter, which also counts the mask stores -- at this point the youngest of those are 8 steps old and done, right behind a group they would be in flight
            if ((g8 & (16 / SWB_GROUP - 1)) == 16 / SWB_GROUP - 1 && !(dbg & 2)) { L.qs.refill(); L.ts.refill(); }
            if (grp_active && !(dbg & 1)) {
#pragma unroll
                for (int s8 = 0; s8 < SWB_GROUP; s8++) tbr[(int64_t)((t - SWB_GROUP) >> 6) * stride + ((t - SWB_GROUP) & 63) + s8] = rec[s8];
            }
        }
        if (blk_active) {
            uint32_t glane = 32u;
            if ((t & (TBS_SEG - 1)) == 0 && active) {      // step t-1 tops a trace-back segment: the lane of its best score is where that segment's walker starts
                int32_t run = 0, bestv = 0;
                glane = 0u;
                for (int k = 0; k < 63; k++) {            // score(lane k+1) - score(lane k) = 2 (Qv[k+1] - Pv[k])
                    run += value_at(L.Q, k + 1) - value_at(L.P, k);
                    if (run > bestv) { bestv = run; glane = (uint32_t)(k + 1); }
                }
            }
            mvr[(t - 1) >> 6] = make_ulonglong2(L.mvacc, (uint64_t)(uint32_t)i0_blk | ((uint64_t)glane << 32));
            if (!active) info[sl] = DpInfo{steps, bt, bl, bt >= 0 ? best : NEGV};
        }
    }
}

// the two DP kernels of a launch run side by side, and it matters which gets onto the chip first: with the bit-sliced kernel's few hundred waves placed first (one to
// a SIMD, spread over the CUs) and the wave-per-read kernel's thousands filling in around them the pair took 16.5 ms on reads of real shape, the other way round 22 ms,
// left to race one or the other.  So the wave-per-read kernel's stream holds this one-thread kernel first, which returns when the bit-sliced kernel's first workgroup has
// said it runs (or after 5 million ticks of the 100 MHz counter, whatever happened: a bound, not a wait anybody should see).
__global__ void k_wait_started(const uint32_t *flag, uint32_t val) {
    const uint64_t t0 = __builtin_readcyclecounter();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < val && __builtin_readcyclecounter() - t0 < 5000000ull) __builtin_amdgcn_s_sleep(16);
}

// ---- the same DP with the band split over a PAIR of lanes (fzp_swb_core.h, Half): 32 reads per wave, half the instructions per step on a wave's
// critical path and twice the waves -- at the bench's job size the chip holds about one DP wave per SIMD, which makes a step's latency (instructions x the
// ~5 cycles a lone wave needs per instruction), not the issue rate, what bounds the launch.  The low lane of a pair owns the contig stream, the high lane
// the read stream; what crosses the middle of the band and the two edge differences the steering compares go through DPP quad swaps.
__device__ __forceinline__ uint32_t pair_swap(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)v, 0xB1, 0xf, 0xf, false); }   // quad_perm [1,0,3,2]

struct SwbPair {
    swb::Half h;
    LaneStream ss;                 // low lane: the contig's bases, high lane: the read's
    int32_t i0, E2, sv0;
    uint32_t down, pdown;
    uint64_t mvacc;
};

template <bool CHECKED>
__device__ __forceinline__ void swb2_step(SwbPair &L, const uint32_t is_hi, const int32_t t, uint2 &rec, const int32_t nq, const int32_t nt, const int32_t max_steps, bool &active,
                                          bool &row_on, bool &col_on, int32_t &Hrow, int32_t &Hcol, int32_t &best, int32_t &bt, int32_t &bl, int32_t &steps) {
    using namespace swb;
    const uint32_t sd = L.down, sr = 1u - sd;
    const uint32_t my = is_hi ? sd : sr, pmy = is_hi ? L.pdown : 1u - L.pdown;
    L.i0 += (int32_t)sd;
    const HalfOut o = half_out(L.h);
    const HalfOut in = {pair_swap(o.a0), pair_swap(o.a1), pair_swap(o.a2), pair_swap(o.w0), pair_swap(o.w1)};
    const uint32_t base = L.ss.pop(my);
    const int32_t kr = nq - 1 - L.i0, kc = t - (nt - 1) - L.i0;      // lanes (cells) of the last row / the last column
    uint32_t bad = 0;
    if (CHECKED) {   // bases past the read's / the window's end never match; low lane: bit c = cell c, high lane: bit b = cell 63 - b
        const int32_t nv = kr + 1;                                    // cells < nv hold read bases, cells >= kc contig bases
        const uint32_t br_lo = nv >= 32 ? 0u : (nv <= 0 ? ~0u : ~0u << nv), br_hi = nv >= 64 ? 0u : (nv <= 32 ? ~0u : (1u << (64 - nv)) - 1u);
        const uint32_t bc_lo = kc <= 0 ? 0u : (kc >= 32 ? ~0u : (1u << kc) - 1u), bc_hi = kc <= 32 ? 0u : (kc >= 64 ? ~0u : ~0u << (64 - kc));
        bad = is_hi ? (br_hi | bc_hi) : (br_lo | bc_lo);
    }
    uint32_t D, G;
    half_step(L.h, my, base, in, my & pmy, bad, &D, &G);
    {   // the record of the step: {D, G} over the 64 cells; the low lane stores D (both halves), the high lane G
        const uint32_t Dn = is_hi ? __brev(D) : D, Gn = is_hi ? __brev(G) : G;      // the high lane's bits are mirrored
        const uint32_t got = pair_swap(is_hi ? Dn : Gn);
        rec = is_hi ? make_uint2(got, Gn) : make_uint2(Dn, got);
    }
    L.mvacc |= (uint64_t)sd << (t & 63);
    const int32_t v = half_edge(L.h, my), pv = (int32_t)pair_swap((uint32_t)v);
    const int32_t v0 = is_hi ? pv : v, v63 = is_hi ? v : pv;
    L.sv0 += v0;
    L.E2 += v63 - v0;
    if (CHECKED) {   // terminal: the best valid cell of the last row / last column (see swb_step); P at cell c: the low lane's A / the high lane's B, Q: B / A
        if (active) {
            const int32_t S0 = -259 + 2 * L.sv0 - 3 * (t + 1);
            const bool kc_in = kc >= 0 && kc <= 63, kr_in = kr >= 0 && kr <= 63;
            const bool col_start = !col_on && sr && kc == 0, row_start = !row_on && sd && kr == 63;
            int32_t pq = 0, pp = 0;
            if (kc_in && (uint32_t)(kc >> 5) == is_hi) pq = is_hi ? value_at(L.h.A, 63 - kc) : value_at(L.h.B, kc);
            if (kr_in && (uint32_t)(kr >> 5) == is_hi) pp = is_hi ? value_at(L.h.B, 63 - kr) : value_at(L.h.A, kr);
            pq += (int32_t)pair_swap((uint32_t)pq); pp += (int32_t)pair_swap((uint32_t)pp);
            if (col_start) Hcol = S0; else if (col_on && kc_in) Hcol += 2 * pq - 3;
            if (row_start) Hrow = S0 + 2 * L.E2; else if (row_on && kr_in) Hrow += 2 * pp - 3;
            col_on = col_on || col_start; row_on = row_on || row_start;
            { const int32_t i = L.i0 + kc; if (col_on && kc_in && i >= 0 && i < nq && Hcol > best) { best = Hcol; bt = t; bl = kc; } }
            { const int32_t jj = t - (nq - 1); if (row_on && kr_in && jj >= 0 && jj < nt && Hrow > best) { best = Hrow; bt = t; bl = kr; } }
            if (L.i0 > nq - 1 || t - (L.i0 + 63) > nt - 1 || t + 1 >= max_steps) { active = false; steps = t + 1; }
        }
    }
    L.pdown = sd;
    L.down = (CHECKED && (t + 1) < 64) ? (uint32_t)(((t + 1) & 1) == 0) : (uint32_t)(L.E2 >= 0);
}

__global__ void __launch_bounds__(256) k_swb2(int64_t first, int64_t count, const int32_t *__restrict__ list, const int32_t *__restrict__ ridx, const uint32_t *__restrict__ read_ori,
                                              const int64_t *__restrict__ ori_woff, const int32_t *__restrict__ read_len, const int32_t *__restrict__ read_ctg,
                                              const uint32_t *__restrict__ ctg_pk, const int64_t *__restrict__ ctg_woff, const int64_t *__restrict__ ctg_len,
                                              const Anchor *__restrict__ anc, const int64_t *__restrict__ tb_off, uint2 *__restrict__ tb, ulonglong2 *__restrict__ mvw,
                                              DpInfo *__restrict__ info, int64_t *__restrict__ tbo, int64_t *__restrict__ mvo, const int64_t *__restrict__ m_off, const int64_t *__restrict__ mv_off,
                                              uint8_t *__restrict__ handled, int32_t steps_limit, int32_t m_stride, int32_t *__restrict__ tbs, uint32_t *__restrict__ start_flag, uint32_t start_val) {
    using namespace swb;
    if (start_flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(start_flag, start_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t is_hi = threadIdx.x & 1u;
    const bool lo = is_hi == 0u;
    // workgroups of four waves -- independent of each other, no LDS, no barrier: a workgroup's waves are spread over its CU's four SIMDs, single-wave workgroups are not
    // (625 of them on 256 CUs ran two to a SIMD here and there -- 12.2 ms instead of 7.4 -- and a SIMD shared by two of these waves runs each at little more than half speed)
    const int64_t li = (int64_t)blockIdx.x * (blockDim.x >> 1) + (threadIdx.x >> 1);
    const int64_t wv = li < count ? list[li] : -1;       // the pair's slot in the chunk, -1: none
    bool active = wv >= 0;
    const int64_t sl = first + (active ? wv : 0);
    const int64_t r = ridx ? ridx[sl] : sl;
    const Anchor a = anc[sl];
    const int64_t toff = m_off ? m_off[sl] : tb_off[sl] - tb_off[first], moff = m_off ? mv_off[sl] : (toff >> 6) + (active ? wv : 0);
    uint2 *tbr = tb + 2 * toff + is_hi;                  // per step 16 B {D, G}: this lane's 8 of them
    ulonglong2 *mvr = mvw + moff;
    const int c_idx = read_ctg[r];
    const int64_t n = read_len[r];
    const int32_t nq = (int32_t)(n - a.i_a);
    int64_t ntl = ctg_len[c_idx] - a.c_a;
    if (ntl > (int64_t)nq + nq / 4 + 64) ntl = (int64_t)nq + nq / 4 + 64;
    const int32_t nt = (int32_t)ntl;
    if (handled) {   // the caller did not sort the slots by kernel: this one takes what fits it and says so, k_sw runs the rest
        const bool mine = active && (!a.aligned || (nq >= 64 && nt >= 64 && nq + nt + 2 <= steps_limit));
        if (active && lo) handled[sl] = mine ? 1 : 0;
        active = mine;
    }
    const int32_t stride = m_off ? m_stride : 64;
    if (active && tbo && lo) { tbo[sl] = toff; mvo[sl] = moff; if (tbs) tbs[sl] = stride; }
    if (active && !a.aligned) { if (lo) info[sl] = DpInfo{0, -1, 0, NEGV}; active = false; }
    const uint32_t *qpk = read_ori + ori_woff[sl];
    const uint32_t *tpk = ctg_pk + ctg_woff[c_idx];
    const int64_t qb = a.i_a, tbase = a.c_a;
    const int32_t max_steps = nq + nt + 2;
    SwbPair L;
    // step -1 (see k_swb): P = 4 on cells >= 33, Q = 4 on cells <= 32; the read's bases 0..30 on cells 33..63, the contig's 32..0 on cells 0..32
    if (lo) {
        L.h.A = {0, 0, 0}; L.h.B = {0, 0, ~0u};
        L.h.Wo0 = L.h.Wo1 = 0; L.h.Wm0 = L.h.Wm1 = 0;
        for (int c = 0; c < 32; c++) { const uint32_t b = base_at(tpk, tbase + (32 - c)); L.h.Wm0 |= (b & 1u) << c; L.h.Wm1 |= (b >> 1) << c; }
        L.ss.init(tpk, tbase + 33, (uint32_t)((tbase + nt + 15) >> 4) + 1u);
    } else {
        L.h.A = {0, 0, 1u << 31}; L.h.B = {0, 0, ~0u >> 1};
        L.h.Wm0 = L.h.Wm1 = 0;
        for (int c = 33; c < 64; c++) { const uint32_t b = base_at(qpk, qb + (c - 33)); L.h.Wm0 |= (b & 1u) << (63 - c); L.h.Wm1 |= (b >> 1) << (63 - c); }
        { const uint32_t b = base_at(tpk, tbase); L.h.Wo0 = (b & 1u) << 31; L.h.Wo1 = (b >> 1) << 31; }
        L.ss.init(qpk, qb + 31, (uint32_t)((qb + nq + 15) >> 4) + 1u);
    }
    L.i0 = -33; L.E2 = 8; L.sv0 = 0;
    L.down = 1; L.pdown = 0;
    bool row_on = false, col_on = false;
    int32_t Hrow = 0, Hcol = 0, best = NEGV, bt = -1, bl = 0, steps = 0;
    int32_t t = 0;                                            // wave-uniform
    while (__ballot(active)) {
        const bool blk_active = active;
        const int32_t i0_blk = L.i0;
        L.mvacc = 0;
        const bool far = !active || (nq - 1 - (L.i0 + 63) > 64 && nt - 1 - (t - L.i0) > 64);
        const bool interior = t >= 64 && __ballot(!far) == 0ull;
        for (int g8 = 0; g8 < 8; g8++) {
            const bool grp_active = active;
            uint2 rec[8];
            if (interior) {
#pragma unroll
                for (int s8 = 0; s8 < 8; s8++) { swb2_step<false>(L, is_hi, t, rec[s8], nq, nt, max_steps, active, row_on, col_on, Hrow, Hcol, best, bt, bl, steps); t++; }
            } else {
#pragma unroll
                for (int s8 = 0; s8 < 8; s8++) { swb2_step<true>(L, is_hi, t, rec[s8], nq, nt, max_steps, active, row_on, col_on, Hrow, Hcol, best, bt, bl, steps); t++; }
            }
            if (g8 & 1) L.ss.refill();      // (ahead of the stores: see k_swb)
            if (grp_active) {
#pragma unroll
                for (int s8 = 0; s8 < 8; s8++) tbr[2 * ((int64_t)((t - 8) >> 6) * stride + ((t - 8) & 63) + s8)] = rec[s8];
            }
        }
        if (blk_active) {
            uint32_t glane = 32u;
            if ((t & (TBS_SEG - 1)) == 0 && active) {      // step t-1 tops a trace-back segment: the lane of its best score is where that segment's walker starts
                // the whole band's planes, in the low lane's view (the high lane's result is not used): P = its A | the partner's B turned round, Q = its B | the partner's A
                const Planes P = {(uint64_t)L.h.A.v0 | ((uint64_t)__brev(pair_swap(L.h.B.v0)) << 32), (uint64_t)L.h.A.v1 | ((uint64_t)__brev(pair_swap(L.h.B.v1)) << 32),
                                  (uint64_t)L.h.A.v2 | ((uint64_t)__brev(pair_swap(L.h.B.v2)) << 32)};
                const Planes Q = {(uint64_t)L.h.B.v0 | ((uint64_t)__brev(pair_swap(L.h.A.v0)) << 32), (uint64_t)L.h.B.v1 | ((uint64_t)__brev(pair_swap(L.h.A.v1)) << 32),
                                  (uint64_t)L.h.B.v2 | ((uint64_t)__brev(pair_swap(L.h.A.v2)) << 32)};
                int32_t run = 0, bestv = 0;
                glane = 0u;
                for (int k = 0; k < 63; k++) {            // score(lane k+1) - score(lane k) = 2 (Qv[k+1] - Pv[k])
                    run += value_at(Q, k + 1) - value_at(P, k);
                    if (run > bestv) { bestv = run; glane = (uint32_t)(k + 1); }
                }
            }
            if (lo) {
                mvr[(t - 1) >> 6] = make_ulonglong2(L.mvacc, (uint64_t)(uint32_t)i0_blk | ((uint64_t)glane << 32));
                if (!active) info[sl] = DpInfo{steps, bt, bl, bt >= 0 ? best : NEGV};
            }
        }
    }
}

// ---- trace-back, part 1: the walk.  One lane per read, 16 reads per wave.
//
// The walk from the best cell back to the anchor is sequential per read, so a lane owns a read; what the
// kernel has to do is keep that serial chain short and never make it wait on memory.
//   * masks are consumed in 64-step chunks (aligned to 64, like the move words).  A step's masks are 128 bits but
//     the path only ever looks at band lanes near its own, so a staged chunk keeps, per step, the 32 bits of D
//     and of G starting at band lane `sh` = clamp(k - 16, 0, 32): 8 B/step, 512 B/chunk, one chunk buffer per
//     read in LDS.  While the lanes walk chunk c, the 16 B/step records of chunk c-1 are already in flight to
//     registers (one coalesced 1 KB load per read); they are cut down and parked in LDS when the walk of
//     chunk c is over.  Small LDS footprint = every read of a 40 000-read launch is resident at once.  A lane whose path left the staged 32 lanes (or whose prefetch was for the wrong
//     chunk) takes a synchronous reload; that is rare.
//   * per step: one LDS read, ~25 VALU ops, no branches.  The moves come from a 64-bit shift register (top bit =
//     move of the step before the current one); the operation of the step (M / I / D) goes into a 2-bit stream,
//     16 ops per word, flushed to HBM when full.  Run-length encoding is k_tb_cigar's job, off this chain.
// HBM traffic: the 16 B/step masks are read once.
constexpr int TBW_STRIDE = 512 + 8;              // bytes per read: one 64-step chunk of {D bits, G bits}; +8 staggers LDS banks
constexpr int TBW_RPW = 16;                      // reads walked per wave
constexpr int TBW_WPG = 1;                       // waves per workgroup (four measured: no faster on uniform reads, 20 % slower on reads of real shape)
// LDS traffic of one wave is processed in program order: what the wave's lanes wrote is there for its later reads; only the compiler has to keep the order
#define TBW_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

struct WalkOut { int32_t ok, i, ts, i_end, j_end, ncol, n_ops, pad_; };   // (i, ts - i) = the cell before the alignment's first

// Segmented form (TBS_*, default): the walk of a read is cut at every TBS_SEG-th DP step.  The walker of the top segment starts at the
// read's best cell; the walker of every other segment starts, at the same time, at the segment's top step in band lane 32 -- a GUESS: the
// steering keeps the path near the band's centre, and trace-backs from neighbouring cells of one anti-diagonal run into each other within
// a few dozen steps (the best way into a cell next to the optimal path is the optimal path plus a gap).  Every walker goes TBS_OV steps
// past its segment's bottom and notes, for the first and the last TBS_OV steps of its walk, which lane it was in and how many ops it had
// emitted (`trail`); k_tb_stitch then finds, boundary by boundary, the first step at which the upper walker and the lower one sit in the
// same cell, takes the upper one's ops up to there and the lower one's from there on, and writes the read's one op stream.  The result
// is the serial walk's, op for op; a read with a boundary that does not merge inside TBS_OV steps is walked again serially (k_tb_walk<false>
// over the flagged reads).  A read's walk is no longer one chain of 2.25 x its length: all walkers are TBS_SEG + TBS_OV steps long.
struct SegOut { int32_t state, i, ts, n_ops, i_start, j_start, k, pad_; };   // state & 3: 0 no such segment, 1 ran to its lower bound, 2 reached the matrix edge; state & 4: a repair walk
struct SegReq { int32_t walker, ts, k, pad_; };                                // repair request: walk this segment again from the cell (ts, k) its upper neighbour stopped in

template <bool SEGMENTED>
__global__ void __launch_bounds__(64 * TBW_WPG) k_tb_walk(int64_t first, int64_t count, const Anchor *__restrict__ anc, const DpInfo *__restrict__ info,
                                                const int64_t *__restrict__ tb_off, const int64_t *__restrict__ tbo, const int64_t *__restrict__ mvo,
                                                const ulonglong2 *__restrict__ tb, const ulonglong2 *__restrict__ mvw, uint32_t *__restrict__ raw,
                                                WalkOut *__restrict__ wout, const int32_t *__restrict__ order, const int32_t *__restrict__ seg_slot,
                                                const int32_t *__restrict__ seg_idx, uint32_t *__restrict__ trail, SegOut *__restrict__ segout, int only_flagged, int guess_lane,
                                                const SegReq *__restrict__ req, const uint32_t *__restrict__ n_req, uint32_t *__restrict__ raw_final, const int32_t *__restrict__ tbs) {
    // TBW_WPG independent waves per workgroup (a workgroup's waves are spread over its CU's SIMDs; single-wave workgroups land two and three to a SIMD while others idle):
    // every wave has its own slice of the LDS buffer and never waits for another
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[TBW_WPG * TBW_RPW * TBW_STRIDE];
    uint8_t *lds = lds_all + (threadIdx.x >> 6) * (TBW_RPW * TBW_STRIDE);
    const int lane = threadIdx.x & 63;
    int64_t wq = ((int64_t)blockIdx.x * TBW_WPG + (threadIdx.x >> 6)) * TBW_RPW + lane;
    // repair launch (segmented form, req != nullptr): lane x walks request x -- the segment of walker req[x].walker again, from the exact cell
    // its upper neighbour stopped in; everything it leaves behind (ops, trail, SegOut) goes where that walker's went
    const bool repair = SEGMENTED && req != nullptr;
    if (repair) count = (int64_t)min(*n_req, (uint32_t)count);
    bool have = lane < TBW_RPW && wq < count;
    SegReq rq = {0, 0, 0, 0};
    if (repair && have) { rq = req[wq]; wq = rq.walker; }
    else if (SEGMENTED && have && order) wq = order[wq];     // launch order: the one-walker reads first (longest first), sixteen to a wave, then the segments
    // serial form: `order` = slots by decreasing read length -- the 16 reads of a wave are of similar length (a wave lasts as long as its
    // longest walk) and the longest start first.  Segmented form: `count` walkers, walker wq = segment seg_idx[wq] of slot seg_slot[wq]
    const int64_t wv = have ? (SEGMENTED ? (int64_t)seg_slot[wq] : (order ? (int64_t)order[wq] : wq)) : 0;
    int32_t seg = (SEGMENTED && have) ? seg_idx[wq] : 0;
    const bool single = SEGMENTED && seg < 0;       // a short read: this walker does all of it, as the serial form would (its ops go to the read's own stream)
    if (single) seg = 0;
    const int64_t r = first + wv;
    if (!SEGMENTED && only_flagged) have = have && wout[r].ok == 2;        // second pass: only the reads whose stitching failed
    Anchor a = {0, 0, 0, 0};
    DpInfo di = {0, -1, 0, NEGV};
    if (have) { a = anc[r]; di = info[r]; }
    bool active = have && a.aligned && di.best_t >= 0;
    const int32_t seg_top = single ? 0 : di.best_t >> TBS_SEG_SHIFT;
    if (SEGMENTED) active = active && seg <= seg_top;
    const int64_t soff = tb_off[r] - tb_off[first];                           // steps before this read in the chunk of reads
    // masks / move words of the winning candidate (second candidates sit behind the first ones in the same buffers)
    int64_t to_ = tbo[r], mo_ = mvo[r];
    asm volatile("" : "+v"(to_), "+v"(mo_));      // both offsets are in registers from here on: no pending load is attributed to the pointers below
    const ulonglong2 *tbr = tb + to_;                                         // per step {D mask, G mask}
    const ulonglong2 *mvr = mvw + mo_;                                        // per 64 steps {move bits, i0 before them}
    uint32_t *rawp = (SEGMENTED && !single) ? raw + wq * TBS_RAW_WORDS : (SEGMENTED ? raw_final : raw) + (soff >> 4);     // 16 ops per word
    const bool spec = SEGMENTED && !repair && seg < seg_top;                  // a walker that starts on the guess
    const int32_t ts0 = repair ? rq.ts : (spec ? (seg + 1) * TBS_SEG - 1 : di.best_t);
    int32_t ts = active ? ts0 : -1;
    int32_t k = repair ? rq.k : di.best_lane, i = -1;
    const int32_t stop_ts = (SEGMENTED && seg > 0) ? seg * TBS_SEG - TBS_OV : (int32_t)0x80000000;   // walk while ts >= stop_ts
    uint32_t *tr_head = SEGMENTED ? trail + wq * (2 * TBS_OV) : nullptr, *tr_tail = SEGMENTED ? tr_head + TBS_OV : nullptr;
    const int32_t tail_top = seg * TBS_SEG - 1;                               // the boundary below this segment
    uint64_t w_prev = 0, pref_word = 0;      // move words: (after the first accept) w_cur = chunk of ts, w_prev = the one below
    uint64_t w_cur = 0;
    if (active) {   // i0 at the start step = i0 before its 64-step chunk + DOWN moves up to and including it
        const ulonglong2 mw = mvr[ts >> 6];
        if (spec) k = guess_lane >= 0 ? guess_lane : (int32_t)(mw.y >> 32);      // k_sw left the lane of the best H of a segment's top step next to its move word
        i = (int32_t)(uint32_t)mw.y + __popcll(mw.x & ((2ull << (ts & 63)) - 1ull)) + k;
        w_prev = mw.x;
        pref_word = (ts >> 6) > 0 ? mvr[(ts >> 6) - 1].x : 0ull;
    }
    const int32_t i_end = i, j_end = ts - i;
    active = active && i >= 0 && ts - i >= 0;
    const bool walked = active;
    int32_t ncol = 0, n_ops = 0, nw = 0;
    uint32_t rawacc = 0, nb = 0;
    const int32_t plo = (int32_t)(uint32_t)(uint64_t)tbr, phi = (int32_t)((uint64_t)tbr >> 32);
    const int32_t rstride = (tbs && have) ? tbs[r] : 64;      // records from one 64-step chunk of the read's masks to the next
    uint8_t *mine = lds + lane * TBW_STRIDE;
    int32_t cur_chunk = -2, sh_cur = 0;
    int32_t pref_chunk = active ? ts >> 6 : -1, pref_sh = min(max(k - 16, 0), 32);
    uint4 pf[TBW_RPW];
#pragma unroll
    for (int l = 0; l < TBW_RPW; l++) pf[l] = make_uint4(0, 0, 0, 0);
    uint64_t rec_base[TBW_RPW];      // every read's mask records: wave-uniform, fetched from the owning lanes once
#pragma unroll
    for (int l = 0; l < TBW_RPW; l++) rec_base[l] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(phi, l) << 32) | (uint32_t)__builtin_amdgcn_readlane(plo, l);
    // records of chunk pref_chunk of every read -> registers (lane x takes step x of the chunk)
    // (the record pointers are rebuilt from lane reads, which leaves them in the generic address space; a FLAT load counts on lgkmcnt as
    //  well as on vmcnt, so the walk's first wait for an LDS read would wait for the whole prefetch: load through global pointers)
    typedef uint32_t tbw_u32x4 __attribute__((ext_vector_type(4)));
    typedef const tbw_u32x4 __attribute__((address_space(1))) *tbw_gptr;
#define TBW_ISSUE()                                                                                                      \
    _Pragma("unroll") for (int l = 0; l < TBW_RPW; l++) {                                                                \
        const int32_t cl = __builtin_amdgcn_readlane(pref_chunk, l);                                                     \
        if (cl >= 0) {                                                                                                   \
            const tbw_u32x4 q_ = ((tbw_gptr)rec_base[l])[(int64_t)cl * __builtin_amdgcn_readlane(rstride, l) + lane];            \
            pf[l] = make_uint4(q_.x, q_.y, q_.z, q_.w);                                                                  \
        }                                                                                                                \
    }
    TBW_ISSUE()
    for (;;) {
        if (!__any(active)) break;
        const int32_t need = active ? ts >> 6 : -1;
        // park the prefetched chunk: 32 lanes' worth of D and G per step (the walk is done with the old contents)
        {
#pragma unroll
            for (int l = 0; l < TBW_RPW; l++) {
                const int32_t cl = __builtin_amdgcn_readlane(pref_chunk, l);
                if (cl >= 0) {
                    const int32_t sh = __builtin_amdgcn_readlane(pref_sh, l);
                    const uint64_t D = ((uint64_t)pf[l].y << 32) | pf[l].x, G = ((uint64_t)pf[l].w << 32) | pf[l].z;
                    *(uint2 *)(lds + l * TBW_STRIDE + lane * 8) = make_uint2((uint32_t)(D >> sh), (uint32_t)(G >> sh));
                }
            }
        }
        const bool ok = need < 0 || (need == pref_chunk && (uint32_t)(k - pref_sh) < 32u);
        if (active && ok) { cur_chunk = pref_chunk; sh_cur = pref_sh; w_cur = w_prev; w_prev = pref_word; }
        const uint64_t redo = __ballot(active && !ok);
        if (redo) {   // rare: the path left the staged lanes, or stalled inside its chunk
            if (active && !ok) {
                cur_chunk = need; sh_cur = min(max(k - 16, 0), 32);
                w_cur = mvr[need].x; w_prev = need > 0 ? mvr[need - 1].x : 0ull;
            }
            for (int l = 0; l < TBW_RPW; l++) {
                if (!((redo >> l) & 1ull)) continue;
                const uint64_t pl = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(phi, l) << 32) | (uint32_t)__builtin_amdgcn_readlane(plo, l);
                const int32_t cl = __builtin_amdgcn_readlane(cur_chunk, l), sh = __builtin_amdgcn_readlane(sh_cur, l);
                const uint4 v = ((const uint4 *)pl)[(int64_t)cl * __builtin_amdgcn_readlane(rstride, l) + lane];
                const uint64_t D = ((uint64_t)v.y << 32) | v.x, G = ((uint64_t)v.w << 32) | v.z;
                *(uint2 *)(lds + l * TBW_STRIDE + lane * 8) = make_uint2((uint32_t)(D >> sh), (uint32_t)(G >> sh));
            }
        }
        // next prefetch: the chunk below, centred on where the path is now
        pref_chunk = (active && cur_chunk > 0) ? cur_chunk - 1 : -1;
        pref_sh = min(max(k - 16, 0), 32);
        if (pref_chunk > 0) pref_word = mvr[pref_chunk - 1].x; else pref_word = 0ull;
        TBW_ISSUE()
        TBW_WAVE_SYNC();
        // ---- walk inside the chunk
        uint32_t d1 = 0;
        uint64_t P = 0;                    // moves of steps ts-1, ts-2, ... from the top bit down (steps before 0 read as RIGHT)
        if (active) {
            const int32_t s_ = ts & 63;
            d1 = (uint32_t)(w_cur >> s_) & 1u;
            P = s_ ? (w_cur << (64 - s_)) | (w_prev >> s_) : w_prev;
        }
        const uint8_t *win = mine;
        const bool notes = SEGMENTED && __any(active && !single);      // (wave-uniform: a wave of whole-read walkers skips the note-taking with a scalar branch)
        while (active && (ts >> 6) == cur_chunk && (uint32_t)(k - sh_cur) < 32u) {
            const uint2 m = *(const uint2 *)(win + (ts & 63) * 8);
            const uint32_t kk = (uint32_t)(k - sh_cur);
            if (notes && !single) {      // where this walker is, for the stitching: its first TBS_OV steps and the TBS_OV steps below its segment
                const uint32_t note = (repair ? 0x80000000u : 0u) | ((uint32_t)n_ops << 8) | (uint32_t)k;      // (a repair walk's notes are told from the stale ones around them by bit 31)
                if (!repair && ts0 - ts < TBS_OV) tr_head[ts0 - ts] = note;
                if ((uint32_t)(tail_top - ts) < (uint32_t)TBS_OV) tr_tail[tail_top - ts] = note;
            }
            const uint32_t db = (m.x >> kk) & 1u, gb = (m.y >> kk) & 1u;
            const uint32_t hi = (uint32_t)(P >> 32);
            const uint32_t d2 = hi >> 31, d3 = (hi >> 30) & 1u;
            // not diagonal: G set after a DOWN move, or clear after a RIGHT move -> the predecessor is the cell above
            const uint32_t ndb = db ^ 1u;
            const uint32_t up = ndb & ((gb ^ d1) ^ 1u);
            const uint32_t op = 2u * ndb - up;                                // M = 0, I = 1, D = 2
            k += (int32_t)(d1 + (db & d2)) - (int32_t)(db + up);
            i -= (int32_t)(db + up);
            ts -= (int32_t)(1u + db);
            d1 = db ? d3 : d2;
            P <<= (1u + db);
            ncol += (int32_t)db;
            n_ops++;
            rawacc |= op << nb;
            nb += 2u;
            if (nb == 32u) { rawp[nw++] = rawacc; rawacc = 0u; nb = 0u; }
            active = (i | (ts - i)) >= 0 && ts >= stop_ts;
        }
        TBW_WAVE_SYNC();
    }
#undef TBW_ISSUE
    if (!have) return;
    if (nb) rawp[nw] = rawacc;
    if (SEGMENTED && !single) {
        SegOut so;
        so.state = !walked ? 0 : (((i | (ts - i)) < 0 ? 2 : 1) | (repair ? 4 : 0));
        so.i = i; so.ts = ts; so.n_ops = n_ops; so.i_start = i_end; so.j_start = j_end; so.k = k; so.pad_ = 0;
        segout[wq] = so;
        return;
    }
    WalkOut o;
    o.ok = walked ? 1 : 0; o.i = i; o.ts = ts; o.i_end = i_end; o.j_end = j_end; o.ncol = ncol; o.n_ops = n_ops; o.pad_ = 0;
    wout[r] = o;
}

__global__ void k_tb_req_reset(uint32_t *counters) { if (threadIdx.x == 0) counters[1] = 0u; }

// ---- trace-back, part 1b: one wave per read joins its segments' walks (k_tb_walk<true>) into the read's op stream.
// First pass (bit 0 of `pass`): every read.  While repair launches are still to come (bit 1), a boundary whose two walkers share no cell inside the
// overlap asks for a repair walk of the lower segment from the cell the upper walker stopped in (if the upper walker is known to be on the path there)
// and the read waits (ok = 3); later passes take the waiting reads only.  In the last pass what still does not join is left to the serial walk (ok = 2).
__global__ void __launch_bounds__(64) k_tb_stitch(int64_t first, int64_t count, const Anchor *__restrict__ anc, const DpInfo *__restrict__ info,
                                                  const int64_t *__restrict__ tb_off, const int32_t *__restrict__ seg_off, const uint32_t *__restrict__ trail,
                                                  const SegOut *__restrict__ segout, const uint32_t *__restrict__ raw_seg, uint32_t *__restrict__ raw,
                                                  WalkOut *__restrict__ wout, uint32_t *__restrict__ counters, SegReq *__restrict__ req, uint32_t req_cap, int pass,
                                                  const uint8_t *__restrict__ seg_single, int ov_limit) {
    __shared__ int32_t p_w[TBS_MAX_PIECES], p_a[TBS_MAX_PIECES], p_out[TBS_MAX_PIECES + 1];     // piece: walker, first op taken from it, first op of the output it fills
    const int lane = lane_id();
    const int64_t wv = blockIdx.x;
    if (wv >= count) return;
    const int64_t r = first + wv;
    if (seg_single[wv]) return;                       // one walker did the whole read and left the stream and the WalkOut itself
    if (!(pass & 1) && wout[r].ok != 3) return;        // pass bit 0: the first pass (every read); bit 1: boundaries that do not join may ask for a repair walk
    const Anchor a = anc[r];
    const DpInfo di = info[r];
    WalkOut o;
    memset(&o, 0, sizeof o);
    if (!(a.aligned && di.best_t >= 0)) { if (lane == 0) wout[r] = o; return; }
    const int32_t S = di.best_t >> TBS_SEG_SHIFT;
    const int32_t w0 = seg_off[wv];
    const SegOut top = segout[w0 + S];
    if (top.state == 0) { if (lane == 0) wout[r] = o; return; }                 // the best cell itself is outside the matrix: no walk (as the serial form)
    o.i_end = top.i_start; o.j_end = top.j_start;
    bool hard = S + 1 > TBS_MAX_PIECES, fail = false;
    bool anchored = true;                                                        // the upper walker of the boundary at hand is on the path where it stopped
    int32_t start = 0, outpos = 0, np = 0, fin_i = 0, fin_ts = 0;
    for (int32_t sg = S; sg >= 0 && !hard; sg--) {                               // wave-uniform
        const SegOut so = segout[w0 + sg];
        if ((so.state & 3) == 0) { hard = true; break; }
        if ((so.state & 3) == 2 || sg == 0) {                                    // the path ends inside this segment: its walker's remaining ops are the last piece
            if (lane == 0) { p_w[np] = w0 + sg; p_a[np] = start; p_out[np] = outpos; }
            outpos += so.n_ops - start; np++;
            fin_i = so.i; fin_ts = so.ts;
            break;
        }
        const SegOut lo = segout[w0 + sg - 1];
        if (lo.state & 4) {                                                      // the lower segment was walked again from this walker's last cell: they join there
            if (lane == 0) { p_w[np] = w0 + sg; p_a[np] = start; p_out[np] = outpos; }
            outpos += so.n_ops - start; np++;
            start = 0; anchored = true;
            continue;
        }
        const uint32_t *ta = trail + (int64_t)(w0 + sg) * (2 * TBS_OV) + TBS_OV, *hb = trail + (int64_t)(w0 + sg - 1) * (2 * TBS_OV);
        const uint32_t want31 = (so.state & 4) ? 1u : 0u;
        int32_t ia = -1, ib = -1;
        for (int d0 = 0; d0 < TBS_OV; d0 += 64) {
            const uint32_t ua = ta[d0 + lane], ub = hb[d0 + lane];
            const uint64_t m = __ballot(d0 + lane < ov_limit && ua != 0xffffffffu && ub != 0xffffffffu && (ua >> 31) == want31 && (ua & 0xffu) == (ub & 0xffu));
            if (m) {
                const int l = __builtin_ctzll(m);                                 // the first common cell below the boundary
                ia = __builtin_amdgcn_readlane((int32_t)((ua >> 8) & 0x7fffffu), l); ib = __builtin_amdgcn_readlane((int32_t)((ub >> 8) & 0x7fffffu), l);
                break;
            }
        }
        if (ia >= start && !fail) {
            if (lane == 0) { p_w[np] = w0 + sg; p_a[np] = start; p_out[np] = outpos; }
            outpos += ia - start; np++;
            start = ib;
        } else if (ia >= 0 && fail) {                                             // below an open boundary: the pieces are not built any more, but this one joins
            anchored = true;
        } else {                                                                  // no common cell inside TBS_OV steps (or one above where this walker joined the path)
            if ((pass & 2) && anchored && lane == 0) {
                const uint32_t q = atomicAdd(&counters[1], 1u);
                atomicAdd(&counters[2], 1u);
                if (q < req_cap) { SegReq rq; rq.walker = w0 + sg - 1; rq.ts = so.ts; rq.k = so.k; rq.pad_ = 0; req[q] = rq; }
            }
            fail = true; anchored = false;
        }
    }
    if (hard || (fail && !(pass & 2))) { o.ok = 2; if (lane == 0) { wout[r] = o; atomicAdd(&counters[0], 1u); } return; }    // k_tb_walk<false> walks this read serially
    if (fail) { o.ok = 3; if (lane == 0) wout[r] = o; return; }                  // waits for the repair walks
    if (lane == 0) p_out[np] = outpos;
    __syncthreads();
    // the pieces, one after the other, into the read's stream: a lane builds an output word from the (at most two) source words under it
    uint32_t *rg = raw + ((tb_off[r] - tb_off[first]) >> 4);
    const int32_t L = outpos, nW = (L + 15) >> 4;
    for (int32_t wi = lane; wi < nW; wi += 64) {
        const int32_t o0 = 16 * wi, o1 = min(o0 + 16, L);
        int32_t pc = 0;
        while (pc + 1 < np && p_out[pc + 1] <= o0) pc++;
        uint32_t word = 0;
        int32_t oo = o0;
        while (oo < o1) {
            const int32_t pe = min(o1, p_out[pc + 1]);                            // ops [oo, pe) come from piece pc
            const int32_t sa = p_a[pc] + (oo - p_out[pc]);                        // first source op
            const uint32_t *src = raw_seg + (int64_t)p_w[pc] * TBS_RAW_WORDS;
            const int32_t sw = sa >> 4, sb = (sa & 15) * 2;
            const uint64_t two = (uint64_t)src[sw] | ((uint64_t)src[sw + 1] << 32);   // (a walker's buffer has a spare word)
            uint32_t bits = (uint32_t)(two >> sb);
            const int32_t cnt = pe - oo;
            if (cnt < 16) bits &= (1u << (2 * cnt)) - 1u;
            word |= bits << (2 * (oo - o0));
            oo = pe; pc++;
        }
        rg[wi] = word;
    }
    o.ok = 1; o.i = fin_i; o.ts = fin_ts; o.ncol = 0; o.n_ops = L;               // ncol (and i, ts after trimming) are k_tb_cigar's pass 0's
    if (lane == 0) wout[r] = o;
}

// ---- trace-back, part 2: one wave per read turns the walk's op stream (alignment end first, 16 ops per word)
// into the forward, run-length encoded CIGAR (M/I/D; '=' / 'X' need the bases and are split on the host where SAM
// text / alnsets are produced -- the phasing stages treat M, = and X alike, phasing.py:81), trims gap runs at both
// ends, adds the soft clips and fills the read's summary.
// A lane owns a word.  Stream position p is forward position L-1-p, so forward runs start where op(p) != op(p+1):
// the flags of a word come from one xor with the stream shifted by one op, and a run is written by the start BELOW
// it (which knows where it ends); suffix scans over the lanes give each word the number of starts and the lowest
// start above it.
__global__ void __launch_bounds__(64) k_tb_cigar(int64_t first, int64_t count, const int32_t *__restrict__ read_len, const Anchor *__restrict__ anc,
                                                 const DpInfo *__restrict__ info, const int64_t *__restrict__ tb_off, uint32_t *__restrict__ raw,
                                                 const WalkOut *__restrict__ wout, const int64_t *__restrict__ cig_off, uint32_t *__restrict__ cig,
                                                 int64_t *__restrict__ cig_start, fzp_aln_summary *__restrict__ summ, int match, int mismatch, int gap,
                                                 int min_pct_identity, const uint32_t *__restrict__ read_ori, const int64_t *__restrict__ read_woff,
                                                 const int32_t *__restrict__ read_ctg, const uint32_t *__restrict__ ctg_pk, const int64_t *__restrict__ ctg_woff) {
    const int lane = lane_id();
    const int64_t wv = blockIdx.x;
    if (wv >= count) return;
    const int64_t r = first + wv;
    const Anchor a = anc[r];
    const DpInfo di = info[r];
    WalkOut w = wout[r];
    const int32_t n = read_len[r];
    fzp_aln_summary out;
    memset(&out, 0, sizeof out);
    out.cells = (int64_t)di.steps * 64;
    if (lane == 0) cig_start[r] = cig_off[r];
    if (!w.ok) { if (lane == 0) summ[r] = out; return; }
    uint32_t *rg = raw + ((tb_off[r] - tb_off[first]) >> 4);    // (pass 0 drops the ops before the alignment's end from it)
    uint32_t *reg = cig + cig_off[r];                     // capacity n + 18 words: [0] leading clip, runs from [1]
    int32_t L = w.n_ops;
    int32_t nW = (L + 15) >> 4;
    constexpr uint32_t EVEN = 0x55555555u;
    auto valid_mask = [&](int32_t wi) -> uint32_t {       // one bit (the even one) per op of word wi that belongs to the stream
        const int32_t nv = min(16, L - 16 * wi);
        return nv >= 16 ? EVEN : (nv <= 0 ? 0u : (((1u << (2 * nv)) - 1u) & EVEN));
    };
    // pass 0 (fzalign v1.5, "best sub-path"): P(k) = score of ops 0..k-1 of the stream (op 0 leaves the forward terminal, the last op reaches the backward
    // one); the alignment is ops e..s with the largest P(s+1) - P(e) (ties: the smallest s, then the largest e) -- both ends are match columns -- and what the
    // walks found outside it (a tail dragged to the matrix border through noise, a head likewise) becomes soft clip.  A lane scores a word: its 16 ops consume at
    // most 16 read and 16 contig bases going down from the word's first cell, which exclusive scans of the words' consumption counts give.  Two sweeps over the
    // word's ops: the first leaves its score, the lowest prefix inside it and which ops are matching columns; a min-scan over the lanes (and the chunks before)
    // then gives every word the lowest prefix before it, and the second sweep finds its best (e, s).
    int32_t S_star = 0;
    {
        const uint32_t *qpk = read_ori + read_woff[r];
        const uint32_t *tpk = ctg_pk + ctg_woff[read_ctg[r]];
        auto scan_incl = [&](int32_t v) -> int32_t {
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int32_t o = __shfl_up(v, d, 64); if (lane >= d) v += o; }
            return v;
        };
        auto window = [&](const uint32_t *pk, int64_t idx_hi, int64_t &lo_idx) -> uint64_t {     // the 32 bases ending in the u32 word of idx_hi
            const int64_t w1 = idx_hi >> 4;
            lo_idx = (w1 - 1) * 16;
            return ((uint64_t)pk[w1] << 32) | (w1 > 0 ? pk[w1 - 1] : 0u);
        };
        int32_t base_S = 0, base_i = 0, base_j = 0, bestS = 0, bestP = -1, bestE = 0;
        int32_t base_min = 0, base_min_pos = 0;                  // lowest prefix P(e) over the chunks so far (P(0) = 0 at e = 0), the largest such e
        for (int32_t wb = 0; wb < nW; wb += 64) {
            const int32_t wi = wb + lane;
            const uint32_t x = wi < nW ? rg[wi] : 0u, vm = wi < nW ? valid_mask(wi) : 0u;
            const uint32_t fM = ~(x | (x >> 1)) & vm, fI = (x & ~(x >> 1)) & vm, fD = (~x & (x >> 1)) & vm;
            const int32_t ci = __popc(fM | fI), cj = __popc(fM | fD);
            const int32_t si = scan_incl(ci), sj = scan_incl(cj);
            int32_t i = w.i_end - (base_i + si - ci), j = w.j_end - (base_j + sj - cj);       // the cell the word's first op leaves
            int64_t qlo = 0, tlo = 0;
            const uint64_t qw = (ci && (int64_t)a.i_a + i >= 0) ? window(qpk, (int64_t)a.i_a + i, qlo) : 0ull;     // (i, j are relative to the anchor: negative in the backward part)
            const uint64_t tw = (cj && (int64_t)a.c_a + j >= 0) ? window(tpk, (int64_t)a.c_a + j, tlo) : 0ull;
            int32_t sl = 0, lmin = 0x3fffffff, lpos = 0;
            uint32_t eq = 0;                                       // even bit of op o: a matching column
#pragma unroll
            for (int o = 0; o < 16; o++) {
                if ((vm >> (2 * o)) & 1u) {
                    if (sl <= lmin) { lmin = sl; lpos = o; }       // prefix BEFORE op o ('<=': the largest e)
                    const uint32_t op = (x >> (2 * o)) & 3u;
                    if (op == 0u) {
                        const uint32_t qb_ = (uint32_t)(qw >> (2 * (uint32_t)((int64_t)a.i_a + i - qlo))) & 3u, tb_ = (uint32_t)(tw >> (2 * (uint32_t)((int64_t)a.c_a + j - tlo))) & 3u;
                        if (qb_ == tb_) { sl += match; eq |= 1u << (2 * o); } else sl -= mismatch;
                        i--; j--;
                    } else { sl -= gap; if (op == 1u) i--; else j--; }
                }
            }
            const int32_t ss = scan_incl(sl);
            const int32_t start = base_S + ss - sl;                // P(16 * wi)
            // lowest prefix over the words up to and including this one: (value, position), later positions win ties
            int32_t mv_ = vm ? start + lmin : 0x3fffffff, mp_ = 16 * wi + lpos;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int32_t ov_ = __shfl_up(mv_, d, 64), op_ = __shfl_up(mp_, d, 64);
                if (lane >= d && ov_ < mv_) { mv_ = ov_; mp_ = op_; }
            }
            int32_t gm = __shfl_up(mv_, 1, 64), gp = __shfl_up(mp_, 1, 64);      // ... before this word
            if (lane == 0 || base_min < gm) { gm = base_min; gp = base_min_pos; }      // (the chunks before hold earlier positions: they win only when strictly lower)
            {
                int32_t Pk = start;
#pragma unroll
                for (int o = 0; o < 16; o++) {
                    if ((vm >> (2 * o)) & 1u) {
                        if (Pk <= gm) { gm = Pk; gp = 16 * wi + o; }
                        const uint32_t op = (x >> (2 * o)) & 3u;
                        Pk += op == 0u ? (((eq >> (2 * o)) & 1u) ? match : -mismatch) : -gap;
                        if (Pk - gm > bestS) { bestS = Pk - gm; bestP = 16 * wi + o; bestE = gp; }      // words ascend within a lane: '>' keeps the smallest s
                    }
                }
            }
            {
                const int32_t cm = __builtin_amdgcn_readlane(mv_, 63), cp = __builtin_amdgcn_readlane(mp_, 63);
                if (cm <= base_min) { base_min = cm; base_min_pos = cp; }
            }
            base_S += __builtin_amdgcn_readlane(ss, 63); base_i += __builtin_amdgcn_readlane(si, 63); base_j += __builtin_amdgcn_readlane(sj, 63);
        }
        // wave argmax: largest score, then smallest s
        int32_t vS = bestS, vP = bestP < 0 ? 0x7fffffff : bestP, vE = bestE;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const int32_t oS = __shfl_xor(vS, d, 64), oP = __shfl_xor(vP, d, 64), oE = __shfl_xor(vE, d, 64);
            if (oS > vS || (oS == vS && oP < vP)) { vS = oS; vP = oP; vE = oE; }
        }
        if (vP == 0x7fffffff || vS <= 0) { if (lane == 0) summ[r] = out; return; }       // not one matching column on the path
        S_star = vS;
        const int32_t e0 = __builtin_amdgcn_readfirstlane(vE), s0 = __builtin_amdgcn_readfirstlane(vP);
        if (e0 > 0) {
            // the ops before e leave the stream: what they consume moves the alignment's end, the rest shifts down (in place: a chunk is read before it is
            // written and only reads at or above what it writes)
            int32_t ce_i = 0, ce_j = 0;
            const int32_t we = e0 >> 4;
            for (int32_t wb = 0; wb <= we; wb += 64) {
                const int32_t wi = wb + lane;
                if (wi <= we) {
                    const uint32_t x = rg[wi];
                    uint32_t vm = valid_mask(wi);
                    if (wi == we) vm &= (1u << (2 * (e0 & 15))) - 1u;
                    const uint32_t fM = ~(x | (x >> 1)) & vm, fI = (x & ~(x >> 1)) & vm, fD = (~x & (x >> 1)) & vm;
                    ce_i += __popc(fM | fI); ce_j += __popc(fM | fD);
                }
            }
            ce_i = wave_sum_i32_dpp(ce_i); ce_j = wave_sum_i32_dpp(ce_j);
            w.i_end -= __builtin_amdgcn_readfirstlane(ce_i); w.j_end -= __builtin_amdgcn_readfirstlane(ce_j);
            const int32_t Ln = s0 - e0 + 1, nWn = (Ln + 15) >> 4;
            const uint32_t sh = 2u * (uint32_t)(e0 & 15);
            for (int32_t wb = 0; wb < nWn; wb += 64) {
                const int32_t wi = wb + lane;
                uint32_t v = 0;
                if (wi < nWn) {
                    const uint32_t lo = rg[we + wi], hi = we + wi + 1 < nW ? rg[we + wi + 1] : 0u;
                    v = sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
                }
                __builtin_amdgcn_wave_barrier();
                if (wi < nWn) rg[wi] = v;
            }
            L = Ln;
        } else L = s0 + 1;
        nW = (L + 15) >> 4;
        // what the kept ops consume, and their aligned columns
        int32_t ci2 = 0, cj2 = 0, nm2 = 0;
        for (int32_t wb = 0; wb < nW; wb += 64) {
            const int32_t wi = wb + lane;
            if (wi < nW) {
                const uint32_t x = rg[wi], vm = valid_mask(wi);
                const uint32_t fM = ~(x | (x >> 1)) & vm, fI = (x & ~(x >> 1)) & vm, fD = (~x & (x >> 1)) & vm;
                ci2 += __popc(fM | fI); cj2 += __popc(fM | fD); nm2 += __popc(fM);
            }
        }
        ci2 = wave_sum_i32_dpp(ci2); cj2 = wave_sum_i32_dpp(cj2); nm2 = wave_sum_i32_dpp(nm2);
        ci2 = __builtin_amdgcn_readfirstlane(ci2); cj2 = __builtin_amdgcn_readfirstlane(cj2); nm2 = __builtin_amdgcn_readfirstlane(nm2);
        w.i = w.i_end - ci2;                      // the cell before the alignment's first op, as the walk would have left it
        w.ts = w.i + (w.j_end - cj2);
        w.ncol = nm2;
        w.n_ops = L;
    }
    // pass A: highest / lowest stream position holding an aligned column
    int32_t pM_hi = -1, pM_lo = 0x7fffffff;
    for (int32_t wb = 0; wb < nW; wb += 64) {
        const int32_t wi = wb + lane;
        if (wi < nW) {
            const uint32_t x = rg[wi];
            const uint32_t mf = ~(x | (x >> 1)) & valid_mask(wi);
            if (mf) { pM_hi = max(pM_hi, 16 * wi + ((31 - __builtin_clz(mf)) >> 1)); pM_lo = min(pM_lo, 16 * wi + (__builtin_ctz(mf) >> 1)); }
        }
    }
    pM_hi = wave_max_i32(pM_hi);
    pM_lo = wave_min_i32(pM_lo);
    // pass B: runs, from the top of the stream (= the alignment's start) down
    const int32_t max_runs = n + 16;
    int32_t n_starts = 0;                 // starts seen in higher chunks
    int32_t low_start = 0x7fffffff;       // lowest of them
    int32_t leadI = 0, leadD = 0, trailI = 0, trailD = 0, lead_runs = 0, trail_runs = 0;
    for (int32_t wtop = nW; wtop > 0; wtop -= 64) {
        const int32_t wi = wtop - 64 + lane;              // lane 63 = highest word of the chunk
        uint32_t sflag = 0, x = 0, above = 0;
        if (wi >= 0) {
            x = rg[wi];
            const uint32_t nx = wi + 1 < nW ? rg[wi + 1] : 0u;
            above = (x >> 2) | (nx << 30);                // op(p+1) lined up with op(p)
            const uint32_t d = x ^ above;
            const uint32_t vm = valid_mask(wi);
            sflag = (d | (d >> 1)) & vm;
            if (wi == ((L - 1) >> 4)) sflag |= 1u << (2 * ((L - 1) & 15));     // the first forward op always starts a run
            // gap ops and run starts outside [pM_lo, pM_hi] are what the trimming removes
            const int32_t p0 = 16 * wi;
            const uint32_t fI = (x & ~(x >> 1)) & vm, fD = (~x & (x >> 1)) & vm;
            const int32_t nh = pM_hi - p0 + 1, nl = pM_lo - p0;      // ops of this word at or below pM_hi / below pM_lo
            const uint32_t m_lead = nh >= 16 ? 0u : (nh <= 0 ? EVEN : (EVEN & ~((1u << (2 * nh)) - 1u)));
            const uint32_t m_trail = nl >= 16 ? EVEN : (nl <= 0 ? 0u : (EVEN & ((1u << (2 * nl)) - 1u)));
            leadI += __popc(fI & m_lead); leadD += __popc(fD & m_lead); lead_runs += __popc(sflag & m_lead);
            trailI += __popc(fI & m_trail); trailD += __popc(fD & m_trail); trail_runs += __popc(sflag & m_trail);
        }
        const int32_t cnt = __popc(sflag);
        const int32_t mylow = sflag ? 16 * wi + (__builtin_ctz(sflag) >> 1) : 0x7fffffff;
        // suffix scans over the lanes above this one
        int32_t cs = cnt, lw = mylow;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int32_t oc = __shfl_down(cs, d, 64), ol = __shfl_down(lw, d, 64);
            if (lane + d < 64) { cs += oc; lw = min(lw, ol); }
        }
        const int32_t tot = __builtin_amdgcn_readlane(cs, 0), chunk_low = __builtin_amdgcn_readlane(lw, 0);
        int32_t k = n_starts + cs - cnt;                                    // starts above this word
        const int32_t lw_up = __shfl_down(lw, 1, 64);                      // (executed by all lanes)
        int32_t prevp = min(low_start, lane < 63 ? lw_up : 0x7fffffff);
        {
            uint32_t bits = sflag;
            while (bits) {
                const int b = 31 - __builtin_clz(bits);                      // even bit of the highest remaining start
                const int32_t pp = 16 * wi + (b >> 1);
                if (k > 0 && k - 1 < max_runs) reg[k] = ((uint32_t)(prevp - pp) << 4) | ((above >> b) & 3u);
                prevp = pp;
                k++;
                bits &= ~(1u << b);
            }
        }
        n_starts += tot;
        low_start = min(low_start, chunk_low);
    }
    const int32_t runs = n_starts;
    if (L > 0 && lane == 0 && runs >= 1 && runs - 1 < max_runs) reg[runs] = ((uint32_t)(low_start + 1) << 4) | (rg[0] & 3u);
    leadI = wave_sum_i32_dpp(leadI); leadD = wave_sum_i32_dpp(leadD);
    trailI = wave_sum_i32_dpp(trailI); trailD = wave_sum_i32_dpp(trailD);
    lead_runs = wave_sum_i32_dpp(lead_runs); trail_runs = wave_sum_i32_dpp(trail_runs);
    if (lane != 0) return;
    if (pM_hi >= 0 && w.ncol > 0 && runs <= max_runs) {
        int32_t fa = 1 + lead_runs, fb = 1 + runs - trail_runs;       // forward ops are reg[fa .. fb)
        const int32_t q_lead = w.i + 1 + leadI, r_lead = (w.ts - w.i) + 1 + leadD;
        const int32_t pos = (int32_t)(a.c_a + r_lead), ref_end = (int32_t)(a.c_a + w.j_end + 1 - trailD);
        const int32_t q_start = (int32_t)(a.i_a + q_lead), q_end = (int32_t)(a.i_a + w.i_end + 1 - trailI);
        // matches from the score of the kept ops (pass 0): they consume i_end - i read and j_end - j contig bases, ncol of each on diagonals,
        // the rest in gaps -- exact, the twin checks it against a direct count
        const int64_t num = (int64_t)S_star + (int64_t)mismatch * w.ncol + (int64_t)gap * ((int64_t)(w.i_end - w.i) + (w.j_end - (w.ts - w.i)) - 2 * (int64_t)w.ncol);
        const int32_t n_match = (int32_t)(num / (match + mismatch));
        const int64_t aln_len = (int64_t)(q_end - q_start) + (ref_end - pos) - w.ncol;      // columns + inserted + deleted bases
        if (min_pct_identity <= 0 || 100 * (int64_t)n_match >= (int64_t)min_pct_identity * aln_len) {   // blasr --minPctIdentity (unzip.py:87)
            out.aligned = 1;
            out.strand = a.strand;
            out.pos = pos;
            out.ref_end = ref_end;
            out.q_start = q_start;
            out.q_end = q_end;
            out.score = S_star;
            out.n_columns = w.ncol;
            out.n_match = n_match;
            int32_t nc = fb - fa;
            if (out.q_start > 0) { reg[--fa] = ((uint32_t)out.q_start << 4) | FZP_OP_S; nc++; }
            if (n - out.q_end > 0) { reg[fb++] = ((uint32_t)(n - out.q_end) << 4) | FZP_OP_S; nc++; }
            out.n_cigar = nc;
            cig_start[r] = cig_off[r] + fa;
        }
    }
    summ[r] = out;
}

// ---- backward extension (v1.4): inputs.  Per read of the chunk: the winner's anchor (i_h, c_h) -> the reversed oriented-read prefix [0, i_h) and the
// reversed contig window of min(c_h, i_h + i_h / 4 + 64) bases before c_h, both 2-bit packed with zero padding, at offsets the host planned from the
// candidates' anchors; k_sw then runs on them as on any read / contig pair (anchor (0, 0), the "contig" of slot r is its own window).
__global__ void __launch_bounds__(256) k_back_prep(int64_t first, const Anchor *__restrict__ anc, const int32_t *__restrict__ read_ctg, const uint32_t *__restrict__ read_ori,
                                                   const int64_t *__restrict__ read_woff, const uint32_t *__restrict__ ctg_pk, const int64_t *__restrict__ ctg_woff,
                                                   const int64_t *__restrict__ bq_off, const int64_t *__restrict__ bt_off, uint32_t *__restrict__ bq, uint32_t *__restrict__ bt,
                                                   Anchor *__restrict__ anc_b, int32_t *__restrict__ b_len, int64_t *__restrict__ b_tlen) {
    const int64_t r = first + blockIdx.x;
    const Anchor a = anc[r];
    const int32_t nqb = (a.aligned && a.i_a > 0 && a.c_a > 0) ? a.i_a : 0;
    const int32_t ntb = nqb ? min(a.c_a, nqb + nqb / 4 + 64) : 0;
    if (threadIdx.x == 0) {
        Anchor ab; ab.aligned = nqb > 0 ? 1 : 0; ab.strand = a.strand; ab.i_a = 0; ab.c_a = 0;
        anc_b[r] = ab; b_len[r] = nqb; b_tlen[r] = ntb;
    }
    const uint32_t *q = read_ori + read_woff[r];
    const uint32_t *t = ctg_pk + ctg_woff[read_ctg[r]];
    uint32_t *dq = bq + bq_off[r], *dt = bt + bt_off[r];
    const int64_t wq = bq_off[r + 1] - bq_off[r], wt = bt_off[r + 1] - bt_off[r];      // capacities: the used words first, zeros behind them
    for (int64_t w = threadIdx.x; w < wq; w += 256) {
        uint32_t v = 0;
        for (int m = 0; m < 16; m++) { const int64_t x = w * 16 + m; if (x < nqb) v |= base_at(q, (int64_t)a.i_a - 1 - x) << (2 * m); }
        dq[w] = v;
    }
    for (int64_t w = threadIdx.x; w < wt; w += 256) {
        uint32_t v = 0;
        for (int m = 0; m < 16; m++) { const int64_t x = w * 16 + m; if (x < ntb) v |= base_at(t, (int64_t)a.c_a - 1 - x) << (2 * m); }
        dt[w] = v;
    }
}
// ---- backward extension: its walk joins the forward one.  The read's op stream so far runs from the alignment's END to the forward walk's exit next to the
// anchor; behind it go the gap moves that exit implies (a walk that leaves through row / column -1 skipped bases there), the ones the backward walk's exit
// implies, and the backward walk's ops turned round (it came from the far end towards the anchor).  k_tb_cigar then sees one path.
__global__ void __launch_bounds__(64) k_back_merge(int64_t first, int64_t count, const int64_t *__restrict__ tb_off, const int64_t *__restrict__ tb_off_b,
                                                   const Anchor *__restrict__ anc_b, const DpInfo *__restrict__ info_b, const WalkOut *__restrict__ wout_b,
                                                   const uint32_t *__restrict__ raw_b, uint32_t *__restrict__ raw, WalkOut *__restrict__ wout, DpInfo *__restrict__ info) {
    const int lane = lane_id();
    const int64_t wv = blockIdx.x;
    if (wv >= count) return;
    const int64_t r = first + wv;
    WalkOut fw = wout[r];
    if (fw.ok != 1 || !anc_b[r].aligned) return;
    const DpInfo ib = info_b[r];
    const WalkOut bw = wout_b[r];
    DpInfo di = info[r];
    di.steps += ib.steps;                                                // the cells of the backward DP count, whatever came of it
    if (bw.ok != 1) { if (lane == 0) info[r] = di; return; }
    const int32_t is = fw.i, js = fw.ts - fw.i, bis = bw.i, bjs = bw.ts - bw.i;
    const int32_t nD = (is < 0 && js >= 0) ? js + 1 : 0, nI = (js < 0 && is >= 0) ? is + 1 : 0;
    const int32_t bD = (bis < 0 && bjs >= 0) ? bjs + 1 : 0, bI = (bjs < 0 && bis >= 0) ? bis + 1 : 0;
    const int32_t nb = bw.n_ops, m = nD + nI + bD + bI + nb, at = fw.n_ops;
    const int64_t cap_ops = tb_off[r + 1] - tb_off[r];
    if ((int64_t)at + m > cap_ops) { if (lane == 0) info[r] = di; return; }      // (cannot happen: a path has fewer ops than the DP had steps)
    uint32_t *rg = raw + ((tb_off[r] - tb_off[first]) >> 4);
    const uint32_t *rb = raw_b + ((tb_off_b[r] - tb_off_b[first]) >> 4);
    const int32_t w0 = at >> 4, w1 = (at + m - 1) >> 4;
    for (int32_t wi = w0 + lane; wi <= w1; wi += 64) {
        uint32_t word = 0;
        if (wi == w0 && (at & 15)) word = rg[wi] & ((1u << (2 * (at & 15))) - 1u);
        for (int sl = 0; sl < 16; sl++) {
            const int32_t p = 16 * wi + sl;
            if (p < at || p >= at + m) continue;
            int32_t x = p - at;
            uint32_t op;
            if (x < nD) op = 2u;
            else if ((x -= nD) < nI) op = 1u;
            else if ((x -= nI) < bD) op = 2u;
            else if ((x -= bD) < bI) op = 1u;
            else { x -= bI; const int32_t src = nb - 1 - x; op = (rb[src >> 4] >> (2 * (src & 15))) & 3u; }
            word |= op << (2 * sl);
        }
        rg[wi] = word;
    }
    if (lane == 0) {
        fw.n_ops = at + m;
        wout[r] = fw;
        info[r] = di;
    }
}

__global__ void __launch_bounds__(256) k_sec_count(int64_t n, const Anchor *__restrict__ ancB, uint32_t *__restrict__ count) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t m = __ballot(r < n && ancB[r].aligned != 0);
    if (lane_id() == 0 && m) atomicAdd(count, (uint32_t)__popcll(m));
}
// ---- candidate selection (blasr --bestn 1, unzip.py:86): a read's second candidate replaces the first when its extension
// scored strictly higher; `steps` of the survivor counts the DP steps of both (fzp_aln_summary.cells)
__global__ void __launch_bounds__(256) k_pick(int64_t w_lo, int64_t w_hi, const int32_t *__restrict__ ridx, const Anchor *__restrict__ anc2, const DpInfo *__restrict__ info2,
                                              const int64_t *__restrict__ tb_off2, int64_t tb_base, int64_t mv_base, Anchor *__restrict__ anc, DpInfo *__restrict__ info,
                                              int64_t *__restrict__ tbo, int64_t *__restrict__ mvo, uint8_t *__restrict__ won, int32_t *__restrict__ tbs) {
    const int64_t w = w_lo + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= w_hi) return;
    const int32_t r = ridx[w];
    DpInfo a = info[r];
    const DpInfo b = info2[w];
    const int32_t total = a.steps + b.steps;
    const bool take = b.best_score > a.best_score;
    if (take) {
        a = b;
        anc[r] = anc2[w];
        const int64_t so = tb_off2[w] - tb_off2[w_lo];
        tbo[r] = tb_base + so;
        if (tbs) tbs[r] = 64;                      // the second candidates' streams are their own
        mvo[r] = mv_base + (so >> 6) + (w - w_lo);
    }
    a.steps = total;
    info[r] = a;
    won[w] = take ? 1 : 0;
}
__global__ void __launch_bounds__(256) k_pick_copy(int64_t w_lo, const int32_t *__restrict__ ridx, const uint8_t *__restrict__ won, const int32_t *__restrict__ read_len,
                                                   const uint32_t *__restrict__ sec_ori, const int64_t *__restrict__ sec_woff, const int64_t *__restrict__ read_woff,
                                                   uint32_t *__restrict__ read_ori) {
    const int64_t w = w_lo + blockIdx.x;
    if (!won[w]) return;
    const int32_t r = ridx[w];
    const int64_t nw = (((int64_t)read_len[r] + 15) / 16 + 8 + 1) & ~1LL;
    const uint32_t *src = sec_ori + sec_woff[w];
    uint32_t *dst = read_ori + read_woff[r];
    for (int64_t x = threadIdx.x; x < nw; x += 256) dst[x] = src[x];
}

// ---- gather accepted records into contiguous CIGAR / ASCII SEQ arrays
__global__ void __launch_bounds__(256) k_gather(int64_t n_rec, const int64_t *__restrict__ rec_read, const int64_t *__restrict__ cig_start, const uint32_t *__restrict__ cig,
                                                const int64_t *__restrict__ out_cig_off, uint32_t *__restrict__ out_cig, const uint32_t *__restrict__ read_ori,
                                                const int64_t *__restrict__ read_woff, const int64_t *__restrict__ out_seq_off, uint8_t *__restrict__ out_seq) {
    const int64_t k = blockIdx.x;
    if (k >= n_rec) return;
    const int64_t r = rec_read[k];
    const int64_t nc = out_cig_off[k + 1] - out_cig_off[k];
    const uint32_t *src = cig + cig_start[r];
    uint32_t *dst = out_cig + out_cig_off[k];
    for (int64_t x = (int64_t)blockIdx.y * 256 + threadIdx.x; x < nc; x += (int64_t)gridDim.y * 256) dst[x] = src[x];
    const int64_t n = out_seq_off[k + 1] - out_seq_off[k];
    const uint32_t *pk = read_ori + read_woff[r];
    uint8_t *sq = out_seq + out_seq_off[k];
    for (int64_t x = (int64_t)blockIdx.y * 256 + threadIdx.x; x < n; x += (int64_t)gridDim.y * 256) sq[x] = (uint8_t)("ACGT"[base_at(pk, x)]);
}
// ---- record planning on the device (what `samtools sort` + make_het_call's record filters do to the aligner's output,
// phasing.py:47-75): per contig the aligned reads ordered by (POS, read index) -- q_id = rank in that order -- the
// filters, and the offsets of every accepted record's CIGAR words, SEQ bytes (segments padded to 16) and 64-op checkpoint chunks.
// Slots: the reads of contig c own slots [slot_off[c], slot_off[c+1]); an aligned read lands in slot_off[c] + rank.
struct PlanSlot { uint64_t rec, cig, seq, ck; };           // scanned in place: flags / sizes -> exclusive prefixes
__global__ void __launch_bounds__(256) k_plan_keys(int64_t n, const int32_t *__restrict__ slot_read, const fzp_aln_summary *__restrict__ summ, uint64_t *__restrict__ key) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const int32_t r = slot_read[s];
    key[s] = summ[r].aligned ? (((uint64_t)(uint32_t)summ[r].pos << 32) | (uint32_t)r) : ~0ull;
}
// rank of an aligned read among its contig's aligned reads by (POS, read index).  Contigs of more than 8192 reads: without comparing all
// pairs -- reads are binned by POS >> 8 (k_rank_hist -> scan -> k_rank_scatter), a read's rank = reads in earlier bins + the smaller keys
// inside its own bin (k_rank_binned); smaller ones: k_rank_allpairs
constexpr int RANK_SHIFT = 8;
__global__ void __launch_bounds__(256) k_rank_hist(int64_t n, const int32_t *__restrict__ slot_ctg, const uint64_t *__restrict__ key, const int64_t *__restrict__ bk_off,
                                                   uint32_t *__restrict__ hist) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n || key[s] == ~0ull) return;
    atomicAdd(&hist[bk_off[slot_ctg[s]] + (int64_t)((uint32_t)(key[s] >> 32) >> RANK_SHIFT)], 1u);
}
__global__ void __launch_bounds__(256) k_rank_scatter(int64_t n, const int32_t *__restrict__ slot_ctg, const uint64_t *__restrict__ key, const int64_t *__restrict__ bk_off,
                                                      const uint32_t *__restrict__ start, uint32_t *__restrict__ fill, uint64_t *__restrict__ members) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n || key[s] == ~0ull) return;
    const int64_t b = bk_off[slot_ctg[s]] + (int64_t)((uint32_t)(key[s] >> 32) >> RANK_SHIFT);
    members[start[b] + atomicAdd(&fill[b], 1u)] = key[s];
}
__global__ void __launch_bounds__(256) k_rank_binned(int64_t n, const int32_t *__restrict__ slot_ctg, const uint64_t *__restrict__ key, const int64_t *__restrict__ bk_off,
                                                     const uint32_t *__restrict__ start, const uint32_t *__restrict__ hist, const uint64_t *__restrict__ members,
                                                     uint32_t *__restrict__ rank_out) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const uint64_t k = key[s];
    if (k == ~0ull) return;
    const int c = slot_ctg[s];
    const int64_t b = bk_off[c] + (int64_t)((uint32_t)(k >> 32) >> RANK_SHIFT);
    const uint32_t b0 = start[b], bn = hist[b];
    uint32_t rank = b0 - start[bk_off[c]];
    for (uint32_t x = 0; x < bn; x++) rank += members[b0 + x] < k ? 1u : 0u;
    rank_out[s] = rank;
}
// small contigs (the usual case: a few thousand reads each): all pairs, the contig's keys passing through LDS in tiles of 1024
__global__ void __launch_bounds__(256) k_rank_allpairs(const int64_t *__restrict__ slot_off, const uint64_t *__restrict__ key, uint32_t *__restrict__ rank_out) {
    __shared__ uint64_t tile[1024];
    const int c = blockIdx.y;
    const int64_t s0 = slot_off[c], s1 = slot_off[c + 1];
    if ((int64_t)blockIdx.x * 256 >= s1 - s0) return;
    const int64_t s = s0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t k = s < s1 ? key[s] : ~0ull;
    uint32_t rank = 0;
    for (int64_t t0 = s0; t0 < s1; t0 += 1024) {
        const int m = (int)min((int64_t)1024, s1 - t0);
        __syncthreads();
        for (int x = threadIdx.x; x < m; x += 256) tile[x] = key[t0 + x];
        __syncthreads();
        if (k != ~0ull)
            for (int x = 0; x < m; x++) rank += tile[x] < k ? 1u : 0u;
    }
    if (k != ~0ull) rank_out[s] = rank;
}
__global__ void __launch_bounds__(256) k_plan_rank(int64_t n, const int32_t *__restrict__ slot_read, const int32_t *__restrict__ slot_ctg, const int64_t *__restrict__ slot_off,
                                                   const uint64_t *__restrict__ key, const uint32_t *__restrict__ rank_in, const fzp_aln_summary *__restrict__ summ,
                                                   const int32_t *__restrict__ read_len,
                                                   uint64_t *__restrict__ v_rec, uint64_t *__restrict__ v_cig, uint64_t *__restrict__ v_seq, uint64_t *__restrict__ v_ck,
                                                   int32_t *__restrict__ g_read, int32_t *__restrict__ g_qid, uint8_t *__restrict__ g_acc, int32_t *__restrict__ last_pos,
                                                   uint32_t *__restrict__ n_aligned, unsigned long long *__restrict__ n_cols) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = s < n && key[s] != ~0ull;          // every lane stays to the end: the per-contig totals are reduced over the wave
    int c = -1;
    bool acc = false;
    int32_t pos = -1, ncol = 0;
    if (valid) {
        c = slot_ctg[s];
        const uint32_t rank = rank_in[s];
        const int32_t r = slot_read[s];
        const fzp_aln_summary sm = summ[r];
        const int64_t nlen = read_len[r];
        const int64_t n_del = (int64_t)(sm.ref_end - sm.pos) - sm.n_columns;
        const int64_t total_aln_pos = nlen + n_del;                       // sum of all CIGAR op lengths
        const int64_t skip_base = (int64_t)sm.q_start + (nlen - sm.q_end);   // soft clips
        // phasing.py:72 in IEEE double exactly as written (no contraction: explicit round-to-nearest ops)
        const double frac = __dsub_rn(1.0, __ddiv_rn(__dmul_rn(1.0, (double)skip_base), (double)total_aln_pos));
        acc = !(frac < 0.1) && !(total_aln_pos < 2000);      // phasing.py:72, 74
        const int64_t g = slot_off[c] + rank;
        g_read[g] = r; g_qid[g] = (int32_t)rank; g_acc[g] = acc ? 1 : 0;
        if (acc) { v_rec[g] = 1; v_cig[g] = (uint64_t)sm.n_cigar; v_seq[g] = (uint64_t)((nlen + 15) & ~15ll); v_ck[g] = (uint64_t)((sm.n_cigar + 63) / 64); }
        pos = sm.pos; ncol = sm.n_columns;
    }
    // per-contig totals: the slots of a wave nearly always belong to one contig -> one atomic per wave, not per read
    // (40 000 same-address atomics issued at once serialise: 0.5 ms)
    const uint64_t vm = __ballot(valid);
    if (vm == 0) return;
    const int c0 = __builtin_amdgcn_readlane(c, __builtin_ctzll(vm));
    if (__all(!valid || c == c0)) {
        int32_t mp = (valid && acc) ? pos : -1;
        unsigned long long cols = (valid && acc) ? (unsigned long long)ncol : 0ull;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { mp = max(mp, __shfl_xor(mp, d, 64)); cols += __shfl_xor(cols, d, 64); }
        if (lane_id() == 0) {
            atomicAdd(&n_aligned[c0], (uint32_t)__popcll(vm));
            if (mp >= 0) atomicMax(&last_pos[c0], mp);
            if (cols) atomicAdd(&n_cols[c0], cols);
        }
    } else if (valid) {
        atomicAdd(&n_aligned[c], 1u);
        if (acc) { atomicMax(&last_pos[c], pos); atomicAdd(&n_cols[c], (unsigned long long)ncol); }
    }
}
__global__ void __launch_bounds__(256) k_plan_emit(int64_t n, int n_ctg, const int32_t *__restrict__ slot_ctg_of_g, const int64_t *__restrict__ slot_off,
                                                   const uint64_t *__restrict__ v_rec, const uint64_t *__restrict__ v_cig, const uint64_t *__restrict__ v_seq,
                                                   const uint64_t *__restrict__ v_ck, const int32_t *__restrict__ g_read, const int32_t *__restrict__ g_qid,
                                                   const uint8_t *__restrict__ g_acc, const fzp_aln_summary *__restrict__ summ, const uint64_t *__restrict__ totals,
                                                   int64_t *__restrict__ rec_read, int32_t *__restrict__ rec_qid, int32_t *__restrict__ rec_pos, int32_t *__restrict__ rec_ctg,
                                                   int64_t *__restrict__ cig_off, int64_t *__restrict__ seq_off, int64_t *__restrict__ ck_off, int64_t *__restrict__ rec_begin) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g == 0) {
        const int64_t nr = (int64_t)totals[0];
        cig_off[nr] = (int64_t)totals[1]; seq_off[nr] = (int64_t)totals[2]; ck_off[nr] = (int64_t)totals[3];
    }
    if (g <= n_ctg) { const int64_t so = slot_off[g]; rec_begin[g] = so < n ? (int64_t)v_rec[so] : (int64_t)totals[0]; }   // also right for contigs without reads
    if (g >= n) return;
    const int c = slot_ctg_of_g[g];
    if (!g_acc[g]) return;
    const int64_t k = (int64_t)v_rec[g];
    const int32_t r = g_read[g];
    rec_read[k] = r; rec_qid[k] = g_qid[g]; rec_pos[k] = summ[r].pos; rec_ctg[k] = c;
    cig_off[k] = (int64_t)v_cig[g]; seq_off[k] = (int64_t)v_seq[g]; ck_off[k] = (int64_t)v_ck[g];
}
// the batch path's variant: SEQ segments are padded to 16 bytes, so a thread turns one packed word into one 16-byte store
__global__ void __launch_bounds__(256) k_gather16(int64_t n_rec, const int64_t *__restrict__ rec_read, const int64_t *__restrict__ cig_start, const uint32_t *__restrict__ cig,
                                                  const int64_t *__restrict__ out_cig_off, uint32_t *__restrict__ out_cig, const uint32_t *__restrict__ read_ori,
                                                  const int64_t *__restrict__ read_woff, const int64_t *__restrict__ out_seq_off, uint8_t *__restrict__ out_seq) {
    const int64_t k = blockIdx.x;
    if (k >= n_rec) return;
    const int64_t r = rec_read[k];
    const int64_t nc = out_cig_off[k + 1] - out_cig_off[k];
    const uint32_t *src = cig + cig_start[r];
    uint32_t *dst = out_cig + out_cig_off[k];
    for (int64_t x = (int64_t)blockIdx.y * 256 + threadIdx.x; x < nc; x += (int64_t)gridDim.y * 256) dst[x] = src[x];
    const int64_t nw = (out_seq_off[k + 1] - out_seq_off[k]) >> 4;
    const uint32_t *pk = read_ori + read_woff[r];
    uint4 *sq = (uint4 *)(out_seq + out_seq_off[k]);
    for (int64_t w = (int64_t)blockIdx.y * 256 + threadIdx.x; w < nw; w += (int64_t)gridDim.y * 256) {
        const uint32_t x = pk[w];
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t t = (x >> (8 * q)) & 0xffu;                 // four 2-bit codes -> one per byte
            t = (t | (t << 12)) & 0x000F000Fu;
            t = (t | (t << 6)) & 0x03030303u;
            const uint32_t c2 = (t >> 1) & 0x01010101u, c3 = c2 & t;     // code >= 2, code == 3
            o[q] = 0x41414141u + 2u * t + 2u * c2 + 11u * c3;            // A=65 C=67 G=71 T=84
        }
        sq[w] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
}  // namespace

// ================================================================================ job
// One forward DP at a time per device.  The DP kernels of two jobs (two contexts in one process: the lanes of fzp_phase_contigs, two steps in flight) running side by side
// land on each other's SIMDs and both run at little more than half speed; queued one behind the other each has the chip to itself, and everything else
// of the two jobs still overlaps.  The chain is made of events: a job's DP launches wait for the event the previous job recorded behind its own (no host thread ever blocks).
static std::mutex g_dp_mu;
static std::map<int, hipEvent_t> g_dp_last;

struct fzp_alnjob {
    int32_t n_ctg = 0;
    int64_t n_reads = 0;
    fzp_align_params P;
    std::vector<std::vector<uint8_t>> h_ctg;     // upper-cased ASCII, for the phasing batch's ref_seq
    std::vector<int64_t> h_ctg_len, h_ctg_woff, h_idx_off, h_read_woff, h_tb_off, h_cig_off;
    std::vector<int32_t> h_idx_bits, h_read_len, h_read_ctg;
    std::vector<fzp_aln_summary> h_summ;
    int64_t ctg_words = 0, read_words = 0, idx_slots = 0;
    DevBuf<uint32_t> ctg_pk, read_pk, read_ori, cig;
    DevBuf<uint8_t> ctg_ascii;                   // upper-cased contigs, concatenated (ref_seq of the phasing batch)
    std::vector<int64_t> h_ctg_aoff;
    DevBuf<int64_t> ctg_woff, ctg_len, idx_off, read_woff, tb_off, cig_off, cig_start;
    DevBuf<int32_t> idx_bits, read_len, read_ctg;
    DevBuf<uint64_t> table;
    DevBuf<uint32_t> part_cursor;                // k-mer index build: staged entries per partition
    DevBuf<int32_t> part_ctg, idx_overflow;
    bool index_built = false;
    DevBuf<int64_t> part_off;
    std::vector<int64_t> h_part_off;
    int64_t n_parts = 0;
    DevBuf<Anchor> anc, ancB, anc2;              // first candidates (per read), second candidates (per read; compacted)
    DevBuf<int64_t> tbo, mvo;                    // per read: where the winning candidate's trace-back masks / move words are
    DevBuf<uint2> hits;                          // seeding: HIT_CAP hit slots per read of a seeding launch
    DevBuf<SeedWin> win;
    DevBuf<uint32_t> n_sec;                      // reads with a second candidate
    DevBuf<int32_t> ridx;                        // compacted second candidates -> read
    DevBuf<int64_t> sec_woff, tb_off2;
    DevBuf<uint32_t> sec_ori;
    DevBuf<DpInfo> info2;
    DevBuf<uint8_t> won;
    int64_t n_second = 0;                        // of the last run
    DevBuf<DpInfo> info;
    DevBuf<uint2> tb2[2];
    DevBuf<uint32_t> raw2[2];                    // the walk's 2-bit op streams
    DevBuf<WalkOut> wout;
    DevBuf<uint8_t> seg_single;                  // per read: 1 = short enough for one walker
    DevBuf<int32_t> seg_order;                   // per launched walker slot: the walker that takes it
    DevBuf<int32_t> seg_off, seg_slot, seg_idx;  // segmented trace-back: per read its first walker (chunk-relative); per walker its slot (chunk-relative) and segment
    std::vector<int64_t> h_seg_base, h_seg_cnt;  // per chunk start (indexed by its first read): first walker in seg_slot / seg_idx, number of walkers
    DevBuf<uint32_t> raw_seg2[2], trail2[2];
    DevBuf<SegOut> segout2[2];
    DevBuf<uint32_t> tb_fallback;                // [0] reads of the last run walked serially after all, [1] repair walks asked for in the chunk at hand, [2] in the whole run
    DevBuf<SegReq> seg_req;
    DevBuf<uint32_t> dp_flag;                    // "the bit-sliced DP kernel of launch dp_seq runs" (k_wait_started)
    uint32_t dp_seq = 0;
    DevBuf<uint8_t> b_handled;                   // per read: its backward extension ran in the bit-sliced kernel
    DevBuf<int32_t> tbs;                         // per read: records from one 64-step block of its masks to the next (64: a stream of its own; 4096: interleaved with its launch group)
    std::vector<int64_t> h_tbm_total;            // per chunk (by its first read): records its planned mask streams span
    DevBuf<int64_t> tbm_off, mvm_off;            // per read: where its masks / move words go in its chunk's buffers, planned in LAUNCH order (longest first)
    DevBuf<int32_t> swb_list, sw_list;           // per run: the chunk's slots by DP kernel (k_swb: 64 per wave, -1 padded; k_sw: its launch order)
    std::vector<int32_t> h_swb_list, h_sw_list, h_lpt;
    std::vector<int64_t> h_swb_at, h_sw_at;      // per chunk (by its first read): where its lists start, and their sizes behind
    DevBuf<int32_t> lpt;                         // per read: the slot (relative to its chunk's first read) that wave / lane number x of the chunk's launches takes --
    int64_t lpt_chunk_steps = -1;                // longest reads first (k_sw, k_tb_walk); rebuilt when the chunking changes
    // record planning: reads grouped by contig (input order inside a contig); built on first use
    DevBuf<int32_t> slot_read, slot_ctg;
    DevBuf<int64_t> slot_off;
    std::vector<int64_t> h_slot_off;
    DevBuf<int64_t> rank_bk_off;                 // per contig: first POS bin of the record planning's rank (k_rank_*)
    int64_t n_rank_buckets = 0;
    bool have_slots = false;
    int64_t max_reads_per_ctg = 1;
    bool summ_on_host = false;
    DevBuf<ulonglong2> mvw2[2];
    // backward extension (v1.4): slot = read; capacities planned by the host from the candidates' anchors at every run
    DevBuf<Anchor> anc_b;
    DevBuf<DpInfo> info_b;
    DevBuf<int32_t> b_len, b_iota, b_order;
    DevBuf<int64_t> b_tlen, bq_off, bt_off, tb_off_b, tbo_b, mvo_b;
    DevBuf<uint32_t> bq, bt, raw_b2[2];
    DevBuf<uint2> tb_b2[2];
    DevBuf<ulonglong2> mvw_b2[2];
    DevBuf<WalkOut> wout_b;
    std::vector<int64_t> h_tb_off_b;
    hipEvent_t ev_sw[2] = {nullptr, nullptr}, ev_tb[2] = {nullptr, nullptr}, ev_bk[2] = {nullptr, nullptr}, ev_l[2] = {nullptr, nullptr};
    DevBuf<fzp_aln_summary> summ;
    bool done = false;
};

extern "C" void fzp_align_params_default(fzp_align_params *p) {
    memset(p, 0, sizeof *p);
    p->kmer = 16; p->seed_stride = 4; p->match = 2; p->mismatch = 4; p->gap = 3; p->min_seed_hits = 8;
    p->min_pct_identity = 70;
}

extern "C" void fzp_align_destroy(fzp_ctx *ctx, fzp_alnjob *job) {
    if (!job) return;
    if (ctx) { (void)fzp_bind(ctx); (void)hipStreamSynchronize(ctx->stream); (void)hipStreamSynchronize(ctx->stream2); }
    for (int k = 0; k < 2; k++) { if (job->ev_sw[k]) (void)hipEventDestroy(job->ev_sw[k]); if (job->ev_tb[k]) (void)hipEventDestroy(job->ev_tb[k]); if (job->ev_bk[k]) (void)hipEventDestroy(job->ev_bk[k]); if (job->ev_l[k]) (void)hipEventDestroy(job->ev_l[k]); }
    delete job;
}

// The contigs' k-mer tables (k_index_stage + k_index_build).  They depend on the contigs and on P.kmer only, so they are built once, by
// fzp_align_create right after the contigs are packed, and every fzp_align_run of the job reuses them (FZP_INDEX_PER_RUN=1: rebuilt per run).
static int build_index(fzp_ctx *ctx, fzp_alnjob *j) {
    hipStream_t st = ctx->stream;
    const fzp_align_params &P = j->P;
    {
        ProfScope ps(ctx, "k1_index");
        int64_t lc_max = 0;
        for (auto v : j->h_ctg_len) lc_max = std::max(lc_max, v);
        const unsigned gx = (unsigned)std::max<int64_t>(1, ((lc_max + CTG_STRIDE - 1) / CTG_STRIDE + STAGE_KMERS - 1) / STAGE_KMERS);
        FZP_HIP(hipMemsetAsync(j->part_cursor.p, 0, (size_t)j->n_parts * 4, st));
        FZP_HIP(hipMemsetAsync(j->idx_overflow.p, 0, 4, st));
        hipLaunchKernelGGL(k_index_stage, dim3(gx, j->n_ctg), dim3(256), 0, st, j->ctg_pk.p, j->ctg_woff.p, j->ctg_len.p, j->idx_off.p, j->idx_bits.p, j->part_off.p, P.kmer,
                           j->table.p, j->part_cursor.p, j->idx_overflow.p);
        const size_t lds = (size_t)(4u << PART_BITS) * 8;
        FZP_HIP(hipFuncSetAttribute((const void *)k_index_build, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_index_build, dim3((unsigned)j->n_parts), dim3(256), lds, st, j->part_ctg.p, j->part_off.p, j->idx_off.p, j->idx_bits.p, j->table.p, j->part_cursor.p,
                           j->idx_overflow.p);
    }
    j->index_built = true;
    return FZP_OK;
}

extern "C" int fzp_align_create(fzp_ctx *ctx, int32_t n_ctg, const uint8_t *const *ctg_seq, const int64_t *ctg_len, int64_t n_reads, const int32_t *read_ctg,
                                const int64_t *read_off, const uint8_t *read_seq, const fzp_align_params *params, fzp_alnjob **out) {
    if (!ctx || !out || n_ctg <= 0 || !ctg_seq || !ctg_len || n_reads < 0 || (n_reads && (!read_ctg || !read_off || !read_seq))) {
        fzp_set_error("fzp_align_create: bad arguments");
        return FZP_EINVAL;
    }
    *out = nullptr;
    FZP_TRY(fzp_bind(ctx));
    fzp_alnjob *j = new fzp_alnjob();
    if (params) j->P = *params; else fzp_align_params_default(&j->P);
    if (j->P.kmer < 8 || j->P.kmer > 16 || j->P.seed_stride < 1 || j->P.match <= 0 || j->P.mismatch < 0 || j->P.gap <= 0 || j->P.min_pct_identity < 0 ||
        j->P.min_pct_identity > 100) {
        delete j; fzp_set_error("fzp_align_create: bad parameters"); return FZP_EINVAL;
    }
    j->n_ctg = n_ctg; j->n_reads = n_reads;
    hipStream_t st = ctx->stream;
    int rc = FZP_OK;
    // contigs
    std::vector<int64_t> coff(1, 0);
    j->h_ctg.resize(n_ctg);                       // upper-cased host copies: fetched from the device when fzp_align_alnset needs them
    for (int c = 0; c < n_ctg; c++) {
        if (ctg_len[c] < 0 || ctg_len[c] > 0x7fff0000LL) { delete j; fzp_set_error("contig %d: length out of range", c); return FZP_EINVAL; }
        j->h_ctg_len.push_back(ctg_len[c]);
        j->h_ctg_woff.push_back(j->ctg_words);
        j->ctg_words += ((ctg_len[c] + 15) / 16 + 8 + 1) & ~1LL;
        int64_t nk = ctg_len[c] - j->P.kmer + 1;
        int bits = 10;
        while ((1LL << bits) < 2 * std::max<int64_t>((nk + CTG_STRIDE - 1) / CTG_STRIDE, 1)) bits++;
        j->h_idx_bits.push_back(bits);
        j->h_idx_off.push_back(j->idx_slots);
        j->idx_slots += 1LL << bits;
        coff.push_back(coff.back() + ((ctg_len[c] + 15) & ~15LL));      // 16-byte aligned segments (k_upper works on whole uint4s)
    }
    // reads
    j->h_read_woff.assign(1, 0);
    j->h_tb_off.assign(1, 0);
    j->h_cig_off.assign(1, 0);
    for (int64_t r = 0; r < n_reads; r++) {
        int64_t n = read_off[r + 1] - read_off[r];
        if (n < 0 || n > 0x3fff0000LL || read_ctg[r] < 0 || read_ctg[r] >= n_ctg) { delete j; fzp_set_error("read %lld: bad length/contig", (long long)r); return FZP_EINVAL; }
        if (n * (int64_t)j->P.match >= (1LL << 26) - (1 << 20)) {   // biased score << 5 must fit 32 bits (k_sw best-cell key)
            delete j; fzp_set_error("read %lld: %lld bases x match %d exceeds the score range of the DP kernel", (long long)r, (long long)n, j->P.match); return FZP_EINVAL;
        }
        j->h_read_len.push_back((int32_t)n);
        j->h_read_ctg.push_back(read_ctg[r]);
        j->read_words += ((n + 15) / 16 + 8 + 1) & ~1LL;
        j->h_read_woff.push_back(j->read_words);
        j->h_tb_off.push_back(j->h_tb_off.back() + (n + n + n / 4 + 64 + 2 + 63) / 64 * 64);   // steps capacity, multiple of 64
        j->h_cig_off.push_back(j->h_cig_off.back() + n + 18);
    }
    DevBuf<uint8_t> d_ascii;
    DevBuf<int64_t> d_off;
    do {
        // contigs: ASCII straight into ctg_ascii (pinned, chunked, threaded staging), upper-cased and packed on the device
        if ((rc = j->ctg_pk.alloc((size_t)j->ctg_words + 8)) || (rc = j->ctg_ascii.alloc((size_t)coff.back() + 16)) || (rc = d_off.upload(coff.data(), coff.size(), st)) ||
            (rc = j->ctg_woff.upload(j->h_ctg_woff.data(), j->h_ctg_woff.size(), st)) || (rc = j->ctg_len.upload(j->h_ctg_len.data(), j->h_ctg_len.size(), st)) ||
            (rc = j->idx_off.upload(j->h_idx_off.data(), j->h_idx_off.size(), st)) || (rc = j->idx_bits.upload(j->h_idx_bits.data(), j->h_idx_bits.size(), st)))
            break;
        {
            std::vector<const void *> srcs; std::vector<size_t> dsts, lens;
            for (int c = 0; c < n_ctg; c++) { srcs.push_back(ctg_seq[c]); dsts.push_back((size_t)coff[(size_t)c]); lens.push_back((size_t)ctg_len[c]); }
            if ((rc = fzp_upload_segments(ctx, j->ctg_ascii.p, srcs, dsts, lens, st))) break;
        }
        hipLaunchKernelGGL(k_upper, dim3((unsigned)((coff.back() / 16 + 255) / 256 + 1)), dim3(256), 0, st, j->ctg_ascii.p, coff.back());
        {   // k_pack reads [seq_off[s], seq_off[s+1]): the contig lengths, not the padded segments -> per-contig begin / end pairs
            std::vector<int64_t> be;
            for (int c = 0; c < n_ctg; c++) { be.push_back(coff[(size_t)c]); be.push_back(coff[(size_t)c] + ctg_len[c]); }
            DevBuf<int64_t> d_be;
            if ((rc = d_be.upload(be.data(), be.size(), st))) break;
            hipLaunchKernelGGL(k_pack2, dim3(n_ctg, 64), dim3(256), 0, st, j->ctg_ascii.p, d_be.p, j->ctg_woff.p, j->ctg_pk.p);
            if (hipStreamSynchronize(st) != hipSuccess) { rc = FZP_EDEVICE; break; }
        }
        j->h_ctg_aoff = coff;
        if (n_reads) {
            const size_t rbytes = (size_t)(read_off[n_reads] - read_off[0]);
            if ((rc = j->read_pk.alloc((size_t)j->read_words + 8)) || (rc = j->read_ori.alloc((size_t)j->read_words + 8)) || (rc = d_ascii.alloc(rbytes + 16)))
                break;
            {
                std::vector<const void *> srcs(1, read_seq + read_off[0]); std::vector<size_t> dsts(1, 0), lens(1, rbytes);
                if ((rc = fzp_upload_segments(ctx, d_ascii.p, srcs, dsts, lens, st))) break;
            }
            std::vector<int64_t> roff((size_t)n_reads + 1);
            for (int64_t r = 0; r <= n_reads; r++) roff[(size_t)r] = read_off[r] - read_off[0];
            if ((rc = d_off.upload(roff.data(), roff.size(), st)) || (rc = j->read_woff.upload(j->h_read_woff.data(), j->h_read_woff.size(), st)) ||
                (rc = j->read_len.upload(j->h_read_len.data(), j->h_read_len.size(), st)) || (rc = j->read_ctg.upload(j->h_read_ctg.data(), j->h_read_ctg.size(), st)) ||
                (rc = j->tb_off.upload(j->h_tb_off.data(), j->h_tb_off.size(), st)) || (rc = j->cig_off.upload(j->h_cig_off.data(), j->h_cig_off.size(), st)))
                break;
            hipLaunchKernelGGL(k_pack, dim3((unsigned)n_reads, 1), dim3(256), 0, st, d_ascii.p, d_off.p, j->read_woff.p, j->read_pk.p);
            if (hipStreamSynchronize(st) != hipSuccess) { rc = FZP_EDEVICE; break; }
        }
        {   // partitions of every contig's table (k_index_stage / k_index_build)
            std::vector<int32_t> pc;
            j->h_part_off.assign(1, 0);
            for (int c = 0; c < n_ctg; c++) {
                const int bbits = j->h_idx_bits[(size_t)c] - 2;
                const int np = 1 << (bbits - std::min(bbits, PART_BITS));
                if (np > 4096) { rc = FZP_EINVAL; fzp_set_error("contig %d: k-mer index of %d partitions (limit 4096: contigs up to ~33 Mb)", c, np); break; }
                for (int q = 0; q < np; q++) pc.push_back(c);
                j->h_part_off.push_back(j->h_part_off.back() + np);
            }
            if (rc) break;
            j->n_parts = j->h_part_off.back();
            if ((rc = j->part_ctg.upload(pc.data(), pc.size(), st)) || (rc = j->part_off.upload(j->h_part_off.data(), j->h_part_off.size(), st)) ||
                (rc = j->part_cursor.alloc((size_t)j->n_parts)) || (rc = j->idx_overflow.alloc(1)))
                break;
            if (hipStreamSynchronize(st) != hipSuccess) { rc = FZP_EDEVICE; break; }
        }
        if ((rc = j->table.alloc((size_t)j->idx_slots)) || (rc = j->anc.alloc((size_t)n_reads)) || (rc = j->ancB.alloc((size_t)n_reads)) || (rc = j->tbo.alloc((size_t)n_reads)) ||
            (rc = j->mvo.alloc((size_t)n_reads)) || (rc = j->n_sec.alloc(1)) || (rc = j->info.alloc((size_t)n_reads)) ||
            (rc = j->summ.alloc((size_t)n_reads)) || (rc = j->cig.alloc((size_t)j->h_cig_off.back())) || (rc = j->cig_start.alloc((size_t)n_reads)))
            break;
        if ((rc = build_index(ctx, j))) break;
    } while (0);
    if (rc == FZP_OK && hipGetLastError() != hipSuccess) rc = FZP_EDEVICE;
    if (rc) { if (rc == FZP_EDEVICE) fzp_set_error("fzp_align_create: device error"); delete j; return rc; }
    *out = j;
    return FZP_OK;
}

// the next fzp_align_run rebuilds the k-mer tables (a job that sees its contigs once pays for them inside its run: bench.py's step)
extern "C" int fzp_align_invalidate_index(fzp_alnjob *j) {
    if (!j) return FZP_EINVAL;
    j->index_built = false;
    return FZP_OK;
}

extern "C" int fzp_align_run(fzp_ctx *ctx, fzp_alnjob *j) {
    if (!ctx || !j) return FZP_EINVAL;
    FZP_TRY(fzp_bind(ctx));
    hipStream_t st = ctx->stream;
    const fzp_align_params &P = j->P;
    if (!j->index_built || getenv("FZP_INDEX_PER_RUN")) FZP_TRY(build_index(ctx, j));     // normally built by fzp_align_create
    const int64_t nr = j->n_reads;
    if (nr > 0) {
        {
            ProfScope ps(ctx, "k1_seed");
            int64_t lc_max = 0, n_max = 0;
            for (auto v : j->h_ctg_len) lc_max = std::max(lc_max, v);
            for (auto v : j->h_read_len) n_max = std::max<int64_t>(n_max, v);
            const int64_t nb_max = std::min<int64_t>(MAX_BINS, ((lc_max + n_max) >> 10) + 2);
            const size_t lds = (size_t)2 * (size_t)nb_max * sizeof(uint32_t);
            FZP_HIP(hipFuncSetAttribute((const void *)k_seed, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            FZP_HIP(hipMemsetAsync(j->n_sec.p, 0, 4, st));
            const int64_t seed_chunk = 65536;            // reads per seeding launch: HIT_CAP x 8 B of hit list each (2 GiB)
            FZP_TRY(j->hits.alloc((size_t)std::min<int64_t>(nr, seed_chunk) * HIT_CAP));
            FZP_TRY(j->win.alloc((size_t)std::min<int64_t>(nr, seed_chunk)));
            for (int64_t f0 = 0; f0 < nr; f0 += seed_chunk) {
                const int64_t cn = std::min<int64_t>(seed_chunk, nr - f0);
                hipLaunchKernelGGL(k_seed, dim3((unsigned)cn), dim3(256), lds, st, f0, j->read_pk.p, j->read_woff.p, j->read_len.p,
                                   j->read_ctg.p, j->ctg_len.p, j->idx_off.p, j->idx_bits.p, j->table.p, P.kmer, P.seed_stride, P.min_seed_hits, j->hits.p, j->win.p);
                hipLaunchKernelGGL(k_chain, dim3((unsigned)(2 * cn)), dim3(64), 0, st, f0, cn, j->read_len.p, j->hits.p, j->win.p, j->anc.p, j->ancB.p);
            }
            hipLaunchKernelGGL(k_sec_count, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, st, nr, j->ancB.p, j->n_sec.p);
        }
        {
            ProfScope ps(ctx, "k1_orient");
            hipLaunchKernelGGL(k_orient, dim3((unsigned)nr, 1), dim3(256), 0, st, (int64_t)0, j->read_pk.p, j->read_woff.p, j->read_len.p, j->anc.p, (const int32_t *)nullptr,
                               j->read_woff.p, j->read_ori.p);
        }
        // second candidates (reads whose votes show a second placement: repeats).  Usually none; then nothing below runs.
        uint32_t n2 = 0;
        int32_t ovf = 0;
        std::vector<Anchor> h_anc((size_t)nr);              // the first candidates' anchors: how far back an extension may have to reach (v1.4)
        FZP_HIP(hipMemcpyAsync(&n2, j->n_sec.p, 4, hipMemcpyDeviceToHost, st));
        FZP_HIP(hipMemcpyAsync(&ovf, j->idx_overflow.p, 4, hipMemcpyDeviceToHost, st));
        FZP_TRY(j->anc.download(h_anc.data(), (size_t)nr, st));
        FZP_HIP(hipStreamSynchronize(st));
        std::vector<int32_t> b_cap((size_t)nr);
        for (int64_t r = 0; r < nr; r++) b_cap[(size_t)r] = (h_anc[(size_t)r].aligned && h_anc[(size_t)r].c_a > 0) ? h_anc[(size_t)r].i_a : 0;
        if (ovf) { fzp_set_error("k-mer index: a table partition overflowed (more than %d distinct k-mers hash into one 64 KB partition)", 4 << PART_BITS); return FZP_EINVAL; }
        j->n_second = n2;
        std::vector<int32_t> h_ridx;
        std::vector<int64_t> h_tb_off2(1, 0);
        if (n2) {
            std::vector<Anchor> hb((size_t)nr), h2;
            FZP_TRY(j->ancB.download(hb.data(), (size_t)nr, st));
            FZP_HIP(hipStreamSynchronize(st));
            std::vector<int64_t> h_woff2(1, 0);
            for (int64_t r = 0; r < nr; r++) {
                if (!hb[(size_t)r].aligned) continue;
                const int64_t n = j->h_read_len[(size_t)r];
                h_ridx.push_back((int32_t)r);
                h2.push_back(hb[(size_t)r]);
                if (hb[(size_t)r].c_a > 0) b_cap[(size_t)r] = std::max(b_cap[(size_t)r], hb[(size_t)r].i_a);     // whichever candidate wins
                h_woff2.push_back(h_woff2.back() + (((n + 15) / 16 + 8 + 1) & ~1LL));
                h_tb_off2.push_back(h_tb_off2.back() + (n + n + n / 4 + 64 + 2 + 63) / 64 * 64);
            }
            n2 = (uint32_t)h_ridx.size();
            FZP_TRY(j->ridx.upload(h_ridx.data(), n2, st)); FZP_TRY(j->anc2.upload(h2.data(), n2, st));
            FZP_TRY(j->sec_woff.upload(h_woff2.data(), h_woff2.size(), st)); FZP_TRY(j->tb_off2.upload(h_tb_off2.data(), h_tb_off2.size(), st));
            FZP_TRY(j->sec_ori.alloc((size_t)h_woff2.back() + 8)); FZP_TRY(j->info2.alloc(n2)); FZP_TRY(j->won.alloc(n2));
            FZP_HIP(hipStreamSynchronize(st));      // the staging vectors die with this scope
            ProfScope ps(ctx, "k1_orient");
            hipLaunchKernelGGL(k_orient, dim3(n2, 1), dim3(256), 0, st, (int64_t)0, j->read_pk.p, j->read_woff.p, j->read_len.p, j->anc2.p, j->ridx.p, j->sec_woff.p, j->sec_ori.p);
        }
        // ---- backward extension (v1.4): per read room for the reversed prefix (as long as the deeper of its candidates' anchors), the reversed contig
        // window and the masks of that DP.  Typical anchors sit a few hundred bases into the read: a few per cent of the forward work.
        {
            std::vector<int64_t> qo((size_t)nr + 1, 0), to((size_t)nr + 1, 0);
            j->h_tb_off_b.assign((size_t)nr + 1, 0);
            for (int64_t r = 0; r < nr; r++) {
                const int64_t cq = b_cap[(size_t)r], ct = cq ? cq + cq / 4 + 64 : 0;
                qo[(size_t)r + 1] = qo[(size_t)r] + (cq ? (((cq + 15) / 16 + 8 + 1) & ~1LL) : 0);
                to[(size_t)r + 1] = to[(size_t)r] + (cq ? (((ct + 15) / 16 + 8 + 1) & ~1LL) : 0);
                j->h_tb_off_b[(size_t)r + 1] = j->h_tb_off_b[(size_t)r] + (cq ? (cq + ct + 2 + 63) / 64 * 64 : 0);
            }
            FZP_TRY(j->bq_off.upload(qo.data(), qo.size(), st)); FZP_TRY(j->bt_off.upload(to.data(), to.size(), st));
            FZP_TRY(j->tb_off_b.upload(j->h_tb_off_b.data(), j->h_tb_off_b.size(), st));
            FZP_TRY(j->bq.alloc((size_t)qo.back() + 16)); FZP_TRY(j->bt.alloc((size_t)to.back() + 16));
            FZP_TRY(j->anc_b.alloc((size_t)nr)); FZP_TRY(j->info_b.alloc((size_t)nr)); FZP_TRY(j->b_len.alloc((size_t)nr)); FZP_TRY(j->b_tlen.alloc((size_t)nr));
            FZP_TRY(j->tbo_b.alloc((size_t)nr)); FZP_TRY(j->mvo_b.alloc((size_t)nr)); FZP_TRY(j->wout_b.alloc((size_t)nr));
            if (j->b_iota.n < (size_t)nr) {
                std::vector<int32_t> io((size_t)nr);
                for (int64_t r = 0; r < nr; r++) io[(size_t)r] = (int32_t)r;
                FZP_TRY(j->b_iota.upload(io.data(), (size_t)nr, st));
            }
            FZP_HIP(hipStreamSynchronize(st));      // the staging vectors die with this scope
        }
        // Trace-back masks live in HBM (16 B per DP step).  Reads go through in chunks: the DP of chunk k+1
        // (integer-VALU bound, every wave slot busy, no LDS) runs on `stream` while the trace-back of chunk k
        // (latency bound, 2 LDS-heavy waves per CU) runs on `stream2`; two mask buffers alternate.
        int64_t budget_steps = (int64_t)48 << 30 >> 4;   // 48 GiB of 16-byte steps over both buffers
        if (const char *e = getenv("FZP_TB_BUDGET_GB")) { long g = atol(e); if (g > 0) budget_steps = ((int64_t)g << 30) >> 4; }
        int n_chunks = 1;   // measured: overlapping the two kernels costs more than it hides (contention, chunk tails)
        if (const char *e = getenv("FZP_SW_CHUNKS")) { int g = atoi(e); if (g > 0) n_chunks = g; }
        bool split_rounds = false;      // measured (r2): 43.4 vs 41.3 ms for K1 at cfg2 -- the trace-back under a second DP launch runs at 1/9 of a SIMD's issue slots
        if (const char *e = getenv("FZP_SW_SPLIT_ROUNDS")) split_rounds = atoi(e) != 0;
        const int64_t total_steps = j->h_tb_off[(size_t)nr];
        int64_t chunk_steps = std::min<int64_t>(budget_steps / 2, (total_steps + n_chunks - 1) / n_chunks);
        FZP_TRY(j->wout.alloc((size_t)nr));
        FZP_TRY(j->tb_fallback.alloc(4));
        FZP_TRY(j->tb_fallback.zero(4, st));
        if (!j->ev_sw[0]) for (int k = 0; k < 2; k++) { FZP_HIP(hipEventCreateWithFlags(&j->ev_l[k], hipEventDisableTiming)); FZP_HIP(hipEventCreateWithFlags(&j->ev_sw[k], hipEventDisableTiming)); FZP_HIP(hipEventCreateWithFlags(&j->ev_tb[k], hipEventDisableTiming)); FZP_HIP(hipEventCreateWithFlags(&j->ev_bk[k], hipEventDisableTiming)); }
        hipStream_t st2 = ctx->stream2;
        if (j->lpt_chunk_steps != chunk_steps || split_rounds) {
            // launch order inside every chunk of reads: longest first (LPT over the wave slots).  One wave per read, workgroups dispatched in
            // index order: with reads of uneven length in input order the grid's tail is whatever long read happened to come last.
            std::vector<int32_t> ord((size_t)nr), sgo((size_t)nr), sgs, sgi, sgw;
            std::vector<int64_t> h_tbm((size_t)nr), h_mvm((size_t)nr);
            std::vector<uint8_t> sg1((size_t)nr);
            j->h_seg_base.assign((size_t)nr + 1, 0); j->h_seg_cnt.assign((size_t)nr + 1, 0);
            for (int64_t f = 0; f < nr;) {
                int64_t l = f;
                while (l < nr && j->h_tb_off[(size_t)l + 1] - j->h_tb_off[(size_t)f] <= chunk_steps) l++;
                if (l == f) l = f + 1;
                if (split_rounds && f == 0 && l == nr) {
                    const int64_t slots = (int64_t)ctx->n_cu * 32, full = nr / slots * slots;
                    if (full >= slots && nr - full >= slots / 8) l = full;
                }
                for (int64_t r = f; r < l; r++) ord[(size_t)r] = (int32_t)(r - f);
                std::stable_sort(ord.begin() + f, ord.begin() + l, [&](int32_t a, int32_t b) { return j->h_read_len[(size_t)(f + a)] > j->h_read_len[(size_t)(f + b)]; });
                {   // mask streams in the same order, interleaved block by block within every group of 64 (= a wave of the bit-sliced kernel, give or take the slots
                    // that go to k_sw): block b of the group's x-th stream starts at record (b * 64 + x) * 64 of the group's region -- what a wave writes
                    // during 64 steps lies within 64 KB instead of in 64 places half a megabyte apart (address translation was a third of that kernel's time)
                    int64_t acc = 0, region = 0;
                    const bool contig = getenv("FZP_TB_CONTIG") != nullptr;      // comparison switch: every stream on its own, in launch order (stride 64)
                    if (j->h_tbm_total.size() < (size_t)nr + 1) j->h_tbm_total.assign((size_t)nr + 1, 0);
                    for (int64_t g0 = 0; g0 < l - f; g0 += 64) {
                        const int64_t gn = std::min<int64_t>(64, l - f - g0);
                        int64_t cap_max = 0;
                        for (int64_t x = g0; x < g0 + gn; x++) { const int64_t r = f + ord[(size_t)(f + x)]; cap_max = std::max(cap_max, j->h_tb_off[(size_t)r + 1] - j->h_tb_off[(size_t)r]); }
                        for (int64_t x = g0; x < g0 + gn; x++) {
                            const int64_t r = f + ord[(size_t)(f + x)];
                            h_tbm[(size_t)r] = contig ? acc : region + (x - g0) * 64; h_mvm[(size_t)r] = (acc >> 6) + x;
                            acc += j->h_tb_off[(size_t)r + 1] - j->h_tb_off[(size_t)r];
                        }
                        region += 64 * cap_max;
                    }
                    j->h_tbm_total[(size_t)f] = contig ? acc : region;
                }
                // walkers of the segmented trace-back: one per TBS_SEG steps of every read's step capacity, longest reads first
                j->h_seg_base[(size_t)f] = (int64_t)sgs.size();
                std::vector<int32_t> w_first((size_t)(l - f));
                int64_t nw_chunk = 0;
                const int64_t single_steps = getenv("FZP_TB_SINGLE_STEPS") ? atol(getenv("FZP_TB_SINGLE_STEPS")) : TBS_SINGLE_STEPS;
                auto n_walkers = [&](int64_t r) { const int64_t cap = j->h_tb_off[(size_t)r + 1] - j->h_tb_off[(size_t)r]; return cap <= single_steps ? (int64_t)1 : (cap + TBS_SEG - 1) / TBS_SEG; };
                for (int64_t r = f; r < l; r++) { sgo[(size_t)r] = (int32_t)nw_chunk; nw_chunk += n_walkers(r); }
                if (nw_chunk >= (1ll << 31)) { fzp_set_error("fzp_align_run: too many trace-back segments in one chunk"); return FZP_EINVAL; }
                sgs.resize(sgs.size() + (size_t)nw_chunk); sgi.resize(sgs.size());
                for (int64_t r = f; r < l; r++) {
                    const int64_t ns = n_walkers(r);
                    const bool one = j->h_tb_off[(size_t)r + 1] - j->h_tb_off[(size_t)r] <= single_steps;
                    sg1[(size_t)r] = one ? 1 : 0;
                    for (int64_t x = 0; x < ns; x++) { sgs[(size_t)(j->h_seg_base[(size_t)f] + sgo[(size_t)r] + x)] = (int32_t)(r - f); sgi[(size_t)(j->h_seg_base[(size_t)f] + sgo[(size_t)r] + x)] = one ? -1 : (int32_t)x; }
                }
                j->h_seg_cnt[(size_t)f] = nw_chunk;
                {   // launch order of the chunk's walkers: whole-read walkers by decreasing read length, then the segment walkers
                    sgw.resize(sgs.size());
                    int32_t *wo = sgw.data() + j->h_seg_base[(size_t)f];
                    int64_t at = 0;
                    for (int64_t x = 0; x < l - f; x++) { const int64_t r = f + ord[(size_t)(f + x)]; if (sg1[(size_t)r]) wo[at++] = sgo[(size_t)r]; }
                    for (int64_t r = f; r < l; r++) if (!sg1[(size_t)r]) for (int64_t x = 0; x < n_walkers(r); x++) wo[at++] = (int32_t)(sgo[(size_t)r] + x);
                }
                f = l;
            }
            FZP_TRY(j->lpt.upload(ord.data(), (size_t)nr, st));
            j->h_lpt = ord;
            FZP_TRY(j->tbm_off.upload(h_tbm.data(), (size_t)nr, st)); FZP_TRY(j->mvm_off.upload(h_mvm.data(), (size_t)nr, st));
            FZP_HIP(hipStreamSynchronize(st));      // (staging vectors)
            FZP_TRY(j->seg_off.upload(sgo.data(), (size_t)nr, st));
            FZP_TRY(j->seg_single.upload(sg1.data(), (size_t)nr, st));
            FZP_TRY(j->seg_order.upload(sgw.data(), sgw.size(), st));
            FZP_TRY(j->seg_slot.upload(sgs.data(), sgs.size(), st)); FZP_TRY(j->seg_idx.upload(sgi.data(), sgi.size(), st));
            j->lpt_chunk_steps = split_rounds ? -1 : chunk_steps;
        }
        const bool use_lpt = getenv("FZP_SW_INPUT_ORDER") == nullptr;      // FZP_SW_INPUT_ORDER=1: the r2 launch order, for comparisons
        const bool use_prio = getenv("FZP_SW_NO_PRIO") == nullptr;
        int guess_lane = -1;                                                    // -1: the lane k_sw recorded (best H of the segment's top step); tests push it to the band's edge to exercise the fallback
        if (const char *e = getenv("FZP_TB_GUESS_LANE")) { const int g = atoi(e); if (g >= 0 && g < 64) guess_lane = g; }
        int ov_limit = TBS_OV, REPAIR_ROUNDS = 3;                               // test switches: a shorter search for the common cell (forces repair walks), fewer repair rounds (forces the serial walk)
        if (const char *e = getenv("FZP_TB_OV_LIMIT")) { const int g = atoi(e); if (g >= 1 && g <= TBS_OV) ov_limit = g; }
        if (const char *e = getenv("FZP_TB_REPAIR_ROUNDS")) { const int g = atoi(e); if (g >= 0 && g <= 8) REPAIR_ROUNDS = g; }
        const bool tb_serial = getenv("FZP_TB_SERIAL") != nullptr;             // FZP_TB_SERIAL=1: the r2 trace-back (one walker per read), for comparisons
        const bool no_masks = getenv("FZP_SW_NO_MASKS") != nullptr;          // MEASUREMENT ONLY: the DP without its trace-back stores; the alignments that follow are garbage
        int64_t sum_len = 0;
        for (int64_t r = 0; r < nr; r++) sum_len += j->h_read_len[(size_t)r];
        const int32_t mean_len = (int32_t)std::max<int64_t>(1, sum_len / std::max<int64_t>(nr, 1));
        // ---- which DP kernel runs which extension (fzalign scores 2 / -4 / -3 are built into the bit-sliced one's cell function)
        bool use_bits = getenv("FZP_SW_NO_BITS") == nullptr && P.match == 2 && P.mismatch == 4 && P.gap == 3 && !split_rounds && use_lpt;
        // the bit-sliced kernel has two forms.  A pair of lanes per read (k_swb2) has the shorter step (112 against 135 instructions on the wave's critical path) but twice the
        // waves, and its instruction mix (v_bitop3, DPP, 3-operand forms) issues at ~4.5 cycles per SIMD however many waves share it: two such waves on one SIMD run at half
        // speed each.  So it is taken when its waves get a SIMD each and nothing else runs beside them; else the whole band sits in one lane (k_swb).  FZP_SWB_64 / FZP_SWB_PAIR force one.
        const int swb_force = getenv("FZP_SWB_64") ? 64 : (getenv("FZP_SWB_PAIR") ? 32 : 0);
        const int32_t m_stride = getenv("FZP_TB_CONTIG") ? 64 : 64 * 64;
        const bool swb_ring = getenv("FZP_SWB_NO_RING") == nullptr;        // k_swb's base streams through LDS rings (the backward extensions are too short to gain: straight from HBM)
        int64_t swb_max_steps = 40960;          // ~ 18 kb reads: a lane's step costs ~330 ns, the chain of a longer extension would outlast the rest of the launch
        if (const char *e = getenv("FZP_SWB_MAX_STEPS")) { const long g = atol(e); if (g > 0) swb_max_steps = g; }
        std::vector<int64_t> &swb_at = j->h_swb_at, &sw_at = j->h_sw_at;
        if (use_bits) {
            j->h_swb_list.clear(); j->h_sw_list.clear(); swb_at.assign(1, 0); sw_at.assign(1, 0);
            const int32_t *ordp = j->h_lpt.data();              // the cached longest-first order of every chunk
            for (int64_t f = 0; f < nr;) {
                int64_t l = f;
                while (l < nr && j->h_tb_off[(size_t)l + 1] - j->h_tb_off[(size_t)f] <= chunk_steps) l++;
                if (l == f) l = f + 1;
                const bool swb_in_order = getenv("FZP_SWB_INPUT_ORDER") != nullptr;
                for (int64_t x = 0; x < l - f; x++) {
                    const int32_t w = swb_in_order ? (int32_t)x : ordp[(size_t)(f + x)];
                    const int64_t r = f + w;
                    const Anchor &a = h_anc[(size_t)r];
                    const int64_t nq = j->h_read_len[(size_t)r] - a.i_a, nt = std::min<int64_t>(j->h_ctg_len[(size_t)j->h_read_ctg[(size_t)r]] - a.c_a, nq + nq / 4 + 64);
                    if (a.aligned && nq >= 64 && nt >= 64 && nq + nt + 2 <= swb_max_steps) j->h_swb_list.push_back(w); else j->h_sw_list.push_back(w);
                }
                while (j->h_swb_list.size() % 64) j->h_swb_list.push_back(-1);
                swb_at.push_back((int64_t)j->h_swb_list.size()); sw_at.push_back((int64_t)j->h_sw_list.size());
                f = l;
            }
            if (j->h_swb_list.empty()) j->h_swb_list.push_back(-1);
            if (j->h_sw_list.empty()) j->h_sw_list.push_back(0);
            FZP_TRY(j->swb_list.upload(j->h_swb_list.data(), j->h_swb_list.size(), st));
            FZP_TRY(j->sw_list.upload(j->h_sw_list.data(), j->h_sw_list.size(), st));
        }
        int64_t first = 0;
        int k = 0;
        int ci = -1;
        bool used[2] = {false, false};
        size_t w_lo = 0;
        while (first < nr) {
            int64_t last = first;
            while (last < nr && j->h_tb_off[(size_t)last + 1] - j->h_tb_off[(size_t)first] <= chunk_steps) last++;
            if (last == first) last = first + 1;
            ci++;
            // whole rounds first: k_sw runs one wave per read on n_CU x 32 wave slots, and reads of similar length finish round by round.
            // Cutting the launch after the last FULL round lets the trace-back of those reads (HBM / latency bound, few waves) run under the
            // DP of the remainder, which leaves slots free anyway.
            if (split_rounds && first == 0 && last == nr) {
                const int64_t slots = (int64_t)ctx->n_cu * 32;
                const int64_t full = nr / slots * slots;
                if (full >= slots && nr - full >= slots / 8) last = full;
            }
            const int64_t cnt = last - first;
            const int64_t steps = j->h_tb_off[(size_t)last] - j->h_tb_off[(size_t)first];
            const int bi = k & 1;
            if (used[bi]) FZP_HIP(hipStreamWaitEvent(st, j->ev_tb[bi], 0));   // buffer free again?
            // the chunk's second candidates sit behind the first ones in the same mask / move-word buffers
            size_t w_hi = w_lo;
            while (w_hi < h_ridx.size() && h_ridx[w_hi] < last) w_hi++;
            const int64_t c2 = (int64_t)(w_hi - w_lo), steps2 = h_tb_off2[w_hi] - h_tb_off2[w_lo];
            const int64_t tsteps = use_bits ? j->h_tbm_total[(size_t)first] : steps;      // records the chunk's first-candidate masks span (planned streams have some slack)
            const int64_t tb_base = tsteps + 64, mv_base = steps / 64 + cnt + 2;
            FZP_TRY(j->tbs.alloc((size_t)nr));
            FZP_TRY(j->tb2[bi].alloc((size_t)(tb_base + steps2) * 2 + 128));
            FZP_TRY(j->mvw2[bi].alloc((size_t)(mv_base + steps2 / 64 + c2 + 2)));
            FZP_TRY(j->raw2[bi].alloc((size_t)(steps / 16 + 64)));
            const bool dp_chain = getenv("FZP_DP_NO_CHAIN") == nullptr;
            std::unique_lock<std::mutex> dp_lk(g_dp_mu, std::defer_lock);
            if (dp_chain) {
                dp_lk.lock();
                auto it = g_dp_last.find(ctx->device);
                if (it != g_dp_last.end()) FZP_HIP(hipStreamWaitEvent(st, it->second, 0));
            }
            {
                ProfScope ps(ctx, "k1_sw");
                if (use_bits) {
                    // every extension goes to one of the two DP kernels: the bit-sliced one (a read per lane) takes those that fit the band on both sides and
                    // are short enough for its per-step latency; the rest -- the long reads first of all -- run a wave each, started before it
                    const int64_t b_at = swb_at[(size_t)ci], b_n = swb_at[(size_t)ci + 1] - b_at, w_at = sw_at[(size_t)ci], w_n = sw_at[(size_t)ci + 1] - w_at;
                    const bool swb64 = swb_force ? swb_force == 64 : !(w_n == 0 && b_n / 32 <= (int64_t)ctx->n_cu * 4);
                    // the bit-sliced kernel first (k_wait_started: which of the two gets onto the chip first decides how the pair runs), the wave-per-read kernel on its
                    // own stream as soon as that one's first workgroup runs: the two share the chip (latency-bound waves of long reads there, one wave per SIMD here)
                    FZP_TRY(j->dp_flag.alloc(1));
                    if (!j->dp_seq) FZP_TRY(j->dp_flag.zero(1, st));
                    uint32_t *flag_p = (w_n > 0 && b_n > 0) ? j->dp_flag.p : (uint32_t *)nullptr;
                    const uint32_t flag_v = ++j->dp_seq;
                    if (w_n > 0) FZP_HIP(hipEventRecord(j->ev_l[0], st));
                    if (b_n > 0 && !swb64)
                        hipLaunchKernelGGL(k_swb2, dim3((unsigned)((b_n + 127) / 128)), dim3(256), 0, st, first, b_n, (const int32_t *)(j->swb_list.p + b_at), (const int32_t *)nullptr, j->read_ori.p, j->read_woff.p,
                                           j->read_len.p, j->read_ctg.p, j->ctg_pk.p, j->ctg_woff.p, j->ctg_len.p, j->anc.p, j->tb_off.p, j->tb2[bi].p, j->mvw2[bi].p, j->info.p, j->tbo.p, j->mvo.p,
                                           (const int64_t *)j->tbm_off.p, (const int64_t *)j->mvm_off.p, (uint8_t *)nullptr, 0x7fffffff, m_stride, j->tbs.p, flag_p, flag_v);
                    if (b_n > 0 && swb64)
                        hipLaunchKernelGGL(swb_ring ? k_swb<true> : k_swb<false>, dim3((unsigned)((b_n + 64 * SWB_WPG - 1) / (64 * SWB_WPG))), dim3(64 * SWB_WPG), 0, st, first, b_n, (const int32_t *)(j->swb_list.p + b_at), (const int32_t *)nullptr, j->read_ori.p, j->read_woff.p,
                                           j->read_len.p, j->read_ctg.p, j->ctg_pk.p, j->ctg_woff.p, j->ctg_len.p, j->anc.p, j->tb_off.p, j->tb2[bi].p, j->mvw2[bi].p, j->info.p, j->tbo.p, j->mvo.p, (const int64_t *)j->tbm_off.p, (const int64_t *)j->mvm_off.p,
                                           (uint8_t *)nullptr, 0x7fffffff, getenv("FZP_SWB_DBG") ? atoi(getenv("FZP_SWB_DBG")) : 0, m_stride, j->tbs.p, flag_p, flag_v);
                    if (w_n > 0) {
                        FZP_HIP(hipStreamWaitEvent(ctx->stream3, j->ev_l[0], 0));
                        if (flag_p) hipLaunchKernelGGL(k_wait_started, dim3(1), dim3(1), 0, ctx->stream3, (const uint32_t *)flag_p, flag_v);
                        hipLaunchKernelGGL(no_masks ? k_sw<false> : k_sw<true>, dim3((unsigned)w_n), dim3(64), 0, ctx->stream3, first, w_n, (const int32_t *)nullptr, j->read_ori.p, j->read_woff.p, j->read_len.p, j->read_ctg.p,
                                           j->ctg_pk.p, j->ctg_woff.p, j->ctg_len.p, j->anc.p, j->tb_off.p, j->tb2[bi].p, j->mvw2[bi].p, P.match, P.mismatch, P.gap, j->info.p, j->tbo.p, j->mvo.p,
                                           (const int32_t *)(j->sw_list.p + w_at), use_prio ? mean_len : 0, (const int64_t *)j->tbm_off.p, (const int64_t *)j->mvm_off.p, (const uint8_t *)nullptr, m_stride, j->tbs.p);
                        FZP_HIP(hipEventRecord(j->ev_l[1], ctx->stream3));
                    }
                    if (w_n > 0) FZP_HIP(hipStreamWaitEvent(st, j->ev_l[1], 0));
                } else
                hipLaunchKernelGGL(no_masks ? k_sw<false> : k_sw<true>, dim3((unsigned)cnt), dim3(64), 0, st, first, cnt, (const int32_t *)nullptr, j->read_ori.p, j->read_woff.p, j->read_len.p, j->read_ctg.p,
                                   j->ctg_pk.p, j->ctg_woff.p, j->ctg_len.p, j->anc.p, j->tb_off.p, j->tb2[bi].p, j->mvw2[bi].p, P.match, P.mismatch, P.gap, j->info.p, j->tbo.p, j->mvo.p,
                                   use_lpt ? (const int32_t *)(j->lpt.p + first) : (const int32_t *)nullptr, use_prio ? mean_len : 0, (const int64_t *)nullptr, (const int64_t *)nullptr, (const uint8_t *)nullptr, 64, j->tbs.p);
            }
            if (c2 > 0) {     // same kernel over the compacted list, then the better extension of each read survives
                {
                    ProfScope ps(ctx, "k1_sw2");
                    hipLaunchKernelGGL(k_sw<true>, dim3((unsigned)c2), dim3(64), 0, st, (int64_t)w_lo, c2, j->ridx.p, j->sec_ori.p, j->sec_woff.p, j->read_len.p, j->read_ctg.p,
                                       j->ctg_pk.p, j->ctg_woff.p, j->ctg_len.p, j->anc2.p, j->tb_off2.p, j->tb2[bi].p + 2 * tb_base, j->mvw2[bi].p + mv_base, P.match, P.mismatch,
                                       P.gap, j->info2.p, (int64_t *)nullptr, (int64_t *)nullptr, (const int32_t *)nullptr, 0, (const int64_t *)nullptr, (const int64_t *)nullptr, (const uint8_t *)nullptr, 64, (int32_t *)nullptr);
                }
                ProfScope ps(ctx, "k1_pick");
                hipLaunchKernelGGL(k_pick, dim3((unsigned)((c2 + 255) / 256)), dim3(256), 0, st, (int64_t)w_lo, (int64_t)w_hi, j->ridx.p, j->anc2.p, j->info2.p, j->tb_off2.p,
                                   tb_base, mv_base, j->anc.p, j->info.p, j->tbo.p, j->mvo.p, j->won.p, j->tbs.p);
                hipLaunchKernelGGL(k_pick_copy, dim3((unsigned)c2), dim3(256), 0, st, (int64_t)w_lo, j->ridx.p, j->won.p, j->read_len.p, j->sec_ori.p, j->sec_woff.p, j->read_woff.p, j->read_ori.p);
            }
            w_lo = w_hi;
            FZP_HIP(hipEventRecord(j->ev_sw[bi], st));        // the forward extensions are final: their walk starts on the second stream while the backward ones run here
            if (dp_chain) {
                hipEvent_t &e = g_dp_last[ctx->device];
                if (!e) FZP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                FZP_HIP(hipEventRecord(e, st));
                dp_lk.unlock();
            }
            const int64_t steps_b = j->h_tb_off_b[(size_t)last] - j->h_tb_off_b[(size_t)first];
            FZP_TRY(j->tb_b2[bi].alloc((size_t)(steps_b + 64) * 2 + 128));
            FZP_TRY(j->mvw_b2[bi].alloc((size_t)(steps_b / 64 + cnt + 2)));
            FZP_TRY(j->raw_b2[bi].alloc((size_t)(steps_b / 16 + 64)));
            {   // backward from the winner's anchor: reversed inputs, then the same DP kernel (slot = read; its "contig" is its own window)
                ProfScope ps(ctx, "k1_back");
                hipLaunchKernelGGL(k_back_prep, dim3((unsigned)cnt), dim3(256), 0, st, first, j->anc.p, j->read_ctg.p, j->read_ori.p, j->read_woff.p, j->ctg_pk.p, j->ctg_woff.p,
                                   j->bq_off.p, j->bt_off.p, j->bq.p, j->bt.p, j->anc_b.p, j->b_len.p, j->b_tlen.p);
                // the backward extensions are short (an anchor sits a few hundred bases into its read): the bit-sliced kernel takes every one that spans the band
                // (lanes in slot order: their mask streams lie side by side), k_sw the rest -- which of the two is decided on the device, the winner's anchor never came to the host
                const bool swb64 = swb_force ? swb_force == 64 : !((cnt + 31) / 32 <= (int64_t)ctx->n_cu * 4);
                if (use_bits && !swb64) {
                    FZP_TRY(j->b_handled.alloc((size_t)nr));
                    hipLaunchKernelGGL(k_swb2, dim3((unsigned)((cnt + 127) / 128)), dim3(256), 0, st, first, cnt, (const int32_t *)j->b_iota.p, (const int32_t *)nullptr, j->bq.p, j->bq_off.p, j->b_len.p, j->b_iota.p,
                                       j->bt.p, j->bt_off.p, j->b_tlen.p, j->anc_b.p, j->tb_off_b.p, j->tb_b2[bi].p, j->mvw_b2[bi].p, j->info_b.p, j->tbo_b.p, j->mvo_b.p,
                                       (const int64_t *)nullptr, (const int64_t *)nullptr, j->b_handled.p, (int32_t)swb_max_steps, 64, (int32_t *)nullptr, (uint32_t *)nullptr, 0u);
                }
                if (use_bits && swb64) {
                    FZP_TRY(j->b_handled.alloc((size_t)nr));
                    hipLaunchKernelGGL(k_swb<false>, dim3((unsigned)((cnt + 64 * SWB_WPG - 1) / (64 * SWB_WPG))), dim3(64 * SWB_WPG), 0, st, first, cnt, (const int32_t *)j->b_iota.p, (const int32_t *)nullptr, j->bq.p, j->bq_off.p, j->b_len.p, j->b_iota.p,
                                       j->bt.p, j->bt_off.p, j->b_tlen.p, j->anc_b.p, j->tb_off_b.p, j->tb_b2[bi].p, j->mvw_b2[bi].p, j->info_b.p, j->tbo_b.p, j->mvo_b.p,
                                       (const int64_t *)nullptr, (const int64_t *)nullptr, j->b_handled.p, (int32_t)swb_max_steps, 0, 64, (int32_t *)nullptr, (uint32_t *)nullptr, 0u);
                }
                hipLaunchKernelGGL(k_sw<true>, dim3((unsigned)cnt), dim3(64), 0, st, first, cnt, (const int32_t *)nullptr, j->bq.p, j->bq_off.p, j->b_len.p, j->b_iota.p,
                                   j->bt.p, j->bt_off.p, j->b_tlen.p, j->anc_b.p, j->tb_off_b.p, j->tb_b2[bi].p, j->mvw_b2[bi].p, P.match, P.mismatch, P.gap, j->info_b.p,
                                   j->tbo_b.p, j->mvo_b.p, (const int32_t *)nullptr, 0, (const int64_t *)nullptr, (const int64_t *)nullptr, use_bits ? (const uint8_t *)j->b_handled.p : (const uint8_t *)nullptr, 64, (int32_t *)nullptr);
            }
            FZP_HIP(hipEventRecord(j->ev_bk[bi], st));
            FZP_HIP(hipStreamWaitEvent(st2, j->ev_sw[bi], 0));
            if (tb_serial) {
                ProfScope ps(ctx, "k1_traceback", st2);
                hipLaunchKernelGGL(k_tb_walk<false>, dim3((unsigned)((cnt + TBW_RPW * TBW_WPG - 1) / (TBW_RPW * TBW_WPG))), dim3(64 * TBW_WPG), 0, st2, first, cnt, j->anc.p, j->info.p, j->tb_off.p,
                                   j->tbo.p, j->mvo.p, (const ulonglong2 *)j->tb2[bi].p, (const ulonglong2 *)j->mvw2[bi].p, j->raw2[bi].p, j->wout.p,
                                   use_lpt ? (const int32_t *)(j->lpt.p + first) : (const int32_t *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr,
                                   (uint32_t *)nullptr, (SegOut *)nullptr, 0, 32, (const SegReq *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const int32_t *)j->tbs.p);
            } else {
                const int64_t nwk = j->h_seg_cnt[(size_t)first], wbase = j->h_seg_base[(size_t)first];
                FZP_TRY(j->raw_seg2[bi].alloc((size_t)nwk * TBS_RAW_WORDS + 64));
                FZP_TRY(j->trail2[bi].alloc((size_t)nwk * 2 * TBS_OV + 64));
                FZP_TRY(j->segout2[bi].alloc((size_t)nwk + 1));
                ProfScope ps(ctx, "k1_traceback", st2);
                FZP_HIP(hipMemsetAsync(j->trail2[bi].p, 0xff, (size_t)nwk * 2 * TBS_OV * 4, st2));
                hipLaunchKernelGGL(k_tb_walk<true>, dim3((unsigned)((nwk + TBW_RPW * TBW_WPG - 1) / (TBW_RPW * TBW_WPG))), dim3(64 * TBW_WPG), 0, st2, first, nwk, j->anc.p, j->info.p, j->tb_off.p,
                                   j->tbo.p, j->mvo.p, (const ulonglong2 *)j->tb2[bi].p, (const ulonglong2 *)j->mvw2[bi].p, j->raw_seg2[bi].p, j->wout.p,
                                   (const int32_t *)(j->seg_order.p + wbase), (const int32_t *)(j->seg_slot.p + wbase), (const int32_t *)(j->seg_idx.p + wbase), j->trail2[bi].p, j->segout2[bi].p, 0, guess_lane, (const SegReq *)nullptr, (const uint32_t *)nullptr, j->raw2[bi].p, (const int32_t *)j->tbs.p);
                const uint32_t req_cap = (uint32_t)std::min<int64_t>(nwk, 1 << 20);
                FZP_TRY(j->seg_req.alloc((size_t)req_cap + 1));
                FZP_HIP(hipMemsetAsync(j->tb_fallback.p + 1, 0, 4, st2));      // this chunk's repair requests
                for (int round = 0; round <= REPAIR_ROUNDS; round++) {
                    hipLaunchKernelGGL(k_tb_stitch, dim3((unsigned)cnt), dim3(64), 0, st2, first, cnt, j->anc.p, j->info.p, j->tb_off.p, (const int32_t *)(j->seg_off.p + first),
                                       (const uint32_t *)j->trail2[bi].p, (const SegOut *)j->segout2[bi].p, (const uint32_t *)j->raw_seg2[bi].p, j->raw2[bi].p, j->wout.p, j->tb_fallback.p,
                                       j->seg_req.p, req_cap, (round == 0 ? 1 : 0) | (round < REPAIR_ROUNDS ? 2 : 0), (const uint8_t *)(j->seg_single.p + first), ov_limit);
                    if (round == REPAIR_ROUNDS) break;
                    // boundaries that did not join: their lower segments again, from the exact cell (a launch of empty waves when there are none)
                    hipLaunchKernelGGL(k_tb_walk<true>, dim3((unsigned)((std::min<int64_t>(req_cap, cnt) + TBW_RPW * TBW_WPG - 1) / (TBW_RPW * TBW_WPG))), dim3(64 * TBW_WPG), 0, st2, first, (int64_t)req_cap, j->anc.p, j->info.p,
                                       j->tb_off.p, j->tbo.p, j->mvo.p, (const ulonglong2 *)j->tb2[bi].p, (const ulonglong2 *)j->mvw2[bi].p, j->raw_seg2[bi].p, j->wout.p,
                                       (const int32_t *)nullptr, (const int32_t *)(j->seg_slot.p + wbase), (const int32_t *)(j->seg_idx.p + wbase), j->trail2[bi].p, j->segout2[bi].p, 0, guess_lane,
                                       (const SegReq *)j->seg_req.p, (const uint32_t *)(j->tb_fallback.p + 1), j->raw2[bi].p, (const int32_t *)j->tbs.p);
                    hipLaunchKernelGGL(k_tb_req_reset, dim3(1), dim3(64), 0, st2, j->tb_fallback.p);   // the walks are queued behind it: the next stitch pass counts from 0
                }
                // reads whose segments did not join (flagged by the stitching) are walked in one piece; every other wave of this launch leaves at once
                hipLaunchKernelGGL(k_tb_walk<false>, dim3((unsigned)((cnt + TBW_RPW * TBW_WPG - 1) / (TBW_RPW * TBW_WPG))), dim3(64 * TBW_WPG), 0, st2, first, cnt, j->anc.p, j->info.p, j->tb_off.p,
                                   j->tbo.p, j->mvo.p, (const ulonglong2 *)j->tb2[bi].p, (const ulonglong2 *)j->mvw2[bi].p, j->raw2[bi].p, j->wout.p,
                                   (const int32_t *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr, (uint32_t *)nullptr, (SegOut *)nullptr, 1, 32, (const SegReq *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const int32_t *)j->tbs.p);
            }
            FZP_HIP(hipStreamWaitEvent(st2, j->ev_bk[bi], 0));
            {   // the backward parts: walked (one walker each: they are short), then joined to the forward streams
                ProfScope ps(ctx, "k1_back_tb", st2);
                hipLaunchKernelGGL(k_tb_walk<false>, dim3((unsigned)((cnt + TBW_RPW * TBW_WPG - 1) / (TBW_RPW * TBW_WPG))), dim3(64 * TBW_WPG), 0, st2, first, cnt, j->anc_b.p, j->info_b.p, j->tb_off_b.p,
                                   j->tbo_b.p, j->mvo_b.p, (const ulonglong2 *)j->tb_b2[bi].p, (const ulonglong2 *)j->mvw_b2[bi].p, j->raw_b2[bi].p, j->wout_b.p,
                                   (const int32_t *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr, (uint32_t *)nullptr, (SegOut *)nullptr, 0, 32,
                                   (const SegReq *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const int32_t *)nullptr);
                hipLaunchKernelGGL(k_back_merge, dim3((unsigned)cnt), dim3(64), 0, st2, first, cnt, j->tb_off.p, j->tb_off_b.p, j->anc_b.p, j->info_b.p, j->wout_b.p,
                                   (const uint32_t *)j->raw_b2[bi].p, j->raw2[bi].p, j->wout.p, j->info.p);
            }
            {
                ProfScope ps(ctx, "k1_cigar", st2);
                hipLaunchKernelGGL(k_tb_cigar, dim3((unsigned)cnt), dim3(64), 0, st2, first, cnt, j->read_len.p, j->anc.p, j->info.p, j->tb_off.p, j->raw2[bi].p,
                                   j->wout.p, j->cig_off.p, j->cig.p, j->cig_start.p, j->summ.p, P.match, P.mismatch, P.gap, P.min_pct_identity,
                                   j->read_ori.p, j->read_woff.p, j->read_ctg.p, j->ctg_pk.p, j->ctg_woff.p);
            }
            FZP_HIP(hipEventRecord(j->ev_tb[bi], st2));
            used[bi] = true;
            first = last;
            k++;
        }
        for (int b2 = 0; b2 < 2; b2++)
            if (used[b2]) FZP_HIP(hipStreamWaitEvent(st, j->ev_tb[b2], 0));   // the main stream continues after all trace-backs
    }
    FZP_HIP(hipStreamSynchronize(st));
    FZP_HIP(hipGetLastError());
    j->summ_on_host = false;      // the batch path plans on the device; the summaries come to the host when someone asks
    j->done = true;
    return FZP_OK;
}

namespace {
int fetch_summaries(fzp_ctx *ctx, fzp_alnjob *j) {
    if (j->summ_on_host) return FZP_OK;
    FZP_TRY(fzp_bind(ctx));
    j->h_summ.resize((size_t)j->n_reads);
    FZP_TRY(j->summ.download(j->h_summ.data(), (size_t)j->n_reads, ctx->stream));
    FZP_HIP(hipStreamSynchronize(ctx->stream));
    j->summ_on_host = true;
    return FZP_OK;
}
}  // namespace

extern "C" int64_t fzp_align_n_second(const fzp_alnjob *j) { return j ? j->n_second : 0; }
extern "C" int fzp_align_tb_fallbacks(fzp_ctx *ctx, fzp_alnjob *j, int64_t *n) {
    if (!ctx || !j || !j->done || !n) { fzp_set_error("fzp_align_tb_fallbacks: run the job first"); return FZP_EINVAL; }
    FZP_TRY(fzp_bind(ctx));
    uint32_t v[4] = {0, 0, 0, 0};
    if (j->tb_fallback.p) { FZP_HIP(hipMemcpyAsync(v, j->tb_fallback.p, 16, hipMemcpyDeviceToHost, ctx->stream)); FZP_HIP(hipStreamSynchronize(ctx->stream)); }
    if (getenv("FZP_TB_DEBUG")) fprintf(stderr, "[fzp_align_tb_fallbacks] counters %u %u %u %u\n", v[0], v[1], v[2], v[3]);
    n[0] = (int64_t)v[0];
    n[1] = (int64_t)v[2];
    return FZP_OK;
}

extern "C" int fzp_align_summaries(fzp_ctx *ctx, fzp_alnjob *j, fzp_aln_summary *out) {
    if (!ctx || !j || !j->done || !out) { fzp_set_error("fzp_align_summaries: run the job first"); return FZP_EINVAL; }
    FZP_TRY(fetch_summaries(ctx, j));
    if (j->n_reads) memcpy(out, j->h_summ.data(), (size_t)j->n_reads * sizeof(fzp_aln_summary));
    return FZP_OK;
}

namespace {
struct RecPlan {
    std::vector<int64_t> rec_read;               // accepted records, (contig, POS, read index) order
    std::vector<int32_t> rec_qid, rec_pos, rec_ctg;
    std::vector<int64_t> cig_off, seq_off, rec_begin;   // rec_begin per contig
    std::vector<std::vector<int64_t>> ctg_reads;        // all aligned reads per contig, q_id order
    std::vector<int32_t> last_pos, max_span;
    std::vector<int64_t> n_columns;
};

// what `samtools sort` + make_het_call's record filters (phasing.py:47-75) do to the aligner's output
void plan_records(const fzp_alnjob *j, int c_lo, int c_hi, RecPlan &p, bool apply_filters = true) {
    const int nc = c_hi - c_lo;
    // bucket the aligned reads by contig (counting sort), then order every bucket by (POS, read index) through one
    // 64-bit key per read
    std::vector<int64_t> cnt((size_t)nc + 1, 0);
    for (int64_t r = 0; r < j->n_reads; r++) {
        const int c = j->h_read_ctg[(size_t)r];
        if (c >= c_lo && c < c_hi && j->h_summ[(size_t)r].aligned) cnt[(size_t)(c - c_lo) + 1]++;
    }
    for (int c = 0; c < nc; c++) cnt[(size_t)c + 1] += cnt[(size_t)c];
    const int64_t n_al = cnt[(size_t)nc];
    std::vector<uint64_t> keys((size_t)n_al);
    {
        std::vector<int64_t> fill(cnt.begin(), cnt.end() - 1);
        for (int64_t r = 0; r < j->n_reads; r++) {
            const int c = j->h_read_ctg[(size_t)r];
            if (c >= c_lo && c < c_hi && j->h_summ[(size_t)r].aligned)
                keys[(size_t)fill[(size_t)(c - c_lo)]++] = ((uint64_t)(uint32_t)j->h_summ[(size_t)r].pos << 32) | (uint32_t)r;   // POS >= 0
        }
    }
    p.ctg_reads.assign(nc, {});
    p.cig_off.assign(1, 0); p.seq_off.assign(1, 0); p.rec_begin.assign(1, 0);
    p.last_pos.assign(nc, -1); p.max_span.assign(nc, 0); p.n_columns.assign(nc, 0);
    p.rec_read.reserve((size_t)n_al); p.rec_qid.reserve((size_t)n_al); p.rec_pos.reserve((size_t)n_al); p.rec_ctg.reserve((size_t)n_al);
    p.cig_off.reserve((size_t)n_al + 1); p.seq_off.reserve((size_t)n_al + 1);
    for (int c = 0; c < nc; c++) {
        uint64_t *k0 = keys.data() + cnt[(size_t)c], *k1 = keys.data() + cnt[(size_t)c + 1];
        std::sort(k0, k1);
        auto &v = p.ctg_reads[c];
        v.resize((size_t)(k1 - k0));
        for (size_t q = 0; q < v.size(); q++) {
            const int64_t r = (int64_t)(uint32_t)k0[q];
            v[q] = r;
            const fzp_aln_summary &s = j->h_summ[(size_t)r];
            const int64_t n = j->h_read_len[(size_t)r];
            const int64_t n_del = (int64_t)(s.ref_end - s.pos) - s.n_columns;
            const int64_t total_aln_pos = n + n_del;                       // sum of all CIGAR op lengths
            const int64_t skip_base = (int64_t)s.q_start + (n - s.q_end);   // soft clips
            if (apply_filters && 1.0 - 1.0 * (double)skip_base / (double)total_aln_pos < 0.1) continue;   // phasing.py:72
            if (apply_filters && total_aln_pos < 2000) continue;                                          // phasing.py:74
            p.rec_read.push_back(r);
            p.rec_qid.push_back((int32_t)q);
            p.rec_pos.push_back(s.pos);
            p.rec_ctg.push_back(c);
            p.cig_off.push_back(p.cig_off.back() + s.n_cigar);
            p.seq_off.push_back(p.seq_off.back() + n);
            p.last_pos[c] = s.pos;
            p.max_span[c] = std::max(p.max_span[c], s.ref_end - s.pos);
            p.n_columns[c] += s.n_columns;
        }
        p.rec_begin.push_back((int64_t)p.rec_read.size());
    }
}

int gather_records(fzp_ctx *ctx, fzp_alnjob *j, const RecPlan &p, DevBuf<uint32_t> &cigar, DevBuf<uint8_t> &seq, DevBuf<int64_t> &d_cig_off, DevBuf<int64_t> &d_seq_off) {
    hipStream_t st = ctx->stream;
    const int64_t nrec = (int64_t)p.rec_read.size();
    DevBuf<int64_t> d_rec_read;
    FZP_TRY(d_rec_read.upload(p.rec_read.data(), (size_t)nrec, st));
    FZP_TRY(d_cig_off.upload(p.cig_off.data(), p.cig_off.size(), st));
    FZP_TRY(d_seq_off.upload(p.seq_off.data(), p.seq_off.size(), st));
    FZP_TRY(cigar.alloc((size_t)p.cig_off.back()));
    FZP_TRY(seq.alloc((size_t)p.seq_off.back()));
    if (nrec > 0) {
        ProfScope ps(ctx, "k1_gather");
        hipLaunchKernelGGL(k_gather, dim3((unsigned)nrec, 4), dim3(256), 0, st, nrec, d_rec_read.p, j->cig_start.p, j->cig.p, d_cig_off.p, cigar.p, j->read_ori.p,
                           j->read_woff.p, d_seq_off.p, seq.p);
    }
    FZP_HIP(hipStreamSynchronize(st));
    FZP_HIP(hipGetLastError());
    return FZP_OK;
}
template <class T>
T *dupv(const std::vector<T> &v) {
    T *p = (T *)malloc((v.size() ? v.size() : 1) * sizeof(T));
    if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}
}  // namespace

static int align_alnset(fzp_ctx *ctx, fzp_alnjob *j, int32_t ctg, const int64_t *name_off, const char *names, fzp_alnset **out, int64_t **read_index, bool apply_filters,
                        std::vector<int32_t> *flags_out, std::shared_ptr<std::vector<uint8_t>> *ref_out);
extern "C" int fzp_align_alnset(fzp_ctx *ctx, fzp_alnjob *j, int32_t ctg, const int64_t *name_off, const char *names, fzp_alnset **out, int64_t **read_index) {
    return align_alnset(ctx, j, ctg, name_off, names, out, read_index, true, nullptr, nullptr);
}
extern "C" int fzp_align_alnset_all(fzp_ctx *ctx, fzp_alnjob *j, int32_t ctg, const int64_t *name_off, const char *names, fzp_alnset **out, int64_t **read_index) {
    return align_alnset(ctx, j, ctg, name_off, names, out, read_index, false, nullptr, nullptr);
}
// every aligned read's record with the device's 'M' CIGARs (no '=' / 'X' split yet), the strand flags of the records and a copy of the upper-cased
// contig: what fzp_pipe.hip hands to a writer thread, which splits (fzp_alnset_split_eqx) and compresses the BAM while the device goes on
int fzp_align_alnset_unsplit(fzp_ctx *ctx, fzp_alnjob *j, int32_t ctg, const int64_t *name_off, const char *names, fzp_alnset **out, std::vector<int32_t> *flags,
                             std::shared_ptr<std::vector<uint8_t>> *ref) {
    return align_alnset(ctx, j, ctg, name_off, names, out, nullptr, false, flags, ref);
}
// device CIGARs carry diagonal runs as 'M'; split them into '=' / 'X' against the contig (host only: any thread)
void fzp_alnset_split_eqx(fzp_alnset *a, const uint8_t *ref) {
    std::vector<uint32_t> out_c;
    std::vector<int64_t> out_off(1, 0);
    out_c.reserve((size_t)a->cig_off[a->n_rec] * 2);
    for (int64_t k = 0; k < a->n_rec; k++) {
        int64_t rp = a->rec_pos[k], qp = 0;
        const uint8_t *sq = a->seq + a->seq_off[k];
        for (int64_t w = a->cig_off[k]; w < a->cig_off[k + 1]; w++) {
            const uint32_t len = a->cigar[w] >> 4, op = a->cigar[w] & 15u;
            if (op != FZP_OP_M) {
                out_c.push_back(a->cigar[w]);
                if (op == FZP_OP_S || op == FZP_OP_I) qp += len; else if (op == FZP_OP_D) rp += len;
                continue;
            }
            uint32_t run = 0; int cur = -1;
            for (uint32_t x = 0; x < len; x++, rp++, qp++) {
                const int eq = code_of(sq[qp]) == code_of(ref[(size_t)rp]) ? FZP_OP_EQ : FZP_OP_X;
                if (eq == cur) run++;
                else { if (run) out_c.push_back((run << 4) | (uint32_t)cur); cur = eq; run = 1; }
            }
            if (run) out_c.push_back((run << 4) | (uint32_t)cur);
        }
        out_off.push_back((int64_t)out_c.size());
    }
    free(a->cigar); free(a->cig_off);
    a->cigar = dupv(out_c); a->cig_off = dupv(out_off);
}
static int align_alnset(fzp_ctx *ctx, fzp_alnjob *j, int32_t ctg, const int64_t *name_off, const char *names, fzp_alnset **out, int64_t **read_index, bool apply_filters,
                        std::vector<int32_t> *flags_out, std::shared_ptr<std::vector<uint8_t>> *ref_out) {
    if (!ctx || !j || !j->done || !out || ctg < 0 || ctg >= j->n_ctg) { fzp_set_error("fzp_align_alnset: bad arguments or job not run"); return FZP_EINVAL; }
    FZP_TRY(fzp_bind(ctx));
    FZP_TRY(fetch_summaries(ctx, j));
    RecPlan p;
    plan_records(j, ctg, ctg + 1, p, apply_filters);
    DevBuf<uint32_t> cigar;
    DevBuf<uint8_t> seq;
    DevBuf<int64_t> dco, dso;
    FZP_TRY(gather_records(ctx, j, p, cigar, seq, dco, dso));
    fzp_alnset *a = (fzp_alnset *)calloc(1, sizeof(fzp_alnset));
    if (!a) return FZP_ENOMEM;
    const int64_t nrec = (int64_t)p.rec_read.size();
    a->n_rec = nrec;
    a->rec_qid = dupv(p.rec_qid); a->rec_pos = dupv(p.rec_pos);
    a->cig_off = dupv(p.cig_off); a->seq_off = dupv(p.seq_off);
    a->cigar = (uint32_t *)malloc((size_t)std::max<int64_t>(p.cig_off.back(), 1) * 4);
    a->seq = (uint8_t *)malloc((size_t)std::max<int64_t>(p.seq_off.back(), 1));
    hipStream_t st = ctx->stream;
    int rc = cigar.download(a->cigar, (size_t)p.cig_off.back(), st);
    if (!rc) rc = seq.download(a->seq, (size_t)p.seq_off.back(), st);
    if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = FZP_EDEVICE;
    if (!rc) {
        std::vector<uint8_t> &ref = j->h_ctg[(size_t)ctg];
        if (ref.size() != (size_t)j->h_ctg_len[(size_t)ctg]) {              // first use: the upper-cased contig comes back from the device
            ref.resize((size_t)j->h_ctg_len[(size_t)ctg]);
            if (!ref.empty() && (hipMemcpyAsync(ref.data(), j->ctg_ascii.p + j->h_ctg_aoff[(size_t)ctg], ref.size(), hipMemcpyDeviceToHost, st) != hipSuccess ||
                                 hipStreamSynchronize(st) != hipSuccess)) rc = FZP_EDEVICE;
        }
        if (!rc && ref_out) {
            *ref_out = std::make_shared<std::vector<uint8_t>>(std::move(ref));   // the writer's copy is the only one kept
            ref.clear();
        } else if (!rc) fzp_alnset_split_eqx(a, ref.data());
        if (!rc && flags_out) {
            flags_out->resize((size_t)nrec);
            for (int64_t k = 0; k < nrec; k++) (*flags_out)[(size_t)k] = j->h_summ[(size_t)p.rec_read[(size_t)k]].strand ? 16 : 0;
        }
    }
    // q_id table: every aligned read of the contig, in sorted order (one SAM line each)
    const auto &ar = p.ctg_reads[0];
    a->n_qid = (int32_t)ar.size();
    std::vector<int64_t> qoff(1, 0);
    std::string qn;
    for (int64_t r : ar) {
        if (names && name_off) qn.append(names + name_off[r], (size_t)(name_off[r + 1] - name_off[r]));
        else { char t[40]; snprintf(t, sizeof t, "read/%lld", (long long)r); qn += t; }
        qoff.push_back((int64_t)qn.size());
    }
    a->qname_off = dupv(qoff);
    a->qnames = (char *)malloc(qn.size() + 1);
    memcpy(a->qnames, qn.c_str(), qn.size() + 1);
    a->last_pos = p.last_pos[0];
    a->max_ref_span = p.max_span[0];
    a->n_columns = p.n_columns[0];
    if (rc) { fzp_alnset_free(a); return rc; }
    if (read_index) *read_index = dupv(p.rec_read);
    *out = a;
    return FZP_OK;
}

extern "C" int fzp_align_to_batch(fzp_ctx *ctx, fzp_alnjob *j, fzp_batch **out) {
    if (!ctx || !j || !j->done || !out) { fzp_set_error("fzp_align_to_batch: job not run"); return FZP_EINVAL; }
    FZP_TRY(fzp_bind(ctx));
    *out = nullptr;
    hipStream_t st = ctx->stream;
    const int nc = j->n_ctg;
    const int64_t nr = j->n_reads;
    if (!j->have_slots) {      // reads grouped by contig, once per job
        std::vector<int64_t> off((size_t)nc + 1, 0);
        for (int64_t r = 0; r < nr; r++) off[(size_t)j->h_read_ctg[(size_t)r] + 1]++;
        for (int c = 0; c < nc; c++) { j->max_reads_per_ctg = std::max(j->max_reads_per_ctg, off[(size_t)c + 1]); off[(size_t)c + 1] += off[(size_t)c]; }
        if (nc > 65535) { fzp_set_error("fzp_align_to_batch: %d contigs in one job (limit 65535)", nc); return FZP_EINVAL; }
        j->h_slot_off = off;
        std::vector<int64_t> bko((size_t)nc + 1, 0);
        for (int c = 0; c < nc; c++) bko[(size_t)c + 1] = bko[(size_t)c] + (j->h_ctg_len[(size_t)c] >> RANK_SHIFT) + 2;
        j->n_rank_buckets = bko.back();
        FZP_TRY(j->rank_bk_off.upload(bko.data(), bko.size(), st));
        std::vector<int32_t> rd((size_t)nr), sc((size_t)nr);
        std::vector<int64_t> fill(off.begin(), off.end() - 1);
        for (int64_t r = 0; r < nr; r++) { const int c = j->h_read_ctg[(size_t)r]; const int64_t s_ = fill[(size_t)c]++; rd[(size_t)s_] = (int32_t)r; sc[(size_t)s_] = c; }
        FZP_TRY(j->slot_read.upload(rd.data(), (size_t)nr, st)); FZP_TRY(j->slot_ctg.upload(sc.data(), (size_t)nr, st)); FZP_TRY(j->slot_off.upload(off.data(), (size_t)nc + 1, st));
        FZP_HIP(hipStreamSynchronize(st));
        j->have_slots = true;
    }
    fzp_batch *b = new fzp_batch();
    struct Guard { fzp_batch *p; ~Guard() { delete p; } } guard{b};
    b->n_ctg = nc;
    // ---- plan on the device
    DevBuf<uint64_t> key, v_rec, v_cig, v_seq, v_ck, totals, rk_members, rk_total;
    DevBuf<uint32_t> rk_hist, rk_start, rk_fill, rk_rank;
    DevBuf<int32_t> &g_read = b->qid_read;      // stays with the batch: q_id q of contig c is read g_read[h_slot_off[c] + q]
    DevBuf<int32_t> g_qid, last_pos;
    DevBuf<uint8_t> g_acc;
    DevBuf<uint32_t> n_aligned;
    DevBuf<unsigned long long> n_cols;
    DevBuf<int64_t> rec_read;
    const size_t ns = (size_t)std::max<int64_t>(nr, 1);
    FZP_TRY(key.alloc(ns)); FZP_TRY(v_rec.alloc(ns)); FZP_TRY(v_cig.alloc(ns)); FZP_TRY(v_seq.alloc(ns)); FZP_TRY(v_ck.alloc(ns)); FZP_TRY(totals.alloc(4));
    FZP_TRY(g_read.alloc(ns)); FZP_TRY(g_qid.alloc(ns)); FZP_TRY(g_acc.alloc(ns)); FZP_TRY(last_pos.alloc((size_t)nc)); FZP_TRY(n_aligned.alloc((size_t)nc)); FZP_TRY(n_cols.alloc((size_t)nc));
    FZP_TRY(v_rec.zero(ns, st)); FZP_TRY(v_cig.zero(ns, st)); FZP_TRY(v_seq.zero(ns, st)); FZP_TRY(v_ck.zero(ns, st)); FZP_TRY(g_acc.zero(ns, st));
    FZP_TRY(n_aligned.zero((size_t)nc, st)); FZP_TRY(n_cols.zero((size_t)nc, st));
    FZP_HIP(hipMemsetAsync(last_pos.p, 0xff, (size_t)nc * 4, st));       // -1
    const unsigned gb = (unsigned)std::max<int64_t>(1, (nr + 255) / 256);
    if (nr > 0) {
        ProfScope ps(ctx, "k1_plan");
        hipLaunchKernelGGL(k_plan_keys, dim3(gb), dim3(256), 0, st, nr, j->slot_read.p, j->summ.p, key.p);
        FZP_TRY(rk_rank.alloc(ns));
        if (j->max_reads_per_ctg <= 8192) {
            hipLaunchKernelGGL(k_rank_allpairs, dim3((unsigned)((j->max_reads_per_ctg + 255) / 256), (unsigned)nc), dim3(256), 0, st, j->slot_off.p, key.p, rk_rank.p);
        } else {       // deep contigs: POS bins instead of all pairs
            const size_t nbk = (size_t)j->n_rank_buckets + 1;
            FZP_TRY(rk_hist.alloc(nbk)); FZP_TRY(rk_start.alloc(nbk)); FZP_TRY(rk_fill.alloc(nbk)); FZP_TRY(rk_members.alloc(ns)); FZP_TRY(rk_total.alloc(1));
            FZP_TRY(rk_hist.zero(nbk, st)); FZP_TRY(rk_fill.zero(nbk, st));
            hipLaunchKernelGGL(k_rank_hist, dim3(gb), dim3(256), 0, st, nr, j->slot_ctg.p, key.p, j->rank_bk_off.p, rk_hist.p);
            FZP_TRY(fzp_exclusive_scan_u32(ctx, rk_hist.p, rk_start.p, nbk, rk_total.p));
            hipLaunchKernelGGL(k_rank_scatter, dim3(gb), dim3(256), 0, st, nr, j->slot_ctg.p, key.p, j->rank_bk_off.p, rk_start.p, rk_fill.p, rk_members.p);
            hipLaunchKernelGGL(k_rank_binned, dim3(gb), dim3(256), 0, st, nr, j->slot_ctg.p, key.p, j->rank_bk_off.p, rk_start.p, rk_hist.p, rk_members.p, rk_rank.p);
        }
        hipLaunchKernelGGL(k_plan_rank, dim3(gb), dim3(256), 0, st, nr, j->slot_read.p, j->slot_ctg.p, j->slot_off.p, key.p, rk_rank.p, j->summ.p, j->read_len.p,
                           v_rec.p, v_cig.p, v_seq.p, v_ck.p, g_read.p, g_qid.p, g_acc.p, last_pos.p, n_aligned.p, n_cols.p);
    }
    FZP_TRY(fzp_exclusive_scan_u64_inplace(ctx, v_rec.p, (size_t)nr, totals.p + 0));
    FZP_TRY(fzp_exclusive_scan_u64_inplace(ctx, v_cig.p, (size_t)nr, totals.p + 1));
    FZP_TRY(fzp_exclusive_scan_u64_inplace(ctx, v_seq.p, (size_t)nr, totals.p + 2));
    FZP_TRY(fzp_exclusive_scan_u64_inplace(ctx, v_ck.p, (size_t)nr, totals.p + 3));
    uint64_t tot[4] = {0, 0, 0, 0};
    std::vector<int32_t> h_last((size_t)nc);
    std::vector<uint32_t> h_nal((size_t)nc);
    std::vector<unsigned long long> h_cols((size_t)nc);
    FZP_HIP(hipMemcpyAsync(tot, totals.p, 32, hipMemcpyDeviceToHost, st));
    FZP_TRY(last_pos.download(h_last.data(), (size_t)nc, st)); FZP_TRY(n_aligned.download(h_nal.data(), (size_t)nc, st)); FZP_TRY(n_cols.download(h_cols.data(), (size_t)nc, st));
    FZP_HIP(hipStreamSynchronize(st));
    b->n_rec = (int64_t)tot[0]; b->n_cig = (int64_t)tot[1]; b->n_seq = (int64_t)tot[2]; b->n_ck = (int64_t)tot[3];
    if (b->n_rec >= (1ll << 31)) { fzp_set_error("fzp_align_to_batch: %lld records (limit 2^31 per batch)", (long long)b->n_rec); return FZP_EINVAL; }
    b->h_goff.assign(1, 0); b->h_qid_off.assign(1, 0);
    for (int c = 0; c < nc; c++) {
        const int32_t limit = h_last[(size_t)c] > 0 ? h_last[(size_t)c] : 0;
        b->h_limit.push_back(limit);
        b->h_ref_len.push_back(j->h_ctg_len[(size_t)c]);
        b->h_goff.push_back(b->h_goff.back() + fzp_pos_pad(limit));
        b->n_eval += limit;
        b->h_qid_off.push_back(b->h_qid_off.back() + (int64_t)h_nal[(size_t)c]);
        b->n_columns += (int64_t)h_cols[(size_t)c];
    }
    b->n_pos = b->h_goff.back();
    b->n_qid = b->h_qid_off.back();
    b->h_slot_off = j->h_slot_off;
    const size_t nrec1 = (size_t)b->n_rec + 1;
    FZP_TRY(rec_read.alloc(nrec1)); FZP_TRY(b->rec_qid.alloc(nrec1)); FZP_TRY(b->rec_pos.alloc(nrec1)); FZP_TRY(b->rec_ctg.alloc(nrec1));
    FZP_TRY(b->cig_off.alloc(nrec1)); FZP_TRY(b->seq_off.alloc(nrec1)); FZP_TRY(b->ck_off.alloc(nrec1)); FZP_TRY(b->ctg_rec_begin.alloc((size_t)nc + 1));
    hipLaunchKernelGGL(k_plan_emit, dim3((unsigned)((std::max<int64_t>(nr, nc + 1) + 255) / 256)), dim3(256), 0, st, nr, nc, j->slot_ctg.p, j->slot_off.p, v_rec.p, v_cig.p, v_seq.p, v_ck.p, g_read.p, g_qid.p, g_acc.p, j->summ.p, totals.p,
                       rec_read.p, b->rec_qid.p, b->rec_pos.p, b->rec_ctg.p, b->cig_off.p, b->seq_off.p, b->ck_off.p, b->ctg_rec_begin.p);
    b->h_rec_begin.resize((size_t)nc + 1);
    FZP_TRY(b->ctg_rec_begin.download(b->h_rec_begin.data(), (size_t)nc + 1, st));
    // ---- gather the accepted records' CIGAR words and SEQ bytes
    FZP_TRY(b->cigar.alloc((size_t)b->n_cig)); FZP_TRY(b->seq.alloc((size_t)b->n_seq));
    if (b->n_rec > 0) {
        ProfScope ps(ctx, "k1_gather");
        hipLaunchKernelGGL(k_gather16, dim3((unsigned)b->n_rec, 4), dim3(256), 0, st, b->n_rec, rec_read.p, j->cig_start.p, j->cig.p, b->cig_off.p, b->cigar.p, j->read_ori.p,
                           j->read_woff.p, b->seq_off.p, b->seq.p);
    }
    FZP_TRY(b->ref.alloc((size_t)b->n_pos));
    for (int c = 0; c < nc; c++)   // evaluated prefix of every contig, device to device
        if (b->h_limit[(size_t)c]) FZP_HIP(hipMemcpyAsync(b->ref.p + b->h_goff[(size_t)c], j->ctg_ascii.p + j->h_ctg_aoff[(size_t)c], (size_t)b->h_limit[(size_t)c], hipMemcpyDeviceToDevice, st));
    FZP_TRY(b->ctg_goff.upload(b->h_goff.data(), b->h_goff.size(), st));
    FZP_TRY(b->ctg_qoff.upload(b->h_qid_off.data(), b->h_qid_off.size(), st));
    FZP_TRY(b->ctg_limit.upload(b->h_limit.data(), b->h_limit.size(), st));
    FZP_HIP(hipStreamSynchronize(st));
    FZP_HIP(hipGetLastError());
    b->have_aln = true;
    guard.p = nullptr;
    *out = b;
    return FZP_OK;
}
