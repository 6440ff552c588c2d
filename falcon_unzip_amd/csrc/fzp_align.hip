// fzp_align.hip -- K1: read -> contig alignment on gfx950 (the role of blasr + samtools sort,
// falcon_unzip/unzip.py:86-91).  Spec "fzalign v1.8" (anchored k-mers since v1.7, a band of 32 cells since v1.8; 64 selectable): oracle/align_oracle.c is its scalar twin and the
// kernels here match it bit-for-bit (summaries, CIGAR words, DP cell counts).  Parity vs blasr itself
// is UNPINNED (third-party binary, not vendored).
//
//   k_pack        ASCII -> 2 bit/base words (16 bases per u32, base m at bits 2m); k_revcomp: the other orientation of every read and contig, once per job
//   k_index_*     contig k-mers -> bucketed table of (key<<32 | position<<1 | strand bit), every sampled position, built partition by partition in LDS
//   k_seed        per read: hit list + diagonal-bin votes of sampled k-mers (both strands) in LDS, the two best windows
//   k_chain       per (read, window): the longest chain of hits (strict links, bridges over seedless stretches) -> anchor + WAYPOINTS every >= `piece` bases
//   k_slot_*      per read: its extension PIECES (waypoint to waypoint, the free end, the backward extension; both candidates) as DP slots;
//   k_sort_*, k_route, k_lists, k_plan_final: slots by decreasing length, dealt to the two DP kernels, their mask streams planned in launch order -- all on the device
//   k_swb / k_swb2 the banded DP, bit-sliced: one slot per lane (per pair of lanes); k_sw: one wave per slot (slots narrower than the band).  Per step the
//                 D and G trace-back masks go to HBM (8 bytes: at band 32 the whole masks, at band 64 their middle 32 lanes).  Integer VALU-bound, no MFMA.
//   k_tb_walk     per slot, one lane: walks the masks back from the terminal, emits a 2-bit op stream
//   k_join        per read, one wave: candidate selection, the winner's pieces joined into one op stream
//   k_tb_cigar    per read, one wave: best sub-path, op stream -> forward run-length CIGAR, clips, summary
//   k_plan_*      per contig: aligned reads ordered by (POS, read), record filters (phasing.py:72-75), record offsets
//   k_gather(16)  accepted records -> contiguous CIGAR + ASCII SEQ arrays for an alnset / the phasing batch
#include <algorithm>
#include <type_traits>
#include <map>
#include <mutex>

#include "fzp_batch.h"
#include "fzp_swb_core.h"
#include "fzp_cigar_core.h"

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
// (r5: a thread takes its 16 bases as five aligned words shifted into place and classifies four bytes at a time -- exact byte equality by carry-free arithmetic, code_of's
// table: C/c 1, G/g 2, T/t 3, everything else 0; sixteen byte loads and sixteen compare chains per word made this the slowest "streaming" kernel of a genome-scale run, 0.5 TB/s)
__device__ __forceinline__ uint32_t codes4(uint32_t x) {      // four ASCII bytes -> 8 bits, byte 0's code in bits 1:0
    const uint32_t y = x & 0xDFDFDFDFu;                       // either case
    auto eq = [](uint32_t v, uint32_t k) { const uint32_t z = v ^ k; const uint32_t t = (z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu; return ~(t | z | 0x7F7F7F7Fu); };      // 0x80 in every byte that equals k's
    const uint32_t c = eq(y, 0x43434343u), g = eq(y, 0x47474747u), t = eq(y, 0x54545454u);
    const uint32_t code = (c >> 7) | (g >> 6) | (t >> 7) | (t >> 6);
    return (code | (code >> 6) | (code >> 12) | (code >> 18)) & 0xffu;
}
__global__ void __launch_bounds__(256) k_pack(const uint8_t *__restrict__ ascii, const int64_t *__restrict__ seq_be, const int64_t *__restrict__ woff,
                                              uint32_t *__restrict__ out) {      // sequence s = ascii[seq_be[2s], seq_be[2s+1]); the buffer holds 32 bytes beyond its last sequence
    const int64_t s = blockIdx.x;
    const int64_t b0 = seq_be[2 * s], n = seq_be[2 * s + 1] - b0;
    const int64_t nw = ((n + 15) / 16 + 8 + 1) & ~1LL;   // zero pad words: 64-bit base windows may run past the end
    uint32_t *dst = out + woff[s];
    for (int64_t w = (int64_t)blockIdx.y * 256 + threadIdx.x; w < nw; w += (int64_t)gridDim.y * 256) {
        uint32_t v = 0;
        const int64_t base = w * 16;
        if (base < n) {
            const uint8_t *a = ascii + b0 + base;
            const uint32_t sh = (uint32_t)((uintptr_t)a & 3u);
            const uint32_t *p = (const uint32_t *)(a - sh);
            const uint32_t d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4];
            v = codes4(__builtin_amdgcn_alignbyte(d1, d0, sh)) | (codes4(__builtin_amdgcn_alignbyte(d2, d1, sh)) << 8) |
                (codes4(__builtin_amdgcn_alignbyte(d3, d2, sh)) << 16) | (codes4(__builtin_amdgcn_alignbyte(d4, d3, sh)) << 24);
            const int64_t left = n - base;
            if (left < 16) v &= (1u << (2 * (uint32_t)left)) - 1u;
        }
        dst[w] = v;
    }
}

__device__ __forceinline__ uint32_t base_at(const uint32_t *__restrict__ pk, int64_t i) { return (pk[i >> 4] >> ((i & 15) * 2)) & 3u; }

// the k-mer at base p: a 32-bit window over two adjacent words, by ONE funnel shift (v_alignbit_b32: ({hi, lo} >> shift) for shift 0..31).
// r6: until r5 this made a 64-bit word of the two and shifted that.  In k_revcomp's first word-at-a-time form the compiler turned that into global_load_dwordx2 +
// v_lshrrev_b64 with a per-lane shift register, and whole rows of 16 lanes came back shifted by the low bits of their STORE ADDRESS instead -- the right two words under
// the wrong shift, other rows every run, gone with any change to the sequence (the shift amount pinned, in an SGPR, 32-bit shifts, one register more); the unaligned 8-byte
// load itself is clean (67 M pairs, four access orders): tools/ubench/unaligned_pair.hip, profiles/r6_kmer_at_anomaly.txt.  Whatever the cause -- it is not in the source --
// a form without a 64-bit variable shift cannot produce that sequence, and it is one instruction where the other was three.
__device__ __forceinline__ uint32_t kmer_at(const uint32_t *__restrict__ pk, int64_t p, int k) {
    const uint32_t lo = pk[p >> 4], hi = pk[(p >> 4) + 1];
    const uint32_t key = __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(p & 15) * 2u);
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
constexpr int BRIDGE_MAX_GAP = 4096;   // v1.6: a chain may cross a seedless stretch of up to this many read bases (looser diagonal tolerance) ...
constexpr int BRIDGE_COST = 4;         // ... for the price of this many hits
constexpr int LONG_READ = 8192, LONG_STRIDE = 3;      // v1.6: reads of at least LONG_READ bases are sampled at LONG_STRIDE times the stride
constexpr int LONG_MS = 3, SAMPLE_CAP = 8192;         // v1.7: of a long read's selected k-mers every LONG_MS-th (x 2, x 3 .. per further 131 072 bases) is looked up, SAMPLE_CAP at most
constexpr int ANCH_DIV = 8;                           // v1.7: table slots are sized for 2 x (k-mer positions / ANCH_DIV) entries (an eighth of the positions is selected on random sequence)
constexpr int PIECE_LEN = 3072;        // v1.6: read bases between waypoints (at least)
constexpr int FZP_DEFAULT_BAND = 32;   // v1.8: cells of the adaptive band (fzp_align_params.band: 64 or 32; the twin's ORC_DEFAULT_BAND says the same)
constexpr int MAX_WP = 31;             // waypoints per candidate: a read has at most 2 x (MAX_WP + 1) = 64 slots, one per lane of k_join
__host__ __device__ __forceinline__ int32_t piece_len(int64_t n) { const int64_t p = (n + MAX_WP - 2) / (MAX_WP - 1); return (int32_t)(p > PIECE_LEN ? p : PIECE_LEN); }
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

// ---- v1.7: anchored k-mers.  A k-mer is SELECTED iff it starts with AC or ends with GT (codes 0 1 / 2 3): decided by the k-mer alone -- contig and read pick the same
// k-mers wherever they agree, on either strand (a k-mer ends with GT exactly when its reverse complement starts with AC) -- and by two of its bases: sixteen positions at
// a time with a dozen bit operations on the packed words.  Bit 2m of the result: position 16 wq + m is selected (k >= 8: positions up to 16 wq + 30 lie in the two words).
__device__ __forceinline__ uint32_t anchored_word(const uint32_t *__restrict__ pk, int64_t wq, int k) {
    const uint64_t x = (uint64_t)pk[wq] | ((uint64_t)pk[wq + 1] << 32);
    constexpr uint64_t M = 0x5555555555555555ull;
    const uint64_t lo = x & M, hi = (x >> 1) & M;
    const uint64_t isA = ~(lo | hi) & M, isC = lo & ~hi, isG = hi & ~lo, isT = hi & lo;
    const uint64_t ac = isA & (isC >> 2), gt = isG & (isT >> 2);
    return (uint32_t)(ac | (gt >> (2 * (k - 2))));
}
// ... restricted to the positions p <= last (= len - k) of the sequence
__device__ __forceinline__ uint32_t anchored_word_upto(const uint32_t *__restrict__ pk, int64_t wq, int k, int64_t last) {
    uint32_t f = anchored_word(pk, wq, k);
    const int64_t nv = last - 16 * wq + 1;                 // valid positions in this word
    if (nv < 16) f &= nv <= 0 ? 0u : ((1u << (2 * nv)) - 1u);
    return f;
}
// k_index_stage's counterpart: a workgroup takes ANCH_WORDS packed words of a contig (16 positions each), every selected position is staged
constexpr int ANCH_WORDS = 8192;
__global__ void __launch_bounds__(256) k_index_stage_anch(const uint32_t *__restrict__ ctg_pk, const int64_t *__restrict__ ctg_woff, const int64_t *__restrict__ ctg_len,
                                                          const int64_t *__restrict__ idx_off, const int32_t *__restrict__ idx_bits, const int64_t *__restrict__ part_off, int k,
                                                          uint64_t *__restrict__ table, uint32_t *__restrict__ cursor, int32_t *__restrict__ overflow) {
    __shared__ uint32_t hist[1 << 12], base[1 << 12];
    const int c = blockIdx.y;
    const int64_t nk = ctg_len[c] - k + 1;
    const int64_t w0 = (int64_t)blockIdx.x * ANCH_WORDS;
    if (w0 * 16 >= nk) return;
    const uint32_t *pk = ctg_pk + ctg_woff[c];
    const int bbits = idx_bits[c] - 2;
    const int pbits = bbits < PART_BITS ? bbits : PART_BITS;
    const int n_part = 1 << (bbits - pbits);
    uint64_t *tab = table + idx_off[c];
    uint32_t *cur = cursor + part_off[c];
    for (int i = threadIdx.x; i < n_part; i += 256) hist[i] = 0;
    __syncthreads();
    for (int pass = 0; pass < 2; pass++) {
        for (int64_t wq = w0 + threadIdx.x; wq < w0 + ANCH_WORDS && wq * 16 < nk; wq += 256) {
            for (uint32_t f = anchored_word_upto(pk, wq, k, nk - 1); f; f &= f - 1) {
                const int64_t pos = 16 * wq + (__builtin_ctz(f) >> 1);
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
        }
        __syncthreads();
        if (pass == 0) {
            for (int i = threadIdx.x; i < n_part; i += 256) { const uint32_t h = hist[i]; base[i] = h ? atomicAdd(&cur[i], h) : 0u; hist[i] = 0; }
            __syncthreads();
        }
    }
}


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
#ifndef FZP_SEED_SPT
#define FZP_SEED_SPT 2
#endif
constexpr int SEED_SPT = FZP_SEED_SPT;      // samples per thread and round (their bucket loads are in flight together).  r5, same box: 4 -> 110 registers, 4 waves per SIMD, k1_seed 1.79 ms; 2 -> 76 registers, 6 waves, 1.56 ms (the kernel waits on memory 60 % of its cycles: what counts is how many waves wait side by side)
__global__ void __launch_bounds__(256) k_seed(int64_t first, const uint32_t *__restrict__ read_pk, const int64_t *__restrict__ read_woff, const int32_t *__restrict__ read_len,
                                              const int32_t *__restrict__ read_ctg, const int64_t *__restrict__ ctg_len, const int64_t *__restrict__ idx_off,
                                              const int32_t *__restrict__ idx_bits, const uint64_t *__restrict__ table, int k, int stride, int min_hits,
                                              uint2 *__restrict__ hits_g, SeedWin *__restrict__ win, int anchored) {
    extern __shared__ uint32_t votes[];   // [NB] words: two vote bins of 16 bits each per word (a read has at most HIT_CAP = 4 096 hits: a bin never reaches 65 536) -- r5: half the LDS
                                          // of 32-bit bins, six workgroups per CU instead of three: the kernel lives on how many bucket probes it keeps in flight
    auto vote_at = [&](int x) -> uint32_t { return (votes[x >> 1] >> (16 * (x & 1))) & 0xffffu; };
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
    for (int i = threadIdx.x; i < NB; i += 256) votes[i] = 0;
    __syncthreads();
    if (n >= LONG_READ) stride *= LONG_STRIDE;  // v1.6: a long read has seeds to spare (a short one needs all of them)
    int64_t ns = (n - k) / stride + 1;         // v1.6: sampled FORWARD read offsets 0, stride, ...
    uint32_t n_hits = 0;                       // block-uniform
    const int lane = lane_id(), wid = threadIdx.x >> 6;
    // v1.7: the samples are the read's selected k-mers (start with AC / end with GT), every ms-th of them in read order.  What LDS keeps is one number per packed word -- how
    // many selected positions lie before it (a thread takes a word, a block scan numbers them) --; sample m = selected position number m * ms is then found where it is
    // needed: the word by bisection, the position inside it from the word's own bits (a few hundred samples of LDS instead of 32 KB: four workgroups still fit a CU).
    uint32_t *pref = votes + NB;                                      // [n / 16 + 2] (the launch sized the dynamic LDS for its longest read)
    const uint32_t ms = n >= LONG_READ ? (uint32_t)LONG_MS * (uint32_t)((n + 131071) / 131072) : 1u;
    const int64_t last = n - k;                                       // last k-mer position
    const int32_t n_words = (int32_t)(last / 16) + 1;
    if (anchored) {
        uint32_t n_sel = 0;                                           // block-uniform: selected positions before this round
        for (int32_t wb4 = 0; wb4 < n_words; wb4 += 1024) {           // four rounds' words are loaded before the first of their scans (a round is a barrier: their loads must not queue behind it)
            uint32_t cnt4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int32_t wq = wb4 + u * 256 + threadIdx.x;
                cnt4[u] = wq < n_words ? (uint32_t)__popc(anchored_word_upto(pk, wq, k, last)) : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int32_t wb = wb4 + u * 256;
                if (wb >= n_words) break;
                const int32_t wq = wb + threadIdx.x;
                const uint32_t cnt = cnt4[u];
                const uint32_t incl = wave_incl_scan_u32(cnt);
                if (lane == 63) wsum[wid] = incl;
                __syncthreads();
                uint32_t o = n_sel + incl - cnt, tot = 0;
#pragma unroll
                for (int w = 0; w < 4; w++) { if (w < wid) o += wsum[w]; tot += wsum[w]; }
                if (wq < n_words) pref[wq] = o;
                n_sel += tot;
                __syncthreads();
            }
        }
        if (threadIdx.x == 0) pref[n_words] = n_sel;
        __syncthreads();
        ns = min((int64_t)SAMPLE_CAP, (int64_t)((n_sel + ms - 1) / ms));
        stride = 1;
    }
    auto sample_pos = [&](int64_t m) -> int64_t {                     // forward offset of sample m < ns
        if (!anchored) return m * stride;
        const uint32_t o = (uint32_t)m * ms;
        int32_t a = 0, b2 = n_words;                                  // largest word a with pref[a] <= o (pref[n_words] = n_sel > o)
        while (b2 - a > 1) { const int32_t mid = (a + b2) >> 1; if (pref[mid] <= o) a = mid; else b2 = mid; }
        uint32_t f = anchored_word_upto(pk, a, k, last);
        for (uint32_t r = o - pref[a]; r; r--) f &= f - 1;
        return 16 * (int64_t)a + (__builtin_ctz(f) >> 1);
    };
    for (int64_t base = 0; base < ns && n_hits < (uint32_t)HIT_CAP; base += 256 * SEED_SPT) {
        uint32_t key[SEED_SPT], orr[SEED_SPT], bkt[SEED_SPT];
        int64_t pfs[SEED_SPT];
        uint4 lo[SEED_SPT], hi[SEED_SPT];
#pragma unroll
        for (int u = 0; u < SEED_SPT; u++) {
            const int64_t m = base + SEED_SPT * threadIdx.x + u;
            key[u] = 0; orr[u] = 0; bkt[u] = 0; pfs[u] = 0;
            lo[u] = hi[u] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
            if (m < ns) {
                pfs[u] = sample_pos(m);
                key[u] = canonical(kmer_at(pk, pfs[u], k), k, &orr[u]);
                bkt[u] = hash_slot(key[u], bbits);
                const uint4 *bp = (const uint4 *)(tab + (size_t)bkt[u] * 4);
                lo[u] = bp[0]; hi[u] = bp[1];
            }
        }
        uint32_t e0[SEED_SPT];
        int cnt[SEED_SPT];
        uint32_t mine = 0;
#pragma unroll
        for (int u = 0; u < SEED_SPT; u++) {
            const int64_t m = base + SEED_SPT * threadIdx.x + u;
            cnt[u] = 0; e0[u] = 0;
            if (m < ns) {
                cnt[u] = index_count(tab, bbits, key[u], bkt[u], lo[u], hi[u], &e0[u]);
                if (cnt[u] > MAX_OCC) cnt[u] = 0;
            }
            mine += (uint32_t)cnt[u];
        }
        // slots in (sample, position) order: exclusive scan of the per-thread counts
        const uint32_t incl = wave_incl_scan_u32(mine);
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        uint32_t off = n_hits + incl - mine, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { if (w < wid) off += wsum[w]; tot += wsum[w]; }
#pragma unroll
        for (int u = 0; u < SEED_SPT; u++) {
            const int64_t pf = pfs[u];
            auto emit = [&](uint32_t hit) {
                const int s_ = (int)((hit & 1u) ^ orr[u]);
                const int64_t cp = hit >> 1, i = s_ ? n - k - pf : pf;
                hits[off] = make_uint2(((uint32_t)s_ << 31) | (uint32_t)i, (uint32_t)cp);
                { const int x_ = s_ * NB + (int)((cp - i + n) >> shift); atomicAdd(&votes[x_ >> 1], 1u << (16 * (x_ & 1))); }
            };
            if (cnt[u] == 1) { if (off < (uint32_t)HIT_CAP) emit(e0[u]); off++; }
            else if (cnt[u] > 1) {                                   // rare: a k-mer with 2..MAX_OCC entries -- all of them, by position (collected here, one sample at a
                uint32_t ent[MAX_OCC];                               // time: four such arrays held across the scan cost the kernel a fourth wave per SIMD)
                (void)index_collect(tab, bbits, key[u], bkt[u], lo[u], hi[u], ent);
                for (int a = 1; a < cnt[u]; a++) {                   // insertion sort
                    const uint32_t v = ent[a];
                    int b = a - 1;
                    while (b >= 0 && ent[b] > v) { ent[b + 1] = ent[b]; b--; }
                    ent[b + 1] = v;
                }
                for (int a = 0; a < cnt[u]; a++, off++) {
                    if (off >= (uint32_t)HIT_CAP) break;
                    emit(ent[a]);
                }
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
        uint64_t sc = (uint64_t)vote_at(x) + vote_at(x + 1);
        uint64_t key = (sc << 32) | (uint64_t)(0xffffffffu - (uint32_t)x);
        best = key > best ? key : best;
    }
    best = block_max_u64(best, red);
    const uint32_t w1 = (uint32_t)(best >> 32);
    if ((int32_t)w1 < min_hits || w1 == 0) { if (threadIdx.x == 0) win[blockIdx.x] = sw; return; }
    const int x1 = (int)(0xffffffffu - (uint32_t)best);
    const int s1 = x1 >= NB, b1 = s1 ? x1 - NB : x1;
    // W2: the best window on the other strand or at least 3 (+ 1 per 16 384 read bases: k_chain's windows widen by as much) bins away
    const int ext = (int)(n >> 14);
    uint64_t best2 = 0;
    for (int x = threadIdx.x; x < 2 * NB; x += 256) {
        const int sx = x >= NB, b = sx ? x - NB : x;
        if (b + 1 >= NB) continue;
        if (sx == s1 && b - b1 < 3 + ext && b1 - b < 3 + ext) continue;
        uint64_t sc = (uint64_t)vote_at(x) + vote_at(x + 1);
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

// ---- seeding, part 2 (spec "chains", "waypoints"): one wave per (read, window).  The window's hits are walked in list order (reversed on
// the reverse strand), 64 list entries per load; the last 64 window hits sit one per lane (lane = running index mod 64), the
// new hit is tested against all of them at once and the best predecessor comes out of one wave-wide max.  The chain is a
// serial dependence per read: thousands of waves in flight hide it.  v1.6: a link is strict (as before) or a bridge (a gap of up to 4 096 bases, a
// looser diagonal bound, BRIDGE_COST hits of penalty); every ring entry also carries the last waypoint of its chain -- a hit at least `piece` bases
// after the waypoint before it becomes one, and says so in wpp[] (list index of the waypoint before it) -- so the longest chain's waypoints are a walk
// of a few links from its last one.
__global__ void __launch_bounds__(64) k_chain(int64_t first, int64_t count, const int32_t *__restrict__ read_len, const uint2 *__restrict__ hits_g,
                                              const SeedWin *__restrict__ win, Anchor *__restrict__ anc, Anchor *__restrict__ ancB, int16_t *__restrict__ wpp_g,
                                              int32_t *__restrict__ n_wp, int2 *__restrict__ wps) {
    const int64_t wv = blockIdx.x;
    if (wv >= 2 * count) return;
    const int64_t slot = wv >> 1;
    const int which = (int)(wv & 1);
    const int64_t r = first + slot;
    const int lane = lane_id();
    const SeedWin sw = win[slot];
    Anchor *out = which ? ancB : anc;
    const Anchor none = {0, 0, 0, 0};
    if (sw.n_hits == 0 || (which && !sw.have2)) { if (lane == 0) { out[r] = none; n_wp[2 * r + which] = 0; } return; }
    const int32_t n = read_len[r];
    const int32_t piece = piece_len(n);
    const int ext = n >> 14;      // a long read's diagonal drifts (CLR reads carry more inserted than deleted bases): its window widens by a bin per 16 384 bases
    const int ws = which ? sw.s2 : sw.s1, wb = which ? sw.b2 : sw.b1, shift = sw.shift;
    const uint2 *hits = hits_g + (size_t)slot * HIT_CAP;
    int16_t *wpp = wpp_g + ((size_t)slot * 2 + (size_t)which) * HIT_CAP;      // (a list of its own per window: the two windows of a read may share hits, and their waves run side by side)
    const int32_t nh = sw.n_hits;
    int32_t ri = 0, rcp = 0, rd = 0, rf = 0, rst = 0, rwp = 0, rwi = 0;      // ring: lane L holds window hit number e with e % 64 == L
    int32_t e = 0, best_f = 0, best_st = -1, best_wp = -1;
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
            inw = b >= wb - 1 - ext && b <= wb + 2 + ext;
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
            const bool near = dist < e && di >= 1 && di <= BRIDGE_MAX_GAP && cp > rcp;
            const bool strict = near && di <= CHAIN_MAX_GAP && dd <= 16 + (di >> 4);
            const bool bridge = near && !strict && dd <= 64 + (di >> 3);
            const int32_t v = strict ? rf + 1 : (bridge ? rf + 1 - BRIDGE_COST : 0);      // (a hit alone is a chain of 1: only v > 1 links it)
            const int32_t key = v > 1 ? ((v << 6) | (63 - dist)) : 0;
            const int32_t K = wave_max_nonneg_dpp(key);
            int32_t f = 1, st = h, wp = h, wi = i, wprev = -1;
            if (K > 0) {
                f = K >> 6;
                const int32_t src = (e - 1 - (63 - (K & 63))) & 63;
                st = __builtin_amdgcn_readlane(rst, src);
                const int32_t pw = __builtin_amdgcn_readlane(rwp, src), pi = __builtin_amdgcn_readlane(rwi, src);
                if (i - pi >= piece) wprev = pw; else { wp = pw; wi = pi; }
            }
            if (wp == h && lane == 0) wpp[h] = (int16_t)wprev;  // this hit is a waypoint of its chain: the one before it (-1: it starts the chain)
            if (lane == (e & 63)) { ri = i; rcp = cp; rd = d; rf = f; rst = st; rwp = wp; rwi = wi; }
            if (f > best_f) { best_f = f; best_st = st; best_wp = wp; }
            e++;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // lane 0 reads back below what it wrote to wpp[] above
    if (lane == 0) {
        Anchor a = none;
        int32_t nw = 0;
        if (best_st >= 0) {
            // v1.4: the chain's first hit IS the anchor (a cell of the true path); v1.6: its waypoints, first to last
            for (int32_t x = best_wp; x >= 0; x = wpp[x]) nw++;
            int2 *wo = wps + (size_t)(2 * r + which) * MAX_WP;
            int32_t at = nw;
            for (int32_t x = best_wp; x >= 0; x = wpp[x]) { at--; if (at < MAX_WP) { const uint2 hv = hits[x]; wo[at] = make_int2((int32_t)(hv.x & 0x7fffffffu), (int32_t)hv.y); } }
            const uint2 hv = hits[best_st];
            a.aligned = 1; a.strand = ws; a.i_a = (int32_t)(hv.x & 0x7fffffffu); a.c_a = (int32_t)hv.y;
        }
        out[r] = a;
        n_wp[2 * r + which] = nw < MAX_WP ? nw : MAX_WP;      // (more cannot be: they lie `piece` bases apart)
    }
}

// ---- the other orientation of every sequence, once per job: out = reverse complement of the packed sequence, same word offsets (LEN: the length array's type).
// A read's alignment on strand s reads orientation s; the BACKWARD extension from an anchor is the forward DP on the opposite orientation of both the read and the
// contig (both complemented: what matches still matches), so no reversed copies are made per run.
template <class LEN>
__global__ void __launch_bounds__(256) k_revcomp(const uint32_t *__restrict__ pk, const int64_t *__restrict__ woff, const LEN *__restrict__ len, uint32_t *__restrict__ out) {
    const int64_t sq = blockIdx.x;
    const int64_t n = (int64_t)len[sq];
    const int64_t nw = ((n + 15) / 16 + 8 + 1) & ~1LL;
    const uint32_t *src = pk + woff[sq];
    uint32_t *dst = out + woff[sq];
    for (int64_t w = (int64_t)blockIdx.y * 256 + threadIdx.x; w < nw; w += (int64_t)gridDim.y * 256) {
        // output bases 16 w .. 16 w + 15 are the complements of source bases n-1-16w down to n-16-16w: one 16-base window of the source, turned round (rc_key) -- r5: sixteen
        // single-base look-ups per word before
        uint32_t v = 0;
        const int64_t left = n - w * 16;                     // output bases of this word that exist
        if (left > 0) {
            const int64_t s0 = left - 16;                    // the window's first source base (negative: the window hangs over the sequence's start)
            uint32_t key;
            if (s0 >= 0) {      // (two plain word loads and a funnel shift: through kmer_at's 64-bit window this loop gave wrong upper halves now and then -- tools/ubench/revcomp_check.hip)
                const uint32_t sh = (uint32_t)(s0 & 15) * 2u;
                const uint32_t lo = src[s0 >> 4], hi = src[(s0 >> 4) + 1];
                key = sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
            } else key = src[0] << (2 * (uint32_t)(-s0));
            v = rc_key(key, 16);
            if (left < 16) v &= (1u << (2 * (uint32_t)left)) - 1u;
        }
        dst[w] = v;
    }
}

// ---- extension pieces as DP slots (v1.6).  A slot is one banded DP: a piece of a candidate's forward extension (from a waypoint to the next: INNER, it has to arrive
// at the sub-matrix's corner; from the last waypoint on: free), or its backward extension (free, on the opposite orientations).
struct Slot { int32_t read, flags, qb, tb, nq, nt, cap, ctg; };      // cap: DP steps it may take, a multiple of 64
constexpr int SLOT_QRC = 1, SLOT_TRC = 2, SLOT_INNER = 4, SLOT_CAND = 8, SLOT_BACK = 16;
__device__ __forceinline__ int32_t slot_cap(int32_t nq, int32_t nt) { return (nq + nt + 2 + 63) & ~63; }

// the slots of one candidate, in the order [backward][forward 0 .. m]; emit(slot) is called for each
template <class F>
__device__ __forceinline__ void cand_slots(const Anchor a, int which, int32_t nw, const int2 *__restrict__ wp, int32_t r, int32_t n, int32_t c, int64_t Lc, F emit) {
    if (!a.aligned || nw <= 0) return;
    const int32_t piece = piece_len(n);
    const int32_t fl = which ? SLOT_CAND : 0;
    if (a.i_a > 0 && a.c_a > 0) {      // backward: the opposite orientation of the read from n - i_a on, the contig's from Lc - c_a on
        const int32_t nq = min(a.i_a, piece), nt = (int32_t)min((int64_t)a.c_a, (int64_t)nq + nq / 4 + 64);
        emit(Slot{r, fl | SLOT_BACK | (a.strand ? 0 : SLOT_QRC) | SLOT_TRC, n - a.i_a, (int32_t)(Lc - a.c_a), nq, nt, slot_cap(nq, nt), c});
    }
    for (int32_t k = 0; k < nw; k++) {
        const int2 o = wp[k];
        int32_t nq, nt, f2 = fl | (a.strand ? SLOT_QRC : 0);
        if (k + 1 < nw) { nq = wp[k + 1].x - o.x; nt = wp[k + 1].y - o.y; f2 |= SLOT_INNER; }
        else { nq = min(n - o.x, 2 * piece); nt = (int32_t)min(Lc - o.y, (int64_t)nq + nq / 4 + 64); }
        emit(Slot{r, f2, o.x, o.y, nq, nt, slot_cap(nq, nt), c});
    }
}
// per read: how many slots, how many DP steps of capacity
__global__ void __launch_bounds__(256) k_slot_count(int64_t nr, const Anchor *__restrict__ anc, const Anchor *__restrict__ ancB, const int32_t *__restrict__ n_wp, const int2 *__restrict__ wps,
                                                    const int32_t *__restrict__ read_len, const int32_t *__restrict__ read_ctg, const int64_t *__restrict__ ctg_len,
                                                    uint32_t *__restrict__ cnt, uint32_t *__restrict__ capq, uint32_t *__restrict__ n_sec, int32_t swb_max_steps, int32_t use_bits,
                                                    unsigned long long *__restrict__ sw_capq) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool second = false;
    uint32_t wq = 0, wn = 0;                               // capacity / 64 and number of this read's slots that the wave-per-slot kernel will take (k_route's test): they keep whole masks
    if (r < nr) {
        uint32_t k = 0, cq = 0;
        const int32_t n = read_len[r], c = read_ctg[r];
        const int64_t Lc = ctg_len[c];
        auto tally = [&](const Slot &sl) { k++; cq += (uint32_t)(sl.cap >> 6); if (!(use_bits && sl.nq >= use_bits && sl.nt >= use_bits && sl.nq + sl.nt + 2 <= swb_max_steps)) { wq += (uint32_t)(sl.cap >> 6); wn++; } };      // (use_bits: the band's cells when the bit-sliced kernel runs, else 0)
        cand_slots(anc[r], 0, n_wp[2 * r], wps + (size_t)(2 * r) * MAX_WP, (int32_t)r, n, c, Lc, tally);
        cand_slots(ancB[r], 1, n_wp[2 * r + 1], wps + (size_t)(2 * r + 1) * MAX_WP, (int32_t)r, n, c, Lc, tally);
        cnt[r] = k; capq[r] = cq;
        second = ancB[r].aligned != 0;
    }
    const uint64_t m = __ballot(second);
    if (lane_id() == 0 && m) atomicAdd(n_sec, (uint32_t)__popcll(m));
    const int32_t ws = wave_sum_i32_dpp((int32_t)wq), wc = wave_sum_i32_dpp((int32_t)wn);
    if (lane_id() == 0 && ws) { atomicAdd(sw_capq, (unsigned long long)(uint32_t)ws); atomicAdd(sw_capq + 1, (unsigned long long)(uint32_t)wc); }
}
__global__ void __launch_bounds__(256) k_slot_emit(int64_t r_lo, int64_t r_hi, const Anchor *__restrict__ anc, const Anchor *__restrict__ ancB, const int32_t *__restrict__ n_wp, const int2 *__restrict__ wps,
                                                   const int32_t *__restrict__ read_len, const int32_t *__restrict__ read_ctg, const int64_t *__restrict__ ctg_len,
                                                   const uint32_t *__restrict__ slot_base, uint32_t s_lo, Slot *__restrict__ slots) {
    const int64_t r = r_lo + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= r_hi) return;
    const int32_t n = read_len[r], c = read_ctg[r];
    const int64_t Lc = ctg_len[c];
    Slot *o = slots + (slot_base[r] - s_lo);
    auto put = [&](const Slot &sl) { *o++ = sl; };
    cand_slots(anc[r], 0, n_wp[2 * r], wps + (size_t)(2 * r) * MAX_WP, (int32_t)r, n, c, Lc, put);
    cand_slots(ancB[r], 1, n_wp[2 * r + 1], wps + (size_t)(2 * r + 1) * MAX_WP, (int32_t)r, n, c, Lc, put);
}

// ---- the chunk's slots by decreasing capacity (a stable counting sort over cap / 64, 1 024 classes): the lanes of a bit-sliced wave run in lockstep, so a wave should
// hold slots of one length; and the longest go first, so the launch's tail is short ones.
constexpr int SORT_CLASSES = 1024;
__device__ __forceinline__ uint32_t sort_class(int32_t cap) { const uint32_t q = (uint32_t)cap >> 6; return (uint32_t)(SORT_CLASSES - 1) - (q < (uint32_t)(SORT_CLASSES - 1) ? q : (uint32_t)(SORT_CLASSES - 1)); }
__global__ void __launch_bounds__(256) k_sort_hist(uint32_t ns, uint32_t nblk, const Slot *__restrict__ slots, uint32_t *__restrict__ bh) {
    __shared__ uint32_t hist[SORT_CLASSES];
    for (int i = threadIdx.x; i < SORT_CLASSES; i += 256) hist[i] = 0;
    __syncthreads();
    const uint32_t x = blockIdx.x * 256 + threadIdx.x;
    if (x < ns) atomicAdd(&hist[sort_class(slots[x].cap)], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < SORT_CLASSES; i += 256) bh[(size_t)i * nblk + blockIdx.x] = hist[i];
}
__global__ void __launch_bounds__(256) k_sort_scatter(uint32_t ns, uint32_t nblk, const Slot *__restrict__ slots, const uint32_t *__restrict__ bh_scan, uint32_t *__restrict__ sorted) {
    __shared__ uint32_t keys[256];
    const uint32_t x = blockIdx.x * 256 + threadIdx.x;
    const uint32_t key = x < ns ? sort_class(slots[x].cap) : 0xffffffffu;
    keys[threadIdx.x] = key;
    __syncthreads();
    if (x >= ns) return;
    uint32_t rank = 0;
    for (uint32_t t = 0; t < threadIdx.x; t++) rank += keys[t] == key ? 1u : 0u;
    sorted[bh_scan[(size_t)key * nblk + blockIdx.x] + rank] = x;
}
// which DP kernel takes a slot: the bit-sliced one what spans the band on both sides and fits its step limit, the wave-per-slot one the rest
__global__ void __launch_bounds__(256) k_route(uint32_t ns, const Slot *__restrict__ slots, const uint32_t *__restrict__ sorted, int32_t swb_max_steps, int32_t use_bits, uint32_t *__restrict__ fits) {
    const uint32_t x = blockIdx.x * 256 + threadIdx.x;
    if (x >= ns) return;
    const Slot sl = slots[sorted[x]];
    fits[x] = (use_bits && sl.nq >= use_bits && sl.nt >= use_bits && sl.nq + sl.nt + 2 <= swb_max_steps) ? 1u : 0u;      // (use_bits: the band's cells, or 0)
}
// the launch lists: list[0, n_b) = the bit-sliced kernel's slots, list[n_b, ns) the others', both in sorted order; lq[y] = cap / 64 of list[y]; gq[g] = cap / 64 of the
// first (longest) slot of the g-th group of 64 bit-sliced slots = what every stream of that group's interleaved region gets
struct PlanTotals { uint64_t n_b, n_groups_q, lq_total, pad_; };
__global__ void __launch_bounds__(256) k_lists(uint32_t ns, const Slot *__restrict__ slots, const uint32_t *__restrict__ sorted, const uint32_t *__restrict__ fits, const uint32_t *__restrict__ pos_b,
                                               const uint64_t *__restrict__ n_b_dev, uint32_t *__restrict__ list, uint32_t *__restrict__ lq, uint32_t *__restrict__ gq) {
    const uint32_t x = blockIdx.x * 256 + threadIdx.x;
    if (x >= ns) return;
    const uint32_t n_b = (uint32_t)*n_b_dev;
    const uint32_t sl = sorted[x], q = (uint32_t)slots[sl].cap >> 6;
    const uint32_t y = fits[x] ? pos_b[x] : n_b + (x - pos_b[x]);
    list[y] = sl; lq[y] = q;
    if (fits[x] && (y & 63u) == 0u) gq[y >> 6] = q;
}
// where every slot's masks, move words and op stream go.  Bit-sliced slots: group g's region starts at record 4096 * (sum of gq before g), the x-th stream of the group at
// + 64 x, its 64-step blocks 4 096 records apart (what a wave writes during 64 steps lies within 64 KB); the others: streams of their own behind all groups.
__global__ void __launch_bounds__(256) k_plan_final(uint32_t ns, const uint32_t *__restrict__ list, const uint32_t *__restrict__ lq_scan, const uint32_t *__restrict__ gq_scan,
                                                    uint64_t *__restrict__ ptot, int64_t *__restrict__ tbo, int64_t *__restrict__ mvo, int32_t *__restrict__ tbs) {
    const uint32_t y = blockIdx.x * 256 + threadIdx.x;
    if (y >= ns) return;
    const uint32_t n_b = (uint32_t)ptot[0];
    const uint32_t sl = list[y];
    mvo[sl] = (int64_t)lq_scan[y];
    // bit-sliced slots: 8-byte records in the chunk's mask buffer, interleaved by launch group; the others: 16-byte records, streams of their own, in the buffer of whole masks
    if (y < n_b) { tbo[sl] = 4096ll * gq_scan[y >> 6] + 32ll * (y & 63u); tbs[sl] = 4096; }      // (swb_rec below: a group's 64-step block is two halves of 32 steps x 64 lanes)
    else { tbo[sl] = 64ll * ((int64_t)lq_scan[y] - (int64_t)lq_scan[n_b]); tbs[sl] = 64; }
    if (y == 0) { ptot[3] = 0; ptot[4] = 64ull * (ptot[2] - (n_b < ns ? (uint64_t)lq_scan[n_b] : ptot[2])); ptot[5] = 0; }      // the fail list is empty; whole masks of slots that land on it go behind the others'; [5]: k_swb's group counter
}

// Where step t of a bit-sliced slot's 8-byte mask records lies, in records from the slot's first (tbo): the 64 slots of a launch group share blocks of 4 096 records per 64
// steps, and inside a block the FIRST 32 steps of all 64 slots come before the second 32 -- [64-step block][half][slot][32 steps].  So the 32 slots one wave of walkers owns
// have the pieces they stage at the same time side by side: 8 KB in one run (r5; [block][slot][64 steps] before: 256-byte pieces 512 bytes apart, which HBM serves at 4.3 TB/s
// where runs of 1 KB and more get 6.1 -- tools/ubench/rand_block_bw.hip).
__device__ __forceinline__ int64_t swb_rec(int32_t t) { return (int64_t)(t >> 6) * 4096 + ((t >> 5) & 1) * 2048 + (t & 31); }

struct DpInfo { int32_t steps, best_t, best_lane, best_score; };
struct ReadPath { int32_t ok, strand, i_end, j_end, n_ops, pad_; int64_t steps; };      // a read's joined path (k_join -> k_tb_cigar): its end cell, its ops, the DP steps of all its slots

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
        bool upd;                                                                                          \
        int32_t val;                                                                                       \
        {                                                                                                  \
            const int32_t ci = i0 + lane, cj = t - ci;                                                     \
            val = inner ? Hn - gap * ((nq - 1 - ci) + (nt - 1 - cj)) : Hn;                                 \
            upd = val > bs && ci >= 0 && ci < nq && cj >= 0 && cj < nt && (ci == nq - 1 || cj == nt - 1);  \
        }                                                                                                  \
        bs = upd ? val : bs;                                                                               \
        bt = upd ? t : bt;                                                                                 \
        mvacc |= (uint64_t)(down ? 1 : 0) << (t & 63);                                                     \
        const int32_t top = __builtin_amdgcn_readlane(Hn, 0), bot = __builtin_amdgcn_readlane(Hn, BAND - 1); \
        X = H;            /* H(t-2) stays in its own lane layout: the next step shifts it as its two moves say */ \
        H = Hn;                                                                                            \
        pdown = down;                                                                                      \
        t++;                                                                                               \
        down = t < BAND ? ((t & 1) == 0) : !(top > bot);                                                   \
    }

// after step t-1 completed a 64-step chunk (or at the very end): the chunk's move record
#define SW_FLUSH(PARTIAL)                                                                                  \
    {                                                                                                      \
        if ((PARTIAL) || ((t - 1) & 63) == 63) {                                                           \
            if (lane == 0) mvr[(t - 1) >> 6] = make_ulonglong2(mvacc, (uint64_t)(uint32_t)(i0 - __popcll(mvacc))); \
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
//     move, so the steps do not keep a "previous move" register up to date (one SALU less per step: the scalar port has no slack, see HISTORY.md section 14).
// The store offset is the step counter: it enters 16 * steps below 2^31 and the signed overflow of its increment ends the block.
// No DPP source is written fewer than two instructions before it is read (gfx9 DPP hazard).
#define SWB_DPP_SHL " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define SWB_DPP_SHR " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
// one DOWN step.  HC: operand holding H of the previous step; XC: operand holding H of two steps ago, receives the new H;
// HDADD: the add that forms the diagonal operand (shifted by what the last two moves say); NP: parity after this step;
// the step ends by jumping to the variant (NP, previous = DOWN, next move) or to the exit of parity NP.
#define SWB_DOWN(LBL, HC, XC, HDADD, NP, KB, ST, LN)                                                   \
    "Lsw%=_" LBL ":\n\t"                                                                          \
    "s_bfe_u64 s[56:57], %[qb], %[qsel]\n\t"                                                      \
    "v_mov_b32_dpp %[qc], %[qc] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"                        \
    "s_add_u32 %[qsel], %[qsel], 2\n\t"                                                           \
    "v_max_i32_dpp %[mm], " HC ", " HC SWB_DPP_SHL "\n\t"                                         \
    "v_writelane_b32 %[qc], s56, " LN "\n\t"                                                      \
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
    "v_readlane_b32 %[bot], " XC ", " LN "\n\t"                                                   \
    "s_addk_i32 %[soff], 16\n\t"                                                                  \
    "s_cbranch_scc1 Lsw%=_end" NP "\n\t"                                                          \
    "s_cmp_gt_i32 %[top], %[bot]\n\t"                                                             \
    "s_cbranch_scc1 Lsw%=_p" NP "DR\n\t"                                                          \
    "s_branch Lsw%=_p" NP "DD\n"
#define SWB_RIGHT(LBL, HC, XC, HDADD, NP, KB, ST, LN)                                                   \
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
    "v_readlane_b32 %[bot], " XC ", " LN "\n\t"                                                   \
    "s_addk_i32 %[soff], 16\n\t"                                                                  \
    "s_cbranch_scc1 Lsw%=_end" NP "\n\t"                                                          \
    "s_cmp_gt_i32 %[top], %[bot]\n\t"                                                             \
    "s_cbranch_scc1 Lsw%=_p" NP "RR\n\t"                                                          \
    "s_branch Lsw%=_p" NP "RD\n"
// label "pPab": parity P (0: H in %[H], X in %[X]; 1: swapped), a = previous move, b = this move
// the diagonal predecessor of lane k is lane k - 1 + (DOWN moves among the last two) of H(t-2)
#define SW_BLOCK_ASM(ST, LN)                                                                        \
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
        SWB_DOWN("p0DD", "%[H]", "%[X]", "v_add_u32_dpp %[hd], %[X], %[sc]" SWB_DPP_SHL, "1", "", ST, LN)                                                                               \
        SWB_DOWN("p0RD", "%[H]", "%[X]", "v_add_u32_e32 %[hd], %[X], %[sc]", "1", "", ST, LN)                                                                                           \
        SWB_RIGHT("p0DR", "%[H]", "%[X]", "v_add_u32_e32 %[hd], %[X], %[sc]", "1", "", ST, LN)                                                                                          \
        SWB_RIGHT("p0RR", "%[H]", "%[X]", "v_add_u32_dpp %[hd], %[X], %[sc]" SWB_DPP_SHR, "1", "", ST, LN)                                                                              \
        SWB_DOWN("p1DD", "%[X]", "%[H]", "v_add_u32_dpp %[hd], %[H], %[sc]" SWB_DPP_SHL, "0", "", ST, LN)                                        \
        SWB_DOWN("p1RD", "%[X]", "%[H]", "v_add_u32_e32 %[hd], %[H], %[sc]", "0", "", ST, LN)                                                    \
        SWB_RIGHT("p1DR", "%[X]", "%[H]", "v_add_u32_e32 %[hd], %[H], %[sc]", "0", "", ST, LN)                                                   \
        SWB_RIGHT("p1RR", "%[X]", "%[H]", "v_add_u32_dpp %[hd], %[H], %[sc]" SWB_DPP_SHR, "0", "", ST, LN)                                       \
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
template <bool STORE, int BAND>
__device__ __forceinline__ void sw_block(int32_t &H, int32_t &X, int32_t &qc, int32_t &tc, const uint64_t qbits, const uint64_t tbits, int32_t &kb,
                                         void *tbp, uint32_t &soff, uint32_t &mv, int32_t &dn, int32_t &pm, const int32_t gapS,
                                         const int32_t vmatS, const int32_t vmisS) {
    int32_t hd, mm, sc, top, bot;
    uint32_t qsel = 2u << 16, tsel = 2u << 16;   // s_bfe_u64 operand: width 2, offset 0
    // (the band's last lane enters the asm as a literal: 63, or 31 in the 32-cell band of v1.8 -- there lanes 32..63 of the wave are switched off, and what a DPP shift
    //  would fetch from them reads as zero, which in the biased scores IS "outside the band")
    if constexpr (BAND == 64) { if constexpr (STORE) { SW_BLOCK_ASM(SWB_STORE, "63") } else { SW_BLOCK_ASM("", "63") } }
    else { if constexpr (STORE) { SW_BLOCK_ASM(SWB_STORE, "31") } else { SW_BLOCK_ASM("", "31") } }
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

template <bool STORE, int BAND>
__global__ void __launch_bounds__(64) k_sw(const uint64_t *__restrict__ lo_dev, const uint64_t *__restrict__ hi_dev, uint32_t hi_host, const uint32_t *__restrict__ list, const Slot *__restrict__ slots,
                                            const uint32_t *__restrict__ read_pk, const uint32_t *__restrict__ read_rc, const int64_t *__restrict__ read_woff,
                                            const uint32_t *__restrict__ ctg_pk, const uint32_t *__restrict__ ctg_rc, const int64_t *__restrict__ ctg_woff,
                                            const int64_t *__restrict__ tbo, const int64_t *__restrict__ mvo, const int32_t *__restrict__ tbs, uint2 *__restrict__ tb, ulonglong2 *__restrict__ mvw,
                                            int match, int mismatch, int gap, DpInfo *__restrict__ info) {
    const int lane = lane_id();
    // wave-uniform on purpose: everything indexed by the slot then lives in SGPRs / scalar loads.  One wave per workgroup: a finished slot frees its place at once.
    // This kernel's slots are list[lo .. hi): the launch list behind the bit-sliced kernel's part (in sorted order, longest first), or the fail list of the 8-byte walk.
    const uint32_t y = (lo_dev ? (uint32_t)*lo_dev : 0u) + blockIdx.x;
    if (y >= (hi_dev ? (uint32_t)*hi_dev : hi_host)) return;
    const uint32_t sl = list[y];
    const Slot S = slots[sl];
    // the records of steps 64 b .. 64 b + 63 of a slot start at record b * stride of its stream (64: a stream of its own)
    const int64_t toff = tbo[sl], moff = mvo[sl];
    const int32_t stride = tbs[sl];
    ulonglong2 *tbr = (ulonglong2 *)tb + toff;   // per step: {D mask, G mask} over the 64 band lanes
    ulonglong2 *mvr = mvw + moff;                // per 64 steps: {move bits, i0 before the chunk}
    const int32_t nq = S.nq, nt = S.nt;
    const bool inner = (S.flags & SLOT_INNER) != 0;      // v1.6: the extension has to arrive at the sub-matrix's corner -- a border cell is valued H - gap x (its distance to the corner)
    const uint32_t *qpk = ((S.flags & SLOT_QRC) ? read_rc : read_pk) + read_woff[S.read];
    const uint32_t *tpk = ((S.flags & SLOT_TRC) ? ctg_rc : ctg_pk) + ctg_woff[S.ctg];
    const int64_t qb = S.qb, tbase = S.tb;
    const int32_t max_steps = nq + nt + 2;
    // v1.8: the band is 64 or 32 cells.  With 32 the wave's upper half is switched off for good: every lane shift below then finds "no lane" (DPP: the fill value /
    // zero) where the band ends, ballots come back with their upper halves clear, and the lane reads and writes name lane BAND - 1.
    constexpr int HB = BAND / 2;
    if (BAND == 32 && lane >= 32) return;

    // state before step 0 (biased): H(-1), and X = H(-2) in the lane layout it was computed in
    int32_t H = (lane == HB || lane == HB + 1) ? SW_BIAS - gap : 0;
    int32_t X = lane == HB ? SW_BIAS : 0;
    bool pdown = false;                                // move of the (virtual) step -1: RIGHT
    int32_t qc, tc;
    {
        int32_t i = lane - (HB + 1), j = HB - lane;
        qc = (i >= 0 && i < nq) ? (int32_t)base_at(qpk, qb + i) : 4;
        tc = (j >= 0 && j < nt) ? (int32_t)base_at(tpk, tbase + j) : 5;
    }
    int32_t bs = 0, bt = -1;
    int32_t i0 = -(HB + 1), t = 0, qpos = HB - 1, tpos = HB + 1;
    uint64_t mvacc = 0;
    bool down = true;
    BaseStream qs, ts;
    qs.init(qpk, qb + (HB - 1));
    ts.init(tpk, tbase + (HB + 1));
    bool done = false;
    while (!done) {
        t = __builtin_amdgcn_readfirstlane(t); i0 = __builtin_amdgcn_readfirstlane(i0);
        // how many steps can run with every lane strictly inside the matrix?  Each step advances i0 or
        // lane 0's column by one, so min(rows left, columns left) steps are safe once the band is inside.
        int32_t safe = swb::sw_interior_safe(t, i0, nq, nt, BAND);      // (fzp_swb_core.h: shared with the host test that pins it)
        if (safe > 0) {
            // ---- interior: asm blocks of <= 32 steps
            int32_t qpos_i = i0 + BAND, tpos_i = t - i0;               // next bases to enter at the band's last lane / lane 0
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
                sw_block<STORE, BAND>(H, X, qc, tc, qbits, tbits, kb, tbp, soff, mv, dn, pm, gapS, vmatS, vmisS);
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
            qpos = i0 + BAND; tpos = t - i0;
            qs.init(qpk, qb + qpos);
            ts.init(tpk, tbase + tpos);
        } else {
            SW_STEP()
            if ((t & 63) == 0) SW_FLUSH(false)
            if (i0 > nq - 1 || (t - 1) - (i0 + BAND - 1) > nt - 1 || t >= max_steps) done = true;
        }
    }
    if ((t & 63) != 0) SW_FLUSH(true)
    asm volatile("s_dcache_wb" ::: "memory");   // the masks went through the scalar cache
    // best cell: max score, then earliest step, then lowest lane
    int32_t s_b = bs, t_b = bt < 0 ? 0x7fffffff : bt, l_b = lane;
#pragma unroll
    for (int d = BAND / 2; d >= 1; d >>= 1) {      // (over the band's lanes: the switched-off half of a 32-cell band has nothing to say)
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
        cur = ((uint64_t)(w0 < lim ? pk[w0] : 0u) | ((uint64_t)(w0 + 1u < lim ? pk[w0 + 1u] : 0u) << 32)) >> sh;      // (words from `lim` on: see LaneStreamL::init)
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
    __device__ __forceinline__ uint32_t peek() const { return (uint32_t)cur; }      // the next base in bits 1:0 (what lies above is the bases after it)
    __device__ __forceinline__ void drop(uint32_t en) { cur >>= 2u * en; have -= (int32_t)en; }
};

// The bit-sliced kernel exists in two register budgets, both compiled in (r6; until r5 the second was a build-time experiment):
//   one wave per SIMD  (RINGW 64, HOLDW 16, GRP 8: ~330 registers) -- the step with the fewest instructions,
//   two waves per SIMD (RINGW 32, HOLDW  8, GRP 4: <= 256 registers, 16 KB of LDS per wave) -- a step takes ~1.4 x as long but two run side by side.
// RINGW: words per lane in a stream's ring; HOLDW: words a bulk load brings; GRP: steps whose mask records leave together.
template <int RINGW, int HOLDW>
struct LaneStreamL {
    static constexpr int BULK_STEPS = 16 * HOLDW;      // steps between bulk loads (a step takes at most one base: HOLDW words at most leave the ring in between)
    const uint32_t *pk;            // the sequence's words
    uint32_t *ring;                // this lane's column of its stream's ring
    uint64_t cur;
    uint32_t pend, rd, wr, gw, lim;
    uint32_t hold[HOLDW], hold_base;     // HOLDW words on their way from HBM to the ring (bulk)
    bool held;
    int32_t have;
    // idx: the next base the stream hands out.  (Also called when a work unit picks a piece up in the middle, r6: then idx may lie at or past the sequence's end -- words from
    // `lim` on read as zero, exactly what the ring hands out there when the stream gets that far by itself.)
    __device__ __forceinline__ void init(const uint32_t *pk_, int64_t idx, uint32_t lim_, uint32_t *ring_) {
        pk = pk_; ring = ring_; lim = lim_; held = false; hold_base = 0;
#pragma unroll
        for (int q = 0; q < HOLDW; q++) hold[q] = 0u;
        const uint32_t w = (uint32_t)(idx >> 4);
        const uint32_t sh = (uint32_t)(idx & 15) * 2u;
        const uint32_t a0 = w < lim ? pk[w] : 0u, a1 = w + 1u < lim ? pk[w + 1u] : 0u;
        cur = ((uint64_t)a0 | ((uint64_t)a1 << 32)) >> sh;
        have = 32 - (int32_t)(idx & 15);
        gw = w + 2;
        for (int q0 = 0; q0 < RINGW; q0 += 16) {
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
        gw += RINGW; wr = RINGW; rd = 1;
        pend = ring[0];
    }
    __device__ __forceinline__ void refill(const int32_t popped) {      // popped: bases taken since the last call (the kernel knows from the moves: pop() does not count)
        have -= popped;
        const bool m = have <= 16;
        cur |= m ? (uint64_t)pend << (2 * have) : 0ull;
        have += m ? 16 : 0;
        const uint32_t nx = ring[(rd & (uint32_t)(RINGW - 1)) * 64];
        pend = m ? nx : pend;
        rd += m ? 1u : 0u;
    }
    // every BULK_STEPS steps (at most HOLDW words leave the ring in between).  What a call loads goes into the ring at the NEXT call: by then more than 63 vector-memory operations
    // have been issued behind the loads, which is more than can be outstanding -- no wait is needed to use them, and none is spent behind the mask stores.
    __device__ __forceinline__ void bulk() {
        if (held) {
#pragma unroll
            for (int q = 0; q < HOLDW; q++) ring[((wr + q) & (uint32_t)(RINGW - 1)) * 64] = hold_base + q < lim ? hold[q] : 0u;
            wr += (uint32_t)HOLDW;
        }
        held = wr - rd <= (uint32_t)(RINGW - 2 * HOLDW);      // (room for these words when they arrive, and for what may still be on its way)
        if (held) {
            hold_base = gw;
#pragma unroll
            for (int q = 0; q < HOLDW; q++) { const uint32_t ix = gw + q; hold[q] = pk[ix < lim ? ix : lim - 1u]; }
            gw += (uint32_t)HOLDW;
        }
    }
    __device__ __forceinline__ uint32_t peek() const { return (uint32_t)cur; }      // the next base in bits 1:0
    __device__ __forceinline__ void drop(uint32_t en) { cur >>= 2u * en; }           // en = 1: that base is taken
};

template <class STREAM, class BW = uint64_t>
struct SwbLaneT {                  // one extension's state (a lane's registers); BW: a word as wide as the band (v1.8: 64 or 32 cells)
    swb::PlanesT<BW> P, Q;         // difference planes of the previous anti-diagonal
    BW R0, R1, C0, C1;             // base windows as bit planes: bit k = read base i0 + k / contig base t - i0 - k
    STREAM qs, ts;
    int32_t i0, E2, sv0;           // E2 = (score of lane 63's cell - score of lane 0's) / 2;  sv0 = sum of lane 0's difference codes: its score is -259 + 2 sv0 - 3 (t + 1)
    uint32_t down, pdown;
    uint64_t mvacc;
};

// one DP step of every lane's extension.  CHECKED = false: the 64 steps of an interior block -- no lane of the wave can reach a border of its matrix in them, so
// there is nothing to validate, no terminal candidate and no end; lanes whose extension is over run along on their stale state (nothing of theirs is stored).
// CHECKED = true: validity of the bases near the ends, terminal candidates, the end of the extension.
// ---- the same step on a band of 32 cells (fzalign v1.8): every plane and window ONE register, the record the WHOLE masks.  Everything else as in swb_step below (the
// comments there): with lane W - 1 = 31 for 63, the border's closed form S(lane 0) = -3 - 8 * 16 = -131 for -259.
template <bool CHECKED, class LANE>
__device__ __forceinline__ void swb_step32(LANE &L, const int32_t t, const int s8, uint32_t &mv8, uint2 &rec, const int32_t nq, const int32_t nt, const int32_t max_steps, const bool inner, bool &active,
                                           bool &row_on, bool &col_on, int32_t &Hrow, int32_t &Hcol, int32_t &best, int32_t &bt, int32_t &bl, int32_t &steps) {
    using namespace swb;
    const uint32_t sd = L.down, sr = 1u - sd;
    L.i0 += (int32_t)sd;
    {
        const uint32_t wq = L.qs.peek(), wt = L.ts.peek();
        L.R0 = __builtin_amdgcn_alignbit(wq, L.R0, sd);
        L.R1 = __builtin_amdgcn_alignbit(wq >> 1, L.R1, sd);
        L.C0 = (L.C0 << sr) | (wt & sr); L.C1 = (L.C1 << sr) | ((wt >> 1) & sr);
        L.qs.drop(sd); L.ts.drop(sr);
    }
    const Planes32 p = {L.P.v0 << sr, L.P.v1 << sr, L.P.v2 << sr}, q = {L.Q.v0 >> sd, L.Q.v1 >> sd, L.Q.v2 >> sd};
    uint32_t xm = lut3<(uint8_t)((TA ^ TB) | TC)>(L.R0, L.C0, L.R1 ^ L.C1);
    const int32_t kr = nq - 1 - L.i0, kc = t - (nt - 1) - L.i0;      // lanes of the last row / the last column
    if (CHECKED) {
        const int32_t nv = kr + 1;
        const uint32_t bad_r = nv >= 32 ? 0u : (nv <= 0 ? ~0u : ~0u << nv);
        const uint32_t bad_c = kc <= 0 ? 0u : (kc >= 32 ? ~0u : ~(~0u << kc));
        xm |= bad_r | bad_c;
    }
    const uint32_t f = ((sd & L.pdown) << 31) | lut3<(uint8_t)(TA & ~TB)>(sr, L.pdown, 0u);
    uint32_t D, G;
    cells<uint32_t>(xm, f, 0u - sd, p, q, &L.P, &L.Q, &D, &G);
    rec = make_uint2(D, G);                                          // the whole band
    mv8 |= sd << s8;
    constexpr uint8_t SEL = (uint8_t)((TA & TB) | (~TA & TC));
    const uint32_t dm = 0u - sd;
    const uint32_t x0 = lut3<SEL>(dm, L.Q.v0, L.P.v0), x1 = lut3<SEL>(dm, L.Q.v1, L.P.v1), x2 = lut3<SEL>(dm, L.Q.v2, L.P.v2);
    const int32_t v0 = (int32_t)(lut3<SEL>(3u, lut3<SEL>(1u, x0, x1 << 1), x2 << 2) & 7u);
    const int32_t v31 = (int32_t)lut3<SEL>(3u, lut3<SEL>(1u, x0 >> 31, x1 >> 30), x2 >> 29);
    L.sv0 += v0;
    L.E2 += v31 - v0;
    if (CHECKED) {
        if (active) {
            const int32_t S0 = -131 + 2 * L.sv0 - 3 * (t + 1);
            const bool kc_in = kc >= 0 && kc <= 31, kr_in = kr >= 0 && kr <= 31;
            const bool col_start = !col_on && sr && kc == 0, row_start = !row_on && sd && kr == 31;
            if (col_start) Hcol = S0; else if (col_on && kc_in) Hcol += 2 * value_at(L.Q, kc & 31) - 3;
            if (row_start) Hrow = S0 + 2 * L.E2; else if (row_on && kr_in) Hrow += 2 * value_at(L.P, kr & 31) - 3;
            col_on = col_on || col_start; row_on = row_on || row_start;
            { const int32_t i = L.i0 + kc, vc = inner ? Hcol - 3 * (nq - 1 - i) : Hcol; if (col_on && kc_in && i >= 0 && i < nq && vc > best) { best = vc; bt = t; bl = kc; } }
            { const int32_t jj = t - (nq - 1), vr = inner ? Hrow - 3 * (nt - 1 - jj) : Hrow; if (row_on && kr_in && jj >= 0 && jj < nt && vr > best) { best = vr; bt = t; bl = kr; } }
            if (L.i0 > nq - 1 || t - (L.i0 + 31) > nt - 1 || t + 1 >= max_steps) { active = false; steps = t + 1; }
        }
    }
    L.pdown = sd;
    L.down = (CHECKED && (t + 1) < 32) ? (uint32_t)(((t + 1) & 1) == 0) : ((uint32_t)L.E2 >> 31) ^ 1u;
}

template <bool CHECKED, class LANE>
__device__ __forceinline__ void swb_step(LANE &L, const int32_t t, const int s8, uint32_t &mv8, uint2 &rec, const int32_t nq, const int32_t nt, const int32_t max_steps, const bool inner, bool &active,
                                         bool &row_on, bool &col_on, int32_t &Hrow, int32_t &Hcol, int32_t &best, int32_t &bt, int32_t &bl, int32_t &steps) {
    using namespace swb;
    const uint32_t sd = L.down, sr = 1u - sd;
    L.i0 += (int32_t)sd;
    {   // the windows slide: a new read base enters at lane 63 (DOWN), a new contig base at lane 0 (RIGHT)
        // (funnel shifts take the entering read base straight from the stream's word: on a RIGHT move they shift by nothing and it stays where it is)
        const uint32_t wq = L.qs.peek(), wt = L.ts.peek();      // the next base of either stream in bits 1:0
        const uint32_t r0l = (uint32_t)L.R0, r0h = (uint32_t)(L.R0 >> 32), r1l = (uint32_t)L.R1, r1h = (uint32_t)(L.R1 >> 32);
        L.R0 = ((uint64_t)__builtin_amdgcn_alignbit(wq, r0h, sd) << 32) | __builtin_amdgcn_alignbit(r0h, r0l, sd);
        L.R1 = ((uint64_t)__builtin_amdgcn_alignbit(wq >> 1, r1h, sd) << 32) | __builtin_amdgcn_alignbit(r1h, r1l, sd);
        L.C0 = (L.C0 << sr) | (uint64_t)(wt & sr); L.C1 = (L.C1 << sr) | (uint64_t)((wt >> 1) & sr);
        L.qs.drop(sd); L.ts.drop(sr);
    }
    const Planes p = {L.P.v0 << sr, L.P.v1 << sr, L.P.v2 << sr}, q = {L.Q.v0 >> sd, L.Q.v1 >> sd, L.Q.v2 >> sd};
    uint64_t xm = lut3<(uint8_t)((TA ^ TB) | TC)>(L.R0, L.C0, L.R1 ^ L.C1);
    const int32_t kr = nq - 1 - L.i0, kc = t - (nt - 1) - L.i0;      // lanes of the last row / the last column
    if (CHECKED) {   // bases past the read's / the window's end never match
        const int32_t nv = kr + 1;                                    // lanes k < nv hold read bases
        const uint64_t bad_r = nv >= 64 ? 0ull : (nv <= 0 ? ~0ull : ~0ull << nv);
        const uint64_t bad_c = kc <= 0 ? 0ull : (kc >= 64 ? ~0ull : ~(~0ull << kc));   // lanes k >= kc hold contig bases
        xm |= bad_r | bad_c;
    }
    const uint64_t f = ((uint64_t)((sd & L.pdown) << 31) << 32) | (uint64_t)lut3<(uint8_t)(TA & ~TB)>(sr, L.pdown, 0u);      // two moves the same way: the edge lane's diagonal predecessor is outside the band
    uint64_t D, G;
    cells<uint64_t>(xm, f, (uint64_t)0 - (uint64_t)sd, p, q, &L.P, &L.Q, &D, &G);
    rec = make_uint2((uint32_t)(D >> 16), (uint32_t)(G >> 16));      // what leaves is the middle of the band: lanes 16..47
    mv8 |= sd << s8;                                                 // the group's moves (the caller puts them into the block's move word)
    // the edge cells' scores: every lane's cell moved down (its vertical difference) or right (its horizontal one).  (Spelled out as three-input selects: left to
    // itself the compiler spends two instructions on each of the six words and as many again on putting the bits together.)
    constexpr uint8_t SEL = (uint8_t)((TA & TB) | (~TA & TC));                 // a ? b : c
    const uint32_t dm = 0u - sd;
    const uint32_t xl0 = lut3<SEL>(dm, (uint32_t)L.Q.v0, (uint32_t)L.P.v0), xl1 = lut3<SEL>(dm, (uint32_t)L.Q.v1, (uint32_t)L.P.v1), xl2 = lut3<SEL>(dm, (uint32_t)L.Q.v2, (uint32_t)L.P.v2);
    const uint32_t xh0 = lut3<SEL>(dm, (uint32_t)(L.Q.v0 >> 32), (uint32_t)(L.P.v0 >> 32)), xh1 = lut3<SEL>(dm, (uint32_t)(L.Q.v1 >> 32), (uint32_t)(L.P.v1 >> 32)),
                   xh2 = lut3<SEL>(dm, (uint32_t)(L.Q.v2 >> 32), (uint32_t)(L.P.v2 >> 32));
    const int32_t v0 = (int32_t)(lut3<SEL>(3u, lut3<SEL>(1u, xl0, xl1 << 1), xl2 << 2) & 7u);                 // bit 0 of the three low words
    const int32_t v63 = (int32_t)lut3<SEL>(3u, lut3<SEL>(1u, xh0 >> 31, xh1 >> 30), xh2 >> 29);             // bit 31 of the three high words
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
            // (v1.6, inner pieces: a border cell is valued by the global alignment through it -- its score minus the gap moves from it to the sub-matrix's corner)
            { const int32_t i = L.i0 + kc, vc = inner ? Hcol - 3 * (nq - 1 - i) : Hcol; if (col_on && kc_in && i >= 0 && i < nq && vc > best) { best = vc; bt = t; bl = kc; } }
            { const int32_t jj = t - (nq - 1), vr = inner ? Hrow - 3 * (nt - 1 - jj) : Hrow; if (row_on && kr_in && jj >= 0 && jj < nt && vr > best) { best = vr; bt = t; bl = kr; } }
            if (L.i0 > nq - 1 || t - (L.i0 + 63) > nt - 1 || t + 1 >= max_steps) { active = false; steps = t + 1; }
        }
    }
    L.pdown = sd;
    L.down = (CHECKED && (t + 1) < 64) ? (uint32_t)(((t + 1) & 1) == 0) : ((uint32_t)L.E2 >> 31) ^ 1u;      // DOWN while lane 63's cell scores at least lane 0's
}

// ---- work units (r6).  A launch group is 64 slots of about one length, and until r5 a persistent wave pulled whole groups: 3 568 groups of two lengths on 1 024 (or, with
// the two-wave budget, 2 048) wave slots end in a round that leaves half of the chip idle, and the longest group of all is a chain of 13 000 steps that must not start
// late or share its SIMD badly (profiles/r5_swb_waves_one_vs_two.txt, profiles/r6_swb_units.txt).  Now the kernel runs a group in UNITS of `ub` 64-step blocks and
// schedules LONGEST REMAINING FIRST: a group's level is the number of units it still has; at every unit boundary the wave looks whether work of a higher level waits
// -- a group nobody has started, or one another wave has parked -- and if so parks its own group and takes that one; otherwise it carries on, at no cost.  Parking is
// cheap because the state of a piece between two blocks is small -- two x three difference planes, four base-window planes, the steering and terminal scalars: 30
// words per lane; the base streams are found again from (i0, t) -- 8 KB per group in HBM.  So the long chains run without a break from the start, groups of one length
// take turns, and the launch ends within a unit on every wave.  Results do not depend on the cuts: a block's outcome is a function of the state at its first step.
//   level k's work: the FRESH groups of exactly k units -- a run [c[k+1], c[k]) of the sorted list, handed out by a counter -- and a bag of PARKED groups (at most
//   c[k+1] entries: a group is parked at a level at most once), filled behind a tail counter and counted in `avail`; a taker first takes one off `avail` -- and gives
//   it back if there was none -- and only then a ticket of the head counter: nobody ever holds a ticket for an entry that no push has been started for, so no wait
//   depends on work that may never come.  (A compare-and-swap on the head counter, the first form, let 2 048 waves retry against each other: 50 ms.)
// How a parked group's state is ordered: the state words and the bag entry are agent-scope RELAXED atomics -- stores that write through and loads that read past an XCD's
// own L2 --, the writer waits for its stores (s_waitcnt) before it posts the entry, the reader looks at the state only after it has seen the entry.  Release / acquire
// FENCES are the textbook form and were measured first: at agent scope each one writes back / invalidates the XCD's whole L2, and ~40 000 of them per launch took
// the launch from 6 to 15 ms.
constexpr int SWB_MAX_LEVELS = 64;                     // a wave looks at all levels at once, one per lane
constexpr int SWB_STATE_WORDS = 32;                    // per lane and group (30 used)
constexpr uint32_t SWB_NONE = 0xffffffffu;
struct SwbUnitPlan {
    uint32_t n_groups, nk_max, ub, hyst;
    uint32_t c[SWB_MAX_LEVELS + 2];                    // c[k] = groups of at least k units (c[0] = c[1] = all); the fresh groups of level k are [c[k+1], c[k])
    uint32_t bag_off[SWB_MAX_LEVELS + 2];              // where level k's bag of parked groups begins
};
struct SwbUnitCtr { uint32_t fresh[SWB_MAX_LEVELS + 2], head[SWB_MAX_LEVELS + 2], tail[SWB_MAX_LEVELS + 2]; int32_t avail[SWB_MAX_LEVELS + 2]; uint32_t done, n_log, pad_[2]; };      // zeroed per launch
__global__ void __launch_bounds__(256) k_swb_units(const uint64_t *__restrict__ n_b_dev, const uint32_t *__restrict__ gq, uint32_t ub_req, uint32_t hyst, SwbUnitPlan *__restrict__ plan) {
    __shared__ uint32_t cc[SWB_MAX_LEVELS + 2];
    const uint32_t n_groups = (uint32_t)((*n_b_dev + 63) / 64);
    const uint32_t g0 = n_groups ? gq[0] : 0u;                                   // blocks of the longest group (gq does not increase)
    uint32_t ub = ub_req ? ub_req : 1u;
    if ((g0 + ub - 1) / ub > (uint32_t)SWB_MAX_LEVELS) ub = (g0 + SWB_MAX_LEVELS - 1) / SWB_MAX_LEVELS;
    const uint32_t nk_max = (g0 + ub - 1) / ub;
    for (uint32_t k = threadIdx.x; k <= (uint32_t)SWB_MAX_LEVELS + 1; k += 256) {
        uint32_t lo = 0;
        if (k <= nk_max) {
            const uint32_t thr = k ? (k - 1) * ub : 0u;                          // groups of more than thr blocks have at least k units
            uint32_t hi = n_groups;                                              // first g with gq[g] <= thr
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (gq[mid] > thr) lo = mid + 1; else hi = mid; }
        }
        cc[k] = lo;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (uint32_t k = 0; k <= (uint32_t)SWB_MAX_LEVELS + 1; k++) { plan->c[k] = cc[k]; plan->bag_off[k] = acc; acc += k <= (uint32_t)SWB_MAX_LEVELS ? cc[k + 1] : 0u; }
        plan->n_groups = n_groups; plan->nk_max = nk_max; plan->ub = ub; plan->hyst = hyst;
    }
}

// a wave claims the best waiting work of a level >= min_level: returns the level (0: none), the group, and whether it is fresh.  All lanes take part (lane x looks at level
// nk_max - x); the claim itself is lane 0's.
__device__ __forceinline__ uint32_t swb_claim(const SwbUnitPlan *__restrict__ plan, SwbUnitCtr *ctr, uint32_t *bag, const uint32_t nk_max, const uint32_t min_level, uint32_t &grp, bool &fresh) {
    const uint32_t lane = threadIdx.x;
    for (;;) {
        const uint32_t k = nk_max > lane ? nk_max - lane : 0u;
        bool av_f = false, av_p = false;
        if (k >= min_level && k >= 1u) {
            const uint32_t nf = plan->c[k] - plan->c[k + 1];
            av_f = __hip_atomic_load(&ctr->fresh[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nf;
            av_p = __hip_atomic_load(&ctr->avail[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0;
        }
        const uint64_t m = __ballot(av_f || av_p);
        if (!m) return 0u;
        const int src = __builtin_ctzll(m);                                      // the highest level with something waiting
        const uint32_t kk = nk_max - (uint32_t)src;
        const bool use_p = __builtin_amdgcn_readlane((int32_t)av_p, src) != 0;   // parked first: its state is waiting in HBM
        uint32_t got = SWB_NONE;
        if (lane == 0) {
            if (use_p) {
                if (atomicAdd(&ctr->avail[kk], -1) > 0) {
                    uint32_t *e = bag + plan->bag_off[kk] + atomicAdd(&ctr->head[kk], 1u);
                    for (;;) { got = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (got != SWB_NONE) break; __builtin_amdgcn_s_sleep(1); }      // (its push is between the tail counter and this store)
                } else atomicAdd(&ctr->avail[kk], 1);
            } else {
                const uint32_t nf = plan->c[kk] - plan->c[kk + 1];
                const uint32_t tk = atomicAdd(&ctr->fresh[kk], 1u);
                if (tk < nf) got = plan->c[kk + 1] + tk;
            }
        }
        got = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)got);
        if (got != SWB_NONE) { grp = got; fresh = !use_p; return kk; }
        // somebody else was faster: look again
    }
}

// RING: the base streams through rings in LDS (LaneStreamL) or straight from HBM (LaneStream; FZP_SWB_NO_RING, for comparisons)
template <bool RING, int RINGW, int HOLDW, int GRP, int WPS, int BAND = 64>
__global__ void __attribute__((amdgpu_waves_per_eu(WPS))) __launch_bounds__(256)
k_swb(const uint64_t *__restrict__ n_b_dev, const uint32_t *__restrict__ list, const Slot *__restrict__ slots,
      const uint32_t *__restrict__ read_pk, const uint32_t *__restrict__ read_rc, const int64_t *__restrict__ read_woff,
      const uint32_t *__restrict__ ctg_pk, const uint32_t *__restrict__ ctg_rc, const int64_t *__restrict__ ctg_woff,
      const int64_t *__restrict__ tbo, const int64_t *__restrict__ mvo, uint2 *__restrict__ tb, ulonglong2 *__restrict__ mvw,
      DpInfo *__restrict__ info, int dbg, uint64_t *__restrict__ wave_log,
      const SwbUnitPlan *__restrict__ plan, SwbUnitCtr *ctr, uint32_t *bag, uint32_t *ustate) {
    using namespace swb;
    typedef LaneStreamL<RINGW, HOLDW> StreamL;
    __shared__ uint32_t srng[RING ? 2 * RINGW * 64 : 1];                   // the two streams' rings
    // PERSISTENT waves (r5): the launch has as many one-wave workgroups as the chip has wave slots for this kernel, and a wave PULLS work -- since r6 by the level
    // scheme above -- until every group is done.
    const uint32_t n_groups = plan->n_groups, nk_max = plan->nk_max, ub = plan->ub, hyst = plan->hyst;
    const int64_t n_b = (int64_t)*n_b_dev;
    uint32_t c_level = 0, c_grp = 0;      // work claimed at a unit boundary, to be taken up next
    bool c_fresh = false;
    // two waves of this kernel on a SIMD do not share it evenly: the arbiter takes the older wave whenever it is ready, and the younger one runs at 800 ns per step beside
    // the older one's 340 (profiles/r6_swb_units.txt).  dbg bit 2: the two take turns at the higher issue priority, ~20 us each by the shared clock
    uint32_t wid = 0;
    if (dbg & 4) { uint32_t hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); wid = hw & 1u; }
  for (;;) {
    if (!c_level) {
        c_level = swb_claim(plan, ctr, bag, nk_max, 1u, c_grp, c_fresh);
        if (!c_level) {      // nothing waits: other waves are still running groups and may park one for a longer chain (they will not while nothing waits, but the counters are read one by one)
            if (__hip_atomic_load(&ctr->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_groups) break;
            __builtin_amdgcn_s_sleep(64);
            continue;
        }
    }
    const uint32_t grp = c_grp;
    uint32_t level = c_level;
    const bool fresh = c_fresh;
    c_level = 0;
    // wave_log (FZP_SWB_WAVE_LOG, a measurement aid): per stretch a wave spent on a group {start, end} of the 100 MHz counter, the hardware id and the group, the steps it ran
    const uint64_t t_begin = wave_log ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // dbg: MEASUREMENT switches (FZP_SWB_DBG, tools/runs/swb_probe.py; the results of such a run are not used): bit 0 = no mask stores, bit 1 = no stream refills
    // (workgroups of one wave: four-wave workgroups, which suit k_swb2, put 256 mask streams on a CU and cost this kernel 10 % -- address translation again)
    const int64_t li = (int64_t)grp * 64 + threadIdx.x;
    bool active = li < n_b;
    const uint32_t sl = list[active ? li : 0];      // the lane's slot; the lanes of a wave stand side by side in the launch list, and so do their mask streams
    const Slot S = slots[sl];
    // (64 streams scattered over the buffer cost twice the time in address translation alone: the plan interleaves a wave's streams block by block, stride 4 096 records)
    uint2 *tbr = tb + tbo[sl];                            // per step {D, G} over band lanes 16..47
    ulonglong2 *mvr = mvw + mvo[sl];
    const int32_t nq = S.nq, nt = S.nt;
    const bool inner = (S.flags & SLOT_INNER) != 0;
    const uint32_t *qpk = ((S.flags & SLOT_QRC) ? read_rc : read_pk) + read_woff[S.read];
    const uint32_t *tpk = ((S.flags & SLOT_TRC) ? ctg_rc : ctg_pk) + ctg_woff[S.ctg];
    const int64_t qb = S.qb, tbase = S.tb;
    const int32_t max_steps = nq + nt + 2;
    const uint32_t qlim = (uint32_t)((qb + nq + 15) >> 4) + 1u, tlim = (uint32_t)((tbase + nt + 15) >> 4) + 1u;
    typedef typename std::conditional<BAND == 64, uint64_t, uint32_t>::type BW;      // a word as wide as the band
    constexpr int HB = BAND / 2;
    SwbLaneT<typename std::conditional<RING, StreamL, LaneStream>::type, BW> L;
    bool row_on = false, col_on = false;
    int32_t Hrow = 0, Hcol = 0, best = NEGV, bt = -1, bl = 0, steps = 0;
    int32_t t = 0;                                            // wave-uniform
    uint32_t *const ust = ustate + (size_t)grp * (SWB_STATE_WORDS * 64) + threadIdx.x;      // the group's state while it is parked: word w of lane x at [w * 64 + x]
    if (fresh) {
        // step -1: the anti-diagonal i + j = -1 of the virtual border, lane k = cell (k - HB - 1, HB - k) (HB = 32: (k - 33, 32 - k)): Pv = 0 where j >= 0 (k <= HB) else 4,
        // Qv = 0 where i >= 0 (k > HB) else 4
        L.P = {0, 0, (BW)(~(BW)0 << (HB + 1))}; L.Q = {0, 0, (BW)(((BW)1 << (HB + 1)) - 1)};
        L.R0 = L.R1 = L.C0 = L.C1 = 0;
        for (int k = HB + 1; k < BAND; k++) { const uint32_t c = base_at(qpk, qb + (k - HB - 1)); L.R0 |= (BW)(c & 1u) << k; L.R1 |= (BW)(c >> 1) << k; }
        for (int k = 0; k <= HB; k++) { const uint32_t c = base_at(tpk, tbase + (HB - k)); L.C0 |= (BW)(c & 1u) << k; L.C1 |= (BW)(c >> 1) << k; }
        L.i0 = -(HB + 1); L.E2 = 8; L.sv0 = 0;                      // at step -1 from the border's closed form: lane 0's cell scores -3 - 8 HB (-259 / -131), the last lane's 16 more
        L.down = 1; L.pdown = 0;
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // (orders the compiler: no instruction)
        auto ld = [&](int w) { return __hip_atomic_load(ust + w * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        auto ld64 = [&](int w) { return (uint64_t)ld(w) | ((uint64_t)ld(w + 1) << 32); };
        L.P = {(BW)ld64(0), (BW)ld64(2), (BW)ld64(4)}; L.Q = {(BW)ld64(6), (BW)ld64(8), (BW)ld64(10)};      // (a 32-cell band leaves the slots' upper words unused)
        L.R0 = (BW)ld64(12); L.R1 = (BW)ld64(14); L.C0 = (BW)ld64(16); L.C1 = (BW)ld64(18);
        L.i0 = (int32_t)ld(20); L.E2 = (int32_t)ld(21); L.sv0 = (int32_t)ld(22);
        const uint32_t fl = ld(23);
        L.down = fl & 1u; L.pdown = (fl >> 1) & 1u; active = (fl & 4u) != 0; row_on = (fl & 8u) != 0; col_on = (fl & 16u) != 0;
        Hrow = (int32_t)ld(24); Hcol = (int32_t)ld(25); best = (int32_t)ld(26); bt = (int32_t)ld(27); bl = (int32_t)ld(28); steps = (int32_t)ld(29);
        t = (int32_t)__builtin_amdgcn_readfirstlane((int32_t)ld(30));
    }
    const int32_t t_first = t;
    // the streams: the next read base to enter is i0 + BAND of the piece (HB - 1 at the start), the next contig base t - i0 (HB + 1 at the start)
    if constexpr (RING) {
        L.qs.init(qpk, qb + L.i0 + BAND, qlim, srng + threadIdx.x);
        L.ts.init(tpk, tbase + t - L.i0, tlim, srng + RINGW * 64 + threadIdx.x);
    } else {
        L.qs.init(qpk, qb + L.i0 + BAND, qlim);
        L.ts.init(tpk, tbase + t - L.i0, tlim);
    }
    int32_t i0_ref = L.i0;                                    // i0 when the streams were last topped up
    for (;;) {      // unit after unit of this group, until it ends or longer work waits
        const int32_t t_end = t + (int32_t)(ub * 64u);
    while (t < t_end && __ballot(active)) {
        const bool blk_active = active;
        const int32_t i0_blk = L.i0, e2_blk = L.E2;      // (E2 at the block's first step rides in the move word's spare half: where in the band the path is likely to be, HISTORY.md section 14)
        L.mvacc = 0;
        if (dbg & 4) { if ((((uint32_t)__builtin_amdgcn_s_memrealtime() >> 11) ^ wid) & 1u) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        // an interior block?  every running lane more than 64 steps away from its last row and its last column (a step brings either one closer by at most one)
        const bool far = !active || (nq - 1 - (L.i0 + BAND - 1) > 64 && nt - 1 - (t - L.i0) > 64);
        const bool interior = t >= 64 && __ballot(!far) == 0ull;
        if constexpr (RING) { if ((t & (StreamL::BULK_STEPS - 1)) == 0 && t > 0) { L.qs.bulk(); L.ts.bulk(); } }
        for (int g8 = 0; g8 < 64 / GRP; g8++) {
            const bool grp_active = active;
            uint2 rec[GRP];
            uint32_t mv8 = 0;
            if (interior) {
#pragma unroll
                for (int s8 = 0; s8 < GRP; s8++) {
                    if constexpr (BAND == 64) swb_step<false>(L, t, s8, mv8, rec[s8], nq, nt, max_steps, inner, active, row_on, col_on, Hrow, Hcol, best, bt, bl, steps);
                    else swb_step32<false>(L, t, s8, mv8, rec[s8], nq, nt, max_steps, inner, active, row_on, col_on, Hrow, Hcol, best, bt, bl, steps);
                    t++;
                }
            } else {
#pragma unroll
                for (int s8 = 0; s8 < GRP; s8++) {
                    if constexpr (BAND == 64) swb_step<true>(L, t, s8, mv8, rec[s8], nq, nt, max_steps, inner, active, row_on, col_on, Hrow, Hcol, best, bt, bl, steps);
                    else swb_step32<true>(L, t, s8, mv8, rec[s8], nq, nt, max_steps, inner, active, row_on, col_on, Hrow, Hcol, best, bt, bl, steps);
                    t++;
                }
            }
            L.mvacc |= (uint64_t)mv8 << ((t - GRP) & 63);
            // the streams top up every 16 steps, and they do it HERE, ahead of a group's stores: taking the word loaded 16 steps ago means waiting on the vector-memory
            // counter, which also counts the mask stores -- at this point the youngest of those are 8 steps old and done, right behind a group they would be in flight
            if ((g8 & (16 / GRP - 1)) == 16 / GRP - 1 && !(dbg & 2)) {
                if constexpr (RING) { const int32_t dq = L.i0 - i0_ref; L.qs.refill(dq); L.ts.refill(16 - dq); i0_ref = L.i0; }      // (16 steps: one base each, a read base on a DOWN move)
                else { L.qs.refill(); L.ts.refill(); }
            }
            if (grp_active && !(dbg & 1)) {      // what leaves is the middle of the band: lanes 16..47 of D and of G, 8 B per step (the walker says so if its path ever needs more)
#pragma unroll
                for (int s8 = 0; s8 < GRP; s8 += 2)
                    *(uint4 *)(tbr + swb_rec(t - GRP) + s8) = make_uint4(rec[s8].x, rec[s8].y, rec[s8 + 1].x, rec[s8 + 1].y);
            }
        }
        if (blk_active) {
            mvr[(t - 1) >> 6] = make_ulonglong2(L.mvacc, (uint64_t)(uint32_t)i0_blk | ((uint64_t)(uint32_t)e2_blk << 32));
            if (!active) info[sl] = DpInfo{steps, bt, bl, bt >= 0 ? best : NEGV};
        }
    }
        if (!__ballot(active)) {      // the group has ended
            if (threadIdx.x == 0) atomicAdd(&ctr->done, 1u);
            break;
        }
        level = level > 1u ? level - 1u : 1u;
        // does a longer chain wait?  (hyst: by how many units longer it has to be)
        c_level = swb_claim(plan, ctr, bag, nk_max, level + 1u + hyst, c_grp, c_fresh);
        if (!c_level) continue;
        {   // park this group at its level
            auto sv = [&](int w, uint32_t v) { __hip_atomic_store(ust + w * 64, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
            auto sv64 = [&](int w, uint64_t v) { sv(w, (uint32_t)v); sv(w + 1, (uint32_t)(v >> 32)); };
            sv64(0, (uint64_t)L.P.v0); sv64(2, (uint64_t)L.P.v1); sv64(4, (uint64_t)L.P.v2); sv64(6, (uint64_t)L.Q.v0); sv64(8, (uint64_t)L.Q.v1); sv64(10, (uint64_t)L.Q.v2);
            sv64(12, (uint64_t)L.R0); sv64(14, (uint64_t)L.R1); sv64(16, (uint64_t)L.C0); sv64(18, (uint64_t)L.C1);
            sv(20, (uint32_t)L.i0); sv(21, (uint32_t)L.E2); sv(22, (uint32_t)L.sv0);
            sv(23, L.down | (L.pdown << 1) | (active ? 4u : 0u) | (row_on ? 8u : 0u) | (col_on ? 16u : 0u));
            sv(24, (uint32_t)Hrow); sv(25, (uint32_t)Hcol); sv(26, (uint32_t)best); sv(27, (uint32_t)bt); sv(28, (uint32_t)bl); sv(29, (uint32_t)steps); sv(30, (uint32_t)t);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (orders the compiler)
            __builtin_amdgcn_s_waitcnt(0);                               // every store of this wave has been written through
            if (threadIdx.x == 0) {
                const uint32_t at = atomicAdd(&ctr->tail[level], 1u);
                __hip_atomic_store(bag + plan->bag_off[level] + at, grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicAdd(&ctr->avail[level], 1);
            }
        }
        break;
    }
    if (wave_log && threadIdx.x == 0) {
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        uint64_t *o = wave_log + 4 * (size_t)atomicAdd(&ctr->n_log, 1u);
        o[0] = t_begin; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = (uint64_t)hw | ((uint64_t)grp << 32); o[3] = (uint64_t)(t - t_first);
    }
  }
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
__device__ __forceinline__ void swb2_step(SwbPair &L, const uint32_t is_hi, const int32_t t, uint2 &rec, const int32_t nq, const int32_t nt, const int32_t max_steps, const bool inner, bool &active,
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
            { const int32_t i = L.i0 + kc, vc = inner ? Hcol - 3 * (nq - 1 - i) : Hcol; if (col_on && kc_in && i >= 0 && i < nq && vc > best) { best = vc; bt = t; bl = kc; } }
            { const int32_t jj = t - (nq - 1), vr = inner ? Hrow - 3 * (nt - 1 - jj) : Hrow; if (row_on && kr_in && jj >= 0 && jj < nt && vr > best) { best = vr; bt = t; bl = kr; } }
            if (L.i0 > nq - 1 || t - (L.i0 + 63) > nt - 1 || t + 1 >= max_steps) { active = false; steps = t + 1; }
        }
    }
    L.pdown = sd;
    L.down = (CHECKED && (t + 1) < 64) ? (uint32_t)(((t + 1) & 1) == 0) : (uint32_t)(L.E2 >= 0);
}

__global__ void __launch_bounds__(256) k_swb2(const uint64_t *__restrict__ n_b_dev, const uint32_t *__restrict__ list, const Slot *__restrict__ slots,
                                              const uint32_t *__restrict__ read_pk, const uint32_t *__restrict__ read_rc, const int64_t *__restrict__ read_woff,
                                              const uint32_t *__restrict__ ctg_pk, const uint32_t *__restrict__ ctg_rc, const int64_t *__restrict__ ctg_woff,
                                              const int64_t *__restrict__ tbo, const int64_t *__restrict__ mvo, uint2 *__restrict__ tb, ulonglong2 *__restrict__ mvw,
                                              DpInfo *__restrict__ info) {
    using namespace swb;
    const uint32_t is_hi = threadIdx.x & 1u;
    const bool lo = is_hi == 0u;
    // workgroups of four waves -- independent of each other, no LDS, no barrier: a workgroup's waves are spread over its CU's four SIMDs, single-wave workgroups are not
    // (625 of them on 256 CUs ran two to a SIMD here and there -- 12.2 ms instead of 7.4 -- and a SIMD shared by two of these waves runs each at little more than half speed)
    const int64_t li = (int64_t)blockIdx.x * (blockDim.x >> 1) + (threadIdx.x >> 1);
    bool active = li < (int64_t)*n_b_dev;
    if (!__ballot(active)) return;
    const uint32_t sl = list[active ? li : 0];      // the pair's slot
    const Slot S = slots[sl];
    uint32_t *tbr = (uint32_t *)(tb + tbo[sl]) + is_hi;   // per step 8 B {D, G} over band lanes 16..47: this lane's 4 of them
    ulonglong2 *mvr = mvw + mvo[sl];
    const int32_t nq = S.nq, nt = S.nt;
    const bool inner = (S.flags & SLOT_INNER) != 0;
    const uint32_t *qpk = ((S.flags & SLOT_QRC) ? read_rc : read_pk) + read_woff[S.read];
    const uint32_t *tpk = ((S.flags & SLOT_TRC) ? ctg_rc : ctg_pk) + ctg_woff[S.ctg];
    const int64_t qb = S.qb, tbase = S.tb;
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
                for (int s8 = 0; s8 < 8; s8++) { swb2_step<false>(L, is_hi, t, rec[s8], nq, nt, max_steps, inner, active, row_on, col_on, Hrow, Hcol, best, bt, bl, steps); t++; }
            } else {
#pragma unroll
                for (int s8 = 0; s8 < 8; s8++) { swb2_step<true>(L, is_hi, t, rec[s8], nq, nt, max_steps, inner, active, row_on, col_on, Hrow, Hcol, best, bt, bl, steps); t++; }
            }
            if (g8 & 1) L.ss.refill();      // (ahead of the stores: see k_swb)
            if (grp_active) {
#pragma unroll
                for (int s8 = 0; s8 < 8; s8++) tbr[2 * (swb_rec(t - 8) + s8)] = (rec[s8].x >> 16) | (rec[s8].y << 16);      // (the low lane holds D, the high lane G, each as {cells 0..31, cells 32..63})
            }
        }
        if (blk_active && lo) {
            mvr[(t - 1) >> 6] = make_ulonglong2(L.mvacc, (uint64_t)(uint32_t)i0_blk);
            if (!active) info[sl] = DpInfo{steps, bt, bl, bt >= 0 ? best : NEGV};
        }
    }
}

// ---- trace-back, part 1: the walk.  One lane per slot, 16 slots per wave.
//
// The walk from a piece's terminal back to its origin is sequential, so a lane owns a slot; what the
// kernel has to do is keep that serial chain short and never make it wait on memory.  (v1.6: a read is walked piece by piece -- every piece its own
// walker, a few thousand steps each -- where v1.5 cut a read's one long walk into speculative segments and stitched them.)
//   * masks are consumed in 64-step chunks (aligned to 64, like the move words).  A step's masks are 128 bits but
//     the path only ever looks at band lanes near its own, so a staged chunk keeps, per step, the 32 bits of D
//     and of G starting at band lane `sh` = clamp(k - 16, 0, 32): 8 B/step, 512 B/chunk, one chunk buffer per
//     slot in LDS.  While the lanes walk chunk c, the 16 B/step records of chunk c-1 are already in flight to
//     registers (one coalesced 1 KB load per slot); they are cut down and parked in LDS when the walk of
//     chunk c is over.  Small LDS footprint = every walker of a launch is resident at once.  A lane whose path left the staged 32 lanes (or whose prefetch was for the wrong
//     chunk) takes a synchronous reload; that is rare.
//   * per step: one LDS read, ~25 VALU ops, no branches.  The moves come from a 64-bit shift register (top bit =
//     move of the step before the current one); the operation of the step (M / I / D) goes into a 2-bit stream,
//     16 ops per word, flushed to HBM when full.  Run-length encoding is k_tb_cigar's job, off this chain.
// HBM traffic: the 16 B/step masks are read once.
constexpr int FAIL_CAP = 8192;                   // slots that may come back from the 8-byte walk per chunk ...
constexpr int64_t FAIL_ROOM = 4ll << 20;         // ... and the DP steps (16-byte records) there is room for
constexpr int TBW_STRIDE = 512 + 8;              // bytes per slot: one 64-step chunk of {D bits, G bits}; +8 staggers LDS banks
#ifndef FZP_TBW_RPW
#define FZP_TBW_RPW 16
#endif
constexpr int TBW_RPW = FZP_TBW_RPW;             // slots walked per wave (build switch: tools/runs/tbw_variants.sh)
constexpr int TBW_WPG = 1;                       // waves per workgroup (four measured: no faster on uniform reads, 20 % slower on reads of real shape)
// LDS traffic of one wave is processed in program order: what the wave's lanes wrote is there for its later reads; only the compiler has to keep the order
#define TBW_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

struct WalkOut { int32_t ok, i, ts, i_end, j_end, ncol, n_ops, pad_; };   // (i_end, j_end) = the terminal; (i, ts - i) = where the walk left the matrix (one of the two is -1)

// FULL = false: the slot's masks are the bit-sliced kernel's 8-byte records (band lanes 16..47 of D and G); a path that leaves those lanes cannot be followed: the slot goes
// on the fail list (WalkOut.ok = 2) and is computed again with whole masks (k_sw) and walked by the FULL form.  (Of 92 000 pieces measured, none left lanes 20..43.)
// FULL = true: 16-byte records (k_sw's), the staged window of 32 lanes follows the path.
// STATS (FZP_TB_STATS, a measurement aid): how far from the band's centre the paths run
template <bool FULL, bool STATS>
__global__ void __launch_bounds__(64 * TBW_WPG) k_tb_walk(const uint32_t *__restrict__ order, const uint64_t *__restrict__ lo_dev, const uint64_t *__restrict__ hi_dev, uint32_t hi_host,
                                                const DpInfo *__restrict__ info, const int64_t *__restrict__ tbo, const int64_t *__restrict__ mvo, const int32_t *__restrict__ tbs,
                                                const void *__restrict__ tb_, const ulonglong2 *__restrict__ mvw, uint32_t *__restrict__ raw,
                                                WalkOut *__restrict__ wout, unsigned long long *__restrict__ stats, uint32_t *__restrict__ fail_list, uint64_t *__restrict__ n_fail,
                                                uint32_t fail_cap, int32_t win_half, int32_t band) {
    // TBW_WPG independent waves per workgroup: every wave has its own slice of the LDS buffer and never waits for another
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[TBW_WPG * TBW_RPW * TBW_STRIDE];
    uint8_t *lds = lds_all + (threadIdx.x >> 6) * (TBW_RPW * TBW_STRIDE);
    const int lane = threadIdx.x & 63;
    // this launch's slots: order[lo .. hi) (the bit-sliced kernel's part of the launch list, the rest of it, or the fail list)
    const int64_t lo = lo_dev ? (int64_t)*lo_dev : 0, hi = hi_dev ? (int64_t)*hi_dev : (int64_t)hi_host;
    const int64_t wq = lo + ((int64_t)blockIdx.x * TBW_WPG + (threadIdx.x >> 6)) * TBW_RPW + lane;
    const bool have = lane < TBW_RPW && wq < hi;
    // `order` = the launch list (slots by decreasing capacity): the 16 walks of a wave are of similar length (a wave lasts as long as its longest) and the longest start first
    const uint32_t sl = have ? order[wq] : 0u;
    DpInfo di = {0, -1, 0, NEGV};
    if (have) di = info[sl];
    bool active = have && di.best_t >= 0;
    int64_t to_ = have ? tbo[sl] : 0, mo_ = have ? mvo[sl] : 0;
    asm volatile("" : "+v"(to_), "+v"(mo_));      // both offsets are in registers from here on: no pending load is attributed to the pointers below
    constexpr int REC = FULL ? 16 : 8;                                        // bytes per step
    const uint8_t *tbr = (const uint8_t *)tb_ + to_ * REC;                    // per step {D mask, G mask}
    const ulonglong2 *mvr = mvw + mo_;                                        // per 64 steps {move bits, i0 before them}
    uint32_t *rawp = raw + 4 * mo_;                                           // 16 ops per word: cap / 16 words per slot
    int32_t ts = active ? di.best_t : -1;
    int32_t k = di.best_lane, i = -1;
    uint64_t w_prev = 0, pref_word = 0;      // move words: (after the first accept) w_cur = chunk of ts, w_prev = the one below
    uint64_t w_cur = 0;
    if (active) {   // i0 at the start step = i0 before its 64-step chunk + DOWN moves up to and including it
        const ulonglong2 mw = mvr[ts >> 6];
        i = (int32_t)(uint32_t)mw.y + __popcll(mw.x & ((2ull << (ts & 63)) - 1ull)) + k;
        w_prev = mw.x;
        pref_word = (ts >> 6) > 0 ? mvr[(ts >> 6) - 1].x : 0ull;
    }
    const int32_t i_end = i, j_end = ts - i;
    active = active && i >= 0 && ts - i >= 0;
    const bool walked = active;
    bool failed = false;
    // v1.8: the band has 64 or 32 cells; the 8-byte records hold lanes rec_lo .. rec_lo + 31 of it -- the middle of a 64-cell band, the whole of a 32-cell one
    const int32_t rec_lo = band == 64 ? 16 : 0;
    const int32_t wlo = band / 2 - win_half;                                 // FULL = false: the path may use band lanes [wlo, wlo + 2 win_half) (all the record holds; tests narrow it)
    const uint32_t wn = 2u * (uint32_t)win_half;
    int32_t ncol = 0, n_ops = 0, nw = 0;
    uint32_t rawacc = 0, nb = 0;
    const int32_t plo = (int32_t)(uint32_t)(uint64_t)tbr, phi = (int32_t)((uint64_t)tbr >> 32);
    const int32_t rstride = have ? tbs[sl] : 64;      // records from one 64-step chunk of the slot's masks to the next
    uint8_t *mine = lds + lane * TBW_STRIDE;
    int32_t cur_chunk = -2, sh_cur = FULL ? 0 : rec_lo;
    uint32_t st_fixed = 0, st_adapt = 0, st_steps = 0, st_maxdev = 0;      // STATS: steps outside lanes [16, 48) / outside the adaptive window
    int32_t st_centre = 32;
    int32_t pref_chunk = active ? ts >> 6 : -1, pref_sh = FULL ? min(max(k - 16, 0), 32) : rec_lo;
    typedef typename std::conditional<FULL, uint4, uint2>::type rec_t;
    rec_t pf[TBW_RPW];
    const int32_t lane_rec = FULL ? lane : (lane & 31) + (lane >> 5) * 2048;      // step `lane` of a 64-step block: whole masks lie step by step, the bit-sliced kernel's in two halves (swb_rec)
#pragma unroll
    for (int l = 0; l < TBW_RPW; l++) memset(&pf[l], 0, sizeof(rec_t));
    uint64_t rec_base[TBW_RPW];      // every slot's mask records: wave-uniform, fetched from the owning lanes once
#pragma unroll
    for (int l = 0; l < TBW_RPW; l++) rec_base[l] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(phi, l) << 32) | (uint32_t)__builtin_amdgcn_readlane(plo, l);
    // records of chunk pref_chunk of every slot -> registers (lane x takes step x of the chunk)
    // (the record pointers are rebuilt from lane reads, which leaves them in the generic address space; a FLAT load counts on lgkmcnt as
    //  well as on vmcnt, so the walk's first wait for an LDS read would wait for the whole prefetch: load through global pointers)
    typedef uint32_t tbw_u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t tbw_u32x2 __attribute__((ext_vector_type(2)));
    typedef const tbw_u32x4 __attribute__((address_space(1))) *tbw_gptr4;
    typedef const tbw_u32x2 __attribute__((address_space(1))) *tbw_gptr2;
#define TBW_ISSUE()                                                                                                      \
    _Pragma("unroll") for (int l = 0; l < TBW_RPW; l++) {                                                                \
        const int32_t cl = __builtin_amdgcn_readlane(pref_chunk, l);                                                     \
        if (cl >= 0) {                                                                                                   \
            const int64_t at_ = (int64_t)cl * __builtin_amdgcn_readlane(rstride, l) + lane_rec;                          \
            if constexpr (FULL) { const tbw_u32x4 q_ = ((tbw_gptr4)rec_base[l])[at_]; pf[l] = make_uint4(q_.x, q_.y, q_.z, q_.w); }   \
            else { const tbw_u32x2 q_ = ((tbw_gptr2)rec_base[l])[at_]; pf[l] = make_uint2(q_.x, q_.y); }               \
        }                                                                                                                \
    }
    auto park = [&](int l, const rec_t &v, int32_t sh) {      // a chunk's records into slot l's LDS buffer: per step the 32 lanes of D and of G the walk will look at
        if constexpr (FULL) {
            const uint64_t D = ((uint64_t)v.y << 32) | v.x, G = ((uint64_t)v.w << 32) | v.z;
            *(uint2 *)(lds + l * TBW_STRIDE + lane * 8) = make_uint2((uint32_t)(D >> sh), (uint32_t)(G >> sh));
        } else *(uint2 *)(lds + l * TBW_STRIDE + lane * 8) = v;
    };
    TBW_ISSUE()
    for (;;) {
        if ((uint32_t)k >= (uint32_t)band) active = false;      // (masks that are not a DP's: a walk that has left the band ends here, marked by i, ts >= 0, instead of spinning on reloads)
        if (!FULL && active && (uint32_t)(k - wlo) >= wn) { active = false; failed = true; }      // the path needs a lane the 8-byte records do not hold
        if (!__any(active)) break;
        const int32_t need = active ? ts >> 6 : -1;
        // park the prefetched chunk (the walk is done with the old contents)
        {
#pragma unroll
            for (int l = 0; l < TBW_RPW; l++) {
                const int32_t cl = __builtin_amdgcn_readlane(pref_chunk, l);
                if (cl >= 0) park(l, pf[l], __builtin_amdgcn_readlane(pref_sh, l));
            }
        }
        const bool ok = need < 0 || (need == pref_chunk && (uint32_t)(k - pref_sh) < 32u);
        if (active && ok) { cur_chunk = pref_chunk; sh_cur = pref_sh; w_cur = w_prev; w_prev = pref_word; }
        const uint64_t redo = __ballot(active && !ok);
        if (redo) {   // rare: the path left the staged lanes, or stalled inside its chunk
            if (active && !ok) {
                cur_chunk = need; sh_cur = FULL ? min(max(k - 16, 0), 32) : rec_lo;
                w_cur = mvr[need].x; w_prev = need > 0 ? mvr[need - 1].x : 0ull;
            }
            for (int l = 0; l < TBW_RPW; l++) {
                if (!((redo >> l) & 1ull)) continue;
                const uint64_t pl = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(phi, l) << 32) | (uint32_t)__builtin_amdgcn_readlane(plo, l);
                const int32_t cl = __builtin_amdgcn_readlane(cur_chunk, l), sh = __builtin_amdgcn_readlane(sh_cur, l);
                const rec_t v = ((const rec_t *)pl)[(int64_t)cl * __builtin_amdgcn_readlane(rstride, l) + lane_rec];
                park(l, v, sh);
            }
        }
        // next prefetch: the chunk below, centred on where the path is now
        pref_chunk = (active && cur_chunk > 0) ? cur_chunk - 1 : -1;
        pref_sh = FULL ? min(max(k - 16, 0), 32) : rec_lo;
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
        if (STATS && active) { const int32_t e2 = (int32_t)(mvr[ts >> 6].y >> 32); st_centre = min(max(32 + e2 / 3, 16), 48); }      // score(lane 63) - score(lane 0) = 2 E2 ~ 6 x (path's lane - 31.5)
        // The loop runs while the walk is inside its matrix (i >= 0, j = ts - i >= 0), inside this chunk (ts >= c_lo: it only goes down) and inside the lanes the staged
        // records hold: one OR of five differences, negative when any of them is -- and the record's LDS address steps down with ts instead of being made from it.
        const int32_t c_lo = cur_chunk << 6;
        const int32_t w_lo = FULL ? sh_cur : wlo, w_hi = FULL ? sh_cur + 31 : wlo + (int32_t)wn - 1;
        const uint8_t *pm = win + (ts & 63) * 8;
        int32_t bad = (active ? 0 : -1) | i | (ts - i) | (ts - c_lo) | (k - w_lo) | (w_hi - k);
        while (bad >= 0) {
            if (STATS) { st_steps++; st_fixed += (uint32_t)(k - 16) >= 32u ? 1u : 0u; st_adapt += (uint32_t)(k - (st_centre - 16)) >= 32u ? 1u : 0u; st_maxdev = max(st_maxdev, (uint32_t)(k >= 32 ? k - 32 : 31 - k)); }
            const uint2 m = *(const uint2 *)pm;
            const uint32_t kk = (uint32_t)(k - sh_cur);
            const uint32_t db = (m.x >> kk) & 1u;
            const uint32_t gx = (m.y >> kk) ^ d1;                             // bit 0: G ^ the move before
            const uint32_t hi = (uint32_t)(P >> 32);
            const uint32_t d2 = hi >> 31, d3 = (hi >> 30) & 1u;
            // not diagonal: G set after a DOWN move, or clear after a RIGHT move -> the predecessor is the cell above
            const uint32_t ndb = db ^ 1u;
            const uint32_t up = ndb & ~gx & 1u;
            const uint32_t op = 2u * ndb - up;                                // M = 0, I = 1, D = 2
            const uint32_t stp = 1u + db, dec = db + up;
            k += (int32_t)(d1 + (db & d2)) - (int32_t)dec;
            i -= (int32_t)dec;
            ts -= (int32_t)stp;
            pm -= 8u * stp;
            d1 = db ? d3 : d2;
            P <<= stp;
            rawacc |= op << nb;
            nb += 2u;
            if (nb == 32u) { rawp[nw++] = rawacc; rawacc = 0u; nb = 0u; }
            bad = i | (ts - i) | (ts - c_lo) | (k - w_lo) | (w_hi - k);
        }
        active = active && (i | (ts - i)) >= 0;
        TBW_WAVE_SYNC();
    }
#undef TBW_ISSUE
    if (!have) return;
    if (STATS && walked && rstride == 4096) {      // (the bit-sliced kernel's slots: the ones whose move words carry E2)
        atomicAdd(&stats[0], 1ull); atomicAdd(&stats[1], st_fixed ? 1ull : 0ull); atomicAdd(&stats[2], st_adapt ? 1ull : 0ull);
        atomicAdd(&stats[3], (unsigned long long)st_steps); atomicAdd(&stats[4], (unsigned long long)st_fixed); atomicAdd(&stats[5], (unsigned long long)st_adapt);
        atomicAdd(&stats[6 + min(st_maxdev >> 1, 9u)], 1ull);      // slots by their largest distance from the band's centre (lanes 31 | 32), in twos
    }
    if (!FULL && failed) {
        const uint32_t f = (uint32_t)atomicAdd((unsigned long long *)n_fail, 1ull);
        if (f < fail_cap) fail_list[f] = sl;
    }
    if (nb) rawp[nw] = rawacc;
    // (neither count is kept in the loop: the op stream knows how many ops went into it, and every op took ts down by one, a diagonal one by one more)
    n_ops = 16 * nw + (int32_t)(nb >> 1);
    ncol = walked ? (di.best_t - ts) - n_ops : 0;
    WalkOut o;
    o.ok = failed ? 2 : (walked ? 1 : 0); o.i = i; o.ts = ts; o.i_end = i_end; o.j_end = j_end; o.ncol = ncol; o.n_ops = n_ops; o.pad_ = 0;
    wout[sl] = o;
}

// ---- the walk of the bit-sliced kernel's slots, r5 form: 32 walkers per wave in step, 32-step staged pieces, 8 KB runs.
// k_tb_walk<false> above keeps 16 of a wave's 64 lanes walking, and what bounds it is what a CU can hold: 520 B of LDS per walker = 256 walkers per CU, whose waves between
// them keep the SIMDs' VALU 88 % busy (r5 counters) with a quarter of the lanes.  Measured on the way here (tools/runs/tbw_variants.sh, tbh_ab.sh, profiles/README.md):
// 32 walkers per wave at the same 256 per CU: the same 3.1 ms; 64 per wave, one wave per SIMD: 3.6 ms; 32 per wave with HALF a 64-step block staged (264 B, 512 walkers
// per CU): 2.7 ms and no longer issue-bound -- bound by HBM, which serves scattered 256-byte pieces at 4.3 TB/s (tools/ubench/rand_block_bw.hip; runs of 1 KB and more: 6.1).
// Hence this form.  The 32 walkers of a wave are 32 neighbours of one launch group, whose mask records the plan lays out [64-step block][half][slot][32 steps] (swb_rec): the
// pieces the wave stages for one half-block are ONE run of 8 KB.  The wave goes down the half-blocks in step -- all pieces end at step 0, a walker joins when the wave reaches
// the half-block its path starts in -- so the 16 loads of an iteration are plain consecutive 512-byte rows off one wave-uniform base, and a walker's lane never has to tell
// anyone where its piece is.  The step itself is k_tb_walk's, with two things taken off its chain: the LDS read (the next step's record is one of the two below the current
// one: both are asked for at the top of a step, one is chosen at its end) and the op stream's store (a piece is at most 32 ops: a 64-bit register takes them, words are cut
// off it once per piece).  WalkOut, the op stream and the fail list are k_tb_walk<false>'s.
constexpr int TBH_RPW = 32;                      // walkers per wave
constexpr int TBH_SUB = 32;                      // steps per staged piece
constexpr int TBH_STRIDE = TBH_SUB * 8 + 8;      // bytes per walker: +8 staggers the banks
__global__ void __launch_bounds__(64) k_tb_walk_h(const uint32_t *__restrict__ order, const uint64_t *__restrict__ hi_dev, const DpInfo *__restrict__ info, const int64_t *__restrict__ tbo,
                                                  const int64_t *__restrict__ mvo, const void *__restrict__ tb_, const ulonglong2 *__restrict__ mvw,
                                                  uint32_t *__restrict__ raw, WalkOut *__restrict__ wout, uint32_t *__restrict__ fail_list, uint64_t *__restrict__ n_fail, uint32_t fail_cap,
                                                  int32_t win_half, int32_t band, uint64_t *__restrict__ wlog) {
    // wlog (builds with -DFZP_TBH_LOG, FZP_TBH_WAVE_LOG=1; tools/runs/tbh_waves.py): per wave {start, end of the 100 MHz counter, shader cycles inside the step loops,
    // shader cycles between "park" and the walk (the prefetch's arrival, the move words, the next prefetch's issue)}
#ifdef FZP_TBH_LOG
    const uint64_t lg_t0 = __builtin_amdgcn_s_memrealtime();
    uint64_t lg_inner = 0, lg_stage = 0, lg_iters = 0;
#define TBH_CLK() __builtin_amdgcn_s_memtime()
#endif
#ifndef FZP_TBH_PAD
#define FZP_TBH_PAD 0      // (measurement builds, tools/runs/tbw_variants.sh: extra LDS per wave = fewer walkers per CU)
#endif
    __shared__ __attribute__((aligned(16))) uint8_t lds_raw[16 + TBH_RPW * TBH_STRIDE + FZP_TBH_PAD];      // (16 bytes in front: the walk asks for the two records below its own, also from a piece's first)
    uint8_t *lds = lds_raw + 16;
    const int lane = threadIdx.x & 63;
    const int64_t hi = (int64_t)*hi_dev;
    const int64_t w0 = (int64_t)blockIdx.x * TBH_RPW;
    if (w0 >= hi) return;                            // (the grid is sized for every slot of the chunk; the bit-sliced kernel's are the first *hi_dev of the launch list)
    const int64_t wq = w0 + lane;
    const bool have = lane < TBH_RPW && wq < hi;
    const uint32_t sl = have ? order[wq] : 0u;
    DpInfo di = {0, -1, 0, NEGV};
    if (have) di = info[sl];
    bool active = have && di.best_t >= 0;
    int64_t mo_ = have ? mvo[sl] : 0;
    // the wave's records: its first walker's (always there) are the run's start, walker l's pieces lie 32 l records on
    const int64_t to0 = tbo[order[w0]];
    const uint8_t *row0 = (const uint8_t *)tb_ + (to0 << 3);
    asm volatile("" : "+v"(mo_));
    const ulonglong2 *mvr = mvw + mo_;                                        // per 64 steps {move bits, i0 before them}
    uint32_t *rawp = raw + 4 * mo_;
    int32_t ts = active ? di.best_t : -1;
    int32_t k = di.best_lane, i = -1;
    uint64_t w_cur = 0, w_prev = 0, pref_word = 0;      // move words: w_cur = 64-step block `wchunk` (the one ts is in once the walker has joined), w_prev the one below, pref_word the one below that
    uint64_t pend_word = 0;                             // ... and the one on its way to become pref_word
    bool pend_take = false, pend_none = true;
    int32_t wchunk = -1;
    if (active) {   // i0 at the start step = i0 before its 64-step block + DOWN moves up to and including it
        const ulonglong2 mw = mvr[ts >> 6];
        i = (int32_t)(uint32_t)mw.y + __popcll(mw.x & ((2ull << (ts & 63)) - 1ull)) + k;
        w_prev = mw.x;                                   // (joining shifts it into w_cur)
        pref_word = (ts >> 6) > 0 ? mvr[(ts >> 6) - 1].x : 0ull;
        wchunk = (ts >> 6) + 1;
    }
    const int32_t i_end = i, j_end = ts - i;
    active = active && i >= 0 && ts - i >= 0;
    const bool walked = active;
    bool failed = false;
    // v1.8: the band has 64 or 32 cells; the records hold lanes rec_lo .. rec_lo + 31 of it -- the middle of a 64-cell band, the WHOLE of a 32-cell one (no path can need more)
    const int32_t rec_lo = band == 64 ? 16 : 0;
    const int32_t wlo = band / 2 - win_half;                                  // the path may use band lanes [wlo, wlo + 2 win_half) (all the record holds; tests narrow it)
    const uint32_t wn = 2u * (uint32_t)win_half;
    int32_t nw = 0;
    uint32_t rawacc = 0, nb = 0;
    const uint8_t *mine = lds + (lane & (TBH_RPW - 1)) * TBH_STRIDE;          // (lanes 32..63 never walk)
    // the half-block the wave starts in: the highest any of its walkers starts in
    int32_t sc = active ? ts >> 5 : -1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) sc = max(sc, __shfl_xor(sc, d, 64));
    sc = __builtin_amdgcn_readfirstlane(sc);
    typedef uint32_t tbh_u32x2 __attribute__((ext_vector_type(2)));
    typedef const tbh_u32x2 __attribute__((address_space(1))) *tbh_gptr2;
    uint2 pf[TBH_RPW / 2];
    uint8_t *park0 = lds + (lane >> 5) * TBH_STRIDE + (lane & 31) * 8;        // load p holds the pieces of walkers 2p (lanes 0..31) and 2p + 1 (lanes 32..63)
    // half-block s of the wave: 256 B per walker, 32 walkers side by side
    // (only the pieces somebody will walk: a walker that has not started yet -- its path begins further down -- or is done asks for nothing.  The wave log said why this
    //  matters: 74 % of a wave's cycles went by between "park" and the walk, waiting for the 8 KB of the iteration -- the walk is bound by what HBM delivers, and part of
    //  what it delivered was pieces of walkers that were not there)
#define TBH_ISSUE(s_)                                                                                                                     \
    {                                                                                                                                     \
        const tbh_gptr2 r_ = (tbh_gptr2)(uint64_t)(row0 + ((int64_t)((s_) >> 1) * 4096 + ((s_) & 1) * 2048) * 8);                          \
        const uint64_t want_ = __ballot(active && (ts >> 5) >= (s_));                                                                     \
        const uint64_t mine_ = want_ >> (lane >> 5);      /* bit 2p: does the walker whose piece this lane fetches in load p want it? */   \
        _Pragma("unroll") for (int p = 0; p < TBH_RPW / 2; p++)                                                                            \
            if ((mine_ >> (2 * p)) & 1ull) { const tbh_u32x2 v_ = r_[64 * p + lane]; pf[p] = make_uint2(v_.x, v_.y); }                       \
    }
    if (sc >= 0) TBH_ISSUE(sc)
    for (; sc >= 0; sc--) {
        if ((uint32_t)k >= (uint32_t)band) active = false;      // (masks that are not a DP's: a walk that has left the band ends here)
        if (active && (uint32_t)(k - wlo) >= wn) { active = false; failed = true; }      // the path needs a lane the 8-byte records do not hold
        if (!__any(active)) break;
        // park the half-block (every walker is done with the old contents)
#ifdef FZP_TBH_LOG
        const uint64_t lg_a = TBH_CLK();
#endif
#pragma unroll
        for (int p = 0; p < TBH_RPW / 2; p++) *(uint2 *)(park0 + 2 * p * TBH_STRIDE) = pf[p];
        if (pend_take) pref_word = pend_none ? 0ull : pend_word;              // (asked for in the last iteration; the parks above have waited for every load of it)
        if (active && (ts >> 5) > sc) { active = false; failed = true; }      // (cannot happen: a step goes down by one or two.  If it does the slot is computed and walked again with whole masks)
        const bool now = active && (ts >> 5) == sc;                           // this walker's path is in the half-block (the others have not started yet)
        pend_take = now && (sc >> 1) != wchunk;
        if (pend_take) { w_cur = w_prev; w_prev = pref_word; wchunk = sc >> 1; }      // it has come down into the next 64-step block
        // The move word two blocks down, for the walkers that have just come down one; looked at in the NEXT iteration (a select right behind the load made the compiler wait
        // for it there, a trip to memory in front of the prefetch's issue).  Only those walkers ask: every lane asking in every iteration was 4 KB of scattered lines per wave
        // and iteration on top of the 8 KB the wave is there for -- and the walk is bound by what HBM delivers (k1_traceback 2.68 -> 2.87 ms)
        if (pend_take) pend_word = mvr[max(wchunk - 2, 0)].x;
        pend_none = wchunk < 2;
        if (sc > 0) TBH_ISSUE(sc - 1)
        TBW_WAVE_SYNC();
        // ---- walk inside the piece
        uint32_t d1 = 0;
        uint64_t P = 0;                    // moves of steps ts-1, ts-2, ... from the top bit down (steps before 0 read as RIGHT)
        if (now) {
            const int32_t s_ = ts & 63;
            d1 = (uint32_t)(w_cur >> s_) & 1u;
            P = s_ ? (w_cur << (64 - s_)) | (w_prev >> s_) : w_prev;
        }
        const int32_t c_lo = sc << 5;
        // (the loop's own variables: kk = k - 16 is the bit the step looks at, p16 the address of the record two below the current one -- both reads of a step hang off it)
        int32_t kk = k - rec_lo;
        const int32_t kw_lo = wlo - rec_lo, kw_hi = wlo + (int32_t)wn - 1 - rec_lo;
        const uint8_t *p16 = mine + (ts & (TBH_SUB - 1)) * 8 - 16;
        int32_t bad = (now ? 0 : -1) | i | (ts - i) | (ts - c_lo) | (kk - kw_lo) | (kw_hi - kk);
        uint64_t acc = 0;                  // this piece's ops, the OLDEST in the top bits (turned round behind the loop)
        uint32_t na = 0;
        uint2 m = *(const uint2 *)(p16 + 16);
#ifdef FZP_TBH_LOG
        const uint64_t lg_b = TBH_CLK();
        lg_stage += lg_b - lg_a; lg_iters++;
#endif
        while (bad >= 0) {
            const uint2 n2 = *(const uint2 *)p16, n1 = *(const uint2 *)(p16 + 8);
            const uint32_t db = (m.x >> kk) & 1u;
            const uint32_t gx = (m.y >> kk) ^ d1;                             // bit 0: G ^ the move before
            const uint32_t hi32 = (uint32_t)(P >> 32);
            const uint32_t d2 = hi32 >> 31;
            const uint32_t ndb = db ^ 1u;
            const uint32_t up = ndb & ~gx & 1u;                               // not diagonal: G set after a DOWN move, or clear after a RIGHT move -> the cell above
            const uint32_t op = 2u * ndb - up;                                // M = 0, I = 1, D = 2
            const uint32_t stp = 1u + db, dec = db + up;
            kk += (int32_t)(d1 + (db & d2)) - (int32_t)dec;
            i -= (int32_t)dec;
            ts -= (int32_t)stp;
            p16 -= 8u * stp;
            d1 = (hi32 << db) >> 31;                                          // the move before the new step: bit 63 of P after a single step, bit 62 after a diagonal one
            P <<= stp;
            acc = (acc << 2) | op;
            na++;
            m = db ? n2 : n1;
            bad = i | (ts - i) | (ts - c_lo) | (kk - kw_lo) | (kw_hi - kk);
        }
        k = kk + rec_lo;
#ifdef FZP_TBH_LOG
        lg_inner += TBH_CLK() - lg_b;
#endif
        if (na) {      // the piece's ops behind the pending bits (nb < 32 of them in rawacc): whole words out, the rest stays pending
            uint64_t r = __builtin_bitreverse64(acc);                         // op g of na -> group 31 - g, its two bits swapped ...
            r = ((r & 0x5555555555555555ull) << 1) | ((r >> 1) & 0x5555555555555555ull);      // ... and swapped back
            const uint64_t ops = r >> (64u - 2u * na);                        // the oldest op in bits 1:0
            const uint64_t lo64 = (uint64_t)rawacc | (ops << nb);
            const uint32_t top = nb ? (uint32_t)(ops >> (64u - nb)) : 0u;
            const uint32_t bits = nb + 2u * na, words = bits >> 5;
            if (words >= 1u) rawp[nw] = (uint32_t)lo64;
            if (words >= 2u) rawp[nw + 1] = (uint32_t)(lo64 >> 32);
            nw += (int32_t)words;
            rawacc = words == 0u ? (uint32_t)lo64 : (words == 1u ? (uint32_t)(lo64 >> 32) : top);
            nb = bits & 31u;
        }
        active = active && (i | (ts - i)) >= 0;
        TBW_WAVE_SYNC();
    }
#undef TBH_ISSUE
#ifdef FZP_TBH_LOG
    if (wlog && lane == 0) { uint64_t *o_ = wlog + 8 * (size_t)blockIdx.x; o_[0] = lg_t0; o_[1] = __builtin_amdgcn_s_memrealtime(); o_[2] = lg_inner; o_[3] = lg_stage; o_[4] = lg_iters; }
#endif
    if (!have) return;
    if (failed) {
        const uint32_t f = (uint32_t)atomicAdd((unsigned long long *)n_fail, 1ull);
        if (f < fail_cap) fail_list[f] = sl;
    }
    if (nb) rawp[nw] = rawacc;
    const int32_t n_ops = 16 * nw + (int32_t)(nb >> 1);
    WalkOut o;
    o.ok = failed ? 2 : (walked ? 1 : 0); o.i = i; o.ts = ts; o.i_end = i_end; o.j_end = j_end; o.ncol = walked ? (di.best_t - ts) - n_ops : 0; o.n_ops = n_ops; o.pad_ = 0;
    wout[sl] = o;
}

// the slots on the fail list get room for whole masks behind the wave-per-slot kernel's own (a bump allocation: which slot lands where is a race, what is computed is not)
__global__ void __launch_bounds__(256) k_fail_plan(const uint32_t *__restrict__ fail_list, const uint64_t *__restrict__ n_fail, uint32_t fail_cap, const Slot *__restrict__ slots,
                                                   unsigned long long *__restrict__ cursor, uint64_t room, int64_t *__restrict__ tbo, int32_t *__restrict__ tbs, uint32_t *__restrict__ overflow) {
    const uint32_t f = blockIdx.x * 256 + threadIdx.x;
    const uint64_t n = *n_fail;
    if (f == 0 && n > fail_cap) atomicOr(overflow, 1u);
    if (f >= n || f >= fail_cap) return;
    const uint32_t sl = fail_list[f];
    const uint64_t cap = (uint64_t)slots[sl].cap;
    const uint64_t at = atomicAdd(cursor, (unsigned long long)cap);
    if (at + cap > room) { atomicOr(overflow, 2u); tbo[sl] = 0; tbs[sl] = 64; return; }      // (the run fails with a message; the walk of this slot stays inside the buffer)
    tbo[sl] = (int64_t)at; tbs[sl] = 64;
}

// ---- trace-back, part 2: one wave per read turns the walk's op stream (alignment end first, 16 ops per word)
// into the forward, run-length encoded CIGAR (M/I/D; '=' / 'X' need the bases and are split on the host where SAM
// text / alnsets are produced -- the phasing stages treat M, = and X alike, phasing.py:81), trims gap runs at both
// ends, adds the soft clips and fills the read's summary.
// A lane owns a word.  Stream position p is forward position L-1-p, so forward runs start where op(p) != op(p+1):
// the flags of a word come from one xor with the stream shifted by one op, and a run is written by the start BELOW
// it (which knows where it ends); suffix scans over the lanes give each word the number of starts and the lowest
// start above it.
__global__ void __launch_bounds__(64) k_tb_cigar(int64_t first, int64_t count, const int32_t *__restrict__ read_len, const ReadPath *__restrict__ rpath,
                                                 const uint32_t *__restrict__ rcapq_scan, uint32_t *__restrict__ raw,
                                                 const int64_t *__restrict__ cig_off, uint32_t *__restrict__ cig,
                                                 int64_t *__restrict__ cig_start, fzp_aln_summary *__restrict__ summ, int match, int mismatch, int gap,
                                                 int min_pct_identity, const uint32_t *__restrict__ read_pk, const uint32_t *__restrict__ read_rc, const int64_t *__restrict__ read_woff,
                                                 const int32_t *__restrict__ read_ctg, const uint32_t *__restrict__ ctg_pk, const int64_t *__restrict__ ctg_woff,
                                                 PkRec *__restrict__ pkrec, int2 *__restrict__ pck, int band) {
    const int lane = lane_id();
    const int64_t wv = blockIdx.x;
    if (wv >= count) return;
    const int64_t r = first + wv;
    const ReadPath rp = rpath[wv];
    // the joined path in the oriented read's / the contig's own coordinates (the anchor offsets of the pieces are folded in by k_join)
    struct { int32_t i_a, c_a, strand; } a = {0, 0, rp.strand};
    WalkOut w;
    w.ok = rp.ok; w.i = 0; w.ts = 0; w.i_end = rp.i_end; w.j_end = rp.j_end; w.ncol = 0; w.n_ops = rp.n_ops; w.pad_ = 0;
    const int32_t n = read_len[r];
    fzp_aln_summary out;
    memset(&out, 0, sizeof out);
    out.cells = rp.steps * band;
    if (lane == 0) cig_start[r] = cig_off[r];
    if (!w.ok) { if (lane == 0) summ[r] = out; return; }
    uint32_t *rg = raw + 4 * (size_t)(rcapq_scan[r] - rcapq_scan[first]);    // the read's op stream (pass 0 drops the ops before the alignment's end from it)
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
    // most 16 read and 16 contig bases going down from the word's first cell, which exclusive scans of the words' consumption counts give.  One sweep over the
    // word's ops (r6; r3-r5: two) leaves its score, the lowest prefix inside it, its highest prefix and its best inner (e, s); a min-scan over the lanes (and the chunks
    // before) gives every word the lowest prefix before it, and the word's best (e, s) follows from those without a second look at the ops.
    int32_t S_star = 0;
    int32_t pM_hi = -1, pM_lo = 0x7fffffff;      // highest / lowest stream position holding an aligned column (of the kept stream: found on the way through it below)
    {
        const uint32_t *qpk = (rp.strand ? read_rc : read_pk) + read_woff[r];
        const uint32_t *tpk = ctg_pk + ctg_woff[read_ctg[r]];
        auto scan_incl = [&](int32_t v) -> int32_t { return (int32_t)wave_incl_scan_u32_dpp((uint32_t)v); };      // (on the DPP network: __shfl_up is a trip through the LDS pipeline per step)
        int32_t base_S = 0, base_i = 0, base_j = 0, bestS = 0, bestP = -1, bestE = 0;
        int32_t base_min = 0, base_min_pos = 0;                  // lowest prefix P(e) over the chunks so far (P(0) = 0 at e = 0), the largest such e
        // r6: ONE sweep over a word's ops, no branches.  (a) Which of its aligned columns match comes from whole-word operations: the 16 read bases and the 16 contig bases
        // going DOWN from the word's first cell (field k = base i - k / j - k), shifted up by two bits per D op / I op before a field -- an I op consumes no contig base, so the
        // contig lags one field further from there on --, six instructions per indel (a word holds two on average), one xor.  (b) The sweep keeps everything in scaled keys
        // (prefix x 256, positions in the low byte) so that "lowest prefix, latest position", "highest prefix, earliest position" and "largest gain, earliest end" are one
        // v_min / v_max each: per op 11 instructions where the two branchy sweeps of r3-r5 spent 80.  (c) What a word needs from outside is the lowest prefix BEFORE it (the
        // min-scan, as before); with G that prefix, B the word's highest prefix-after-an-op and A its best (end - start inside the word), the serial rule "P(s+1) - min(G,
        // lowest prefix inside up to s)" has the value max(A, B - G) -- and its tie rules (smallest s, then largest e; a prefix inside the word wins a tie against G)
        // come out as: A's pair if A > B - G, or if they tie and A's s is not behind B's.  Ops past the stream's end are scored as gaps: prefixes only fall behind the last
        // op, so no position there can win anything (gap > 0).  The scores times 4 096 have to fit the keys: fzp_align_create checks match, mismatch, gap <= 4 096.
        auto desc16 = [&](const uint32_t *pk, int32_t hi) -> uint32_t {      // bases hi, hi - 1, .., hi - 15 in fields 0..15 (a base below 0: 0)
            uint32_t W = 0;
            const int32_t p0 = hi - 15;
            if (hi >= 0) {
                if (p0 >= 0) { const int32_t wl = p0 >> 4; const uint32_t sh = 2u * (uint32_t)(p0 & 15); W = __builtin_amdgcn_alignbit(sh ? pk[wl + 1] : 0u, pk[wl], sh); }
                else W = pk[0] << (2u * (uint32_t)(-p0));
            }
            return cigc::fields_of_reversed_bits(__builtin_bitreverse32(W));     // (field 15 - k with its two bits swapped, put right)
        };
        // the table of the four-op groups (fzp_cigar_core.h: what a lane does with its word is plain C++ there, held against the serial rule on the host by tests/test_cigar_core.py)
        __shared__ cigc::Ent lut[256];
        {
            const cigc::Scores sc = cigc::scores_of(match, mismatch, gap);
            for (int c = lane; c < 256; c += 64) lut[c] = cigc::lut_entry(c, sc);
        }
        __syncthreads();
        // (r6: with a third of the instructions the kernel waits for its loads -- a wave makes ~120 dependent round trips per read; every loop over the stream now asks for
        // the next chunk's words before it works on this one's)
        uint32_t x_next = lane < nW ? rg[lane] : 0u;
        for (int32_t wb = 0; wb < nW; wb += 64) {
            const int32_t wi = wb + lane;
            const uint32_t x = x_next, vm = wi < nW ? valid_mask(wi) : 0u;
            x_next = wi + 64 < nW ? rg[wi + 64] : 0u;
            const uint32_t fM = ~(x | (x >> 1)) & vm, fI = (x & ~(x >> 1)) & vm, fD = (~x & (x >> 1)) & vm;
            const int32_t ci = __popc(fM | fI), cj = __popc(fM | fD);
            const int32_t si = scan_incl(ci), sj = scan_incl(cj);
            const int32_t i = w.i_end - (base_i + si - ci), j = w.j_end - (base_j + sj - cj);       // the cell the word's first op leaves (relative to the anchor: negative in the backward part)
            const uint32_t Dq = ci ? desc16(qpk, a.i_a + i) : 0u, Dt = cj ? desc16(tpk, a.c_a + j) : 0u;
            // a D op leaves its read base to the next op: every field above it takes the field below (a two-bit hole opens at the op); an I op does the same to the contig's
            // bases.  The k-th D and the k-th I of a word in the same turn, as many turns as the wave's busiest word needs (a word that is through changes nothing)
            uint32_t Qs = Dq, Ts = Dt;
            for (uint32_t rd = fD, ri = fI; __builtin_amdgcn_ballot_w64((rd | ri) != 0u);) { cigc::hole_turn(Qs, rd); cigc::hole_turn(Ts, ri); }
            const uint32_t E = Qs ^ Ts;                            // per field: read base xor contig base
            const uint32_t eqw = fM & ~(E | (E >> 1));             // even bit of op o: a matching column
            const uint32_t gapw = fI | fD | (EVEN & ~vm);
            // four ops at a time: the table (filled above, one entry per spelling of four ops) holds what a sweep over them leaves -- their score, the lowest prefix before
            // one of them, the highest prefix behind one, their best inner (e, s) --, and a group joins the word's running values in eight instructions
            const uint32_t cw = eqw | (gapw << 1);                 // per op: 0 a mismatching column, 1 a matching one, 2 a gap
            cigc::Ent ent[4];
#pragma unroll
            for (int g = 0; g < 4; g++) ent[g] = lut[(cw >> (8 * g)) & 255u];
            cigc::Word W = cigc::word_begin();
#pragma unroll
            for (int g = 0; g < 4; g++) cigc::word_join(W, ent[g], g);
            const cigc::WordOut O = cigc::word_end(W);
            const int32_t sl = O.tot, lmin = O.lmin, lpos = O.lpos;
            const int32_t ss = scan_incl(sl);
            const int32_t start = base_S + ss - sl;                // P(16 * wi)
            // lowest prefix over the words up to and including this one: (value, position), later positions win ties
            int32_t mv_ = start + lmin, mp_ = 16 * wi + lpos;
            {   // inclusive (min, position) scan over the lanes on the DPP network: four shifts inside each row of 16, then the row ends passed on (row_bcast 15 / 31); a lane
                // without a source sees (+inf, -): "ov_ < mv_" then never holds.  (r5: twelve __shfl_up per chunk -- LDS-pipeline round trips on the wave's critical path -- gone)
#define MINPOS_STEP(ctrl, row_mask)                                                                                                        \
    { const int32_t ov_ = __builtin_amdgcn_update_dpp(0x3fffffff, mv_, ctrl, row_mask, 0xf, false), op_ = __builtin_amdgcn_update_dpp(0, mp_, ctrl, row_mask, 0xf, false); \
      if (ov_ < mv_) { mv_ = ov_; mp_ = op_; } }
                MINPOS_STEP(0x111, 0xf) MINPOS_STEP(0x112, 0xf) MINPOS_STEP(0x114, 0xf) MINPOS_STEP(0x118, 0xf)      // row_shr 1, 2, 4, 8
                MINPOS_STEP(0x142, 0xa) MINPOS_STEP(0x143, 0xc)                                                        // row_bcast:15 (rows 1, 3), row_bcast:31 (rows 2, 3)
#undef MINPOS_STEP
            }
            int32_t gm = wave_shr1(mv_, 0x3fffffff), gp = wave_shr1(mp_, 0);      // ... before this word (lane 0 takes the chunks before, below)
            if (lane == 0 || base_min < gm) { gm = base_min; gp = base_min_pos; }      // (the chunks before hold earlier positions: they win only when strictly lower)
            {
                const cigc::Pick pick = cigc::word_pick(O, start, gm, gp, wi);
                if (wi < nW && pick.V > bestS) { bestS = pick.V; bestP = pick.s; bestE = pick.e; }      // words ascend within a lane: '>' keeps the smallest s
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
            uint32_t lo_n = lane < nWn ? rg[we + lane] : 0u, hi_n = (lane < nWn && we + lane + 1 < nW) ? rg[we + lane + 1] : 0u;
            for (int32_t wb = 0; wb < nWn; wb += 64) {
                const int32_t wi = wb + lane;
                const uint32_t lo = lo_n, hi = hi_n;
                if (wi + 64 < nWn) { lo_n = rg[we + wi + 64]; hi_n = we + wi + 65 < nW ? rg[we + wi + 65] : 0u; }      // (the next chunk's words, all above what this chunk writes)
                const uint32_t v = sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
                __builtin_amdgcn_wave_barrier();
                if (wi < nWn) rg[wi] = v;
            }
            L = Ln;
        } else L = s0 + 1;
        nW = (L + 15) >> 4;
        // what the kept ops consume, and their aligned columns; on the way every 16th word's place -- {read bases, contig bases} consumed before it -- goes out as
        // a checkpoint of the packed hand-off to K2 (fzp_batch.h: PkSrc; the stream itself stays where it is, in the job's op buffer)
        int32_t ci2 = 0, cj2 = 0, nm2 = 0;
        int2 *ckp = pck + ((size_t)(rcapq_scan[r] >> 2) + (size_t)r);
        __builtin_amdgcn_wave_barrier();
        x_next = lane < nW ? rg[lane] : 0u;
        for (int32_t wb = 0; wb < nW; wb += 64) {
            const int32_t wi = wb + lane;
            uint32_t ci = 0, cj = 0;
            const uint32_t x = x_next;
            x_next = wi + 64 < nW ? rg[wi + 64] : 0u;
            if (wi < nW) {
                const uint32_t vm = valid_mask(wi);
                const uint32_t fM = ~(x | (x >> 1)) & vm, fI = (x & ~(x >> 1)) & vm, fD = (~x & (x >> 1)) & vm;
                ci = (uint32_t)__popc(fM | fI); cj = (uint32_t)__popc(fM | fD); nm2 += __popc(fM);
                // (pass A of r1-r5, a loop of its own: the highest / lowest stream position holding an aligned column)
                if (fM) { pM_hi = max(pM_hi, 16 * wi + ((31 - __builtin_clz(fM)) >> 1)); pM_lo = min(pM_lo, 16 * wi + (__builtin_ctz(fM) >> 1)); }
            }
            const uint32_t si = wave_incl_scan_u32_dpp(ci), sj = wave_incl_scan_u32_dpp(cj);
            if (wi < nW && (wi & 15) == 0) ckp[wi >> 4] = make_int2(ci2 + (int32_t)(si - ci), cj2 + (int32_t)(sj - cj));
            ci2 += __builtin_amdgcn_readlane((int32_t)si, 63); cj2 += __builtin_amdgcn_readlane((int32_t)sj, 63);
        }
        nm2 = wave_sum_i32_dpp(nm2);
        nm2 = __builtin_amdgcn_readfirstlane(nm2);
        w.i = w.i_end - ci2;                      // the cell before the alignment's first op, as the walk would have left it
        w.ts = w.i + (w.j_end - cj2);
        w.ncol = nm2;
        w.n_ops = L;
    }
    pM_hi = wave_max_i32(pM_hi);
    pM_lo = wave_min_i32(pM_lo);
    // pass B: runs, from the top of the stream (= the alignment's start) down
    const int32_t max_runs = n + 16;
    int32_t n_starts = 0;                 // starts seen in higher chunks
    int32_t low_start = 0x7fffffff;       // lowest of them
    int32_t leadI = 0, leadD = 0, trailI = 0, trailD = 0, lead_runs = 0, trail_runs = 0;
    uint32_t xb_n = nW - 1 - lane >= 0 ? rg[nW - 1 - lane] : 0u, nxb_n = (nW - 1 - lane >= 0 && lane > 0) ? rg[nW - lane] : 0u;      // (wi + 1 < nW: every lane but the top word's)
    for (int32_t wtop = nW; wtop > 0; wtop -= 64) {
        const int32_t wi = wtop - 1 - lane;               // lane 0 = highest word of the chunk: "the words above this one" are the lanes below, and the scans over them run
                                                          // as prefix scans on the DPP network (r5: twelve __shfl_down per chunk were LDS-pipeline round trips)
        uint32_t sflag = 0, x = 0, above = 0;
        const uint32_t xb = xb_n, nxb = nxb_n;
        if (wi - 64 >= 0) { xb_n = rg[wi - 64]; nxb_n = rg[wi - 63]; }      // the next chunk down
        if (wi >= 0) {
            x = xb;
            const uint32_t nx = wi + 1 < nW ? nxb : 0u;
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
        // inclusive scans over this word and the words above it (= lanes 0 .. lane): number of starts, lowest start
        const int32_t cs = (int32_t)wave_incl_scan_u32_dpp((uint32_t)cnt);
        int32_t lw = mylow;
#define MIN_STEP(ctrl, row_mask) lw = min(lw, __builtin_amdgcn_update_dpp(0x7fffffff, lw, ctrl, row_mask, 0xf, false));
        MIN_STEP(0x111, 0xf) MIN_STEP(0x112, 0xf) MIN_STEP(0x114, 0xf) MIN_STEP(0x118, 0xf) MIN_STEP(0x142, 0xa) MIN_STEP(0x143, 0xc)
#undef MIN_STEP
        const int32_t tot = __builtin_amdgcn_readlane(cs, 63), chunk_low = __builtin_amdgcn_readlane(lw, 63);
        int32_t k = n_starts + cs - cnt;                                    // starts above this word
        const int32_t lw_up = wave_shr1(lw, 0x7fffffff);                   // lowest start among the words above this one in the chunk (lane 0: none)
        int32_t prevp = min(low_start, lw_up);
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
            pkrec[r] = PkRec{w.i_end, w.j_end, L, a.strand};      // the packed hand-off: the kept stream rg[0 .. L) leaves cell (i_end, j_end)
        }
    }
    summ[r] = out;
}

// ---- candidate selection and joining (v1.6), one wave per read, one lane per slot of the read.  Selection (blasr --bestn 1, unzip.py:86): the candidate whose FORWARD
// pieces' terminal scores sum highest (ties: the first).  Its path, END first: the forward pieces from the last (free) one down to the anchor's -- for an inner piece the gap
// moves from the sub-matrix's corner to its terminal, then the walk, then the gap moves its exit through row / column -1 implies -- and behind them the backward piece: the
// gap moves at the anchor's corner, then its walk turned round (it came from the far end towards the anchor).  k_tb_cigar then sees one path.
constexpr int JOIN_SEGS = 5 * (MAX_WP + 1) + 8;
__global__ void __launch_bounds__(64) k_join(int64_t r_lo, int64_t count, const uint32_t *__restrict__ slot_base, const uint32_t *__restrict__ cnt, uint32_t s_lo,
                                             const Slot *__restrict__ slots, const DpInfo *__restrict__ info, const WalkOut *__restrict__ wout, const int64_t *__restrict__ mvo,
                                             const uint32_t *__restrict__ raw, const uint32_t *__restrict__ rcapq_scan, uint32_t *__restrict__ rraw, ReadPath *__restrict__ rpath) {
    __shared__ int32_t l_eD[64], l_eI[64], l_w[64], l_xD[64], l_xI[64];
    __shared__ int64_t l_src[64];
    __shared__ int32_t g_start[JOIN_SEGS + 1], g_kind[JOIN_SEGS];      // kind: 0 copy, 1 run of I, 2 run of D, 3 copy turned round
    __shared__ int32_t g_len[JOIN_SEGS];
    __shared__ int64_t g_src[JOIN_SEGS];
    __shared__ int32_t g_n;
    const int lane = lane_id();
    const int64_t wv = blockIdx.x;
    if (wv >= count) return;
    const int64_t r = r_lo + wv;
    const int32_t ns = (int32_t)cnt[r];
    const uint32_t base = slot_base[r] - s_lo;
    ReadPath out;
    memset(&out, 0, sizeof out);
    if (ns == 0) { if (lane == 0) rpath[wv] = out; return; }
    const bool have = lane < ns;
    Slot S = {0, 0, 0, 0, 0, 0, 0, 0};
    DpInfo I = {0, -1, 0, NEGV};
    WalkOut W;
    memset(&W, 0, sizeof W);
    if (have) { S = slots[base + lane]; I = info[base + lane]; W = wout[base + lane]; }
    const bool back = have && (S.flags & SLOT_BACK) != 0;
    const int cand = have ? ((S.flags & SLOT_CAND) ? 1 : 0) : -1;
    out.steps = (int64_t)wave_sum_i32_dpp(have ? I.steps : 0);
    const bool f0 = cand == 0 && !back, f1 = cand == 1 && !back;
    const int32_t s0 = wave_sum_i32_dpp(f0 ? I.best_score : 0), s1 = wave_sum_i32_dpp(f1 ? I.best_score : 0);
    const bool bad0 = __any(f0 && I.best_t < 0), bad1 = __any(f1 && I.best_t < 0), has1 = __any(cand == 1);
    const int32_t sc0 = bad0 ? NEGV : s0, sc1 = bad1 ? NEGV : s1;
    const int win = (has1 && sc1 > sc0) ? 1 : 0;
    if (win ? bad1 : bad0) { if (lane == 0) rpath[wv] = out; return; }      // (a piece without a valid border cell: cannot happen for nq, nt >= 1)
    const uint64_t mine = __ballot(cand == win);
    const int w_first = __builtin_ctzll(mine), w_last = 63 - __builtin_clzll(mine);
    const bool has_back = __builtin_amdgcn_readlane((int32_t)(back ? 1 : 0), w_first) != 0;
    {   // what every piece of the winner contributes
        const bool inner = (S.flags & SLOT_INNER) != 0;
        const int32_t is = W.i, js = W.ts - W.i;
        l_eD[lane] = inner ? S.nt - 1 - W.j_end : 0;
        l_eI[lane] = inner ? S.nq - 1 - W.i_end : 0;
        l_w[lane] = (have && I.best_t >= 0 && W.ok) ? W.n_ops : -1;
        l_xD[lane] = (is < 0 && js >= 0) ? js + 1 : 0;
        l_xI[lane] = (js < 0 && is >= 0) ? is + 1 : 0;
        l_src[lane] = have ? 4 * mvo[base + lane] : 0;
    }
    // the path's end = the free piece's terminal, in the oriented read's / the contig's coordinates
    out.strand = (__builtin_amdgcn_readlane(S.flags, w_last) & SLOT_QRC) ? 1 : 0;
    out.i_end = __builtin_amdgcn_readlane(S.qb + W.i_end, w_last);
    out.j_end = __builtin_amdgcn_readlane(S.tb + W.j_end, w_last);
    __syncthreads();
    if (lane == 0) {
        int32_t ng = 0, at = 0;
        auto seg = [&](int kind, int32_t len, int64_t src) { if (len > 0) { g_start[ng] = at; g_len[ng] = len; g_kind[ng] = kind; g_src[ng] = src; ng++; at += len; } };
        for (int x = w_last; x >= w_first + (has_back ? 1 : 0); x--) {
            seg(2, l_eD[x], 0); seg(1, l_eI[x], 0);
            seg(0, l_w[x], l_src[x]);
            seg(2, l_xD[x], 0); seg(1, l_xI[x], 0);
        }
        if (has_back && l_w[w_first] >= 0) {
            seg(2, l_xD[w_first], 0); seg(1, l_xI[w_first], 0);
            seg(3, l_w[w_first], l_src[w_first]);
        }
        g_start[ng] = at;
        g_n = ng;
    }
    __syncthreads();
    const int32_t ng = g_n, L = g_start[ng], nW = (L + 15) >> 4;
    uint32_t *rg = rraw + 4 * (size_t)(rcapq_scan[r] - rcapq_scan[r_lo]);
    for (int32_t wi = lane; wi < nW; wi += 64) {
        const int32_t o0 = 16 * wi, o1 = min(o0 + 16, L);
        int32_t lo = 0, hi = ng - 1;                       // the segment holding op o0
        while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (g_start[mid] <= o0) lo = mid; else hi = mid - 1; }
        int32_t pc = lo, oo = o0;
        uint32_t word = 0;
        while (oo < o1) {
            const int32_t pe = min(o1, g_start[pc + 1]), c = pe - oo, sa = oo - g_start[pc];      // ops [oo, pe) come from segment pc, from its op sa on
            const int kind = g_kind[pc];
            uint32_t bits;
            if (kind == 0) {
                const uint32_t *src = raw + g_src[pc];
                const int32_t sw = sa >> 4, sb = (sa & 15) * 2;
                const uint64_t two = (uint64_t)src[sw] | ((uint64_t)src[sw + 1] << 32);          // (the op buffer has spare words behind every slot's stream)
                bits = (uint32_t)(two >> sb);
            } else if (kind == 3) {
                const uint32_t *src = raw + g_src[pc];
                const int32_t top = g_len[pc] - 1 - sa;                                           // source op of output op oo, going down from there
                bits = 0;
                for (int32_t q = 0; q < c; q++) { const int32_t so = top - q; bits |= ((src[so >> 4] >> (2 * (so & 15))) & 3u) << (2 * q); }
            } else bits = kind == 1 ? 0x55555555u : 0xAAAAAAAAu;
            if (c < 16) bits &= (1u << (2 * c)) - 1u;
            word |= bits << (2 * (oo - o0));
            oo = pe; pc++;
        }
        rg[wi] = word;
    }
    if (lane == 0) { out.ok = 1; out.n_ops = L; rpath[wv] = out; }
}

// ---- gather accepted records into contiguous CIGAR / ASCII SEQ arrays
__global__ void __launch_bounds__(256) k_gather(int64_t n_rec, const int64_t *__restrict__ rec_read, const int64_t *__restrict__ cig_start, const uint32_t *__restrict__ cig,
                                                const int64_t *__restrict__ out_cig_off, uint32_t *__restrict__ out_cig, const uint32_t *__restrict__ read_pk, const uint32_t *__restrict__ read_rc,
                                                const fzp_aln_summary *__restrict__ summ, const int64_t *__restrict__ read_woff, const int64_t *__restrict__ out_seq_off, uint8_t *__restrict__ out_seq) {
    const int64_t k = blockIdx.x;
    if (k >= n_rec) return;
    const int64_t r = rec_read[k];
    const int64_t nc = out_cig_off[k + 1] - out_cig_off[k];
    const uint32_t *src = cig + cig_start[r];
    uint32_t *dst = out_cig + out_cig_off[k];
    for (int64_t x = (int64_t)blockIdx.y * 256 + threadIdx.x; x < nc; x += (int64_t)gridDim.y * 256) dst[x] = src[x];
    const int64_t n = out_seq_off[k + 1] - out_seq_off[k];
    const uint32_t *pk = (summ[r].strand ? read_rc : read_pk) + read_woff[r];      // SEQ on the reference strand
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
                                                   uint32_t *__restrict__ n_aligned, unsigned long long *__restrict__ n_cols, int32_t *__restrict__ max_span, int n_ctg) {
    // n_aligned: [c] aligned reads of contig c, [n_ctg + c] the accepted ones among them (= its records)
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = s < n && key[s] != ~0ull;          // every lane stays to the end: the per-contig totals are reduced over the wave
    int c = -1;
    bool acc = false;
    int32_t pos = -1, ncol = 0, rspan = 0;
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
        pos = sm.pos; ncol = sm.n_columns; rspan = sm.ref_end - sm.pos;
    }
    // per-contig totals: the slots of a wave nearly always belong to one contig -> one atomic per wave, not per read
    // (40 000 same-address atomics issued at once serialise: 0.5 ms)
    const uint64_t vm = __ballot(valid);
    if (vm == 0) return;
    const int c0 = __builtin_amdgcn_readlane(c, __builtin_ctzll(vm));
    const uint64_t am = __ballot(valid && acc);
    if (__all(!valid || c == c0)) {
        int32_t mp = (valid && acc) ? pos : -1, msp = (valid && acc) ? rspan : 0;
        unsigned long long cols = (valid && acc) ? (unsigned long long)ncol : 0ull;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { mp = max(mp, __shfl_xor(mp, d, 64)); msp = max(msp, __shfl_xor(msp, d, 64)); cols += __shfl_xor(cols, d, 64); }
        if (lane_id() == 0) {
            atomicAdd(&n_aligned[c0], (uint32_t)__popcll(vm));
            if (am) atomicAdd(&n_aligned[n_ctg + c0], (uint32_t)__popcll(am));
            if (mp >= 0) atomicMax(&last_pos[c0], mp);
            if (msp > 0) atomicMax(&max_span[c0], msp);
            if (cols) atomicAdd(&n_cols[c0], cols);
        }
    } else if (valid) {
        atomicAdd(&n_aligned[c], 1u);
        if (acc) { atomicAdd(&n_aligned[n_ctg + c], 1u); atomicMax(&last_pos[c], pos); atomicMax(&max_span[c], rspan); atomicAdd(&n_cols[c], (unsigned long long)ncol); }
    }
}
__global__ void __launch_bounds__(256) k_plan_emit(int64_t n, int n_ctg, const int32_t *__restrict__ slot_ctg_of_g, const int64_t *__restrict__ slot_off,
                                                   const uint64_t *__restrict__ v_rec, const uint64_t *__restrict__ v_cig, const uint64_t *__restrict__ v_seq,
                                                   const uint64_t *__restrict__ v_ck, const int32_t *__restrict__ g_read, const int32_t *__restrict__ g_qid,
                                                   const uint8_t *__restrict__ g_acc, const fzp_aln_summary *__restrict__ summ, const uint64_t *__restrict__ totals,
                                                   int64_t *__restrict__ rec_read, int32_t *__restrict__ rec_qid, int32_t *__restrict__ rec_pos, int32_t *__restrict__ rec_ctg,
                                                   int64_t *__restrict__ cig_off, int64_t *__restrict__ seq_off, int64_t *__restrict__ ck_off, int64_t *__restrict__ rec_begin,
                                                   int32_t *__restrict__ rec_span) {
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
    rec_read[k] = r; rec_qid[k] = g_qid[g]; rec_pos[k] = summ[r].pos; rec_ctg[k] = c; rec_span[k] = summ[r].ref_end - summ[r].pos;
    cig_off[k] = (int64_t)v_cig[g]; seq_off[k] = (int64_t)v_seq[g]; ck_off[k] = (int64_t)v_ck[g];
}
// the evaluated prefix of every contig (ref_seq of the phasing batch) from the job's upper-cased contigs into the batch's position layout: one launch for all contigs
// (r5: a hipMemcpyAsync per contig was twenty dispatches of a few microseconds each with a gap behind every one)
__global__ void __launch_bounds__(256) k_copy_ref(const uint8_t *__restrict__ src, const int64_t *__restrict__ aoff, const int64_t *__restrict__ goff, const int32_t *__restrict__ limit,
                                                  uint8_t *__restrict__ dst) {
    const int c = blockIdx.y;
    const int64_t n = limit[c];
    const uint8_t *s = src + aoff[c];          // 16-byte aligned segments on both sides (the positions are laid out tile-aligned)
    uint8_t *d = dst + goff[c];
    const int64_t n16 = n >> 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) ((uint4 *)d)[i] = ((const uint4 *)s)[i];
    if (blockIdx.x == 0) for (int64_t i = (n16 << 4) + threadIdx.x; i < n; i += 256) d[i] = s[i];
}
// ---- a 64-bit fingerprint of every read's CIGAR (checker's aid: tests and bench.py hold all 40 000 reads of the bench workload against the twin's CIGARs, gap placement
// included, without bringing 800 MB of words over): sum over the words of splitmix64(index << 32 | word), wave per read.  Words as the device keeps them: M / I / D / S runs.
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ void __launch_bounds__(256) k_cigar_hash(int64_t n_reads, const fzp_aln_summary *__restrict__ summ, const int64_t *__restrict__ cig_start, const uint32_t *__restrict__ cig,
                                                    uint64_t *__restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_reads) return;
    const int lane = lane_id();
    uint64_t h = 0;
    if (summ[r].aligned) {
        const uint32_t *w = cig + cig_start[r];
        const int32_t n = summ[r].n_cigar;
        for (int32_t k = lane; k < n; k += 64) h += mix64(((uint64_t)(uint32_t)k << 32) | w[k]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) h += __shfl_xor(h, d, 64);
    if (lane == 0) out[r] = h;
}
// the batch path's variant: SEQ segments are padded to 16 bytes, so a thread turns one packed word into one 16-byte store
__global__ void __launch_bounds__(256) k_gather16(int64_t n_rec, const int64_t *__restrict__ rec_read, const int64_t *__restrict__ cig_start, const uint32_t *__restrict__ cig,
                                                  const int64_t *__restrict__ out_cig_off, uint32_t *__restrict__ out_cig, const uint32_t *__restrict__ read_pk, const uint32_t *__restrict__ read_rc,
                                                  const fzp_aln_summary *__restrict__ summ, const int64_t *__restrict__ read_woff, const int64_t *__restrict__ out_seq_off, uint8_t *__restrict__ out_seq) {
    const int64_t k = blockIdx.x;
    if (k >= n_rec) return;
    const int64_t r = rec_read[k];
    const int64_t nc = out_cig_off[k + 1] - out_cig_off[k];
    const uint32_t *src = cig + cig_start[r];
    uint32_t *dst = out_cig + out_cig_off[k];
    for (int64_t x = (int64_t)blockIdx.y * 256 + threadIdx.x; x < nc; x += (int64_t)gridDim.y * 256) dst[x] = src[x];
    const int64_t nw = (out_seq_off[k + 1] - out_seq_off[k]) >> 4;
    const uint32_t *pk = (summ[r].strand ? read_rc : read_pk) + read_woff[r];      // SEQ on the reference strand
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
// land on each other's SIMDs and both run at little more than half speed (HISTORY.md section 14); queued one behind the other each has the chip to itself, and everything else
// of the two jobs still overlaps.  The chain is made of events: a job's DP launches wait for the event the previous job recorded behind its own (no host thread ever blocks).
static std::mutex g_dp_mu;
static std::map<int, hipEvent_t> g_dp_last;

// what one chunk of reads needs between its planning and its trace-back; two sets alternate, so that the DP of chunk k + 1 runs while chunk k is walked
struct ChunkBufs {
    DevBuf<Slot> slots;
    DevBuf<uint32_t> bh, sorted, fits, pos_b, list, lq, lq_scan, gq, gq_scan;
    DevBuf<uint64_t> ptot;                       // [0] slots of the bit-sliced kernel, [1] sum of gq (its mask regions, in units of 4 096 records), [2] sum of lq (cap / 64 over all slots),
                                                 // [3] slots on the fail list, [4] records used in tbw
    DevBuf<int64_t> tbo, mvo;                    // per slot: where its trace-back masks / move words are
    DevBuf<int32_t> tbs;                         // per slot: records from one 64-step block of its masks to the next (4 096: interleaved with its launch group; 64: a stream of its own)
    DevBuf<DpInfo> info;
    DevBuf<WalkOut> wout;
    DevBuf<uint2> tb;                            // the bit-sliced kernel's masks: 8 B per DP step (band lanes 16..47 of D and G)
    DevBuf<ulonglong2> tbw;                      // whole masks, 16 B per DP step: the wave-per-slot kernel's slots, and behind them the slots whose walk needed more than the middle lanes
    DevBuf<uint32_t> fail_list;                  // slots whose 8-byte walk left the recorded lanes (computed again with whole masks)
    DevBuf<ulonglong2> mvw;                      // move words, one per 64 steps
    DevBuf<uint32_t> raw;                        // 2-bit op streams per slot (the walks); the reads' joined streams live with the job (opk)
    DevBuf<ReadPath> rpath;
    // k_swb's work units (r6): the unit plan, per launch group how many of its units are done, and the group's state between two units (8 KB per group)
    DevBuf<SwbUnitPlan> uplan;
    DevBuf<SwbUnitCtr> uctr;
    DevBuf<uint32_t> ubag, ustate;
};

struct fzp_alnjob {
    int32_t n_ctg = 0;
    int64_t n_reads = 0;
    fzp_align_params P;
    std::vector<std::vector<uint8_t>> h_ctg;     // upper-cased ASCII, for the phasing batch's ref_seq
    std::vector<int64_t> h_ctg_len, h_ctg_woff, h_idx_off, h_read_woff, h_cig_off;
    std::vector<int32_t> h_idx_bits, h_read_len, h_read_ctg;
    std::vector<fzp_aln_summary> h_summ;
    int64_t ctg_words = 0, read_words = 0, idx_slots = 0;
    DevBuf<uint32_t> ctg_pk, ctg_rc, read_pk, read_rc, cig;      // 2-bit packed sequences, both orientations (the second one made once, by k_revcomp)
    DevBuf<uint8_t> ctg_ascii;                   // upper-cased contigs, concatenated (ref_seq of the phasing batch)
    std::vector<int64_t> h_ctg_aoff;
    DevBuf<int64_t> ctg_aoff;
    DevBuf<int64_t> ctg_woff, ctg_len, idx_off, read_woff, cig_off, cig_start;
    DevBuf<int32_t> idx_bits, read_len, read_ctg;
    DevBuf<uint64_t> table;
    DevBuf<uint32_t> part_cursor;                // k-mer index build: staged entries per partition
    DevBuf<int32_t> part_ctg, idx_overflow;
    bool index_built = false;
    DevBuf<int64_t> part_off;
    std::vector<int64_t> h_part_off;
    int64_t n_parts = 0;
    DevBuf<Anchor> anc, ancB;                    // first / second candidate per read
    DevBuf<int32_t> n_wp;                        // waypoints per (read, candidate)
    DevBuf<int16_t> wpp;                         // seeding scratch: per (read, window, hit) of a seeding launch, the waypoint before it
    DevBuf<int2> wps;                            // the waypoints (oriented read offset, contig position), MAX_WP per (read, candidate)
    DevBuf<uint2> hits;                          // seeding: HIT_CAP hit slots per read of a seeding launch
    DevBuf<SeedWin> win;
    DevBuf<uint32_t> n_sec;                      // reads with a second candidate
    int64_t n_second = 0;                        // of the last run
    DevBuf<uint32_t> r_cnt, r_capq, slot_base, rcapq_scan;      // per read: slots, their capacity / 64, and the exclusive scans of both
    DevBuf<uint64_t> rtot;                       // [0] slots of the run, [1] capacity / 64 of the run, [2] capacity / 64 and [3] number of the slots the wave-per-slot kernel takes
    DevBuf<uint32_t> fb_overflow;                // the fail list or its mask room overflowed (the run reports it)
    ChunkBufs cb[2];
    // the packed hand-off to K2 (fzp_batch.h: PkSrc), for the whole run: every read's joined op stream at opk + 4 * rcapq_scan[read] (k_join writes it, k_tb_cigar
    // keeps the alignment's stretch of it in place), its PkRec and its 256-op checkpoints
    DevBuf<uint32_t> opk;
    DevBuf<PkRec> pkrec;
    DevBuf<int2> pck;
    DevBuf<unsigned long long> tb_stats;         // FZP_TB_STATS (measurement aid): how often a path leaves a 32-lane window of its band, summed over the job's runs
    DevBuf<uint64_t> wave_log;                   // FZP_SWB_WAVE_LOG (measurement aid): per k_swb workgroup of the last chunk {start, end, hardware id, steps}
    int64_t wave_log_n = 0;
    bool summ_on_host = false;
    bool whole_masks_only = false;               // the second attempt of a run whose fail list overflowed: no bit-sliced kernel, whole masks for every piece
    bool overflow_unchecked = false;             // fzp_align_run_deferred: nobody has asked the device yet whether the fail list overflowed (fzp_align_to_batch does)
    // record planning: reads grouped by contig (input order inside a contig); built on first use
    DevBuf<int32_t> slot_read, slot_ctg;
    DevBuf<int64_t> slot_off;
    std::vector<int64_t> h_slot_off;
    DevBuf<int64_t> rank_bk_off;                 // per contig: first POS bin of the record planning's rank (k_rank_*)
    int64_t n_rank_buckets = 0;
    bool have_slots = false;
    int64_t max_reads_per_ctg = 1;
    hipEvent_t ev_sw[2] = {nullptr, nullptr}, ev_tb[2] = {nullptr, nullptr}, ev_l[2] = {nullptr, nullptr};
    DevBuf<fzp_aln_summary> summ;
    bool done = false;
    std::shared_ptr<fzp_job_life> life = std::make_shared<fzp_job_life>();      // shared with the batches fzp_align_to_batch makes (fzp_batch.h)
};

extern "C" void fzp_align_params_default(fzp_align_params *p) {
    memset(p, 0, sizeof *p);
    p->kmer = 16; p->seed_stride = 4; p->match = 2; p->mismatch = 4; p->gap = 3; p->min_seed_hits = 8;
    p->min_pct_identity = 70;
    p->seed_anchored = 1;
    if (const char *e = getenv("FZP_SEED_ANCHORED")) p->seed_anchored = atoi(e) != 0;      // (A/B runs: v1.6's fixed strides without touching the caller)
    p->band = FZP_DEFAULT_BAND;
    if (const char *e = getenv("FZP_ALIGN_BAND")) { const int g = atoi(e); if (g == 32 || g == 64) p->band = g; }      // (the twin reads the same switch)
}

extern "C" void fzp_align_destroy(fzp_ctx *ctx, fzp_alnjob *job) {
    if (!job) return;
    if (ctx) { (void)fzp_bind(ctx); (void)hipStreamSynchronize(ctx->stream); (void)hipStreamSynchronize(ctx->stream2); (void)hipStreamSynchronize(ctx->stream3); }
    if (job->life->batches.load() > 0) job->life->dead.store(true);      // (a batch made from this job is still open: whatever it is asked next fails instead of reading freed memory)
    for (int k = 0; k < 2; k++) { if (job->ev_sw[k]) (void)hipEventDestroy(job->ev_sw[k]); if (job->ev_tb[k]) (void)hipEventDestroy(job->ev_tb[k]); if (job->ev_l[k]) (void)hipEventDestroy(job->ev_l[k]); }
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
        { const fzp_fill_piece fl[2] = {fzp_zeroes(j->part_cursor, (size_t)j->n_parts), fzp_zeroes(j->idx_overflow, 1)}; FZP_TRY(fzp_fill(ctx, st, fl, 2)); }
        if (P.seed_anchored)
            hipLaunchKernelGGL(k_index_stage_anch, dim3((unsigned)std::max<int64_t>(1, ((lc_max + 15) / 16 + ANCH_WORDS - 1) / ANCH_WORDS), j->n_ctg), dim3(256), 0, st, j->ctg_pk.p, j->ctg_woff.p,
                               j->ctg_len.p, j->idx_off.p, j->idx_bits.p, j->part_off.p, P.kmer, j->table.p, j->part_cursor.p, j->idx_overflow.p);
        else
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

// contigs that lie in a device buffer (dev mode below) into the job's 16-byte aligned segments
__global__ void __launch_bounds__(256) k_copy_spans(const uint8_t *__restrict__ src, const int64_t *__restrict__ be, const int64_t *__restrict__ doff, uint8_t *__restrict__ dst) {
    const int c = blockIdx.y;
    const int64_t b = be[2 * c], n = be[2 * c + 1] - b;
    const uint8_t *s = src + b;
    uint8_t *d = dst + doff[c];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d[i] = s[i];
}

// reads given as spans of one host buffer: read r = buf[read_be[2r], read_be[2r + 1]).  What lies between the spans travels with them (the bytes from the first span's
// begin to the last span's end go to the device in one piece), so the caller can hand over a FASTA file's bytes as they are: headers and line ends stay behind at the pack.
// dev != nullptr (r6, fzp_phase_contigs_files): the bytes are on the device already and so are the spans -- contig c = d_raw[d_ctg_be[2c], d_ctg_be[2c + 1]), read r
// likewise from d_read_be (fzp_fasta.hip found them there); the host brings only the lengths (ctg_len, read_len).  Nothing is uploaded but the small tables.
static int align_create_core(fzp_ctx *ctx, int32_t n_ctg, const uint8_t *const *ctg_seq, const int64_t *ctg_len, int64_t n_reads, const int32_t *read_ctg,
                             const int64_t *read_be, const uint8_t *buf, const fzp_align_params *params, fzp_alnjob **out, const fzp_aln_dev_src *dev, const int64_t *read_len) {
    if (!ctx || !out || n_ctg <= 0 || (!dev && !ctg_seq) || !ctg_len || n_reads < 0 || (n_reads && (!read_ctg || (!dev && (!read_be || !buf)) || (dev && !read_len)))) {
        fzp_set_error("fzp_align_create: bad arguments");
        return FZP_EINVAL;
    }
    *out = nullptr;
    FZP_TRY(fzp_bind(ctx));
    fzp_alnjob *j = new fzp_alnjob();
    if (params) j->P = *params; else fzp_align_params_default(&j->P);
    if (j->P.band == 0) j->P.band = FZP_DEFAULT_BAND;      // (a caller's zeroed struct)
    if (j->P.kmer < 8 || j->P.kmer > 16 || j->P.seed_stride < 1 || j->P.match <= 0 || j->P.mismatch < 0 || j->P.gap <= 0 || j->P.match > 4096 || j->P.mismatch > 4096 || j->P.gap > 4096 || j->P.min_pct_identity < 0 ||
        j->P.min_pct_identity > 100 || (j->P.band != 32 && j->P.band != 64)) {
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
        const int64_t idx_div = j->P.seed_anchored ? ANCH_DIV : CTG_STRIDE;      // expected entries = positions / idx_div
        while ((1LL << bits) < 2 * std::max<int64_t>((nk + idx_div - 1) / idx_div, 1)) bits++;
        j->h_idx_bits.push_back(bits);
        j->h_idx_off.push_back(j->idx_slots);
        j->idx_slots += 1LL << bits;
        coff.push_back(coff.back() + ((ctg_len[c] + 15) & ~15LL));      // 16-byte aligned segments (k_upper works on whole uint4s)
    }
    // reads
    j->h_read_woff.assign(1, 0);
    j->h_cig_off.assign(1, 0);
    int64_t span_lo = INT64_MAX, span_hi = 0;
    for (int64_t r = 0; r < n_reads; r++) {
        int64_t n = dev ? read_len[r] : read_be[2 * r + 1] - read_be[2 * r];
        if (n < 0 || n > 0x3fff0000LL || (!dev && read_be[2 * r] < 0) || read_ctg[r] < 0 || read_ctg[r] >= n_ctg) { delete j; fzp_set_error("read %lld: bad length/contig", (long long)r); return FZP_EINVAL; }
        if (!dev) { span_lo = std::min(span_lo, read_be[2 * r]); span_hi = std::max(span_hi, read_be[2 * r + 1]); }
        if (n * (int64_t)j->P.match >= (1LL << 26) - (1 << 20)) {   // biased score << 5 must fit 32 bits (k_sw best-cell key)
            delete j; fzp_set_error("read %lld: %lld bases x match %d exceeds the score range of the DP kernel", (long long)r, (long long)n, j->P.match); return FZP_EINVAL;
        }
        j->h_read_len.push_back((int32_t)n);
        j->h_read_ctg.push_back(read_ctg[r]);
        j->read_words += ((n + 15) / 16 + 8 + 1) & ~1LL;
        j->h_read_woff.push_back(j->read_words);
        j->h_cig_off.push_back(j->h_cig_off.back() + n + 18);
    }
    DevBuf<uint8_t> d_ascii;
    DevBuf<int64_t> d_off;
    do {
        // contigs: ASCII straight into ctg_ascii (pinned, chunked, threaded staging), upper-cased and packed on the device
        if ((rc = j->ctg_pk.alloc((size_t)j->ctg_words + 8)) || (rc = j->ctg_rc.alloc((size_t)j->ctg_words + 8)) || (rc = j->ctg_ascii.alloc((size_t)coff.back() + 32)) || (rc = d_off.upload(coff.data(), coff.size(), st)) ||
            (rc = j->ctg_woff.upload(j->h_ctg_woff.data(), j->h_ctg_woff.size(), st)) || (rc = j->ctg_len.upload(j->h_ctg_len.data(), j->h_ctg_len.size(), st)) ||
            (rc = j->idx_off.upload(j->h_idx_off.data(), j->h_idx_off.size(), st)) || (rc = j->idx_bits.upload(j->h_idx_bits.data(), j->h_idx_bits.size(), st)))
            break;
        if (dev) hipLaunchKernelGGL(k_copy_spans, dim3(64, (unsigned)n_ctg), dim3(256), 0, st, dev->d_raw, dev->d_ctg_be, (const int64_t *)d_off.p, j->ctg_ascii.p);
        else {
            std::vector<const void *> srcs; std::vector<size_t> dsts, lens;
            for (int c = 0; c < n_ctg; c++) { srcs.push_back(ctg_seq[c]); dsts.push_back((size_t)coff[(size_t)c]); lens.push_back((size_t)ctg_len[c]); }
            if ((rc = fzp_upload_segments(ctx, j->ctg_ascii.p, srcs, dsts, lens, st))) break;
        }
        hipLaunchKernelGGL(k_upper, dim3((unsigned)((coff.back() / 16 + 255) / 256 + 1)), dim3(256), 0, st, j->ctg_ascii.p, coff.back());
        {   // the contigs' own lengths, not the padded segments
            std::vector<int64_t> be;
            for (int c = 0; c < n_ctg; c++) { be.push_back(coff[(size_t)c]); be.push_back(coff[(size_t)c] + ctg_len[c]); }
            DevBuf<int64_t> d_be;
            if ((rc = d_be.upload(be.data(), be.size(), st))) break;
            hipLaunchKernelGGL(k_pack, dim3(n_ctg, 64), dim3(256), 0, st, j->ctg_ascii.p, d_be.p, j->ctg_woff.p, j->ctg_pk.p);
            hipLaunchKernelGGL(k_revcomp<int64_t>, dim3(n_ctg, 64), dim3(256), 0, st, (const uint32_t *)j->ctg_pk.p, (const int64_t *)j->ctg_woff.p, (const int64_t *)j->ctg_len.p, j->ctg_rc.p);
            if (hipStreamSynchronize(st) != hipSuccess) { rc = FZP_EDEVICE; break; }
        }
        j->h_ctg_aoff = coff;
        if ((rc = j->ctg_aoff.upload(coff.data(), coff.size(), st))) break;
        if (hipStreamSynchronize(st) != hipSuccess) { rc = FZP_EDEVICE; break; }
        if (n_reads) {
            const size_t rbytes = dev ? 0 : (size_t)(span_hi - span_lo);
            if ((rc = j->read_pk.alloc((size_t)j->read_words + 8)) || (rc = j->read_rc.alloc((size_t)j->read_words + 8)) || (!dev && (rc = d_ascii.alloc(rbytes + 32))))
                break;
            if (!dev) {
                std::vector<const void *> srcs(1, buf + span_lo); std::vector<size_t> dsts(1, 0), lens(1, rbytes);
                if ((rc = fzp_upload_segments(ctx, d_ascii.p, srcs, dsts, lens, st))) break;
                std::vector<int64_t> rbe((size_t)n_reads * 2);
                for (int64_t r = 0; r < 2 * n_reads; r++) rbe[(size_t)r] = read_be[r] - span_lo;
                if ((rc = d_off.upload(rbe.data(), rbe.size(), st))) break;
            }
            if ((rc = j->read_woff.upload(j->h_read_woff.data(), j->h_read_woff.size(), st)) ||
                (rc = j->read_len.upload(j->h_read_len.data(), j->h_read_len.size(), st)) || (rc = j->read_ctg.upload(j->h_read_ctg.data(), j->h_read_ctg.size(), st)) ||
                (rc = j->cig_off.upload(j->h_cig_off.data(), j->h_cig_off.size(), st)))
                break;
            hipLaunchKernelGGL(k_pack, dim3((unsigned)n_reads, 1), dim3(256), 0, st, dev ? dev->d_raw : (const uint8_t *)d_ascii.p, dev ? dev->d_read_be : (const int64_t *)d_off.p, j->read_woff.p, j->read_pk.p);
            hipLaunchKernelGGL(k_revcomp<int32_t>, dim3((unsigned)n_reads, 1), dim3(256), 0, st, (const uint32_t *)j->read_pk.p, (const int64_t *)j->read_woff.p, (const int32_t *)j->read_len.p, j->read_rc.p);
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
        if ((rc = j->table.alloc((size_t)j->idx_slots)) || (rc = j->anc.alloc((size_t)n_reads)) || (rc = j->ancB.alloc((size_t)n_reads)) || (rc = j->n_sec.alloc(1)) ||
            (rc = j->n_wp.alloc((size_t)2 * n_reads)) || (rc = j->wps.alloc((size_t)2 * n_reads * MAX_WP)) || (rc = j->r_cnt.alloc((size_t)n_reads + 1)) || (rc = j->r_capq.alloc((size_t)n_reads + 1)) ||
            (rc = j->slot_base.alloc((size_t)n_reads + 1)) || (rc = j->rcapq_scan.alloc((size_t)n_reads + 1)) || (rc = j->rtot.alloc(4)) || (rc = j->fb_overflow.alloc(1)) ||
            (rc = j->summ.alloc((size_t)n_reads)) || (rc = j->cig.alloc((size_t)j->h_cig_off.back())) || (rc = j->cig_start.alloc((size_t)n_reads)))
            break;
        if ((rc = build_index(ctx, j))) break;
    } while (0);
    if (rc == FZP_OK && hipGetLastError() != hipSuccess) rc = FZP_EDEVICE;
    if (rc) { if (rc == FZP_EDEVICE) fzp_set_error("fzp_align_create: device error"); delete j; return rc; }
    *out = j;
    return FZP_OK;
}

extern "C" int fzp_align_create_spans(fzp_ctx *ctx, int32_t n_ctg, const uint8_t *const *ctg_seq, const int64_t *ctg_len, int64_t n_reads, const int32_t *read_ctg,
                                      const int64_t *read_be, const uint8_t *buf, const fzp_align_params *params, fzp_alnjob **out) {
    return align_create_core(ctx, n_ctg, ctg_seq, ctg_len, n_reads, read_ctg, read_be, buf, params, out, nullptr, nullptr);
}
int fzp_align_create_dev(fzp_ctx *ctx, int32_t n_ctg, const int64_t *ctg_len, int64_t n_reads, const int32_t *read_ctg, const int64_t *read_len, const fzp_aln_dev_src *dev,
                         const fzp_align_params *params, fzp_alnjob **out) {
    if (!dev || !dev->d_raw || !dev->d_ctg_be || (n_reads && !dev->d_read_be)) { fzp_set_error("fzp_align_create_dev: bad arguments"); return FZP_EINVAL; }
    return align_create_core(ctx, n_ctg, nullptr, ctg_len, n_reads, read_ctg, nullptr, nullptr, params, out, dev, read_len);
}

// the reads back to back in one buffer: read r = read_seq[read_off[r], read_off[r + 1])
extern "C" int fzp_align_create(fzp_ctx *ctx, int32_t n_ctg, const uint8_t *const *ctg_seq, const int64_t *ctg_len, int64_t n_reads, const int32_t *read_ctg,
                                const int64_t *read_off, const uint8_t *read_seq, const fzp_align_params *params, fzp_alnjob **out) {
    if (n_reads < 0 || (n_reads && !read_off)) { fzp_set_error("fzp_align_create: bad arguments"); return FZP_EINVAL; }
    std::vector<int64_t> be((size_t)n_reads * 2);
    for (int64_t r = 0; r < n_reads; r++) { be[(size_t)(2 * r)] = read_off[r]; be[(size_t)(2 * r + 1)] = read_off[r + 1]; }
    return fzp_align_create_spans(ctx, n_ctg, ctg_seq, ctg_len, n_reads, read_ctg, be.data(), read_seq, params, out);
}

// the next fzp_align_run rebuilds the k-mer tables (a job that sees its contigs once pays for them inside its run: bench.py's step)
extern "C" int fzp_align_invalidate_index(fzp_alnjob *j) {
    if (!j) return FZP_EINVAL;
    j->index_built = false;
    return FZP_OK;
}

// defer_overflow (fzp_pipe.hip, through fzp_align_run_deferred): the run's one question to the device that nothing before fzp_align_to_batch needs answered -- "did the
// fail list overflow?" -- rides in that call's fetch instead of costing the step a read-back of its own; fzp_align_to_batch does the retry when the answer is yes.
static int align_run(fzp_ctx *ctx, fzp_alnjob *j, bool defer_overflow);
// (a run rewrites the packed records a batch of this job reads in place: not while one is open)
static int no_open_batch(const fzp_alnjob *j) {
    if (j && j->life->batches.load() > 0) { fzp_set_error("fzp_align_run: a batch made by fzp_align_to_batch from this job is still open -- it reads the job's packed records where they lie; destroy it first"); return FZP_EINVAL; }
    return FZP_OK;
}
extern "C" int fzp_align_run(fzp_ctx *ctx, fzp_alnjob *j) { FZP_TRY(no_open_batch(j)); return align_run(ctx, j, false); }
int fzp_align_run_deferred(fzp_ctx *ctx, fzp_alnjob *j) { FZP_TRY(no_open_batch(j)); return align_run(ctx, j, true); }
static int align_run(fzp_ctx *ctx, fzp_alnjob *j, bool defer_overflow) {
    if (!ctx || !j) return FZP_EINVAL;
    j->overflow_unchecked = false;
    FZP_TRY(fzp_bind(ctx));
    hipStream_t st = ctx->stream, st2 = ctx->stream2, st3 = ctx->stream3;
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
            const size_t lds = ((size_t)nb_max + (P.seed_anchored ? (size_t)(n_max / 16 + 4) : 0)) * sizeof(uint32_t);      // vote bins (two strands x nb_max, 16 bits each) + (v1.7's anchored k-mers only) one count per packed word of the longest read: reads of up to ~480 kb
            if (lds > 150 * 1024) { fzp_set_error("fzp_align_run: a read of %lld bases against a contig of %lld: the seeding kernel's tables (%zu KB) do not fit a CU's LDS", (long long)n_max, (long long)lc_max, lds >> 10); return FZP_EINVAL; }
            FZP_HIP(hipFuncSetAttribute((const void *)k_seed, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            FZP_TRY(j->pkrec.alloc((size_t)nr));      // (the packed hand-off's per-read records: zeroed here, with the counters, instead of by a runtime fill of its own behind the plan's read-back)
            { const fzp_fill_piece fl[4] = {fzp_zeroes(j->n_sec, 1), {j->rtot.p, 32, 0u}, fzp_zeroes(j->fb_overflow, 1), fzp_zeroes(j->pkrec, (size_t)nr)}; FZP_TRY(fzp_fill(ctx, st, fl, 4)); }      // (the plan's counters too: one launch)
            const int64_t seed_chunk = 65536;            // reads per seeding launch: HIT_CAP x 12 B of hit list and waypoint links each (3 GiB)
            FZP_TRY(j->hits.alloc((size_t)std::min<int64_t>(nr, seed_chunk) * HIT_CAP));
            FZP_TRY(j->wpp.alloc((size_t)std::min<int64_t>(nr, seed_chunk) * 2 * HIT_CAP));
            FZP_TRY(j->win.alloc((size_t)std::min<int64_t>(nr, seed_chunk)));
            for (int64_t f0 = 0; f0 < nr; f0 += seed_chunk) {
                const int64_t cn = std::min<int64_t>(seed_chunk, nr - f0);
                hipLaunchKernelGGL(k_seed, dim3((unsigned)cn), dim3(256), lds, st, f0, j->read_pk.p, j->read_woff.p, j->read_len.p,
                                   j->read_ctg.p, j->ctg_len.p, j->idx_off.p, j->idx_bits.p, j->table.p, P.kmer, P.seed_stride, P.min_seed_hits, j->hits.p, j->win.p, P.seed_anchored ? 1 : 0);
                hipLaunchKernelGGL(k_chain, dim3((unsigned)(2 * cn)), dim3(64), 0, st, f0, cn, j->read_len.p, j->hits.p, j->win.p, j->anc.p, j->ancB.p, j->wpp.p, j->n_wp.p, j->wps.p);
            }
        }
        // ---- which DP kernel runs which slot (fzalign scores 2 / -4 / -3 are built into the bit-sliced one's cell function)
        const int band = P.band == 32 ? 32 : 64;      // fzalign v1.8: the band's cells
        auto ksw = band == 32 ? k_sw<true, 32> : k_sw<true, 64>;
        const bool use_bits = getenv("FZP_SW_NO_BITS") == nullptr && P.match == 2 && P.mismatch == 4 && P.gap == 3 && !j->whole_masks_only;
        // the bit-sliced kernel has two forms.  A pair of lanes per slot (k_swb2) has the shorter step but twice the waves, and its instruction mix issues at ~4.5 cycles per
        // SIMD however many waves share it: two such waves on one SIMD run at half speed each.  So it is taken when its waves get a SIMD each; else the whole band sits in one
        // lane (k_swb).  FZP_SWB_64 / FZP_SWB_PAIR force one.
        const int swb_force = getenv("FZP_SWB_64") ? 64 : (getenv("FZP_SWB_PAIR") ? 32 : 0);
        const bool swb_ring = getenv("FZP_SWB_NO_RING") == nullptr;        // k_swb's base streams through LDS rings
        int64_t swb_max_steps = 40960;          // a lane's step costs ~300 ns: a longer extension's chain would outlast the rest of the launch (pieces are a few thousand steps; only reads of > 90 kb have longer ones)
        if (const char *e = getenv("FZP_SWB_MAX_STEPS")) { const long g = atol(e); if (g > 0) swb_max_steps = g; }
        // ---- the extension pieces of every read (v1.6) as DP slots: counted per read on the device, the two scans give every read its first slot and its share of the
        // mask capacity; the scans come to the host (8 bytes per read), which only cuts the reads into chunks that fit the mask budget -- everything else is planned on the device
        {
            ProfScope ps(ctx, "k1_plan_dp");
            hipLaunchKernelGGL(k_slot_count, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, st, nr, j->anc.p, j->ancB.p, j->n_wp.p, j->wps.p, j->read_len.p, j->read_ctg.p, j->ctg_len.p,
                               j->r_cnt.p, j->r_capq.p, j->n_sec.p, (int32_t)swb_max_steps, use_bits ? band : 0, (unsigned long long *)(j->rtot.p + 2));
        }
        FZP_TRY(fzp_exclusive_scan_u32_end(ctx, j->r_cnt.p, j->slot_base.p, (size_t)nr, j->rtot.p + 0));       // (r6: the arrays' closing entries come from the scans -- two 4-byte uploads
        FZP_TRY(fzp_exclusive_scan_u32_end(ctx, j->r_capq.p, j->rcapq_scan.p, (size_t)nr, j->rtot.p + 1));     //  from the stack stood here, each a runtime copy with 20-40 us of idle device around it)
        uint32_t n2 = 0;
        int32_t ovf = 0;
        uint64_t rtot[4] = {0, 0, 0, 0};
        {   // the run's totals: all the host needs unless the run has to be cut into chunks
            const fzp_fetch_piece fp[3] = {{&n2, j->n_sec.p, 4}, {&ovf, j->idx_overflow.p, 4}, {rtot, j->rtot.p, 32}};
            FZP_TRY(fzp_fetch(ctx, st, fp, 3));
        }
        if (ovf) { fzp_set_error("k-mer index: a table partition overflowed (more than %d distinct k-mers hash into one 64 KB partition)", 4 << PART_BITS); return FZP_EINVAL; }
        if (rtot[0] >= (1ull << 31) || rtot[1] >= (1ull << 31)) { fzp_set_error("fzp_align_run: %llu extension pieces / %llu x 64 DP steps in one job (limit 2^31 each)", (unsigned long long)rtot[0], (unsigned long long)rtot[1]); return FZP_EINVAL; }
        const uint32_t end_sb = (uint32_t)rtot[0], end_cq = (uint32_t)rtot[1];
        j->n_second = n2;
        // the run's joined op streams + what K2 needs to read them (the packed hand-off): 16 B per 64 DP steps of capacity, 16 B per read, 8 B per 256 ops
        FZP_TRY(j->opk.alloc((size_t)rtot[1] * 4 + 128));
        FZP_TRY(j->pck.alloc((size_t)(rtot[1] >> 2) + (size_t)nr + 2));
        // Trace-back masks live in HBM (8 B per DP step).  Reads go through in chunks: the DP of chunk k+1 runs on `stream` while the trace-back of chunk k
        // runs on `stream2`; two sets of buffers alternate.
        int64_t budget_steps = (int64_t)48 << 30 >> 3;   // 48 GiB of 8-byte steps over both buffers
        if (const char *e = getenv("FZP_TB_BUDGET_GB")) { long g = atol(e); if (g > 0) budget_steps = ((int64_t)g << 30) >> 3; }
        if (!use_bits) budget_steps /= 2;               // every piece keeps whole masks: 16 B per step
        int n_chunks = 1;
        if (const char *e = getenv("FZP_SW_CHUNKS")) { int g = atoi(e); if (g > 0) n_chunks = g; }
        const int64_t total_steps = (int64_t)rtot[1] * 64;
        const int64_t chunk_steps = std::min<int64_t>(budget_steps / 2, (total_steps + n_chunks - 1) / n_chunks);
        if (!j->ev_sw[0]) for (int k = 0; k < 2; k++) { FZP_HIP(hipEventCreateWithFlags(&j->ev_l[k], hipEventDisableTiming)); FZP_HIP(hipEventCreateWithFlags(&j->ev_sw[k], hipEventDisableTiming)); FZP_HIP(hipEventCreateWithFlags(&j->ev_tb[k], hipEventDisableTiming)); }
        // one chunk for the whole run (the usual case): the scans stay on the device.  Otherwise they come over, to be cut where the mask budget says
        const bool one_chunk = total_steps <= chunk_steps;
        std::vector<uint32_t> h_sb, h_cq;
        if (!one_chunk) {
            h_sb.resize((size_t)nr + 1); h_cq.resize((size_t)nr + 1);
            FZP_TRY(j->slot_base.download(h_sb.data(), (size_t)nr, st));
            FZP_TRY(j->rcapq_scan.download(h_cq.data(), (size_t)nr, st));
            FZP_HIP(hipStreamSynchronize(st));
            h_sb[(size_t)nr] = end_sb; h_cq[(size_t)nr] = end_cq;
        } else { h_sb = {0u, end_sb}; h_cq = {0u, end_cq}; }
        auto sb_at = [&](int64_t r) { return one_chunk ? h_sb[r ? 1 : 0] : h_sb[(size_t)r]; };      // (one chunk: only r = 0 and r = nr are asked for)
        auto cq_at = [&](int64_t r) { return one_chunk ? h_cq[r ? 1 : 0] : h_cq[(size_t)r]; };
        int64_t first = 0;
        int k = 0;
        bool used[2] = {false, false};
        while (first < nr) {
            int64_t last = first;
            if (one_chunk) last = nr;
            else while (last < nr && ((int64_t)h_cq[(size_t)last + 1] - (int64_t)h_cq[(size_t)first]) * 64 <= chunk_steps) last++;
            if (last == first) last = first + 1;
            const int64_t cnt = last - first;
            const uint32_t s_lo = sb_at(first), ns = sb_at(last) - s_lo;
            const int64_t capq = (int64_t)cq_at(last) - (int64_t)cq_at(first);
            const int bi = k & 1;
            ChunkBufs &B = j->cb[bi];
            if (used[bi]) FZP_HIP(hipStreamWaitEvent(st, j->ev_tb[bi], 0));   // buffers free again?
            FZP_TRY(B.rpath.alloc((size_t)cnt));
            // whole masks: the wave-per-slot kernel's slots + room for slots that come back from the 8-byte walk.  (the wave-per-slot slots of a CHUNK never hold more
            // than the chunk's own capacity -- sizing their room by the run's total made a multi-chunk run in whole-mask mode ask for the whole run's masks twice over, exactly
            // where chunking was meant to bound them)
            const int64_t tbw_room = std::min<int64_t>((int64_t)rtot[2], capq) * 64 + FAIL_ROOM;
            uint32_t *const rraw = j->opk.p + 4 * (size_t)cq_at(first);      // this chunk's reads' joined streams inside the job's buffer: read r at + 4 * (rcapq_scan[r] - rcapq_scan[first])
            if (ns > 0) {
                const uint32_t nblk = (ns + 255) / 256, ngrp = (ns + 63) / 64;
                FZP_TRY(B.slots.alloc(ns)); FZP_TRY(B.bh.alloc((size_t)SORT_CLASSES * nblk)); FZP_TRY(B.sorted.alloc(ns)); FZP_TRY(B.fits.alloc(ns)); FZP_TRY(B.pos_b.alloc(ns));
                FZP_TRY(B.list.alloc(ns)); FZP_TRY(B.lq.alloc(ns)); FZP_TRY(B.lq_scan.alloc(ns)); FZP_TRY(B.gq.alloc(ngrp)); FZP_TRY(B.gq_scan.alloc(ngrp)); FZP_TRY(B.ptot.alloc(8));
                FZP_TRY(B.tbo.alloc(ns)); FZP_TRY(B.mvo.alloc(ns)); FZP_TRY(B.tbs.alloc(ns)); FZP_TRY(B.info.alloc(ns)); FZP_TRY(B.wout.alloc(ns));
                // masks: the bit-sliced slots' interleaved groups take 64 x (their longest member) each -- in sorted order at most the slots' own capacity + one group of the longest;
                // whole masks: the wave-per-slot kernel's slots (their capacity over the whole run bounds every chunk's) and room for slots that come back from the 8-byte walk
                const int64_t swb_cap = (swb_max_steps + 2 + 63) / 64 * 64;
                // (the 8-byte buffer is not used at all in whole-mask mode)
                FZP_TRY(B.tb.alloc(use_bits ? (size_t)(capq * 64 + 64 * swb_cap + 64) + 128 : (size_t)128));
                FZP_TRY(B.tbw.alloc((size_t)tbw_room + 64));
                FZP_TRY(B.fail_list.alloc(FAIL_CAP));
                FZP_TRY(B.ptot.alloc(8));
                FZP_TRY(B.mvw.alloc((size_t)capq + 2));
                FZP_TRY(B.raw.alloc((size_t)capq * 4 + 64));
                {
                    ProfScope ps(ctx, "k1_plan_dp");
                    hipLaunchKernelGGL(k_slot_emit, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, first, last, j->anc.p, j->ancB.p, j->n_wp.p, j->wps.p, j->read_len.p, j->read_ctg.p, j->ctg_len.p,
                                       (const uint32_t *)j->slot_base.p, s_lo, B.slots.p);
                    hipLaunchKernelGGL(k_sort_hist, dim3(nblk), dim3(256), 0, st, ns, nblk, (const Slot *)B.slots.p, B.bh.p);
                }
                FZP_TRY(fzp_exclusive_scan_u32(ctx, B.bh.p, B.bh.p, (size_t)SORT_CLASSES * nblk, nullptr));
                {
                    ProfScope ps(ctx, "k1_plan_dp");
                    hipLaunchKernelGGL(k_sort_scatter, dim3(nblk), dim3(256), 0, st, ns, nblk, (const Slot *)B.slots.p, (const uint32_t *)B.bh.p, B.sorted.p);
                    hipLaunchKernelGGL(k_route, dim3(nblk), dim3(256), 0, st, ns, (const Slot *)B.slots.p, (const uint32_t *)B.sorted.p, (int32_t)swb_max_steps, use_bits ? band : 0, B.fits.p);
                }
                FZP_TRY(fzp_exclusive_scan_u32(ctx, B.fits.p, B.pos_b.p, ns, B.ptot.p + 0));
                {
                    ProfScope ps(ctx, "k1_plan_dp");
                    FZP_HIP(hipMemsetAsync(B.gq.p, 0, (size_t)ngrp * 4, st));
                    hipLaunchKernelGGL(k_lists, dim3(nblk), dim3(256), 0, st, ns, (const Slot *)B.slots.p, (const uint32_t *)B.sorted.p, (const uint32_t *)B.fits.p, (const uint32_t *)B.pos_b.p,
                                       (const uint64_t *)B.ptot.p, B.list.p, B.lq.p, B.gq.p);
                }
                FZP_TRY(fzp_exclusive_scan_u32(ctx, B.gq.p, B.gq_scan.p, ngrp, B.ptot.p + 1));
                FZP_TRY(fzp_exclusive_scan_u32(ctx, B.lq.p, B.lq_scan.p, ns, B.ptot.p + 2));
                {
                    ProfScope ps(ctx, "k1_plan_dp");
                    hipLaunchKernelGGL(k_plan_final, dim3(nblk), dim3(256), 0, st, ns, (const uint32_t *)B.list.p, (const uint32_t *)B.lq_scan.p, (const uint32_t *)B.gq_scan.p,
                                       B.ptot.p, B.tbo.p, B.mvo.p, B.tbs.p);
                }
                // One forward DP at a time per device (g_dp_mu / g_dp_last above): a job's DP launches wait for the event the previous job recorded behind its own
                const bool dp_chain = getenv("FZP_DP_NO_CHAIN") == nullptr;
                std::unique_lock<std::mutex> dp_lk(g_dp_mu, std::defer_lock);
                if (dp_chain) {
                    dp_lk.lock();
                    auto it = g_dp_last.find(ctx->device);
                    if (it != g_dp_last.end()) FZP_HIP(hipStreamWaitEvent(st, it->second, 0));
                }
                {
                    ProfScope ps(ctx, "k1_sw");
                    // every slot goes to one of the two DP kernels (k_route): the bit-sliced one (a slot per lane) takes those that span the band on both sides, the
                    // wave-per-slot one the rest -- mostly backward extensions of a few dozen bases -- beside it on a stream of its own
                    const bool swb64 = band == 32 || (swb_force ? swb_force == 64 : !((int64_t)ns / 32 <= (int64_t)ctx->n_cu * 4));      // (the pair-of-lanes form exists for the 64-cell band only: a lane of the 32-cell band already holds one register per plane)
                    // the bit-sliced kernel's two register budgets (FZP_SWB_WAVES = 1 | 2), its work units (FZP_SWB_UNIT: 64-step blocks per unit; FZP_SWB_HYST: by how many
                    // units waiting work has to be longer before a wave parks its group for it) and the SIMD-sharing switch (FZP_SWB_DBG bit 2).  DEFAULT: one wave per SIMD,
                    // whole groups -- r5's schedule, through the same code (one level, every group fresh).  What r6 built and measured (profiles/r6_swb_units.txt): with units
                    // the launch IS uniform (2 047 of 2 048 slots busy until its last 0.6 ms), but the second wave of a SIMD gets 0.42 of the first one's issue rate, two
                    // waves together 1.18 x one wave's throughput, not the 1.4 x the median step time had suggested: two waves + units of 16 blocks + turns at the priority
                    // run the bench step's launch in 5.34 ms against 5.83 in a loop of K1 runs, in 5.9 against 6.0 ms inside the whole step, and in 6.35 against 5.39 ms
                    // on reads of real shape -- so they stay switches.
                    int swb_wps = 1;
                    if (const char *e = getenv("FZP_SWB_WAVES")) { const int g = atoi(e); if (g == 1 || g == 2) swb_wps = g; }
                    unsigned swb_slots = (unsigned)ctx->n_cu * 4u * (unsigned)swb_wps;      // persistent waves: as many as the chip holds of this kernel, no more than there are groups
                    if (const char *e = getenv("FZP_SWB_GRID")) { const long g = atol(e); if (g > 0) swb_slots = (unsigned)std::min<long>(g, swb_slots); }      // (tests: few waves, so that a small job's groups wait for each other)
                    uint32_t ub_req = 1u << 20;
                    if (const char *e = getenv("FZP_SWB_UNIT")) { const long g = atol(e); if (g > 0) ub_req = (uint32_t)std::min<long>(g, 1l << 20); }
                    if (ngrp <= swb_slots) ub_req = 1u << 20;      // (no more groups than slots: every group in one piece -- its waves all start at once, cutting would only add hand-overs)
                    uint32_t swb_hyst = 0;
                    if (const char *e = getenv("FZP_SWB_HYST")) { const long g = atol(e); if (g >= 0 && g < 64) swb_hyst = (uint32_t)g; }
                    const int swb_dbg = getenv("FZP_SWB_DBG") ? atoi(getenv("FZP_SWB_DBG")) : 0;
                    uint64_t *wave_log = nullptr;
                    if (use_bits && swb64) {
                        const size_t n_bag = (size_t)(capq / ub_req) + ngrp + 64;      // every (group, level) pair at most once
                        FZP_TRY(B.uplan.alloc(1)); FZP_TRY(B.uctr.alloc(1)); FZP_TRY(B.ubag.alloc(n_bag)); FZP_TRY(B.ustate.alloc((size_t)ngrp * SWB_STATE_WORDS * 64));
                        { const fzp_fill_piece fl[2] = {{B.uctr.p, sizeof(SwbUnitCtr), 0u}, {B.ubag.p, n_bag * 4, 0xffffffffu}}; FZP_TRY(fzp_fill(ctx, st, fl, 2)); }
                        hipLaunchKernelGGL(k_swb_units, dim3(1), dim3(256), 0, st, (const uint64_t *)B.ptot.p, (const uint32_t *)B.gq.p, ub_req, swb_hyst, B.uplan.p);
                        if (getenv("FZP_SWB_WAVE_LOG")) {
                            const size_t nlog = 2 * n_bag + 64;
                            FZP_TRY(j->wave_log.alloc(4 * nlog)); FZP_TRY(j->wave_log.zero(4 * nlog, st)); wave_log = j->wave_log.p; j->wave_log_n = (int64_t)nlog;
                        }
                    }
                    FZP_HIP(hipEventRecord(j->ev_l[0], st));
                    if (use_bits && !swb64)
                        hipLaunchKernelGGL(k_swb2, dim3((ns + 127) / 128), dim3(256), 0, st, (const uint64_t *)B.ptot.p, (const uint32_t *)B.list.p, (const Slot *)B.slots.p,
                                           (const uint32_t *)j->read_pk.p, (const uint32_t *)j->read_rc.p, (const int64_t *)j->read_woff.p, (const uint32_t *)j->ctg_pk.p, (const uint32_t *)j->ctg_rc.p,
                                           (const int64_t *)j->ctg_woff.p, (const int64_t *)B.tbo.p, (const int64_t *)B.mvo.p, B.tb.p, B.mvw.p, B.info.p);
                    if (use_bits && swb64) {
                        auto kfn = band == 32 ? (swb_wps == 2 ? (swb_ring ? k_swb<true, 32, 8, 4, 2, 32> : k_swb<false, 32, 8, 4, 2, 32>) : (swb_ring ? k_swb<true, 64, 16, 8, 1, 32> : k_swb<false, 64, 16, 8, 1, 32>))
                                              : (swb_wps == 2 ? (swb_ring ? k_swb<true, 32, 8, 4, 2> : k_swb<false, 32, 8, 4, 2>) : (swb_ring ? k_swb<true, 64, 16, 8, 1> : k_swb<false, 64, 16, 8, 1>));
                        hipLaunchKernelGGL(kfn, dim3(std::min<unsigned>(ngrp, swb_slots)), dim3(64), 0, st, (const uint64_t *)B.ptot.p, (const uint32_t *)B.list.p,
                                           (const Slot *)B.slots.p, (const uint32_t *)j->read_pk.p, (const uint32_t *)j->read_rc.p, (const int64_t *)j->read_woff.p, (const uint32_t *)j->ctg_pk.p,
                                           (const uint32_t *)j->ctg_rc.p, (const int64_t *)j->ctg_woff.p, (const int64_t *)B.tbo.p, (const int64_t *)B.mvo.p, B.tb.p, B.mvw.p, B.info.p,
                                           swb_dbg, wave_log, (const SwbUnitPlan *)B.uplan.p, B.uctr.p, B.ubag.p, B.ustate.p);
                    }
                    hipStream_t st_sw = getenv("FZP_SW_SERIAL") ? st : st3;      // (comparison switch: the wave-per-piece kernel behind the bit-sliced one instead of beside it)
                    if (st_sw == st3) FZP_HIP(hipStreamWaitEvent(st3, j->ev_l[0], 0));
                    hipLaunchKernelGGL(ksw, dim3((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(ns, rtot[3]))), dim3(64), 0, st_sw, (const uint64_t *)B.ptot.p, (const uint64_t *)nullptr, ns, /* (no chunk has more such slots than the run) */ (const uint32_t *)B.list.p, (const Slot *)B.slots.p,
                                       (const uint32_t *)j->read_pk.p, (const uint32_t *)j->read_rc.p, (const int64_t *)j->read_woff.p, (const uint32_t *)j->ctg_pk.p, (const uint32_t *)j->ctg_rc.p,
                                       (const int64_t *)j->ctg_woff.p, (const int64_t *)B.tbo.p, (const int64_t *)B.mvo.p, (const int32_t *)B.tbs.p, (uint2 *)B.tbw.p, B.mvw.p, P.match, P.mismatch, P.gap, B.info.p);
                    if (st_sw == st3) { FZP_HIP(hipEventRecord(j->ev_l[1], st3)); FZP_HIP(hipStreamWaitEvent(st, j->ev_l[1], 0)); }
                }
                if (dp_chain) {
                    hipEvent_t &e = g_dp_last[ctx->device];
                    if (!e) FZP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    FZP_HIP(hipEventRecord(e, st));
                    dp_lk.unlock();
                }
            }
            FZP_HIP(hipEventRecord(j->ev_sw[bi], st));
            FZP_HIP(hipStreamWaitEvent(st2, j->ev_sw[bi], 0));
            if (ns > 0) {
                ProfScope ps(ctx, "k1_traceback", st2);
                const bool tb_stats = getenv("FZP_TB_STATS") != nullptr;
                if (tb_stats && !j->tb_stats.p) { FZP_TRY(j->tb_stats.alloc(16)); FZP_TRY(j->tb_stats.zero(16, st2)); }
                int32_t win_half = 16;                                   // test switch: a narrower window sends paths to the fail list that the 8-byte records could follow
                if (const char *e = getenv("FZP_TB_WINDOW")) { const int g = atoi(e); if (g >= 1 && g <= 16) win_half = g; }
                const unsigned wg = (ns + TBW_RPW * TBW_WPG - 1) / (TBW_RPW * TBW_WPG);
                auto kw_mid = tb_stats ? k_tb_walk<false, true> : k_tb_walk<false, false>;
                auto kw_full = k_tb_walk<true, false>;
                // the bit-sliced kernel's slots: 8-byte records; the others: whole masks
                uint64_t *tbh_log = nullptr;
                if (getenv("FZP_TBH_WAVE_LOG")) {      // (measurement aid; the kernel writes it in -DFZP_TBH_LOG builds only)
                    const size_t nwv = (size_t)(ns + TBH_RPW - 1) / TBH_RPW;
                    FZP_TRY(j->wave_log.alloc(8 * nwv + 8)); FZP_TRY(j->wave_log.zero(8 * nwv + 8, st2)); tbh_log = j->wave_log.p; j->wave_log_n = (int64_t)(2 * nwv);
                }
                const bool walk_old = getenv("FZP_TBW_OLD") != nullptr;      // (A/B and the parity test: the 16-walker form on the same records)
                if (tb_stats || walk_old)
                    hipLaunchKernelGGL(kw_mid, dim3(wg), dim3(64 * TBW_WPG), 0, st2, (const uint32_t *)B.list.p, (const uint64_t *)nullptr, (const uint64_t *)B.ptot.p, 0u,
                                       (const DpInfo *)B.info.p, (const int64_t *)B.tbo.p, (const int64_t *)B.mvo.p, (const int32_t *)B.tbs.p, (const void *)B.tb.p, (const ulonglong2 *)B.mvw.p, B.raw.p, B.wout.p,
                                       (unsigned long long *)j->tb_stats.p, B.fail_list.p, B.ptot.p + 3, (uint32_t)FAIL_CAP, win_half, band);
                else
                    hipLaunchKernelGGL(k_tb_walk_h, dim3((ns + TBH_RPW - 1) / TBH_RPW), dim3(64), 0, st2, (const uint32_t *)B.list.p, (const uint64_t *)B.ptot.p, (const DpInfo *)B.info.p,
                                       (const int64_t *)B.tbo.p, (const int64_t *)B.mvo.p, (const void *)B.tb.p, (const ulonglong2 *)B.mvw.p, B.raw.p, B.wout.p,
                                       B.fail_list.p, B.ptot.p + 3, (uint32_t)FAIL_CAP, win_half, band, tbh_log);
                hipLaunchKernelGGL(kw_full, dim3(wg), dim3(64 * TBW_WPG), 0, st2, (const uint32_t *)B.list.p, (const uint64_t *)B.ptot.p, (const uint64_t *)nullptr, ns,
                                   (const DpInfo *)B.info.p, (const int64_t *)B.tbo.p, (const int64_t *)B.mvo.p, (const int32_t *)B.tbs.p, (const void *)B.tbw.p, (const ulonglong2 *)B.mvw.p, B.raw.p, B.wout.p,
                                   (unsigned long long *)nullptr, (uint32_t *)nullptr, (uint64_t *)nullptr, 0u, 16, band);
                // slots whose path left the recorded lanes (normally none: these launches find an empty list): whole masks from the wave-per-slot kernel, walked again
                hipLaunchKernelGGL(k_fail_plan, dim3(FAIL_CAP / 256), dim3(256), 0, st2, (const uint32_t *)B.fail_list.p, (const uint64_t *)(B.ptot.p + 3), (uint32_t)FAIL_CAP, (const Slot *)B.slots.p,
                                   (unsigned long long *)(B.ptot.p + 4), (uint64_t)tbw_room, B.tbo.p, B.tbs.p, j->fb_overflow.p);
                hipLaunchKernelGGL(ksw, dim3(FAIL_CAP), dim3(64), 0, st2, (const uint64_t *)nullptr, (const uint64_t *)(B.ptot.p + 3), (uint32_t)FAIL_CAP, (const uint32_t *)B.fail_list.p, (const Slot *)B.slots.p,
                                   (const uint32_t *)j->read_pk.p, (const uint32_t *)j->read_rc.p, (const int64_t *)j->read_woff.p, (const uint32_t *)j->ctg_pk.p, (const uint32_t *)j->ctg_rc.p,
                                   (const int64_t *)j->ctg_woff.p, (const int64_t *)B.tbo.p, (const int64_t *)B.mvo.p, (const int32_t *)B.tbs.p, (uint2 *)B.tbw.p, B.mvw.p, P.match, P.mismatch, P.gap, B.info.p);
                hipLaunchKernelGGL(kw_full, dim3(FAIL_CAP / (TBW_RPW * TBW_WPG)), dim3(64 * TBW_WPG), 0, st2, (const uint32_t *)B.fail_list.p, (const uint64_t *)nullptr, (const uint64_t *)(B.ptot.p + 3), (uint32_t)FAIL_CAP,
                                   (const DpInfo *)B.info.p, (const int64_t *)B.tbo.p, (const int64_t *)B.mvo.p, (const int32_t *)B.tbs.p, (const void *)B.tbw.p, (const ulonglong2 *)B.mvw.p, B.raw.p, B.wout.p,
                                   (unsigned long long *)nullptr, (uint32_t *)nullptr, (uint64_t *)nullptr, 0u, 16, band);
            }
            {
                ProfScope ps(ctx, "k1_join", st2);
                hipLaunchKernelGGL(k_join, dim3((unsigned)cnt), dim3(64), 0, st2, first, cnt, (const uint32_t *)j->slot_base.p, (const uint32_t *)j->r_cnt.p, s_lo, (const Slot *)B.slots.p,
                                   (const DpInfo *)B.info.p, (const WalkOut *)B.wout.p, (const int64_t *)B.mvo.p, (const uint32_t *)B.raw.p, (const uint32_t *)j->rcapq_scan.p, rraw, B.rpath.p);
            }
            {
                ProfScope ps(ctx, "k1_cigar", st2);
                hipLaunchKernelGGL(k_tb_cigar, dim3((unsigned)cnt), dim3(64), 0, st2, first, cnt, j->read_len.p, (const ReadPath *)B.rpath.p, (const uint32_t *)j->rcapq_scan.p, rraw,
                                   j->cig_off.p, j->cig.p, j->cig_start.p, j->summ.p, P.match, P.mismatch, P.gap, P.min_pct_identity,
                                   (const uint32_t *)j->read_pk.p, (const uint32_t *)j->read_rc.p, (const int64_t *)j->read_woff.p, j->read_ctg.p, (const uint32_t *)j->ctg_pk.p, (const int64_t *)j->ctg_woff.p,
                                   j->pkrec.p, j->pck.p, band);
            }
            FZP_HIP(hipEventRecord(j->ev_tb[bi], st2));
            used[bi] = true;
            first = last;
            k++;
        }
        for (int b2 = 0; b2 < 2; b2++)
            if (used[b2]) FZP_HIP(hipStreamWaitEvent(st, j->ev_tb[b2], 0));   // the main stream continues after all trace-backs
    }
    uint32_t fbo = 0;
    if (nr > 0 && defer_overflow && !j->whole_masks_only) {
        FZP_HIP(hipGetLastError());
        j->overflow_unchecked = true;
        j->summ_on_host = false;
        j->done = true;
        return FZP_OK;
    }
    if (nr > 0) FZP_TRY(fzp_fetch(ctx, st, &fbo, j->fb_overflow.p, 4));
    FZP_HIP(hipStreamSynchronize(st));
    FZP_HIP(hipGetLastError());
    if (fbo) {      // more pieces needed whole trace-back masks than there was room for (FAIL_CAP pieces / FAIL_ROOM steps per chunk): the run again, every piece through the wave-per-piece kernel
        if (j->whole_masks_only) { fzp_set_error("fzp_align_run: the fail list overflowed although nothing is on it in this mode"); return FZP_EINVAL; }
        if (getenv("FZP_TB_NO_RETRY")) { fzp_set_error("fzp_align_run: more than %d extension pieces (or %lld DP steps of them) per chunk needed whole trace-back masks (FZP_TB_NO_RETRY)", FAIL_CAP, (long long)FAIL_ROOM); return FZP_EINVAL; }
        j->whole_masks_only = true;
        const int rc = align_run(ctx, j, false);
        j->whole_masks_only = false;
        return rc;
    }
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
void fzp_align_templates(const fzp_alnjob *j, const uint8_t **ascii, const int64_t **aoff) { *ascii = j->ctg_ascii.p; *aoff = j->ctg_aoff.p; }
// which read belongs to which contig, as the job was created (host copy; fzp_pipe.hip resolves the read-map rows against the contigs' read names while the device aligns)
void fzp_align_host_reads(const fzp_alnjob *j, int32_t *n_ctg, int64_t *n_reads, const int32_t **read_ctg) {
    *n_ctg = (int32_t)j->h_ctg_len.size(); *n_reads = j->n_reads; *read_ctg = j->h_read_ctg.size() == (size_t)j->n_reads ? j->h_read_ctg.data() : nullptr;
}
// measurement aid (tools/runs/tb_window_stats.py): with FZP_TB_STATS set, the walkers' window statistics summed over the job's runs (16 counters)
extern "C" int fzp_debug_tb_stats(fzp_ctx *ctx, fzp_alnjob *j, unsigned long long *out) {
    if (!ctx || !j || !j->tb_stats.p || fzp_bind(ctx) != FZP_OK) return FZP_EINVAL;
    return hipMemcpy(out, j->tb_stats.p, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? FZP_OK : FZP_EDEVICE;
}
// measurement aid (tools/runs/swb_waves.py): with FZP_SWB_WAVE_LOG set, the last run's k_swb workgroups as {start, end (100 MHz ticks), hardware id, steps}; returns their number
extern "C" int64_t fzp_debug_swb_waves(fzp_ctx *ctx, fzp_alnjob *j, uint64_t *out, int64_t cap) {
    if (!ctx || !j || !j->wave_log.p || fzp_bind(ctx) != FZP_OK) return 0;
    const int64_t n = std::min<int64_t>(cap, j->wave_log_n);
    if (n > 0 && (hipMemcpy(out, j->wave_log.p, (size_t)n * 32, hipMemcpyDeviceToHost) != hipSuccess)) return 0;
    return n;
}
// ---- K1's intermediates for the checker (tests/test_gpu_align.py; VERDICT r5: "no test compares the k-mer table or the hit lists with the twin's")
// the index of contig `ctg` as a sorted list of its entries (key << 32 | position << 1 | "the canonical form is the reverse complement"): what the twin's sorted (key, pos)
// table holds -- WHICH slot an entry sits in is a race between inserts (it never decides what a look-up finds), the multiset is not
extern "C" int fzp_debug_index_entries(fzp_ctx *ctx, fzp_alnjob *j, int32_t ctg, uint64_t **entries, int64_t *n) {
    if (!ctx || !j || ctg < 0 || ctg >= j->n_ctg || !entries || !n || !j->index_built) { fzp_set_error("fzp_debug_index_entries: bad arguments (or no index built)"); return FZP_EINVAL; }
    FZP_TRY(fzp_bind(ctx));
    const size_t slots = (size_t)1 << j->h_idx_bits[(size_t)ctg];
    std::vector<uint64_t> t(slots);
    FZP_HIP(hipStreamSynchronize(ctx->stream));
    FZP_HIP(hipMemcpy(t.data(), j->table.p + j->h_idx_off[(size_t)ctg], slots * 8, hipMemcpyDeviceToHost));
    size_t m = 0;
    for (size_t i = 0; i < slots; i++) if (t[i] != EMPTY) t[m++] = t[i];
    std::sort(t.begin(), t.begin() + (long)m);
    uint64_t *o = (uint64_t *)malloc((m ? m : 1) * 8);
    if (!o) return FZP_ENOMEM;
    memcpy(o, t.data(), m * 8);
    *entries = o; *n = (int64_t)m;
    return FZP_OK;
}
__global__ void __launch_bounds__(256) k_table_fingerprint(const uint64_t *__restrict__ table, int64_t slots, unsigned long long *__restrict__ out) {
    unsigned long long cnt = 0, sum = 0, x = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < slots; i += (int64_t)gridDim.x * 256) {
        const uint64_t e = table[i];
        if (e != EMPTY) { const uint64_t h = mix64(e); cnt++; sum += h; x ^= h; }
    }
    for (int d = 32; d >= 1; d >>= 1) { cnt += __shfl_xor(cnt, d, 64); sum += __shfl_xor(sum, d, 64); x ^= __shfl_xor(x, d, 64); }
    if (lane_id() == 0) { atomicAdd(out, cnt); atomicAdd(out + 1, sum); atomicXor(out + 2, x); }
}
// an order-free fingerprint of ALL contigs' tables: {entries, sum and xor of mix64(entry)} -- the same whatever slots the entries landed in
extern "C" int fzp_debug_index_fingerprint(fzp_ctx *ctx, fzp_alnjob *j, uint64_t *out3) {
    if (!ctx || !j || !out3 || !j->index_built) { fzp_set_error("fzp_debug_index_fingerprint: bad arguments (or no index built)"); return FZP_EINVAL; }
    FZP_TRY(fzp_bind(ctx));
    DevBuf<unsigned long long> d;
    FZP_TRY(d.alloc(3)); FZP_TRY(d.zero(3, ctx->stream));
    hipLaunchKernelGGL(k_table_fingerprint, dim3(2048), dim3(256), 0, ctx->stream, (const uint64_t *)j->table.p, j->idx_slots, d.p);
    FZP_HIP(hipMemcpyAsync(out3, d.p, 24, hipMemcpyDeviceToHost, ctx->stream));
    FZP_HIP(hipStreamSynchronize(ctx->stream));
    return FZP_OK;
}
// the k-mer tables again, now (what a run does when the job's index was invalidated; the race between inserts runs again)
extern "C" int fzp_debug_rebuild_index(fzp_ctx *ctx, fzp_alnjob *j) {
    if (!ctx || !j) return FZP_EINVAL;
    FZP_TRY(fzp_bind(ctx));
    FZP_TRY(build_index(ctx, j));
    FZP_HIP(hipStreamSynchronize(ctx->stream));
    return FZP_OK;
}
// the hit list of read `read` as the last run's seeding left it: n_hits pairs (strand << 31 | oriented read offset, contig position), spec order (sample, then position)
extern "C" int fzp_debug_read_hits(fzp_ctx *ctx, fzp_alnjob *j, int64_t read, uint32_t *out, int32_t *n_hits) {
    if (!ctx || !j || !j->done || read < 0 || read >= j->n_reads || !out || !n_hits) { fzp_set_error("fzp_debug_read_hits: bad arguments (run the job first)"); return FZP_EINVAL; }
    if (j->n_reads > 65536) { fzp_set_error("fzp_debug_read_hits: jobs of up to 65 536 reads (one seeding launch: the hit lists of earlier launches are gone)"); return FZP_EINVAL; }
    FZP_TRY(fzp_bind(ctx));
    SeedWin w;
    FZP_HIP(hipStreamSynchronize(ctx->stream));
    FZP_HIP(hipMemcpy(&w, j->win.p + read, sizeof w, hipMemcpyDeviceToHost));
    const int32_t nh = std::min<int32_t>(std::max<int32_t>(w.n_hits, 0), HIT_CAP);
    if (nh) FZP_HIP(hipMemcpy(out, j->hits.p + (size_t)read * HIT_CAP, (size_t)nh * 8, hipMemcpyDeviceToHost));
    *n_hits = nh;
    return FZP_OK;
}

extern "C" int fzp_align_cigar_hashes(fzp_ctx *ctx, fzp_alnjob *j, uint64_t *out) {
    if (!ctx || !j || !j->done || !out) { fzp_set_error("fzp_align_cigar_hashes: run the job first"); return FZP_EINVAL; }
    FZP_TRY(fzp_bind(ctx));
    if (j->n_reads == 0) return FZP_OK;
    DevBuf<uint64_t> d;
    FZP_TRY(d.alloc((size_t)j->n_reads));
    hipLaunchKernelGGL(k_cigar_hash, dim3((unsigned)((j->n_reads + 3) / 4)), dim3(256), 0, ctx->stream, j->n_reads, (const fzp_aln_summary *)j->summ.p, (const int64_t *)j->cig_start.p,
                       (const uint32_t *)j->cig.p, d.p);
    FZP_TRY(d.download(out, (size_t)j->n_reads, ctx->stream));
    FZP_HIP(hipStreamSynchronize(ctx->stream));
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
        hipLaunchKernelGGL(k_gather, dim3((unsigned)nrec, 4), dim3(256), 0, st, nrec, d_rec_read.p, j->cig_start.p, j->cig.p, d_cig_off.p, cigar.p, j->read_pk.p, j->read_rc.p,
                           j->summ.p, j->read_woff.p, d_seq_off.p, seq.p);
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
    DevBuf<int64_t> &rec_read = b->rec_read;      // stays with the batch: K2 reads every record through its read
    const size_t ns = (size_t)std::max<int64_t>(nr, 1);
    FZP_TRY(key.alloc(ns)); FZP_TRY(v_rec.alloc(ns)); FZP_TRY(v_cig.alloc(ns)); FZP_TRY(v_seq.alloc(ns)); FZP_TRY(v_ck.alloc(ns)); FZP_TRY(totals.alloc(4));
    FZP_TRY(g_read.alloc(ns)); FZP_TRY(g_qid.alloc(ns)); FZP_TRY(last_pos.alloc((size_t)nc)); FZP_TRY(n_aligned.alloc((size_t)nc * 2)); FZP_TRY(n_cols.alloc((size_t)nc));
    FZP_TRY(b->ctg_maxspan.alloc((size_t)nc));      // the contigs' longest reference span: from the summaries, no CIGAR pass
    {
        const size_t ns4 = (ns + 3) & ~(size_t)3;
        FZP_TRY(g_acc.alloc(ns4));
        const fzp_fill_piece fl[9] = {fzp_zeroes(v_rec, ns), fzp_zeroes(v_cig, ns), fzp_zeroes(v_seq, ns), fzp_zeroes(v_ck, ns), fzp_zeroes(g_acc, ns4), fzp_zeroes(n_aligned, (size_t)nc * 2),
                                      fzp_zeroes(n_cols, (size_t)nc), fzp_zeroes(b->ctg_maxspan, (size_t)nc), fzp_ones(last_pos, (size_t)nc) /* -1 */};
        FZP_TRY(fzp_fill(ctx, st, fl, 9));      // one launch (nine runtime fills before)
    }
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
                           v_rec.p, v_cig.p, v_seq.p, v_ck.p, g_read.p, g_qid.p, g_acc.p, last_pos.p, n_aligned.p, n_cols.p, b->ctg_maxspan.p, nc);
    }
    FZP_TRY(fzp_exclusive_scan_u64_inplace(ctx, v_rec.p, (size_t)nr, totals.p + 0));
    FZP_TRY(fzp_exclusive_scan_u64_inplace(ctx, v_cig.p, (size_t)nr, totals.p + 1));
    FZP_TRY(fzp_exclusive_scan_u64_inplace(ctx, v_seq.p, (size_t)nr, totals.p + 2));
    FZP_TRY(fzp_exclusive_scan_u64_inplace(ctx, v_ck.p, (size_t)nr, totals.p + 3));
    uint64_t tot[4] = {0, 0, 0, 0};
    std::vector<int32_t> h_last((size_t)nc);
    std::vector<uint32_t> h_nal((size_t)nc * 2);      // aligned reads per contig, then accepted ones (records) per contig
    std::vector<unsigned long long> h_cols((size_t)nc);
    uint32_t fbo = 0;
    const bool ask_overflow = j->overflow_unchecked && nr > 0;
    if (nc <= 32) {      // (a few contigs: everything the host needs here fits one fetch)
        const fzp_fetch_piece fp[5] = {{tot, totals.p, 32}, {h_last.data(), last_pos.p, (size_t)nc * 4}, {h_nal.data(), n_aligned.p, (size_t)nc * 8}, {h_cols.data(), n_cols.p, (size_t)nc * 8},
                                       {&fbo, j->fb_overflow.p, 4}};
        FZP_TRY(fzp_fetch(ctx, st, fp, ask_overflow ? 5 : 4));
    } else {
        if (ask_overflow) FZP_HIP(hipMemcpyAsync(&fbo, j->fb_overflow.p, 4, hipMemcpyDeviceToHost, st));
        FZP_HIP(hipMemcpyAsync(tot, totals.p, 32, hipMemcpyDeviceToHost, st));
        FZP_TRY(last_pos.download(h_last.data(), (size_t)nc, st)); FZP_TRY(n_aligned.download(h_nal.data(), (size_t)nc * 2, st)); FZP_TRY(n_cols.download(h_cols.data(), (size_t)nc, st));
        FZP_HIP(hipStreamSynchronize(st));
    }
    j->overflow_unchecked = false;
    if (fbo) {      // the deferred answer is "it overflowed": the run again with whole masks (fzp_align_run's own retry), then this call from the top
        if (getenv("FZP_TB_NO_RETRY")) { fzp_set_error("fzp_align_run: more than %d extension pieces (or %lld DP steps of them) per chunk needed whole trace-back masks (FZP_TB_NO_RETRY)", FAIL_CAP, (long long)FAIL_ROOM); return FZP_EINVAL; }
        FZP_HIP(hipStreamSynchronize(st));      // (the plan's buffers of this attempt die with this frame)
        j->whole_masks_only = true;
        const int rc = align_run(ctx, j, false);
        j->whole_masks_only = false;
        if (rc != FZP_OK) return rc;
        return fzp_align_to_batch(ctx, j, out);
    }
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
    FZP_TRY(b->rec_span.alloc(nrec1));
    hipLaunchKernelGGL(k_plan_emit, dim3((unsigned)((std::max<int64_t>(nr, nc + 1) + 255) / 256)), dim3(256), 0, st, nr, nc, j->slot_ctg.p, j->slot_off.p, v_rec.p, v_cig.p, v_seq.p, v_ck.p, g_read.p, g_qid.p, g_acc.p, j->summ.p, totals.p,
                       rec_read.p, b->rec_qid.p, b->rec_pos.p, b->rec_ctg.p, b->cig_off.p, b->seq_off.p, b->ck_off.p, b->ctg_rec_begin.p, b->rec_span.p);
    b->h_rec_begin.assign((size_t)nc + 1, 0);      // (what k_plan_emit writes to ctg_rec_begin, from the per-contig record counts: no download, no wait)
    for (int c = 0; c < nc; c++) b->h_rec_begin[(size_t)c + 1] = b->h_rec_begin[(size_t)c] + (int64_t)h_nal[(size_t)nc + c];
    if (b->h_rec_begin[(size_t)nc] != b->n_rec) { fzp_set_error("fzp_align_to_batch: the contigs' records (%lld) do not add up to the batch's (%lld)", (long long)b->h_rec_begin[(size_t)nc], (long long)b->n_rec); return FZP_EDEVICE; }
    // ---- the records stay in K1's own form (r5: the packed hand-off, fzp_batch.h): K2 reads the alignments' 2-bit op streams and the 2-bit reads where K1 left them.  The
    // accepted records' run-length CIGAR words and byte SEQ (k_gather16: 1.1 GB written per bench step, and 0.44 ms of checkpoint pass behind it) are made only when
    // somebody asks (fzp_batch_need_bytes: K6); the job must outlive the batch.
    b->packed = true; b->have_bytes = false;
    b->pk.ops = j->opk.p; b->pk.rcapq_scan = j->rcapq_scan.p; b->pk.prec = j->pkrec.p; b->pk.ck = j->pck.p;
    b->pk.read_pk = j->read_pk.p; b->pk.read_rc = j->read_rc.p; b->pk.read_woff = j->read_woff.p;
    b->make_bytes = [j](fzp_ctx *cx, fzp_batch *bb) -> int {
        FZP_TRY(bb->cigar.alloc((size_t)bb->n_cig)); FZP_TRY(bb->seq.alloc((size_t)bb->n_seq));
        if (bb->n_rec > 0) {
            ProfScope ps(cx, "k1_gather");
            hipLaunchKernelGGL(k_gather16, dim3((unsigned)bb->n_rec, 4), dim3(256), 0, cx->stream, bb->n_rec, bb->rec_read.p, j->cig_start.p, j->cig.p, bb->cig_off.p, bb->cigar.p, j->read_pk.p, j->read_rc.p,
                               j->summ.p, j->read_woff.p, bb->seq_off.p, bb->seq.p);
        }
        FZP_HIP(hipGetLastError());
        return FZP_OK;
    };
    FZP_TRY(b->ref.alloc((size_t)b->n_pos + 16));
    FZP_TRY(b->ctg_goff.upload(b->h_goff.data(), b->h_goff.size(), st));
    FZP_TRY(b->ctg_qoff.upload(b->h_qid_off.data(), b->h_qid_off.size(), st));
    FZP_TRY(b->ctg_limit.upload(b->h_limit.data(), b->h_limit.size(), st));
    if (b->n_pos > 0) hipLaunchKernelGGL(k_copy_ref, dim3(64, (unsigned)nc), dim3(256), 0, st, (const uint8_t *)j->ctg_ascii.p, (const int64_t *)j->ctg_aoff.p, (const int64_t *)b->ctg_goff.p,
                                         (const int32_t *)b->ctg_limit.p, b->ref.p);      // evaluated prefix of every contig, device to device
    FZP_HIP(hipGetLastError());      // (no wait here: what was uploaded lives in the batch, and whoever runs the batch next is on the same stream)
    b->have_aln = true;
    b->life = j->life;
    j->life->batches.fetch_add(1);
    guard.p = nullptr;
    *out = b;
    return FZP_OK;
}
