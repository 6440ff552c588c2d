// fzp_batch.h -- device-resident state of a batch of contigs going through K2..K5.
//
// HBM layout (all struct-of-arrays, contigs concatenated):
//   records      rec_pos/rec_qid/rec_ctg [n_rec], cig_off/seq_off [n_rec+1], cigar (u32 len<<4|op), seq (u8)
//   positions    one global index g = ctg_goff[c] + p for every EVALUATED contig position p < limit[c]
//                (limit = POS of the last accepted record: later positions are never evaluated,
//                phasing.py:98-102).  cnt[4g..4g+3] = A,C,G,T column counts, oth[g] = non-ACGT symbol
//                tracker, flag8[g] = het call, site_idx[g] / row_off[g] = exclusive scans.
//   sites        fzp_site[n_sites] ascending g; site_g[s]; site_begin[c]
//   variant_map  vmap_qid[n_rows] grouped per site (major allele rows, then minor), record order
//   sets         setq[n_rows]: per site the two alleles' DISTINCT q_ids, ascending, in A<C<T<G allele
//                order: allele x=0 at row_off, x=1 at row_off + count[allele of x=0]; set_n[2s+x]
//   atable       fzp_arow[n_arows] (global site indices on the device), arow_begin[c]
//   blocks       fzp_pvar[n_pvars], pvar_begin[c]; site_blk[s], site_b1[s]
//   reads        fzp_pread[n_preads], pread_begin[c]; q indices are global: qid_off[c] + q_id
#pragma once
#include <functional>

#include <atomic>
#include <memory>
#include "fzp_common.h"

// ---- the packed hand-off K1 -> K2 (r5; unzip.py:86-91 <-> phasing.py:27,42-96 without the BAM, the SAM text, the byte SEQ or the run-length CIGAR in between).
// K1 leaves, per aligned read, the alignment's joined 2-bit op stream (0 = aligned column, 1 = inserted read base, 2 = deleted contig base; 16 ops per word, the
// alignment's END first) at ops + 4 * rcapq_scan[read], a PkRec -- the cell the stream's first op leaves, in the oriented read's / the contig's own coordinates -- and
// per 16 words (256 ops) a checkpoint {read bases, contig bases consumed before it} at ck + (rcapq_scan[read] >> 2) + read.  An aligned column of the CIGAR walk
// (phasing.py:77-96) is an op 0 at cell (i, j): reference position j, symbol = base i of the oriented 2-bit read.  K2 reads exactly that.
struct PkRec { int32_t i_end, j_end, n_ops, strand; };
struct PkSrc {
    const uint32_t *ops = nullptr, *rcapq_scan = nullptr;
    const PkRec *prec = nullptr;
    const int2 *ck = nullptr;
    const uint32_t *read_pk = nullptr, *read_rc = nullptr;
    const int64_t *read_woff = nullptr;
};

// K2's position tiles: every contig's evaluated range starts on a tile boundary of the global position index (ctg_goff is a multiple of
// FZP_POS_TILE), so a tile is exactly FZP_POS_TILE / 256 of the 256-position blocks the call / compaction kernels work on, and a tile
// that no record overlaps is never written, read or scanned (blk_live)
constexpr int FZP_POS_TILE = 2048;
inline int64_t fzp_pos_pad(int64_t limit) { return (limit + FZP_POS_TILE - 1) / FZP_POS_TILE * FZP_POS_TILE; }

// A batch made by fzp_align_to_batch BORROWS device memory of its alnjob (PkSrc below, and make_bytes): the job has to stay as it is while the batch lives.  The two share
// this record: fzp_align_run refuses while a batch is open, fzp_align_destroy marks the record dead, and every entry point that reads the borrowed memory asks
// fzp_batch_source_ok first (the rule was neither written down nor enforced).
struct fzp_job_life { std::atomic<int> batches{0}; std::atomic<bool> dead{false}; };

struct fzp_batch {
    int32_t n_ctg = 0;
    std::shared_ptr<fzp_job_life> life;      // set by fzp_align_to_batch
    // host mirrors
    std::vector<int64_t> h_rec_begin, h_goff, h_qid_off, h_ref_len;
    std::vector<int32_t> h_limit;
    std::vector<int64_t> h_site_begin, h_arow_begin, h_pvar_begin, h_pread_begin;
    int64_t n_rec = 0, n_cig = 0, n_seq = 0, n_pos = 0, n_columns = 0, n_qid = 0;
    int64_t n_eval = 0;        // evaluated positions (sum of the contigs' limits); n_pos counts the tile-aligned layout
    int64_t n_sites = 0, n_rows = 0, n_arows = 0, n_pvars = 0, n_preads = 0;
    bool have_aln = false, have_sites = false, have_sets = false, have_arows = false, have_blocks = false,
         have_preads = false;
    // inputs
    DevBuf<int32_t> rec_pos, rec_qid, rec_ctg;
    DevBuf<int64_t> cig_off, seq_off;
    DevBuf<uint32_t> cigar;
    DevBuf<uint8_t> seq, ref;
    DevBuf<int64_t> ctg_goff, ctg_qoff, ctg_rec_begin;
    DevBuf<int32_t> ctg_limit;
    // batches made by fzp_align_to_batch: the aligned reads of contig c in q_id order are qid_read[h_slot_off[c] .. + n_qid(c))
    DevBuf<int32_t> qid_read;
    std::vector<int64_t> h_slot_off;
    // ... and their records stay in K1's packed form (PkSrc points into the alnjob, which must outlive the batch): `cigar` / `seq` / the CIGAR checkpoints are made
    // only when somebody asks for them (fzp_batch_need_bytes: K6, whose tally walks the D / I ops of the run-length form)
    bool packed = false, have_bytes = true;
    PkSrc pk;
    DevBuf<int64_t> rec_read;          // [n_rec] the read behind every record
    std::function<int(fzp_ctx *, fzp_batch *)> make_bytes;
    // CIGAR checkpoints: per record, per 64-op chunk, the (reference, query) offsets at the chunk's start
    std::vector<int64_t> h_ck_off;     // [n_rec+1] prefix of ceil(n_ops/64) (batches built from host records)
    int64_t n_ck = 0;                  // total 64-op chunks = size of ck_ref / ck_q
    DevBuf<int64_t> ck_off;
    DevBuf<int32_t> ck_ref, ck_q, rec_span, ctg_maxspan;
    std::vector<int32_t> h_tile_ctg, h_tile_start;   // K2 position tiles (never span contigs)
    DevBuf<int32_t> tile_ctg, tile_start;
    // K2
    DevBuf<uint32_t> cnt, oth, site_idx, row_off32;
    DevBuf<uint8_t> flag8, blk_live;         // blk_live: per 256-position block, 1 if its tile saw a record
    DevBuf<fzp_site> sites;
    DevBuf<int64_t> site_g;
    DevBuf<int32_t> site_ctg;
    DevBuf<int64_t> site_begin;
    DevBuf<uint64_t> vtmp;
    DevBuf<uint32_t> vfill;
    DevBuf<int32_t> vmap_qid;
    // K3
    DevBuf<int32_t> setq;
    DevBuf<uint32_t> set_n;
    DevBuf<uint32_t> cand_n, cap_off, nkept, kept_off;
    DevBuf<fzp_arow> arows_tmp, arows;
    DevBuf<int64_t> arow_begin;
    // K4
    DevBuf<uint32_t> lk_flag, lk_idx;
    DevBuf<int32_t> lk_i1, lk_i2, lk_cis, lk_trans;
    DevBuf<uint32_t> left_n, left_off, left_fill, right_n, right_off, fr2;
    DevBuf<int32_t> left_lk;
    DevBuf<int4> left_pk;             // per left link, CSR order: (left site, cis, trans, the site itself)
    DevBuf<uint32_t> pj;              // pointer-jumping words: parent<<1 | flip
    DevBuf<uint8_t> orient;
    DevBuf<int32_t> lext, rext, lscore, rscore, rawblk, blkcnt, blknew;
    DevBuf<fzp_pvar> pvars_tmp, pvars;
    DevBuf<uint32_t> pv_n, pv_off;
    DevBuf<int64_t> pvar_begin;
    DevBuf<int32_t> site_blk;
    DevBuf<uint8_t> site_b1;
    // K5
    DevBuf<int32_t> bmin, bmax;
    DevBuf<uint32_t> rng_n, rng_off, c0, c1, pr_flag, pr_idx;
    DevBuf<fzp_pread> preads;
    DevBuf<int64_t> pread_begin;
    // this batch's pinned host staging block (from the ctx's cache, back to it when the batch dies): the K2/K3 records start
    // their way into it early (pf_early), fzp_batch_result_all adds the rest and hands out views that live as long as the batch
    fzp_ctx *pin_ctx = nullptr;
    void *pin = nullptr;
    size_t pin_cap = 0;
    bool pf_early = false;
    size_t pf_sites = 0, pf_vmap = 0, pf_arows = 0, pf_end = 0;
    // a caller that takes variant_map and atable as device-made text (fzp_pipe.hip) does not want their rows on the host: 14.7 MB per bench step that no one reads, and a
    // device-to-host copy in flight holds up every kernel that writes memory beside it (measured: K4's 5 us fills took 77 and 266 us under these two copies)
    bool host_skip_rows = false;
    // fzp_batch_result_begin: the block and read records are on their way (main stream); fzp_batch_result_all then only waits
    bool late_begun = false, late_early = false, late_event = false;
    size_t late_off[5] = {0, 0, 0, 0, 0};
    ~fzp_batch() { if (pin) fzp_pinned_release(pin_ctx, pin); if (life) life->batches.fetch_sub(1); }
    // scratch
    DevBuf<uint64_t> totals;          // a few device u64 scalars
    DevBuf<int32_t> errflag;
};

// sequences that are on the device already (fzp_phase_contigs_files, r6: the group's files uploaded as they are, records found by fzp_fasta.hip): contig c =
// d_raw[d_ctg_be[2c], d_ctg_be[2c + 1]), read r = d_raw[d_read_be[2r], d_read_be[2r + 1]) (device arrays)
struct fzp_aln_dev_src { const uint8_t *d_raw = nullptr; const int64_t *d_ctg_be = nullptr, *d_read_be = nullptr; };
int fzp_align_create_dev(fzp_ctx *ctx, int32_t n_ctg, const int64_t *ctg_len, int64_t n_reads, const int32_t *read_ctg, const int64_t *read_len, const fzp_aln_dev_src *dev,
                         const fzp_align_params *params, fzp_alnjob **out);      // (fzp_align.hip)

// polishing (fzp_polish_tigs, fzp_cns.hip): the templates as the alnjob keeps them on the device (upper-cased ASCII, contig c at ref + ref_off[c]) and their lengths (host)
struct fzp_cns_polish { const uint8_t *ref = nullptr; const int64_t *ref_off = nullptr; const int64_t *len = nullptr; };
void fzp_align_templates(const fzp_alnjob *job, const uint8_t **ascii, const int64_t **aoff);      // (fzp_align.hip)
void fzp_align_host_reads(const fzp_alnjob *job, int32_t *n_ctg, int64_t *n_reads, const int32_t **read_ctg);      // (fzp_align.hip)

inline int fzp_batch_source_ok(const fzp_batch *b) {
    if (b->life && b->life->dead.load()) { fzp_set_error("this batch reads the packed records of an alignment job that has been destroyed (fzp_align_destroy before fzp_batch_destroy)"); return FZP_EINVAL; }
    return FZP_OK;
}

// stage drivers (fzp_phase.hip)
int fzp_align_run_deferred(fzp_ctx *ctx, fzp_alnjob *job);   // fzp_align_run whose fail-list overflow question is answered by the fzp_align_to_batch that follows (fzp_align.hip)
int fzp_batch_result_begin(fzp_ctx *ctx, fzp_batch *b);     // the record copies fzp_batch_result_all waits for, started now on the main stream (fzp_api.hip)
int fzp_batch_need_bytes(fzp_ctx *ctx, fzp_batch *b);      // packed batches: run-length CIGAR words, byte SEQ and the 64-op checkpoints, now (fzp_hetcall.hip)
int fzp_k2_het_call(fzp_ctx *ctx, fzp_batch *b);
int fzp_k3_sets(fzp_ctx *ctx, fzp_batch *b);
int fzp_k3_assoc(fzp_ctx *ctx, fzp_batch *b);
int fzp_k4_blocks(fzp_ctx *ctx, fzp_batch *b);
int fzp_k5_reads(fzp_ctx *ctx, fzp_batch *b);
