/*
 * fzphase.h -- C-ABI of libfzphase.so, the MI355X (gfx950) phasing engine behind FALCON_unzip's
 * per-contig task interface.
 *
 * The reference (PacificBiosciences/FALCON_unzip, /root/reference) has no FFI: its boundary is
 * process + files.  Each entry point below therefore replaces the BODY of one
 * reference task function and exchanges the same records that function reads/writes as text;
 * the text formats themselves are produced by the fzp_format_* serializers, byte-for-byte.
 *
 *   entry point             replaces (reference file:line)
 *   ----------------------  ---------------------------------------------------------------
 *   fzp_parse_sam           falcon_unzip/phasing.py:42-75   (SAM text -> q_id, filters, CIGAR)
 *   fzp_het_call            falcon_unzip/phasing.py:14-135  make_het_call
 *   fzp_assoc_table         falcon_unzip/phasing.py:137-206 generate_association_table
 *   fzp_phase_blocks        falcon_unzip/phasing.py:208-421 get_score + get_phased_blocks
 *   fzp_phase_reads         falcon_unzip/phasing.py:423-480 get_phased_reads
 *   fzp_batch_*             falcon_unzip/phasing.py:482-553 phasing() chain, many contigs per launch
 *   fzp_readmap             falcon_unzip/phasing_readmap.py:8-51 get_phasing_readmap
 *   fzp_align_*             falcon_unzip/unzip.py:61-99     task_run_blasr (blasr + samtools sort;
 *                                                           own aligner spec, parity unpinned)
 *   fzp_format_*            the `print >>f` statements at phasing.py:124-134,199,418-421,478-480
 *                           and phasing_readmap.py:49-51
 *
 * Conventions: plain C, no exceptions cross the boundary; every function returns FZP_OK (0) or a
 * negative FZP_E* code and records a message readable through fzp_last_error() (thread-local).
 * Inputs are caller-owned host buffers; outputs are library-allocated host buffers released with
 * fzp_free() (or the matching *_free).  Any number of fzp_ctx per process and device; a ctx is not
 * thread-safe, different ctxs are independent (own streams, own device-block and pinned-block caches) and may be
 * driven from different threads.  All compute runs in HIP kernels on the ctx's
 * device: there is NO CPU fallback -- without a usable gfx950 device fzp_ctx_create fails.
 */
#ifndef FZPHASE_H
#define FZPHASE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FZP_OK 0
#define FZP_EINVAL (-1)     /* malformed input (where the reference raises IndexError/KeyError/ValueError) */
#define FZP_EZERODIV (-2)   /* CIGAR without ops: ZeroDivisionError at phasing.py:72 */
#define FZP_ENOMEM (-3)
#define FZP_EUNSORTED (-4)  /* accepted records not sorted by POS: reference behaviour is order-dependent
                               (`--bam path to sorted bam file`, phasing.py:562); refused */
#define FZP_EDEVICE (-5)    /* HIP runtime error */
#define FZP_ENODEVICE (-6)  /* no gfx950 device / library built without device code */
#define FZP_EIO (-7)        /* a file could not be opened, read or written (the message carries the path and strerror) */

/* BAM CIGAR op numbering; a cigar word is  len << 4 | op  */
enum { FZP_OP_M = 0, FZP_OP_I = 1, FZP_OP_D = 2, FZP_OP_N = 3, FZP_OP_S = 4, FZP_OP_H = 5, FZP_OP_P = 6,
       FZP_OP_EQ = 7, FZP_OP_X = 8 };

typedef struct fzp_ctx fzp_ctx;

const char *fzp_version(void);
const char *fzp_last_error(void);
void fzp_free(void *p);

int fzp_ctx_create(int device_id, unsigned flags, fzp_ctx **out);
void fzp_ctx_destroy(fzp_ctx *ctx);
int fzp_ctx_synchronize(fzp_ctx *ctx);
/* What hipSetDeviceFlags answered when the last context asked for its scheduling mode (FZP_SCHED, default blocking sync): 0 = accepted, -1 = nobody asked yet.  A host that
 * brought the device up before (torch under backend nccl) is fine -- measured: the flag is accepted afterwards; what must be in the environment BEFORE the runtime comes up is
 * ROC_SIGNAL_POOL_SIZE (fzp_ctx_create sets 4096 when it is the first to touch HIP; a host that initialises HIP itself exports it: bench.py and scripts/ do). */
int fzp_sched_status(void);
int fzp_mem_info(fzp_ctx *ctx, size_t *free_bytes, size_t *total_bytes);   /* hipMemGetInfo of the ctx's device (blocks cached by the ctx count as used) */

/* Per-kernel device timing (HIP events on the ctx stream).  fzp_prof_enable(ctx,1) starts
 * collecting; fzp_prof_get returns the summed duration and launch count of kernel `name`
 * since the last fzp_prof_reset.  Names: see DESIGN.md section 5.  on = 2 brackets the DP stage ("k1_sw") only:
 * every bracket costs the stream ~10 us and the host ~80 us of CPU, sixty of them per step are not free. */
int fzp_prof_enable(fzp_ctx *ctx, int on);
int fzp_prof_reset(fzp_ctx *ctx);
int fzp_prof_get(fzp_ctx *ctx, const char *name, double *total_ms, int64_t *launches);
int fzp_prof_names(fzp_ctx *ctx, char **names_nl_separated);

/* ---------------------------------------------------------------------------------------------
 * Alignment records of ONE contig: what make_het_call extracts from `samtools view` text
 * (phasing.py:42-75).  Only ACCEPTED records (those passing the two filters at phasing.py:72-75)
 * are listed, in input order; the q_id table lists every QNAME seen, accepted or not.
 * --------------------------------------------------------------------------------------------- */
typedef struct {
    int64_t n_rec;
    int32_t *rec_qid;     /* [n_rec]   q_id = order of first appearance of QNAME (phasing.py:47-54) */
    int32_t *rec_pos;     /* [n_rec]   0-based POS (phasing.py:57) */
    int64_t *cig_off;     /* [n_rec+1] */
    uint32_t *cigar;      /* [cig_off[n_rec]]  len<<4|op */
    int64_t *seq_off;     /* [n_rec+1] */
    uint8_t *seq;         /* [seq_off[n_rec]]  raw SEQ bytes, any symbol */
    int32_t n_qid;
    int64_t *qname_off;   /* [n_qid+1] */
    char *qnames;         /* concatenated, no separators */
    int32_t last_pos;     /* POS of the last accepted record, -1 if none: only positions < last_pos
                             are ever evaluated (phasing.py:98-102, no final flush) */
    int32_t max_ref_span; /* max over records of reference bases consumed */
    int64_t n_columns;    /* total aligned columns (M,=,X bases) of accepted records */
} fzp_alnset;

/* SAM text (header lines allowed) -> records.  Host-side C++; mirrors phasing.py:42-75 incl. the
 * IEEE-double clip filter.  FZP_EUNSORTED if accepted records are not POS-sorted. */
int fzp_parse_sam(const char *sam, size_t len, fzp_alnset **out);
void fzp_alnset_free(fzp_alnset *a);

/* ---------------------------------------------------------------------------------------------
 * Stage records
 * --------------------------------------------------------------------------------------------- */
typedef struct {           /* one variant_pos row (phasing.py:124) + where its variant_map rows are */
    int32_t pos;           /* 0-based */
    uint8_t ref_base;
    uint8_t base[4];       /* ranked: count desc, ties larger letter first (phasing.py:116-117) */
    uint8_t pad_[3];
    int32_t total;
    int32_t count[4];
    int64_t row_off;       /* variant_map rows [row_off, row_off+count[0]) carry base[0],
                              the next count[1] rows carry base[1] (phasing.py:125-128) */
} fzp_site;

typedef struct {           /* one atable row (phasing.py:199); alleles in A<C<T<G order (py2 dict order) */
    int32_t site1, site2;  /* indices into the contig's fzp_site array */
    int32_t n[4];          /* n11 n12 n21 n22 */
} fzp_arow;

typedef struct {           /* one 'V' line of phased_variants (phasing.py:421) */
    int32_t block;         /* 1-based phase block id */
    int32_t site;
    uint8_t b1, b2;        /* phase-0 allele, phase-1 allele */
    uint8_t pad_[2];
    int32_t lext, rext;    /* 1-based positions, as printed */
    int32_t lscore, rscore;
} fzp_pvar;

typedef struct {           /* one phased_reads row (phasing.py:478-480) */
    int32_t q_id, block, phase, n0, n1;
} fzp_pread;

typedef struct {           /* everything phasing() produces for one contig */
    int64_t n_sites;  fzp_site *sites;
    int64_t n_rows;   int32_t *vmap_qid;      /* variant_map q_id column, grouped per fzp_site.row_off */
    int64_t n_arows;  fzp_arow *arows;
    int64_t n_pvars;  fzp_pvar *pvars;
    int64_t n_preads; fzp_pread *preads;      /* ascending (q_id, block) */
} fzp_result;
void fzp_result_free(fzp_result *r);           /* frees the arrays, not the struct */

/* ---------------------------------------------------------------------------------------------
 * Single-contig stage entry points (host buffers in, host records out).  Each one uploads, runs
 * the HIP kernels of that stage and downloads.  ref_seq is the upper-cased contig (phasing.py:494).
 * --------------------------------------------------------------------------------------------- */
int fzp_het_call(fzp_ctx *ctx, const fzp_alnset *aln, const uint8_t *ref_seq, int64_t ref_len,
                 fzp_site **sites, int64_t *n_sites, int32_t **vmap_qid, int64_t *n_rows);
int fzp_assoc_table(fzp_ctx *ctx, const fzp_site *sites, int64_t n_sites, const int32_t *vmap_qid, int64_t n_rows,
                    fzp_arow **arows, int64_t *n_arows);
int fzp_phase_blocks(fzp_ctx *ctx, const fzp_site *sites, int64_t n_sites, const fzp_arow *arows, int64_t n_arows,
                     fzp_pvar **pvars, int64_t *n_pvars);
int fzp_phase_reads(fzp_ctx *ctx, const fzp_site *sites, int64_t n_sites, const int32_t *vmap_qid, int64_t n_rows,
                    const fzp_pvar *pvars, int64_t n_pvars, int32_t n_qid, fzp_pread **preads, int64_t *n_preads);

/* ---------------------------------------------------------------------------------------------
 * Many contigs per launch, inputs resident in HBM (the phasing() chain, phasing.py:482-553).
 *   fzp_batch_create   uploads the records of n_ctg contigs (one alnset + one ref_seq each)
 *   fzp_batch_run      K2..K5 over all contigs in batched launches; results stay on the device
 *   fzp_batch_result   downloads contig `ctg`'s records
 * --------------------------------------------------------------------------------------------- */
typedef struct fzp_batch fzp_batch;
#define FZP_STAGE_HET 1u
#define FZP_STAGE_ASSOC 2u
#define FZP_STAGE_BLOCKS 4u
#define FZP_STAGE_READS 8u
#define FZP_STAGE_ALL 15u
int fzp_batch_create(fzp_ctx *ctx, int32_t n_ctg, const fzp_alnset *const *aln, const uint8_t *const *ref_seq,
                     const int64_t *ref_len, fzp_batch **out);
int fzp_batch_run(fzp_ctx *ctx, fzp_batch *b, unsigned stages);
int fzp_batch_result(fzp_ctx *ctx, fzp_batch *b, int32_t ctg, fzp_result *out);
/* all contigs at once: arrays concatenated in contig order with GLOBAL site / row indices (site indices in
 * arows / pvars and row_off in sites count from the start of the batch); begin[] arrays have n_ctg+1 entries.
 * One D2H copy per array instead of one per contig.  The record arrays in `all` are BORROWED: they point into
 * the batch's own pinned staging block and stay valid until the next fzp_batch_run / fzp_batch_result_all on the
 * SAME batch or its fzp_batch_destroy (other batches of the ctx do not touch them); fzp_result_all_free releases
 * only the begin[] arrays. */
typedef struct {
    fzp_result all;
    int64_t *site_begin, *row_begin, *arow_begin, *pvar_begin, *pread_begin;
} fzp_result_all;
int fzp_batch_result_all(fzp_ctx *ctx, fzp_batch *b, fzp_result_all *out);
void fzp_result_all_free(fzp_result_all *r);
int fzp_batch_counts(fzp_ctx *ctx, fzp_batch *b, int64_t *n_rec, int64_t *n_columns, int64_t *n_positions,
                     int64_t *n_sites, int64_t *n_rows, int64_t *n_arows, int64_t *n_pvars, int64_t *n_preads);
void fzp_batch_destroy(fzp_ctx *ctx, fzp_batch *b);

/* ---------------------------------------------------------------------------------------------
 * Text serializers: the reference's on-disk formats, byte-for-byte (Python-2 `print`).
 * --------------------------------------------------------------------------------------------- */
int fzp_format_variant_pos(const fzp_site *sites, int64_t n_sites, char **text, size_t *len);
int fzp_format_variant_map(const fzp_site *sites, int64_t n_sites, const int32_t *vmap_qid, char **text, size_t *len);
int fzp_format_q_id_map(const fzp_alnset *aln, char **text, size_t *len);
int fzp_format_atable(const fzp_site *sites, const fzp_arow *arows, int64_t n_arows, char **text, size_t *len);
int fzp_format_phased_variants(const fzp_site *sites, const fzp_pvar *pvars, int64_t n_pvars, char **text, size_t *len);
int fzp_format_phased_reads(const fzp_pread *preads, int64_t n_preads, const char *ctg_id,
                            const int64_t *qname_off, const char *qnames, int32_t n_qid, char **text, size_t *len);

/* get_phasing_readmap (phasing_readmap.py:8-51): text in, `rid_to_phase.<ctg>` text out plus the
 * 16-byte records the multi-GPU gather carries: (pread id, contig index, block, phase). */
typedef struct { int32_t arid, ctg, block, phase; } fzp_r2p;
int fzp_readmap(const char *phased_reads, size_t pr_len, const char *rawread_ids, size_t rr_len,
                const char *pread_ids, size_t pi_len, const char *pread_to_contigs, size_t pc_len,
                const char *ctg_id, int32_t ctg_index, fzp_r2p **recs, int64_t *n_recs, char **text, size_t *len);

/* ---------------------------------------------------------------------------------------------
 * K1: read -> contig banded alignment (role of blasr + samtools sort, unzip.py:86-91).
 * Own deterministic spec "fzalign v1.7" (DESIGN.md section 6): k-mer seeding over the ANCHORED k-mers (those that start with AC or end with GT: an eighth of the
 * positions, the same ones on the contig and on a read wherever they agree; v1.6 -- seed_anchored = 0 -- indexed every 2nd position and looked up every 4th / 12th read k-mer), up to two
 * candidate placements per read (chained anchors); the chain's hits every >= 3 072 read bases are waypoints and the extension is a sequence of
 * independent banded DPs from one waypoint to the next (the role of blasr's alignment between chain anchors), a free one past the last waypoint and one
 * backward from the anchor -- each in an adaptive anti-diagonal band of 64 cells with linear-gap scores --, the better forward extension kept
 * (--bestn 1), the best-scoring stretch of the joined path reported, identity gate (--minPctIdentity 70), traceback to =/X/I/D/S CIGARs.
 * Parity vs blasr is UNPINNED; the kernels are bit-exact against their scalar CPU twin in oracle/align_oracle.c.  With the default scores (2, 4, 3)
 * the DP runs bit-sliced (one piece per lane); other scores run the wave-per-piece kernel: same spec, same results.
 * --------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t kmer;            /* seed length (<=16), default 16 */
    int32_t seed_stride;     /* query every seed_stride-th read k-mer, default 4 */
    int32_t match, mismatch, gap;   /* scores: +match, -mismatch, -gap (defaults 2,4,3) */
    int32_t min_seed_hits;   /* reads with fewer votes in the best window are unaligned, default 8 */
    int32_t min_pct_identity; /* alignments below this identity are dropped (blasr --minPctIdentity 70.0, unzip.py:87); default 70, 0 = off */
    int32_t seed_anchored;   /* 1 (default, v1.7): index and look-ups use the anchored k-mers; 0: v1.6's fixed strides (every 2nd contig position, every seed_stride-th read k-mer) */
    int32_t band;            /* fzalign v1.8: cells of the adaptive band, 64 or 32 (0: the default).  On every test set the 32-cell band finds the same scores (profiles/r6_band32_go_nogo.txt)
                              * at about half the DP's instructions: planes and windows are one register instead of two, and the 8-byte trace-back record is the whole mask */
    int32_t reserved[7];
} fzp_align_params;
void fzp_align_params_default(fzp_align_params *p);

typedef struct {             /* per read, input order */
    int32_t aligned;         /* 0 = no placement */
    int32_t strand;          /* 0 forward, 1 reverse-complement */
    int32_t pos;             /* 0-based contig position of the first aligned column */
    int32_t ref_end;         /* one past the last aligned contig position */
    int32_t q_start, q_end;  /* aligned part of the (oriented) read */
    int32_t score;
    int32_t n_cigar;         /* CIGAR words incl. soft clips */
    int64_t cells;           /* DP cells evaluated for this read (steps * 64 over all its extension pieces, both candidates) */
    int32_t n_columns;       /* aligned columns (= and X bases) */
    int32_t n_match;         /* '=' columns; 100 * n_match / (columns + inserted + deleted bases) is the identity the gate tests */
} fzp_aln_summary;

/* reads/contigs are ASCII (ACGT, any case; any other symbol is treated as 'A' throughout, and the
 * SEQ handed to the phasing stages is the oriented read re-spelled in upper-case ACGT).  All reads in one call belong to contig `ctg_seq` (the reference
 * aligns <ctg>_reads.fa to <ctg>_ref.fa, unzip.py:233-234).
 * fzp_align_create uploads and 2-bit-packs reads and contigs and builds the contigs' k-mer tables (they depend on the contigs and k only);
 * fzp_align_run seeds, chains, extends and traces back, and may be repeated on the same job. */
typedef struct fzp_alnjob fzp_alnjob;
int fzp_align_create(fzp_ctx *ctx, int32_t n_ctg, const uint8_t *const *ctg_seq, const int64_t *ctg_len,
                     int64_t n_reads, const int32_t *read_ctg, const int64_t *read_off, const uint8_t *read_seq,
                     const fzp_align_params *params, fzp_alnjob **out);
/* the same with the reads given as spans of one host buffer: read r = buf[read_be[2r], read_be[2r + 1]).  The bytes between the first span's begin and the last
 * span's end go to the device in one piece, so a FASTA file's bytes can be handed over as they are (one line per sequence, as falcon_kit's
 * fetch_reads -- unzip.py:49-50 -- writes <ctg>_reads.fa): headers and line ends are left behind when the reads are packed.  fzp_phase_contigs_files uses it. */
int fzp_align_create_spans(fzp_ctx *ctx, int32_t n_ctg, const uint8_t *const *ctg_seq, const int64_t *ctg_len,
                           int64_t n_reads, const int32_t *read_ctg, const int64_t *read_be /* [2 n_reads] */, const uint8_t *buf,
                           const fzp_align_params *params, fzp_alnjob **out);
int fzp_align_run(fzp_ctx *ctx, fzp_alnjob *job);
/* forget the k-mer tables fzp_align_create built: the next fzp_align_run builds them again, inside the run */
int fzp_align_invalidate_index(fzp_alnjob *job);
/* K1's intermediates, for checkers (tests/test_gpu_align.py): the k-mer table of a contig as a sorted list of entries (key << 32 | position << 1 | strand bit;
 * malloc'ed, fzp_free), an order-free fingerprint of all tables {entries, sum, xor of a 64-bit mix of every entry}, the tables built again, and a read's hit list as the
 * last run's seeding left it (out: room for 2 x 4 096 words; pairs (strand << 31 | oriented read offset, contig position) in spec order) */
int fzp_debug_index_entries(fzp_ctx *ctx, fzp_alnjob *job, int32_t ctg, uint64_t **entries, int64_t *n);
int fzp_debug_index_fingerprint(fzp_ctx *ctx, fzp_alnjob *job, uint64_t *out3);
int fzp_debug_rebuild_index(fzp_ctx *ctx, fzp_alnjob *job);
int fzp_debug_read_hits(fzp_ctx *ctx, fzp_alnjob *job, int64_t read, uint32_t *out, int32_t *n_hits);
int fzp_align_summaries(fzp_ctx *ctx, fzp_alnjob *job, fzp_aln_summary *out /* [n_reads] */);
/* a 64-bit fingerprint of every read's CIGAR as the device keeps it (runs of M / I / D / S, the clips included): sum over the words of
 * splitmix64(index << 32 | word), 0 for a read without an alignment.  Lets a checker hold every CIGAR of a large run against another aligner's -- gap placement and
 * all -- for 8 bytes per read (tests/test_gpu_scale.py, bench.py's cpu_baseline). */
int fzp_align_cigar_hashes(fzp_ctx *ctx, fzp_alnjob *job, uint64_t *out /* [n_reads] */);
/* reads of the last run that had a second candidate placement extended (repeats; blasr --bestn 1 keeps the better one) */
int64_t fzp_align_n_second(const fzp_alnjob *job);
/* alignment records of contig `ctg` in (POS, read index) order, q_id = rank in that order; names
 * (optional, may be NULL -> "read/<index>") fill the q_id table */
int fzp_align_alnset(fzp_ctx *ctx, fzp_alnjob *job, int32_t ctg, const int64_t *name_off, const char *names,
                     fzp_alnset **out, int64_t **read_index /* [n_rec] input index of each record, optional */);
/* the same without make_het_call's two record filters (phasing.py:72-75): EVERY aligned read, which is what the blasr task's
 * <ctg>_sorted.bam holds (unzip.py:86-91) -- the BAM must carry the filtered-out reads too, or a later `fc_phasing.py` run on it would
 * number its q_ids differently */
int fzp_align_alnset_all(fzp_ctx *ctx, fzp_alnjob *job, int32_t ctg, const int64_t *name_off, const char *names,
                         fzp_alnset **out, int64_t **read_index);
/* hand the aligned records of all contigs to the phasing stages without leaving the device.
 * LIFETIME: the batch BORROWS the job's device memory -- it reads the alignments' packed op streams and the packed reads where K1 left them -- so the job has to stay as
 * it is while the batch lives: destroy the batch (fzp_batch_destroy) before the job is run again or destroyed.  Enforced: fzp_align_run returns FZP_EINVAL while a batch
 * made from the job is open; after fzp_align_destroy of its job every entry point that would read the borrowed memory (fzp_batch_run, fzp_batch_consensus*) returns
 * FZP_EINVAL -- results already computed stay readable (fzp_batch_result*), and fzp_batch_destroy is always valid. */
int fzp_align_to_batch(fzp_ctx *ctx, fzp_alnjob *job, fzp_batch **out);
void fzp_align_destroy(fzp_ctx *ctx, fzp_alnjob *job);
/* SAM text of an alnset (what `samtools view` would print), for users who want the alignments */
int fzp_format_sam(const fzp_alnset *aln, const char *ctg_id, const int32_t *flags, char **text, size_t *len);

/* ---- K6: phased-pile consensus (BASELINE config 4; "next" row n3's device part).  The reference has no consensus code
 * of its own (falcon_sense is falcon_kit's, Arrow is `variantCaller`'s, run_quiver.py:82-97): parity unpinned, the
 * definition is oracle/cns_oracle.c ("fzcns v2").  For every (contig, block, phase) with at least one record
 * in its pile: the consensus of the block's span [first site, last site] over the records of the reads K5 gave that
 * phase.  Needs a batch with alignment records on which FZP_STAGE_ALL has run. */
typedef struct {
    int32_t ctg, block, phase;   /* block ids as in phased_variants (1-based, per contig) */
    int32_t lo, hi;              /* 0-based inclusive span on the contig */
    int32_t n_records;           /* alignment records in the pile */
    int64_t seq_off, seq_len;    /* into fzp_tigs.seq */
} fzp_tig;
typedef struct { int64_t n_tigs; fzp_tig *tigs; int64_t n_seq; uint8_t *seq; } fzp_tigs;
int fzp_batch_consensus(fzp_ctx *ctx, fzp_batch *b, fzp_tigs *out);                 /* fzcns v3 */
/* version 1: at most one inserted base per position, decided on the count of I ops (the first definition; faster);
 * version 2: insertions of up to 8 bases, every base gated on the support of its prefix-linked tag (2 * count > coverage), the weight
 * falcon_sense gives a tag link in its (t_pos, delta, base) graph */
/* version 3 (the default): the LENGTH of an insertion is decided first -- the median inserted length over the pile's reads, i.e. the largest l with more than half of
 * the coverage carrying an I op of at least l bases (one base only with a majority for the same base) -- then its bases level by level among the I ops long enough
 * to have one: an inserted het that noisy reads spell as 2, 3, 4 or 5 bases is still called */
int fzp_batch_consensus_v(fzp_ctx *ctx, fzp_batch *b, int version, fzp_tigs *out);

/* Polishing: a tig as the template -- the consensus ROLE of run_quiver.py:82-97 (`pbalign` of a tig's routed reads to that tig, then `variantCaller`'s per-tig call; both
 * external and closed here, so this is the repo's own aligner + pile vote: parity unpinned, twin = oracle/cns_oracle.c over one pile per tig).  tig_seq[c] / tig_len[c]:
 * the layout's tigs (graphs_to_h_tigs.py:406-410,558-562: the records of p_ctg.<ctg>.fa / h_ctg_all.<ctg>.fa); read r = read_seq[read_off[r], read_off[r + 1]) belongs to
 * tig read_tig[r] (what fzp_track_reads / fzp_bam_route assigned it).  K1 aligns every read to its tig, K6's packed tally (fzcns v3) runs over the whole tig as ONE pile.
 * out: every tig once, input order (ctg = its index, block 1, phase 0, lo 0, hi len - 1, n_records = its aligned reads that passed the record filters); a tig without
 * records comes back upper-cased and otherwise unchanged.  Release with fzp_tigs_free. */
int fzp_polish_tigs(fzp_ctx *ctx, int32_t n_tigs, const uint8_t *const *tig_seq, const int64_t *tig_len, int64_t n_reads, const int32_t *read_tig,
                    const int64_t *read_off, const uint8_t *read_seq, const fzp_align_params *params, fzp_tigs *out);
void fzp_tigs_free(fzp_tigs *t);     /* frees the arrays, not the struct */
/* FASTA of one contig's tigs: ">{ctg_id}_{block:03d}_{phase} {lo+1} {hi+1} {n_records}\n{sequence}\n" */
int fzp_format_tigs(const fzp_tigs *t, int32_t ctg, const char *ctg_id, char **text, size_t *len);

/* ---- the two large files of a batch, serialised on the device: `het_call/variant_map` (phasing.py:125-128) and `g_atable/atable`
 * (phasing.py:199) of ALL contigs, contig after contig, byte-identical to fzp_format_variant_map / fzp_format_atable;
 * ctg_begin[n_ctg + 1] = byte offset of every contig's part.  text and ctg_begin are malloc'ed (fzp_free). */
#define FZP_TEXT_VARIANT_MAP 1
#define FZP_TEXT_ATABLE 2
int fzp_batch_text(fzp_ctx *ctx, fzp_batch *b, int what, char **text, size_t *len, int64_t **ctg_begin);

/* ---- many contigs, one call: the job fan-out of unzip_all (unzip.py:221-288: per contig one blasr task, unzip.py:61-99, and one
 * phasing task, unzip.py:102-133 = fc_phasing.py + fc_phasing_readmap.py), files included.  SURVEY 8b's fzp_phase_contigs. */
#define FZP_PIPE_CONSENSUS 1u     /* also K6 (fzp_batch_consensus): <ctg>/cns/phased_blocks.fa */
#define FZP_PIPE_ASYNC_WRITES 2u  /* the call returns once every text exists and its write is queued; a few background threads of the ctx write the files while
                                     the caller goes on (e.g. with the next job's kernels); fzp_pipe_flush(ctx) waits for them and reports the first error */
#define FZP_PIPE_REBUILD_INDEX 4u /* fzp_job_phase_write: build the contigs' k-mer tables again inside the call (fzp_align_create built them once; a job that
                                     sees its contigs once pays for them in its one run, and bench.py's step asks for exactly that) */
#define FZP_PIPE_BAM 8u           /* also the blasr task's artefacts, from the SAME alignment pass: <ctg>/blasr/<ctg>_sorted.bam + .bai (unzip.py:86-91,239-240) with
                                     every aligned read (fzp_align_alnset_all + fzp_format_bam); compressed on the writer threads */
#define FZP_PIPE_SENTINELS 16u    /* the job_done files pypeflow looks for (unzip.py:241,268; scripts at unzip.py:81,94,120,128): <ctg>/phasing/p_<ctg>_done once every
                                     phasing file of the contig is written, and with FZP_PIPE_BAM <ctg>/blasr/aln_<ctg>_done once the BAM is; `<job_done>.exit` is
                                     touched whether or not the contig succeeded (the scripts' `trap ... EXIT`), `<job_done>` only on success */
typedef struct {
    int32_t n_ctg;
    const char *const *ctg_id;     /* [n_ctg] names: directory names and the ctg column of phased_reads / rid_to_phase */
    const int64_t *name_off;       /* [n_reads + 1] read names by read index, concatenated (optional: NULL -> "read/<index>") */
    const char *names;
} fzp_names;
typedef struct {
    const char *out_dir;           /* files go to <out_dir>/<ctg_id>/{het_call/<three files>, g_atable/atable, get_phased_blocks/phased_variants,
                                      phased_reads, rid_to_phase.<ctg_id>} (phasing.py:501-503,520,534,543; unzip.py:269); NULL = texts are
                                      produced but nothing is written */
    const char *rawread_ids; size_t rr_len;       /* the three read_map files of fc_phasing_readmap.py (phasing_readmap.py:15-16,36), */
    const char *pread_ids; size_t pi_len;         /* whole-file texts; pread_to_contigs == NULL: no rid_to_phase output */
    const char *pread_to_contigs; size_t pc_len;
    const int32_t *ctg_index;      /* [n_ctg] contig index stored in the rid_to_phase records (the job-wide sorted contig list); NULL = 0..n-1 */
    int32_t n_threads;             /* host threads for the small files and the writes; 0 = all cores */
    int32_t n_lanes;               /* fzp_phase_contigs: contig groups in flight (each lane = a host thread with its own ctx); 0 = 2 */
    int64_t group_bases;           /* fzp_phase_contigs: read bases per contig group; 0 = from the device memory (~40 B of HBM per read base in flight) */
    unsigned flags;                /* FZP_PIPE_* */
    fzp_align_params align;        /* fzp_phase_contigs: aligner parameters */
} fzp_pipe_opts;
typedef struct {
    fzp_r2p *r2p;                  /* rid_to_phase records of all contigs, contig order (fzp_pipe_out_free) */
    int64_t n_r2p;
    int64_t n_reads, n_aligned, n_rec, n_sites, n_rows, n_arows, n_pvars, n_preads, n_groups, bytes_written;
    double dp_cells;
    double ms_upload, ms_k1, ms_phase, ms_results, ms_text;   /* host wall per section, summed over groups (lanes overlap) */
} fzp_pipe_out;
void fzp_pipe_opts_default(fzp_pipe_opts *o);
/* inputs resident in HBM: K1 -> K5 of every contig of `job`, all files, rid_to_phase records */
int fzp_job_phase_write(fzp_ctx *ctx, fzp_alnjob *job, const fzp_names *names, const fzp_pipe_opts *opts, fzp_pipe_out *out);
/* inputs in host memory (arguments as fzp_align_create): groups of contigs streamed through the device */
int fzp_phase_contigs(fzp_ctx *ctx, int32_t n_ctg, const uint8_t *const *ctg_seq, const int64_t *ctg_len, int64_t n_reads, const int32_t *read_ctg,
                      const int64_t *read_off, const uint8_t *read_seq, const fzp_names *names, const fzp_pipe_opts *opts, fzp_pipe_out *out);
/* inputs in the reference's own files: <reads_dir>/<ctg_id>_ref.fa (the record named <ctg_id> is the contig, phasing.py:489-494) and <ctg_id>_reads.fa
 * (unzip.py:204,233-234) for every names->ctg_id; names->name_off / names are ignored (read names come from the FASTA headers, first word).  The files are
 * parsed by host threads one contig group ahead of the lanes; host memory holds the groups in flight, never the whole call's reads. */
int fzp_phase_contigs_files(fzp_ctx *ctx, const char *reads_dir, const fzp_names *names, const fzp_pipe_opts *opts, fzp_pipe_out *out);
void fzp_pipe_out_free(fzp_pipe_out *o);
/* test hook (CPU only): the FASTA reader's view of one contig group -- what fzp_phase_contigs_files hands to fzp_align_create; outputs malloc'ed (fzp_free) */
int fzp_debug_load_fasta_group(const char *reads_dir, const char *const *ctg_id, int32_t n_ctg, int32_t n_threads, uint8_t **ref, int64_t **ref_off, uint8_t **blob,
                               int64_t **off, char **names, int64_t **name_off, int32_t **read_ctg, int64_t *n_reads);
/* the same through the r6 reader: the files read as they are, uploaded, their records found on the device (csrc/fzp_fasta.hip); what the packer would read comes back as bytes */
int fzp_debug_load_fasta_group_dev(fzp_ctx *ctx, const char *reads_dir, const char *const *ctg_id, int32_t n_ctg, int32_t n_threads, uint8_t **ref, int64_t **ref_off, uint8_t **blob,
                               int64_t **off, char **names, int64_t **name_off, int32_t **read_ctg, int64_t *n_reads);
/* test hook (CPU only): one contig's read map as fzp_job_phase_write makes it -- from phased-read records and the q_id name table (name q = names[name_off[q], name_off[q + 1]))
 * -- to be held against fzp_readmap, which works from the files' text; outputs malloc'ed (fzp_free) */
int fzp_debug_readmap_records(const char *rawread_ids, size_t rr_len, const char *pread_ids, size_t pi_len, const char *pread_to_contigs, size_t pc_len, const char *ctg_id,
                              int32_t ctg_index, const fzp_pread *preads, int64_t n_preads, const int64_t *name_off, const char *names, int32_t n_q, fzp_r2p **recs,
                              int64_t *n_recs, char **text, size_t *text_len);
int fzp_pipe_flush(fzp_ctx *ctx);   /* every queued file of FZP_PIPE_ASYNC_WRITES calls on this ctx (and its lanes) is on the file system when this returns */

/* ---- the one exchange step of the multi-GPU path (get_rid_to_phase_all, unzip.py:303-314): every rank contributes the
 * rid_to_phase records of its contigs, every rank receives all of them ordered by (contig index, pread id) -- the order of
 * rid_to_phase.all.  Two ncclAllGather calls over RCCL (counts, then the payload padded to the largest shard) on the ctx's
 * stream.  RCCL (librccl.so.1) is loaded at run time; FZP_ENODEVICE if it is missing.  One process per GPU: rank 0 obtains an id
 * (fzp_comm_unique_id), hands it to the other ranks by any means, every rank calls fzp_comm_create. */
#define FZP_COMM_ID_BYTES 128
typedef struct fzp_comm fzp_comm;
int fzp_comm_unique_id(char id[FZP_COMM_ID_BYTES]);
int fzp_comm_create(fzp_ctx *ctx, int rank, int world, const char id[FZP_COMM_ID_BYTES], fzp_comm **out);
int fzp_comm_ranks(fzp_comm *c, int *rank, int *world);      /* as ncclCommUserRank / ncclCommCount report them */
/* which RCCL the process bound: the file ncclAllGather lives in (dladdr) and ncclGetVersion -- a process that imported torch first gets torch's copy under the same soname.
 * FZP_COMM_TIMEOUT_S (default 600, 0 = none): fzp_comm_create and fzp_allgather_rid_to_phase return FZP_EDEVICE with a message instead of waiting for ever for a peer. */
int fzp_comm_library(char *path, size_t cap, int *version);
void fzp_comm_destroy(fzp_comm *c);
int fzp_allgather_rid_to_phase(fzp_comm *c, const fzp_r2p *local, int64_t n_local, fzp_r2p **all /* fzp_free */, int64_t *n_all);
/* rid_to_phase.all (unzip.py:285, 303-314) from gathered records in the order given: '%09d ctg block phase' rows (phasing_readmap.py:47-51), ctg = ctg_ids[record.ctg] */
int fzp_format_rid_to_phase_all(const fzp_r2p *recs, int64_t n, const char *const *ctg_ids, int32_t n_ctg, char **text, size_t *len);

/* ---- BAM emitter / reader ("next" row n1).  The reference's blasr task writes <ctg>_sorted.bam + index
 * (unzip.py:86-91) and make_het_call reads it through `samtools view <bam> <ctg>` (phasing.py:27).
 * fzp_format_bam: coordinate-sorted BAM (BGZF, EOF marker) of an alnset and, if bai != NULL, its .bai; MAPQ 254,
 * QUAL absent, as fzp_format_sam prints them.  fzp_bam_to_sam: the `samtools view [region]` role -- text lines of the
 * 11 mandatory columns (optional fields dropped; phasing.py reads columns 0,1,2,3,5,9); region NULL or "" = all
 * records, else RNAME.  Outputs are malloc'ed (fzp_free). */
int fzp_format_bam(const fzp_alnset *aln, const char *ctg_id, int64_t ctg_len, const int32_t *flags,
                   uint8_t **bam, size_t *bam_len, uint8_t **bai, size_t *bai_len);
int fzp_bam_to_sam(const uint8_t *bam, size_t len, const char *region, char **text, size_t *text_len);

/* raw record access for select_reads_from_bam.py (select_reads_from_bam.py:44-87 does this through pysam): every record of a BAM as its
 * bytes (block_size prefix included) plus its read name; and a writer that puts whole records under a given header. */
typedef struct {
    char *header_text; size_t header_len;          /* SAM header text, NUL padding stripped */
    int32_t n_ref; uint8_t *ref_block; size_t ref_block_len;   /* the reference dictionary as it sits in the file (after n_ref) */
    int64_t n_rec;
    int64_t *rec_off;                              /* [n_rec + 1] into records */
    uint8_t *records; size_t records_len;
    int64_t *name_off;                             /* [n_rec + 1] into names */
    char *names;
} fzp_bam_view;
int fzp_bam_open(const uint8_t *bam, size_t len, fzp_bam_view **out);
void fzp_bam_view_free(fzp_bam_view *v);
int fzp_bam_write(const char *header_text, size_t header_len, int32_t n_ref, const uint8_t *ref_block, size_t ref_block_len, int32_t n_parts,
                  const uint8_t *const *parts, const size_t *part_lens, uint8_t **bam, size_t *bam_len);
/* the same job for inputs of ANY size (select_reads_from_bam.py:69-90 streams record by record through pysam; subreads BAMs are tens of GB):
 * fzp_bam_read_header decodes only the first BGZF blocks of a file (header text without NUL padding, the reference block: n_ref x
 * {l_name, name, l_ref}); fzp_bam_route reads the inputs in order, one BGZF block at a time, and appends every record whose read name is in
 * the table (names[name_off[i] .. name_off[i+1]) -> destination name_dest[i]) to that destination's BAM -- header_text / ref_block first,
 * records from a block boundary, EOF marker at the end; a destination that receives no record gets no file.  Memory: one block per input
 * being read + one pending block per destination; no file descriptor is held between a destination's appends.
 * dest_records[n_dest] (optional) = records written per destination; first_use[n_dest] (optional) = destinations in the order they
 * received their first record, -1 padded.  FZP_EIO if a file cannot be opened / read / written, FZP_EINVAL on a malformed BAM. */
int fzp_bam_read_header(const char *path, char **text /* fzp_free */, size_t *text_len, int32_t *n_ref, uint8_t **ref_block /* fzp_free */, size_t *ref_block_len);
int fzp_bam_route(int32_t n_in, const char *const *in_paths, int64_t n_names, const int64_t *name_off, const char *names, const int32_t *name_dest,
                  int32_t n_dest, const char *const *dest_paths, const char *header_text, size_t header_len, int32_t n_ref, const uint8_t *ref_block,
                  size_t ref_block_len, int64_t *dest_records, int32_t *first_use);

/* ======================================================================== overlap filter ("next" row n2)
 * falcon_unzip/ovlp_filter_with_phase.py: the consumer of rid_to_phase.all.  It reads `LA4Falcon -mo` text
 * (13 whitespace-separated columns per overlap: q_id t_id -len idt q_strand q_s q_e q_l t_strand t_s t_e t_l tag)
 * three times: filter_stage1 (:49-143) counts 5'/3' overlaps per query and builds the ignore list, filter_stage2
 * (:145-186) collects contained reads, filter_stage3 (:188-277) keeps the best-n overlaps per read end with
 * in-phase partners first.  Here the text is tokenised once on the host, the three stages run on the device. */
typedef struct {
    int64_t max_diff;   /* --max_diff  :285 */
    int64_t max_cov;    /* --max_cov   :286 */
    int64_t min_cov;    /* --min_cov   :287 */
    int64_t min_len;    /* --min_len   :288 (default 2500) */
    int64_t bestn;      /* --bestn     :289 (default 10) */
} fzp_ovlp_params;
typedef struct fzp_ovlset fzp_ovlset;
/* Tokenise the dumps of n_files .las files (in fofn order) -- on the device: the text goes to HBM once and one thread
 * per line does str.split(), the id look-ups and int()/float() -- and read the rid_to_phase.all map (main :306-309:
 * later rows overwrite earlier ones; ids, contigs, blocks and phases are compared as strings).  The texts are BORROWED:
 * `texts[k]` must stay valid and unchanged until fzp_ovlset_free (fzp_ovl_filter and fzp_ovl_format read lines back from them; only a dump
 * without its final newline is copied); rid_map is copied.  The parsed columns stay resident in HBM inside the ovlset.  At most 4 GiB of text / 2^31 lines per call.
 * FZP_EINVAL where the reference would raise while reading (a line with fewer than 2 tokens, a map row with fewer
 * than 4). */
int fzp_ovl_parse(fzp_ctx *ctx, int32_t n_files, const char *const *texts, const size_t *lens, const char *rid_map,
                  size_t map_len, fzp_ovlset **out);
void fzp_ovlset_free(fzp_ovlset *s);
int64_t fzp_ovl_n_lines(const fzp_ovlset *s);   /* input lines */
int64_t fzp_ovl_n_rows(const fzp_ovlset *s);    /* lines whose q_id and t_id are both in the map (the rest never matter) */
/* filter_stage1..3 on the device.  rows: selected input lines (global 0-based line numbers over all files) in the
 * order the reference prints them; ignore / contained: the ids of the two sets as indices into the map's distinct
 * keys in first-appearance order (fzp_ovl_id_name).  All three are malloc'ed; release with fzp_free.  FZP_EINVAL if
 * a line that passes the phase checks has a field the reference's int()/float() would reject (or an integer beyond
 * 32 bits -- deliberate limit). */
int fzp_ovl_filter(fzp_ctx *ctx, const fzp_ovlset *s, const fzp_ovlp_params *params, int64_t **rows, int64_t *n_rows,
                   int32_t **ignore, int64_t *n_ignore, int32_t **contained, int64_t *n_contained);
/* the `print " ".join(l)` at :353 for the selected lines: the line's tokens, then "ctg.block.phase" of q and of t */
int fzp_ovl_format(const fzp_ovlset *s, const int64_t *rows, int64_t n_rows, char **text, size_t *len);
int fzp_ovl_id_name(const fzp_ovlset *s, int32_t id, const char **name, int32_t *len);

/* ======================================================================== raw-read tracker ("next" row n4)
 * falcon_unzip/rr_hctg_track.py (run_track_reads :68-139, tr_stage1 :31-66): from `LA4Falcon -m` dumps of the raw-read
 * overlaps, keep for every B-read its bestn best A-reads (phase-incompatible pairs vetoed, :49-57) and score the contigs
 * those A-reads map to.  text = the rawread_to_contigs file, "bread ctg count rank score in_ctg" per line, in CANONICAL
 * order (B-read id, score, contig name): the reference's own line order is a Python dict order.  Read ids must be
 * 9-digit decimals (what LA4Falcon and fc_get_read_hctg_map write); anything else is FZP_EINVAL. */
int fzp_track_reads(fzp_ctx *ctx, int32_t n_files, const char *const *texts, const size_t *lens,
                    const char *phased_reads, size_t pr_len, const char *read_to_contig_map, size_t rc_len,
                    const char *rawread_ids, size_t ri_len, int64_t min_len, int64_t bestn, char **text, size_t *len);

#ifdef __cplusplus
}
#endif
#endif /* FZPHASE_H */
