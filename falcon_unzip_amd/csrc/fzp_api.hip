// fzp_api.hip -- batch management and the C-ABI stage entry points (include/fzphase.h).
#include <algorithm>

#include "fzp_batch.h"

namespace {
template <class T>
T *host_copy(const DevBuf<T> &d, size_t off, size_t n, hipStream_t st, int *rc) {
    T *h = (T *)malloc((n ? n : 1) * sizeof(T));
    if (!h) { *rc = FZP_ENOMEM; return nullptr; }
    if (n) {
        hipError_t e = hipMemcpyAsync(h, d.p + off, n * sizeof(T), hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) { fzp_set_error("D2H copy: %s", hipGetErrorString(e)); free(h); *rc = FZP_EDEVICE; return nullptr; }
    }
    *rc = FZP_OK;
    return h;
}

int upload_contig_tables(fzp_ctx *ctx, fzp_batch *b) {
    hipStream_t st = ctx->stream;
    FZP_TRY(b->ctg_goff.upload(b->h_goff.data(), b->h_goff.size(), st));
    FZP_TRY(b->ctg_qoff.upload(b->h_qid_off.data(), b->h_qid_off.size(), st));
    FZP_TRY(b->ctg_limit.upload(b->h_limit.data(), b->h_limit.size(), st));
    return FZP_OK;
}
}  // namespace

// Sites, variant_map rows and atable rows are final after K3: start their way to the host on stream2 while K4 / K5 run on
// the main stream.  fzp_batch_result_all then only adds the (small) block and read records.  Best effort: on any
// failure the records are simply downloaded later.
namespace {
inline size_t al64(size_t x) { return (x + 63) & ~(size_t)63; }
bool batch_pinned(fzp_ctx *ctx, fzp_batch *b, size_t need) {      // the batch's staging block holds `need` bytes (contents are not kept when it grows)
    if (b->pin && b->pin_cap >= need) return true;
    if (b->pin) { (void)hipStreamSynchronize(ctx->stream2); fzp_pinned_release(b->pin_ctx, b->pin); b->pin = nullptr; b->pin_cap = 0; b->pf_early = false; }
    b->pin = fzp_pinned_acquire(ctx, need, &b->pin_cap);
    b->pin_ctx = ctx;
    return b->pin != nullptr;
}
void prefetch_early_records(fzp_ctx *ctx, fzp_batch *b) {
    b->pf_early = false;
    const size_t s_sites = (size_t)b->n_sites * sizeof(fzp_site), s_vmap = (size_t)b->n_rows * sizeof(int32_t), s_arows = (size_t)b->n_arows * sizeof(fzp_arow);
    // room for what comes later as well: at most one block record per site, one read record per variant_map row
    const size_t need = al64(s_sites) + al64(s_vmap) + al64(s_arows) + al64((size_t)b->n_sites * sizeof(fzp_pvar)) + al64((size_t)b->n_rows * sizeof(fzp_pread)) + 4096;
    if (!batch_pinned(ctx, b, need)) return;
    if (!ctx->ev_pf && hipEventCreateWithFlags(&ctx->ev_pf, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ctx->ev_pf = nullptr; return; }
    if (hipEventRecord(ctx->ev_pf, ctx->stream) != hipSuccess || hipStreamWaitEvent(ctx->stream2, ctx->ev_pf, 0) != hipSuccess) { (void)hipGetLastError(); return; }
    char *base = (char *)b->pin;
    b->pf_sites = 0; b->pf_vmap = al64(s_sites); b->pf_arows = b->pf_vmap + al64(s_vmap); b->pf_end = b->pf_arows + al64(s_arows);
    bool ok = true;
    if (s_sites) ok = ok && hipMemcpyAsync(base + b->pf_sites, b->sites.p, s_sites, hipMemcpyDeviceToHost, ctx->stream2) == hipSuccess;
    if (s_vmap && !b->host_skip_rows) ok = ok && hipMemcpyAsync(base + b->pf_vmap, b->vmap_qid.p, s_vmap, hipMemcpyDeviceToHost, ctx->stream2) == hipSuccess;
    if (s_arows && !b->host_skip_rows) ok = ok && hipMemcpyAsync(base + b->pf_arows, b->arows.p, s_arows, hipMemcpyDeviceToHost, ctx->stream2) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); return; }
    if (!ctx->ev_pf_done && hipEventCreateWithFlags(&ctx->ev_pf_done, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ctx->ev_pf_done = nullptr; }
    if (ctx->ev_pf_done && hipEventRecord(ctx->ev_pf_done, ctx->stream2) != hipSuccess) { (void)hipGetLastError(); }
    b->pf_early = true;
}
}  // namespace

// ================================================================================ batch
extern "C" int fzp_batch_create(fzp_ctx *ctx, int32_t n_ctg, const fzp_alnset *const *aln, const uint8_t *const *ref_seq, const int64_t *ref_len,
                                fzp_batch **out) {
    if (!ctx || !out || n_ctg <= 0 || !aln || !ref_seq || !ref_len) { fzp_set_error("fzp_batch_create: bad arguments"); return FZP_EINVAL; }
    *out = nullptr;
    FZP_TRY(fzp_bind(ctx));
    fzp_batch *b = new fzp_batch();
    b->n_ctg = n_ctg;
    b->h_rec_begin.assign(1, 0); b->h_goff.assign(1, 0); b->h_qid_off.assign(1, 0);
    int64_t n_cig = 0, n_seq = 0;
    for (int c = 0; c < n_ctg; c++) {
        const fzp_alnset *a = aln[c];
        if (!a) { delete b; fzp_set_error("contig %d: null alnset", c); return FZP_EINVAL; }
        int32_t limit = a->last_pos > 0 ? a->last_pos : 0;     // positions beyond the contig's end may be piled up; only a het call there fails, as in
                                                                // the reference (IndexError at ref_seq[pos], phasing.py:124) -- checked after K2
        b->h_limit.push_back(limit);
        b->h_ref_len.push_back(ref_len[c]);
        b->h_rec_begin.push_back(b->h_rec_begin.back() + a->n_rec);
        b->h_goff.push_back(b->h_goff.back() + fzp_pos_pad(limit));
        b->n_eval += limit;
        b->h_qid_off.push_back(b->h_qid_off.back() + a->n_qid);
        n_cig += a->cig_off[a->n_rec];
        n_seq += a->seq_off[a->n_rec];
        b->n_columns += a->n_columns;
    }
    b->n_rec = b->h_rec_begin.back();
    b->n_pos = b->h_goff.back();
    b->n_qid = b->h_qid_off.back();
    b->n_cig = n_cig; b->n_seq = n_seq;
    // concatenate on the host, then one upload per array
    std::vector<int32_t> rec_pos((size_t)b->n_rec), rec_qid((size_t)b->n_rec), rec_ctg((size_t)b->n_rec);
    std::vector<int64_t> cig_off((size_t)b->n_rec + 1), seq_off((size_t)b->n_rec + 1);
    std::vector<uint32_t> cigar((size_t)n_cig);
    std::vector<uint8_t> seq((size_t)n_seq), ref((size_t)b->n_pos);
    int64_t r0 = 0, c0 = 0, s0 = 0;
    for (int c = 0; c < n_ctg; c++) {
        const fzp_alnset *a = aln[c];
        for (int64_t r = 0; r < a->n_rec; r++) {
            rec_pos[(size_t)(r0 + r)] = a->rec_pos[r];
            rec_qid[(size_t)(r0 + r)] = a->rec_qid[r];
            rec_ctg[(size_t)(r0 + r)] = c;
            cig_off[(size_t)(r0 + r)] = c0 + a->cig_off[r];
            seq_off[(size_t)(r0 + r)] = s0 + a->seq_off[r];
        }
        if (a->cig_off[a->n_rec]) memcpy(cigar.data() + c0, a->cigar, (size_t)a->cig_off[a->n_rec] * sizeof(uint32_t));
        if (a->seq_off[a->n_rec]) memcpy(seq.data() + s0, a->seq, (size_t)a->seq_off[a->n_rec]);
        if (b->h_limit[c]) memcpy(ref.data() + b->h_goff[c], ref_seq[c], (size_t)std::min<int64_t>(b->h_limit[c], ref_len[c]));      // the rest stays 0
        r0 += a->n_rec; c0 += a->cig_off[a->n_rec]; s0 += a->seq_off[a->n_rec];
    }
    cig_off[(size_t)b->n_rec] = c0;
    seq_off[(size_t)b->n_rec] = s0;
    b->h_ck_off.assign((size_t)b->n_rec + 1, 0);
    for (int64_t r = 0; r < b->n_rec; r++) b->h_ck_off[(size_t)r + 1] = b->h_ck_off[(size_t)r] + (cig_off[(size_t)r + 1] - cig_off[(size_t)r] + 63) / 64;
    b->n_ck = b->h_ck_off.back();
    hipStream_t st = ctx->stream;
    int rc = FZP_OK;
    if ((rc = b->rec_pos.upload(rec_pos.data(), rec_pos.size(), st)) || (rc = b->rec_qid.upload(rec_qid.data(), rec_qid.size(), st)) ||
        (rc = b->rec_ctg.upload(rec_ctg.data(), rec_ctg.size(), st)) || (rc = b->cig_off.upload(cig_off.data(), cig_off.size(), st)) ||
        (rc = b->seq_off.upload(seq_off.data(), seq_off.size(), st)) || (rc = b->cigar.upload(cigar.data(), cigar.size(), st)) ||
        (rc = b->seq.upload(seq.data(), seq.size(), st)) || (rc = b->ref.upload(ref.data(), ref.size(), st)) || (rc = upload_contig_tables(ctx, b)) ||
        (rc = b->ck_off.upload(b->h_ck_off.data(), b->h_ck_off.size(), st)) || (rc = b->ctg_rec_begin.upload(b->h_rec_begin.data(), b->h_rec_begin.size(), st))) {
        delete b;
        return rc;
    }
    hipError_t e = hipStreamSynchronize(st);   // host vectors go out of scope
    if (e != hipSuccess) { delete b; fzp_set_error("upload: %s", hipGetErrorString(e)); return FZP_EDEVICE; }
    b->have_aln = true;
    *out = b;
    return FZP_OK;
}

extern "C" void fzp_batch_destroy(fzp_ctx *ctx, fzp_batch *b) {
    if (!b) return;
    if (ctx) { (void)fzp_bind(ctx); (void)hipStreamSynchronize(ctx->stream); if (b->pin) (void)hipStreamSynchronize(ctx->stream2); }
    delete b;
}

extern "C" int fzp_batch_run(fzp_ctx *ctx, fzp_batch *b, unsigned stages) {
    if (!ctx || !b) return FZP_EINVAL;
    FZP_TRY(fzp_batch_source_ok(b));
    FZP_TRY(fzp_bind(ctx));
    if (stages & FZP_STAGE_HET) {
        if (!b->have_aln) { fzp_set_error("batch holds no alignment records"); return FZP_EINVAL; }
        FZP_TRY(fzp_k2_het_call(ctx, b));
    }
    if (stages & FZP_STAGE_ASSOC) FZP_TRY(fzp_k3_assoc(ctx, b));
    if ((stages & FZP_STAGE_ALL) == FZP_STAGE_ALL && b->have_sites && b->have_arows) prefetch_early_records(ctx, b);   // best effort
    if (stages & FZP_STAGE_BLOCKS) FZP_TRY(fzp_k4_blocks(ctx, b));
    if (stages & FZP_STAGE_READS) FZP_TRY(fzp_k5_reads(ctx, b));
    return FZP_OK;
}

extern "C" int fzp_batch_counts(fzp_ctx *ctx, fzp_batch *b, int64_t *n_rec, int64_t *n_columns, int64_t *n_positions, int64_t *n_sites, int64_t *n_rows,
                                int64_t *n_arows, int64_t *n_pvars, int64_t *n_preads) {
    (void)ctx;
    if (!b) return FZP_EINVAL;
    if (n_rec) *n_rec = b->n_rec;
    if (n_columns) *n_columns = b->n_columns;
    if (n_positions) *n_positions = b->n_eval;
    if (n_sites) *n_sites = b->n_sites;
    if (n_rows) *n_rows = b->n_rows;
    if (n_arows) *n_arows = b->n_arows;
    if (n_pvars) *n_pvars = b->n_pvars;
    if (n_preads) *n_preads = b->n_preads;
    return FZP_OK;
}

extern "C" int fzp_batch_result(fzp_ctx *ctx, fzp_batch *b, int32_t ctg, fzp_result *out) {
    if (!ctx || !b || !out || ctg < 0 || ctg >= b->n_ctg) return FZP_EINVAL;
    FZP_TRY(fzp_bind(ctx));
    memset(out, 0, sizeof *out);
    hipStream_t st = ctx->stream;
    int rc = FZP_OK;
    int64_t sb = 0, row_base = 0;
    if (b->have_sites) {
        sb = b->h_site_begin[ctg];
        int64_t se = b->h_site_begin[ctg + 1];
        out->n_sites = se - sb;
        out->sites = host_copy(b->sites, (size_t)sb, (size_t)(se - sb), st, &rc);
        if (rc) { fzp_result_free(out); return rc; }
        // row range of this contig: [row_off(first site), row_off(first site of the next contig))
        int64_t row_end = b->n_rows;
        fzp_site nxt;
        if (se < b->n_sites) {
            FZP_HIP(hipMemcpyAsync(&nxt, b->sites.p + se, sizeof nxt, hipMemcpyDeviceToHost, st));
        }
        FZP_HIP(hipStreamSynchronize(st));
        if (se < b->n_sites) row_end = nxt.row_off;
        row_base = out->n_sites ? out->sites[0].row_off : row_end;
        for (int64_t i = 0; i < out->n_sites; i++) out->sites[i].row_off -= row_base;
        out->n_rows = row_end - row_base;
        out->vmap_qid = host_copy(b->vmap_qid, (size_t)row_base, (size_t)out->n_rows, st, &rc);
        if (rc) { fzp_result_free(out); return rc; }
    }
    if (b->have_arows) {
        int64_t ab = b->h_arow_begin[ctg], ae = b->h_arow_begin[ctg + 1];
        out->n_arows = ae - ab;
        out->arows = host_copy(b->arows, (size_t)ab, (size_t)(ae - ab), st, &rc);
        if (rc) { fzp_result_free(out); return rc; }
        FZP_HIP(hipStreamSynchronize(st));
        for (int64_t i = 0; i < out->n_arows; i++) { out->arows[i].site1 -= (int32_t)sb; out->arows[i].site2 -= (int32_t)sb; }
    }
    if (b->have_blocks) {
        int64_t pb = b->h_pvar_begin[ctg], pe = b->h_pvar_begin[ctg + 1];
        out->n_pvars = pe - pb;
        out->pvars = host_copy(b->pvars, (size_t)pb, (size_t)(pe - pb), st, &rc);
        if (rc) { fzp_result_free(out); return rc; }
        FZP_HIP(hipStreamSynchronize(st));
        for (int64_t i = 0; i < out->n_pvars; i++) out->pvars[i].site -= (int32_t)sb;
    }
    if (b->have_preads) {
        int64_t rb = b->h_pread_begin[ctg], re = b->h_pread_begin[ctg + 1];
        out->n_preads = re - rb;
        out->preads = host_copy(b->preads, (size_t)rb, (size_t)(re - rb), st, &rc);
        if (rc) { fzp_result_free(out); return rc; }
    }
    FZP_HIP(hipStreamSynchronize(st));
    return FZP_OK;
}

extern "C" void fzp_result_all_free(fzp_result_all *r) {
    if (!r) return;   // r->all is borrowed from the batch's pinned block
    free(r->site_begin); free(r->row_begin); free(r->arow_begin); free(r->pvar_begin); free(r->pread_begin);
    memset(r, 0, sizeof *r);
}

namespace {
struct ResPart { const void *src; size_t bytes; };
// the record arrays a batch can hand to the host, in their fixed order: sites, variant_map q_ids, atable rows, block records, read records (bytes 0 = not there / not wanted)
void result_parts(const fzp_batch *b, ResPart parts[5]) {
    const bool rows = !b->host_skip_rows;
    parts[0] = {b->sites.p, b->have_sites ? (size_t)b->n_sites * sizeof(fzp_site) : 0};
    parts[1] = {b->vmap_qid.p, b->have_sites && rows ? (size_t)b->n_rows * sizeof(int32_t) : 0};
    parts[2] = {b->arows.p, b->have_arows && rows ? (size_t)b->n_arows * sizeof(fzp_arow) : 0};
    parts[3] = {b->pvars.p, b->have_blocks ? (size_t)b->n_pvars * sizeof(fzp_pvar) : 0};
    parts[4] = {b->preads.p, b->have_preads ? (size_t)b->n_preads * sizeof(fzp_pread) : 0};
}
}  // namespace

// The copies of whatever is not on its way yet, on the main stream, without waiting: a caller with device work still to launch (fzp_pipe.hip: the two texts) puts it behind them.
int fzp_batch_result_begin(fzp_ctx *ctx, fzp_batch *b) {
    if (!ctx || !b) return FZP_EINVAL;
    FZP_TRY(fzp_bind(ctx));
    hipStream_t st = ctx->stream;
    ResPart parts[5];
    result_parts(b, parts);
    // parts 0..2 (sites, variant_map, atable) may already be on their way (prefetch_early_records)
    const bool early = b->pf_early && b->pin && b->have_sites && b->have_arows;
    size_t total = early ? b->pf_end : 0;
    for (int k = early ? 3 : 0; k < 5; k++) total += al64(parts[k].bytes);
    const bool use_early = early && total <= b->pin_cap;
    if (!use_early) {
        (void)hipStreamSynchronize(ctx->stream2);                     // nothing may still be writing into a block we are about to give back
        total = 0;
        for (int k = 0; k < 5; k++) total += al64(parts[k].bytes);
        if (!batch_pinned(ctx, b, total + 64)) { fzp_set_error("pinned host allocation of %zu bytes failed", total); return FZP_ENOMEM; }
    }
    size_t off = use_early ? b->pf_end : 0;
    for (int k = 0; k < 5; k++) {
        if (use_early && k < 3) { b->late_off[k] = k == 0 ? b->pf_sites : (k == 1 ? b->pf_vmap : b->pf_arows); continue; }
        b->late_off[k] = off;
        if (parts[k].bytes && hipMemcpyAsync((char *)b->pin + off, parts[k].src, parts[k].bytes, hipMemcpyDeviceToHost, st) != hipSuccess) { (void)hipGetLastError(); fzp_set_error("D2H copy failed"); return FZP_EDEVICE; }
        off += al64(parts[k].bytes);
    }
    // "the late records are over": what fzp_batch_result_all waits for (not the stream: the caller may have put more work behind the copies)
    if (!ctx->ev_late && hipEventCreateWithFlags(&ctx->ev_late, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ctx->ev_late = nullptr; }
    b->late_event = ctx->ev_late && hipEventRecord(ctx->ev_late, st) == hipSuccess;
    if (!b->late_event) (void)hipGetLastError();
    b->late_begun = true; b->late_early = use_early;
    return FZP_OK;
}

extern "C" int fzp_batch_result_all(fzp_ctx *ctx, fzp_batch *b, fzp_result_all *out) {
    if (!ctx || !b || !out) return FZP_EINVAL;
    FZP_TRY(fzp_bind(ctx));
    memset(out, 0, sizeof *out);
    hipStream_t st = ctx->stream;
    const size_t nb = (size_t)b->n_ctg + 1;
    auto dup = [&](const std::vector<int64_t> &v) {
        int64_t *p = (int64_t *)calloc(nb, sizeof(int64_t));
        if (p && v.size() == nb) memcpy(p, v.data(), nb * sizeof(int64_t));
        return p;
    };
    fzp_result &r = out->all;
    if (!b->late_begun) FZP_TRY(fzp_batch_result_begin(ctx, b));
    // the early records: wait for THEIR copies (the event recorded behind them), not for whatever else the caller has put on stream2 since
    if ((b->late_event ? hipEventSynchronize(ctx->ev_late) : hipStreamSynchronize(st)) != hipSuccess ||
        (b->late_early && (ctx->ev_pf_done ? hipEventSynchronize(ctx->ev_pf_done) : hipStreamSynchronize(ctx->stream2)) != hipSuccess)) { (void)hipGetLastError(); fzp_set_error("D2H sync failed"); return FZP_EDEVICE; }
    b->pf_early = false;                                               // a later run of the batch refills the block
    b->late_begun = false;
    ResPart parts[5];
    result_parts(b, parts);
    auto view = [&](int k) -> void * { return parts[k].bytes ? (char *)b->pin + b->late_off[k] : nullptr; };   // borrowed views into the batch's pinned block
    if (b->have_sites) {
        r.n_sites = b->n_sites; r.n_rows = b->n_rows;
        r.sites = (fzp_site *)view(0); r.vmap_qid = (int32_t *)view(1);
        out->site_begin = dup(b->h_site_begin);
    }
    if (b->have_arows) { r.n_arows = b->n_arows; r.arows = (fzp_arow *)view(2); out->arow_begin = dup(b->h_arow_begin); }
    if (b->have_blocks) { r.n_pvars = b->n_pvars; r.pvars = (fzp_pvar *)view(3); out->pvar_begin = dup(b->h_pvar_begin); }
    if (b->have_preads) { r.n_preads = b->n_preads; r.preads = (fzp_pread *)view(4); out->pread_begin = dup(b->h_pread_begin); }
    if (b->have_sites) {
        out->row_begin = (int64_t *)calloc(nb, sizeof(int64_t));
        for (int c = 0; c <= b->n_ctg; c++) {
            int64_t s0 = b->h_site_begin[c];
            out->row_begin[c] = s0 < b->n_sites ? r.sites[s0].row_off : b->n_rows;
        }
    }
    return FZP_OK;
}

// ================================================================================ stage injection
namespace {
// a 1-contig batch that starts from het-call output instead of alignment records
int batch_from_sites(fzp_ctx *ctx, const fzp_site *sites, int64_t n_sites, const int32_t *vmap_qid, int64_t n_rows, int32_t n_qid, fzp_batch **out) {
    if (n_sites < 0 || n_rows < 0 || (n_sites && !sites) || (n_rows && !vmap_qid)) { fzp_set_error("bad site/variant_map arguments"); return FZP_EINVAL; }
    int64_t rows = 0;
    int32_t max_q = -1;
    for (int64_t i = 0; i < n_sites; i++) {
        if (i && sites[i].pos <= sites[i - 1].pos) { fzp_set_error("sites must be in ascending position order (site %lld)", (long long)i); return FZP_EINVAL; }
        if (sites[i].row_off != rows || sites[i].count[0] < 0 || sites[i].count[1] < 0) { fzp_set_error("site %lld: variant_map rows are not contiguous", (long long)i); return FZP_EINVAL; }
        rows += (int64_t)sites[i].count[0] + sites[i].count[1];
    }
    if (rows != n_rows) { fzp_set_error("variant_map has %lld rows, sites account for %lld", (long long)n_rows, (long long)rows); return FZP_EINVAL; }
    for (int64_t i = 0; i < n_rows; i++) {
        if (vmap_qid[i] < 0) { fzp_set_error("negative q_id in variant_map"); return FZP_EINVAL; }
        max_q = std::max(max_q, vmap_qid[i]);
    }
    if (n_qid < max_q + 1) n_qid = max_q + 1;
    fzp_batch *b = new fzp_batch();
    b->n_ctg = 1;
    int64_t span = n_sites ? (int64_t)sites[n_sites - 1].pos + 1 : 0;
    b->h_goff = {0, fzp_pos_pad(span)};
    b->h_qid_off = {0, n_qid};
    b->h_limit = {(int32_t)span};
    b->h_rec_begin = {0, 0};
    b->h_site_begin = {0, n_sites};
    b->n_pos = fzp_pos_pad(span); b->n_eval = span; b->n_qid = n_qid; b->n_sites = n_sites; b->n_rows = n_rows;
    hipStream_t st = ctx->stream;
    std::vector<int64_t> site_g((size_t)n_sites);
    std::vector<int32_t> site_ctg((size_t)n_sites, 0);
    for (int64_t i = 0; i < n_sites; i++) site_g[(size_t)i] = sites[i].pos;
    int rc;
    if ((rc = upload_contig_tables(ctx, b)) || (rc = b->sites.upload(sites, (size_t)n_sites, st)) || (rc = b->site_g.upload(site_g.data(), (size_t)n_sites, st)) ||
        (rc = b->site_ctg.upload(site_ctg.data(), (size_t)n_sites, st)) || (rc = b->site_begin.upload(b->h_site_begin.data(), 2, st)) ||
        (rc = b->vmap_qid.upload(vmap_qid, (size_t)n_rows, st))) {
        delete b;
        return rc;
    }
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) { delete b; fzp_set_error("upload: %s", hipGetErrorString(e)); return FZP_EDEVICE; }
    b->have_sites = true;
    *out = b;
    return FZP_OK;
}

int batch_add_arows(fzp_ctx *ctx, fzp_batch *b, const fzp_arow *arows, int64_t n_arows) {
    if (n_arows < 0 || (n_arows && !arows)) return FZP_EINVAL;
    for (int64_t i = 0; i < n_arows; i++) {
        const fzp_arow &r = arows[i];
        if (r.site1 < 0 || r.site2 <= r.site1 || r.site2 >= b->n_sites) { fzp_set_error("atable row %lld: site indices out of order/range", (long long)i); return FZP_EINVAL; }
        if (i && (r.site1 < arows[i - 1].site1 || (r.site1 == arows[i - 1].site1 && r.site2 <= arows[i - 1].site2))) {
            fzp_set_error("atable row %lld: rows must ascend by (site1, site2)", (long long)i);
            return FZP_EINVAL;
        }
    }
    hipStream_t st = ctx->stream;
    FZP_TRY(b->arows.upload(arows, (size_t)n_arows, st));
    b->n_arows = n_arows;
    b->h_arow_begin = {0, n_arows};
    FZP_TRY(b->arow_begin.upload(b->h_arow_begin.data(), 2, st));
    FZP_HIP(hipStreamSynchronize(st));
    b->have_arows = true;
    return FZP_OK;
}
}  // namespace

extern "C" int fzp_het_call(fzp_ctx *ctx, const fzp_alnset *aln, const uint8_t *ref_seq, int64_t ref_len, fzp_site **sites, int64_t *n_sites,
                            int32_t **vmap_qid, int64_t *n_rows) {
    if (!ctx || !aln || !sites || !n_sites || !vmap_qid || !n_rows) return FZP_EINVAL;
    fzp_batch *b = nullptr;
    FZP_TRY(fzp_batch_create(ctx, 1, &aln, &ref_seq, &ref_len, &b));
    int rc = fzp_batch_run(ctx, b, FZP_STAGE_HET);
    fzp_result r;
    if (rc == FZP_OK) rc = fzp_batch_result(ctx, b, 0, &r);
    fzp_batch_destroy(ctx, b);
    if (rc) return rc;
    *sites = r.sites; *n_sites = r.n_sites; *vmap_qid = r.vmap_qid; *n_rows = r.n_rows;
    return FZP_OK;
}

extern "C" int fzp_assoc_table(fzp_ctx *ctx, const fzp_site *sites, int64_t n_sites, const int32_t *vmap_qid, int64_t n_rows, fzp_arow **arows,
                               int64_t *n_arows) {
    if (!ctx || !arows || !n_arows) return FZP_EINVAL;
    FZP_TRY(fzp_bind(ctx));
    fzp_batch *b = nullptr;
    FZP_TRY(batch_from_sites(ctx, sites, n_sites, vmap_qid, n_rows, 0, &b));
    int rc = fzp_batch_run(ctx, b, FZP_STAGE_ASSOC);
    fzp_result r;
    if (rc == FZP_OK) rc = fzp_batch_result(ctx, b, 0, &r);
    fzp_batch_destroy(ctx, b);
    if (rc) return rc;
    *arows = r.arows; *n_arows = r.n_arows;
    r.arows = nullptr;
    fzp_result_free(&r);
    return FZP_OK;
}

extern "C" int fzp_phase_blocks(fzp_ctx *ctx, const fzp_site *sites, int64_t n_sites, const fzp_arow *arows, int64_t n_arows, fzp_pvar **pvars,
                                int64_t *n_pvars) {
    if (!ctx || !pvars || !n_pvars) return FZP_EINVAL;
    FZP_TRY(fzp_bind(ctx));
    // the variant_map rows are not needed here (the reference only reads ref_base from it, phasing.py:230-238)
    std::vector<fzp_site> s2(sites, sites + n_sites);
    for (auto &s : s2) { s.row_off = 0; s.count[0] = 0; s.count[1] = 0; }
    fzp_batch *b = nullptr;
    FZP_TRY(batch_from_sites(ctx, s2.data(), n_sites, nullptr, 0, 0, &b));
    int rc = batch_add_arows(ctx, b, arows, n_arows);
    if (rc == FZP_OK) rc = fzp_batch_run(ctx, b, FZP_STAGE_BLOCKS);
    fzp_result r;
    if (rc == FZP_OK) rc = fzp_batch_result(ctx, b, 0, &r);
    fzp_batch_destroy(ctx, b);
    if (rc) return rc;
    *pvars = r.pvars; *n_pvars = r.n_pvars;
    r.pvars = nullptr;
    fzp_result_free(&r);
    return FZP_OK;
}

extern "C" int fzp_phase_reads(fzp_ctx *ctx, const fzp_site *sites, int64_t n_sites, const int32_t *vmap_qid, int64_t n_rows, const fzp_pvar *pvars,
                               int64_t n_pvars, int32_t n_qid, fzp_pread **preads, int64_t *n_preads) {
    if (!ctx || !preads || !n_preads || n_pvars < 0 || (n_pvars && !pvars)) return FZP_EINVAL;
    FZP_TRY(fzp_bind(ctx));
    fzp_batch *b = nullptr;
    FZP_TRY(batch_from_sites(ctx, sites, n_sites, vmap_qid, n_rows, n_qid, &b));
    // variant_to_phase (phasing.py:454-463): block and phase-0 allele per site
    std::vector<int32_t> blk((size_t)n_sites, 0);
    std::vector<uint8_t> b1((size_t)n_sites, 0);
    int rc = FZP_OK;
    for (int64_t i = 0; i < n_pvars && rc == FZP_OK; i++) {
        if (pvars[i].site < 0 || pvars[i].site >= n_sites || pvars[i].block <= 0) { fzp_set_error("phased variant %lld: bad site/block", (long long)i); rc = FZP_EINVAL; break; }
        blk[(size_t)pvars[i].site] = pvars[i].block;
        b1[(size_t)pvars[i].site] = pvars[i].b1;
    }
    hipStream_t st = ctx->stream;
    if (rc == FZP_OK) rc = b->site_blk.upload(blk.data(), (size_t)n_sites, st);
    if (rc == FZP_OK) rc = b->site_b1.upload(b1.data(), (size_t)n_sites, st);
    if (rc == FZP_OK && hipStreamSynchronize(st) != hipSuccess) rc = FZP_EDEVICE;
    if (rc == FZP_OK) { b->have_blocks = true; b->h_pvar_begin = {0, 0}; b->n_pvars = 0; b->have_blocks = true; }
    if (rc == FZP_OK) rc = fzp_k5_reads(ctx, b);
    fzp_result r;
    memset(&r, 0, sizeof r);
    if (rc == FZP_OK) {
        b->have_blocks = false;   // nothing to download for that stage
        rc = fzp_batch_result(ctx, b, 0, &r);
    }
    fzp_batch_destroy(ctx, b);
    if (rc) return rc;
    *preads = r.preads; *n_preads = r.n_preads;
    r.preads = nullptr;
    fzp_result_free(&r);
    return FZP_OK;
}
