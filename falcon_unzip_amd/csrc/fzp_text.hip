// fzp_text.hip -- the two large text files of a phasing batch, serialised on the device.
//
// `het_call/variant_map` (phasing.py:125-128, one row per (site, read)) and `g_atable/atable` (phasing.py:199, one row
// per kept site pair) are > 90 % of the bytes phasing() writes (cfg2: 1.0 M + 0.66 M rows, ~55 MB per 40 000 reads).
// A thread owns a row: row length -> exclusive scan -> the row's characters at its offset; rows are already in the
// reference's order (they were compacted by scans), so the text is the file, contig after contig, and the host only
// cuts it at the contigs' byte offsets.  HBM-bound: 12-24 B of records in, ~25-45 B of text out per row.
#include "fzp_batch.h"

namespace {
__device__ __forceinline__ int ndig(uint32_t v) {
    return v < 10u ? 1 : v < 100u ? 2 : v < 1000u ? 3 : v < 10000u ? 4 : v < 100000u ? 5 : v < 1000000u ? 6 : v < 10000000u ? 7 : v < 100000000u ? 8 : v < 1000000000u ? 9 : 10;
}
__device__ __forceinline__ int nint(int32_t v) { return v < 0 ? 1 + ndig(0u - (uint32_t)v) : ndig((uint32_t)v); }
__device__ __forceinline__ char *put_u(char *p, uint32_t v) {
    const int n = ndig(v);
    for (int k = n - 1; k >= 0; k--) { p[k] = (char)('0' + v % 10u); v /= 10u; }
    return p + n;
}
__device__ __forceinline__ char *put_i(char *p, int32_t v) {
    if (v < 0) { *p++ = '-'; return put_u(p, 0u - (uint32_t)v); }
    return put_u(p, (uint32_t)v);
}
// site of variant_map row r: the last site whose row_off <= r
__device__ __forceinline__ int64_t site_of_row(const fzp_site *__restrict__ sites, int64_t n_sites, int64_t r) {
    int64_t lo = 0, hi = n_sites - 1;
    while (lo < hi) {
        const int64_t m = (lo + hi + 1) >> 1;
        if (sites[m].row_off <= r) lo = m; else hi = m - 1;
    }
    return lo;
}
__global__ void __launch_bounds__(256) k_vmap_len(int64_t n_rows, const fzp_site *__restrict__ sites, int64_t n_sites, const int32_t *__restrict__ qid, uint32_t *__restrict__ len) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const fzp_site &s = sites[site_of_row(sites, n_sites, r)];
    len[r] = (uint32_t)(ndig((uint32_t)s.pos + 1u) + 5 + nint(qid[r]) + 1);          // "pos R B qid\n"
}
__global__ void __launch_bounds__(256) k_vmap_put(int64_t n_rows, const fzp_site *__restrict__ sites, int64_t n_sites, const int32_t *__restrict__ qid,
                                                  const uint32_t *__restrict__ off, char *__restrict__ text) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const fzp_site &s = sites[site_of_row(sites, n_sites, r)];
    char *p = text + off[r];
    p = put_u(p, (uint32_t)s.pos + 1u);
    *p++ = ' '; *p++ = (char)s.ref_base; *p++ = ' ';
    *p++ = (char)s.base[r - s.row_off < s.count[0] ? 0 : 1];                           // major-allele rows first (phasing.py:125-128)
    *p++ = ' ';
    p = put_i(p, qid[r]);
    *p = '\n';
}
// the two alleles of a site in CPython-2.7 dict order A < C < T < G (phasing.py:175,181)
__device__ __forceinline__ void actg_pair(const fzp_site &s, char *x, char *y) {
    const char a = (char)s.base[0], c = (char)s.base[1];
    if (py2_rank((uint8_t)a) < py2_rank((uint8_t)c)) { *x = a; *y = c; } else { *x = c; *y = a; }
}
__global__ void __launch_bounds__(256) k_arow_len(int64_t n, const fzp_site *__restrict__ sites, const fzp_arow *__restrict__ rows, uint32_t *__restrict__ len) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const fzp_arow r = rows[i];
    len[i] = (uint32_t)(ndig((uint32_t)sites[r.site1].pos + 1u) + 5 + ndig((uint32_t)sites[r.site2].pos + 1u) + 4 + 4 + nint(r.n[0]) + nint(r.n[1]) + nint(r.n[2]) + nint(r.n[3]) + 1);
}
__global__ void __launch_bounds__(256) k_arow_put(int64_t n, const fzp_site *__restrict__ sites, const fzp_arow *__restrict__ rows, const uint32_t *__restrict__ off,
                                                  char *__restrict__ text) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const fzp_arow r = rows[i];
    const fzp_site &s1 = sites[r.site1], &s2 = sites[r.site2];
    char b11, b12, b21, b22;
    actg_pair(s1, &b11, &b12);
    actg_pair(s2, &b21, &b22);
    char *p = text + off[i];
    p = put_u(p, (uint32_t)s1.pos + 1u); *p++ = ' '; *p++ = b11; *p++ = ' '; *p++ = b12; *p++ = ' ';
    p = put_u(p, (uint32_t)s2.pos + 1u); *p++ = ' '; *p++ = b21; *p++ = ' '; *p++ = b22;
    for (int k = 0; k < 4; k++) { *p++ = ' '; p = put_i(p, r.n[k]); }
    *p = '\n';
}
__global__ void __launch_bounds__(256) k_pick_offsets(int n, const int64_t *__restrict__ row_begin, int64_t n_rows, const uint32_t *__restrict__ off, const uint64_t *__restrict__ total,
                                                      int64_t *__restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    const int64_t r = row_begin[c];
    out[c] = r < n_rows ? (int64_t)off[r] : (int64_t)*total;
}
}  // namespace

// what: 1 = variant_map, 2 = atable.  text: device buffer (grow-only, reused); *bytes and ctg_begin[n_ctg + 1] (byte offset of every
// contig's part) come back on the host.  One small sync.
int fzp_batch_text_dev(fzp_ctx *ctx, fzp_batch *b, int what, DevBuf<char> &text, size_t *bytes, std::vector<int64_t> &ctg_begin) {
    FZP_TRY(fzp_bind(ctx));
    hipStream_t st = ctx->stream;
    const int nc = b->n_ctg;
    ctg_begin.assign((size_t)nc + 1, 0);
    *bytes = 0;
    const int64_t n = what == 1 ? b->n_rows : b->n_arows;
    if ((what == 1 && !b->have_sites) || (what == 2 && !b->have_arows)) { fzp_set_error("fzp_batch_text: stage has not run"); return FZP_EINVAL; }
    if (n == 0) return FZP_OK;
    if (n >= (1ll << 31)) { fzp_set_error("fzp_batch_text: %lld rows (limit 2^31 per batch)", (long long)n); return FZP_EINVAL; }
    DevBuf<uint32_t> len;
    DevBuf<uint64_t> total;
    DevBuf<int64_t> d_rb, d_out;
    FZP_TRY(len.alloc((size_t)n)); FZP_TRY(total.alloc(1));
    const unsigned g = (unsigned)((n + 255) / 256);
    std::vector<int64_t> rb((size_t)nc + 1);
    {
        ProfScope ps(ctx, what == 1 ? "text_vmap" : "text_atable");
        if (what == 1) hipLaunchKernelGGL(k_vmap_len, dim3(g), dim3(256), 0, st, n, b->sites.p, b->n_sites, b->vmap_qid.p, len.p);
        else hipLaunchKernelGGL(k_arow_len, dim3(g), dim3(256), 0, st, n, b->sites.p, b->arows.p, len.p);
        FZP_TRY(fzp_exclusive_scan_u32(ctx, len.p, len.p, (size_t)n, total.p));
    }
    uint64_t tot = 0;
    // contig c's first row: variant_map -> row_off of its first site; atable -> arow_begin
    if (what == 2) { for (int c = 0; c <= nc; c++) rb[(size_t)c] = b->h_arow_begin[(size_t)c]; FZP_TRY(d_rb.upload(rb.data(), rb.size(), st)); }
    FZP_TRY(fzp_fetch(ctx, st, &tot, total.p, 8));
    if (tot >= (1ull << 32)) { fzp_set_error("fzp_batch_text: %llu bytes of text (limit 4 GiB per batch)", (unsigned long long)tot); return FZP_EINVAL; }
    if (what == 1) {
        // row_off of the contigs' first sites: read back with the sites (tiny gather through the same kernel needs them on the device)
        if (b->pf_early && b->pin) {      // the sites are (on their way) in the batch's pinned block already: fzp_batch_run's early download
            if (ctx->ev_pf_done) FZP_HIP(hipEventSynchronize(ctx->ev_pf_done)); else FZP_HIP(hipStreamSynchronize(ctx->stream2));
            const fzp_site *hs = (const fzp_site *)((const char *)b->pin + b->pf_sites);
            for (int c = 0; c <= nc; c++) rb[(size_t)c] = b->h_site_begin[(size_t)c] < b->n_sites ? hs[b->h_site_begin[(size_t)c]].row_off : b->n_rows;
        } else {
            std::vector<fzp_site> first((size_t)nc + 1);
            for (int c = 0; c <= nc; c++) {
                const int64_t s0 = b->h_site_begin[(size_t)c];
                if (s0 < b->n_sites) FZP_HIP(hipMemcpyAsync(&first[(size_t)c], b->sites.p + s0, sizeof(fzp_site), hipMemcpyDeviceToHost, st));
            }
            FZP_HIP(hipStreamSynchronize(st));
            for (int c = 0; c <= nc; c++) rb[(size_t)c] = b->h_site_begin[(size_t)c] < b->n_sites ? first[(size_t)c].row_off : b->n_rows;
        }
        FZP_TRY(d_rb.upload(rb.data(), rb.size(), st));
    }
    FZP_TRY(text.alloc((size_t)tot + 16));
    FZP_TRY(d_out.alloc((size_t)nc + 1));
    {
        ProfScope ps(ctx, what == 1 ? "text_vmap" : "text_atable");
        if (what == 1) hipLaunchKernelGGL(k_vmap_put, dim3(g), dim3(256), 0, st, n, b->sites.p, b->n_sites, b->vmap_qid.p, len.p, text.p);
        else hipLaunchKernelGGL(k_arow_put, dim3(g), dim3(256), 0, st, n, b->sites.p, b->arows.p, len.p, text.p);
        hipLaunchKernelGGL(k_pick_offsets, dim3((unsigned)((nc + 256) / 256)), dim3(256), 0, st, nc + 1, d_rb.p, n, len.p, total.p, d_out.p);
    }
    FZP_TRY(d_out.download(ctg_begin.data(), (size_t)nc + 1, st));
    FZP_HIP(hipStreamSynchronize(st));
    FZP_HIP(hipGetLastError());
    *bytes = (size_t)tot;
    return FZP_OK;
}

// Both texts with ONE wait (r5; fzp_pipe.hip).  The lengths of both are scanned, the contigs' first rows are taken from where K2 / K3 left them on the device, the contigs'
// byte offsets picked from the scans -- and only then the host asks: both totals and both offset tables in one fetch.  It sizes the two buffers and launches the kernels
// that write the text; nothing waits for those here (the caller's copies go behind them on the stream, or behind an event recorded after this returns).
namespace {
__global__ void k_vmap_row_begin(const int64_t *__restrict__ site_begin, int n_ctg, const fzp_site *__restrict__ sites, int64_t n_sites, int64_t n_rows, int64_t *__restrict__ row_begin) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_ctg) return;
    const int64_t s0 = site_begin[c];
    row_begin[c] = s0 < n_sites ? (int64_t)sites[s0].row_off : n_rows;
}
}  // namespace
int fzp_batch_texts_dev(fzp_ctx *ctx, fzp_batch *b, DevBuf<char> &t_vmap, size_t *n_vmap, std::vector<int64_t> &vb, DevBuf<char> &t_atab, size_t *n_atab, std::vector<int64_t> &ab) {
    FZP_TRY(fzp_bind(ctx));
    hipStream_t st = ctx->stream;
    const int nc = b->n_ctg;
    vb.assign((size_t)nc + 1, 0); ab.assign((size_t)nc + 1, 0);
    *n_vmap = 0; *n_atab = 0;
    if (!b->have_sites || !b->have_arows) { fzp_set_error("fzp_batch_texts: stage has not run"); return FZP_EINVAL; }
    const int64_t nv = b->n_rows, na = b->n_arows;
    if (nv >= (1ll << 31) || na >= (1ll << 31)) { fzp_set_error("fzp_batch_texts: %lld / %lld rows (limit 2^31 per batch)", (long long)nv, (long long)na); return FZP_EINVAL; }
    if ((size_t)(nc + 1) * 8 > 256) {      // many contigs: the offset tables do not fit a fetch -- one text after the other, as before
        FZP_TRY(fzp_batch_text_dev(ctx, b, FZP_TEXT_VARIANT_MAP, t_vmap, n_vmap, vb));
        return fzp_batch_text_dev(ctx, b, FZP_TEXT_ATABLE, t_atab, n_atab, ab);
    }
    DevBuf<uint32_t> len_v, len_a;
    DevBuf<uint64_t> totals;
    DevBuf<int64_t> rb_v, out;      // out: [0, nc]: variant_map's byte offsets per contig, [nc + 1, 2 nc + 1]: atable's
    FZP_TRY(len_v.alloc((size_t)std::max<int64_t>(nv, 1))); FZP_TRY(len_a.alloc((size_t)std::max<int64_t>(na, 1))); FZP_TRY(totals.alloc(2));
    FZP_TRY(rb_v.alloc((size_t)nc + 1)); FZP_TRY(out.alloc(2 * ((size_t)nc + 1)));
    const unsigned gv = (unsigned)((nv + 255) / 256), ga = (unsigned)((na + 255) / 256), gc = (unsigned)((nc + 256) / 256);
    {
        ProfScope ps(ctx, "text_vmap");
        if (nv > 0) hipLaunchKernelGGL(k_vmap_len, dim3(gv), dim3(256), 0, st, nv, b->sites.p, b->n_sites, b->vmap_qid.p, len_v.p);
        FZP_TRY(fzp_exclusive_scan_u32(ctx, len_v.p, len_v.p, (size_t)nv, totals.p + 0));
    }
    {
        ProfScope ps(ctx, "text_atable");
        if (na > 0) hipLaunchKernelGGL(k_arow_len, dim3(ga), dim3(256), 0, st, na, b->sites.p, b->arows.p, len_a.p);
        FZP_TRY(fzp_exclusive_scan_u32(ctx, len_a.p, len_a.p, (size_t)na, totals.p + 1));
    }
    hipLaunchKernelGGL(k_vmap_row_begin, dim3((nc + 1 + 63) / 64), dim3(64), 0, st, b->site_begin.p, nc, b->sites.p, b->n_sites, nv, rb_v.p);
    hipLaunchKernelGGL(k_pick_offsets, dim3(gc), dim3(256), 0, st, nc + 1, rb_v.p, nv, len_v.p, totals.p + 0, out.p);
    hipLaunchKernelGGL(k_pick_offsets, dim3(gc), dim3(256), 0, st, nc + 1, b->arow_begin.p, na, len_a.p, totals.p + 1, out.p + nc + 1);
    uint64_t tot[2] = {0, 0};
    const fzp_fetch_piece fp[3] = {{tot, totals.p, 16}, {vb.data(), out.p, ((size_t)nc + 1) * 8}, {ab.data(), out.p + nc + 1, ((size_t)nc + 1) * 8}};
    FZP_TRY(fzp_fetch(ctx, st, fp, 3));
    if (tot[0] >= (1ull << 32) || tot[1] >= (1ull << 32)) { fzp_set_error("fzp_batch_texts: %llu / %llu bytes of text (limit 4 GiB per batch)", (unsigned long long)tot[0], (unsigned long long)tot[1]); return FZP_EINVAL; }
    FZP_TRY(t_vmap.alloc((size_t)tot[0] + 16)); FZP_TRY(t_atab.alloc((size_t)tot[1] + 16));
    if (nv > 0) { ProfScope ps(ctx, "text_vmap"); hipLaunchKernelGGL(k_vmap_put, dim3(gv), dim3(256), 0, st, nv, b->sites.p, b->n_sites, b->vmap_qid.p, len_v.p, t_vmap.p); }
    if (na > 0) { ProfScope ps(ctx, "text_atable"); hipLaunchKernelGGL(k_arow_put, dim3(ga), dim3(256), 0, st, na, b->sites.p, b->arows.p, len_a.p, t_atab.p); }
    FZP_HIP(hipGetLastError());
    *n_vmap = (size_t)tot[0]; *n_atab = (size_t)tot[1];
    return FZP_OK;      // (len_v / len_a go back to the pool here: whatever takes them next is launched on this stream, behind the kernels that read them)
}

extern "C" int fzp_batch_text(fzp_ctx *ctx, fzp_batch *b, int what, char **text, size_t *len, int64_t **ctg_begin) {
    if (!ctx || !b || !text || !len || (what != FZP_TEXT_VARIANT_MAP && what != FZP_TEXT_ATABLE)) { fzp_set_error("fzp_batch_text: bad arguments"); return FZP_EINVAL; }
    DevBuf<char> d;
    size_t n = 0;
    std::vector<int64_t> cb;
    FZP_TRY(fzp_batch_text_dev(ctx, b, what, d, &n, cb));
    char *h = (char *)malloc(n + 1);
    if (!h) return FZP_ENOMEM;
    if (n) {
        if (hipMemcpyAsync(h, d.p, n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { free(h); fzp_set_error("fzp_batch_text: download failed"); return FZP_EDEVICE; }
    }
    h[n] = 0;
    *text = h; *len = n;
    if (ctg_begin) {
        *ctg_begin = (int64_t *)malloc(cb.size() * sizeof(int64_t));
        if (!*ctg_begin) { free(h); return FZP_ENOMEM; }
        memcpy(*ctg_begin, cb.data(), cb.size() * sizeof(int64_t));
    }
    return FZP_OK;
}
