// fzp_fasta.hip -- the records of a group's FASTA files found and measured ON THE DEVICE (r6).
//
// The reference's inputs are files: 3-unzip/reads/<ctg>_ref.fa and <ctg>_reads.fa (unzip.py:204,233-234), read by falcon_kit's FastaReader (phasing.py:489-494: a record
// = a '>' line, its name the header's first word, its sequence the following lines joined with the white space at their ends dropped) and by blasr.  Until r5 host threads
// looked for the line ends (memchr over every byte), built the records and handed spans to the packer; the bytes were uploaded anyway.  Now the host only READS the files
// (pread into one pinned buffer, a '\n' behind every file so that no line runs from one file into the next) and the device does the rest:
//   k_fa_nl_count / k_fa_nl_emit   where the line ends are (16 KB per workgroup; ordered by a scan over the workgroups' counts)
//   k_fa_lines                     per line: its trimmed span, "is a header" (a '>' in column 0), the file it lies in; a header's name span
//   scan of the header flags       -> every line's record
//   k_fa_heads, k_fa_seqlen        per record its header line and file; per sequence line its bases (0 for lines no header of the SAME file precedes)
//   scan of the bases              -> every line's place in its record's joined sequence
//   k_fa_recs                      per record: length; a record whose bases lie on ONE line is used where it lies (falcon_kit writes those: unzip.py:49-50), any other
//                                  (wrapped lines, blank lines inside) is joined by k_fa_join into a side buffer -- the packer (k_pack) reads begin / end pairs either way
// What comes back to the host: per record its file, its length and where its name stands (the host has the bytes: names are cut from its own copy).
// HBM-bound byte work: the buffer is read twice (count, emit) + once by the packer.
#include <algorithm>
#include "fzp_batch.h"
#include "fzp_fasta.h"

namespace {
constexpr int FA_TILE = 16384;      // bytes per workgroup of the line-end kernels
__device__ __forceinline__ bool fa_sp(uint8_t c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }
// bit k set where byte k of the 16 is '\n'
__device__ __forceinline__ uint32_t nl_mask16(const uint4 w) {
    const uint32_t x[4] = {w.x, w.y, w.z, w.w};
    uint32_t m = 0;
#pragma unroll
    for (int z = 0; z < 4; z++) {
        const uint32_t v = x[z] ^ 0x0a0a0a0au;                                   // zero bytes where '\n'
        const uint32_t t = (v & 0x7f7f7f7fu) + 0x7f7f7f7fu;
        const uint32_t hit = ~(t | v | 0x7f7f7f7fu);                               // 0x80 in every zero byte
        m |= (((hit >> 7) & 1u) | ((hit >> 14) & 2u) | ((hit >> 21) & 4u) | ((hit >> 28) & 8u)) << (4 * z);
    }
    return m;
}
__device__ __forceinline__ uint4 load16(const uint8_t *__restrict__ raw, int64_t at, int64_t n) {      // bytes [at, at + 16), zeros past n (the buffer is 16-byte aligned and padded)
    if (at + 16 <= n) return *(const uint4 *)(raw + at);
    uint32_t x[4] = {0, 0, 0, 0};
    for (int k = 0; k < 16 && at + k < n; k++) x[k >> 2] |= (uint32_t)raw[at + k] << (8 * (k & 3));
    return make_uint4(x[0], x[1], x[2], x[3]);
}
__global__ void __launch_bounds__(256) k_fa_nl_count(const uint8_t *__restrict__ raw, int64_t n, uint32_t *__restrict__ cnt) {
    __shared__ uint32_t ws[4];
    uint32_t c = 0;
    const int64_t t0 = (int64_t)blockIdx.x * FA_TILE;
    for (int it = 0; it < FA_TILE / 4096; it++) {
        const int64_t at = t0 + it * 4096 + (int64_t)threadIdx.x * 16;
        if (at < n) c += (uint32_t)__popc(nl_mask16(load16(raw, at, n)));
    }
    const int32_t s = wave_sum_i32_dpp((int32_t)c);
    if (lane_id() == 63) ws[threadIdx.x >> 6] = (uint32_t)s;
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ void __launch_bounds__(256) k_fa_nl_emit(const uint8_t *__restrict__ raw, int64_t n, const uint32_t *__restrict__ base, int64_t *__restrict__ nl) {
    __shared__ uint32_t ws[4];
    uint32_t run = base[blockIdx.x];
    const int64_t t0 = (int64_t)blockIdx.x * FA_TILE;
    for (int it = 0; it < FA_TILE / 4096; it++) {
        const int64_t at = t0 + it * 4096 + (int64_t)threadIdx.x * 16;
        uint32_t m = at < n ? nl_mask16(load16(raw, at, n)) : 0u;
        const uint32_t c = (uint32_t)__popc(m);
        const uint32_t incl = wave_incl_scan_u32_dpp(c);
        if (lane_id() == 63) ws[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (int w = 0; w < 4; w++) { if (w < (int)(threadIdx.x >> 6)) before += ws[w]; all += ws[w]; }
        uint32_t k = run + before + incl - c;
        while (m) { const int b = __builtin_ctz(m); nl[k++] = at + b; m &= m - 1; }
        run += all;
        __syncthreads();
    }
}
// per line: header? file, trimmed span (a header's: its first word)
__global__ void __launch_bounds__(256) k_fa_lines(const uint8_t *__restrict__ raw, const int64_t *__restrict__ nl, int64_t n_lines, const int64_t *__restrict__ foff, int nf,
                                                  int64_t *__restrict__ l_u, int64_t *__restrict__ l_v, uint32_t *__restrict__ l_hdr, int32_t *__restrict__ l_file) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_lines) return;
    const int64_t s = i ? nl[i - 1] + 1 : 0, e = nl[i];
    int a = 0, b = nf;                                        // the file: last t with foff[t] <= s
    while (b - a > 1) { const int m = (a + b) >> 1; if (foff[m] <= s) a = m; else b = m; }
    const bool hdr = e > s && raw[s] == '>';
    int64_t u = s, v = e;
    if (hdr) {      // the name: the first word behind '>' (falcon_kit: header.split()[0])
        u = s + 1;
        while (u < e && fa_sp(raw[u])) u++;
        v = u;
        while (v < e && !fa_sp(raw[v])) v++;
    } else {
        while (u < v && fa_sp(raw[u])) u++;
        while (v > u && fa_sp(raw[v - 1])) v--;
    }
    l_u[i] = u; l_v[i] = v; l_hdr[i] = hdr ? 1u : 0u; l_file[i] = a;
}
__global__ void __launch_bounds__(256) k_fa_heads(int64_t n_lines, const uint32_t *__restrict__ l_hdr, const uint32_t *__restrict__ l_rec, const int32_t *__restrict__ l_file,
                                                  const int64_t *__restrict__ l_u, const int64_t *__restrict__ l_v, int64_t *__restrict__ r_line, int32_t *__restrict__ r_file,
                                                  int64_t *__restrict__ r_nb, int64_t *__restrict__ r_ne) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_lines || !l_hdr[i]) return;
    const uint32_t r = l_rec[i];
    r_line[r] = i; r_file[r] = l_file[i]; r_nb[r] = l_u[i]; r_ne[r] = l_v[i];
}
__global__ void __launch_bounds__(256) k_fa_seqlen(int64_t n_lines, const uint32_t *__restrict__ l_hdr, const uint32_t *__restrict__ l_rec, const int32_t *__restrict__ l_file,
                                                   const int64_t *__restrict__ l_u, const int64_t *__restrict__ l_v, const int32_t *__restrict__ r_file, uint64_t *__restrict__ l_boff) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_lines) return;
    uint64_t len = 0;
    if (!l_hdr[i]) {
        const uint32_t before = l_rec[i];                      // headers before this line
        if (before > 0 && r_file[before - 1] == l_file[i]) len = (uint64_t)(l_v[i] - l_u[i]);      // (lines in front of a file's first header belong to nobody)
    }
    l_boff[i] = len;
}
// per record: its length, and whether its bases lie on one line (then: where)
__global__ void __launch_bounds__(256) k_fa_recs(int64_t n_rec, int64_t n_lines, const int64_t *__restrict__ r_line, const uint64_t *__restrict__ l_boff, uint64_t total,
                                                 const int64_t *__restrict__ l_u, const int64_t *__restrict__ l_v, const uint32_t *__restrict__ l_hdr,
                                                 int64_t *__restrict__ r_len, int64_t *__restrict__ be, uint64_t *__restrict__ r_join) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rec) return;
    const int64_t first = r_line[r] + 1, last = r + 1 < n_rec ? r_line[r + 1] : n_lines;
    const uint64_t b0 = first < n_lines ? l_boff[first] : total, b1 = last < n_lines ? l_boff[last] : total;
    const int64_t len = (int64_t)(b1 - b0);
    r_len[r] = len;
    int64_t at = -1;
    bool plain = true;
    if (len > 0) {      // the first line that carries bases: if it carries all of them the record is used where it lies
        for (int64_t i = first; i < last; i++) {
            const uint64_t nx = i + 1 < n_lines ? l_boff[i + 1] : total;
            if (nx > l_boff[i]) { at = l_u[i]; plain = (int64_t)(nx - l_boff[i]) == len; break; }
        }
    }
    if (len == 0) { be[2 * r] = 0; be[2 * r + 1] = 0; r_join[r] = 0; }
    else if (plain) { be[2 * r] = at; be[2 * r + 1] = at + len; r_join[r] = 0; }
    else { be[2 * r] = -1; be[2 * r + 1] = (int64_t)b0; r_join[r] = (uint64_t)len; }      // (patched by k_fa_join_be once the joined records have their places)
}
// joined records: begin / end as offsets from `raw` INTO the side buffer (the packer adds them to one base pointer: both are device allocations of one flat address space)
__global__ void __launch_bounds__(256) k_fa_join_be(int64_t n_rec, const int64_t *__restrict__ r_len, const uint64_t *__restrict__ r_joff, int64_t join_minus_raw, int64_t *__restrict__ be,
                                                    int64_t *__restrict__ r_b0) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rec) return;
    if (be[2 * r] == -1) {
        r_b0[r] = be[2 * r + 1];                               // the record's first base in the lines' numbering
        be[2 * r] = join_minus_raw + (int64_t)r_joff[r]; be[2 * r + 1] = be[2 * r] + r_len[r];
    } else r_b0[r] = -1;
}
// one wave per line of a joined record: its bases to their place
__global__ void __launch_bounds__(64) k_fa_join(const uint8_t *__restrict__ raw, int64_t n_lines, const uint32_t *__restrict__ l_hdr, const uint32_t *__restrict__ l_rec,
                                                const int64_t *__restrict__ l_u, const uint64_t *__restrict__ l_boff, uint64_t total, const int64_t *__restrict__ r_b0,
                                                const uint64_t *__restrict__ r_joff, uint8_t *__restrict__ join) {
    const int64_t i = blockIdx.x;
    if (i >= n_lines || l_hdr[i] || l_rec[i] == 0) return;
    const uint32_t r = l_rec[i] - 1;
    if (r_b0[r] < 0) return;
    const uint64_t nx = i + 1 < n_lines ? l_boff[i + 1] : total;
    const int64_t len = (int64_t)(nx - l_boff[i]);
    uint8_t *dst = join + r_joff[r] + (l_boff[i] - (uint64_t)r_b0[r]);
    const uint8_t *src = raw + l_u[i];
    for (int64_t k = threadIdx.x; k < len; k += 64) dst[k] = src[k];
}
// every record's name (the header's first word) to its place in one block: a thread a record, names are a few dozen bytes
__global__ void __launch_bounds__(256) k_fa_names(const uint8_t *__restrict__ raw, const int64_t *__restrict__ nb, const int64_t *__restrict__ noff, int64_t n, uint8_t *__restrict__ dst) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const uint8_t *s = raw + nb[r];
    uint8_t *d = dst + noff[r];
    const int64_t len = noff[r + 1] - noff[r];
    for (int64_t i = 0; i < len; i++) d[i] = s[i];
}
// records `first` .. `first + n` joined back to back (test hook: what the packer would read, as bytes)
__global__ void __launch_bounds__(256) k_fa_gather(const uint8_t *__restrict__ raw, const int64_t *__restrict__ be, int64_t first, const int64_t *__restrict__ doff, uint8_t *__restrict__ dst) {
    const int64_t r = first + blockIdx.y;
    const int64_t b = be[2 * r], n = be[2 * r + 1] - b;
    const uint8_t *s = raw + b;
    uint8_t *d = dst + doff[blockIdx.y];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d[i] = s[i];
}
}  // namespace

int fzp_fasta_fetch_seqs(fzp_ctx *ctx, hipStream_t st, const uint8_t *d_raw, const FaIndex &X, int64_t first, int64_t n, uint8_t *host_out) {
    if (n <= 0) return FZP_OK;
    std::vector<int64_t> off((size_t)n + 1, 0);
    for (int64_t k = 0; k < n; k++) off[(size_t)k + 1] = off[(size_t)k] + X.h_len[(size_t)(first + k)];
    DevBuf<int64_t> d_off;
    DevBuf<uint8_t> d_out;
    FZP_TRY(d_off.upload(off.data(), off.size(), st)); FZP_TRY(d_out.alloc((size_t)off.back() + 16));
    for (int64_t k0 = 0; k0 < n; k0 += 32768) {      // (grid.y limit)
        const int64_t m = std::min<int64_t>(32768, n - k0);
        hipLaunchKernelGGL(k_fa_gather, dim3(8, (unsigned)m), dim3(256), 0, st, d_raw, (const int64_t *)X.d_be.p, first + k0, (const int64_t *)(d_off.p + k0), d_out.p);
    }
    if (off.back()) FZP_HIP(hipMemcpyAsync(host_out, d_out.p, (size_t)off.back(), hipMemcpyDeviceToHost, st));
    FZP_HIP(hipStreamSynchronize(st));
    return FZP_OK;
}

int fzp_fasta_index_dev(fzp_ctx *ctx, hipStream_t st, const uint8_t *d_raw, int64_t n_bytes, const int64_t *foff, int nf, FaIndex &X) {
    X.n_rec = 0; X.n_lines = 0; X.join_bytes = 0;
    X.h_file.clear(); X.h_len.clear(); X.h_name_b.clear(); X.h_name_e.clear(); X.h_noff.assign(1, 0); X.h_names.clear();
    if (n_bytes <= 0 || nf <= 0) return FZP_OK;
    ProfScope ps(ctx, "fa_index");
    const int64_t n_tiles = (n_bytes + FA_TILE - 1) / FA_TILE;
    DevBuf<uint32_t> cnt, base;
    DevBuf<uint64_t> tot;
    FZP_TRY(cnt.alloc((size_t)n_tiles)); FZP_TRY(base.alloc((size_t)n_tiles)); FZP_TRY(tot.alloc(4));
    hipLaunchKernelGGL(k_fa_nl_count, dim3((unsigned)n_tiles), dim3(256), 0, st, d_raw, n_bytes, cnt.p);
    FZP_TRY(fzp_exclusive_scan_u32(ctx, cnt.p, base.p, (size_t)n_tiles, tot.p));
    uint64_t n_nl = 0;
    FZP_TRY(fzp_fetch(ctx, st, &n_nl, tot.p, sizeof(uint64_t)));
    if (n_nl == 0) return FZP_OK;
    if (n_nl >= (1ull << 32)) { fzp_set_error("FASTA group: %llu lines (limit 2^32 per group)", (unsigned long long)n_nl); return FZP_EINVAL; }
    const int64_t nL = (int64_t)n_nl;      // (the host put a '\n' behind every file: the buffer's last byte is one, every line ends in one)
    X.n_lines = nL;
    DevBuf<int64_t> nl, l_u, l_v, d_foff;
    DevBuf<uint32_t> l_hdr, l_rec;
    DevBuf<int32_t> l_file;
    DevBuf<uint64_t> l_boff;
    FZP_TRY(nl.alloc((size_t)nL)); FZP_TRY(l_u.alloc((size_t)nL)); FZP_TRY(l_v.alloc((size_t)nL)); FZP_TRY(l_hdr.alloc((size_t)nL)); FZP_TRY(l_rec.alloc((size_t)nL));
    FZP_TRY(l_file.alloc((size_t)nL)); FZP_TRY(l_boff.alloc((size_t)nL)); FZP_TRY(d_foff.upload(foff, (size_t)nf, st));
    hipLaunchKernelGGL(k_fa_nl_emit, dim3((unsigned)n_tiles), dim3(256), 0, st, d_raw, n_bytes, (const uint32_t *)base.p, nl.p);
    const unsigned gl = (unsigned)((nL + 255) / 256);
    hipLaunchKernelGGL(k_fa_lines, dim3(gl), dim3(256), 0, st, d_raw, (const int64_t *)nl.p, nL, (const int64_t *)d_foff.p, nf, l_u.p, l_v.p, l_hdr.p, l_file.p);
    FZP_TRY(fzp_exclusive_scan_u32(ctx, l_hdr.p, l_rec.p, (size_t)nL, tot.p + 1));      // l_rec[i] = headers before line i (a header line's own record index)
    uint64_t n_rec = 0;
    FZP_TRY(fzp_fetch(ctx, st, &n_rec, tot.p + 1, sizeof(uint64_t)));
    X.n_rec = (int64_t)n_rec;
    if (n_rec == 0) return FZP_OK;
    DevBuf<int64_t> r_line, r_nb, r_ne, r_len, r_b0;
    DevBuf<int32_t> r_file;
    DevBuf<uint64_t> r_join;
    const size_t nr = (size_t)n_rec;
    FZP_TRY(r_line.alloc(nr)); FZP_TRY(r_nb.alloc(nr)); FZP_TRY(r_ne.alloc(nr)); FZP_TRY(r_len.alloc(nr)); FZP_TRY(r_b0.alloc(nr)); FZP_TRY(r_file.alloc(nr)); FZP_TRY(r_join.alloc(nr));
    FZP_TRY(X.d_be.alloc(2 * nr));
    hipLaunchKernelGGL(k_fa_heads, dim3(gl), dim3(256), 0, st, nL, (const uint32_t *)l_hdr.p, (const uint32_t *)l_rec.p, (const int32_t *)l_file.p, (const int64_t *)l_u.p, (const int64_t *)l_v.p,
                       r_line.p, r_file.p, r_nb.p, r_ne.p);
    hipLaunchKernelGGL(k_fa_seqlen, dim3(gl), dim3(256), 0, st, nL, (const uint32_t *)l_hdr.p, (const uint32_t *)l_rec.p, (const int32_t *)l_file.p, (const int64_t *)l_u.p, (const int64_t *)l_v.p,
                       (const int32_t *)r_file.p, l_boff.p);
    FZP_TRY(fzp_exclusive_scan_u64_inplace(ctx, l_boff.p, (size_t)nL, tot.p + 2));
    uint64_t total = 0;
    FZP_TRY(fzp_fetch(ctx, st, &total, tot.p + 2, sizeof(uint64_t)));
    const unsigned gr = (unsigned)((nr + 255) / 256);
    hipLaunchKernelGGL(k_fa_recs, dim3(gr), dim3(256), 0, st, (int64_t)nr, nL, (const int64_t *)r_line.p, (const uint64_t *)l_boff.p, total, (const int64_t *)l_u.p, (const int64_t *)l_v.p,
                       (const uint32_t *)l_hdr.p, r_len.p, X.d_be.p, r_join.p);
    FZP_TRY(fzp_exclusive_scan_u64_inplace(ctx, r_join.p, nr, tot.p + 3));
    uint64_t jb = 0;
    FZP_TRY(fzp_fetch(ctx, st, &jb, tot.p + 3, sizeof(uint64_t)));
    X.join_bytes = (int64_t)jb;
    FZP_TRY(X.d_join.alloc((size_t)jb + 64));
    hipLaunchKernelGGL(k_fa_join_be, dim3(gr), dim3(256), 0, st, (int64_t)nr, (const int64_t *)r_len.p, (const uint64_t *)r_join.p, (int64_t)((intptr_t)X.d_join.p - (intptr_t)d_raw), X.d_be.p, r_b0.p);
    if (jb > 0)
        hipLaunchKernelGGL(k_fa_join, dim3((unsigned)nL), dim3(64), 0, st, d_raw, nL, (const uint32_t *)l_hdr.p, (const uint32_t *)l_rec.p, (const int64_t *)l_u.p, (const uint64_t *)l_boff.p, total,
                           (const int64_t *)r_b0.p, (const uint64_t *)r_join.p, X.d_join.p);
    X.h_file.resize(nr); X.h_len.resize(nr); X.h_name_b.resize(nr); X.h_name_e.resize(nr);
    FZP_TRY(r_file.download(X.h_file.data(), nr, st)); FZP_TRY(r_len.download(X.h_len.data(), nr, st));
    FZP_TRY(r_nb.download(X.h_name_b.data(), nr, st)); FZP_TRY(r_ne.download(X.h_name_e.data(), nr, st));
    FZP_HIP(hipStreamSynchronize(st));
    // the names: their lengths are on the host now; one more kernel puts them side by side, one copy brings them over (r6: the host kept the files' bytes for this alone)
    X.h_noff.assign(nr + 1, 0);
    for (size_t r = 0; r < nr; r++) {
        const int64_t l = X.h_name_e[r] - X.h_name_b[r];
        if (l < 0) { fzp_set_error("FASTA index: a record's name ends before it begins"); return FZP_EINVAL; }
        X.h_noff[r + 1] = X.h_noff[r] + l;
    }
    X.h_names.resize((size_t)X.h_noff[nr]);
    if (X.h_noff[nr] > 0) {
        DevBuf<int64_t> d_noff;
        DevBuf<uint8_t> d_names;
        FZP_TRY(d_noff.upload(X.h_noff.data(), nr + 1, st));
        FZP_TRY(d_names.alloc((size_t)X.h_noff[nr]));
        hipLaunchKernelGGL(k_fa_names, dim3(gr), dim3(256), 0, st, d_raw, (const int64_t *)r_nb.p, (const int64_t *)d_noff.p, (int64_t)nr, d_names.p);
        FZP_HIP(hipMemcpyAsync(X.h_names.data(), d_names.p, (size_t)X.h_noff[nr], hipMemcpyDeviceToHost, st));
        FZP_HIP(hipStreamSynchronize(st));
    }
    FZP_HIP(hipGetLastError());
    return FZP_OK;
}
