// fzp_bam.hip -- BAM / BGZF / BAI emitter and reader (host code).
//
// The reference's blasr task leaves `<ctg>_sorted.bam` + index behind (unzip.py:86-91) and make_het_call reads it back
// through `samtools view <bam> <ctg>` (phasing.py:27,42-59).  K1 hands its records to K2 without leaving HBM, so these
// files are only for users who want the alignments -- but they should exist and be real BAM: fzp_format_bam writes a
// coordinate-sorted BAM (BGZF blocks <= 64 KiB, EOF marker) and its .bai (binning + 16 kb linear index); fzp_bam_to_sam
// is the `samtools view [region]` role for installations without samtools (text lines of the 11 mandatory columns).
// Formats per the SAM/BAM specification (SAMv1 sections 4.1, 4.2, 5.2); zlib does the deflate.
#include <zlib.h>
#include <algorithm>
#include <cerrno>
#include <unordered_map>
#include "fzp_common.h"

namespace {
struct Bytes {
    std::vector<uint8_t> v;
    void u8(uint8_t x) { v.push_back(x); }
    void u16(uint16_t x) { v.push_back((uint8_t)x); v.push_back((uint8_t)(x >> 8)); }
    void u32(uint32_t x) { for (int i = 0; i < 4; i++) v.push_back((uint8_t)(x >> (8 * i))); }
    void i32(int32_t x) { u32((uint32_t)x); }
    void u64(uint64_t x) { for (int i = 0; i < 8; i++) v.push_back((uint8_t)(x >> (8 * i))); }
    void raw(const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; v.insert(v.end(), b, b + n); }
};

// SAMv1 5.3: bin of a zero-based half-open interval
inline int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

struct Bgzf {
    Bytes out;
    std::vector<uint8_t> pend;            // uncompressed bytes of the block being filled
    static constexpr size_t BLOCK = 0xff00;
    uint64_t tell() const { return ((uint64_t)out.v.size() << 16) | (uint64_t)pend.size(); }
    int flush() {
        if (pend.empty()) return FZP_OK;
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { fzp_set_error("deflateInit2 failed"); return FZP_ENOMEM; }
        std::vector<uint8_t> comp(deflateBound(&zs, (uLong)pend.size()) + 16);
        zs.next_in = pend.data(); zs.avail_in = (uInt)pend.size();
        zs.next_out = comp.data(); zs.avail_out = (uInt)comp.size();
        const int rc = deflate(&zs, Z_FINISH);
        const size_t clen = comp.size() - zs.avail_out;
        deflateEnd(&zs);
        if (rc != Z_STREAM_END || clen + 26 > 65536) { fzp_set_error("BGZF block does not fit (%zu compressed bytes)", clen); return FZP_EINVAL; }
        static const uint8_t hdr[12] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0};
        out.raw(hdr, 12);
        out.u8('B'); out.u8('C'); out.u16(2); out.u16((uint16_t)(clen + 25));      // BSIZE = block size - 1
        out.raw(comp.data(), clen);
        out.u32((uint32_t)crc32(crc32(0L, Z_NULL, 0), pend.data(), (uInt)pend.size()));
        out.u32((uint32_t)pend.size());
        pend.clear();
        return FZP_OK;
    }
    int write(const void *p, size_t n) {
        const uint8_t *b = (const uint8_t *)p;
        while (n) {
            const size_t k = std::min(n, BLOCK - pend.size());
            pend.insert(pend.end(), b, b + k);
            b += k; n -= k;
            if (pend.size() == BLOCK) FZP_TRY(flush());
        }
        return FZP_OK;
    }
    int finish() {
        FZP_TRY(flush());
        static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        out.raw(eof, 28);
        return FZP_OK;
    }
};

inline uint8_t nib(uint8_t c) {
    switch (c) {
        case 'A': case 'a': return 1;
        case 'C': case 'c': return 2;
        case 'G': case 'g': return 4;
        case 'T': case 't': return 8;
        case '=': return 0;
        default: return 15;
    }
}
int give(const std::vector<uint8_t> &v, uint8_t **out, size_t *len) {
    uint8_t *p = (uint8_t *)malloc(v.size() ? v.size() : 1);
    if (!p) return FZP_ENOMEM;
    if (!v.empty()) memcpy(p, v.data(), v.size());
    *out = p; *len = v.size();
    return FZP_OK;
}
}  // namespace

extern "C" int fzp_format_bam(const fzp_alnset *a, const char *ctg_id, int64_t ctg_len, const int32_t *flags, uint8_t **bam, size_t *bam_len, uint8_t **bai,
                              size_t *bai_len) {
    if (!a || !ctg_id || !bam || !bam_len || ctg_len < 0 || ctg_len > 0x7fffffff) { fzp_set_error("fzp_format_bam: bad arguments"); return FZP_EINVAL; }
    Bgzf z;
    {   // header
        Bytes h;
        char text[512];
        const int tl = snprintf(text, sizeof text, "@HD\tVN:1.5\tSO:coordinate\n@SQ\tSN:%s\tLN:%lld\n@PG\tID:fzphase\tPN:fzphase\n", ctg_id, (long long)ctg_len);
        if (tl <= 0 || tl >= (int)sizeof text) { fzp_set_error("contig id too long for the BAM header"); return FZP_EINVAL; }
        h.raw("BAM\1", 4); h.i32(tl); h.raw(text, (size_t)tl);
        h.i32(1); h.i32((int32_t)strlen(ctg_id) + 1); h.raw(ctg_id, strlen(ctg_id) + 1); h.i32((int32_t)ctg_len);
        FZP_TRY(z.write(h.v.data(), h.v.size()));
    }
    // index state (one reference)
    std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
    std::vector<uint64_t> linear;
    uint64_t first_off = 0, last_off = 0;
    int64_t n_mapped = 0;
    int32_t prev_pos = -1;
    for (int64_t r = 0; r < a->n_rec; r++) {
        const int32_t q = a->rec_qid[r];
        const size_t nl = (size_t)(a->qname_off[q + 1] - a->qname_off[q]);
        const int64_t sl = a->seq_off[r + 1] - a->seq_off[r];
        const int64_t nc = a->cig_off[r + 1] - a->cig_off[r];
        const int32_t pos = a->rec_pos[r];
        if (nl + 1 > 255 || pos < prev_pos) { fzp_set_error("record %lld: name longer than 254 or not coordinate-sorted", (long long)r); return FZP_EINVAL; }
        const bool long_cigar = nc > 65535;           // SAMv1 4.2.2: the real CIGAR goes into a CG:B,I tag, the CIGAR field holds <l_seq>S<ref_len>N
        prev_pos = pos;
        int64_t rlen = 0;
        for (int64_t k = a->cig_off[r]; k < a->cig_off[r + 1]; k++) {
            const uint32_t op = a->cigar[k] & 15u;
            if (op == FZP_OP_M || op == FZP_OP_D || op == FZP_OP_N || op == FZP_OP_EQ || op == FZP_OP_X) rlen += a->cigar[k] >> 4;
        }
        const int64_t end = pos + (rlen > 0 ? rlen : 1);
        const int bin = reg2bin(pos, end);
        Bytes rec;
        rec.i32(0);                                   // block_size, patched below
        rec.i32(0); rec.i32(pos);
        rec.u8((uint8_t)(nl + 1)); rec.u8(254); rec.u16((uint16_t)bin); rec.u16((uint16_t)(long_cigar ? 2 : nc)); rec.u16((uint16_t)(flags ? flags[r] : 0));
        rec.i32((int32_t)sl); rec.i32(-1); rec.i32(-1); rec.i32(0);
        rec.raw(a->qnames + a->qname_off[q], nl); rec.u8(0);
        if (long_cigar) { rec.u32(((uint32_t)sl << 4) | FZP_OP_S); rec.u32(((uint32_t)(rlen > 0 ? rlen : 0) << 4) | FZP_OP_N); }
        else for (int64_t k = a->cig_off[r]; k < a->cig_off[r + 1]; k++) rec.u32(a->cigar[k]);
        const uint8_t *sq = a->seq + a->seq_off[r];
        for (int64_t k = 0; k < sl; k += 2) rec.u8((uint8_t)((nib(sq[k]) << 4) | (k + 1 < sl ? nib(sq[k + 1]) : 0)));
        for (int64_t k = 0; k < sl; k++) rec.u8(0xff);
        if (long_cigar) {
            rec.u8('C'); rec.u8('G'); rec.u8('B'); rec.u8('I'); rec.i32((int32_t)nc);
            for (int64_t k = a->cig_off[r]; k < a->cig_off[r + 1]; k++) rec.u32(a->cigar[k]);
        }
        const uint32_t bs = (uint32_t)rec.v.size() - 4;
        for (int i = 0; i < 4; i++) rec.v[(size_t)i] = (uint8_t)(bs >> (8 * i));
        const uint64_t v0 = z.tell();
        FZP_TRY(z.write(rec.v.data(), rec.v.size()));
        const uint64_t v1 = z.tell();
        // index: chunk list per bin (adjacent records of a bin merge), linear index per 16 kb window
        auto &ch = bins[(uint32_t)bin];
        if (!ch.empty() && ch.back().second == v0) ch.back().second = v1; else ch.push_back({v0, v1});
        for (int64_t w = pos >> 14; w <= (end - 1) >> 14; w++) {
            if ((size_t)w >= linear.size()) linear.resize((size_t)w + 1, 0);
            if (linear[(size_t)w] == 0) linear[(size_t)w] = v0;
        }
        if (n_mapped == 0) first_off = v0;
        last_off = v1;
        n_mapped++;
    }
    FZP_TRY(z.finish());
    FZP_TRY(give(z.out.v, bam, bam_len));
    if (bai && bai_len) {
        // a record's virtual offset recorded while its block was still open stays valid: the block's file offset is
        // fixed at that moment (blocks are emitted in order), only a write that exactly fills a block moves `tell()`
        // to the next block's start -- equally valid as an end offset.
        Bytes ix;
        ix.raw("BAI\1", 4); ix.i32(1);
        ix.i32((int32_t)bins.size() + (n_mapped ? 1 : 0));
        for (auto &kv : bins) {
            ix.u32(kv.first); ix.i32((int32_t)kv.second.size());
            for (auto &c : kv.second) { ix.u64(c.first); ix.u64(c.second); }
        }
        if (n_mapped) {   // samtools' metadata pseudo-bin
            ix.u32(37450); ix.i32(2); ix.u64(first_off); ix.u64(last_off); ix.u64((uint64_t)n_mapped); ix.u64(0);
        }
        for (size_t w = linear.size(); w-- > 1;) if (linear[w - 1] == 0) linear[w - 1] = linear[w];   // empty windows take the next one's offset (htslib)
        ix.i32((int32_t)linear.size());
        for (uint64_t x : linear) ix.u64(x);
        ix.u64(0);                                    // n_no_coor
        int rc = give(ix.v, bai, bai_len);
        if (rc) { free(*bam); *bam = nullptr; return rc; }
    }
    return FZP_OK;
}

// `samtools view <bam> [region]`: one text line per record (11 mandatory columns, optional fields dropped -- the
// phasing code reads columns 0, 1, 2, 3, 5, 9 only, phasing.py:47-59).  region: NULL = every record, else RNAME.
namespace {
// BGZF -> one byte stream (every block checked: header, BSIZE, ISIZE <= 64 KiB, CRC)
int bgzf_inflate_all(const uint8_t *bam, size_t len, std::vector<uint8_t> &d) {
    size_t p = 0;
    while (p < len) {
        if (len - p < 18 || bam[p] != 0x1f || bam[p + 1] != 0x8b || bam[p + 2] != 8 || !(bam[p + 3] & 4)) { fzp_set_error("not a BGZF block at offset %zu", p); return FZP_EINVAL; }
        const size_t xlen = bam[p + 10] | (bam[p + 11] << 8);
        size_t bsize = 0;
        for (size_t x = p + 12; x + 4 <= p + 12 + xlen && x + 4 <= len;) {
            const size_t slen = bam[x + 2] | (bam[x + 3] << 8);
            if (bam[x] == 'B' && bam[x + 1] == 'C' && slen == 2 && x + 6 <= len) bsize = (size_t)(bam[x + 4] | (bam[x + 5] << 8)) + 1;
            x += 4 + slen;
        }
        if (!bsize || p + bsize > len || bsize < 12 + xlen + 8) { fzp_set_error("truncated BGZF block at offset %zu", p); return FZP_EINVAL; }
        const uint8_t *cd = bam + p + 12 + xlen;
        const size_t clen = bsize - 12 - xlen - 8;
        const uint8_t *tail = bam + p + bsize - 8;
        const uint32_t crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
        const uint32_t isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
        if (isize > 65536) { fzp_set_error("BGZF block at offset %zu claims %u bytes (limit 65536)", p, isize); return FZP_EINVAL; }
        if (isize) {
            const size_t at = d.size();
            d.resize(at + isize);
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) { fzp_set_error("inflateInit2 failed"); return FZP_ENOMEM; }
            zs.next_in = (Bytef *)cd; zs.avail_in = (uInt)clen;
            zs.next_out = d.data() + at; zs.avail_out = isize;
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END || zs.avail_out != 0 || (uint32_t)crc32(crc32(0L, Z_NULL, 0), d.data() + at, isize) != crc) { fzp_set_error("corrupt BGZF block at offset %zu", p); return FZP_EINVAL; }
        }
        p += bsize;
    }
    return FZP_OK;
}
}  // namespace

extern "C" int fzp_bam_to_sam(const uint8_t *bam, size_t len, const char *region, char **text, size_t *text_len) {
    if ((!bam && len) || !text || !text_len) { fzp_set_error("fzp_bam_to_sam: bad arguments"); return FZP_EINVAL; }
    std::vector<uint8_t> d;
    FZP_TRY(bgzf_inflate_all(bam, len, d));
    // ---- BAM
    auto rd32 = [&](size_t o) { return (int32_t)(d[o] | (d[o + 1] << 8) | (d[o + 2] << 16) | ((uint32_t)d[o + 3] << 24)); };
    if (d.size() < 12 || memcmp(d.data(), "BAM\1", 4) != 0) { fzp_set_error("not a BAM stream"); return FZP_EINVAL; }
    size_t o = 4;
    const int32_t l_text = rd32(o); o += 4;
    if (l_text < 0 || o + (size_t)l_text + 4 > d.size()) { fzp_set_error("truncated BAM header"); return FZP_EINVAL; }
    o += (size_t)l_text;
    const int32_t n_ref = rd32(o); o += 4;
    std::vector<std::string> names;
    for (int32_t i = 0; i < n_ref; i++) {
        if (o + 4 > d.size()) { fzp_set_error("truncated BAM reference list"); return FZP_EINVAL; }
        const int32_t ln = rd32(o); o += 4;
        if (ln < 1 || o + (size_t)ln + 4 > d.size()) { fzp_set_error("truncated BAM reference list"); return FZP_EINVAL; }
        names.emplace_back((const char *)d.data() + o, (size_t)ln - 1);
        o += (size_t)ln + 4;
    }
    int32_t want = -2;                                  // -2: all
    if (region && *region) {
        want = -3;                                      // named but absent: nothing matches
        for (int32_t i = 0; i < n_ref; i++) if (names[(size_t)i] == region) want = i;
    }
    std::string out;
    static const char OPS[] = "MIDNSHP=X???????";
    static const char NIB[] = "=ACMGRSVTWYHKDBN";
    char num[32];
    while (o + 4 <= d.size()) {
        const int32_t bs = rd32(o); o += 4;
        if (bs < 32 || o + (size_t)bs > d.size()) { fzp_set_error("truncated BAM record"); return FZP_EINVAL; }
        const size_t r0 = o;
        o += (size_t)bs;
        const int32_t ref = rd32(r0), pos = rd32(r0 + 4);
        const uint32_t l_name = d[r0 + 8], mapq = d[r0 + 9];
        const uint32_t n_cig = d[r0 + 12] | (d[r0 + 13] << 8), flag = d[r0 + 14] | (d[r0 + 15] << 8);
        const int32_t l_seq = rd32(r0 + 16), nref = rd32(r0 + 20), npos = rd32(r0 + 24), tlen = rd32(r0 + 28);
        if (l_seq < 0 || 32 + (size_t)l_name + 4 * (size_t)n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq > (size_t)bs || l_name < 1) { fzp_set_error("malformed BAM record"); return FZP_EINVAL; }
        if (want != -2 && ref != want) continue;
        out.append((const char *)d.data() + r0 + 32, l_name - 1); out.push_back('\t');
        out.append(num, (size_t)snprintf(num, sizeof num, "%u\t", flag));
        if (ref >= 0 && ref < n_ref) out.append(names[(size_t)ref]); else out.push_back('*');
        out.append(num, (size_t)snprintf(num, sizeof num, "\t%d\t%u\t", pos + 1, mapq));
        const size_t c0 = r0 + 32 + l_name;
        size_t cg0 = c0;
        uint32_t cg_n = n_cig;
        if (n_cig == 2) {     // a CIGAR of more than 65535 ops lives in the CG:B,I tag (SAMv1 4.2.2): it is the first tag as written by fzp_format_bam
            const size_t t0 = c0 + 8 + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
            const uint32_t w0 = (uint32_t)rd32(c0), w1 = (uint32_t)rd32(c0 + 4);
            if ((w0 & 15) == FZP_OP_S && (int32_t)(w0 >> 4) == l_seq && (w1 & 15) == FZP_OP_N && t0 + 8 <= r0 + (size_t)bs && d[t0] == 'C' && d[t0 + 1] == 'G' &&
                d[t0 + 2] == 'B' && d[t0 + 3] == 'I') {
                const int32_t cnt = rd32(t0 + 4);
                if (cnt > 0 && t0 + 8 + 4 * (size_t)cnt <= r0 + (size_t)bs) { cg0 = t0 + 8; cg_n = (uint32_t)cnt; }
            }
        }
        if (!cg_n) out.push_back('*');
        for (uint32_t k = 0; k < cg_n; k++) {
            const uint32_t w = (uint32_t)rd32(cg0 + 4 * k);
            out.append(num, (size_t)snprintf(num, sizeof num, "%u%c", w >> 4, OPS[w & 15]));
        }
        out.push_back('\t');
        if (nref < 0) out.push_back('*'); else if (nref == ref) out.push_back('='); else if (nref < n_ref) out.append(names[(size_t)nref]); else out.push_back('*');
        out.append(num, (size_t)snprintf(num, sizeof num, "\t%d\t%d\t", npos + 1, tlen));
        const size_t s0 = c0 + 4 * (size_t)n_cig;
        if (!l_seq) out.push_back('*');
        for (int32_t k = 0; k < l_seq; k++) out.push_back(NIB[(d[s0 + (size_t)k / 2] >> ((k & 1) ? 0 : 4)) & 15]);
        out.push_back('\t');
        const size_t q0 = s0 + ((size_t)l_seq + 1) / 2;
        if (!l_seq || d[q0] == 0xff) out.push_back('*');
        else for (int32_t k = 0; k < l_seq; k++) out.push_back((char)(d[q0 + (size_t)k] + 33));
        out.push_back('\n');
    }
    char *t = (char *)malloc(out.size() + 1);
    if (!t) return FZP_ENOMEM;
    memcpy(t, out.data(), out.size());
    t[out.size()] = 0;
    *text = t; *text_len = out.size();
    return FZP_OK;
}

// ---- raw record access: what select_reads_from_bam.py does through pysam (open every input BAM, route whole records by
// read name into per-contig BAM files under one merged header, select_reads_from_bam.py:44-87)
extern "C" int fzp_bam_open(const uint8_t *bam, size_t len, fzp_bam_view **out) {
    if ((!bam && len) || !out) { fzp_set_error("fzp_bam_open: bad arguments"); return FZP_EINVAL; }
    *out = nullptr;
    std::vector<uint8_t> d;
    FZP_TRY(bgzf_inflate_all(bam, len, d));
    auto rd32 = [&](size_t o) { return (int32_t)(d[o] | (d[o + 1] << 8) | (d[o + 2] << 16) | ((uint32_t)d[o + 3] << 24)); };
    if (d.size() < 12 || memcmp(d.data(), "BAM\1", 4) != 0) { fzp_set_error("not a BAM stream"); return FZP_EINVAL; }
    size_t o = 4;
    const int32_t l_text = rd32(o); o += 4;
    if (l_text < 0 || o + (size_t)l_text + 4 > d.size()) { fzp_set_error("truncated BAM header"); return FZP_EINVAL; }
    const size_t text_at = o;
    o += (size_t)l_text;
    const int32_t n_ref = rd32(o); o += 4;
    const size_t ref_at = o;
    for (int32_t i = 0; i < n_ref; i++) {
        if (o + 4 > d.size()) { fzp_set_error("truncated BAM reference list"); return FZP_EINVAL; }
        const int32_t ln = rd32(o); o += 4;
        if (ln < 1 || o + (size_t)ln + 4 > d.size()) { fzp_set_error("truncated BAM reference list"); return FZP_EINVAL; }
        o += (size_t)ln + 4;
    }
    const size_t rec_at = o;
    std::vector<int64_t> rec_off(1, 0), name_off(1, 0);
    std::string names;
    while (o + 4 <= d.size()) {
        const int32_t bs = rd32(o);
        if (bs < 32 || o + 4 + (size_t)bs > d.size()) { fzp_set_error("truncated BAM record"); return FZP_EINVAL; }
        const uint32_t l_name = d[o + 4 + 8];
        if (l_name < 1 || 32 + (size_t)l_name > (size_t)bs) { fzp_set_error("malformed BAM record"); return FZP_EINVAL; }
        names.append((const char *)d.data() + o + 4 + 32, l_name - 1);
        name_off.push_back((int64_t)names.size());
        o += 4 + (size_t)bs;
        rec_off.push_back((int64_t)(o - rec_at));
    }
    if (o != d.size()) { fzp_set_error("trailing bytes after the last BAM record"); return FZP_EINVAL; }
    fzp_bam_view *v = (fzp_bam_view *)calloc(1, sizeof(fzp_bam_view));
    if (!v) return FZP_ENOMEM;
    auto dup = [](const void *src, size_t n) { void *q = malloc(n ? n : 1); if (q && n) memcpy(q, src, n); return q; };
    size_t tl = (size_t)l_text;
    while (tl && d[text_at + tl - 1] == 0) tl--;                       // l_text may count NUL padding
    v->header_text = (char *)dup(d.data() + text_at, tl); v->header_len = tl;
    v->n_ref = n_ref;
    v->ref_block = (uint8_t *)dup(d.data() + ref_at, rec_at - ref_at); v->ref_block_len = rec_at - ref_at;
    v->n_rec = (int64_t)rec_off.size() - 1;
    v->rec_off = (int64_t *)dup(rec_off.data(), rec_off.size() * 8);
    v->records = (uint8_t *)dup(d.data() + rec_at, d.size() - rec_at); v->records_len = d.size() - rec_at;
    v->name_off = (int64_t *)dup(name_off.data(), name_off.size() * 8);
    v->names = (char *)dup(names.data(), names.size());
    if (!v->header_text || !v->ref_block || !v->rec_off || !v->records || !v->name_off || !v->names) { fzp_bam_view_free(v); return FZP_ENOMEM; }
    *out = v;
    return FZP_OK;
}

extern "C" void fzp_bam_view_free(fzp_bam_view *v) {
    if (!v) return;
    free(v->header_text); free(v->ref_block); free(v->rec_off); free(v->records); free(v->name_off); free(v->names);
    free(v);
}

extern "C" int fzp_bam_write(const char *header_text, size_t header_len, int32_t n_ref, const uint8_t *ref_block, size_t ref_block_len, int32_t n_parts,
                             const uint8_t *const *parts, const size_t *part_lens, uint8_t **bam, size_t *bam_len) {
    if ((!header_text && header_len) || n_ref < 0 || (!ref_block && ref_block_len) || n_parts < 0 || (n_parts && (!parts || !part_lens)) || !bam || !bam_len) {
        fzp_set_error("fzp_bam_write: bad arguments");
        return FZP_EINVAL;
    }
    Bgzf z;
    Bytes h;
    h.raw("BAM\1", 4); h.u32((uint32_t)header_len); h.raw(header_text, header_len); h.u32((uint32_t)n_ref); h.raw(ref_block, ref_block_len);
    FZP_TRY(z.write(h.v.data(), h.v.size()));
    FZP_TRY(z.flush());                                                  // records start on a block boundary, as samtools writes them
    for (int32_t k = 0; k < n_parts; k++) {
        // whole records only: walk the block_size prefixes
        size_t o = 0;
        while (o < part_lens[k]) {
            if (o + 4 > part_lens[k]) { fzp_set_error("fzp_bam_write: part %d is not a sequence of BAM records", k); return FZP_EINVAL; }
            const uint8_t *q = parts[k] + o;
            const size_t bs = (size_t)(q[0] | (q[1] << 8) | (q[2] << 16) | ((uint32_t)q[3] << 24));
            if (bs < 32 || o + 4 + bs > part_lens[k]) { fzp_set_error("fzp_bam_write: part %d is not a sequence of BAM records", k); return FZP_EINVAL; }
            o += 4 + bs;
        }
        if (part_lens[k]) FZP_TRY(z.write(parts[k], part_lens[k]));
    }
    FZP_TRY(z.finish());
    return give(z.out.v, bam, bam_len);
}

// ---- streaming routing of whole records from many BAM files into per-destination BAM files (select_reads_from_bam.py:69-90 does this
// record by record through pysam).  Memory is bounded by one BGZF block per open input and one pending block per destination: input
// files of any size go through; a destination's bytes are appended to its file as its blocks fill (opened per append: no descriptor is
// held between blocks, so thousands of destinations are fine).
namespace {
struct BgzfIn {
    FILE *f = nullptr;
    std::string path;
    std::vector<uint8_t> buf, blk;       // decoded bytes not yet consumed (from `at`), one compressed block
    size_t at = 0;
    bool eof = false;
    ~BgzfIn() { if (f) fclose(f); }
    int open(const char *p) {
        path = p;
        f = fopen(p, "rb");
        if (!f) { fzp_set_error("cannot open %s: %s", p, strerror(errno)); return FZP_EIO; }
        return FZP_OK;
    }
    // decode the next block behind the unconsumed bytes
    int fill() {
        if (at) { buf.erase(buf.begin(), buf.begin() + (long)at); at = 0; }
        uint8_t h[12];
        const size_t got = fread(h, 1, 12, f);
        if (got == 0) { eof = true; return FZP_OK; }
        if (got < 12 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) { fzp_set_error("%s: not a BGZF block", path.c_str()); return FZP_EINVAL; }
        const size_t xlen = h[10] | (h[11] << 8);
        blk.resize(xlen);
        if (fread(blk.data(), 1, xlen, f) != xlen) { fzp_set_error("%s: truncated BGZF block", path.c_str()); return FZP_EINVAL; }
        size_t bsize = 0;
        for (size_t x = 0; x + 4 <= xlen;) {
            const size_t slen = blk[x + 2] | (blk[x + 3] << 8);
            if (blk[x] == 'B' && blk[x + 1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = (size_t)(blk[x + 4] | (blk[x + 5] << 8)) + 1;
            x += 4 + slen;
        }
        if (bsize < 12 + xlen + 8) { fzp_set_error("%s: BGZF block without a BC field", path.c_str()); return FZP_EINVAL; }
        const size_t rest = bsize - 12 - xlen;
        blk.resize(rest);
        if (fread(blk.data(), 1, rest, f) != rest) { fzp_set_error("%s: truncated BGZF block", path.c_str()); return FZP_EINVAL; }
        const uint8_t *tail = blk.data() + rest - 8;
        const uint32_t crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
        const uint32_t isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
        if (isize > 65536) { fzp_set_error("%s: BGZF block claims %u bytes (limit 65536)", path.c_str(), isize); return FZP_EINVAL; }
        if (!isize) return FZP_OK;
        const size_t o = buf.size();
        buf.resize(o + isize);
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) { fzp_set_error("inflateInit2 failed"); return FZP_ENOMEM; }
        zs.next_in = blk.data(); zs.avail_in = (uInt)(rest - 8);
        zs.next_out = buf.data() + o; zs.avail_out = isize;
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END || zs.avail_out != 0 || (uint32_t)crc32(crc32(0L, Z_NULL, 0), buf.data() + o, isize) != crc) { fzp_set_error("%s: corrupt BGZF block", path.c_str()); return FZP_EINVAL; }
        return FZP_OK;
    }
    // at least n unconsumed bytes, unless the file ends first
    int need(size_t n) {
        while (buf.size() - at < n && !eof) FZP_TRY(fill());
        return FZP_OK;
    }
    size_t have() const { return buf.size() - at; }
    const uint8_t *ptr() const { return buf.data() + at; }
    static int32_t rd32(const uint8_t *p) { return (int32_t)(p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24)); }
    // magic, header text, reference block; leaves the stream at the first record
    int header(std::string &text, int32_t &n_ref, std::vector<uint8_t> &refs) {
        FZP_TRY(need(12));
        if (have() < 12 || memcmp(ptr(), "BAM\1", 4) != 0) { fzp_set_error("%s: not a BAM stream", path.c_str()); return FZP_EINVAL; }
        const int32_t l_text = rd32(ptr() + 4);
        if (l_text < 0) { fzp_set_error("%s: bad BAM header", path.c_str()); return FZP_EINVAL; }
        FZP_TRY(need(8 + (size_t)l_text + 4));
        if (have() < 8 + (size_t)l_text + 4) { fzp_set_error("%s: truncated BAM header", path.c_str()); return FZP_EINVAL; }
        size_t tl = (size_t)l_text;
        while (tl && ptr()[8 + tl - 1] == 0) tl--;                      // l_text may count NUL padding
        text.assign((const char *)ptr() + 8, tl);
        n_ref = rd32(ptr() + 8 + l_text);
        at += 8 + (size_t)l_text + 4;
        refs.clear();
        for (int32_t i = 0; i < n_ref; i++) {
            FZP_TRY(need(4));
            if (have() < 4) { fzp_set_error("%s: truncated BAM reference list", path.c_str()); return FZP_EINVAL; }
            const int32_t ln = rd32(ptr());
            if (ln < 1) { fzp_set_error("%s: bad BAM reference list", path.c_str()); return FZP_EINVAL; }
            FZP_TRY(need(8 + (size_t)ln));
            if (have() < 8 + (size_t)ln) { fzp_set_error("%s: truncated BAM reference list", path.c_str()); return FZP_EINVAL; }
            refs.insert(refs.end(), ptr(), ptr() + 8 + ln);
            at += 8 + (size_t)ln;
        }
        return FZP_OK;
    }
};

struct BamOut {
    std::string path;
    Bgzf z;
    bool created = false;
    int64_t n_rec = 0;
    int spill() {
        if (z.out.v.empty()) return FZP_OK;
        FILE *f = fopen(path.c_str(), created ? "ab" : "wb");
        if (!f) { fzp_set_error("cannot write %s: %s", path.c_str(), strerror(errno)); return FZP_EIO; }
        const bool ok = fwrite(z.out.v.data(), 1, z.out.v.size(), f) == z.out.v.size();
        const int e = errno;
        if (fclose(f) != 0 || !ok) { fzp_set_error("write to %s failed: %s", path.c_str(), strerror(ok ? errno : e)); return FZP_EIO; }
        created = true;
        z.out.v.clear();
        return FZP_OK;
    }
};
}  // namespace

extern "C" int fzp_bam_read_header(const char *path, char **text, size_t *text_len, int32_t *n_ref, uint8_t **ref_block, size_t *ref_block_len) {
    if (!path || !text || !text_len || !n_ref || !ref_block || !ref_block_len) { fzp_set_error("fzp_bam_read_header: bad arguments"); return FZP_EINVAL; }
    BgzfIn in;
    FZP_TRY(in.open(path));
    std::string t;
    std::vector<uint8_t> refs;
    FZP_TRY(in.header(t, *n_ref, refs));
    char *tp = (char *)malloc(t.size() + 1);
    if (!tp) return FZP_ENOMEM;
    memcpy(tp, t.data(), t.size());
    tp[t.size()] = 0;
    size_t rl = 0;
    int rc = give(refs, ref_block, &rl);
    if (rc) { free(tp); return rc; }
    *text = tp; *text_len = t.size(); *ref_block_len = rl;
    return FZP_OK;
}

extern "C" int fzp_bam_route(int32_t n_in, const char *const *in_paths, int64_t n_names, const int64_t *name_off, const char *names, const int32_t *name_dest,
                             int32_t n_dest, const char *const *dest_paths, const char *header_text, size_t header_len, int32_t n_ref, const uint8_t *ref_block,
                             size_t ref_block_len, int64_t *dest_records, int32_t *first_use) {
    if (n_in < 0 || (n_in && !in_paths) || n_names < 0 || (n_names && (!name_off || !names || !name_dest)) || n_dest < 0 || (n_dest && !dest_paths) ||
        (!header_text && header_len) || n_ref < 0 || (!ref_block && ref_block_len)) {
        fzp_set_error("fzp_bam_route: bad arguments");
        return FZP_EINVAL;
    }
    std::unordered_map<std::string, int32_t> where;
    where.reserve((size_t)n_names * 2);
    for (int64_t i = 0; i < n_names; i++) {
        if (name_dest[i] < 0 || name_dest[i] >= n_dest) { fzp_set_error("fzp_bam_route: name %lld has destination %d of %d", (long long)i, name_dest[i], n_dest); return FZP_EINVAL; }
        where.emplace(std::string(names + name_off[i], (size_t)(name_off[i + 1] - name_off[i])), name_dest[i]);
    }
    std::vector<BamOut> outs((size_t)n_dest);
    int32_t n_used = 0;
    for (int32_t d = 0; d < n_dest; d++) { outs[(size_t)d].path = dest_paths[d]; if (first_use) first_use[d] = -1; if (dest_records) dest_records[d] = 0; }
    std::string key;
    for (int32_t k = 0; k < n_in; k++) {
        BgzfIn in;
        FZP_TRY(in.open(in_paths[k]));
        std::string t;
        int32_t nr = 0;
        std::vector<uint8_t> refs;
        FZP_TRY(in.header(t, nr, refs));
        for (;;) {
            FZP_TRY(in.need(4));
            if (in.have() == 0) break;
            if (in.have() < 4) { fzp_set_error("%s: truncated BAM record", in.path.c_str()); return FZP_EINVAL; }
            const int32_t bs = BgzfIn::rd32(in.ptr());
            if (bs < 32) { fzp_set_error("%s: malformed BAM record", in.path.c_str()); return FZP_EINVAL; }
            FZP_TRY(in.need(4 + (size_t)bs));
            if (in.have() < 4 + (size_t)bs) { fzp_set_error("%s: truncated BAM record", in.path.c_str()); return FZP_EINVAL; }
            const uint8_t *r = in.ptr();
            const uint32_t l_name = r[4 + 8];
            if (l_name < 1 || 32 + (size_t)l_name > (size_t)bs) { fzp_set_error("%s: malformed BAM record", in.path.c_str()); return FZP_EINVAL; }
            key.assign((const char *)r + 4 + 32, l_name - 1);
            auto it = where.find(key);
            if (it != where.end()) {
                BamOut &o = outs[(size_t)it->second];
                if (o.n_rec == 0) {                                    // first record of this destination: header first, records from a block boundary
                    Bytes h;
                    h.raw("BAM\1", 4); h.u32((uint32_t)header_len); h.raw(header_text, header_len); h.u32((uint32_t)n_ref); h.raw(ref_block, ref_block_len);
                    FZP_TRY(o.z.write(h.v.data(), h.v.size()));
                    FZP_TRY(o.z.flush());
                    if (first_use) first_use[n_used] = it->second;
                    n_used++;
                }
                FZP_TRY(o.z.write(r, 4 + (size_t)bs));
                o.n_rec++;
                if (o.z.out.v.size() >= (1u << 20)) FZP_TRY(o.spill());
            }
            in.at += 4 + (size_t)bs;
        }
    }
    for (int32_t d = 0; d < n_dest; d++) {
        BamOut &o = outs[(size_t)d];
        if (!o.n_rec) continue;
        FZP_TRY(o.z.finish());
        FZP_TRY(o.spill());
        if (dest_records) dest_records[d] = o.n_rec;
    }
    return FZP_OK;
}
