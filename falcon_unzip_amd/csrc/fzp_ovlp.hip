// fzp_ovlp.hip -- overlap filter with phase (falcon_unzip/ovlp_filter_with_phase.py:49-354), SURVEY section 8f row n2.
//
// The reference streams `LA4Falcon -mo` text three times and keeps Python dicts / sets keyed by id strings.  Here:
//   host    the rid map (small): ids / contigs / blocks / phases interned (strings stay strings: they are only ever
//           compared for equality, plus one lexicographic rank for sort ties);
//   device  K_nl/K_ls the text goes to HBM once; newline counts per 16 B -> scan -> line starts
//           K_tok     one thread per line: str.split() state machine, ids -> map entries through a direct table
//                     (rid_to_phase.all keys are '%09d' decimals, phasing_readmap.py:47-51; when a map has any other
//                     key shape the lines are tokenised on the host instead), int() / float() of the numeric columns
//                     (leniently: a bad field only matters if the line passes the phase checks, as in the reference
//                     where such lines are never parsed; `float(l[3]) < 90` is decided exactly in integers for plain
//                     decimals of <= 15 digits, anything else is settled by the host's strtod)
//           K_pre     the four phase checks of every stage (:64-73)
//           K_heads   query groups = runs of equal q among the lines that passed, per file (:77, :215)
//           K_stage1  5'/3' counts per group -> ignore flags (:79-87, :96-119)
//           K_stage2  containment flags (:165-181)
//           K_rank    per group and read end: rank of every candidate under the reference's sort key
//                     (-inphase, -overlap_len, t_l - (t_e - t_s), then the token list) and the best-n cut (:221-235)
//           K_emit    selected lines in print order (slots from scans: no order depends on atomics)
//   host    groups whose candidates tie on the whole numeric key AND partner id (the same pair listed twice with equal
//           length and span) are re-ordered by the reference's last key, the comparison of the lines' token lists.
// HBM-bound byte / integer scans: the text once (1 B per byte), 42 B per line of columns, 8 B per selected line out; groups are small (tens to hundreds of rows), so
// the all-pairs ranking inside a group is cheap and needs no global sort.
#include <algorithm>
#include <thread>
#include <string_view>
#include <unordered_map>
#include "fzp_common.h"

namespace {
struct Tok { int64_t off; int32_t len; };

inline bool is_space(unsigned char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f' || c == '\n'; }

// str.split() of text[b, e): token offsets (absolute) into out, returns the count
int split_line(const char *text, int64_t b, int64_t e, Tok *out, int max) {
    int n = 0;
    int64_t s = b;
    while (s < e) {
        while (s < e && is_space((unsigned char)text[s])) s++;
        if (s >= e) break;
        const int64_t t0 = s;
        while (s < e && !is_space((unsigned char)text[s])) s++;
        if (n < max) { out[n].off = t0; out[n].len = (int32_t)(s - t0); }
        n++;
    }
    return n;
}

bool parse_int32(const char *p, int n, int64_t *out) {   // int(str), limited to 32 bits
    if (n <= 0 || n > 20) return false;
    int i = 0;
    bool neg = false;
    if (p[0] == '+' || p[0] == '-') { neg = p[0] == '-'; i = 1; }
    if (i >= n) return false;
    int64_t v = 0;
    for (; i < n; i++) {
        if (p[i] < '0' || p[i] > '9') return false;
        v = v * 10 + (p[i] - '0');
        if (v > (1ll << 40)) return false;
    }
    v = neg ? -v : v;
    if (v < -2147483647ll || v > 2147483647ll) return false;
    *out = v;
    return true;
}
bool parse_float(const char *p, int n, double *out) {    // float(str)
    if (n <= 0 || n > 62) return false;
    char buf[64];
    memcpy(buf, p, (size_t)n); buf[n] = 0;
    for (int k = 0; k < n; k++) if (buf[k] == 'x' || buf[k] == 'X' || buf[k] == 'p' || buf[k] == 'P') return false;
    char *endp;
    *out = strtod(buf, &endp);
    return !(*endp || endp == buf);
}
constexpr int MAXTOK = 64;
constexpr uint8_t F_PARSE_OK = 1, F_IDT_OK = 2, F_CONTAINS = 4, F_CONTAINED = 8;
}  // namespace

// The dumps on the host, as the formatter, the tie fix-up and the fallback tokeniser read them back: one "text" made of the caller's
// buffers laid end to end, each ending with '\n'.  The buffers are BORROWED (they stay valid until the set is freed: include/fzphase.h);
// only a dump that lacks its final newline is copied (with one).  Offsets are global; base(off) gives a pointer p such that p + o is the
// byte at global offset o for every o inside the segment that holds `off` (a line never spans segments).  materialise() makes the one
// contiguous copy the host tokeniser wants.
struct HostText {
    std::vector<const char *> seg;
    std::vector<int64_t> beg;               // [n_seg + 1] global offset of every segment, then the total
    std::vector<char *> owned;              // segments that had to be copied
    char *whole = nullptr;                  // contiguous copy (materialise)
    HostText() { beg.push_back(0); }
    ~HostText() { for (auto q : owned) free(q); free(whole); }
    bool add(const char *s, size_t k) {     // one dump
        if (k && s[k - 1] != '\n') {
            char *c = (char *)malloc(k + 1);
            if (!c) return false;
            memcpy(c, s, k); c[k] = '\n';
            owned.push_back(c);
            s = c; k++;
        }
        seg.push_back(s);
        beg.push_back(beg.back() + (int64_t)k);
        return true;
    }
    size_t size() const { return (size_t)beg.back(); }
    const char *base(int64_t off) const {
        if (whole) return whole;
        size_t k = (size_t)(std::upper_bound(beg.begin(), beg.end(), off) - beg.begin());
        k = k ? k - 1 : 0;
        if (k >= seg.size()) k = seg.size() - 1;
        return seg[k] - beg[k];
    }
    bool materialise() {
        if (whole) return true;
        whole = (char *)malloc(size() + 16);
        if (!whole) return false;
        for (size_t k = 0; k < seg.size(); k++) memcpy(whole + beg[k], seg[k], (size_t)(beg[k + 1] - beg[k]));
        return true;
    }
    const char *data() const { return whole; }    // after materialise()
};
// all segments into one device buffer (16-byte padded, the pad zeroed)
static int upload_text(const HostText &t, DevBuf<uint8_t> &d_text, hipStream_t st) {
    const int64_t n16 = ((int64_t)t.size() + 15) / 16;
    if (n16 <= 0) return FZP_OK;
    FZP_TRY(d_text.alloc((size_t)n16 * 16));
    FZP_HIP(hipMemsetAsync(d_text.p + (n16 - 1) * 16, 0, 16, st));
    for (size_t k = 0; k < t.seg.size(); k++)
        if (t.beg[k + 1] > t.beg[k]) FZP_HIP(hipMemcpyAsync(d_text.p + t.beg[k], t.seg[k], (size_t)(t.beg[k + 1] - t.beg[k]), hipMemcpyHostToDevice, st));
    return FZP_OK;
}

struct fzp_ovlset {
    int device = 0;
    HostText text;                          // all dumps, each ending with '\n' (borrowed from the caller)
    std::vector<int64_t> file_end;          // end offset of every dump inside `text`
    std::vector<int64_t> line_off;          // [n_lines + 1]
    std::string map_text;
    std::vector<Tok> key, ctg, blk, ph;     // per distinct map key (first-appearance order), offsets into map_text
    std::vector<int32_t> ctg_code, blk_code, ph_code, lex_rank;
    std::unordered_map<std::string_view, int32_t> ids;   // key string -> index (views into map_text)
    int64_t n_lines = 0, n_rows = 0;        // rows = lines whose q_id and t_id are both map keys
    bool tokenised_on_device = false;
    // per line, resident in HBM: q / t = map entry or -1
    DevBuf<int32_t> d_q, d_t, d_file, d_ovl, d_q_s, d_q_e, d_q_l, d_t_s, d_t_e, d_t_l;
    DevBuf<uint8_t> d_flags;
    DevBuf<int32_t> d_ctg, d_blk, d_ph, d_lex;
};

extern "C" void fzp_ovlset_free(fzp_ovlset *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    delete s;
}
extern "C" int64_t fzp_ovl_n_lines(const fzp_ovlset *s) { return s ? s->n_lines : 0; }
extern "C" int64_t fzp_ovl_n_rows(const fzp_ovlset *s) { return s ? s->n_rows : 0; }
extern "C" int fzp_ovl_id_name(const fzp_ovlset *s, int32_t id, const char **name, int32_t *len) {
    if (!s || id < 0 || (size_t)id >= s->key.size() || !name || !len) { fzp_set_error("fzp_ovl_id_name: bad arguments"); return FZP_EINVAL; }
    *name = s->map_text.data() + s->key[(size_t)id].off;
    *len = s->key[(size_t)id].len;
    return FZP_OK;
}

namespace {
// one line on the host: what K_tok computes on the device (same rules), for the fallback tokeniser, the tie fix-up and
// the formatter
struct HostLine { int nt; int32_t q, t; int64_t v[7]; uint8_t flags; };
bool host_line(const fzp_ovlset *s, int64_t line, HostLine *o, Tok *toks /* [MAXTOK] */) {
    const char *tx = s->text.base(s->line_off[(size_t)line]);
    const int nt = split_line(tx, s->line_off[(size_t)line], s->line_off[(size_t)line + 1], toks, MAXTOK);
    o->nt = nt; o->q = o->t = -1; o->flags = 0;
    for (int c = 0; c < 7; c++) o->v[c] = 0;
    if (nt < 2) return false;
    auto iq = s->ids.find(std::string_view(tx + toks[0].off, (size_t)toks[0].len));
    auto it = s->ids.find(std::string_view(tx + toks[1].off, (size_t)toks[1].len));
    o->q = iq == s->ids.end() ? -1 : iq->second;
    o->t = it == s->ids.end() ? -1 : it->second;
    static const int col[7] = {2, 5, 6, 7, 9, 10, 11};
    double idt = 0;
    bool ok = nt >= 12 && nt <= MAXTOK;
    for (int c = 0; ok && c < 7; c++) ok = parse_int32(tx + toks[col[c]].off, toks[col[c]].len, &o->v[c]);
    if (ok) ok = parse_float(tx + toks[3].off, toks[3].len, &idt);
    if (ok) {
        o->flags |= F_PARSE_OK;
        if (!(idt < 90)) o->flags |= F_IDT_OK;
        const std::string_view tag(tx + toks[nt - 1].off, (size_t)toks[nt - 1].len);
        if (tag == "contains") o->flags |= F_CONTAINS;
        if (tag == "contained") o->flags |= F_CONTAINED;
    }
    return true;
}
// '%09d'-shaped key: decimal digits, at least 9 of them, no leading zero beyond the padding
bool canonical_id(const char *p, int n, int64_t *val) {
    if (n < 9 || n > 10) return false;
    if (n > 9 && p[0] == '0') return false;
    int64_t v = 0;
    for (int i = 0; i < n; i++) { if (p[i] < '0' || p[i] > '9') return false; v = v * 10 + (p[i] - '0'); }
    if (v > 2147483646ll) return false;
    *val = v;
    return true;
}
int ovl_tokenise_device(fzp_ctx *ctx, fzp_ovlset *s, int64_t max_id, const std::vector<int32_t> &arid_of, DevBuf<uint8_t> &text);
int ovl_tokenise_host(fzp_ctx *ctx, fzp_ovlset *s);
}  // namespace

extern "C" int fzp_ovl_parse(fzp_ctx *ctx, int32_t n_files, const char *const *texts, const size_t *lens, const char *rid_map, size_t map_len, fzp_ovlset **out) {
    if (!ctx || !out || n_files < 0 || (n_files && (!texts || !lens)) || (!rid_map && map_len)) { fzp_set_error("fzp_ovl_parse: bad arguments"); return FZP_EINVAL; }
    *out = nullptr;
    FZP_TRY(fzp_bind(ctx));
    fzp_ovlset *s = new fzp_ovlset();
    s->device = ctx->device;
    // ---- the dumps: the caller's buffers, end to end, every dump ending with '\n'
    for (int k = 0; k < n_files; k++) {
        if (!s->text.add(texts[k], lens[k])) { delete s; fzp_set_error("fzp_ovl_parse: host allocation failed"); return FZP_ENOMEM; }
        s->file_end.push_back((int64_t)s->text.size());
    }
    if (s->text.size() >= (1ull << 32)) { fzp_set_error("fzp_ovl_parse: %zu bytes of text (limit 4 GiB per call; split the fofn)", s->text.size()); delete s; return FZP_EINVAL; }
    // the text starts its way to HBM now; the map is read while the DMA runs
    hipStream_t st = ctx->stream;
    DevBuf<uint8_t> d_text;
    { const int rc0 = upload_text(s->text, d_text, st); if (rc0) { delete s; return rc0; } }
    // ---- rid_to_phase.all (:306-309)
    s->map_text.assign(rid_map ? rid_map : "", map_len);
    std::unordered_map<std::string_view, int32_t> strs;
    auto intern = [&](Tok t) {
        std::string_view v(s->map_text.data() + t.off, (size_t)t.len);
        auto it = strs.find(v);
        if (it != strs.end()) return it->second;
        const int32_t c = (int32_t)strs.size();
        strs.emplace(v, c);
        return c;
    };
    {
        const char *mt = s->map_text.data();
        const int64_t n = (int64_t)s->map_text.size();
        int64_t b = 0, row = 0;
        while (b < n) {
            const void *nlp = memchr(mt + b, '\n', (size_t)(n - b));
            const int64_t e = nlp ? (const char *)nlp - mt : n;
            Tok t[5];
            const int nt = split_line(mt, b, e, t, 5);
            if (nt < 4) { fzp_set_error("rid_phase_map row %lld has %d fields (IndexError at ovlp_filter_with_phase.py:309)", (long long)row, nt); delete s; return FZP_EINVAL; }
            std::string_view k(mt + t[0].off, (size_t)t[0].len);
            auto it = s->ids.find(k);
            size_t at;
            if (it == s->ids.end()) {
                at = s->key.size();
                s->ids.emplace(k, (int32_t)at);
                s->key.push_back(t[0]); s->ctg.push_back(t[1]); s->blk.push_back(t[2]); s->ph.push_back(t[3]);
            } else {
                at = (size_t)it->second;
                s->ctg[at] = t[1]; s->blk[at] = t[2]; s->ph[at] = t[3];
            }
            b = nlp ? e + 1 : n;
            row++;
        }
    }
    const size_t na = s->key.size();
    s->ctg_code.resize(na); s->blk_code.resize(na); s->ph_code.resize(na); s->lex_rank.resize(na);
    for (size_t i = 0; i < na; i++) { s->ctg_code[i] = intern(s->ctg[i]); s->blk_code[i] = intern(s->blk[i]); s->ph_code[i] = intern(s->ph[i]); }
    bool canonical = true;
    int64_t max_id = -1;
    {
        std::vector<int32_t> ord(na);
        for (size_t i = 0; i < na; i++) ord[i] = (int32_t)i;
        const char *mt = s->map_text.data();
        std::sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) {
            return std::string_view(mt + s->key[(size_t)a].off, (size_t)s->key[(size_t)a].len) < std::string_view(mt + s->key[(size_t)b].off, (size_t)s->key[(size_t)b].len);
        });
        for (size_t r = 0; r < na; r++) s->lex_rank[(size_t)ord[r]] = (int32_t)r;
        for (size_t i = 0; i < na && canonical; i++) {
            int64_t v;
            canonical = canonical_id(mt + s->key[i].off, s->key[i].len, &v);
            if (canonical) max_id = std::max(max_id, v);
        }
    }
    int rc = FZP_OK;
    if (!(rc = s->d_ctg.upload(s->ctg_code.data(), na, st)) && !(rc = s->d_blk.upload(s->blk_code.data(), na, st)) && !(rc = s->d_ph.upload(s->ph_code.data(), na, st)))
        rc = s->d_lex.upload(s->lex_rank.data(), na, st);
    if (!rc) {
        const bool force_host = getenv("FZP_OVL_HOST_TOKENISER") != nullptr;
        if (canonical && max_id <= (1 << 28) && !force_host) {      // direct table <= 1 GiB
            std::vector<int32_t> arid_of((size_t)max_id + 2, -1);
            const char *mt = s->map_text.data();
            for (size_t i = 0; i < na; i++) { int64_t v = 0; canonical_id(mt + s->key[i].off, s->key[i].len, &v); arid_of[(size_t)v] = (int32_t)i; }
            rc = ovl_tokenise_device(ctx, s, max_id, arid_of, d_text);
        } else {
            rc = ovl_tokenise_host(ctx, s);
        }
    }
    if (rc) { delete s; return rc; }
    *out = s;
    return FZP_OK;
}

// ================================================================================ device
namespace {
struct OvlView {
    const int32_t *q, *t, *file, *ovl, *q_s, *q_e, *q_l, *t_s, *t_e, *t_l;
    const uint8_t *flags;
    const int32_t *ctg, *blk, *ph, *lex;     // per id
    int64_t n;
};

constexpr uint8_t F_NEED_HOST = 16;   // idt is not a plain short decimal: the host's strtod decides

// K_nl: newlines per 16 bytes;  K_ls: line k+1 starts after the k-th newline
__global__ void __launch_bounds__(256) k_ovl_nl(const uint4 *__restrict__ text16, int64_t n16, uint32_t *__restrict__ cnt) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n16) return;
    const uint4 w = text16[j];
    const uint32_t x[4] = {w.x, w.y, w.z, w.w};
    uint32_t c = 0;
#pragma unroll
    for (int z = 0; z < 4; z++)
#pragma unroll
        for (int y = 0; y < 4; y++) c += ((x[z] >> (8 * y)) & 0xffu) == (uint32_t)'\n';
    cnt[j] = c;
}
__global__ void __launch_bounds__(256) k_ovl_ls(const uint4 *__restrict__ text16, int64_t n16, const uint32_t *__restrict__ pos, int64_t *__restrict__ ls) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n16) return;
    const uint4 w = text16[j];
    const uint32_t x[4] = {w.x, w.y, w.z, w.w};
    int64_t k = pos[j];
#pragma unroll
    for (int z = 0; z < 4; z++)
#pragma unroll
        for (int y = 0; y < 4; y++)
            if (((x[z] >> (8 * y)) & 0xffu) == (uint32_t)'\n') ls[++k] = j * 16 + z * 4 + y + 1;
}

struct OvlCols { int32_t *q, *t, *file, *ovl, *q_s, *q_e, *q_l, *t_s, *t_e, *t_l; uint8_t *flags; };

// K_tok: one thread per line -- str.split(), the two ids, int() of columns 2,5,6,7,9,10,11, float() of column 3, the tag
__global__ void __launch_bounds__(256) k_ovl_tok(const uint8_t *__restrict__ text, const int64_t *__restrict__ ls, int64_t n_lines, const int64_t *__restrict__ file_end,
                                                 int n_files, const int32_t *__restrict__ arid_of, int64_t max_id, OvlCols o, int32_t *__restrict__ err,
                                                 uint32_t *__restrict__ counters /* [0] rows, [1] lines for the host */) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool is_row = false, need_host = false;
    if (i < n_lines) {
        const int64_t b = ls[i], e = ls[i + 1];
        int tk = -1;                       // index of the current token
        bool in_tok = false;
        int64_t ids[2] = {-1, -1};
        int64_t v[7] = {0, 0, 0, 0, 0, 0, 0};
        uint32_t ok_mask = 0;              // bit c: integer column c parsed
        // current token
        int64_t acc = 0;
        int nd = 0, len = 0, fd = 0, ns = 0;
        bool neg = false, good = true, dot = false, plain = true;
        uint8_t first = 0;
        int64_t last_start = b;
        int last_len = 0;
        int f_state = 0;                   // 0 not seen, 1 decided by the device, 2 host, 3 invalid
        bool idt_lt90 = false;
        auto finish = [&]() {
            if (tk == 0 || tk == 1) {
                const bool shape = good && len >= 9 && len <= 10 && (len == 9 || first != (uint8_t)'0');
                if (arid_of) ids[tk] = (shape && acc <= max_id) ? (int64_t)arid_of[acc] : -1;
                else ids[tk] = (shape && acc <= 0x7ffffffe) ? acc : -1;           // raw mode: the id's value (-1: not '%09d'-shaped)
            } else if (tk == 3) {
                if (!plain || ns > 15 || fd > 15) f_state = 2;
                else if (nd == 0) f_state = 3;                       // no digit at all: float() raises
                else {
                    f_state = 1;
                    int64_t lim = 90;
                    for (int z = 0; z < fd; z++) lim *= 10;
                    idt_lt90 = neg ? true : acc < lim;
                }
            } else {
                int c = tk == 2 ? 0 : (tk >= 5 && tk <= 7) ? tk - 4 : (tk >= 9 && tk <= 11) ? tk - 5 : -1;
                if (c >= 0) {
                    const int64_t val = neg ? -acc : acc;
                    if (good && nd > 0 && nd <= 10 && val >= -2147483647ll && val <= 2147483647ll) { v[c] = val; ok_mask |= 1u << c; }
                }
            }
            last_len = len;
        };
        for (int64_t p = b; p < e; p++) {
            const uint8_t c = text[p];
            const bool sp = c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f' || c == '\n';
            if (sp) {
                if (in_tok) { finish(); in_tok = false; }
                continue;
            }
            if (!in_tok) {
                in_tok = true; tk++;
                acc = 0; nd = 0; len = 0; fd = 0; ns = 0; neg = false; good = true; dot = false; plain = true; first = c;
                last_start = p;
            }
            const bool dg = c >= (uint8_t)'0' && c <= (uint8_t)'9';
            if (tk == 3) {
                if (dg) {
                    if (ns > 0 || c != (uint8_t)'0') ns++;
                    if (ns <= 17) acc = acc * 10 + (c - (uint8_t)'0');
                    nd++;
                    if (dot) fd++;
                } else if (c == (uint8_t)'.' && !dot) dot = true;
                else if ((c == (uint8_t)'+' || c == (uint8_t)'-') && len == 0) neg = c == (uint8_t)'-';
                else plain = false;
            } else {
                if (dg) { if (nd < 12) acc = acc * 10 + (c - (uint8_t)'0'); nd++; }
                else if ((c == (uint8_t)'+' || c == (uint8_t)'-') && len == 0 && tk >= 2) neg = c == (uint8_t)'-';
                else good = false;
                if (tk < 2 && nd > 10) good = false;
            }
            len++;
        }
        if (in_tok) finish();
        const int nt = tk + 1;
        if (nt < 2) atomicMin(err, (int32_t)min(i, (int64_t)0x7ffffffe));
        uint8_t fl = 0;
        if (nt >= 12 && nt <= MAXTOK && ok_mask == 0x7fu && (f_state == 1 || f_state == 2)) {
            fl |= F_PARSE_OK;
            if (f_state == 2) { fl |= F_NEED_HOST; need_host = true; }
            else if (!idt_lt90) fl |= F_IDT_OK;
            // tag = the last token
            const char *t1 = "contains", *t2 = "contained";
            bool m1 = last_len == 8, m2 = last_len == 9;
            for (int z = 0; z < 9 && z < last_len; z++) {
                const uint8_t c = text[last_start + z];
                if (z < 8 && c != (uint8_t)t1[z]) m1 = false;
                if (c != (uint8_t)t2[z]) m2 = false;
            }
            if (m1) fl |= F_CONTAINS;
            if (m2) fl |= F_CONTAINED;
        }
        int lo = 0, hi = n_files - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (file_end[mid] > b) hi = mid; else lo = mid + 1; }
        o.q[i] = (int32_t)ids[0]; o.t[i] = (int32_t)ids[1]; o.file[i] = lo;
        o.ovl[i] = (int32_t)-v[0];
        o.q_s[i] = (int32_t)v[1]; o.q_e[i] = (int32_t)v[2]; o.q_l[i] = (int32_t)v[3];
        o.t_s[i] = (int32_t)v[4]; o.t_e[i] = (int32_t)v[5]; o.t_l[i] = (int32_t)v[6];
        o.flags[i] = fl;
        is_row = ids[0] >= 0 && ids[1] >= 0;
        need_host = need_host && is_row;
    }
    const uint64_t br = __ballot(is_row), bh = __ballot(need_host);
    if (lane_id() == 0) {
        if (br) atomicAdd(&counters[0], (uint32_t)__popcll(br));
        if (bh) atomicAdd(&counters[1], (uint32_t)__popcll(bh));
    }
}

int alloc_cols(fzp_ovlset *s, size_t n) {
    FZP_TRY(s->d_q.alloc(n)); FZP_TRY(s->d_t.alloc(n)); FZP_TRY(s->d_file.alloc(n)); FZP_TRY(s->d_ovl.alloc(n));
    FZP_TRY(s->d_q_s.alloc(n)); FZP_TRY(s->d_q_e.alloc(n)); FZP_TRY(s->d_q_l.alloc(n));
    FZP_TRY(s->d_t_s.alloc(n)); FZP_TRY(s->d_t_e.alloc(n)); FZP_TRY(s->d_t_l.alloc(n)); FZP_TRY(s->d_flags.alloc(n));
    return FZP_OK;
}
inline unsigned blocks_for(int64_t n, int per) { return (unsigned)std::max<int64_t>(1, (n + per - 1) / per); }

int ovl_tokenise_device(fzp_ctx *ctx, fzp_ovlset *s, int64_t max_id, const std::vector<int32_t> &arid_of, DevBuf<uint8_t> &text) {
    hipStream_t st = ctx->stream;
    const int64_t nb = (int64_t)s->text.size();
    s->tokenised_on_device = true;
    if (nb == 0) { s->n_lines = 0; s->line_off.assign(1, 0); return FZP_OK; }
    const int64_t n16 = (nb + 15) / 16;
    DevBuf<uint32_t> cnt, pos, counters;
    DevBuf<uint64_t> totals;
    DevBuf<int64_t> ls, fend;
    DevBuf<int32_t> d_arid, err;
    FZP_TRY(cnt.alloc((size_t)n16)); FZP_TRY(pos.alloc((size_t)n16)); FZP_TRY(totals.alloc(2)); FZP_TRY(counters.alloc(2)); FZP_TRY(err.alloc(1));
    { ProfScope ps(ctx, "ovl_lines"); hipLaunchKernelGGL(k_ovl_nl, dim3(blocks_for(n16, 256)), dim3(256), 0, st, (const uint4 *)text.p, n16, cnt.p); }
    FZP_TRY(fzp_exclusive_scan_u32(ctx, cnt.p, pos.p, (size_t)n16, totals.p));
    uint64_t tot = 0;
    FZP_HIP(hipMemcpyAsync(&tot, totals.p, 8, hipMemcpyDeviceToHost, st));
    FZP_HIP(hipStreamSynchronize(st));
    const int64_t nl = (int64_t)tot;
    if (nl >= (1ll << 31) - 1) { fzp_set_error("fzp_ovl_parse: %lld lines (limit 2^31 per call)", (long long)nl); return FZP_EINVAL; }
    s->n_lines = nl;
    FZP_TRY(ls.alloc((size_t)nl + 1));
    FZP_HIP(hipMemsetAsync(ls.p, 0, 8, st));
    { ProfScope ps(ctx, "ovl_lines"); hipLaunchKernelGGL(k_ovl_ls, dim3(blocks_for(n16, 256)), dim3(256), 0, st, (const uint4 *)text.p, n16, pos.p, ls.p); }
    FZP_TRY(alloc_cols(s, (size_t)nl));
    FZP_TRY(d_arid.upload(arid_of.data(), arid_of.size(), st));
    FZP_TRY(fend.upload(s->file_end.data(), s->file_end.size(), st));
    const int32_t no_err = 0x7fffffff;
    FZP_HIP(hipMemcpyAsync(err.p, &no_err, 4, hipMemcpyHostToDevice, st));
    FZP_HIP(hipMemsetAsync(counters.p, 0, 8, st));
    OvlCols o = {s->d_q.p, s->d_t.p, s->d_file.p, s->d_ovl.p, s->d_q_s.p, s->d_q_e.p, s->d_q_l.p, s->d_t_s.p, s->d_t_e.p, s->d_t_l.p, s->d_flags.p};
    { ProfScope ps(ctx, "ovl_tokenise"); hipLaunchKernelGGL(k_ovl_tok, dim3(blocks_for(nl, 256)), dim3(256), 0, st, text.p, ls.p, nl, fend.p, (int)s->file_end.size(), arid_of.empty() ? (const int32_t *)nullptr : d_arid.p, max_id, o, err.p, counters.p); }
    s->line_off.resize((size_t)nl + 1);
    uint32_t h_cnt[2] = {0, 0};
    int32_t h_err = 0;
    FZP_TRY(ls.download(s->line_off.data(), (size_t)nl + 1, st));
    FZP_HIP(hipMemcpyAsync(h_cnt, counters.p, 8, hipMemcpyDeviceToHost, st));
    FZP_HIP(hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, st));
    FZP_HIP(hipStreamSynchronize(st));
    FZP_HIP(hipGetLastError());
    if (h_err != no_err) { fzp_set_error("overlap line %d has fewer than 2 tokens (ValueError at ovlp_filter_with_phase.py:62)", h_err); return FZP_EINVAL; }
    s->n_rows = h_cnt[0];
    if (h_cnt[1]) {   // idt columns the device left to strtod (exponents, nan/inf, > 15 digits)
        std::vector<uint8_t> fl((size_t)nl);
        FZP_TRY(s->d_flags.download(fl.data(), (size_t)nl, st));
        FZP_HIP(hipStreamSynchronize(st));
        Tok toks[MAXTOK];
        for (int64_t i = 0; i < nl; i++)
            if (fl[(size_t)i] & F_NEED_HOST) { HostLine hl; host_line(s, i, &hl, toks); fl[(size_t)i] = hl.flags; }
        FZP_TRY(s->d_flags.upload(fl.data(), (size_t)nl, st));
        FZP_HIP(hipStreamSynchronize(st));
    }
    return FZP_OK;
}

int ovl_tokenise_host(fzp_ctx *ctx, fzp_ovlset *s) {
    hipStream_t st = ctx->stream;
    if (!s->text.materialise()) { fzp_set_error("fzp_ovl_parse: host allocation failed"); return FZP_ENOMEM; }
    const char *tx = s->text.data();
    const int64_t n = (int64_t)s->text.size();
    s->line_off.assign(1, 0);
    for (int64_t b = 0; b < n;) {
        const void *nlp = memchr(tx + b, '\n', (size_t)(n - b));
        b = nlp ? ((const char *)nlp - tx) + 1 : n;
        s->line_off.push_back(b);
    }
    const int64_t nl = (int64_t)s->line_off.size() - 1;
    if (nl >= (1ll << 31) - 1) { fzp_set_error("fzp_ovl_parse: %lld lines (limit 2^31 per call)", (long long)nl); return FZP_EINVAL; }
    s->n_lines = nl;
    std::vector<int32_t> q((size_t)nl), t((size_t)nl), file((size_t)nl), col[7];
    for (auto &c : col) c.resize((size_t)nl);
    std::vector<uint8_t> fl((size_t)nl);
    Tok toks[MAXTOK];
    int f = 0;
    for (int64_t i = 0; i < nl; i++) {
        HostLine hl;
        if (!host_line(s, i, &hl, toks)) { fzp_set_error("overlap line %lld has fewer than 2 tokens (ValueError at ovlp_filter_with_phase.py:62)", (long long)i); return FZP_EINVAL; }
        while (f + 1 < (int)s->file_end.size() && s->line_off[(size_t)i] >= s->file_end[(size_t)f]) f++;
        q[(size_t)i] = hl.q; t[(size_t)i] = hl.t; file[(size_t)i] = f; fl[(size_t)i] = hl.flags;
        col[0][(size_t)i] = (int32_t)-hl.v[0];
        for (int c = 1; c < 7; c++) col[c][(size_t)i] = (int32_t)hl.v[c];
        s->n_rows += hl.q >= 0 && hl.t >= 0;
    }
    if (nl) {
        FZP_TRY(s->d_q.upload(q.data(), (size_t)nl, st)); FZP_TRY(s->d_t.upload(t.data(), (size_t)nl, st)); FZP_TRY(s->d_file.upload(file.data(), (size_t)nl, st));
        FZP_TRY(s->d_ovl.upload(col[0].data(), (size_t)nl, st)); FZP_TRY(s->d_q_s.upload(col[1].data(), (size_t)nl, st)); FZP_TRY(s->d_q_e.upload(col[2].data(), (size_t)nl, st));
        FZP_TRY(s->d_q_l.upload(col[3].data(), (size_t)nl, st)); FZP_TRY(s->d_t_s.upload(col[4].data(), (size_t)nl, st)); FZP_TRY(s->d_t_e.upload(col[5].data(), (size_t)nl, st));
        FZP_TRY(s->d_t_l.upload(col[6].data(), (size_t)nl, st)); FZP_TRY(s->d_flags.upload(fl.data(), (size_t)nl, st));
        FZP_HIP(hipStreamSynchronize(st));
    }
    return FZP_OK;
}

// K_pre: the four checks every stage starts with (:64-73)
__global__ void __launch_bounds__(256) k_ovl_pre(OvlView v, uint32_t *__restrict__ pre, int32_t *__restrict__ err) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= v.n) return;
    const int32_t q = v.q[i], t = v.t[i];
    const bool keep = q >= 0 && t >= 0 && v.ctg[q] == v.ctg[t] && !(v.blk[q] == v.blk[t] && v.ph[q] != v.ph[t]);
    pre[i] = keep ? 1u : 0u;
    if (keep && !(v.flags[i] & F_PARSE_OK)) atomicMin(err, (int32_t)min(i, (int64_t)0x7ffffffe));
}
__global__ void __launch_bounds__(256) k_ovl_compact(int64_t n, const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pos, int32_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && flag[i]) out[pos[i]] = (int32_t)i;
}
// K_heads: a group starts where q (or the file) changes among the kept rows (:77, :212-216)
__global__ void __launch_bounds__(256) k_ovl_heads(OvlView v, int64_t np, const int32_t *__restrict__ P, uint32_t *__restrict__ head) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= np) return;
    bool h = k == 0;
    if (!h) { const int32_t a = P[k], b = P[k - 1]; h = v.file[a] != v.file[b] || v.q[a] != v.q[b]; }
    head[k] = h ? 1u : 0u;
}
__device__ __forceinline__ bool pass2(const OvlView &v, int32_t i, int32_t min_len) {   // :98-102
    return (v.flags[i] & F_IDT_OK) && v.q_l[i] >= min_len && v.t_l[i] >= min_len;
}
// K_stage1: one wave per group (:79-87 with the counts of :108-119)
__global__ void __launch_bounds__(256) k_ovl_stage1(OvlView v, int64_t ng, const int32_t *__restrict__ gstart, const int32_t *__restrict__ P, int64_t max_diff,
                                                    int64_t max_cov, int64_t min_cov, int32_t min_len, uint8_t *__restrict__ ignore) {
    const int lane = lane_id();
    for (int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); g < ng; g += (int64_t)gridDim.x * 4) {
        int32_t left = 0, right = 0;
        for (int32_t k = gstart[g] + lane; k < gstart[g + 1]; k += 64) {
            const int32_t i = P[k];
            if (!pass2(v, i, min_len)) continue;
            left += v.q_s[i] == 0;
            right += v.q_e[i] == v.q_l[i];
        }
        left = wave_sum_i32_dpp(left);
        right = wave_sum_i32_dpp(right);
        const int64_t d = left > right ? left - right : right - left;
        const bool ig = d > max_diff || left > max_cov || right > max_cov || left < min_cov || right < min_cov;
        if (ig && lane == 0) ignore[v.q[P[gstart[g]]]] = 1;
    }
}
// K_stage2 (:165-181)
__global__ void __launch_bounds__(256) k_ovl_stage2(OvlView v, int64_t np, const int32_t *__restrict__ P, int32_t min_len, const uint8_t *__restrict__ ignore,
                                                    uint8_t *__restrict__ contained) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= np) return;
    const int32_t i = P[k];
    if (!pass2(v, i, min_len)) return;
    const int32_t q = v.q[i], t = v.t[i];
    if (ignore[q] || ignore[t]) return;
    if (v.flags[i] & F_CONTAINED) contained[q] = 1;
    if (v.flags[i] & F_CONTAINS) contained[t] = 1;
}
// candidate of stage 3 (:239-260): -1 none, 0 = 5' list, 1 = 3' list
__device__ __forceinline__ int cand_end(const OvlView &v, int32_t i, int32_t min_len, const uint8_t *ignore, const uint8_t *contained) {
    const int32_t q = v.q[i], t = v.t[i];
    if (contained[q] || contained[t] || ignore[q] || ignore[t]) return -1;
    if (!pass2(v, i, min_len)) return -1;
    if (v.q_s[i] == 0) return 0;
    if (v.q_e[i] == v.q_l[i]) return 1;
    return -1;
}
struct CandKey { int32_t ninph, negovl, m_range, trank; };
__device__ __forceinline__ CandKey cand_key(const OvlView &v, int32_t i) {
    const int32_t q = v.q[i], t = v.t[i];
    CandKey k;
    k.ninph = (v.ctg[q] == v.ctg[t] && v.blk[q] == v.blk[t] && v.ph[q] == v.ph[t]) ? 0 : 1;   // -inphase, shifted by one
    k.negovl = -v.ovl[i];
    k.m_range = v.t_l[i] - (v.t_e[i] - v.t_s[i]);
    k.trank = v.lex[t];
    return k;
}
// K_rank: one wave per group.  rank[k] = position of row k in its sorted list, or -1 if it is not printed;
// cnt[2g + end] = rows printed of that list; tie[2g + end] = two candidates agree on the whole key and the partner.
__global__ void __launch_bounds__(256) k_ovl_rank(OvlView v, int64_t ng, const int32_t *__restrict__ gstart, const int32_t *__restrict__ P, int32_t min_len, int64_t bestn,
                                                  const uint8_t *__restrict__ ignore, const uint8_t *__restrict__ contained, int32_t *__restrict__ rank,
                                                  uint32_t *__restrict__ cnt, uint8_t *__restrict__ tie) {
    const int lane = lane_id();
    for (int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); g < ng; g += (int64_t)gridDim.x * 4) {
        const int32_t gs = gstart[g], ge = gstart[g + 1];
        int32_t n_end[2] = {0, 0}, cut[2] = {0x7fffffff, 0x7fffffff};
        bool any_tie[2] = {false, false};
        for (int32_t k0 = gs; k0 < ge; k0 += 64) {
            const int32_t k = k0 + lane;
            int e = -1;
            CandKey a = {0, 0, 0, 0};
            if (k < ge) { e = cand_end(v, P[k], min_len, ignore, contained); if (e >= 0) a = cand_key(v, P[k]); }
            int32_t r = 0;
            bool tied = false;
            for (int32_t kb = gs; kb < ge; kb++) {            // wave-uniform walk over the group
                const int32_t ib = P[kb];
                const int eb = cand_end(v, ib, min_len, ignore, contained);
                if (eb < 0) continue;
                const CandKey b = cand_key(v, ib);
                if (eb != e || kb == k) continue;
                const bool lt = b.ninph != a.ninph ? b.ninph < a.ninph
                              : b.negovl != a.negovl ? b.negovl < a.negovl
                              : b.m_range != a.m_range ? b.m_range < a.m_range
                              : b.trank != a.trank ? b.trank < a.trank : kb < k;
                r += lt ? 1 : 0;
                tied |= b.ninph == a.ninph && b.negovl == a.negovl && b.m_range == a.m_range && b.trank == a.trank;
            }
            if (k < ge) rank[k] = e >= 0 ? r : -1;
#pragma unroll
            for (int z = 0; z < 2; z++) {
                n_end[z] += __popcll(__ballot(e == z));
                any_tie[z] |= __any(e == z && tied);
                const int32_t c = (e == z && r >= bestn && a.m_range > 1000) ? r : 0x7fffffff;   // the first such row is the last one printed (:232-233)
                cut[z] = min(cut[z], wave_min_i32(c));
            }
        }
        int32_t kept[2];
#pragma unroll
        for (int z = 0; z < 2; z++) kept[z] = cut[z] == 0x7fffffff ? n_end[z] : min(n_end[z], cut[z] + 1);
        for (int32_t k = gs + lane; k < ge; k += 64) {
            const int32_t r = rank[k];
            if (r < 0) continue;
            const int e = cand_end(v, P[k], min_len, ignore, contained);
            if (r >= kept[e]) rank[k] = -1;
        }
        if (lane == 0) {
            cnt[2 * g] = (uint32_t)kept[0]; cnt[2 * g + 1] = (uint32_t)kept[1];
            tie[2 * g] = any_tie[0]; tie[2 * g + 1] = any_tie[1];
        }
    }
}
// K_emit: print order = groups in input order, 5' list then 3' list, each by rank
__global__ void __launch_bounds__(256) k_ovl_emit(OvlView v, int64_t np, const int32_t *__restrict__ P, const uint32_t *__restrict__ gid, int32_t min_len,
                                                  const uint8_t *__restrict__ ignore, const uint8_t *__restrict__ contained, const int32_t *__restrict__ rank,
                                                  const uint32_t *__restrict__ off, int32_t *__restrict__ out) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= np) return;
    const int32_t r = rank[k];
    if (r < 0) return;
    const int e = cand_end(v, P[k], min_len, ignore, contained);
    out[off[2 * (int64_t)gid[k] + e] + r] = P[k];
}
__global__ void __launch_bounds__(256) k_ovl_gid(int64_t np, const uint32_t *__restrict__ head, const uint32_t *__restrict__ hscan, uint32_t *__restrict__ gid) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < np) gid[k] = hscan[k] + head[k] - 1u;     // inclusive scan - 1
}
}  // namespace

extern "C" int fzp_ovl_filter(fzp_ctx *ctx, const fzp_ovlset *s, const fzp_ovlp_params *pr, int64_t **rows_out, int64_t *n_rows_out, int32_t **ignore_out,
                              int64_t *n_ignore, int32_t **contained_out, int64_t *n_contained) {
    if (!ctx || !s || !pr || !rows_out || !n_rows_out) { fzp_set_error("fzp_ovl_filter: bad arguments"); return FZP_EINVAL; }
    *rows_out = nullptr; *n_rows_out = 0;
    if (ignore_out) { *ignore_out = nullptr; if (n_ignore) *n_ignore = 0; }
    if (contained_out) { *contained_out = nullptr; if (n_contained) *n_contained = 0; }
    FZP_TRY(fzp_bind(ctx));
    hipStream_t st = ctx->stream;
    const int64_t n = s->n_lines;
    const size_t na = s->key.size();
    const int32_t min_len = (int32_t)std::max<int64_t>(-2147483647ll, std::min<int64_t>(pr->min_len, 2147483647ll));
    std::vector<uint8_t> h_ignore(na + 1, 0), h_contained(na + 1, 0);
    std::vector<int32_t> h_out;
    std::vector<int32_t> h_P, h_gstart;
    std::vector<uint32_t> h_off, h_cnt;
    std::vector<uint8_t> h_tie;
    int64_t np = 0, ng = 0;
    if (n > 0) {
        DevBuf<int32_t> P, gstart, rank, out, err;
        DevBuf<uint8_t> ignore, contained, tie;
        DevBuf<uint32_t> pre, pos, head, hscan, gid, cnt, off;
        DevBuf<uint64_t> totals;
        OvlView v = {s->d_q.p, s->d_t.p, s->d_file.p, s->d_ovl.p, s->d_q_s.p, s->d_q_e.p, s->d_q_l.p, s->d_t_s.p, s->d_t_e.p, s->d_t_l.p, s->d_flags.p,
                     s->d_ctg.p, s->d_blk.p, s->d_ph.p, s->d_lex.p, n};
        FZP_TRY(pre.alloc((size_t)n)); FZP_TRY(pos.alloc((size_t)n)); FZP_TRY(err.alloc(1)); FZP_TRY(totals.alloc(4));
        const int32_t no_err = 0x7fffffff;
        FZP_HIP(hipMemcpyAsync(err.p, &no_err, 4, hipMemcpyHostToDevice, st));
        { ProfScope ps(ctx, "ovl_pre"); hipLaunchKernelGGL(k_ovl_pre, dim3(blocks_for(n, 256)), dim3(256), 0, st, v, pre.p, err.p); }
        FZP_TRY(fzp_exclusive_scan_u32(ctx, pre.p, pos.p, (size_t)n, totals.p + 0));
        uint64_t tot = 0;
        int32_t h_err = 0;
        FZP_HIP(hipMemcpyAsync(&tot, totals.p, 8, hipMemcpyDeviceToHost, st));
        FZP_HIP(hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, st));
        FZP_HIP(hipStreamSynchronize(st));
        if (h_err != no_err) {
            fzp_set_error("overlap line %d passes the phase checks but has a field int()/float() would reject (ValueError/IndexError in filter_stage1) or an integer beyond 32 bits", h_err);
            return FZP_EINVAL;
        }
        np = (int64_t)tot;
        if (np > 0) {
            FZP_TRY(P.alloc((size_t)np)); FZP_TRY(head.alloc((size_t)np)); FZP_TRY(hscan.alloc((size_t)np)); FZP_TRY(gid.alloc((size_t)np)); FZP_TRY(rank.alloc((size_t)np));
            hipLaunchKernelGGL(k_ovl_compact, dim3(blocks_for(n, 256)), dim3(256), 0, st, n, pre.p, pos.p, P.p);
            { ProfScope ps(ctx, "ovl_heads"); hipLaunchKernelGGL(k_ovl_heads, dim3(blocks_for(np, 256)), dim3(256), 0, st, v, np, P.p, head.p); }
            FZP_TRY(fzp_exclusive_scan_u32(ctx, head.p, hscan.p, (size_t)np, totals.p + 1));
            FZP_HIP(hipMemcpyAsync(&tot, totals.p + 1, 8, hipMemcpyDeviceToHost, st));
            FZP_HIP(hipStreamSynchronize(st));
            ng = (int64_t)tot;
            FZP_TRY(gstart.alloc((size_t)ng + 1)); FZP_TRY(cnt.alloc((size_t)ng * 2)); FZP_TRY(off.alloc((size_t)ng * 2)); FZP_TRY(tie.alloc((size_t)ng * 2));
            FZP_TRY(ignore.alloc(na + 1)); FZP_TRY(contained.alloc(na + 1));
            FZP_TRY(ignore.zero(na + 1, st)); FZP_TRY(contained.zero(na + 1, st));
            hipLaunchKernelGGL(k_ovl_compact, dim3(blocks_for(np, 256)), dim3(256), 0, st, np, head.p, hscan.p, gstart.p);
            const int32_t np32 = (int32_t)np;
            FZP_HIP(hipMemcpyAsync(gstart.p + ng, &np32, 4, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_ovl_gid, dim3(blocks_for(np, 256)), dim3(256), 0, st, np, head.p, hscan.p, gid.p);
            const unsigned gblocks = (unsigned)std::min<int64_t>(blocks_for(ng, 4), 1 << 16);
            { ProfScope ps(ctx, "ovl_stage1"); hipLaunchKernelGGL(k_ovl_stage1, dim3(gblocks), dim3(256), 0, st, v, ng, gstart.p, P.p, pr->max_diff, pr->max_cov, pr->min_cov, min_len, ignore.p); }
            { ProfScope ps(ctx, "ovl_stage2"); hipLaunchKernelGGL(k_ovl_stage2, dim3(blocks_for(np, 256)), dim3(256), 0, st, v, np, P.p, min_len, ignore.p, contained.p); }
            { ProfScope ps(ctx, "ovl_rank"); hipLaunchKernelGGL(k_ovl_rank, dim3(gblocks), dim3(256), 0, st, v, ng, gstart.p, P.p, min_len, pr->bestn, ignore.p, contained.p, rank.p, cnt.p, tie.p); }
            FZP_TRY(fzp_exclusive_scan_u32(ctx, cnt.p, off.p, (size_t)ng * 2, totals.p + 2));
            FZP_HIP(hipMemcpyAsync(&tot, totals.p + 2, 8, hipMemcpyDeviceToHost, st));
            FZP_HIP(hipStreamSynchronize(st));
            const int64_t n_out = (int64_t)tot;
            FZP_TRY(out.alloc((size_t)n_out));
            { ProfScope ps(ctx, "ovl_emit"); hipLaunchKernelGGL(k_ovl_emit, dim3(blocks_for(np, 256)), dim3(256), 0, st, v, np, P.p, gid.p, min_len, ignore.p, contained.p, rank.p, off.p, out.p); }
            h_out.resize((size_t)n_out); h_tie.resize((size_t)ng * 2);
            FZP_TRY(out.download(h_out.data(), (size_t)n_out, st));
            FZP_TRY(tie.download(h_tie.data(), (size_t)ng * 2, st));
            FZP_TRY(ignore.download(h_ignore.data(), na + 1, st));
            FZP_TRY(contained.download(h_contained.data(), na + 1, st));
            FZP_HIP(hipStreamSynchronize(st));
            bool any_tie = false;
            for (uint8_t x : h_tie) any_tie |= x != 0;
            if (any_tie) {
                h_P.resize((size_t)np); h_gstart.resize((size_t)ng + 1); h_off.resize((size_t)ng * 2); h_cnt.resize((size_t)ng * 2);
                FZP_TRY(P.download(h_P.data(), (size_t)np, st)); FZP_TRY(gstart.download(h_gstart.data(), (size_t)ng + 1, st));
                FZP_TRY(off.download(h_off.data(), (size_t)ng * 2, st)); FZP_TRY(cnt.download(h_cnt.data(), (size_t)ng * 2, st));
                FZP_HIP(hipStreamSynchronize(st));
            }
            FZP_HIP(hipGetLastError());
        }
    }
    // ---- lists that tie on (numeric key, partner): the reference falls through to comparing the token lists (:218-219)
    if (!h_P.empty()) {
        Tok toks[MAXTOK];
        auto cand = [&](int32_t i, HostLine *hl) -> int {
            host_line(s, i, hl, toks);
            const int32_t q = hl->q, t = hl->t;
            if (h_contained[(size_t)q] || h_contained[(size_t)t] || h_ignore[(size_t)q] || h_ignore[(size_t)t]) return -1;
            if (!(hl->flags & F_IDT_OK) || hl->v[3] < min_len || hl->v[6] < min_len) return -1;
            if (hl->v[1] == 0) return 0;
            if (hl->v[2] == hl->v[3]) return 1;
            return -1;
        };
        struct HC { int32_t ninph, negovl, m_range, row; int64_t seq; };
        for (int64_t g = 0; g < ng; g++)
            for (int e = 0; e < 2; e++) {
                if (!h_tie[(size_t)(2 * g + e)]) continue;
                std::vector<HC> c;
                for (int32_t k = h_gstart[(size_t)g]; k < h_gstart[(size_t)g + 1]; k++) {
                    const int32_t i = h_P[(size_t)k];
                    HostLine hl;
                    if (cand(i, &hl) != e) continue;
                    const int32_t q = hl.q, t = hl.t;
                    const bool inph = s->ctg_code[(size_t)q] == s->ctg_code[(size_t)t] && s->blk_code[(size_t)q] == s->blk_code[(size_t)t] && s->ph_code[(size_t)q] == s->ph_code[(size_t)t];
                    c.push_back({inph ? 0 : 1, (int32_t)hl.v[0], (int32_t)(hl.v[6] - (hl.v[5] - hl.v[4])), i, (int64_t)k});
                }
                std::stable_sort(c.begin(), c.end(), [&](const HC &a, const HC &b) {
                    if (a.ninph != b.ninph) return a.ninph < b.ninph;
                    if (a.negovl != b.negovl) return a.negovl < b.negovl;
                    if (a.m_range != b.m_range) return a.m_range < b.m_range;
                    Tok ta[MAXTOK], tb[MAXTOK];
                    const int64_t la = a.row, lb = b.row;
                    const char *txa = s->text.base(s->line_off[(size_t)la]), *txb = s->text.base(s->line_off[(size_t)lb]);
                    const int na_ = split_line(txa, s->line_off[(size_t)la], s->line_off[(size_t)la + 1], ta, MAXTOK);
                    const int nb_ = split_line(txb, s->line_off[(size_t)lb], s->line_off[(size_t)lb + 1], tb, MAXTOK);
                    for (int z = 0; z < std::min(na_, nb_); z++) {
                        const std::string_view x(txa + ta[z].off, (size_t)ta[z].len), y(txb + tb[z].off, (size_t)tb[z].len);
                        if (x != y) return x < y;
                    }
                    return na_ < nb_;
                });
                const uint32_t o = h_off[(size_t)(2 * g + e)], m = h_cnt[(size_t)(2 * g + e)];
                for (uint32_t z = 0; z < m && z < c.size(); z++) h_out[(size_t)o + z] = c[z].row;
            }
    }
    // ---- results
    int64_t *rows = (int64_t *)malloc((h_out.size() ? h_out.size() : 1) * sizeof(int64_t));
    if (!rows) return FZP_ENOMEM;
    for (size_t z = 0; z < h_out.size(); z++) rows[z] = h_out[z];
    *rows_out = rows; *n_rows_out = (int64_t)h_out.size();
    auto list = [&](const std::vector<uint8_t> &f, int32_t **o, int64_t *no) -> int {
        if (!o) return FZP_OK;
        int64_t c = 0;
        for (size_t i = 0; i < na; i++) c += f[i] != 0;
        int32_t *p = (int32_t *)malloc((size_t)(c ? c : 1) * sizeof(int32_t));
        if (!p) return FZP_ENOMEM;
        int64_t w = 0;
        for (size_t i = 0; i < na; i++) if (f[i]) p[w++] = (int32_t)i;
        *o = p;
        if (no) *no = c;
        return FZP_OK;
    };
    FZP_TRY(list(h_ignore, ignore_out, n_ignore));
    FZP_TRY(list(h_contained, contained_out, n_contained));
    return FZP_OK;
}

extern "C" int fzp_ovl_format(const fzp_ovlset *s, const int64_t *rows, int64_t n_rows, char **text, size_t *len) {
    if (!s || (!rows && n_rows) || !text || !len || n_rows < 0) { fzp_set_error("fzp_ovl_format: bad arguments"); return FZP_EINVAL; }
    // the selected lines, re-split (tokens joined by single spaces, :277) with the two phase tags appended; chunks of rows on host threads
    const char *mt = s->map_text.data();
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)16, (int64_t)std::max(1u, std::thread::hardware_concurrency()), n_rows / 2048 + 1}));
    std::vector<std::string> part((size_t)T);
    std::vector<int64_t> bad((size_t)T, -1);
    auto work = [&](int w) {
        const int64_t z0 = n_rows * w / T, z1 = n_rows * (w + 1) / T;
        std::string &out = part[(size_t)w];
        out.reserve((size_t)(z1 - z0) * 96);
        Tok t[MAXTOK];
        for (int64_t z = z0; z < z1; z++) {
            const int64_t line = rows[z];
            if (line < 0 || line >= s->n_lines) { bad[(size_t)w] = line; return; }
            const char *tx = s->text.base(s->line_off[(size_t)line]);
            const int nt = split_line(tx, s->line_off[(size_t)line], s->line_off[(size_t)line + 1], t, MAXTOK);
            int32_t ids[2] = {-1, -1};
            if (nt >= 2)
                for (int k = 0; k < 2; k++) {
                    auto it = s->ids.find(std::string_view(tx + t[k].off, (size_t)t[k].len));
                    if (it != s->ids.end()) ids[k] = it->second;
                }
            if (ids[0] < 0 || ids[1] < 0) { bad[(size_t)w] = line; return; }
            for (int k = 0; k < nt && k < MAXTOK; k++) { if (k) out.push_back(' '); out.append(tx + t[k].off, (size_t)t[k].len); }
            for (int k = 0; k < 2; k++) {
                const size_t a = (size_t)ids[k];
                out.push_back(' ');
                out.append(mt + s->ctg[a].off, (size_t)s->ctg[a].len); out.push_back('.');
                out.append(mt + s->blk[a].off, (size_t)s->blk[a].len); out.push_back('.');
                out.append(mt + s->ph[a].off, (size_t)s->ph[a].len);
            }
            out.push_back('\n');
        }
    };
    {
        std::vector<std::thread> th;
        for (int w = 1; w < T; w++) th.emplace_back(work, w);
        work(0);
        for (auto &x : th) x.join();
    }
    for (int w = 0; w < T; w++) if (bad[(size_t)w] != -1) { fzp_set_error("fzp_ovl_format: line %lld is not a filterable row", (long long)bad[(size_t)w]); return FZP_EINVAL; }
    size_t total = 0;
    for (auto &q : part) total += q.size();
    char *p = (char *)malloc(total + 1);
    if (!p) return FZP_ENOMEM;
    size_t at = 0;
    for (auto &q : part) { memcpy(p + at, q.data(), q.size()); at += q.size(); }
    p[total] = 0;
    *text = p; *len = total;
    return FZP_OK;
}


// ================================================================================ raw-read tracker (rr_hctg_track.py)
// run_track_reads (:68-139): for every raw read that shows up as the B-read of an overlap keep its bestn best A-reads
// -- the heap of (overlap_len, q_id) tuples at :60-64 / :99-106 is "the bestn largest tuples" -- then score the contigs
// those A-reads map to.  Same dumps, same tokeniser as the overlap filter (ids in raw mode).  Device: filter + phase
// veto per line -> histogram of B-reads -> scan -> scatter into per-B-read segments -> one wave per B-read: rank the
// segment, keep bestn, accumulate (contig -> score, count) in LDS, rank the contigs -> rows.  The reference's line
// order is a dict order (:111), contigs of equal score keep dict order (:125): unspecified -> canonical order here
// (B-read, score, contig string), as in the oracle and the fixtures.
namespace {
struct TrkPhase { int32_t has, ctg, block, phase; };

__global__ void __launch_bounds__(256) k_trk_filter(int64_t n, const int32_t *__restrict__ qv, const int32_t *__restrict__ tv, const int32_t *__restrict__ t_l,
                                                    const uint8_t *__restrict__ flags, int32_t min_len, const int32_t *__restrict__ rid_of, int64_t max_rid,
                                                    const TrkPhase *__restrict__ rph, int64_t n_rid, uint32_t *__restrict__ pass, uint32_t *__restrict__ bcount,
                                                    int32_t *__restrict__ err_parse, int32_t *__restrict__ err_id) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t ok = 0;
    if (!(flags[i] & F_PARSE_OK)) atomicMin(err_parse, (int32_t)i);            // every line is parsed before any test (:38-42)
    else if (t_l[i] >= min_len) {
        const int32_t q = qv[i], t = tv[i];
        if (q < 0 || t < 0) atomicMin(err_id, (int32_t)i);                     // not a '%09d' id: outside this implementation
        else if (q <= max_rid && rid_of[q] >= 0) {                             // `q_id in rid_to_ctg`
            if (t >= n_rid) atomicMin(err_id, (int32_t)i);                     // IndexError at :50
            else {
                ok = 1;
                const TrkPhase tp = rph[t];
                if (tp.has && tp.block != -1) {
                    if (q >= n_rid) atomicMin(err_id, (int32_t)i);
                    else {
                        const TrkPhase qp = rph[q];
                        if (qp.has && qp.ctg == tp.ctg && qp.block == tp.block && qp.phase != tp.phase) ok = 0;
                    }
                }
                if (ok) atomicAdd(&bcount[t], 1u);
            }
        }
    }
    pass[i] = ok;
}
__global__ void __launch_bounds__(256) k_trk_scatter(int64_t n, const uint32_t *__restrict__ pass, const int32_t *__restrict__ qv, const int32_t *__restrict__ tv,
                                                     const int32_t *__restrict__ ovl, const uint32_t *__restrict__ boff, uint32_t *__restrict__ bfill,
                                                     int2 *__restrict__ seg) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !pass[i]) return;
    const int32_t t = tv[i];
    seg[boff[t] + atomicAdd(&bfill[t], 1u)] = make_int2(ovl[i], qv[i]);
}
struct TrkRow { int32_t bread, ctg, count, rank; int64_t score; int32_t in_ctg, pad_; };
constexpr int TRK_CAP = 64;     // distinct contigs one B-read's kept A-reads may map to

// one wave per B-read with hits
__global__ void __launch_bounds__(256) k_trk_reads(int64_t n_b, const int32_t *__restrict__ blist, const uint32_t *__restrict__ boff, const uint32_t *__restrict__ bcount,
                                                   const int2 *__restrict__ seg, int64_t bestn, const int32_t *__restrict__ rid_of, int64_t max_rid,
                                                   const int64_t *__restrict__ rc_off, const int32_t *__restrict__ rc_ctg, const int32_t *__restrict__ ctg_lex,
                                                   TrkRow *__restrict__ rows, uint32_t *__restrict__ n_rows, uint32_t row_cap, int32_t *__restrict__ err_cap) {
    __shared__ int32_t l_key[4][TRK_CAP];
    __shared__ int32_t l_cnt[4][TRK_CAP];
    __shared__ long long l_score[4][TRK_CAP];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    for (int64_t bi = (int64_t)blockIdx.x * 4 + wv; bi < n_b; bi += (int64_t)gridDim.x * 4) {
        const int32_t t = blist[bi];
        const uint32_t s0 = boff[t], m = bcount[t];
        for (int x = lane; x < TRK_CAP; x += 64) { l_key[wv][x] = -1; l_cnt[wv][x] = 0; l_score[wv][x] = 0; }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t k0 = 0; k0 < m; k0 += 64) {
            const uint32_t k = k0 + lane;
            int2 a = make_int2(0, 0);
            uint32_t r = 0;
            if (k < m) a = seg[s0 + k];
            for (uint32_t kb = 0; kb < m; kb++) {               // the heap keeps the largest (overlap_len, q_id) tuples
                const int2 b = seg[s0 + kb];
                r += (b.x > a.x || (b.x == a.x && (b.y > a.y || (b.y == a.y && kb < k)))) ? 1u : 0u;
            }
            if (k < m && (int64_t)r < bestn) {
                const int32_t rq = rid_of[a.y];                  // >= 0: the line passed `q_id in rid_to_ctg`
                for (int64_t z = rc_off[rq]; z < rc_off[rq + 1]; z++) {
                    const int32_t c = rc_ctg[z];
                    int slot = c & (TRK_CAP - 1), tries = 0;
                    for (; tries < TRK_CAP; tries++, slot = (slot + 1) & (TRK_CAP - 1)) {
                        const int32_t old = atomicCAS(&l_key[wv][slot], -1, c);
                        if (old == -1 || old == c) break;
                    }
                    if (tries == TRK_CAP) { atomicMin(err_cap, t); continue; }
                    atomicAdd(&l_cnt[wv][slot], 1);
                    atomicAdd((unsigned long long *)&l_score[wv][slot], (unsigned long long)(long long)(-(long long)a.x));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // rank the contigs by (score, contig string) and write the rows
        const int32_t c = lane < TRK_CAP ? l_key[wv][lane] : -1;
        const long long sc = lane < TRK_CAP ? l_score[wv][lane] : 0;
        int32_t rk = 0;
        for (int x = 0; x < TRK_CAP; x++) {
            const int32_t oc = l_key[wv][x];
            if (oc < 0 || oc == c) continue;
            const long long os = l_score[wv][x];
            rk += (os < sc || (os == sc && ctg_lex[oc] < ctg_lex[c < 0 ? 0 : c])) ? 1 : 0;
        }
        const uint64_t have = __ballot(c >= 0);
        uint32_t base = 0;
        if (lane == 0 && have) base = atomicAdd(n_rows, (uint32_t)__popcll(have));
        base = __builtin_amdgcn_readfirstlane(base);
        if (c >= 0) {
            const uint32_t at = base + (uint32_t)__popcll(have & ((1ull << lane) - 1ull));
            if (at < row_cap) {
                TrkRow o;
                o.bread = t; o.ctg = c; o.count = l_cnt[wv][lane]; o.rank = rk; o.score = sc; o.pad_ = 0;
                o.in_ctg = 0;
                if (t <= max_rid && rid_of[t] >= 0) {
                    const int32_t rt = rid_of[t];
                    for (int64_t z = rc_off[rt]; z < rc_off[rt + 1]; z++) if (rc_ctg[z] == c) o.in_ctg = 1;
                }
                rows[at] = o;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}
__global__ void __launch_bounds__(256) k_trk_blist(int64_t n_rid, const uint32_t *__restrict__ bcount, const uint32_t *__restrict__ pos, int32_t *__restrict__ blist,
                                                   uint32_t *__restrict__ flag) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_rid) return;
    if (flag) { flag[t] = bcount[t] ? 1u : 0u; return; }
    if (bcount[t]) blist[pos[t]] = (int32_t)t;
}
}  // namespace

extern "C" int fzp_track_reads(fzp_ctx *ctx, int32_t n_files, const char *const *texts, const size_t *lens, const char *phased_reads, size_t pr_len,
                               const char *read_to_contig_map, size_t rc_len, const char *rawread_ids, size_t ri_len, int64_t min_len, int64_t bestn,
                               char **text_out, size_t *len_out) {
    if (!ctx || n_files < 0 || (n_files && (!texts || !lens)) || !text_out || !len_out) { fzp_set_error("fzp_track_reads: bad arguments"); return FZP_EINVAL; }
    *text_out = nullptr; *len_out = 0;
    FZP_TRY(fzp_bind(ctx));
    hipStream_t st = ctx->stream;
    // ---- read_to_contig_map (:14-23): rid -> set of contigs.  rids must be '%09d'-shaped (fc_get_read_hctg_map writes them so)
    std::string rc_text(read_to_contig_map ? read_to_contig_map : "", rc_len), pr_text(phased_reads ? phased_reads : "", pr_len);
    std::unordered_map<std::string_view, int32_t> ctg_ids;
    std::vector<std::string_view> ctg_names;
    auto ctg_of = [&](std::string_view v) {
        auto it = ctg_ids.find(v);
        if (it != ctg_ids.end()) return it->second;
        const int32_t c = (int32_t)ctg_names.size();
        ctg_ids.emplace(v, c); ctg_names.push_back(v);
        return c;
    };
    std::vector<std::pair<int32_t, int32_t>> pairs;     // (rid value, contig)
    int64_t max_rid = -1;
    {
        const char *mt = rc_text.data();
        const int64_t n = (int64_t)rc_text.size();
        for (int64_t b = 0, row = 0; b < n; row++) {
            const void *nlp = memchr(mt + b, '\n', (size_t)(n - b));
            const int64_t e = nlp ? (const char *)nlp - mt : n;
            Tok t[6];
            const int nt = split_line(mt, b, e, t, 6);
            b = nlp ? e + 1 : n;
            if (nt != 4) { fzp_set_error("read_to_contig_map row %lld has %d fields (ValueError at rr_hctg_track.py:20)", (long long)row, nt); return FZP_EINVAL; }
            int64_t v;
            if (!canonical_id(mt + t[1].off, t[1].len, &v) || t[1].len != 9) { fzp_set_error("read_to_contig_map row %lld: read id is not a 9-digit decimal (unsupported)", (long long)row); return FZP_EINVAL; }
            pairs.push_back({(int32_t)v, ctg_of(std::string_view(mt + t[3].off, (size_t)t[3].len))});
            max_rid = std::max(max_rid, v);
        }
    }
    // ---- phased reads -> oid -> (ctg, block, phase), later rows win (:73-81); rawread_ids.split('\n') (:83)
    std::unordered_map<std::string_view, TrkPhase> oid_phase;
    {
        const char *mt = pr_text.data();
        const int64_t n = (int64_t)pr_text.size();
        for (int64_t b = 0, row = 0; b < n; row++) {
            const void *nlp = memchr(mt + b, '\n', (size_t)(n - b));
            const int64_t e = nlp ? (const char *)nlp - mt : n;
            Tok t[8];
            const int nt = split_line(mt, b, e, t, 8);
            b = nlp ? e + 1 : n;
            int64_t blk, ph;
            if (nt < 7 || !parse_int32(mt + t[2].off, t[2].len, &blk) || !parse_int32(mt + t[3].off, t[3].len, &ph)) {
                fzp_set_error("phased-read-file row %lld is malformed (IndexError/ValueError at rr_hctg_track.py:76-80)", (long long)row);
                return FZP_EINVAL;
            }
            oid_phase[std::string_view(mt + t[6].off, (size_t)t[6].len)] = TrkPhase{1, ctg_of(std::string_view(mt + t[1].off, (size_t)t[1].len)), (int32_t)blk, (int32_t)ph};
        }
    }
    std::vector<TrkPhase> rph;
    {
        size_t b = 0;
        for (size_t i = 0; i <= ri_len; i++)
            if (i == ri_len || rawread_ids[i] == '\n') {
                auto it = oid_phase.find(std::string_view(rawread_ids + b, i - b));
                rph.push_back(it == oid_phase.end() ? TrkPhase{0, 0, 0, 0} : it->second);
                b = i + 1;
            }
    }
    const int64_t n_rid = (int64_t)rph.size();
    // CSR rid value -> distinct contigs; contig lexicographic ranks
    std::sort(pairs.begin(), pairs.end());
    pairs.erase(std::unique(pairs.begin(), pairs.end()), pairs.end());
    std::vector<int32_t> rid_of((size_t)max_rid + 2, -1), rc_ctg;
    std::vector<int64_t> rc_off(1, 0);
    int64_t k_max = 1;
    for (size_t i = 0; i < pairs.size();) {
        size_t j2 = i;
        while (j2 < pairs.size() && pairs[j2].first == pairs[i].first) { rc_ctg.push_back(pairs[j2].second); j2++; }
        rid_of[(size_t)pairs[i].first] = (int32_t)rc_off.size() - 1;
        rc_off.push_back((int64_t)rc_ctg.size());
        k_max = std::max<int64_t>(k_max, (int64_t)(j2 - i));
        i = j2;
    }
    std::vector<int32_t> ctg_lex(ctg_names.size() + 1, 0);
    {
        std::vector<int32_t> ord(ctg_names.size());
        for (size_t i = 0; i < ord.size(); i++) ord[i] = (int32_t)i;
        std::sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) { return ctg_names[(size_t)a] < ctg_names[(size_t)b]; });
        for (size_t r = 0; r < ord.size(); r++) ctg_lex[(size_t)ord[r]] = (int32_t)r;
    }
    // ---- the dumps: tokenise on the device (raw-id mode)
    fzp_ovlset *s = new fzp_ovlset();
    struct Guard { fzp_ovlset *p; ~Guard() { delete p; } } guard{s};
    s->device = ctx->device;
    for (int k = 0; k < n_files; k++) {
        if (!s->text.add(texts[k], lens[k])) { fzp_set_error("fzp_track_reads: host allocation failed"); return FZP_ENOMEM; }
        s->file_end.push_back((int64_t)s->text.size());
    }
    if (s->text.size() >= (1ull << 32)) { fzp_set_error("fzp_track_reads: %zu bytes of text (limit 4 GiB per call)", s->text.size()); return FZP_EINVAL; }
    std::vector<TrkRow> h_rows;
    if (s->text.size() > 0) {
        DevBuf<uint8_t> d_text;
        FZP_TRY(upload_text(s->text, d_text, st));
        FZP_TRY(ovl_tokenise_device(ctx, s, 0, std::vector<int32_t>(), d_text));
        const int64_t n = s->n_lines;
        DevBuf<int32_t> d_rid_of, d_rc_ctg, d_ctg_lex, d_blist, e1, e2, e3;
        DevBuf<int64_t> d_rc_off;
        DevBuf<TrkPhase> d_rph;
        DevBuf<uint32_t> pass, bcount, boff, bfill, bflag, bpos, n_rows;
        DevBuf<int2> seg;
        DevBuf<TrkRow> rows;
        DevBuf<uint64_t> totals;
        FZP_TRY(d_rid_of.upload(rid_of.data(), rid_of.size(), st)); FZP_TRY(d_rc_ctg.upload(rc_ctg.data(), rc_ctg.size(), st)); FZP_TRY(d_rc_off.upload(rc_off.data(), rc_off.size(), st));
        FZP_TRY(d_ctg_lex.upload(ctg_lex.data(), ctg_lex.size(), st)); FZP_TRY(d_rph.upload(rph.data(), rph.size(), st));
        FZP_TRY(pass.alloc((size_t)n)); FZP_TRY(bcount.alloc((size_t)n_rid)); FZP_TRY(boff.alloc((size_t)n_rid)); FZP_TRY(bfill.alloc((size_t)n_rid));
        FZP_TRY(bflag.alloc((size_t)n_rid)); FZP_TRY(bpos.alloc((size_t)n_rid)); FZP_TRY(totals.alloc(2)); FZP_TRY(n_rows.alloc(1));
        FZP_TRY(e1.alloc(1)); FZP_TRY(e2.alloc(1)); FZP_TRY(e3.alloc(1));
        FZP_TRY(bcount.zero((size_t)n_rid, st)); FZP_TRY(bfill.zero((size_t)n_rid, st)); FZP_TRY(n_rows.zero(1, st));
        const int32_t none = 0x7fffffff;
        FZP_HIP(hipMemcpyAsync(e1.p, &none, 4, hipMemcpyHostToDevice, st)); FZP_HIP(hipMemcpyAsync(e2.p, &none, 4, hipMemcpyHostToDevice, st));
        FZP_HIP(hipMemcpyAsync(e3.p, &none, 4, hipMemcpyHostToDevice, st));
        const int32_t ml = (int32_t)std::max<int64_t>(-2147483647ll, std::min<int64_t>(min_len, 2147483647ll));
        { ProfScope ps(ctx, "trk_filter"); hipLaunchKernelGGL(k_trk_filter, dim3(blocks_for(n, 256)), dim3(256), 0, st, n, s->d_q.p, s->d_t.p, s->d_t_l.p, s->d_flags.p, ml, d_rid_of.p, max_rid, d_rph.p, n_rid, pass.p, bcount.p, e1.p, e2.p); }
        FZP_TRY(fzp_exclusive_scan_u32(ctx, bcount.p, boff.p, (size_t)n_rid, totals.p));
        hipLaunchKernelGGL(k_trk_blist, dim3(blocks_for(n_rid, 256)), dim3(256), 0, st, n_rid, bcount.p, (const uint32_t *)nullptr, (int32_t *)nullptr, bflag.p);
        FZP_TRY(fzp_exclusive_scan_u32(ctx, bflag.p, bpos.p, (size_t)n_rid, totals.p + 1));
        uint64_t tot[2] = {0, 0};
        int32_t h_e1 = 0, h_e2 = 0;
        FZP_HIP(hipMemcpyAsync(tot, totals.p, 16, hipMemcpyDeviceToHost, st));
        FZP_HIP(hipMemcpyAsync(&h_e1, e1.p, 4, hipMemcpyDeviceToHost, st)); FZP_HIP(hipMemcpyAsync(&h_e2, e2.p, 4, hipMemcpyDeviceToHost, st));
        FZP_HIP(hipStreamSynchronize(st));
        if (h_e1 != none) { fzp_set_error("overlap line %d has a field int()/float() rejects or fewer than 12 columns (rr_hctg_track.py:36-42)", h_e1); return FZP_EINVAL; }
        if (h_e2 != none) { fzp_set_error("overlap line %d: read id is not a 9-digit decimal, or is beyond rawread_ids (IndexError at rr_hctg_track.py:50)", h_e2); return FZP_EINVAL; }
        const int64_t n_hits = (int64_t)tot[0], n_b = (int64_t)tot[1];
        if (n_b > 0) {
            const uint64_t cap64 = (uint64_t)std::min<int64_t>(n_hits, n_b * std::max<int64_t>(bestn, 0)) * (uint64_t)k_max + 64;
            if (cap64 >= (1ull << 31)) { fzp_set_error("fzp_track_reads: output bound of %llu rows is too large for one call", (unsigned long long)cap64); return FZP_EINVAL; }
            FZP_TRY(seg.alloc((size_t)n_hits)); FZP_TRY(d_blist.alloc((size_t)n_b)); FZP_TRY(rows.alloc((size_t)cap64));
            hipLaunchKernelGGL(k_trk_scatter, dim3(blocks_for(n, 256)), dim3(256), 0, st, n, pass.p, s->d_q.p, s->d_t.p, s->d_ovl.p, boff.p, bfill.p, seg.p);
            hipLaunchKernelGGL(k_trk_blist, dim3(blocks_for(n_rid, 256)), dim3(256), 0, st, n_rid, bcount.p, bpos.p, d_blist.p, (uint32_t *)nullptr);
            { ProfScope ps(ctx, "trk_reads"); hipLaunchKernelGGL(k_trk_reads, dim3((unsigned)std::min<int64_t>(blocks_for(n_b, 4), 1 << 16)), dim3(256), 0, st, n_b, d_blist.p, boff.p, bcount.p, seg.p, bestn, d_rid_of.p, max_rid, d_rc_off.p, d_rc_ctg.p, d_ctg_lex.p, rows.p, n_rows.p, (uint32_t)cap64, e3.p); }
            uint32_t h_n = 0;
            int32_t h_e3 = 0;
            FZP_HIP(hipMemcpyAsync(&h_n, n_rows.p, 4, hipMemcpyDeviceToHost, st)); FZP_HIP(hipMemcpyAsync(&h_e3, e3.p, 4, hipMemcpyDeviceToHost, st));
            FZP_HIP(hipStreamSynchronize(st));
            if (h_e3 != none) { fzp_set_error("read %d: its best hits map to more than %d contigs (limit of this implementation)", h_e3, TRK_CAP); return FZP_EINVAL; }
            h_rows.resize(h_n);
            FZP_TRY(rows.download(h_rows.data(), h_n, st));
            FZP_HIP(hipStreamSynchronize(st));
        }
        FZP_HIP(hipGetLastError());
    }
    // ---- canonical order and text (:134): bread ctg count rank score in_ctg
    std::sort(h_rows.begin(), h_rows.end(), [](const TrkRow &a, const TrkRow &b) { return a.bread != b.bread ? a.bread < b.bread : a.rank < b.rank; });
    std::string out;
    char num[160];
    for (const TrkRow &r : h_rows) {
        const int k = snprintf(num, sizeof num, "%09d ", r.bread);
        out.append(num, (size_t)k);
        out.append(ctg_names[(size_t)r.ctg]);
        const int k2 = snprintf(num, sizeof num, " %d %d %lld %d\n", r.count, r.rank, (long long)r.score, r.in_ctg);
        out.append(num, (size_t)k2);
    }
    char *p = (char *)malloc(out.size() + 1);
    if (!p) return FZP_ENOMEM;
    memcpy(p, out.data(), out.size());
    p[out.size()] = 0;
    *text_out = p; *len_out = out.size();
    return FZP_OK;
}
