// fzp_ovlp.hip -- overlap filter with phase (falcon_unzip/ovlp_filter_with_phase.py:49-354), SURVEY section 8f row n2.
//
// The reference streams `LA4Falcon -mo` text three times and keeps Python dicts / sets keyed by id strings.  Here:
//   host    one pass over the text: tokenise, intern ids / contigs / blocks / phases (strings stay strings: they are
//           only ever compared for equality, plus one lexicographic rank for sort ties), parse the numeric columns of
//           every line whose two ids are in the map (leniently: a bad field only matters if the line passes the phase
//           checks, which is decided on the device, as in the reference where such lines are never parsed);
//   device  K_pre     the four phase checks of every stage (:64-73)
//           K_heads   query groups = runs of equal q among the lines that passed, per file (:77, :215)
//           K_stage1  5'/3' counts per group -> ignore flags (:79-87, :96-119)
//           K_stage2  containment flags (:165-181)
//           K_rank    per group and read end: rank of every candidate under the reference's sort key
//                     (-inphase, -overlap_len, t_l - (t_e - t_s), then the token list) and the best-n cut (:221-235)
//           K_emit    selected lines in print order (slots from scans: no order depends on atomics)
//   host    groups whose candidates tie on the whole numeric key AND partner id (the same pair listed twice with equal
//           length and span) are re-ordered by the reference's last key, the comparison of the lines' token lists.
// HBM-bound integer scans: 42 B per row in, 8 B per selected row out; groups are small (tens to hundreds of rows), so
// the all-pairs ranking inside a group is cheap and needs no global sort.
#include <algorithm>
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

struct fzp_ovlset {
    std::string text;                       // all dumps, each ending with '\n'
    std::vector<int64_t> line_off;          // [n_lines + 1]
    std::string map_text;
    std::vector<Tok> key, ctg, blk, ph;     // per distinct map key (first-appearance order), offsets into map_text
    std::vector<int32_t> ctg_code, blk_code, ph_code, lex_rank;
    // rows: lines whose q_id and t_id are both map keys
    std::vector<int64_t> row_line;
    std::vector<int32_t> row_file, q, t, ovl, q_s, q_e, q_l, t_s, t_e, t_l;
    std::vector<uint8_t> flags;
};

extern "C" void fzp_ovlset_free(fzp_ovlset *s) { delete s; }
extern "C" int64_t fzp_ovl_n_lines(const fzp_ovlset *s) { return s ? (int64_t)s->line_off.size() - 1 : 0; }
extern "C" int64_t fzp_ovl_n_rows(const fzp_ovlset *s) { return s ? (int64_t)s->row_line.size() : 0; }
extern "C" int fzp_ovl_id_name(const fzp_ovlset *s, int32_t id, const char **name, int32_t *len) {
    if (!s || id < 0 || (size_t)id >= s->key.size() || !name || !len) { fzp_set_error("fzp_ovl_id_name: bad arguments"); return FZP_EINVAL; }
    *name = s->map_text.data() + s->key[(size_t)id].off;
    *len = s->key[(size_t)id].len;
    return FZP_OK;
}

extern "C" int fzp_ovl_parse(int32_t n_files, const char *const *texts, const size_t *lens, const char *rid_map, size_t map_len, fzp_ovlset **out) {
    if (!out || n_files < 0 || (n_files && (!texts || !lens)) || (!rid_map && map_len)) { fzp_set_error("fzp_ovl_parse: bad arguments"); return FZP_EINVAL; }
    *out = nullptr;
    fzp_ovlset *s = new fzp_ovlset();
    // ---- rid_to_phase.all (:306-309)
    s->map_text.assign(rid_map ? rid_map : "", map_len);
    std::unordered_map<std::string_view, int32_t> ids, strs;
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
        int64_t b = 0;
        int64_t row = 0;
        while (b < n) {
            const void *nlp = memchr(mt + b, '\n', (size_t)(n - b));
            const int64_t e = nlp ? (const char *)nlp - mt : n;
            Tok t[5];
            const int nt = split_line(mt, b, e, t, 5);
            if (nt < 4) { fzp_set_error("rid_phase_map row %lld has %d fields (IndexError at ovlp_filter_with_phase.py:309)", (long long)row, nt); delete s; return FZP_EINVAL; }
            std::string_view k(mt + t[0].off, (size_t)t[0].len);
            auto it = ids.find(k);
            size_t at;
            if (it == ids.end()) {
                at = s->key.size();
                ids.emplace(k, (int32_t)at);
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
    {
        std::vector<int32_t> ord(na);
        for (size_t i = 0; i < na; i++) ord[i] = (int32_t)i;
        const char *mt = s->map_text.data();
        std::sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) {
            return std::string_view(mt + s->key[(size_t)a].off, (size_t)s->key[(size_t)a].len) < std::string_view(mt + s->key[(size_t)b].off, (size_t)s->key[(size_t)b].len);
        });
        for (size_t r = 0; r < na; r++) s->lex_rank[(size_t)ord[r]] = (int32_t)r;
    }
    // ---- the dumps
    size_t total = 0;
    for (int k = 0; k < n_files; k++) total += lens[k] + 1;
    s->text.reserve(total);
    std::vector<int64_t> file_end;
    for (int k = 0; k < n_files; k++) {
        if (lens[k]) s->text.append(texts[k], lens[k]);
        if (lens[k] && s->text.back() != '\n') s->text.push_back('\n');
        file_end.push_back((int64_t)s->text.size());
    }
    const char *tx = s->text.data();
    const int64_t n = (int64_t)s->text.size();
    int64_t b = 0;
    int file = 0;
    s->line_off.push_back(0);
    while (b < n) {
        while (file < n_files && b >= file_end[(size_t)file]) file++;
        const void *nlp = memchr(tx + b, '\n', (size_t)(n - b));
        const int64_t e = nlp ? (const char *)nlp - tx : n;
        Tok t[MAXTOK];
        const int nt = split_line(tx, b, e, t, MAXTOK);
        const int64_t line = (int64_t)s->line_off.size() - 1;
        if (nt < 2) { fzp_set_error("overlap line %lld has %d tokens (ValueError at ovlp_filter_with_phase.py:62)", (long long)line, nt); delete s; return FZP_EINVAL; }
        auto iq = ids.find(std::string_view(tx + t[0].off, (size_t)t[0].len));
        auto it = iq == ids.end() ? ids.end() : ids.find(std::string_view(tx + t[1].off, (size_t)t[1].len));
        if (iq != ids.end() && it != ids.end()) {
            int64_t v[7] = {0, 0, 0, 0, 0, 0, 0};
            double idt = 0;
            uint8_t fl = 0;
            static const int col[7] = {2, 5, 6, 7, 9, 10, 11};
            bool ok = nt >= 12 && nt <= MAXTOK;
            for (int c = 0; ok && c < 7; c++) ok = parse_int32(tx + t[col[c]].off, t[col[c]].len, &v[c]);
            if (ok) ok = parse_float(tx + t[3].off, t[3].len, &idt);
            if (ok && -v[0] > 2147483647ll) ok = false;
            if (ok) {
                fl |= F_PARSE_OK;
                if (!(idt < 90)) fl |= F_IDT_OK;
                const std::string_view tag(tx + t[nt - 1].off, (size_t)t[nt - 1].len);
                if (tag == "contains") fl |= F_CONTAINS;
                if (tag == "contained") fl |= F_CONTAINED;
            }
            s->row_line.push_back(line);
            s->row_file.push_back(file);
            s->q.push_back(iq->second); s->t.push_back(it->second);
            s->ovl.push_back((int32_t)-v[0]);
            s->q_s.push_back((int32_t)v[1]); s->q_e.push_back((int32_t)v[2]); s->q_l.push_back((int32_t)v[3]);
            s->t_s.push_back((int32_t)v[4]); s->t_e.push_back((int32_t)v[5]); s->t_l.push_back((int32_t)v[6]);
            s->flags.push_back(fl);
        }
        b = nlp ? e + 1 : n;
        s->line_off.push_back(b);
    }
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

// K_pre: the four checks every stage starts with (:64-73)
__global__ void __launch_bounds__(256) k_ovl_pre(OvlView v, uint32_t *__restrict__ pre, int32_t *__restrict__ err) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= v.n) return;
    const int32_t q = v.q[i], t = v.t[i];
    const bool keep = v.ctg[q] == v.ctg[t] && !(v.blk[q] == v.blk[t] && v.ph[q] != v.ph[t]);
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
inline unsigned blocks_for(int64_t n, int per) { return (unsigned)std::max<int64_t>(1, (n + per - 1) / per); }
}  // namespace

extern "C" int fzp_ovl_filter(fzp_ctx *ctx, const fzp_ovlset *s, const fzp_ovlp_params *pr, int64_t **rows_out, int64_t *n_rows_out, int32_t **ignore_out,
                              int64_t *n_ignore, int32_t **contained_out, int64_t *n_contained) {
    if (!ctx || !s || !pr || !rows_out || !n_rows_out) { fzp_set_error("fzp_ovl_filter: bad arguments"); return FZP_EINVAL; }
    *rows_out = nullptr; *n_rows_out = 0;
    if (ignore_out) { *ignore_out = nullptr; if (n_ignore) *n_ignore = 0; }
    if (contained_out) { *contained_out = nullptr; if (n_contained) *n_contained = 0; }
    FZP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t n = (int64_t)s->row_line.size();
    const size_t na = s->key.size();
    if (n >= (1ll << 31)) { fzp_set_error("fzp_ovl_filter: %lld rows (limit 2^31 per call)", (long long)n); return FZP_EINVAL; }
    const int32_t min_len = (int32_t)std::max<int64_t>(-2147483647ll, std::min<int64_t>(pr->min_len, 2147483647ll));
    std::vector<uint8_t> h_ignore(na + 1, 0), h_contained(na + 1, 0);
    std::vector<int32_t> h_out;
    std::vector<int32_t> h_P, h_gstart;
    std::vector<uint32_t> h_off, h_cnt;
    std::vector<uint8_t> h_tie;
    int64_t np = 0, ng = 0;
    if (n > 0) {
        DevBuf<int32_t> q, t, file, ovl, q_s, q_e, q_l, t_s, t_e, t_l, ctg, blk, ph, lex, P, gstart, rank, out, err;
        DevBuf<uint8_t> flags, ignore, contained, tie;
        DevBuf<uint32_t> pre, pos, head, hscan, gid, cnt, off;
        DevBuf<uint64_t> totals;
        FZP_TRY(q.upload(s->q.data(), (size_t)n, st)); FZP_TRY(t.upload(s->t.data(), (size_t)n, st)); FZP_TRY(file.upload(s->row_file.data(), (size_t)n, st));
        FZP_TRY(ovl.upload(s->ovl.data(), (size_t)n, st)); FZP_TRY(q_s.upload(s->q_s.data(), (size_t)n, st)); FZP_TRY(q_e.upload(s->q_e.data(), (size_t)n, st));
        FZP_TRY(q_l.upload(s->q_l.data(), (size_t)n, st)); FZP_TRY(t_s.upload(s->t_s.data(), (size_t)n, st)); FZP_TRY(t_e.upload(s->t_e.data(), (size_t)n, st));
        FZP_TRY(t_l.upload(s->t_l.data(), (size_t)n, st)); FZP_TRY(flags.upload(s->flags.data(), (size_t)n, st));
        FZP_TRY(ctg.upload(s->ctg_code.data(), na, st)); FZP_TRY(blk.upload(s->blk_code.data(), na, st)); FZP_TRY(ph.upload(s->ph_code.data(), na, st));
        FZP_TRY(lex.upload(s->lex_rank.data(), na, st));
        OvlView v = {q.p, t.p, file.p, ovl.p, q_s.p, q_e.p, q_l.p, t_s.p, t_e.p, t_l.p, flags.p, ctg.p, blk.p, ph.p, lex.p, n};
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
            fzp_set_error("overlap line %lld passes the phase checks but has a field int()/float() would reject (ValueError/IndexError in filter_stage1) or an integer beyond 32 bits",
                          (long long)s->row_line[(size_t)h_err]);
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
        const char *tx = s->text.data();
        auto cand = [&](int32_t i) -> int {
            const int32_t q = s->q[(size_t)i], t = s->t[(size_t)i];
            if (h_contained[(size_t)q] || h_contained[(size_t)t] || h_ignore[(size_t)q] || h_ignore[(size_t)t]) return -1;
            if (!(s->flags[(size_t)i] & F_IDT_OK) || s->q_l[(size_t)i] < min_len || s->t_l[(size_t)i] < min_len) return -1;
            if (s->q_s[(size_t)i] == 0) return 0;
            if (s->q_e[(size_t)i] == s->q_l[(size_t)i]) return 1;
            return -1;
        };
        struct HC { int32_t ninph, negovl, m_range, row; int64_t seq; };
        for (int64_t g = 0; g < ng; g++)
            for (int e = 0; e < 2; e++) {
                if (!h_tie[(size_t)(2 * g + e)]) continue;
                std::vector<HC> c;
                for (int32_t k = h_gstart[(size_t)g]; k < h_gstart[(size_t)g + 1]; k++) {
                    const int32_t i = h_P[(size_t)k];
                    if (cand(i) != e) continue;
                    const int32_t q = s->q[(size_t)i], t = s->t[(size_t)i];
                    const bool inph = s->ctg_code[(size_t)q] == s->ctg_code[(size_t)t] && s->blk_code[(size_t)q] == s->blk_code[(size_t)t] && s->ph_code[(size_t)q] == s->ph_code[(size_t)t];
                    c.push_back({inph ? 0 : 1, -s->ovl[(size_t)i], s->t_l[(size_t)i] - (s->t_e[(size_t)i] - s->t_s[(size_t)i]), i, (int64_t)k});
                }
                std::stable_sort(c.begin(), c.end(), [&](const HC &a, const HC &b) {
                    if (a.ninph != b.ninph) return a.ninph < b.ninph;
                    if (a.negovl != b.negovl) return a.negovl < b.negovl;
                    if (a.m_range != b.m_range) return a.m_range < b.m_range;
                    Tok ta[MAXTOK], tb[MAXTOK];
                    const int64_t la = s->row_line[(size_t)a.row], lb = s->row_line[(size_t)b.row];
                    const int na_ = split_line(tx, s->line_off[(size_t)la], s->line_off[(size_t)la + 1], ta, MAXTOK);
                    const int nb_ = split_line(tx, s->line_off[(size_t)lb], s->line_off[(size_t)lb + 1], tb, MAXTOK);
                    for (int z = 0; z < std::min(na_, nb_); z++) {
                        const std::string_view x(tx + ta[z].off, (size_t)ta[z].len), y(tx + tb[z].off, (size_t)tb[z].len);
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
    for (size_t z = 0; z < h_out.size(); z++) rows[z] = s->row_line[(size_t)h_out[z]];
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
    std::string out;
    const char *tx = s->text.data(), *mt = s->map_text.data();
    // line -> row (for the ids of the line): rows are ascending by line
    for (int64_t z = 0; z < n_rows; z++) {
        const int64_t line = rows[z];
        auto it = std::lower_bound(s->row_line.begin(), s->row_line.end(), line);
        if (line < 0 || it == s->row_line.end() || *it != line) { fzp_set_error("fzp_ovl_format: line %lld is not a filterable row", (long long)line); return FZP_EINVAL; }
        const size_t r = (size_t)(it - s->row_line.begin());
        Tok t[MAXTOK];
        const int nt = split_line(tx, s->line_off[(size_t)line], s->line_off[(size_t)line + 1], t, MAXTOK);
        for (int k = 0; k < nt && k < MAXTOK; k++) { if (k) out.push_back(' '); out.append(tx + t[k].off, (size_t)t[k].len); }
        const int32_t ids[2] = {s->q[r], s->t[r]};
        for (int k = 0; k < 2; k++) {
            const size_t a = (size_t)ids[k];
            out.push_back(' ');
            out.append(mt + s->ctg[a].off, (size_t)s->ctg[a].len); out.push_back('.');
            out.append(mt + s->blk[a].off, (size_t)s->blk[a].len); out.push_back('.');
            out.append(mt + s->ph[a].off, (size_t)s->ph[a].len);
        }
        out.push_back('\n');
    }
    char *p = (char *)malloc(out.size() + 1);
    if (!p) return FZP_ENOMEM;
    memcpy(p, out.data(), out.size());
    p[out.size()] = 0;
    *text = p; *len = out.size();
    return FZP_OK;
}
