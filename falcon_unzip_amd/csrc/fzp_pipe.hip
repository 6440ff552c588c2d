// fzp_pipe.hip -- the per-contig task chain of unzip_all as ONE call over many contigs (host orchestration, C++).
//
// The reference starts, per contig, a blasr job (task_run_blasr, unzip.py:61-99) and a phasing job (task_phasing,
// unzip.py:102-133 = fc_phasing.py + fc_phasing_readmap.py), each reading and writing files.  Here:
//   fzp_job_phase_write   inputs resident in HBM (an fzp_alnjob): K1 -> K5 in batched launches, records to the host, the
//                         two big files serialised on the device (fzp_text.hip), the small ones by host threads, every
//                         file of every contig written, rid_to_phase records returned for the gather;
//   fzp_phase_contigs     inputs in host memory: contigs are cut into groups sized to the trace-back budget and dealt to
//                         `n_lanes` host threads, each with its own ctx (own streams and allocator caches): while one lane
//                         stages, uploads or writes, the other lanes' kernels keep the GPU busy.
// File layout = the reference's: <out_dir>/<ctg>/{het_call/{variant_pos,variant_map,q_id_map}, g_atable/atable,
// get_phased_blocks/phased_variants, phased_reads, rid_to_phase.<ctg>}  (phasing.py:501-503,520,534,543; unzip.py:269).
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <sys/uio.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <string_view>
#include <thread>
#include <unordered_map>

#include "fzp_batch.h"
#include "fzp_fasta.h"

int fzp_batch_text_dev(fzp_ctx *ctx, fzp_batch *b, int what, DevBuf<char> &text, size_t *bytes, std::vector<int64_t> &ctg_begin);
int fzp_batch_consensus_dev(fzp_ctx *ctx, fzp_batch *b, int version, std::vector<fzp_tig> &tigs, DevBuf<uint8_t> &seq, uint64_t *n_seq, const fzp_cns_polish *polish = nullptr);      // K6 with the sequence bytes left on the device (fzp_cns.hip)
int fzp_batch_texts_dev(fzp_ctx *ctx, fzp_batch *b, DevBuf<char> &t_vmap, size_t *n_vmap, std::vector<int64_t> &vb, DevBuf<char> &t_atab, size_t *n_atab, std::vector<int64_t> &ab);      // both, one wait (fzp_text.hip)

// ---- background file writes (FZP_PIPE_ASYNC_WRITES): a few threads per ctx drain a queue of per-contig write tasks, so that the
// page-cache copies (and the file system's occasional throttling) of one call overlap the kernels of the next
struct FileWriter {
    std::mutex mu;
    std::condition_variable cv, idle;
    std::deque<std::function<void()>> q;
    std::vector<std::thread> th;
    int running = 0;
    bool stop = false;
    std::string first_error;
    std::atomic<int64_t> extra{0};      // bytes the tasks wrote beyond what their call had counted when it returned (the small texts are made by the task): taken by whoever flushes
    void start(int n) {
        for (int i = 0; i < n; i++)
            th.emplace_back([this, i] {
                { char nm[16]; snprintf(nm, sizeof nm, "fzp-wr%d", i); (void)pthread_setname_np(pthread_self(), nm); }
                // A rank with few cores puts its writers into the IDLE scheduling class (r5): they write 39 MB per bench step and are the largest consumer of such a rank's CPU
                // (13 of 28 ms), and whenever one of them holds a core the launch thread -- woken by the device, with a kernel to launch -- waits for a time slice.  Idle-class
                // threads run when nothing else wants the core: the two-core step went 21.4 -> 18.6 ms (1.28 -> 1.12 x the unconstrained one; tools/runs/writer_sched_ab.sh);
                // with sixteen cores nothing changes.  FZP_WRITER_SCHED = other | idle | batch overrides; push() below keeps a starved queue from growing.
                {
                    const char *e = getenv("FZP_WRITER_SCHED");
                    const int cls = e ? (!strcmp(e, "idle") ? SCHED_IDLE : (!strcmp(e, "batch") ? SCHED_BATCH : SCHED_OTHER)) : (cores_per_rank() <= 4 ? SCHED_IDLE : SCHED_OTHER);
                    if (cls != SCHED_OTHER) { struct sched_param sp; memset(&sp, 0, sizeof sp); (void)pthread_setschedparam(pthread_self(), cls, &sp); }
                }
                for (;;) {
                    std::function<void()> job;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [this] { return stop || !q.empty(); });
                        if (q.empty()) return;
                        job = std::move(q.front());
                        q.pop_front();
                        running++;
                    }
                    job();
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        running--;
                        if (q.empty() && running == 0) idle.notify_all();
                    }
                }
            });
    }
    static constexpr size_t MAX_QUEUED = 160;      // tasks (eight bench steps' worth): beyond that the pushing thread writes the oldest one itself -- a queue whose writers get no
                                                   // core (idle class on a saturated node) must not grow by a step's pinned blocks per step
    void push(std::function<void()> f) {
        std::function<void()> mine;
        {
            std::lock_guard<std::mutex> lk(mu);
            q.push_back(std::move(f));
            if (q.size() > MAX_QUEUED) { mine = std::move(q.front()); q.pop_front(); running++; }
        }
        cv.notify_one();
        if (mine) {
            mine();
            std::lock_guard<std::mutex> lk(mu);
            running--;
            if (q.empty() && running == 0) idle.notify_all();
        }
    }
    void fail(const std::string &e) { std::lock_guard<std::mutex> lk(mu); if (first_error.empty()) first_error = e; }
    std::string drain() {
        std::unique_lock<std::mutex> lk(mu);
        idle.wait(lk, [this] { return q.empty() && running == 0; });
        std::string e;
        e.swap(first_error);
        return e;
    }
    ~FileWriter() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto &t : th) t.join();
    }
};
// The per-contig host work of a call (small texts, read map) runs on threads that live as long as the context: spawning twenty threads per
// call costs about as much as the work they do.  run(n, fn): fn(worker, task) for every task in [0, n), the caller's thread included; returns
// when all are done.  One run at a time per pool (a context is used by one host thread).
struct WorkPool {
    std::mutex mu;
    std::condition_variable done_cv;
    std::vector<std::unique_ptr<std::condition_variable>> wake;      // one per worker: a run wakes the workers it has tasks for, not the whole pool (r5: fifteen sleepers woken
                                                                      // per run to find nothing cost a rank half a millisecond of CPU each and step)
    std::vector<std::thread> th;
    const std::function<void(int, int)> *fn = nullptr;
    std::atomic<int> next{0};
    int n = 0, active = 0, limit = 0;
    uint64_t gen = 0;
    bool stop = false;
    int size() const { return (int)th.size() + 1; }
    void start(int workers, int device) {
        for (int i = 0; i < workers; i++) wake.emplace_back(new std::condition_variable());
        for (int i = 0; i < workers; i++)
            th.emplace_back([this, i, device] {
                { char nm[16]; snprintf(nm, sizeof nm, "fzp-wk%d", i); (void)pthread_setname_np(pthread_self(), nm); }
                (void)hipSetDevice(device);                  // the tasks wait on events of the context's device
                uint64_t seen = 0;
                for (;;) {
                    const std::function<void(int, int)> *f;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        wake[(size_t)i]->wait(lk, [&] { return stop || gen != seen; });
                        if (stop) return;
                        seen = gen;
                        f = fn;
                        if (!f || i + 1 >= limit) continue;        // nothing left, or this run wants fewer threads than the pool has
                        active++;
                    }
                    for (int t; (t = next.fetch_add(1)) < n;) (*f)(i + 1, t);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (--active == 0) done_cv.notify_all();
                    }
                }
            });
    }
    void run(int n_tasks, const std::function<void(int, int)> &f, int max_threads) {
        int helpers;
        {
            std::lock_guard<std::mutex> lk(mu);
            fn = &f; n = n_tasks; limit = max_threads; next.store(0); gen++;
            helpers = std::min((int)th.size(), std::min(max_threads, n_tasks) - 1);      // the caller's thread takes tasks too
        }
        for (int i = 0; i < helpers; i++) wake[(size_t)i]->notify_one();
        for (int t; (t = next.fetch_add(1)) < n_tasks;) f(0, t);
        std::unique_lock<std::mutex> lk(mu);
        fn = nullptr;                                        // a worker that wakes up late finds nothing to do
        done_cv.wait(lk, [&] { return active == 0; });
    }
    ~WorkPool() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        for (auto &w : wake) w->notify_all();
        for (auto &t : th) t.join();
    }
};
static int writer_threads() {      // background file writers of a context: FZP_WRITER_THREADS, default = the rank's cores (2..16)
    if (const char *e = getenv("FZP_WRITER_THREADS")) { const int g = atoi(e); if (g >= 1 && g <= 64) return g; }
    return std::min(4, std::max(2, cores_per_rank()));      // (r5: 16 writers cost a rank 13 ms of CPU per step in wake-ups and page-cache contention, 4 cost 9.6 and the files are down as soon)
}
struct GroupPool;
static void group_pool_destroy(GroupPool *p);
void fzp_writer_destroy(fzp_ctx *ctx) {
    if (!ctx) return;
    group_pool_destroy(ctx->gpool);
    ctx->gpool = nullptr;
    if (ctx->writer) {
        (void)ctx->writer->drain();
        delete ctx->writer;
        ctx->writer = nullptr;
    }
    delete ctx->workers;
    ctx->workers = nullptr;
}

namespace {
using clk = std::chrono::steady_clock;
double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

struct Tok { const char *s; size_t n; };
inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; }
int split_ws(const char *l, size_t n, Tok *t, int maxt) {
    int k = 0;
    size_t i = 0;
    while (i < n) {
        while (i < n && is_ws(l[i])) i++;
        if (i >= n) break;
        size_t j = i;
        while (j < n && !is_ws(l[j])) j++;
        if (k < maxt) t[k] = {l + i, j - i};
        k++;
        i = j;
    }
    return k;
}
bool tok_int(Tok t, long long *v) {
    if (t.n == 0 || t.n > 19) return false;
    size_t i = 0;
    bool neg = false;
    if (t.s[0] == '-' || t.s[0] == '+') { neg = t.s[0] == '-'; i = 1; }
    if (i == t.n) return false;
    long long x = 0;
    for (; i < t.n; i++) {
        if (t.s[i] < '0' || t.s[i] > '9') return false;
        x = x * 10 + (t.s[i] - '0');
    }
    *v = neg ? -x : x;
    return true;
}
void split_nl(const char *s, size_t n, std::vector<Tok> &out) {   // text.split('\n')
    size_t off = 0;
    for (;;) {
        const char *e = (const char *)memchr(s + off, '\n', n - off);
        if (!e) { out.push_back({s + off, n - off}); break; }
        out.push_back({s + off, (size_t)(e - (s + off))});
        off = (size_t)(e - s) + 1;
    }
}

// ---- the three read_map tables of get_phasing_readmap (phasing_readmap.py:15-16,36), tokenised ONCE for all contigs
struct ReadMaps {
    std::vector<Tok> rid_to_oid, pid_to_fid;          // lines 17-18
    struct Row { Tok pid, rank; int nt; };
    std::vector<std::string> names;                   // distinct contig names of pread_to_contigs (column 1), sorted
    std::vector<std::vector<Row>> rows;               // rows per name, file order
    bool short_row = false;                           // a row with fewer than 2 tokens: the reference raises on it whatever the contig
};
void parse_maps(const fzp_pipe_opts *o, ReadMaps &m) {
    split_nl(o->rawread_ids ? o->rawread_ids : "", o->rawread_ids ? o->rr_len : 0, m.rid_to_oid);
    split_nl(o->pread_ids ? o->pread_ids : "", o->pread_ids ? o->pi_len : 0, m.pid_to_fid);
    std::unordered_map<std::string, int> idx;
    std::vector<std::pair<std::string, std::vector<ReadMaps::Row>>> tmp;
    const char *p2c = o->pread_to_contigs;
    size_t off = 0;
    int last = -1;
    while (off < o->pc_len) {
        const char *l = p2c + off;
        const char *e = (const char *)memchr(l, '\n', o->pc_len - off);
        size_t ln = e ? (size_t)(e - l) : o->pc_len - off;
        off += ln + (e ? 1 : 0);
        Tok t[4];
        int nt = split_ws(l, ln, t, 4);
        if (nt < 2) { m.short_row = true; continue; }
        // (a contig's rows stand together: the row before names the same contig nearly always -- no string, no hash for those; r6: the parse was 2 ms of CPU per bench step)
        if (last >= 0 && tmp[(size_t)last].first.size() == t[1].n && memcmp(tmp[(size_t)last].first.data(), t[1].s, t[1].n) == 0) {
            tmp[(size_t)last].second.push_back({t[0], nt >= 4 ? t[3] : Tok{nullptr, 0}, nt});
            continue;
        }
        std::string name(t[1].s, t[1].n);
        auto it = idx.find(name);
        if (it == idx.end()) { it = idx.emplace(name, (int)tmp.size()).first; tmp.push_back({name, {}}); }
        last = it->second;
        tmp[(size_t)last].second.push_back({t[0], nt >= 4 ? t[3] : Tok{nullptr, 0}, nt});
    }
    std::sort(tmp.begin(), tmp.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    for (auto &kv : tmp) { m.names.push_back(kv.first); m.rows.push_back(std::move(kv.second)); }
}

// the read maps are parsed on their own thread while the device works (2 ms for 40 000 reads); whoever needs them first waits
struct MapsHolder {
    mutable ReadMaps maps;
    std::shared_future<void> ready;
    bool have = false;
    // lazy (r6, fzp_job_phase_write): parsed by whoever asks first -- the thread that resolves the read-map rows, once K1's kernels are queued.  A thread of its own from the
    // call's first instruction on took a core from the launch thread exactly while it had K1's plan to launch (a rank with two cores: +0.3-0.9 ms of K1)
    const fzp_pipe_opts *lazy_opts = nullptr;
    mutable std::once_flag lazy_once;
    void start(const fzp_pipe_opts *o) {
        have = o->pread_to_contigs != nullptr;
        if (have) ready = std::async(std::launch::async, [this, o]() { parse_maps(o, maps); }).share();
    }
    void start_lazy(const fzp_pipe_opts *o) { have = o->pread_to_contigs != nullptr; lazy_opts = o; }
    const ReadMaps *get() const {
        if (!have) return nullptr;
        if (lazy_opts) std::call_once(lazy_once, [this]() { parse_maps(lazy_opts, maps); });
        else ready.wait();
        return &maps;
    }
};

struct TextBuf {
    std::string s;
    inline void puti(long long v, int min_digits = 0) {      // (min_digits: zero-padded like '%0Nd' of a non-negative number)
        static const char P2[] = "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
                                 "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
        char t[24];
        int k = 24;
        const bool neg = v < 0;
        unsigned long long u = neg ? 0ULL - (unsigned long long)v : (unsigned long long)v;
        while (u >> 32) { const unsigned d = (unsigned)(u % 100); u /= 100; t[--k] = P2[2 * d + 1]; t[--k] = P2[2 * d]; }
        uint32_t w = (uint32_t)u;
        while (w >= 100) { const uint32_t d = w % 100; w /= 100; t[--k] = P2[2 * d + 1]; t[--k] = P2[2 * d]; }
        if (w >= 10) { t[--k] = P2[2 * w + 1]; t[--k] = P2[2 * w]; } else t[--k] = (char)('0' + w);
        while (24 - k < min_digits && k > 1) t[--k] = '0';
        if (neg) s.push_back('-');
        s.append(t + k, (size_t)(24 - k));
    }
};

// get_phasing_readmap for one contig from records instead of text, in two halves.  readmap_rows: everything the contig's rows of pread_to_contigs decide on their
// own -- which rows count (name starts with the contig id, rank 0: phasing_readmap.py:41-43), the raw read behind each pread (lines 20-23, 44-45), later rows of a pread
// replacing earlier ones, the canonical output order -- resolved down to "the aligned read of this NAME, if there is one".  It needs the q_id -> name table only, so it
// runs on the host threads while the device is still phasing.  readmap_fill: the phased reads (ascending (q_id, block); the last row of a name wins, lines 29-33) ->
// rid_to_phase.<ctg> text + records.  Same semantics and messages as fzp_readmap.
struct ReadmapRows {
    std::vector<int32_t> name_of_q;              // per q_id: the id of its name among the contig's distinct read names
    int32_t n_names = 0;
    std::vector<long long> pid;                  // the rows that count, canonical order
    std::vector<int32_t> nid;                    // ... and the name id of each one's raw read (-1: no aligned read of that name)
};
inline uint64_t name_hash(const char *s, size_t n) {      // FNV-1a, folded
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= (uint8_t)s[i]; h *= 1099511628211ull; }
    return h ^ (h >> 29);
}
// (r6) The name table need not be the q_id table: readmap_fill only asks "the name id of q" and "the name id of a row's raw read", so the rows can be resolved against the names
// of ALL reads of the contig -- known before the device has aligned anything -- and `name_of_q` filled in from the q_id -> read table afterwards (job_phase_write does that: the
// 2.4 ms of CPU per bench step this costs used to fall into the 2 ms in which the phasing kernels are launched, and a rank with two cores felt it).
template <class NameOf>
int readmap_rows_t(const ReadMaps &m, const char *ctg_id, size_t nq, NameOf name_of, ReadmapRows &R, std::string &err) {
    if (m.short_row) { err = "pread_to_contigs: short row"; return FZP_EINVAL; }
    // the contig's distinct read names: an open-addressing table of q ids (no node per name: twenty of these run side by side)
    size_t cap = 16;
    while (cap < 2 * nq + 2) cap <<= 1;
    std::vector<int32_t> slot(cap, -1);                                    // -> the first q_id of the name
    std::vector<int32_t> id_of_q(nq, 0);
    auto find = [&](std::string_view nm) -> int32_t {                      // the slot of the name, or of the free place it would take
        size_t h = (size_t)name_hash(nm.data(), nm.size()) & (cap - 1);
        while (slot[h] >= 0 && name_of((size_t)slot[h]) != nm) h = (h + 1) & (cap - 1);
        return (int32_t)h;
    };
    R.name_of_q.resize(nq);
    int32_t n_names = 0;
    for (size_t q = 0; q < nq; q++) {
        const int32_t h = find(name_of(q));
        if (slot[(size_t)h] < 0) { slot[(size_t)h] = (int32_t)q; id_of_q[q] = n_names++; }
        R.name_of_q[q] = id_of_q[(size_t)slot[(size_t)h]];
    }
    R.n_names = n_names;
    const size_t cn = strlen(ctg_id);
    // names that start with ctg_id (startswith, line 41): a contiguous range of the sorted names
    auto lo = std::lower_bound(m.names.begin(), m.names.end(), std::string(ctg_id));
    struct Pick { size_t name, row; };
    std::vector<std::pair<const char *, Pick>> order;                     // file order matters for "later rows overwrite": the picked rows by address
    size_t n_match = 0;
    for (auto it = lo; it != m.names.end() && it->size() >= cn && memcmp(it->data(), ctg_id, cn) == 0; ++it, ++n_match) {
        const size_t ni = (size_t)(it - m.names.begin());
        for (size_t r = 0; r < m.rows[ni].size(); r++) order.push_back({m.rows[ni][r].pid.s, {ni, r}});
    }
    if (n_match > 1) std::sort(order.begin(), order.end(), [](const auto &a, const auto &b) { return a.first < b.first; });      // (one name: its rows are in file order already)
    struct Out { long long pid; uint32_t seq; int32_t nid; };
    std::vector<Out> out;
    out.reserve(order.size());
    for (auto &o : order) {
        const ReadMaps::Row &row = m.rows[o.second.name][o.second.row];
        long long rank, pid;
        if (row.nt < 4 || !tok_int(row.rank, &rank)) { err = "pread_to_contigs: bad rank"; return FZP_EINVAL; }
        if (rank != 0) continue;                                                      // line 43
        if (!tok_int(row.pid, &pid) || pid < 0 || (size_t)pid >= m.pid_to_fid.size() || pid > 0x7fffffffLL) { err = "pread_to_contigs: pread id out of range"; return FZP_EINVAL; }
        Tok fid = m.pid_to_fid[(size_t)pid];                                          // lines 20-23
        const char *s1 = (const char *)memchr(fid.s, '/', fid.n);
        if (!s1) { err = "pread_ids: '" + std::string(fid.s, fid.n) + "' has no '/'"; return FZP_EINVAL; }
        s1++;
        size_t rem = fid.n - (size_t)(s1 - fid.s);
        const char *s2 = (const char *)memchr(s1, '/', rem);
        Tok mid = {s1, s2 ? (size_t)(s2 - s1) : rem};
        long long raw;
        if (!tok_int(mid, &raw) || raw < 0) { err = "pread_ids: bad raw-read field"; return FZP_EINVAL; }
        raw /= 10;                                                                    // py2 int division
        if ((size_t)raw >= m.rid_to_oid.size()) { err = "rawread_ids: id " + std::to_string(raw) + " out of range"; return FZP_EINVAL; }
        Tok oid = m.rid_to_oid[(size_t)raw];
        const int32_t h = find(std::string_view(oid.s, oid.n));
        out.push_back({pid, (uint32_t)out.size(), slot[(size_t)h] < 0 ? -1 : id_of_q[(size_t)slot[(size_t)h]]});      // line 46: .get(oid, (-1, 0)) once the phases are known
    }
    // canonical order: ascending '%09d' string == ascending pread id below 10^9 (py2 dict order is unspecified); of a pread's rows the LAST one in the file counts
    std::sort(out.begin(), out.end(), [](const Out &a, const Out &b) {
        if (a.pid == b.pid) return a.seq < b.seq;
        if (a.pid < 1000000000LL && b.pid < 1000000000LL) return a.pid < b.pid;      // nine digits: string order == numeric order
        char ka[32], kb[32];
        snprintf(ka, sizeof ka, "%09lld", a.pid); snprintf(kb, sizeof kb, "%09lld", b.pid);
        return strcmp(ka, kb) < 0;
    });
    R.pid.clear(); R.nid.clear();
    R.pid.reserve(out.size()); R.nid.reserve(out.size());
    for (size_t i = 0; i < out.size(); i++)
        if (i + 1 == out.size() || out[i + 1].pid != out[i].pid) { R.pid.push_back(out[i].pid); R.nid.push_back(out[i].nid); }
    return FZP_OK;
}
int readmap_rows(const ReadMaps &m, const char *ctg_id, const std::vector<int64_t> &qoff, const std::string &qnames, ReadmapRows &R, std::string &err) {
    return readmap_rows_t(m, ctg_id, qoff.size() - 1, [&](size_t q) { return std::string_view(qnames.data() + qoff[q], (size_t)(qoff[q + 1] - qoff[q])); }, R, err);
}
// the records first (they are what the caller's gather needs), the text from them (lines 49-51) -- by a write task when the files are written in the background
void readmap_text(const fzp_r2p *recs, size_t n, const char *ctg_id, std::string &text) {
    const size_t cn = strlen(ctg_id);
    TextBuf b;
    b.s.reserve(n * (cn + 20));
    for (size_t i = 0; i < n; i++) {
        if (recs[i].arid >= 0) b.puti((long long)recs[i].arid, 9);      // '%09d' (phasing_readmap.py:49-51)
        else { char key[32]; snprintf(key, sizeof key, "%09lld", (long long)recs[i].arid); b.s += key; }
        b.s.push_back(' '); b.s.append(ctg_id, cn); b.s.push_back(' ');
        b.puti(recs[i].block); b.s.push_back(' '); b.puti(recs[i].phase); b.s.push_back('\n');
    }
    text.swap(b.s);
}
void readmap_fill(const ReadmapRows &R, const char *ctg_id, int32_t ctg_index, const fzp_pread *pr, int64_t n_pr, std::vector<fzp_r2p> &recs, std::string *text) {
    std::vector<std::pair<int, int>> val((size_t)R.n_names, {-1, 0});      // rid_to_phase by name (lines 29-33: the last line of a name wins; a name without one: (-1, 0), line 46)
    for (int64_t i = 0; i < n_pr; i++) val[(size_t)R.name_of_q[(size_t)pr[i].q_id]] = {pr[i].block, pr[i].phase};
    const size_t first = recs.size();
    for (size_t i = 0; i < R.pid.size(); i++) {
        const std::pair<int, int> v = R.nid[i] < 0 ? std::pair<int, int>{-1, 0} : val[(size_t)R.nid[i]];
        recs.push_back({(int32_t)R.pid[i], ctg_index, v.first, v.second});
    }
    if (text) readmap_text(recs.data() + first, recs.size() - first, ctg_id, *text);
}

bool mkdir_p(const std::string &path) {
    std::string cur;
    for (size_t i = 0; i <= path.size(); i++) {
        if (i == path.size() || path[i] == '/') {
            if (!cur.empty() && mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST) return false;
        }
        if (i < path.size()) cur.push_back(path[i]);
    }
    return true;
}
// A contig's files are made relative to ONE handle of its directory (mkdirat / openat): twenty writers that each resolved seven full paths -- and three mkdir -p walks
// from the root -- took turns on the locks of the ancestors they all share; now a contig touches its parent once (mkdir of its own directory) and after that only itself.
// measurement aid (FZP_PIPE_TIMING): CPU time of the write tasks by what they do, summed over the threads that ran them; printed and cleared by fzp_pipe_flush
std::atomic<int64_t> g_wt_fmt_ns{0}, g_wt_wait_ns{0}, g_wt_dir_ns{0}, g_wt_big_ns{0}, g_wt_small_ns{0}, g_wt_tasks{0};
inline int64_t thread_cpu_ns() { struct timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (int64_t)ts.tv_sec * 1000000000ll + ts.tv_nsec; }
struct DirWriter {
    int fd = -1;
    ~DirWriter() { if (fd >= 0) close(fd); }
    bool open_base(const std::string &base) {
        if (mkdir(base.c_str(), 0777) != 0 && errno != EEXIST && !mkdir_p(base)) return false;      // (the parent is made once per call, by the calling thread)
        fd = open(base.c_str(), O_RDONLY | O_DIRECTORY | O_CLOEXEC);
        return fd >= 0;
    }
    bool subdir(const char *name) { return mkdirat(fd, name, 0777) == 0 || errno == EEXIST; }
    // A file that is already there (a re-run into the same tree: what a restarted unzip job does, and what every bench step after the first does) is OVERWRITTEN IN PLACE and cut
    // to its new length afterwards instead of being truncated first: O_TRUNC hands every page of the old file back and the writes then allocate, clear and charge as many
    // new ones -- on a memory file system that was most of a writer's CPU time (r6: 13 ms per bench step for 39.5 MB in 140 files).  A fresh file costs what it did.
    // FZP_PIPE_TRUNC=1 restores the truncating open.
    static bool trunc_first() { static const bool t = [] { const char *e = getenv("FZP_PIPE_TRUNC"); return e && atoi(e) != 0; }(); return t; }
    static bool cut_to(int f, int64_t n) {
        if (trunc_first()) return true;
        struct stat sb;
        if (fstat(f, &sb) != 0) return false;
        return sb.st_size <= n || ftruncate(f, (off_t)n) == 0;
    }
    bool file(const char *rel, const char *data, size_t n, std::atomic<int64_t> &bytes) {
        const int f = openat(fd, rel, O_WRONLY | O_CREAT | O_CLOEXEC | (trunc_first() ? O_TRUNC : 0), 0666);
        if (f < 0) return false;
        size_t off = 0;
        while (off < n) {
            const ssize_t w = write(f, data + off, n - off);
            if (w < 0) { if (errno == EINTR) continue; const int e = errno; close(f); errno = e; return false; }
            off += (size_t)w;
        }
        if (!cut_to(f, (int64_t)n)) { const int e = errno; close(f); errno = e; return false; }
        close(f);
        bytes += (int64_t)n;
        return true;
    }
    // the same from pieces that lie where they lie (a contig's tigs in the pinned block, their headers beside them): writev, IOV_MAX pieces per call
    bool filev(const char *rel, std::vector<struct iovec> &iov, std::atomic<int64_t> &bytes) {
        const int f = openat(fd, rel, O_WRONLY | O_CREAT | O_CLOEXEC | (trunc_first() ? O_TRUNC : 0), 0666);
        if (f < 0) return false;
        size_t k = 0;
        int64_t total = 0;
        while (k < iov.size()) {
            const int n = (int)std::min<size_t>(iov.size() - k, 1024);
            const ssize_t w = writev(f, iov.data() + k, n);
            if (w < 0) { if (errno == EINTR) continue; const int e = errno; close(f); errno = e; return false; }
            total += w;
            size_t left = (size_t)w;      // (a short write: step over what went out)
            while (k < iov.size() && left >= iov[k].iov_len) { left -= iov[k].iov_len; k++; }
            if (left) { iov[k].iov_base = (char *)iov[k].iov_base + left; iov[k].iov_len -= left; }
        }
        if (!cut_to(f, total)) { const int e = errno; close(f); errno = e; return false; }
        close(f);
        bytes += total;
        return true;
    }
};
void add(fzp_pipe_out *a, const fzp_pipe_out &b) {
    a->n_reads += b.n_reads; a->n_aligned += b.n_aligned; a->n_rec += b.n_rec; a->n_sites += b.n_sites; a->n_rows += b.n_rows; a->n_arows += b.n_arows;
    a->n_pvars += b.n_pvars; a->n_preads += b.n_preads; a->n_groups += b.n_groups; a->bytes_written += b.bytes_written; a->dp_cells += b.dp_cells;
    a->ms_upload += b.ms_upload; a->ms_k1 += b.ms_k1; a->ms_phase += b.ms_phase; a->ms_results += b.ms_results; a->ms_text += b.ms_text;
}
}  // namespace

// test hook (tests/test_host_logic.py, CPU only): the read map of one contig the way fzp_job_phase_write makes it -- from phased-read RECORDS and the q_id name table, in its
// two halves (readmap_rows under the phasing kernels, readmap_fill behind them) -- so that it can be held against fzp_readmap, which works from the files' text.
extern "C" int fzp_debug_readmap_records(const char *rawread_ids, size_t rr_len, const char *pread_ids, size_t pi_len, const char *pread_to_contigs, size_t pc_len, const char *ctg_id,
                                         int32_t ctg_index, const fzp_pread *preads, int64_t n_preads, const int64_t *name_off, const char *names, int32_t n_q, fzp_r2p **recs,
                                         int64_t *n_recs, char **text, size_t *text_len) {
    if (!ctg_id || !name_off || !names || n_q < 0 || n_preads < 0 || (n_preads && !preads) || !recs || !n_recs || !text || !text_len) { fzp_set_error("fzp_debug_readmap_records: bad arguments"); return FZP_EINVAL; }
    fzp_pipe_opts o;
    fzp_pipe_opts_default(&o);
    o.rawread_ids = rawread_ids; o.rr_len = rr_len; o.pread_ids = pread_ids; o.pi_len = pi_len; o.pread_to_contigs = pread_to_contigs ? pread_to_contigs : ""; o.pc_len = pread_to_contigs ? pc_len : 0;
    ReadMaps m;
    parse_maps(&o, m);
    for (int64_t i = 0; i < n_preads; i++) if (preads[i].q_id < 0 || preads[i].q_id >= n_q) { fzp_set_error("fzp_debug_readmap_records: q_id out of range"); return FZP_EINVAL; }
    const std::vector<int64_t> qoff(name_off, name_off + n_q + 1);
    const std::string qn(names, (size_t)name_off[n_q]);
    ReadmapRows R;
    std::string err, txt;
    const int rc = readmap_rows(m, ctg_id, qoff, qn, R, err);
    if (rc != FZP_OK) { fzp_set_error("%s", err.c_str()); return rc; }
    std::vector<fzp_r2p> out;
    readmap_fill(R, ctg_id, ctg_index, preads, n_preads, out, &txt);
    *recs = (fzp_r2p *)malloc((out.size() ? out.size() : 1) * sizeof(fzp_r2p));
    *text = (char *)malloc(txt.size() + 1);
    if (!*recs || !*text) { free(*recs); free(*text); fzp_set_error("fzp_debug_readmap_records: host memory"); return FZP_ENOMEM; }
    if (!out.empty()) memcpy(*recs, out.data(), out.size() * sizeof(fzp_r2p));
    memcpy(*text, txt.data(), txt.size()); (*text)[txt.size()] = 0;
    *n_recs = (int64_t)out.size(); *text_len = txt.size();
    return FZP_OK;
}

extern "C" void fzp_pipe_opts_default(fzp_pipe_opts *o) {
    memset(o, 0, sizeof *o);
    fzp_align_params_default(&o->align);
}

// everything after the upload, for the contigs of one job
static int job_phase_write(fzp_ctx *ctx, fzp_alnjob *job, const fzp_names *nm, const fzp_pipe_opts *o, const MapsHolder &mh, const int32_t *ctg_index, fzp_pipe_out *out,
                           std::vector<fzp_r2p> &r2p) {
    FZP_TRY(fzp_bind(ctx));
    auto t0 = clk::now();
    const auto t_body = t0;
    // ---- (r6) while the device aligns: every contig's rows of pread_to_contigs resolved against the names of the contig's READS (readmap_rows_t's note).  The job knows which
    // read belongs to which contig; the names are the caller's.
    int32_t njc = 0;
    int64_t njr = 0;
    const int32_t *j_read_ctg = nullptr;
    fzp_align_host_reads(job, &njc, &njr, &j_read_ctg);
    int T = o->n_threads > 0 ? o->n_threads : std::min(64, std::max(2, cores_per_rank()));      // (the cores this rank may use, not the machine's: a rank of eight behind a 16-CPU quota has two)
    if (!ctx->workers || ctx->workers->size() < std::min(T, (int)njc)) {        // grown on demand, kept for the next call
        delete ctx->workers;
        ctx->workers = new WorkPool();
        ctx->workers->start(std::max(0, std::min(T, std::max((int)njc, 8)) - 1), ctx->device);
    }
    struct EarlyRows { ReadmapRows rows; int rc = FZP_OK; std::string err; };
    std::vector<EarlyRows> erows;
    std::vector<int32_t> local_of;                 // per read of the job: its place among its contig's reads
    std::vector<int64_t> rd_first;                 // per contig: where its reads begin in rd_list
    std::vector<int64_t> rd_list;
    const bool rows_early = mh.have && nm->names && nm->name_off && j_read_ctg && njc > 0;
    std::thread rows_thread;
    struct JoinRows { std::thread &t; ~JoinRows() { if (t.joinable()) t.join(); } } rows_join{rows_thread};      // (declared after everything the thread touches)
    if (o->flags & FZP_PIPE_REBUILD_INDEX) FZP_TRY(fzp_align_invalidate_index(job));
    FZP_TRY(fzp_align_run_deferred(ctx, job));      // (whether the fail list overflowed is asked by fzp_align_to_batch, in the fetch it makes anyway)
    // (K1's kernels are queued: this thread now sleeps until they are through -- the DP and the walk, 7 ms of the bench step --, which is when a rank with two cores has one to spare)
    if (rows_early) {
        rows_thread = std::thread([&, T]() {
            (void)pthread_setname_np(pthread_self(), "fzp-rows");
            // which reads a contig has, in input order (here and not on the calling thread: nothing of this may stand between a step's last copy and the next step's first kernel)
            rd_first.assign((size_t)njc + 1, 0);
            for (int64_t r = 0; r < njr; r++) if (j_read_ctg[r] >= 0 && j_read_ctg[r] < njc) rd_first[(size_t)j_read_ctg[r] + 1]++;
            for (int32_t c = 0; c < njc; c++) rd_first[(size_t)c + 1] += rd_first[(size_t)c];
            rd_list.resize((size_t)rd_first[(size_t)njc]);
            local_of.assign((size_t)njr, -1);
            {
                std::vector<int64_t> at(rd_first.begin(), rd_first.end() - 1);
                for (int64_t r = 0; r < njr; r++) if (j_read_ctg[r] >= 0 && j_read_ctg[r] < njc) { const int32_t c = j_read_ctg[r]; local_of[(size_t)r] = (int32_t)(at[(size_t)c] - rd_first[(size_t)c]); rd_list[(size_t)at[(size_t)c]++] = r; }
            }
            erows.resize((size_t)njc);
            const ReadMaps *mp = mh.get();
            if (!mp) return;
            const std::function<void(int, int)> w = [&](int, int c) {
                const int64_t *rl = rd_list.data() + rd_first[(size_t)c];
                erows[(size_t)c].rc = readmap_rows_t(*mp, nm->ctg_id[c], (size_t)(rd_first[(size_t)c + 1] - rd_first[(size_t)c]),
                                                     [&](size_t k) { const int64_t r = rl[k]; return std::string_view(nm->names + nm->name_off[r], (size_t)(nm->name_off[r + 1] - nm->name_off[r])); },
                                                     erows[(size_t)c].rows, erows[(size_t)c].err);
            };
            ctx->workers->run((int)njc, w, std::min(std::max(1, std::min(T, (int)njc)), 6));
        });
    }
    fzp_batch *b = nullptr;
    FZP_TRY(fzp_align_to_batch(ctx, job, &b));
    struct BG { fzp_ctx *c; fzp_batch *b; ~BG() { fzp_batch_destroy(c, b); } } bg{ctx, b};
    out->ms_k1 += ms_since(t0);
    t0 = clk::now();
    const int nc = b->n_ctg;
    static const bool timing = getenv("FZP_PIPE_TIMING") != nullptr;
    std::atomic<int64_t> us_names{0}, us_fmt{0}, us_map{0}, us_write{0};
    if (nc != njc) { fzp_set_error("fzp_job_phase_write: the batch has %d contigs, its job %d", nc, (int)njc); return FZP_EINVAL; }
    const int want_threads = std::max(1, std::min(T, nc));
    T = ctx->workers->size();
    // ---- what K1 alone decides goes to the host threads NOW, under the phasing kernels: the q_id -> read table comes over, and per contig the q_id names, q_id_map and the
    // read-map rows resolved down to read names (readmap_rows) are made while the device runs K2..K5
    const int64_t n_slots = b->h_slot_off.empty() ? 0 : b->h_slot_off.back();
    hipStream_t st2 = ctx->stream2;
    struct Ev { hipEvent_t e = nullptr; ~Ev() { if (e) (void)hipEventDestroy(e); } } ev_k1, ev_q;
    FZP_HIP(hipEventCreateWithFlags(&ev_k1.e, hipEventDisableTiming));
    FZP_HIP(hipEventCreateWithFlags(&ev_q.e, hipEventDisableTiming));
    struct PinQ { fzp_ctx *c; void *p = nullptr; ~PinQ() { if (p) fzp_pinned_release(c, p); } } pin_q{ctx};
    size_t pin_q_cap = 0;
    pin_q.p = fzp_pinned_acquire(ctx, (size_t)n_slots * 4 + 64, &pin_q_cap);
    if (!pin_q.p) { fzp_set_error("pinned host allocation failed"); return FZP_ENOMEM; }
    struct StreamGuard { hipStream_t s; bool armed = true; ~StreamGuard() { if (armed) (void)hipStreamSynchronize(s); } } sg2{st2};     // nothing below may leave while copies into pinned blocks are in flight
    FZP_HIP(hipEventRecord(ev_k1.e, ctx->stream));
    FZP_HIP(hipStreamWaitEvent(st2, ev_k1.e, 0));
    if (n_slots) FZP_HIP(hipMemcpyAsync(pin_q.p, b->qid_read.p, (size_t)n_slots * 4, hipMemcpyDeviceToHost, st2));
    FZP_HIP(hipEventRecord(ev_q.e, st2));
    const int32_t *qid_read = (const int32_t *)pin_q.p;
    struct PreCtg { std::vector<int64_t> qoff; std::string qn, qmap; ReadmapRows rows; int rc = FZP_OK; std::string err; };
    std::vector<PreCtg> pre((size_t)nc);
    const std::function<void(int, int)> pre_work = [&](int, int c) {
        PreCtg &P = pre[(size_t)c];
        auto tq = clk::now();
        // q_id table of the contig: aligned reads in (POS, read) order
        const int64_t nq = b->h_qid_off[(size_t)c + 1] - b->h_qid_off[(size_t)c];
        const int32_t *qr = qid_read + b->h_slot_off[(size_t)c];
        P.qoff.assign((size_t)nq + 1, 0);
        for (int64_t q = 0; q < nq; q++) {
            const int64_t r = qr[q];
            if (nm->names && nm->name_off) P.qn.append(nm->names + nm->name_off[r], (size_t)(nm->name_off[r + 1] - nm->name_off[r]));
            else { char tt[40]; snprintf(tt, sizeof tt, "read/%lld", (long long)r); P.qn += tt; }
            P.qoff[(size_t)q + 1] = (int64_t)P.qn.size();
        }
        us_names += (int64_t)(ms_since(tq) * 1e3); tq = clk::now();
        {
            TextBuf tb;
            tb.s.reserve(P.qn.size() + (size_t)nq * 12);
            for (int64_t q = 0; q < nq; q++) { tb.puti(q); tb.s.push_back(' '); tb.s.append(P.qn, (size_t)P.qoff[(size_t)q], (size_t)(P.qoff[(size_t)q + 1] - P.qoff[(size_t)q])); tb.s.push_back('\n'); }   // phasing.py:132-134
            P.qmap.swap(tb.s);
        }
        us_fmt += (int64_t)(ms_since(tq) * 1e3); tq = clk::now();
        if (rows_early) {                                   // the rows are resolved (above); what K1 adds is which read every q_id is
            EarlyRows &E = erows[(size_t)c];
            P.rc = E.rc; P.err = E.err;
            std::vector<int32_t> name_of_read;
            name_of_read.swap(E.rows.name_of_q);
            P.rows = std::move(E.rows);
            P.rows.name_of_q.resize((size_t)nq);
            for (int64_t q = 0; q < nq && P.rc == FZP_OK; q++) {
                const int32_t k = qr[q] >= 0 && qr[q] < njr ? local_of[(size_t)qr[q]] : -1;
                if (k < 0 || (size_t)k >= name_of_read.size() || j_read_ctg[qr[q]] != c) { P.rc = FZP_EINVAL; P.err = "q_id table names a read of another contig"; break; }
                P.rows.name_of_q[(size_t)q] = name_of_read[(size_t)k];
            }
        } else if (const ReadMaps *mp = mh.get()) P.rc = readmap_rows(*mp, nm->ctg_id[c], P.qoff, P.qn, P.rows, P.err);
        us_map += (int64_t)(ms_since(tq) * 1e3);
    };
    std::string early_err;
    std::thread early([&]() {
        (void)pthread_setname_np(pthread_self(), "fzp-early");
        if (hipSetDevice(ctx->device) != hipSuccess || hipEventSynchronize(ev_q.e) != hipSuccess) { (void)hipGetLastError(); early_err = "the q_id table did not arrive"; return; }
        if (rows_thread.joinable()) rows_thread.join();      // (one run at a time per pool; the rows have had all of K1 to get done)
        ctx->workers->run(nc, pre_work, std::min(want_threads, 6));      // (a few threads: there are milliseconds to do this in, and the thread that launches the kernels wants a core)
    });
    struct Join { std::thread &t; ~Join() { if (t.joinable()) t.join(); } } early_join{early};      // (declared after everything the thread touches)
    b->host_skip_rows = true;      // variant_map and atable leave as device-made text: their rows stay on the device
    FZP_TRY(fzp_batch_run(ctx, b, FZP_STAGE_ALL));
    out->ms_phase += ms_since(t0);
    t0 = clk::now();
    const bool want_cns = (o->flags & FZP_PIPE_CONSENSUS) != 0;
    std::vector<fzp_tig> tigs;      // K6 of every (block, phase) pile -> <ctg>/cns/phased_blocks.fa; the sequence bytes stay on the device until they join the texts' copy (below)
    DevBuf<uint8_t> d_cns_tmp;
    uint64_t n_cns = 0;
    if (want_cns) FZP_TRY(fzp_batch_consensus_dev(ctx, b, 3, tigs, d_cns_tmp, &n_cns));
    // the two big texts: serialised on the device, brought over while the records come.  (Serialising them right behind K3 and copying them under K4 / K5 was
    // measured: the phasing stage grew by more than the host section shrank -- that section is bound by its own formatting, not by these copies.)
    struct Owned {
        fzp_ctx *c; void *pin; void *rec_pin = nullptr; std::vector<char *> texts; std::mutex mu;
        std::vector<int64_t> site_begin, pvar_begin, pread_begin;      // per contig, into the batch-wide record arrays below
        const fzp_site *sites = nullptr; const fzp_pvar *pvars = nullptr; const fzp_pread *preads = nullptr;      // views into rec_pin (the batch's pinned result block, taken over)
        // the two device-made texts on their way into `pin`: the copy runs under whatever the device does next; a write task waits for `ev_text` before it touches them, and
        // the device blocks they come from stay out of the allocator's hands until the last task is done
        DevBuf<char> d_vmap, d_atab;
        DevBuf<uint8_t> d_cns;
        hipEvent_t ev_text = nullptr;
        int device = 0;
        bool texts_there() { return hipSetDevice(device) == hipSuccess && hipEventSynchronize(ev_text) == hipSuccess; }
        ~Owned() { if (ev_text) { (void)hipEventSynchronize(ev_text); (void)hipEventDestroy(ev_text); } fzp_pinned_release(c, pin); if (rec_pin) fzp_pinned_release(c, rec_pin); for (auto t : texts) free(t); }
    };
    std::shared_ptr<Owned> owned = std::make_shared<Owned>();
    owned->c = ctx; owned->pin = nullptr; owned->device = ctx->device;
    FZP_HIP(hipEventCreateWithFlags(&owned->ev_text, hipEventDisableTiming));
    // ---- what the caller waits for comes over FIRST: the block and read records go onto the main stream right behind K5; the kernels that serialise the two big texts run behind
    // them, and the texts' own copies (which only the file writers wait for) start when the records are here.  (r5: with the texts' 30 MB copied first the 1.6 MB of records
    // queued behind and beside them for 0.7 ms -- a device-to-host copy in flight holds up everything else that moves memory.)
    FZP_TRY(fzp_batch_result_begin(ctx, b));
    size_t n_vmap = 0, n_atab = 0;
    std::vector<int64_t> vb, ab;
    FZP_TRY(fzp_batch_texts_dev(ctx, b, owned->d_vmap, &n_vmap, vb, owned->d_atab, &n_atab, ab));      // (their sizes come back through ONE fetch on the same stream: the records are on the host when it returns; the kernels that write the texts are launched, not waited for)
    struct Ev2 { hipEvent_t e = nullptr; ~Ev2() { if (e) (void)hipEventDestroy(e); } } ev_put;
    FZP_HIP(hipEventCreateWithFlags(&ev_put.e, hipEventDisableTiming));
    FZP_HIP(hipEventRecord(ev_put.e, ctx->stream));
    fzp_result_all ra;
    FZP_TRY(fzp_batch_result_all(ctx, b, &ra));
    struct RG { fzp_result_all *r; ~RG() { fzp_result_all_free(r); } } rg{&ra};
    const ReadMaps *maps = mh.get();
    early.join();                                                // (the early half: long done)
    if (!early_err.empty()) { fzp_set_error("%s", early_err.c_str()); return FZP_EDEVICE; }
    // the rid_to_phase RECORDS (an array pass per contig) on the pool's threads while this one queues the texts' copies and makes the write tasks
    std::vector<std::vector<fzp_r2p>> recs((size_t)nc);
    {
        owned->site_begin.assign(ra.site_begin, ra.site_begin + nc + 1); owned->pvar_begin.assign(ra.pvar_begin, ra.pvar_begin + nc + 1); owned->pread_begin.assign(ra.pread_begin, ra.pread_begin + nc + 1);
        owned->sites = ra.all.sites; owned->pvars = ra.all.pvars; owned->preads = ra.all.preads;
        owned->rec_pin = b->pin; b->pin = nullptr;                // (fzp_batch_result_all's views point into it; the batch no longer gives it back)
    }
    for (int c = 0; c < nc && maps; c++) if (pre[(size_t)c].rc != FZP_OK) { fzp_set_error("%s", pre[(size_t)c].err.c_str()); return pre[(size_t)c].rc; }
    const std::function<void(int, int)> fill_work = [&](int, int c) {
        PreCtg &P = pre[(size_t)c];
        recs[(size_t)c].reserve(P.rows.pid.size());
        readmap_fill(P.rows, nm->ctg_id[c], ctg_index ? ctg_index[c] : c, owned->preads + owned->pread_begin[(size_t)c], owned->pread_begin[(size_t)c + 1] - owned->pread_begin[(size_t)c], recs[(size_t)c], nullptr);
    };
    // (a rank with two cores keeps the pass on this thread: its cores are busy with the previous call's write tasks, and a thread that has to
    // be woken there waits for a time slice -- measured on two cores: 22.5 ms per step with the hand-off, 1.34 x the unconstrained step instead of 1.12 x)
    const bool fill_beside = maps && cores_per_rank() > 2;
    std::thread filler;
    if (fill_beside) filler = std::thread([&]() { (void)pthread_setname_np(pthread_self(), "fzp-fill"); ctx->workers->run(nc, fill_work, std::min(want_threads, 4)); });
    struct JoinF { std::thread &t; ~JoinF() { if (t.joinable()) t.join(); } } filler_join{filler};
    size_t pin_cap = 0;
    const size_t o_atab = (n_vmap + 63) & ~(size_t)63, o_cns = o_atab + ((n_atab + 63) & ~(size_t)63), o_end = o_cns + (((size_t)n_cns + 63) & ~(size_t)63);
    char *pin = (char *)fzp_pinned_acquire(ctx, o_end + 64, &pin_cap);
    if (!pin) { fzp_set_error("pinned host allocation failed"); return FZP_ENOMEM; }
    owned->pin = pin;
    // the block (and the small texts) live until their last file is written: shared by the per-contig write tasks
    const bool async = (o->flags & FZP_PIPE_ASYNC_WRITES) != 0 && o->out_dir;
    if (async && !ctx->writer) { ctx->writer = new FileWriter(); ctx->writer->start(writer_threads()); }
    if (async) { const std::string e = [&] { std::lock_guard<std::mutex> lk(ctx->writer->mu); std::string x; x.swap(ctx->writer->first_error); return x; }(); if (!e.empty()) { fzp_set_error("%s", e.c_str()); return FZP_EINVAL; } }
    // The texts only have to be there when a contig's write task reaches them: their copies go behind the kernels that write them
    FZP_HIP(hipStreamWaitEvent(st2, ev_put.e, 0));
    if (n_vmap) FZP_HIP(hipMemcpyAsync(pin, owned->d_vmap.p, n_vmap, hipMemcpyDeviceToHost, st2));
    if (n_atab) FZP_HIP(hipMemcpyAsync(pin + o_atab, owned->d_atab.p, n_atab, hipMemcpyDeviceToHost, st2));
    if (n_cns) {      // the tigs' bases with them (r5: they used to come over into pageable memory inside the consensus call and were copied twice more into the contig's FASTA text, on this thread)
        std::swap(owned->d_cns.p, d_cns_tmp.p); std::swap(owned->d_cns.n, d_cns_tmp.n);
        FZP_HIP(hipMemcpyAsync(pin + o_cns, owned->d_cns.p, (size_t)n_cns, hipMemcpyDeviceToHost, st2));
    }
    FZP_HIP(hipEventRecord(owned->ev_text, st2));
    // the blasr task's BAM from the same pass: records come down here (device part), are split into '=' / 'X' and compressed by the contig's write task
    struct BamJob {
        fzp_alnset *aln = nullptr; std::vector<int32_t> flags; std::shared_ptr<std::vector<uint8_t>> ref;
        ~BamJob() { fzp_alnset_free(aln); }
    };
    const bool want_bam = (o->flags & FZP_PIPE_BAM) != 0 && o->out_dir, want_done = (o->flags & FZP_PIPE_SENTINELS) != 0 && o->out_dir;
    std::vector<std::shared_ptr<BamJob>> bams((size_t)nc);
    if (want_bam) for (int c = 0; c < nc; c++) {
        bams[(size_t)c] = std::make_shared<BamJob>();
        FZP_TRY(fzp_align_alnset_unsplit(ctx, job, c, nm->name_off, nm->names, &bams[(size_t)c]->aln, &bams[(size_t)c]->flags, &bams[(size_t)c]->ref));
    }
    out->ms_results += ms_since(t0);
    t0 = clk::now();
    // ---- per contig: the small files on host threads, all files written
    std::atomic<int64_t> bytes{0};
    const std::string out_dir = o->out_dir ? o->out_dir : "";
    if (o->out_dir && !mkdir_p(out_dir)) { fzp_set_error("cannot create %s: %s", out_dir.c_str(), strerror(errno)); return FZP_EIO; }      // once, here: the contigs' writers only make their own directories
    // The texts of the small files are made where the files are written: by the contig's write task -- on the background writers when FZP_PIPE_ASYNC_WRITES (under the next
    // call's kernels: r5, the host section at the end of a step was 1.1 ms of idle GPU on sixteen cores and 3.5 ms on two), on the pool otherwise.  The records those texts are
    // made from live in the batch's pinned result block, which the tasks have taken over.
    if (filler.joinable()) filler.join();
    else if (maps) for (int c = 0; c < nc; c++) fill_work(0, c);
    std::vector<std::function<bool()>> tasks((size_t)nc);     // per contig: make the small texts, write all files
    std::vector<std::shared_ptr<std::string>> whys((size_t)nc);   // why a contig's write task failed, recorded by the thread it failed on
    for (auto &w : whys) w = std::make_shared<std::string>();
    size_t tig_at = 0;      // (the tigs come in contig order)
    if (o->out_dir) for (int c = 0; c < nc; c++) {
        const char *ctg = nm->ctg_id[c];
        // the contig's tigs: the header lines are made here (a few dozen per contig), the bases are written from where the copy puts them
        struct CnsFa { std::vector<std::string> hdr; std::vector<const char *> seq; std::vector<size_t> len; size_t bytes = 0; };
        auto cns_p = std::make_shared<CnsFa>();
        if (want_cns) {
            char hdr[320];
            for (; tig_at < tigs.size() && tigs[tig_at].ctg < c; tig_at++) {}
            for (; tig_at < tigs.size() && tigs[tig_at].ctg == c; tig_at++) {
                const fzp_tig &g = tigs[tig_at];
                const int hl = snprintf(hdr, sizeof hdr, ">%s_%03d_%d %d %d %d\n", ctg, g.block, g.phase, g.lo + 1, g.hi + 1, g.n_records);      // (fzp_format_tigs' line)
                if (hl <= 0 || hl >= (int)sizeof hdr) { fzp_set_error("contig id too long"); return FZP_EINVAL; }
                cns_p->hdr.emplace_back(hdr, (size_t)hl); cns_p->seq.push_back(pin + o_cns + g.seq_off); cns_p->len.push_back((size_t)g.seq_len);
                cns_p->bytes += (size_t)hl + (size_t)g.seq_len + 1;
            }
        }
        const size_t fl = cns_p->bytes;
        const std::string base = out_dir + "/" + ctg, ctg_s = ctg;
        const char *pv = pin + vb[(size_t)c], *pa = pin + o_atab + ab[(size_t)c];
        const size_t lv = (size_t)(vb[(size_t)c + 1] - vb[(size_t)c]), la = (size_t)(ab[(size_t)c + 1] - ab[(size_t)c]);
        auto pre_p = std::make_shared<PreCtg>(std::move(pre[(size_t)c]));                    // q_id names, q_id_map
        const bool have_r2p = maps != nullptr;
        auto recs_p = std::make_shared<std::vector<fzp_r2p>>(have_r2p ? recs[(size_t)c] : std::vector<fzp_r2p>());
        std::atomic<int64_t> *bytes_p = &bytes;
        FileWriter *fw = async ? ctx->writer : nullptr;
        std::shared_ptr<BamJob> bam = bams[(size_t)c];
        std::shared_ptr<std::string> why = whys[(size_t)c];
        const int64_t known = (int64_t)(lv + la + pre_p->qmap.size() + (want_cns ? fl : 0));
        if (async) bytes += known;      // (what is known of the queued task's bytes when the call returns: the small texts do not exist yet -- the rest goes to the writer's `extra`)
        tasks[(size_t)c] = [owned, c, base, ctg_s, pv, pa, lv, la, pre_p, recs_p, have_r2p, want_cns, cns_p, bytes_p, fw, bam, want_done, why, known]() -> bool {
            std::atomic<int64_t> local{0};
            std::atomic<int64_t> &bt = fw ? local : *bytes_p;
            struct Extra { FileWriter *w; std::atomic<int64_t> &l; int64_t k; ~Extra() { if (w) w->extra += l.load() - k; } } extra_guard{fw, local, known};
            const std::string aln_done = "blasr/aln_" + ctg_s + "_done", p_done = "phasing/p_" + ctg_s + "_done";
            auto failed = [&](const std::string &what) { *why = what; if (fw) fw->fail(what); return false; };      // the cause is taken where it happens, on this thread
            // the small texts (phasing.py:124, 412-421, 478-480; phasing_readmap.py:49-51)
            const int64_t s0 = owned->site_begin[(size_t)c], s1 = owned->site_begin[(size_t)c + 1], p0 = owned->pvar_begin[(size_t)c], p1 = owned->pvar_begin[(size_t)c + 1],
                          r0 = owned->pread_begin[(size_t)c], r1 = owned->pread_begin[(size_t)c + 1];
            char *txt[3] = {nullptr, nullptr, nullptr};
            size_t len[3] = {0, 0, 0};
            struct FreeTxt { char **t; ~FreeTxt() { for (int i = 0; i < 3; i++) free(t[i]); } } free_txt{txt};
            const int64_t nq = (int64_t)pre_p->qoff.size() - 1;
            int64_t tc0 = thread_cpu_ns(), tc1;
            auto lap = [&](std::atomic<int64_t> &acc) { tc1 = thread_cpu_ns(); acc += tc1 - tc0; tc0 = tc1; };
            int rc = fzp_format_variant_pos(owned->sites + s0, s1 - s0, &txt[0], &len[0]);
            if (rc == FZP_OK) rc = fzp_format_phased_variants(owned->sites, owned->pvars + p0, p1 - p0, &txt[1], &len[1]);       // pvars carry batch-wide site indices
            if (rc == FZP_OK) rc = fzp_format_phased_reads(owned->preads + r0, r1 - r0, ctg_s.c_str(), pre_p->qoff.data(), pre_p->qn.data(), (int32_t)nq, &txt[2], &len[2]);
            if (rc != FZP_OK) return failed(fzp_last_error());
            lap(g_wt_fmt_ns);
            if (!owned->texts_there()) { (void)hipGetLastError(); return failed("the device-made texts of " + ctg_s + " did not arrive"); }
            lap(g_wt_wait_ns);
            std::string r2p_text;
            if (have_r2p) readmap_text(recs_p->data(), recs_p->size(), ctg_s.c_str(), r2p_text);
            lap(g_wt_fmt_ns);
            DirWriter dw;
            if (!dw.open_base(base)) return failed("cannot create " + base + ": " + strerror(errno));
            if (bam) {                                           // <ctg>_sorted.bam + index, then the blasr task's sentinels
                uint8_t *bb = nullptr, *bi = nullptr;
                size_t bl = 0, il = 0;
                bool okb = dw.subdir("blasr");
                std::string cause = okb ? "" : std::string("cannot create ") + base + "/blasr: " + strerror(errno);
                if (okb) {
                    fzp_alnset_split_eqx(bam->aln, bam->ref->data());
                    if (fzp_format_bam(bam->aln, ctg_s.c_str(), (int64_t)bam->ref->size(), bam->flags.data(), &bb, &bl, &bi, &il) != FZP_OK) { okb = false; cause = fzp_last_error(); }
                }
                if (okb && !(dw.file(("blasr/" + ctg_s + "_sorted.bam").c_str(), (const char *)bb, bl, bt) && dw.file(("blasr/" + ctg_s + "_sorted.bam.bai").c_str(), (const char *)bi, il, bt))) {
                    okb = false; cause = std::string("cannot write under ") + base + "/blasr: " + strerror(errno);
                }
                free(bb); free(bi);
                if (want_done) { if (okb) (void)dw.file(aln_done.c_str(), "", 0, bt); (void)dw.file((aln_done + ".exit").c_str(), "", 0, bt); }
                if (!okb) {
                    if (want_done) { (void)dw.subdir("phasing"); (void)dw.file((p_done + ".exit").c_str(), "", 0, bt); }      // the phasing task never ran
                    return failed(cause);
                }
            }
            bool ok = dw.subdir("het_call") && dw.subdir("g_atable") && dw.subdir("get_phased_blocks");
            lap(g_wt_dir_ns);
            ok = ok && dw.file("het_call/variant_map", pv, lv, bt) && dw.file("g_atable/atable", pa, la, bt);
            lap(g_wt_big_ns);
            ok = ok && dw.file("het_call/variant_pos", txt[0], len[0], bt) &&
                 dw.file("het_call/q_id_map", pre_p->qmap.data(), pre_p->qmap.size(), bt) &&
                 dw.file("get_phased_blocks/phased_variants", txt[1], len[1], bt) && dw.file("phased_reads", txt[2], len[2], bt);
            if (ok && have_r2p) ok = dw.file(("rid_to_phase." + ctg_s).c_str(), r2p_text.data(), r2p_text.size(), bt);
            lap(g_wt_small_ns);
            g_wt_tasks++;
            if (ok && want_cns) {
                std::vector<struct iovec> iov;
                iov.reserve(3 * cns_p->hdr.size());
                static const char nl = '\n';
                for (size_t k = 0; k < cns_p->hdr.size(); k++) {
                    iov.push_back({(void *)cns_p->hdr[k].data(), cns_p->hdr[k].size()}); iov.push_back({(void *)cns_p->seq[k], cns_p->len[k]}); iov.push_back({(void *)&nl, 1});
                }
                ok = dw.subdir("cns") && dw.filev("cns/phased_blocks.fa", iov, bt);
            }
            const std::string cause = ok ? "" : "cannot write under " + base + ": " + strerror(errno);
            if (want_done && dw.subdir("phasing")) { if (ok) (void)dw.file(p_done.c_str(), "", 0, bt); (void)dw.file((p_done + ".exit").c_str(), "", 0, bt); }
            return ok ? true : failed(cause);
        };
    }
    if (o->out_dir) {
        if (async) { for (int c = 0; c < nc; c++) if (tasks[(size_t)c]) ctx->writer->push([tk = std::move(tasks[(size_t)c])]() { (void)tk(); }); }
        else {
            std::atomic<int> failed{-1};
            const std::function<void(int, int)> wr = [&](int, int c) { if (tasks[(size_t)c] && !tasks[(size_t)c]()) failed.store(c); };
            ctx->workers->run(nc, wr, want_threads);
            if (failed.load() >= 0) { fzp_set_error("%s", whys[(size_t)failed.load()]->c_str()); return FZP_EIO; }
        }
    }
    { size_t tot = r2p.size(); for (int c = 0; c < nc; c++) tot += recs[(size_t)c].size(); r2p.reserve(tot); }      // (one block, not a doubling series of them)
    for (int c = 0; c < nc; c++) r2p.insert(r2p.end(), recs[(size_t)c].begin(), recs[(size_t)c].end());
    out->ms_text += ms_since(t0);
    if (timing) fprintf(stderr, "[fzp_pipe] host section %.2f ms on %d threads; summed over contigs: names %.2f fmt %.2f readmap %.2f write %.2f ms\n", ms_since(t0), T,
                        us_names.load() / 1e3, us_fmt.load() / 1e3, us_map.load() / 1e3, us_write.load() / 1e3);
    out->bytes_written += bytes.load();
    out->n_groups += 1;
    out->n_rec += b->n_rec; out->n_sites += b->n_sites; out->n_rows += b->n_rows; out->n_arows += b->n_arows; out->n_pvars += b->n_pvars; out->n_preads += b->n_preads;
    out->n_aligned += b->n_qid;
    if (timing) fprintf(stderr, "[fzp_pipe] body done at %.2f ms\n", ms_since(t_body));
    sg2.armed = false;      // what is still in flight on stream2 -- the two big texts -- lands in a block the write tasks own, and they wait for it themselves
    return FZP_OK;
}

extern "C" int fzp_job_phase_write(fzp_ctx *ctx, fzp_alnjob *job, const fzp_names *nm, const fzp_pipe_opts *opts, fzp_pipe_out *out) {
    if (!ctx || !job || !nm || !nm->ctg_id || !out) { fzp_set_error("fzp_job_phase_write: bad arguments"); return FZP_EINVAL; }
    fzp_pipe_opts o;
    if (opts) o = *opts; else fzp_pipe_opts_default(&o);
    memset(out, 0, sizeof *out);
    const auto t_call = clk::now();
    MapsHolder mh;
    mh.start_lazy(&o);
    std::vector<fzp_r2p> r2p;
    FZP_TRY(job_phase_write(ctx, job, nm, &o, mh, o.ctg_index, out, r2p));
    if (getenv("FZP_PIPE_TIMING")) fprintf(stderr, "[fzp_job_phase_write] %.2f ms in the call: k1 %.2f phase %.2f results %.2f text %.2f\n", ms_since(t_call), out->ms_k1, out->ms_phase,
                                           out->ms_results, out->ms_text);
    out->n_r2p = (int64_t)r2p.size();
    out->r2p = (fzp_r2p *)malloc((r2p.size() ? r2p.size() : 1) * sizeof(fzp_r2p));
    if (!out->r2p) return FZP_ENOMEM;
    if (!r2p.empty()) memcpy(out->r2p, r2p.data(), r2p.size() * sizeof(fzp_r2p));
    return FZP_OK;
}

extern "C" int fzp_phase_contigs(fzp_ctx *ctx, int32_t n_ctg, const uint8_t *const *ctg_seq, const int64_t *ctg_len, int64_t n_reads, const int32_t *read_ctg,
                                 const int64_t *read_off, const uint8_t *read_seq, const fzp_names *nm, const fzp_pipe_opts *opts, fzp_pipe_out *out) {
    if (!ctx || n_ctg <= 0 || !ctg_seq || !ctg_len || n_reads < 0 || (n_reads && (!read_ctg || !read_off || !read_seq)) || !nm || !nm->ctg_id || !out) {
        fzp_set_error("fzp_phase_contigs: bad arguments");
        return FZP_EINVAL;
    }
    fzp_pipe_opts o;
    if (opts) o = *opts; else fzp_pipe_opts_default(&o);
    if (!getenv("FZP_PIPE_SYNC_WRITES")) o.flags |= FZP_PIPE_ASYNC_WRITES;      // (this call flushes before it returns: a group's files go down under the next group's kernels whether the caller asked or not)
    if (ctx->writer) (void)ctx->writer->extra.exchange(0);                       // (what earlier fzp_job_phase_write calls left there is not this call's)
    for (auto l : ctx->lanes) if (l->writer) (void)l->writer->extra.exchange(0);
    memset(out, 0, sizeof *out);
    for (int64_t r = 0; r < n_reads; r++) if (read_ctg[r] < 0 || read_ctg[r] >= n_ctg) { fzp_set_error("read %lld: bad contig", (long long)r); return FZP_EINVAL; }
    const bool timing = getenv("FZP_PIPE_TIMING") != nullptr;
    const auto t_call = clk::now();
    MapsHolder mh;
    mh.start(&o);
    const double ms_maps = ms_since(t_call);
    // reads of every contig (input order inside a contig), read bases per contig
    std::vector<std::vector<int64_t>> ctg_reads((size_t)n_ctg);
    std::vector<int64_t> bases((size_t)n_ctg, 0);
    for (int64_t r = 0; r < n_reads; r++) { ctg_reads[(size_t)read_ctg[r]].push_back(r); bases[(size_t)read_ctg[r]] += read_off[r + 1] - read_off[r]; }
    // groups of consecutive contigs: trace-back masks cost ~18 B per read base (8 B per DP step, ~2.25 steps per base), everything else of K1 about as much again
    int64_t group_bases = o.group_bases;
    if (group_bases <= 0) {
        size_t fr = 0, tot = 0;
        FZP_TRY(fzp_bind(ctx));
        FZP_HIP(hipMemGetInfo(&fr, &tot));
        const int lanes = o.n_lanes > 0 ? o.n_lanes : 2;
        int64_t all = 0;
        for (auto v : bases) all += v;
        // as large as the device allows (long launches, few tails), but at least two groups per lane so that uploads and file
        // writes of one group hide behind the kernels of another
        group_bases = std::min<int64_t>((int64_t)((double)tot * 0.55 / lanes / 40.0), std::max<int64_t>(64ll << 20, all / (2 * lanes)));
    }
    {   // a contig is never split over groups (its reads are chunked inside K1 so that the trace-back masks fit); what must fit at once is
        // the rest of its per-base state: ASCII + two packed copies + CIGAR room + hit lists ~ 8 B per read base
        size_t fr = 0, tot = 0;
        FZP_TRY(fzp_bind(ctx));
        FZP_HIP(hipMemGetInfo(&fr, &tot));
        for (int c = 0; c < n_ctg; c++)
            if ((double)bases[(size_t)c] * 8.0 + (double)ctg_len[c] * 12.0 > 0.8 * (double)tot) {
                fzp_set_error("fzp_phase_contigs: contig %s (%lld bp, %lld read bases) does not fit the device (%.0f GB) as one group", nm->ctg_id[c], (long long)ctg_len[c],
                              (long long)bases[(size_t)c], (double)tot / 1e9);
                return FZP_ENOMEM;
            }
    }
    struct Group { int c0, c1; };
    std::vector<Group> groups;
    for (int c = 0; c < n_ctg;) {
        int e = c;
        int64_t acc = 0;
        while (e < n_ctg && (e == c || acc + bases[(size_t)e] <= group_bases)) { acc += bases[(size_t)e]; e++; }
        groups.push_back({c, e});
        c = e;
    }
    int lanes = o.n_lanes > 0 ? o.n_lanes : 2;
    lanes = std::min<int>(lanes, (int)groups.size());
    std::atomic<size_t> next{0};
    std::vector<int> rcs((size_t)lanes, FZP_OK);
    std::vector<std::string> errs((size_t)lanes);
    std::vector<fzp_pipe_out> outs((size_t)lanes);
    std::vector<std::vector<std::vector<fzp_r2p>>> r2p_g(1);
    r2p_g[0].resize(groups.size());
    const int device = ctx->device;
    while ((int)ctx->lanes.size() < lanes - 1) {           // the extra lanes' contexts live as long as `ctx` (warm caches on the next call)
        fzp_ctx *lc = nullptr;
        FZP_TRY(fzp_ctx_create(device, 0, &lc));
        ctx->lanes.push_back(lc);
    }
    std::mutex up_mu;
    auto lane = [&](int li) {
        fzp_ctx *lc = li == 0 ? ctx : ctx->lanes[(size_t)li - 1];
        if (fzp_bind(lc) != FZP_OK) { rcs[(size_t)li] = FZP_EDEVICE; errs[(size_t)li] = fzp_last_error(); return; }
        fzp_pipe_out &po = outs[(size_t)li];
        memset(&po, 0, sizeof po);
        for (;;) {
            const size_t g = next.fetch_add(1);
            if (g >= groups.size()) break;
            const Group &G = groups[g];
            // the group's inputs: contig pointers as they are, reads gathered per contig (pointers into the caller's blob where they are contiguous)
            const int gc = G.c1 - G.c0;
            std::vector<int64_t> r_idx;
            for (int c = G.c0; c < G.c1; c++) r_idx.insert(r_idx.end(), ctg_reads[(size_t)c].begin(), ctg_reads[(size_t)c].end());
            const int64_t gr = (int64_t)r_idx.size();
            std::vector<int32_t> g_ctg((size_t)gr);
            std::vector<int64_t> g_off((size_t)gr + 1, 0), g_noff((size_t)gr + 1, 0);
            bool contiguous = true;
            for (int64_t k = 0; k < gr; k++) {
                const int64_t r = r_idx[(size_t)k];
                g_ctg[(size_t)k] = read_ctg[r] - G.c0;
                g_off[(size_t)k + 1] = g_off[(size_t)k] + (read_off[r + 1] - read_off[r]);
                if (k && r != r_idx[(size_t)k - 1] + 1) contiguous = false;
            }
            auto t0 = clk::now();
            std::vector<uint8_t> gathered;
            const uint8_t *g_seq = read_seq;
            std::vector<int64_t> abs_off;
            if (contiguous && gr) { abs_off.resize((size_t)gr + 1); for (int64_t k = 0; k <= gr; k++) abs_off[(size_t)k] = read_off[r_idx[0]] + g_off[(size_t)k]; }
            else if (gr) {
                gathered.resize((size_t)g_off[(size_t)gr]);
                for (int64_t k = 0; k < gr; k++) memcpy(gathered.data() + g_off[(size_t)k], read_seq + read_off[r_idx[(size_t)k]], (size_t)(g_off[(size_t)k + 1] - g_off[(size_t)k]));
                g_seq = gathered.data();
                abs_off = g_off;
            } else abs_off.assign(1, 0);
            std::string g_names;
            for (int64_t k = 0; k < gr; k++) {        // names by the caller's read index ("read/<index>" when none were given)
                const int64_t r = r_idx[(size_t)k];
                if (nm->names && nm->name_off) g_names.append(nm->names + nm->name_off[r], (size_t)(nm->name_off[r + 1] - nm->name_off[r]));
                else { char tt[40]; snprintf(tt, sizeof tt, "read/%lld", (long long)r); g_names += tt; }
                g_noff[(size_t)k + 1] = (int64_t)g_names.size();
            }
            fzp_alnjob *job = nullptr;
            int rc;
            {   // one upload at a time: the first group then has the whole PCIe link and its kernels start early; the next group's upload
                // runs under them (two uploads side by side would finish together and leave the device idle until then)
                std::lock_guard<std::mutex> lk(up_mu);
                rc = fzp_align_create(lc, gc, ctg_seq + G.c0, ctg_len + G.c0, gr, g_ctg.data(), abs_off.data(), g_seq, &o.align, &job);
            }
            po.ms_upload += ms_since(t0);
            if (rc == FZP_OK) {
                fzp_names gn;
                gn.n_ctg = gc; gn.ctg_id = nm->ctg_id + G.c0;
                gn.name_off = g_noff.data();
                gn.names = g_names.data();
                std::vector<int32_t> gi;
                for (int c = G.c0; c < G.c1; c++) gi.push_back(o.ctg_index ? o.ctg_index[c] : c);
                rc = job_phase_write(lc, job, &gn, &o, mh, gi.data(), &po, r2p_g[0][g]);
                if (rc == FZP_OK) {
                    std::vector<fzp_aln_summary> sm((size_t)gr);
                    if (gr && fzp_align_summaries(lc, job, sm.data()) == FZP_OK) for (auto &s : sm) po.dp_cells += (double)s.cells;
                }
            }
            if (rc != FZP_OK) errs[(size_t)li] = fzp_last_error();
            fzp_align_destroy(lc, job);
            po.n_reads += gr;
            if (rc != FZP_OK) { rcs[(size_t)li] = rc; break; }
        }
    };
    const double ms_prep = ms_since(t_call);
    {
        std::vector<std::thread> th;
        for (int li = 1; li < lanes; li++) th.emplace_back(lane, li);
        lane(0);
        for (auto &x : th) x.join();
    }
    const double ms_lanes = ms_since(t_call);
    (void)fzp_bind(ctx);
    const int frc = fzp_pipe_flush(ctx);                    // the groups' files overlapped other groups' kernels; all are down now
    int64_t late_bytes = 0;
    if (ctx->writer) late_bytes += ctx->writer->extra.exchange(0);
    for (auto l : ctx->lanes) if (l->writer) late_bytes += l->writer->extra.exchange(0);
    if (timing) fprintf(stderr, "[fzp_phase_contigs] read maps started by %.2f ms, grouping until %.2f, lanes until %.2f, flush until %.2f (%d lanes, %zu groups)\n", ms_maps, ms_prep, ms_lanes,
                        ms_since(t_call), lanes, groups.size());
    for (int li = 0; li < lanes; li++) if (rcs[(size_t)li] != FZP_OK) { fzp_set_error("%s", errs[(size_t)li].c_str()); return rcs[(size_t)li]; }
    if (frc != FZP_OK) return frc;
    for (int li = 0; li < lanes; li++) add(out, outs[(size_t)li]);
    out->bytes_written += late_bytes;
    std::vector<fzp_r2p> all;
    for (auto &v : r2p_g[0]) all.insert(all.end(), v.begin(), v.end());
    out->n_r2p = (int64_t)all.size();
    out->r2p = (fzp_r2p *)malloc((all.size() ? all.size() : 1) * sizeof(fzp_r2p));
    if (!out->r2p) return FZP_ENOMEM;
    if (!all.empty()) memcpy(out->r2p, all.data(), all.size() * sizeof(fzp_r2p));
    return FZP_OK;
}

// ---- inputs from the reference's own files: <reads_dir>/<ctg>_ref.fa, <ctg>_reads.fa (unzip.py:204,233-234; read at phasing.py:489-494 and by blasr).
// A group's files are mapped and cut into pieces of a few megabytes at record starts; host threads scan the pieces (memchr) for their records, a prefix sum gives every
// record its place in the group's buffers, the same threads copy the sequences there -- one copy, no per-file intermediates.  Groups are parsed one AHEAD of the lanes
// that align them, and their buffers are kept with the context between calls, so what is in host memory at any time is the groups in flight -- not the rank's reads.
// a grow-only byte buffer that is neither zeroed nor copied when it grows: its pages are first touched by the threads that fill it (a std::vector's resize would
// fill 1 GB from one thread -- 250 ms -- before sixteen threads overwrite it in 15)
struct RawBuf {
    uint8_t *p = nullptr;
    size_t cap = 0;
    RawBuf() = default;
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    ~RawBuf() { free(p); }
    bool need(size_t n) {
        if (n <= cap && p) return true;
        free(p);
        const size_t want = ((n + n / 8 + (2u << 20)) + ((2u << 20) - 1)) & ~(size_t)((2u << 20) - 1);      // some slack: the next group is rarely smaller
        p = (uint8_t *)aligned_alloc(2u << 20, want);
        cap = p ? want : 0;
        if (p) (void)madvise(p, want, MADV_HUGEPAGE);
        return p != nullptr;
    }
    uint8_t *data() { return p; }
};
constexpr int FA_UP_STREAMS = 4;               // upload streams of the files loader: one stream keeps one copy engine busy (~35 GB/s here), the pieces are dealt over several
struct GroupIn {
    RawBuf raw;                                  // the group's files as they are: the reads files first, contig after contig, then the contig files
    RawBuf ref;                                  // contigs whose FASTA record is not one plain line, joined (normally empty: the others are used where they lie in `raw`)
    std::vector<const uint8_t *> ctg_ptr;        // [gc] every contig's bases (in raw or in ref)
    std::vector<int64_t> ctg_len;                // [gc]
    RawBuf blob;                                 // the group's reads joined, contig after contig -- only when some read's record is not one plain line
    const uint8_t *seq_base = nullptr;           // raw or blob: what the spans below index
    std::vector<int64_t> be;                     // [2 n_reads] read r = seq_base[be[2r], be[2r + 1])
    std::vector<int64_t> noff;
    std::vector<int32_t> read_ctg;
    RawBuf names;
    bool in_place = false;                       // the reads are spans of `raw`
    // r6, the default: the files as they are in ONE device block (a '\n' behind every file), records found on the device (fzp_fasta.hip); the host keeps no copy
    int64_t pin_bytes = 0;                       // the block's bytes
    std::vector<int64_t> foff;                   // [2 gc + 1] file t (reads files first, then the contig files) at d_raw + foff[t]
    bool dev_parse = false;
    double ms_read = 0;
    // ... and on their way to the device while the rest is still being read: every 4 MB piece is handed to the DMA engine by the thread that read it (upload stream of the
    // pool), the lane waits for ev_up on its own stream
    DevBuf<uint8_t> d_raw;
    hipEvent_t ev_up[FA_UP_STREAMS] = {};
    bool streamed = false;
    int n_up = 0;
    int rc = FZP_OK;
    std::string err;
    ~GroupIn() { for (auto e : ev_up) if (e) (void)hipEventDestroy(e); }
};
struct GroupPool {
    hipStream_t up[FA_UP_STREAMS] = {};          // the loader's upload streams (pieces of a group's files, pinned -> device)
    int n_up = 0;
    hipEvent_t prev[FA_UP_STREAMS] = {};         // where the previous group's copies end on every stream: a group's copies start behind ALL of them (group order on the link)
    bool have_prev = false;
    // The pieces' way to the device: a ring of pinned buffers of one piece each, taken in the order the pieces are handed out (a buffer is free again when the copy out of
    // it is through; 64 of them: the readers, a little faster than the link, run up to 256 MB ahead before one of them has to wait -- two buffers per reader had every
    // reader waiting for its own last copy, 0.3 ms of spinning each time: the bench's from-files leg 31 -> 35 ms, its host CPU 150 -> 225 ms).
    // r6 first pinned a group's files WHOLE (1.2-1.6 GB a group at genome scale): hipHostMalloc takes ~0.2 s per GB, stalls the other lane's launches while it runs, and the
    // pool of such blocks took three calls to settle (configs[4] from files: 1.0 / 0.41 / 0.42 / 0.24 s).  The host needs the bytes for nothing: names come back from the device.
    fzp_ctx *stage_ctx = nullptr;
    std::vector<uint8_t *> stage;
    std::vector<hipEvent_t> stage_ev;
    std::vector<char> stage_used;
    uint8_t *nl_line = nullptr;                               // 64 pinned '\n'
    std::unique_ptr<std::atomic<int64_t>[]> stage_turn;      // per buffer: the ticket that may fill it next (the one before has queued its copy)
    int64_t stage_ticket = 0;                                 // tickets handed out so far (a group takes a run of them)
    size_t stage_bytes = 0;
    int ensure_stage(fzp_ctx *ctx, int n_bufs, size_t bytes) {
        if (stage_ctx == ctx && (int)stage.size() >= n_bufs && stage_bytes >= bytes) return FZP_OK;
        for (int u = 0; u < n_up; u++) (void)hipStreamSynchronize(up[u]);      // (nothing may still be on its way out of a buffer that goes back)
        drop_stage();
        stage_ctx = ctx; stage_bytes = bytes;
        nl_line = (uint8_t *)fzp_pinned_acquire(ctx, 4096, nullptr);
        if (!nl_line) return FZP_ENOMEM;
        memset(nl_line, '\n', 64);
        for (int i = 0; i < n_bufs; i++) {
            uint8_t *b = (uint8_t *)fzp_pinned_acquire(ctx, bytes, nullptr);
            hipEvent_t e = nullptr;
            if (!b || hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { if (b) fzp_pinned_release(ctx, b); (void)hipGetLastError(); return FZP_ENOMEM; }
            stage.push_back(b); stage_ev.push_back(e); stage_used.push_back(0);
        }
        stage_turn.reset(new std::atomic<int64_t>[(size_t)n_bufs]);
        for (int i = 0; i < n_bufs; i++) stage_turn[(size_t)i].store(stage_ticket + ((i - stage_ticket % n_bufs) % n_bufs + n_bufs) % n_bufs);      // the next ticket that maps to buffer i
        return FZP_OK;
    }
    void drop_stage() {
        for (auto b : stage) fzp_pinned_release(stage_ctx, b);
        if (nl_line) { fzp_pinned_release(stage_ctx, nl_line); nl_line = nullptr; }
        for (auto e : stage_ev) if (e) (void)hipEventDestroy(e);
        stage.clear(); stage_ev.clear(); stage_used.clear(); stage_turn.reset(); stage_bytes = 0;
    }
    ~GroupPool() { drop_stage(); for (auto u : up) if (u) (void)hipStreamDestroy(u); for (auto e : prev) if (e) (void)hipEventDestroy(e); }
    std::mutex mu;
    std::vector<std::unique_ptr<GroupIn>> idle;
    std::unique_ptr<GroupIn> take() {
        std::lock_guard<std::mutex> lk(mu);
        if (idle.empty()) return std::unique_ptr<GroupIn>(new GroupIn());
        auto g = std::move(idle.back());
        idle.pop_back();
        return g;
    }
    void give(std::unique_ptr<GroupIn> g) { std::lock_guard<std::mutex> lk(mu); if (idle.size() < 4) idle.push_back(std::move(g)); }
};
static void group_pool_destroy(GroupPool *p) { delete p; }
namespace {
struct FaRec { const char *name; int32_t name_len; const char *s0, *s1; int64_t len; bool plain; };      // header's first word; the sequence's lines lie in [s0, s1); len = its bases; plain: one line, nothing to trim
struct FaPiece { int file; const char *a, *b; std::vector<int64_t> nl; std::vector<FaRec> recs; int64_t bases = 0, name_bytes = 0; bool all_plain = true; };
inline bool fa_sp(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }
// a record's sequence, joined: white space at line ends dropped (falcon_kit's FastaReader as phasing.py:490-494 uses it)
inline void copy_seq(const FaRec &R, uint8_t *dst) {
    const char *q = R.s0;
    while (q < R.s1) {
        const char *n2 = (const char *)memchr(q, '\n', (size_t)(R.s1 - q));
        const char *e = n2 ? n2 : R.s1;
        const char *u = q, *v = e;
        while (u < v && fa_sp(*u)) u++;
        while (v > u && fa_sp(v[-1])) v--;
        memcpy(dst, u, (size_t)(v - u));
        dst += v - u;
        q = n2 ? n2 + 1 : R.s1;
    }
}
// The loader.  Pass 1, per 4 MB piece of a file: read it (pread into the group's buffer) and, while it is warm, note where its line ends are.  Pass 2, per piece: the
// records whose header line STARTS in the piece, from the line ends alone (a record runs on into later pieces through their lists; only line starts and ends are looked at).
// A record of one plain line -- what falcon_kit writes -- is used where it lies: its span of the buffer goes to fzp_align_create_spans, nothing is copied.
struct LineCursor {      // line ends of a file at or after a position, across its pieces
    const std::vector<FaPiece> *pieces; size_t k, k_end, i;
    // the next line end at or after `from`, or f1 (the file's end) when there is none
    const char *next(const char *base, const char *f1) {
        while (k < k_end) {
            const auto &v = (*pieces)[k].nl;
            if (i < v.size()) return base + v[i++];
            k++; i = 0;
        }
        return f1;
    }
};
void records_of_piece(std::vector<FaPiece> &pieces, size_t k, size_t k_end, const char *base, const char *f0, const char *f1) {
    FaPiece &P = pieces[k];
    // line starts in [a, b): the file's start, and one behind every line end in [a - 1, b - 1)
    LineCursor cur{&pieces, k, k_end, 0};
    const char *s = nullptr;
    if (P.a == f0) s = f0;
    else if (P.a[-1] == '\n') s = P.a;
    else {      // the first line start of this piece lies behind its first line end
        const char *e = cur.k == k && !P.nl.empty() ? base + P.nl[0] : nullptr;
        if (!e || e + 1 >= P.b) return;
        cur.i = 1;
        s = e + 1;
    }
    while (s < P.b && s < f1) {
        const char *le = cur.next(base, f1);      // end of the line that starts at s
        if (*s != '>') { s = le + 1; continue; }  // (a sequence line of a record an earlier piece owns, or lines before a file's first header)
        FaRec R;
        const char *x = s + 1;
        while (x < le && fa_sp(*x)) x++;
        const char *y = x;
        while (y < le && !fa_sp(*y)) y++;
        R.name = x; R.name_len = (int32_t)(y - x);
        R.s0 = le < f1 ? le + 1 : f1;
        R.len = 0;
        int n_lines = 0;
        bool trimmed = false;
        const char *q = R.s0;
        // (the cursor may run past this piece: a record belongs to the piece of its header; what it leaves behind in the later lists is skipped by their own pieces)
        LineCursor c2 = cur;
        while (q < f1 && *q != '>') {
            const char *e = c2.next(base, f1);
            const char *u = q, *v = e;
            while (u < v && fa_sp(*u)) u++;
            while (v > u && fa_sp(v[-1])) v--;
            if (u != q || v != e) trimmed = true;
            R.len += v - u;
            n_lines++;
            q = e < f1 ? e + 1 : f1;
        }
        R.s1 = q;
        R.plain = n_lines == 1 && !trimmed;
        if (!R.plain && R.len > 0) P.all_plain = false;
        if (R.len == 0) R.plain = true;           // (nothing to copy either way)
        P.bases += R.len; P.name_bytes += R.name_len;
        P.recs.push_back(R);
        // go on at the next header: only line starts inside this piece count
        if (q >= P.b) break;
        cur = c2;
        s = q;
    }
}
struct Mapped { const char *p = nullptr; size_t n = 0; };
void load_group(const std::string &dir, const char *const *ctg_id, int c0, int c1, int n_threads, GroupIn &G) {
    const int gc = c1 - c0, nf = 2 * gc;            // file t < gc: <ctg t>_reads.fa, file gc + t: <ctg t>_ref.fa
    G.rc = FZP_OK; G.err.clear();
    const bool timing = getenv("FZP_PIPE_TIMING") != nullptr;
    const auto t_0 = clk::now();
    double t_map = 0, t_scan = 0;
    auto path_of = [&](int t) { return dir + "/" + ctg_id[c0 + (t < gc ? t : t - gc)] + (t < gc ? "_reads.fa" : "_ref.fa"); };
    // the files' bytes into one buffer of the group (kept with the context: warm pages), read in pieces of 4 MB by all threads -- mapping the files instead costs more
    // in page-table set-up and tear-down (10 ms each way per 300 MB) than the copy does
    std::vector<Mapped> mp((size_t)nf);
    std::vector<int> fds((size_t)nf, -1);
    std::vector<size_t> foff((size_t)nf + 1, 0);
    std::vector<std::string> errs((size_t)nf);
    std::mutex err_mu;
    std::atomic<int> next{0};
    auto par = [&](const std::function<void()> &work) {
        std::vector<std::thread> th;
        for (int i = 1; i < n_threads; i++) th.emplace_back(work);
        work();
        for (auto &x : th) x.join();
    };
    auto close_all = [&]() { for (int &fd : fds) if (fd >= 0) { close(fd); fd = -1; } };
    for (int t = 0; t < nf; t++) {
        const std::string path = path_of(t);
        fds[(size_t)t] = open(path.c_str(), O_RDONLY);
        struct stat sb;
        if (fds[(size_t)t] < 0 || fstat(fds[(size_t)t], &sb) != 0) { G.rc = FZP_EIO; G.err = path + ": " + strerror(errno); close_all(); return; }
        mp[(size_t)t].n = (size_t)sb.st_size;
        foff[(size_t)t + 1] = foff[(size_t)t] + mp[(size_t)t].n;
    }
    if (!G.raw.need(foff[(size_t)nf] + 1)) { G.rc = FZP_ENOMEM; G.err = "host memory for the group's files"; close_all(); return; }
    const char *base = (const char *)G.raw.data();
    for (int t = 0; t < nf; t++) mp[(size_t)t].p = base + foff[(size_t)t];
    // pieces of ~4 MB, in file order
    std::vector<FaPiece> pieces;
    std::vector<size_t> file_piece0((size_t)nf + 1, 0);
    size_t PIECE = 4u << 20;
    if (const char *e = getenv("FZP_FASTA_PIECE")) { const long v = atol(e); if (v > 0) PIECE = (size_t)v; }      // (tests: records that straddle many pieces)
    for (int t = 0; t < nf; t++) {
        file_piece0[(size_t)t] = pieces.size();
        for (size_t a = 0; a < mp[(size_t)t].n; a += PIECE) {
            FaPiece P;
            P.file = t; P.a = mp[(size_t)t].p + a; P.b = mp[(size_t)t].p + std::min(mp[(size_t)t].n, a + PIECE);
            pieces.push_back(std::move(P));
        }
    }
    file_piece0[(size_t)nf] = pieces.size();
    par([&]() {
        for (int k; (k = next.fetch_add(1)) < (int)pieces.size();) {
            FaPiece &P = pieces[(size_t)k];
            const int t = P.file;
            size_t at = (size_t)(P.a - mp[(size_t)t].p);
            const size_t end = (size_t)(P.b - mp[(size_t)t].p);
            bool ok = true;
            while (at < end) {
                const ssize_t got = pread(fds[(size_t)t], (void *)(mp[(size_t)t].p + at), end - at, (off_t)at);
                if (got <= 0) {      // (several pieces of one file are read by different threads: the first failure takes the file's slot, under a lock)
                    const std::string why = path_of(t) + ": " + (got < 0 ? strerror(errno) : "file shrank while it was read");
                    std::lock_guard<std::mutex> lk(err_mu);
                    if (errs[(size_t)t].empty()) errs[(size_t)t] = why;
                    ok = false; break;
                }
                at += (size_t)got;
            }
            if (!ok) continue;
            for (const char *q = P.a; q < P.b;) {      // its line ends, as offsets into the group's buffer
                const char *e = (const char *)memchr(q, '\n', (size_t)(P.b - q));
                if (!e) break;
                P.nl.push_back((int64_t)(e - base));
                q = e + 1;
            }
        }
    });
    close_all();
    t_map = ms_since(t_0);
    for (int t = 0; t < nf; t++) if (!errs[(size_t)t].empty()) { G.rc = FZP_EIO; G.err = errs[(size_t)t]; return; }
    next.store(0);
    par([&]() {
        for (int k; (k = next.fetch_add(1)) < (int)pieces.size();) {
            const int t = pieces[(size_t)k].file;
            records_of_piece(pieces, (size_t)k, file_piece0[(size_t)t + 1], base, mp[(size_t)t].p, mp[(size_t)t].p + mp[(size_t)t].n);
        }
    });
    t_scan = ms_since(t_0);
    // the contigs: of <ctg>_ref.fa the LAST record named <ctg> (the loop at phasing.py:490-494 leaves that one); none -> an empty contig
    std::vector<const FaRec *> ref_rec((size_t)gc, nullptr);
    for (auto &P : pieces)
        if (P.file >= gc) {
            const int c = P.file - gc;
            const size_t want = strlen(ctg_id[c0 + c]);
            for (auto &R : P.recs) if ((size_t)R.name_len == want && memcmp(R.name, ctg_id[c0 + c], want) == 0) ref_rec[(size_t)c] = &R;
        }
    G.ctg_ptr.assign((size_t)gc, (const uint8_t *)base);
    G.ctg_len.assign((size_t)gc, 0);
    std::vector<int64_t> ref_at((size_t)gc + 1, 0);      // room in G.ref for the contigs that have to be joined
    for (int c = 0; c < gc; c++) {
        const FaRec *R = ref_rec[(size_t)c];
        G.ctg_len[(size_t)c] = R ? R->len : 0;
        ref_at[(size_t)c + 1] = ref_at[(size_t)c] + (R && !R->plain ? R->len : 0);
    }
    if (!G.ref.need((size_t)ref_at[(size_t)gc] + 1)) { G.rc = FZP_ENOMEM; G.err = "host memory for the group's contigs"; return; }
    for (int c = 0; c < gc; c++) {
        const FaRec *R = ref_rec[(size_t)c];
        if (R) G.ctg_ptr[(size_t)c] = R->plain ? (const uint8_t *)R->s0 : G.ref.data() + ref_at[(size_t)c];
    }
    // the reads: every record of <ctg>_reads.fa, file order; a prefix sum over the pieces gives every piece its first read, base and name byte
    std::vector<int64_t> p_rec(pieces.size() + 1, 0), p_base(pieces.size() + 1, 0), p_name(pieces.size() + 1, 0);
    bool in_place = true;
    for (size_t k = 0; k < pieces.size(); k++) {
        const bool rd = pieces[k].file < gc;
        p_rec[k + 1] = p_rec[k] + (rd ? (int64_t)pieces[k].recs.size() : 0);
        p_base[k + 1] = p_base[k] + (rd ? pieces[k].bases : 0);
        p_name[k + 1] = p_name[k] + (rd ? pieces[k].name_bytes : 0);
        if (rd && !pieces[k].all_plain) in_place = false;
    }
    if (getenv("FZP_FASTA_JOIN")) in_place = false;      // (tests: the joining path on plain files)
    const int64_t nr = p_rec.back();
    if ((!in_place && !G.blob.need((size_t)p_base.back() + 1)) || !G.names.need((size_t)p_name.back() + 1)) { G.rc = FZP_ENOMEM; G.err = "host memory for the group's reads"; return; }
    G.in_place = in_place;
    G.seq_base = in_place ? G.raw.data() : G.blob.data();
    G.be.resize((size_t)nr * 2); G.noff.resize((size_t)nr + 1); G.read_ctg.resize((size_t)nr);
    G.noff[0] = 0;
    next.store(0);
    const int n_work = (int)pieces.size() + gc;
    par([&]() {
        for (int k; (k = next.fetch_add(1)) < n_work;) {
            if (k >= (int)pieces.size()) {      // a contig that has to be joined
                const int c = k - (int)pieces.size();
                if (ref_rec[(size_t)c] && !ref_rec[(size_t)c]->plain) copy_seq(*ref_rec[(size_t)c], G.ref.data() + ref_at[(size_t)c]);
                continue;
            }
            const FaPiece &P = pieces[(size_t)k];
            if (P.file >= gc) continue;
            int64_t r = p_rec[(size_t)k], ab = p_base[(size_t)k], an = p_name[(size_t)k];
            for (const FaRec &R : P.recs) {
                if (in_place) { const int64_t at = R.len ? (int64_t)(R.s0 - base) : 0; G.be[(size_t)(2 * r)] = at; G.be[(size_t)(2 * r + 1)] = at + R.len; }
                else { copy_seq(R, G.blob.data() + ab); G.be[(size_t)(2 * r)] = ab; G.be[(size_t)(2 * r + 1)] = ab + R.len; }
                memcpy(G.names.data() + an, R.name, (size_t)R.name_len);
                ab += R.len; an += R.name_len;
                G.noff[(size_t)r + 1] = an; G.read_ctg[(size_t)r] = P.file;
                r++;
            }
        }
    });
    if (timing) fprintf(stderr, "[load_group] %d contigs, %lld reads, %.1f MB on %d threads: read + line ends by %.2f ms, records by %.2f, %s by %.2f\n", gc, (long long)nr,
                        (double)p_base.back() / 1e6, n_threads, t_map, t_scan, in_place ? "spans (reads used in place)" : "reads joined", ms_since(t_0));
}
// r6: the loader that only READS.  The group's files go as they are into one DEVICE block -- 4 MB pieces, pread by the rank's threads into pinned staging buffers and sent
// from there -- with a '\n' behind every file (no line runs from one file into the next; an empty line more is nothing to a FASTA reader).  Line ends, records, lengths,
// names: fzp_fasta.hip, on the device.  The host parser above stays as the checker (FZP_FASTA_HOST=1, tests).
void load_group_raw(fzp_ctx *ctx, const std::string &dir, const char *const *ctg_id, int c0, int c1, int n_threads, GroupIn &G, const hipStream_t *ups, int n_up, GroupPool *gp = nullptr) {
    const int gc = c1 - c0, nf = 2 * gc;
    G.rc = FZP_OK; G.err.clear(); G.dev_parse = true; G.streamed = false;
    if (fzp_bind(ctx) != FZP_OK) { G.rc = FZP_EDEVICE; G.err = fzp_last_error(); return; }
    const auto t_0 = clk::now();
    auto path_of = [&](int t) { return dir + "/" + ctg_id[c0 + (t < gc ? t : t - gc)] + (t < gc ? "_reads.fa" : "_ref.fa"); };
    std::vector<int> fds((size_t)nf, -1);
    std::vector<size_t> fsz((size_t)nf, 0);
    auto close_all = [&]() { for (int &fd : fds) if (fd >= 0) { close(fd); fd = -1; } };
    G.foff.assign((size_t)nf + 1, 0);
    for (int t = 0; t < nf; t++) {
        const std::string path = path_of(t);
        fds[(size_t)t] = open(path.c_str(), O_RDONLY);
        struct stat sb;
        if (fds[(size_t)t] < 0 || fstat(fds[(size_t)t], &sb) != 0) { G.rc = FZP_EIO; G.err = path + ": " + strerror(errno); close_all(); return; }
        fsz[(size_t)t] = (size_t)sb.st_size;
        G.foff[(size_t)t + 1] = G.foff[(size_t)t] + (int64_t)fsz[(size_t)t] + 1;      // + the '\n' behind it
    }
    G.pin_bytes = G.foff[(size_t)nf];
    if (G.d_raw.alloc((size_t)G.pin_bytes + 64) != FZP_OK) { G.rc = FZP_ENOMEM; G.err = "device memory for the group's files"; close_all(); return; }
    for (int u = 0; u < n_up; u++)
        if (!G.ev_up[u] && hipEventCreateWithFlags(&G.ev_up[u], hipEventDisableTiming) != hipSuccess) { G.rc = FZP_EDEVICE; G.err = "hipEventCreate"; close_all(); return; }
    hipStream_t up = ups[0];
    // several streams keep several copy engines busy, but the link is one: without an order the next group's pieces (read while this group's are still on their way) would
    // share it with them and BOTH groups would arrive late.  Every stream first waits for the end of the previous group's copies on every stream.
    if (gp && n_up > 1 && gp->have_prev)
        for (int u = 0; u < n_up; u++) for (int v = 0; v < n_up; v++) (void)hipStreamWaitEvent(ups[u], gp->prev[v], 0);
    std::atomic<int> hip_bad{0};
    struct Piece { int t; size_t a, b; };
    std::vector<Piece> pieces;
    size_t PIECE = 4u << 20;
    if (const char *e = getenv("FZP_FASTA_PIECE")) { const long v = atol(e); if (v > 0) PIECE = (size_t)v; }
    for (int t = 0; t < nf; t++) for (size_t a = 0; a < fsz[(size_t)t]; a += PIECE) pieces.push_back({t, a, std::min(fsz[(size_t)t], a + PIECE)});
    if (hipMemsetAsync(G.d_raw.p + G.pin_bytes, 0, 64, up) != hipSuccess) hip_bad.store(1);
    int cap = 8;      // readers: eight keep ahead of the link (each moves 5-7 GB/s out of the page cache, the link takes ~37); sixteen only contend (profiles/r6_from_files.txt)
    if (const char *e = getenv("FZP_FASTA_READERS")) { const int g = atoi(e); if (g >= 1) cap = g; }
    const int T = (int)std::min<size_t>((size_t)std::max(1, std::min(n_threads, cap)), std::max<size_t>(1, pieces.size()));
    GroupPool own;      // (a call without a pool -- the test hook -- stages through buffers of its own)
    GroupPool *sp = gp ? gp : &own;
    if (!gp) { own.n_up = 0; }
    int ring = 64;
    if (const char *e = getenv("FZP_FASTA_RING")) { const int g = atoi(e); if (g >= 2 * T && g <= 1024) ring = g; }
    ring = std::max(ring, 2 * T);
    if (sp->ensure_stage(ctx, ring, PIECE) != FZP_OK) { G.rc = FZP_ENOMEM; G.err = "pinned staging buffers for the group's files"; close_all(); return; }
    std::vector<std::string> errs((size_t)nf);
    std::mutex err_mu;
    std::atomic<int> next{0};
    // (the separators: one-byte copies out of a pinned line of them -- copies like the pieces, so that the stream stays on the copy engine; fills are kernels)
    for (int t = 0; t < nf; t++)
        if (hipMemcpyAsync(G.d_raw.p + G.foff[(size_t)t] + (int64_t)fsz[(size_t)t], sp->nl_line, 1, hipMemcpyHostToDevice, up) != hipSuccess) hip_bad.store(1);
    const int64_t ticket0 = sp->stage_ticket;
    const int64_t R = (int64_t)sp->stage.size();
    sp->stage_ticket += (int64_t)pieces.size();
    auto work = [&](int) {
        if (fzp_bind(ctx) != FZP_OK) { hip_bad.store(1); return; }
        for (int k; (k = next.fetch_add(1)) < (int)pieces.size();) {
            const Piece &P = pieces[(size_t)k];
            const int64_t tk = ticket0 + k;
            const size_t bi = (size_t)(tk % R);
            uint8_t *buf = sp->stage[bi];
            while (sp->stage_turn[bi].load(std::memory_order_acquire) != tk) std::this_thread::yield();      // (the ticket a ring before this one has not queued its copy yet: a reader that fell far behind)
            if (sp->stage_used[bi] && hipEventSynchronize(sp->stage_ev[bi]) != hipSuccess) { hip_bad.store(1); sp->stage_turn[bi].store(tk + R, std::memory_order_release); break; }      // the copy that last left this buffer
            size_t at = P.a;
            while (at < P.b) {
                const ssize_t got = pread(fds[(size_t)P.t], buf + (at - P.a), P.b - at, (off_t)at);
                if (got <= 0) {
                    const std::string why = path_of(P.t) + ": " + (got < 0 ? strerror(errno) : "file shrank while it was read");
                    std::lock_guard<std::mutex> lk(err_mu);
                    if (errs[(size_t)P.t].empty()) errs[(size_t)P.t] = why;
                    break;
                }
                at += (size_t)got;
            }
            if (at != P.b) { sp->stage_turn[bi].store(tk + R, std::memory_order_release); continue; }
            // the piece is in pinned memory: on its way while the next one is read
            hipStream_t us = ups[k % n_up];
            const bool ok = hipMemcpyAsync(G.d_raw.p + G.foff[(size_t)P.t] + (int64_t)P.a, buf, P.b - P.a, hipMemcpyHostToDevice, us) == hipSuccess && hipEventRecord(sp->stage_ev[bi], us) == hipSuccess;
            if (ok) sp->stage_used[bi] = 1;
            sp->stage_turn[bi].store(tk + R, std::memory_order_release);      // the buffer's next ticket may come (it waits for the event first)
            if (!ok) { hip_bad.store(1); break; }
        }
    };
    {
        std::vector<std::thread> th;
        for (int i = 1; i < T; i++) th.emplace_back(work, i);
        work(0);
        for (auto &x : th) x.join();
    }
    close_all();
    bool bad = hip_bad.load() != 0;
    for (int t = 0; t < nf; t++) if (!errs[(size_t)t].empty()) { G.rc = FZP_EIO; G.err = errs[(size_t)t]; bad = true; break; }
    for (int u = 0; u < n_up && !bad; u++) if (hipEventRecord(G.ev_up[u], ups[u]) != hipSuccess) bad = true;
    G.n_up = n_up;
    if (gp && n_up > 1 && !bad) {
        for (int u = 0; u < n_up; u++) {
            if (!gp->prev[u] && hipEventCreateWithFlags(&gp->prev[u], hipEventDisableTiming) != hipSuccess) { bad = true; break; }
            if (hipEventRecord(gp->prev[u], ups[u]) != hipSuccess) { bad = true; break; }
        }
        gp->have_prev = !bad;
    }
    if (bad || !gp) {      // (a failed load, or buffers of this call's own: nothing may still be on its way out of them when they go)
        for (int u = 0; u < n_up; u++) (void)hipStreamSynchronize(ups[u]);
        if (bad) {
            (void)hipGetLastError();
            // (readers that gave up may have left tickets nobody took: the ring starts afresh at the next ticket -- every copy out of it is through)
            for (int64_t i = 0; i < R; i++) sp->stage_turn[(size_t)i].store(sp->stage_ticket + ((i - sp->stage_ticket % R) % R + R) % R);
            if (G.rc == FZP_OK) { G.rc = FZP_EDEVICE; G.err = "upload of the group's files failed"; }
            return;
        }
    }
    G.streamed = true;
    G.ms_read = ms_since(t_0);
    if (getenv("FZP_PIPE_TIMING")) fprintf(stderr, "[load_group_raw] %d contigs, %.1f MB in %zu pieces on %d threads: %.2f ms\n", gc, (double)G.pin_bytes / 1e6, pieces.size(), T, G.ms_read);
}

// the group on the device: its bytes (one DMA), its records (fzp_fasta.hip), and from those what the aligner and the writers need -- which records are reads (the reads
// files come first: a prefix), every read's contig and length, the names cut from the host's own copy, and per contig the LAST record of <ctg>_ref.fa named <ctg>
// (the loop at phasing.py:490-494 leaves that one; none: an empty contig)
struct GroupDev {
    const uint8_t *d_raw = nullptr;      // (the GroupIn's: uploaded by its loader)
    FaIndex X;
    DevBuf<int64_t> d_ctg_be;
    std::vector<int64_t> ctg_len, read_len;
    std::vector<int64_t> ctg_rec;      // per contig: the record it is (-1: none)
    int64_t n_reads = 0;
};
int group_to_device(fzp_ctx *lc, GroupIn &G, const char *const *ctg_id, int c0, int gc, GroupDev &D, std::mutex &up_mu) {
    hipStream_t st = lc->stream;
    const int nf = 2 * gc;
    (void)up_mu;
    if (!G.streamed || !G.d_raw.p) { fzp_set_error("group_to_device: the group's files are not on their way to the device"); return FZP_EINVAL; }
    for (int u = 0; u < G.n_up; u++) FZP_HIP(hipStreamWaitEvent(st, G.ev_up[u], 0));      // the loader's copies (its own streams)
    D.d_raw = G.d_raw.p;
    FZP_TRY(fzp_fasta_index_dev(lc, st, D.d_raw, G.pin_bytes, G.foff.data(), nf, D.X));
    const FaIndex &X = D.X;
    int64_t nr = 0;
    while (nr < X.n_rec && X.h_file[(size_t)nr] < gc) nr++;
    D.n_reads = nr;
    G.read_ctg.resize((size_t)nr); D.read_len.resize((size_t)nr); G.noff.resize((size_t)nr + 1);
    G.noff[0] = 0;
    for (int64_t r = 0; r < nr; r++) { G.read_ctg[(size_t)r] = X.h_file[(size_t)r]; D.read_len[(size_t)r] = X.h_len[(size_t)r]; G.noff[(size_t)r + 1] = G.noff[(size_t)r] + (X.h_name_e[(size_t)r] - X.h_name_b[(size_t)r]); }
    if (!G.names.need((size_t)G.noff[(size_t)nr] + 1)) { fzp_set_error("host memory for the group's read names"); return FZP_ENOMEM; }
    if (nr) memcpy(G.names.data(), X.h_names.data(), (size_t)G.noff[(size_t)nr]);      // (the reads' records come first: their names are the block's beginning)
    D.ctg_rec.assign((size_t)gc, -1);
    for (int64_t r = nr; r < X.n_rec; r++) {
        const int c = X.h_file[(size_t)r] - gc;
        const size_t want = strlen(ctg_id[c0 + c]);
        if ((size_t)(X.h_name_e[(size_t)r] - X.h_name_b[(size_t)r]) == want && memcmp(X.h_names.data() + X.h_noff[(size_t)r], ctg_id[c0 + c], want) == 0) D.ctg_rec[(size_t)c] = r;
    }
    D.ctg_len.assign((size_t)gc, 0);
    FZP_TRY(D.d_ctg_be.alloc((size_t)2 * gc));
    FZP_HIP(hipMemsetAsync(D.d_ctg_be.p, 0, (size_t)2 * gc * sizeof(int64_t), st));
    for (int c = 0; c < gc; c++)
        if (D.ctg_rec[(size_t)c] >= 0) {
            D.ctg_len[(size_t)c] = X.h_len[(size_t)D.ctg_rec[(size_t)c]];
            FZP_HIP(hipMemcpyAsync(D.d_ctg_be.p + 2 * c, X.d_be.p + 2 * D.ctg_rec[(size_t)c], 2 * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
        }
    return FZP_OK;
}
}  // namespace

// test hook (tests/test_host_logic.py, CPU only): what the FASTA reader makes of <reads_dir>/<ctg>_{ref,reads}.fa of the given contigs -- one group, as fzp_phase_contigs_files
// would hand it to fzp_align_create.  Outputs are malloc'ed (fzp_free): the contigs back to back with ref_off[n_ctg + 1], the reads with off / name offsets / contig per read.
extern "C" int fzp_debug_load_fasta_group(const char *reads_dir, const char *const *ctg_id, int32_t n_ctg, int32_t n_threads, uint8_t **ref, int64_t **ref_off, uint8_t **blob,
                                          int64_t **off, char **names, int64_t **name_off, int32_t **read_ctg, int64_t *n_reads) {
    if (!reads_dir || !ctg_id || n_ctg <= 0 || !ref || !ref_off || !blob || !off || !names || !name_off || !read_ctg || !n_reads) { fzp_set_error("fzp_debug_load_fasta_group: bad arguments"); return FZP_EINVAL; }
    GroupIn G;
    load_group(reads_dir, ctg_id, 0, n_ctg, n_threads > 0 ? n_threads : std::min(32, cores_per_rank()), G);
    if (G.rc != FZP_OK) { fzp_set_error("%s", G.err.c_str()); return G.rc; }
    auto dup = [](const void *p, size_t bytes) { void *q = malloc(bytes ? bytes : 1); if (q && bytes) memcpy(q, p, bytes); return q; };
    const size_t nr = G.read_ctg.size();
    {   // contigs and reads joined back to back, whichever way the loader holds them
        std::vector<int64_t> ro((size_t)n_ctg + 1, 0), o(nr + 1, 0);
        for (int c = 0; c < n_ctg; c++) ro[(size_t)c + 1] = ro[(size_t)c] + G.ctg_len[(size_t)c];
        for (size_t r = 0; r < nr; r++) o[r + 1] = o[r] + (G.be[2 * r + 1] - G.be[2 * r]);
        uint8_t *rb = (uint8_t *)malloc((size_t)ro[(size_t)n_ctg] + 1), *bb = (uint8_t *)malloc((size_t)o[nr] + 1);
        if (!rb || !bb) { free(rb); free(bb); fzp_set_error("fzp_debug_load_fasta_group: host memory"); return FZP_ENOMEM; }
        for (int c = 0; c < n_ctg; c++) memcpy(rb + ro[(size_t)c], G.ctg_ptr[(size_t)c], (size_t)G.ctg_len[(size_t)c]);
        for (size_t r = 0; r < nr; r++) memcpy(bb + o[r], G.seq_base + G.be[2 * r], (size_t)(o[r + 1] - o[r]));
        *ref = rb; *ref_off = (int64_t *)dup(ro.data(), ro.size() * 8);
        *blob = bb; *off = (int64_t *)dup(o.data(), o.size() * 8);
    }
    *names = (char *)dup(G.names.data(), (size_t)G.noff[nr]); *name_off = (int64_t *)dup(G.noff.data(), (nr + 1) * 8);
    *read_ctg = (int32_t *)dup(G.read_ctg.data(), nr * 4);
    *n_reads = (int64_t)nr;
    return FZP_OK;
}

// the same through the r6 reader (tests/test_gpu_pipe.py): the files read as they are, uploaded, indexed on the device (fzp_fasta.hip) -- and what the packer would read
// brought back as bytes, to be held against the host reader's output above, hostile files included
extern "C" int fzp_debug_load_fasta_group_dev(fzp_ctx *ctx, const char *reads_dir, const char *const *ctg_id, int32_t n_ctg, int32_t n_threads, uint8_t **ref, int64_t **ref_off, uint8_t **blob,
                                              int64_t **off, char **names, int64_t **name_off, int32_t **read_ctg, int64_t *n_reads) {
    if (!ctx || !reads_dir || !ctg_id || n_ctg <= 0 || !ref || !ref_off || !blob || !off || !names || !name_off || !read_ctg || !n_reads) { fzp_set_error("fzp_debug_load_fasta_group_dev: bad arguments"); return FZP_EINVAL; }
    FZP_TRY(fzp_bind(ctx));
    GroupIn G;
    load_group_raw(ctx, reads_dir, ctg_id, 0, n_ctg, n_threads > 0 ? n_threads : std::min(32, cores_per_rank()), G, &ctx->stream2, 1);
    if (G.rc != FZP_OK) { fzp_set_error("%s", G.err.c_str()); return G.rc; }
    GroupDev D;
    std::mutex mu;
    FZP_TRY(group_to_device(ctx, G, ctg_id, 0, n_ctg, D, mu));
    auto dup = [](const void *p, size_t bytes) { void *q = malloc(bytes ? bytes : 1); if (q && bytes) memcpy(q, p, bytes); return q; };
    const size_t nr = (size_t)D.n_reads;
    std::vector<int64_t> ro((size_t)n_ctg + 1, 0), o(nr + 1, 0);
    for (int c = 0; c < n_ctg; c++) ro[(size_t)c + 1] = ro[(size_t)c] + D.ctg_len[(size_t)c];
    for (size_t r = 0; r < nr; r++) o[r + 1] = o[r] + D.read_len[r];
    uint8_t *rb = (uint8_t *)malloc((size_t)ro[(size_t)n_ctg] + 1), *bb = (uint8_t *)malloc((size_t)o[nr] + 1);
    if (!rb || !bb) { free(rb); free(bb); fzp_set_error("fzp_debug_load_fasta_group_dev: host memory"); return FZP_ENOMEM; }
    int rc = fzp_fasta_fetch_seqs(ctx, ctx->stream, D.d_raw, D.X, 0, (int64_t)nr, bb);
    for (int c = 0; c < n_ctg && rc == FZP_OK; c++)
        if (D.ctg_rec[(size_t)c] >= 0) rc = fzp_fasta_fetch_seqs(ctx, ctx->stream, D.d_raw, D.X, D.ctg_rec[(size_t)c], 1, rb + ro[(size_t)c]);
    if (rc != FZP_OK) { free(rb); free(bb); return rc; }
    *ref = rb; *ref_off = (int64_t *)dup(ro.data(), ro.size() * 8);
    *blob = bb; *off = (int64_t *)dup(o.data(), o.size() * 8);
    *names = (char *)dup(G.names.data(), (size_t)G.noff[nr]); *name_off = (int64_t *)dup(G.noff.data(), (nr + 1) * 8);
    *read_ctg = (int32_t *)dup(G.read_ctg.data(), nr * 4);
    *n_reads = (int64_t)nr;
    return FZP_OK;
}

extern "C" int fzp_phase_contigs_files(fzp_ctx *ctx, const char *reads_dir, const fzp_names *nm, const fzp_pipe_opts *opts, fzp_pipe_out *out) {
    if (!ctx || !reads_dir || !nm || !nm->ctg_id || nm->n_ctg <= 0 || !out) { fzp_set_error("fzp_phase_contigs_files: bad arguments"); return FZP_EINVAL; }
    fzp_pipe_opts o;
    if (opts) o = *opts; else fzp_pipe_opts_default(&o);
    if (!getenv("FZP_PIPE_SYNC_WRITES")) o.flags |= FZP_PIPE_ASYNC_WRITES;      // (as in fzp_phase_contigs: flushed before the call returns)
    if (ctx->writer) (void)ctx->writer->extra.exchange(0);
    for (auto l : ctx->lanes) if (l->writer) (void)l->writer->extra.exchange(0);
    memset(out, 0, sizeof *out);
    const int n_ctg = nm->n_ctg;
    const std::string dir(reads_dir);
    const bool timing = getenv("FZP_PIPE_TIMING") != nullptr;
    const auto t_call = clk::now();
    MapsHolder mh;
    mh.start(&o);
    // read bases per contig ~ the size of its reads file (names are a few per cent of it): sizes the contig groups before anything is parsed
    std::vector<int64_t> bases((size_t)n_ctg, 0);
    for (int c = 0; c < n_ctg; c++) {
        struct stat sb;
        const std::string p = dir + "/" + nm->ctg_id[c] + "_reads.fa";
        if (stat(p.c_str(), &sb) != 0) { fzp_set_error("%s: %s", p.c_str(), strerror(errno)); return FZP_EIO; }
        bases[(size_t)c] = (int64_t)sb.st_size;
    }
    FZP_TRY(fzp_bind(ctx));
    size_t fr = 0, tot = 0;
    FZP_HIP(hipMemGetInfo(&fr, &tot));
    int lanes = o.n_lanes > 0 ? o.n_lanes : 2;
    int64_t group_bases = o.group_bases;
    if (group_bases <= 0) {
        int64_t all = 0;
        for (auto v : bases) all += v;
        group_bases = std::min<int64_t>((int64_t)((double)tot * 0.55 / lanes / 40.0), std::max<int64_t>(64ll << 20, all / (2 * lanes)));
    }
    struct Group { int c0, c1; };
    std::vector<Group> groups;
    for (int c = 0; c < n_ctg;) {
        int e = c;
        int64_t acc = 0;
        while (e < n_ctg && (e == c || acc + bases[(size_t)e] <= group_bases)) { acc += bases[(size_t)e]; e++; }
        groups.push_back({c, e});
        c = e;
    }
    lanes = std::min<int>(lanes, (int)groups.size());
    const int device = ctx->device;
    while ((int)ctx->lanes.size() < lanes - 1) {
        fzp_ctx *lc = nullptr;
        FZP_TRY(fzp_ctx_create(device, 0, &lc));
        ctx->lanes.push_back(lc);
    }
    const int host_threads = o.n_threads > 0 ? o.n_threads : std::min(32, cores_per_rank());
    const bool fa_host = getenv("FZP_FASTA_HOST") != nullptr;      // (the r5 reader: host threads find line ends and records; kept as the checker)
    // the loader: group g's files are parsed when a lane takes group g - 1 at the latest (one group ahead of every lane; each load uses every host thread, so
    // loads run one after the other, in group order: the first group is there as soon as it can be)
    if (!ctx->gpool) ctx->gpool = new GroupPool();
    GroupPool *pool = ctx->gpool;
    if (!pool->n_up) {
        int want = 1;      // (measured: 2 and 4 streams, with and without group order on the link, are within the noise of one -- 32-36 ms -- and cost the host more: profiles/r6_from_files.txt)
        if (const char *e = getenv("FZP_FASTA_UP_STREAMS")) { const int g = atoi(e); if (g >= 1 && g <= FA_UP_STREAMS) want = g; }
        for (int u = 0; u < want; u++) FZP_HIP(hipStreamCreateWithFlags(&pool->up[u], hipStreamNonBlocking));
        pool->n_up = want;
    }
    std::vector<std::unique_ptr<GroupIn>> gin(groups.size());
    std::vector<std::future<void>> loading(groups.size());
    std::mutex ld_mu, ld_serial;
    std::condition_variable ld_turn;
    size_t ld_next = 0;                          // the group whose load may run: loads take every host thread, so they run one at a time -- and in group order (a ticket, not
                                                 // just a mutex: group 1 must not slip in ahead of group 0 and keep the first lane waiting)
    size_t n_started = 0;
    auto start_loads = [&](size_t upto) {        // (callers hold ld_mu)
        for (; n_started < std::min(upto, groups.size()); n_started++) {
            const size_t g = n_started;
            gin[g] = pool->take();
            GroupIn *G = gin[g].get();
            const Group gr = groups[g];
            loading[g] = std::async(std::launch::async, [&, G, gr, g]() {
                { std::unique_lock<std::mutex> lk(ld_serial); ld_turn.wait(lk, [&] { return ld_next == g; }); }
                if (fa_host) load_group(dir, nm->ctg_id, gr.c0, gr.c1, host_threads, *G);
                else load_group_raw(ctx, dir, nm->ctg_id, gr.c0, gr.c1, host_threads, *G, pool->up, pool->n_up, pool);
                { std::lock_guard<std::mutex> lk(ld_serial); ld_next = g + 1; }
                ld_turn.notify_all();
            });
        }
    };
    { std::lock_guard<std::mutex> lk(ld_mu); start_loads((size_t)lanes + 1); }
    std::atomic<size_t> next{0};
    std::vector<int> rcs((size_t)lanes, FZP_OK);
    std::vector<std::string> errs((size_t)lanes);
    std::vector<fzp_pipe_out> outs((size_t)lanes);
    std::vector<std::vector<fzp_r2p>> r2p_g(groups.size());
    std::vector<double> ms_parse((size_t)lanes, 0.0);
    std::mutex up_mu;
    auto lane = [&](int li) {
        fzp_ctx *lc = li == 0 ? ctx : ctx->lanes[(size_t)li - 1];
        if (fzp_bind(lc) != FZP_OK) { rcs[(size_t)li] = FZP_EDEVICE; errs[(size_t)li] = fzp_last_error(); return; }
        fzp_pipe_out &po = outs[(size_t)li];
        memset(&po, 0, sizeof po);
        for (;;) {
            const size_t g = next.fetch_add(1);
            if (g >= groups.size()) break;
            const Group &Gr = groups[g];
            GroupIn *G;
            {
                std::lock_guard<std::mutex> lk(ld_mu);
                start_loads(g + (size_t)lanes + 1);
                G = gin[g].get();
            }
            auto t0 = clk::now();
            loading[g].wait();                                  // (normally done long ago: it was started a group ahead)
            ms_parse[(size_t)li] += ms_since(t0);
            int rc = G->rc;
            if (rc != FZP_OK) { errs[(size_t)li] = G->err; rcs[(size_t)li] = rc; break; }
            const int gc = Gr.c1 - Gr.c0;
            t0 = clk::now();
            fzp_alnjob *job = nullptr;
            int64_t gr_n = 0;
            if (G->dev_parse) {      // r6: the bytes as they are, the records found on the device
                GroupDev D;
                rc = group_to_device(lc, *G, nm->ctg_id, Gr.c0, gc, D, up_mu);
                gr_n = D.n_reads;
                if (timing) fprintf(stderr, "[lane %d group %zu] +%.2f ms: files on the device and indexed (%lld records, %lld lines, %lld joined bytes)\n", li, g, ms_since(t_call), (long long)D.X.n_rec, (long long)D.X.n_lines, (long long)D.X.join_bytes);
                if (rc == FZP_OK) {
                    fzp_aln_dev_src src;
                    src.d_raw = D.d_raw; src.d_ctg_be = D.d_ctg_be.p; src.d_read_be = D.X.d_be.p;
                    rc = fzp_align_create_dev(lc, gc, D.ctg_len.data(), gr_n, G->read_ctg.data(), D.read_len.data(), &src, &o.align, &job);
                }
            } else {
                gr_n = (int64_t)G->read_ctg.size();
                std::vector<const uint8_t *> cptr((size_t)gc);
                std::vector<int64_t> clen((size_t)gc);
                for (int c = 0; c < gc; c++) { clen[(size_t)c] = G->ctg_len[(size_t)c]; cptr[(size_t)c] = G->ctg_ptr[(size_t)c]; }
                // one upload at a time (see fzp_phase_contigs)
                std::lock_guard<std::mutex> lk(up_mu);
                rc = fzp_align_create_spans(lc, gc, cptr.data(), clen.data(), gr_n, G->read_ctg.data(), G->be.data(), G->seq_base, &o.align, &job);
            }
            po.ms_upload += ms_since(t0);
            if (timing) fprintf(stderr, "[lane %d group %zu] +%.2f ms: alignment job created (packed, index built)\n", li, g, ms_since(t_call));
            if (rc == FZP_OK) {
                fzp_names gn;
                gn.n_ctg = gc; gn.ctg_id = nm->ctg_id + Gr.c0;
                gn.name_off = G->noff.data();
                gn.names = (const char *)G->names.data();
                std::vector<int32_t> gi;
                for (int c = Gr.c0; c < Gr.c1; c++) gi.push_back(o.ctg_index ? o.ctg_index[c] : c);
                rc = job_phase_write(lc, job, &gn, &o, mh, gi.data(), &po, r2p_g[g]);
                if (timing) fprintf(stderr, "[lane %d group %zu] +%.2f ms: phased, write tasks queued\n", li, g, ms_since(t_call));
                if (rc == FZP_OK) {
                    std::vector<fzp_aln_summary> sm((size_t)gr_n);
                    if (gr_n && fzp_align_summaries(lc, job, sm.data()) == FZP_OK) for (auto &s : sm) po.dp_cells += (double)s.cells;
                }
            }
            if (rc != FZP_OK) errs[(size_t)li] = fzp_last_error();
            fzp_align_destroy(lc, job);
            if (timing) fprintf(stderr, "[lane %d group %zu] +%.2f ms: summaries read, job destroyed\n", li, g, ms_since(t_call));
            po.n_reads += gr_n;
            { std::lock_guard<std::mutex> lk(ld_mu); pool->give(std::move(gin[g])); }      // its buffers serve a later group (of this call or the next)
            if (rc != FZP_OK) { rcs[(size_t)li] = rc; break; }
        }
    };
    {
        std::vector<std::thread> th;
        for (int li = 1; li < lanes; li++) th.emplace_back(lane, li);
        lane(0);
        for (auto &x : th) x.join();
    }
    for (size_t g = 0; g < n_started; g++) if (loading[g].valid()) loading[g].wait();      // (after an error: loads still running hold references into this frame)
    for (auto &g : gin) if (g) pool->give(std::move(g));
    (void)fzp_bind(ctx);
    const int frc = fzp_pipe_flush(ctx);
    int64_t late_bytes = 0;
    if (ctx->writer) late_bytes += ctx->writer->extra.exchange(0);
    for (auto l : ctx->lanes) if (l->writer) late_bytes += l->writer->extra.exchange(0);
    if (timing) { double w = 0; for (auto v : ms_parse) w += v; fprintf(stderr, "[fzp_phase_contigs_files] %.2f ms in the call, %.2f ms of it waiting for the parser (%d lanes, %zu groups)\n", ms_since(t_call), w, lanes, groups.size()); }
    for (int li = 0; li < lanes; li++) if (rcs[(size_t)li] != FZP_OK) { fzp_set_error("%s", errs[(size_t)li].c_str()); return rcs[(size_t)li]; }
    if (frc != FZP_OK) return frc;
    for (int li = 0; li < lanes; li++) add(out, outs[(size_t)li]);
    out->bytes_written += late_bytes;
    std::vector<fzp_r2p> all;
    for (auto &v : r2p_g) all.insert(all.end(), v.begin(), v.end());
    out->n_r2p = (int64_t)all.size();
    out->r2p = (fzp_r2p *)malloc((all.size() ? all.size() : 1) * sizeof(fzp_r2p));
    if (!out->r2p) return FZP_ENOMEM;
    if (!all.empty()) memcpy(out->r2p, all.data(), all.size() * sizeof(fzp_r2p));
    return FZP_OK;
}

extern "C" int fzp_pipe_flush(fzp_ctx *ctx) {
    if (!ctx) return FZP_EINVAL;
    std::string err;
    if (ctx->writer) err = ctx->writer->drain();
    for (auto l : ctx->lanes) if (l->writer) { const std::string e = l->writer->drain(); if (err.empty()) err = e; }
    static const bool timing = getenv("FZP_PIPE_TIMING") != nullptr;
    if (timing && g_wt_tasks.load()) {
        const double n = (double)g_wt_tasks.exchange(0);
        fprintf(stderr, "[fzp_pipe] write tasks: %.0f; CPU ms per task: small texts %.3f, wait for the big texts %.3f, directories %.3f, the two big files %.3f, the small files %.3f\n", n,
                g_wt_fmt_ns.exchange(0) / 1e6 / n, g_wt_wait_ns.exchange(0) / 1e6 / n, g_wt_dir_ns.exchange(0) / 1e6 / n, g_wt_big_ns.exchange(0) / 1e6 / n, g_wt_small_ns.exchange(0) / 1e6 / n);
    }
    if (!err.empty()) { fzp_set_error("%s", err.c_str()); return FZP_EINVAL; }
    return FZP_OK;
}

extern "C" void fzp_pipe_out_free(fzp_pipe_out *o) {
    if (!o) return;
    free(o->r2p);
    o->r2p = nullptr; o->n_r2p = 0;
}
