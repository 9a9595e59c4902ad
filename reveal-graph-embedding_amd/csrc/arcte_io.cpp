// The two file formats either side of the ARCTE hot path, natively: the edge-list reader of
// reveal_graph_embedding/datautil/datarw.py:54-120 and the triplet feature writer of :123-143 (the console script
// entry_points/arcte.py:63-84 reads with the one and writes with the other).  Host C++, no GPU: the per-line Python of
// the reference parses 50 M lines and formats 0.9 G at BASELINE configs[2], next to a 0.7 s kernel.
//
// Reader semantics kept: a line is `line.strip().split(separator)` (common.py:36-49); a line whose first field begins
// with '#' is skipped; fields 0 and 1 are integers, field 2 a float (Python's int() / float() of the stripped field);
// node ids are renumbered in first-seen order, source before target (datarw.py:87-92); with `undirected` every non-loop
// edge is followed by its reciprocal (:105-109).  Lines are parsed by several threads, the renumbering is one ordered
// pass.  Writer: one line per stored entry in row-major order, `<original id><sep><column><sep><int(value)>`.
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "arcte_hip.h"

// arcte_hip.hip: fills the thread's arcte_hip_last_error() (library-internal, not part of the ABI)
extern "C" __attribute__((visibility("hidden"))) int arcte_io_set_error(int code, const char *msg);

namespace {

int io_fail(int code, const std::string &msg) { return arcte_io_set_error(code, msg.c_str()); }

int io_threads(size_t work_items, size_t per_thread)
{
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 8;
    size_t t = std::min<size_t>(std::min<unsigned>(hw, 32u), std::max<size_t>(1, work_items / std::max<size_t>(per_thread, 1)));
    if (const char *e = getenv("ARCTE_HIP_IO_THREADS")) {
        long v = atol(e);
        if (v > 0) t = (size_t)v;
    }
    return (int)std::max<size_t>(1, t);
}

bool is_space(unsigned char ch) { return ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r' || ch == '\f' || ch == '\v'; }

// PEP 515: int() and float() take single underscores BETWEEN digits ("1_000"); copies the field without them into buf
// (room for cap bytes), returns the new length or -1 when an underscore stands anywhere else / the field is too long
int drop_underscores(const char *b, const char *e, char *buf, int cap)
{
    int n = 0;
    for (const char *p = b; p < e; p++) {
        if (*p == '_') {
            if (p == b || p + 1 == e || p[-1] < '0' || p[-1] > '9' || p[1] < '0' || p[1] > '9') return -1;
            continue;
        }
        if (n + 1 >= cap) return -1;
        buf[n++] = *p;
    }
    buf[n] = 0;
    return n;
}

// Python's int(field) for the shapes an id takes: optional surrounding whitespace, optional sign, decimal digits (single
// underscores between them)
bool parse_int(const char *b, const char *e, int64_t &out)
{
    while (b < e && is_space((unsigned char)*b)) b++;
    while (e > b && is_space((unsigned char)e[-1])) e--;
    if (b == e) return false;
    char buf[64];
    if (memchr(b, '_', (size_t)(e - b))) {
        const int n = drop_underscores(b, e, buf, (int)sizeof(buf));
        if (n < 0) return false;
        b = buf;
        e = buf + n;
    }
    bool neg = false;
    if (*b == '+' || *b == '-') { neg = *b == '-'; b++; }
    if (b == e) return false;
    uint64_t v = 0;
    for (; b < e; b++) {
        if (*b < '0' || *b > '9') return false;
        if (v > (UINT64_MAX - 9) / 10) return false;
        v = v * 10 + (uint64_t)(*b - '0');
    }
    if (v > (uint64_t)INT64_MAX + (neg ? 1 : 0)) return false;
    out = neg ? (int64_t)(0 - v) : (int64_t)v;
    return true;
}

// Python's float(field): strtod over the stripped field, all of it -- minus what strtod takes and float() does not (hexadecimal
// floats "0x10", "nan(...)"), plus what float() takes and strtod does not (single underscores between digits).  strtod reads
// the decimal point of the C locale's LC_NUMERIC; this library never calls setlocale and CPython leaves LC_NUMERIC at "C".
bool parse_float(const char *b, const char *e, double &out)
{
    while (b < e && is_space((unsigned char)*b)) b++;
    while (e > b && is_space((unsigned char)e[-1])) e--;
    if (b == e || e - b > 120) return false;
    char buf[128];
    const int n = drop_underscores(b, e, buf, (int)sizeof(buf));
    if (n <= 0) return false;
    for (int i = 0; i < n; i++)
        if (buf[i] == 'x' || buf[i] == 'X' || buf[i] == '(') return false;
    char *end = nullptr;
    errno = 0;
    out = strtod(buf, &end);
    return end == buf + n;
}

struct RawEdge { int64_t src, dst; double w; };

struct ChunkResult {
    std::vector<RawEdge> edges;
    int64_t first_line = 0;      // (lines before this chunk: filled afterwards for the error message)
    int64_t lines = 0;
    int64_t bad_line = -1;       // first malformed line inside the chunk, counted from its start
    std::string bad_why;
};

void parse_chunk(const char *b, const char *e, const std::string &sep, ChunkResult &r)
{
    const size_t sl = sep.size();
    while (b < e) {
        // a line ends at "\n", at "\r\n" and at a lone "\r" (the reference iterates a text file: universal newlines)
        const char *nl = (const char *)memchr(b, '\n', (size_t)(e - b));
        const char *le = nl ? nl : e;
        const char *next = nl ? nl + 1 : e;
        if (const char *cr = (const char *)memchr(b, '\r', (size_t)(le - b))) {
            if (cr + 1 < le || (cr + 1 == le && !nl)) {      // not the "\r" of a "\r\n": the line ends here
                le = cr;
                next = cr + 1;
            }
        }
        // line.strip()
        const char *lb = b;
        while (lb < le && is_space((unsigned char)*lb)) lb++;
        const char *lend = le;
        while (lend > lb && is_space((unsigned char)lend[-1])) lend--;
        r.lines++;
        b = next;
        if (lb == lend) {      // words == [''] -> words[0][0] raises IndexError in the reference
            if (r.bad_line < 0) { r.bad_line = r.lines; r.bad_why = "empty line (the reference raises IndexError at datarw.py:80)"; }
            continue;
        }
        if (*lb == '#') continue;
        // .split(separator): the first three fields
        const char *f[4];
        int nf = 0;
        f[nf++] = lb;
        const char *p = lb;
        const char *fe[3] = {lend, lend, lend};
        while (nf < 4) {
            const char *hit = nullptr;
            if (sl == 1) hit = (const char *)memchr(p, sep[0], (size_t)(lend - p));
            else if (sl > 1 && (size_t)(lend - p) >= sl) {
                for (const char *q = p; q + sl <= lend; q++)
                    if (memcmp(q, sep.data(), sl) == 0) { hit = q; break; }
            }
            if (!hit) break;
            fe[nf - 1] = hit;
            p = hit + sl;
            f[nf++] = p;
        }
        if (nf < 3) {
            if (r.bad_line < 0) { r.bad_line = r.lines; r.bad_why = "fewer than three fields"; }
            continue;
        }
        RawEdge ed;
        if (!parse_int(f[0], fe[0], ed.src) || !parse_int(f[1], fe[1], ed.dst) || !parse_float(f[2], nf > 3 ? fe[2] : lend, ed.w)) {
            if (r.bad_line < 0) { r.bad_line = r.lines; r.bad_why = "a field is not a number"; }
            continue;
        }
        r.edges.push_back(ed);
    }
}

// open addressing, int64 key -> int32 value, insertion only
struct IdMap {
    std::vector<int64_t> keys;
    std::vector<int32_t> vals;
    size_t mask = 0, used = 0;
    void init(size_t cap_pow2) { keys.assign(cap_pow2, 0); vals.assign(cap_pow2, -1); mask = cap_pow2 - 1; used = 0; }
    static uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
    void grow()
    {
        std::vector<int64_t> ok;
        std::vector<int32_t> ov;
        ok.swap(keys); ov.swap(vals);
        init((mask + 1) * 2);
        for (size_t i = 0; i < ok.size(); i++)
            if (ov[i] >= 0) { size_t h = mix((uint64_t)ok[i]) & mask; while (vals[h] >= 0) h = (h + 1) & mask; keys[h] = ok[i]; vals[h] = ov[i]; used++; }
    }
    int32_t get_or_add(int64_t key, int32_t next)
    {
        if ((used + 1) * 2 > mask + 1) grow();
        size_t h = mix((uint64_t)key) & mask;
        while (vals[h] >= 0) {
            if (keys[h] == key) return vals[h];
            h = (h + 1) & mask;
        }
        keys[h] = key; vals[h] = next; used++;
        return next;
    }
};

// decimal digits of v into out (back to front), returns the length
inline int fmt_i64(int64_t v, char *out)
{
    char tmp[24];
    int n = 0;
    uint64_t u = v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    int len = 0;
    if (v < 0) out[len++] = '-';
    while (n) out[len++] = tmp[--n];
    return len;
}

}  // namespace

struct arcte_hip_edge_list {
    int64_t n_nodes = 0;
    std::vector<int32_t> row, col;
    std::vector<double> val;
    std::vector<int64_t> node_ids;        // new id -> original id
};

extern "C" {

int arcte_hip_edge_list_read(const char *path, const char *separator, int undirected, arcte_hip_edge_list **out)
{
    if (!path || !separator || !out) return io_fail(ARCTE_HIP_EINVAL, "bad argument");
    *out = nullptr;
    const std::string sep(separator);
    if (sep.empty()) return io_fail(ARCTE_HIP_EINVAL, "empty separator");
    int fd = open(path, O_RDONLY);
    if (fd < 0) return io_fail(ARCTE_HIP_EINVAL, std::string("cannot open ") + path + ": " + strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return io_fail(ARCTE_HIP_EINVAL, std::string("cannot stat ") + path); }
    const size_t size = (size_t)st.st_size;
    arcte_hip_edge_list *el = new arcte_hip_edge_list();
    if (size == 0) { close(fd); *out = el; return 0; }
    const char *base = (const char *)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (base == MAP_FAILED) { delete el; return io_fail(ARCTE_HIP_EINVAL, std::string("cannot map ") + path); }
    // chunks that end at line ends
    const int nt = io_threads(size, (size_t)4 << 20);
    std::vector<size_t> cut((size_t)nt + 1, size);
    cut[0] = 0;
    for (int t = 1; t < nt; t++) {
        size_t p = size / (size_t)nt * (size_t)t;
        if (p < cut[(size_t)t - 1]) p = cut[(size_t)t - 1];
        const char *nl = (const char *)memchr(base + p, '\n', size - p);
        cut[(size_t)t] = nl ? (size_t)(nl - base) + 1 : size;
    }
    std::vector<ChunkResult> res((size_t)nt);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++)
            th.emplace_back([&, t]() { parse_chunk(base + cut[(size_t)t], base + cut[(size_t)t + 1], sep, res[(size_t)t]); });
        for (auto &x : th) x.join();
    }
    munmap((void *)base, size);
    int64_t lines = 0, total = 0;
    for (int t = 0; t < nt; t++) {
        if (res[(size_t)t].bad_line >= 0) {
            std::string why = "line " + std::to_string(lines + res[(size_t)t].bad_line) + " of " + path + ": " + res[(size_t)t].bad_why;
            delete el;
            return io_fail(ARCTE_HIP_EINVAL, why);
        }
        lines += res[(size_t)t].lines;
        total += (int64_t)res[(size_t)t].edges.size();
    }
    // first-seen renumbering, source before target (datarw.py:87-92), reciprocal edges (:105-109): one ordered pass
    const int64_t cap = undirected ? 2 * total : total;
    if (cap >= ((int64_t)1 << 31)) { delete el; return io_fail(ARCTE_HIP_EINVAL, "more than 2^31 triplets"); }
    el->row.reserve((size_t)cap);
    el->col.reserve((size_t)cap);
    el->val.reserve((size_t)cap);
    IdMap map;
    // (sized from the edge count / 8, not 2x: ids repeat -- a 50M-edge file over 1M nodes would zero-fill 1.6 GB of table and
    //  probe it at random; get_or_add doubles the table when it fills to one half)
    size_t mcap = 1024;
    while (mcap < (size_t)std::min<int64_t>(total / 8, (int64_t)1 << 26)) mcap <<= 1;
    map.init(mcap);
    for (int t = 0; t < nt; t++) {
        for (const RawEdge &e : res[(size_t)t].edges) {
            const int32_t s = map.get_or_add(e.src, (int32_t)el->node_ids.size());
            if (s == (int32_t)el->node_ids.size()) el->node_ids.push_back(e.src);
            const int32_t d = map.get_or_add(e.dst, (int32_t)el->node_ids.size());
            if (d == (int32_t)el->node_ids.size()) el->node_ids.push_back(e.dst);
            if (el->node_ids.size() >= ((size_t)1 << 31) - 2) { delete el; return io_fail(ARCTE_HIP_EINVAL, "more than 2^31 nodes"); }
            el->row.push_back(s); el->col.push_back(d); el->val.push_back(e.w);
            if (undirected && s != d) { el->row.push_back(d); el->col.push_back(s); el->val.push_back(e.w); }
        }
        std::vector<RawEdge>().swap(res[(size_t)t].edges);
    }
    el->n_nodes = (int64_t)el->node_ids.size();
    *out = el;
    return 0;
}

int arcte_hip_edge_list_sizes(arcte_hip_edge_list *el, int64_t *n_nodes, int64_t *n_triplets)
{
    if (!el) return io_fail(ARCTE_HIP_EINVAL, "edge list is NULL");
    if (n_nodes) *n_nodes = el->n_nodes;
    if (n_triplets) *n_triplets = (int64_t)el->row.size();
    return 0;
}

int arcte_hip_edge_list_fetch(arcte_hip_edge_list *el, int32_t *row, int32_t *col, double *val, int64_t *node_ids)
{
    if (!el) return io_fail(ARCTE_HIP_EINVAL, "edge list is NULL");
    const size_t m = el->row.size();
    if (row && m) memcpy(row, el->row.data(), m * sizeof(int32_t));
    if (col && m) memcpy(col, el->col.data(), m * sizeof(int32_t));
    if (val && m) memcpy(val, el->val.data(), m * sizeof(double));
    if (node_ids && el->n_nodes) memcpy(node_ids, el->node_ids.data(), (size_t)el->n_nodes * sizeof(int64_t));
    return 0;
}

int arcte_hip_edge_list_destroy(arcte_hip_edge_list *el)
{
    delete el;
    return 0;
}

int arcte_hip_write_feature_triplets(const char *path, int64_t n_rows, const int64_t *indptr, const int32_t *indices, const int64_t *node_ids,
                                     const int64_t *doubled_diagonal, int64_t n_doubled, const char *separator)
{
    if (!path || !separator || n_rows < 0 || (n_rows && (!indptr || !node_ids)) || (n_doubled && !doubled_diagonal))
        return io_fail(ARCTE_HIP_EINVAL, "bad argument");
    const std::string sep(separator);
    const int64_t nnz = n_rows ? indptr[n_rows] : 0;
    if (nnz && !indices) return io_fail(ARCTE_HIP_EINVAL, "indices is NULL");
    std::vector<char> doubled((size_t)std::max<int64_t>(n_rows, 1), 0);
    for (int64_t k = 0; k < n_doubled; k++) {
        if (doubled_diagonal[k] < 0 || doubled_diagonal[k] >= n_rows) return io_fail(ARCTE_HIP_EINVAL, "doubled diagonal node out of range");
        doubled[(size_t)doubled_diagonal[k]] = 1;
    }
    // Pass 1 (parallel): the byte length of every block of rows (~1 M entries each) -- digits are counted, nothing is
    // formatted; pass 2 (parallel): every block is formatted straight into its place in the memory-mapped output file.
    const int64_t per_block = (int64_t)1 << 20;
    std::vector<int64_t> block_start{0};
    for (int64_t i = 0; i < n_rows;) {
        int64_t j = i;
        const int64_t from = indptr[i];
        while (j < n_rows && indptr[j + 1] - from <= per_block) j++;
        if (j == i) j = i + 1;
        block_start.push_back(j);
        i = j;
    }
    const int64_t nblocks = (int64_t)block_start.size() - 1;
    auto digits = [](uint64_t u) -> int { int d = 1; while (u >= 10) { u /= 10; d++; } return d; };
    std::vector<int64_t> block_bytes((size_t)std::max<int64_t>(nblocks, 1), 0);
    const int nt = io_threads((size_t)nnz, (size_t)1 << 20);
    auto for_blocks = [&](auto body) {
        std::atomic<int64_t> next{0};
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++)
            th.emplace_back([&]() { for (int64_t b = next.fetch_add(1); b < nblocks; b = next.fetch_add(1)) body(b); });
        for (auto &x : th) x.join();
    };
    const int64_t fixed = (int64_t)sep.size() * 2 + 2;          // two separators, the value, the newline
    for_blocks([&](int64_t b) {
        int64_t bytes = 0;
        for (int64_t i = block_start[(size_t)b]; i < block_start[(size_t)b + 1]; i++) {
            const int64_t id = node_ids[i];
            const int idlen = digits(id < 0 ? (uint64_t)0 - (uint64_t)id : (uint64_t)id) + (id < 0 ? 1 : 0);
            bytes += (indptr[i + 1] - indptr[i]) * (idlen + fixed);
            for (int64_t k = indptr[i]; k < indptr[i + 1]; k++) {
                const uint32_t c = (uint32_t)indices[k];
                bytes += c < 10 ? 1 : c < 100 ? 2 : c < 1000 ? 3 : c < 10000 ? 4 : c < 100000 ? 5 : c < 1000000 ? 6 : c < 10000000 ? 7 : c < 100000000 ? 8 : c < 1000000000 ? 9 : 10;
            }
        }
        block_bytes[(size_t)b] = bytes;
    });
    std::vector<int64_t> block_off((size_t)nblocks + 1, 0);
    for (int64_t b = 0; b < nblocks; b++) block_off[(size_t)b + 1] = block_off[(size_t)b] + block_bytes[(size_t)b];
    const int64_t total = block_off[(size_t)nblocks];
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return io_fail(ARCTE_HIP_EINVAL, std::string("cannot open ") + path + " for writing: " + strerror(errno));
    if (total == 0) { close(fd); return 0; }
    std::atomic<int> wrong{0};
    for_blocks([&](int64_t b) {
        // (formatted into a buffer of the thread and written at the block's offset: parallel pwrite() into the page cache
        //  measured twice as fast as storing through a shared file mapping, whose every 4 KB page is a fault)
        thread_local std::vector<char> buf;
        buf.resize((size_t)block_bytes[(size_t)b]);
        char *o = buf.data();
        char idbuf[24];
        for (int64_t i = block_start[(size_t)b]; i < block_start[(size_t)b + 1]; i++) {
            const int idlen = fmt_i64(node_ids[i], idbuf);
            for (int64_t k = indptr[i]; k < indptr[i + 1]; k++) {
                const int32_t c = indices[k];
                memcpy(o, idbuf, (size_t)idlen); o += idlen;
                memcpy(o, sep.data(), sep.size()); o += sep.size();
                o += fmt_i64(c, o);
                memcpy(o, sep.data(), sep.size()); o += sep.size();
                *o++ = (doubled[(size_t)i] && c == (int32_t)i) ? '2' : '1';
                *o++ = '\n';
            }
        }
        if (o != buf.data() + buf.size()) { wrong.store(1); return; }
        size_t done = 0;
        while (done < buf.size()) {
            const ssize_t w = pwrite(fd, buf.data() + done, buf.size() - done, (off_t)(block_off[(size_t)b] + (int64_t)done));
            if (w <= 0) { wrong.store(2); return; }
            done += (size_t)w;
        }
    });
    if (wrong.load() == 2) { close(fd); return io_fail(ARCTE_HIP_EINVAL, std::string("write to ") + path + " failed: " + strerror(errno)); }
    if (close(fd) != 0) return io_fail(ARCTE_HIP_EINVAL, std::string("closing ") + path + " failed: " + strerror(errno));
    if (wrong.load()) return io_fail(ARCTE_HIP_EINVAL, "internal error: a block of the feature file did not have its computed length");
    return 0;
}

}  // extern "C"
