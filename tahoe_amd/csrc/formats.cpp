// Host-side pieces of the Tahoe surface that need no device: node encoding, the two text file
// formats, deterministic synthetic inputs, error text.
//
// Reference behaviour restated here (file:line into sampathrg/Tahoe):
//   encode_node / dense_node_decode            Struct.h:103-117
//   generate_forest_from_file                  BaseTahoeTest.h:267-352
//   generate_data_from_file (host half)        BaseTahoeTest.h:354-402
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace tahoe {

static thread_local std::string g_last_error;

tahoe_status fail(tahoe_status code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
void clear_error() { g_last_error.clear(); }

namespace {

constexpr int32_t kFidMask = (int32_t)((1u << 30) - 1u);  // Struct.h:57
constexpr int32_t kDefLeftMask = (int32_t)(1u << 30);     // Struct.h:58
constexpr int32_t kIsLeafMask = (int32_t)(1u << 31);      // Struct.h:59

constexpr size_t kMaxLine = 1024;  // MAX_LINE, BaseTahoeTest.h:269,356

// Hands out the input one "fgets(buf, 1024, fp)" unit at a time: at most 1023 characters, ending
// after a newline.  At end of input the previous unit stays current, which is what the reference's
// unchecked fgets calls observe (BaseTahoeTest.h:298-307, :384).  Reads a memory-mapped file when it
// is given one, a FILE otherwise (pipes, files that cannot be mapped).
class LineFeed {
   public:
    explicit LineFeed(FILE *fp) : fp_(fp), chunk_(1 << 22) { unit_[0] = '\0'; }
    LineFeed(const char *mem, size_t len) : mem_(mem), len_(len) { unit_[0] = '\0'; }
    // Advances to the next unit; returns false (unit unchanged) at end of input.
    bool next()
    {
        size_t n = 0;
        while (n < kMaxLine - 1) {
            if (pos_ == len_) {
                if (!fp_) break;
                len_ = fread(chunk_.data(), 1, chunk_.size(), fp_);
                pos_ = 0;
                mem_ = chunk_.data();
                if (len_ == 0) break;
            }
            char c = mem_[pos_++];
            scratch_[n++] = c;
            if (c == '\n') break;
        }
        if (n == 0) return false;
        memcpy(unit_, scratch_, n);
        unit_[n] = '\0';
        return true;
    }
    int as_int() const { return (int)strtol(unit_, nullptr, 10); }  // atoi
    float as_float() const { return (float)strtod(unit_, nullptr); }  // atof, then double -> float
    size_t pos() const { return pos_; }  // memory mode: offset of the next unread byte
    void set_unit(const char *s, size_t n)
    {
        memcpy(unit_, s, n);
        unit_[n] = '\0';
    }

   private:
    FILE *fp_ = nullptr;
    std::vector<char> chunk_;
    const char *mem_ = nullptr;
    size_t pos_ = 0, len_ = 0;
    char scratch_[kMaxLine];
    char unit_[kMaxLine];
};

// A read-only mapping of a whole file (or nothing, when the file is empty, not regular or cannot be mapped).
struct Mapping {
    const char *p = nullptr;
    size_t n = 0;
    explicit Mapping(FILE *fp)
    {
        struct stat st;
        if (fstat(fileno(fp), &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0) return;
        void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(fp), 0);
        if (m == MAP_FAILED) return;
        (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
        p = static_cast<const char *>(m);
        n = (size_t)st.st_size;
    }
    ~Mapping()
    {
        if (p) munmap(const_cast<char *>(p), n);
    }
    Mapping(const Mapping &) = delete;
    Mapping &operator=(const Mapping &) = delete;
};

int loader_threads()
{
    if (const char *e = getenv("TAHOE_LOADER_THREADS")) {
        const int v = atoi(e);
        if (v >= 1) return std::min(v, 64);
    }
    return (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
}

// Parses the body [off, n) of a mapped file with several threads: `store(k, unit)` receives the k-th fgets unit
// (NUL-terminated copy, newline included) for k < wanted, from whichever thread owns its byte range.  The single-
// thread reader above costs ~60-100 ns per line in strtod (K3: 41 M model lines, 256 M data lines).
// Returns the number of units the body holds (capped at `wanted`), or SIZE_MAX when some line exceeds one fgets
// unit -- then units and lines no longer coincide and the caller falls back to the serial reader.  `last` receives
// the final unit stored (what a short file keeps repeating), unchanged if the body is empty.
template <class Store>
size_t parse_body_parallel(const char *p, size_t off, size_t n, size_t wanted, int threads, Store store, char *last)
{
    const size_t body = n - off;
    const int P = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, body / (1u << 16)));
    std::vector<size_t> cut(P + 1);
    cut[0] = off;
    cut[P] = n;
    for (int t = 1; t < P; ++t) {
        size_t c = std::max(cut[t - 1], off + body / P * t);
        const void *nl = c < n ? memchr(p + c, '\n', n - c) : nullptr;
        cut[t] = nl ? (size_t)(static_cast<const char *>(nl) - p) + 1 : n;
    }
    // pass 1: lines per range; a line is one unit iff it has <= 1022 characters before its newline (<= 1023 when
    // the file ends without one)
    std::vector<size_t> count(P, 0);
    std::atomic<bool> long_line{false};
    auto scan = [&](int t) {
        size_t c = cut[t], k = 0;
        while (c < cut[t + 1]) {
            const void *nl = memchr(p + c, '\n', cut[t + 1] - c);
            const size_t end = nl ? (size_t)(static_cast<const char *>(nl) - p) : cut[t + 1];
            if (end - c + (nl ? 1 : 0) > kMaxLine - 1) {
                long_line = true;
                return;
            }
            ++k;
            c = end + 1;
        }
        count[t] = k;
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < P; ++t) th.emplace_back(scan, t);
        scan(0);
        for (std::thread &x : th) x.join();
    }
    if (long_line) return SIZE_MAX;
    std::vector<size_t> first(P + 1, 0);
    for (int t = 0; t < P; ++t) first[t + 1] = first[t] + count[t];
    // pass 2: convert
    auto parse = [&](int t) {
        char unit[kMaxLine];
        size_t c = cut[t], k = first[t];
        while (c < cut[t + 1] && k < wanted) {
            const void *nl = memchr(p + c, '\n', cut[t + 1] - c);
            const size_t end = nl ? (size_t)(static_cast<const char *>(nl) - p) + 1 : cut[t + 1];
            memcpy(unit, p + c, end - c);
            unit[end - c] = '\0';
            store(k, unit);
            ++k;
            c = end;
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < P; ++t) th.emplace_back(parse, t);
        parse(0);
        for (std::thread &x : th) x.join();
    }
    const size_t have = std::min(first[P], wanted);
    if (first[P] > 0) {
        // the last unit a sequential reader would have consumed: unit number have-1, found from the end of its range
        int t = P - 1;
        while (t > 0 && first[t] > have - 1) --t;
        size_t c = cut[t], k = first[t];
        for (;;) {
            const void *nl = memchr(p + c, '\n', cut[t + 1] - c);
            const size_t end = nl ? (size_t)(static_cast<const char *>(nl) - p) + 1 : cut[t + 1];
            if (k == have - 1) {
                memcpy(last, p + c, end - c);
                last[end - c] = '\0';
                break;
            }
            ++k;
            c = end;
        }
    }
    return have;
}

}  // namespace
}  // namespace tahoe

using namespace tahoe;

extern "C" {

const char *tahoe_last_error(void) { return g_last_error.c_str(); }
int tahoe_abi_version(void) { return TAHOE_AMD_ABI_VERSION; }

int tahoe_tree_num_nodes(int depth) { return (1 << (depth + 1)) - 1; }

void tahoe_encode_node(tahoe_dense_node *n, int fid, float value, int def_left, float weight, int is_leaf)
{
    n->weight = weight;
    n->val = value;
    n->bits = (fid & kFidMask) | (def_left ? kDefLeftMask : 0) | (is_leaf ? kIsLeafMask : 0);
}

void tahoe_decode_node(const tahoe_dense_node *n, float *value, float *weight, int *fid, int *def_left,
                       int *is_leaf)
{
    if (value) *value = n->val;
    if (weight) *weight = n->weight;
    if (fid) *fid = n->bits & kFidMask;
    if (def_left) *def_left = (n->bits & kDefLeftMask) != 0;
    if (is_leaf) *is_leaf = (n->bits & kIsLeafMask) != 0;
}

tahoe_status tahoe_load_model(const char *path, int *num_trees, int *depth, tahoe_dense_node **nodes_out)
{
    if (!path || !num_trees || !depth || !nodes_out)
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_load_model: null argument");
    FILE *fp = fopen(path, "r");
    if (!fp) return fail(TAHOE_ERR_IO, "fail to read: %s: %s", path, strerror(errno));
    Mapping map(fp);
    LineFeed file_feed(fp), mem_feed(map.p, map.n);
    LineFeed &in = map.p ? mem_feed : file_feed;
    if (in.next()) *num_trees = in.as_int();
    if (in.next()) *depth = in.as_int() - 1;  // the file stores levels = depth + 1
    if (*num_trees < 0 || *depth < 0 || *depth > 30) {
        fclose(fp);
        return fail(TAHOE_ERR_INVALID_ARG, "model header out of range: num_trees=%d depth=%d", *num_trees,
                    *depth);
    }
    const size_t total = (size_t)*num_trees * (size_t)tahoe_tree_num_nodes(*depth);
    tahoe_dense_node *nodes = (tahoe_dense_node *)calloc(total ? total : 1, sizeof(tahoe_dense_node));
    if (!nodes) {
        fclose(fp);
        return fail(TAHOE_ERR_NO_MEMORY, "tahoe_load_model: %zu nodes", total);
    }
    size_t done = 0;  // nodes already filled by the parallel reader
    const int threads = loader_threads();
    if (map.p && threads > 1 && total > 0) {
        // unit k of the body is field k % 5 of node k / 5; the three flag fields of a node may come from two threads
        auto store = [nodes](size_t k, const char *unit) {
            tahoe_dense_node &nd = nodes[k / 5];
            switch (k % 5) {
                case 0: __atomic_fetch_or(&nd.bits, (int32_t)strtol(unit, nullptr, 10) & kFidMask, __ATOMIC_RELAXED); break;
                case 1: nd.val = (float)strtod(unit, nullptr); break;
                case 2: if ((int)strtol(unit, nullptr, 10) != 0) __atomic_fetch_or(&nd.bits, kDefLeftMask, __ATOMIC_RELAXED); break;
                case 3: nd.weight = (float)strtod(unit, nullptr); break;
                default: if ((int)strtol(unit, nullptr, 10) != 0) __atomic_fetch_or(&nd.bits, kIsLeafMask, __ATOMIC_RELAXED); break;
            }
        };
        char last[kMaxLine];
        last[0] = '\0';
        const size_t have = parse_body_parallel(map.p, in.pos(), map.n, total * 5, threads, store, last);
        if (have != SIZE_MAX) {
            // a short file: the remaining fields all read the last unit again (the header's if the body is empty)
            if (have > 0) in.set_unit(last, strlen(last));
            for (size_t k = have; k < total * 5; ++k) {
                tahoe_dense_node &nd = nodes[k / 5];
                switch (k % 5) {
                    case 0: nd.bits |= in.as_int() & kFidMask; break;
                    case 1: nd.val = in.as_float(); break;
                    case 2: if (in.as_int() != 0) nd.bits |= kDefLeftMask; break;
                    case 3: nd.weight = in.as_float(); break;
                    default: if (in.as_int() != 0) nd.bits |= kIsLeafMask; break;
                }
            }
            done = total;
        } else {
            memset(nodes, 0, total * sizeof(tahoe_dense_node));
        }
    }
    for (size_t i = done; i < total; ++i) {
        in.next();
        const int fid = in.as_int();
        in.next();
        const float value = in.as_float();
        in.next();
        const bool def_left = in.as_int() != 0;
        in.next();
        const float weight = in.as_float();
        in.next();
        const bool is_leaf = in.as_int() != 0;
        tahoe_encode_node(&nodes[i], fid, value, def_left, weight, is_leaf);
    }
    fclose(fp);
    *nodes_out = nodes;
    return TAHOE_OK;
}

tahoe_status tahoe_load_data(const char *path, int *num_rows, int *num_cols, float *missing, float **data_out)
{
    if (!path || !num_rows || !num_cols || !missing || !data_out)
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_load_data: null argument");
    FILE *fp = fopen(path, "r");
    if (!fp) return fail(TAHOE_ERR_IO, "fail to read: %s: %s", path, strerror(errno));
    Mapping map(fp);
    LineFeed file_feed(fp), mem_feed(map.p, map.n);
    LineFeed &in = map.p ? mem_feed : file_feed;
    if (in.next()) *num_rows = in.as_int();
    if (in.next()) *num_cols = in.as_int();
    if (in.next()) *missing = in.as_float();
    if (*num_rows < 0 || *num_cols < 0) {
        fclose(fp);
        return fail(TAHOE_ERR_INVALID_ARG, "data header out of range: rows=%d cols=%d", *num_rows, *num_cols);
    }
    const size_t total = (size_t)*num_rows * (size_t)*num_cols;  // 64-bit (int in the reference, :377)
    float *data = (float *)malloc((total ? total : 1) * sizeof(float));
    if (!data) {
        fclose(fp);
        return fail(TAHOE_ERR_NO_MEMORY, "tahoe_load_data: %zu values", total);
    }
    size_t done = 0;
    const int threads = loader_threads();
    if (map.p && threads > 1 && total > 0) {
        auto store = [data](size_t k, const char *unit) { data[k] = (float)strtod(unit, nullptr); };
        char last[kMaxLine];
        last[0] = '\0';
        const size_t have = parse_body_parallel(map.p, in.pos(), map.n, total, threads, store, last);
        if (have != SIZE_MAX) {
            if (have > 0) in.set_unit(last, strlen(last));
            const float v = in.as_float();
            for (size_t k = have; k < total; ++k) data[k] = v;
            done = total;
        }
    }
    for (size_t i = done; i < total; ++i) {
        in.next();
        data[i] = in.as_float();
    }
    fclose(fp);
    *data_out = data;
    return TAHOE_OK;
}

tahoe_status tahoe_write_model(const char *path, int num_trees, int depth, const tahoe_dense_node *nodes)
{
    if (!path || !nodes || num_trees < 0 || depth < 0)
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_write_model: bad argument");
    FILE *fp = fopen(path, "w");
    if (!fp) return fail(TAHOE_ERR_IO, "cannot write %s: %s", path, strerror(errno));
    fprintf(fp, "%d\n%d\n", num_trees, depth + 1);
    const size_t total = (size_t)num_trees * (size_t)tahoe_tree_num_nodes(depth);
    for (size_t i = 0; i < total; ++i) {
        int fid, def_left, is_leaf;
        float value, weight;
        tahoe_decode_node(&nodes[i], &value, &weight, &fid, &def_left, &is_leaf);
        // %.9g round-trips every float32 through strtod -> float.
        fprintf(fp, "%d\n%.9g\n%d\n%.9g\n%d\n", fid, value, def_left, weight, is_leaf);
    }
    if (fclose(fp) != 0) return fail(TAHOE_ERR_IO, "write error on %s", path);
    return TAHOE_OK;
}

tahoe_status tahoe_write_data(const char *path, int num_rows, int num_cols, float missing, const float *data)
{
    if (!path || (!data && num_rows * (size_t)num_cols) || num_rows < 0 || num_cols < 0)
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_write_data: bad argument");
    FILE *fp = fopen(path, "w");
    if (!fp) return fail(TAHOE_ERR_IO, "cannot write %s: %s", path, strerror(errno));
    fprintf(fp, "%d\n%d\n%.9g\n", num_rows, num_cols, missing);
    const size_t total = (size_t)num_rows * (size_t)num_cols;
    for (size_t i = 0; i < total; ++i) fprintf(fp, "%.9g\n", data[i]);
    if (fclose(fp) != 0) return fail(TAHOE_ERR_IO, "write error on %s", path);
    return TAHOE_OK;
}

// ---- binary cache: the text formats cost seconds to minutes at K3 size -------------------------
// One 64-byte header, then the payload exactly as it sits in memory (dense_node_t AoS / row-major float32), so a
// load is one read() into the array the caller gets.
namespace {
struct BinHeader {
    char magic[8];  // "TAHOEBIN"
    uint32_t version;
    uint32_t kind;  // 1 model, 2 data
    int32_t a, b;   // model: num_trees, depth; data: num_rows, num_cols
    float missing;  // data only
    uint32_t reserved;
    uint64_t payload_bytes;
    uint64_t checksum;    // of the payload
    int64_t src_size;     // size and mtime of the text file this was parsed from (0 = not a cache of a text file)
    int64_t src_mtime_ns;
};
static_assert(sizeof(BinHeader) == 64, "BinHeader is the on-disk layout");
constexpr uint32_t kBinVersion = 1;

// four independent multiply-add lanes over 64-bit words, folded; the tail bytes are zero-extended
static uint64_t payload_checksum(const void *p, size_t n)
{
    const uint64_t K = 0x9E3779B97F4A7C15ull;
    uint64_t h[4] = {1, 2, 3, 4};
    const unsigned char *b = static_cast<const unsigned char *>(p);
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        memcpy(w, b + i, 32);
        for (int l = 0; l < 4; ++l) h[l] = (h[l] ^ w[l]) * K + 0x632BE59BD9B4E019ull;
    }
    uint64_t tail[4] = {0, 0, 0, 0};
    memcpy(tail, b + i, n - i);
    for (int l = 0; l < 4; ++l) h[l] = (h[l] ^ tail[l]) * K;
    uint64_t r = n;
    for (int l = 0; l < 4; ++l) r = (r ^ (h[l] >> 29) ^ h[l]) * K;
    return r ^ (r >> 32);
}

static tahoe_status write_bin(const char *path, uint32_t kind, int a, int b, float missing, const void *payload, size_t bytes,
                       int64_t src_size, int64_t src_mtime_ns)
{
    BinHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "TAHOEBIN", 8);
    h.version = kBinVersion;
    h.kind = kind;
    h.a = a;
    h.b = b;
    h.missing = missing;
    h.payload_bytes = bytes;
    h.checksum = payload_checksum(payload, bytes);
    h.src_size = src_size;
    h.src_mtime_ns = src_mtime_ns;
    // write beside the target, then rename: a reader never sees a half-written cache
    const std::string tmp = std::string(path) + ".tmp" + std::to_string((long)getpid());
    FILE *fp = fopen(tmp.c_str(), "wb");
    if (!fp) return fail(TAHOE_ERR_IO, "cannot write %s: %s", tmp.c_str(), strerror(errno));
    bool ok = fwrite(&h, sizeof(h), 1, fp) == 1 && (bytes == 0 || fwrite(payload, 1, bytes, fp) == bytes);
    ok = (fclose(fp) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path) != 0) {
        (void)remove(tmp.c_str());
        return fail(TAHOE_ERR_IO, "write error on %s", path);
    }
    return TAHOE_OK;
}

// Reads and validates; *payload is malloc'ed.  want_src_*: when non-zero the header must name that text file state.
static tahoe_status read_bin(const char *path, uint32_t kind, BinHeader *h, void **payload, size_t elem_bytes)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return fail(TAHOE_ERR_IO, "fail to read: %s: %s", path, strerror(errno));
    if (fread(h, sizeof(*h), 1, fp) != 1 || memcmp(h->magic, "TAHOEBIN", 8) != 0) {
        fclose(fp);
        return fail(TAHOE_ERR_IO, "%s: not a tahoe binary file", path);
    }
    if (h->version != kBinVersion || h->kind != kind) {
        fclose(fp);
        return fail(TAHOE_ERR_IO, "%s: version %u kind %u, expected version %u kind %u", path, h->version, h->kind,
                    kBinVersion, kind);
    }
    size_t count = 0;
    bool shape_ok = false;
    if (kind == 1 && h->a >= 0 && h->b >= 0 && h->b <= 30) {
        count = (size_t)h->a * (size_t)tahoe_tree_num_nodes(h->b);
        shape_ok = true;
    } else if (kind == 2 && h->a >= 0 && h->b >= 0) {
        count = (size_t)h->a * (size_t)h->b;
        shape_ok = true;
    }
    if (!shape_ok || h->payload_bytes != count * elem_bytes) {
        fclose(fp);
        return fail(TAHOE_ERR_IO, "%s: header shape (%d, %d) does not match a payload of %llu bytes", path, h->a, h->b,
                    (unsigned long long)h->payload_bytes);
    }
    void *buf = malloc(h->payload_bytes ? h->payload_bytes : 1);
    if (!buf) {
        fclose(fp);
        return fail(TAHOE_ERR_NO_MEMORY, "%s: %llu bytes", path, (unsigned long long)h->payload_bytes);
    }
    const size_t got = h->payload_bytes ? fread(buf, 1, h->payload_bytes, fp) : 0;
    const bool trailing = fgetc(fp) != EOF;
    fclose(fp);
    if (got != h->payload_bytes || trailing) {
        free(buf);
        return fail(TAHOE_ERR_IO, "%s: payload is %s than the header says", path, trailing ? "longer" : "shorter");
    }
    if (payload_checksum(buf, h->payload_bytes) != h->checksum) {
        free(buf);
        return fail(TAHOE_ERR_IO, "%s: payload checksum mismatch (corrupt file)", path);
    }
    *payload = buf;
    return TAHOE_OK;
}

static bool stat_source(const char *path, int64_t *size, int64_t *mtime_ns)
{
    struct stat st;
    if (stat(path, &st) != 0) return false;
    *size = (int64_t)st.st_size;
    *mtime_ns = (int64_t)st.st_mtim.tv_sec * 1000000000ll + (int64_t)st.st_mtim.tv_nsec;
    return true;
}
}  // namespace

tahoe_status tahoe_save_model_bin(const char *path, int num_trees, int depth, const tahoe_dense_node *nodes)
{
    if (!path || num_trees < 0 || depth < 0 || depth > 30 || (!nodes && num_trees))
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_save_model_bin: bad argument");
    const size_t total = (size_t)num_trees * (size_t)tahoe_tree_num_nodes(depth);
    return write_bin(path, 1, num_trees, depth, 0.f, nodes, total * sizeof(tahoe_dense_node), 0, 0);
}

tahoe_status tahoe_load_model_bin(const char *path, int *num_trees, int *depth, tahoe_dense_node **nodes_out)
{
    if (!path || !num_trees || !depth || !nodes_out)
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_load_model_bin: null argument");
    BinHeader h;
    void *buf = nullptr;
    tahoe_status st = read_bin(path, 1, &h, &buf, sizeof(tahoe_dense_node));
    if (st != TAHOE_OK) return st;
    *num_trees = h.a;
    *depth = h.b;
    *nodes_out = static_cast<tahoe_dense_node *>(buf);
    return TAHOE_OK;
}

tahoe_status tahoe_save_data_bin(const char *path, int num_rows, int num_cols, float missing, const float *data)
{
    if (!path || num_rows < 0 || num_cols < 0 || (!data && (size_t)num_rows * (size_t)num_cols))
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_save_data_bin: bad argument");
    return write_bin(path, 2, num_rows, num_cols, missing, data, (size_t)num_rows * (size_t)num_cols * sizeof(float), 0, 0);
}

tahoe_status tahoe_load_data_bin(const char *path, int *num_rows, int *num_cols, float *missing, float **data_out)
{
    if (!path || !num_rows || !num_cols || !missing || !data_out)
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_load_data_bin: null argument");
    BinHeader h;
    void *buf = nullptr;
    tahoe_status st = read_bin(path, 2, &h, &buf, sizeof(float));
    if (st != TAHOE_OK) return st;
    *num_rows = h.a;
    *num_cols = h.b;
    *missing = h.missing;
    *data_out = static_cast<float *>(buf);
    return TAHOE_OK;
}

// Text file with a binary cache beside it ("<path>.tbin"): the cache is used when it names the text file's current
// size and mtime, otherwise the text is parsed (exactly as tahoe_load_model / tahoe_load_data) and the cache is
// rewritten, best effort.  The in/out defaults only matter on the text path; a cache stores what that path produced.
tahoe_status tahoe_load_model_cached(const char *path, int *num_trees, int *depth, tahoe_dense_node **nodes_out, int *from_cache)
{
    if (!path || !num_trees || !depth || !nodes_out)
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_load_model_cached: null argument");
    if (from_cache) *from_cache = 0;
    const std::string bin = std::string(path) + ".tbin";
    int64_t size = 0, mtime = 0;
    const bool have_src = stat_source(path, &size, &mtime);
    if (have_src) {
        BinHeader h;
        void *buf = nullptr;
        if (read_bin(bin.c_str(), 1, &h, &buf, sizeof(tahoe_dense_node)) == TAHOE_OK) {
            if (h.src_size == size && h.src_mtime_ns == mtime) {
                *num_trees = h.a;
                *depth = h.b;
                *nodes_out = static_cast<tahoe_dense_node *>(buf);
                if (from_cache) *from_cache = 1;
                clear_error();
                return TAHOE_OK;
            }
            free(buf);
        }
        clear_error();
    }
    tahoe_status st = tahoe_load_model(path, num_trees, depth, nodes_out);
    if (st != TAHOE_OK) return st;
    const size_t total = (size_t)*num_trees * (size_t)tahoe_tree_num_nodes(*depth);
    if (write_bin(bin.c_str(), 1, *num_trees, *depth, 0.f, *nodes_out, total * sizeof(tahoe_dense_node), size, mtime) != TAHOE_OK)
        clear_error();  // a read-only directory is not an error for the load
    return TAHOE_OK;
}

tahoe_status tahoe_load_data_cached(const char *path, int *num_rows, int *num_cols, float *missing, float **data_out, int *from_cache)
{
    if (!path || !num_rows || !num_cols || !missing || !data_out)
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_load_data_cached: null argument");
    if (from_cache) *from_cache = 0;
    const std::string bin = std::string(path) + ".tbin";
    int64_t size = 0, mtime = 0;
    const bool have_src = stat_source(path, &size, &mtime);
    if (have_src) {
        BinHeader h;
        void *buf = nullptr;
        if (read_bin(bin.c_str(), 2, &h, &buf, sizeof(float)) == TAHOE_OK) {
            if (h.src_size == size && h.src_mtime_ns == mtime) {
                *num_rows = h.a;
                *num_cols = h.b;
                *missing = h.missing;
                *data_out = static_cast<float *>(buf);
                if (from_cache) *from_cache = 1;
                clear_error();
                return TAHOE_OK;
            }
            free(buf);
        }
        clear_error();
    }
    tahoe_status st = tahoe_load_data(path, num_rows, num_cols, missing, data_out);
    if (st != TAHOE_OK) return st;
    if (write_bin(bin.c_str(), 2, *num_rows, *num_cols, *missing, *data_out,
                  (size_t)*num_rows * (size_t)*num_cols * sizeof(float), size, mtime) != TAHOE_OK)
        clear_error();
    return TAHOE_OK;
}

void tahoe_free_host(void *p) { free(p); }

void tahoe_synth_forest(tahoe_dense_node *nodes, int num_trees, int depth, int num_cols, uint64_t seed,
                        float leaf_prob)
{
    const size_t per_tree = (size_t)tahoe_tree_num_nodes(depth);
    const size_t first_bottom = ((size_t)1 << depth) - 1;
    const uint64_t cols = num_cols > 0 ? (uint64_t)num_cols : 1;
    for (size_t t = 0; t < (size_t)num_trees; ++t) {
        for (size_t j = 0; j < per_tree; ++j) {
            const uint64_t g = (t * per_tree + j) * 4;
            const uint64_t x0 = splitmix64_at(seed, g), x1 = splitmix64_at(seed, g + 1),
                           x2 = splitmix64_at(seed, g + 2), x3 = splitmix64_at(seed, g + 3);
            const int fid = (int)(x0 % cols);
            const float value = 2.0f * u01(x1) - 1.0f;
            const bool bottom = j >= first_bottom;
            const bool is_leaf = bottom || (u01(x3) < leaf_prob);
            tahoe_encode_node(&nodes[t * per_tree + j], is_leaf ? 0 : fid, value, (int)(x2 & 1), u01(x2), is_leaf);
        }
    }
}

void tahoe_synth_data(float *out, size_t first_row, size_t rows, int num_cols, uint64_t seed, float missing_prob,
                      float missing, float nan_prob)
{
    const size_t cols = (size_t)num_cols;
    for (size_t r = 0; r < rows; ++r) {
        for (size_t c = 0; c < cols; ++c) {
            const uint64_t e = ((first_row + r) * cols + c) * 2;
            float v = 2.0f * u01(splitmix64_at(seed, e)) - 1.0f;
            if (missing_prob > 0.0f || nan_prob > 0.0f) {
                const float p = u01(splitmix64_at(seed, e + 1));
                if (p < missing_prob)
                    v = missing;
                else if (p < missing_prob + nan_prob)
                    v = std::numeric_limits<float>::quiet_NaN();
            }
            out[r * cols + c] = v;
        }
    }
}

}  // extern "C"
