// main.cpp — `unikmer`-compatible command line driver over the C ABI of libunikmer_hip.so.
//
// The reference's L3/L4 layers (cobra sub-commands in /root/reference/unikmer/cmd/*.go) are
// Go; there is no Go toolchain in this image, so the host side of the drop-in is C++ (the
// INTEGRATION.md cgo shim documents the Go binding).  Commands keep the reference's names,
// flags and output-naming rules; all k-mer arithmetic that the reference does per record on
// the CPU goes through the HIP library:
//   count  (count.go)   FASTA/Q -> ukm_encode_kmers | ukm_nthash -> ukm_sort_* -> ukm_unique
//   sort   (sort.go)    ukm_sort_u64 | ukm_sort_pairs -> ukm_unique(-u/-d)
//   union / inter / diff / common / merge -> ukm_union / ukm_inter / ukm_diff / ukm_common / ukm_merge_k
// CPU-only commands (no GPU needed): view, dump, num, info/stats, concat, head, encode, decode.
// `count` keeps the window values on the device from encode to the final set (chunked, double-buffered upload).
// Not implemented (SURVEY.md §2a out of scope): grep, filter, rfilter, tsplit, locate, map, sample,
// autocompletion; count -S (syncmer sketch: third-party rule not reconstructable from the tree).
#include <dirent.h>
#include <getopt.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cerrno>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <ctime>
#include <deque>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <regex>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/unikmer_hip.h"
#include "unik.hpp"

using std::string;
using std::vector;
typedef uint64_t u64;
typedef uint32_t u32;

static const char *VERSION = "0.21.0-hip";
static const string EXT = ".unik";

// ---- errors / logging (util-cli.go:39-44 checkError -> log + os.Exit(-1)) -------------------------
// A worker thread (the FASTA/Q parser of `count`) must not call exit(): atexit handlers and static destructors --
// the HIP runtime's among them -- would run while the main thread is inside HIP calls.  There die() throws; the
// thread's catch hands the message to the main thread, which reports it after join() (round-2 advice).
static thread_local bool t_worker_thread = false;
[[noreturn]] static void die(const char *fmt, ...) {
    char buf[2048];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (t_worker_thread) throw std::runtime_error(buf);
    fprintf(stderr, "[ERRO] %s\n", buf);
    // _exit, not exit: a fatal error on the main thread may come while the parser thread of `count` is still filling
    // page-locked chunks; exit() would run atexit handlers and static destructors (the HIP runtime's among them) under it
    fflush(nullptr);
    _exit(255);
}
static bool g_verbose = false;
static void info(const char *fmt, ...) {
    if (!g_verbose) return;
    va_list ap;
    va_start(ap, fmt);
    fprintf(stderr, "[INFO] ");
    vfprintf(stderr, fmt, ap);
    fprintf(stderr, "\n");
    va_end(ap);
}

static void warn(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    fprintf(stderr, "[WARN] ");
    vfprintf(stderr, fmt, ap);
    fprintf(stderr, "\n");
    va_end(ap);
}

// ---- tiny flag parser -----------------------------------------------------------------------------
struct FlagSpec {
    char shortname;  // 0 = none
    const char *longname;
    bool takes_value;
};
struct Args {
    std::map<string, string> val;
    std::set<string> present;
    vector<string> files;
    bool has(const string &k) const { return present.count(k) > 0; }
    string str(const string &k, const string &d = "") const { auto it = val.find(k); return it == val.end() ? d : it->second; }
    long long num(const string &k, long long d) const { auto it = val.find(k); return it == val.end() ? d : atoll(it->second.c_str()); }
    double real(const string &k, double d) const { auto it = val.find(k); return it == val.end() ? d : atof(it->second.c_str()); }
};

static const vector<FlagSpec> GLOBAL_FLAGS = {
    {'j', "threads", true}, {0, "verbose", false}, {'C', "no-compress", false}, {0, "compression-level", true},
    {'c', "compact", false}, {'i', "infile-list", true}, {0, "max-taxid", true}, {'I', "ignore-taxid", false},
    {0, "data-dir", true}, {0, "skip-flag-check", false}, {0, "skip-file-check", false}, {0, "gpu", true},
    {'h', "help", false},
};

static Args parse_args(int argc, char **argv, vector<FlagSpec> specs) {
    specs.insert(specs.end(), GLOBAL_FLAGS.begin(), GLOBAL_FLAGS.end());
    Args a;
    for (int i = 0; i < argc; i++) {
        string s = argv[i];
        if (s == "-" || s.empty() || s[0] != '-') { a.files.push_back(s); continue; }
        if (s == "--") { for (int j = i + 1; j < argc; j++) a.files.push_back(argv[j]); break; }
        if (s[1] == '-') {
            string name = s.substr(2), value;
            bool has_eq = false;
            size_t eq = name.find('=');
            if (eq != string::npos) { value = name.substr(eq + 1); name = name.substr(0, eq); has_eq = true; }
            const FlagSpec *f = nullptr;
            for (auto &sp : specs) if (name == sp.longname) f = &sp;
            if (!f) die("unknown flag: --%s", name.c_str());
            a.present.insert(f->longname);
            if (f->takes_value) {
                if (!has_eq) { if (i + 1 >= argc) die("flag needs an argument: --%s", name.c_str()); value = argv[++i]; }
                a.val[f->longname] = value;
            }
        } else {
            for (size_t p = 1; p < s.size(); p++) {
                const FlagSpec *f = nullptr;
                for (auto &sp : specs) if (sp.shortname && sp.shortname == s[p]) f = &sp;
                if (!f) die("unknown shorthand flag: '%c' in %s", s[p], s.c_str());
                a.present.insert(f->longname);
                if (f->takes_value) {
                    string value = s.substr(p + 1);
                    if (value.empty()) { if (i + 1 >= argc) die("flag needs an argument: -%c", s[p]); value = argv[++i]; }
                    a.val[f->longname] = value;
                    break;
                }
            }
        }
    }
    return a;
}

// ---- options shared by all commands (util.go:52-109) -----------------------------------------------
struct Options {
    bool compress = true, compact = false, ignore_taxid = false, skip_file_check = false;
    int level = -1, gpu = 0;
    u32 max_taxid = 0xFFFFFFFFu;
    string data_dir;
};
static Options get_options(const Args &a) {
    Options o;
    g_verbose = a.has("verbose");
    o.compress = !a.has("no-compress");
    o.level = (int)a.num("compression-level", -1);
    o.compact = a.has("compact");
    o.ignore_taxid = a.has("ignore-taxid");
    o.skip_file_check = a.has("skip-file-check");
    o.max_taxid = (u32)a.num("max-taxid", 0xFFFFFFFFLL);
    const char *env = getenv("UNIKMER_DB");  // util.go:75-83
    const char *home = getenv("HOME");
    o.data_dir = a.has("data-dir") ? a.str("data-dir") : (env ? string(env) : (string(home ? home : ".") + "/.unikmer"));
    const char *g = getenv("UNIKMER_GPU");
    o.gpu = (int)a.num("gpu", g ? atoi(g) : 0);
    return o;
}

static bool file_exists(const string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }

// util-cli.go:192-264 : files from the command line plus the optional list file; "-" = stdin
static vector<string> get_files(const Args &a, const Options &o, bool allow_stdin_default = true) {
    vector<string> files = a.files;
    if (a.has("infile-list")) {
        std::ifstream fh(a.str("infile-list"));
        if (!fh) die("fail to read file list: %s", a.str("infile-list").c_str());
        string line;
        while (std::getline(fh, line)) {
            while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
            if (!line.empty()) files.push_back(line);
        }
    }
    if (files.empty() && allow_stdin_default) files.push_back("-");
    if (!o.skip_file_check)
        for (auto &f : files)
            if (f != "-" && !file_exists(f)) die("file does not exist: %s", f.c_str());
    return files;
}
static string out_name(const string &prefix) {  // union.go:79-81
    if (prefix == "-") return prefix;
    if (prefix.size() >= EXT.size() && prefix.compare(prefix.size() - EXT.size(), EXT.size(), EXT) == 0) return prefix;
    return prefix + EXT;
}

// ---- GPU context + device buffers ---------------------------------------------------------------
struct Gpu {
    ukm_ctx *c = nullptr;
    explicit Gpu(int device) {
        if (ukm_ctx_create(device, &c) != UKM_OK) die("%s", ukm_last_error());
    }
    ~Gpu() { if (c) ukm_ctx_destroy(c); }
};
static void ck(int rc) { if (rc != UKM_OK) die("%s", ukm_last_error()); }

// ---- taxonomy (util.go:119-171): nodes.dmp (+ merged.dmp) -> ukm_taxonomy_load ---------------------
static bool parse_two_ids(const string &line, u32 &a, u32 &b) {
    const char *p = line.c_str();
    char *e = nullptr;
    unsigned long x = strtoul(p, &e, 10);
    if (e == p) return false;
    const char *q = strchr(e, '|');
    if (!q) return false;
    q++;
    unsigned long y = strtoul(q, &e, 10);
    if (e == q) return false;
    a = (u32)x; b = (u32)y;
    return true;
}
static u32 load_taxonomy(Gpu &g, const Options &o) {
    const string nodes = o.data_dir + "/nodes.dmp", merged = o.data_dir + "/merged.dmp";
    if (!file_exists(nodes))
        die("taxonomy file not found: %s (set --data-dir or UNIKMER_DB)", nodes.c_str());
    info("loading Taxonomy from: %s", o.data_dir.c_str());
    vector<u32> child, parent, mo, mn;
    std::ifstream fh(nodes);
    string line;
    u32 a, b;
    while (std::getline(fh, line)) if (parse_two_ids(line, a, b)) { child.push_back(a); parent.push_back(b); }
    if (file_exists(merged)) {
        std::ifstream mh(merged);
        while (std::getline(mh, line)) if (parse_two_ids(line, a, b)) { mo.push_back(a); mn.push_back(b); }
    }
    info("%zu nodes loaded, %zu merged nodes loaded", child.size(), mo.size());
    ck(ukm_taxonomy_load(g.c, child.data(), parent.data(), child.size(), mo.empty() ? nullptr : mo.data(),
                         mn.empty() ? nullptr : mn.data(), mo.size()));
    u32 mx = 0;
    ck(ukm_taxonomy_max_taxid(g.c, &mx));
    return mx;  // util.go:169: opt.MaxTaxid follows the taxonomy
}

// ---- one loaded .unik file ---------------------------------------------------------------------------
struct Loaded {
    unik::Header h;
    vector<u64> codes;
    vector<u32> taxids;  // filled when has_taxid and the records carry their own (UnikIncludeTaxID)
    // the header's global taxid (`count -t`, count.go:466-468) when the records carry none: unik.Reader hands it out with
    // every record; the library takes it as ONE number per file (ukm_*_ft) -- nothing is expanded on the host
    u32 file_taxid = 0;
    bool has_taxid = false;
    bool per_record() const { return has_taxid && h.is_include_taxid(); }
    u32 taxid_of(size_t i) const { return per_record() ? taxids[i] : file_taxid; }
};
static Loaded load_unik(const string &file, const Options &o) {
    unik::Reader r(file);
    Loaded L;
    L.h = r.h;
    L.has_taxid = !o.ignore_taxid && r.h.has_taxid_info();
    r.read_all(L.codes, L.per_record() ? &L.taxids : nullptr);
    if (L.has_taxid && !L.per_record()) L.file_taxid = r.h.global_taxid;
    return L;
}
static void check_compat(const unik::Header &a, const unik::Header &b, const string &file) {  // util-binary-file.go:31-44
    if (a.k != b.k) die("k-mer length not consistent (%d != %d), please check with \"unikmer stats\": %s", a.k, b.k, file.c_str());
    if (a.is_canonical() != b.is_canonical()) die("'canonical' flags not consistent, please check with \"unikmer stats\": %s", file.c_str());
    if (a.is_hashed() != b.is_hashed()) die("'hashed' flags not consistent, please check with \"unikmer stats\": %s", file.c_str());
    if (a.is_scaled() != b.is_scaled()) die("'scaled' flags not consistent, please check with \"unikmer stats\": %s", file.c_str());
}

static void write_unik(const string &out_file, const Options &o, int k, u32 mode, u32 max_taxid, u32 global_taxid,
                       const unik::Header *scale_from, const u64 *codes, const u32 *taxids, u64 n) {
    unik::OutStream os(out_file, o.compress, o.level);
    unik::Writer w(os, k, mode);
    w.set_max_taxid(max_taxid);
    if (global_taxid) w.set_global_taxid(global_taxid);
    if (scale_from && scale_from->is_scaled()) w.set_scale(scale_from->scale, scale_from->max_hash);
    w.set_number(n);
    const bool tx = (mode & unik::UnikIncludeTaxID) != 0;
    for (u64 i = 0; i < n; i++) {
        if (tx) w.write_code_with_taxid(codes[i], taxids[i]);
        else w.write_code(codes[i]);
    }
    w.flush();
    os.close();
    info("%llu k-mers saved to %s", (unsigned long long)n, out_file.c_str());
}

// ---- out-of-core protocol of sort -m / split / merge (sort.go:241-447, split.go:288-401, merge.go:228-341) ----
// util.go:291-335 ParseByteSize: plain number, or B/K/M/G suffix (powers of 1024); "" = 0
static long long parse_byte_size(string v) {
    while (!v.empty() && strchr(" \t\r\n", v.back())) v.pop_back();
    while (!v.empty() && strchr(" \t\r\n", v.front())) v.erase(v.begin());
    if (v.empty()) return 0;
    double unit = 0;
    switch (v.back()) {
    case 'B': case 'b': unit = 1; break;
    case 'K': case 'k': unit = 1 << 10; break;
    case 'M': case 'm': unit = 1 << 20; break;
    case 'G': case 'g': unit = 1 << 30; break;
    default: break;
    }
    if (unit != 0) {
        v.pop_back();
        if (v.empty()) return 0;
    } else {
        unit = 1;
    }
    char *e = nullptr;
    const double x = strtod(v.c_str(), &e);
    if (e == v.c_str() || *e) die("parsing byte size: invalid byte size: %s", v.c_str());
    return x < 0 ? 0 : (long long)(x * unit);
}
static bool dir_exists(const string &p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
static vector<string> list_dir(const string &d) {
    vector<string> names;
    DIR *dh = opendir(d.c_str());
    if (!dh) die("check given directory '%s': %s", d.c_str(), strerror(errno));
    while (struct dirent *e = readdir(dh)) { string n = e->d_name; if (n != "." && n != "..") names.push_back(n); }
    closedir(dh);
    std::sort(names.begin(), names.end());
    return names;
}
// sort.go:113-137 / split.go:108-126: a non-empty directory needs --force; it is then emptied
static void prepare_dir(const string &d, bool force, const char *what) {
    if (dir_exists(d)) {
        vector<string> names = list_dir(d);
        if (!names.empty() && !force) die("%s not empty: %s, choose another one or use --force to overwrite", what, d.c_str());
        for (auto &n : names) if (remove((d + "/" + n).c_str()) != 0) die("fail to remove %s/%s", d.c_str(), n.c_str());
    } else if (mkdir(d.c_str(), 0777) != 0) {
        // parents (mkdir -p)
        string acc;
        for (size_t i = 0; i <= d.size(); i++)
            if (i == d.size() || d[i] == '/') { acc = d.substr(0, i); if (!acc.empty() && !dir_exists(acc) && mkdir(acc.c_str(), 0777) != 0) die("fail to create directory: %s", acc.c_str()); }
    }
}
static string chunk_file_name(const string &dir, int i) {  // util-sort.go:192-194
    char b[64];
    snprintf(b, sizeof b, "chunk_%03d", i);
    return dir + "/" + b + EXT;
}
static string base_of(const string &p) { size_t s = p.find_last_of('/'); return s == string::npos ? p : p.substr(s + 1); }

struct ChunkJob {
    int k = 0;
    u32 mode = 0;  // includes UnikSorted
    bool tax = false, uniq = false, rep = false;
    int key_bits = 64;
    u32 max_taxid = 0xFFFFFFFFu;
    const unik::Header *h0 = nullptr;
};
// one chunk: device sort + the scan of dumpCodes2File / dumpCodesTaxids2File (util-sort.go:35-190):
// -u collapses runs (LCA of their taxids), -d writes every code once and repeated ones twice, else plain
static u64 sort_chunk_to_file(Gpu &g, const Options &o, const ChunkJob &j, vector<u64> &codes, vector<u32> &taxids, const string &file) {
    const u64 n = codes.size();
    vector<u64> out(2 * n + 1);
    vector<u32> tout(j.tax ? 2 * n + 1 : 0);
    u64 m = 0;
    if (n) {
        if (j.tax) ck(ukm_sort_pairs(g.c, codes.data(), taxids.data(), n, j.key_bits));
        else ck(ukm_sort_u64(g.c, codes.data(), n, j.key_bits));
        ck(ukm_unique(g.c, codes.data(), j.tax ? taxids.data() : nullptr, n, j.uniq ? UKM_UNIQUE : (j.rep ? UKM_REPEATED_CHUNK : UKM_PLAIN),
                      out.data(), j.tax ? tout.data() : nullptr, 2 * n + 1, &m));
    }
    write_unik(file, o, j.k, j.mode, j.max_taxid, 0, j.h0, out.data(), j.tax ? tout.data() : nullptr, m);
    return m;
}

// kmers v0.1.0 text <-> code on the host (view / dump / encode / decode only; the throughput
// path is ukm_encode_kmers)
static int base2bit(unsigned char c) {
    switch (c) {
    case 'A': case 'a': case 'N': case 'n': case 'M': case 'm': case 'V': case 'v': case 'H': case 'h':
    case 'R': case 'r': case 'D': case 'd': case 'W': case 'w': return 0;
    case 'C': case 'c': case 'S': case 's': case 'B': case 'b': case 'Y': case 'y': return 1;
    case 'G': case 'g': case 'K': case 'k': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return 4;
    }
}
static bool encode_kmer(const string &s, u64 &code) {
    if (s.empty() || s.size() > 32) return false;
    u64 c = 0;
    for (unsigned char ch : s) { int b = base2bit(ch); if (b > 3) return false; c = (c << 2) | (u64)b; }
    code = c;
    return true;
}
static u64 revcomp(u64 code, int k) {
    u64 c = ~code, r = 0;
    for (int i = 0; i < k; i++) { r = (r << 2) | (c & 3); c >>= 2; }
    return r;
}
static string decode_kmer(u64 code, int k) {
    string s((size_t)k, 'A');
    for (int i = k - 1; i >= 0; i--) { s[(size_t)i] = "ACGT"[code & 3]; code >>= 2; }
    return s;
}

// ---- FASTA/Q reader (bio/seqio/fastx as used by count.go:289-299) ----------------------------------
struct SeqBatch {
    vector<uint8_t> bases;
    vector<u64> off{0};
    vector<string> names;
};
// FASTA/Q (+gzip) parser (bio/seqio/fastx, count.go:289-299).  The sink decides per record whether it is kept
// (begin_record) and receives the sequence line by line; blanks inside sequence lines are dropped as
// bio/fastx does.
struct FastxSink {
    virtual ~FastxSink() {}
    virtual bool begin_record(const string &name) = 0;  // false: skip this record
    virtual void add_seq(const char *p, size_t n) = 0;
    virtual void end_record() = 0;
};
static void parse_fastx(const string &file, FastxSink &sink) {
    unik::InStream in(file);
    vector<char> buf(1 << 20);
    bool have = false, keep = false, fastq = false;
    int fq_state = 0;  // 0 header, 1 seq, 3 qual
    u64 seq_len = 0, qual_len = 0;
    auto finish = [&]() {
        if (have && keep) sink.end_record();
        have = false;
    };
    auto seq_line = [&](const string &l) -> size_t {
        size_t cnt = 0, s0 = 0;
        for (size_t i = 0; i <= l.size(); i++) {
            if (i == l.size() || l[i] == ' ' || l[i] == '\t') {
                if (i > s0) { if (keep) sink.add_seq(l.data() + s0, i - s0); cnt += i - s0; }
                s0 = i + 1;
            }
        }
        return cnt;
    };
    string carry;
    auto handle = [&](const string &l) {
        if (fastq) {
            if (fq_state == 0) { if (l.empty()) return; if (l[0] != '@') die("invalid FASTQ record in %s", file.c_str()); finish(); have = true; keep = sink.begin_record(l.substr(1)); seq_len = qual_len = 0; fq_state = 1; }
            else if (fq_state == 1) { if (!l.empty() && l[0] == '+') fq_state = 3; else seq_len += seq_line(l); }
            else if (fq_state == 3) { qual_len += l.size(); if (qual_len >= seq_len) fq_state = 0; }
            return;
        }
        if (!l.empty() && l[0] == '>') { finish(); have = true; keep = sink.begin_record(l.substr(1)); return; }
        if (!have) { if (l.empty()) return; die("invalid FASTA/Q format: %s", file.c_str()); }
        seq_line(l);
    };
    bool first = true;
    for (;;) {
        size_t n = in.read(buf.data(), buf.size());
        if (n == 0) break;
        size_t s0 = 0;
        for (size_t i = 0; i < n; i++) {
            if (buf[i] == '\n') {
                carry.append(buf.data() + s0, i - s0);
                if (!carry.empty() && carry.back() == '\r') carry.pop_back();
                if (first) { first = false; fastq = !carry.empty() && carry[0] == '@'; }
                handle(carry);
                carry.clear();
                s0 = i + 1;
            }
        }
        carry.append(buf.data() + s0, n - s0);
    }
    if (!carry.empty()) { if (first) fastq = carry[0] == '@'; handle(carry); }
    finish();
}
struct BatchSink : FastxSink {
    SeqBatch &b;
    bool keep_names;
    BatchSink(SeqBatch &bb, bool kn) : b(bb), keep_names(kn) {}
    bool begin_record(const string &name) override { if (keep_names) b.names.push_back(name); return true; }
    void add_seq(const char *p, size_t n) override { b.bases.insert(b.bases.end(), p, p + n); }
    void end_record() override { b.off.push_back(b.bases.size()); }
};
static void read_fastx(const string &file, SeqBatch &b, bool keep_names) {
    BatchSink sink(b, keep_names);
    parse_fastx(file, sink);
}

// ---- count: device-resident pipeline ------------------------------------------------------------------------
// FASTA/Q bases go to the GPU in record-aligned chunks through two page-locked staging buffers: chunk i+1 is
// uploaded on the context's transfer stream while chunk i is encoded / hashed (count.go:285-299 reads record
// by record; here a "record batch" is a chunk).  The window values never come back to the host: they are
// written to ONE device array, sorted and reduced there (count.go:424-436, 571-581), and only the final set
// crosses PCIe.  Returns the result in `codes`.
struct DevMem {
    ukm_ctx *c;
    void *p = nullptr;
    DevMem(ukm_ctx *ctx, u64 bytes) : c(ctx) { ck(ukm_dev_alloc(c, bytes ? bytes : 8, &p)); }
    ~DevMem() { if (p) ukm_dev_free(c, p); }
    DevMem(const DevMem &) = delete;
};
struct PinMem {
    ukm_ctx *c;
    void *p = nullptr;
    PinMem(ukm_ctx *ctx, u64 bytes) : c(ctx) { ck(ukm_host_alloc(c, bytes ? bytes : 8, &p)); }
    ~PinMem() { if (p) ukm_host_free(c, p); }
    PinMem(const PinMem &) = delete;
};
static double now_ms() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
// One staging chunk: page-locked bases + record boundaries (relative to the chunk).
struct HostChunk {
    ukm_ctx *c = nullptr;
    uint8_t *bases = nullptr;
    u64 cap = 0, nb = 0;
    vector<u64> off{0};
    vector<u32> wtax;  // -T: the taxid of every window of the chunk, in window order (count.go:334-344)
    void reserve(u64 want) {  // grows the page-locked buffer (a record longer than a chunk)
        if (want <= cap) return;
        u64 ncap = std::max<u64>(want, cap + cap / 2);
        void *p = nullptr;
        ck(ukm_host_alloc(c, ncap, &p));
        if (nb) memcpy(p, bases, nb);
        if (bases) ukm_host_free(c, bases);
        bases = (uint8_t *)p;
        cap = ncap;
    }
    void reset() { nb = 0; off.assign(1, 0); wtax.clear(); }
};
// parser thread -> device thread hand-off: three chunks go round (one being filled, one travelling, one in the kernels)
struct ChunkPipe {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<int> full, free_;
    bool done = false;
    string error;
    int get_free() { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return !free_.empty(); }); int i = free_.front(); free_.pop_front(); return i; }
    void put_full(int i) { { std::lock_guard<std::mutex> l(mu); full.push_back(i); } cv.notify_all(); }
    int get_full() { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return !full.empty() || done; }); if (full.empty()) return -1; int i = full.front(); full.pop_front(); return i; }
    void put_free(int i) { { std::lock_guard<std::mutex> l(mu); free_.push_back(i); } cv.notify_all(); }
    void finish() { { std::lock_guard<std::mutex> l(mu); done = true; } cv.notify_all(); }
};
struct ChunkSink : FastxSink {
    ChunkPipe &pipe;
    HostChunk *ch;
    u64 chunk_bytes;
    int cur;
    const std::regex *skip_name;
    // -T: the taxid is parsed from the header when the record ENDS, and only if the record is long enough to have a
    // window (count.go:323-344: ErrShortSeq is hit before the header is looked at); every window gets its record's taxid
    const std::regex *tax_re;
    int k;
    bool circular;
    string cur_name;
    u64 rec_start = 0;
    ChunkSink(ChunkPipe &p, HostChunk *c, u64 cb, const std::regex *skip, const std::regex *tre = nullptr, int kk = 0, bool circ = false)
        : pipe(p), ch(c), chunk_bytes(cb), skip_name(skip), tax_re(tre), k(kk), circular(circ) { cur = pipe.get_free(); ch[cur].reset(); }
    bool begin_record(const string &name) override {
        if (skip_name && std::regex_search(name, *skip_name)) return false;  // -B (count.go:300-312)
        if (tax_re) cur_name = name;
        rec_start = ch[cur].nb;
        return true;
    }
    void add_seq(const char *p, size_t n) override {
        HostChunk &h = ch[cur];
        h.reserve(h.nb + n);
        memcpy(h.bases + h.nb, p, n);
        h.nb += n;
    }
    void end_record() override {
        HostChunk &h = ch[cur];
        h.off.push_back(h.nb);
        if (tax_re) {
            const u64 len = h.nb - rec_start;
            if (len >= (u64)k) {
                std::smatch m;
                if (!std::regex_search(cur_name, m, *tax_re) || m.size() < 2) die("failed to parse taxid in header: %s", cur_name.c_str());
                const u32 t = (u32)strtoul(m[1].str().c_str(), nullptr, 10);
                h.wtax.insert(h.wtax.end(), circular ? len : len - (u64)k + 1, t);
            }
        }
        if (h.nb >= chunk_bytes) flush();
    }
    void flush() {
        if (ch[cur].off.size() > 1) { pipe.put_full(cur); cur = pipe.get_free(); ch[cur].reset(); }
    }
};

// device array that grows (the number of windows is unknown while the file is still being read)
struct DevArray {
    ukm_ctx *c;
    u64 *p = nullptr;
    u64 cap = 0;
    explicit DevArray(ukm_ctx *ctx) : c(ctx) {}
    ~DevArray() { if (p) ukm_dev_free(c, p); }
    void ensure(u64 used, u64 want) {
        if (want <= cap) return;
        const u64 ncap = std::max<u64>(want, cap * 2);
        void *q = nullptr;
        ck(ukm_dev_alloc(c, ncap * 8, &q));
        if (used) ck(ukm_copy(c, q, p, used * 8));
        if (p) ukm_dev_free(c, p);
        p = (u64 *)q;
        cap = ncap;
    }
};

static u64 count_on_device(Gpu &g, int device, const vector<string> &files, const std::regex *skip_name, int k, bool canonical, bool circular,
                           bool hashed, u64 max_hash, bool linear, int uniq_mode, int key_bits, vector<u64> &codes,
                           int minimizer_w = 0, const std::regex *tax_re = nullptr, vector<u32> *taxids_out = nullptr) {
    const char *ce = getenv("UNIKMER_CHUNK_MB");
    const u64 CH = (u64)(ce ? std::max(1, atoi(ce)) : 32) << 20;
    const double t0 = now_ms();
    Gpu gp(device);  // the parser thread's own context (page-locked allocations only)
    HostChunk ch[3];
    ChunkPipe pipe;
    for (int i = 0; i < 3; i++) { ch[i].c = gp.c; ch[i].reserve(CH + (1 << 20)); pipe.free_.push_back(i); }
    std::thread parser([&]() {
        t_worker_thread = true;  // die() throws here instead of exiting under the main thread's HIP calls
        try {
            ChunkSink sink(pipe, ch, CH, skip_name, tax_re, k, circular);
            for (auto &f : files) { info("reading sequence file: %s", f.c_str()); parse_fastx(f, sink); }
            sink.flush();
        } catch (const std::exception &e) {
            std::lock_guard<std::mutex> l(pipe.mu);
            pipe.error = e.what();
        }
        pipe.finish();
    });
    // device side: chunk i+1 travels (transfer stream) while chunk i is encoded / hashed
    struct Slot { void *bases = nullptr; u64 bcap = 0; void *off = nullptr; u64 ocap = 0; void *tax = nullptr; u64 tcap = 0; int host = -1; u64 nrec = 0; };
    Slot sl[2];
    DevArray dcodes(g.c);
    // -T: the windows' taxids travel with their chunk and are appended to a second growing device array (u32, kept in
    // a u64-granular DevArray: capacity counts pairs of taxids)
    DevArray dtax(g.c);
    const bool with_tax = tax_re != nullptr;
    u64 n = 0, total_bases = 0, nchunks = 0;
    auto upload = [&](Slot &d, int hi) {
        HostChunk &h = ch[hi];
        if (h.nb > d.bcap) { if (d.bases) ukm_dev_free(g.c, d.bases); d.bcap = h.nb + h.nb / 8; ck(ukm_dev_alloc(g.c, d.bcap, &d.bases)); }
        const u64 ob = h.off.size() * 8;
        if (ob > d.ocap) { if (d.off) ukm_dev_free(g.c, d.off); d.ocap = ob + ob / 8; ck(ukm_dev_alloc(g.c, d.ocap, &d.off)); }
        ck(ukm_copy_async(g.c, d.bases, h.bases, h.nb));
        ck(ukm_copy_async(g.c, d.off, h.off.data(), ob));   // (pageable source: staged by the runtime)
        if (with_tax && !h.wtax.empty()) {
            const u64 tb = h.wtax.size() * 4;
            if (tb > d.tcap) { if (d.tax) ukm_dev_free(g.c, d.tax); d.tcap = tb + tb / 8; ck(ukm_dev_alloc(g.c, d.tcap, &d.tax)); }
            ck(ukm_copy_async(g.c, d.tax, h.wtax.data(), tb));
        }
        d.host = hi;
        d.nrec = h.off.size() - 1;
        total_bases += h.nb;
        nchunks++;
    };
    auto compute = [&](Slot &d) {
        HostChunk &h = ch[d.host];
        dcodes.ensure(n, n + h.nb);  // windows <= bases
        u64 m = 0;
        if (minimizer_w > 0) ck(ukm_minimizer(g.c, (const uint8_t *)d.bases, (const u64 *)d.off, d.nrec, k, minimizer_w, circular, max_hash, dcodes.p + n, nullptr, dcodes.cap - n, &m));
        else if (hashed) ck(ukm_nthash(g.c, (const uint8_t *)d.bases, (const u64 *)d.off, d.nrec, k, canonical, circular, max_hash, dcodes.p + n, dcodes.cap - n, &m));
        else ck(ukm_encode_kmers(g.c, (const uint8_t *)d.bases, (const u64 *)d.off, d.nrec, k, canonical, circular, dcodes.p + n, dcodes.cap - n, &m));
        if (with_tax) {
            if (m != h.wtax.size()) die("count -T: %llu windows but %zu taxids in a chunk", (unsigned long long)m, h.wtax.size());
            dtax.ensure((n + 1) / 2, (n + m + 1) / 2 + 1);
            if (m) ck(ukm_copy(g.c, (u32 *)dtax.p + n, d.tax, m * 4));  // device to device, ordered behind the upload by the fence above
        }
        n += m;
        pipe.put_free(d.host);  // its upload is complete (the kernels waited for it): the parser may refill it
        d.host = -1;
    };
    int cur = 0;
    int hi = pipe.get_full();
    if (hi >= 0) upload(sl[cur], hi);
    while (hi >= 0) {
        ck(ukm_copy_fence(g.c));            // kernels of the current chunk start after its upload
        const int nxt = pipe.get_full();    // (blocks while the parser is still filling it)
        if (nxt >= 0) upload(sl[cur ^ 1], nxt);
        compute(sl[cur]);
        cur ^= 1;
        hi = nxt;
    }
    parser.join();
    ck(ukm_copy_sync(g.c));
    for (auto &d : sl) { if (d.bases) ukm_dev_free(g.c, d.bases); if (d.off) ukm_dev_free(g.c, d.off); if (d.tax) ukm_dev_free(g.c, d.tax); }
    for (auto &h : ch) if (h.bases) ukm_host_free(gp.c, h.bases);
    if (!pipe.error.empty()) die("%s", pipe.error.c_str());
    const double t1 = now_ms();
    codes.clear();
    u64 nout = n;
    if (!linear && n) {
        float ms_sort = 0, ms_uniq = 0;
        if (with_tax) ck(ukm_sort_pairs(g.c, dcodes.p, (u32 *)dtax.p, n, key_bits));
        else ck(ukm_sort_u64(g.c, dcodes.p, n, key_bits));
        ukm_last_call_ms(g.c, &ms_sort);
        const double t1b = now_ms();
        DevMem dout(g.c, n * 8);
        DevMem dtout(g.c, with_tax ? n * 4 : 8);
        const double t1c = now_ms();
        ck(ukm_unique(g.c, dcodes.p, with_tax ? (u32 *)dtax.p : nullptr, n, uniq_mode, (u64 *)dout.p, with_tax ? (u32 *)dtout.p : nullptr, n, &nout));
        ukm_last_call_ms(g.c, &ms_uniq);
        if (with_tax && taxids_out) {
            taxids_out->resize(nout ? nout : 1);
            ck(ukm_copy(g.c, taxids_out->data(), dtout.p, nout * 4));
            taxids_out->resize(nout);
        }
        const double t2 = now_ms();
        info("  sort: %.2f ms wall (%.2f ms on the device), output allocation %.2f ms, unique: %.2f ms wall (%.2f ms on the device)",
             t1b - t1, ms_sort, t1c - t1b, t2 - t1c, ms_uniq);
        codes.resize(nout ? nout : 1);
        ck(ukm_copy(g.c, codes.data(), dout.p, nout * 8));
        info("device pipeline: %llu bases in %llu chunk(s); parse+upload+encode %.2f ms (overlapped), sort+unique %.2f ms, download of %llu codes %.2f ms",
             (unsigned long long)total_bases, (unsigned long long)nchunks, t1 - t0, t2 - t1, (unsigned long long)nout, now_ms() - t2);
    } else {
        codes.resize(n ? n : 1);
        if (n) ck(ukm_copy(g.c, codes.data(), dcodes.p, n * 8));
        if (with_tax && taxids_out) {
            taxids_out->resize(n ? n : 1);
            if (n) ck(ukm_copy(g.c, taxids_out->data(), dtax.p, n * 4));
            taxids_out->resize(n);
        }
        info("device pipeline: %llu bases in %llu chunk(s); parse+upload+encode %.2f ms (overlapped), download of %llu codes %.2f ms",
             (unsigned long long)total_bases, (unsigned long long)nchunks, t1 - t0, (unsigned long long)n, now_ms() - t1);
    }
    codes.resize(nout);
    return nout;
}

// =================================================================================================
// count (count.go:41-602)
// =================================================================================================
static int cmd_count(int argc, char **argv) {
    Args a = parse_args(argc, argv, {{'o', "out-prefix", true}, {'k', "kmer-len", true}, {'K', "canonical", false},
                                     {'s', "sort", false}, {'t', "taxid", true}, {'T', "parse-taxid", false},
                                     {'r', "parse-taxid-regexp", true}, {'d', "repeated", false}, {'u', "unique", false},
                                     {'V', "more-verbose", false}, {'H', "hash", false}, {0, "circular", false},
                                     {'D', "scale", true}, {'W', "minimizer-w", true}, {'S', "syncmer-s", true},
                                     {'l', "linear", false}, {'B', "seq-name-filter", true}});
    Options o = get_options(a);
    vector<string> files = get_files(a, o);
    const int k = (int)a.num("kmer-len", 0);
    if (k <= 0) die("value of flag -k/--kmer-len should be positive");
    const bool canonical = a.has("canonical"), sortk = a.has("sort"), linear = a.has("linear");
    bool hashed = a.has("hash");
    const bool repeated = a.has("repeated"), unique = a.has("unique"), circular = a.has("circular");
    if (k > 32 && !hashed) { hashed = true; fprintf(stderr, "[WARN] flag -H/--hash is switched on for k > 32\n"); }  // count.go:81-84
    if (hashed && k > 64) die("k-mer size (%d) should be <=64", k);
    const long long scale = a.num("scale", 1);
    if (scale < 1 || scale > 0x7fffffffLL) die("value of flag --scale is too big");
    const bool scaled = scale > 1;
    if (scaled && !hashed) { hashed = true; fprintf(stderr, "[WARN] flag -H/--hash is switched on for scale > 1\n"); }
    const long long minimizer_w = a.num("minimizer-w", 0);
    if (minimizer_w < 0 || minimizer_w > 0x7fffffffLL) die("value of flag --minimizer-w is too big");
    const bool minimizer = minimizer_w > 0;
    if (minimizer) {  // count.go:100-113 (the sketch itself is always canonical; the -K file flag stays as given)
        if (!hashed) { hashed = true; fprintf(stderr, "[WARN] flag -H/--hash is switched on for minimizer-w > 1\n"); }
        if (!canonical) fprintf(stderr, "[WARN] flag -K/--canonical is switched on for minimizer-w > 1\n");
        if (k > 64) die("k-mer size (%d) should be <=64", k);
    }
    if (a.num("syncmer-s", 0) > 0) die("-S/--syncmer-s is not supported in this build");
    if (repeated && unique) die("flag -d/--repeated and -u/--unique are not compatible");
    const u32 gtaxid = (u32)a.num("taxid", 0);
    const bool parse_taxid = a.has("parse-taxid");
    if (parse_taxid && !a.has("parse-taxid-regexp")) die("flag -r/--parse-taxid-regexp needed when given flag -T/--parse-taxid");
    if (parse_taxid && gtaxid) die("flag -t/--taxid and -T/--parse-taxid can not given simultaneously");
    if (linear && (repeated || unique || sortk)) die("flag -l/--linear is not compatible with -s, -u and -d");
    const string out_file = out_name(a.str("out-prefix", "-"));

    // every mode goes through the device pipeline (round 3: also -T and -W): the parser thread fills page-locked
    // chunks -- with -T it also parses each record's taxid and expands it to the record's windows --, the device thread
    // uploads chunk i + 1 while chunk i is encoded / hashed / sketched, and sort + dedup run on the device
    if (parse_taxid && (scaled || minimizer)) die("-T/--parse-taxid together with -D/--scale or -W/--minimizer-w is not supported in this build");
    Gpu g(o.gpu);
    u32 max_taxid = o.max_taxid;
    if (parse_taxid) max_taxid = load_taxonomy(g, o);
    const u64 max_hash = scaled ? ukm_max_hash((u64)scale) : 0;
    // Scaled sketch: every kept hash is <= maxHash, so the radix sort needs only its significant bits
    // (scale 1000: 55 bits = 7 passes instead of 8)
    int key_bits = hashed ? 64 : 2 * k;
    if (hashed && max_hash != 0) { key_bits = 1; while (key_bits < 64 && (max_hash >> key_bits) != 0) key_bits++; }
    const int uniq_mode = unique ? UKM_SINGLETON : (repeated ? UKM_REPEATED : UKM_UNIQUE);  // count.go:424-436
    vector<u64> codes;
    vector<u32> taxids;
    std::regex re_skip, re_tax;
    if (a.has("seq-name-filter")) re_skip = std::regex(a.str("seq-name-filter"), std::regex::icase);
    if (parse_taxid) re_tax = std::regex(a.str("parse-taxid-regexp"));
    const u64 n = count_on_device(g, o.gpu, files, a.has("seq-name-filter") ? &re_skip : nullptr, k, canonical, circular, hashed, max_hash, linear,
                                  uniq_mode, key_bits, codes, minimizer ? (int)minimizer_w : 0, parse_taxid ? &re_tax : nullptr, &taxids);
    u32 mode = 0;
    if (canonical) mode |= unik::UnikCanonical;
    if (parse_taxid) mode |= unik::UnikIncludeTaxID;
    if (hashed) mode |= unik::UnikHashed;
    unik::Header sh;
    if (scaled) { sh.flag |= unik::UnikScaled; sh.scale = (u32)scale; sh.max_hash = max_hash; }
    if (!linear) {
        // without -s the reference writes Go-map order; any order is valid there, we keep the
        // sorted order but only set the Sorted flag (and its encoding) when -s is given
        if (sortk) mode |= unik::UnikSorted;
        else if (o.compact && !hashed) mode |= unik::UnikCompact;
    } else if (o.compact && !hashed) {
        mode |= unik::UnikCompact;
    }
    write_unik(out_file, o, k, mode, max_taxid, gtaxid, scaled ? &sh : nullptr, codes.data(), taxids.data(), n);
    return 0;
}

// =================================================================================================
// n-way commands: sort / union / inter / diff / common / merge
// =================================================================================================
struct Inputs {
    vector<Loaded> files;
    int k = 0;
    bool canonical = false, hashed = false, has_taxid = false, any_taxid = false;
    unik::Header h0;
    u64 total = 0;
};
static Inputs load_inputs(const vector<string> &files, const Options &o, bool require_sorted_all, bool require_sorted_first,
                          bool allow_mix) {
    Inputs in;
    for (size_t i = 0; i < files.size(); i++) {
        info("processing file (%zu/%zu): %s", i + 1, files.size(), files[i].c_str());
        in.files.push_back(load_unik(files[i], o));
        Loaded &L = in.files.back();
        if (i == 0) {
            in.h0 = L.h; in.k = L.h.k; in.canonical = L.h.is_canonical(); in.hashed = L.h.is_hashed();
            in.has_taxid = L.has_taxid;
        } else {
            check_compat(in.h0, L.h, files[i]);
            if (!allow_mix && !o.ignore_taxid && L.has_taxid != in.has_taxid)
                die(L.has_taxid ? "taxid information not found in previous files, but found in this: %s"
                                : "taxid information found in previous files, but missing in this: %s", files[i].c_str());
        }
        in.any_taxid |= L.has_taxid;
        if ((require_sorted_all || (require_sorted_first && i == 0)) && !L.h.is_sorted())
            die(require_sorted_all ? "input should be sorted: %s" : "the first file should be sorted: %s", files[i].c_str());
        in.total += L.codes.size();
    }
    return in;
}
struct Ptrs {
    vector<const u64 *> k;
    vector<const u32 *> t;
    vector<u32> ft;  // per-file taxids (ukm_*_ft): the global taxid of a file whose records carry none
    vector<u64> n;
};
static Ptrs ptrs_of(const Inputs &in, bool tax) {
    Ptrs p;
    for (auto &L : in.files) {
        p.k.push_back(L.codes.data());
        p.t.push_back(tax && L.per_record() ? L.taxids.data() : nullptr);
        p.ft.push_back(tax && L.has_taxid && !L.per_record() ? L.file_taxid : 0u);
        p.n.push_back(L.codes.size());
    }
    return p;
}
static u32 out_mode(const Inputs &in, bool sorted, bool tax, const Options &o) {
    u32 mode = 0;
    if (sorted) mode |= unik::UnikSorted;
    else if (o.compact && !in.hashed) mode |= unik::UnikCompact;
    if (in.canonical) mode |= unik::UnikCanonical;
    if (tax) mode |= unik::UnikIncludeTaxID;
    if (in.hashed) mode |= unik::UnikHashed;
    return mode;
}

// mergeChunksFile (util-sort.go:227-606) over sorted chunk files, on the device.  Files are loaded
// whole (HBM holds far more than the reference's open-file budget); when there are at least
// `max_open` of them the reference's two rounds are kept: groups of max_open files are merged with
// finalRound = false into new chunk files, then those are merged with finalRound = true.
static u64 merge_files_once(Gpu &g, const Options &o, const ChunkJob &j, const vector<string> &files, const string &out_file, bool final_round) {
    vector<Loaded> ls;
    u64 total = 0;
    Options oo = o;
    oo.ignore_taxid = !j.tax;
    for (auto &f : files) {
        ls.push_back(load_unik(f, oo));
        if (j.h0) check_compat(*j.h0, ls.back().h, f);
        if (!ls.back().h.is_sorted()) die("chunk file should be sorted: %s", f.c_str());
        if (j.tax && !ls.back().has_taxid) die("taxid information found in previous files, but missing in this: %s", f.c_str());
        total += ls.back().codes.size();
    }
    vector<const u64 *> pk; vector<const u32 *> pt; vector<u32> pf; vector<u64> pn;
    for (auto &L : ls) {
        pk.push_back(L.codes.data());
        pt.push_back(j.tax && L.per_record() ? L.taxids.data() : nullptr);
        pf.push_back(j.tax && !L.per_record() ? L.file_taxid : 0u);
        pn.push_back(L.codes.size());
    }
    const u64 cap = 2 * total + 1;
    vector<u64> out(cap);
    vector<u32> tout(j.tax ? cap : 0);
    u64 n = 0;
    if (!ls.empty())
        ck(ukm_merge_k_ft(g.c, pk.data(), j.tax ? pt.data() : nullptr, j.tax ? pf.data() : nullptr, pn.data(), (int)ls.size(),
                          j.uniq ? UKM_UNIQUE : (j.rep ? UKM_REPEATED : UKM_PLAIN), final_round ? 1 : 0, out.data(), j.tax ? tout.data() : nullptr, cap, &n));
    write_unik(out_file, o, j.k, j.mode, j.max_taxid, 0, j.h0, out.data(), j.tax ? tout.data() : nullptr, n);
    return n;
}
static u64 merge_rounds(Gpu &g, const Options &o, const ChunkJob &j, vector<string> files, const string &out_file, int max_open,
                        const string &tmp_dir, int &i_tmp, vector<string> &made) {
    if ((int)files.size() < max_open) {
        info("======= Stage 2: merging from %zu chunks =======", files.size());
        return merge_files_once(g, o, j, files, out_file, true);
    }
    info("======= Stage 2: merging from %zu chunks (round: 1/2) =======", files.size());
    vector<string> next, group;
    auto flush = [&]() {
        if (group.empty()) return;
        const string f = chunk_file_name(tmp_dir, ++i_tmp);
        info("[chunk %d] sorting k-mers from %zu tmp files", i_tmp, group.size());
        merge_files_once(g, o, j, group, f, false);
        next.push_back(f);
        made.push_back(f);
        group.clear();
    };
    for (auto &f : files) { group.push_back(f); if ((int)group.size() == max_open) flush(); }
    flush();
    info("======= Stage 3: merging from %zu chunks (round: 2/2) =======", next.size());
    return merge_files_once(g, o, j, next, out_file, true);
}
static string tmp_dir_for(const string &tmp_root, const string &out_prefix) {  // sort.go:113-118
    const string root = tmp_root.empty() ? string("./") : tmp_root;
    return root + (root.back() == '/' ? "" : "/") + (out_prefix == "-" ? string("stdout.tmp") : base_of(out_prefix) + ".tmp");
}
static void cleanup_tmp(const vector<string> &files, const string &dir, bool keep) {  // sort.go:425-446
    if (keep) return;
    info("removing %zu intermediate files", files.size());
    for (auto &f : files) if (remove(f.c_str()) != 0) die("fail to remove intermediate file: %s", f.c_str());
    info("removing tmp dir: %s", dir.c_str());
    if (rmdir(dir.c_str()) != 0) die("fail to remove temp directory, please manually delete it: %s", dir.c_str());
}

enum SetCmd { C_UNION, C_INTER, C_DIFF, C_COMMON, C_SORT, C_MERGE, C_SPLIT };

static int cmd_setop(SetCmd which, int argc, char **argv) {
    vector<FlagSpec> specs = {{'o', "out-prefix", true}};
    if (which == C_UNION) specs.push_back({'s', "sort", false});
    if (which == C_INTER) specs.push_back({'m', "mix-taxid", false});
    if (which == C_DIFF) { specs.push_back({'s', "sort", false}); specs.push_back({'t', "compare-taxid", false}); }
    if (which == C_COMMON) { specs.push_back({'m', "mix-taxid", false}); specs.push_back({'p', "proportion", true}); specs.push_back({'n', "number", true}); }
    if (which == C_SORT || which == C_MERGE) {
        specs.push_back({'u', "unique", false}); specs.push_back({'d', "repeated", false});
        specs.push_back({'M', "max-open-files", true}); specs.push_back({'t', "tmp-dir", true});
        specs.push_back({'k', "keep-tmp-dir", false}); specs.push_back({0, "force", false});
        if (which == C_SORT) specs.push_back({'m', "chunk-size", true});
        else { specs.push_back({'D', "is-dir", false}); specs.push_back({'p', "pattern", true}); }
    }
    if (which == C_SPLIT) {
        specs = {{'O', "out-dir", true}, {'m', "chunk-size", true}, {0, "force", false}, {'u', "unique", false}, {'d', "repeated", false}};
    }
    Args a = parse_args(argc, argv, specs);
    Options o = get_options(a);
    vector<string> files = get_files(a, o);
    const string out_prefix = a.str("out-prefix", "-");
    const string out_file = out_name(out_prefix);
    const bool mix = a.has("mix-taxid");
    const bool uniq = a.has("unique"), rep = a.has("repeated");
    if (uniq && rep) die("flag -u/--unique overides -d/--repeated, do not give both");
    const int max_open = (int)a.num("max-open-files", 400);
    if (max_open <= 0) die("value of flag --max-open-files should be positive");
    if (which == C_MERGE && a.has("is-dir")) {  // merge.go:77-130: chunk files of the given directories
        std::regex re(a.str("pattern", "^chunk_\\d+\\.unik$"));
        vector<string> found;
        for (auto d : files) {
            if (d == "-") d = "./";
            if (!dir_exists(d)) { fprintf(stderr, "[WARN] skip unexisted dir: %s\n", d.c_str()); continue; }
            size_t nf = 0;
            for (auto &nme : list_dir(d)) {
                if (nme[0] == '.' || dir_exists(d + "/" + nme) || !std::regex_search(nme, re)) continue;
                found.push_back(d + (d.back() == '/' ? "" : "/") + nme);
                nf++;
            }
            info("%zu chunk files found in dir: %s", nf, d.c_str());
        }
        if (found.empty()) { fprintf(stderr, "[WARN] 0 chunk files found in %zu dir(s)\n", files.size()); return 0; }
        files = found;
    }

    // single-input fast path of union / inter: the file is copied byte for byte (union.go:97-112, inter.go:96-120)
    if ((which == C_UNION || which == C_INTER) && files.size() == 1 && files[0] != "-") {
        if (which == C_INTER) { unik::Reader r(files[0]); if (!r.h.is_sorted() && !a.has("skip-flag-check")) die("input should be sorted: %s", files[0].c_str()); }
        std::ifstream src(files[0], std::ios::binary);
        if (out_file == "-") std::cout << src.rdbuf();
        else { std::ofstream dst(out_file, std::ios::binary); dst << src.rdbuf(); }
        return 0;
    }

    if (which == C_MERGE) {  // merge.go:228-341
        unik::Reader r0(files[0]);
        ChunkJob j;
        j.k = r0.h.k; j.h0 = &r0.h; j.uniq = uniq; j.rep = rep;
        j.tax = !o.ignore_taxid && r0.h.has_taxid_info();
        j.key_bits = r0.h.is_hashed() ? 64 : 2 * r0.h.k;
        j.mode = unik::UnikSorted | (r0.h.is_canonical() ? unik::UnikCanonical : 0) | (r0.h.is_hashed() ? unik::UnikHashed : 0) |
                 (j.tax ? unik::UnikIncludeTaxID : 0);
        Gpu g(o.gpu);
        if (j.tax && (uniq || rep)) j.max_taxid = load_taxonomy(g, o);
        else j.max_taxid = o.max_taxid;
        vector<string> made;
        int i_tmp = 0;
        string tmp_dir;
        if ((int)files.size() >= max_open) {
            tmp_dir = tmp_dir_for(a.str("tmp-dir", "./"), out_prefix);
            prepare_dir(tmp_dir, a.has("force"), "tmp dir");
        }
        merge_rounds(g, o, j, files, out_file, max_open, tmp_dir, i_tmp, made);
        if (!tmp_dir.empty()) cleanup_tmp(made, tmp_dir, a.has("keep-tmp-dir"));
        return 0;
    }

    Inputs in = load_inputs(files, o, which == C_INTER && !a.has("skip-flag-check"), which == C_DIFF, mix);
    bool tax = in.has_taxid || (mix && in.any_taxid);
    if (which == C_DIFF) tax = in.has_taxid;  // taxid always from file 1 (diff.go:496-515)
    Gpu g(o.gpu);
    u32 max_taxid = o.max_taxid;
    bool cmp_taxid = which == C_DIFF && a.has("compare-taxid");
    if (cmp_taxid && !in.has_taxid) {  // diff.go:124-133: a warning, the flag is ignored
        fprintf(stderr, "[WARN] no taxids found in the first file, flag -t/--compare-taxid ignored\n");
        cmp_taxid = false;
    }
    bool need_lca = (tax && which != C_DIFF) || cmp_taxid;
    if (which == C_SORT || which == C_SPLIT) need_lca = tax && (uniq || rep);  // sort.go:196-198
    if (need_lca) max_taxid = load_taxonomy(g, o);
    Ptrs p = ptrs_of(in, tax || cmp_taxid);
    const int ns = (int)in.files.size();
    u64 cap = in.total;
    if (which == C_INTER || which == C_DIFF) cap = in.files[0].codes.size();
    if (which == C_SORT || which == C_SPLIT) cap = 2 * in.total;
    vector<u64> out(cap ? cap : 1);
    vector<u32> tout((tax || cmp_taxid) ? (cap ? cap : 1) : 0);
    u64 n = 0;
    const bool with_t = tax || cmp_taxid;
    const u32 *const *tp = with_t ? p.t.data() : nullptr;
    const u32 *fp = with_t ? p.ft.data() : nullptr;
    bool sorted_out = true;
    switch (which) {
    case C_UNION:
        ck(ukm_union_ft(g.c, p.k.data(), tp, fp, p.n.data(), ns, 0, out.data(), with_t ? tout.data() : nullptr, cap, &n));
        sorted_out = a.has("sort");  // same stream; the flag/encoding follow -s (union.go:220-235)
        break;
    case C_INTER:
        ck(ukm_inter_ft(g.c, p.k.data(), tp, fp, p.n.data(), ns, mix ? UKM_F_MIX_TAXID : 0, out.data(), with_t ? tout.data() : nullptr, cap, &n));
        if (n == 0) info("no intersection found");
        break;
    case C_DIFF: {
        vector<uint8_t> sf;
        for (auto &L : in.files) sf.push_back(L.h.is_sorted() ? 1 : 0);
        // files equal (by path) to the first one are skipped (diff.go:464-466)
        Ptrs q; vector<uint8_t> sf2;
        for (int i = 0; i < ns; i++) if (i == 0 || files[(size_t)i] != files[0]) { q.k.push_back(p.k[(size_t)i]); q.t.push_back(p.t[(size_t)i]); q.ft.push_back(p.ft[(size_t)i]); q.n.push_back(p.n[(size_t)i]); sf2.push_back(sf[(size_t)i]); }
        ck(ukm_diff_ft(g.c, q.k.data(), with_t ? q.t.data() : nullptr, with_t ? q.ft.data() : nullptr, q.n.data(), (int)q.k.size(), sf2.data(),
                       cmp_taxid ? UKM_F_CMP_TAXID : 0, out.data(), with_t ? tout.data() : nullptr, cap, &n));
        if (n == 0) fprintf(stderr, "[WARN] no set difference found\n");
        sorted_out = a.has("sort");
        tax = in.has_taxid;
        break;
    }
    case C_COMMON: {
        const double prop = a.real("proportion", 1.0);
        if (prop <= 0 || prop > 1) die("value of -p/--proportion should be in range of (0, 1]");
        if (ns > 65535) die("at most 65535 files supported");
        const u32 thr = ukm_common_threshold((u32)ns, prop, (u32)a.num("number", 0));
        info("searching k-mers shared by >= %u files ...", thr);
        ck(ukm_common_ft(g.c, p.k.data(), tp, fp, p.n.data(), ns, thr, 0, out.data(), with_t ? tout.data() : nullptr, cap, &n));
        break;
    }
    case C_SPLIT:
    case C_SORT: {
        // sort.go:227-572.  Without -m (or when the input is smaller than one chunk) everything is
        // sorted in HBM in one go; with -m N the reference's chunk protocol runs: stage 1 sorts
        // chunks of N k-mers into tmp files (-u: deduplicated, -d: every code once and repeated ones
        // twice), stage 2/3 merge them (merge_rounds).  `split` is stage 1 alone (split.go).
        const long long max_elem = parse_byte_size(a.str("chunk-size", ""));
        const bool limit = max_elem > 0 && (in.total >= (u64)max_elem || which == C_SPLIT);
        if (limit || which == C_SPLIT) {
            ChunkJob j;
            j.k = in.k; j.h0 = &in.h0; j.uniq = uniq; j.rep = rep; j.tax = tax;
            j.key_bits = in.hashed ? 64 : 2 * in.k;
            j.mode = out_mode(in, true, tax, o);
            j.max_taxid = max_taxid;
            string dir;
            if (which == C_SPLIT) {
                dir = a.str("out-dir", "");
                if (dir.empty()) dir = (files[0] == "-" ? string("stdin") : files[0]) + ".split";
                if (dir != "./" && dir != ".") prepare_dir(dir, a.has("force"), "outDir");
            } else {
                dir = tmp_dir_for(a.str("tmp-dir", "./"), out_prefix);
                prepare_dir(dir, a.has("force"), "tmp dir");
                info("======= Stage 1: spliting k-mers into chunks =======");
            }
            const u64 step = max_elem > 0 ? (u64)max_elem : (in.total ? in.total : 1);
            vector<string> chunks;
            vector<u64> cc; vector<u32> ct;
            int i_tmp = 0;
            auto flush_chunk = [&]() {
                if (cc.empty()) return;
                const string f = chunk_file_name(dir, ++i_tmp);
                info("[chunk %d] sorting %zu k-mers", i_tmp, cc.size());
                const u64 m = sort_chunk_to_file(g, o, j, cc, ct, f);
                info("[chunk %d] %llu k-mers saved to tmp file: %s", i_tmp, (unsigned long long)m, f.c_str());
                chunks.push_back(f);
                cc.clear(); ct.clear();
            };
            for (auto &L : in.files)
                for (size_t i = 0; i < L.codes.size(); i++) {
                    cc.push_back(L.codes[i]);
                    if (tax) ct.push_back(L.taxid_of(i));
                    if (cc.size() >= step) flush_chunk();
                }
            flush_chunk();
            if (which == C_SPLIT) { info("%llu k-mers saved to %zu chunk files in %s", (unsigned long long)in.total, chunks.size(), dir.c_str()); return 0; }
            vector<string> made = chunks;
            merge_rounds(g, o, j, chunks, out_file, max_open, dir, i_tmp, made);
            cleanup_tmp(made, dir, a.has("keep-tmp-dir"));
            return 0;
        }
        vector<u64> all; vector<u32> allt;
        all.reserve(in.total);
        for (auto &L : in.files) {
            all.insert(all.end(), L.codes.begin(), L.codes.end());
            if (tax) {
                if (L.per_record()) allt.insert(allt.end(), L.taxids.begin(), L.taxids.end());
                else allt.insert(allt.end(), L.codes.size(), L.file_taxid);  // (one sort over all inputs: the array is what it takes)
            }
        }
        const int key_bits = in.hashed ? 64 : 2 * in.k;
        if (!all.empty()) {
            if (tax) ck(ukm_sort_pairs(g.c, all.data(), allt.data(), all.size(), key_bits));
            else ck(ukm_sort_u64(g.c, all.data(), all.size(), key_bits));
            ck(ukm_unique(g.c, all.data(), tax ? allt.data() : nullptr, all.size(), uniq ? UKM_UNIQUE : (rep ? UKM_REPEATED : UKM_PLAIN),
                          out.data(), tax ? tout.data() : nullptr, cap, &n));
        }
        break;
    }
    case C_MERGE:
        break;  // handled above
    }
    const u32 mode = out_mode(in, sorted_out, tax, o);
    // a global taxid shared by every input survives as the header's global taxid when no per-record taxids are written
    write_unik(out_file, o, in.k, mode, max_taxid, 0, &in.h0, out.data(), tax ? tout.data() : nullptr, n);
    return 0;
}

// =================================================================================================
// CPU-only commands
// =================================================================================================
static std::unique_ptr<unik::OutStream> text_out(const string &file) {
    const bool gz = file.size() > 3 && file.compare(file.size() - 3, 3, ".gz") == 0;  // "suffix .gz for gzipped out"
    return std::unique_ptr<unik::OutStream>(new unik::OutStream(file, gz, 6));
}
static void put(unik::OutStream &o, const string &s) { o.write(s.data(), s.size()); }

static int cmd_view(int argc, char **argv) {  // view.go:163-218
    Args a = parse_args(argc, argv, {{'o', "out-file", true}, {'n', "show-code", false}, {'N', "show-code-only", false}, {'a', "fasta", false},
                                     {'q', "fastq", false}, {'t', "show-taxid", false}, {'T', "show-taxid-only", false}, {'g', "genome", true}});
    Options o = get_options(a);
    vector<string> files = get_files(a, o);
    vector<string> genomes;
    if (a.has("genome")) {  // comma-separated list (a StringSlice flag in view.go:235)
        std::istringstream gs(a.str("genome", ""));
        string gfile;
        while (std::getline(gs, gfile, ',')) if (!gfile.empty()) genomes.push_back(gfile);
    }
    auto out = text_out(a.str("out-file", "-"));
    string buf;
    // -g: hash -> first location in the genomes (loadHash2Loc, util.go:344-393: canonical ntHash of every window of
    // every CIRCULAR record, the first occurrence wins).  On the device: ukm_nthash of all records, a stable sort of
    // (hash, window index), one lower bound per queried code.
    SeqBatch gb;
    vector<u64> gh, gwoff;  // sorted hashes; first window index of every record
    vector<u32> gpos;       // window index of gh[i]
    std::unique_ptr<Gpu> gpu;
    bool g_loaded = false;
    auto load_genomes = [&](int k) {
        for (auto &gf : genomes) read_fastx(gf, gb, false);
        const u64 nrec = gb.off.size() - 1;
        gwoff.assign(nrec + 1, 0);
        for (u64 r = 0; r < nrec; r++) { const u64 len = gb.off[r + 1] - gb.off[r]; gwoff[r + 1] = gwoff[r] + (len >= (u64)k ? len : 0); }
        const u64 nw = gwoff[nrec];
        if (nw > 0xFFFFFFFFull) die("-g/--genome: more than 2^32 k-mers in the genomes");
        gh.resize(nw); gpos.resize(nw);
        if (nw == 0) return;
        gpu.reset(new Gpu(o.gpu));
        u64 m = 0;
        ck(ukm_nthash(gpu->c, gb.bases.data(), gb.off.data(), nrec, k, 1, 1, 0, gh.data(), nw, &m));
        if (m != nw) die("-g/--genome: %llu hashes for %llu windows", (unsigned long long)m, (unsigned long long)nw);
        for (u64 i = 0; i < nw; i++) gpos[i] = (u32)i;
        ck(ukm_sort_pairs(gpu->c, gh.data(), gpos.data(), nw, 64));
        info("%llu hash-k-mers pairs from %llu sequences loaded", (unsigned long long)nw, (unsigned long long)nrec);
    };
    for (auto &f : files) {
        unik::Reader r(f);
        const int k = r.h.k;
        const bool hashed = r.h.is_hashed();
        const string qual((size_t)k, 'g');
        bool use_genomes = false;
        if (!genomes.empty()) {
            if (!hashed) warn("-g/--genome ignored since k-mers not hashed");
            else if (!r.h.is_canonical()) warn("-g/--genome ignored since 'canonical' flag is off");
            else { if (!g_loaded) { load_genomes(k); g_loaded = true; } use_genomes = true; }
        }
        vector<u64> qcodes;
        vector<u32> qtax;
        vector<string> qkmer;
        if (use_genomes) {  // all codes of the file first, then one batch of lower bounds on the device
            u64 c; u32 t;
            while (r.read(c, t)) { qcodes.push_back(c); qtax.push_back(t); }
            qkmer.resize(qcodes.size());
            vector<u64> cuts;
            const size_t batch = 1u << 28;
            for (size_t b0 = 0; b0 < qcodes.size(); b0 += batch) {
                const size_t nq = std::min(batch, qcodes.size() - b0);
                cuts.resize(nq);
                if (!gh.empty()) ck(ukm_partition_points(gpu->c, gh.data(), gh.size(), qcodes.data() + b0, (int)nq, cuts.data()));
                for (size_t i = 0; i < nq; i++) {
                    const u64 code = qcodes[b0 + i];
                    const u64 lb = gh.empty() ? 0 : cuts[i];
                    if (lb < gh.size() && gh[lb] == code) {
                        const u64 w = gpos[lb];
                        const u64 rec = (u64)(std::upper_bound(gwoff.begin(), gwoff.end(), w) - gwoff.begin()) - 1;
                        const u64 len = gb.off[rec + 1] - gb.off[rec], idx = w - gwoff[rec];
                        string km((size_t)k, 'N');
                        for (int j = 0; j < k; j++) km[(size_t)j] = (char)gb.bases[gb.off[rec] + (idx + (u64)j) % len];
                        qkmer[b0 + i] = km;
                    } else {
                        qkmer[b0 + i] = std::to_string(code);
                        warn("fail to decode hash: %llu, which is not found in given genomes", (unsigned long long)code);
                    }
                }
            }
        }
        u64 code; u32 taxid;
        size_t qi = 0;
        for (;;) {
            if (use_genomes) { if (qi >= qcodes.size()) break; code = qcodes[qi]; taxid = qtax[qi]; }
            else if (!r.read(code, taxid)) break;
            const string kmer = use_genomes ? qkmer[qi++] : (hashed ? std::to_string(code) : decode_kmer(code, k));
            if (a.has("fasta")) buf += ">" + std::to_string(code) + (a.has("show-taxid") ? " " + std::to_string(taxid) : "") + "\n" + kmer + "\n";
            else if (a.has("fastq")) buf += "@" + std::to_string(code) + (a.has("show-taxid") ? " " + std::to_string(taxid) : "") + "\n" + kmer + "\n+\n" + qual + "\n";
            else if (a.has("show-taxid")) buf += kmer + "\t" + std::to_string(taxid) + "\n";
            else if (a.has("show-taxid-only")) buf += std::to_string(taxid) + "\n";
            else if (a.has("show-code-only")) buf += std::to_string(code) + "\n";
            else if (a.has("show-code")) buf += kmer + "\t" + std::to_string(code) + "\n";
            else buf += kmer + "\n";
            if (buf.size() > (1u << 20)) { put(*out, buf); buf.clear(); }
        }
    }
    put(*out, buf);
    return 0;
}

// ntHash of whole k-mers given as text (dump -H, encode -H: dump.go:250-275, encode.go:106-113): every line is one
// record of k bases with exactly one window, hashed on the device in batches
static void hash_kmers_on_device(Gpu &g, const vector<string> &kmers, int k, bool canonical, vector<u64> &out) {
    out.resize(kmers.size());
    const size_t batch = 1u << 22;
    vector<uint8_t> bases;
    vector<u64> off;
    for (size_t b0 = 0; b0 < kmers.size(); b0 += batch) {
        const size_t n = std::min(batch, kmers.size() - b0);
        bases.resize(n * (size_t)k);
        off.resize(n + 1);
        for (size_t i = 0; i < n; i++) {
            memcpy(&bases[i * (size_t)k], kmers[b0 + i].data(), (size_t)k);
            off[i] = (u64)i * (u64)k;
        }
        off[n] = (u64)n * (u64)k;
        u64 m = 0;
        ck(ukm_nthash(g.c, bases.data(), off.data(), n, k, canonical ? 1 : 0, 0, 0, out.data() + b0, n, &m));
        if (m != n) die("ntHash: %llu hashes for %zu k-mers", (unsigned long long)m, n);
    }
}

static int cmd_dump(int argc, char **argv) {  // dump.go:128-315
    Args a = parse_args(argc, argv, {{'o', "out-prefix", true}, {'u', "unique", false}, {'K', "canonical", false}, {'O', "canonical-only", false},
                                     {'s', "sorted", false}, {'t', "taxid", true}, {'H', "hash", false}, {0, "hashed", false}, {'k', "kmer-len", true}});
    Options o = get_options(a);
    vector<string> files = get_files(a, o);
    const bool hashed = a.has("hash");  // compute the ntHash of the given k-mers (dump.go:73)
    if (hashed && a.has("canonical-only")) die("flag -H/--hash and -k/--canonical-only are not compatible");  // dump.go:76-78
    const bool hashed_already = a.has("hashed");
    bool canonical = a.has("canonical");
    const bool canonical_only = a.has("canonical-only"), sorted = a.has("sorted"), unique = a.has("unique");
    int k = -1;
    if (hashed_already) { canonical = true; k = (int)a.num("kmer-len", 0); if (k == 0) die("flag -k/--kmer-len should be given when --hashed given"); }
    const u32 gtaxid = (u32)a.num("taxid", 0);
    const string out_file = out_name(a.str("out-prefix", "-"));
    unik::OutStream os(out_file, o.compress, o.level);
    std::unique_ptr<unik::Writer> w;
    std::set<u64> seen;
    u64 n = 0;
    bool include_taxid = false;
    vector<string> hk;  // -H: the k-mers of all files, hashed on the device after the last line
    vector<u32> ht;
    for (auto &f : files) {
        unik::InStream in(f);
        string text, chunk(1 << 20, '\0');
        for (;;) { size_t got = in.read(&chunk[0], chunk.size()); if (!got) break; text.append(chunk.data(), got); }
        std::istringstream ss(text);
        string line;
        while (std::getline(ss, line)) {
            while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
            if (line.empty()) continue;
            u32 taxid = 0;
            string kmer = line;
            size_t tab = line.find_first_of("\t ");
            bool has_t = false;
            if (tab != string::npos) { kmer = line.substr(0, tab); taxid = (u32)strtoul(line.c_str() + tab + 1, nullptr, 10); has_t = true; }
            if (!w) {  // the first line fixes k and whether taxids are included (dump.go:139-223)
                if (!hashed_already) k = (int)kmer.size();
                if (k > 32 && !hashed_already && !hashed) die("k-mer size (%d) should be <= 32", k);
                if (k > 64) die("k-mer size (%d) should be <= 64", k);
                include_taxid = has_t;
                u32 mode = 0;
                if (sorted) mode |= unik::UnikSorted;
                else if (o.compact && !hashed_already && !hashed) mode |= unik::UnikCompact;
                if (canonical || canonical_only) mode |= unik::UnikCanonical;
                if (include_taxid) mode |= unik::UnikIncludeTaxID;
                if (hashed_already || hashed) mode |= unik::UnikHashed;
                w.reset(new unik::Writer(os, k, mode));
                w->set_max_taxid(o.max_taxid);
                if (gtaxid && !include_taxid) w->set_global_taxid(gtaxid);
            }
            u64 code;
            if (hashed) {
                if ((int)kmer.size() != k) die("K-mer length mismatch, previous: %d, current: %zu. %s", k, kmer.size(), kmer.c_str());
                hk.push_back(kmer);
                ht.push_back(taxid);
                continue;
            }
            if (hashed_already) code = strtoull(kmer.c_str(), nullptr, 10);
            else {
                if ((int)kmer.size() != k) die("K-mer length mismatch, previous: %d, current: %zu. %s", k, kmer.size(), kmer.c_str());
                if (!encode_kmer(kmer, code)) die("fail to encode '%s': illegal base", kmer.c_str());
                const u64 rc = revcomp(code, k);
                if (canonical_only) { if (rc < code) continue; }
                else if (canonical && rc < code) code = rc;
            }
            if (unique && !seen.insert(code).second) continue;
            if (include_taxid) w->write_code_with_taxid(code, taxid); else w->write_code(code);
            n++;
        }
    }
    if (!w) die("no k-mers given");
    if (hashed) {
        Gpu g(o.gpu);
        vector<u64> hv;
        hash_kmers_on_device(g, hk, k, canonical, hv);
        for (size_t i = 0; i < hv.size(); i++) {
            if (unique && !seen.insert(hv[i]).second) continue;
            if (include_taxid) w->write_code_with_taxid(hv[i], ht[i]); else w->write_code(hv[i]);
            n++;
        }
    }
    w->flush();
    os.close();
    info("%llu unique k-mers saved to %s", (unsigned long long)n, out_file.c_str());
    return 0;
}

static string basename_of(const string &p) { size_t s = p.find_last_of('/'); return s == string::npos ? p : p.substr(s + 1); }

static int cmd_num(int argc, char **argv) {  // num.go:60-131
    Args a = parse_args(argc, argv, {{'o', "out-file", true}, {'n', "file-name", false}, {'b', "basename", false}, {'f', "force", false}});
    Options o = get_options(a);
    vector<string> files = get_files(a, o);
    auto out = text_out(a.str("out-file", "-"));
    for (auto &f : files) {
        unik::Reader r(f);
        long long n = (r.h.number == ~0ull || r.h.number == 0) ? -1 : (long long)r.h.number;
        if (n < 0 && a.has("force")) { u64 c; u32 t; n = 0; while (r.read(c, t)) n++; }
        string line = std::to_string(n);
        if (a.has("file-name")) line += "\t" + (a.has("basename") ? basename_of(f) : f);
        put(*out, line + "\n");
    }
    return 0;
}

static int cmd_info(int argc, char **argv) {  // info.go:377-421 (tabular form)
    Args a = parse_args(argc, argv, {{'o', "out-file", true}, {'a', "all", false}, {'T', "tabular", false}, {'e', "skip-err", false},
                                     {0, "symbol-true", true}, {0, "symbol-false", true}, {'b', "basename", false}});
    Options o = get_options(a);
    vector<string> files = get_files(a, o);
    auto out = text_out(a.str("out-file", "-"));
    const string T = a.str("symbol-true", "✓"), F = a.str("symbol-false", "✕");
    string hdr = "file\tk\tcanonical\thashed\tscaled\tinclude-taxid\tglobal-taxid\tsorted";
    if (a.has("all")) hdr += "\tcompact\tgzipped\tversion\tnumber\tdescription";
    put(*out, hdr + "\n");
    for (auto &f : files) {
        unik::Reader r(f);
        auto b = [&](bool v) { return v ? T : F; };
        string line = (a.has("basename") ? basename_of(f) : f) + "\t" + std::to_string(r.h.k) + "\t" + b(r.h.is_canonical()) + "\t" + b(r.h.is_hashed()) +
                      "\t" + b(r.h.is_scaled()) + "\t" + b(r.h.is_include_taxid()) + "\t" + (r.h.has_global_taxid() ? std::to_string(r.h.global_taxid) : "") +
                      "\t" + b(r.h.is_sorted());
        if (a.has("all")) {
            long long n = (r.h.number == ~0ull || r.h.number == 0) ? -1 : (long long)r.h.number;
            if (n < 0) { u64 c; u32 t; n = 0; while (r.read(c, t)) n++; }
            line += "\t" + b(r.h.is_compact()) + "\t" + b(r.gzipped()) + "\tv" + std::to_string(r.h.main_version) + "." + std::to_string(r.h.minor_version) +
                    "\t" + std::to_string(n) + "\t" + r.h.description;
        }
        put(*out, line + "\n");
    }
    return 0;
}

static int cmd_concat(int argc, char **argv) {  // concat.go:60-206
    Args a = parse_args(argc, argv, {{'o', "out-prefix", true}, {'s', "sorted", false}, {'t', "taxid", true}, {'n', "number", true}});
    Options o = get_options(a);
    vector<string> files = get_files(a, o);
    const string out_file = out_name(a.str("out-prefix", "-"));
    unik::OutStream os(out_file, o.compress, o.level);
    std::unique_ptr<unik::Writer> w;
    unik::Header h0;
    bool tax = false;
    u64 n = 0;
    for (size_t i = 0; i < files.size(); i++) {
        unik::Reader r(files[i]);
        if (!w) {
            h0 = r.h;
            tax = !o.ignore_taxid && r.h.has_taxid_info();
            u32 mode = 0;
            if (a.has("sorted")) mode |= unik::UnikSorted;
            else if (o.compact && !r.h.is_hashed()) mode |= unik::UnikCompact;
            if (r.h.is_canonical()) mode |= unik::UnikCanonical;
            if (tax) mode |= unik::UnikIncludeTaxID;
            if (r.h.is_hashed()) mode |= unik::UnikHashed;
            w.reset(new unik::Writer(os, r.h.k, mode));
            w->set_max_taxid(o.max_taxid);
            if (r.h.is_scaled()) w->set_scale(r.h.scale, r.h.max_hash);
            const long long num = a.num("number", -1);
            w->set_number(num < 0 ? ~0ull : (u64)num);  // concat.go:69,143-145: unknown unless given
        } else {
            check_compat(h0, r.h, files[i]);
        }
        u64 c; u32 t;
        while (r.read(c, t)) { if (tax) w->write_code_with_taxid(c, t); else w->write_code(c); n++; }
    }
    if (!w) die("no input");
    w->flush();
    os.close();
    info("%llu k-mers saved to %s", (unsigned long long)n, out_file.c_str());
    return 0;
}

static int cmd_head(int argc, char **argv) {  // head.go:60-163
    Args a = parse_args(argc, argv, {{'o', "out-prefix", true}, {'n', "number", true}});
    Options o = get_options(a);
    vector<string> files = get_files(a, o);
    if (files.size() > 1) die("no more than one file should be given");
    const long long number = a.num("number", 10);
    if (number <= 0) die("value of flag -n/--number should be positive");
    unik::Reader r(files[0]);
    const string out_file = out_name(a.str("out-prefix", "-"));
    unik::OutStream os(out_file, o.compress, o.level);
    unik::Header h = r.h;
    unik::Writer w(os, h.k, h.flag & ~(u32)unik::UnikScaled);
    w.h.taxid_bytes = h.taxid_bytes; w.h.global_taxid = h.global_taxid; w.h.description = h.description;
    if (h.is_scaled()) w.set_scale(h.scale, h.max_hash);
    u64 c; u32 t; long long n = 0;
    vector<std::pair<u64, u32>> recs;
    while (n < number && r.read(c, t)) { recs.push_back({c, t}); n++; }
    w.set_number((u64)n);
    for (auto &x : recs) { if (h.is_include_taxid()) w.write_code_with_taxid(x.first, x.second); else w.write_code(x.first); }
    w.flush();
    os.close();
    return 0;
}

static int cmd_encode(int argc, char **argv) {  // encode.go:60-136
    Args a = parse_args(argc, argv, {{'o', "out-file", true}, {'a', "all", false}, {'K', "canonical", false}, {'H', "hash", false}});
    Options o = get_options(a);
    const bool hashed = a.has("hash");
    vector<string> files = get_files(a, o);
    auto out = text_out(a.str("out-file", "-"));
    if (hashed) {  // encode.go:106-113: one ntHash per line, the first line fixes k (<= 64)
        vector<string> kmers;
        int k = -1;
        for (auto &f : files) {
            unik::InStream in(f);
            string text, chunk(1 << 20, '\0');
            for (;;) { size_t got = in.read(&chunk[0], chunk.size()); if (!got) break; text.append(chunk.data(), got); }
            std::istringstream ss(text);
            string line;
            while (std::getline(ss, line)) {
                while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
                if (line.empty()) continue;
                if (k == -1) { k = (int)line.size(); if (k > 64) die("k-mer size (%d) should be <=64", k); }
                else if ((int)line.size() != k) die("K-mer length mismatch, previous: %d, current: %zu. %s", k, line.size(), line.c_str());
                kmers.push_back(line);
            }
        }
        if (kmers.empty()) return 0;
        Gpu g(o.gpu);
        vector<u64> hv;
        hash_kmers_on_device(g, kmers, k, a.has("canonical"), hv);
        string buf;
        for (u64 h : hv) { buf += std::to_string(h) + "\n"; if (buf.size() > (1u << 20)) { put(*out, buf); buf.clear(); } }
        put(*out, buf);
        return 0;
    }
    for (auto &f : files) {
        unik::InStream in(f);
        string text, chunk(1 << 20, '\0');
        for (;;) { size_t got = in.read(&chunk[0], chunk.size()); if (!got) break; text.append(chunk.data(), got); }
        std::istringstream ss(text);
        string line;
        while (std::getline(ss, line)) {
            while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
            if (line.empty()) continue;
            u64 code;
            if (!encode_kmer(line, code)) die("fail to encode '%s'", line.c_str());
            const int k = (int)line.size();
            if (a.has("canonical")) code = std::min(code, revcomp(code, k));
            if (a.has("all")) put(*out, line + "\t" + decode_kmer(code, k) + "\t" + std::to_string(code) + "\n");
            else put(*out, std::to_string(code) + "\n");
        }
    }
    return 0;
}

static int cmd_decode(int argc, char **argv) {  // decode.go:60-124
    Args a = parse_args(argc, argv, {{'o', "out-file", true}, {'k', "kmer-len", true}, {'a', "all", false}});
    Options o = get_options(a);
    const int k = (int)a.num("kmer-len", 0);
    if (k <= 0 || k > 32) die("invalid k: %d", k);
    vector<string> files = get_files(a, o);
    auto out = text_out(a.str("out-file", "-"));
    const u64 maxcode = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    for (auto &f : files) {
        unik::InStream in(f);
        string text, chunk(1 << 20, '\0');
        for (;;) { size_t got = in.read(&chunk[0], chunk.size()); if (!got) break; text.append(chunk.data(), got); }
        std::istringstream ss(text);
        string line;
        while (std::getline(ss, line)) {
            if (line.empty()) continue;
            const u64 code = strtoull(line.c_str(), nullptr, 10);
            if (code > maxcode) die("encode integer overflows for k=%d, max: %llu", k, (unsigned long long)maxcode);  // decode.go:102-104
            if (a.has("all")) put(*out, std::to_string(code) + "\t" + decode_kmer(code, k) + "\n");
            else put(*out, decode_kmer(code, k) + "\n");
        }
    }
    return 0;
}

static void usage() {
    fprintf(stderr,
            "unikmer (HIP) - k-mer set operations on AMD MI355X behind the unikmer command line\n\n"
            "Usage: unikmer <command> [flags] [files]\n\n"
            "GPU commands : count sort split merge union inter diff common\n"
            "CPU commands : view dump num info(stats) concat head encode decode version\n"
            "Global flags : -j --verbose -C --compression-level -c -i -I --max-taxid --data-dir --gpu\n");
}

int main(int argc, char **argv) {
    if (argc < 2) { usage(); return 0; }
    const string cmd = argv[1];
    argc -= 2; argv += 2;
    try {
        if (cmd == "count") return cmd_count(argc, argv);
        if (cmd == "sort") return cmd_setop(C_SORT, argc, argv);
        if (cmd == "union") return cmd_setop(C_UNION, argc, argv);
        if (cmd == "inter") return cmd_setop(C_INTER, argc, argv);
        if (cmd == "diff") return cmd_setop(C_DIFF, argc, argv);
        if (cmd == "common") return cmd_setop(C_COMMON, argc, argv);
        if (cmd == "merge") return cmd_setop(C_MERGE, argc, argv);
        if (cmd == "split") return cmd_setop(C_SPLIT, argc, argv);
        if (cmd == "view") return cmd_view(argc, argv);
        if (cmd == "dump") return cmd_dump(argc, argv);
        if (cmd == "num") return cmd_num(argc, argv);
        if (cmd == "info" || cmd == "stats") return cmd_info(argc, argv);
        if (cmd == "concat") return cmd_concat(argc, argv);
        if (cmd == "head") return cmd_head(argc, argv);
        if (cmd == "encode") return cmd_encode(argc, argv);
        if (cmd == "decode") return cmd_decode(argc, argv);
        if (cmd == "version") { printf("unikmer v%s\n", VERSION); return 0; }
        if (cmd == "-h" || cmd == "--help" || cmd == "help") { usage(); return 0; }
        die("unknown command \"%s\" for \"unikmer\"", cmd.c_str());
    } catch (const unik::Error &e) {
        die("%s", e.what());
    } catch (const std::exception &e) {
        die("%s", e.what());
    }
}
