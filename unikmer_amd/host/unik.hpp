// unik.hpp — reader/writer of the `.unik` v5.0 container, the C++ stand-in for
// github.com/shenwei356/unik/v5 v5.0.1 as the reference uses it (unik.NewReader /
// ReadCodeWithTaxid / NewWriter / WriteCode / WriteCodeWithTaxid / WriteTaxid / Flush /
// SetMaxTaxid / SetGlobalTaxid / SetScale / Number; e.g. union.go:136,171,187,251).
//
// PARITY UNPINNED: the reference tree holds neither a `.unik` file nor a format description
// (`.gitignore` excludes *.unik) and the module source is not vendored.  The byte layout
// below is the reconstruction of SURVEY.md Appendix B6; it is self-consistent (everything this
// tool writes it reads back) but has not been checked against a file written by the Go tool.
//
// Layout (big-endian throughout):
//   magic ".unikmer" (8) | main=5, minor=0, K, 0 (4 x u8) | Flag u32 | Number u64
//   | globalTaxid u32 | taxidBytesLen u8 | 3 x 0 | descLen u32 | desc | scale u32 | maxHash u64
//   | 52 reserved zero bytes
//   body, unsorted: per record  code (8 bytes, or (k+3)/4 bytes when Compact) [taxid]
//   body, sorted  : records in pairs: ctrl = ((len0-1)<<3)|(len1-1), then len0 bytes of
//                   (c0 - prev) and len1 bytes of (c1 - c0), prev = c1, [taxid0][taxid1];
//                   a trailing odd record: ctrl = 128, 8 bytes of the full code [taxid]
//   taxid = taxidBytesLen big-endian bytes, present when IncludeTaxID.
// The whole stream is optionally gzip (zlib's gz* API reads plain and gzip transparently,
// like util-io.go:99-101 sniffing 1f 8b).
#pragma once

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace unik {

enum : uint32_t {
    UnikCompact = 1u << 0,
    UnikCanonical = 1u << 1,
    UnikSorted = 1u << 2,
    UnikIncludeTaxID = 1u << 3,
    UnikHashed = 1u << 4,
    UnikScaled = 1u << 5,
};

struct Header {
    uint8_t main_version = 5, minor_version = 0;
    int k = 0;
    uint32_t flag = 0;
    uint64_t number = ~0ull;  // 0xFFFF.. (prints as -1) or 0 = unknown (concat.go:69, info.go:379)
    uint32_t global_taxid = 0;
    uint8_t taxid_bytes = 4;
    std::string description;
    uint32_t scale = 1;
    uint64_t max_hash = ~0ull;

    bool is_compact() const { return flag & UnikCompact; }
    bool is_canonical() const { return flag & UnikCanonical; }
    bool is_sorted() const { return flag & UnikSorted; }
    bool is_include_taxid() const { return flag & UnikIncludeTaxID; }
    bool is_hashed() const { return flag & UnikHashed; }
    bool is_scaled() const { return flag & UnikScaled; }
    bool has_global_taxid() const { return global_taxid > 0; }
    bool has_taxid_info() const { return is_include_taxid() || has_global_taxid(); }
};

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

inline int taxid_bytes_for(uint32_t max_taxid) {  // util.go:340-342 maxUint32N inverse
    if (max_taxid <= 0xFF) return 1;
    if (max_taxid <= 0xFFFF) return 2;
    if (max_taxid <= 0xFFFFFF) return 3;
    return 4;
}

// ---- byte streams over zlib ---------------------------------------------------------------------
class InStream {
  public:
    explicit InStream(const std::string &path) : path_(path) {
        if (path == "-") gz_ = gzdopen(fileno(stdin), "rb");
        else gz_ = gzopen(path.c_str(), "rb");
        if (!gz_) throw Error("fail to open file: " + path);
        gzbuffer(gz_, 1 << 20);
    }
    ~InStream() { if (gz_) gzclose(gz_); }
    InStream(const InStream &) = delete;
    // returns number of bytes read (< n only at EOF)
    size_t read(void *dst, size_t n) {
        size_t got = 0;
        while (got < n) {
            int r = gzread(gz_, (char *)dst + got, (unsigned)std::min<size_t>(n - got, 1u << 30));
            if (r < 0) throw Error("read error: " + path_);
            if (r == 0) break;
            got += (size_t)r;
        }
        return got;
    }
    void must_read(void *dst, size_t n) {
        if (read(dst, n) != n) throw Error("unexpected EOF: " + path_);
    }
    bool gzipped() const { return !gzdirect(gz_); }
    gzFile raw() { return gz_; }

  private:
    std::string path_;
    gzFile gz_ = nullptr;
};

class OutStream {
  public:
    OutStream(const std::string &path, bool compress, int level) : path_(path) {
        if (compress) {
            std::string mode = "wb" + std::to_string(level < 0 ? 6 : (level > 9 ? 9 : level));
            gz_ = (path == "-") ? gzdopen(fileno(stdout), mode.c_str()) : gzopen(path.c_str(), mode.c_str());
            if (!gz_) throw Error("fail to write file: " + path);
            gzbuffer(gz_, 1 << 20);
        } else {
            fp_ = (path == "-") ? stdout : fopen(path.c_str(), "wb");
            if (!fp_) throw Error("fail to write file: " + path);
        }
    }
    ~OutStream() { close(); }
    OutStream(const OutStream &) = delete;
    void write(const void *src, size_t n) {
        if (n == 0) return;
        if (gz_) {
            size_t done = 0;
            while (done < n) {
                unsigned chunk = (unsigned)std::min<size_t>(n - done, 1u << 30);
                if (gzwrite(gz_, (const char *)src + done, chunk) != (int)chunk) throw Error("write error: " + path_);
                done += chunk;
            }
        } else if (fwrite(src, 1, n, fp_) != n) {
            throw Error("write error: " + path_);
        }
    }
    void close() {
        if (gz_) { gzclose(gz_); gz_ = nullptr; }
        if (fp_) { if (fp_ != stdout) fclose(fp_); else fflush(fp_); fp_ = nullptr; }
    }

  private:
    std::string path_;
    gzFile gz_ = nullptr;
    FILE *fp_ = nullptr;
};

static inline void put_be(uint8_t *p, uint64_t v, int n) {
    for (int i = n - 1; i >= 0; i--) { p[i] = (uint8_t)v; v >>= 8; }
}
static inline uint64_t get_be(const uint8_t *p, int n) {
    uint64_t v = 0;
    for (int i = 0; i < n; i++) v = (v << 8) | p[i];
    return v;
}
static inline int byte_len(uint64_t v) {
    int n = 1;
    while (v >>= 8) n++;
    return n;
}

// ---- Reader -------------------------------------------------------------------------------------
class Reader {
  public:
    Header h;
    explicit Reader(const std::string &path) : in_(path), path_(path) { read_header(); }

    bool gzipped() const { return in_.gzipped(); }

    // unik.Reader.ReadCodeWithTaxid: false at EOF.  When the file has only a global taxid it is
    // returned for every record.
    bool read(uint64_t &code, uint32_t &taxid) {
        taxid = h.global_taxid;
        if (!h.is_sorted()) {
            uint8_t b[8];
            const int n = h.is_compact() ? (h.k + 3) / 4 : 8;
            size_t got = in_.read(b, (size_t)n);
            if (got == 0) return false;
            if (got != (size_t)n) throw Error("truncated record: " + path_);
            code = get_be(b, n);
            if (h.is_include_taxid()) taxid = read_taxid();
            return true;
        }
        if (have_second_) {
            code = second_;
            taxid = h.is_include_taxid() ? second_taxid_ : h.global_taxid;
            have_second_ = false;
            return true;
        }
        uint8_t ctrl;
        if (in_.read(&ctrl, 1) == 0) return false;
        uint8_t b[16];
        if (ctrl & 128) {  // trailing single record: the full code
            in_.must_read(b, 8);
            code = get_be(b, 8);
            prev_ = code;
            if (h.is_include_taxid()) taxid = read_taxid();
            return true;
        }
        const int l0 = ((ctrl >> 3) & 7) + 1, l1 = (ctrl & 7) + 1;
        in_.must_read(b, (size_t)(l0 + l1));
        const uint64_t c0 = prev_ + get_be(b, l0);
        const uint64_t c1 = c0 + get_be(b + l0, l1);
        prev_ = c1;
        code = c0;
        second_ = c1;
        have_second_ = true;
        if (h.is_include_taxid()) {
            taxid = read_taxid();
            second_taxid_ = read_taxid();
        }
        return true;
    }

    // bulk: append every record; taxids filled when the file carries taxid information
    void read_all(std::vector<uint64_t> &codes, std::vector<uint32_t> *taxids) {
        uint64_t c;
        uint32_t t;
        if (h.number != ~0ull && h.number != 0) {
            codes.reserve(codes.size() + h.number);
            if (taxids) taxids->reserve(taxids->size() + h.number);
        }
        while (read(c, t)) {
            codes.push_back(c);
            if (taxids) taxids->push_back(t);
        }
    }

  private:
    uint32_t read_taxid() {
        uint8_t b[4];
        in_.must_read(b, h.taxid_bytes);
        return (uint32_t)get_be(b, h.taxid_bytes);
    }
    void read_header() {
        uint8_t b[64];
        if (in_.read(b, 8) != 8 || memcmp(b, ".unikmer", 8) != 0) throw Error("invalid binary format: " + path_);
        in_.must_read(b, 4);
        h.main_version = b[0]; h.minor_version = b[1]; h.k = b[2];
        if (h.main_version != 5) throw Error("version mismatch (need v5.x): " + path_);
        in_.must_read(b, 4); h.flag = (uint32_t)get_be(b, 4);
        in_.must_read(b, 8); h.number = get_be(b, 8);
        in_.must_read(b, 4); h.global_taxid = (uint32_t)get_be(b, 4);
        in_.must_read(b, 4); h.taxid_bytes = b[0];
        if (h.taxid_bytes < 1 || h.taxid_bytes > 4) throw Error("bad taxid byte length: " + path_);
        in_.must_read(b, 4);
        const uint32_t dl = (uint32_t)get_be(b, 4);
        if (dl > 1024) throw Error("description too long: " + path_);
        h.description.resize(dl);
        if (dl) in_.must_read(&h.description[0], dl);
        in_.must_read(b, 4); h.scale = (uint32_t)get_be(b, 4);
        in_.must_read(b, 8); h.max_hash = get_be(b, 8);
        in_.must_read(b, 52);
    }

    InStream in_;
    std::string path_;
    uint64_t prev_ = 0, second_ = 0;
    uint32_t second_taxid_ = 0;
    bool have_second_ = false;
};

// ---- Writer -------------------------------------------------------------------------------------
class Writer {
  public:
    Header h;
    Writer(OutStream &out, int k, uint32_t mode) : out_(out) {
        h.k = k;
        h.flag = mode;
        buf_.reserve(1 << 20);
    }
    void set_max_taxid(uint32_t m) { h.taxid_bytes = (uint8_t)taxid_bytes_for(m); }
    void set_global_taxid(uint32_t t) { h.global_taxid = t; }
    void set_scale(uint32_t scale, uint64_t max_hash) {  // count.go:254-256,469-471
        h.scale = scale;
        h.max_hash = max_hash;
        h.flag |= UnikScaled;
    }
    void set_number(uint64_t n) { h.number = n; }

    void write_header() {
        if (wrote_header_) return;
        wrote_header_ = true;
        uint8_t b[64] = {0};
        out_.write(".unikmer", 8);
        b[0] = h.main_version; b[1] = h.minor_version; b[2] = (uint8_t)h.k; b[3] = 0;
        out_.write(b, 4);
        put_be(b, h.flag, 4); out_.write(b, 4);
        put_be(b, h.number, 8); out_.write(b, 8);
        put_be(b, h.global_taxid, 4); out_.write(b, 4);
        b[0] = h.taxid_bytes; b[1] = b[2] = b[3] = 0; out_.write(b, 4);
        put_be(b, h.description.size(), 4); out_.write(b, 4);
        out_.write(h.description.data(), h.description.size());
        put_be(b, h.scale, 4); out_.write(b, 4);
        put_be(b, h.max_hash, 8); out_.write(b, 8);
        memset(b, 0, 52); out_.write(b, 52);
    }

    void write_code(uint64_t code) { write_code_with_taxid(code, 0); }

    void write_code_with_taxid(uint64_t code, uint32_t taxid) {
        write_header();
        const bool tx = h.is_include_taxid();
        if (!h.is_sorted()) {
            uint8_t b[12];
            const int n = h.is_compact() ? (h.k + 3) / 4 : 8;
            put_be(b, code, n);
            int m = n;
            if (tx) { put_be(b + n, taxid, h.taxid_bytes); m += h.taxid_bytes; }
            push(b, m);
            return;
        }
        if (!have_first_) {
            first_ = code;
            first_taxid_ = taxid;
            have_first_ = true;
            return;
        }
        if (first_ < prev_ || code < first_) throw Error("codes written to a sorted .unik must be ascending");
        uint8_t b[32];
        const uint64_t d0 = first_ - prev_, d1 = code - first_;
        const int l0 = byte_len(d0), l1 = byte_len(d1);
        b[0] = (uint8_t)(((l0 - 1) << 3) | (l1 - 1));
        put_be(b + 1, d0, l0);
        put_be(b + 1 + l0, d1, l1);
        int m = 1 + l0 + l1;
        if (tx) {
            put_be(b + m, first_taxid_, h.taxid_bytes); m += h.taxid_bytes;
            put_be(b + m, taxid, h.taxid_bytes); m += h.taxid_bytes;
        }
        push(b, m);
        prev_ = code;
        have_first_ = false;
    }

    void flush() {
        write_header();
        if (h.is_sorted() && have_first_) {
            uint8_t b[16];
            b[0] = 128;
            put_be(b + 1, first_, 8);
            int m = 9;
            if (h.is_include_taxid()) { put_be(b + m, first_taxid_, h.taxid_bytes); m += h.taxid_bytes; }
            push(b, m);
            have_first_ = false;
        }
        if (!buf_.empty()) { out_.write(buf_.data(), buf_.size()); buf_.clear(); }
    }

  private:
    void push(const uint8_t *b, int n) {
        buf_.insert(buf_.end(), b, b + n);
        if (buf_.size() >= (1u << 20)) { out_.write(buf_.data(), buf_.size()); buf_.clear(); }
    }
    OutStream &out_;
    std::vector<uint8_t> buf_;
    bool wrote_header_ = false, have_first_ = false;
    uint64_t prev_ = 0, first_ = 0;
    uint32_t first_taxid_ = 0;
};

}  // namespace unik
