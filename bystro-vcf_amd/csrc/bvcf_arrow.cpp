// bvcf_arrow.cpp — Arrow IPC *file* writer for the dosage matrix (--dosageOutput).
//
// Replaces the reference's arrow/arrow.go (ArrowWriter + ArrowRowBuilder over apache/arrow/go, an
// un-vendored dependency) for the one table shape main.go:306-342 writes: a utf8 column "locus" and
// one int8 column per sample, record batches of 5 000 rows (main.go:518), body buffers compressed
// with zstd (ipc.WithZstd()).  The format is the published Arrow columnar IPC format (Message.fbs,
// Schema.fbs, File.fbs, metadata V5); the flatbuffers are built by the small back-to-front builder
// below.  Host-only: no device code here.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bvcf.h"

// libzstd.so.1 ships in the image without its header: the three entry points used, as published
extern "C" {
size_t ZSTD_compress(void *dst, size_t dstCapacity, const void *src, size_t srcSize, int compressionLevel);
size_t ZSTD_compressBound(size_t srcSize);
unsigned ZSTD_isError(size_t code);
typedef struct ZSTD_CCtx_s ZSTD_CCtx;
ZSTD_CCtx *ZSTD_createCCtx(void);
size_t ZSTD_freeCCtx(ZSTD_CCtx *cctx);
size_t ZSTD_compressCCtx(ZSTD_CCtx *cctx, void *dst, size_t dstCapacity, const void *src, size_t srcSize, int compressionLevel);
}

namespace {

// ---------------------------------------------------------------- minimal flatbuffer builder
// Back to front, like the reference implementations: children are written before (= at higher
// addresses than) their parents, every offset is "distance from the end of the buffer".
class Fb {
 public:
  Fb() : buf_(1024), head_(1024) {}
  size_t size() const { return buf_.size() - head_; }
  const uint8_t *data() const { return buf_.data() + head_; }

  void prep(size_t sz, size_t additional) {
    if (sz > minalign_) minalign_ = sz;
    const size_t pad = (~(size() + additional) + 1) & (sz - 1);
    need(pad + sz + additional);
    head_ -= pad;
    memset(&buf_[head_], 0, pad);
  }
  template <typename T>
  void put(T v) {  // after prep
    head_ -= sizeof(T);
    memcpy(&buf_[head_], &v, sizeof(T));
  }
  template <typename T>
  void scalar(T v) {
    prep(sizeof(T), 0);
    put(v);
  }
  void uoffset(uint32_t target) {  // a reference to something already written
    prep(4, 0);
    put<uint32_t>((uint32_t)size() - target + 4u);
  }
  uint32_t string(const char *s, size_t n) {
    prep(4, n + 1);
    need(n + 1);
    head_ -= n + 1;
    memcpy(&buf_[head_], s, n);
    buf_[head_ + n] = 0;
    put<uint32_t>((uint32_t)n);
    return (uint32_t)size();
  }
  // vectors: start, add the elements LAST FIRST, end
  void start_vector(size_t elem_size, size_t count, size_t align) {
    prep(4, elem_size * count);
    prep(align, elem_size * count);
  }
  uint32_t end_vector(size_t count) {
    put<uint32_t>((uint32_t)count);
    return (uint32_t)size();
  }
  // tables
  void start_table(int n_slots) {
    slots_.assign(n_slots, 0);
    table_start_ = (uint32_t)size();
  }
  template <typename T>
  void field(int slot, T v, T dflt) {
    if (v == dflt) return;
    scalar(v);
    slots_[slot] = (uint32_t)size();
  }
  void field_offset(int slot, uint32_t target) {
    if (!target) return;
    uoffset(target);
    slots_[slot] = (uint32_t)size();
  }
  uint32_t end_table() {
    scalar<int32_t>(0);  // soffset to the vtable, patched below
    const uint32_t table = (uint32_t)size();
    int n = (int)slots_.size();
    while (n > 0 && slots_[n - 1] == 0) n--;
    for (int i = n - 1; i >= 0; i--) {
      prep(2, 0);
      put<uint16_t>(slots_[i] ? (uint16_t)(table - slots_[i]) : (uint16_t)0);
    }
    prep(2, 0);
    put<uint16_t>((uint16_t)(table - table_start_));
    prep(2, 0);
    put<uint16_t>((uint16_t)((n + 2) * 2));
    const int32_t soff = (int32_t)size() - (int32_t)table;
    memcpy(&buf_[buf_.size() - table], &soff, 4);
    return table;
  }
  void finish(uint32_t root) {
    prep(minalign_ < 8 ? 8 : minalign_, 4);
    uoffset(root);
  }

 private:
  void need(size_t n) {
    if (head_ >= n) return;
    const size_t old = buf_.size(), used = old - head_;
    size_t cap = old * 2;
    while (cap - used < n) cap *= 2;
    std::vector<uint8_t> nb(cap);
    memcpy(&nb[cap - used], &buf_[head_], used);
    buf_.swap(nb);
    head_ = cap - used;
  }
  std::vector<uint8_t> buf_;
  size_t head_;
  size_t minalign_ = 1;
  std::vector<uint32_t> slots_;
  uint32_t table_start_ = 0;
};

constexpr int16_t kMetadataV5 = 4;
enum { kHeaderSchema = 1, kHeaderRecordBatch = 3 };
enum { kTypeInt = 2, kTypeUtf8 = 5 };

struct Block {
  int64_t offset;
  int32_t meta_len;
  int64_t body_len;
};

uint32_t build_schema(Fb &fb, const std::vector<std::string> &names) {
  // Field { name, nullable, type_type, type, dictionary, children, custom_metadata }
  std::vector<uint32_t> fields(names.size());
  for (size_t i = names.size(); i-- > 0;) {
    uint32_t type;
    if (i == 0) {
      fb.start_table(0);  // Utf8 {}
      type = fb.end_table();
    } else {
      fb.start_table(2);  // Int { bitWidth, is_signed }
      fb.field<int32_t>(0, 8, 0);
      fb.field<uint8_t>(1, 1, 0);
      type = fb.end_table();
    }
    fb.start_vector(4, 0, 4);
    const uint32_t children = fb.end_vector(0);
    const uint32_t name = fb.string(names[i].data(), names[i].size());
    fb.start_table(7);
    fb.field_offset(0, name);
    fb.field<uint8_t>(1, 1, 0);  // nullable, like the schema the Go writer builds from arrow.Field{Nullable: true}
    fb.field<uint8_t>(2, i == 0 ? kTypeUtf8 : kTypeInt, 0);
    fb.field_offset(3, type);
    fb.field_offset(5, children);
    fields[i] = fb.end_table();
  }
  fb.start_vector(4, fields.size(), 4);
  for (size_t i = fields.size(); i-- > 0;) fb.uoffset(fields[i]);
  const uint32_t fvec = fb.end_vector(fields.size());
  fb.start_table(4);  // Schema { endianness, fields, custom_metadata, features }
  fb.field_offset(1, fvec);
  return fb.end_table();
}

uint32_t build_message(Fb &fb, uint8_t header_type, uint32_t header, int64_t body_len) {
  fb.start_table(5);  // Message { version, header_type, header, bodyLength, custom_metadata }
  fb.field<int16_t>(0, kMetadataV5, 0);
  fb.field<uint8_t>(1, header_type, 0);
  fb.field_offset(2, header);
  fb.field<int64_t>(3, body_len, 0);
  return fb.end_table();
}

struct Writer {
  FILE *f = nullptr;
  std::vector<std::string> names;  // "locus" + samples
  uint32_t ns = 0, rows_per_batch = 5000;
  int level = 3;  // < 0: buffers stored uncompressed
  int64_t pos = 0;
  std::vector<Block> blocks;
  // the batch being filled
  uint32_t n_rows = 0;
  std::vector<int32_t> locus_off;
  std::string locus_data;
  std::vector<int8_t> rows;  // the batch as it arrives, row-major: rows[row * ns + s]; transposed at flush
  std::string err;
  // the batch being written: a full batch is handed to a background thread (transpose, zstd, file write) while the
  // next one fills; only that thread touches the file between open and close
  std::thread bg;
  bool bg_ok = true;
  uint32_t bg_n_rows = 0;
  std::vector<int32_t> bg_locus_off;
  std::string bg_locus_data;
  std::vector<int8_t> bg_rows;

  bool wait_bg() {
    if (bg.joinable()) bg.join();
    return bg_ok;
  }
  // hands the filled batch to the background writer (after the previous one has finished)
  bool flush_async() {
    if (!wait_bg()) return false;
    if (n_rows == 0) return true;
    std::swap(rows, bg_rows);
    std::swap(locus_off, bg_locus_off);
    std::swap(locus_data, bg_locus_data);
    bg_n_rows = n_rows;
    if (rows.size() != bg_rows.size()) rows.resize(bg_rows.size());
    n_rows = 0;
    locus_off.clear();
    locus_data.clear();
    bg = std::thread([this]() { bg_ok = write_batch(bg_rows, bg_locus_off, bg_locus_data, bg_n_rows); });
    return true;
  }
  bool flush() {
    if (!flush_async()) return false;
    return wait_bg();
  }

  bool put(const void *p, size_t n) {
    if (n && fwrite(p, 1, n, f) != n) {
      err = "write failed";
      return false;
    }
    pos += (int64_t)n;
    return true;
  }
  bool pad8() {
    static const char z[8] = {0};
    return put(z, (size_t)((8 - (pos & 7)) & 7));
  }
  // continuation marker, metadata length, flatbuffer, padding; returns the bytes written
  bool put_message(const Fb &fb, int32_t *meta_len) {
    const uint32_t cont = 0xFFFFFFFFu;
    const int32_t len = (int32_t)((fb.size() + 7) & ~(size_t)7);
    *meta_len = len + 8;
    return put(&cont, 4) && put(&len, 4) && put(fb.data(), fb.size()) && pad8();
  }
  // one body buffer: int64 uncompressed length + zstd frame (or -1 + the raw bytes)
  void add_buffer(std::string &body, std::vector<std::pair<int64_t, int64_t>> &bufs, const void *p, size_t n,
                  ZSTD_CCtx *cctx = nullptr) {
    const int64_t at = (int64_t)body.size();
    if (n == 0) {
      bufs.emplace_back(at, 0);
      return;
    }
    if (level < 0) {
      body.append((const char *)p, n);
      bufs.emplace_back(at, (int64_t)n);
    } else {
      const size_t bound = ZSTD_compressBound(n);
      body.resize(body.size() + 8 + bound);
      // (a context per thread: the one-shot call builds and tears down its own for every 5 000-byte column)
      const size_t got = cctx ? ZSTD_compressCCtx(cctx, &body[(size_t)at + 8], bound, p, n, level)
                              : ZSTD_compress(&body[(size_t)at + 8], bound, p, n, level);
      int64_t raw = (int64_t)n;
      size_t used = got;
      if (ZSTD_isError(got) || got >= n) {  // not worth it: the format's "stored" form
        raw = -1;
        memcpy(&body[(size_t)at + 8], p, n);
        used = n;
      }
      memcpy(&body[(size_t)at], &raw, 8);
      body.resize((size_t)at + 8 + used);
      bufs.emplace_back(at, (int64_t)(8 + used));
    }
    body.resize((body.size() + 7) & ~(size_t)7, '\0');
  }
  bool write_batch(const std::vector<int8_t> &rows, std::vector<int32_t> &locus_off, const std::string &locus_data,
                   const uint32_t n_rows) {
    if (n_rows == 0) return true;
    std::string body;
    std::vector<std::pair<int64_t, int64_t>> bufs;
    locus_off.push_back((int32_t)locus_data.size());
    add_buffer(body, bufs, nullptr, 0);  // locus validity: no nulls
    add_buffer(body, bufs, locus_off.data(), locus_off.size() * 4);
    add_buffer(body, bufs, locus_data.data(), locus_data.size());
    // the sample columns are independent buffers: compressed by a few threads, each into its own piece of
    // the body, then laid end to end (every piece is a multiple of 8 bytes long)
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned n_thr = (unsigned)std::min<uint64_t>(std::min(16u, hw ? hw : 1u), ((uint64_t)ns * n_rows >> 16) + 1);
    std::vector<std::string> piece(n_thr);
    std::vector<std::vector<std::pair<int64_t, int64_t>>> piece_bufs(n_thr);
    auto work = [&](unsigned t) {
      // 64 columns at a time: one cache line of every row feeds 64 column buffers that stay in L1/L2
      constexpr uint32_t kTile = 64;
      std::vector<int8_t> col((size_t)kTile * n_rows);
      ZSTD_CCtx *cctx = level >= 0 ? ZSTD_createCCtx() : nullptr;
      const uint32_t s_lo = (uint32_t)((uint64_t)ns * t / n_thr), s_hi = (uint32_t)((uint64_t)ns * (t + 1) / n_thr);
      for (uint32_t s0 = s_lo; s0 < s_hi; s0 += kTile) {
        const uint32_t w = std::min(kTile, s_hi - s0);
        for (uint32_t r = 0; r < n_rows; r++) {
          const int8_t *src = &rows[(size_t)r * ns + s0];
          for (uint32_t j = 0; j < w; j++) col[(size_t)j * n_rows + r] = src[j];
        }
        for (uint32_t j = 0; j < w; j++) {
          add_buffer(piece[t], piece_bufs[t], nullptr, 0);
          add_buffer(piece[t], piece_bufs[t], &col[(size_t)j * n_rows], n_rows, cctx);
        }
      }
      if (cctx) ZSTD_freeCCtx(cctx);
    };
    if (n_thr <= 1) {
      work(0);
    } else {
      std::vector<std::thread> th;
      for (unsigned t = 0; t < n_thr; t++) th.emplace_back(work, t);
      for (auto &x : th) x.join();
    }
    for (unsigned t = 0; t < n_thr; t++) {
      const int64_t at = (int64_t)body.size();
      for (auto &b : piece_bufs[t]) bufs.emplace_back(b.first + at, b.second);
      body.append(piece[t]);
    }
    Fb fb;
    uint32_t compression = 0;
    if (level >= 0) {
      fb.start_table(2);           // BodyCompression { codec = LZ4_FRAME, method = BUFFER }
      fb.field<int8_t>(0, 1, 0);   // ZSTD
      compression = fb.end_table();
    }
    fb.start_vector(16, bufs.size(), 8);  // struct Buffer { offset, length }
    for (size_t i = bufs.size(); i-- > 0;) {
      fb.put<int64_t>(bufs[i].second);
      fb.put<int64_t>(bufs[i].first);
    }
    const uint32_t bvec = fb.end_vector(bufs.size());
    fb.start_vector(16, (size_t)ns + 1, 8);  // struct FieldNode { length, null_count }
    for (uint32_t i = 0; i <= ns; i++) {
      fb.put<int64_t>(0);
      fb.put<int64_t>((int64_t)n_rows);
    }
    const uint32_t nvec = fb.end_vector((size_t)ns + 1);
    fb.start_table(5);  // RecordBatch { length, nodes, buffers, compression, variadicBufferCounts }
    fb.field<int64_t>(0, (int64_t)n_rows, 0);
    fb.field_offset(1, nvec);
    fb.field_offset(2, bvec);
    fb.field_offset(3, compression);
    const uint32_t rb = fb.end_table();
    fb.finish(build_message(fb, kHeaderRecordBatch, rb, (int64_t)body.size()));
    Block b;
    b.offset = pos;
    b.body_len = (int64_t)body.size();
    if (!put_message(fb, &b.meta_len) || !put(body.data(), body.size())) return false;
    blocks.push_back(b);
    return true;
  }
};

}  // namespace

struct bvcf_arrow {
  Writer w;
};

extern "C" {

int bvcf_arrow_open(bvcf_arrow **out, const char *path, const char *const *sample_names,
                    const uint32_t *sample_name_lens, uint32_t n_samples, uint32_t rows_per_batch, int zstd_level) {
  if (!out || !path) return BVCF_E_ARG;
  *out = nullptr;
  bvcf_arrow *a = new bvcf_arrow();
  Writer &w = a->w;
  w.f = fopen(path, "wb");
  if (!w.f) {
    delete a;
    return BVCF_E_IO;
  }
  w.ns = n_samples;
  if (rows_per_batch) w.rows_per_batch = rows_per_batch;
  w.level = zstd_level == 0 ? 3 : zstd_level;
  w.names.emplace_back("locus");
  for (uint32_t i = 0; i < n_samples; i++) w.names.emplace_back(sample_names[i], sample_name_lens[i]);
  w.rows.resize((size_t)n_samples * w.rows_per_batch);
  Fb fb;
  fb.finish(build_message(fb, kHeaderSchema, build_schema(fb, w.names), 0));
  int32_t meta_len;
  if (!w.put("ARROW1\0\0", 8) || !w.put_message(fb, &meta_len)) {
    fclose(w.f);
    delete a;
    return BVCF_E_IO;
  }
  *out = a;
  return BVCF_OK;
}

int bvcf_arrow_append(bvcf_arrow *a, const char *locus, uint32_t locus_len, const int8_t *dosage) {
  if (!a) return BVCF_E_ARG;
  Writer &w = a->w;
  w.locus_off.push_back((int32_t)w.locus_data.size());
  w.locus_data.append(locus, locus_len);
  if (w.ns) memcpy(&w.rows[(size_t)w.n_rows * w.ns], dosage, w.ns);
  if (++w.n_rows == w.rows_per_batch && !w.flush_async()) return BVCF_E_IO;
  return BVCF_OK;
}

int bvcf_arrow_close(bvcf_arrow *a) {
  if (!a) return BVCF_E_ARG;
  Writer &w = a->w;
  bool ok = w.flush();
  const uint32_t eos[2] = {0xFFFFFFFFu, 0u};
  ok = ok && w.put(eos, 8);
  Fb fb;
  fb.start_vector(24, w.blocks.size(), 8);  // struct Block { offset: long; metaDataLength: int; bodyLength: long }
  for (size_t i = w.blocks.size(); i-- > 0;) {
    fb.put<int64_t>(w.blocks[i].body_len);
    fb.put<int32_t>(0);  // padding
    fb.put<int32_t>(w.blocks[i].meta_len);
    fb.put<int64_t>(w.blocks[i].offset);
  }
  const uint32_t rbs = fb.end_vector(w.blocks.size());
  fb.start_vector(24, 0, 8);
  const uint32_t dicts = fb.end_vector(0);
  const uint32_t schema = build_schema(fb, w.names);
  fb.start_table(5);  // Footer { version, schema, dictionaries, recordBatches, custom_metadata }
  fb.field<int16_t>(0, kMetadataV5, 0);
  fb.field_offset(1, schema);
  fb.field_offset(2, dicts);
  fb.field_offset(3, rbs);
  fb.finish(fb.end_table());
  const int32_t flen = (int32_t)fb.size();
  ok = ok && w.put(fb.data(), fb.size()) && w.put(&flen, 4) && w.put("ARROW1", 6);
  ok = (fclose(w.f) == 0) && ok;
  delete a;
  return ok ? BVCF_OK : BVCF_E_IO;
}

}  // extern "C"
