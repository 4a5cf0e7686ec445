// bvcf_input.h — the byte source in front of the path: plain text, gzip, or BGZF.
//
// The reference reads uncompressed text and leaves decompression to `pigz -d -c` in front of the
// pipe (README.md:10,46 — that single-threaded inflate is the stated bottleneck of its headline
// run).  Here the driver accepts the compressed file itself: a single-stream gzip is inflated in the
// reader thread (nothing more is possible: one deflate stream is sequential), a BGZF file (bgzip /
// htslib, the usual form of .vcf.gz) is inflated block-parallel.
#pragma once

#include <stddef.h>
#include <stdint.h>
#include <sys/types.h>

#include <string>
#include <vector>

struct z_stream_s;

namespace bvcf_input {

class ByteSource {
 public:
  ByteSource(int fd, unsigned n_threads);
  ~ByteSource();
  ByteSource(const ByteSource &) = delete;
  ByteSource &operator=(const ByteSource &) = delete;

  // Up to `cap` bytes of text into dst.  Returns the count, 0 at end of input, -1 on error, kNoRoom
  // when the next indivisible piece (a BGZF block, <= 64 KiB) does not fit into cap.
  static constexpr ssize_t kNoRoom = -2;
  ssize_t read(uint8_t *dst, size_t cap);
  const std::string &error() const { return err_; }
  const char *kind() const;  // "text", "gzip" or "bgzf" (after the first read / sniff)
  // Looks at the first bytes without consuming them: true if the input is BGZF (block-gzip).
  bool sniff_bgzf();
  // The raw (still compressed) bytes of the input, for a caller that inflates elsewhere (the device): up to cap bytes,
  // 0 at end of input, -1 on error.
  ssize_t read_raw(uint8_t *dst, size_t cap);

 private:
  enum Kind { kUnknown, kText, kGzip, kBgzf };
  bool fill_compressed();                       // append raw bytes from fd to cbuf_
  ssize_t read_text(uint8_t *dst, size_t cap);
  ssize_t read_gzip(uint8_t *dst, size_t cap);
  ssize_t read_bgzf(uint8_t *dst, size_t cap);

  int fd_;
  unsigned n_threads_;
  Kind kind_ = kUnknown;
  bool fd_eof_ = false;
  std::vector<uint8_t> cbuf_;  // raw bytes not yet consumed
  size_t cpos_ = 0;            // consumed prefix of cbuf_
  z_stream_s *z_ = nullptr;    // gzip mode
  bool z_member_done_ = false;
  std::string err_;
};

}  // namespace bvcf_input
