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
  // copier threads a text pipe is read with for a given thread allowance (0: one plain read(); see read_fifo_fanout)
  static unsigned fanout_threads(unsigned n_threads) { return n_threads >= 12 ? 4u : (n_threads >= 6 ? 2u : 0u); }
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
  ssize_t read_fifo_fanout(uint8_t *dst, size_t cap);  // kFanoutUnavailable: not possible here, nothing consumed
  ssize_t read_gzip(uint8_t *dst, size_t cap);
  ssize_t read_bgzf(uint8_t *dst, size_t cap);

  static constexpr ssize_t kFanoutUnavailable = -3;
  int fd_;
  unsigned n_threads_;
  // text from a pipe: the pipe's pages are handed on to a few private pipes (splice(2): no copy) whose readers copy them
  // out side by side -- one thread's read() of a pipe copies ~8 GB/s, and a pipe's read() copies under the pipe's lock
  bool fifo_ = false;
  bool fanout_ok_ = true;
  unsigned n_fan_ = 0;
  int fan_[4][2] = {{-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}};
  Kind kind_ = kUnknown;
  bool fd_eof_ = false;
  std::vector<uint8_t> cbuf_;  // raw bytes not yet consumed
  size_t cpos_ = 0;            // consumed prefix of cbuf_
  z_stream_s *z_ = nullptr;    // gzip mode
  bool z_member_done_ = false;
  std::string err_;
};

}  // namespace bvcf_input
