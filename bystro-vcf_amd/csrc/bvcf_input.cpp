// bvcf_input.cpp — see bvcf_input.h
#include "bvcf_input.h"

#include <errno.h>
#include <fcntl.h>
#include <poll.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace bvcf_input {

namespace {

constexpr size_t kRawChunk = 8u << 20;    // bytes asked from the fd at a time

struct BgzfBlock {
  size_t payload_off;  // deflate data inside cbuf_
  uint32_t payload_len;
  uint32_t crc, isize;
  size_t out_off;
};

// a complete BGZF block at p[0..n)?  Returns its total size (BSIZE + 1), 0 if more bytes are needed,
// -1 if p does not start a BGZF block.  (SAM spec §4.1: gzip member with FEXTRA subfield 'B','C')
long bgzf_block_size(const uint8_t *p, size_t n, uint32_t *xlen_out) {
  if (n < 18) return 0;
  if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return -1;
  const uint32_t xlen = p[10] | (p[11] << 8);
  if (n < 12 + (size_t)xlen) return 0;
  size_t q = 12;
  const size_t xend = 12 + xlen;
  while (q + 4 <= xend) {
    const uint32_t slen = p[q + 2] | (p[q + 3] << 8);
    if (p[q] == 'B' && p[q + 1] == 'C' && slen == 2 && q + 6 <= xend) {
      *xlen_out = xlen;
      return (long)(p[q + 4] | (p[q + 5] << 8)) + 1;
    }
    q += 4 + slen;
  }
  return -1;
}

}  // namespace

ByteSource::ByteSource(int fd, unsigned n_threads) : fd_(fd), n_threads_(n_threads ? n_threads : 1) {
  if (const char *e = getenv("BVCF_INFLATE_THREADS")) n_threads_ = (unsigned)std::max(1, atoi(e));  // tuning
  // `pigz -dc in.vcf.gz | bystro-vcf`: a pipe hands over 64 KiB per read() by default; ask for the most the system
  // gives an unprivileged process (1 MiB, /proc/sys/fs/pipe-max-size) -- fewer system calls and context switches
  struct stat st;
  if (fstat(fd_, &st) == 0 && S_ISFIFO(st.st_mode)) {
    (void)fcntl(fd_, F_SETPIPE_SZ, 1 << 20);
    fifo_ = true;
  }
  if (const char *e = getenv("BVCF_PIPE_FANOUT")) fanout_ok_ = *e != '0';  // (A/B and tests)
}

ByteSource::~ByteSource() {
  if (z_) {
    inflateEnd(z_);
    delete z_;
  }
  for (auto &p : fan_)
    for (int &f : p)
      if (f >= 0) close(f);
}

// Text from a pipe, `cap` bytes or to the end of input.  This thread hands the pipe's pages on, up to 1 MiB at a time and
// in turn, to n_fan_ private pipes -- splice(2) between pipes moves page references, it copies nothing --; one thread per
// private pipe copies what arrives there to its place in dst.  (A pipe's read() copies with the pipe locked, so several
// readers of ONE pipe would only take turns.)
ssize_t ByteSource::read_fifo_fanout(uint8_t *dst, size_t cap) {
  if (!n_fan_) {
    const unsigned k = fanout_threads(n_threads_);
    if (!k) return kFanoutUnavailable;
    for (unsigned i = 0; i < k; i++) {
      if (pipe2(fan_[i], O_CLOEXEC) != 0) {
        for (unsigned j = 0; j < i; j++) {
          close(fan_[j][0]);
          close(fan_[j][1]);
          fan_[j][0] = fan_[j][1] = -1;
        }
        fan_[i][0] = fan_[i][1] = -1;
        return kFanoutUnavailable;
      }
      (void)fcntl(fan_[i][1], F_SETPIPE_SZ, 1 << 20);  // (refused past the user's pipe-page allowance: 64 KiB then)
    }
    n_fan_ = k;
  }
  struct Job {
    uint8_t *dst;
    size_t n;
  };
  struct Lane {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> q;
    bool done = false;
    int err = 0;
  };
  std::vector<Lane> lanes(n_fan_);
  std::atomic<bool> copier_failed{false};  // (a copier that cannot read its pipe: stop handing pages on, they would pile up there)
  auto copier = [&](unsigned k) {
    Lane &L = lanes[k];
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(L.mu);
        L.cv.wait(lk, [&] { return !L.q.empty() || L.done; });
        if (L.q.empty()) return;
        j = L.q.front();
        L.q.pop_front();
      }
      size_t got = 0;
      while (got < j.n) {  // (the bytes are in the private pipe already: this never waits for the producer)
        const ssize_t g = ::read(fan_[k][0], j.dst + got, j.n - got);
        if (g < 0 && errno == EINTR) continue;
        if (g <= 0) {
          std::lock_guard<std::mutex> lk(L.mu);
          L.err = g < 0 ? errno : EIO;
          copier_failed.store(true);
          return;
        }
        got += (size_t)g;
      }
    }
  };
  size_t total = 0;
  int splice_err = 0;
  unsigned k = 0;
  std::vector<std::thread> th;
  // (a copier that has failed no longer empties its pipe: a blocking splice into it would never return, so the wait for
  // room there and for input is a poll that looks at the flag every 100 ms)
  auto wait_ready = [&](int fd, short ev) {
    struct pollfd pf = {fd, ev, 0};
    while (!copier_failed.load()) {
      const int r = poll(&pf, 1, 100);
      if (r > 0 || (r < 0 && errno != EINTR)) return;
    }
  };
  while (total < cap && !copier_failed.load()) {
    const size_t want = std::min<size_t>(1u << 20, cap - total);
    wait_ready(fan_[k][1], POLLOUT);
    wait_ready(fd_, POLLIN);
    if (copier_failed.load()) break;
    const ssize_t m = splice(fd_, nullptr, fan_[k][1], nullptr, want, SPLICE_F_MOVE | SPLICE_F_NONBLOCK);
    if (m < 0 && (errno == EINTR || errno == EAGAIN)) continue;
    if (m < 0) {
      if (total == 0 && (errno == EINVAL || errno == ENOSYS || errno == EBADF)) return kFanoutUnavailable;  // (no threads yet)
      splice_err = errno;
      break;
    }
    if (m == 0) break;  // end of input
    if (th.empty())
      for (unsigned i = 0; i < n_fan_; i++) th.emplace_back(copier, i);
    {
      std::lock_guard<std::mutex> lk(lanes[k].mu);
      lanes[k].q.push_back(Job{dst + total, (size_t)m});
    }
    lanes[k].cv.notify_one();
    total += (size_t)m;
    k = (k + 1) % n_fan_;
  }
  for (auto &L : lanes) {
    {
      std::lock_guard<std::mutex> lk(L.mu);
      L.done = true;
    }
    L.cv.notify_one();
  }
  for (auto &x : th) x.join();
  for (auto &L : lanes)
    if (L.err) {
      err_ = std::string("read: ") + strerror(L.err);
      return -1;
    }
  if (splice_err) {
    err_ = std::string("splice: ") + strerror(splice_err);
    return -1;
  }
  return (ssize_t)total;
}

const char *ByteSource::kind() const {
  switch (kind_) {
    case kText: return "text";
    case kGzip: return "gzip";
    case kBgzf: return "bgzf";
    default: return "unknown";
  }
}

bool ByteSource::fill_compressed() {
  if (fd_eof_) return false;
  if (cpos_ > 0 && cpos_ == cbuf_.size()) {
    cbuf_.clear();
    cpos_ = 0;
  }
  const size_t old = cbuf_.size();
  cbuf_.resize(old + kRawChunk);
  size_t got_total = 0;
  for (;;) {
    ssize_t got = ::read(fd_, cbuf_.data() + old, kRawChunk);
    if (got < 0) {
      if (errno == EINTR) continue;
      err_ = std::string("read: ") + strerror(errno);
      cbuf_.resize(old);
      fd_eof_ = true;
      return false;
    }
    got_total = (size_t)got;
    break;
  }
  cbuf_.resize(old + got_total);
  if (got_total == 0) fd_eof_ = true;
  return got_total > 0;
}

bool ByteSource::sniff_bgzf() {
  if (kind_ == kUnknown) {
    while (cbuf_.size() < 18 && fill_compressed()) {
    }
    if (!err_.empty()) return false;
    kind_ = kText;
    if (cbuf_.size() >= 2 && cbuf_[0] == 0x1f && cbuf_[1] == 0x8b) {
      uint32_t xlen;
      kind_ = bgzf_block_size(cbuf_.data(), cbuf_.size(), &xlen) != -1 ? kBgzf : kGzip;
    }
  }
  return kind_ == kBgzf;
}

ssize_t ByteSource::read_raw(uint8_t *dst, size_t cap) {
  if (!err_.empty()) return -1;
  return read_text(dst, cap);  // what the sniffing buffered, then the fd itself (parallel pread for files)
}

ssize_t ByteSource::read(uint8_t *dst, size_t cap) {
  if (!err_.empty()) return -1;
  if (kind_ == kUnknown) {
    sniff_bgzf();
    if (!err_.empty()) return -1;
  }
  switch (kind_) {
    case kGzip: return read_gzip(dst, cap);
    case kBgzf: return read_bgzf(dst, cap);
    default: return read_text(dst, cap);
  }
}

ssize_t ByteSource::read_text(uint8_t *dst, size_t cap) {
  if (cpos_ < cbuf_.size()) {  // what the sniffing consumed from the fd
    const size_t n = std::min(cap, cbuf_.size() - cpos_);
    memcpy(dst, cbuf_.data() + cpos_, n);
    cpos_ += n;
    if (cpos_ == cbuf_.size()) {
      std::vector<uint8_t>().swap(cbuf_);
      cpos_ = 0;
    }
    return (ssize_t)n;
  }
  // a regular file: a few threads pread() disjoint parts of the request (one thread copies out of the page cache
  // at about 10 GB/s, which is what the CLI waited for once the device and the formatter were done)
  if (cap >= (8u << 20)) {
    struct stat st;
    const off_t at = lseek(fd_, 0, SEEK_CUR);
    if (at >= 0 && fstat(fd_, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > at) {
      const size_t want = (size_t)std::min<off_t>((off_t)cap, st.st_size - at);
      unsigned max_thr = 8;
      if (const char *e = getenv("BVCF_READ_THREADS")) max_thr = (unsigned)std::max(1, atoi(e));  // tuning
      const unsigned n_thr = std::max(1u, std::min({max_thr, n_threads_ ? n_threads_ : 8u, (unsigned)(want >> 21)}));
      std::vector<size_t> got_n(n_thr, 0);
      std::vector<int> err_n(n_thr, 0);
      auto part = [&](unsigned t) {
        const size_t lo = want * t / n_thr, hi = want * (t + 1) / n_thr;
        size_t done = 0;
        while (lo + done < hi) {
          const ssize_t g = pread(fd_, dst + lo + done, hi - lo - done, at + (off_t)(lo + done));
          if (g < 0 && errno == EINTR) continue;
          if (g < 0) err_n[t] = errno;
          if (g <= 0) break;  // error, or the file got shorter
          done += (size_t)g;
        }
        got_n[t] = done;
      };
      std::vector<std::thread> th;
      for (unsigned t = 1; t < n_thr; t++) th.emplace_back(part, t);
      part(0);
      for (auto &x : th) x.join();
      size_t total = 0;
      for (unsigned t = 0; t < n_thr; t++) {
        if (err_n[t]) {
          err_ = std::string("read: ") + strerror(err_n[t]);
          return -1;
        }
        total += got_n[t];
        if (got_n[t] != want * (t + 1) / n_thr - want * t / n_thr) break;  // short part: what follows it is not contiguous
      }
      lseek(fd_, at + (off_t)total, SEEK_SET);
      return (ssize_t)total;
    }
  }
  if (fifo_ && fanout_ok_ && cap >= (4u << 20)) {
    const ssize_t got = read_fifo_fanout(dst, cap);
    if (got != kFanoutUnavailable) return got;
    fanout_ok_ = false;
  }
  for (;;) {
    ssize_t got = ::read(fd_, dst, cap);
    if (got < 0 && errno == EINTR) continue;
    if (got < 0) err_ = std::string("read: ") + strerror(errno);
    return got;
  }
}

ssize_t ByteSource::read_gzip(uint8_t *dst, size_t cap) {
  if (!z_) {
    z_ = new z_stream_s();
    memset(z_, 0, sizeof *z_);
    if (inflateInit2(z_, 15 + 32) != Z_OK) {
      err_ = "inflateInit2 failed";
      return -1;
    }
  }
  size_t produced = 0;
  while (produced < cap) {
    if (cpos_ == cbuf_.size()) {
      if (!fill_compressed()) {
        if (!err_.empty()) return -1;
        if (!z_member_done_ && produced == 0) {
          // input ended inside a member
          if (z_->total_in > 0) {
            err_ = "gzip: unexpected end of file";
            return -1;
          }
        }
        break;
      }
    }
    if (z_member_done_) {  // concatenated members (pigz -i, cat a.gz b.gz)
      if (cbuf_[cpos_] == 0) {  // zero padding after the last member: ignore, as gzip(1) does
        cpos_++;
        continue;
      }
      inflateReset(z_);
      z_member_done_ = false;
    }
    z_->next_in = cbuf_.data() + cpos_;
    z_->avail_in = (uInt)std::min<size_t>(cbuf_.size() - cpos_, 1u << 30);
    z_->next_out = dst + produced;
    z_->avail_out = (uInt)std::min<size_t>(cap - produced, 1u << 30);
    const uInt in0 = z_->avail_in, out0 = z_->avail_out;
    const int rc = inflate(z_, Z_NO_FLUSH);
    cpos_ += in0 - z_->avail_in;
    produced += out0 - z_->avail_out;
    if (rc == Z_STREAM_END) {
      z_member_done_ = true;
    } else if (rc != Z_OK && rc != Z_BUF_ERROR) {
      err_ = std::string("gzip: ") + (z_->msg ? z_->msg : "corrupt input");
      return -1;
    }
  }
  return (ssize_t)produced;
}

ssize_t ByteSource::read_bgzf(uint8_t *dst, size_t cap) {
  for (;;) {
    // ---- as many whole blocks as fit into dst
    std::vector<BgzfBlock> blocks;
    size_t pos = cpos_, out = 0;
    bool need_more = false, no_room = false;
    while (pos < cbuf_.size()) {
      uint32_t xlen = 0;
      const long bs = bgzf_block_size(cbuf_.data() + pos, cbuf_.size() - pos, &xlen);
      if (bs < 0) {
        err_ = "bgzf: not a BGZF block (plain gzip member inside a BGZF file?)";
        return -1;
      }
      if (bs == 0 || pos + (size_t)bs > cbuf_.size()) {
        need_more = true;
        break;
      }
      if ((size_t)bs < 12 + (size_t)xlen + 8) {
        err_ = "bgzf: corrupt block size";
        return -1;
      }
      const uint8_t *tail = cbuf_.data() + pos + bs - 8;
      BgzfBlock b;
      b.payload_off = pos + 12 + xlen;
      b.payload_len = (uint32_t)(bs - 12 - xlen - 8);
      b.crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
      b.isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
      if (b.isize > (1u << 16)) {
        err_ = "bgzf: block larger than 64 KiB";
        return -1;
      }
      if (out + b.isize > cap) {
        no_room = true;
        break;
      }
      b.out_off = out;
      out += b.isize;
      blocks.push_back(b);
      pos += (size_t)bs;
    }
    if (!blocks.empty() && (out > 0 || !need_more)) {
      // ---- inflate them in parallel, each to its own place
      std::atomic<size_t> next{0};
      std::atomic<int> bad{0};
      auto work = [&]() {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) {
          bad.store(1);
          return;
        }
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= blocks.size() || bad.load()) break;
          const BgzfBlock &b = blocks[i];
          inflateReset(&zs);
          zs.next_in = cbuf_.data() + b.payload_off;
          zs.avail_in = b.payload_len;
          zs.next_out = dst + b.out_off;
          zs.avail_out = b.isize;
          const int rc = inflate(&zs, Z_FINISH);
          if (rc != Z_STREAM_END || zs.avail_out != 0 ||
              (uint32_t)crc32(crc32(0L, Z_NULL, 0), dst + b.out_off, b.isize) != b.crc)
            bad.store(2);
        }
        inflateEnd(&zs);
      };
      const unsigned nt = (unsigned)std::min<size_t>(n_threads_, blocks.size());
      if (nt <= 1) {
        work();
      } else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back(work);
        for (auto &x : th) x.join();
      }
      if (bad.load()) {
        err_ = "bgzf: corrupt block (inflate or CRC mismatch)";
        return -1;
      }
      cpos_ = pos;
      if (out > 0) return (ssize_t)out;
      continue;  // only empty blocks (the EOF marker): look further
    }
    if (!blocks.empty()) cpos_ = pos;
    if (no_room) return kNoRoom;  // the next block does not fit into what is left of dst
    // ---- need more compressed bytes
    if (cpos_ > 0) {
      cbuf_.erase(cbuf_.begin(), cbuf_.begin() + (ptrdiff_t)cpos_);
      cpos_ = 0;
    }
    if (!fill_compressed()) {
      if (!err_.empty()) return -1;
      if (!cbuf_.empty()) {
        err_ = "bgzf: truncated block at end of file";
        return -1;
      }
      return 0;
    }
  }
}

}  // namespace bvcf_input
