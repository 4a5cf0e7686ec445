// bvcf_plan.cpp — the pure half of bvcf_run_fd's input side: which byte ranges there are, which lines a range owns, where
// BGZF blocks start, how many host threads each stage gets.  No I/O, no device, no threads: tests/test_plan.py drives
// these through the C entries of include/bvcf_plan.h.  Counterpart of the reference's single producer cutting 64-line
// work items (main.go:345-380).
#include "bvcf_pipeline.h"

namespace bvcf_host {

// The byte ranges of a text file.  A reader reads range + spare straight into one pinned buffer of `cap` bytes.
bvcf_range_plan plan_text_ranges(uint64_t file_size, uint64_t data_off, uint64_t cap, uint64_t first_line_bytes) {
  bvcf_range_plan p;
  memset(&p, 0, sizeof p);
  if (!cap) cap = 64ull << 20;
  p.data_off = data_off;
  // up to an eighth of the buffer for the line that straddles a range's end (tiny buffers: up to half, 64 KiB if it fits)
  p.range_bytes = cap - std::max<uint64_t>(cap / 8, std::min<uint64_t>(cap / 2, 64u << 10));
  if (!p.range_bytes) p.range_bytes = 1;
  p.spare_bytes = cap - p.range_bytes;
  // ... of which eight times the first data line are read, 64 KiB at least: a line that needs more takes the slow way
  p.spare_bytes = std::min<uint64_t>(p.spare_bytes, std::max<uint64_t>(8 * first_line_bytes, 64u << 10));
  const uint64_t body = file_size > data_off ? file_size - data_off : 0;
  p.n_ranges = (body + p.range_bytes - 1) / p.range_bytes;
  return p;
}

uint64_t bgzf_range_bytes(uint64_t total, unsigned n_workers, uint64_t cap) {
  if (!cap) cap = 256ull << 20;
  if (!n_workers) n_workers = 1;
  uint64_t rb = std::min<uint64_t>(std::max<uint64_t>(total / (4ull * n_workers), 1u << 20), std::max<uint64_t>(cap / 4, 1u << 20));
  return (rb + 0xFFFFu) & ~(uint64_t)0xFFFFu;
}

bvcf_range_plan plan_bgzf_ranges(uint64_t file_size, uint64_t data_off, unsigned n_workers, uint64_t cap) {
  bvcf_range_plan p;
  memset(&p, 0, sizeof p);
  p.data_off = data_off;
  p.range_bytes = bgzf_range_bytes(file_size, n_workers, cap);
  const uint64_t body = file_size > data_off ? file_size - data_off : 0;
  p.n_ranges = (body + p.range_bytes - 1) / p.range_bytes;
  return p;
}

// Range [a, b) owns the bytes (T(a), T(b)], T(x) = the first terminator at or after x.  buf[0, n) was read at a.
bvcf_text_cut cut_text_range(const uint8_t *buf, size_t n, size_t own_len, bool first_range, bool last_range, uint8_t eol) {
  bvcf_text_cut c;
  memset(&c, 0, sizeof c);
  own_len = std::min(own_len, n);
  // where this range's lines start: after the first terminator at or past `a` (the first range: at the first data line)
  size_t s = 0;
  if (!first_range) {
    const uint8_t *t = own_len ? (const uint8_t *)memchr(buf, eol, own_len) : nullptr;
    if (!t) {
      c.kind = BVCF_CUT_NONE;  // one line covers the whole range: it belongs to an earlier range
      return c;
    }
    s = (size_t)(t - buf) + 1;
  }
  c.start = s;
  if (last_range) {
    // the run's last line ends the range; an unterminated tail is dropped (main.go:354-358)
    const uint8_t *t = n > s ? (const uint8_t *)memrchr(buf + s, eol, n - s) : nullptr;
    c.end = t ? (size_t)(t - buf) + 1 : s;
    return c;
  }
  // ... and where they end: after the first terminator at or past `b` (the line that straddles the end is ours)
  const uint8_t *t = n > own_len ? (const uint8_t *)memchr(buf + own_len, eol, n - own_len) : nullptr;
  if (t) {
    c.end = (size_t)(t - buf) + 1;
    return c;
  }
  // the straddling line does not end within the spare room: the lines before it go as they are
  const uint8_t *tl = own_len > s ? (const uint8_t *)memrchr(buf + s, eol, own_len - s) : nullptr;
  c.kind = BVCF_CUT_LONG;
  c.long_start = tl ? (size_t)(tl - buf) + 1 : s;
  c.end = c.long_start;
  return c;
}

long find_block_chain(const uint8_t *buf, size_t n, size_t from) {
  for (size_t p = from; p + 18 <= n; p++) {
    if (buf[p] != 0x1f) {
      const uint8_t *q = (const uint8_t *)memchr(buf + p, 0x1f, n - p);
      if (!q) return -1;
      p = (size_t)(q - buf);
      if (p + 18 > n) return -1;
    }
    size_t at = p;
    bool ok = true;
    for (int hop = 0; hop < 3 && ok; hop++) {
      uint32_t xlen = 0;
      const long bs = bvcf_bgzf::block_size(buf + at, n - at, &xlen);
      if (bs < 0) ok = false;
      if (bs <= 0) break;  // 0: the header runs past the buffer
      if ((size_t)bs < 12 + (size_t)xlen + 8) ok = false;
      at += (size_t)bs;
      if (at >= n) break;
    }
    if (ok) return (long)p;
  }
  return -1;
}

int frame_at(const uint8_t *buf, size_t n, size_t off, Frame *f) {
  if (off >= n) return 0;
  uint32_t xlen = 0;
  const long bs = bvcf_bgzf::block_size(buf + off, n - off, &xlen);
  if (bs < 0) return -1;
  if (bs == 0 || off + (size_t)bs > n) return 0;
  if ((size_t)bs < 12 + (size_t)xlen + 8) return -1;
  const uint8_t *tail = buf + off + bs - 8;
  f->off = off;
  f->total = (uint32_t)bs;
  f->in_off = 12 + xlen;
  f->in_len = (uint32_t)(bs - 12 - xlen - 8);
  f->isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
  if (f->isize > (1u << 16)) return -1;
  return 1;
}

int block_has_eol(z_stream &zs, const uint8_t *p, uint32_t n, uint8_t eol) {
  uint8_t out[16384];
  inflateReset(&zs);
  zs.next_in = const_cast<uint8_t *>(p);
  zs.avail_in = n;
  for (;;) {
    zs.next_out = out;
    zs.avail_out = sizeof out;
    const int zr = inflate(&zs, Z_NO_FLUSH);
    if (zr != Z_OK && zr != Z_STREAM_END && zr != Z_BUF_ERROR) return -1;
    const size_t got = sizeof out - zs.avail_out;
    if (got && memchr(out, eol, got)) return 1;
    if (zr == Z_STREAM_END) return 0;
    if (zr == Z_BUF_ERROR && !got) return -1;  // the payload ends inside the stream
  }
}

// One batch of a BGZF input: the own blocks from `pos` on (those that start before `limit`, while their text leaves
// room for the look-ahead and their bytes fit `small`), then the look-ahead -- found, not guessed: the blocks after the
// own ones are inflated just far enough to see a terminator.  frame(off, &f) returns the block at window offset off
// (1), a clean end of the input (0) or a malformed / truncated block (-1), growing the window as it needs.
BgzfBatch cut_bgzf_batch(const std::function<int(size_t, Frame *)> &frame, const std::function<const uint8_t *(size_t)> &at,
                         z_stream &zs, size_t pos, size_t limit, size_t cap, size_t small, size_t la_reserve, uint8_t eol) {
  BgzfBatch b;
  std::vector<Frame> own;
  size_t q = pos;
  for (;;) {
    Frame f;
    if (q >= limit) break;
    const int r = frame(q, &f);
    if (r < 0) b.bad = true;
    if (r != 1) break;
    if (!own.empty() && (b.own_text + f.isize + la_reserve > cap ||
                         (small && b.own_bytes + f.total + (la_reserve >> 1) + (1u << 17) > small)))
      break;
    own.push_back(f);
    b.own_text += f.isize;
    b.own_bytes += f.total;
    q += f.total;
  }
  if (b.bad || own.empty()) return b;
  for (;;) {
    b.la = 0;
    b.la_text = 0;
    b.at_eof = false;
    for (;;) {
      Frame f;
      const int r = frame(pos + b.own_bytes + b.la, &f);
      if (r < 0) b.bad = true;
      if (r == 0) b.at_eof = true;
      if (r != 1) break;
      b.la += f.total;
      b.la_text += f.isize;
      const int he = f.isize ? block_has_eol(zs, at(f.off + f.in_off), f.in_len, eol) : 0;
      if (he < 0) b.bad = true;
      if (he != 0) break;
    }
    if (b.bad || (b.own_text + b.la_text <= cap && (!small || b.own_bytes + b.la <= small))) break;
    if (own.size() == 1) {
      b.too_long = true;  // "a line is longer than max_batch_bytes"
      break;
    }
    // the line that straddles the end of the own blocks is longer than the room kept for it: fewer own blocks
    while (own.size() > 1 && (b.own_text + b.la_text > cap || (small && b.own_bytes + b.la > small))) {
      b.own_text -= own.back().isize;
      b.own_bytes -= own.back().total;
      own.pop_back();
    }
  }
  b.n_own = own.size();
  return b;
}

// Host threads of a run.  What burns CPU is the readers' copies out of the page cache (about 10 GB/s per thread; one
// GPU's H2D link takes 46 GB/s) and the TSV assembly; device threads wait for the GPU, the sink writes.  The two
// kinds share the worker's part of the quota half and half: measured on a 16-core share with one device, 2 readers x 4
// copy threads + 8 formatter threads run the 63 GB of configs[2] as fast as 2 x 8 + 16 did, without
// asking the scheduler for twice the quota.  A BGZF file needs one reader thread per worker (80 x less to copy).
bvcf_thread_budget plan_threads(unsigned cpus, unsigned n_workers, int mode) {
  bvcf_thread_budget b;
  memset(&b, 0, sizeof b);
  cpus = std::max(1u, cpus);
  n_workers = std::max(1u, n_workers);
  const unsigned share = std::max(2u, cpus / n_workers);  // (fewer CPUs than 2 per worker: one reader + one formatter each)
  if (mode == BVCF_MODE_TEXT_RANGES) {
    const unsigned copy = std::max(1u, share / 2);
    b.readers = copy >= 2 ? 2u : 1u;
    b.copy_threads = std::min(8u, std::max(1u, copy / b.readers));
    b.format_threads = std::min(32u, std::max(1u, share - b.readers * b.copy_threads));
    b.busy_total = n_workers * (b.readers * b.copy_threads + b.format_threads);
  } else if (mode == BVCF_MODE_BGZF_RANGES) {
    b.readers = 1;
    b.copy_threads = 1;
    b.format_threads = std::min(32u, std::max(1u, share - 1));
    b.busy_total = n_workers * (1 + b.format_threads);
  } else {
    // one reader for the whole stream; a text pipe's pages are copied out by a few threads beside it
    // (bvcf_input::ByteSource::fanout_threads: 4 from 12 CPUs up, 2 from 6, none below)
    b.readers = 0;
    const unsigned fan = cpus >= 12 ? 4u : (cpus >= 6 ? 2u : 0u);
    b.copy_threads = fan ? fan : 1u;
    b.format_threads = std::min(32u, std::max(1u, cpus > n_workers + fan ? (cpus - 1 - fan) / n_workers : 1u));
    b.busy_total = 1 + fan + n_workers * b.format_threads;
  }
  return b;
}

}  // namespace bvcf_host

using namespace bvcf_host;

extern "C" {

int bvcf_plan_text_ranges(uint64_t file_size, uint64_t data_off, uint64_t max_batch_bytes, uint64_t first_line_bytes,
                          bvcf_range_plan *out) {
  if (!out || data_off > file_size) return BVCF_E_ARG;
  *out = plan_text_ranges(file_size, data_off, max_batch_bytes, first_line_bytes);
  return BVCF_OK;
}

int bvcf_plan_bgzf_ranges(uint64_t file_size, uint64_t data_off, uint32_t n_workers, uint64_t max_batch_bytes, bvcf_range_plan *out) {
  if (!out || data_off > file_size || !n_workers) return BVCF_E_ARG;
  *out = plan_bgzf_ranges(file_size, data_off, n_workers, max_batch_bytes);
  return BVCF_OK;
}

int bvcf_cut_text_range(const uint8_t *window, uint64_t n, uint64_t own_len, int is_first, int is_last, uint8_t eol, bvcf_text_cut *out) {
  if (!out || (!window && n)) return BVCF_E_ARG;
  *out = cut_text_range(window, (size_t)n, (size_t)own_len, is_first != 0, is_last != 0, eol);
  return BVCF_OK;
}

long bvcf_find_bgzf_chain(const uint8_t *buf, size_t n, size_t from) { return buf ? find_block_chain(buf, n, from) : -1; }

int bvcf_plan_threads(uint32_t cpus, uint32_t n_workers, int mode, bvcf_thread_budget *out) {
  if (!out || !n_workers || mode < 0 || mode > 2) return BVCF_E_ARG;
  *out = plan_threads(cpus, n_workers, mode);
  return BVCF_OK;
}

}  // extern "C"
