// bvcf_readers.cpp — the input side of bvcf_run_fd: the header, and the reader threads that turn the input into blocks of
// whole lines (text) or whole BGZF blocks for their device workers.  The counterpart of readVcf's preamble and producer
// loop (main.go:250-304, 349-380), replicated per device for a regular file.  Which bytes a range owns is decided by the
// pure functions of bvcf_plan.cpp; what is left here is the I/O around them.
#include "bvcf_pipeline.h"

namespace bvcf_host {

namespace {

Block text_block(std::shared_ptr<BufHold> hold, const uint8_t *p, size_t n, uint64_t file_off, uint64_t range, uint32_t piece, bool last) {
  Block b;
  b.hold = std::move(hold);
  b.data = p;
  b.nb = n;
  b.file_off = file_off;
  b.range = range;
  b.piece = piece;
  b.last_piece = last;
  return b;
}

// the BGZF blocks at the head of an input, inflated one at a time until the preamble is complete
struct HeadInflater {
  std::vector<uint8_t> text;
  std::vector<std::pair<size_t, size_t>> marks;  // (compressed offset, text offset) of each inflated block
  size_t hoff = 0;                               // compressed bytes consumed
  z_stream zs;
  bool ok;
  HeadInflater() {
    memset(&zs, 0, sizeof zs);
    ok = inflateInit2(&zs, -15) == Z_OK;
  }
  ~HeadInflater() {
    if (ok) inflateEnd(&zs);
  }
  // the block f of the compressed bytes at blk: true if it inflates to its ISIZE with its CRC
  bool add(const uint8_t *blk, const Frame &f) {
    marks.emplace_back(hoff, text.size());
    const size_t at = text.size();
    text.resize(at + f.isize);
    inflateReset(&zs);
    zs.next_in = const_cast<uint8_t *>(blk + f.in_off);
    zs.avail_in = f.in_len;
    zs.next_out = text.data() + at;
    zs.avail_out = f.isize;
    const int zr = f.isize ? inflate(&zs, Z_FINISH) : Z_STREAM_END;
    const uint8_t *tail = blk + f.total - 8;
    const uint32_t want_crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
    if ((f.isize && (zr != Z_STREAM_END || zs.avail_out != 0)) || (uint32_t)crc32(crc32(0L, Z_NULL, 0), text.data() + at, f.isize) != want_crc)
      return false;
    hoff += f.total;
    return true;
  }
  // the block that holds text offset data_off: its compressed offset and data_off's place in its text
  void locate(size_t data_off, size_t *comp_off, uint32_t *first_off) const {
    *comp_off = hoff;
    *first_off = 0;
    for (size_t i = 0; i < marks.size(); i++) {
      const size_t t_end = i + 1 < marks.size() ? marks[i + 1].second : text.size();
      if (data_off < t_end) {
        *comp_off = marks[i].first;
        *first_off = (uint32_t)(data_off - marks[i].second);
        return;
      }
    }
  }
};

}  // namespace

ssize_t pread_parallel(int fd, uint8_t *dst, size_t n, off_t off, unsigned n_thr, int *err) {
  n_thr = (unsigned)std::max<size_t>(1, std::min<size_t>(n_thr, n >> 21));
  std::vector<size_t> got_n(n_thr, 0);
  std::vector<int> err_n(n_thr, 0);
  auto part = [&](unsigned t) {
    const size_t lo = n * t / n_thr, hi = n * (t + 1) / n_thr;
    size_t done = 0;
    while (lo + done < hi) {
      const ssize_t g = pread(fd, dst + lo + done, hi - lo - done, off + (off_t)(lo + done));
      if (g < 0 && errno == EINTR) continue;
      if (g < 0) err_n[t] = errno;
      if (g <= 0) break;
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
      *err = err_n[t];
      return -1;
    }
    total += got_n[t];
    if (got_n[t] != n * (t + 1) / n_thr - n * t / n_thr) break;  // short part: what follows it is not contiguous
  }
  return (ssize_t)total;
}

// ---- the header of a regular file (range modes): this thread, while the workers warm their devices up

bool Driver::read_text_header() {
  const size_t total = (size_t)(file_size_ - file_base_);
  std::string msg;
  std::vector<uint8_t> head;
  size_t look = 1u << 20;
  for (;;) {
    look = std::min<size_t>(look, total);
    head.resize(look);
    int err = 0;
    const ssize_t g = pread_parallel(fd_in_, head.data(), look, file_base_, 1, &err);
    if (g < 0) {
      fail(std::string("read: ") + strerror(err), BVCF_E_FATAL);
      return false;
    }
    head.resize((size_t)g);
    R_.pre = Preamble();
    // (as the single reader does it: a header that does not end within max_batch_bytes is "No header found")
    const bool all = head.size() >= total || head.size() >= cap_;
    const int pr = parse_preamble(head.data(), head.size(), all, c_->normalize_header, &R_.pre, &msg);
    if (pr < 0) {
      fail(msg, BVCF_E_FATAL);
      return false;
    }
    if (pr == 0) break;
    look *= 4;
  }
  // some data lines for prepare_run (the shape of the lines, the reservation)
  const size_t want = std::min<size_t>(total, R_.pre.data_off + (4u << 20));
  if (head.size() < want) {
    const size_t old = head.size();
    head.resize(want);
    int err = 0;
    const ssize_t g2 = pread_parallel(fd_in_, head.data() + old, want - old, file_base_ + (off_t)old, 1, &err);
    head.resize(old + (g2 > 0 ? (size_t)g2 : 0));
  }
  const uint8_t *d = head.data() + R_.pre.data_off;
  const size_t nd = head.size() - R_.pre.data_off;
  const uint8_t *e = nd ? (const uint8_t *)memchr(d, R_.pre.eol_byte, nd) : nullptr;
  const size_t first_line = e ? (size_t)(e - d) + 1 : nd;
  plan_.p = plan_text_ranges((uint64_t)file_size_, (uint64_t)file_base_ + R_.pre.data_off, cap_, first_line);
  return adopt_preamble(d, nd);
}

bool Driver::read_bgzf_header() {
  // the leading blocks are inflated here until the #CHROM line is complete
  const size_t total = (size_t)(file_size_ - file_base_);
  HeadInflater hi;
  std::vector<uint8_t> comp;
  std::string msg;
  bool bad = !hi.ok, at_end = false;
  auto inflate_next = [&]() -> int {  // 1 = a block was inflated, 0 = end of input, -1 = bad
    for (;;) {
      Frame f;
      const int r = frame_at(comp.data(), comp.size(), hi.hoff, &f);
      if (r < 0) return -1;
      if (r == 1) return hi.add(comp.data() + hi.hoff, f) ? 1 : -1;
      if (comp.size() >= total) return hi.hoff >= comp.size() ? 0 : -1;
      const size_t old = comp.size(), step = std::min<size_t>(4u << 20, total - old);
      comp.resize(old + step);
      int err = 0;
      const ssize_t g = pread_parallel(fd_in_, comp.data() + old, step, file_base_ + (off_t)old, 1, &err);
      comp.resize(old + (g > 0 ? (size_t)g : 0));
      if (g <= 0) return -1;
    }
  };
  int pr = 1;
  while (!bad && pr == 1) {
    const int ir = inflate_next();
    if (ir < 0) {
      bad = true;
      break;
    }
    at_end = ir == 0;
    R_.pre = Preamble();
    pr = parse_preamble(hi.text.data(), hi.text.size(), at_end, c_->normalize_header, &R_.pre, &msg);
    if (at_end) break;
  }
  if (bad) {
    fail("bgzf: corrupt block (inflate or CRC mismatch)", BVCF_E_FATAL);
    return false;
  }
  if (pr != 0) {
    fail(pr < 0 ? msg : std::string("No header found"), BVCF_E_FATAL);
    return false;
  }
  const size_t data_off = R_.pre.data_off;
  // a few data lines for prepare_run (path choice, reservation): make sure at least one whole line is in view
  for (int extra = 0; extra < 8; extra++) {
    if (memchr(hi.text.data() + data_off, R_.pre.eol_byte, hi.text.size() - data_off)) break;
    if (inflate_next() != 1) break;
  }
  size_t c0 = 0;
  hi.locate(data_off, &c0, &plan_.first_off);
  // (range_bytes was chosen before the header was read, bgzf_range_bytes: the buffers are being pinned meanwhile)
  plan_.p.n_ranges = ((total > c0 ? total - c0 : 0) + plan_.p.range_bytes - 1) / plan_.p.range_bytes;
  plan_.p.data_off = (uint64_t)file_base_ + c0;
  // one more batch in flight per device than for text: two batches' blocks inflate side by side while a third is in its
  // kernel chain / on its way back
  R_.n_slots = (uint32_t)bgzf_in_flight_ + 1;
  max_in_flight_.store(bgzf_in_flight_);
  return adopt_preamble(hi.text.data() + data_off, hi.text.size() - data_off);
}

bool Driver::read_file_header() { return mode_ == kRangeText ? read_text_header() : read_bgzf_header(); }

// the worker's readers hand their ranges over in the order of the ranges (see OrderedSink)
void Driver::push_in_turn(DevWorker *W, uint64_t local, std::vector<Block> &blocks) {
  {
    std::unique_lock<std::mutex> lk(W->mu);
    W->cv.wait(lk, [&] { return W->next_push == local || failed_.load(); });
  }
  if (!failed_.load())
    for (Block &b : blocks) W->q.push(std::move(b));
  blocks.clear();
  {
    std::lock_guard<std::mutex> lk(W->mu);
    if (W->next_push == local) W->next_push = local + 1;
  }
  W->cv.notify_all();
}

// ---- range mode, text: worker k reads ranges k, k + N, ... of the file into its own pinned buffers
void Driver::range_text_reader(DevWorker *W, unsigned r) {
  if (several_devices_) bind_here(W->cpus);
  wait_plan();
  if (failed_.load()) return;
  const uint8_t eol = eol_byte_.load();
  const size_t Rb = (size_t)plan_.p.range_bytes;
  double t_read = 0;
  // (with a dosage file a block keeps its buffer until it is its turn in the output: a second reader running ahead
  // could then hold every buffer while the first one waits for one -- a single reader takes the ranges in order)
  const unsigned stride = R_.arrow ? 1u : std::max(1u, budget_.readers);
  if (r >= stride) return;
  std::vector<Block> out;
  for (uint64_t j = r; !failed_.load(); j += stride) {
    const uint64_t i = W->idx + (uint64_t)n_dev_ * j;
    if (i >= plan_.p.n_ranges) break;
    const off_t a = (off_t)plan_.p.data_off + (off_t)(i * Rb);
    const off_t b = std::min<off_t>(a + (off_t)Rb, file_size_);
    auto hold = W->pool->get();
    if (!hold) {
      if (!failed_.load()) fail("cannot allocate pinned host memory", BVCF_E_NOMEM);
      break;
    }
    const double t0 = now_s();
    const size_t want = (size_t)std::min<off_t>((off_t)(Rb + plan_.p.spare_bytes), file_size_ - a);
    int err = 0;
    const ssize_t got = pread_parallel(fd_in_, hold->p, want, a, budget_.copy_threads, &err);
    t_read += now_s() - t0;
    if (got < (ssize_t)want) {
      fail(got < 0 ? std::string("read: ") + strerror(err) : std::string("read: the input file got shorter"), BVCF_E_FATAL);
      break;
    }
    const uint8_t *buf = hold->p;
    const size_t n = (size_t)got;
    const bvcf_text_cut cut = cut_text_range(buf, n, (size_t)(b - a), i == 0, b >= file_size_, eol);
    const size_t s = (size_t)cut.start, e = (size_t)cut.end;
    if (cut.kind == BVCF_CUT_NONE) {
      out.push_back(text_block(nullptr, nullptr, 0, (uint64_t)a, i, 0, true));
    } else if (cut.kind == BVCF_CUT_LINES) {
      out.push_back(text_block(e > s ? hold : nullptr, buf + s, e - s, (uint64_t)a + s, i, 0, true));
    } else {
      // The straddling line does not end within the buffer's spare room: the lines before it go as they are, the long
      // line is read into memory of its own (up to max_batch_bytes, as for the single reader).
      const size_t s_long = (size_t)cut.long_start;
      auto big = std::make_shared<BufHold>();
      big->heap.assign(buf + s_long, buf + n);
      bool found = false, too_long = false, at_eof = false;
      off_t pos = a + (off_t)n;
      while (!found && !too_long && !at_eof) {
        const size_t old = big->heap.size(), step = 4u << 20;
        big->heap.resize(old + step);
        int e2 = 0;
        const ssize_t g = pread_parallel(fd_in_, big->heap.data() + old, (size_t)std::min<off_t>((off_t)step, file_size_ - pos), pos, 1, &e2);
        if (g < 0) {
          fail(std::string("read: ") + strerror(e2), BVCF_E_FATAL);
          break;
        }
        big->heap.resize(old + (size_t)g);
        pos += g;
        if (g == 0 || pos >= file_size_) at_eof = true;
        const uint8_t *t2 = g > 0 ? (const uint8_t *)memchr(big->heap.data() + old, eol, (size_t)g) : nullptr;
        if (t2) {
          big->heap.resize((size_t)(t2 - big->heap.data()) + 1);
          found = true;
        } else if (big->heap.size() > cap_) {
          too_long = true;
        }
      }
      if (failed_.load()) break;
      if (too_long || (found && big->heap.size() > cap_)) {
        fail("a line is longer than max_batch_bytes", BVCF_E_TOO_BIG);
        break;
      }
      out.push_back(text_block(s_long > s ? hold : nullptr, buf + s, s_long - s, (uint64_t)a + s, i, 0, false));
      if (found) {
        const uint8_t *p = big->heap.data();
        const size_t nb = big->heap.size();
        out.push_back(text_block(big, p, nb, (uint64_t)a + s_long, i, 1, true));
      } else {
        out.push_back(text_block(nullptr, nullptr, 0, (uint64_t)a + s_long, i, 1, true));  // the file ends inside the line: dropped
      }
    }
    hold.reset();
    push_in_turn(W, j, out);
  }
  // (the pool's allocator threads stop pinning more buffers once nobody will ask for one: only when the LAST reader is
  // through -- a reader with no range of its own must not end the pool under its sibling, which may not hold a buffer yet)
  if (W->readers_done.fetch_add(1) + 1 == stride) W->pool->stop();
  {
    std::lock_guard<std::mutex> lk(W->mu);
    W->t_read += t_read;
  }
  W->cv.notify_all();
}

// ---- range mode, BGZF: worker k reads compressed ranges k, k + N, ... and cuts them into batches of whole blocks for
// bvcf_submit_bgzf (cut_bgzf_batch)
void Driver::range_bgzf_reader(DevWorker *W) {
  if (several_devices_) bind_here(W->cpus);
  wait_plan();
  if (failed_.load()) return;
  const uint8_t eol = eol_byte_.load();
  const size_t Rb = (size_t)plan_.p.range_bytes;
  const size_t buf_bytes = W->pool->bytes();
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  if (inflateInit2(&zs, -15) != Z_OK) {
    fail("inflateInit2 failed", BVCF_E_FATAL);
    return;
  }
  size_t la_reserve = 4u << 16;  // text kept free for the look-ahead when a batch's own blocks are chosen
  for (uint64_t i = W->idx; i < plan_.p.n_ranges && !failed_.load(); i += n_dev_) {
    const off_t a = (off_t)plan_.p.data_off + (off_t)(i * Rb);
    const off_t b = std::min<off_t>(a + (off_t)Rb, file_size_);
    auto hold = W->pool->get();
    if (!hold) {
      if (!failed_.load()) fail("cannot allocate pinned host memory", BVCF_E_NOMEM);
      break;
    }
    const double t0 = now_s();
    const size_t want = (size_t)std::min<off_t>((off_t)buf_bytes, file_size_ - a);
    int err = 0;
    const ssize_t got = pread_parallel(fd_in_, hold->p, want, a, budget_.copy_threads, &err);
    W->t_read += now_s() - t0;
    if (got < (ssize_t)want) {
      fail(got < 0 ? std::string("read: ") + strerror(err) : std::string("read: the input file got shorter"), BVCF_E_FATAL);
      break;
    }
    // the window: the pinned buffer; if a batch's look-ahead runs past it, a copy in memory of its own that grows
    const uint8_t *win = hold->p;
    size_t win_n = (size_t)got;
    std::shared_ptr<BufHold> big;  // set once the window has moved
    auto window_reaches_eof = [&]() { return a + (off_t)win_n >= file_size_; };
    auto grow_window = [&]() -> bool {
      if (window_reaches_eof()) return false;
      auto nb = std::make_shared<BufHold>();
      const size_t step = 8u << 20;
      nb->heap.resize(win_n + step);
      memcpy(nb->heap.data(), win, win_n);
      int e2 = 0;
      const ssize_t g = pread_parallel(fd_in_, nb->heap.data() + win_n, (size_t)std::min<off_t>((off_t)step, file_size_ - a - (off_t)win_n),
                                       a + (off_t)win_n, 1, &e2);
      if (g <= 0) return false;
      nb->heap.resize(win_n + (size_t)g);
      big = nb;
      win = big->heap.data();
      win_n = big->heap.size();
      return true;
    };
    const size_t own_len = (size_t)(b - a);
    // the first block that starts in [a, b): ours from there on
    size_t p0 = 0;
    if (i > 0) {
      const long f = find_block_chain(win, win_n, 0);
      if (f < 0 || (size_t)f >= own_len) {
        if (f < 0 && own_len > (1u << 17)) {
          fail("bgzf: not a BGZF block, or a truncated file", BVCF_E_FATAL);
          break;
        }
        W->q.push(text_block(nullptr, nullptr, 0, (uint64_t)a, i, 0, true));  // no block starts in this range
        continue;
      }
      p0 = (size_t)f;
    }
    // frame of the block at window offset off, growing the window when it ends inside the block
    auto frame = [&](size_t off, Frame *f) -> int {
      for (;;) {
        const int r = frame_at(win, win_n, off, f);
        if (r != 0) return r;
        if (off >= win_n && window_reaches_eof()) return 0;  // a clean end of the input
        if (!grow_window()) return -1;                       // the file ends inside a block
      }
    };
    auto at = [&](size_t off) { return win + off; };
    size_t pos = p0;
    uint32_t piece = 0;
    bool first_batch = true, bad = false;
    while (!bad && !failed_.load()) {
      const BgzfBatch bt = cut_bgzf_batch(frame, at, zs, pos, own_len, cap_, 0, la_reserve, eol);
      if (bt.too_long) {
        fail("a line is longer than max_batch_bytes", BVCF_E_TOO_BIG);
        break;
      }
      if (bt.bad) {
        bad = true;
        break;
      }
      if (!bt.n_own) {
        // (only when the range's last block ended exactly at `b` on the previous batch: close the range)
        W->q.push(text_block(nullptr, nullptr, 0, (uint64_t)a + pos, i, piece, true));
        break;
      }
      la_reserve = std::max(la_reserve, std::min<size_t>(2 * bt.la_text, cap_ / 2));
      const size_t next = pos + bt.own_bytes;
      // (the end-of-file marker block and anything else without text after the last terminator: nothing follows)
      const bool last = next >= own_len || (window_reaches_eof() && next >= win_n);
      Block blk;
      blk.hold = big ? big : hold;
      blk.data = win + pos;
      blk.nb = bt.own_bytes + bt.la;
      blk.own = bt.own_bytes;
      blk.bgzf = true;
      blk.bgzf_flags = ((i == 0 && first_batch) ? 0 : BVCF_BGZF_SKIP_FIRST_LINE) | (bt.at_eof ? BVCF_BGZF_END_OF_STREAM : 0);
      blk.first_off = (i == 0 && first_batch) ? plan_.first_off : 0;
      blk.file_off = (uint64_t)a + pos;
      blk.range = i;
      blk.piece = piece++;
      blk.last_piece = last;
      W->q.push(blk);
      first_batch = false;
      pos = next;
      if (last) break;
    }
    if (bad && !failed_.load()) fail("bgzf: not a BGZF block, or a truncated file", BVCF_E_FATAL);
    if (i + n_dev_ >= plan_.p.n_ranges) W->pool->stop();
  }
  inflateEnd(&zs);
  W->pool->stop();
}

// ---- stream mode: ONE reader cuts the blocks (pipes, single-stream gzip), the run's thread deals them

void Driver::push_end(bool read_error, bool too_long) {
  Block e;
  e.end = true;
  e.bgzf_flags = (read_error ? 1 : 0) | (too_long ? 2 : 0);  // (on an end marker: why the stream ended early)
  ready_q_->push(e);
}

// BGZF on a pipe, inflated on the device: whole compressed blocks per buffer.  The header has to be read here, so the
// leading blocks are inflated with zlib until the #CHROM line is complete; everything from the block that holds the
// first data line on is handed over compressed, each batch with the following blocks as look-ahead.
void Driver::stream_bgzf(bvcf_input::ByteSource &src) {
  input_is_bgzf_device_.store(true);
  // (batches of 256 MiB of text, as for a BGZF file: a batch takes as long as its slowest block however few it has)
  if (!c_->max_batch_bytes) R_.max_batch = cap_ = 256ull << 20;
  std::vector<uint8_t> pend;  // compressed bytes read from the input; pend[pp..] not yet handed over
  size_t pp = 0;
  bool raw_eof = false;
  auto more = [&]() -> bool {
    if (raw_eof) return false;
    if (pp > (32u << 20)) {
      pend.erase(pend.begin(), pend.begin() + (ptrdiff_t)pp);
      pp = 0;
    }
    const size_t old = pend.size(), step = 8u << 20;
    pend.resize(old + step);
    const ssize_t got = src.read_raw(pend.data() + old, step);
    pend.resize(old + (got > 0 ? (size_t)got : 0));
    if (got <= 0) {
      raw_eof = true;
      if (got < 0) source_err_ = src.error();
      return false;
    }
    return true;
  };
  // frame of the block at pend[pp + off]: 1, 0 at a clean end of input, -1 malformed / truncated / read error
  auto frame = [&](size_t off, Frame *f) -> int {
    for (;;) {
      const int r = frame_at(pend.data() + pp, pend.size() - pp, off, f);
      if (r != 0) return r;
      const bool had = pend.size() - pp > off;
      if (!more()) return (!source_err_.empty() || had) ? -1 : 0;
    }
  };
  auto at = [&](size_t off) -> const uint8_t * { return pend.data() + pp + off; };
  auto fail_read = [&](const std::string &m) {
    source_err_ = m;
    push_end(true, false);
  };
  // ---- the header, from blocks inflated here
  HeadInflater hi;
  std::string msg;
  if (!hi.ok) return fail_read("inflateInit2 failed");
  auto inflate_next = [&]() -> int {  // 1 = a block was inflated, 0 = end of input, -1 = bad
    Frame f;
    const int r = frame(hi.hoff, &f);
    if (r <= 0) return r;
    return hi.add(pend.data() + pp + hi.hoff, f) ? 1 : -1;
  };
  int pr = 1;
  while (pr == 1) {
    const int ir = inflate_next();
    if (ir < 0) return fail_read(source_err_.empty() ? std::string("bgzf: corrupt block (inflate or CRC mismatch)") : source_err_);
    const bool hdr_eof = ir == 0;
    pr = parse_preamble(hi.text.data(), hi.text.size(), hdr_eof, c_->normalize_header, &R_.pre, &msg);
    if (hdr_eof) break;
  }
  if (pr != 0) {
    fail(pr < 0 ? msg : std::string("No header found"), BVCF_E_FATAL);
    return push_end(false, false);
  }
  const size_t data_off = R_.pre.data_off;
  for (int extra = 0; extra < 8; extra++) {
    if (memchr(hi.text.data() + data_off, R_.pre.eol_byte, hi.text.size() - data_off)) break;
    if (inflate_next() != 1) break;
  }
  R_.n_slots = (uint32_t)bgzf_in_flight_ + 1;
  max_in_flight_.store(bgzf_in_flight_);
  if (!adopt_preamble(hi.text.data() + data_off, hi.text.size() - data_off)) return push_end(false, false);
  const uint8_t eol = R_.pre.eol_byte;
  size_t c0 = 0;
  uint32_t first_off = 0;
  hi.locate(data_off, &c0, &first_off);
  uint64_t consumed = c0;  // compressed bytes of the input in front of pend[pp]
  pp += c0;
  // the compressed bytes of a batch (own + look-ahead blocks) go into pinned buffers a quarter of the text's size
  const size_t small = std::max<size_t>(cap_ / 4, 1u << 20);
  stream_pool_.reset(new BufPool(dry_ ? -1 : dev_list_[0], small, (int)std::min<size_t>(4 * n_dev_ + 4, 64)));
  stream_pool_->start();
  plan_ready();
  size_t la_reserve = 4u << 16;
  bool first = true;
  uint64_t seq = 0;
  for (;;) {
    if (stop_.load() || failed_.load()) break;
    auto hold = stream_pool_->get();
    if (!hold) break;
    const BgzfBatch bt = cut_bgzf_batch(frame, at, hi.zs, 0, (size_t)-1, cap_, small, la_reserve, eol);
    if (bt.bad) return fail_read(source_err_.empty() ? std::string("bgzf: not a BGZF block, or a truncated file") : source_err_);
    if (bt.too_long) return push_end(false, true);
    if (!bt.n_own) break;  // a clean end of the input
    la_reserve = std::max(la_reserve, std::min<size_t>(2 * bt.la_text, cap_ / 2));
    memcpy(hold->p, pend.data() + pp, bt.own_bytes + bt.la);
    Block b;
    b.data = hold->p;
    b.hold = std::move(hold);
    b.nb = bt.own_bytes + bt.la;
    b.own = bt.own_bytes;
    b.bgzf = true;
    b.bgzf_flags = (first ? 0 : BVCF_BGZF_SKIP_FIRST_LINE) | (bt.at_eof ? BVCF_BGZF_END_OF_STREAM : 0);
    b.first_off = first ? first_off : 0;
    b.file_off = consumed;
    b.range = seq++;
    pp += bt.own_bytes;
    consumed += bt.own_bytes;
    first = false;
    ready_q_->push(b);
    if (bt.at_eof && bt.la == 0) break;
  }
  stream_pool_->stop();
  push_end(false, false);
}

void Driver::stream_reader() {
  bvcf_input::ByteSource src(fd_in_, std::min(32u, hw_));
  {
    // BGZF input (bgzip / htslib .vcf.gz): the blocks go to the device compressed and are inflated there
    // (bvcf_submit_bgzf) unless BVCF_DEVICE_INFLATE=0 (then this thread's workers inflate them with zlib)
    const char *e = getenv("BVCF_DEVICE_INFLATE");
    const bool device_inflate = dry_ ? dry_device_inflate_ != 0 : !(e && *e == '0');
    if (device_inflate && src.sniff_bgzf()) {
      stream_bgzf(src);
      return;
    }
  }
  // being read into, two on each device, up to two with each formatter, one spare
  stream_pool_.reset(new BufPool(dry_ ? -1 : dev_list_[0], cap_, (int)std::min<size_t>(4 * n_dev_ + 3, 64)));
  stream_pool_->start(1);
  std::vector<uint8_t> carry;
  bool first = true, eof = false;
  uint64_t seq = 0, consumed = 0;  // consumed: text bytes of the stream in front of the buffer being filled (less the carry)
  uint8_t eol = '\n';
  while (!eof && !stop_.load() && !failed_.load()) {
    auto hold = stream_pool_->get();
    if (!hold) {
      if (stream_pool_->failed()) fail("cannot allocate pinned host memory", BVCF_E_NOMEM);
      break;
    }
    uint8_t *buf = hold->p;
    size_t fill = carry.size();
    if (fill) memcpy(buf, carry.data(), fill);
    const uint64_t buf_off = consumed - fill;
    carry.clear();
    bool read_error = false;
    while (!eof && fill < cap_) {
      ssize_t got = src.read(buf + fill, cap_ - fill);
      if (got == bvcf_input::ByteSource::kNoRoom) break;  // this buffer is as full as it gets
      if (got < 0) {
        source_err_ = src.error();
        read_error = true;
        eof = true;
        break;
      }
      if (got == 0) {
        eof = true;
        break;
      }
      fill += (size_t)got;
      consumed += (uint64_t)got;
    }
    if (eof) stream_pool_->stop();
    if (read_error) return push_end(true, false);
    size_t start = 0;
    if (first) {
      // readVcf's preamble (main.go:250-304); the terminator is learnt from line 1 (parse.FindEndOfLine)
      std::string msg;
      const int pr = parse_preamble(buf, fill, true, c_->normalize_header, &R_.pre, &msg);
      if (pr != 0) {
        fail(msg, BVCF_E_FATAL);
        return push_end(false, false);
      }
      start = R_.pre.data_off;
      if (!adopt_preamble(buf + start, fill > start ? fill - start : 0)) return push_end(false, false);
      eol = R_.pre.eol_byte;
      plan_ready();
      first = false;
    }
    const uint8_t *lastp = fill > start ? (const uint8_t *)memrchr(buf + start, eol, fill - start) : nullptr;
    if (!lastp) {
      if (!eof && fill == cap_) return push_end(false, true);
      // at EOF an unterminated tail is dropped (main.go:354-358); otherwise the line continues in the next buffer
      if (!eof) carry.assign(buf + start, buf + fill);
      continue;
    }
    const size_t nb = (size_t)(lastp - buf) + 1 - start;
    if (!eof) carry.assign(buf + start + nb, buf + fill);
    ready_q_->push(text_block(std::move(hold), buf + start, nb, buf_off + start, seq++, 0, true));
  }
  stream_pool_->stop();
  push_end(false, false);
}

}  // namespace bvcf_host
