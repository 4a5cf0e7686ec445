// bvcf_driver.cpp — bvcf_run_fd: the reference's main() + readVcf (main.go:134-217, 241-396) as a pipeline that is
// replicated per device.
//
// The reference has ONE producer (readVcf's loop, main.go:349-380) feeding NumCPU workers (main.go:345-347).  One
// producer cannot feed several GPUs (it tops out at one PCIe link's worth of text), so here everything between the
// input file and the ordered output belongs to a device worker:
//
//   per device worker k        reader thread     its own byte ranges of the file -> its own pinned buffers
//                              device thread     one ctx: bvcf_submit one block ahead, bvcf_collect the oldest
//                              formatter thread  TSV assembly (main.go:566-695) on the worker's thread pool
//   one for the run            ordered sink      rows and log lines to fd_out / fd_err in input order (main.go:524-532)
//
// Range mode (the input is a regular file, text or BGZF): range i of the data goes to worker i mod N.  Ranges are cut
// at fixed byte offsets, so their ends fall inside lines (BGZF: inside blocks, and the blocks' ends inside lines);
// the rule that gives every line exactly one owner needs no communication: with T(x) = the first terminator at an
// offset >= x, range [a, b) owns the bytes (T(a), T(b)] -- it skips the line it starts in and owns the line that
// straddles its end (the rule k_cuts applies to BGZF batches).  The first range starts where the data lines start.
// Stream mode (a pipe, a single-stream gzip): one reader thread cuts the blocks and this thread deals block k to
// worker k mod N, as the reference's single producer does.
// Either way the output is the same bytes in the same order for any device list: blocks are ordered by (range, piece).
#include "bvcf_host_internal.h"

#include <ctype.h>
#include <sched.h>
#include <sys/stat.h>

#include <set>

namespace {

typedef std::vector<std::string> Parts;

// a pinned buffer on loan from a pool (it goes home when the last block cut from it is done with), or bytes of its own
struct BufHold {
  uint8_t *p = nullptr;
  Channel<uint8_t *> *home = nullptr;
  std::vector<uint8_t> heap;
  ~BufHold() {
    if (home && p) home->push(p);
  }
};

// one block: whole lines of text, or whole BGZF blocks (own + look-ahead) for bvcf_submit_bgzf
struct Block {
  std::shared_ptr<BufHold> hold;
  const uint8_t *data = nullptr;
  size_t nb = 0;
  // the order of the output: (range, piece); last_piece closes the range
  uint64_t range = 0;
  uint32_t piece = 0;
  bool last_piece = true;
  bool bgzf = false;
  size_t own = 0;
  int bgzf_flags = 0;
  uint32_t first_off = 0;
  bool end = false;  // queue terminator
};

// pinned buffers of one size, allocated in the background (pinning 64 MiB takes ~25 ms: the first block is being read
// while the next buffers are pinned), at most `max` of them, none after stop()
class BufPool {
 public:
  BufPool(int device, size_t bytes, int max) : device_(device), bytes_(bytes), max_(max), free_(1024) {}
  ~BufPool() { join(); }
  void start(int n_threads = 2) {
    for (int t = 0; t < n_threads; t++)
      th_.emplace_back([this]() {
        for (;;) {
          if (stop_.load()) return;
          const int i = next_.fetch_add(1);
          if (i >= max_) return;
          uint8_t *p = (uint8_t *)bvcf_alloc_pinned_near(device_, bytes_);
          if (!p) {
            failed_.store(true);
            free_.push(nullptr);
            return;
          }
          {
            std::lock_guard<std::mutex> lk(mu_);
            all_.push_back(p);
          }
          free_.push(p);
        }
      });
  }
  // a free buffer (nullptr: pinning failed, or unblock() was called)
  std::shared_ptr<BufHold> get() {
    uint8_t *p = free_.pop();
    if (!p) {
      free_.push(nullptr);  // the next caller sees it too
      return nullptr;
    }
    auto h = std::make_shared<BufHold>();
    h->p = p;
    h->home = &free_;
    return h;
  }
  void stop() { stop_.store(true); }
  void unblock() { free_.push(nullptr); }
  void join() {
    for (auto &t : th_)
      if (t.joinable()) t.join();
  }
  bool failed() const { return failed_.load(); }
  size_t bytes() const { return bytes_; }
  void free_all() {
    join();
    std::vector<std::thread> th;
    for (uint8_t *p : all_) th.emplace_back([p]() { bvcf_free_pinned(p); });
    for (auto &t : th) t.join();
    all_.clear();
  }

 private:
  int device_;
  size_t bytes_;
  int max_;
  Channel<uint8_t *> free_;
  std::vector<std::thread> th_;
  std::atomic<int> next_{0};
  std::atomic<bool> stop_{false}, failed_{false};
  std::mutex mu_;
  std::vector<uint8_t *> all_;
};

// ---- the ordered output (main.go:524-532, 705-711): rows of block (range, piece) go out when every earlier block's have
struct OutItem {
  uint64_t range = 0;
  uint32_t piece = 0;
  bool last_piece = true;
  Parts *parts = nullptr;
  std::string log;
  std::function<void()> in_order;  // runs in output order once the rows are out (the dosage rows of the block)
};

class OrderedSink {
 public:
  OrderedSink(int fd_out, int fd_err) : fd_out_(fd_out), fd_err_(fd_err) {}
  ~OrderedSink() {
    for (auto &kv : held_) delete kv.second.parts;
    for (Parts *p : spares_) delete p;
  }
  void start() {
    th_ = std::thread([this]() { loop(); });
  }
  void put(OutItem &&it) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      held_.emplace(std::make_pair(it.range, it.piece), std::move(it));
    }
    cv_.notify_all();
  }
  Parts *spare() {
    std::lock_guard<std::mutex> lk(mu_);
    if (spares_.empty()) return new Parts();
    Parts *p = spares_.back();
    spares_.pop_back();
    return p;
  }
  void close() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      closed_ = true;
    }
    cv_.notify_all();
  }
  void abort() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      aborted_ = true;
    }
    cv_.notify_all();
  }
  void join() {
    if (th_.joinable()) th_.join();
  }
  bool write_failed() const { return write_failed_.load(); }
  double t_last_write() const { return t_last_write_; }

 private:
  void loop() {
    uint64_t range = 0;
    uint32_t piece = 0;
    for (;;) {
      OutItem it;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return aborted_ || closed_ || held_.count(std::make_pair(range, piece)); });
        if (aborted_) return;
        auto f = held_.find(std::make_pair(range, piece));
        if (f == held_.end()) return;  // closed, and that block is not coming
        it = std::move(f->second);
        held_.erase(f);
      }
      if (!it.log.empty()) write_all(fd_err_, it.log.data(), it.log.size());
      if (it.parts)
        for (const std::string &s : *it.parts)
          if (!s.empty() && !write_failed_.load() && write_all(fd_out_, s.data(), s.size())) write_failed_.store(true);
      t_last_write_ = now_s();
      if (it.in_order) it.in_order();
      if (it.parts) {
        std::lock_guard<std::mutex> lk(mu_);
        if (spares_.size() < 8)
          spares_.push_back(it.parts);
        else
          delete it.parts;
        it.parts = nullptr;
      }
      if (it.last_piece) {
        range++;
        piece = 0;
      } else {
        piece++;
      }
    }
  }
  int fd_out_, fd_err_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::map<std::pair<uint64_t, uint32_t>, OutItem> held_;
  std::vector<Parts *> spares_;
  bool closed_ = false, aborted_ = false;
  std::atomic<bool> write_failed_{false};
  double t_last_write_ = 0;
  std::thread th_;
};

// ---- helpers of the readers

// n bytes at file offset off into dst, by up to n_thr threads (one thread copies out of the page cache at about
// 10 GB/s).  Returns the bytes read (short only at the end of the file), or -1 with errno in *err.
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

// The threads of a device worker run on the cores of the NUMA node its GPU hangs off (the pinned buffers they fill
// are placed there by bvcf_alloc_pinned_near).  Only with several devices; BVCF_NUMA=0 turns it off.  Best effort.
struct NodeCpus {
  cpu_set_t set;
  bool valid = false;
};

NodeCpus cpus_near_device(int device) {
  NodeCpus r;
  CPU_ZERO(&r.set);
  char path[256], buf[4096];
  char bdf[64];
  if (bvcf_device_pci_bus_id(device, bdf, sizeof bdf) != 0) return r;
  for (char *q = bdf; *q; q++) *q = (char)tolower(*q);
  snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
  FILE *f = fopen(path, "r");
  if (!f) return r;
  int node = -1;
  const int got = fscanf(f, "%d", &node);
  fclose(f);
  if (got != 1 || node < 0) return r;
  snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
  f = fopen(path, "r");
  if (!f) return r;
  const size_t n = fread(buf, 1, sizeof buf - 1, f);
  fclose(f);
  buf[n] = 0;
  int count = 0;
  for (char *q = buf; *q;) {
    char *e = nullptr;
    const long lo = strtol(q, &e, 10);
    if (e == q) break;
    long hi = lo;
    if (*e == '-') {
      q = e + 1;
      hi = strtol(q, &e, 10);
    }
    for (long c = lo; c <= hi && c < CPU_SETSIZE; c++) {
      CPU_SET((int)c, &r.set);
      count++;
    }
    q = (*e == ',') ? e + 1 : e;
    if (*e != ',') break;
  }
  r.valid = count > 0;
  return r;
}

void bind_here(const NodeCpus &nc) {
  if (nc.valid) sched_setaffinity(0, sizeof nc.set, &nc.set);
}

// ---- BGZF framing for the readers

// Is there a terminator in the text of the block whose deflate payload is p[0, n)?  Inflates only as far as needed.
// 1 yes, 0 no, -1 the data is not valid DEFLATE.
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

// The first offset p >= from in buf[0, n) where BGZF blocks start: the block at p is well formed and so are the two
// after it (a chain that runs into the end of the buffer counts).  -1 if there is none.
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

struct Frame {  // one BGZF block inside a reader's window
  size_t off;   // of the block in the window
  uint32_t total, in_off, in_len, isize;
};

// the block at window offset off: 1 and *f filled, 0 the window ends before the block does, -1 malformed
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

}  // namespace

extern "C" {

int bvcf_run_fd(const bvcf_config *c, int fd_in, int fd_out, int fd_err, uint64_t *n_lines_in) {
  if (!c) return BVCF_E_ARG;
  const char *timing_env = getenv("BVCF_TIMING");
  const bool timing = timing_env != nullptr;
  const bool timing_json = timing && strcmp(timing_env, "json") == 0;
  const double t_start = now_s();
  double t_init = 0, t_prepare = 0, t_deal = 0, t_wait_read_stream = 0;
  // ---- how the input is read
  enum Mode { kStream, kRangeText, kRangeBgzf };
  Mode mode = kStream;
  off_t file_base = 0, file_size = 0;
  {
    struct stat st;
    const char *e = getenv("BVCF_RANGE_READ");
    const off_t at = lseek(fd_in, 0, SEEK_CUR);
    if (!(e && *e == '0') && at >= 0 && fstat(fd_in, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > at) {
      uint8_t magic[18];
      const ssize_t g = pread(fd_in, magic, sizeof magic, at);
      file_base = at;
      file_size = st.st_size;
      if (g >= 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
        uint32_t xlen = 0;
        const char *di = getenv("BVCF_DEVICE_INFLATE");
        if (g == 18 && bvcf_bgzf::block_size(magic, 18, &xlen) != -1 && !(di && *di == '0')) mode = kRangeBgzf;
      } else if (g > 0) {
        mode = kRangeText;
      }
    }
  }

  Run R;
  R.cfg = c;
  size_t text_in_flight = 2;
  if (const char *e = getenv("BVCF_TEXT_IN_FLIGHT")) text_in_flight = (size_t)std::min(6, std::max(1, atoi(e)));  // tuning
  R.n_slots = (uint32_t)text_in_flight + 1;  // two batches on the device, one more being formatted
  // (BGZF inflated on the device: a batch takes as long as its slowest block -- one wave decodes a block from start to
  // end -- and every batch costs the device thread ~0.8 ms of launches and waits, so the batches are made larger: 256 MiB
  // of text = ~4 000 blocks fill the decoder's wave slots; 400 k rows of configs[2] then take 0.034 s instead of 0.063)
  R.max_batch = c->max_batch_bytes ? c->max_batch_bytes : ((mode == kRangeBgzf ? 256ull : 64ull) << 20);
  const size_t cap = R.max_batch;
  std::atomic<uint64_t> lines_in{0};

  // the devices of the run
  std::vector<int> dev_list;
  if (c->n_devices && c->devices)
    dev_list.assign(c->devices, c->devices + c->n_devices);
  else
    dev_list.push_back(c->device);
  const size_t n_dev = dev_list.size();
  // host threads are bound to the CPUs of their device's NUMA node when the run spans devices (BVCF_NUMA=0: never,
  // BVCF_NUMA=1: also with one device -- the binding code can then be exercised on a one-GPU box)
  const char *numa_env = getenv("BVCF_NUMA");
  const bool numa_off = numa_env && *numa_env == '0', numa_force = numa_env && *numa_env == '1';
  const bool several_devices = (std::set<int>(dev_list.begin(), dev_list.end()).size() > 1 || numa_force) && !numa_off;

  // fmt.Fprintln(writer, stringHeader(config)), main.go:196-200
  if (!c->no_out) {
    char h[512];
    size_t hn = bvcf_string_header(c, h, sizeof h);
    h[hn] = '\n';
    if (write_all(fd_out, h, hn + 1)) {
      dprintf(fd_err, "write failed\n");
      return BVCF_E_FATAL;
    }
  }
  if (bvcf_device_count() <= 0) {
    // fail loudly: there is no CPU path
    dprintf(fd_err, "cannot allocate pinned host memory (no usable HIP device?)\n");
    return BVCF_E_NODEV;
  }

  // ---- first error wins; everything then drains
  OrderedSink sink(fd_out, fd_err);
  std::mutex fail_mu;
  int rc = BVCF_OK;
  std::string log;
  std::atomic<bool> failed{false};

  // ---- what the workers wait for: the header is known, the ctx parameters are set, the ranges are laid out
  struct Plan {
    std::mutex mu;
    std::condition_variable cv;
    bool ready = false;
    // range modes
    off_t data_off = 0;      // text: file offset of the first data line;  BGZF: of the block that holds it
    uint32_t first_off = 0;  // BGZF: where in that block's text the data lines start
    size_t range_bytes = 0;
    size_t spare_bytes = 0;  // text: what a reader reads past its range for the line that straddles the end
    uint64_t n_ranges = 0;
  } plan;
  auto plan_ready = [&]() {
    {
      std::lock_guard<std::mutex> lk(plan.mu);
      plan.ready = true;
    }
    plan.cv.notify_all();
  };
  auto wait_plan = [&]() {
    std::unique_lock<std::mutex> lk(plan.mu);
    plan.cv.wait(lk, [&] { return plan.ready || failed.load(); });
  };

  struct FmtJob {
    Block b;
    bvcf_result res;
    bool has_res = false;
    uint64_t job = 0;  // its number among the worker's collected batches
    bool end = false;
  };
  struct DevWorker {
    uint32_t idx = 0;
    int device = 0;
    bvcf_ctx *ctx = nullptr;
    Channel<Block> q{3};     // blocks for this device
    Channel<FmtJob> fq{8};   // collected batches for its formatter
    std::thread dev_th, fmt_th;
    std::vector<std::thread> rd_th;
    std::unique_ptr<BufPool> pool;       // range modes: the worker's own pinned buffers
    std::unique_ptr<WorkPool> fmt_pool;  // TSV assembly threads
    NodeCpus cpus;
    // result slots: jobs of this worker that are done with (formatted; with a dosage file: appended in order)
    std::mutex mu;
    std::condition_variable cv;
    uint64_t fmt_done = 0;            // every job below this number is done with
    std::set<uint64_t> done_early;  // ... and these above it
    // timing
    double t_warm = 0, t_ctx = 0, t_submit = 0, t_gpu = 0, t_fmt_wait = 0, t_first_submit = 0, t_starved = 0, t_fmt = 0,
           t_read = 0, t_first_block = 0;
    uint64_t n_blocks = 0, n_bytes = 0;
  };
  std::vector<std::unique_ptr<DevWorker>> workers;
  for (size_t d = 0; d < n_dev; d++) {
    workers.emplace_back(new DevWorker());
    workers.back()->idx = (uint32_t)d;
    workers.back()->device = dev_list[d];
  }
  auto fail = [&](const std::string &m, int code) {
    {
      std::lock_guard<std::mutex> lk(fail_mu);
      if (rc == BVCF_OK) {
        rc = code;
        log.append(m + "\n");
      }
      failed.store(true);
    }
    sink.abort();
    plan.cv.notify_all();
    for (auto &W : workers) W->cv.notify_all();
  };
  std::atomic<size_t> max_in_flight{text_in_flight};  // batches a device worker keeps submitted
  // BGZF batches on the device: a block's DEFLATE stream is decoded by one wave from start to end, so a batch takes as
  // long as its slowest block however few blocks it has, and a batch of ~1 000 blocks fills a fifth of the wave slots the
  // decoder's LDS footprint allows: several batches inflate side by side while another is in its kernel chain
  size_t bgzf_in_flight = 3;
  if (const char *e = getenv("BVCF_BGZF_IN_FLIGHT")) bgzf_in_flight = (size_t)std::min(8, std::max(1, atoi(e)));  // tuning
  std::atomic<bool> input_is_bgzf_device{mode == kRangeBgzf};
  const unsigned hw = usable_cpus();
  // (measured on a 16-core share of the host, one device, 24 GB from /dev/shm: 2 readers 0.84 s steady -- the device waits
  // for them --, 4 readers 0.58 s, 8 readers 0.53 s = 46 GB/s)
  unsigned n_read_thr = (unsigned)std::min<size_t>(8, std::max<size_t>(2, hw / (2 * n_dev)));
  if (const char *e = getenv("BVCF_READ_THREADS")) n_read_thr = (unsigned)std::max(1, atoi(e));  // tuning

  // ---- device thread of a worker: the goroutine of main.go:345-347 with a GPU behind it
  auto device_main = [&](DevWorker *W) {
    if (several_devices) bind_here(W->cpus);
    {
      const double t0 = now_s();
      bvcf_warmup(W->device);  // runtime + kernels onto the device while the header is read (errors: bvcf_create reports them)
      W->t_warm = now_s() - t0;
    }
    wait_plan();
    std::string wmsg;
    // The ctx is created while the worker's reader fills its first buffer -- when it is known that a block will come:
    // in range mode worker k has work iff there are more than k ranges; of a stream only the first worker is sure of one.
    if (!failed.load() && (mode == kStream ? W->idx == 0 : W->idx < plan.n_ranges)) {
      const double tc = now_s();
      const int r = create_ctx(R, W->device, &W->ctx, &wmsg);
      W->t_ctx = now_s() - tc;
      if (r) fail(wmsg, r);
    }
    std::deque<Block> in_flight;  // submitted, not yet collected (oldest first)
    // Collect number q of the ctx lands in result slot q % n_slots, whose arrays the formatter may still be reading for
    // the batch collected n_slots collects ago.
    uint64_t n_collects = 0, n_jobs = 0;
    std::deque<std::pair<uint64_t, uint64_t>> outstanding;  // (job number, collect number) of jobs not known finished
    auto wait_formatted = [&](uint64_t n) {
      const double t0 = now_s();
      std::unique_lock<std::mutex> lk(W->mu);
      W->cv.wait(lk, [&] { return W->fmt_done >= n || failed.load(); });
      W->t_fmt_wait += now_s() - t0;
    };
    auto slot_is_free = [&]() {
      uint64_t need = 0;
      while (!outstanding.empty() && outstanding.front().second + R.n_slots <= n_collects) {
        need = outstanding.front().first + 1;
        outstanding.pop_front();
      }
      if (need) wait_formatted(need);
    };
    auto submit = [&](const Block &q) {
      return q.bgzf ? bvcf_submit_bgzf(W->ctx, q.data, q.nb, q.own, q.bgzf_flags, q.first_off, q.range)
                    : bvcf_submit(W->ctx, q.data, q.nb, q.range);
    };
    auto finish_oldest = [&]() {
      Block b = in_flight.front();
      bvcf_result res;
      slot_is_free();
      if (failed.load()) return;
      const double t0 = now_s();
      int r = bvcf_collect(W->ctx, &res);
      n_collects++;
      if (r == BVCF_E_CAPACITY) {
        // drop what is in flight here, let the formatter finish with the arrays that are about to be reallocated,
        // grow, resubmit everything still queued on this device
        wait_formatted(n_jobs);
        outstanding.clear();
        for (size_t k = 1; k < in_flight.size(); k++) {
          bvcf_result tmp;
          bvcf_collect(W->ctx, &tmp);
          n_collects++;
        }
        r = bvcf_reserve(W->ctx, res.need_lines + res.need_lines / 4 + 64, res.need_alleles + res.need_alleles / 4 + 64,
                         res.need_cmap_bytes + res.need_cmap_bytes / 4 + 4096);
        for (size_t k = 0; k < in_flight.size() && r == BVCF_OK; k++) r = submit(in_flight[k]);
        if (r == BVCF_OK) {
          r = bvcf_collect(W->ctx, &res);
          n_collects++;
        }
      }
      W->t_gpu += now_s() - t0;
      if (r != BVCF_OK) {
        fail(std::string("bvcf: ") + bvcf_last_error(W->ctx), r);
        return;
      }
      in_flight.pop_front();
      lines_in.fetch_add(res.n_lines_seen);
      FmtJob j;
      j.b = b;
      j.res = res;
      j.has_res = true;
      j.job = n_jobs;
      outstanding.emplace_back(n_jobs, n_collects - 1);
      n_jobs++;
      W->fq.push(std::move(j));
    };
    for (;;) {
      const double tp = now_s();
      Block b = W->q.pop();
      if (W->n_blocks) W->t_starved += now_s() - tp;
      if (b.end) break;
      if (failed.load()) continue;
      if (!b.nb) {
        // nothing in this block (a range inside one long line): it still takes its turn in the output
        FmtJob j;
        j.b = b;
        W->fq.push(std::move(j));
        continue;
      }
      if (!W->ctx) {
        const double tc = now_s();
        const int r = create_ctx(R, W->device, &W->ctx, &wmsg);
        W->t_ctx = now_s() - tc;
        if (r) {
          fail(wmsg, r);
          continue;
        }
      }
      // keep one block (BGZF on the device: two) ahead of the one being collected
      if (in_flight.size() >= max_in_flight.load()) finish_oldest();
      if (failed.load()) continue;
      const double ts = now_s();
      const int r = submit(b);
      if (!W->n_blocks) W->t_first_submit = now_s() - t_start;
      W->t_submit += now_s() - ts;
      if (r) {
        fail(std::string("bvcf_submit: ") + bvcf_last_error(W->ctx), r);
        continue;
      }
      W->n_blocks++;
      W->n_bytes += b.bgzf ? b.own : b.nb;
      in_flight.push_back(b);
    }
    while (!failed.load() && !in_flight.empty()) finish_oldest();
    FmtJob e;
    e.end = true;
    W->fq.push(std::move(e));
  };

  // ---- formatter thread of a worker: main.go:566-695 for its collected batches
  std::atomic<bool> dosage_failed{false};
  auto formatter_main = [&](DevWorker *W) {
    if (several_devices) bind_here(W->cpus);
    for (;;) {
      FmtJob j = W->fq.pop();
      if (j.end) break;
      const uint64_t job = j.job;
      auto release = [W, job]() {
        {
          std::lock_guard<std::mutex> lk(W->mu);
          W->done_early.insert(job);
          while (W->done_early.erase(W->fmt_done)) W->fmt_done++;
        }
        W->cv.notify_all();
      };
      if (failed.load()) {
        if (j.has_res) release();
        continue;
      }
      const double t0 = now_s();
      OutItem it;
      it.range = j.b.range;
      it.piece = j.b.piece;
      it.last_piece = j.b.last_piece;
      it.parts = sink.spare();
      if (j.has_res) {
        const uint8_t *text = j.b.bgzf ? j.res.text : j.b.data;  // (inflated on the device: the copy that came back)
        format_log(&j.res, text, it.log);
        if (R.want_rows)
          format_parts(c, &j.res, text, *R.names, R.ratios.get(), W->fmt_pool.get(), *it.parts);
        else
          for (auto &q : *it.parts) q.clear();
        if (R.arrow && j.res.dosage) {
          // the dosage rows go into the file in input order (main.go:576-584): the sink runs this when it is the block's
          // turn; the result slot and the text stay on loan until then
          auto keep = std::make_shared<FmtJob>(std::move(j));
          it.in_order = [&R, &dosage_failed, &fail, keep, text, release]() {
            if (!dosage_failed.load() && append_dosage(R, &keep->res, text)) {
              dosage_failed.store(true);
              fail("dosage matrix: write failed", BVCF_E_FATAL);
            }
            release();
          };
        } else {
          release();
        }
      } else {
        for (auto &q : *it.parts) q.clear();
      }
      W->t_fmt += now_s() - t0;
      sink.put(std::move(it));
      if (sink.write_failed()) fail("write failed", BVCF_E_FATAL);
    }
  };

  // ---- blocks of a text range / of the text stream: `text[s, e)` of the buffer held by `hold`
  auto text_block = [](std::shared_ptr<BufHold> hold, const uint8_t *p, size_t n, uint64_t range, uint32_t piece, bool last) {
    Block b;
    b.hold = std::move(hold);
    b.data = p;
    b.nb = n;
    b.range = range;
    b.piece = piece;
    b.last_piece = last;
    return b;
  };
  std::atomic<uint8_t> eol_byte{'\n'};

  // ---- range mode, text: worker k reads ranges k, k + N, ... of the file into its own pinned buffers
  const unsigned n_text_readers = 2;  // per worker: one reads range i + N while the other cuts and hands over range i
  auto range_text_reader = [&](DevWorker *W, unsigned r) {
    if (several_devices) bind_here(W->cpus);
    wait_plan();
    if (failed.load()) return;
    const uint8_t eol = eol_byte.load();
    const size_t Rb = plan.range_bytes;
    double t_read = 0;
    // (with a dosage file a block keeps its buffer until it is its turn in the output: a second reader running ahead
    // could then hold every buffer while the first one waits for one -- a single reader takes the ranges in order)
    const unsigned stride = R.arrow ? 1u : n_text_readers;
    if (r >= stride) return;
    for (uint64_t j = r; !failed.load(); j += stride) {
      const uint64_t i = W->idx + (uint64_t)n_dev * j;
      if (i >= plan.n_ranges) break;
      const off_t a = plan.data_off + (off_t)(i * Rb);
      const off_t b = std::min<off_t>(a + (off_t)Rb, file_size);
      const bool last_range = b >= file_size;
      auto hold = W->pool->get();
      if (!hold) {
        if (!failed.load()) fail("cannot allocate pinned host memory", BVCF_E_NOMEM);
        break;
      }
      const double t0 = now_s();
      const size_t want = (size_t)std::min<off_t>((off_t)(Rb + plan.spare_bytes), file_size - a);
      int err = 0;
      const ssize_t got = pread_parallel(fd_in, hold->p, want, a, n_read_thr, &err);
      t_read += now_s() - t0;
      if (got < (ssize_t)want) {
        fail(got < 0 ? std::string("read: ") + strerror(err) : std::string("read: the input file got shorter"), BVCF_E_FATAL);
        break;
      }
      const uint8_t *buf = hold->p;
      const size_t n = (size_t)got, own_len = (size_t)(b - a);
      // where this range's lines start: after the first terminator at or past `a` (range 0: at the first data line)
      size_t s = 0;
      if (i > 0) {
        const uint8_t *t = (const uint8_t *)memchr(buf, eol, own_len);
        if (!t) {
          // one line covers the whole range: it belongs to an earlier range
          W->q.push(text_block(nullptr, nullptr, 0, i, 0, true));
          continue;
        }
        s = (size_t)(t - buf) + 1;
      }
      if (last_range) {
        // the run's last line ends the range; an unterminated tail is dropped (main.go:354-358)
        const uint8_t *t = n > s ? (const uint8_t *)memrchr(buf + s, eol, n - s) : nullptr;
        const size_t e = t ? (size_t)(t - buf) + 1 : s;
        W->q.push(text_block(e > s ? hold : nullptr, buf + s, e - s, i, 0, true));
        continue;
      }
      // ... and where they end: after the first terminator at or past `b` (the line that straddles the end is ours)
      const uint8_t *t = (const uint8_t *)memchr(buf + own_len, eol, n - own_len);
      if (t) {
        const size_t e = (size_t)(t - buf) + 1;
        W->q.push(text_block(hold, buf + s, e - s, i, 0, true));
        continue;
      }
      // The straddling line does not end within the buffer's spare room: the lines before it go as they are, the long
      // line is read into memory of its own (up to max_batch_bytes, as for the single reader).
      const uint8_t *tl = own_len > s ? (const uint8_t *)memrchr(buf + s, eol, own_len - s) : nullptr;
      const size_t s_long = tl ? (size_t)(tl - buf) + 1 : s;
      auto big = std::make_shared<BufHold>();
      big->heap.assign(buf + s_long, buf + n);
      bool found = false, too_long = false, at_eof = false;
      off_t pos = a + (off_t)n;
      while (!found && !too_long && !at_eof) {
        const size_t old = big->heap.size(), step = 4u << 20;
        big->heap.resize(old + step);
        int e2 = 0;
        const ssize_t g = pread_parallel(fd_in, big->heap.data() + old, (size_t)std::min<off_t>((off_t)step, file_size - pos), pos, 1, &e2);
        if (g < 0) {
          fail(std::string("read: ") + strerror(e2), BVCF_E_FATAL);
          at_eof = true;
          break;
        }
        big->heap.resize(old + (size_t)g);
        pos += g;
        if (g == 0 || pos >= file_size) at_eof = true;
        const uint8_t *t2 = (const uint8_t *)memchr(big->heap.data() + old, eol, (size_t)g);
        if (t2) {
          big->heap.resize((size_t)(t2 - big->heap.data()) + 1);
          found = true;
        } else if (big->heap.size() > cap) {
          too_long = true;
        }
      }
      if (failed.load()) break;
      if (too_long || (found && big->heap.size() > cap)) {
        fail("a line is longer than max_batch_bytes", BVCF_E_TOO_BIG);
        break;
      }
      W->q.push(text_block(s_long > s ? hold : nullptr, buf + s, s_long - s, i, 0, false));
      if (found) {
        const uint8_t *p = big->heap.data();
        const size_t nb = big->heap.size();
        W->q.push(text_block(big, p, nb, i, 1, true));
      } else {
        W->q.push(text_block(nullptr, nullptr, 0, i, 1, true));  // the file ends inside the line: dropped
      }
    }
    W->pool->stop();
    std::lock_guard<std::mutex> lk(W->mu);
    W->t_read += t_read;
  };

  // ---- range mode, BGZF: worker k reads compressed ranges k, k + N, ... and cuts them into batches of whole blocks
  // for bvcf_submit_bgzf.  The look-ahead of a batch is found, not guessed: the blocks after its own are inflated here,
  // just far enough to see a terminator (a fraction of a block per batch).
  auto range_bgzf_reader = [&](DevWorker *W) {
    if (several_devices) bind_here(W->cpus);
    wait_plan();
    if (failed.load()) return;
    const uint8_t eol = eol_byte.load();
    const size_t Rb = plan.range_bytes;
    const size_t buf_bytes = W->pool->bytes();
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) {
      fail("inflateInit2 failed", BVCF_E_FATAL);
      return;
    }
    size_t la_reserve = 4u << 16;  // text kept free for the look-ahead when a batch's own blocks are chosen
    for (uint64_t i = W->idx; i < plan.n_ranges && !failed.load(); i += n_dev) {
      const off_t a = plan.data_off + (off_t)(i * Rb);
      const off_t b = std::min<off_t>(a + (off_t)Rb, file_size);
      auto hold = W->pool->get();
      if (!hold) {
        if (!failed.load()) fail("cannot allocate pinned host memory", BVCF_E_NOMEM);
        break;
      }
      const double t0 = now_s();
      size_t want = (size_t)std::min<off_t>((off_t)buf_bytes, file_size - a);
      int err = 0;
      const ssize_t got = pread_parallel(fd_in, hold->p, want, a, n_read_thr, &err);
      W->t_read += now_s() - t0;
      if (got < (ssize_t)want) {
        fail(got < 0 ? std::string("read: ") + strerror(err) : std::string("read: the input file got shorter"), BVCF_E_FATAL);
        break;
      }
      // the window: the pinned buffer; if a batch's look-ahead runs past it, a copy in memory of its own that grows
      const uint8_t *win = hold->p;
      size_t win_n = (size_t)got;
      std::shared_ptr<BufHold> big;  // set once the window has moved
      auto window_reaches_eof = [&]() { return a + (off_t)win_n >= file_size; };
      auto grow_window = [&]() -> bool {
        if (window_reaches_eof()) return false;
        auto nb = std::make_shared<BufHold>();
        const size_t step = 8u << 20;
        nb->heap.resize(win_n + step);
        memcpy(nb->heap.data(), win, win_n);
        int e2 = 0;
        const ssize_t g = pread_parallel(fd_in, nb->heap.data() + win_n, (size_t)std::min<off_t>((off_t)step, file_size - a - (off_t)win_n),
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
          W->q.push(text_block(nullptr, nullptr, 0, i, 0, true));  // no block starts in this range
          continue;
        }
        p0 = (size_t)f;
      }
      // frame of the block at window offset off, growing the window when it ends inside the block
      bool bad = false;
      auto frame = [&](size_t off, Frame *f) -> int {
        for (;;) {
          const int r = frame_at(win, win_n, off, f);
          if (r != 0) {
            if (r < 0) bad = true;
            return r;
          }
          if (off >= win_n && window_reaches_eof()) return 0;  // a clean end of the input
          if (!grow_window()) {
            bad = true;  // the file ends inside a block
            return -1;
          }
        }
      };
      size_t pos = p0;
      uint32_t piece = 0;
      bool first_batch = true;
      while (!bad && !failed.load()) {
        // own blocks: those that start before `b`, while their text leaves room for the look-ahead
        std::vector<Frame> own;
        size_t own_text = 0, q = pos;
        for (;;) {
          Frame f;
          if (q >= own_len || frame(q, &f) != 1) break;
          if (!own.empty() && own_text + f.isize + la_reserve > cap) break;
          own.push_back(f);
          own_text += f.isize;
          q += f.total;
        }
        if (bad) break;
        if (own.empty()) {
          // (only when the range's last block ended exactly at `b` on the previous batch: close the range)
          W->q.push(text_block(nullptr, nullptr, 0, i, piece, true));
          break;
        }
        // the look-ahead: blocks after the own ones until the text shows a terminator
        size_t la = 0, la_text = 0;
        bool at_eof = false;
        for (;;) {
          size_t own_bytes = 0;
          for (const Frame &f : own) own_bytes += f.total;
          la = 0;
          la_text = 0;
          at_eof = false;
          for (;;) {
            Frame f;
            const int r = frame(pos + own_bytes + la, &f);
            if (r < 0) break;
            if (r == 0) {
              at_eof = true;
              break;
            }
            la += f.total;
            la_text += f.isize;
            const int he = f.isize ? block_has_eol(zs, win + f.off + f.in_off, f.in_len, eol) : 0;
            if (he < 0) bad = true;
            if (he != 0) break;
          }
          if (bad || own_text + la_text <= cap) break;
          if (own.size() == 1) {
            fail("a line is longer than max_batch_bytes", BVCF_E_TOO_BIG);
            bad = true;
            break;
          }
          // the line that straddles the end of the own blocks is longer than the room kept for it: fewer own blocks
          while (own.size() > 1 && own_text + la_text > cap) {
            own_text -= own.back().isize;
            own.pop_back();
          }
        }
        if (bad) break;
        la_reserve = std::max(la_reserve, std::min<size_t>(2 * la_text, cap / 2));
        size_t own_bytes = 0;
        for (const Frame &f : own) own_bytes += f.total;
        const size_t next = pos + own_bytes;
        // (the end-of-file marker block and anything else without text after the last terminator: nothing follows)
        const bool last = next >= own_len || (window_reaches_eof() && next >= win_n);
        Block blk;
        blk.hold = big ? big : hold;
        blk.data = win + pos;
        blk.nb = own_bytes + la;
        blk.own = own_bytes;
        blk.bgzf = true;
        blk.bgzf_flags = ((i == 0 && first_batch) ? 0 : BVCF_BGZF_SKIP_FIRST_LINE) | (at_eof ? BVCF_BGZF_END_OF_STREAM : 0);
        blk.first_off = (i == 0 && first_batch) ? plan.first_off : 0;
        blk.range = i;
        blk.piece = piece++;
        blk.last_piece = last;
        W->q.push(blk);
        first_batch = false;
        pos = next;
        if (last) break;
      }
      if (bad && !failed.load()) fail("bgzf: not a BGZF block, or a truncated file", BVCF_E_FATAL);
      if (i + n_dev >= plan.n_ranges) W->pool->stop();
    }
    inflateEnd(&zs);
    W->pool->stop();
  };

  // ---- stream mode: ONE reader cuts the blocks (pipes, single-stream gzip), this thread deals them
  std::string source_err;
  Channel<Block> ready_q(4 * n_dev + 4);
  std::unique_ptr<BufPool> stream_pool;
  std::atomic<bool> stop{false};
  struct StreamPre {  // what the reader learnt before the first block: preamble and a sample of the data lines
    bool have = false;
    std::vector<uint8_t> sample;
  } stream_pre;
  auto push_end = [&](bool read_error, bool too_long) {
    Block e;
    e.end = true;
    e.bgzf_flags = (read_error ? 1 : 0) | (too_long ? 2 : 0);  // (on an end marker: why the stream ended early)
    ready_q.push(e);
  };
  // the preamble of the stream (main.go:250-304), the ctx parameters; then the workers may start
  auto adopt_preamble = [&](const uint8_t *data, size_t n_data) -> bool {
    const double t0 = now_s();
    std::string msg;
    const int r = prepare_run(R, &msg, data, n_data, false);
    if (r) {
      fail(msg, r);
      return false;
    }
    if (R.pre.header.size() == 9) {
      const char *m = "Found 9 header fields. When genotypes present, we expect 1+ samples after FORMAT (10 fields minimum)\n";
      write_all(fd_err, m, strlen(m));
    }
    eol_byte.store(R.pre.eol_byte);
    const unsigned per_worker = std::max(1u, (unsigned)(R.n_threads / n_dev));
    for (auto &W : workers)
      if (R.want_rows && per_worker > 1) W->fmt_pool.reset(new WorkPool(per_worker));
    t_prepare = now_s() - t0;
    return true;
  };

  // BGZF on a pipe, inflated on the device: whole compressed blocks per buffer.  The header has to be read here, so the
  // leading blocks are inflated with zlib until the #CHROM line is complete; everything from the block that holds the
  // first data line on is handed over compressed, each batch with the following blocks as look-ahead.
  auto stream_bgzf = [&](bvcf_input::ByteSource &src) {
    input_is_bgzf_device.store(true);
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
        if (got < 0) source_err = src.error();
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
        if (!more()) return (!source_err.empty() || had) ? -1 : 0;
      }
    };
    auto fail_read = [&](const std::string &m) {
      source_err = m;
      push_end(true, false);
    };
    // ---- the header, from blocks inflated here
    std::vector<uint8_t> htext;
    std::vector<std::pair<size_t, size_t>> marks;  // (compressed offset from pp, text offset) of each inflated block
    size_t hoff = 0;
    std::string msg;
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return fail_read("inflateInit2 failed");
    auto inflate_next = [&]() -> int {  // 1 = a block was inflated, 0 = end of input, -1 = bad
      Frame f;
      const int r = frame(hoff, &f);
      if (r <= 0) return r;
      marks.emplace_back(hoff, htext.size());
      const size_t at = htext.size();
      htext.resize(at + f.isize);
      inflateReset(&zs);
      const uint8_t *blk = pend.data() + pp + hoff;
      zs.next_in = const_cast<uint8_t *>(blk + f.in_off);
      zs.avail_in = f.in_len;
      zs.next_out = htext.data() + at;
      zs.avail_out = f.isize;
      const int zr = f.isize ? inflate(&zs, Z_FINISH) : Z_STREAM_END;
      const uint8_t *tail = blk + f.total - 8;
      const uint32_t want_crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
      if ((f.isize && (zr != Z_STREAM_END || zs.avail_out != 0)) ||
          (uint32_t)crc32(crc32(0L, Z_NULL, 0), htext.data() + at, f.isize) != want_crc)
        return -1;
      hoff += f.total;
      return 1;
    };
    int pr = 1;
    bool hdr_eof = false;
    while (pr == 1) {
      const int ir = inflate_next();
      if (ir < 0) {
        inflateEnd(&zs);
        return fail_read(source_err.empty() ? std::string("bgzf: corrupt block (inflate or CRC mismatch)") : source_err);
      }
      hdr_eof = ir == 0;
      pr = parse_preamble(htext.data(), htext.size(), hdr_eof, c->normalize_header, &R.pre, &msg);
      if (hdr_eof) break;
    }
    if (pr != 0) {
      inflateEnd(&zs);
      fail(pr < 0 ? msg : std::string("No header found"), BVCF_E_FATAL);
      return push_end(false, false);
    }
    const size_t data_off = R.pre.data_off;
    // a few data lines for prepare_run (path choice, reservation): make sure at least one whole line is in view
    for (int extra = 0; extra < 8; extra++) {
      if (memchr(htext.data() + data_off, R.pre.eol_byte, htext.size() - data_off)) break;
      if (inflate_next() != 1) break;
    }
    // one more batch in flight per device than for text: two batches' blocks inflate side by side (k_inflate_w16)
    // while a third is in its kernel chain / on its way back
    R.n_slots = (uint32_t)bgzf_in_flight + 1;
    max_in_flight.store(bgzf_in_flight);
    if (!adopt_preamble(htext.data() + data_off, htext.size() - data_off)) {
      inflateEnd(&zs);
      return push_end(false, false);
    }
    const uint8_t eol = R.pre.eol_byte;
    // the block that holds the first data byte (or the end of what was inflated)
    size_t b0 = marks.size();
    for (size_t i = 0; i < marks.size(); i++) {
      const size_t t_end = i + 1 < marks.size() ? marks[i + 1].second : htext.size();
      if (data_off < t_end) {
        b0 = i;
        break;
      }
    }
    uint32_t first_off = 0;
    if (b0 < marks.size()) {
      first_off = (uint32_t)(data_off - marks[b0].second);
      pp += marks[b0].first;
    } else {
      pp += hoff;
    }
    // the compressed bytes of a batch (own + look-ahead blocks) go into pinned buffers a quarter of the text's size
    const size_t small = std::max<size_t>(cap / 4, 1u << 20);
    stream_pool.reset(new BufPool(dev_list[0], small, (int)std::min<size_t>(4 * n_dev + 4, 64)));
    stream_pool->start();
    plan_ready();
    size_t la_reserve = 4u << 16;
    bool first = true;
    uint64_t seq = 0;
    for (;;) {
      if (stop.load() || failed.load()) break;
      auto hold = stream_pool->get();
      if (!hold) break;
      std::vector<Frame> own;
      size_t own_text = 0, own_bytes = 0;
      bool bad = false;
      for (;;) {
        Frame f;
        const int r = frame(own_bytes, &f);
        if (r < 0) bad = true;
        if (r <= 0) break;
        if (!own.empty() && (own_text + f.isize + la_reserve > cap || own_bytes + f.total + (la_reserve >> 1) + (1u << 17) > small)) break;
        own.push_back(f);
        own_text += f.isize;
        own_bytes += f.total;
      }
      size_t la = 0, la_text = 0;
      bool at_eof = false, too_long = false;
      while (!bad && !own.empty()) {
        la = 0;
        la_text = 0;
        at_eof = false;
        for (;;) {
          Frame f;
          const int r = frame(own_bytes + la, &f);
          if (r < 0) bad = true;
          if (r == 0) at_eof = true;
          if (r <= 0) break;
          la += f.total;
          la_text += f.isize;
          const int he = f.isize ? block_has_eol(zs, pend.data() + pp + f.off + f.in_off, f.in_len, eol) : 0;
          if (he < 0) bad = true;
          if (he != 0) break;
        }
        if (bad || (own_text + la_text <= cap && own_bytes + la <= small)) break;
        if (own.size() == 1) {
          too_long = true;
          break;
        }
        while (own.size() > 1 && (own_text + la_text > cap || own_bytes + la > small)) {
          own_text -= own.back().isize;
          own_bytes -= own.back().total;
          own.pop_back();
        }
      }
      if (bad) {
        inflateEnd(&zs);
        return fail_read(source_err.empty() ? std::string("bgzf: not a BGZF block, or a truncated file") : source_err);
      }
      if (too_long) {
        inflateEnd(&zs);
        return push_end(false, true);
      }
      if (own.empty()) break;  // a clean end of the input
      la_reserve = std::max(la_reserve, std::min<size_t>(2 * la_text, cap / 2));
      memcpy(hold->p, pend.data() + pp, own_bytes + la);
      Block b;
      b.data = hold->p;
      b.hold = std::move(hold);
      b.nb = own_bytes + la;
      b.own = own_bytes;
      b.bgzf = true;
      b.bgzf_flags = (first ? 0 : BVCF_BGZF_SKIP_FIRST_LINE) | (at_eof ? BVCF_BGZF_END_OF_STREAM : 0);
      b.first_off = first ? first_off : 0;
      b.range = seq++;
      pp += own_bytes;
      first = false;
      ready_q.push(b);
      if (at_eof && la == 0) break;
    }
    inflateEnd(&zs);
    stream_pool->stop();
    push_end(false, false);
  };

  auto stream_reader = [&]() {
    bvcf_input::ByteSource src(fd_in, std::min(32u, hw));
    {
      // BGZF input (bgzip / htslib .vcf.gz): the blocks go to the device compressed and are inflated there
      // (bvcf_submit_bgzf) unless BVCF_DEVICE_INFLATE=0 (then this thread's workers inflate them with zlib)
      const char *e = getenv("BVCF_DEVICE_INFLATE");
      if (!(e && *e == '0') && src.sniff_bgzf()) {
        stream_bgzf(src);
        return;
      }
    }
    // being read into, two on each device, up to two with each formatter, one spare
    stream_pool.reset(new BufPool(dev_list[0], cap, (int)std::min<size_t>(4 * n_dev + 3, 64)));
    stream_pool->start(1);
    std::vector<uint8_t> carry;
    bool first = true, eof = false;
    uint64_t seq = 0;
    uint8_t eol = '\n';
    while (!eof && !stop.load() && !failed.load()) {
      auto hold = stream_pool->get();
      if (!hold) {
        if (stream_pool->failed()) fail("cannot allocate pinned host memory", BVCF_E_NOMEM);
        break;
      }
      uint8_t *buf = hold->p;
      size_t fill = carry.size();
      if (fill) memcpy(buf, carry.data(), fill);
      carry.clear();
      bool read_error = false;
      while (!eof && fill < cap) {
        ssize_t got = src.read(buf + fill, cap - fill);
        if (got == bvcf_input::ByteSource::kNoRoom) break;  // this buffer is as full as it gets
        if (got < 0) {
          source_err = src.error();
          read_error = true;
          eof = true;
          break;
        }
        if (got == 0) {
          eof = true;
          break;
        }
        fill += (size_t)got;
      }
      if (eof) stream_pool->stop();
      if (read_error) return push_end(true, false);
      size_t start = 0;
      if (first) {
        // readVcf's preamble (main.go:250-304); the terminator is learnt from line 1 (parse.FindEndOfLine)
        std::string msg;
        const int pr = parse_preamble(buf, fill, true, c->normalize_header, &R.pre, &msg);
        if (pr != 0) {
          fail(msg, BVCF_E_FATAL);
          return push_end(false, false);
        }
        start = R.pre.data_off;
        if (!adopt_preamble(buf + start, fill > start ? fill - start : 0)) return push_end(false, false);
        eol = R.pre.eol_byte;
        plan_ready();
        first = false;
      }
      const uint8_t *lastp = fill > start ? (const uint8_t *)memrchr(buf + start, eol, fill - start) : nullptr;
      if (!lastp) {
        if (!eof && fill == cap) return push_end(false, true);
        // at EOF an unterminated tail is dropped (main.go:354-358); otherwise the line continues in the next buffer
        if (!eof) carry.assign(buf + start, buf + fill);
        continue;
      }
      const size_t nb = (size_t)(lastp - buf) + 1 - start;
      if (!eof) carry.assign(buf + start + nb, buf + fill);
      ready_q.push(text_block(std::move(hold), buf + start, nb, seq++, 0, true));
    }
    stream_pool->stop();
    push_end(false, false);
  };

  // ---- start: the sink, the workers (their devices warm up while the header is read), the reader(s)
  sink.start();
  for (auto &W : workers) {
    if (several_devices) W->cpus = cpus_near_device(W->device);
    DevWorker *w = W.get();
    W->dev_th = std::thread(device_main, w);
    W->fmt_th = std::thread(formatter_main, w);
  }
  std::thread stream_th;
  bool have_pre = false;
  if (mode == kStream) {
    stream_th = std::thread(stream_reader);
  } else {
    // the worker's buffers are pinned while the header is read; a text buffer holds a range and the spare room for the
    // line that straddles its end, a BGZF buffer the compressed range and some look-ahead blocks
    const size_t total = (size_t)(file_size - file_base);
    size_t buf_bytes = cap;
    if (mode == kRangeBgzf) {
      size_t rb = std::min<size_t>(std::max<size_t>(total / (4 * n_dev), 1u << 20), std::max<size_t>(cap / 4, 1u << 20));
      rb = (rb + 0xFFFFu) & ~(size_t)0xFFFFu;
      plan.range_bytes = rb;
      buf_bytes = rb + (1u << 20);
    } else {
      // (up to an eighth of the buffer for the line that straddles a range's end -- how much of it is read is decided
      // below, from the first data line; a line that needs more takes the slow way round)
      plan.range_bytes = cap - std::max<size_t>(cap / 8, std::min<size_t>(cap / 2, 64u << 10));
      plan.spare_bytes = cap - plan.range_bytes;
    }
    for (auto &W : workers) {
      // being read into, one queued, two (BGZF: three) on the device, up to two with the formatter
      W->pool.reset(new BufPool(W->device, buf_bytes, mode == kRangeBgzf ? (int)bgzf_in_flight + 2 : 5 + (int)text_in_flight));
      W->pool->start(mode == kRangeBgzf ? 1 : 2);
      DevWorker *w = W.get();
      if (mode == kRangeBgzf)
        W->rd_th.emplace_back(range_bgzf_reader, w);
      else
        for (unsigned r = 0; r < n_text_readers; r++) W->rd_th.emplace_back(range_text_reader, w, r);
    }
    // ---- this thread: the header
    std::string msg;
    std::vector<uint8_t> head;
    bool ok = false;
    if (mode == kRangeText) {
      size_t look = 1u << 20;
      for (;;) {
        look = std::min<size_t>(look, total);
        head.resize(look);
        int err = 0;
        const ssize_t g = pread_parallel(fd_in, head.data(), look, file_base, 1, &err);
        if (g < 0) {
          fail(std::string("read: ") + strerror(err), BVCF_E_FATAL);
          break;
        }
        head.resize((size_t)g);
        R.pre = Preamble();
        // (as the single reader does it: a header that does not end within max_batch_bytes is "No header found")
        const bool all = head.size() >= total || head.size() >= cap;
        const int pr = parse_preamble(head.data(), head.size(), all, c->normalize_header, &R.pre, &msg);
        if (pr < 0) {
          fail(msg, BVCF_E_FATAL);
          break;
        }
        if (pr == 0) {
          // some data lines for prepare_run (the shape of the lines, the reservation)
          const size_t want = std::min<size_t>(total, R.pre.data_off + (4u << 20));
          if (head.size() < want) {
            const size_t old = head.size();
            head.resize(want);
            const ssize_t g2 = pread_parallel(fd_in, head.data() + old, want - old, file_base + (off_t)old, 1, &err);
            head.resize(old + (g2 > 0 ? (size_t)g2 : 0));
          }
          ok = true;
          break;
        }
        look *= 4;
      }
      if (ok) {
        have_pre = true;
        plan.data_off = file_base + (off_t)R.pre.data_off;
        {
          // eight times the first data line, 64 KiB at least
          const uint8_t *d = head.data() + R.pre.data_off;
          const size_t nd = head.size() - R.pre.data_off;
          const uint8_t *e = nd ? (const uint8_t *)memchr(d, R.pre.eol_byte, nd) : nullptr;
          const size_t first_line = e ? (size_t)(e - d) + 1 : nd;
          plan.spare_bytes = std::min(plan.spare_bytes, std::max<size_t>(8 * first_line, 64u << 10));
        }
        const size_t body = total - R.pre.data_off;
        plan.n_ranges = (body + plan.range_bytes - 1) / plan.range_bytes;
        ok = adopt_preamble(head.data() + R.pre.data_off, head.size() - R.pre.data_off);
      }
    } else {
      // BGZF: the leading blocks are inflated here until the #CHROM line is complete
      z_stream zs;
      memset(&zs, 0, sizeof zs);
      std::vector<uint8_t> comp, htext;
      std::vector<std::pair<size_t, size_t>> marks;
      size_t hoff = 0;
      bool bad = inflateInit2(&zs, -15) != Z_OK, at_end = false;
      int pr = 1;
      auto inflate_next = [&]() -> int {
        for (;;) {
          Frame f;
          const int r = frame_at(comp.data(), comp.size(), hoff, &f);
          if (r < 0) return -1;
          if (r == 0) {
            if ((off_t)comp.size() >= file_size - file_base) return hoff >= comp.size() ? 0 : -1;
            const size_t old = comp.size(), step = std::min<size_t>(4u << 20, total - old);
            comp.resize(old + step);
            int err = 0;
            const ssize_t g = pread_parallel(fd_in, comp.data() + old, step, file_base + (off_t)old, 1, &err);
            comp.resize(old + (g > 0 ? (size_t)g : 0));
            if (g <= 0) return -1;
            continue;
          }
          marks.emplace_back(hoff, htext.size());
          const size_t at = htext.size();
          htext.resize(at + f.isize);
          inflateReset(&zs);
          zs.next_in = comp.data() + hoff + f.in_off;
          zs.avail_in = f.in_len;
          zs.next_out = htext.data() + at;
          zs.avail_out = f.isize;
          const int zr = f.isize ? inflate(&zs, Z_FINISH) : Z_STREAM_END;
          const uint8_t *tail = comp.data() + hoff + f.total - 8;
          const uint32_t want_crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
          if ((f.isize && (zr != Z_STREAM_END || zs.avail_out != 0)) ||
              (uint32_t)crc32(crc32(0L, Z_NULL, 0), htext.data() + at, f.isize) != want_crc)
            return -1;
          hoff += f.total;
          return 1;
        }
      };
      while (!bad && pr == 1) {
        const int ir = inflate_next();
        if (ir < 0) {
          bad = true;
          break;
        }
        at_end = ir == 0;
        R.pre = Preamble();
        pr = parse_preamble(htext.data(), htext.size(), at_end, c->normalize_header, &R.pre, &msg);
        if (at_end) break;
      }
      if (bad) {
        fail("bgzf: corrupt block (inflate or CRC mismatch)", BVCF_E_FATAL);
      } else if (pr != 0) {
        fail(pr < 0 ? msg : std::string("No header found"), BVCF_E_FATAL);
      } else {
        have_pre = true;
        const size_t data_off = R.pre.data_off;
        for (int extra = 0; extra < 8; extra++) {
          if (memchr(htext.data() + data_off, R.pre.eol_byte, htext.size() - data_off)) break;
          if (inflate_next() != 1) break;
        }
        size_t b0 = marks.size();
        for (size_t i = 0; i < marks.size(); i++) {
          const size_t t_end = i + 1 < marks.size() ? marks[i + 1].second : htext.size();
          if (data_off < t_end) {
            b0 = i;
            break;
          }
        }
        size_t c0 = hoff;
        if (b0 < marks.size()) {
          plan.first_off = (uint32_t)(data_off - marks[b0].second);
          c0 = marks[b0].first;
        }
        plan.data_off = file_base + (off_t)c0;
        const size_t body = total > c0 ? total - c0 : 0;
        plan.n_ranges = (body + plan.range_bytes - 1) / plan.range_bytes;
        R.n_slots = (uint32_t)bgzf_in_flight + 1;
        max_in_flight.store(bgzf_in_flight);
        ok = adopt_preamble(htext.data() + data_off, htext.size() - data_off);
      }
      inflateEnd(&zs);
    }
    t_init = now_s() - t_start;
    if (ok) plan_ready();
  }

  // ---- stream mode: deal the blocks (workQueue <- buff, main.go:366)
  if (mode == kStream) {
    uint64_t k = 0;
    for (;;) {
      const double t0 = now_s();
      Block b = ready_q.pop();
      t_wait_read_stream += now_s() - t0;
      if (b.end) {
        if (b.bgzf_flags & 1) fail(source_err.empty() ? std::string("read error") : source_err, BVCF_E_FATAL);
        if (b.bgzf_flags & 2) fail("a line is longer than max_batch_bytes", BVCF_E_TOO_BIG);
        break;
      }
      if (failed.load()) continue;
      have_pre = true;
      if (!k) t_init = now_s() - t_start;
      const double t1 = now_s();
      workers[k % n_dev]->q.push(b);
      t_deal += now_s() - t1;
      k++;
    }
    if (plan.ready) have_pre = true;
    stop.store(true);
    if (stream_pool) stream_pool->unblock();
    stream_th.join();
  } else {
    for (auto &W : workers)
      for (auto &t : W->rd_th) t.join();
  }

  // ---- shut down
  if (!have_pre && !failed.load()) fail("EOF", BVCF_E_FATAL);
  if (!plan.ready) plan_ready();  // (after a failure, or an input without data: the device threads go on to their queues)
  for (auto &W : workers) {
    Block end;
    end.end = true;
    W->q.push(end);
  }
  for (auto &W : workers) W->dev_th.join();
  for (auto &W : workers) W->fmt_th.join();
  sink.close();
  sink.join();
  if (sink.write_failed()) fail("write failed", BVCF_E_FATAL);
  if (close_dosage(R)) fail("dosage matrix: write failed", BVCF_E_FATAL);
  if (!log.empty()) write_all(fd_err, log.data(), log.size());
  const double t_end0 = now_s();

  // the final count gather.  The sum is formed on the host; the RCCL all-reduce over the devices that took part (the
  // path's one collective) costs a communicator bring-up (seconds with eight devices, inside the run's
  // wall time) and is what a caller asks for with BVCF_RCCL=1; the totals are the same either way.
  uint64_t totals[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int used_rccl = 0;
  double t_gather = 0;
  {
    std::vector<bvcf_ctx *> live;
    for (auto &W : workers)
      if (W->ctx) {
        bvcf_result tmp;  // collect anything left after a failure so the ctx can be destroyed
        while (bvcf_collect(W->ctx, &tmp) != BVCF_E_EMPTY) {
        }
        live.push_back(W->ctx);
      }
    const char *force = getenv("BVCF_RCCL");
    const bool want_rccl = force && *force == '1';
    const double tg = now_s();
    if (!live.empty() && rc == BVCF_OK) {
      // (the summary is informational: never fail the run on it)
      if (!want_rccl) {
        bvcf_sum_counters(live.data(), (int)live.size(), totals);
      } else if (bvcf_allreduce_counters(live.data(), (int)live.size(), totals, &used_rccl) != BVCF_OK) {
        if (timing) {
          const std::string m = std::string("[bvcf timing] count gather over RCCL failed (") + bvcf_last_error(live[0]) +
                                "): summed on the host\n";
          write_all(fd_err, m.data(), m.size());
        }
        used_rccl = 0;
        bvcf_sum_counters(live.data(), (int)live.size(), totals);
      }
    }
    t_gather = now_s() - tg;
    if (!c->leave_teardown_to_exit)
      for (bvcf_ctx *x : live) bvcf_destroy(x);
  }
  if (stream_pool) stream_pool->join();
  for (auto &W : workers)
    if (W->pool) W->pool->join();
  if (!c->leave_teardown_to_exit) {
    if (stream_pool) stream_pool->free_all();
    for (auto &W : workers)
      if (W->pool) W->pool->free_all();
  }
  if (timing) {
    const double t_total = now_s() - t_start;
    double t_ctx = 0, t_gpu = 0, t_submit = 0, t_fmt_wait = 0, t_first = 0, t_fmt = 0, t_starved = 0, t_read = 0, t_warm = 0;
    size_t used = 0;
    bool have_first = false;
    for (auto &W : workers) {
      if (!W->n_blocks) continue;
      used++;
      t_ctx = std::max(t_ctx, W->t_ctx);
      t_warm = std::max(t_warm, W->t_warm);
      t_gpu = std::max(t_gpu, W->t_gpu);
      t_submit = std::max(t_submit, W->t_submit);
      t_fmt_wait = std::max(t_fmt_wait, W->t_fmt_wait);
      t_fmt = std::max(t_fmt, W->t_fmt);
      t_starved = std::max(t_starved, W->t_starved);
      t_read = std::max(t_read, W->t_read);
      if (!have_first || W->t_first_submit < t_first) t_first = W->t_first_submit;
      have_first = true;
    }
    const double t_last_write = sink.t_last_write();
    const double t_steady = t_last_write > t_start + t_first ? t_last_write - t_start - t_first : 0.0;
    const char *mode_name = mode == kStream ? "one reader for the stream" : "per-device readers over byte ranges of the file";
    if (timing_json) {
      std::string j = "[bvcf timing-json] {";
      char tmp[320];
      auto num = [&](const char *k, double v, bool comma = true) {
        j.append(tmp, (size_t)snprintf(tmp, sizeof tmp, "\"%s\": %.6f%s", k, v, comma ? ", " : ""));
      };
      num("total_s", t_total);
      num("init_s", t_init);
      num("warmup_max_s", t_warm);
      num("prepare_s", t_prepare);
      num("ctx_create_max_s", t_ctx);
      num("first_submit_at_s", t_first);
      num("last_write_at_s", t_last_write > t_start ? t_last_write - t_start : 0.0);
      num("steady_s", t_steady);
      // the time a device thread sat without a block after its first one (stream mode: the dealer waiting for the reader)
      num("wait_for_reader_s", mode == kStream ? t_wait_read_stream : t_starved);
      num("reader_busy_max_s", t_read);
      num("deal_wait_s", t_deal);
      num("submit_max_s", t_submit);
      num("gpu_wait_max_s", t_gpu);
      num("wait_for_formatter_max_s", t_fmt_wait);
      num("formatter_busy_s", t_fmt);
      num("count_gather_s", t_gather);
      num("teardown_s", now_s() - t_end0);
      j.append(tmp, (size_t)snprintf(tmp, sizeof tmp,
                                     "\"lines_in\": %llu, \"devices_used\": %zu, \"count_gather\": \"%s\", \"input\": \"%s\", \"readers\": \"%s\", ",
                                     (unsigned long long)lines_in.load(), used, used_rccl ? "rccl" : "host",
                                     input_is_bgzf_device.load() ? "bgzf, inflated on the device" : "text, gzip or bgzf through the host",
                                     mode_name));
      j.append("\"devices\": [");
      for (size_t d = 0; d < n_dev; d++)
        j.append(tmp, (size_t)snprintf(tmp, sizeof tmp,
                                       "%s{\"device\": %d, \"blocks\": %llu, \"bytes\": %llu, \"gpu_wait_s\": %.6f, \"starved_s\": %.6f, "
                                       "\"read_s\": %.6f, \"format_s\": %.6f, \"cpus_bound\": %d}",
                                       d ? ", " : "", workers[d]->device, (unsigned long long)workers[d]->n_blocks,
                                       (unsigned long long)workers[d]->n_bytes, workers[d]->t_gpu, workers[d]->t_starved,
                                       workers[d]->t_read, workers[d]->t_fmt,
                                       workers[d]->cpus.valid ? CPU_COUNT(&workers[d]->cpus.set) : 0));
      j.append("], \"counters\": [");
      for (int k = 0; k < 8; k++) j.append(tmp, (size_t)snprintf(tmp, sizeof tmp, "%s%llu", k ? ", " : "", (unsigned long long)totals[k]));
      j.append("]}\n");
      write_all(fd_err, j.data(), j.size());
    } else {
      dprintf(fd_err,
              "[bvcf timing] init %.3f (warm-up %.3f, prepare %.3f, ctx %.3f) first submit at %.3f wait-for-reader %.3f "
              "(reader busy %.3f) deal %.3f submit %.3f gpu(wait) %.3f wait-for-formatter %.3f (formatter busy %.3f) count gather "
              "%.3f teardown %.3f total %.3f s; steady %.3f s; %zu of %zu device(s), count gather: %s; %s\n",
              t_init, t_warm, t_prepare, t_ctx, t_first, mode == kStream ? t_wait_read_stream : t_starved, t_read, t_deal, t_submit,
              t_gpu, t_fmt_wait, t_fmt, t_gather, now_s() - t_end0, t_total, t_steady, used, n_dev, used_rccl ? "rccl" : "host",
              mode_name);
    }
  }
  if (n_lines_in) *n_lines_in = lines_in.load();
  return rc;
}

}  // extern "C"
