// bvcf_driver.cpp — bvcf_run_fd: the reference's main() + readVcf (main.go:134-217, 241-396) as a pipeline that is
// replicated per device (bvcf_pipeline.h has the picture).  This file: the run itself, the device and formatter threads
// of a worker, the ordered sink, the timing report; the input side is bvcf_readers.cpp, the partition rules bvcf_plan.cpp.
#include "bvcf_pipeline.h"

#include <ctype.h>
#include <sys/resource.h>
#include <sys/stat.h>

namespace bvcf_host {

// ---- BufPool

void BufPool::start(int n_threads) {
  for (int t = 0; t < n_threads; t++)
    th_.emplace_back([this]() {
      for (;;) {
        if (stop_.load()) return;
        const int i = next_.fetch_add(1);
        if (i >= max_) return;
        uint8_t *p = device_ < 0 ? (uint8_t *)malloc(bytes_) : (uint8_t *)bvcf_alloc_pinned_near(device_, bytes_);
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

std::shared_ptr<BufHold> BufPool::get() {
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

void BufPool::join() {
  for (auto &t : th_)
    if (t.joinable()) t.join();
}

void BufPool::free_all() {
  join();
  std::vector<std::thread> th;
  const bool heap = device_ < 0;
  for (uint8_t *p : all_) th.emplace_back([p, heap]() { heap ? free(p) : bvcf_free_pinned(p); });
  for (auto &t : th) t.join();
  all_.clear();
}

// ---- OrderedSink

OrderedSink::~OrderedSink() {
  for (auto &kv : held_) delete kv.second.parts;
  for (Parts *p : spares_) delete p;
}

void OrderedSink::start() {
  th_ = std::thread([this]() { loop(); });
}

void OrderedSink::put(OutItem &&it) {
  size_t nb = it.log.size();
  if (it.parts)
    for (const std::string &s : *it.parts) nb += s.size();
  {
    std::unique_lock<std::mutex> lk(mu_);
    // (the awaited item is always admitted: whoever holds it has handed over everything before it)
    room_.wait(lk, [&] {
      return aborted_ || closed_ || held_bytes_ + nb <= max_held_ || (it.range == want_range_ && it.piece == want_piece_);
    });
    const auto key = std::make_pair(it.range, it.piece);
    held_bytes_ += nb;
    max_seen_ = std::max(max_seen_, held_bytes_);
    held_bytes_of_[key] = nb;
    held_.emplace(key, std::move(it));
  }
  cv_.notify_all();
}

Parts *OrderedSink::spare() {
  std::lock_guard<std::mutex> lk(mu_);
  if (spares_.empty()) return new Parts();
  Parts *p = spares_.back();
  spares_.pop_back();
  return p;
}

void OrderedSink::close() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    closed_ = true;
  }
  cv_.notify_all();
  room_.notify_all();
}

void OrderedSink::abort() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    aborted_ = true;
  }
  cv_.notify_all();
  room_.notify_all();
}

void OrderedSink::join() {
  if (th_.joinable()) th_.join();
}

void OrderedSink::loop() {
  for (;;) {
    OutItem it;
    {
      std::unique_lock<std::mutex> lk(mu_);
      const auto key = std::make_pair(want_range_, want_piece_);
      cv_.wait(lk, [&] { return aborted_ || closed_ || held_.count(key); });
      if (aborted_) return;
      auto f = held_.find(key);
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
    {
      std::lock_guard<std::mutex> lk(mu_);
      auto hb = held_bytes_of_.find(std::make_pair(it.range, it.piece));
      if (hb != held_bytes_of_.end()) {
        held_bytes_ -= hb->second;
        held_bytes_of_.erase(hb);
      }
      if (it.parts) {
        if (spares_.size() < 8)
          spares_.push_back(it.parts);
        else
          delete it.parts;
        it.parts = nullptr;
      }
      if (it.last_piece) {
        want_range_++;
        want_piece_ = 0;
      } else {
        want_piece_++;
      }
    }
    room_.notify_all();
  }
}

// ---- NUMA placement of a worker's host threads

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

// ---- the run

Driver::Driver(const bvcf_config *c, int fd_in, int fd_out, int fd_err, bool dry, unsigned dry_workers, int dry_device_inflate)
    : c_(c), fd_in_(fd_in), fd_out_(fd_out), fd_err_(fd_err), dry_(dry), dry_device_inflate_(dry_device_inflate) {
  const char *timing_env = getenv("BVCF_TIMING");
  timing_ = timing_env != nullptr && !dry;
  timing_json_ = timing_ && strcmp(timing_env, "json") == 0;
  t_start_ = now_s();
  R_.cfg = c;
  if (const char *e = getenv("BVCF_TEXT_IN_FLIGHT")) text_in_flight_ = (size_t)std::min(6, std::max(1, atoi(e)));  // tuning
  if (const char *e = getenv("BVCF_BGZF_IN_FLIGHT")) bgzf_in_flight_ = (size_t)std::min(8, std::max(1, atoi(e)));  // tuning
  max_in_flight_.store(text_in_flight_);
  // the devices of the run
  if (dry)
    dev_list_.assign(std::max(1u, dry_workers), -1);
  else if (c->n_devices && c->devices)
    dev_list_.assign(c->devices, c->devices + c->n_devices);
  else
    dev_list_.push_back(c->device);
  n_dev_ = dev_list_.size();
  // host threads are bound to the CPUs of their device's NUMA node when the run spans devices (BVCF_NUMA=0: never,
  // BVCF_NUMA=1: also with one device -- the binding code can then be exercised on a one-GPU box)
  const char *numa_env = getenv("BVCF_NUMA");
  const bool numa_off = numa_env && *numa_env == '0', numa_force = numa_env && *numa_env == '1';
  several_devices_ = !dry && (std::set<int>(dev_list_.begin(), dev_list_.end()).size() > 1 || numa_force) && !numa_off;
  hw_ = usable_cpus();
  for (size_t d = 0; d < n_dev_; d++) {
    workers_.emplace_back(new DevWorker());
    workers_.back()->idx = (uint32_t)d;
    workers_.back()->device = dev_list_[d];
  }
  ready_q_.reset(new Channel<Block>(4 * n_dev_ + 4));
}

// how the input is read: byte ranges of a regular file (text or BGZF), or one stream
void Driver::sniff_input() {
  struct stat st;
  const char *e = getenv("BVCF_RANGE_READ");
  const off_t at = lseek(fd_in_, 0, SEEK_CUR);
  if (!(e && *e == '0' && !dry_) && at >= 0 && fstat(fd_in_, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > at) {
    uint8_t magic[18];
    const ssize_t g = pread(fd_in_, magic, sizeof magic, at);
    file_base_ = at;
    file_size_ = st.st_size;
    if (g >= 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
      uint32_t xlen = 0;
      const char *di = getenv("BVCF_DEVICE_INFLATE");
      const bool device_inflate = dry_ ? dry_device_inflate_ != 0 : !(di && *di == '0');
      if (g == 18 && bvcf_bgzf::block_size(magic, 18, &xlen) != -1 && device_inflate) mode_ = kRangeBgzf;
    } else if (g > 0) {
      mode_ = kRangeText;
    }
  }
  R_.n_slots = (uint32_t)text_in_flight_ + 1;  // two batches on the device, one more being formatted
  // (BGZF inflated on the device: a batch takes as long as its slowest block -- one wave decodes a block from start to
  // end -- and every batch costs the device thread ~0.8 ms of launches and waits, so the batches are made larger: 256 MiB
  // of text = ~4 000 blocks fill the decoder's wave slots; 400 k rows of configs[2] then take 0.034 s instead of 0.063)
  R_.max_batch = c_->max_batch_bytes ? c_->max_batch_bytes : ((mode_ == kRangeBgzf ? 256ull : 64ull) << 20);
  cap_ = R_.max_batch;
  input_is_bgzf_device_.store(mode_ == kRangeBgzf);
  budget_ = plan_threads(hw_, (unsigned)n_dev_, (int)mode_);
  if (const char *t = getenv("BVCF_READ_THREADS")) budget_.copy_threads = (unsigned)std::max(1, atoi(t));  // tuning
  if (c_->n_format_threads) budget_.format_threads = std::max(1u, (unsigned)(c_->n_format_threads / n_dev_));
  if (const char *t = getenv("BVCF_FORMAT_THREADS")) budget_.format_threads = (unsigned)std::max(1, atoi(t));  // tuning
}

void Driver::fail(const std::string &m, int code) {
  {
    std::lock_guard<std::mutex> lk(fail_mu_);
    if (rc_ == BVCF_OK) {
      rc_ = code;
      log_.append(m + "\n");
    }
    failed_.store(true);
  }
  if (sink_) sink_->abort();
  // (the waiters read `failed_` under these locks: taking each once orders the store before their next check)
  {
    std::lock_guard<std::mutex> lk(plan_.mu);
  }
  plan_.cv.notify_all();
  for (auto &W : workers_) {
    {
      std::lock_guard<std::mutex> lk(W->mu);
    }
    W->cv.notify_all();
  }
}

void Driver::plan_ready() {
  {
    std::lock_guard<std::mutex> lk(plan_.mu);
    plan_.ready = true;
  }
  plan_.cv.notify_all();
}

void Driver::wait_plan() {
  std::unique_lock<std::mutex> lk(plan_.mu);
  plan_.cv.wait(lk, [&] { return plan_.ready || failed_.load(); });
}

bvcf_range_plan Driver::plan_of_run() const { return plan_.p; }

// the preamble of the input is known (main.go:250-304): the ctx parameters; then the workers may start
bool Driver::adopt_preamble(const uint8_t *data, size_t n_data) {
  const double t0 = now_s();
  eol_byte_.store(R_.pre.eol_byte);
  if (dry_) return true;
  std::string msg;
  const int r = prepare_run(R_, &msg, data, n_data, false);
  if (r) {
    fail(msg, r);
    return false;
  }
  if (R_.pre.header.size() == 9) {
    const char *m = "Found 9 header fields. When genotypes present, we expect 1+ samples after FORMAT (10 fields minimum)\n";
    write_all(fd_err_, m, strlen(m));
  }
  for (auto &W : workers_)
    if (R_.want_rows && budget_.format_threads > 1) W->fmt_pool.reset(new WorkPool(budget_.format_threads));
  t_prepare_ = now_s() - t0;
  return true;
}

// ---- device thread of a worker: the goroutine of main.go:345-347 with a GPU behind it
void Driver::device_main(DevWorker *W) {
  if (several_devices_) bind_here(W->cpus);
  {
    const double t0 = now_s();
    bvcf_warmup(W->device);  // runtime + kernels onto the device while the header is read (errors: bvcf_create reports them)
    W->t_warm = now_s() - t0;
  }
  wait_plan();
  std::string wmsg;
  auto make_ctx = [&]() {
    const double tc = now_s();
    const int r = create_ctx(R_, W->device, &W->ctx, &wmsg);
    W->t_ctx = now_s() - tc;
    if (r) fail(wmsg, r);
    return r == 0;
  };
  // The ctx is created while the worker's reader fills its first buffer -- when it is known that a block will come:
  // in range mode worker k has work iff there are more than k ranges; of a stream only the first worker is sure of one.
  if (!failed_.load() && (mode_ == kStream ? W->idx == 0 : W->idx < plan_.p.n_ranges)) make_ctx();
  std::deque<Block> in_flight;  // submitted, not yet collected (oldest first)
  // Collect number q of the ctx lands in result slot q % n_slots, whose arrays the formatter may still be reading for
  // the batch collected n_slots collects ago.
  uint64_t n_collects = 0, n_jobs = 0;
  std::deque<std::pair<uint64_t, uint64_t>> outstanding;  // (job number, collect number) of jobs not known finished
  auto wait_formatted = [&](uint64_t n) {
    const double t0 = now_s();
    std::unique_lock<std::mutex> lk(W->mu);
    W->cv.wait(lk, [&] { return W->fmt_done >= n || failed_.load(); });
    W->t_fmt_wait += now_s() - t0;
  };
  auto slot_is_free = [&]() {
    uint64_t need = 0;
    while (!outstanding.empty() && outstanding.front().second + R_.n_slots <= n_collects) {
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
    if (failed_.load()) return;
    const double t0 = now_s();
    int r = bvcf_collect(W->ctx, &res);
    n_collects++;
    // (a batch can ask twice: what it needs of one array may only show once another has grown)
    for (int grown = 0; r == BVCF_E_CAPACITY && grown < 4; grown++) {
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
    lines_in_.fetch_add(res.n_lines_seen);
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
    if (failed_.load()) continue;
    if (!b.nb) {
      // nothing in this block (a range inside one long line): it still takes its turn in the output
      FmtJob j;
      j.b = b;
      W->fq.push(std::move(j));
      continue;
    }
    if (!W->ctx && !make_ctx()) continue;
    // keep one block (BGZF on the device: two) ahead of the one being collected
    if (in_flight.size() >= max_in_flight_.load()) finish_oldest();
    if (failed_.load()) continue;
    const double ts = now_s();
    const int r = submit(b);
    if (!W->n_blocks) W->t_first_submit = now_s() - t_start_;
    W->t_submit += now_s() - ts;
    if (r) {
      fail(std::string("bvcf_submit: ") + bvcf_last_error(W->ctx), r);
      continue;
    }
    W->n_blocks++;
    W->n_bytes += b.bgzf ? b.own : b.nb;
    in_flight.push_back(b);
  }
  while (!failed_.load() && !in_flight.empty()) finish_oldest();
  FmtJob e;
  e.end = true;
  W->fq.push(std::move(e));
}

// bvcf_plan_fd: what the device thread would have been given
void Driver::dry_main(DevWorker *W) {
  for (;;) {
    Block b = W->q.pop();
    if (b.end) break;
    bvcf_plan_block pb;
    memset(&pb, 0, sizeof pb);
    pb.worker = W->idx;
    pb.piece = b.piece;
    pb.range = b.range;
    pb.file_off = b.file_off;
    pb.nbytes = b.nb;
    pb.own = b.own;
    pb.first_off = b.first_off;
    pb.bgzf = b.bgzf;
    pb.bgzf_flags = (uint8_t)b.bgzf_flags;
    pb.last_piece = b.last_piece;
    std::lock_guard<std::mutex> lk(dry_mu_);
    dry_blocks.push_back(pb);
  }
}

// ---- formatter thread of a worker: main.go:566-695 for its collected batches
void Driver::formatter_main(DevWorker *W) {
  if (several_devices_) bind_here(W->cpus);
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
    if (failed_.load()) {
      if (j.has_res) release();
      continue;
    }
    const double t0 = now_s();
    OutItem it;
    it.range = j.b.range;
    it.piece = j.b.piece;
    it.last_piece = j.b.last_piece;
    it.parts = sink_->spare();
    if (j.has_res) {
      const uint8_t *text = j.b.bgzf ? j.res.text : j.b.data;  // (inflated on the device: the copy that came back)
      format_log(&j.res, text, it.log);
      if (R_.want_rows)
        format_parts(c_, &j.res, text, *R_.names, R_.ratios.get(), W->fmt_pool.get(), *it.parts);
      else
        for (auto &q : *it.parts) q.clear();
      if (R_.arrow && j.res.dosage) {
        // the dosage rows go into the file in input order (main.go:576-584): the sink runs this when it is the block's
        // turn; the result slot and the text stay on loan until then
        auto keep = std::make_shared<FmtJob>(std::move(j));
        it.in_order = [this, keep, text, release]() {
          if (!dosage_failed_.load() && append_dosage(R_, &keep->res, text)) {
            dosage_failed_.store(true);
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
    sink_->put(std::move(it));
    if (sink_->write_failed()) fail("write failed", BVCF_E_FATAL);
  }
}

int Driver::run(uint64_t *n_lines_in) {
  sniff_input();
  // fmt.Fprintln(writer, stringHeader(config)), main.go:196-200
  if (!dry_ && !c_->no_out) {
    char h[512];
    size_t hn = bvcf_string_header(c_, h, sizeof h);
    h[hn] = '\n';
    if (write_all(fd_out_, h, hn + 1)) {
      dprintf(fd_err_, "write failed\n");
      return BVCF_E_FATAL;
    }
  }
  if (!dry_ && bvcf_device_count() <= 0) {
    // fail loudly: there is no CPU path
    dprintf(fd_err_, "cannot allocate pinned host memory (no usable HIP device?)\n");
    return BVCF_E_NODEV;
  }
  size_t sink_mb = 512;
  if (const char *e = getenv("BVCF_SINK_MB")) sink_mb = (size_t)std::max(1, atoi(e));
  sink_.reset(new OrderedSink(fd_out_, fd_err_, sink_mb << 20));

  // ---- start: the sink, the workers (their devices warm up while the header is read), the reader(s)
  if (!dry_) sink_->start();
  for (auto &W : workers_) {
    if (several_devices_) W->cpus = cpus_near_device(W->device);
    DevWorker *w = W.get();
    if (dry_) {
      W->dev_th = std::thread([this, w] { dry_main(w); });
    } else {
      W->dev_th = std::thread([this, w] { device_main(w); });
      W->fmt_th = std::thread([this, w] { formatter_main(w); });
    }
  }
  bool have_pre = false;
  if (mode_ == kStream) {
    std::thread stream_th([this] { stream_reader(); });
    // deal the blocks (workQueue <- buff, main.go:366)
    uint64_t k = 0;
    for (;;) {
      const double t0 = now_s();
      Block b = ready_q_->pop();
      t_wait_read_stream_ += now_s() - t0;
      if (b.end) {
        if (b.bgzf_flags & 1) fail(source_err_.empty() ? std::string("read error") : source_err_, BVCF_E_FATAL);
        if (b.bgzf_flags & 2) fail("a line is longer than max_batch_bytes", BVCF_E_TOO_BIG);
        break;
      }
      if (failed_.load()) continue;
      if (!k) t_init_ = now_s() - t_start_;
      const double t1 = now_s();
      workers_[k % n_dev_]->q.push(b);
      t_deal_ += now_s() - t1;
      k++;
    }
    have_pre = plan_.ready || k > 0;
    stop_.store(true);
    if (stream_pool_) stream_pool_->unblock();
    stream_th.join();
  } else {
    // the worker's buffers are pinned while the header is read; a text buffer holds a range and the spare room for the
    // line that straddles its end, a BGZF buffer the compressed range and some look-ahead blocks
    size_t buf_bytes = cap_;
    if (mode_ == kRangeBgzf) {
      plan_.p.range_bytes = bgzf_range_bytes((uint64_t)(file_size_ - file_base_), (unsigned)n_dev_, cap_);
      buf_bytes = (size_t)plan_.p.range_bytes + (1u << 20);
    }
    for (auto &W : workers_) {
      // being read into, one queued, two (BGZF: three) on the device, up to two with the formatter
      W->pool.reset(new BufPool(W->device, buf_bytes, mode_ == kRangeBgzf ? (int)bgzf_in_flight_ + 2 : 5 + (int)text_in_flight_));
      W->pool->start(mode_ == kRangeBgzf ? 1 : 2);
      DevWorker *w = W.get();
      if (mode_ == kRangeBgzf)
        W->rd_th.emplace_back([this, w] { range_bgzf_reader(w); });
      else
        for (unsigned r = 0; r < std::max(1u, budget_.readers); r++) W->rd_th.emplace_back([this, w, r] { range_text_reader(w, r); });
    }
    have_pre = read_file_header();
    t_init_ = now_s() - t_start_;
    // (after a failure the readers and device threads leave wait_plan through `failed_`)
    if (have_pre)
      plan_ready();
    else if (!failed_.load())
      fail("EOF", BVCF_E_FATAL);
    for (auto &W : workers_)
      for (auto &t : W->rd_th) t.join();
    have_pre = have_pre || failed_.load();
  }

  // ---- shut down
  if (!have_pre && !failed_.load()) fail("EOF", BVCF_E_FATAL);
  if (!plan_.ready) plan_ready();  // (after a failure, or an input without data: the device threads go on to their queues)
  for (auto &W : workers_) {
    Block end;
    end.end = true;
    W->q.push(end);
  }
  for (auto &W : workers_) W->dev_th.join();
  for (auto &W : workers_)
    if (W->fmt_th.joinable()) W->fmt_th.join();
  sink_->close();
  sink_->join();
  if (sink_->write_failed()) fail("write failed", BVCF_E_FATAL);
  if (close_dosage(R_)) fail("dosage matrix: write failed", BVCF_E_FATAL);
  if (!log_.empty()) write_all(fd_err_, log_.data(), log_.size());
  const double t_end0 = now_s();

  // the final count gather.  The sum is formed on the host; the RCCL all-reduce over the devices that took part (the
  // path's one collective) costs a communicator bring-up (seconds with eight devices, inside the run's
  // wall time) and is what a caller asks for with BVCF_RCCL=1; the totals are the same either way.
  uint64_t totals[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int used_rccl = 0;
  double t_gather = 0;
  {
    std::vector<bvcf_ctx *> live;
    for (auto &W : workers_)
      if (W->ctx) {
        bvcf_result tmp;  // collect anything left after a failure so the ctx can be destroyed
        while (bvcf_collect(W->ctx, &tmp) != BVCF_E_EMPTY) {
        }
        live.push_back(W->ctx);
      }
    const char *force = getenv("BVCF_RCCL");
    const bool want_rccl = force && *force == '1';
    const double tg = now_s();
    if (!live.empty() && rc_ == BVCF_OK) {
      // (the summary is informational: never fail the run on it)
      if (!want_rccl) {
        bvcf_sum_counters(live.data(), (int)live.size(), totals);
      } else if (bvcf_allreduce_counters(live.data(), (int)live.size(), totals, &used_rccl) != BVCF_OK) {
        if (timing_) {
          const std::string m = std::string("[bvcf timing] count gather over RCCL failed (") + bvcf_last_error(live[0]) + "): summed on the host\n";
          write_all(fd_err_, m.data(), m.size());
        }
        used_rccl = 0;
        bvcf_sum_counters(live.data(), (int)live.size(), totals);
      }
    }
    t_gather = now_s() - tg;
    if (!c_->leave_teardown_to_exit)
      for (bvcf_ctx *x : live) bvcf_destroy(x);
  }
  if (stream_pool_) stream_pool_->join();
  for (auto &W : workers_)
    if (W->pool) W->pool->join();
  if (dry_ || !c_->leave_teardown_to_exit) {
    if (stream_pool_) stream_pool_->free_all();
    for (auto &W : workers_)
      if (W->pool) W->pool->free_all();
  }
  if (timing_) report_timing(t_end0, totals, used_rccl, t_gather);
  if (dry_)
    std::sort(dry_blocks.begin(), dry_blocks.end(), [](const bvcf_plan_block &x, const bvcf_plan_block &y) {
      return x.range != y.range ? x.range < y.range : x.piece < y.piece;
    });
  if (n_lines_in) *n_lines_in = lines_in_.load();
  return rc_;
}

// BVCF_TIMING=1: one line; BVCF_TIMING=json: one JSON object with the stage times of the run
void Driver::report_timing(double t_end0, const uint64_t totals[8], int used_rccl, double t_gather) {
  const double t_total = now_s() - t_start_;
  double t_ctx = 0, t_gpu = 0, t_submit = 0, t_fmt_wait = 0, t_first = 0, t_fmt = 0, t_starved = 0, t_read = 0, t_warm = 0;
  size_t used = 0;
  bool have_first = false;
  for (auto &W : workers_) {
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
  const double t_last_write = sink_->t_last_write();
  const double t_steady = t_last_write > t_start_ + t_first ? t_last_write - t_start_ - t_first : 0.0;
  const char *mode_name = mode_ == kStream ? "one reader for the stream" : "per-device readers over byte ranges of the file";
  const double t_wait_reader = mode_ == kStream ? t_wait_read_stream_ : t_starved;
  if (!timing_json_) {
    dprintf(fd_err_,
            "[bvcf timing] init %.3f (warm-up %.3f, prepare %.3f, ctx %.3f) first submit at %.3f wait-for-reader %.3f "
            "(reader busy %.3f) deal %.3f submit %.3f gpu(wait) %.3f wait-for-formatter %.3f (formatter busy %.3f) count gather "
            "%.3f teardown %.3f total %.3f s; steady %.3f s; %zu of %zu device(s), count gather: %s; %s\n",
            t_init_, t_warm, t_prepare_, t_ctx, t_first, t_wait_reader, t_read, t_deal_, t_submit, t_gpu, t_fmt_wait, t_fmt, t_gather,
            now_s() - t_end0, t_total, t_steady, used, n_dev_, used_rccl ? "rccl" : "host", mode_name);
    return;
  }
  std::string j = "[bvcf timing-json] {";
  char tmp[320];
  auto num = [&](const char *k, double v) { j.append(tmp, (size_t)snprintf(tmp, sizeof tmp, "\"%s\": %.6f, ", k, v)); };
  num("total_s", t_total);
  num("init_s", t_init_);
  num("warmup_max_s", t_warm);
  num("prepare_s", t_prepare_);
  num("ctx_create_max_s", t_ctx);
  num("first_submit_at_s", t_first);
  num("last_write_at_s", t_last_write > t_start_ ? t_last_write - t_start_ : 0.0);
  num("steady_s", t_steady);
  // the time a device thread sat without a block after its first one (stream mode: the dealer waiting for the reader)
  num("wait_for_reader_s", t_wait_reader);
  num("reader_busy_max_s", t_read);
  num("deal_wait_s", t_deal_);
  num("submit_max_s", t_submit);
  num("gpu_wait_max_s", t_gpu);
  num("wait_for_formatter_max_s", t_fmt_wait);
  num("formatter_busy_s", t_fmt);
  num("count_gather_s", t_gather);
  num("teardown_s", now_s() - t_end0);
  {
    // what the process paid the kernel for so far (a first exec on a box pages the HIP runtime's libraries in from
    // the image: major faults and blocks read; a later one finds them in the page cache)
    struct rusage ru;
    memset(&ru, 0, sizeof ru);
    getrusage(RUSAGE_SELF, &ru);
    num("major_faults", (double)ru.ru_majflt);
    num("minor_faults", (double)ru.ru_minflt);
    num("in_blocks", (double)ru.ru_inblock);
    num("user_cpu_s", (double)ru.ru_utime.tv_sec + 1e-6 * (double)ru.ru_utime.tv_usec);
    num("system_cpu_s", (double)ru.ru_stime.tv_sec + 1e-6 * (double)ru.ru_stime.tv_usec);
  }
  num("sink_max_held_MB", (double)sink_->max_held_seen() / 1048576.0);
  j.append(tmp, (size_t)snprintf(tmp, sizeof tmp,
                                 "\"threads\": {\"cpus\": %u, \"readers_per_worker\": %u, \"copy_threads_per_reader\": %u, "
                                 "\"format_threads_per_worker\": %u, \"busy_total\": %u}, ",
                                 hw_, budget_.readers, budget_.copy_threads, budget_.format_threads, budget_.busy_total));
  j.append(tmp, (size_t)snprintf(tmp, sizeof tmp,
                                 "\"lines_in\": %llu, \"devices_used\": %zu, \"count_gather\": \"%s\", \"input\": \"%s\", \"readers\": \"%s\", ",
                                 (unsigned long long)lines_in_.load(), used, used_rccl ? "rccl" : "host",
                                 input_is_bgzf_device_.load() ? "bgzf, inflated on the device" : "text, gzip or bgzf through the host", mode_name));
  j.append("\"devices\": [");
  for (size_t d = 0; d < n_dev_; d++) {
    const DevWorker &W = *workers_[d];
    j.append(tmp, (size_t)snprintf(tmp, sizeof tmp,
                                   "%s{\"device\": %d, \"blocks\": %llu, \"bytes\": %llu, \"gpu_wait_s\": %.6f, \"starved_s\": %.6f, "
                                   "\"read_s\": %.6f, \"format_s\": %.6f, \"warmup_s\": %.6f, \"cpus_bound\": %d}",
                                   d ? ", " : "", W.device, (unsigned long long)W.n_blocks, (unsigned long long)W.n_bytes, W.t_gpu,
                                   W.t_starved, W.t_read, W.t_fmt, W.t_warm, W.cpus.valid ? CPU_COUNT(&W.cpus.set) : 0));
  }
  j.append("], \"counters\": [");
  for (int k = 0; k < 8; k++) j.append(tmp, (size_t)snprintf(tmp, sizeof tmp, "%s%llu", k ? ", " : "", (unsigned long long)totals[k]));
  j.append("]}\n");
  write_all(fd_err_, j.data(), j.size());
}

}  // namespace bvcf_host

using namespace bvcf_host;

extern "C" {

int bvcf_run_fd(const bvcf_config *c, int fd_in, int fd_out, int fd_err, uint64_t *n_lines_in) {
  if (!c) return BVCF_E_ARG;
  Driver d(c, fd_in, fd_out, fd_err);
  return d.run(n_lines_in);
}

int bvcf_plan_fd(int fd_in, int fd_err, uint32_t n_workers, uint64_t max_batch_bytes, int device_inflate, bvcf_plan_block *out,
                 size_t cap, size_t *n_out, int *mode_out, bvcf_range_plan *plan_out) {
  if (!n_workers || n_workers > 64 || (!out && cap) || !n_out) return BVCF_E_ARG;
  bvcf_config c;
  bvcf_config_defaults(&c);
  c.max_batch_bytes = max_batch_bytes;
  Driver d(&c, fd_in, -1, fd_err, true, n_workers, device_inflate);
  const int rc = d.run(nullptr);
  *n_out = d.dry_blocks.size();
  for (size_t i = 0; i < d.dry_blocks.size() && i < cap; i++) out[i] = d.dry_blocks[i];
  if (mode_out) *mode_out = (int)d.mode();
  if (plan_out) *plan_out = d.plan_of_run();
  return rc;
}

}  // extern "C"
