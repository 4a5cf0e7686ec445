#pragma once
// bvcf_host_internal.h — declarations shared by the host half of the path: bvcf_host.cpp (TSV assembly entry points, the
// in-memory driver), bvcf_driver.cpp / bvcf_readers.cpp (the stream driver).  Definitions: bvcf_host_common.cpp.  The
// counterpart of readVcf's preamble (main.go:241-304) and of processLines' TSV assembly (main.go:566-695).
//
// Nothing here computes what the kernels compute: rows are assembled from bvcf_result only.
#include "../../include/bvcf.h"
#include "bvcf_input.h"
#include "bvcf_bgzf.h"

#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace bvcf_host {

extern const char *const kBaseHeader[15];  // parse.Header (main.go:224)
extern const char *const kSiteNames[5];    // parse.Snp / Ins / Del / Mnp / Multi

const char *or_default(const char *s, const char *d);
void append_ll(std::string &o, long long v);
void append_g3(std::string &o, double x);  // strconv.FormatFloat(x, 'G', 3, 64), main.go:627

// sample names for the het / hom / missing lists: one contiguous arena of "name<delimiter>" entries, so that
// joining is a run of short memcpys from one array (the header's std::strings live all over the heap)
struct Names {
  std::string arena;
  std::vector<uint32_t> off;  // entry s is arena[off[s], off[s + 1]); the delimiter is its last n_delim bytes
  size_t n_delim = 0;
  uint32_t max_entry = 0;  // longest name + delimiter; the arena is padded so that 32 bytes can be read at any entry
  Names(const char *const *ptr, const uint32_t *len, size_t n, const char *delim) {
    n_delim = strlen(delim);
    off.reserve(n + 1);
    for (size_t s = 0; s < n; s++) {
      off.push_back((uint32_t)arena.size());
      arena.append(ptr[s], len[s]);
      arena.append(delim, n_delim);
      max_entry = std::max<uint32_t>(max_entry, (uint32_t)(len[s] + n_delim));
    }
    off.push_back((uint32_t)arena.size());
    arena.append(32, '\0');
  }
};

// "%.3G" of the ratios nearly every row prints: n / n_samples (heterozygosity, homozygosity, missingness of a line
// without missing genotypes) and ac / (2 n_samples) (sampleMaf).  The doubles are formed exactly as format_lines
// forms them, so a cached string is the string snprintf would produce.
struct Ratios {
  uint32_t ns = 0;
  std::vector<char> of_ns, of_2ns;  // 12 bytes per entry: length, then the characters
  explicit Ratios(uint32_t n_samples) {
    if (n_samples == 0 || n_samples > 50000) return;  // (big cohorts: 150 000 snprintf calls are not worth it up front)
    ns = n_samples;
    auto fill = [](std::vector<char> &t, uint32_t n_max, double denom) {
      t.assign((size_t)(n_max + 1) * 12, 0);
      for (uint32_t n = 0; n <= n_max; n++) {
        char tmp[64];
        const int k = snprintf(tmp, sizeof tmp, "%.3G", (double)n / denom);
        if (k > 0 && k <= 11) {
          t[(size_t)n * 12] = (char)k;
          memcpy(&t[(size_t)n * 12 + 1], tmp, (size_t)k);
        }
      }
    };
    fill(of_ns, ns, (double)ns);
    fill(of_2ns, 2 * ns, (double)(2 * ns));
  }
  // appends "%.3G" of num / den
  void append(std::string &o, uint32_t num, double den_d, uint64_t den) const {
    const std::vector<char> *t = nullptr;
    if (ns && den == ns && num <= ns)
      t = &of_ns;
    else if (ns && den == 2ull * ns && num <= 2 * ns)
      t = &of_2ns;
    if (t && (*t)[(size_t)num * 12]) {
      o.append(&(*t)[(size_t)num * 12 + 1], (size_t)(*t)[(size_t)num * 12]);
      return;
    }
    append_g3(o, (double)num / den_d);
  }
};

void join_class(std::string &o, const uint8_t *cmap, bool sparse, uint32_t ns, unsigned want, uint32_t count, const Names &nm);
const char *err_text(uint32_t code);
// where line li's bytes are: in the block the batch was submitted as, or -- bvcf_submit_bgzf with head_off -- in the
// compact copy of the line heads that came back
const char *row_of(const bvcf_result *r, const uint8_t *block, uint32_t li, const bvcf_line &L);
// line li of a batch as full records, whichever form the batch came back in (see the definition)
struct LineView {
  const bvcf_line *L;
  const bvcf_allele *A0;      // its first output allele
  const char *row = nullptr;  // where the line's bytes are when not at row_of(): rendered rows of a BGZF batch, whose text came
                              // back as the lines of the cuts only (bvcf_row_cut.text_off)
};
LineView line_view(const bvcf_result *r, uint32_t li, bvcf_line *tmp_line, bvcf_allele *tmp_allele);
void append_err(std::string &log, const bvcf_err &e, const bvcf_line &L, const char *row);
// rows of lines [lo, hi), main.go:566-695
void format_lines(const bvcf_config *c, const bvcf_result *r, const uint8_t *block, const Names &nm, const Ratios *rt,
                  uint32_t lo, uint32_t hi, std::string &out);
// CPUs this process may actually use: the smallest of the hardware's count, the affinity mask and the cgroup's CPU quota
unsigned usable_cpus();

// Persistent workers for the per-batch TSV assembly: a batch is a few thousand rows, too short to pay for
// thread creation every time.  run() hands out task indices [0, n_tasks); the caller works too.
class WorkPool {
 public:
  explicit WorkPool(unsigned n_threads) {
    for (unsigned i = 1; i < n_threads; i++) th_.emplace_back([this] { loop(); });
  }
  ~WorkPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_ = true;
    }
    wake_.notify_all();
    for (auto &t : th_) t.join();
  }
  unsigned size() const { return (unsigned)th_.size() + 1; }
  template <class F>
  void run(uint32_t n_tasks, F &&fn) {
    if (!n_tasks) return;
    auto job = std::make_shared<Job>();
    job->fn = std::forward<F>(fn);
    job->total = n_tasks;
    job->left.store(n_tasks);
    {
      std::lock_guard<std::mutex> lk(mu_);
      job_ = job;
      gen_++;
    }
    wake_.notify_all();
    work(*job);
    std::unique_lock<std::mutex> lk(job->mu);
    job->done.wait(lk, [&] { return job->left.load() == 0; });
  }

 private:
  struct Job {
    std::function<void(uint32_t)> fn;
    uint32_t total = 0;
    std::atomic<uint32_t> next{0}, left{0};
    std::mutex mu;
    std::condition_variable done;
  };
  static void work(Job &j) {
    for (;;) {
      const uint32_t t = j.next.fetch_add(1);
      if (t >= j.total) return;
      j.fn(t);
      if (j.left.fetch_sub(1) == 1) {
        std::lock_guard<std::mutex> lk(j.mu);
        j.done.notify_all();
      }
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      std::shared_ptr<Job> job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        wake_.wait(lk, [&] { return quit_ || gen_ != seen; });
        if (quit_) return;
        seen = gen_;
        job = job_;  // a worker only ever touches the job it took under the lock
      }
      work(*job);
    }
  }
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable wake_;
  std::shared_ptr<Job> job_;
  uint64_t gen_ = 0;
  bool quit_ = false;
};

void format_log(const bvcf_result *r, const uint8_t *block, std::string &log);
void format_parts(const bvcf_config *c, const bvcf_result *r, const uint8_t *block, const Names &nm, const Ratios *rt,
                  WorkPool *pool, std::vector<std::string> &parts);
void format_batch(const bvcf_config *c, const bvcf_result *r, const uint8_t *block, const Names &nm, const Ratios *rt,
                  WorkPool *pool, std::string &out, std::string &log);
char *dup_out(const std::string &s, size_t *n);

// ---- readVcf's preamble, main.go:250-304
struct Preamble {
  uint8_t eol_byte = '\n';
  uint32_t eol_chars = 1;
  std::vector<std::string> header;  // normalised
  size_t data_off = 0;              // first byte after the #CHROM line
};

// returns 0, 1 = need more input, <0 = fatal (message in *msg)
int parse_preamble(const uint8_t *in, size_t n, bool at_eof, bool normalize, Preamble *pre, std::string *msg);

struct Run {
  const bvcf_config *cfg;
  bvcf_ctx *ctx = nullptr;
  Preamble pre;
  std::vector<const char *> name_ptr;
  std::vector<uint32_t> name_len;
  unsigned n_threads = 1;
  uint64_t max_batch = 0;
  std::unique_ptr<Names> names; // built once the header is known
  std::unique_ptr<Ratios> ratios;
  bvcf_arrow *arrow = nullptr;  // --dosageOutput
  bool want_rows = true;        // !noOut
  std::unique_ptr<WorkPool> pool;  // TSV assembly workers (n_threads of them, this thread included)
  uint32_t n_slots = 2;            // result slots of the ctx
  bvcf_params params;              // what every ctx of the run is created with (prepare_run), bar the device
};

uint32_t choose_path(const Run &R, const uint8_t *data, size_t n);
int write_sample_list(const Run &R);
// What every ctx of the run shares: the sample list file, the ctx parameters (R.params), the name arena, the ratio
// strings, the formatter's worker pool, the dosage file.  Once per run, after the header is known.
int prepare_run(Run &R, std::string *msg, const uint8_t *data = nullptr, size_t n_data = 0, bool make_pool = true);
// one ctx of the run on `device` (the counterpart of one `go processLines(...)`, main.go:345-347)
int create_ctx(const Run &R, int device, bvcf_ctx **ctx, std::string *msg);
int open_ctx(Run &R, std::string *msg, const uint8_t *data = nullptr, size_t n_data = 0);
int append_dosage(Run &R, const bvcf_result *r, const uint8_t *block);
int close_dosage(Run &R);
int process_block(Run &R, const uint8_t *block, size_t n, uint64_t seq, bvcf_result *res, std::string *msg);

// a bounded FIFO between pipeline stages
template <class T>
class Channel {
 public:
  explicit Channel(size_t cap) : cap_(cap) {}
  void push(T v) {
    std::unique_lock<std::mutex> lk(mu_);
    not_full_.wait(lk, [&] { return q_.size() < cap_; });
    q_.push_back(std::move(v));
    not_empty_.notify_one();
  }
  T pop() {
    std::unique_lock<std::mutex> lk(mu_);
    not_empty_.wait(lk, [&] { return !q_.empty(); });
    T v = std::move(q_.front());
    q_.pop_front();
    not_full_.notify_one();
    return v;
  }
  bool try_pop(T *v) {
    std::lock_guard<std::mutex> lk(mu_);
    if (q_.empty()) return false;
    *v = std::move(q_.front());
    q_.pop_front();
    not_full_.notify_one();
    return true;
  }

 private:
  size_t cap_;
  std::deque<T> q_;
  std::mutex mu_;
  std::condition_variable not_full_, not_empty_;
};

int write_all(int fd, const char *p, size_t n);
double now_s();

}  // namespace bvcf_host
