#pragma once
// bvcf_pipeline.h — the pieces bvcf_run_fd is assembled from (bvcf_driver.cpp: the run and its device / formatter
// threads; bvcf_readers.cpp: the input side; bvcf_plan.cpp: the pure partition logic).
//
// The reference has ONE producer (readVcf's loop, main.go:349-380) feeding NumCPU workers (main.go:345-347).  One
// producer cannot feed several GPUs (it tops out at one PCIe link's worth of text), so here everything between the
// input file and the ordered output belongs to a device worker:
//
//   per device worker k        reader thread(s)  its own byte ranges of the file -> its own pinned buffers
//                              device thread     one ctx: bvcf_submit one block ahead, bvcf_collect the oldest
//                              formatter thread  TSV assembly (main.go:566-695) on the worker's thread pool
//   one for the run            ordered sink      rows and log lines to fd_out / fd_err in input order (main.go:524-532)
//
// Range mode (the input is a regular file, text or BGZF): range i of the data goes to worker i mod N; which lines a
// range owns is include/bvcf_plan.h's rule.  Stream mode (a pipe, a single-stream gzip): one reader thread cuts the
// blocks and the run's thread deals block k to worker k mod N, as the reference's single producer does.  Either way
// the output is the same bytes in the same order for any device list: blocks are ordered by (range, piece).
#include "bvcf_host_internal.h"
#include "../../include/bvcf_plan.h"

#include <atomic>
#include <sched.h>

#include <set>

namespace bvcf_host {

typedef std::vector<std::string> Parts;

// ---- bvcf_plan.cpp
struct Frame {  // one BGZF block inside a reader's window
  size_t off;   // of the block in the window
  uint32_t total, in_off, in_len, isize;
};
struct BgzfBatch {
  size_t n_own = 0, own_bytes = 0, own_text = 0, la = 0, la_text = 0;
  bool at_eof = false, bad = false, too_long = false;
};
bvcf_range_plan plan_text_ranges(uint64_t file_size, uint64_t data_off, uint64_t cap, uint64_t first_line_bytes);
uint64_t bgzf_range_bytes(uint64_t total, unsigned n_workers, uint64_t cap);
bvcf_range_plan plan_bgzf_ranges(uint64_t file_size, uint64_t data_off, unsigned n_workers, uint64_t cap);
bvcf_text_cut cut_text_range(const uint8_t *buf, size_t n, size_t own_len, bool first_range, bool last_range, uint8_t eol);
long find_block_chain(const uint8_t *buf, size_t n, size_t from);
// the block at window offset off: 1 and *f filled, 0 the window ends before the block does, -1 malformed
int frame_at(const uint8_t *buf, size_t n, size_t off, Frame *f);
// Is there a terminator in the text of the block whose deflate payload is p[0, n)?  Inflates only as far as needed.
// 1 yes, 0 no, -1 the data is not valid DEFLATE.
int block_has_eol(z_stream &zs, const uint8_t *p, uint32_t n, uint8_t eol);
BgzfBatch cut_bgzf_batch(const std::function<int(size_t, Frame *)> &frame, const std::function<const uint8_t *(size_t)> &at,
                         z_stream &zs, size_t pos, size_t limit, size_t cap, size_t small, size_t la_reserve, uint8_t eol);
bvcf_thread_budget plan_threads(unsigned cpus, unsigned n_workers, int mode);

// ---- buffers and blocks

// a buffer on loan from a pool (it goes home when the last block cut from it is done with), or bytes of its own
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
  uint64_t file_off = 0;  // where data[0] is in the input file (range modes; reported by bvcf_plan_fd)
  bool end = false;       // queue terminator
};

// pinned buffers of one size, allocated in the background (pinning 64 MiB takes ~25 ms: the first block is being read
// while the next buffers are pinned), at most `max` of them, none after stop().  device < 0: plain heap memory
// (bvcf_plan_fd, which runs the readers without a device).
class BufPool {
 public:
  BufPool(int device, size_t bytes, int max) : device_(device), bytes_(bytes), max_(max), free_(1024) {}
  ~BufPool() {
    join();
    if (device_ < 0) free_all();
  }
  void start(int n_threads = 2);
  // a free buffer (nullptr: pinning failed, or unblock() was called)
  std::shared_ptr<BufHold> get();
  void stop() { stop_.store(true); }
  void unblock() { free_.push(nullptr); }
  void join();
  bool failed() const { return failed_.load(); }
  size_t bytes() const { return bytes_; }
  void free_all();

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

// Bounded: put() waits while more than `max_held_bytes` of formatted rows are waiting for their turn, unless the item is
// the one the writer is waiting for.  Every worker hands its blocks over in (range, piece) order (the readers take
// turns), so the owner of the awaited item is never stuck behind one of its own later items: no deadlock.  A slow
// fd_out (a pipe into gzip, a disk) thus holds the formatters, they hold their result slots, and the device threads
// stop submitting -- as the bounded write queue of the reference's workers does (main.go:524-532).
class OrderedSink {
 public:
  OrderedSink(int fd_out, int fd_err, size_t max_held_bytes) : fd_out_(fd_out), fd_err_(fd_err), max_held_(max_held_bytes) {}
  ~OrderedSink();
  void start();
  void put(OutItem &&it);
  Parts *spare();
  void close();
  void abort();
  void join();
  bool write_failed() const { return write_failed_.load(); }
  double t_last_write() const { return t_last_write_; }
  size_t max_held_seen() const { return max_seen_; }

 private:
  void loop();
  int fd_out_, fd_err_;
  size_t max_held_;
  std::mutex mu_;
  std::condition_variable cv_, room_;
  std::map<std::pair<uint64_t, uint32_t>, OutItem> held_;
  std::map<std::pair<uint64_t, uint32_t>, size_t> held_bytes_of_;
  size_t held_bytes_ = 0, max_seen_ = 0;
  uint64_t want_range_ = 0;
  uint32_t want_piece_ = 0;
  std::vector<Parts *> spares_;
  bool closed_ = false, aborted_ = false;
  std::atomic<bool> write_failed_{false};
  double t_last_write_ = 0;
  std::thread th_;
};

// The threads of a device worker run on the cores of the NUMA node its GPU hangs off (the pinned buffers they fill
// are placed there by bvcf_alloc_pinned_near).  Only with several devices; BVCF_NUMA=0 turns it off.  Best effort.
struct NodeCpus {
  cpu_set_t set;
  bool valid = false;
};
NodeCpus cpus_near_device(int device);
inline void bind_here(const NodeCpus &nc) {
  if (nc.valid) sched_setaffinity(0, sizeof nc.set, &nc.set);
}

// n bytes at file offset off into dst, by up to n_thr threads (one thread copies out of the page cache at about
// 10 GB/s).  Returns the bytes read (short only at the end of the file), or -1 with errno in *err.
ssize_t pread_parallel(int fd, uint8_t *dst, size_t n, off_t off, unsigned n_thr, int *err);

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
  Channel<Block> q{3};    // blocks for this device
  Channel<FmtJob> fq{8};  // collected batches for its formatter
  std::thread dev_th, fmt_th;
  std::vector<std::thread> rd_th;
  std::unique_ptr<BufPool> pool;       // range modes: the worker's own pinned buffers
  std::unique_ptr<WorkPool> fmt_pool;  // TSV assembly threads
  NodeCpus cpus;
  // result slots: jobs of this worker that are done with (formatted; with a dosage file: appended in order)
  std::mutex mu;
  std::condition_variable cv;
  uint64_t fmt_done = 0;          // every job below this number is done with
  std::set<uint64_t> done_early;  // ... and these above it
  // the worker's readers hand their ranges over in order: next_push = the worker-local number of the range whose turn it is
  uint64_t next_push = 0;
  std::atomic<unsigned> readers_done{0};  // range_text_reader: the LAST of the worker's readers stops its buffer pool
  // timing
  double t_warm = 0, t_ctx = 0, t_submit = 0, t_gpu = 0, t_fmt_wait = 0, t_first_submit = 0, t_starved = 0, t_fmt = 0, t_read = 0;
  uint64_t n_blocks = 0, n_bytes = 0;
};

enum Mode { kStream = BVCF_MODE_STREAM, kRangeText = BVCF_MODE_TEXT_RANGES, kRangeBgzf = BVCF_MODE_BGZF_RANGES };

// One run of bvcf_run_fd (main.go:134-217 + readVcf, main.go:241-396), or -- dry -- of bvcf_plan_fd.
class Driver {
 public:
  Driver(const bvcf_config *c, int fd_in, int fd_out, int fd_err, bool dry = false, unsigned dry_workers = 1,
         int dry_device_inflate = 1);
  int run(uint64_t *n_lines_in);
  // bvcf_plan_fd: the blocks the workers received, in (range, piece) order
  std::vector<bvcf_plan_block> dry_blocks;
  bvcf_range_plan plan_of_run() const;
  Mode mode() const { return mode_; }

 private:
  // bvcf_driver.cpp
  void sniff_input();
  void fail(const std::string &m, int code);
  void plan_ready();
  void wait_plan();
  bool adopt_preamble(const uint8_t *data, size_t n_data);
  void device_main(DevWorker *W);
  void dry_main(DevWorker *W);
  void formatter_main(DevWorker *W);
  void report_timing(double t_end0, const uint64_t totals[8], int used_rccl, double t_gather);
  // bvcf_readers.cpp
  bool read_file_header();                  // range modes: the preamble, the plan
  bool read_text_header();
  bool read_bgzf_header();
  void push_in_turn(DevWorker *W, uint64_t local, std::vector<Block> &blocks);
  void range_text_reader(DevWorker *W, unsigned r);
  void range_bgzf_reader(DevWorker *W);
  void stream_reader();
  void stream_bgzf(bvcf_input::ByteSource &src);
  void push_end(bool read_error, bool too_long);

  const bvcf_config *c_;
  int fd_in_, fd_out_, fd_err_;
  bool dry_;
  int dry_device_inflate_;
  bool timing_ = false, timing_json_ = false;
  double t_start_ = 0, t_init_ = 0, t_prepare_ = 0, t_deal_ = 0, t_wait_read_stream_ = 0;
  Mode mode_ = kStream;
  off_t file_base_ = 0, file_size_ = 0;
  Run R_;
  size_t cap_ = 0;  // max_batch_bytes of the run
  std::atomic<uint64_t> lines_in_{0};
  std::vector<int> dev_list_;
  size_t n_dev_ = 1;
  bool several_devices_ = false;
  bvcf_thread_budget budget_;
  unsigned hw_ = 1;
  // first error wins; everything then drains
  std::unique_ptr<OrderedSink> sink_;
  std::mutex fail_mu_;
  int rc_ = BVCF_OK;
  std::string log_;
  std::atomic<bool> failed_{false};
  // what the workers wait for: the header is known, the ctx parameters are set, the ranges are laid out
  struct {
    std::mutex mu;
    std::condition_variable cv;
    bool ready = false;
    bvcf_range_plan p = {0, 0, 0, 0};
    uint32_t first_off = 0;  // BGZF: where in the first block's text the data lines start
  } plan_;
  std::vector<std::unique_ptr<DevWorker>> workers_;
  std::atomic<size_t> max_in_flight_{2};  // batches a device worker keeps submitted
  size_t text_in_flight_ = 2, bgzf_in_flight_ = 3;
  std::atomic<bool> input_is_bgzf_device_{false};
  std::atomic<uint8_t> eol_byte_{'\n'};
  std::atomic<bool> dosage_failed_{false};
  // stream mode
  std::string source_err_;
  std::unique_ptr<Channel<Block>> ready_q_;
  std::unique_ptr<BufPool> stream_pool_;
  std::atomic<bool> stop_{false};
  std::mutex dry_mu_;
};

}  // namespace bvcf_host
