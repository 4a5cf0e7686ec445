// bvcf_host.cpp — host half of the path, above the C-ABI: the counterpart of readVcf's preamble and
// producer loop (main.go:241-396) and of processLines' TSV assembly (main.go:566-695).
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

namespace {

// parse.Header (main.go:224), pinned by main_test.go:79-80
const char *const kBaseHeader[15] = {"chrom",       "pos",           "type",         "ref",          "alt",
                                     "trTv",        "heterozygotes", "heterozygosity", "homozygotes", "homozygosity",
                                     "missingGenos", "missingness",  "ac",           "an",           "sampleMaf"};

// parse.Snp / Ins / Del / Mnp / Multi
const char *const kSiteNames[5] = {"SNP", "INS", "DEL", "MNP", "MULTIALLELIC"};

const char *or_default(const char *s, const char *d) { return s ? s : d; }

void append_ll(std::string &o, long long v) {
  char tmp[24];
  char *e = tmp + sizeof tmp, *p = e;
  unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
  do {
    *--p = (char)('0' + u % 10);
    u /= 10;
  } while (u);
  if (v < 0) *--p = '-';
  o.append(p, (size_t)(e - p));
}

// strconv.FormatFloat(x, 'G', 3, 64) (main.go:627); "%.3G" is identical on [0, 1]
void append_g3(std::string &o, double x) {
  char tmp[64];
  int n = snprintf(tmp, sizeof tmp, "%.3G", x);
  o.append(tmp, (size_t)n);
}

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

// strings.Join(names of the `count` samples with class `want`, fieldDelimiter); `sparse`: the map is a list of its
// non-zero bytes (BVCF_ALLELE_CMAP_SPARSE).  The output is sized for `count` entries up front and written with
// fixed-size copies; the map is read eight bytes (32 samples) at a time.
void join_class(std::string &o, const uint8_t *cmap, bool sparse, uint32_t ns, unsigned want, uint32_t count,
                const Names &nm) {
  const size_t at = o.size();
  o.resize(at + (size_t)count * nm.max_entry + 32);
  char *const w0 = &o[at];
  char *w = w0;
  const char *const arena = nm.arena.data();
  const uint32_t wide = nm.max_entry <= 16 ? 16u : (nm.max_entry <= 32 ? 32u : 0u);
  uint32_t k = 0;
  // groups j (2 bits each) of x that hold `want`, for sample base s0; false once `count` names are out
  auto emit = [&](uint64_t x, uint32_t s0) -> bool {
    const uint64_t y = x ^ (want * 0x5555555555555555ull);
    uint64_t m = ~(y | (y >> 1)) & 0x5555555555555555ull;
    while (m) {
      const uint32_t sidx = s0 + ((uint32_t)__builtin_ctzll(m) >> 1);
      m &= m - 1;
      if (sidx >= ns || k == count) return false;
      const uint32_t a = nm.off[sidx], n = nm.off[sidx + 1] - a;  // name + delimiter
      if (wide == 16)
        memcpy(w, arena + a, 16);
      else if (wide == 32)
        memcpy(w, arena + a, 32);
      else
        memcpy(w, arena + a, n);
      w += n;
      k++;
    }
    return true;
  };
  if (sparse) {
    uint32_t n;
    memcpy(&n, cmap, 4);
    for (uint32_t i = 0; i < n && i < BVCF_CMAP_SPARSE_MAX; i++) {
      uint32_t e;
      memcpy(&e, cmap + 4 + 4 * i, 4);
      // the bits above the byte must not look like class-`want` groups: 0 never is (want != 0)
      if (!emit(e & 0xFFu, (e >> 8) * 4u)) break;
    }
  } else {
    const uint32_t nbytes = (ns + 3) / 4;
    for (uint32_t b = 0; b < nbytes; b += 8) {
      uint64_t x = 0;
      memcpy(&x, cmap + b, std::min<uint32_t>(8u, nbytes - b));
      if (x && !emit(x, b * 4u)) break;
    }
  }
  size_t len = (size_t)(w - w0);
  if (len) len -= nm.n_delim;  // the last entry's delimiter goes
  o.resize(at + len);
}

const char *err_text(uint32_t code) {
  switch (code) {
    case BVCF_ERR_SAME: return "REF == ALT";
    case BVCF_ERR_BAD_ALT1:
    case BVCF_ERR_BAD_ALT: return "ALT not ACTG";
    case BVCF_ERR_DEL1_1:
    case BVCF_ERR_DEL1: return "1st base REF != ALT";
    case BVCF_ERR_POS1:
    case BVCF_ERR_POS: return "Invalid POS";
    case BVCF_ERR_INS1: return "1st base ALT != REF";
    case BVCF_ERR_MIXED: return "Mixed indel/snp sites not supported";
    case BVCF_ERR_EMPTY_REF: return "empty REF";
  }
  return "?";
}

// one log line in the reference's formats (main.go:730-986)
// where line li's bytes are: in the block the batch was submitted as, or -- bvcf_submit_bgzf with head_off -- in the
// compact copy of the line heads that came back
inline const char *row_of(const bvcf_result *r, const uint8_t *block, uint32_t li) {
  return (const char *)block + (r->head_off ? r->head_off[li] : r->lines[li].off);
}

void append_err(std::string &log, const bvcf_err &e, const bvcf_line &L, const char *row) {
  log.append(row, L.fend[0]);  // chrom
  log.push_back(':');
  log.append(row + L.fend[0] + 1, L.fend[1] - L.fend[0] - 1);  // pos
  char tmp[64];
  switch (e.code) {
    case BVCF_ERR_SAME: log.append(" : "); break;
    case BVCF_ERR_BAD_ALT1:
    case BVCF_ERR_DEL1_1:
    case BVCF_ERR_POS1: log.append(" ALT #1 "); break;
    case BVCF_ERR_BAD_ALT:
    case BVCF_ERR_INS1: log.append(tmp, (size_t)snprintf(tmp, sizeof tmp, " ALT #%u ", e.alt_no)); break;
    case BVCF_ERR_DEL1:
    case BVCF_ERR_MIXED: log.append(tmp, (size_t)snprintf(tmp, sizeof tmp, " ALT#%u ", e.alt_no)); break;
    case BVCF_ERR_EMPTY_REF: log.append(e.alt_no == 1 ? " ALT #1 " : " "); break;
    default: log.push_back(' '); break;
  }
  log.append(err_text(e.code));
  log.push_back('\n');
}

// rows of lines [lo, hi), main.go:566-695
void format_lines(const bvcf_config *c, const bvcf_result *r, const uint8_t *block, const Names &nm,
                  const Ratios *rt, uint32_t lo, uint32_t hi, std::string &out) {
  const char *empty = or_default(c->empty_field, "!");
  const uint32_t ns = r->n_samples;
  const double num_samples = (double)ns;
  for (uint32_t li = lo; li < hi; li++) {
    const bvcf_line &L = r->lines[li];
    if (L.status != BVCF_LINE_OK) continue;
    const char *row = row_of(r, block, li);
    auto fstart = [&](int i) -> uint32_t { return i ? L.fend[i - 1] + 1 : 0; };
    for (uint32_t k = 0; k < L.n_rec; k++) {
      const uint32_t slot = k ? L.rec_first + k - 1 : li;
      const bvcf_allele &A = r->alleles[slot];
      // main.go:555-560: with samples, an allele nobody carries is skipped
      if (ns > 0 && A.ac == 0) continue;
      const bvcf_names *NL = r->name_lists ? &r->name_lists[slot] : nullptr;
      // main.go:570-574
      const uint32_t nchrom = L.fend[0];
      if (nchrom < 4 || row[0] != 'c') out.append("chr");
      out.append(row, nchrom);
      out.push_back('\t');
      if (A.flags & BVCF_ALLELE_POS_TEXT)
        out.append(row + fstart(1), L.fend[1] - fstart(1));
      else
        append_ll(out, A.pos);
      out.push_back('\t');
      out.append(kSiteNames[A.site_type < 5 ? A.site_type : 0]);
      out.push_back('\t');
      out.push_back((char)A.ref);
      out.push_back('\t');
      if (A.kind == BVCF_ALT_BASE) {
        out.push_back((char)A.alt_base);
      } else if (A.kind == BVCF_ALT_INS) {
        out.push_back('+');
        out.append(row + (A.alt_off - L.off), A.alt_len);  // (alt_off is a block offset inside the line's ALT column)
      } else {
        out.push_back('-');
        append_ll(out, A.alt_len);
      }
      out.push_back('\t');
      out.push_back((char)('0' + A.trtv));  // main.go:602-606
      out.push_back('\t');

      const double effective = num_samples - (double)A.n_miss;  // main.go:563
      const uint8_t *cm = (A.cmap_off != BVCF_NO_CMAP && r->cmap) ? r->cmap + A.cmap_off : nullptr;
      const uint64_t n_eff = ns >= A.n_miss ? ns - A.n_miss : 0;
      struct {
        uint32_t n;
        unsigned cls;
        double denom;
        uint64_t den;
      } lists[3] = {{A.n_het, BVCF_CLS_HET, effective, n_eff}, {A.n_hom, BVCF_CLS_HOM, effective, n_eff},
                    {A.n_miss, BVCF_CLS_MISSING, num_samples, ns}};
      for (int q = 0; q < 3; q++) {  // main.go:612-656
        if (lists[q].n == 0 || !cm) {
          out.append(empty);
          out.append("\t0");
        } else {
          if (NL)  // rendered on the device (bvcf_params.want_name_lists): one copy per list
            out.append(r->names + NL->off[q], NL->len[q]);
          else
            join_class(out, cm, (A.flags & BVCF_ALLELE_CMAP_SPARSE) != 0, ns, lists[q].cls, lists[q].n, nm);
          out.push_back('\t');
          if (rt)
            rt->append(out, lists[q].n, lists[q].denom, lists[q].den);
          else
            append_g3(out, (double)lists[q].n / lists[q].denom);
        }
        out.push_back('\t');
      }
      append_ll(out, A.ac);  // main.go:661-671
      out.push_back('\t');
      append_ll(out, A.an);
      out.push_back('\t');
      if (A.ac == 0)
        out.push_back('0');
      else if (rt)
        rt->append(out, A.ac, (double)A.an, A.an);
      else
        append_g3(out, (double)A.ac / (double)A.an);
      if (c->keep_pos) {  // main.go:674-692
        out.push_back('\t');
        out.append(row + fstart(1), L.fend[1] - fstart(1));
      }
      if (c->keep_id) {
        out.push_back('\t');
        out.append(row + fstart(2), L.fend[2] - fstart(2));
      }
      if (c->keep_info) {
        out.push_back('\t');
        append_ll(out, A.alt_idx);
        out.push_back('\t');
        out.append(row + fstart(7), L.fend[7] - fstart(7));
      }
      out.push_back('\n');
    }
  }
}

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

// the batch's log lines in input order (stable: one line's messages keep their ALT order)
void format_log(const bvcf_result *r, const uint8_t *block, std::string &log) {
  if (!r->n_errs) return;
  std::vector<uint32_t> idx(r->n_errs);
  for (uint32_t i = 0; i < r->n_errs; i++) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return r->errs[x].line < r->errs[y].line; });
  for (uint32_t i : idx) append_err(log, r->errs[i], r->lines[r->errs[i].line], row_of(r, block, r->errs[i].line));
}

// rows of one batch as consecutive pieces (parts[0] + parts[1] + ... is the batch's TSV): runs of lines are claimed
// by the pool's threads, a few per thread so that lines with long sample lists do not leave the others idle
void format_parts(const bvcf_config *c, const bvcf_result *r, const uint8_t *block, const Names &nm, const Ratios *rt,
                  WorkPool *pool, std::vector<std::string> &parts) {
  const unsigned nt = pool ? pool->size() : 1;
  uint32_t n_parts = 1;
  if (nt > 1 && r->n_lines >= 4 * nt) n_parts = std::min<uint32_t>(4 * nt, r->n_lines / 32u);
  if (n_parts < 1) n_parts = 1;
  if (parts.size() < n_parts) parts.resize(n_parts);
  for (auto &p : parts) p.clear();  // keeps the capacity of a recycled vector
  auto one = [&](uint32_t t) {
    const uint32_t lo = (uint32_t)((uint64_t)r->n_lines * t / n_parts);
    const uint32_t hi = (uint32_t)((uint64_t)r->n_lines * (t + 1) / n_parts);
    format_lines(c, r, block, nm, rt, lo, hi, parts[t]);
  };
  if (n_parts == 1)
    one(0);
  else
    pool->run(n_parts, one);
}

void format_batch(const bvcf_config *c, const bvcf_result *r, const uint8_t *block, const Names &nm, const Ratios *rt,
                  WorkPool *pool, std::string &out, std::string &log) {
  format_log(r, block, log);
  std::vector<std::string> parts;
  format_parts(c, r, block, nm, rt, pool, parts);
  for (auto &p : parts) out.append(p);
}

char *dup_out(const std::string &s, size_t *n) {
  char *p = (char *)malloc(s.size() + 1);
  if (!p) return nullptr;
  memcpy(p, s.data(), s.size());
  p[s.size()] = 0;
  *n = s.size();
  return p;
}

// ---- readVcf's preamble, main.go:250-304

struct Preamble {
  uint8_t eol_byte = '\n';
  uint32_t eol_chars = 1;
  std::vector<std::string> header;  // normalised
  size_t data_off = 0;              // first byte after the #CHROM line
};

// returns 0, 1 = need more input, <0 = fatal (message in *msg)
int parse_preamble(const uint8_t *in, size_t n, bool at_eof, bool normalize, Preamble *pre, std::string *msg) {
  // parse.FindEndOfLine(reader, ""): consume line 1, learn the terminator ("\r\n"/"\r": unpinned)
  size_t i = 0;
  for (;; i++) {
    if (i >= n) {
      if (!at_eof) return 1;
      *msg = "EOF";
      return -1;
    }
    if (in[i] == '\n') break;
    if (in[i] == '\r') {
      if (i + 1 >= n) {
        if (!at_eof) return 1;
        *msg = "EOF";
        return -1;
      }
      if (in[i + 1] == '\n') {
        pre->eol_chars = 2;
      } else {
        pre->eol_byte = '\r';
      }
      break;
    }
  }
  // main.go:256-264
  if (!memmem(in, i, "##fileformat=VCFv4", 18)) {
    *msg = "Not a VCF file";
    return -1;
  }
  size_t pos = i + pre->eol_chars;
  // main.go:266-294
  while (pos < n) {
    const uint8_t *e = (const uint8_t *)memchr(in + pos, pre->eol_byte, n - pos);
    if (!e) break;
    const size_t row_len = (size_t)(e - (in + pos)) + 1;
    const uint8_t *row = in + pos;
    pos += row_len;
    if (row_len < pre->eol_chars) continue;
    const size_t body = row_len - pre->eol_chars;
    const uint8_t *tab = (const uint8_t *)memchr(row, '\t', body);
    const size_t f0 = tab ? (size_t)(tab - row) : body;
    if (f0 == 6 && memcmp(row, "#CHROM", 6) == 0) {
      size_t s = 0;
      for (size_t k = 0; k <= body; k++) {
        if (k != body && row[k] != '\t') continue;
        std::string f((const char *)row + s, k - s);
        // parse.NormalizeHeader, main.go:296 (restated: '.' -> '_'; parity unpinned)
        if (normalize) std::replace(f.begin(), f.end(), '.', '_');
        pre->header.push_back(std::move(f));
        s = k + 1;
      }
      pre->data_off = pos;
      return 0;
    }
  }
  if (!at_eof) return 1;
  *msg = "No header found";
  return -1;
}

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

// Which device path suits this file: the streaming path reads the text once -- the bare 4-byte "x|y<TAB>" fields of
// a FORMAT == GT file (1000-Genomes style) through its regular scan, fields with further sub-fields through its
// general stream -- as long as a line's class map fits the LDS stage (16 384 samples); beyond that, lines that are
// not regular would all be left to k_gt, for which the census path is the better frame.
uint32_t choose_path(const Run &R, const uint8_t *data, size_t n) {
  if (R.pre.header.size() < 256) return 0;  // the library's own rule (census for narrow files)
  if (R.pre.header.size() >= 9 + (size_t)BVCF_WIDE_SAMPLES) return 0;  // very wide lines: the census path's split scan
  if (R.pre.header.size() <= 9 + 16384u) return 2;
  // FORMAT column (index 8) of the first record
  size_t pos = 0;
  for (int tabs = 0; pos < n && tabs < 8; pos++) {
    if (data[pos] == R.pre.eol_byte) return 0;
    tabs += data[pos] == '\t';
  }
  size_t e = pos;
  while (e < n && data[e] != '\t' && data[e] != R.pre.eol_byte) e++;
  return (e - pos == 2 && data[pos] == 'G' && data[pos + 1] == 'T') ? 2u : 1u;
}

// writeSampleListIfWanted + makeSampleList, main.go:398-445: header fields 9.. one per line; the file is
// opened O_WRONLY|O_CREATE (no truncation), and stays empty when the header has fewer than 10 fields
int write_sample_list(const Run &R) {
  const char *path = R.cfg->sample_list_path;
  if (!path || !*path) return 0;
  int fd = open(path, O_WRONLY | O_CREAT, 0644);
  if (fd < 0) return -1;
  std::string s;
  if (R.pre.header.size() >= 10)
    for (size_t i = 9; i < R.pre.header.size(); i++) {
      s.append(R.pre.header[i]);
      s.push_back('\n');
    }
  size_t off = 0;
  while (off < s.size()) {
    ssize_t w = write(fd, s.data() + off, s.size() - off);
    if (w < 0) {
      if (errno == EINTR) continue;
      close(fd);
      return -1;
    }
    off += (size_t)w;
  }
  fsync(fd);
  return close(fd);
}

// What every ctx of the run shares: the sample list file, the ctx parameters (R.params), the name arena, the ratio
// strings, the formatter's worker pool, the dosage file.  Once per run, after the header is known.
int prepare_run(Run &R, std::string *msg, const uint8_t *data = nullptr, size_t n_data = 0) {
  if (R.pre.header.size() < 8) {
    // the reference indexes record[6] / record[7] unguarded: out of contract
    *msg = "Malformed header: fewer than 8 fields";
    return BVCF_E_FATAL;
  }
  if (write_sample_list(R)) {  // main.go:298-304
    *msg = "Couldn't write sample list file";
    return BVCF_E_FATAL;
  }
  bvcf_params &p = R.params;
  memset(&p, 0, sizeof p);
  p.abi_version = BVCF_ABI_VERSION;
  p.device = R.cfg->device;
  p.n_header_fields = (uint32_t)R.pre.header.size();
  p.eol_chars = R.pre.eol_chars;
  p.eol_byte = R.pre.eol_byte;
  R.want_rows = !R.cfg->no_out;
  p.want_class_maps = R.want_rows;  // needsLabels, main.go:502
  // BVCF_DEVICE_NAMES=1: the sample-name lists of the rows come off the device as text instead of being
  // joined by the formatter from the class maps.  Off by default: measured on the dense profile (every row a common
  // variant, 10 KB of names per row) the text is 16 x the class maps over PCIe and the run gets slower, while the
  // formatter's worker pool is not what a one-GPU run waits for (profiles/r02_e2e_cli_dense_*.log).
  {
    const char *e = getenv("BVCF_DEVICE_NAMES");
    p.want_name_lists = R.want_rows && R.pre.header.size() > 9 && e && *e == '1' &&
                        strlen(or_default(R.cfg->field_delimiter, ";")) <= 16;
  }
  p.want_dosage = R.cfg->dosage_path && *R.cfg->dosage_path && R.pre.header.size() > 9;
  p.allow_filter = R.cfg->allow_filter;
  p.exclude_filter = R.cfg->exclude_filter;
  p.max_batch_bytes = R.max_batch;
  p.n_slots = R.n_slots;
  // The library sizes its result arrays for the shortest line that could pass (48 bytes for a sites-only file:
  // 1.4 M lines per 64 MiB batch, a gigabyte of pinned result memory over three slots).  The first block says how
  // long the lines of this file are: reserve for lines half that long; a batch that needs more grows the
  // reservation (BVCF_E_CAPACITY, bvcf_reserve).
  if (data && n_data) {
    const size_t look = std::min<size_t>(n_data, 4u << 20);
    size_t n_eol = 0;
    for (const uint8_t *q = data, *e = data + look; (q = (const uint8_t *)memchr(q, R.pre.eol_byte, (size_t)(e - q))); q++) n_eol++;
    if (n_eol >= 16) {
      const uint64_t avg = look / n_eol;
      const uint64_t floor_len = std::max<uint64_t>(48, 2ull * R.pre.header.size());  // the library's own bound
      const uint64_t per_line = std::max<uint64_t>(floor_len, avg / 2);
      // (the slack for short lines between the records, as the library computes it: what 32 MiB of class maps hold)
      const uint64_t ns = R.pre.header.size() > 9 ? R.pre.header.size() - 9 : 0;
      const uint64_t stride = std::max<uint64_t>(16, ((ns + 3) / 4 + 15) & ~15ull);
      const uint64_t slack = std::min<uint64_t>(4096, std::max<uint64_t>(64, (32ull << 20) / stride));
      p.max_lines = (uint32_t)std::min<uint64_t>(R.max_batch / per_line + slack, 0x7FFFFFFFu);
    }
  }
  p.path = data ? choose_path(R, data, n_data) : 0;
  for (size_t i = 9; i < R.pre.header.size(); i++) {
    R.name_ptr.push_back(R.pre.header[i].data());
    R.name_len.push_back((uint32_t)R.pre.header[i].size());
  }
  R.names.reset(new Names(R.name_ptr.data(), R.name_len.data(), R.name_ptr.size(), or_default(R.cfg->field_delimiter, ";")));
  R.ratios.reset(new Ratios((uint32_t)R.name_ptr.size()));
  R.n_threads = R.cfg->n_format_threads ? R.cfg->n_format_threads
                                         : std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
  if (R.want_rows && R.n_threads > 1) R.pool.reset(new WorkPool(R.n_threads));
  if (R.cfg->dosage_path && *R.cfg->dosage_path) {  // main.go:306-342
    if (R.pre.header.size() <= 9) {
      // "No samples found in VCF file; writing empty dosage matrix file"
      FILE *f = fopen(R.cfg->dosage_path, "wb");
      if (!f) {
        *msg = std::string("open ") + R.cfg->dosage_path + ": " + strerror(errno);
        return BVCF_E_FATAL;
      }
      fclose(f);
    } else if (bvcf_arrow_open(&R.arrow, R.cfg->dosage_path, R.name_ptr.data(), R.name_len.data(),
                               (uint32_t)R.name_ptr.size(), 0, 0) != BVCF_OK) {
      *msg = std::string("open ") + R.cfg->dosage_path + ": " + strerror(errno);
      return BVCF_E_FATAL;
    }
  }
  return BVCF_OK;
}

// one ctx of the run on `device` (the counterpart of one `go processLines(...)`, main.go:345-347)
int create_ctx(const Run &R, int device, bvcf_ctx **ctx, std::string *msg) {
  bvcf_params p = R.params;
  p.device = device;
  int rc = bvcf_create(ctx, &p);
  if (rc) {
    *msg = std::string("bvcf_create: ") + bvcf_last_error(nullptr);
    return rc;
  }
  if (p.want_name_lists) {
    rc = bvcf_set_sample_names(*ctx, R.name_ptr.data(), R.name_len.data(), (uint32_t)R.name_ptr.size(),
                               or_default(R.cfg->field_delimiter, ";"));
    if (rc) {
      *msg = std::string("bvcf_set_sample_names: ") + bvcf_last_error(*ctx);
      bvcf_destroy(*ctx);
      *ctx = nullptr;
    }
  }
  return rc;
}

int open_ctx(Run &R, std::string *msg, const uint8_t *data = nullptr, size_t n_data = 0) {
  int rc = prepare_run(R, msg, data, n_data);
  if (rc == BVCF_OK) rc = create_ctx(R, R.cfg->device, &R.ctx, msg);
  return rc;
}

// the Arrow rows of one collected batch, in input order (main.go:576-584): "chrom:pos:ref:alt" + one int8 per sample
int append_dosage(Run &R, const bvcf_result *r, const uint8_t *block) {
  if (!R.arrow || !r->dosage) return BVCF_OK;
  std::string locus;
  for (uint32_t li = 0; li < r->n_lines; li++) {
    const bvcf_line &L = r->lines[li];
    if (L.status != BVCF_LINE_OK) continue;
    const char *row = row_of(r, block, li);
    for (uint32_t k = 0; k < L.n_rec; k++) {
      const uint32_t slot = k ? L.rec_first + k - 1 : li;
      const bvcf_allele &A = r->alleles[slot];
      if (A.ac == 0) continue;  // main.go:558-560
      locus.clear();
      if (L.fend[0] < 4 || row[0] != 'c') locus.append("chr");
      locus.append(row, L.fend[0]);
      locus.push_back(':');
      if (A.flags & BVCF_ALLELE_POS_TEXT)
        locus.append(row + L.fend[0] + 1, L.fend[1] - L.fend[0] - 1);
      else
        append_ll(locus, A.pos);
      locus.push_back(':');
      locus.push_back((char)A.ref);
      locus.push_back(':');
      if (A.kind == BVCF_ALT_BASE) {
        locus.push_back((char)A.alt_base);
      } else if (A.kind == BVCF_ALT_INS) {
        locus.push_back('+');
        locus.append(row + (A.alt_off - L.off), A.alt_len);
      } else {
        locus.push_back('-');
        append_ll(locus, A.alt_len);
      }
      if (bvcf_arrow_append(R.arrow, locus.data(), (uint32_t)locus.size(), r->dosage + (size_t)slot * r->dosage_stride))
        return BVCF_E_FATAL;
    }
  }
  return BVCF_OK;
}

int close_dosage(Run &R) {
  if (!R.arrow) return BVCF_OK;
  const int rc = bvcf_arrow_close(R.arrow);
  R.arrow = nullptr;
  return rc;
}

// submit one block and collect it, growing the result reservation when the batch asks for it
int process_block(Run &R, const uint8_t *block, size_t n, uint64_t seq, bvcf_result *res, std::string *msg) {
  for (int attempt = 0; attempt < 4; attempt++) {
    int rc = bvcf_submit(R.ctx, block, n, seq);
    if (rc) {
      *msg = std::string("bvcf_submit: ") + bvcf_last_error(R.ctx);
      return rc;
    }
    rc = bvcf_collect(R.ctx, res);
    if (rc == BVCF_OK) return rc;
    if (rc != BVCF_E_CAPACITY) {
      *msg = std::string("bvcf_collect: ") + bvcf_last_error(R.ctx);
      return rc;
    }
    rc = bvcf_reserve(R.ctx, res->need_lines + res->need_lines / 4 + 64, res->need_alleles + res->need_alleles / 4 + 64,
                      res->need_cmap_bytes + res->need_cmap_bytes / 4 + 4096);
    if (rc) {
      *msg = std::string("bvcf_reserve: ") + bvcf_last_error(R.ctx);
      return rc;
    }
  }
  *msg = "result reservation did not converge";
  return BVCF_E_CAPACITY;
}

}  // namespace

namespace {

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

 private:
  size_t cap_;
  std::deque<T> q_;
  std::mutex mu_;
  std::condition_variable not_full_, not_empty_;
};

// one block of whole lines in a pinned buffer
struct Block {
  uint8_t *buf = nullptr;
  size_t start = 0, nb = 0;  // lines live in buf[start, start + nb)
  size_t fill = 0;           // bytes read into buf (preamble parsing needs this on the first block)
  bool first = false, last = false, too_long = false, read_error = false;
  uint64_t seq = 0;          // block number: the order of the output
  // BGZF input inflated on the device: buf[0, nb) holds whole compressed blocks, the first `own` bytes of them the
  // batch's own, the rest look-ahead (bvcf_submit_bgzf)
  bool bgzf = false, skip_first = false;
  size_t own = 0;
  uint32_t first_off = 0;
  // ... and what the reader learnt from the blocks it inflated itself to get at the header
  struct Pre {
    Preamble pre;
    std::vector<uint8_t> sample;  // the first data lines, for prepare_run
  };
  std::shared_ptr<Pre> pre;
};

}  // namespace


extern "C" {

void bvcf_config_defaults(bvcf_config *c) {
  memset(c, 0, sizeof *c);
  c->empty_field = "!";
  c->field_delimiter = ";";
  c->allow_filter = "PASS,.";
  c->exclude_filter = "";
  c->normalize_header = 1;
}

size_t bvcf_string_header(const bvcf_config *c, char *out, size_t cap) {
  std::string h;
  for (int i = 0; i < 15; i++) {
    if (i) h.push_back('\t');
    h.append(kBaseHeader[i]);
  }
  if (c->keep_pos) h.append("\tvcfPos");
  if (c->keep_id) h.append("\tid");
  if (c->keep_info) h.append("\talleleIdx\tinfo");
  if (out && cap > h.size()) memcpy(out, h.c_str(), h.size() + 1);
  return h.size();
}

void bvcf_free(void *p) { free(p); }

int bvcf_decompress_fd(int fd_in, int fd_out, uint32_t n_threads, char *kind_out) {
  if (!n_threads) n_threads = std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
  bvcf_input::ByteSource src(fd_in, n_threads);
  std::vector<uint8_t> buf(8u << 20);
  for (;;) {
    ssize_t got = src.read(buf.data(), buf.size());
    if (got == 0) break;
    if (got < 0) {
      if (kind_out) snprintf(kind_out, 8, "%s", src.kind());
      return BVCF_E_FATAL;
    }
    size_t off = 0;
    while (off < (size_t)got) {
      ssize_t w = write(fd_out, buf.data() + off, (size_t)got - off);
      if (w < 0) {
        if (errno == EINTR) continue;
        return BVCF_E_FATAL;
      }
      off += (size_t)w;
    }
  }
  if (kind_out) snprintf(kind_out, 8, "%s", src.kind());
  return BVCF_OK;
}

int bvcf_format_tsv(const bvcf_config *c, const bvcf_result *r, const uint8_t *block, const char *const *sample_names,
                    const uint32_t *sample_name_lens, char **out, size_t *n_out, char **log, size_t *n_log) {
  if (!c || !r || !out || !n_out) return BVCF_E_ARG;
  if (r->n_samples && (!sample_names || !sample_name_lens)) return BVCF_E_ARG;
  std::string o, l;
  Names nm(sample_names, sample_name_lens, r->n_samples, or_default(c->field_delimiter, ";"));
  const unsigned nt =
      c->n_format_threads ? c->n_format_threads : std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
  std::unique_ptr<WorkPool> pool;
  if (nt > 1 && r->n_lines >= 4 * nt) pool.reset(new WorkPool(nt));
  format_batch(c, r, block, nm, nullptr, pool.get(), o, l);
  *out = dup_out(o, n_out);
  if (log && n_log) *log = dup_out(l, n_log);
  return *out ? BVCF_OK : BVCF_E_NOMEM;
}

int bvcf_run_buffer(const bvcf_config *c, const uint8_t *vcf, size_t n, char **out, size_t *n_out, char **log,
                    size_t *n_log, uint64_t *n_lines_in) {
  if (!c || (!vcf && n) || !out || !n_out || !log || !n_log) return BVCF_E_ARG;
  std::string o, l, msg;
  Run R;
  R.cfg = c;
  R.max_batch = c->max_batch_bytes ? c->max_batch_bytes : (64ull << 20);
  uint64_t lines_in = 0;
  int rc = parse_preamble(vcf, n, true, c->normalize_header, &R.pre, &msg);
  if (rc < 0) {
    l = msg + "\n";
    rc = BVCF_E_FATAL;
  } else {
    rc = open_ctx(R, &msg, vcf + R.pre.data_off, n - R.pre.data_off);
    if (rc) l = msg + "\n";
  }
  if (rc == BVCF_OK) {
    if (R.pre.header.size() == 9)  // main.go:507-509
      l.append("Found 9 header fields. When genotypes present, we expect 1+ samples after FORMAT (10 fields minimum)\n");
    const Names &nm = *R.names;
    size_t pos = R.pre.data_off;
    uint64_t seq = 0;
    while (pos < n) {
      // whole lines only; an unterminated tail is dropped (main.go:354-358)
      size_t end = std::min<size_t>(n, pos + R.max_batch);
      const uint8_t *last = (const uint8_t *)memrchr(vcf + pos, R.pre.eol_byte, end - pos);
      if (!last) {
        if (end == n) break;
        msg = "a line is longer than max_batch_bytes";
        l.append(msg + "\n");
        rc = BVCF_E_TOO_BIG;
        break;
      }
      const size_t nb = (size_t)(last - (vcf + pos)) + 1;
      bvcf_result res;
      rc = process_block(R, vcf + pos, nb, seq++, &res, &msg);
      if (rc) {
        l.append(msg + "\n");
        break;
      }
      lines_in += res.n_lines_seen;
      if (R.want_rows) {
        format_batch(c, &res, vcf + pos, nm, R.ratios.get(), R.pool.get(), o, l);
      } else {
        format_log(&res, vcf + pos, l);
      }
      if (append_dosage(R, &res, vcf + pos)) {
        l.append("dosage matrix: write failed\n");
        rc = BVCF_E_FATAL;
        break;
      }
      pos += nb;
    }
  }
  if (close_dosage(R) && rc == BVCF_OK) {
    l.append("dosage matrix: write failed\n");
    rc = BVCF_E_FATAL;
  }
  if (R.ctx) bvcf_destroy(R.ctx);
  if (n_lines_in) *n_lines_in = lines_in;
  *out = dup_out(o, n_out);
  *log = dup_out(l, n_log);
  return rc;
}

static int write_all(int fd, const char *p, size_t n) {
  while (n) {
    ssize_t w = write(fd, p, n);
    if (w < 0) {
      if (errno == EINTR) continue;
      return -1;
    }
    p += w;
    n -= (size_t)w;
  }
  return 0;
}

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// The reference's main() + readVcf (main.go:134-217, 241-396) as a pipeline:
//   reader thread      fd -> pinned buffers, cut at the last terminator                 (main.go:349-380)
//   this thread        preamble, then deals block k to device worker k % N              (workQueue <- buff, main.go:366)
//   N device workers   one ctx each: bvcf_submit one block ahead, bvcf_collect the oldest
//                                                                  (the goroutines of main.go:345-347)
//   format thread      takes the collected batches in block order, TSV assembly on the worker pool  (main.go:566-695)
//   writer thread      ordered write to fd_out                                          (main.go:524-532,705-711)
// A collected batch's result arrays stay valid until its slot is collected into again, n_slots batches later on the
// same ctx, so formatting runs one or two batches behind the devices instead of between two submits.  The output is
// the same bytes for any device list: blocks are cut by the reader alone and merged by block number.
int bvcf_run_fd(const bvcf_config *c, int fd_in, int fd_out, int fd_err, uint64_t *n_lines_in) {
  if (!c) return BVCF_E_ARG;
  const char *timing_env = getenv("BVCF_TIMING");
  const bool timing = timing_env != nullptr;
  const bool timing_json = timing && strcmp(timing_env, "json") == 0;
  const double t_start = now_s();
  double t_wait_read = 0, t_fmt = 0, t_init = 0, t_prepare = 0, t_deal = 0;
  std::string msg;
  Run R;
  R.cfg = c;
  R.n_slots = 3;  // two batches on the device, one more being formatted
  R.max_batch = c->max_batch_bytes ? c->max_batch_bytes : (64ull << 20);
  std::atomic<uint64_t> lines_in{0};

  // the devices of the run
  std::vector<int> dev_list;
  if (c->n_devices && c->devices)
    dev_list.assign(c->devices, c->devices + c->n_devices);
  else
    dev_list.push_back(c->device);
  const size_t n_dev = dev_list.size();

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

  const size_t cap = R.max_batch;
  // being read into, two (BGZF inflated on the device: three) on each device, up to two with the formatter, one spare
  const int kBufs = (int)std::min<size_t>(3 * n_dev + 4, 64);
  // Pinning memory costs about 25 ms per 64 MiB: only the first buffer is allocated before the reader starts, the
  // others follow in the background while the first block is read and the ctx is created, and stop at end of input
  // (a small file never pays for all of them).
  std::vector<uint8_t *> bufs((size_t)kBufs, nullptr);
  bufs[0] = (uint8_t *)bvcf_alloc_pinned(cap);
  const double t_pinned = now_s() - t_start;
  if (!bufs[0]) {
    // no device => no pinned memory either; fail loudly, there is no CPU path
    dprintf(fd_err, "cannot allocate pinned host memory (no usable HIP device?)\n");
    return BVCF_E_NODEV;
  }

  Channel<uint8_t *> free_q(256);
  Channel<Block> ready_q((size_t)kBufs);
  typedef std::vector<std::string> Parts;
  Channel<Parts *> write_q(4);
  free_q.push(bufs[0]);
  std::atomic<bool> stop{false};
  std::atomic<bool> stop_alloc{false}, alloc_failed{false};
  // BGZF input inflated on the device fills a buffer with compressed bytes only: the buffers after the first are then
  // a quarter of the size (pinning is what they cost), and batches are cut to fit them.  0 = not known yet.
  std::atomic<size_t> later_buf_bytes{0};
  std::thread allocator([&]() {
    while (!later_buf_bytes.load() && !stop_alloc.load()) std::this_thread::sleep_for(std::chrono::microseconds(200));
    const size_t bytes = later_buf_bytes.load() ? later_buf_bytes.load() : cap;
    for (int i = 1; i < kBufs && !stop_alloc.load(); i++) {
      bufs[i] = (uint8_t *)bvcf_alloc_pinned(bytes);
      if (!bufs[i]) {
        // the pipeline needs three buffers to make progress: end the run (reported below) rather than stall
        alloc_failed.store(true);
        stop.store(true);
        free_q.push(nullptr);
        return;
      }
      free_q.push(bufs[i]);
    }
  });
  auto free_bufs = [&]() {
    std::vector<std::thread> th;
    for (int i = 0; i < kBufs; i++)
      if (bufs[i]) th.emplace_back([&, i]() { bvcf_free_pinned(bufs[i]); });
    for (auto &t : th) t.join();
  };
  std::atomic<uint8_t> eol_byte{'\n'};

  std::string source_err;
  std::atomic<bool> input_is_bgzf_device{false};
  std::atomic<size_t> max_in_flight{2};  // batches a device worker keeps submitted
  // ---- reader, BGZF on the device: whole compressed blocks per buffer.  The header has to be read here, so the
  // leading blocks are inflated with zlib until the #CHROM line is complete; everything from the block that holds the
  // first data line on is handed over compressed, each buffer with the next buffer's first blocks as look-ahead.
  auto read_bgzf_raw = [&](bvcf_input::ByteSource &src) {
    input_is_bgzf_device.store(true);
    std::vector<uint8_t> pend;  // compressed bytes read from the input; pend[pp..] not yet handed over
    size_t pp = 0;
    bool raw_eof = false;
    auto fail_read = [&](const std::string &m) {
      if (!later_buf_bytes.load()) later_buf_bytes.store(cap);
      source_err = m;
      Block b;
      b.read_error = true;
      b.last = true;
      ready_q.push(b);
    };
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
    // the whole block at pend[pp + off]: its size; 0 at a clean end of input; -1 malformed / truncated / read error
    auto block_at = [&](size_t off, bvcf_bgzf::Block *b) -> long {
      for (;;) {
        const size_t have = pend.size() - pp - off;
        uint32_t xlen = 0;
        const long bs = have ? bvcf_bgzf::block_size(pend.data() + pp + off, have, &xlen) : 0;
        if (bs < 0) return -1;
        if (bs > 0 && (size_t)bs <= have) {
          std::vector<bvcf_bgzf::Block> one;
          if (bvcf_bgzf::scan(pend.data() + pp + off, (size_t)bs, &one) != bs || one.size() != 1) return -1;
          *b = one[0];
          return bs;
        }
        if (!more()) return (!source_err.empty() || have) ? -1 : 0;
      }
    };
    // ---- the header, from blocks inflated here
    auto pre = std::make_shared<Block::Pre>();
    std::vector<uint8_t> htext;
    std::vector<std::pair<size_t, size_t>> marks;  // (compressed offset from pp, text offset) of each inflated block
    size_t hoff = 0;
    std::string msg;
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return fail_read("inflateInit2 failed");
    auto inflate_next = [&]() -> int {  // 1 = a block was inflated, 0 = end of input, -1 = bad
      bvcf_bgzf::Block b;
      const long bs = block_at(hoff, &b);
      if (bs <= 0) return (int)bs;
      marks.emplace_back(hoff, htext.size());
      const size_t at = htext.size();
      htext.resize(at + b.isize);
      inflateReset(&zs);
      zs.next_in = pend.data() + pp + hoff + b.in_off;
      zs.avail_in = b.in_len;
      zs.next_out = htext.data() + at;
      zs.avail_out = b.isize;
      const int zr = b.isize ? inflate(&zs, Z_FINISH) : Z_STREAM_END;
      if ((b.isize && (zr != Z_STREAM_END || zs.avail_out != 0)) ||
          (uint32_t)crc32(crc32(0L, Z_NULL, 0), htext.data() + at, b.isize) != b.crc)
        return -1;
      hoff += (size_t)bs;
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
      pr = parse_preamble(htext.data(), htext.size(), hdr_eof, c->normalize_header, &pre->pre, &msg);
      if (hdr_eof) break;
    }
    if (pr != 0) {
      inflateEnd(&zs);
      return fail_read(pr < 0 ? msg : std::string("No header found"));
    }
    const size_t data_off = pre->pre.data_off;
    // a few data lines for prepare_run (path choice, reservation): make sure at least one whole line is in view
    for (int extra = 0; extra < 8; extra++) {
      if (memchr(htext.data() + data_off, pre->pre.eol_byte, htext.size() - data_off)) break;
      if (inflate_next() != 1) break;
    }
    inflateEnd(&zs);
    pre->sample.assign(htext.begin() + (ptrdiff_t)data_off, htext.end());
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
    // look-ahead: the longest line must end within it.  Twice the first line, or 16 bytes per column.
    size_t line_len = 16 * pre->pre.header.size() + 4096;
    if (const uint8_t *e = (const uint8_t *)memchr(pre->sample.data(), pre->pre.eol_byte, pre->sample.size()))
      line_len = std::max<size_t>(line_len, 2 * (size_t)(e - pre->sample.data()));
    const size_t look = std::min<size_t>(line_len / 65280 + 2, std::max<size_t>(2, cap / (4 * 65536)));
    const size_t text_limit = cap > (look + 1) * 65536 ? cap - (look + 1) * 65536 : cap / 2;
    // the compressed bytes of a batch (own + look-ahead blocks) must fit the smaller buffers
    const size_t small = std::min<size_t>(cap, std::max<size_t>(cap / 4, (2 * look + 8) * 66000));
    later_buf_bytes.store(small);
    const size_t comp_limit = small - (look + 1) * 66000;

    bool first = true;
    for (;;) {
      if (stop.load()) break;
      Block b;
      b.buf = free_q.pop();
      if (!b.buf) break;
      b.bgzf = true;
      b.first = first;
      b.skip_first = !first;
      b.first_off = first ? first_off : 0;
      if (first) b.pre = pre;
      size_t off = 0, own_text = 0;
      bool bad = false;
      for (;;) {
        bvcf_bgzf::Block k;
        const long bs = block_at(off, &k);
        if (bs < 0) bad = true;
        if (bs <= 0) break;
        if (off && (own_text + k.isize > text_limit || off + (size_t)bs > comp_limit)) break;
        off += (size_t)bs;
        own_text += k.isize;
      }
      size_t la = 0, la_text = 0;
      for (size_t n = 0; n < look && !bad; n++) {
        bvcf_bgzf::Block k;
        const long bs = block_at(off + la, &k);
        if (bs < 0) bad = true;
        if (bs <= 0) break;
        if (own_text + la_text + k.isize > cap || off + la + (size_t)bs > small) break;
        la += (size_t)bs;
        la_text += k.isize;
      }
      if (bad) {
        free_q.push(b.buf);
        return fail_read(source_err.empty() ? std::string("bgzf: not a BGZF block, or a truncated file") : source_err);
      }
      if (off) memcpy(b.buf, pend.data() + pp, off + la);
      b.nb = off + la;
      b.own = off;
      pp += off;
      // the stream ends with this buffer if nothing follows its own blocks
      b.last = la == 0 && raw_eof && pend.size() == pp;
      if (b.last) stop_alloc.store(true);
      first = false;
      const bool last = b.last;
      ready_q.push(b);
      if (last) return;
    }
    Block end;
    end.last = true;
    ready_q.push(end);
  };

  // ---- reader: whole lines per block; the partial last line is carried into the next buffer
  std::thread reader([&]() {
    bvcf_input::ByteSource src(fd_in, std::min(32u, std::max(1u, std::thread::hardware_concurrency())));
    {
      // BGZF input (bgzip / htslib .vcf.gz): the blocks go to the device compressed and are inflated there
      // (bvcf_submit_bgzf) unless BVCF_DEVICE_INFLATE=0 (then this thread's workers inflate them with zlib)
      const char *e = getenv("BVCF_DEVICE_INFLATE");
      if (!(e && *e == '0') && src.sniff_bgzf()) {
        read_bgzf_raw(src);
        return;
      }
      later_buf_bytes.store(cap);
    }
    std::vector<uint8_t> carry;
    bool first = true, eof = false;
    while (!eof && !stop.load()) {
      Block b;
      b.buf = free_q.pop();
      if (!b.buf) break;
      b.first = first;
      size_t fill = carry.size();
      if (fill) memcpy(b.buf, carry.data(), fill);
      carry.clear();
      while (!eof && fill < cap) {
        ssize_t got = src.read(b.buf + fill, cap - fill);
        if (got == bvcf_input::ByteSource::kNoRoom) break;  // this buffer is as full as it gets
        if (got < 0) {
          source_err = src.error();
          b.read_error = true;
          eof = true;
          break;
        }
        if (got == 0) {
          eof = true;
          break;
        }
        fill += (size_t)got;
      }
      b.fill = fill;
      b.last = eof;
      if (eof) stop_alloc.store(true);
      if (first) {
        // the terminator is learnt from line 1 (parse.FindEndOfLine, main.go:250)
        uint8_t e = '\n';
        for (size_t i = 0; i < fill; i++) {
          if (b.buf[i] == '\n') break;
          if (b.buf[i] == '\r') {
            if (i + 1 < fill && b.buf[i + 1] != '\n') e = '\r';
            break;
          }
        }
        eol_byte.store(e);
        first = false;
      }
      const uint8_t *lastp = fill ? (const uint8_t *)memrchr(b.buf, eol_byte.load(), fill) : nullptr;
      if (!lastp) {
        if (!eof && fill == cap) b.too_long = true;
        b.nb = 0;  // at EOF an unterminated tail is dropped (main.go:354-358)
      } else {
        b.nb = (size_t)(lastp - b.buf) + 1;
        if (!eof) carry.assign(b.buf + b.nb, b.buf + fill);
      }
      const bool fatal_block = b.too_long || b.read_error;
      ready_q.push(b);
      if (fatal_block) break;
    }
    if (!eof || stop.load()) {
      Block end;
      end.last = true;
      ready_q.push(end);
    }
  });

  // ---- writer; written-out part vectors go back to the formatter with their capacity
  std::atomic<bool> write_failed{false};
  std::mutex spare_mu;
  std::vector<Parts *> spares;
  double t_last_write = 0;
  std::thread writer([&]() {
    for (;;) {
      Parts *ps = write_q.pop();
      if (!ps) break;
      for (const std::string &s : *ps)
        if (!s.empty() && !write_failed.load() && write_all(fd_out, s.data(), s.size())) write_failed.store(true);
      t_last_write = now_s();
      std::lock_guard<std::mutex> lk(spare_mu);
      if (spares.size() < 4)
        spares.push_back(ps);
      else
        delete ps;
    }
  });

  // ---- first error wins; everything then drains
  std::mutex fail_mu;
  int rc = BVCF_OK;
  std::string log;
  std::atomic<bool> failed{false};
  auto fail = [&](const std::string &m, int code) {
    std::lock_guard<std::mutex> lk(fail_mu);
    if (rc == BVCF_OK) {
      rc = code;
      log.append(m + "\n");
    }
    failed.store(true);
  };

  // ---- device workers -> formatter: collected batches, taken in block order
  struct FmtJob {
    Block b;
    bvcf_result res;
    uint32_t worker = 0;
  };
  struct Reorder {
    std::mutex mu;
    std::condition_variable cv;
    std::map<uint64_t, FmtJob> held;
    uint64_t next = 0;
    bool closed = false;
    void put(uint64_t seq, const FmtJob &j) {
      {
        std::lock_guard<std::mutex> lk(mu);
        held.emplace(seq, j);
      }
      cv.notify_all();
    }
    // the job of block `next`; false once closed and that block is not coming
    bool take(FmtJob *j) {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return closed || held.count(next); });
      auto it = held.find(next);
      if (it == held.end()) return false;
      *j = it->second;
      held.erase(it);
      next++;
      return true;
    }
    void close() {
      {
        std::lock_guard<std::mutex> lk(mu);
        closed = true;
      }
      cv.notify_all();
    }
  } reorder;

  struct DevWorker {
    uint32_t idx = 0;
    int device = 0;
    bvcf_ctx *ctx = nullptr;
    Channel<Block> q{3};  // blocks dealt to this device; buf == nullptr ends the worker
    std::thread th;
    bool started = false;
    // slots: jobs of this worker the formatter has finished
    std::mutex mu;
    std::condition_variable cv;
    uint64_t fmt_done = 0;
    // timing
    double t_ctx = 0, t_submit = 0, t_gpu = 0, t_fmt_wait = 0, t_first_submit = 0;
    uint64_t n_blocks = 0, n_bytes = 0;
  };
  std::vector<std::unique_ptr<DevWorker>> workers;
  for (size_t d = 0; d < n_dev; d++) {
    workers.emplace_back(new DevWorker());
    workers.back()->idx = (uint32_t)d;
    workers.back()->device = dev_list[d];
  }
  std::atomic<bool> dosage_failed{false};

  auto worker_main = [&](DevWorker *W) {
    std::deque<Block> in_flight;  // submitted, not yet collected (oldest first)
    // Collect number q of the ctx lands in result slot q % n_slots, whose arrays the formatter may still be
    // reading for the batch collected n_slots collects ago.
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
    std::string wmsg;
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
        for (size_t k = 0; k < in_flight.size() && r == BVCF_OK; k++) {
          const Block &q = in_flight[k];
          r = q.bgzf ? bvcf_submit_bgzf(W->ctx, q.buf, q.nb, q.own, q.skip_first, q.first_off, q.seq)
                     : bvcf_submit(W->ctx, q.buf + q.start, q.nb, q.seq);
        }
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
      j.worker = W->idx;
      outstanding.emplace_back(n_jobs, n_collects - 1);
      n_jobs++;
      reorder.put(b.seq, j);
    };
    for (;;) {
      Block b = W->q.pop();
      if (!b.buf) break;
      if (failed.load()) continue;  // (the buffers are released at shutdown)
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
      const int r = b.bgzf ? bvcf_submit_bgzf(W->ctx, b.buf, b.nb, b.own, b.skip_first, b.first_off, b.seq)
                           : bvcf_submit(W->ctx, b.buf + b.start, b.nb, b.seq);
      if (!W->n_blocks) W->t_first_submit = now_s() - t_start;
      W->t_submit += now_s() - ts;
      if (r) {
        fail(std::string("bvcf_submit: ") + bvcf_last_error(W->ctx), r);
        continue;
      }
      W->n_blocks++;
      W->n_bytes += b.nb;
      in_flight.push_back(b);
    }
    while (!failed.load() && !in_flight.empty()) finish_oldest();
    if (failed.load()) reorder.close();  // a block of this worker may never arrive: do not let the formatter wait for it
  };

  // ---- formatter: collected batches in block order
  std::thread formatter([&]() {
    for (;;) {
      FmtJob j;
      if (!reorder.take(&j)) break;
      const double t0 = now_s();
      Parts *ps = nullptr;
      {
        std::lock_guard<std::mutex> lk(spare_mu);
        if (!spares.empty()) {
          ps = spares.back();
          spares.pop_back();
        }
      }
      if (!ps) ps = new Parts();
      const uint8_t *text = j.b.bgzf ? j.res.text : j.b.buf + j.b.start;  // (inflated on the device: the copy that came back)
      std::string jlog;
      format_log(&j.res, text, jlog);
      if (R.want_rows)
        format_parts(c, &j.res, text, *R.names, R.ratios.get(), R.pool.get(), *ps);
      else
        for (auto &q : *ps) q.clear();
      if (append_dosage(R, &j.res, text)) {
        dosage_failed.store(true);
        fail("dosage matrix: write failed", BVCF_E_FATAL);
      }
      t_fmt += now_s() - t0;
      write_q.push(ps);
      if (!jlog.empty()) write_all(fd_err, jlog.data(), jlog.size());
      free_q.push(j.b.buf);
      DevWorker *W = workers[j.worker].get();
      {
        std::lock_guard<std::mutex> lk(W->mu);
        W->fmt_done++;
      }
      W->cv.notify_all();
      if (write_failed.load()) fail("write failed", BVCF_E_FATAL);
    }
    // after a failure nobody may keep waiting for a slot
    for (auto &W : workers) W->cv.notify_all();
  });

  // ---- this thread: preamble, then deal the blocks
  bool have_pre = false, done = false;
  uint64_t seq = 0;
  t_init = now_s() - t_start;
  while (!done) {
    double t0 = now_s();
    Block b = ready_q.pop();
    t_wait_read += now_s() - t0;
    if (b.read_error) fail(source_err.empty() ? std::string("read error") : source_err, BVCF_E_FATAL);
    if (b.too_long) fail("a line is longer than max_batch_bytes", BVCF_E_TOO_BIG);
    if (!failed.load() && b.buf && !have_pre && b.bgzf) {
      // BGZF on the device: the reader inflated the header blocks itself
      t0 = now_s();
      if (!b.pre) {
        fail("internal: BGZF block without its header", BVCF_E_FATAL);
      } else {
        R.pre = b.pre->pre;
        have_pre = true;
        // one more batch in flight per device than for text: two batches' blocks inflate side by side (k_inflate_w16)
        // while a third is in its kernel chain / on its way back
        R.n_slots = 4;
        max_in_flight.store(3);
        int r = prepare_run(R, &msg, b.pre->sample.data(), b.pre->sample.size());
        if (r) fail(msg, r);
        if (!failed.load() && R.pre.header.size() == 9) {
          const char *m = "Found 9 header fields. When genotypes present, we expect 1+ samples after FORMAT (10 fields minimum)\n";
          write_all(fd_err, m, strlen(m));
        }
      }
      t_prepare = now_s() - t0;
      t_init += t_prepare;
    }
    if (!failed.load() && b.buf && !have_pre) {
      t0 = now_s();
      int pr = parse_preamble(b.buf, b.fill, true, c->normalize_header, &R.pre, &msg);
      if (pr != 0) {
        fail(msg, BVCF_E_FATAL);
      } else {
        have_pre = true;
        int r = prepare_run(R, &msg, b.buf + R.pre.data_off, b.fill > R.pre.data_off ? b.fill - R.pre.data_off : 0);
        if (r) fail(msg, r);
        if (!failed.load() && R.pre.header.size() == 9) {
          const char *m = "Found 9 header fields. When genotypes present, we expect 1+ samples after FORMAT (10 fields minimum)\n";
          write_all(fd_err, m, strlen(m));
        }
        b.start = R.pre.data_off;
        b.nb = b.nb > b.start ? b.nb - b.start : 0;
      }
      t_prepare = now_s() - t0;
      t_init += t_prepare;
    }
    if (!failed.load() && b.buf && b.nb) {
      DevWorker *W = workers[seq % n_dev].get();
      if (!W->started) {
        W->started = true;
        W->th = std::thread(worker_main, W);
      }
      b.seq = seq++;
      t0 = now_s();
      W->q.push(b);
      t_deal += now_s() - t0;
    } else if (b.buf) {
      free_q.push(b.buf);
    }
    if (b.last || failed.load()) done = true;
  }

  // ---- shut down
  for (auto &W : workers)
    if (W->started) {
      Block end;
      W->q.push(end);
    }
  for (auto &W : workers)
    if (W->started) W->th.join();
  if (alloc_failed.load()) fail("cannot allocate pinned host memory", BVCF_E_NOMEM);
  if (!have_pre && !failed.load()) fail("EOF", BVCF_E_FATAL);
  reorder.close();
  formatter.join();
  stop.store(true);
  stop_alloc.store(true);
  allocator.join();
  for (int i = 0; i < kBufs; i++) free_q.push(nullptr);  // unblock a reader waiting for a buffer
  // drain blocks the reader may still push so that it can exit
  std::thread drain([&]() {
    for (;;) {
      Block b = ready_q.pop();
      if (b.last) break;
    }
  });
  reader.join();
  {
    Block end;
    end.last = true;
    ready_q.push(end);
  }
  drain.join();
  write_q.push(nullptr);
  writer.join();
  for (Parts *ps : spares) delete ps;
  if (write_failed.load()) fail("write failed", BVCF_E_FATAL);
  if (close_dosage(R)) fail("dosage matrix: write failed", BVCF_E_FATAL);
  if (!log.empty()) write_all(fd_err, log.data(), log.size());
  const double t_end0 = now_s();

  // the final count gather: one RCCL all-reduce over the devices that took part (host sum for one device)
  uint64_t totals[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int used_rccl = 0;
  {
    std::vector<bvcf_ctx *> live;
    for (auto &W : workers)
      if (W->ctx) {
        bvcf_result tmp;  // collect anything left after a failure so the ctx can be destroyed
        while (bvcf_collect(W->ctx, &tmp) != BVCF_E_EMPTY) {
        }
        live.push_back(W->ctx);
      }
    if (!live.empty() && rc == BVCF_OK && bvcf_allreduce_counters(live.data(), (int)live.size(), totals, &used_rccl) != BVCF_OK)
      bvcf_sum_counters(live.data(), (int)live.size(), totals);  // the summary is informational: never fail the run on it
    if (!c->leave_teardown_to_exit)
      for (bvcf_ctx *x : live) bvcf_destroy(x);
  }
  if (!c->leave_teardown_to_exit) free_bufs();
  if (timing) {
    const double t_total = now_s() - t_start;
    double t_ctx = 0, t_gpu = 0, t_submit = 0, t_fmt_wait = 0, t_first = 0;
    size_t used = 0;
    for (auto &W : workers) {
      if (!W->n_blocks) continue;
      used++;
      t_ctx = std::max(t_ctx, W->t_ctx);
      t_gpu = std::max(t_gpu, W->t_gpu);
      t_submit = std::max(t_submit, W->t_submit);
      t_fmt_wait = std::max(t_fmt_wait, W->t_fmt_wait);
      if (W->idx == 0) t_first = W->t_first_submit;
    }
    const double t_steady = t_last_write > t_start + t_first ? t_last_write - t_start - t_first : 0.0;
    if (timing_json) {
      std::string j = "[bvcf timing-json] {";
      char tmp[256];
      auto num = [&](const char *k, double v, bool comma = true) {
        j.append(tmp, (size_t)snprintf(tmp, sizeof tmp, "\"%s\": %.6f%s", k, v, comma ? ", " : ""));
      };
      num("total_s", t_total);
      num("init_s", t_init);
      num("pinned_first_buffer_s", t_pinned);
      num("prepare_s", t_prepare);
      num("ctx_create_max_s", t_ctx);
      num("first_submit_at_s", t_first);
      num("last_write_at_s", t_last_write > t_start ? t_last_write - t_start : 0.0);
      num("steady_s", t_steady);
      num("wait_for_reader_s", t_wait_read);
      num("deal_wait_s", t_deal);
      num("submit_max_s", t_submit);
      num("gpu_wait_max_s", t_gpu);
      num("wait_for_formatter_max_s", t_fmt_wait);
      num("formatter_busy_s", t_fmt);
      num("teardown_s", now_s() - t_end0);
      j.append(tmp, (size_t)snprintf(tmp, sizeof tmp, "\"lines_in\": %llu, \"devices_used\": %zu, \"count_gather\": \"%s\", \"input\": \"%s\", ",
                                     (unsigned long long)lines_in.load(), used, used_rccl ? "rccl" : "host",
                                     input_is_bgzf_device.load() ? "bgzf, inflated on the device" : "text, gzip or bgzf through the host"));
      j.append("\"devices\": [");
      for (size_t d = 0; d < n_dev; d++)
        j.append(tmp, (size_t)snprintf(tmp, sizeof tmp, "%s{\"device\": %d, \"blocks\": %llu, \"bytes\": %llu, \"gpu_wait_s\": %.6f}",
                                       d ? ", " : "", workers[d]->device, (unsigned long long)workers[d]->n_blocks,
                                       (unsigned long long)workers[d]->n_bytes, workers[d]->t_gpu));
      j.append("], \"counters\": [");
      for (int k = 0; k < 8; k++) j.append(tmp, (size_t)snprintf(tmp, sizeof tmp, "%s%llu", k ? ", " : "", (unsigned long long)totals[k]));
      j.append("]}\n");
      write_all(fd_err, j.data(), j.size());
    } else {
      dprintf(fd_err,
              "[bvcf timing] init %.3f (pinned buffer %.3f, prepare %.3f, ctx %.3f) wait-for-reader %.3f deal %.3f submit %.3f "
              "gpu(wait) %.3f wait-for-formatter %.3f (formatter busy %.3f) teardown %.3f total %.3f s; steady %.3f s; "
              "%zu of %zu device(s), count gather: %s\n",
              t_init, t_pinned, t_prepare, t_ctx, t_wait_read, t_deal, t_submit, t_gpu, t_fmt_wait, t_fmt, now_s() - t_end0,
              t_total, t_steady, used, n_dev, used_rccl ? "rccl" : "host");
    }
  }
  if (n_lines_in) *n_lines_in = lines_in.load();
  return rc;
}

}  // extern "C"
