// bvcf_host.cpp — host half of the path, above the C-ABI: the counterpart of readVcf's preamble and
// producer loop (main.go:241-396) and of processLines' TSV assembly (main.go:566-695).
//
// Nothing here computes what the kernels compute: rows are assembled from bvcf_result only.
#include "../../include/bvcf.h"
#include "bvcf_input.h"

#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
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
  char tmp[32];
  int n = snprintf(tmp, sizeof tmp, "%lld", v);
  o.append(tmp, (size_t)n);
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
  Names(const char *const *ptr, const uint32_t *len, size_t n, const char *delim) {
    n_delim = strlen(delim);
    off.reserve(n + 1);
    for (size_t s = 0; s < n; s++) {
      off.push_back((uint32_t)arena.size());
      arena.append(ptr[s], len[s]);
      arena.append(delim, n_delim);
    }
    off.push_back((uint32_t)arena.size());
  }
};

// strings.Join(names of samples with class `want`, fieldDelimiter); `sparse`: the map is a list of its non-zero
// bytes (BVCF_ALLELE_CMAP_SPARSE)
void join_class(std::string &o, const uint8_t *cmap, bool sparse, uint32_t ns, unsigned want, const Names &nm) {
  const size_t at = o.size();
  auto emit = [&](uint32_t b, unsigned byte) {
    for (unsigned j = 0; j < 4; j++) {
      if (((byte >> (2 * j)) & 3u) != want) continue;
      const uint32_t s = b * 4 + j;
      if (s >= ns) break;
      o.append(nm.arena.data() + nm.off[s], nm.off[s + 1] - nm.off[s]);  // name + delimiter
    }
  };
  struct Trim {  // the last entry's delimiter goes
    std::string &o;
    size_t at, n;
    ~Trim() {
      if (o.size() > at) o.resize(o.size() - n);
    }
  } trim{o, at, nm.n_delim};
  if (sparse) {
    uint32_t n;
    memcpy(&n, cmap, 4);
    for (uint32_t i = 0; i < n && i < BVCF_CMAP_SPARSE_MAX; i++) {
      uint32_t e;
      memcpy(&e, cmap + 4 + 4 * i, 4);
      emit(e >> 8, e & 0xFFu);
    }
    return;
  }
  const uint32_t nbytes = (ns + 3) / 4;
  for (uint32_t b = 0; b < nbytes; b++) {
    const unsigned byte = cmap[b];
    if (byte) emit(b, byte);
  }
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
void append_err(std::string &log, const bvcf_err &e, const bvcf_line &L, const uint8_t *block) {
  const char *row = (const char *)block + L.off;
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
void format_lines(const bvcf_config *c, const bvcf_result *r, const uint8_t *block, const Names &nm, uint32_t lo,
                  uint32_t hi, std::string &out) {
  const char *empty = or_default(c->empty_field, "!");
  const uint32_t ns = r->n_samples;
  const double num_samples = (double)ns;
  for (uint32_t li = lo; li < hi; li++) {
    const bvcf_line &L = r->lines[li];
    if (L.status != BVCF_LINE_OK) continue;
    const char *row = (const char *)block + L.off;
    auto fstart = [&](int i) -> uint32_t { return i ? L.fend[i - 1] + 1 : 0; };
    for (uint32_t k = 0; k < L.n_rec; k++) {
      const bvcf_allele &A = r->alleles[k ? L.rec_first + k - 1 : li];
      // main.go:555-560: with samples, an allele nobody carries is skipped
      if (ns > 0 && A.ac == 0) continue;
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
        out.append((const char *)block + A.alt_off, A.alt_len);
      } else {
        out.push_back('-');
        append_ll(out, A.alt_len);
      }
      out.push_back('\t');
      out.push_back((char)('0' + A.trtv));  // main.go:602-606
      out.push_back('\t');

      const double effective = num_samples - (double)A.n_miss;  // main.go:563
      const uint8_t *cm = (A.cmap_off != BVCF_NO_CMAP && r->cmap) ? r->cmap + A.cmap_off : nullptr;
      struct {
        uint32_t n;
        unsigned cls;
        double denom;
      } lists[3] = {{A.n_het, BVCF_CLS_HET, effective}, {A.n_hom, BVCF_CLS_HOM, effective},
                    {A.n_miss, BVCF_CLS_MISSING, num_samples}};
      for (int q = 0; q < 3; q++) {  // main.go:612-656
        if (lists[q].n == 0 || !cm) {
          out.append(empty);
          out.append("\t0");
        } else {
          join_class(out, cm, (A.flags & BVCF_ALLELE_CMAP_SPARSE) != 0, ns, lists[q].cls, nm);
          out.push_back('\t');
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

void format_batch(const bvcf_config *c, const bvcf_result *r, const uint8_t *block, const Names &nm,
                  unsigned n_threads, std::string &out, std::string &log) {
  // log lines in input order (stable: one line's messages keep their ALT order)
  if (r->n_errs) {
    std::vector<uint32_t> idx(r->n_errs);
    for (uint32_t i = 0; i < r->n_errs; i++) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return r->errs[x].line < r->errs[y].line; });
    for (uint32_t i : idx) append_err(log, r->errs[i], r->lines[r->errs[i].line], block);
  }
  if (n_threads <= 1 || r->n_lines < 4 * n_threads) {
    format_lines(c, r, block, nm, 0, r->n_lines, out);
    return;
  }
  std::vector<std::string> parts(n_threads);
  std::vector<std::thread> th;
  for (unsigned t = 0; t < n_threads; t++) {
    const uint32_t lo = (uint32_t)((uint64_t)r->n_lines * t / n_threads);
    const uint32_t hi = (uint32_t)((uint64_t)r->n_lines * (t + 1) / n_threads);
    th.emplace_back([=, &parts, &nm]() { format_lines(c, r, block, nm, lo, hi, parts[t]); });
  }
  for (auto &x : th) x.join();
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
  bvcf_arrow *arrow = nullptr;  // --dosageOutput
  bool want_rows = true;        // !noOut
};

// Which device path suits this file: the streaming path shines when sample fields are the bare
// 4-byte "x|y<TAB>" of a FORMAT == GT file (1000-Genomes style); files whose FORMAT carries more
// sub-fields are scanned by the general path, for which the census path is the faster frame.
uint32_t choose_path(const Run &R, const uint8_t *data, size_t n) {
  if (R.pre.header.size() < 256) return 0;  // the library's own rule (census for narrow files)
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

int open_ctx(Run &R, std::string *msg, const uint8_t *data = nullptr, size_t n_data = 0) {
  if (R.pre.header.size() < 8) {
    // the reference indexes record[6] / record[7] unguarded: out of contract
    *msg = "Malformed header: fewer than 8 fields";
    return BVCF_E_FATAL;
  }
  if (write_sample_list(R)) {  // main.go:298-304
    *msg = "Couldn't write sample list file";
    return BVCF_E_FATAL;
  }
  bvcf_params p;
  memset(&p, 0, sizeof p);
  p.abi_version = BVCF_ABI_VERSION;
  p.device = R.cfg->device;
  p.n_header_fields = (uint32_t)R.pre.header.size();
  p.eol_chars = R.pre.eol_chars;
  p.eol_byte = R.pre.eol_byte;
  R.want_rows = !R.cfg->no_out;
  p.want_class_maps = R.want_rows;  // needsLabels, main.go:502
  p.want_dosage = R.cfg->dosage_path && *R.cfg->dosage_path && R.pre.header.size() > 9;
  p.allow_filter = R.cfg->allow_filter;
  p.exclude_filter = R.cfg->exclude_filter;
  p.max_batch_bytes = R.max_batch;
  p.n_slots = 2;
  p.path = data ? choose_path(R, data, n_data) : 0;
  int rc = bvcf_create(&R.ctx, &p);
  if (rc) {
    *msg = std::string("bvcf_create: ") + bvcf_last_error(nullptr);
    return rc;
  }
  for (size_t i = 9; i < R.pre.header.size(); i++) {
    R.name_ptr.push_back(R.pre.header[i].data());
    R.name_len.push_back((uint32_t)R.pre.header[i].size());
  }
  R.names.reset(new Names(R.name_ptr.data(), R.name_len.data(), R.name_ptr.size(), or_default(R.cfg->field_delimiter, ";")));
  R.n_threads = R.cfg->n_format_threads ? R.cfg->n_format_threads
                                         : std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
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

// the Arrow rows of one collected batch, in input order (main.go:576-584): "chrom:pos:ref:alt" + one int8 per sample
int append_dosage(Run &R, const bvcf_result *r, const uint8_t *block) {
  if (!R.arrow || !r->dosage) return BVCF_OK;
  std::string locus;
  for (uint32_t li = 0; li < r->n_lines; li++) {
    const bvcf_line &L = r->lines[li];
    if (L.status != BVCF_LINE_OK) continue;
    const char *row = (const char *)block + L.off;
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
        locus.append((const char *)block + A.alt_off, A.alt_len);
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
  format_batch(c, r, block, nm, nt, o, l);
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
        format_batch(c, &res, vcf + pos, nm, R.n_threads, o, l);
      } else {
        std::string none;
        format_batch(c, &res, vcf + pos, nm, 1, none, l);  // the log lines only
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

// The reference's main() + readVcf (main.go:134-217, 241-396) as a four-stage pipeline:
//   reader thread  fd -> pinned buffers, cut at the last terminator         (main.go:349-380)
//   this thread    bvcf_submit one block ahead, bvcf_collect the oldest     (processLines' input side)
//   format pool    TSV assembly of a collected batch                        (main.go:566-695)
//   writer thread  ordered write to fd_out                                  (main.go:524-532,705-711)
int bvcf_run_fd(const bvcf_config *c, int fd_in, int fd_out, int fd_err, uint64_t *n_lines_in) {
  if (!c) return BVCF_E_ARG;
  const bool timing = getenv("BVCF_TIMING") != nullptr;
  const double t_start = now_s();
  double t_wait_read = 0, t_gpu = 0, t_fmt = 0, t_init = 0;
  std::string msg;
  Run R;
  R.cfg = c;
  R.max_batch = c->max_batch_bytes ? c->max_batch_bytes : (64ull << 20);
  uint64_t lines_in = 0;

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
  constexpr int kBufs = 4;
  uint8_t *bufs[kBufs];
  for (int i = 0; i < kBufs; i++) bufs[i] = (uint8_t *)bvcf_alloc_pinned(cap);
  auto free_bufs = [&]() {
    for (int i = 0; i < kBufs; i++) bvcf_free_pinned(bufs[i]);
  };
  for (int i = 0; i < kBufs; i++)
    if (!bufs[i]) {
      // no device => no pinned memory either; fail loudly, there is no CPU path
      free_bufs();
      dprintf(fd_err, "cannot allocate pinned host memory (no usable HIP device?)\n");
      return BVCF_E_NODEV;
    }

  Channel<uint8_t *> free_q(64);
  Channel<Block> ready_q(kBufs);
  Channel<std::string *> write_q(4);
  for (int i = 0; i < kBufs; i++) free_q.push(bufs[i]);
  std::atomic<bool> stop{false};
  std::atomic<uint8_t> eol_byte{'\n'};
  std::atomic<bool> eol_known{false};

  // ---- reader: whole lines per block; the partial last line is carried into the next buffer
  std::string source_err;
  std::thread reader([&]() {
    bvcf_input::ByteSource src(fd_in, std::min(32u, std::max(1u, std::thread::hardware_concurrency())));
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
        eol_known.store(true);
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

  // ---- writer
  std::atomic<bool> write_failed{false};
  std::thread writer([&]() {
    for (;;) {
      std::string *s = write_q.pop();
      if (!s) break;
      if (!write_failed.load() && write_all(fd_out, s->data(), s->size())) write_failed.store(true);
      delete s;
    }
  });

  int rc = BVCF_OK;
  bool have_pre = false, done = false;
  std::string log;
  std::deque<Block> in_flight;  // submitted, not yet collected (oldest first)
  uint64_t seq = 0;

  auto fail = [&](const std::string &m, int code) {
    if (rc == BVCF_OK) {
      rc = code;
      log.append(m + "\n");
    }
    done = true;
  };

  // collect the oldest in-flight block, format it, queue its rows; grows the reservation on demand
  auto finish_oldest = [&]() {
    Block b = in_flight.front();
    bvcf_result res;
    double t0 = now_s();
    int r = bvcf_collect(R.ctx, &res);
    if (r == BVCF_E_CAPACITY) {
      // drop what is in flight, grow, resubmit everything still queued on the device side
      for (size_t k = 1; k < in_flight.size(); k++) {
        bvcf_result tmp;
        bvcf_collect(R.ctx, &tmp);
      }
      r = bvcf_reserve(R.ctx, res.need_lines + res.need_lines / 4 + 64, res.need_alleles + res.need_alleles / 4 + 64,
                       res.need_cmap_bytes + res.need_cmap_bytes / 4 + 4096);
      for (size_t k = 0; k < in_flight.size() && r == BVCF_OK; k++)
        r = bvcf_submit(R.ctx, in_flight[k].buf + in_flight[k].start, in_flight[k].nb, seq++);
      if (r == BVCF_OK) r = bvcf_collect(R.ctx, &res);
    }
    t_gpu += now_s() - t0;
    if (r != BVCF_OK) {
      fail(std::string("bvcf: ") + bvcf_last_error(R.ctx), r);
      return;
    }
    in_flight.pop_front();
    lines_in += res.n_lines_seen;
    t0 = now_s();
    std::string *out = new std::string();
    const Names &nm = *R.names;
    format_batch(c, &res, b.buf + b.start, nm, R.want_rows ? R.n_threads : 1, *out, log);
    if (!R.want_rows) out->clear();
    if (append_dosage(R, &res, b.buf + b.start)) fail("dosage matrix: write failed", BVCF_E_FATAL);
    t_fmt += now_s() - t0;
    write_q.push(out);
    if (!log.empty()) {
      write_all(fd_err, log.data(), log.size());
      log.clear();
    }
    free_q.push(b.buf);
    if (write_failed.load()) fail("write failed", BVCF_E_FATAL);
  };

  t_init = now_s() - t_start;
  while (!done) {
    double t0 = now_s();
    Block b = ready_q.pop();
    t_wait_read += now_s() - t0;
    if (b.read_error) fail(source_err.empty() ? std::string("read error") : source_err, BVCF_E_FATAL);
    if (b.too_long) fail("a line is longer than max_batch_bytes", BVCF_E_TOO_BIG);
    if (!done && b.buf && !have_pre) {
      t0 = now_s();
      int pr = parse_preamble(b.buf, b.fill, true, c->normalize_header, &R.pre, &msg);
      if (pr != 0) {
        fail(msg, BVCF_E_FATAL);
      } else {
        have_pre = true;
        int r = open_ctx(R, &msg, b.buf + R.pre.data_off, b.fill > R.pre.data_off ? b.fill - R.pre.data_off : 0);
        if (r) fail(msg, r);
        if (!done && R.pre.header.size() == 9)
          log.append("Found 9 header fields. When genotypes present, we expect 1+ samples after FORMAT (10 fields minimum)\n");
        b.start = R.pre.data_off;
        b.nb = b.nb > b.start ? b.nb - b.start : 0;
      }
      t_init += now_s() - t0;
    }
    if (!done && b.buf && b.nb) {
      // keep one block ahead of the one being formatted
      if (in_flight.size() >= 2) finish_oldest();
      if (!done) {
        int r = bvcf_submit(R.ctx, b.buf + b.start, b.nb, seq++);
        if (r)
          fail(std::string("bvcf_submit: ") + bvcf_last_error(R.ctx), r);
        else
          in_flight.push_back(b);
      }
    } else if (b.buf) {
      free_q.push(b.buf);
    }
    if (b.last) {
      while (!done && !in_flight.empty()) finish_oldest();
      if (!have_pre && rc == BVCF_OK) fail("EOF", BVCF_E_FATAL);
      done = true;
    }
  }

  // ---- shut down
  stop.store(true);
  for (int i = 0; i < kBufs; i++) free_q.push(nullptr);  // unblock a reader waiting for a buffer
  // drain blocks the reader may still push so that it can exit
  std::thread drain([&]() {
    for (;;) {
      Block b = ready_q.pop();
      if (b.last && !b.buf) break;
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
  if (close_dosage(R) && rc == BVCF_OK) {
    log.append("dosage matrix: write failed\n");
    rc = BVCF_E_FATAL;
  }
  if (!log.empty()) write_all(fd_err, log.data(), log.size());
  const double t_end0 = now_s();
  if (R.ctx) {
    // collect anything left after a failure so the ctx can be destroyed
    bvcf_result tmp;
    while (bvcf_collect(R.ctx, &tmp) != BVCF_E_EMPTY) {
    }
    bvcf_destroy(R.ctx);
  }
  free_bufs();
  if (timing)
    dprintf(fd_err,
            "[bvcf timing] init %.3f wait-for-reader %.3f gpu(wait) %.3f format %.3f teardown %.3f total %.3f s\n",
            t_init, t_wait_read, t_gpu, t_fmt, now_s() - t_end0, now_s() - t_start);
  if (n_lines_in) *n_lines_in = lines_in;
  return rc;
}

}  // extern "C"
