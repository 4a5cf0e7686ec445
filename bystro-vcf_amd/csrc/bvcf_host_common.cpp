// bvcf_host_common.cpp — what bvcf_host.cpp (in-memory driver), bvcf_driver.cpp / bvcf_readers.cpp (the stream driver)
// share: readVcf's preamble (main.go:241-304), processLines' TSV assembly (main.go:566-695), the run's shared state.
// Declarations: bvcf_host_internal.h.
//
// Nothing here computes what the kernels compute: rows are assembled from bvcf_result only.
#include "bvcf_host_internal.h"

#include <sched.h>

namespace bvcf_host {

// parse.Header (main.go:224), pinned by main_test.go:79-80
extern const char *const kBaseHeader[15] = {"chrom",       "pos",           "type",         "ref",          "alt",
                                     "trTv",        "heterozygotes", "heterozygosity", "homozygotes", "homozygosity",
                                     "missingGenos", "missingness",  "ac",           "an",           "sampleMaf"};

// parse.Snp / Ins / Del / Mnp / Multi
extern const char *const kSiteNames[5] = {"SNP", "INS", "DEL", "MNP", "MULTIALLELIC"};

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
const char *row_of(const bvcf_result *r, const uint8_t *block, uint32_t li, const bvcf_line &L) {
  return (const char *)block + (r->head_off ? r->head_off[li] : L.off);
}

// Line li of a batch as full records: the batch's own -- the unpacked form, or a BVCF_SITE_FULL line of the packed form
// (bvcf_params.packed_sites) -- or expanded from the line's 32-byte site record into *tl / *ta: a line that passed is a
// biallelic SNP with its position taken verbatim (main.go:735-745).
LineView line_view(const bvcf_result *r, uint32_t li, bvcf_line *tl, bvcf_allele *ta) {
  LineView v;
  if (!r->sites && r->row_cuts) {
    // rendered rows (bvcf_params.render_sites): only the lines left to the host are ever looked at, and their records are
    // where their cut says (the cuts are in line order)
    const bvcf_row_cut *lo = r->row_cuts, *hi = r->row_cuts + r->n_row_cuts;
    const bvcf_row_cut *it = std::lower_bound(lo, hi, li, [](const bvcf_row_cut &q, uint32_t x) { return q.line < x; });
    const bool found = it != hi && it->line == li && it->slot < r->n_full_lines;
    const uint32_t slot = found ? it->slot : 0u;
    v.L = &r->lines[slot];
    v.A0 = &r->alleles[slot];
    if (found && r->text && it->text_off != BVCF_NO_TEXT_OFF) v.row = (const char *)r->text + it->text_off;
    return v;
  }
  if (!r->sites) {
    v.L = &r->lines[li];
    v.A0 = &r->alleles[li];
    return v;
  }
  const bvcf_site &s = r->sites[li];
  if (s.status & BVCF_SITE_FULL) {
    v.L = &r->lines[s.full_idx];
    v.A0 = &r->alleles[s.full_idx];
    return v;
  }
  memset(tl, 0, sizeof *tl);
  tl->off = s.off;
  tl->len = s.len;
  for (int k = 0; k < 9; k++) tl->fend[k] = (k < 8 && s.fend[k] != 0xFFu) ? s.fend[k] : s.len;
  tl->n_rec = s.status == BVCF_LINE_OK ? 1u : 0u;
  tl->n_fields = s.n_fields;
  tl->gt_task = li;
  tl->status = s.status;
  tl->site_type = BVCF_SITE_SNP;
  memset(ta, 0, sizeof *ta);
  ta->line = li;
  ta->alt_len = 1;
  ta->cmap_off = BVCF_NO_CMAP;
  ta->ref = s.ref;
  ta->alt_base = s.alt_base;
  ta->kind = BVCF_ALT_BASE;
  ta->site_type = BVCF_SITE_SNP;
  ta->trtv = s.trtv;
  ta->flags = BVCF_ALLELE_POS_TEXT;
  ta->gt_task = 0xFFFFFFFFu;
  v.L = tl;
  v.A0 = ta;
  return v;
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
  bvcf_line tmp_line;
  bvcf_allele tmp_allele;
  // The packed form's common line -- a biallelic SNP of a file without samples (main.go:735-745): its row is the CHROM
  // and POS bytes, REF, ALT, trTv and a tail that is the same for every such line (no carriers: three empty lists, zero
  // counts).  It is put together in a local buffer and appended once; a sites-only run is bound by this function (20 M
  // rows: 0.6 s of eight formatter threads through the generic code below, against 0.04 s of GPU time).
  std::string site_tail;
  if (r->sites && ns == 0) {
    site_tail.push_back('\t');
    for (int q = 0; q < 3; q++) {
      site_tail.append(empty);
      site_tail.append("\t0\t");
    }
    site_tail.append("0\t0\t0");
  }
  for (uint32_t li = lo; li < hi; li++) {
    if (r->sites && ns == 0) {
      const bvcf_site &s = r->sites[li];
      if (!(s.status & BVCF_SITE_FULL)) {
        if (s.status != BVCF_LINE_OK) continue;
        const uint32_t f0 = s.fend[0], f1 = s.fend[1] != 0xFFu ? s.fend[1] : s.len;
        const uint32_t f2 = s.fend[2] != 0xFFu ? s.fend[2] : s.len, f6 = s.fend[6] != 0xFFu ? s.fend[6] : s.len;
        const uint32_t f7 = s.fend[7] != 0xFFu ? s.fend[7] : s.len;
        const uint32_t n_pos = f1 - f0 - 1, n_id = c->keep_id ? f2 - f1 - 1 : 0, n_info = c->keep_info ? f7 - f6 - 1 : 0;
        char buf[1024];
        if (f0 + 2u * n_pos + n_id + n_info + site_tail.size() + 32u <= sizeof buf) {
          const char *row = (const char *)block + (r->head_off ? r->head_off[li] : s.off);
          char *p = buf;
          if (f0 < 4 || row[0] != 'c') {  // main.go:570-574
            memcpy(p, "chr", 3);
            p += 3;
          }
          memcpy(p, row, f0 + 1 + n_pos);  // CHROM, the TAB, POS verbatim
          p += f0 + 1 + n_pos;
          memcpy(p, "\tSNP\t", 5);
          p += 5;
          *p++ = (char)s.ref;
          *p++ = '\t';
          *p++ = (char)s.alt_base;
          *p++ = '\t';
          *p++ = (char)('0' + s.trtv);  // main.go:602-606
          memcpy(p, site_tail.data(), site_tail.size());
          p += site_tail.size();
          if (c->keep_pos) {  // main.go:674-692
            *p++ = '\t';
            memcpy(p, row + f0 + 1, n_pos);
            p += n_pos;
          }
          if (c->keep_id) {
            *p++ = '\t';
            memcpy(p, row + f1 + 1, n_id);
            p += n_id;
          }
          if (c->keep_info) {
            memcpy(p, "\t0\t", 3);  // (alleleIdx of a biallelic line's one allele, main.go:687)
            p += 3;
            memcpy(p, row + f6 + 1, n_info);
            p += n_info;
          }
          *p++ = '\n';
          out.append(buf, (size_t)(p - buf));
          continue;
        }
      }
    }
    const LineView view = line_view(r, li, &tmp_line, &tmp_allele);
    const bvcf_line &L = *view.L;
    if (L.status != BVCF_LINE_OK) continue;
    const char *row = view.row ? view.row : row_of(r, block, li, L);
    auto fstart = [&](int i) -> uint32_t { return i ? L.fend[i - 1] + 1 : 0; };
    for (uint32_t k = 0; k < L.n_rec; k++) {
      const uint32_t slot = k ? L.rec_first + k - 1 : li;  // (k == 0, packed form: the record is *view.A0, wherever it lies)
      const bvcf_allele &A = k ? r->alleles[slot] : *view.A0;
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

// CPUs this process may actually use: the smallest of the hardware's count, the affinity mask and the cgroup's CPU
// quota (a container on a 256-thread host with "cpu.max 1600000 100000" gets 16 cores' worth of time: thread pools sized
// by the hardware count only buy throttling -- whole scheduling periods in which every thread of the process stands
// still, readers and device threads included).
unsigned usable_cpus() {
  static const unsigned cached = [] {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
      const int k = CPU_COUNT(&set);
      if (k > 0) n = std::min(n, (unsigned)k);
    }
    auto apply = [&](double cores) {
      if (cores > 0) n = std::min(n, std::max(1u, (unsigned)(cores + 0.999)));
    };
    // cgroup v2: "<quota|max> <period>" in cpu.max of the process's group and of every group above it
    std::string rel;
    if (FILE *f = fopen("/proc/self/cgroup", "r")) {
      char line[1024];
      while (fgets(line, sizeof line, f))
        if (strncmp(line, "0::", 3) == 0) {
          rel = line + 3;
          while (!rel.empty() && (rel.back() == '\n' || rel.back() == '/')) rel.pop_back();
        }
      fclose(f);
    }
    for (;;) {
      const std::string path = "/sys/fs/cgroup" + rel + "/cpu.max";
      if (FILE *f = fopen(path.c_str(), "r")) {
        char q[64];
        long long period = 0;
        if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) apply((double)atoll(q) / (double)period);
        fclose(f);
      }
      if (rel.empty()) break;
      const size_t cut = rel.rfind('/');
      rel = cut == std::string::npos ? std::string() : rel.substr(0, cut);
    }
    // cgroup v1
    long long quota = -1, period = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
      if (fscanf(f, "%lld", &quota) != 1) quota = -1;
      fclose(f);
    }
    if (FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
      if (fscanf(f, "%lld", &period) != 1) period = 0;
      fclose(f);
    }
    if (quota > 0 && period > 0) apply((double)quota / (double)period);
    if (const char *e = getenv("BVCF_CPUS")) n = (unsigned)std::max(1, atoi(e));  // tuning
    return n;
  }();
  return cached;
}


// the batch's log lines in input order (stable: one line's messages keep their ALT order)
void format_log(const bvcf_result *r, const uint8_t *block, std::string &log) {
  if (!r->n_errs) return;
  std::vector<uint32_t> idx(r->n_errs);
  for (uint32_t i = 0; i < r->n_errs; i++) idx[i] = i;
  // (bvcf_err.pad: the message's place among its line's, where they were not logged in order)
  std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) {
    return r->errs[x].line != r->errs[y].line ? r->errs[x].line < r->errs[y].line : r->errs[x].pad < r->errs[y].pad;
  });
  bvcf_line tl;
  bvcf_allele ta;
  for (uint32_t i : idx) {
    const uint32_t li = r->errs[i].line;
    const LineView view = line_view(r, li, &tl, &ta);
    const bvcf_line &L = *view.L;
    append_err(log, r->errs[i], L, view.row ? view.row : row_of(r, block, li, L));
  }
}

// rows of one batch as consecutive pieces (parts[0] + parts[1] + ... is the batch's TSV): runs of lines are claimed
// by the pool's threads, a few per thread so that lines with long sample lists do not leave the others idle
void format_parts(const bvcf_config *c, const bvcf_result *r, const uint8_t *block, const Names &nm, const Ratios *rt,
                  WorkPool *pool, std::vector<std::string> &parts) {
  const unsigned nt = pool ? pool->size() : 1;
  if (!r->sites && r->row_cuts) {
    // The rows of this batch were made on the device (bvcf_params.render_sites): the batch's TSV is the stream, with the
    // rows of the lines left to the host (indels, several ALTs, ...: their cuts, in line order and so in stream order) put in
    // at their offsets.  Parts are byte ranges of the stream: a part copies its bytes and formats the cut lines that fall
    // into it.
    const uint64_t B = r->n_row_bytes;
    uint32_t n_parts = 1;
    if (nt > 1 && B >= (1u << 20)) n_parts = (uint32_t)std::min<uint64_t>(4 * nt, B >> 18);
    if (parts.size() < n_parts) parts.resize(n_parts);
    for (auto &p : parts) p.clear();
    auto one = [&](uint32_t t) {
      const uint64_t lo = B * t / n_parts, hi = B * (t + 1) / n_parts;
      const bool last = t + 1 == n_parts;
      const bvcf_row_cut *cb = r->row_cuts, *ce = r->row_cuts + r->n_row_cuts;
      const bvcf_row_cut *it = std::lower_bound(cb, ce, lo, [](const bvcf_row_cut &q, uint64_t x) { return q.off < x; });
      std::string &out = parts[t];
      out.reserve((size_t)(hi - lo) + 256);
      uint64_t pos = lo;
      for (; it != ce && (it->off < hi || (last && it->off == hi)); ++it) {
        out.append((const char *)r->rows + pos, (size_t)(it->off - pos));
        pos = it->off;
        format_lines(c, r, block, nm, rt, it->line, it->line + 1, out);
      }
      out.append((const char *)r->rows + pos, (size_t)(hi - pos));
    };
    if (n_parts == 1)
      one(0);
    else
      pool->run(n_parts, one);
    return;
  }
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


// Which device path suits this file: the streaming path reads the text once -- the bare 4-byte "x|y<TAB>" fields of
// a FORMAT == GT file (1000-Genomes style) through its regular scan, fields with further sub-fields through its
// general stream -- as long as a line's class map fits the LDS stage (16 384 samples); beyond that, lines that are
// not regular would all be left to k_gt, for which the census path is the better frame.
uint32_t choose_path(const Run &R, const uint8_t *data, size_t n) {
  if (R.pre.header.size() < 256) return 0;  // the library's own rule (census for narrow files)
  if (R.pre.header.size() >= 9 + (size_t)BVCF_WIDE_SAMPLES) return 0;  // very wide lines: the census path's split scan
  if (R.pre.header.size() <= 9 + 16384u) {
    // the first data line says which streaming kernel the first batch should take (a BGZF batch is launched before anyone
    // has seen its text): FORMAT is not plain "GT" -> 3
    size_t pos = 0;
    for (int tabs = 0; pos < n && tabs < 8; pos++) {
      if (data[pos] == R.pre.eol_byte) return 2;
      tabs += data[pos] == '\t';
    }
    size_t e = pos;
    while (e < n && data[e] != '\t' && data[e] != R.pre.eol_byte) e++;
    if (e >= n || e == pos) return 2;
    return (e - pos == 2 && data[pos] == 'G' && data[pos + 1] == 'T') ? 2u : 3u;
  }
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
int prepare_run(Run &R, std::string *msg, const uint8_t *data, size_t n_data, bool make_pool) {
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
  {
    // a file without samples comes back in the packed form: 32 bytes per line instead of 128 (BVCF_PACKED_SITES=0: the
    // full form, for A/B and parity tests)
    const char *e = getenv("BVCF_PACKED_SITES");
    p.packed_sites = R.pre.header.size() <= 9 && !(e && *e == '0');
    // ... and the rows of the lines the packed form settles are made on the device (bvcf_params.render_sites, ABI 7: a
    // sites-only run is otherwise bound by the formatter threads; BVCF_RENDER_SITES=0: the host formats every row, for A/B
    // and parity tests)
    const char *e2 = getenv("BVCF_RENDER_SITES");
    p.render_sites = p.packed_sites && R.want_rows && !(e2 && *e2 == '0') && strlen(or_default(R.cfg->empty_field, "!")) <= 16;
  }
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
                                         : std::min(32u, usable_cpus());
  if (make_pool && R.want_rows && R.n_threads > 1) R.pool.reset(new WorkPool(R.n_threads));
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
  if (rc == BVCF_OK && p.render_sites) {
    rc = bvcf_set_row_format(*ctx, or_default(R.cfg->empty_field, "!"), R.cfg->keep_pos, R.cfg->keep_id, R.cfg->keep_info);
    if (rc) {
      *msg = std::string("bvcf_set_row_format: ") + bvcf_last_error(*ctx);
      bvcf_destroy(*ctx);
      *ctx = nullptr;
    }
  }
  return rc;
}

int open_ctx(Run &R, std::string *msg, const uint8_t *data, size_t n_data) {
  int rc = prepare_run(R, msg, data, n_data);
  if (rc == BVCF_OK) rc = create_ctx(R, R.cfg->device, &R.ctx, msg);
  return rc;
}

// the Arrow rows of one collected batch, in input order (main.go:576-584): "chrom:pos:ref:alt" + one int8 per sample
int append_dosage(Run &R, const bvcf_result *r, const uint8_t *block) {
  if (!R.arrow || !r->dosage) return BVCF_OK;
  std::string locus;
  for (uint32_t li = 0; li < r->n_lines; li++) {  // (a dosage matrix needs samples: never the packed form)
    const bvcf_line &L = r->lines[li];
    if (L.status != BVCF_LINE_OK) continue;
    const char *row = row_of(r, block, li, L);
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





int write_all(int fd, const char *p, size_t n) {
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

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}


}  // namespace bvcf_host
