// bvcf_host.cpp — host half of the path, above the C-ABI: TSV assembly of collected batches, the in-memory driver
// (bvcf_run_buffer), the byte source on its own.  The stream driver (bvcf_run_fd) is bvcf_driver.cpp.
//
// Nothing here computes what the kernels compute: rows are assembled from bvcf_result only.
#include "bvcf_host_internal.h"

using namespace bvcf_host;



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
  if (!n_threads) n_threads = std::min(32u, usable_cpus());
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
      c->n_format_threads ? c->n_format_threads : std::min(32u, usable_cpus());
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

}  // extern "C"
