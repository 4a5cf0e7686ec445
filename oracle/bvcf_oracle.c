/*
 * bvcf_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of bystro-vcf's per-line variant pipeline, following
 * /root/reference/main.go @ 2024_10_08 function by function.  It keeps the
 * reference's *structure* (split every line into a field array, per-allele
 * rescan of all samples, N workers over 64-line batches) so that it is also a
 * fair "reference CPU path" for bench.py's cpu_baseline leg.
 *
 * Third-party arithmetic (github.com/bystrogenomics/bystro-utils/parse
 * @ v0.0.0-20180921004542-b5183a523f20, not vendored in the reference) is
 * restated from its call sites, the reference's tests and golden output:
 *   parse.Header / Snp / Ins / Del / Mnp / Multi / NotTrTv / GetTrTv : pinned
 *   parse.FindEndOfLine "\n" : pinned;  "\r\n", "\r" : PARITY UNPINNED
 *   parse.NormalizeHeader ('.' -> '_')               : PARITY UNPINNED
 *
 * Only tests/, __graft_entry__.smoke() and bench.py (cpu_baseline) use this.
 */
#define _GNU_SOURCE
#include "bvcf_oracle.h"

#include <errno.h>
#include <limits.h>
#include <pthread.h>
#include <time.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef ORC_MAIN
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#endif

/* ------------------------------------------------------------------ buffers */

void orc_buf_init(orc_buf *b) {
  b->data = NULL;
  b->len = 0;
  b->cap = 0;
}

void orc_buf_free(orc_buf *b) {
  free(b->data);
  orc_buf_init(b);
}

static void buf_reserve(orc_buf *b, size_t extra) {
  if (b->len + extra + 1 <= b->cap) return;
  size_t ncap = b->cap ? b->cap * 2 : 4096;
  while (ncap < b->len + extra + 1) ncap *= 2;
  char *p = (char *)realloc(b->data, ncap);
  if (!p) {
    fprintf(stderr, "oracle: out of memory\n");
    abort();
  }
  b->data = p;
  b->cap = ncap;
}

static void buf_write(orc_buf *b, const char *s, size_t n) {
  buf_reserve(b, n);
  memcpy(b->data + b->len, s, n);
  b->len += n;
  b->data[b->len] = 0;
}

static void buf_puts(orc_buf *b, const char *s) { buf_write(b, s, strlen(s)); }

static void buf_putc(orc_buf *b, char c) { buf_write(b, &c, 1); }

/* log.Printf: one line per call; a trailing '\n' in the format is not doubled */
static void log_printf(orc_buf *log, const char *fmt, ...) {
  if (!log) return;
  char tmp[1024];
  va_list ap;
  va_start(ap, fmt);
  int n = vsnprintf(tmp, sizeof tmp, fmt, ap);
  va_end(ap);
  if (n < 0) return;
  if ((size_t)n < sizeof tmp) {
    buf_write(log, tmp, (size_t)n);
  } else {
    char *big = (char *)malloc((size_t)n + 1);
    va_start(ap, fmt);
    vsnprintf(big, (size_t)n + 1, fmt, ap);
    va_end(ap);
    buf_write(log, big, (size_t)n);
    free(big);
  }
  if (log->len == 0 || log->data[log->len - 1] != '\n') buf_putc(log, '\n');
}

/* one getAlleles message: "<chrom>:<pos><rest>\n" with chrom/pos as raw bytes (Go strings may hold NULs) */
static void log_site(orc_buf *log, const char *chrom, size_t nchrom, const char *pos, size_t npos,
                     const char *fmt, ...) {
  if (!log) return;
  char tmp[256];
  va_list ap;
  va_start(ap, fmt);
  int n = vsnprintf(tmp, sizeof tmp, fmt, ap);
  va_end(ap);
  buf_write(log, chrom, nchrom);
  buf_putc(log, ':');
  buf_write(log, pos, npos);
  if (n > 0) buf_write(log, tmp, (size_t)n < sizeof tmp ? (size_t)n : sizeof tmp - 1);
  buf_putc(log, '\n');
}

/* strconv.Itoa */
static void buf_itoa(orc_buf *b, long long v) {
  char tmp[32];
  int n = snprintf(tmp, sizeof tmp, "%lld", v);
  buf_write(b, tmp, (size_t)n);
}

/* strconv.FormatFloat(x, 'G', 3, 64); C "%.3G" is identical on [0,1] */
static void buf_float_g3(orc_buf *b, double x) {
  char tmp[64];
  int n = snprintf(tmp, sizeof tmp, "%.3G", x);
  buf_write(b, tmp, (size_t)n);
}

/* ------------------------------------------------------------------ config */

void orc_config_defaults(orc_config *c) {
  /* main.go:84-99 */
  c->empty_field = "!";
  c->field_delimiter = ";";
  c->keep_id = 0;
  c->keep_info = 0;
  c->keep_pos = 0;
  c->allow_filter = "PASS,.";
  c->exclude_filter = "";
  c->n_threads = 1;
  c->normalize_header = 1;
}

/* a parsed --allowFilter / --excludeFilter set; is_nil mirrors a nil Go map */
typedef struct {
  int is_nil;
  int n;
  char **vals;
  size_t *lens;
} filter_set;

static int is_space(char c) {
  return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r';
}

/* main.go:108-123: strings.Split(v, ",") then strings.TrimSpace each */
static void filter_set_parse(filter_set *fs, const char *text, int star_is_nil) {
  fs->is_nil = 1;
  fs->n = 0;
  fs->vals = NULL;
  fs->lens = NULL;
  if (!text || !*text) return;
  if (star_is_nil && strcmp(text, "*") == 0) return;
  fs->is_nil = 0;
  size_t n = strlen(text);
  int cnt = 1;
  for (size_t i = 0; i < n; i++)
    if (text[i] == ',') cnt++;
  fs->vals = (char **)calloc((size_t)cnt, sizeof(char *));
  fs->lens = (size_t *)calloc((size_t)cnt, sizeof(size_t));
  size_t start = 0;
  for (size_t i = 0; i <= n; i++) {
    if (i == n || text[i] == ',') {
      size_t a = start, e = i;
      while (a < e && is_space(text[a])) a++;
      while (e > a && is_space(text[e - 1])) e--;
      char *v = (char *)malloc(e - a + 1);
      memcpy(v, text + a, e - a);
      v[e - a] = 0;
      fs->vals[fs->n] = v;
      fs->lens[fs->n] = e - a;
      fs->n++;
      start = i + 1;
    }
  }
}

static void filter_set_free(filter_set *fs) {
  for (int i = 0; i < fs->n; i++) free(fs->vals[i]);
  free(fs->vals);
  free(fs->lens);
}

static int filter_set_has(const filter_set *fs, const char *s, size_t n) {
  for (int i = 0; i < fs->n; i++)
    if (fs->lens[i] == n && memcmp(fs->vals[i], s, n) == 0) return 1;
  return 0;
}

/* ------------------------------------------------------------------ header */

/* parse.Header (pinned by main_test.go:79-80 and the golden's first line) */
static const char *const k_base_header[15] = {
    "chrom",       "pos",          "type",         "ref",         "alt",
    "trTv",        "heterozygotes", "heterozygosity", "homozygotes", "homozygosity",
    "missingGenos", "missingness", "ac",           "an",          "sampleMaf"};

/* main.go:219-239 */
void orc_string_header(const orc_config *cfg, orc_buf *out) {
  for (int i = 0; i < 15; i++) {
    if (i) buf_putc(out, '\t');
    buf_puts(out, k_base_header[i]);
  }
  if (cfg->keep_pos) buf_puts(out, "\tvcfPos");
  if (cfg->keep_id) buf_puts(out, "\tid");
  if (cfg->keep_info) buf_puts(out, "\talleleIdx\tinfo");
}

/* ------------------------------------------------------------------ QC */

static int is_actg(char c) { return c == 'A' || c == 'C' || c == 'T' || c == 'G'; }

/* main.go:456-474.  The reference indexes alt[0] on an empty string (a Go
 * panic); the restatement treats empty as invalid. */
int orc_alt_is_valid(const char *alt, size_t n) {
  if (n == 0) return 0;
  for (size_t i = 0; i < n; i++)
    if (!is_actg(alt[i])) return 0;
  return 1;
}

/* main.go:447-454 */
static int line_passes(size_t n_record, size_t n_header, const char *filter, size_t nfilter,
                       const filter_set *allowed, const filter_set *excluded) {
  return n_record == n_header && (allowed->is_nil || filter_set_has(allowed, filter, nfilter)) &&
         (excluded->is_nil || !filter_set_has(excluded, filter, nfilter));
}

/* strconv.Atoi: optional sign, decimal digits only, must fit in int64 */
static int go_atoi(const char *s, size_t n, long long *out) {
  if (n == 0) return 0;
  size_t i = 0;
  int neg = 0;
  if (s[0] == '+' || s[0] == '-') {
    neg = s[0] == '-';
    i = 1;
    if (n == 1) return 0;
  }
  unsigned long long v = 0;
  const unsigned long long lim = neg ? 9223372036854775808ULL : 9223372036854775807ULL;
  for (; i < n; i++) {
    if (s[i] < '0' || s[i] > '9') return 0;
    unsigned d = (unsigned)(s[i] - '0');
    if (v > (lim - d) / 10) return 0;
    v = v * 10 + d;
  }
  *out = neg ? (long long)(0ULL - v) : (long long)v;
  return 1;
}

/* ------------------------------------------------------------------ getAlleles */

static void alleles_init(orc_alleles *a) {
  memset(a, 0, sizeof *a);
}

void orc_alleles_free(orc_alleles *a) {
  for (int i = 0; i < a->n; i++) {
    free(a->positions[i]);
    free(a->alts[i]);
  }
  free(a->positions);
  free(a->refs);
  free(a->alts);
  free(a->alt_indices);
  alleles_init(a);
}

static char *dup_n(const char *s, size_t n) {
  char *p = (char *)malloc(n + 1);
  memcpy(p, s, n);
  p[n] = 0;
  return p;
}

static char *dup_ll(long long v) {
  char tmp[32];
  int n = snprintf(tmp, sizeof tmp, "%lld", v);
  return dup_n(tmp, (size_t)n);
}

static void alleles_push(orc_alleles *a, char *pos, char ref, char *alt, int idx) {
  int n = a->n + 1;
  a->positions = (char **)realloc(a->positions, (size_t)n * sizeof(char *));
  a->refs = (char *)realloc(a->refs, (size_t)n);
  a->alts = (char **)realloc(a->alts, (size_t)n * sizeof(char *));
  a->alt_indices = (int *)realloc(a->alt_indices, (size_t)n * sizeof(int));
  a->positions[a->n] = pos;
  a->refs[a->n] = ref;
  a->alts[a->n] = alt;
  a->alt_indices[a->n] = idx;
  a->n = n;
}

static char *plus_alt(const char *s, size_t n) {
  char *p = (char *)malloc(n + 2);
  p[0] = '+';
  memcpy(p + 1, s, n);
  p[n + 1] = 0;
  return p;
}

/* main.go:723-1038 */
void orc_get_alleles(const char *chrom, size_t nchrom, const char *pos, size_t npos,
                     const char *ref, size_t nref, const char *alt, size_t nalt,
                     orc_alleles *out, orc_buf *log) {
  alleles_init(out);

  /* main.go:729-732 */
  if (nalt == nref && memcmp(alt, ref, nalt) == 0) {
    log_site(log, chrom, nchrom, pos, npos, " : %s", "REF == ALT");
    return;
  }

  /* main.go:735-765 */
  if (nalt == 1) {
    if (alt[0] != 'A' && alt[0] != 'C' && alt[0] != 'G' && alt[0] != 'T') {
      log_site(log, chrom, nchrom, pos, npos, " ALT #1 %s", "ALT not ACTG");
      return;
    }
    if (nref == 1) {
      strcpy(out->site_type, "SNP");
      alleles_push(out, dup_n(pos, npos), ref[0], dup_n(alt, 1), 0);
      return;
    }
    /* nref == 0 would index ref[0] out of range in Go (panic); reject */
    if (nref == 0) {
      log_site(log, chrom, nchrom, pos, npos, " ALT #1 %s", "empty REF");
      return;
    }
    if (alt[0] != ref[0]) {
      log_site(log, chrom, nchrom, pos, npos, " ALT #1 %s", "1st base REF != ALT");
      return;
    }
    long long int_pos;
    if (!go_atoi(pos, npos, &int_pos)) {
      log_site(log, chrom, nchrom, pos, npos, " ALT #1 %s", "Invalid POS");
      return;
    }
    strcpy(out->site_type, "DEL");
    alleles_push(out, dup_ll(int_pos + 1), ref[1], dup_ll(1 - (long long)nref), 0);
    return;
  }

  if (nref == 0) { /* Go would panic on ref[0] below; reject */
    log_site(log, chrom, nchrom, pos, npos, " %s", "empty REF");
    return;
  }

  long long int_pos = 0;
  int multi = 0;
  int alt_idx = -1;
  size_t start = 0;
  /* main.go:774: strings.Split(alt, ",") */
  for (size_t i = 0; i <= nalt; i++) {
    if (i != nalt && alt[i] != ',') continue;
    const char *t = alt + start;
    const size_t nt = i - start;
    start = i + 1;
    alt_idx++;

    if (!multi && alt_idx > 0) multi = 1; /* main.go:777-779 */

    if (!orc_alt_is_valid(t, nt)) { /* main.go:781-784 */
      log_site(log, chrom, nchrom, pos, npos, " ALT #%d %s", alt_idx + 1, "ALT not ACTG");
      continue;
    }

    if (nref == 1) { /* main.go:786-815 */
      if (nt == 1) {
        alleles_push(out, dup_n(pos, npos), ref[0], dup_n(t, 1), alt_idx);
        continue;
      }
      if (t[0] != ref[0]) {
        log_site(log, chrom, nchrom, pos, npos, " ALT #%d %s", alt_idx + 1,
                   "1st base ALT != REF");
        continue;
      }
      alleles_push(out, dup_n(pos, npos), ref[0], plus_alt(t + 1, nt - 1), alt_idx);
      continue;
    }

    /* main.go:822-830 */
    if (int_pos == 0) {
      if (!go_atoi(pos, npos, &int_pos)) {
        log_site(log, chrom, nchrom, pos, npos, " %s", "Invalid POS");
        break;
      }
    }

    if (nt == 1) { /* main.go:832-847 */
      if (t[0] != ref[0]) {
        log_site(log, chrom, nchrom, pos, npos, " ALT#%d %s", alt_idx + 1,
                   "1st base REF != ALT");
        continue;
      }
      alleles_push(out, dup_ll(int_pos + 1), ref[1], dup_ll(1 - (long long)nref), alt_idx);
      continue;
    }

    if (nref == nt) { /* main.go:855-873 */
      for (size_t k = 0; k < nref; k++) {
        if (ref[k] != t[k]) alleles_push(out, dup_ll(int_pos + (long long)k), ref[k], dup_n(t + k, 1), alt_idx);
      }
      continue;
    }

    if (nt > nref) { /* main.go:899-958 */
      long long r = 0;
      const long long lt = (long long)nt, lr = (long long)nref;
      while (lt + r > 0 && lr + r > 1 && t[lt + r - 1] == ref[lr + r - 1]) r--;
      const long long offset = lr + r;
      if (memcmp(ref, t, (size_t)offset) != 0) {
        log_site(log, chrom, nchrom, pos, npos, " ALT#%d %s", alt_idx + 1,
                   "Mixed indel/snp sites not supported");
        continue;
      }
      alleles_push(out, dup_ll(int_pos + offset - 1), ref[offset - 1],
                   plus_alt(t + offset, (size_t)(lt + r - offset)), alt_idx);
      continue;
    }

    { /* main.go:971-998 */
      long long r = 0;
      const long long lt = (long long)nt, lr = (long long)nref;
      while (lt + r > 1 && lr + r > 0 && t[lt + r - 1] == ref[lr + r - 1]) r--;
      const long long offset = lt + r;
      if (memcmp(ref, t, (size_t)offset) != 0) {
        log_site(log, chrom, nchrom, pos, npos, " ALT#%d %s", alt_idx + 1,
                   "Mixed indel/snp sites not supported");
        continue;
      }
      alleles_push(out, dup_ll(int_pos + offset), ref[offset], dup_ll(-(lr + r - offset)), alt_idx);
      continue;
    }
  }

  /* main.go:1004-1037 */
  if (out->n == 0) return;
  if (multi) {
    strcpy(out->site_type, "MULTIALLELIC");
    return;
  }
  if (strlen(out->alts[0]) > 1) {
    strcpy(out->site_type, out->alts[0][0] == '-' ? "DEL" : "INS");
    return;
  }
  strcpy(out->site_type, out->n > 1 ? "MNP" : "SNP");
}

/* parse.GetTrTv(ref, alt) as used at main.go:605.  Pinned on all 12 ACGT
 * pairs and on indel alts by the golden's column 6.  A REF base
 * outside ACGT is not pinned by any reference fixture; restated as "0". */
char orc_get_trtv(char ref, const char *alt, size_t nalt) {
  if (nalt != 1) return '0';
  const char a = alt[0];
  if (!is_actg(ref) || !is_actg(a)) return '0';
  if ((ref == 'A' && a == 'G') || (ref == 'G' && a == 'A') || (ref == 'C' && a == 'T') ||
      (ref == 'T' && a == 'C'))
    return '1';
  return '2';
}

/* ------------------------------------------------------------------ makeHetHomozygotes */

static const char *mem_chr(const char *s, size_t n, char c) { return (const char *)memchr(s, c, n); }

/* main.go:1042-1194 */
void orc_make_het_hom(const char *const *fields, const size_t *flen, int n_header,
                      const char *allele_num, uint8_t *cls_out, int8_t *dosage_out, int *ac,
                      int *an) {
  const size_t na = strlen(allele_num);
  int total_alt = 0, total_gt = 0;

  for (int i = 9; i < n_header; i++) { /* main.go:1057 */
    const char *g = fields[i];
    const size_t n = flen[i];
    uint8_t cls = ORC_CLS_NONE;
    int8_t dosage = 0;

    /* main.go:1063-1124: 3-byte diploid fast path */
    if ((n == 3 || (n > 3 && g[3] == ':')) && (g[1] == '|' || g[1] == '/')) {
      if (g[0] == '0' && g[2] == '0') {
        total_gt += 2;
        goto next_sample;
      }
      if (na == 1) {
        if ((g[0] == '0' && g[2] == allele_num[0]) || (g[0] == allele_num[0] && g[2] == '0')) {
          total_gt += 2;
          total_alt += 1;
          cls = ORC_CLS_HET;
          dosage = 1;
          goto next_sample;
        }
        if (g[0] == allele_num[0] && g[2] == allele_num[0]) {
          total_gt += 2;
          total_alt += 2;
          cls = ORC_CLS_HOM;
          dosage = 2;
          goto next_sample;
        }
      }
      if (g[0] == '.' || g[2] == '.') {
        cls = ORC_CLS_MISSING;
        dosage = -1;
        goto next_sample;
      }
    }

    { /* main.go:1126-1190: general path */
      const char *colon = mem_chr(g, n, ':');
      const size_t nf = colon ? (size_t)(colon - g) : n; /* strings.SplitN(g, ":", 2)[0] */
      char sep = 0;
      if (mem_chr(g, nf, '|'))
        sep = '|';
      else if (mem_chr(g, nf, '/'))
        sep = '/';

      int alt_count = 0, gt_count = 0;
      size_t start = 0;
      for (size_t k = 0; k <= nf; k++) {
        if (k != nf && !(sep && g[k] == sep)) continue;
        const char *tok = g + start;
        const size_t ntok = k - start;
        start = k + 1;
        if (ntok == 1 && tok[0] == '.') { /* main.go:1150-1160 */
          cls = ORC_CLS_MISSING;
          dosage = -1;
          goto next_sample;
        }
        if (ntok == na && memcmp(tok, allele_num, na) == 0) alt_count++;
        gt_count++;
      }
      total_gt += gt_count;
      total_alt += alt_count;
      dosage = (int8_t)(alt_count <= 127 ? alt_count : 127);
      if (alt_count != 0) cls = alt_count == gt_count ? ORC_CLS_HOM : ORC_CLS_HET;
    }

  next_sample:
    if (cls_out) cls_out[i - 9] = cls;
    if (dosage_out) dosage_out[i - 9] = dosage;
  }
  *ac = total_alt;
  *an = total_gt;
}

/* ------------------------------------------------------------------ processLines */

typedef struct {
  const char **ptr;
  size_t *len;
  size_t n, cap;
} field_vec;

static void split_tabs(field_vec *fv, const char *s, size_t n) {
  /* strings.Split(row, "\t"), main.go:535 */
  fv->n = 0;
  size_t start = 0;
  for (size_t i = 0; i <= n; i++) {
    if (i != n && s[i] != '\t') continue;
    if (fv->n == fv->cap) {
      fv->cap = fv->cap ? fv->cap * 2 : 64;
      fv->ptr = (const char **)realloc((void *)fv->ptr, fv->cap * sizeof(char *));
      fv->len = (size_t *)realloc(fv->len, fv->cap * sizeof(size_t));
    }
    fv->ptr[fv->n] = s + start;
    fv->len[fv->n] = i - start;
    fv->n++;
    start = i + 1;
  }
}

typedef struct {
  const orc_config *cfg;
  filter_set allowed, excluded;
  char **header; /* normalised header fields */
  size_t *header_len;
  int n_header;
  int num_chars;
  orc_buf *dosage; /* optional: one "locus<TAB>d0,d1,...\n" text row per output row (main.go:576-584) */
} run_ctx;

/* join the names of samples whose class == want, main.go:617,639,653 */
static int join_names(orc_buf *out, const run_ctx *rc, const uint8_t *cls, uint8_t want) {
  int cnt = 0;
  const int ns = rc->n_header - 9;
  for (int s = 0; s < ns; s++) {
    if (cls[s] != want) continue;
    if (cnt) buf_puts(out, rc->cfg->field_delimiter);
    buf_write(out, rc->header[9 + s], rc->header_len[9 + s]);
    cnt++;
  }
  return cnt;
}

static int count_cls(const uint8_t *cls, int ns, uint8_t want) {
  int c = 0;
  for (int s = 0; s < ns; s++) c += cls[s] == want;
  return c;
}

/* the body of the `for _, row := range lines` loop, main.go:534-698 */
static void process_line(const run_ctx *rc, const char *row, size_t nrow, field_vec *fv,
                         uint8_t *cls, orc_buf *out, orc_buf *log) {
  const orc_config *cfg = rc->cfg;
  if (nrow < (size_t)rc->num_chars) return; /* Go would panic on the slice; skip */
  split_tabs(fv, row, nrow - (size_t)rc->num_chars);

  const char *filter = fv->n > 6 ? fv->ptr[6] : "";
  const size_t nfilter = fv->n > 6 ? fv->len[6] : 0;
  if (!line_passes(fv->n, (size_t)rc->n_header, filter, nfilter, &rc->allowed, &rc->excluded))
    return;

  orc_alleles al;
  orc_get_alleles(fv->ptr[0], fv->len[0], fv->ptr[1], fv->len[1], fv->ptr[3], fv->len[3],
                  fv->ptr[4], fv->len[4], &al, log);
  if (al.n == 0) {
    orc_alleles_free(&al);
    return;
  }

  const int multiallelic = strcmp(al.site_type, "MULTIALLELIC") == 0;
  const int ns = rc->n_header > 9 ? rc->n_header - 9 : 0;
  const double num_samples = (double)ns;

  for (int i = 0; i < al.n; i++) {
    int ac = 0, an = 0, n_het = 0, n_hom = 0, n_miss = 0;
    double effective = 0;
    if (ns > 0) { /* main.go:555-564 */
      char strAlt[16];
      snprintf(strAlt, sizeof strAlt, "%d", al.alt_indices[i] + 1);
      int8_t *dos = rc->dosage ? (int8_t *)malloc((size_t)ns) : NULL;
      orc_make_het_hom(fv->ptr, fv->len, rc->n_header, strAlt, cls, dos, &ac, &an);
      if (ac == 0) {
        free(dos);
        continue;
      }
      if (dos) { /* the Arrow row, main.go:576-584: "chrom:pos:ref:alt" then one int8 per sample */
        orc_buf *d = rc->dosage;
        if (fv->len[0] < 4 || fv->ptr[0][0] != 'c') buf_puts(d, "chr");
        buf_write(d, fv->ptr[0], fv->len[0]);
        buf_putc(d, ':');
        buf_puts(d, al.positions[i]);
        buf_putc(d, ':');
        buf_putc(d, al.refs[i]);
        buf_putc(d, ':');
        buf_puts(d, al.alts[i]);
        buf_putc(d, '\t');
        for (int k = 0; k < ns; k++) {
          if (k) buf_putc(d, ',');
          buf_itoa(d, dos[k]);
        }
        buf_putc(d, '\n');
        free(dos);
      }
      n_het = count_cls(cls, ns, ORC_CLS_HET);
      n_hom = count_cls(cls, ns, ORC_CLS_HOM);
      n_miss = count_cls(cls, ns, ORC_CLS_MISSING);
      effective = num_samples - (double)n_miss;
    }

    /* main.go:570-574 */
    if (fv->len[0] < 4 || fv->ptr[0][0] != 'c') buf_puts(out, "chr");
    buf_write(out, fv->ptr[0], fv->len[0]);
    buf_putc(out, '\t');
    buf_puts(out, al.positions[i]);
    buf_putc(out, '\t');
    buf_puts(out, al.site_type);
    buf_putc(out, '\t');
    buf_putc(out, al.refs[i]);
    buf_putc(out, '\t');
    buf_puts(out, al.alts[i]);
    buf_putc(out, '\t');
    /* main.go:602-606 */
    buf_putc(out, multiallelic ? '0' : orc_get_trtv(al.refs[i], al.alts[i], strlen(al.alts[i])));
    buf_putc(out, '\t');

    /* main.go:612-656 */
    if (n_het == 0) {
      buf_puts(out, cfg->empty_field);
      buf_puts(out, "\t0");
    } else {
      join_names(out, rc, cls, ORC_CLS_HET);
      buf_putc(out, '\t');
      buf_float_g3(out, (double)n_het / effective);
    }
    buf_putc(out, '\t');
    if (n_hom == 0) {
      buf_puts(out, cfg->empty_field);
      buf_puts(out, "\t0");
    } else {
      join_names(out, rc, cls, ORC_CLS_HOM);
      buf_putc(out, '\t');
      buf_float_g3(out, (double)n_hom / effective);
    }
    buf_putc(out, '\t');
    if (n_miss == 0) {
      buf_puts(out, cfg->empty_field);
      buf_puts(out, "\t0");
    } else {
      join_names(out, rc, cls, ORC_CLS_MISSING);
      buf_putc(out, '\t');
      buf_float_g3(out, (double)n_miss / num_samples);
    }

    /* main.go:659-671 */
    buf_putc(out, '\t');
    buf_itoa(out, ac);
    buf_putc(out, '\t');
    buf_itoa(out, an);
    buf_putc(out, '\t');
    if (ac == 0)
      buf_putc(out, '0');
    else
      buf_float_g3(out, (double)ac / (double)an);

    /* main.go:674-692 */
    if (cfg->keep_pos) {
      buf_putc(out, '\t');
      buf_write(out, fv->ptr[1], fv->len[1]);
    }
    if (cfg->keep_id) {
      buf_putc(out, '\t');
      buf_write(out, fv->ptr[2], fv->len[2]);
    }
    if (cfg->keep_info) {
      buf_putc(out, '\t');
      buf_itoa(out, al.alt_indices[i]);
      buf_putc(out, '\t');
      buf_write(out, fv->ptr[7], fv->len[7]);
    }
    buf_putc(out, '\n');
  }
  orc_alleles_free(&al);
}

/* ------------------------------------------------------------------ readVcf */

typedef struct {
  const char *p;
  size_t n;
} line_ref;

typedef struct {
  const run_ctx *rc;
  const line_ref *lines;
  size_t n_lines;
  size_t n_batches;
  size_t next_batch; /* guarded by mu */
  pthread_mutex_t mu;
  orc_buf *batch_out; /* one per batch, stitched in order afterwards */
  orc_buf *batch_log;
} pool;

#define ORC_BATCH 64 /* maxCapacity, main.go:349 */

static void *worker(void *arg) {
  pool *pl = (pool *)arg;
  field_vec fv = {0};
  const int ns = pl->rc->n_header > 9 ? pl->rc->n_header - 9 : 0;
  uint8_t *cls = (uint8_t *)malloc((size_t)(ns ? ns : 1));
  for (;;) {
    pthread_mutex_lock(&pl->mu);
    size_t b = pl->next_batch++;
    pthread_mutex_unlock(&pl->mu);
    if (b >= pl->n_batches) break;
    size_t lo = b * ORC_BATCH, hi = lo + ORC_BATCH;
    if (hi > pl->n_lines) hi = pl->n_lines;
    for (size_t i = lo; i < hi; i++)
      process_line(pl->rc, pl->lines[i].p, pl->lines[i].n, &fv, cls, &pl->batch_out[b],
                   &pl->batch_log[b]);
  }
  free(cls);
  free((void *)fv.ptr);
  free(fv.len);
  return NULL;
}

static int fatal(orc_buf *err, const char *msg) {
  buf_puts(err, msg);
  buf_putc(err, '\n');
  return 1;
}

/* main.go:241-396 */
static int read_vcf_impl(const orc_config *cfg, const char *in, size_t n_in, orc_buf *out, orc_buf *err,
                         uint64_t *n_rows_in, orc_buf *dosage) {
  if (n_rows_in) *n_rows_in = 0;
  /* parse.FindEndOfLine(reader, ""): consume the first line, learn the terminator */
  size_t i = 0;
  char eol = '\n';
  int num_chars = 1;
  for (;; i++) {
    if (i >= n_in) return fatal(err, "EOF");
    if (in[i] == '\n') break;
    if (in[i] == '\r') {
      if (i + 1 >= n_in) return fatal(err, "EOF");
      if (in[i + 1] == '\n') {
        num_chars = 2;
      } else {
        eol = '\r';
      }
      break;
    }
  }
  const size_t vlen = i; /* version line without terminator */
  size_t pos = i + (size_t)num_chars;

  /* main.go:256-264: regexp.MatchString("##fileformat=VCFv4", versionLine) */
  if (!memmem(in, vlen, "##fileformat=VCFv4", 18)) return fatal(err, "Not a VCF file");

  /* main.go:266-294 */
  const char *hdr = NULL;
  size_t hdr_len = 0;
  while (pos < n_in) {
    const char *e = (const char *)memchr(in + pos, eol, n_in - pos);
    if (!e) break; /* io.EOF before a terminator */
    size_t row_len = (size_t)(e - (in + pos)) + 1;
    const char *row = in + pos;
    pos += row_len;
    if (row_len < (size_t)num_chars) continue;
    size_t body = row_len - (size_t)num_chars;
    const char *tab = (const char *)memchr(row, '\t', body);
    size_t f0 = tab ? (size_t)(tab - row) : body;
    if (f0 == 6 && memcmp(row, "#CHROM", 6) == 0) {
      hdr = row;
      hdr_len = body;
      break;
    }
  }
  if (!hdr) return fatal(err, "No header found");

  run_ctx rc;
  memset(&rc, 0, sizeof rc);
  rc.cfg = cfg;
  rc.num_chars = num_chars;
  rc.dosage = dosage;
  filter_set_parse(&rc.allowed, cfg->allow_filter, 1);
  filter_set_parse(&rc.excluded, cfg->exclude_filter, 0);

  field_vec hv = {0};
  split_tabs(&hv, hdr, hdr_len);
  rc.n_header = (int)hv.n;
  rc.header = (char **)calloc(hv.n, sizeof(char *));
  rc.header_len = (size_t *)calloc(hv.n, sizeof(size_t));
  for (size_t k = 0; k < hv.n; k++) {
    rc.header[k] = dup_n(hv.ptr[k], hv.len[k]);
    rc.header_len[k] = hv.len[k];
    /* parse.NormalizeHeader(header), main.go:296 (unpinned restatement) */
    if (cfg->normalize_header)
      for (size_t c = 0; c < hv.len[k]; c++)
        if (rc.header[k][c] == '.') rc.header[k][c] = '_';
  }
  free((void *)hv.ptr);
  free(hv.len);

  int rv = 0;
  if (rc.n_header < 8) {
    /* the reference indexes record[7] / record[6] unguarded: out of contract */
    rv = fatal(err, "Malformed header: fewer than 8 fields");
    goto done;
  }
  if (rc.n_header == 9) /* main.go:507-509 */
    log_printf(err, "Found 9 header fields. When genotypes present, we expect 1+ samples after FORMAT (10 fields minimum)");

  { /* main.go:349-380: deliver terminator-delimited lines; drop an unterminated tail */
    size_t cap = 1024, nl = 0;
    line_ref *lines = (line_ref *)malloc(cap * sizeof(line_ref));
    while (pos < n_in) {
      const char *e = (const char *)memchr(in + pos, eol, n_in - pos);
      if (!e) break;
      size_t row_len = (size_t)(e - (in + pos)) + 1;
      if (nl == cap) {
        cap *= 2;
        lines = (line_ref *)realloc(lines, cap * sizeof(line_ref));
      }
      lines[nl].p = in + pos;
      lines[nl].n = row_len;
      nl++;
      pos += row_len;
    }
    if (n_rows_in) *n_rows_in = nl;

    pool pl;
    memset(&pl, 0, sizeof pl);
    pl.rc = &rc;
    pl.lines = lines;
    pl.n_lines = nl;
    pl.n_batches = (nl + ORC_BATCH - 1) / ORC_BATCH;
    pl.batch_out = (orc_buf *)calloc(pl.n_batches ? pl.n_batches : 1, sizeof(orc_buf));
    pl.batch_log = (orc_buf *)calloc(pl.n_batches ? pl.n_batches : 1, sizeof(orc_buf));
    pthread_mutex_init(&pl.mu, NULL);

    int nt = cfg->n_threads > 1 && !dosage ? cfg->n_threads : 1; /* one dosage sink: serial */
    if (nt == 1) {
      worker(&pl);
    } else {
      pthread_t *th = (pthread_t *)calloc((size_t)nt, sizeof(pthread_t));
      for (int t = 0; t < nt; t++) pthread_create(&th[t], NULL, worker, &pl);
      for (int t = 0; t < nt; t++) pthread_join(th[t], NULL);
      free(th);
    }
    for (size_t b = 0; b < pl.n_batches; b++) {
      if (pl.batch_out[b].len) buf_write(out, pl.batch_out[b].data, pl.batch_out[b].len);
      if (pl.batch_log[b].len) buf_write(err, pl.batch_log[b].data, pl.batch_log[b].len);
      orc_buf_free(&pl.batch_out[b]);
      orc_buf_free(&pl.batch_log[b]);
    }
    free(pl.batch_out);
    free(pl.batch_log);
    pthread_mutex_destroy(&pl.mu);
    free(lines);
  }

done:
  for (int k = 0; k < rc.n_header; k++) free(rc.header[k]);
  free(rc.header);
  free(rc.header_len);
  filter_set_free(&rc.allowed);
  filter_set_free(&rc.excluded);
  return rv;
}

/* ------------------------------------------------------------------ flat wrappers */

size_t orc_get_alleles_flat(const char *chrom, const char *pos, const char *ref, const char *alt,
                            char *out, size_t out_cap, char *log, size_t log_cap) {
  orc_alleles al;
  orc_buf lg, o;
  orc_buf_init(&lg);
  orc_buf_init(&o);
  orc_get_alleles(chrom, strlen(chrom), pos, strlen(pos), ref, strlen(ref), alt, strlen(alt), &al,
                  &lg);
  buf_puts(&o, al.site_type);
  buf_putc(&o, '\n');
  for (int i = 0; i < al.n; i++) {
    buf_puts(&o, al.positions[i]);
    buf_putc(&o, '\t');
    buf_putc(&o, al.refs[i]);
    buf_putc(&o, '\t');
    buf_puts(&o, al.alts[i]);
    buf_putc(&o, '\t');
    buf_itoa(&o, al.alt_indices[i]);
    buf_putc(&o, '\n');
  }
  size_t n = o.len < out_cap ? o.len : (out_cap ? out_cap - 1 : 0);
  if (out_cap) {
    memcpy(out, o.data ? o.data : "", n);
    out[n] = 0;
  }
  if (log_cap) {
    size_t m = lg.len < log_cap ? lg.len : log_cap - 1;
    memcpy(log, lg.data ? lg.data : "", m);
    log[m] = 0;
  }
  orc_alleles_free(&al);
  orc_buf_free(&lg);
  orc_buf_free(&o);
  return n;
}

int orc_make_het_hom_flat(const char *line, size_t n, int n_header, const char *allele_num,
                          uint8_t *cls_out, int8_t *dosage_out, int *ac, int *an) {
  field_vec fv = {0};
  split_tabs(&fv, line, n);
  int rv = 0;
  if ((int)fv.n < n_header) {
    rv = -1;
  } else {
    orc_make_het_hom(fv.ptr, fv.len, n_header, allele_num, cls_out, dosage_out, ac, an);
  }
  free((void *)fv.ptr);
  free(fv.len);
  return rv;
}

int orc_read_vcf(const orc_config *cfg, const char *in, size_t n_in, orc_buf *out, orc_buf *err,
                 uint64_t *n_rows_in) {
  return read_vcf_impl(cfg, in, n_in, out, err, n_rows_in, NULL);
}

/* the rows of --dosageOutput as text (test infrastructure for the Arrow writer's parity test) */
int orc_run_dosage(const orc_config *cfg, const char *in, size_t n_in, char **dos, size_t *n_dos) {
  orc_buf o, e, d;
  orc_buf_init(&o);
  orc_buf_init(&e);
  orc_buf_init(&d);
  int rv = read_vcf_impl(cfg, in, n_in, &o, &e, NULL, &d);
  orc_buf_free(&o);
  orc_buf_free(&e);
  buf_reserve(&d, 0);
  d.data[d.len] = 0;
  *dos = d.data;
  *n_dos = d.len;
  return rv;
}

int orc_run(const orc_config *cfg, const char *in, size_t n_in, char **out, size_t *n_out,
            char **err, size_t *n_err, uint64_t *n_rows_in) {
  orc_buf o, e;
  orc_buf_init(&o);
  orc_buf_init(&e);
  int rv = orc_read_vcf(cfg, in, n_in, &o, &e, n_rows_in);
  buf_reserve(&o, 0);
  buf_reserve(&e, 0);
  o.data[o.len] = 0;
  e.data[e.len] = 0;
  *out = o.data;
  *n_out = o.len;
  *err = e.data;
  *n_err = e.len;
  return rv;
}

void orc_free(void *p) { free(p); }

/* ------------------------------------------------------------------ CLI */
#ifdef ORC_MAIN
/* stdin -> stdout driver used to pin the oracle against the golden file and as
 * the "reference CPU path" timed by bench.py.  Flags as main.go:84-99. */
static int flag_bool(const char *arg, const char *name, int *val) {
  const char *a = arg;
  if (*a == '-') a++;
  if (*a == '-') a++;
  size_t n = strlen(name);
  if (strncmp(a, name, n) != 0) return 0;
  if (a[n] == 0) {
    *val = 1;
    return 1;
  }
  if (a[n] == '=') {
    *val = strcmp(a + n + 1, "false") != 0 && strcmp(a + n + 1, "0") != 0;
    return 1;
  }
  return 0;
}

static int flag_str(int argc, char **argv, int *i, const char *name, const char **val) {
  const char *a = argv[*i];
  if (*a == '-') a++;
  if (*a == '-') a++;
  size_t n = strlen(name);
  if (strncmp(a, name, n) != 0) return 0;
  if (a[n] == '=') {
    *val = a + n + 1;
    return 1;
  }
  if (a[n] == 0 && *i + 1 < argc) {
    *val = argv[++*i];
    return 1;
  }
  return 0;
}

int main(int argc, char **argv) {
  orc_config cfg;
  orc_config_defaults(&cfg);
  const char *in_path = NULL, *threads = NULL, *raw = NULL;
  int timing = 0; /* --timing: one "[oracle timing] ..." line on stderr (bench.py's cpu_baseline leg) */
  for (int i = 1; i < argc; i++) {
    if (flag_bool(argv[i], "keepId", &cfg.keep_id) || flag_bool(argv[i], "keepInfo", &cfg.keep_info) ||
        flag_bool(argv[i], "keepPos", &cfg.keep_pos) || flag_bool(argv[i], "timing", &timing))
      continue;
    if (flag_str(argc, argv, &i, "emptyField", &cfg.empty_field) ||
        flag_str(argc, argv, &i, "fieldDelimiter", &cfg.field_delimiter) ||
        flag_str(argc, argv, &i, "allowFilter", &cfg.allow_filter) ||
        flag_str(argc, argv, &i, "excludeFilter", &cfg.exclude_filter) ||
        flag_str(argc, argv, &i, "in", &in_path) || flag_str(argc, argv, &i, "threads", &threads) ||
        flag_str(argc, argv, &i, "normalizeHeader", &raw))
      continue;
    fprintf(stderr, "flag provided but not defined: %s\n", argv[i]);
    return 2;
  }
  if (threads) cfg.n_threads = atoi(threads);
  if (raw) cfg.normalize_header = atoi(raw);

  FILE *f = in_path ? fopen(in_path, "rb") : stdin;
  if (!f) {
    perror(in_path);
    return 1;
  }
  struct timespec ts0, ts1, ts2, ts3;
  clock_gettime(CLOCK_MONOTONIC, &ts0);
  orc_buf in;
  orc_buf_init(&in);
  /* a regular file is mapped, not copied (bench.py's file is 63 GB in /dev/shm: a second copy of it is 15 s and as much
   * memory again); MAP_POPULATE so that the page-table work stays in the "read" figure, not in "process" */
  char *mapped = NULL;
  size_t mapped_len = 0;
  {
    struct stat st;
    if (in_path && fstat(fileno(f), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
      void *m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fileno(f), 0);
      if (m != MAP_FAILED) {
        mapped = (char *)m;
        mapped_len = (size_t)st.st_size;
      }
    }
  }
  while (!mapped) {
    buf_reserve(&in, 1 << 24);
    size_t r = fread(in.data + in.len, 1, 1 << 24, f);
    if (!r) break;
    in.len += r;
  }
  orc_buf out, err;
  orc_buf_init(&out);
  orc_buf_init(&err);
  orc_string_header(&cfg, &out);
  buf_putc(&out, '\n');
  uint64_t n_rows = 0;
  clock_gettime(CLOCK_MONOTONIC, &ts1);
  int rv = mapped ? orc_read_vcf(&cfg, mapped, mapped_len, &out, &err, &n_rows)
                  : orc_read_vcf(&cfg, in.data ? in.data : "", in.len, &out, &err, &n_rows);
  clock_gettime(CLOCK_MONOTONIC, &ts2);
  fwrite(out.data, 1, out.len, stdout);
  if (err.len) fwrite(err.data, 1, err.len, stderr);
  clock_gettime(CLOCK_MONOTONIC, &ts3);
  if (timing) {
    /* read = input into memory; process = readVcf (line delivery + the workers, main.go:241-396); write = output */
    fprintf(stderr, "[oracle timing] rows %llu threads %d read %.3f process %.3f write %.3f s\n", (unsigned long long)n_rows,
            cfg.n_threads > 1 ? cfg.n_threads : 1,
            (double)(ts1.tv_sec - ts0.tv_sec) + 1e-9 * (double)(ts1.tv_nsec - ts0.tv_nsec),
            (double)(ts2.tv_sec - ts1.tv_sec) + 1e-9 * (double)(ts2.tv_nsec - ts1.tv_nsec),
            (double)(ts3.tv_sec - ts2.tv_sec) + 1e-9 * (double)(ts3.tv_nsec - ts2.tv_nsec));
  }
  if (mapped) munmap(mapped, mapped_len);
  orc_buf_free(&in);
  orc_buf_free(&out);
  orc_buf_free(&err);
  if (f != stdin) fclose(f);
  return rv;
}
#endif
