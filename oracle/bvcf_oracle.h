/*
 * bvcf_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of bystro-vcf's per-line variant pipeline
 * (reference: /root/reference/main.go @ 2024_10_08), used only as the
 * checker for the HIP path: tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may call it; nothing under bystro-vcf_amd/ may.
 *
 * Pinning: the restatement reproduces the reference's own golden output
 * (previous_out_check/out_check_new_10_3_18.vcf.gz, 19 821 rows) byte for
 * byte after sort, and every known-answer table of main_test.go
 * (tests/golden/known_answers.json).  See oracle/README.md.
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef BVCF_ORACLE_H
#define BVCF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* growable byte buffer */
typedef struct {
  char *data;
  size_t len;
  size_t cap;
} orc_buf;

void orc_buf_init(orc_buf *b);
void orc_buf_free(orc_buf *b);

/* mirrors main.go:63-80 `Config` (only the fields the path reads) */
typedef struct {
  const char *empty_field;     /* --emptyField,     default "!"  (main.go:91) */
  const char *field_delimiter; /* --fieldDelimiter, default ";"  (main.go:92) */
  int keep_id;                 /* --keepId   (main.go:93) */
  int keep_info;               /* --keepInfo (main.go:96) */
  int keep_pos;                /* --keepPos  (main.go:95) */
  /* raw flag text; NULL, "" or "*" => allow all (main.go:108-114) */
  const char *allow_filter;
  /* raw flag text; NULL or "" => exclude none (main.go:117-123) */
  const char *exclude_filter;
  int n_threads;               /* worker pool size; <=1 => serial (main.go:345-347) */
  int normalize_header;        /* parse.NormalizeHeader restatement: '.' -> '_' (unpinned) */
} orc_config;

void orc_config_defaults(orc_config *c); /* setup() defaults, main.go:84-99 */

/* class codes shared with the HIP path's 2-bit class map */
enum { ORC_CLS_NONE = 0, ORC_CLS_HET = 1, ORC_CLS_HOM = 2, ORC_CLS_MISSING = 3 };

#define ORC_MAX_ALLELES 4096

/* getAlleles result, main.go:723 */
typedef struct {
  char site_type[16];          /* "", SNP, INS, DEL, MNP, MULTIALLELIC */
  int n;                       /* number of output alleles */
  /* per output allele */
  char **positions;            /* decimal text */
  char *refs;                  /* one byte each */
  char **alts;                 /* "X", "+XXX", "-N" */
  int *alt_indices;            /* 0-based VCF ALT index */
} orc_alleles;

void orc_alleles_free(orc_alleles *a);

/* main.go:456-474 */
int orc_alt_is_valid(const char *alt, size_t n);

/* main.go:723-1038; messages the reference logs go to `log` (one line each) */
void orc_get_alleles(const char *chrom, size_t nchrom, const char *pos, size_t npos,
                     const char *ref, size_t nref, const char *alt, size_t nalt,
                     orc_alleles *out, orc_buf *log);

/* third-party parse.GetTrTv restated from call site main.go:605 + golden col 6 */
char orc_get_trtv(char ref, const char *alt, size_t nalt);

/*
 * main.go:1042-1194.  fields/flen: the split record; n_header: len(header).
 * cls_out[n_header-9] receives ORC_CLS_* per sample, dosage_out likewise
 * (either may be NULL).  Returns ac/an through pointers.
 */
void orc_make_het_hom(const char *const *fields, const size_t *flen, int n_header,
                      const char *allele_num, uint8_t *cls_out, int8_t *dosage_out,
                      int *ac, int *an);

/*
 * readVcf + processLines, main.go:241-396,476-721, on an in-memory input.
 * Output rows are appended to `out` in INPUT ORDER (the reference's order with
 * one worker); the header line of main.go:199 is NOT written (see
 * orc_string_header).  Returns 0, or 1 for the reference's log.Fatal paths
 * (message in `err`).  If n_rows_in is non-NULL it receives the number of data
 * lines delivered to the workers.
 */
int orc_read_vcf(const orc_config *cfg, const char *in, size_t n_in, orc_buf *out, orc_buf *err,
                 uint64_t *n_rows_in);

/* stringHeader(config), main.go:219-239, without trailing newline */
void orc_string_header(const orc_config *cfg, orc_buf *out);

/* ---- flat wrappers for ctypes-driven tests ---- */

/* text result: "TYPE\n" then one "pos\tref\talt\tidx\n" per allele; returns bytes written */
size_t orc_get_alleles_flat(const char *chrom, const char *pos, const char *ref, const char *alt,
                            char *out, size_t out_cap, char *log, size_t log_cap);

/* line = tab-joined record (no terminator) */
int orc_make_het_hom_flat(const char *line, size_t n, int n_header, const char *allele_num,
                          uint8_t *cls_out, int8_t *dosage_out, int *ac, int *an);

/* whole-file convenience: returns malloc'd output (caller frees with orc_free) */
int orc_run(const orc_config *cfg, const char *in, size_t n_in, char **out, size_t *n_out,
            char **err, size_t *n_err, uint64_t *n_rows_in);
void orc_free(void *p);

/* the rows the reference would hand to its Arrow writer (--dosageOutput, main.go:576-584), as text:
 * "chrom:pos:ref:alt<TAB>d0,d1,...\n" per output row, in input order; malloc'd, free with orc_free */
int orc_run_dosage(const orc_config *cfg, const char *in, size_t n_in, char **dos, size_t *n_dos);

#ifdef __cplusplus
}
#endif
#endif
