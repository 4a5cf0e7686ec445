/*
 * bvcf.h — C-ABI of libbvcf.so: the MI355X-native replacement for bystro-vcf's
 * per-line variant pipeline (reference /root/reference/main.go @ 2024_10_08).
 *
 * The reference has no FFI boundary; the path is the goroutine body
 *     processLines(header, numChars, config, queue, writer, complete, arrowWriter)   main.go:476-477
 * and the pure functions it calls:
 *     linePasses(record, header, allowed, excluded)            main.go:447-454
 *     altIsValid(alt)                                          main.go:456-474
 *     getAlleles(chrom, pos, ref, alt)                         main.go:723-1038
 *     makeHetHomozygotes(fields, header, alleleNum, ...)       main.go:1042-1194
 *     parse.GetTrTv(ref, alt)                                  main.go:602-606
 * fed by readVcf's 64-line batches (main.go:349-380).  A cgo caller replaces
 * `workQueue <- buff` with bvcf_submit() and the body of processLines with
 * bvcf_collect() + its own TSV assembly (or bvcf_format_tsv()); see INTEGRATION.md.
 *
 * Everything is plain C: pointers, sizes, fixed-width integers.  No callbacks,
 * no exceptions, nothing aborts: every function returns 0 or a negative
 * bvcf_status.  A ctx is single-caller; distinct ctxs (one per GPU) are
 * independent.  There is NO CPU fallback: without a HIP device bvcf_create fails.
 */
#ifndef BVCF_H
#define BVCF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BVCF_ABI_VERSION 7

typedef enum {
  BVCF_OK = 0,
  BVCF_E_ARG = -1,        /* bad argument */
  BVCF_E_HIP = -2,        /* HIP runtime error (bvcf_last_error has the text) */
  BVCF_E_NODEV = -3,      /* no usable HIP device */
  BVCF_E_BUSY = -4,       /* all slots in flight: collect first */
  BVCF_E_EMPTY = -5,      /* nothing to collect */
  BVCF_E_TOO_BIG = -6,    /* block larger than max_batch_bytes */
  BVCF_E_CAPACITY = -7,   /* batch needs more lines/alleles/class-map bytes than reserved;
                             bvcf_result.need_* say how many; grow with bvcf_reserve or split */
  BVCF_E_NOMEM = -8,
  BVCF_E_FATAL = -9,      /* the reference's log.Fatal paths (bvcf_run_*) */
  BVCF_E_IO = -10         /* file could not be opened / written */
} bvcf_status;

/* line verdicts, cf. main.go:537-545 */
enum {
  BVCF_LINE_OK = 0,        /* passed the gate and produced >= 1 allele */
  BVCF_LINE_FIELDS = 1,    /* len(record) != len(header)            (linePasses) */
  BVCF_LINE_FILTER = 2,    /* FILTER not allowed / excluded         (linePasses) */
  BVCF_LINE_NOALLELE = 3   /* getAlleles returned no alleles (reasons in the error list) */
};

/* parse.Snp / Ins / Del / Mnp / Multi, main.go:743,764,1013-1037 */
enum { BVCF_SITE_SNP = 0, BVCF_SITE_INS = 1, BVCF_SITE_DEL = 2, BVCF_SITE_MNP = 3, BVCF_SITE_MULTI = 4 };

/* how bvcf_allele describes the output `alt` text */
enum {
  BVCF_ALT_BASE = 0,  /* one base: alt_base */
  BVCF_ALT_INS = 1,   /* "+" followed by block[alt_off .. alt_off+alt_len) */
  BVCF_ALT_DEL = 2    /* "-" followed by decimal alt_len */
};

/* 2-bit per-sample classes of the class map (sample s: byte s/4, bits 2*(s%4)) */
enum { BVCF_CLS_NONE = 0, BVCF_CLS_HET = 1, BVCF_CLS_HOM = 2, BVCF_CLS_MISSING = 3 };

/* reasons getAlleles logs, main.go:42-50 with the formats of main.go:730-986 */
enum {
  BVCF_ERR_SAME = 1,        /* "%s:%s : REF == ALT"                              main.go:730 */
  BVCF_ERR_BAD_ALT1 = 2,    /* "%s:%s ALT #1 ALT not ACTG"   (single-ALT path)   main.go:737 */
  BVCF_ERR_DEL1_1 = 3,      /* "%s:%s ALT #1 1st base REF != ALT"                main.go:748 */
  BVCF_ERR_POS1 = 4,        /* "%s:%s ALT #1 Invalid POS"                        main.go:755 */
  BVCF_ERR_BAD_ALT = 5,     /* "%s:%s ALT #%d ALT not ACTG"                      main.go:782 */
  BVCF_ERR_INS1 = 6,        /* "%s:%s ALT #%d 1st base ALT != REF"               main.go:798 */
  BVCF_ERR_POS = 7,         /* "%s:%s Invalid POS"                               main.go:827 */
  BVCF_ERR_DEL1 = 8,        /* "%s:%s ALT#%d 1st base REF != ALT"                main.go:835 */
  BVCF_ERR_MIXED = 9,       /* "%s:%s ALT#%d Mixed indel/snp sites not supported" main.go:934,986 */
  BVCF_ERR_EMPTY_REF = 10   /* empty REF (a Go panic in the reference; rejected here) */
};

#define BVCF_ALLELE_POS_TEXT 1u /* bvcf_allele.flags: output pos is the POS field verbatim */
/* bvcf_allele.flags: the class map at cmap_off is a short list, not the 2-bit map: uint32 n (<= BVCF_CMAP_SPARSE_MAX),
 * then n entries (map byte index << 8 | map byte) in ascending order; every map byte that is not listed is 0.
 * The streaming path writes this form for alleles that few samples carry (most of a cohort file). */
#define BVCF_ALLELE_CMAP_SPARSE 2u
#define BVCF_CMAP_SPARSE_MAX 15u
/* from this many samples up, path 0 / 1 split the regular genotype scan of one line over several waves */
#define BVCF_WIDE_SAMPLES 32768u
#define BVCF_NO_CMAP 0xFFFFFFFFu
#define BVCF_DEVICE_PAD 64      /* bytes a device-resident block must own past nbytes */

typedef struct bvcf_ctx bvcf_ctx;

/* ctx configuration: what processLines closes over (main.go:494-509) */
typedef struct {
  uint32_t abi_version;       /* BVCF_ABI_VERSION */
  int32_t device;             /* HIP device ordinal */
  uint32_t n_header_fields;   /* len(header) incl. the 9 fixed columns (main.go:449,505) */
  uint32_t eol_chars;         /* numChars: 1 for "\n", 2 for "\r\n" (main.go:250,535) */
  uint8_t eol_byte;           /* endOfLineByte, '\n' unless the file uses lone '\r' */
  uint8_t want_class_maps;    /* needsLabels: emit the 2-bit class maps (main.go:502) */
  uint8_t want_dosage;        /* needsDosages: emit one int8 per sample per output allele (main.go:503,1069-1178) */
  uint8_t want_name_lists;    /* render the het / hom / missing sample-name lists on the device (bvcf_result.name_lists);
                                 takes effect once bvcf_set_sample_names has been called; needs want_class_maps */
  const char *allow_filter;   /* --allowFilter text; NULL, "" or "*" = allow all (main.go:98,108-114) */
  const char *exclude_filter; /* --excludeFilter text; NULL or "" = none (main.go:99,117-123) */
  uint64_t max_batch_bytes;   /* largest block bvcf_submit accepts (0 = 64 MiB; below 4 GiB - 1 MiB: offsets are 32-bit) */
  uint32_t max_lines;         /* 0 = derived from max_batch_bytes and n_header_fields */
  uint32_t max_alleles;       /* slots of alleles[] (>= max_lines); 0 = 2 * max_lines + 1024 */
  uint64_t cmap_bytes;        /* class-map arena, one map per (line, ALT index); 0 = 1.5 maps per line */
  uint32_t n_slots;           /* batches in flight (0 = 3: the short kernels that end two batches' chains then run beside the
                                 third's scan -- +5 % with 20 % multiallelic lines, +7 % on sites-only input, even elsewhere) */
  uint32_t path;              /* 0 = choose (streaming for 256 .. BVCF_WIDE_SAMPLES header fields, census otherwise);
                                 1 = census path (separate newline census, every line listed; from
                                 BVCF_WIDE_SAMPLES samples up the genotype scan of one line is split over waves);
                                 2 = streaming path when there are samples (lines found and ALT #1 scanned
                                 in one pass; only lines with the right field count are listed).  The ctx
                                 walks a batch with the kernel made for the shape of the previous batch's
                                 lines -- bare "x|y" sample fields, or fields with sub-fields beyond GT --
                                 the results do not depend on which;
                                 3 = as 2, and the caller knows that the sample fields carry more than GT (FORMAT
                                 "GT:DP:..."): the first batch already takes the kernel for such lines.  bvcf_submit
                                 finds that out from a host block by itself; a device-resident or BGZF first batch
                                 cannot be looked at before it is launched */
  uint32_t packed_sites;      /* files WITHOUT sample columns (n_header_fields <= 9): 1 = return the packed form of the
                                 batch -- one 32-byte bvcf_site per line (bvcf_result.sites) and full bvcf_line /
                                 bvcf_allele records only for the lines that need them (anything but a plain SNP).  A
                                 sites-only line is ~140 bytes of text; its full records are 128.  Ignored (sites == NULL)
                                 when the file has samples */
  uint32_t render_sites;      /* ABI 7, packed ctxs only: 1 = the TSV rows of the lines the packed form settles (a biallelic SNP of
                                 a file without samples: its row is the CHROM and POS bytes, REF, ALT, trTv and a constant
                                 tail, main.go:586-695,735-745) are rendered ON THE DEVICE, in input order, into
                                 bvcf_result.rows; the site records then stay on the device (sites == NULL) and the lines
                                 the host still has to format -- the BVCF_SITE_FULL ones -- are listed in row_cuts with the
                                 place of their rows in the stream.  Needs bvcf_set_row_format. */
} bvcf_params;

/* one input line; 64 bytes */
typedef struct {
  uint32_t off;        /* line start, bytes from block start */
  uint32_t len;        /* bytes without the terminator */
  uint32_t fend[9];    /* end (exclusive, relative to off) of fields 0..8; missing fields = len.
                          field i starts at (i ? fend[i-1]+1 : 0) */
  uint32_t rec_first;  /* where this line's 2nd.. output alleles start (see bvcf_result.alleles) */
  uint32_t n_rec;      /* number of output alleles (MNPs expand, rejected ALTs vanish) */
  uint32_t n_fields;   /* len(record) */
  uint32_t gt_task;    /* internal: first genotype-scan task of this line */
  uint8_t status;      /* BVCF_LINE_* */
  uint8_t site_type;   /* BVCF_SITE_* when status == OK */
  uint8_t pad[2];
} bvcf_line;

/* one output allele == one candidate TSV row (dropped by the caller when samples exist and ac == 0,
 * main.go:558-560); 64 bytes */
typedef struct {
  int64_t pos;         /* output position unless flags & BVCF_ALLELE_POS_TEXT */
  uint32_t line;       /* index into lines[] */
  uint32_t alt_idx;    /* 0-based VCF ALT index: the alleleIdx column (main.go:687) */
  uint32_t alt_off;    /* BVCF_ALT_INS: block offset of the inserted bases */
  uint32_t alt_len;    /* INS: inserted bases; DEL: deleted bases (the N of "-N"); BASE: 1 */
  uint32_t ac;         /* totalAltCount  (main.go:1170) */
  uint32_t an;         /* totalGtCount   (main.go:1169) */
  uint32_t n_het;      /* len(hets)      */
  uint32_t n_hom;      /* len(homs)      */
  uint32_t n_miss;     /* len(missing)   */
  uint32_t cmap_off;   /* byte offset of this allele's class map in cmap[], or BVCF_NO_CMAP */
  uint8_t ref;         /* refs[i] */
  uint8_t alt_base;    /* BVCF_ALT_BASE: the base */
  uint8_t kind;        /* BVCF_ALT_* */
  uint8_t site_type;   /* BVCF_SITE_* */
  uint8_t trtv;        /* 0 / 1 / 2 (main.go:602-606) */
  uint8_t flags;
  uint8_t pad[2];
  uint32_t gt_task;    /* internal: genotype-scan task that produced ac..n_miss */
  uint32_t pad2;
} bvcf_allele;

/* packed form of one line of a file without samples (bvcf_params.packed_sites); 32 bytes.  Without BVCF_SITE_FULL the
 * line is settled here: status is its verdict (BVCF_LINE_OK / FIELDS / FILTER) and, when OK, it is a biallelic SNP whose
 * single output allele is {pos = the POS field verbatim, ref, alt_base, trtv, alt_idx 0, type SNP} -- main.go:735-745;
 * every TAB that bounds a fixed column lies in the line's first 64 bytes.  With BVCF_SITE_FULL the line's records are
 * lines[full_idx] / alleles[full_idx] (+ alleles[lines[full_idx].rec_first ..]) as in the unpacked form. */
#define BVCF_SITE_FULL 0x80u
typedef struct {
  uint32_t off;        /* line start, bytes from block start */
  uint32_t len;        /* bytes without the terminator */
  uint8_t fend[8];     /* end (exclusive, relative to off) of fields 0..7; 0xFF = len (the line's last field, or missing) */
  uint8_t ref;
  uint8_t alt_base;
  uint8_t trtv;        /* 0 / 1 / 2 (main.go:602-606) */
  uint8_t status;      /* BVCF_LINE_OK / BVCF_LINE_FIELDS / BVCF_LINE_FILTER, or BVCF_SITE_FULL */
  uint32_t full_idx;   /* BVCF_SITE_FULL: index into lines[] (and of the line's first record in alleles[]) */
  uint32_t n_fields;   /* len(record) */
  uint32_t reserved;
} bvcf_site;

/* one message getAlleles would log; 16 bytes */
typedef struct {
  uint32_t line;
  uint32_t alt_no;     /* the %d of "ALT #%d" (1-based), 0 if the format has none */
  uint32_t code;       /* BVCF_ERR_* */
  uint32_t pad;
} bvcf_err;

/* the three strings.Join(names, fieldDelimiter) of one output allele (main.go:612-656), rendered on the device:
 * list q (0 heterozygotes, 1 homozygotes, 2 missingGenos) is names[off[q] .. off[q] + len[q]); len 0 = empty list
 * (the caller prints emptyField).  Valid for the alleles[] slots that hold a record with ac > 0. */
typedef struct {
  uint32_t off[3];
  uint32_t len[3];
} bvcf_names;

/* one line whose rows the host makes, and where they go in bvcf_result.rows; 24 bytes */
#define BVCF_NO_TEXT_OFF 0xFFFFFFFFu
typedef struct bvcf_row_cut {
  uint32_t line;     /* line number in the batch */
  uint32_t slot;     /* its records: lines[slot], alleles[slot] */
  uint64_t off;      /* byte offset in rows[] in front of which its rows belong */
  uint32_t text_off; /* bvcf_submit_bgzf batches: the line's bytes start at bvcf_result.text[text_off] -- only the lines of
                        the cuts come back, not the batch's whole text -- or BVCF_NO_TEXT_OFF: at text[lines[slot].off] (the
                        whole text came back), as in the block of a batch submitted as text */
  uint32_t reserved;
} bvcf_row_cut;

/* a collected batch.  All pointers are library-owned pinned host memory of the slot the batch ran in.  Collects fill
 * the ctx's n_slots slots in turn, so the pointers stay valid until the n_slots-th following bvcf_collect on the same
 * ctx (with n_slots = 1: the next one), or bvcf_reserve / bvcf_destroy. */
typedef struct {
  uint64_t batch_seq;
  int32_t status;            /* BVCF_OK or BVCF_E_CAPACITY (then only need_* are meaningful) */
  uint32_t n_lines;
  uint32_t n_alleles;
  uint32_t n_errs;
  uint64_t n_cmap_bytes;
  uint32_t cmap_stride;      /* bytes per class map: ceil(n_samples/4) rounded up to 16 */
  uint32_t n_samples;
  const bvcf_line *lines;
  /* output allele j of line i is alleles[i] for j == 0 and alleles[lines[i].rec_first + j - 1] after
   * that: slot i always belongs to line i (no allocation, no atomics for biallelic lines), further
   * alleles of multiallelic / MNP lines follow the first n_lines slots.  n_alleles is the array
   * length, not the number of valid records. */
  const bvcf_allele *alleles;
  const bvcf_err *errs;      /* unordered; filter by lines[e.line].status != FIELDS/FILTER is done on device */
  const uint8_t *cmap;
  uint64_t need_lines, need_alleles, need_cmap_bytes;
  float kernel_ms;           /* device time of the kernel chain for this batch (HIP events) */
  uint32_t reserved;
  uint64_t n_lines_seen;     /* terminated lines in the block (>= n_lines: the streaming path does not list
                                lines that fail len(record) == len(header), main.go:449) */
  /* bvcf_params.want_dosage: the row of alleles[k] is dosage[k * dosage_stride .. + n_samples):
   * the number of GT alleles equal to the ALT index, 127 at most, -1 if any allele is '.'
   * (main.go:1069-1178).  Rows of slots without a record are not written.  NULL otherwise. */
  const int8_t *dosage;
  uint32_t dosage_stride;    /* n_samples rounded up to 16 */
  uint32_t reserved2;
  /* bvcf_params.want_name_lists + bvcf_set_sample_names: name_lists[k] describes alleles[k]; NULL otherwise */
  const bvcf_names *name_lists;
  const char *names;
  uint64_t n_name_bytes;
  /* bvcf_submit_bgzf: the text the caller's TSV assembly needs; NULL for batches submitted as text (the caller has it).
   * head_off == NULL (no sample columns): the whole inflated text from the batch's first line on; lines[i].off index
   * into it.  head_off != NULL (files with samples): only the HEAD of every line that passed the gate -- its bytes up
   * to the end of the INFO column, fend[7] -- packed back to back: line i's bytes start at text[head_off[i]] (the
   * samples, 98 % of a cohort file, do not cross PCIe a second time).  Offsets inside a line (fend[], alt_off -
   * lines[i].off) apply to both forms. */
  const uint8_t *text;
  uint64_t n_text_bytes;
  const uint32_t *head_off;  /* [n_lines]; entries of lines that did not pass (status FIELDS / FILTER) are undefined */
  /* bvcf_params.packed_sites on a file without samples: sites[i] describes line i (n_lines of them); lines[] then
   * holds only the n_full_lines records of the lines marked BVCF_SITE_FULL, in no particular order (lines[j].gt_task is
   * the line number), alleles[j] is the first output allele of lines[j], and the further alleles of such lines follow the
   * n_full_lines first records (lines[j].rec_first >= n_full_lines; n_alleles = n_full_lines + the further ones).
   * bvcf_err.line stays the line number.  NULL otherwise. */
  const bvcf_site *sites;
  uint32_t n_full_lines;
  uint32_t n_row_cuts;
  /* bvcf_params.render_sites (ABI 7): rows[0 .. n_row_bytes) are the rows of the lines the packed form settles, in input
   * order, each with its "\n".  row_cuts lists, by line number, the n_row_cuts == n_full_lines lines that are NOT in there
   * (BVCF_SITE_FULL: their records are lines[slot] / alleles[slot] ...): the rows the caller makes of line row_cuts[j].line
   * belong at byte row_cuts[j].off of the stream.  n_ok_sites = the rendered rows (for the caller's counts). */
  const uint8_t *rows;
  uint64_t n_row_bytes;
  const bvcf_row_cut *row_cuts;
  uint64_t n_ok_sites;
} bvcf_result;

/* ---- lifecycle ---- */
int bvcf_create(bvcf_ctx **out, const bvcf_params *p);
void bvcf_destroy(bvcf_ctx *ctx);
const char *bvcf_last_error(const bvcf_ctx *ctx);
const char *bvcf_version(void);
int bvcf_reserve(bvcf_ctx *ctx, uint64_t lines, uint64_t alleles, uint64_t cmap_bytes);
/* The (normalised) sample names, header fields 9.., and the --fieldDelimiter text (at most 16 bytes), for
 * bvcf_params.want_name_lists: with them on the device, every collected batch carries the het / hom / missing name
 * lists of its output alleles as text, and the caller's TSV assembly copies three strings per row instead
 * of walking the class map name by name.  Call once, before the first bvcf_submit.  n must be the ctx's sample count. */
int bvcf_set_sample_names(bvcf_ctx *ctx, const char *const *names, const uint32_t *lens, uint32_t n, const char *delimiter);
/* bvcf_params.render_sites: what the rendered rows depend on besides the line -- the --emptyField text (at most 16 bytes;
 * NULL = "!") and which of the optional columns are on (main.go:674-692).  Call once, before the first bvcf_submit. */
int bvcf_set_row_format(bvcf_ctx *ctx, const char *empty_field, int keep_pos, int keep_id, int keep_info);

/* pinned host memory for blocks handed to bvcf_submit (hipHostMalloc) */
void *bvcf_alloc_pinned(size_t nbytes);
/* the same, placed near `device` (on a multi-socket host: the NUMA node the device hangs off); NULL if there is no
 * such device */
void *bvcf_alloc_pinned_near(int device, size_t nbytes);
void bvcf_free_pinned(void *p);
/* optional: brings up the HIP runtime on `device` and loads the library's kernels onto it, so that a later
 * bvcf_create does not pay for that (bvcf_run_fd calls it while it reads the header of its input) */
int bvcf_warmup(int device);

/* ---- the hot path ---- */
/* block: whole lines only (ends with a terminator; an unterminated tail is ignored, cf. main.go:354-358).
 * Asynchronous; the block must stay valid until the matching bvcf_collect returns. */
int bvcf_submit(bvcf_ctx *ctx, const uint8_t *block, size_t nbytes, uint64_t batch_seq);
/* same, block already resident on ctx's device; it must own BVCF_DEVICE_PAD bytes past nbytes */
int bvcf_submit_device(bvcf_ctx *ctx, const void *dblock, size_t nbytes, uint64_t batch_seq);
/* A batch given as BGZF blocks (bgzip / htslib .vcf.gz): the compressed bytes cross to the device and are inflated
 * and CRC-checked there (one wavefront per block), so no host core inflates and PCIe carries a fraction of the text.
 * comp holds whole BGZF blocks, n_own bytes of which are the batch's own; the blocks after them are look-ahead (the
 * next batch's first blocks, which that batch submits again as its own).  Batches are cut in the compressed domain,
 * so their text begins and ends inside lines; the rule that makes every line belong to exactly one batch:
 *   - the batch's text ends after the first terminator at or past the end of its own blocks' text (found in the
 *     look-ahead; without look-ahead -- the stream's last batch -- it ends where the text ends);
 *   - flags & BVCF_BGZF_SKIP_FIRST_LINE: the text before the batch's first terminator belongs to the previous batch
 *     and is skipped; otherwise the batch starts at byte first_off of its text (the stream's first batch: first_off
 *     is where the data lines begin);
 *   - flags & BVCF_BGZF_END_OF_STREAM: the look-ahead blocks are the last of the stream, so a final line without a
 *     terminator simply ends the batch (it is dropped, main.go:354-358) instead of counting as too little look-ahead.
 * bvcf_collect then also returns the text the line offsets refer to (bvcf_result.text: a pinned host copy).  Errors
 * surface at bvcf_collect: BVCF_E_FATAL for a corrupt block (inflate error or CRC mismatch) and for a line that does
 * not end within the look-ahead (give more look-ahead blocks).  The text of own + look-ahead blocks must fit
 * max_batch_bytes. */
#define BVCF_BGZF_SKIP_FIRST_LINE 1
#define BVCF_BGZF_END_OF_STREAM 2
int bvcf_submit_bgzf(bvcf_ctx *ctx, const uint8_t *comp, size_t n_comp, size_t n_own, int flags,
                     uint32_t first_off, uint64_t batch_seq);
/* blocks until the oldest submitted batch is done */
int bvcf_collect(bvcf_ctx *ctx, bvcf_result *r);
/* 1 = census path, 2 = streaming path (see bvcf_params.path) */
int bvcf_path(const bvcf_ctx *ctx);

/* running totals since bvcf_create: {lines_in, lines_ok, alleles_out, alleles_ac0, errs,
 * bytes_in, cmap_bytes, kernel_ns} */
int bvcf_counters(bvcf_ctx *ctx, uint64_t out[8]);
/* the run summary over several ctxs driven by one process: element-wise sum of their counters, on the host */
int bvcf_sum_counters(bvcf_ctx *const *ctxs, int n, uint64_t out[8]);
/* the same sum as the final count gather of a multi-GPU run (the only collective the path has): when
 * the n ctxs sit on n distinct devices (n >= 2) each ctx's totals are uploaded to its device and summed with one
 * RCCL ncclAllReduce(ncclSum, uint64[8]) over xGMI (single process, ncclCommInitAll; librccl.so.1 is dlopen'ed at
 * the first call), and out is read back from ctxs[0]'s device.  With one ctx, or ctxs that share a device (RCCL
 * has one rank per device), the sum is formed on the host as bvcf_sum_counters does.  *used_rccl (optional)
 * says which of the two happened.  BVCF_RCCL=1 in the environment forces the RCCL path for n == 1 too. */
int bvcf_allreduce_counters(bvcf_ctx *const *ctxs, int n, uint64_t out[8], int *used_rccl);
/* HIP devices visible to the process (0 when there is none or no runtime) */
int bvcf_device_count(void);
/* "domain:bus:device.function" of a device, as under /sys/bus/pci/devices (bvcf_run_fd keeps a device worker's host
 * threads on the NUMA node its GPU hangs off); cap >= 16 */
int bvcf_device_pci_bus_id(int device, char *out, int cap);

/* ---- host side of the path: header, TSV assembly, whole-stream driver ---- */

/* mirrors main.go:63-80 `Config` */
typedef struct {
  const char *empty_field;      /* --emptyField      "!" */
  const char *field_delimiter;  /* --fieldDelimiter  ";" */
  const char *allow_filter;     /* --allowFilter     "PASS,." */
  const char *exclude_filter;   /* --excludeFilter   "" */
  uint8_t keep_id, keep_info, keep_pos, keep_qual; /* keepQual is parsed and never read (main.go:94) */
  uint8_t normalize_header;     /* parse.NormalizeHeader restatement ('.' -> '_'), default 1 */
  uint8_t leave_teardown_to_exit; /* bvcf_run_fd: the process exits right after the call (the CLI): skip destroying the
                                   ctxs and unpinning the buffers, the OS reclaims them (~0.1 s of a 0.9 s run) */
  uint8_t reserved[2];
  int32_t device;               /* HIP device ordinal */
  uint32_t n_format_threads;    /* 0 = the CPUs the process may use (affinity, cgroup quota), at most 32 */
  uint64_t max_batch_bytes;     /* 0 = 64 MiB (256 MiB of text for a BGZF file inflated on the device) */
  const char *sample_list_path; /* --sample: write the sample names, one per line (main.go:398-445); NULL/"" = no */
  const char *dosage_path;      /* --dosageOutput: Arrow IPC file of the dosage matrix (main.go:306-342); NULL/"" = no */
  uint8_t no_out;               /* --noOut: no TSV rows and no header line (main.go:196-208,502) */
  uint8_t reserved3[3];
  /* bvcf_run_fd: the HIP devices the blocks of the stream are dealt to, round-robin in input order (the
   * counterpart of the reference's NumCPU workers, main.go:345-347).  One ctx and one host thread per entry; an
   * ordinal may repeat (two ctxs sharing a device).  n_devices == 0: the single device `device`.  A device only
   * gets a ctx once a block is dealt to it, so a short stream does not pay for the devices it does not reach. */
  uint32_t n_devices;
  const int32_t *devices;
} bvcf_config;

void bvcf_config_defaults(bvcf_config *c); /* setup() defaults, main.go:84-99 */

/* stringHeader(config), main.go:219-239: writes the tab-joined header (no newline), returns its
 * length (or the length needed if cap is too small) */
size_t bvcf_string_header(const bvcf_config *c, char *out, size_t cap);

/* TSV rows for one collected batch, in input order (main.go:566-695).  sample_names[i] /
 * sample_name_lens[i]: the (normalised) header fields 9.. ; appends to a malloc'd buffer the caller
 * releases with bvcf_free.  Also renders the error list as the reference's log lines into *log. */
int bvcf_format_tsv(const bvcf_config *c, const bvcf_result *r, const uint8_t *block,
                    const char *const *sample_names, const uint32_t *sample_name_lens,
                    char **out, size_t *n_out, char **log, size_t *n_log);

/* readVcf(config, reader, writer) on an in-memory VCF (main.go:241-396): preamble checks, header,
 * block cutting, submit/collect, TSV.  Output rows in input order, without the header line.
 * Returns BVCF_OK, or BVCF_E_FATAL with the reference's message in *log. */
int bvcf_run_buffer(const bvcf_config *c, const uint8_t *vcf, size_t n, char **out, size_t *n_out,
                    char **log, size_t *n_log, uint64_t *n_lines_in);

/* the same over file descriptors (the CLI): reads fd_in to EOF, writes header + rows to fd_out and
 * log lines to fd_err.  With bvcf_config.n_devices > 1 the blocks are dealt round-robin to one ctx per device and the
 * results merged by block number, so the output is the same bytes in the same order for any device list; the
 * per-ctx counters are summed at the end with bvcf_allreduce_counters.
 * Environment: BVCF_TIMING=1 adds one "[bvcf timing] ..." line to fd_err, BVCF_TIMING=json one JSON object
 * "[bvcf timing-json] {...}" (stage times in seconds, the devices used, the summed counters). */
int bvcf_run_fd(const bvcf_config *c, int fd_in, int fd_out, int fd_err, uint64_t *n_lines_in);

/* the byte source in front of bvcf_run_fd on its own: copies fd_in to fd_out, inflating gzip (streaming)
 * or BGZF (block-parallel, n_threads workers; 0 = up to 32) on the way.  Host-only: needs no device. */
int bvcf_decompress_fd(int fd_in, int fd_out, uint32_t n_threads, char *kind_out /* >= 8 bytes or NULL */);

/* BGZF blocks inflated ON THE DEVICE (one wavefront per block): comp holds whole BGZF blocks
 * (bgzip / htslib output; the empty end-of-file block may be among them), out receives their text, *n_out its length.
 * Each block is checked against its ISIZE and CRC32.  Returns BVCF_E_FATAL for data that is not BGZF or is corrupt,
 * BVCF_E_TOO_BIG if out is too small (*n_out then holds the size needed).  This is the building block of
 * bvcf_run_fd's compressed path, exported for tests and for callers that keep compressed data resident. */
int bvcf_bgzf_inflate_device(int device, const uint8_t *comp, size_t n_comp, uint8_t *out, size_t cap, size_t *n_out);

void bvcf_free(void *p);

/* ---- the dosage matrix file (--dosageOutput) ----
 * Replaces arrow/arrow.go's ArrowWriter + ArrowRowBuilder (NewArrowIPCFileWriter / WriteRow / Close,
 * called from main.go:334, 576-584, 698-716) for the table main.go:319-329 declares: a utf8 column
 * "locus" ("chrom:pos:ref:alt") and one int8 column per sample.  Writes an Arrow IPC file with record
 * batches of rows_per_batch rows (0 = 5 000, main.go:518) whose buffers are zstd-compressed
 * (zstd_level 0 = 3, < 0 = uncompressed).  Host-only. */
typedef struct bvcf_arrow bvcf_arrow;
int bvcf_arrow_open(bvcf_arrow **out, const char *path, const char *const *sample_names,
                    const uint32_t *sample_name_lens, uint32_t n_samples, uint32_t rows_per_batch, int zstd_level);
int bvcf_arrow_append(bvcf_arrow *w, const char *locus, uint32_t locus_len, const int8_t *dosage /* n_samples */);
int bvcf_arrow_close(bvcf_arrow *w); /* flushes the last batch, writes the footer, frees w */

#ifdef __cplusplus
}
#endif
#endif /* BVCF_H */
