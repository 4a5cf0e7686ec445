/*
 * bvcf_plan.h — the host-side partition logic of bvcf_run_fd, exported so that it can be tested without a device.
 * NOT part of the drop-in ABI (include/bvcf.h): only tests/ call these.
 *
 * The reference has ONE producer goroutine that cuts the input into 64-line work items for NumCPU workers
 * (/root/reference/main.go:345-380).  bvcf_run_fd replaces that with per-device readers over byte ranges of the input
 * file; which reader owns which line is decided by the pure functions below, with no communication between readers:
 *
 *   text   range i = file bytes [data_off + i*range_bytes, +range_bytes).  With T(x) = the first terminator at an
 *          offset >= x, range [a, b) owns the bytes (T(a), T(b)]: it skips the line it starts in, and owns the line
 *          that straddles its end.  Range 0 starts at the first data line.  The last range ends at the file's last
 *          terminator (an unterminated tail is dropped, main.go:354-358).
 *   BGZF   range i = compressed bytes [data_off + i*range_bytes, +range_bytes); it owns the BGZF blocks that START in
 *          it, found by scanning for a chain of three well-formed block headers.  Its batches are cut at block
 *          boundaries, so their text begins and ends inside lines; bvcf_submit_bgzf's rule (include/bvcf.h) gives each
 *          line to exactly one batch.
 */
#ifndef BVCF_PLAN_H
#define BVCF_PLAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  uint64_t data_off;    /* file offset of the first data line (text) / of the BGZF block that holds it */
  uint64_t range_bytes; /* bytes of the file per range */
  uint64_t spare_bytes; /* text: what a reader reads past its range for the line that straddles the range's end */
  uint64_t n_ranges;    /* range i belongs to worker i % n_workers */
} bvcf_range_plan;

/* text file of file_size bytes whose data lines start at data_off; max_batch_bytes = the pinned buffer a range and
 * its spare have to fit (0 = 64 MiB); first_line_bytes = length of the first data line incl. its terminator (sizes
 * the spare: eight such lines, 64 KiB at least, an eighth of the buffer at most) */
int bvcf_plan_text_ranges(uint64_t file_size, uint64_t data_off, uint64_t max_batch_bytes, uint64_t first_line_bytes,
                          bvcf_range_plan *out);
/* BGZF file: ranges of ~1/(4 n_workers) of the file, between 1 MiB and a quarter of max_batch_bytes (0 = 256 MiB),
 * rounded up to 64 KiB; data_off = offset of the block that holds the first data line */
int bvcf_plan_bgzf_ranges(uint64_t file_size, uint64_t data_off, uint32_t n_workers, uint64_t max_batch_bytes,
                          bvcf_range_plan *out);

/* what a text range owns inside its reader's window */
enum {
  BVCF_CUT_LINES = 0, /* window[start, end) are the range's lines, all of them */
  BVCF_CUT_NONE = 1,  /* one line covers the whole range: it belongs to an earlier range, nothing is owned */
  BVCF_CUT_LONG = 2   /* window[start, end) are the range's lines but the last: the line that straddles the range's end
                         starts at long_start and does not end inside the window (the reader fetches the rest) */
};
typedef struct {
  int32_t kind;
  uint32_t reserved;
  uint64_t start, end, long_start;
} bvcf_text_cut;
/* window = the n bytes read at the range's start (range_bytes + spare_bytes, or up to the file's end); own_len = the
 * range's own bytes in it; is_first: the range starts at the first data line; is_last: the range ends the file */
int bvcf_cut_text_range(const uint8_t *window, uint64_t n, uint64_t own_len, int is_first, int is_last, uint8_t eol,
                        bvcf_text_cut *out);

/* the first offset p >= from in buf[0, n) where a chain of BGZF blocks starts (the block at p is well formed and so are
 * the two after it; a chain that runs into the end of the buffer counts); -1 if there is none */
long bvcf_find_bgzf_chain(const uint8_t *buf, size_t n, size_t from);

/* host threads of a run, from the CPUs the process may use (affinity mask, cgroup quota) and the number of device
 * workers: everything that burns CPU -- the readers' copy threads and the formatter pools -- adds up to at most
 * `cpus` (when cpus >= 2 * n_workers; below that every worker still gets one of each); device threads wait */
typedef struct {
  uint32_t readers;        /* reader threads per worker (text ranges: 2 when the worker has >= 2 copy threads) */
  uint32_t copy_threads;   /* threads one reader splits a pread over (itself included) */
  uint32_t format_threads; /* TSV assembly threads per worker (the formatter thread included) */
  uint32_t busy_total;     /* n_workers * (readers * copy_threads + format_threads), stream mode: + 1 reader */
} bvcf_thread_budget;
enum { BVCF_MODE_STREAM = 0, BVCF_MODE_TEXT_RANGES = 1, BVCF_MODE_BGZF_RANGES = 2 };
int bvcf_plan_threads(uint32_t cpus, uint32_t n_workers, int mode, bvcf_thread_budget *out);

/* one block a reader of bvcf_run_fd hands to its device worker */
typedef struct {
  uint32_t worker;      /* the device worker that gets it (range % n_workers; stream mode: block number % n_workers) */
  uint32_t piece;       /* blocks of one range are ordered by piece */
  uint64_t range;
  uint64_t file_off;    /* text: file offset of the block's first byte; BGZF: of its first compressed block.  A line
                           that outran the spare room travels as a block of its own with the same meaning */
  uint64_t nbytes;      /* text: bytes of whole lines (0: nothing, the block only takes its turn in the output);
                           BGZF: compressed bytes, own blocks + look-ahead */
  uint64_t own;         /* BGZF: compressed bytes of the batch's own blocks */
  uint32_t first_off;   /* BGZF: where the batch's text starts in its first block (the run's first batch) */
  uint8_t bgzf;         /* 1: a batch of BGZF blocks (the compressed submit of include/bvcf.h, with bgzf_flags and first_off) */
  uint8_t bgzf_flags;   /* BVCF_BGZF_SKIP_FIRST_LINE | BVCF_BGZF_END_OF_STREAM */
  uint8_t last_piece;   /* closes its range */
  uint8_t reserved;
} bvcf_plan_block;

/* Runs bvcf_run_fd's input side over fd_in for n_workers device workers WITHOUT any device: the header is parsed, the
 * ranges planned, the same reader threads run (buffers from the heap instead of pinned memory), and every block a
 * device worker would receive is reported instead of submitted, in the order (range, piece).  *n_out = the number
 * of blocks (may exceed cap: then only cap were stored).  mode_out: BVCF_MODE_*.  device_inflate: 0 = BGZF input is
 * inflated by the host reader (stream mode), 1 = handed over compressed (the default of bvcf_run_fd).
 * Returns BVCF_OK or the status bvcf_run_fd would fail with (message on fd_err). */
int bvcf_plan_fd(int fd_in, int fd_err, uint32_t n_workers, uint64_t max_batch_bytes, int device_inflate,
                 bvcf_plan_block *out, size_t cap, size_t *n_out, int *mode_out, bvcf_range_plan *plan_out);

#ifdef __cplusplus
}
#endif
#endif /* BVCF_PLAN_H */
