// bvcf_core.hip — ctx, slots, launches and the device half of the C-ABI (include/bvcf.h).
//
// A ctx owns n_slots independent batch slots on one GPU.  submit = H2D (or adopt a resident
// block) + the five-kernel chain + D2H of the 24-byte counter block, all on the slot's stream;
// collect = wait, size check, D2H of exactly the used parts of the result arrays.
#include "bvcf_device.hip.h"
#include "../../include/bvcf_bench.h"
#include "bvcf_bgzf.h"

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: librccl.so.1 is dlopen'ed by bvcf_allreduce_counters
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace bvcf_dev;

namespace {

struct Slot {
  hipStream_t stream = nullptr;
  bool used_gen = false;  // the batch in flight went through k_stream_gen
  hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr, ev_ctr = nullptr;
  hipEvent_t ev_in = nullptr, ev_scan = nullptr;  // scan-stream hand-over (launch_chain)
  // device
  uint8_t *d_in = nullptr;
  uint32_t *d_census = nullptr, *d_group = nullptr, *d_line_off = nullptr;
  uint32_t group_words = 0;
  uint32_t *d_s2_groups = nullptr;  // k_census_tiles: two sets of group totals, used by the slot's batches in turn
  uint32_t s2_parity = 0;           // ... which one the batch being launched adds to (make_args)
  bvcf_line *d_lines = nullptr;
  bvcf_allele *d_alleles = nullptr;
  bvcf_site *d_sites = nullptr, *h_sites = nullptr;  // packed ctxs only
  // bvcf_params.render_sites: the stream of rendered rows, the lines left to the host, the scan's group arrays and totals
  // packed ctxs: the pinned copies of the full records are sized for what such files need (a few lines in a hundred
  // leave the fast lanes), not for every line -- pinning costs 0.25 ms per megabyte at ctx set-up; bvcf_collect grows them
  uint64_t hcap_recs = 0, hcap_errs = 0;  // h_lines: hcap_recs; h_alleles: 2 * hcap_recs (first records, then the further ones)
  uint8_t *d_rows = nullptr, *h_rows = nullptr;
  uint64_t cap_rows = 0;
  bvcf_row_cut *d_row_cuts = nullptr, *h_row_cuts = nullptr;
  uint32_t cap_row_cuts = 0, cap_render_groups = 0;
  uint64_t cap_host_cuts = 0;
  unsigned long long *d_rgroup_bytes = nullptr, *d_rtotals = nullptr, *h_rtotals = nullptr;
  unsigned long long *d_rgroup_ctext = nullptr;  // BGZF batches: the text of the lines left to the host, packed (k_render_rows)
  uint8_t *d_cut_text = nullptr;
  uint64_t cap_cut_text = 0;
  bool cut_text_on = false;  // the batch in flight was launched with it
  uint32_t *d_rgroup_full = nullptr;
  bvcf_err *d_errs = nullptr;
  uint8_t *d_cmap = nullptr;
  int8_t *d_dosage = nullptr;
  GtTask *d_tasks = nullptr;
  GtResult *d_results = nullptr;
  uint32_t *d_win_tabs = nullptr;  // wide ctxs only
  uint32_t win_tabs_cap = 0;
  StreamEntry *d_entries = nullptr;
  uint32_t *d_line_len = nullptr, *d_line_cmap = nullptr, *d_line_bits = nullptr, *d_finish_items = nullptr;
  uint16_t *d_head_bits = nullptr;
  // device-side name lists (want_name_lists)
  bvcf_names *d_name_lists = nullptr, *h_name_lists = nullptr;
  uint32_t *d_name_tot = nullptr;
  uint8_t *d_names = nullptr;
  char *h_names = nullptr;
  unsigned long long *d_name_total = nullptr, *h_name_total = nullptr;
  uint64_t cap_names = 0;
  // bvcf_submit_bgzf: compressed bytes, block descriptors, inflate results, the batch's text on the host
  uint8_t *d_comp = nullptr;
  uint64_t cap_comp = 0;
  uint32_t *d_bgzf = nullptr, *h_bgzf = nullptr;  // per block: BgzfDesc (4 words), expected crc, then status[], crc[]
  uint64_t cap_bgzf_blocks = 0;
  uint32_t *d_cuts = nullptr, *h_cuts = nullptr;  // {start, end, flags, first bad block}
  uint8_t *h_text = nullptr;
  uint64_t cap_h_text = 0;
  // ... of which only the line heads come back when there are samples (k_heads_*)
  uint32_t *d_head_off = nullptr, *h_head_off = nullptr;
  uint8_t *d_heads = nullptr;
  unsigned long long *d_head_total = nullptr, *h_head_total = nullptr;
  uint64_t cap_head_lines = 0;
  bool heads = false;  // the batch in flight returns heads
  hipEvent_t ev_cut = nullptr;
  bool await_cuts = false;  // inflate enqueued, the kernel chain not yet (it needs the cut points)
  bool is_bgzf = false;     // the batch in flight came through bvcf_submit_bgzf
  int bgzf_rc = 0;          // ... and was refused (corrupt block, line past the look-ahead): reported by bvcf_collect
  const char *bgzf_err = nullptr;
  uint32_t text_start = 0, text_total = 0;
  BatchCounters *d_counters = nullptr;
  // pinned host
  BatchCounters *h_counters = nullptr;
  bvcf_line *h_lines = nullptr;
  bvcf_allele *h_alleles = nullptr;
  bvcf_err *h_errs = nullptr;
  uint8_t *h_cmap = nullptr;
  int8_t *h_dosage = nullptr;
  // capacities this slot was allocated with
  uint64_t cap_lines = 0, cap_alleles = 0, cap_cmap = 0, cap_census = 0;
  // in-flight batch
  bool busy = false;
  uint64_t seq = 0;
  size_t nbytes = 0;
  const uint8_t *src = nullptr;  // the device text of the batch in flight
};

}  // namespace

struct bvcf_ctx {
  bvcf_params p{};
  int device = 0;
  int n_cu = 0;
  int gt_grid = 0, stream_grid = 0;
  hipStream_t scan_stream = nullptr;  // see launch_chain
  uint32_t stream_lds_pad = 0;  // dynamic LDS asked for with k_stream (it uses none): caps the k_stream workgroups of ALL batches per CU
  bool fused = false;
  // streaming path: which kernel walks the next batch -- k_stream (made for the 4-byte sample grid; other lines are
  // left to k_gt) or k_stream_gen (any fields, one pass).  Adaptive: a batch whose lines were mostly of the other
  // kernel's shape switches (the results are the same either way); BVCF_GEN_STREAM=0 / 1 pins it.
  bool gen_mode = false;
  uint32_t last_real = 0xFFFFFFFFu, last_finish = 0xFFFFFFFFu;  // the last collected batch's counters->n_real / n_finish (unknown: full grids)
  bool shape_seen = false;  // gen_mode has had its first hint (peek_line_shape) or a batch's counters
  int gen_policy = -1;  // -1 adaptive, 0 never, 1 always
  uint32_t gen_grid = 0;
  bool wide = false;  // census path with k_gt_wide in front of k_gt (see kWideSamples)
  uint64_t avg_line_bytes = 0;  // of the last collected batch (bvcf_submit_bgzf picks its inflate kernel by it)
  bool names_on = false;  // want_name_lists and bvcf_set_sample_names called: the chain ends with the k_name_* kernels
  uint32_t *d_name_off = nullptr;
  uint8_t *d_name_text = nullptr;
  NameTable name_table{};
  bool sites = false; // no sample columns: k_sites after the census instead of k_scatter_eol + k_head + k_finish
  bool sites1 = false;  // ... or k_sites1 on its own, no census, the line numbers by look-back (BVCF_SITES=3)
  bool sites2 = false;  // ... or k_sites2 behind the census: tiles, the common lines on fast lanes (the default for such input)
  bool packed = false;  // ... and the batch comes back in the packed form (bvcf_params.packed_sites; k_sites2 only)
  bool sites2_tile_census = true;  // ... its census per tile (k_census_tiles, no scan kernel) instead of per chunk (BVCF_S2_CENSUS=chunk)
  uint32_t s2_groups_cap = 0;      // entries of one of a slot's two sets of group totals (k_census_tiles)
  int sites_grid = 0, sites1_grid = 0;
  uint32_t win_bytes = 64u << 10;  // wide: bytes of a line's sample region per wave of the split general scan
  uint32_t tile_bytes = 0, tile_quota = 0;
  // bvcf_params.render_sites (packed ctxs): rows made on the device; the format comes with bvcf_set_row_format
  bool render = false, row_fmt_set = false;
  uint8_t *d_row_fmt = nullptr;  // "chr" | "\tSNP\t" | the constant tail
  uint32_t row_tail_len = 0, row_keep_pos = 0, row_keep_id = 0, row_keep_info = 0;
  uint32_t n_samples = 0;
  uint32_t cmap_stride = 0;
  uint32_t dosage_stride = 0;  // 0 unless want_dosage
  uint64_t max_lines = 0, max_alleles = 0, max_cmap = 0;
  uint64_t need_extras = 0;  // packed / k_sites1 ctxs: extra ALT records of the last batch that did not fit (they sit behind slot max_lines)
  FilterTable *d_filters = nullptr;
  uint32_t s1_fmode = 0, s1_fkey[4] = {0, 0, 0, 0}, s1_flen[4] = {0, 0, 0, 0};  // k_sites1's view of the allow list
  std::vector<Slot> slots;
  size_t head = 0, tail = 0, in_flight = 0;  // ring of busy slots, oldest at tail
  uint64_t totals[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  std::string err;
};

namespace {

thread_local std::string g_create_err;
// where HIP_TRY leaves its message when several threads work for one ctx (bvcf_create allocates the slots side by side)
thread_local std::string *g_err_sink = nullptr;

// Offsets into a block are 32-bit; the kernels read up to a few KiB past the last line (clamped loads, tile
// rounding), so a block stays a megabyte short of 4 GiB.
constexpr uint64_t kMaxBlockBytes = 0xFFF00000ull;

#define HIP_TRY(ctx, expr)                                                               \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      *(g_err_sink ? g_err_sink : &(ctx)->err) = std::string(#expr) + ": " + hipGetErrorString(e_); \
      return BVCF_E_HIP;                                                                 \
    }                                                                                    \
  } while (0)

bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; }

// main.go:108-123: strings.Split(v, ",") then strings.TrimSpace
int fill_filter(const char *text, bool star_is_nil, uint32_t *nil, uint32_t *n, uint16_t *off, uint16_t *len,
                uint8_t *pool, uint32_t *pool_used) {
  *nil = 1;
  *n = 0;
  if (!text || !*text) return 0;
  if (star_is_nil && strcmp(text, "*") == 0) return 0;
  *nil = 0;
  size_t L = strlen(text), start = 0;
  for (size_t i = 0; i <= L; i++) {
    if (i != L && text[i] != ',') continue;
    size_t a = start, e = i;
    while (a < e && is_space(text[a])) a++;
    while (e > a && is_space(text[e - 1])) e--;
    if (*n >= 32 || *pool_used + (e - a) > 2048) return -1;
    off[*n] = (uint16_t)*pool_used;
    len[*n] = (uint16_t)(e - a);
    memcpy(pool + *pool_used, text + a, e - a);
    *pool_used += (uint32_t)(e - a);
    (*n)++;
    start = i + 1;
  }
  return 0;
}

void free_slot(Slot &s) {
  if (s.stream) hipStreamSynchronize(s.stream);
  hipFree(s.d_in);
  hipFree(s.d_census);
  hipFree(s.d_group);
  hipFree(s.d_s2_groups);
  hipFree(s.d_line_off);
  hipFree(s.d_lines);
  hipFree(s.d_alleles);
  hipFree(s.d_sites);
  hipHostFree(s.h_sites);
  hipFree(s.d_rows);
  hipHostFree(s.h_rows);
  hipFree(s.d_row_cuts);
  hipHostFree(s.h_row_cuts);
  hipFree(s.d_rgroup_bytes);
  hipFree(s.d_rgroup_full);
  hipFree(s.d_rgroup_ctext);
  hipFree(s.d_cut_text);
  hipFree(s.d_rtotals);
  hipHostFree(s.h_rtotals);
  hipFree(s.d_errs);
  hipFree(s.d_cmap);
  hipFree(s.d_dosage);
  hipFree(s.d_tasks);
  hipFree(s.d_results);
  hipFree(s.d_win_tabs);
  hipFree(s.d_entries);
  hipFree(s.d_line_len);
  hipFree(s.d_line_cmap);
  hipFree(s.d_line_bits);
  hipFree(s.d_finish_items);
  hipFree(s.d_head_bits);
  hipFree(s.d_name_lists);
  hipFree(s.d_name_tot);
  hipFree(s.d_names);
  hipFree(s.d_name_total);
  hipHostFree(s.h_name_lists);
  hipHostFree(s.h_names);
  hipHostFree(s.h_name_total);
  hipFree(s.d_comp);
  hipFree(s.d_bgzf);
  hipHostFree(s.h_bgzf);
  hipFree(s.d_cuts);
  hipHostFree(s.h_cuts);
  hipHostFree(s.h_text);
  hipFree(s.d_head_off);
  hipHostFree(s.h_head_off);
  hipFree(s.d_heads);
  hipFree(s.d_head_total);
  hipHostFree(s.h_head_total);
  if (s.ev_cut) hipEventDestroy(s.ev_cut);
  hipFree(s.d_counters);
  hipHostFree(s.h_counters);
  hipHostFree(s.h_lines);
  hipHostFree(s.h_alleles);
  hipHostFree(s.h_errs);
  hipHostFree(s.h_cmap);
  hipHostFree(s.h_dosage);
  if (s.ev_k0) hipEventDestroy(s.ev_k0);
  if (s.ev_k1) hipEventDestroy(s.ev_k1);
  if (s.ev_in) hipEventDestroy(s.ev_in);
  if (s.ev_scan) hipEventDestroy(s.ev_scan);
  if (s.ev_ctr) hipEventDestroy(s.ev_ctr);
  if (s.stream) hipStreamDestroy(s.stream);
  s = Slot{};
}

// the text arena of a slot's name lists (bvcf_collect grows it when a batch needs more)
int alloc_name_arena(bvcf_ctx *c, Slot &s, uint64_t want_bytes) {
  if (want_bytes <= s.cap_names) return BVCF_OK;
  hipFree(s.d_names);
  hipHostFree(s.h_names);
  s.d_names = nullptr;
  s.h_names = nullptr;
  s.cap_names = 0;
  HIP_TRY(c, hipMalloc(&s.d_names, want_bytes + 64));
  HIP_TRY(c, hipHostMalloc(&s.h_names, want_bytes + 64, hipHostMallocDefault));
  s.cap_names = want_bytes;
  return BVCF_OK;
}

// the name-list buffers of a slot: lists / totals follow max_alleles, the text arena keeps its size
int alloc_names(bvcf_ctx *c, Slot &s, uint64_t want_bytes) {
  if (!c->names_on) return BVCF_OK;
  hipFree(s.d_name_lists);
  hipFree(s.d_name_tot);
  hipHostFree(s.h_name_lists);
  s.d_name_lists = nullptr;
  s.d_name_tot = nullptr;
  s.h_name_lists = nullptr;
  HIP_TRY(c, hipMalloc(&s.d_name_lists, c->max_alleles * sizeof(bvcf_names)));
  HIP_TRY(c, hipMalloc(&s.d_name_tot, (c->max_alleles + 1) * sizeof(uint32_t)));
  HIP_TRY(c, hipHostMalloc(&s.h_name_lists, c->max_alleles * sizeof(bvcf_names), hipHostMallocDefault));
  if (!s.d_name_total) {
    HIP_TRY(c, hipMalloc(&s.d_name_total, sizeof(unsigned long long)));
    HIP_TRY(c, hipHostMalloc(&s.h_name_total, sizeof(unsigned long long), hipHostMallocDefault));
  }
  return alloc_name_arena(c, s, want_bytes);
}

NameArgs make_name_args(bvcf_ctx *c, Slot &s) {
  NameArgs na{};
  na.nt = c->name_table;
  na.lists = s.d_name_lists;
  na.tot = s.d_name_tot;
  na.out = s.d_names;
  na.cap = s.cap_names;
  na.total = s.d_name_total;
  return na;
}

// (re)allocate the result arrays of a slot for the ctx's current capacities
int alloc_results(bvcf_ctx *c, Slot &s) {
  if (s.cap_lines == c->max_lines && s.cap_alleles == c->max_alleles && s.cap_cmap == c->max_cmap) return BVCF_OK;
  hipFree(s.d_line_off);
  hipFree(s.d_lines);
  hipFree(s.d_alleles);
  hipFree(s.d_sites);
  hipHostFree(s.h_sites);
  s.d_sites = nullptr;
  s.h_sites = nullptr;
  hipFree(s.d_errs);
  hipFree(s.d_cmap);
  hipFree(s.d_dosage);
  hipFree(s.d_tasks);
  hipFree(s.d_results);
  hipFree(s.d_win_tabs);
  s.d_win_tabs = nullptr;
  hipFree(s.d_line_len);
  hipFree(s.d_line_cmap);
  hipFree(s.d_line_bits);
  hipFree(s.d_finish_items);
  s.d_line_bits = nullptr;
  s.d_finish_items = nullptr;
  hipHostFree(s.h_lines);
  hipHostFree(s.h_alleles);
  hipHostFree(s.h_errs);
  hipHostFree(s.h_cmap);
  hipHostFree(s.h_dosage);
  s.d_dosage = nullptr;
  s.h_dosage = nullptr;
  s.d_line_off = nullptr;
  s.d_lines = nullptr;
  s.d_alleles = nullptr;
  s.d_errs = nullptr;
  s.d_cmap = nullptr;
  s.d_tasks = nullptr;
  s.d_results = nullptr;
  s.d_line_len = nullptr;
  s.d_line_cmap = nullptr;
  s.h_lines = nullptr;
  s.h_alleles = nullptr;
  s.h_errs = nullptr;
  s.h_cmap = nullptr;
  s.cap_lines = s.cap_alleles = s.cap_cmap = 0;
  HIP_TRY(c, hipMalloc(&s.d_line_off, (c->max_lines + 1) * sizeof(uint32_t)));
  HIP_TRY(c, hipMalloc(&s.d_lines, c->max_lines * sizeof(bvcf_line)));
  HIP_TRY(c, hipMalloc(&s.d_alleles, c->max_alleles * sizeof(bvcf_allele)));
  if (c->packed) {
    HIP_TRY(c, hipMalloc(&s.d_sites, (c->max_lines + 64) * sizeof(bvcf_site)));
    if (!c->render)  // (rendered rows: the site records never leave the device)
      HIP_TRY(c, hipHostMalloc(&s.h_sites, (c->max_lines + 64) * sizeof(bvcf_site), hipHostMallocDefault));
  }
  HIP_TRY(c, hipMalloc(&s.d_errs, c->max_alleles * sizeof(bvcf_err)));
  HIP_TRY(c, hipMalloc(&s.d_cmap, c->max_cmap + 64));
  HIP_TRY(c, hipMalloc(&s.d_tasks, c->max_alleles * sizeof(GtTask)));
  HIP_TRY(c, hipMalloc(&s.d_results, c->max_alleles * sizeof(GtResult)));
  if (c->wide) {
    s.win_tabs_cap = (uint32_t)std::min<uint64_t>(2 * (c->p.max_batch_bytes / c->win_bytes) + c->max_lines + 64, 0x7FFFFFFFu);
    HIP_TRY(c, hipMalloc(&s.d_win_tabs, (size_t)s.win_tabs_cap * sizeof(uint32_t)));
  }
  HIP_TRY(c, hipMalloc(&s.d_line_len, c->max_lines * sizeof(uint32_t)));
  HIP_TRY(c, hipMalloc(&s.d_line_cmap, c->max_lines * sizeof(uint32_t)));
  if (c->fused) {
    HIP_TRY(c, hipMalloc(&s.d_line_bits, c->max_lines * 8 * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc(&s.d_finish_items, (c->max_lines + c->max_alleles) * sizeof(uint32_t)));
  }
  if (c->packed) {
    s.hcap_recs = std::max<uint64_t>(4096, c->max_lines / 16);
    s.hcap_errs = std::max<uint64_t>(4096, c->max_lines / 16);
    HIP_TRY(c, hipHostMalloc(&s.h_lines, s.hcap_recs * sizeof(bvcf_line), hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc(&s.h_alleles, 2 * s.hcap_recs * sizeof(bvcf_allele), hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc(&s.h_errs, s.hcap_errs * sizeof(bvcf_err), hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc(&s.h_cmap, 4096, hipHostMallocDefault));  // (no samples: no class maps)
  } else {
    s.hcap_recs = s.hcap_errs = 0;
    HIP_TRY(c, hipHostMalloc(&s.h_lines, c->max_lines * sizeof(bvcf_line), hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc(&s.h_alleles, c->max_alleles * sizeof(bvcf_allele), hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc(&s.h_errs, c->max_alleles * sizeof(bvcf_err), hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc(&s.h_cmap, c->max_cmap + 64, hipHostMallocDefault));
  }
  if (c->dosage_stride) {
    HIP_TRY(c, hipMalloc(&s.d_dosage, c->max_alleles * c->dosage_stride + 64));
    HIP_TRY(c, hipHostMalloc(&s.h_dosage, c->max_alleles * c->dosage_stride + 64, hipHostMallocDefault));
  }
  {
    const int rc = alloc_names(c, s, std::max<uint64_t>(s.cap_names, c->p.max_batch_bytes / 2 + (1u << 20)));
    if (rc) return rc;
  }
  s.cap_lines = c->max_lines;
  s.cap_alleles = c->max_alleles;
  s.cap_cmap = c->max_cmap;
  return BVCF_OK;
}

int alloc_slot(bvcf_ctx *c, Slot &s) {
  HIP_TRY(c, hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
  HIP_TRY(c, hipEventCreate(&s.ev_k0));
  HIP_TRY(c, hipEventCreate(&s.ev_k1));
  HIP_TRY(c, hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
  HIP_TRY(c, hipEventCreateWithFlags(&s.ev_scan, hipEventDisableTiming));
  HIP_TRY(c, hipEventCreate(&s.ev_ctr));
  const uint64_t in_cap = c->p.max_batch_bytes + BVCF_DEVICE_PAD;
  HIP_TRY(c, hipMalloc(&s.d_in, in_cap));
  s.cap_census = std::max<uint64_t>((c->p.max_batch_bytes + kChunk - 1) / kChunk + 1, s1_state_words((uint32_t)c->p.max_batch_bytes));
  {  // (k_census_tiles: a count per tile and, behind them, a total per bundle; two sets of group totals, both zero to begin with)
    const uint32_t nt = s2_n_tiles((uint32_t)c->p.max_batch_bytes) + 1u;
    s.cap_census = std::max<uint64_t>(s.cap_census, (uint64_t)s2_bundle_off(nt) + s2_n_bundles(nt) + 64u);
    c->s2_groups_cap = s2_n_groups(nt) + 2u;
  }
  HIP_TRY(c, hipMalloc(&s.d_census, s.cap_census * sizeof(uint32_t)));
  // (group totals of the census scan, then -- from a 16-byte boundary -- the lines per producer wave of the streaming path)
  s.group_words = ((uint32_t)(s.cap_census / kScanGroup) + 2u + 3u) & ~3u;
  HIP_TRY(c, hipMalloc(&s.d_group, ((size_t)s.group_words + kMaxProducerWaves) * sizeof(uint32_t)));
  HIP_TRY(c, hipMalloc(&s.d_s2_groups, 2ull * c->s2_groups_cap * sizeof(uint32_t)));
  HIP_TRY(c, hipMemset(s.d_s2_groups, 0, 2ull * c->s2_groups_cap * sizeof(uint32_t)));
  s.s2_parity = 0;
  HIP_TRY(c, hipMalloc(&s.d_counters, sizeof(BatchCounters)));
  HIP_TRY(c, hipHostMalloc(&s.h_counters, sizeof(BatchCounters), hipHostMallocDefault));
  if (c->fused) {
    const uint64_t max_tiles = (c->p.max_batch_bytes + c->tile_bytes - 1) / c->tile_bytes + 1;
    HIP_TRY(c, hipMalloc(&s.d_entries, max_tiles * c->tile_quota * sizeof(StreamEntry)));
    HIP_TRY(c, hipMalloc(&s.d_head_bits, max_tiles * c->tile_quota * 16 * sizeof(uint16_t)));
  }
  return alloc_results(c, s);
}

KernelArgs make_args(bvcf_ctx *c, Slot &s, const uint8_t *d_src, size_t nbytes) {
  KernelArgs a{};
  a.buf = d_src;
  a.nbytes = (uint32_t)nbytes;
  a.cap = (uint32_t)(nbytes + BVCF_DEVICE_PAD);
  a.n_header = c->p.n_header_fields;
  a.n_samples = c->n_samples;
  a.eol_chars = c->p.eol_chars;
  a.eol_byte = c->p.eol_byte;
  a.want_cmap = c->p.want_class_maps;
  a.cmap_stride = c->cmap_stride;
  a.max_lines = (uint32_t)c->max_lines;
  a.max_alleles = (uint32_t)c->max_alleles;
  a.max_errs = (uint32_t)c->max_alleles;
  a.max_tasks = (uint32_t)c->max_alleles;
  a.max_cmap = c->max_cmap;
  a.filters = c->d_filters;
  a.census = s.d_census;
  a.group_base = s.d_group;
  a.run_lines = s.d_group + s.group_words;
  a.s2_groups = s.d_s2_groups + (s.s2_parity & 1u) * c->s2_groups_cap;
  a.s2_groups_next = s.d_s2_groups + ((s.s2_parity & 1u) ^ 1u) * c->s2_groups_cap;
  a.line_off = s.d_line_off;
  a.lines = s.d_lines;
  a.alleles = s.d_alleles;
  a.sites = c->packed ? s.d_sites : nullptr;
  a.errs = s.d_errs;
  a.cmap = s.d_cmap;
  a.dosage = s.d_dosage;
  a.dosage_stride = c->dosage_stride;
  a.tasks = s.d_tasks;
  a.results = s.d_results;
  a.counters = s.d_counters;
  a.fused = c->fused ? 1u : 0u;
  a.gen_stream = c->gen_mode ? 1u : 0u;
  a.prod_waves = (uint32_t)(c->gen_mode ? c->gen_grid : c->stream_grid) * kWavesPerWg;
  a.wide = c->wide ? 1u : 0u;
  a.win_bytes = c->win_bytes;
  a.win_tabs = s.d_win_tabs;
  a.win_tabs_cap = s.d_win_tabs ? s.win_tabs_cap : 0u;
  a.tile_bytes = c->tile_bytes;
  a.tile_quota = c->tile_quota;
  a.n_tiles = c->fused ? (uint32_t)((nbytes + c->tile_bytes - 1) / c->tile_bytes) : 0u;
  a.entries = s.d_entries;
  a.line_len = s.d_line_len;
  a.line_cmap = s.d_line_cmap;
  a.head_bits = s.d_head_bits;
  a.line_bits = s.d_line_bits;
  a.finish_items = s.d_finish_items;
  a.real_tasks = s.d_finish_items ? s.d_finish_items + c->max_lines : nullptr;  // (one allocation: [max_lines] + [max_alleles])
  a.s1_fmode = c->s1_fmode;
  for (int i = 0; i < 4; i++) {
    a.s1_fkey[i] = c->s1_fkey[i];
    a.s1_flen[i] = c->s1_flen[i];
  }
  return a;
}

// the kernel chain for one resident block; ev_gt0 / ev_gt1 (optional) bracket the dominant kernel
// (k_gt on the census path, k_stream on the streaming path)
void launch_names(bvcf_ctx *c, const KernelArgs &a, const NameArgs &na, hipStream_t st) {
  hipLaunchKernelGGL(k_name_len, dim3(c->n_cu * 4), dim3(kWgThreads), 0, st, a, na);
  hipLaunchKernelGGL(k_name_scan, dim3(1), dim3(1024), 0, st, a, na);
  hipLaunchKernelGGL(k_name_write, dim3(c->n_cu * 4), dim3(kWgThreads), 0, st, a, na);
}

// a batch's counters are in: should the next one go through the other streaming kernel?
void adapt_stream_kernel(bvcf_ctx *c, bool was_gen, const BatchCounters &ctr) {
  // how many scans the last batch left to k_gt, and lines to k_finish: the grids of the next batch's (launch_chain)
  c->last_real = ctr.n_real;
  c->last_finish = ctr.n_finish;
  if (c->gen_policy >= 0 || !c->fused || ctr.n_lines < 16) return;
  c->shape_seen = true;
  if ((uint64_t)ctr.n_other_shape * 2u > ctr.n_lines) c->gen_mode = !was_gen;
}

void launch_chain(bvcf_ctx *c, const KernelArgs &a, hipStream_t st, hipEvent_t ev_gt0, hipEvent_t ev_gt1, Slot *slot = nullptr) {
  if (a.fused) {
    // (experiments builds, BVCF_SCAN_STREAM=1: the one-pass kernels of ALL batches through one stream of the ctx, what follows
    // a batch's pass on the slot's stream behind events -- the chain then runs nearly serially; scan_stream is null otherwise)
    const bool split = c->scan_stream && slot && slot->ev_in && slot->ev_scan;
    hipStream_t ss = split ? c->scan_stream : st;
    if (split) {
      hipEventRecord(slot->ev_in, st);  // the text is in, and the slot's last batch is through
      hipStreamWaitEvent(ss, slot->ev_in, 0);
    }
    hipMemsetAsync(a.counters, 0, sizeof(BatchCounters), ss);
    if (ev_gt0) hipEventRecord(ev_gt0, ss);
    if (a.gen_stream)
      hipLaunchKernelGGL(k_stream_gen, dim3(c->gen_grid), dim3(kWgThreads), gen_lds_bytes(a.n_samples), ss, a);
    else
      hipLaunchKernelGGL(k_stream, dim3(c->stream_grid), dim3(kWgThreads), c->stream_lds_pad, ss, a);
    if (ev_gt1) hipEventRecord(ev_gt1, ss);
    if (split) {
      hipEventRecord(slot->ev_scan, ss);
      hipStreamWaitEvent(st, slot->ev_scan, 0);
    }
    hipLaunchKernelGGL(k_order, dim3(c->n_cu * 4), dim3(kWgThreads), 0, st, a);
#ifdef BVCF_EXPERIMENTS
    // experiment (results are then wrong): which follower costs what with blocks in flight -- 1: no k_head, 2: no k_gt, 4: no k_finish
    static const int skip = getenv("BVCF_EXP_SKIP") ? atoi(getenv("BVCF_EXP_SKIP")) : 0;
    static const int head_wgs = getenv("BVCF_EXP_HEAD_WGS") ? atoi(getenv("BVCF_EXP_HEAD_WGS")) : 4;
    static const int gt_div = getenv("BVCF_EXP_GT_DIV") ? atoi(getenv("BVCF_EXP_GT_DIV")) : 1;
#else
    constexpr int skip = 0, head_wgs = 4, gt_div = 1;
#endif
    if (skip & 1) {
    } else if (c->p.n_slots > 1)
      hipLaunchKernelGGL(k_head_lean, dim3(c->n_cu * head_wgs), dim3(kWgThreads), 0, st, a);
    else
      hipLaunchKernelGGL(k_head, dim3(c->n_cu * head_wgs), dim3(kWgThreads), 0, st, a);
    // k_gt and k_finish walk lists (real_tasks, finish_items) with grid strides: any grid is right.  A file of biallelic lines
    // leaves both empty, and a thousand workgroups that start to find that out hold wave slots the next blocks' scans would
    // use; the grids follow what the last collected batch needed (a wave of k_gt per two scans, a thread of k_finish per line).
    const uint32_t gt_full = (uint32_t)c->gt_grid / (uint32_t)gt_div;
    const uint32_t gt_wgs = c->last_real == 0xFFFFFFFFu ? gt_full
                            : std::min<uint32_t>(gt_full, std::max<uint32_t>((uint32_t)c->n_cu / 4u, c->last_real / (2u * kWavesPerWg) + 1u));
    const uint32_t fin_full = (uint32_t)c->n_cu * 4u;
    const uint32_t fin_wgs = c->last_finish == 0xFFFFFFFFu ? fin_full
                             : std::min<uint32_t>(fin_full, std::max<uint32_t>((uint32_t)c->n_cu / 4u, c->last_finish / kWgThreads + 1u));
    if (!(skip & 2)) hipLaunchKernelGGL(k_gt, dim3(gt_wgs), dim3(kWgThreads), 0, st, a);
    if (!(skip & 4)) hipLaunchKernelGGL(k_finish, dim3(fin_wgs), dim3(kWgThreads), 0, st, a);
    if (a.dosage) hipLaunchKernelGGL(k_dosage, dim3(c->gt_grid), dim3(kWgThreads), 0, st, a);
    return;
  }
#ifdef BVCF_EXPERIMENTS
  if (c->sites1) {
    // sites-only input, one pass: the counters and the tiles' look-back state start from zero
    const uint32_t n_words = s1_state_words(a.nbytes);
    hipLaunchKernelGGL(k_s1_zero, dim3(std::min<uint32_t>((n_words + 255u) / 256u, (uint32_t)c->n_cu)), dim3(256), 0, st, a, n_words);
    if (ev_gt0) hipEventRecord(ev_gt0, st);
    const uint32_t n_tiles = s1_n_tiles(a.nbytes);
    // (the first generation of workgroups starts spread over ~a tile's lifetime, see the kernel; wall_clock64 ticks at 100 MHz)
    static const uint32_t stagger_us = [] {
      const char *e = getenv("BVCF_S1_STAGGER_US");
      return e ? (uint32_t)atoi(e) : 0u;
    }();
    if (n_tiles)
      hipLaunchKernelGGL(k_sites1, dim3(s1_n_wgs(n_tiles)), dim3(kS1Threads), 0, st, a, n_tiles, (uint32_t)c->sites1_grid, stagger_us * 100u);
    if (ev_gt1) hipEventRecord(ev_gt1, st);
    return;
  }
#endif
  const uint32_t n_chunks = (a.nbytes + kChunk - 1) / kChunk;
  const uint32_t n_groups = (n_chunks + kScanGroup - 1) / kScanGroup;
  const uint32_t stream_grid = (uint32_t)std::min<uint64_t>((n_chunks + kWavesPerWg - 1) / kWavesPerWg,
                                                            (uint64_t)c->n_cu * 8);
  if (c->sites2 && c->sites2_tile_census) {
    // the census per tile, one scan level
    const uint32_t n_tiles = s2_n_tiles(a.nbytes);
    if (a.sites) {  // the packed form: no scan kernel, k_sites2p sums the census' three levels itself
      const uint32_t grid = s2_n_bundles(n_tiles);
      hipLaunchKernelGGL(k_census_tiles, dim3(grid ? grid : 1), dim3(kWgThreads), 0, st, a, n_tiles, c->s2_groups_cap);
    } else {
      const uint32_t grid = (uint32_t)std::min<uint64_t>((n_tiles + kWavesPerWg - 1) / kWavesPerWg, (uint64_t)c->n_cu * 8);
      hipLaunchKernelGGL(k_count_tiles, dim3(grid ? grid : 1), dim3(kWgThreads), 0, st, a, n_tiles);
      hipLaunchKernelGGL(k_scan_flat, dim3(1), dim3(1024), 0, st, a, n_tiles);
    }
    if (ev_gt0) hipEventRecord(ev_gt0, st);
    if (a.sites)
      hipLaunchKernelGGL(k_sites2p, dim3(c->sites1_grid), dim3(kS1Threads), 0, st, a, n_tiles, 0u);
    else
      hipLaunchKernelGGL(k_sites2, dim3(c->sites1_grid), dim3(kS1Threads), 0, st, a, n_tiles, 0u);
    if (ev_gt1) hipEventRecord(ev_gt1, st);
    return;
  }
  hipLaunchKernelGGL(k_count_eol, dim3(stream_grid ? stream_grid : 1), dim3(kWgThreads), 0, st, a, n_chunks);
  hipLaunchKernelGGL(k_scan_groups, dim3(n_groups ? n_groups : 1), dim3(kWgThreads), 0, st, a, n_chunks);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, st, a, n_groups);
  if (c->sites2) {
    if (ev_gt0) hipEventRecord(ev_gt0, st);
    if (a.sites)
      hipLaunchKernelGGL(k_sites2p, dim3(c->sites1_grid), dim3(kS1Threads), 0, st, a, s2_n_tiles(a.nbytes), n_chunks);
    else
      hipLaunchKernelGGL(k_sites2, dim3(c->sites1_grid), dim3(kS1Threads), 0, st, a, s2_n_tiles(a.nbytes), n_chunks);
    if (ev_gt1) hipEventRecord(ev_gt1, st);
    return;
  }
#ifdef BVCF_EXPERIMENTS
  if (c->sites) {
    // sites-only input: line records and allele records straight from one pass over the text
    if (ev_gt0) hipEventRecord(ev_gt0, st);
    hipLaunchKernelGGL(k_sites, dim3(c->sites_grid), dim3(kSitesThreads), 0, st, a, n_chunks);
    if (ev_gt1) hipEventRecord(ev_gt1, st);
    return;
  }
#endif
  hipLaunchKernelGGL(k_scatter_eol, dim3(stream_grid ? stream_grid : 1), dim3(kWgThreads), 0, st, a, n_chunks);
  // (k_head_lean when batches overlap: at 132 registers three of its workgroups fit on a CU beside the kernels of
  // the neighbouring batch; sites-only benchmark with two slots 4.4 -> 4.9 G variants/s, with one slot 3.5 -> 3.4)
  if (c->p.n_slots > 1)
    hipLaunchKernelGGL(k_head_lean, dim3(c->n_cu * 4), dim3(kWgThreads), 0, st, a);
  else
    hipLaunchKernelGGL(k_head, dim3(c->n_cu * 4), dim3(kWgThreads), 0, st, a);
  if (a.n_samples) {
    if (ev_gt0) hipEventRecord(ev_gt0, st);
    if (c->wide) {
      hipMemsetAsync(a.results, 0, (size_t)a.max_tasks * sizeof(GtResult), st);
      hipLaunchKernelGGL(k_gt_wide, dim3(c->gt_grid), dim3(kWgThreads), 0, st, a);
      if (a.win_tabs) {
        hipLaunchKernelGGL(k_tabs_wide, dim3(c->gt_grid), dim3(kWgThreads), 0, st, a);
        hipLaunchKernelGGL(k_gt_wide_general, dim3(c->gt_grid), dim3(kWgThreads), 0, st, a);
      }
    }
    hipLaunchKernelGGL(k_gt, dim3(c->gt_grid), dim3(kWgThreads), 0, st, a);
    if (ev_gt1) hipEventRecord(ev_gt1, st);
    hipLaunchKernelGGL(k_finish, dim3(c->n_cu * 4), dim3(kWgThreads), 0, st, a);
    if (a.dosage) hipLaunchKernelGGL(k_dosage, dim3(c->gt_grid), dim3(kWgThreads), 0, st, a);
    if (a.dosage && c->wide && a.win_tabs) hipLaunchKernelGGL(k_dosage_wide, dim3(c->gt_grid), dim3(kWgThreads), 0, st, a);
  } else {
    if (ev_gt0) hipEventRecord(ev_gt0, st);
    if (ev_gt1) hipEventRecord(ev_gt1, st);
  }
}

// k_crc32's tables (bvcf_inflate.hip.h), built once and kept on every device that inflates: slicing-by-4, the same
// tables moved on by 1 008 zero bytes, x^(8 * 16 * (63 - L)) per lane (zlib's multmodp on the reflected CRC-32 polynomial)
static const CrcTabs *crc_tabs_on_device() {
  static std::mutex mu;
  static const CrcTabs *on_dev[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (on_dev[dev]) return on_dev[dev];
  auto mul = [](uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 31; i >= 0; i--) {
      if ((a >> i) & 1u) p ^= b;
      b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
  };
  auto xpow = [&](uint64_t n) {  // x^n mod P
    uint32_t r = 0x80000000u, b = 0x40000000u;  // x^0, x^1
    for (; n; n >>= 1) {
      if (n & 1u) r = mul(r, b);
      b = mul(b, b);
    }
    return r;
  };
  static CrcTabs t;
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t c = i;
    for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
    t.std4[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t c = t.std4[0][i];
    for (int j = 1; j < 4; j++) {
      c = t.std4[0][c & 0xFFu] ^ (c >> 8);
      t.std4[j][i] = c;
    }
  }
  const uint32_t k1008 = xpow(8u * 1008u);
  for (int j = 0; j < 4; j++)
    for (uint32_t i = 0; i < 256; i++) t.jump[j][i] = mul(k1008, t.std4[j][i]);
  for (uint32_t L = 0; L < 64; L++) t.lane_k[L] = xpow(8u * 16u * (63u - L));
  CrcTabs *d = nullptr;
  if (hipMalloc(&d, sizeof t) != hipSuccess) return nullptr;
  if (hipMemcpy(d, &t, sizeof t, hipMemcpyHostToDevice) != hipSuccess) {
    hipFree(d);
    return nullptr;
  }
  on_dev[dev] = d;
  return d;
}

// inflate + CRC of BGZF blocks whose compressed bytes are at d_comp (device): text to d_text.  desc/crc/status are
// device arrays of n_blocks entries.
// w16: the 16 KiB-window kernel (two batches of blocks resident at once), for text whose lines are well under 16 KB
// false: the CRC tables could not be put on the device -- nothing was launched (a batch must not go unchecked)
static bool launch_inflate(int n_cu, const uint8_t *d_comp, const BgzfDesc *d_desc, uint32_t n_blocks, uint8_t *d_text,
                           uint32_t *d_status, uint32_t *d_crc, hipStream_t st, bool w16) {
  const CrcTabs *crc_tabs = crc_tabs_on_device();
  if (!crc_tabs) return false;
  static const int per_cu32 = [] {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_inflate, kInfThreads, 0) != hipSuccess || n < 1) n = 4;
    return n;
  }();
  static const int per_cu16 = [] {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_inflate_w16, kInfThreads, 0) != hipSuccess || n < 1) n = 7;
    if (getenv("BVCF_DEBUG")) fprintf(stderr, "[bvcf debug] k_inflate_w16: %d workgroups per CU\n", n);
    return n;
  }();
  static const int per_cu4 = [] {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_inflate_w4, kInfThreads, 0) != hipSuccess || n < 1) n = 12;
    if (getenv("BVCF_DEBUG")) fprintf(stderr, "[bvcf debug] k_inflate_w4: %d workgroups per CU\n", n);
    return n;
  }();
  // The 4 KiB window is the default: the decoder is a serial chain of ~250 instructions per symbol, so what counts is
  // how many blocks a SIMD interleaves -- 15 waves per CU against 7 (16 KiB) and 4 (32 KiB): 117 / 80 / 64 GB/s of text
  // on configs[2] rows, although most of its matches (a line repeats the one before it, 10 KB back) are then read back
  // from memory.  (`w16` is what the caller knows about the lines; kept for the A/B switch.)
  bool w4 = true;
  w16 = false;
  if (const char *e = getenv("BVCF_INFLATE_W16")) {  // tests / tuning: force the variant (0: 32 KiB, 1: 16 KiB, 2: 4 KiB)
    w16 = *e == '1';
    w4 = *e == '2';
  }
  const uint32_t grid = std::min<uint32_t>(n_blocks, (uint32_t)n_cu * (uint32_t)(w4 ? per_cu4 : (w16 ? per_cu16 : per_cu32)));
  if (w4)
    hipLaunchKernelGGL(k_inflate_w4, dim3(grid ? grid : 1), dim3(kInfThreads), 0, st, d_comp, d_desc, n_blocks, d_text, d_status);
  else if (w16)
    hipLaunchKernelGGL(k_inflate_w16, dim3(grid ? grid : 1), dim3(kInfThreads), 0, st, d_comp, d_desc, n_blocks, d_text, d_status);
  else
    hipLaunchKernelGGL(k_inflate, dim3(grid ? grid : 1), dim3(kInfThreads), 0, st, d_comp, d_desc, n_blocks, d_text, d_status);
  hipLaunchKernelGGL(k_crc32, dim3(std::min<uint32_t>(n_blocks ? n_blocks : 1, (uint32_t)n_cu * 16u)), dim3(kWave), 0, st,
                     (const uint8_t *)d_text, d_desc, n_blocks, crc_tabs, d_crc);
  return true;
}

// bvcf_params.render_sites: the slot's row stream and the list of the lines left to the host
int ensure_render_buffers(bvcf_ctx *c, Slot &s, uint64_t need_rows = 0, uint64_t need_host_cuts = 0) {
  // (what a batch of a typical file takes, not the worst case -- a row is a third of its line in a dbSNP-like file, and
  // pinned memory costs 0.25 ms per megabyte to get: a batch whose rows outgrow the stream grows it, bvcf_collect; the
  // pinned copy of the cuts is sized like the full records of a packed ctx: a sixteenth of the lines, grown on demand)
  const uint64_t extra = c->row_keep_info ? c->p.max_batch_bytes : 0;
  const uint64_t want_rows = std::max<uint64_t>(need_rows, c->p.max_batch_bytes / 8 * 3 + extra + (1u << 20));
  const uint32_t want_cuts = (uint32_t)std::min<uint64_t>(c->max_lines + 64, 0xFFFFFFF0ull);
  const uint64_t want_host_cuts = std::max<uint64_t>(need_host_cuts, std::max<uint64_t>(4096, c->max_lines / 16));
  const uint32_t want_groups = (uint32_t)((c->max_lines + kRenderGroup - 1) / kRenderGroup + 2);
  // (the stream on its own: when a batch's rows outgrew it, the prefixes k_render_scan left in the group arrays are what
  // the second k_render_rows works from)
  if (!s.d_rows || s.cap_rows < want_rows) {
    HIP_TRY(c, hipStreamSynchronize(s.stream));
    hipFree(s.d_rows);
    hipHostFree(s.h_rows);
    s.d_rows = s.h_rows = nullptr;
    s.cap_rows = 0;
    HIP_TRY(c, hipMalloc(&s.d_rows, want_rows));
    HIP_TRY(c, hipHostMalloc(&s.h_rows, want_rows, hipHostMallocDefault));
    s.cap_rows = want_rows;
  }
  if (!s.h_row_cuts || s.cap_host_cuts < want_host_cuts) {
    hipHostFree(s.h_row_cuts);
    s.h_row_cuts = nullptr;
    s.cap_host_cuts = 0;
    HIP_TRY(c, hipHostMalloc(&s.h_row_cuts, (size_t)want_host_cuts * sizeof(bvcf_row_cut), hipHostMallocDefault));
    s.cap_host_cuts = want_host_cuts;
  }
  if (!s.d_row_cuts || s.cap_row_cuts < want_cuts || s.cap_render_groups < want_groups) {
    HIP_TRY(c, hipStreamSynchronize(s.stream));
    hipFree(s.d_row_cuts);
    hipFree(s.d_rgroup_bytes);
    hipFree(s.d_rgroup_full);
    hipFree(s.d_rgroup_ctext);
    s.d_row_cuts = nullptr;
    s.d_rgroup_bytes = nullptr;
    s.d_rgroup_full = nullptr;
    s.d_rgroup_ctext = nullptr;
    s.cap_row_cuts = s.cap_render_groups = 0;
    HIP_TRY(c, hipMalloc(&s.d_row_cuts, (size_t)want_cuts * sizeof(bvcf_row_cut)));
    HIP_TRY(c, hipMalloc(&s.d_rgroup_bytes, (size_t)want_groups * sizeof(unsigned long long)));
    HIP_TRY(c, hipMalloc(&s.d_rgroup_full, (size_t)want_groups * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc(&s.d_rgroup_ctext, (size_t)want_groups * sizeof(unsigned long long)));
    s.cap_row_cuts = want_cuts;
    s.cap_render_groups = want_groups;
  }
  if (s.is_bgzf && !s.d_cut_text) {  // (a few lines in a hundred: an eighth of the batch and 1 MiB; more falls back to the whole text)
    s.cap_cut_text = c->p.max_batch_bytes / 8 + (1u << 20);
    HIP_TRY(c, hipMalloc(&s.d_cut_text, s.cap_cut_text));
  }
  if (!s.d_rtotals) {
    HIP_TRY(c, hipMalloc(&s.d_rtotals, 4 * sizeof(unsigned long long)));
    HIP_TRY(c, hipHostMalloc(&s.h_rtotals, 4 * sizeof(unsigned long long), hipHostMallocDefault));
  }
  return BVCF_OK;
}

// ... and its three kernels behind k_sites2p (bvcf_render.hip.h), the totals on their way to the host
RenderArgs make_render_args(bvcf_ctx *c, Slot &s, const KernelArgs &a) {
  RenderArgs ra{};
  ra.sites = a.sites;
  ra.text = a.buf;
  ra.rows = s.d_rows;
  ra.rows_cap = s.cap_rows;
  ra.cuts = s.d_row_cuts;
  ra.cuts_cap = s.cap_row_cuts;
  ra.n_groups_cap = s.cap_render_groups - 2u;
  ra.max_lines = a.max_lines;
  ra.group_bytes = s.d_rgroup_bytes;
  ra.group_full = s.d_rgroup_full;
  ra.totals = s.d_rtotals;
  ra.counters = a.counters;
  ra.fmt = c->d_row_fmt;
  ra.tail_len = c->row_tail_len;
  ra.keep_pos = c->row_keep_pos;
  ra.keep_id = c->row_keep_id;
  ra.keep_info = c->row_keep_info;
  ra.lines = a.lines;
  static const bool cut_text_off = [] {  // (BVCF_CUT_TEXT=0: the whole text of a BGZF batch comes back, for A/B and parity tests)
    const char *e = getenv("BVCF_CUT_TEXT");
    return e && *e == '0';
  }();
  ra.cut_text = (s.is_bgzf && s.d_cut_text && !cut_text_off) ? s.d_cut_text : nullptr;
  ra.cut_text_cap = s.cap_cut_text;
  ra.group_ctext = s.d_rgroup_ctext;
  return ra;
}
int launch_render(bvcf_ctx *c, Slot &s, const KernelArgs &a) {
  const RenderArgs ra = make_render_args(c, s, a);
  s.cut_text_on = ra.cut_text != nullptr;
  HIP_TRY(c, hipMemsetAsync(s.d_rtotals, 0, 4 * sizeof(unsigned long long), s.stream));
  const uint32_t grid = (uint32_t)c->n_cu * 8u;
  hipLaunchKernelGGL(k_render_len, dim3(grid), dim3(kWgThreads), 0, s.stream, ra);
  hipLaunchKernelGGL(k_render_scan, dim3(1), dim3(1024), 0, s.stream, ra);
  hipLaunchKernelGGL(k_render_rows, dim3(grid), dim3(kWgThreads), 0, s.stream, ra);
  HIP_TRY(c, hipGetLastError());
  return BVCF_OK;
}

// the kernel chain of the batch in slot s over the resident text src[0 .. nbytes), the counter read-back and the event
// bvcf_collect waits for
int launch_batch(bvcf_ctx *c, Slot &s, const uint8_t *src, size_t nbytes) {
  if (c->render) {
    if (!c->row_fmt_set) {
      c->err = "bvcf_params.render_sites needs bvcf_set_row_format before the first batch";
      return BVCF_E_ARG;
    }
    const int rc = ensure_render_buffers(c, s);
    if (rc) return rc;
  }
  HIP_TRY(c, hipEventRecord(s.ev_k0, s.stream));
  // (k_census_tiles of THIS chain zeroes the other parity's group totals for the slot's next batch: the flip and the
  // launch go together, nothing that can return early sits between them)
  s.s2_parity ^= 1u;
  KernelArgs a = make_args(c, s, src, nbytes);
  s.used_gen = a.gen_stream != 0;
  launch_chain(c, a, s.stream, nullptr, nullptr, &s);
  const bool names = c->names_on && s.d_name_lists;
  if (names) launch_names(c, a, make_name_args(c, s), s.stream);
  HIP_TRY(c, hipGetLastError());
  if (c->render) {
    const int rc = launch_render(c, s, a);
    if (rc) return rc;
  }
  HIP_TRY(c, hipEventRecord(s.ev_k1, s.stream));
  HIP_TRY(c, hipMemcpyAsync(s.h_counters, s.d_counters, sizeof(BatchCounters), hipMemcpyDeviceToHost, s.stream));
  if (c->render) {
    HIP_TRY(c, hipMemcpyAsync(s.h_rtotals, s.d_rtotals, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s.stream));
    // (Tried: as much of the stream as the last batch's rows-per-text ratio predicts copied to the host right here,
    // behind the kernels, so that it crosses beside the next batches' uploads.  The slot's host buffer may still be read by
    // the caller then -- results stay valid until the n_slots-th following COLLECT, a submit comes earlier -- and with a
    // second buffer to make it legal the run was no faster: a sites-only run waits for the uploads, 40 GB/s.)
  }
  if (names)
    HIP_TRY(c, hipMemcpyAsync(s.h_name_total, s.d_name_total, sizeof(unsigned long long), hipMemcpyDeviceToHost, s.stream));
  s.heads = s.is_bgzf && c->n_samples > 0 && s.d_head_off && nbytes > 0;
  if (s.heads) {
    HeadArgs h;
    h.off = s.d_head_off;
    h.out = s.d_heads;
    h.cap = c->p.max_batch_bytes;
    h.total = s.d_head_total;
    hipLaunchKernelGGL(k_heads_len, dim3(c->n_cu * 2), dim3(kWgThreads), 0, s.stream, a, h);
    hipLaunchKernelGGL(k_heads_scan, dim3(1), dim3(1024), 0, s.stream, a, h);
    hipLaunchKernelGGL(k_heads_copy, dim3(c->n_cu * 4), dim3(kWgThreads), 0, s.stream, a, h);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(s.h_head_total, s.d_head_total, sizeof(unsigned long long), hipMemcpyDeviceToHost, s.stream));
  }
  s.src = src;
  s.nbytes = nbytes;
  HIP_TRY(c, hipEventRecord(s.ev_ctr, s.stream));
  return BVCF_OK;
}

int submit_common(bvcf_ctx *c, const uint8_t *host_block, const void *dev_block, size_t nbytes, uint64_t seq) {
  if (!c) return BVCF_E_ARG;
  if (nbytes > c->p.max_batch_bytes || nbytes >= kMaxBlockBytes) {
    c->err = "block larger than max_batch_bytes";
    return BVCF_E_TOO_BIG;
  }
  if (c->in_flight == c->slots.size()) {
    c->err = "all slots in flight";
    return BVCF_E_BUSY;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  Slot &s = c->slots[c->head];
  int rc = alloc_results(c, s);
  if (rc) return rc;
  const uint8_t *src = (const uint8_t *)dev_block;
  if (host_block) {
    HIP_TRY(c, hipMemcpyAsync(s.d_in, host_block, nbytes, hipMemcpyHostToDevice, s.stream));
    // the pad is read (and masked) by the last lanes of the last chunk: keep it defined
    HIP_TRY(c, hipMemsetAsync(s.d_in + nbytes, '\n', BVCF_DEVICE_PAD, s.stream));
    src = s.d_in;
  }
  s.await_cuts = false;
  s.is_bgzf = false;
  s.bgzf_rc = 0;
  rc = launch_batch(c, s, src, nbytes);
  if (rc) return rc;
  s.busy = true;
  s.seq = seq;
  c->head = (c->head + 1) % c->slots.size();
  c->in_flight++;
  return BVCF_OK;
}

// second half of bvcf_submit_bgzf, once the cut points of the slot's text are on the host: the kernel chain over
// text[start, end) and the copy of that text for the caller's TSV assembly.  wait: block until they are.
int launch_after_cuts(bvcf_ctx *c, Slot &s, bool wait) {
  if (!s.await_cuts) return BVCF_OK;
  if (!wait && hipEventQuery(s.ev_cut) != hipSuccess) return BVCF_OK;  // not yet (or an error: collect reports it)
  hipError_t e = hipEventSynchronize(s.ev_cut);
  s.await_cuts = false;
  if (e != hipSuccess) {
    c->err = std::string("BGZF inflate failed: ") + hipGetErrorString(e);
    return BVCF_E_HIP;
  }
  const uint32_t start = s.h_cuts[0], end = s.h_cuts[1], flags = s.h_cuts[2];
  if (flags) {
    s.bgzf_err = (flags & kCutInflateError) ? "bgzf: corrupt block (inflate)"
                 : (flags & kCutCrcMismatch) ? "bgzf: corrupt block (CRC mismatch)"
                                              : "bgzf: a line does not end within the look-ahead blocks";
    s.bgzf_rc = BVCF_E_FATAL;
    return launch_batch(c, s, s.d_in, 0);  // (the slot still has to be collected: an empty batch carries the error)
  }
  s.text_start = start;
  // (the text itself is copied back by bvcf_collect, with the other result arrays: h_text may still be read by the
  // caller for the batch this slot held before)
  return launch_batch(c, s, s.d_in + start, end - start);
}

}  // namespace

extern "C" {

#ifdef BVCF_EXPERIMENTS
const char *bvcf_version(void) { return "bvcf-mi355x 0.1 (gfx950) +experiments"; }
#else
const char *bvcf_version(void) { return "bvcf-mi355x 0.1 (gfx950)"; }
#endif

const char *bvcf_last_error(const bvcf_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

void *bvcf_alloc_pinned(size_t nbytes) {
  void *p = nullptr;
  // portable: blocks are handed to the ctx of whichever device they are dealt to (bvcf_run_fd)
  if (hipHostMalloc(&p, nbytes ? nbytes : 1, hipHostMallocPortable) != hipSuccess) return nullptr;
  return p;
}

void *bvcf_alloc_pinned_near(int device, size_t nbytes) {
  // (the runtime places pinned memory on the NUMA node closest to the calling thread's current device)
  // The calling thread's current device is put back afterwards.
  int n_dev = 0, was = -1;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) return nullptr;
  if (hipGetDevice(&was) != hipSuccess) was = -1;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  void *p = bvcf_alloc_pinned(nbytes);
  if (was >= 0 && was != device) hipSetDevice(was);
  return p;
}

int bvcf_device_pci_bus_id(int device, char *out, int cap) {
  if (!out || cap < 16) return BVCF_E_ARG;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) return BVCF_E_NODEV;
  return hipDeviceGetPCIBusId(out, cap, device) == hipSuccess ? BVCF_OK : BVCF_E_HIP;
}

__global__ void k_warm(uint32_t *p) {
  if (p) *p = 1u;
}

int bvcf_warmup(int device) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || device < 0 || device >= n_dev) return BVCF_E_NODEV;
  if (hipSetDevice(device) != hipSuccess || hipFree(nullptr) != hipSuccess) return BVCF_E_HIP;
  // the first query about a kernel loads the library's code object onto the device
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_finish, kWgThreads, 0) != hipSuccess) return BVCF_E_HIP;
  // ... and the first launch sets up the queue behind the stream (0.16 s in a bare HIP program: tools/hostreg_bench.hip)
  hipLaunchKernelGGL(k_warm, dim3(1), dim3(64), 0, 0, (uint32_t *)nullptr);
  return hipDeviceSynchronize() == hipSuccess ? BVCF_OK : BVCF_E_HIP;
}

void bvcf_free_pinned(void *p) {
  if (p) hipHostFree(p);
}

void bvcf_destroy(bvcf_ctx *c) {
  if (!c) return;
  hipSetDevice(c->device);
  for (auto &s : c->slots) free_slot(s);
  if (c->scan_stream) hipStreamDestroy(c->scan_stream);
  hipFree(c->d_filters);
  hipFree(c->d_row_fmt);
  hipFree(c->d_name_off);
  hipFree(c->d_name_text);
  delete c;
}

int bvcf_create(bvcf_ctx **out, const bvcf_params *p) {
  if (!out || !p) return BVCF_E_ARG;
  *out = nullptr;
  if (p->abi_version != BVCF_ABI_VERSION || p->n_header_fields < 1 || p->eol_chars < 1 || p->eol_chars > 2) {
    g_create_err = "bad bvcf_params";
    return BVCF_E_ARG;
  }
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || p->device < 0 || p->device >= n_dev) {
    // there is deliberately no CPU fallback
    g_create_err = "no usable HIP device (libbvcf has no CPU fallback)";
    return BVCF_E_NODEV;
  }
  bvcf_ctx *c = new bvcf_ctx();
  c->p = *p;
  c->device = p->device;
  if (!c->p.max_batch_bytes) c->p.max_batch_bytes = 64ull << 20;
  if (c->p.max_batch_bytes >= kMaxBlockBytes) c->p.max_batch_bytes = kMaxBlockBytes - 1;
  if (!c->p.n_slots) c->p.n_slots = 3;  // (measured better than 2 or equal on every input shape: profiles/r05_blocks_in_flight_2_vs_3_all_profiles.txt)
  if (!c->p.eol_byte) c->p.eol_byte = '\n';
  c->n_samples = p->n_header_fields > 9 ? p->n_header_fields - 9 : 0;
  c->cmap_stride = ((c->n_samples + 3) / 4 + 15) & ~15u;
  c->dosage_stride = p->want_dosage && c->n_samples ? ((c->n_samples + 15) & ~15u) : 0u;
  const uint64_t min_line = std::max<uint64_t>(48, 2ull * p->n_header_fields);
  // (the slack is for short lines -- comments, junk -- between the records; every listed line owns a class-map slot
  // on the census path, so for very wide cohorts the slack is what 32 MiB of maps can hold: a batch that needs more
  // grows the reservation, BVCF_E_CAPACITY)
  const uint64_t slack = std::min<uint64_t>(4096, std::max<uint64_t>(64, (32ull << 20) / std::max<uint32_t>(c->cmap_stride, 1u)));
  c->max_lines = p->max_lines ? p->max_lines : c->p.max_batch_bytes / min_line + slack;
  c->max_alleles = p->max_alleles ? p->max_alleles : 2 * c->max_lines + 1024;
  if (c->max_alleles < c->max_lines + 64) c->max_alleles = c->max_lines + 64;  // slot i belongs to line i
  c->max_cmap = p->cmap_bytes ? p->cmap_bytes : (c->max_lines + c->max_lines / 2) * (uint64_t)c->cmap_stride + (1ull << 20);
  if (c->max_cmap > 0xFFFFFF00ull) c->max_cmap = 0xFFFFFF00ull;  // cmap_off is 32-bit
  if (c->max_cmap < 4096) c->max_cmap = 4096;  // (k_gt's prefetch reads a raw-list area's worth from the start of the arena)
  c->max_cmap = (c->max_cmap + 63) & ~63ull;
  // streaming path: lines are found by the genotype scan itself.  Its tile-local entry quota is
  // bounded because a line that passes the field count is at least n_header - 1 bytes long; for
  // narrow files the quota would dwarf the text, so they stay on the census path unless asked.
  c->tile_bytes = 64u << 10;  // (8-64 KiB measure alike now that the runs are balanced)
  if (const char *e = getenv("BVCF_TILE_KB")) {
    const unsigned kb = (unsigned)atoi(e);
    if (kb >= 4 && kb <= 1024) c->tile_bytes = kb << 10;
  }
  uint32_t path = p->path;
  if (const char *e = getenv("BVCF_PATH")) path = (uint32_t)atoi(e);  // test / tuning override
  // From kWideSamples samples up a line is hundreds of kilobytes and a batch holds too few of them to fill the GPU
  // with one wave per line: the census path then splits the regular scan of a line over several waves, and is
  // what `choose` picks.
  const bool many_samples = c->n_samples >= kWideSamples;
  c->fused = c->n_samples > 0 && (path == 2 || path == 3 || (path == 0 && p->n_header_fields >= 256 && !many_samples));
  c->wide = !c->fused && many_samples;
  if (const char *e = getenv("BVCF_GEN_STREAM")) c->gen_policy = atoi(e) != 0 ? 1 : 0;
  if (!c->fused || c->n_samples > 4u * kStageBytes) c->gen_policy = 0;  // (a line's dense class map is staged in LDS)
  c->gen_mode = c->gen_policy == 1;
  if (path == 3 && c->gen_policy < 0) {  // the caller has seen a line: its sample fields carry more than GT
    c->gen_mode = true;
    c->shape_seen = true;
  }
  if (const char *e = getenv("BVCF_WIDE")) c->wide = !c->fused && c->n_samples > 0 && atoi(e) != 0;  // test / tuning override
  if (const char *e = getenv("BVCF_WIDE_WIN")) {  // test / tuning: window of the split general scan, bytes
    const long v = atol(e);
    if (v >= 64 && v <= (64l << 20)) c->win_bytes = (uint32_t)v;
  }
  c->tile_quota = c->tile_bytes / (p->n_header_fields - 1 + p->eol_chars) + 2;
  auto fail = [&](int rc) {
    g_create_err = c->err;
    bvcf_destroy(c);
    return rc;
  };
  if (hipSetDevice(c->device) != hipSuccess) {
    c->err = "hipSetDevice failed";
    return fail(BVCF_E_HIP);
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) {
    c->err = "hipGetDeviceProperties failed";
    return fail(BVCF_E_HIP);
  }
  c->n_cu = prop.multiProcessorCount;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gt, kWgThreads, 0) != hipSuccess || per_cu < 1)
    per_cu = 4;
  c->gt_grid = c->n_cu * per_cu;
  per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_stream, kWgThreads, 0) != hipSuccess || per_cu < 1)
    per_cu = 3;
  // Two waves per SIMD run the scan as fast as three (it is bound by VALU issue).  With more than one batch in
  // flight the third wave's registers are better spent on the previous batch's k_head_lean / k_gt / k_finish, which
  // then run beside this kernel instead of waiting for its workgroups to finish (+7 % on the two-slot benchmark).
  if (c->p.n_slots > 1 && per_cu > 2) per_cu = 2;
#ifdef BVCF_EXPERIMENTS
  if (const char *e = getenv("BVCF_STREAM_WGS")) {  // experiment: workgroups per CU, up to the occupancy limit
    const int w = atoi(e);
    if (w >= 1 && w <= 4) per_cu = w;
  }
#endif
  c->stream_grid = c->n_cu * per_cu;
#ifdef BVCF_EXPERIMENTS
  // (round 5, measured and not adopted: every batch's one-pass kernel on one stream of the ctx; LDS asked for with k_stream
  // to cap its workgroups per CU over all batches -- profiles/r05_c4_in_flight_what_the_ten_percent_are.txt)
  if (c->p.n_slots > 1 && getenv("BVCF_SCAN_STREAM") && atoi(getenv("BVCF_SCAN_STREAM")) == 1 &&
      hipStreamCreateWithFlags(&c->scan_stream, hipStreamNonBlocking) != hipSuccess) {
    c->err = "hipStreamCreate failed";
    return fail(BVCF_E_HIP);
  }
  if (const char *e = getenv("BVCF_EXP_STREAM_LDS")) c->stream_lds_pad = (uint32_t)atoi(e);
#endif
  per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_stream_gen, kWgThreads, gen_lds_bytes(c->n_samples)) != hipSuccess || per_cu < 1)
    per_cu = 2;
  per_cu = std::min(per_cu, 6);  // (7 fit a cohort of a few thousand samples; 6 measured best)
#ifdef BVCF_EXPERIMENTS
  if (const char *e = getenv("BVCF_GEN_WGS")) {  // experiment: workgroups per CU
    const int w = atoi(e);
    if (w >= 1 && w <= 8) per_cu = std::min(per_cu, w);
  }
#endif
  c->gen_grid = c->n_cu * per_cu;
  // sites-only input takes k_sites2 behind its census (BVCF_SITES=0: the census chain with k_head, for A/B and parity
  // tests; builds with -DBVCF_EXPERIMENTS also know 1: k_sites, round 2's kernel, and 3: k_sites1, no census, the line
  // numbers by look-back -- both slower, kept out of the product library)
  c->sites = c->n_samples == 0;
  c->sites2 = c->sites;
  if (const char *e = getenv("BVCF_SITES")) {
    const int m = atoi(e);
#ifdef BVCF_EXPERIMENTS
    c->sites = c->sites && m != 0;
    c->sites2 = c->sites && m == 2;
    c->sites1 = c->sites && m == 3;
#else
    c->sites = c->sites2 = c->sites && m != 0;
#endif
  }
  if (const char *e = getenv("BVCF_S2_CENSUS")) c->sites2_tile_census = strcmp(e, "chunk") != 0;  // (A/B and parity tests)
  c->packed = c->sites2 && p->packed_sites != 0;
  c->render = c->packed && p->render_sites != 0;
  per_cu = 0;
#ifdef BVCF_EXPERIMENTS
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sites, kSitesThreads, 0) != hipSuccess || per_cu < 1)
    per_cu = 3;
  c->sites_grid = c->n_cu * per_cu;
  per_cu = 0;
#endif
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sites2, kS1Threads, 0) != hipSuccess || per_cu < 1)
    per_cu = 2;
#ifdef BVCF_EXPERIMENTS
  if (const char *e = getenv("BVCF_SITES1_WGS")) {  // experiment: workgroups per CU
    const int w = atoi(e);
    if (w >= 1 && w <= 8) per_cu = std::min(per_cu, w);
  }
  if (const char *e = getenv("BVCF_S2_WGS")) {  // experiment: workgroups per CU
    const int w = atoi(e);
    if (w >= 1 && w <= 4) per_cu = std::min(per_cu, w);
  }
#endif
  c->sites1_grid = c->n_cu * per_cu;
  // the streaming kernel gives every wave its own range of class-map slots (two of them slack): room for that
  if (!p->cmap_bytes) {
    c->max_cmap += (uint64_t)c->stream_grid * kWavesPerWg * 2u * c->cmap_stride + c->max_lines / 16 * (uint64_t)c->cmap_stride;
    c->max_cmap = std::min<uint64_t>((c->max_cmap + 63) & ~63ull, 0xFFFFFF00ull);
  }

  FilterTable ft;
  memset(&ft, 0, sizeof ft);
  uint32_t used = 0;
  if (fill_filter(p->allow_filter, true, &ft.allow_nil, &ft.allow_n, ft.allow_off, ft.allow_len, ft.text, &used) ||
      fill_filter(p->exclude_filter, false, &ft.deny_nil, &ft.deny_n, ft.deny_off, ft.deny_len, ft.text, &used)) {
    c->err = "too many / too long FILTER values (32 values, 2048 bytes)";
    return fail(BVCF_E_ARG);
  }
  // k_sites1 tests FILTER values of up to four bytes as dwords: possible when nothing is excluded and the allow list is
  // up to four values of one to four bytes (the default "PASS,." is), or allows everything
  if (ft.deny_nil) {
    if (ft.allow_nil) {
      c->s1_fmode = 2;
    } else if (ft.allow_n <= 4) {
      c->s1_fmode = 1;
      for (uint32_t i = 0; i < ft.allow_n; i++) {
        const uint32_t l = ft.allow_len[i];
        if (l == 0 || l > 4) {
          c->s1_fmode = 0;
          break;
        }
        uint32_t k = 0;
        for (uint32_t q = 0; q < l; q++) k |= (uint32_t)ft.text[ft.allow_off[i] + q] << (8 * q);
        c->s1_fkey[i] = k;
        c->s1_flen[i] = l;
      }
      if (!c->s1_fmode)
        for (int i = 0; i < 4; i++) c->s1_flen[i] = 0;
    }
  }
  if (hipMalloc(&c->d_filters, sizeof ft) != hipSuccess ||
      hipMemcpy(c->d_filters, &ft, sizeof ft, hipMemcpyHostToDevice) != hipSuccess) {
    c->err = "filter table upload failed";
    return fail(BVCF_E_HIP);
  }
  c->slots.resize(c->p.n_slots);
  {
    // the slots' buffers side by side: pinning the result arrays is most of what creating a ctx costs
    std::vector<int> rcs(c->slots.size(), BVCF_OK);
    std::vector<std::string> errs(c->slots.size());
    std::vector<std::thread> th;
    for (size_t i = 1; i < c->slots.size(); i++)
      th.emplace_back([c, i, &rcs, &errs]() {
        g_err_sink = &errs[i];
        if (hipSetDevice(c->device) != hipSuccess) {
          rcs[i] = BVCF_E_HIP;
          errs[i] = "hipSetDevice failed";
          return;
        }
        rcs[i] = alloc_slot(c, c->slots[i]);
      });
    g_err_sink = &errs[0];
    rcs[0] = alloc_slot(c, c->slots[0]);
    g_err_sink = nullptr;
    for (auto &t : th) t.join();
    for (size_t i = 0; i < rcs.size(); i++)
      if (rcs[i]) {
        c->err = errs[i];
        return fail(rcs[i]);
      }
  }
  *out = c;
  return BVCF_OK;
}

int bvcf_set_sample_names(bvcf_ctx *c, const char *const *names, const uint32_t *lens, uint32_t n, const char *delimiter) {
  if (!c || (n && (!names || !lens)) || !delimiter) return BVCF_E_ARG;
  if (c->in_flight) {
    c->err = "bvcf_set_sample_names with batches in flight";
    return BVCF_E_BUSY;
  }
  const size_t dl = strlen(delimiter);
  if (n != c->n_samples || dl > sizeof c->name_table.delim) {
    c->err = "bvcf_set_sample_names: sample count differs from the ctx's, or the delimiter is longer than 16 bytes";
    return BVCF_E_ARG;
  }
  if (!c->p.want_name_lists || !c->p.want_class_maps || !n) return BVCF_OK;  // nothing to render
  std::vector<uint32_t> off(n + 1);
  std::string text;
  for (uint32_t i = 0; i < n; i++) {
    off[i] = (uint32_t)text.size();
    text.append(names[i], lens[i]);
  }
  off[n] = (uint32_t)text.size();
  HIP_TRY(c, hipSetDevice(c->device));
  hipFree(c->d_name_off);
  hipFree(c->d_name_text);
  c->d_name_off = nullptr;
  c->d_name_text = nullptr;
  HIP_TRY(c, hipMalloc(&c->d_name_off, off.size() * sizeof(uint32_t)));
  HIP_TRY(c, hipMalloc(&c->d_name_text, text.size() + 16));
  HIP_TRY(c, hipMemcpy(c->d_name_off, off.data(), off.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_name_text, text.data(), text.size(), hipMemcpyHostToDevice));
  c->name_table.off = c->d_name_off;
  c->name_table.text = c->d_name_text;
  c->name_table.delim_len = (uint32_t)dl;
  memset(c->name_table.delim, 0, sizeof c->name_table.delim);
  memcpy(c->name_table.delim, delimiter, dl);
  c->names_on = true;
  for (auto &s : c->slots) {
    const int rc = alloc_names(c, s, std::max<uint64_t>(s.cap_names, c->p.max_batch_bytes / 2 + (1u << 20)));
    if (rc) return rc;
  }
  return BVCF_OK;
}

int bvcf_set_row_format(bvcf_ctx *c, const char *empty_field, int keep_pos, int keep_id, int keep_info) {
  if (!c) return BVCF_E_ARG;
  if (c->in_flight) {
    c->err = "bvcf_set_row_format with batches in flight";
    return BVCF_E_BUSY;
  }
  const char *empty = empty_field ? empty_field : "!";
  if (strlen(empty) > 16) {
    c->err = "bvcf_set_row_format: --emptyField longer than 16 bytes (render the rows on the host: render_sites = 0)";
    return BVCF_E_ARG;
  }
  // "chr" | "\tSNP\t" | the tail of a line without samples: three empty lists with their zero ratios, ac, an, sampleMaf
  // (main.go:612-670 with no carriers)
  std::string fmt = "chr\tSNP\t";
  std::string tail = "\t";
  for (int q = 0; q < 3; q++) {
    tail += empty;
    tail += "\t0\t";
  }
  tail += "0\t0\t0";
  fmt += tail;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!c->d_row_fmt) HIP_TRY(c, hipMalloc(&c->d_row_fmt, 128));
  HIP_TRY(c, hipMemcpy(c->d_row_fmt, fmt.data(), fmt.size(), hipMemcpyHostToDevice));
  c->row_tail_len = (uint32_t)tail.size();
  c->row_keep_pos = keep_pos != 0;
  c->row_keep_id = keep_id != 0;
  c->row_keep_info = keep_info != 0;
  c->row_fmt_set = true;
  // (now, while the caller is still setting up, not inside its first submits; the slots side by side: pinning is what
  // takes the time, and the runtime pins from several threads at once)
  std::vector<int> rcs(c->slots.size(), BVCF_OK);
  std::vector<std::string> errs(c->slots.size());  // (HIP_TRY's message of a thread goes to its own string: g_err_sink)
  std::vector<std::thread> th;
  for (size_t k = 0; k < c->slots.size(); k++)
    th.emplace_back([c, k, &rcs, &errs]() {
      g_err_sink = &errs[k];
      hipSetDevice(c->device);
      rcs[k] = ensure_render_buffers(c, c->slots[k]);
      g_err_sink = nullptr;
    });
  for (auto &t : th) t.join();
  for (size_t k = 0; k < rcs.size(); k++)
    if (rcs[k]) {
      c->err = errs[k];
      return rcs[k];
    }
  return BVCF_OK;
}

int bvcf_reserve(bvcf_ctx *c, uint64_t lines, uint64_t alleles, uint64_t cmap_bytes) {
  if (!c) return BVCF_E_ARG;
  if (c->in_flight) {
    c->err = "bvcf_reserve with batches in flight";
    return BVCF_E_BUSY;
  }
  if (lines > 0xFFFFFFF0ull || alleles > 0xFFFFFFF0ull) return BVCF_E_ARG;
  c->max_lines = std::max<uint64_t>(c->max_lines, lines);
  c->max_alleles = std::max<uint64_t>(std::max<uint64_t>(c->max_alleles, alleles), c->max_lines + 64);
  if (c->sites1 || c->packed)
    c->max_alleles = std::min<uint64_t>(std::max<uint64_t>(c->max_alleles, c->max_lines + c->need_extras + c->need_extras / 4 + 64), 0xFFFFFFF0ull);
  c->max_cmap = std::min<uint64_t>(std::max<uint64_t>(c->max_cmap, (cmap_bytes + 63) & ~63ull), 0xFFFFFF00ull);
  HIP_TRY(c, hipSetDevice(c->device));
  for (auto &s : c->slots) {
    int rc = alloc_results(c, s);
    if (rc) return rc;
  }
  return BVCF_OK;
}

// The very first batch of a ctx has no predecessor to tell the shape of the file's lines: when its text is on the host
// anyway, the first line with ten fields says it (a single-batch run of a GATK file would otherwise go through k_stream +
// k_gt).  Only a hint -- either kernel handles every line.
static void peek_line_shape(bvcf_ctx *c, const uint8_t *block, size_t nbytes) {
  c->shape_seen = true;
  if (c->gen_policy >= 0 || !c->fused || !c->n_samples) return;
  const uint8_t eol = (uint8_t)c->p.eol_byte;
  size_t ls = 0;
  for (int tries = 0; tries < 8 && ls < nbytes; tries++) {
    const uint8_t *e = (const uint8_t *)memchr(block + ls, eol, nbytes - ls);
    if (!e) return;
    const size_t le = (size_t)(e - block);  // the terminator
    size_t pos = ls;
    int tabs = 0;
    for (; pos < le && tabs < 9; pos++) tabs += block[pos] == '\t';
    if (tabs == 9) {
      const size_t cend = le + 1 - c->p.eol_chars;  // content end
      c->gen_mode = !(cend >= pos && cend - pos + 1 == 4ull * c->n_samples);
      return;
    }
    ls = le + 1;
  }
}

int bvcf_submit(bvcf_ctx *c, const uint8_t *block, size_t nbytes, uint64_t batch_seq) {
  if (!c || (!block && nbytes)) return BVCF_E_ARG;
  static const uint8_t empty = 0;
  if (!c->shape_seen && block && nbytes) peek_line_shape(c, block, nbytes);
  return submit_common(c, block ? block : &empty, nullptr, nbytes, batch_seq);
}

int bvcf_submit_device(bvcf_ctx *c, const void *dblock, size_t nbytes, uint64_t batch_seq) {
  if (!c || !dblock) return BVCF_E_ARG;
  return submit_common(c, nullptr, dblock, nbytes, batch_seq);
}

int bvcf_submit_bgzf(bvcf_ctx *c, const uint8_t *comp, size_t n_comp, size_t n_own, int flags, uint32_t first_off,
                     uint64_t batch_seq) {
  if (!c || !comp || !n_comp || n_own > n_comp) return BVCF_E_ARG;
  const bool skip_first_line = (flags & BVCF_BGZF_SKIP_FIRST_LINE) != 0;
  if (c->in_flight == c->slots.size()) {
    c->err = "all slots in flight";
    return BVCF_E_BUSY;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  // chains of older bgzf batches whose cut points have arrived go first (keeps the device busy between collects)
  for (size_t k = 0, i = c->tail; k < c->in_flight; k++, i = (i + 1) % c->slots.size()) {
    const int rc = launch_after_cuts(c, c->slots[i], false);
    if (rc) return rc;
  }
  std::vector<bvcf_bgzf::Block> blocks;
  const long used = bvcf_bgzf::scan(comp, n_comp, &blocks);
  if (used < 0 || (size_t)used != n_comp || blocks.empty()) {
    c->err = "bvcf_submit_bgzf: not whole BGZF blocks";
    return BVCF_E_ARG;
  }
  uint64_t total = 0, own = 0, own_bytes = 0;
  for (const auto &b : blocks) {
    if (own_bytes < n_own) {
      own_bytes += b.total;
      own += b.isize;
    }
    total += b.isize;
  }
  if (own_bytes != n_own) {
    c->err = "bvcf_submit_bgzf: n_own does not fall on a block boundary";
    return BVCF_E_ARG;
  }
  if (total > c->p.max_batch_bytes || total >= kMaxBlockBytes || n_comp >= kMaxBlockBytes) {
    c->err = "bgzf batch inflates to more than max_batch_bytes";
    return BVCF_E_TOO_BIG;
  }
  if (!skip_first_line && first_off > total) return BVCF_E_ARG;
  Slot &s = c->slots[c->head];
  int rc = alloc_results(c, s);
  if (rc) return rc;
  // ---- buffers of the compressed path, on first use / growth
  const size_t nb = blocks.size();
  if (!s.ev_cut) HIP_TRY(c, hipEventCreateWithFlags(&s.ev_cut, hipEventDisableTiming));
  if (!s.h_text) {
    // the host copies of the text, for every slot at once and side by side (pinning 64 MiB takes 13-25 ms).  With
    // samples only the line heads come back, a few percent of the text: a sixteenth of a batch to start with (bvcf_collect
    // grows a slot's buffer when a batch needs more)
    // (no samples, rows rendered on the device: only the lines left to the host come back -- an eighth of a batch and 1 MiB,
    // the size of the device's buffer for them; a batch that falls back to its whole text grows the slot's copy, bvcf_collect)
    const uint64_t want = c->n_samples ? c->p.max_batch_bytes / 16 + (1u << 20)
                                       : (c->render ? c->p.max_batch_bytes / 8 + (1u << 20) : c->p.max_batch_bytes + BVCF_DEVICE_PAD);
    std::vector<std::thread> th;
    for (auto &q : c->slots)
      if (!q.h_text)
        th.emplace_back([c, &q, want]() {
          hipSetDevice(c->device);
          if (hipHostMalloc(&q.h_text, want, hipHostMallocDefault) != hipSuccess) q.h_text = nullptr;
          q.cap_h_text = q.h_text ? want : 0;
        });
    for (auto &t : th) t.join();
    if (!s.h_text) {
      c->err = "hipHostMalloc failed (text copy of a BGZF batch)";
      return BVCF_E_NOMEM;
    }
  }
  if (c->n_samples && s.cap_head_lines < c->max_lines) {
    hipFree(s.d_head_off);
    hipHostFree(s.h_head_off);
    s.d_head_off = nullptr;
    s.h_head_off = nullptr;
    s.cap_head_lines = 0;
    HIP_TRY(c, hipMalloc(&s.d_head_off, (c->max_lines + 1) * sizeof(uint32_t)));
    HIP_TRY(c, hipHostMalloc(&s.h_head_off, (c->max_lines + 1) * sizeof(uint32_t), hipHostMallocDefault));
    if (!s.d_heads) HIP_TRY(c, hipMalloc(&s.d_heads, c->p.max_batch_bytes + 64));
    if (!s.d_head_total) {
      HIP_TRY(c, hipMalloc(&s.d_head_total, sizeof(unsigned long long)));
      HIP_TRY(c, hipHostMalloc(&s.h_head_total, sizeof(unsigned long long), hipHostMallocDefault));
    }
    s.cap_head_lines = c->max_lines;
  }
  if (!s.d_cuts) {
    HIP_TRY(c, hipMalloc(&s.d_cuts, 4 * sizeof(uint32_t)));
    HIP_TRY(c, hipHostMalloc(&s.h_cuts, 4 * sizeof(uint32_t), hipHostMallocDefault));
  }
  if (n_comp > s.cap_comp) {
    hipFree(s.d_comp);
    s.d_comp = nullptr;
    s.cap_comp = 0;
    const uint64_t want = std::max<uint64_t>(n_comp + n_comp / 4, 1u << 20);
    HIP_TRY(c, hipMalloc(&s.d_comp, want + 64));
    s.cap_comp = want;
  }
  if (nb > s.cap_bgzf_blocks) {
    hipFree(s.d_bgzf);
    hipHostFree(s.h_bgzf);
    s.d_bgzf = nullptr;
    s.h_bgzf = nullptr;
    s.cap_bgzf_blocks = 0;
    const uint64_t want = std::max<uint64_t>(nb + nb / 4, 4096);
    HIP_TRY(c, hipMalloc(&s.d_bgzf, want * 7 * sizeof(uint32_t)));
    HIP_TRY(c, hipHostMalloc(&s.h_bgzf, want * 5 * sizeof(uint32_t), hipHostMallocDefault));
    s.cap_bgzf_blocks = want;
  }
  // descriptors (4 words per block), then the expected CRCs; status[] and crc[] follow on the device
  BgzfDesc *h_desc = reinterpret_cast<BgzfDesc *>(s.h_bgzf);
  uint32_t *h_crc = s.h_bgzf + 4 * nb;
  uint64_t out_off = 0;
  for (size_t i = 0; i < nb; i++) {
    h_desc[i].in_off = blocks[i].in_off;
    h_desc[i].in_len = blocks[i].in_len;
    h_desc[i].out_off = (uint32_t)out_off;
    h_desc[i].isize = blocks[i].isize;
    h_crc[i] = blocks[i].crc;
    out_off += blocks[i].isize;
  }
  BgzfDesc *d_desc = reinterpret_cast<BgzfDesc *>(s.d_bgzf);
  uint32_t *d_want = s.d_bgzf + 4 * nb, *d_status = s.d_bgzf + 5 * nb, *d_crc = s.d_bgzf + 6 * nb;
  HIP_TRY(c, hipMemcpyAsync(s.d_comp, comp, n_comp, hipMemcpyHostToDevice, s.stream));
  HIP_TRY(c, hipMemcpyAsync(s.d_bgzf, s.h_bgzf, 5 * nb * sizeof(uint32_t), hipMemcpyHostToDevice, s.stream));
  // lines well under 16 KB (the batches so far say): the small-window kernel, so that two batches inflate side by side
  if (!launch_inflate(c->n_cu, s.d_comp, d_desc, (uint32_t)nb, s.d_in, d_status, d_crc, s.stream, c->avg_line_bytes && c->avg_line_bytes <= 12000)) {
    c->err = "bvcf_submit_bgzf: the CRC tables could not be placed on the device";
    return BVCF_E_NOMEM;
  }
  // the pad behind the text is read (and masked) by the scans: keep it defined
  HIP_TRY(c, hipMemsetAsync(s.d_in + total, '\n', BVCF_DEVICE_PAD, s.stream));
  CutArgs ca;
  ca.text = s.d_in;
  ca.total = (uint32_t)total;
  ca.own = (uint32_t)own;
  ca.skip_first = skip_first_line ? 1u : 0u;
  ca.at_eof = (flags & BVCF_BGZF_END_OF_STREAM) ? 1u : 0u;
  ca.first_off = first_off;
  ca.eol_byte = c->p.eol_byte;
  ca.n_blocks = (uint32_t)nb;
  ca.status = d_status;
  ca.crc = d_crc;
  ca.want_crc = d_want;
  ca.out = s.d_cuts;
  hipLaunchKernelGGL(k_cuts, dim3(1), dim3(kWave), 0, s.stream, ca);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(s.h_cuts, s.d_cuts, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s.stream));
  HIP_TRY(c, hipEventRecord(s.ev_cut, s.stream));
  s.await_cuts = true;
  s.is_bgzf = true;
  s.bgzf_rc = 0;
  s.text_total = (uint32_t)total;
  s.text_start = 0;
  s.busy = true;
  s.seq = batch_seq;
  s.nbytes = 0;
  c->head = (c->head + 1) % c->slots.size();
  c->in_flight++;
  return BVCF_OK;
}

int bvcf_collect(bvcf_ctx *c, bvcf_result *r) {
  if (!c || !r) return BVCF_E_ARG;
  if (!c->in_flight) {
    c->err = "nothing to collect";
    return BVCF_E_EMPTY;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  Slot &s = c->slots[c->tail];
  auto release = [&]() {
    s.busy = false;
    c->tail = (c->tail + 1) % c->slots.size();
    c->in_flight--;
  };
  const bool was_bgzf = s.is_bgzf;
  {
    const int rc2 = launch_after_cuts(c, s, true);
    if (rc2) {
      release();
      return rc2;
    }
  }
  hipError_t e = hipEventSynchronize(s.ev_ctr);
  if (e != hipSuccess) {
    c->err = std::string("kernel chain failed: ") + hipGetErrorString(e);
    release();
    return BVCF_E_HIP;
  }
  if (s.bgzf_rc) {
    const int rc2 = s.bgzf_rc;
    c->err = s.bgzf_err ? s.bgzf_err : "bgzf: batch refused";
    s.bgzf_rc = 0;
    release();
    return rc2;
  }
  memset(r, 0, sizeof *r);
  r->batch_seq = s.seq;
  r->n_samples = c->n_samples;
  r->cmap_stride = c->cmap_stride;
  float ms = 0;
  hipEventElapsedTime(&ms, s.ev_k0, s.ev_k1);
  r->kernel_ms = ms;
  const BatchCounters ctr = *s.h_counters;
  adapt_stream_kernel(c, s.used_gen, ctr);
  // slot i of alleles[] / tasks / class maps belongs to line i; the counters count the extras, which follow the lines'
  // slots (k_sites1 does not know the number of lines while it runs: there they follow slot max_lines)
  const uint64_t extras_at = (c->sites1 || c->packed) ? (uint64_t)s.cap_lines : (uint64_t)ctr.n_lines;
  const uint64_t n_alleles = extras_at + ctr.n_alleles;
  const uint64_t n_tasks = (uint64_t)ctr.n_lines + ctr.n_tasks;
  const uint64_t need_alleles = std::max<uint64_t>(std::max<uint64_t>(n_alleles, ctr.n_errs), n_tasks);
  const bool maps = c->p.want_class_maps && c->n_samples;
  const uint64_t cmap_bytes = !maps ? 0 : (c->fused ? (uint64_t)ctr.cmap_maps : n_tasks) * c->cmap_stride;
  if (ctr.pad[0]) {
    c->err = "internal error: streaming tile quota or class-map slot range exceeded";
    release();
    return BVCF_E_HIP;
  }
  if (ctr.n_lines > s.cap_lines || need_alleles > s.cap_alleles || cmap_bytes > s.cap_cmap) {
    r->status = BVCF_E_CAPACITY;
    r->need_lines = ctr.n_lines;
    r->need_alleles = need_alleles;
    r->need_cmap_bytes = cmap_bytes;
    // (the extras of a packed ctx follow slot cap_lines: once the lines grow, so does where they start -- bvcf_reserve
    // adds them to the NEW line capacity, need_alleles alone is relative to the old one)
    c->need_extras = (c->sites1 || c->packed) ? (uint64_t)ctr.n_alleles : 0;
    c->err = "batch exceeds reserved result capacity";
    release();
    return BVCF_E_CAPACITY;
  }
  // the packed form: a site record per line, full records (lines[], their first alleles) only for the n_full lines that
  // asked for them
  const uint32_t n_first = c->packed ? std::min<uint32_t>(ctr.n_full, ctr.n_lines) : ctr.n_lines;
  // (rendered rows: the site records stay on the device; the stream and the list of the lines left to the host come back)
  uint64_t row_bytes = 0, n_row_cuts = 0, n_ok_sites = 0;
  if (c->render) {
    row_bytes = s.h_rtotals[0];
    n_row_cuts = s.h_rtotals[1];
    n_ok_sites = s.h_rtotals[2];
    if (n_row_cuts > s.cap_row_cuts) {
      c->err = "internal error: more lines left to the host than the batch has lines";
      release();
      return BVCF_E_HIP;
    }
    if (n_row_cuts > s.cap_host_cuts) {
      const int rc = ensure_render_buffers(c, s, 0, n_row_cuts + n_row_cuts / 2 + 64);
      if (rc) {
        release();
        return rc;
      }
    }
    if (row_bytes > s.cap_rows) {
      // the stream was too small and k_render_rows wrote nothing: grow it and write again (the prefixes stand)
      const int rc = ensure_render_buffers(c, s, row_bytes + row_bytes / 4 + (1u << 20));
      if (rc) {
        release();
        return rc;
      }
      KernelArgs a = make_args(c, s, s.src, s.nbytes);
      hipLaunchKernelGGL(k_render_rows, dim3((uint32_t)c->n_cu * 8u), dim3(kWgThreads), 0, s.stream, make_render_args(c, s, a));
      HIP_TRY(c, hipGetLastError());
    }
    if (row_bytes) HIP_TRY(c, hipMemcpyAsync(s.h_rows, s.d_rows, row_bytes, hipMemcpyDeviceToHost, s.stream));
    if (n_row_cuts)
      HIP_TRY(c, hipMemcpyAsync(s.h_row_cuts, s.d_row_cuts, n_row_cuts * sizeof(bvcf_row_cut), hipMemcpyDeviceToHost, s.stream));
  } else if (c->packed && ctr.n_lines)
    HIP_TRY(c, hipMemcpyAsync(s.h_sites, s.d_sites, ctr.n_lines * sizeof(bvcf_site), hipMemcpyDeviceToHost, s.stream));
  if (c->packed) {
    // the host's copies hold the n_first full records and, right behind them, the further alleles (on the device those
    // follow slot cap_lines: rec_first is moved accordingly once they are here); grown when a batch needs more
    const uint64_t need_recs = std::max<uint64_t>(n_first, ((uint64_t)n_first + ctr.n_alleles + 1) / 2);  // (h_alleles holds 2 * hcap_recs)
    if (need_recs > s.hcap_recs || ctr.n_errs > s.hcap_errs) {
      const uint64_t want_recs = std::max<uint64_t>(s.hcap_recs, need_recs + need_recs / 2 + 64);
      const uint64_t want_errs = std::max<uint64_t>(s.hcap_errs, (uint64_t)ctr.n_errs + ctr.n_errs / 2 + 64);
      hipHostFree(s.h_lines);
      hipHostFree(s.h_alleles);
      hipHostFree(s.h_errs);
      s.h_lines = nullptr;
      s.h_alleles = nullptr;
      s.h_errs = nullptr;
      s.hcap_recs = s.hcap_errs = 0;
      if (hipHostMalloc(&s.h_lines, want_recs * sizeof(bvcf_line), hipHostMallocDefault) != hipSuccess ||
          hipHostMalloc(&s.h_alleles, 2 * want_recs * sizeof(bvcf_allele), hipHostMallocDefault) != hipSuccess ||
          hipHostMalloc(&s.h_errs, want_errs * sizeof(bvcf_err), hipHostMallocDefault) != hipSuccess) {
        c->err = "hipHostMalloc failed (full records of a packed batch)";
        s.cap_lines = 0;  // (alloc_results starts over at the slot's next use)
        release();
        return BVCF_E_NOMEM;
      }
      s.hcap_recs = want_recs;
      s.hcap_errs = want_errs;
    }
  }
  if (n_first)
    HIP_TRY(c, hipMemcpyAsync(s.h_lines, s.d_lines, (size_t)n_first * sizeof(bvcf_line), hipMemcpyDeviceToHost, s.stream));
  if (c->packed) {
    if (n_first)
      HIP_TRY(c, hipMemcpyAsync(s.h_alleles, s.d_alleles, (size_t)n_first * sizeof(bvcf_allele), hipMemcpyDeviceToHost, s.stream));
    if (ctr.n_alleles)
      HIP_TRY(c, hipMemcpyAsync(s.h_alleles + n_first, s.d_alleles + extras_at, (size_t)ctr.n_alleles * sizeof(bvcf_allele),
                                hipMemcpyDeviceToHost, s.stream));
  } else if (c->sites1) {
    if (n_first)
      HIP_TRY(c, hipMemcpyAsync(s.h_alleles, s.d_alleles, (size_t)n_first * sizeof(bvcf_allele), hipMemcpyDeviceToHost, s.stream));
    if (ctr.n_alleles)
      HIP_TRY(c, hipMemcpyAsync(s.h_alleles + extras_at, s.d_alleles + extras_at, (size_t)ctr.n_alleles * sizeof(bvcf_allele),
                                hipMemcpyDeviceToHost, s.stream));
  } else if (n_alleles)
    HIP_TRY(c, hipMemcpyAsync(s.h_alleles, s.d_alleles, n_alleles * sizeof(bvcf_allele), hipMemcpyDeviceToHost,
                              s.stream));
  if (ctr.n_errs)
    HIP_TRY(c, hipMemcpyAsync(s.h_errs, s.d_errs, ctr.n_errs * sizeof(bvcf_err), hipMemcpyDeviceToHost, s.stream));
  if (cmap_bytes)
    HIP_TRY(c, hipMemcpyAsync(s.h_cmap, s.d_cmap, cmap_bytes, hipMemcpyDeviceToHost, s.stream));
  if (c->dosage_stride && n_alleles)
    HIP_TRY(c, hipMemcpyAsync(s.h_dosage, s.d_dosage, n_alleles * c->dosage_stride, hipMemcpyDeviceToHost, s.stream));
  uint64_t text_bytes = 0;
  if (was_bgzf && s.heads) {
    // the packed line heads and where each line's is
    text_bytes = *s.h_head_total;
    if (text_bytes > c->p.max_batch_bytes) {
      c->err = "internal error: line heads larger than the batch";
      release();
      return BVCF_E_HIP;
    }
    if (text_bytes > s.cap_h_text) {  // (the caller is done with what this slot returned n_slots collects ago)
      hipHostFree(s.h_text);
      s.h_text = nullptr;
      s.cap_h_text = 0;
      const uint64_t want = std::min<uint64_t>(text_bytes + text_bytes / 2, c->p.max_batch_bytes + BVCF_DEVICE_PAD);
      if (hipHostMalloc(&s.h_text, want, hipHostMallocDefault) != hipSuccess) {
        s.h_text = nullptr;
        c->err = "hipHostMalloc failed (text copy of a BGZF batch)";
        release();
        return BVCF_E_NOMEM;
      }
      s.cap_h_text = want;
    }
    if (ctr.n_lines) HIP_TRY(c, hipMemcpyAsync(s.h_head_off, s.d_head_off, ctr.n_lines * sizeof(uint32_t), hipMemcpyDeviceToHost, s.stream));
    if (text_bytes) HIP_TRY(c, hipMemcpyAsync(s.h_text, s.d_heads, text_bytes, hipMemcpyDeviceToHost, s.stream));
  } else if (was_bgzf && c->render && s.cut_text_on && s.h_rtotals[3] <= s.cap_cut_text && s.h_rtotals[3] < 0xFFFFFFFFull) {
    // rendered rows: the host only reads the lines left to it -- their bytes, packed (bvcf_row_cut.text_off), not the
    // batch's whole text
    text_bytes = s.h_rtotals[3];
    if (text_bytes) HIP_TRY(c, hipMemcpyAsync(s.h_text, s.d_cut_text, text_bytes, hipMemcpyDeviceToHost, s.stream));
  } else if (was_bgzf && s.nbytes) {
    text_bytes = s.nbytes;
    if (text_bytes > s.cap_h_text) {  // (a rendered ctx keeps a small copy buffer: see bvcf_submit_bgzf)
      hipHostFree(s.h_text);
      s.h_text = nullptr;
      s.cap_h_text = 0;
      const uint64_t want = c->p.max_batch_bytes + BVCF_DEVICE_PAD;
      if (hipHostMalloc(&s.h_text, want, hipHostMallocDefault) != hipSuccess) {
        s.h_text = nullptr;
        c->err = "hipHostMalloc failed (text copy of a BGZF batch)";
        release();
        return BVCF_E_NOMEM;
      }
      s.cap_h_text = want;
    }
    HIP_TRY(c, hipMemcpyAsync(s.h_text, s.src, s.nbytes, hipMemcpyDeviceToHost, s.stream));
  }
  const bool names = c->names_on && s.d_name_lists;
  uint64_t name_bytes = 0;
  if (names && n_alleles) {
    name_bytes = *s.h_name_total;
    if (name_bytes >= 0xFFFFFFF0ull) {
      c->err = "the sample-name lists of one batch pass 4 GiB: submit smaller blocks";
      release();
      return BVCF_E_TOO_BIG;
    }
    if (name_bytes > s.cap_names) {
      // the arena was too small and k_name_write wrote nothing: grow it and write again (the offsets stand)
      const int rc = alloc_name_arena(c, s, name_bytes + name_bytes / 4 + (1u << 20));
      if (rc) {
        release();
        return rc;
      }
      KernelArgs a = make_args(c, s, s.src, s.nbytes);
      hipLaunchKernelGGL(k_name_write, dim3(c->n_cu * 4), dim3(kWgThreads), 0, s.stream, a, make_name_args(c, s));
    }
    HIP_TRY(c, hipMemcpyAsync(s.h_name_lists, s.d_name_lists, n_alleles * sizeof(bvcf_names), hipMemcpyDeviceToHost, s.stream));
    if (name_bytes) HIP_TRY(c, hipMemcpyAsync(s.h_names, s.d_names, name_bytes, hipMemcpyDeviceToHost, s.stream));
  }
  e = hipStreamSynchronize(s.stream);
  if (e != hipSuccess) {
    c->err = std::string("result copy failed: ") + hipGetErrorString(e);
    release();
    return BVCF_E_HIP;
  }
  // getAlleles' messages were recorded before the field-count verdict was known: a line that
  // fails linePasses (main.go:537-539) never reaches getAlleles, so its messages are dropped here
  // (packed form: the verdict of line i is in its site record, or in the full record that one points at)
  auto verdict_of = [&](uint32_t i) -> uint32_t {
    if (!c->packed) return s.h_lines[i].status;
    if (c->render) {
      // (only lines with full records log anything: found among the cuts, which are in line order)
      const bvcf_row_cut *lo = s.h_row_cuts, *hi = s.h_row_cuts + n_row_cuts;
      const bvcf_row_cut *it = std::lower_bound(lo, hi, i, [](const bvcf_row_cut &q, uint32_t v) { return q.line < v; });
      if (it == hi || it->line != i || it->slot >= n_first) return (uint32_t)BVCF_LINE_FIELDS;
      return s.h_lines[it->slot].status;
    }
    const bvcf_site &st = s.h_sites[i];
    if (!(st.status & BVCF_SITE_FULL)) return st.status;
    return st.full_idx < n_first ? s.h_lines[st.full_idx].status : (uint32_t)BVCF_LINE_FIELDS;
  };
  if (c->packed)  // (the further alleles sit right behind the n_first first records here, behind slot cap_lines on the device)
    for (uint32_t j = 0; j < n_first; j++)
      if (s.h_lines[j].rec_first >= extras_at) s.h_lines[j].rec_first = s.h_lines[j].rec_first - (uint32_t)extras_at + n_first;
  uint32_t n_errs = 0;
  for (uint32_t i = 0; i < ctr.n_errs; i++) {
    const bvcf_err &er = s.h_errs[i];
    if (er.line < ctr.n_lines && verdict_of(er.line) != BVCF_LINE_FIELDS) s.h_errs[n_errs++] = er;
  }
  r->status = BVCF_OK;
  r->n_lines = ctr.n_lines;
  r->n_alleles = c->packed ? n_first + ctr.n_alleles : (uint32_t)n_alleles;
  r->n_errs = n_errs;
  r->n_cmap_bytes = cmap_bytes;
  r->n_lines_seen = ctr.lines_seen;
  r->lines = s.h_lines;
  r->alleles = s.h_alleles;
  r->errs = s.h_errs;
  r->cmap = s.h_cmap;
  r->dosage = c->dosage_stride ? s.h_dosage : nullptr;
  r->dosage_stride = c->dosage_stride;
  r->text = was_bgzf ? s.h_text : nullptr;
  r->n_text_bytes = was_bgzf ? text_bytes : 0;
  r->head_off = (was_bgzf && s.heads) ? s.h_head_off : nullptr;
  r->sites = (c->packed && !c->render) ? s.h_sites : nullptr;
  r->n_full_lines = c->packed ? n_first : 0u;
  r->rows = c->render ? s.h_rows : nullptr;
  r->n_row_bytes = row_bytes;
  r->row_cuts = c->render ? s.h_row_cuts : nullptr;
  r->n_row_cuts = (uint32_t)n_row_cuts;
  r->n_ok_sites = n_ok_sites;
  r->name_lists = names ? s.h_name_lists : nullptr;
  r->names = names ? s.h_names : nullptr;
  r->n_name_bytes = name_bytes;

  uint64_t ok = 0, ac0 = 0, recs = 0;
  if (c->render) {
    ok += n_ok_sites;
    recs += n_ok_sites;
  } else if (c->packed)
    for (uint32_t i = 0; i < ctr.n_lines; i++) {
      const bvcf_site &st = s.h_sites[i];
      if (!(st.status & BVCF_SITE_FULL)) {
        ok += st.status == BVCF_LINE_OK;
        recs += st.status == BVCF_LINE_OK;
      }
    }
  for (uint32_t i = 0; i < n_first; i++) {
    const bvcf_line &L = s.h_lines[i];
    if (L.status != BVCF_LINE_OK) continue;
    ok++;
    recs += L.n_rec;
    if (c->n_samples)
      for (uint32_t j = 0; j < L.n_rec; j++) ac0 += s.h_alleles[j ? L.rec_first + j - 1 : i].ac == 0;
  }
  if (ctr.lines_seen) c->avg_line_bytes = s.nbytes / ctr.lines_seen;
  c->totals[0] += ctr.lines_seen;
  c->totals[1] += ok;
  c->totals[2] += recs;
  c->totals[3] += ac0;
  c->totals[4] += n_errs;
  c->totals[5] += s.nbytes;
  c->totals[6] += r->n_cmap_bytes;
  c->totals[7] += (uint64_t)(ms * 1e6);
  release();
  return BVCF_OK;
}

int bvcf_path(const bvcf_ctx *c) { return c ? (c->fused ? 2 : 1) : BVCF_E_ARG; }
int bvcf_bench_stream_kernel(const bvcf_ctx *c) { return (c && c->fused) ? (c->gen_mode ? 1 : 0) : -1; }

int bvcf_counters(bvcf_ctx *c, uint64_t out[8]) {
  if (!c || !out) return BVCF_E_ARG;
  memcpy(out, c->totals, sizeof c->totals);
  return BVCF_OK;
}

int bvcf_sum_counters(bvcf_ctx *const *ctxs, int n, uint64_t out[8]) {
  if (!ctxs || n < 0 || !out) return BVCF_E_ARG;
  for (int k = 0; k < 8; k++) out[k] = 0;
  for (int i = 0; i < n; i++) {
    if (!ctxs[i]) return BVCF_E_ARG;
    for (int k = 0; k < 8; k++) out[k] += ctxs[i]->totals[k];
  }
  return BVCF_OK;
}

int bvcf_bgzf_inflate_device(int device, const uint8_t *comp, size_t n_comp, uint8_t *out, size_t cap, size_t *n_out) {
  if ((!comp && n_comp) || !n_out || (!out && cap)) return BVCF_E_ARG;
  *n_out = 0;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) return BVCF_E_NODEV;
  std::vector<bvcf_bgzf::Block> blocks;
  const long used = bvcf_bgzf::scan(comp, n_comp, &blocks);
  if (used < 0 || (size_t)used != n_comp) return BVCF_E_FATAL;  // not BGZF, or a truncated last block
  if (n_comp >= 0xFFF00000ull) return BVCF_E_TOO_BIG;
  std::vector<BgzfDesc> desc(blocks.size());
  uint64_t total = 0;
  for (size_t i = 0; i < blocks.size(); i++) {
    desc[i].in_off = blocks[i].in_off;
    desc[i].in_len = blocks[i].in_len;
    desc[i].out_off = (uint32_t)total;
    desc[i].isize = blocks[i].isize;
    total += blocks[i].isize;
  }
  *n_out = (size_t)total;
  if (total > cap || total >= 0xFFF00000ull) return BVCF_E_TOO_BIG;
  if (blocks.empty()) return BVCF_OK;
  if (hipSetDevice(device) != hipSuccess) return BVCF_E_HIP;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return BVCF_E_HIP;
  uint8_t *d_comp = nullptr, *d_text = nullptr;
  BgzfDesc *d_desc = nullptr;
  uint32_t *d_status = nullptr, *d_crc = nullptr;
  const size_t nb = blocks.size();
  int rc = BVCF_OK;
  std::vector<uint32_t> status(nb), crc(nb);
  if (hipMalloc(&d_comp, n_comp + 64) != hipSuccess || hipMalloc(&d_text, total + 64) != hipSuccess ||
      hipMalloc(&d_desc, nb * sizeof(BgzfDesc)) != hipSuccess || hipMalloc(&d_status, nb * 4) != hipSuccess ||
      hipMalloc(&d_crc, nb * 4) != hipSuccess) {
    rc = BVCF_E_NOMEM;
  } else if (hipMemcpy(d_comp, comp, n_comp, hipMemcpyHostToDevice) != hipSuccess ||
             hipMemcpy(d_desc, desc.data(), nb * sizeof(BgzfDesc), hipMemcpyHostToDevice) != hipSuccess) {
    rc = BVCF_E_HIP;
  } else {
    if (!launch_inflate(prop.multiProcessorCount, d_comp, d_desc, (uint32_t)nb, d_text, d_status, d_crc, nullptr, false)) rc = BVCF_E_NOMEM;
    if (rc || hipDeviceSynchronize() != hipSuccess || hipMemcpy(status.data(), d_status, nb * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(crc.data(), d_crc, nb * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        (total && hipMemcpy(out, d_text, total, hipMemcpyDeviceToHost) != hipSuccess))
      rc = rc ? rc : BVCF_E_HIP;
  }
  hipFree(d_comp);
  hipFree(d_text);
  hipFree(d_desc);
  hipFree(d_status);
  hipFree(d_crc);
  if (rc) return rc;
  for (size_t i = 0; i < nb; i++)
    if (status[i] != kInfOk || crc[i] != blocks[i].crc) return BVCF_E_FATAL;  // corrupt block (inflate or CRC mismatch)
  return BVCF_OK;
}

int bvcf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// The final count gather.  One process drives all ctxs, so the communicator is built with
// ncclCommInitAll and the n all-reduces are issued as one group.
int bvcf_allreduce_counters(bvcf_ctx *const *ctxs, int n, uint64_t out[8], int *used_rccl) {
  if (used_rccl) *used_rccl = 0;
  if (!ctxs || n < 0 || !out) return BVCF_E_ARG;
  for (int i = 0; i < n; i++)
    if (!ctxs[i]) return BVCF_E_ARG;
  bool distinct = true;
  for (int i = 0; i < n && distinct; i++)
    for (int j = 0; j < i; j++)
      if (ctxs[i]->device == ctxs[j]->device) distinct = false;
  const char *force = getenv("BVCF_RCCL");
  const bool want = distinct && (n >= 2 || (n == 1 && force && *force == '1'));
  if (!want) return bvcf_sum_counters(ctxs, n, out);

  bvcf_ctx *c0 = ctxs[0];
  // dlopen by SONAME: a process that already holds an RCCL (torch ships its own copy) gets that one
  static void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) {
    c0->err = std::string("dlopen librccl.so.1: ") + dlerror();
    return BVCF_E_HIP;
  }
  auto p_init = (decltype(&ncclCommInitAll))dlsym(lib, "ncclCommInitAll");
  auto p_destroy = (decltype(&ncclCommDestroy))dlsym(lib, "ncclCommDestroy");
  auto p_allreduce = (decltype(&ncclAllReduce))dlsym(lib, "ncclAllReduce");
  auto p_gstart = (decltype(&ncclGroupStart))dlsym(lib, "ncclGroupStart");
  auto p_gend = (decltype(&ncclGroupEnd))dlsym(lib, "ncclGroupEnd");
  auto p_errstr = (decltype(&ncclGetErrorString))dlsym(lib, "ncclGetErrorString");
  if (!p_init || !p_destroy || !p_allreduce || !p_gstart || !p_gend || !p_errstr) {
    c0->err = "librccl.so.1 lacks an expected symbol";
    return BVCF_E_HIP;
  }
  std::vector<int> devs(n);
  std::vector<ncclComm_t> comms(n, nullptr);
  std::vector<uint64_t *> d_buf(n, nullptr);
  for (int i = 0; i < n; i++) devs[i] = ctxs[i]->device;
  int rc = BVCF_OK;
  auto nccl_fail = [&](ncclResult_t r, const char *what) {
    c0->err = std::string(what) + ": " + p_errstr(r);
    rc = BVCF_E_HIP;
  };
  ncclResult_t r = p_init(comms.data(), n, devs.data());
  if (r != ncclSuccess) {
    nccl_fail(r, "ncclCommInitAll");
    return rc;
  }
  for (int i = 0; i < n && rc == BVCF_OK; i++) {
    if (hipSetDevice(devs[i]) != hipSuccess || hipMalloc(&d_buf[i], 8 * sizeof(uint64_t)) != hipSuccess ||
        hipMemcpy(d_buf[i], ctxs[i]->totals, 8 * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) {
      c0->err = "bvcf_allreduce_counters: counter upload failed";
      rc = BVCF_E_HIP;
    }
  }
  if (rc == BVCF_OK) {
    r = p_gstart();
    for (int i = 0; i < n && r == ncclSuccess; i++) {
      hipSetDevice(devs[i]);
      r = p_allreduce(d_buf[i], d_buf[i], 8, ncclUint64, ncclSum, comms[i], ctxs[i]->slots[0].stream);
    }
    const ncclResult_t r2 = p_gend();
    if (r != ncclSuccess || r2 != ncclSuccess) nccl_fail(r != ncclSuccess ? r : r2, "ncclAllReduce");
  }
  for (int i = 0; i < n && rc == BVCF_OK; i++) {
    hipSetDevice(devs[i]);
    if (hipStreamSynchronize(ctxs[i]->slots[0].stream) != hipSuccess) {
      c0->err = "bvcf_allreduce_counters: all-reduce failed";
      rc = BVCF_E_HIP;
    }
  }
  if (rc == BVCF_OK) {
    hipSetDevice(devs[0]);
    if (hipMemcpy(out, d_buf[0], 8 * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) {
      c0->err = "bvcf_allreduce_counters: counter download failed";
      rc = BVCF_E_HIP;
    }
  }
  for (int i = 0; i < n; i++) {
    hipSetDevice(devs[i]);
    if (d_buf[i]) hipFree(d_buf[i]);
    if (comms[i]) p_destroy(comms[i]);
  }
  if (rc == BVCF_OK && used_rccl) *used_rccl = 1;
  return rc;
}

int bvcf_bench_device(bvcf_ctx *c, const void *const *dblocks, const size_t *nbytes, int n_blocks, int iters,
                      float *chain_ms, float *gt_ms, uint64_t counts[5]) {
  return bvcf_bench_device_slots(c, dblocks, nbytes, n_blocks, iters, 0, chain_ms, gt_ms, counts);
}

int bvcf_bench_device_slots(bvcf_ctx *c, const void *const *dblocks, const size_t *nbytes, int n_blocks, int iters,
                            uint32_t slots_in_use, float *chain_ms, float *gt_ms, uint64_t counts[5]) {
  if (!c || !dblocks || !nbytes || n_blocks < 1 || iters < 1) return BVCF_E_ARG;
  if (c->in_flight) {
    c->err = "bvcf_bench_device with batches in flight";
    return BVCF_E_BUSY;
  }
  for (int b = 0; b < n_blocks; b++)
    if (!dblocks[b] || nbytes[b] > c->p.max_batch_bytes || nbytes[b] >= kMaxBlockBytes) return BVCF_E_TOO_BIG;
  HIP_TRY(c, hipSetDevice(c->device));
  // Batch i runs on slot i % n_use, each slot on its own stream, exactly as bvcf_submit deals them: with two
  // slots the short latency-bound kernels that end one batch's chain overlap the next batch's scan.
  const size_t n_use = slots_in_use ? std::min<size_t>(slots_in_use, c->slots.size()) : c->slots.size();
  for (size_t k = 0; k < n_use; k++) {
    int rc = alloc_results(c, c->slots[k]);
    if (rc) return rc;
  }
  Slot &s = c->slots[(size_t)(iters - 1) % n_use];  // the slot whose counters are reported
  std::vector<hipEvent_t> ev((size_t)iters * 4);
  for (auto &e : ev) HIP_TRY(c, hipEventCreate(&e));
  for (int i = 0; i < iters; i++) {
    Slot &si = c->slots[(size_t)i % n_use];
    si.s2_parity ^= 1u;
    KernelArgs a = make_args(c, si, (const uint8_t *)dblocks[i % n_blocks], nbytes[i % n_blocks]);
    HIP_TRY(c, hipEventRecord(ev[4 * i], si.stream));
    launch_chain(c, a, si.stream, ev[4 * i + 1], ev[4 * i + 2], &si);
    HIP_TRY(c, hipEventRecord(ev[4 * i + 3], si.stream));
  }
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(s.h_counters, s.d_counters, sizeof(BatchCounters), hipMemcpyDeviceToHost, s.stream));
  for (size_t k = 0; k < n_use; k++) HIP_TRY(c, hipStreamSynchronize(c->slots[k].stream));
  for (int i = 0; i < iters; i++) {
    float t0 = 0, t1 = 0;
    hipEventElapsedTime(&t0, ev[4 * i], ev[4 * i + 3]);
    hipEventElapsedTime(&t1, ev[4 * i + 1], ev[4 * i + 2]);
    if (chain_ms) chain_ms[i] = t0;
    if (gt_ms) gt_ms[i] = t1;
  }
  for (auto &e : ev) hipEventDestroy(e);
  const BatchCounters ctr = *s.h_counters;
  adapt_stream_kernel(c, c->gen_mode, ctr);  // (every launch of this call went through the same kernel)
  if (counts) {
    counts[0] = ctr.n_lines;
    counts[1] = (uint64_t)ctr.n_lines + ctr.n_alleles;
    counts[2] = ctr.n_errs;
    counts[3] = (c->fused ? (uint64_t)ctr.cmap_maps : (uint64_t)ctr.n_lines + ctr.n_tasks) * c->cmap_stride;
    counts[4] = (uint64_t)ctr.n_lines + ctr.n_tasks;
  }
  const uint64_t b_need = std::max<uint64_t>(std::max<uint64_t>((uint64_t)ctr.n_lines + ctr.n_alleles, ctr.n_errs),
                                             (uint64_t)ctr.n_lines + ctr.n_tasks);
  if ((ctr.n_lines > s.cap_lines || b_need > s.cap_alleles ||
      (c->p.want_class_maps && c->n_samples &&
       (c->fused ? (uint64_t)ctr.cmap_maps : (uint64_t)ctr.n_lines + ctr.n_tasks) * c->cmap_stride > s.cap_cmap))) {
    c->err = "bench block exceeds reserved result capacity: lines " + std::to_string(ctr.n_lines) + " records " +
             std::to_string(b_need) + " maps " + std::to_string(ctr.cmap_maps);
    return BVCF_E_CAPACITY;
  }
  return BVCF_OK;
}

}  // extern "C"

#ifdef BVCF_EXP_GT_KINDS
// tasks k_gt ran since the last call, by kind (see g_gt_kinds)
extern "C" int bvcf_debug_gt_kinds(unsigned int out[4]) {
  const unsigned int z[4] = {0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(bvcf_dev::g_gt_kinds), sizeof(z), 0, hipMemcpyDeviceToHost) != hipSuccess) return BVCF_E_HIP;
  return hipMemcpyToSymbol(HIP_SYMBOL(bvcf_dev::g_gt_kinds), z, sizeof(z), 0, hipMemcpyHostToDevice) == hipSuccess ? BVCF_OK : BVCF_E_HIP;
}
#endif
#ifdef BVCF_EXP_TIMES
extern "C" int bvcf_debug_head_times(unsigned long long *out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bvcf_dev::g_head_t), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost);
}
extern "C" int bvcf_debug_phase_times(unsigned long long *out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bvcf_dev::g_phase_t), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost);
}
extern "C" int bvcf_debug_wave_hw(unsigned int *out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bvcf_dev::g_wave_hw), sizeof(unsigned int) * n, 0, hipMemcpyDeviceToHost);
}
extern "C" int bvcf_debug_wave_times(unsigned long long *out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bvcf_dev::g_wave_t), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost);
}
#endif
