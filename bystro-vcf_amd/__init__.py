"""bystro-vcf_amd — Python binding of libbvcf.so (the MI355X-native per-line variant pipeline).

This package is plumbing for tests and benchmarks: every call goes through the C-ABI declared in
include/bvcf.h.  There is no Python or CPU implementation of the path here; if the HIP library is
missing the import fails.
"""
import ctypes as C
import os

import numpy as np

# One HIP runtime per process: torch bundles its own libamdhip64 (same SONAME as /opt/rocm's).
# Loading torch first makes libbvcf.so bind to that copy instead of pulling in a second runtime,
# which would leave the later one without a usable device.
try:
    import torch  # noqa: F401
except ImportError:  # the library then binds to /opt/rocm's runtime through its RUNPATH
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BVCF_LIB") or os.path.join(_HERE, "libbvcf.so")  # BVCF_LIB: A/B another build

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "bystro-vcf_amd: %s not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C bystro-vcf_amd/csrc`); there is no fallback path" % LIB_PATH)

lib = C.CDLL(LIB_PATH)

ABI_VERSION = 7
DEVICE_PAD = 64
NO_CMAP = 0xFFFFFFFF

OK, E_ARG, E_HIP, E_NODEV, E_BUSY, E_EMPTY, E_TOO_BIG, E_CAPACITY, E_NOMEM, E_FATAL = 0, -1, -2, -3, -4, -5, -6, -7, -8, -9
LINE_OK, LINE_FIELDS, LINE_FILTER, LINE_NOALLELE = 0, 1, 2, 3
SITE_NAMES = ["SNP", "INS", "DEL", "MNP", "MULTIALLELIC"]
ALT_BASE, ALT_INS, ALT_DEL = 0, 1, 2
CLS_NONE, CLS_HET, CLS_HOM, CLS_MISSING = 0, 1, 2, 3

# the measurement hooks of include/bvcf_bench.h (not part of the drop-in ABI)
BENCH_EXPORTS = ["bvcf_bench_device", "bvcf_bench_device_slots", "bvcf_bench_stream_kernel"]

# the partition logic of bvcf_run_fd, exported for host-only tests (include/bvcf_plan.h; not part of the drop-in ABI)
PLAN_EXPORTS = ["bvcf_plan_text_ranges", "bvcf_plan_bgzf_ranges", "bvcf_cut_text_range", "bvcf_find_bgzf_chain",
                "bvcf_plan_threads", "bvcf_plan_fd"]

# every symbol include/bvcf.h declares
EXPORTS = [
    "bvcf_create", "bvcf_destroy", "bvcf_last_error", "bvcf_version", "bvcf_reserve", "bvcf_set_sample_names", "bvcf_set_row_format", "bvcf_alloc_pinned", "bvcf_alloc_pinned_near", "bvcf_warmup",
    "bvcf_free_pinned", "bvcf_submit", "bvcf_submit_device", "bvcf_submit_bgzf", "bvcf_collect", "bvcf_counters", "bvcf_sum_counters",
    "bvcf_allreduce_counters", "bvcf_device_count", "bvcf_device_pci_bus_id", "bvcf_path", "bvcf_config_defaults", "bvcf_string_header", "bvcf_format_tsv", "bvcf_run_buffer", "bvcf_run_fd", "bvcf_decompress_fd", "bvcf_bgzf_inflate_device", "bvcf_free",
    "bvcf_arrow_open", "bvcf_arrow_append", "bvcf_arrow_close",
]


class Params(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("device", C.c_int32), ("n_header_fields", C.c_uint32),
        ("eol_chars", C.c_uint32), ("eol_byte", C.c_uint8), ("want_class_maps", C.c_uint8),
        ("want_dosage", C.c_uint8), ("want_name_lists", C.c_uint8), ("allow_filter", C.c_char_p), ("exclude_filter", C.c_char_p),
        ("max_batch_bytes", C.c_uint64), ("max_lines", C.c_uint32), ("max_alleles", C.c_uint32),
        ("cmap_bytes", C.c_uint64), ("n_slots", C.c_uint32), ("path", C.c_uint32),
        ("packed_sites", C.c_uint32), ("render_sites", C.c_uint32),
    ]


class Config(C.Structure):
    _fields_ = [
        ("empty_field", C.c_char_p), ("field_delimiter", C.c_char_p), ("allow_filter", C.c_char_p),
        ("exclude_filter", C.c_char_p), ("keep_id", C.c_uint8), ("keep_info", C.c_uint8),
        ("keep_pos", C.c_uint8), ("keep_qual", C.c_uint8), ("normalize_header", C.c_uint8),
        ("leave_teardown_to_exit", C.c_uint8), ("reserved", C.c_uint8 * 2), ("device", C.c_int32), ("n_format_threads", C.c_uint32),
        ("max_batch_bytes", C.c_uint64), ("sample_list_path", C.c_char_p),
        ("dosage_path", C.c_char_p), ("no_out", C.c_uint8), ("reserved3", C.c_uint8 * 3),
        ("n_devices", C.c_uint32), ("devices", C.POINTER(C.c_int32)),
    ]


class Result(C.Structure):
    _fields_ = [
        ("batch_seq", C.c_uint64), ("status", C.c_int32), ("n_lines", C.c_uint32), ("n_alleles", C.c_uint32),
        ("n_errs", C.c_uint32), ("n_cmap_bytes", C.c_uint64), ("cmap_stride", C.c_uint32),
        ("n_samples", C.c_uint32), ("lines", C.c_void_p), ("alleles", C.c_void_p), ("errs", C.c_void_p),
        ("cmap", C.c_void_p), ("need_lines", C.c_uint64), ("need_alleles", C.c_uint64),
        ("need_cmap_bytes", C.c_uint64), ("kernel_ms", C.c_float), ("reserved", C.c_uint32),
        ("n_lines_seen", C.c_uint64), ("dosage", C.c_void_p), ("dosage_stride", C.c_uint32), ("reserved2", C.c_uint32),
        ("name_lists", C.c_void_p), ("names", C.c_void_p), ("n_name_bytes", C.c_uint64),
        ("text", C.c_void_p), ("n_text_bytes", C.c_uint64), ("head_off", C.c_void_p),
        ("sites", C.c_void_p), ("n_full_lines", C.c_uint32), ("n_row_cuts", C.c_uint32),
        ("rows", C.c_void_p), ("n_row_bytes", C.c_uint64), ("row_cuts", C.c_void_p), ("n_ok_sites", C.c_uint64),
    ]


ROW_CUT_DTYPE = np.dtype([("line", "<u4"), ("slot", "<u4"), ("off", "<u8"), ("text_off", "<u4"), ("reserved", "<u4")])
NO_TEXT_OFF = 0xFFFFFFFF
LINE_DTYPE = np.dtype([
    ("off", "<u4"), ("len", "<u4"), ("fend", "<u4", (9,)), ("rec_first", "<u4"), ("n_rec", "<u4"),
    ("n_fields", "<u4"), ("gt_task", "<u4"), ("status", "u1"), ("site_type", "u1"), ("pad", "u1", (2,))])
ALLELE_DTYPE = np.dtype([
    ("pos", "<i8"), ("line", "<u4"), ("alt_idx", "<u4"), ("alt_off", "<u4"), ("alt_len", "<u4"), ("ac", "<u4"),
    ("an", "<u4"), ("n_het", "<u4"), ("n_hom", "<u4"), ("n_miss", "<u4"), ("cmap_off", "<u4"), ("ref", "u1"),
    ("alt_base", "u1"), ("kind", "u1"), ("site_type", "u1"), ("trtv", "u1"), ("flags", "u1"), ("pad", "u1", (2,)),
    ("gt_task", "<u4"), ("pad2", "<u4")])
ERR_DTYPE = np.dtype([("line", "<u4"), ("alt_no", "<u4"), ("code", "<u4"), ("pad", "<u4")])
SITE_DTYPE = np.dtype([("off", "<u4"), ("len", "<u4"), ("fend", "u1", (8,)), ("ref", "u1"), ("alt_base", "u1"), ("trtv", "u1"),
                       ("status", "u1"), ("full_idx", "<u4"), ("n_fields", "<u4"), ("reserved", "<u4")])
SITE_FULL = 0x80
NAMES_DTYPE = np.dtype([("off", "<u4", (3,)), ("len", "<u4", (3,))])
assert LINE_DTYPE.itemsize == 64 and ALLELE_DTYPE.itemsize == 64 and ERR_DTYPE.itemsize == 16 and SITE_DTYPE.itemsize == 32

lib.bvcf_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(Params)]
lib.bvcf_create.restype = C.c_int
lib.bvcf_destroy.argtypes = [C.c_void_p]
lib.bvcf_destroy.restype = None
lib.bvcf_last_error.argtypes = [C.c_void_p]
lib.bvcf_last_error.restype = C.c_char_p
lib.bvcf_version.restype = C.c_char_p
lib.bvcf_reserve.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
lib.bvcf_set_sample_names.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_char_p]
lib.bvcf_alloc_pinned.argtypes = [C.c_size_t]
lib.bvcf_alloc_pinned.restype = C.c_void_p
lib.bvcf_free_pinned.argtypes = [C.c_void_p]
lib.bvcf_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64]
lib.bvcf_submit_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64]
lib.bvcf_submit_bgzf.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, C.c_uint32, C.c_uint64]
lib.bvcf_collect.argtypes = [C.c_void_p, C.POINTER(Result)]
lib.bvcf_bench_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_int,
                                  C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
lib.bvcf_bench_device_slots.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_int,
                                        C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
lib.bvcf_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
lib.bvcf_path.argtypes = [C.c_void_p]
lib.bvcf_sum_counters.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_uint64)]
lib.bvcf_allreduce_counters.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
lib.bvcf_device_count.restype = C.c_int
lib.bvcf_config_defaults.argtypes = [C.POINTER(Config)]
lib.bvcf_config_defaults.restype = None
lib.bvcf_string_header.argtypes = [C.POINTER(Config), C.c_char_p, C.c_size_t]
lib.bvcf_string_header.restype = C.c_size_t
lib.bvcf_format_tsv.argtypes = [C.POINTER(Config), C.POINTER(Result), C.c_void_p, C.POINTER(C.c_char_p),
                                C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
lib.bvcf_run_buffer.argtypes = [C.POINTER(Config), C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p),
                                C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                C.POINTER(C.c_uint64)]
lib.bvcf_run_fd.argtypes = [C.POINTER(Config), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
lib.bvcf_decompress_fd.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_char_p]
lib.bvcf_free.argtypes = [C.c_void_p]
lib.bvcf_free.restype = None


class RangePlan(C.Structure):
    _fields_ = [("data_off", C.c_uint64), ("range_bytes", C.c_uint64), ("spare_bytes", C.c_uint64), ("n_ranges", C.c_uint64)]


class TextCut(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_uint32), ("start", C.c_uint64), ("end", C.c_uint64), ("long_start", C.c_uint64)]


class ThreadBudget(C.Structure):
    _fields_ = [("readers", C.c_uint32), ("copy_threads", C.c_uint32), ("format_threads", C.c_uint32), ("busy_total", C.c_uint32)]


class PlanBlock(C.Structure):
    _fields_ = [("worker", C.c_uint32), ("piece", C.c_uint32), ("range", C.c_uint64), ("file_off", C.c_uint64),
                ("nbytes", C.c_uint64), ("own", C.c_uint64), ("first_off", C.c_uint32), ("bgzf", C.c_uint8),
                ("bgzf_flags", C.c_uint8), ("last_piece", C.c_uint8), ("reserved", C.c_uint8)]


CUT_LINES, CUT_NONE, CUT_LONG = 0, 1, 2
MODE_STREAM, MODE_TEXT_RANGES, MODE_BGZF_RANGES = 0, 1, 2
BGZF_SKIP_FIRST_LINE, BGZF_END_OF_STREAM = 1, 2
lib.bvcf_plan_text_ranges.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(RangePlan)]
lib.bvcf_plan_bgzf_ranges.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.POINTER(RangePlan)]
lib.bvcf_cut_text_range.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_uint8, C.POINTER(TextCut)]
lib.bvcf_find_bgzf_chain.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t]
lib.bvcf_find_bgzf_chain.restype = C.c_long
lib.bvcf_plan_threads.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.POINTER(ThreadBudget)]
lib.bvcf_plan_fd.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint64, C.c_int, C.POINTER(PlanBlock), C.c_size_t,
                             C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(RangePlan)]


def plan_fd(fd_in, n_workers, max_batch_bytes=0, device_inflate=1, fd_err=2, cap=1 << 16):
    """the blocks bvcf_run_fd's readers would hand to n_workers device workers (no device involved)
    -> (rc, mode, RangePlan, [PlanBlock])"""
    out = (PlanBlock * cap)()
    n = C.c_size_t(0)
    mode = C.c_int(-1)
    plan = RangePlan()
    rc = lib.bvcf_plan_fd(fd_in, fd_err, n_workers, max_batch_bytes, device_inflate, out, cap, C.byref(n), C.byref(mode), C.byref(plan))
    assert n.value <= cap, "more blocks than room"
    return rc, mode.value, plan, [out[i] for i in range(n.value)]


class BvcfError(RuntimeError):
    def __init__(self, rc, msg):
        super().__init__("bvcf error %d: %s" % (rc, msg))
        self.rc = rc


def make_config(cfg=None, device=0, max_batch_bytes=0, n_format_threads=0):
    """cfg uses the key names of tests/golden/known_answers.json (emptyField, keepId, allow, ...).
    The returned object keeps the byte strings alive."""
    cfg = cfg or {}
    c = Config()
    lib.bvcf_config_defaults(C.byref(c))
    keep = [cfg.get("emptyField", "!").encode(), cfg.get("fieldDelimiter", ";").encode(),
            cfg.get("allow", "PASS,.").encode(), cfg.get("exclude", "").encode()]
    c.empty_field, c.field_delimiter, c.allow_filter, c.exclude_filter = keep
    c.keep_id = int(cfg.get("keepId", False))
    c.keep_info = int(cfg.get("keepInfo", False))
    c.keep_pos = int(cfg.get("keepPos", False))
    c.normalize_header = int(cfg.get("normalizeHeader", True))
    c.device = device
    c.max_batch_bytes = max_batch_bytes
    c.n_format_threads = n_format_threads
    if cfg.get("sample"):
        keep.append(cfg["sample"].encode())
        c.sample_list_path = keep[-1]
    if cfg.get("dosageOutput"):
        keep.append(str(cfg["dosageOutput"]).encode())
        c.dosage_path = keep[-1]
    c.no_out = int(cfg.get("noOut", False))
    if cfg.get("devices"):  # bvcf_run_fd only: the device list the blocks are dealt to
        arr = (C.c_int32 * len(cfg["devices"]))(*cfg["devices"])
        keep.append(arr)
        c.devices = arr
        c.n_devices = len(cfg["devices"])
    c._keep = keep
    return c


def string_header(cfg=None):
    c = make_config(cfg)
    buf = C.create_string_buffer(512)
    n = lib.bvcf_string_header(C.byref(c), buf, len(buf))
    return buf.raw[:n].decode()


def run_buffer(vcf_bytes, cfg=None, device=0, max_batch_bytes=0, n_format_threads=0):
    """readVcf on an in-memory VCF through the HIP path.
    -> (rc, TSV body bytes (no header line), log text, n data lines)"""
    c = make_config(cfg, device, max_batch_bytes, n_format_threads)
    out, log = C.c_void_p(), C.c_void_p()
    n_out, n_log, n_lines = C.c_size_t(), C.c_size_t(), C.c_uint64()
    rc = lib.bvcf_run_buffer(C.byref(c), vcf_bytes, len(vcf_bytes), C.byref(out), C.byref(n_out), C.byref(log),
                             C.byref(n_log), C.byref(n_lines))
    o = C.string_at(out, n_out.value) if out.value else b""
    e = C.string_at(log, n_log.value).decode(errors="replace") if log.value else ""
    lib.bvcf_free(out)
    lib.bvcf_free(log)
    return rc, o, e, n_lines.value


def run_fd(fd_in, fd_out, fd_err, cfg=None, device=0, max_batch_bytes=0):
    """bvcf_run_fd over open file descriptors -> (rc, n data lines)"""
    c = make_config(cfg, device, max_batch_bytes)
    n_lines = C.c_uint64()
    rc = lib.bvcf_run_fd(C.byref(c), fd_in, fd_out, fd_err, C.byref(n_lines))
    return rc, n_lines.value


def allreduce_counters(ctxs):
    """the final count gather over a list of Ctx -> (totals[8], used_rccl)"""
    arr = (C.c_void_p * len(ctxs))(*[x.h for x in ctxs])
    out = (C.c_uint64 * 8)()
    used = C.c_int(0)
    rc = lib.bvcf_allreduce_counters(arr, len(ctxs), out, C.byref(used))
    if rc:
        raise BvcfError(rc, lib.bvcf_last_error(ctxs[0].h).decode())
    return list(out), bool(used.value)


lib.bvcf_bgzf_inflate_device.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]


def bgzf_inflate_device(comp, cap=None, device=0):
    """whole BGZF blocks -> (rc, text) through k_inflate / k_crc32 on the device"""
    cap = cap if cap is not None else max(64, 66000 * (comp.count(b"\x1f\x8b\x08\x04") + 1))
    buf = C.create_string_buffer(cap)
    n = C.c_size_t()
    rc = lib.bvcf_bgzf_inflate_device(device, comp, len(comp), buf, cap, C.byref(n))
    return rc, (buf.raw[:n.value] if rc == 0 else b""), n.value


def decompress(data, n_threads=0):
    """gzip / BGZF / plain bytes -> (rc, bytes, kind) through the driver's byte source (host only)"""
    import tempfile
    with tempfile.TemporaryFile() as fi, tempfile.TemporaryFile() as fo:
        fi.write(data)
        fi.flush()
        fi.seek(0)
        kind = C.create_string_buffer(8)
        rc = lib.bvcf_decompress_fd(fi.fileno(), fo.fileno(), n_threads, kind)
        fo.seek(0)
        return rc, fo.read(), kind.value.decode()


def decompress_pipe(data, n_threads=16, piece=0, seed=0):
    """the same through a PIPE (stdin's shape): a writer thread feeds `data` in pieces of `piece` bytes (0: random sizes up
    to 3 MiB) -- text through a pipe is read by bvcf_input's splice fan-out when n_threads allows -> (rc, bytes, kind)"""
    import random
    import tempfile
    import threading
    r, w = os.pipe()
    rng = random.Random(seed)

    def feed():
        try:
            mv, off = memoryview(data), 0
            while off < len(mv):
                n = piece or rng.choice([1, 7, 4096, 65536, 100_000, 1 << 20, 3 << 20])
                off += os.write(w, mv[off:off + n])
        finally:
            os.close(w)

    t = threading.Thread(target=feed)
    t.start()
    try:
        with tempfile.TemporaryFile() as fo:
            kind = C.create_string_buffer(8)
            rc = lib.bvcf_decompress_fd(r, fo.fileno(), n_threads, kind)
            fo.seek(0)
            return rc, fo.read(), kind.value.decode()
    finally:
        os.close(r)
        t.join()


class Batch:
    """numpy views (copied) of one collected bvcf_result"""

    def __init__(self, r):
        self.batch_seq = r.batch_seq
        self.n_samples = r.n_samples
        self.cmap_stride = r.cmap_stride
        self.kernel_ms = r.kernel_ms
        self.n_lines_seen = r.n_lines_seen

        def arr(ptr, n, dt):
            if not n:
                return np.zeros(0, dtype=dt)
            return np.frombuffer(C.string_at(ptr, n * dt.itemsize), dtype=dt).copy()

        self.sites = None
        # bvcf_params.render_sites: the rows of the lines the packed form settles, and where the other lines' rows belong
        self.rows = C.string_at(r.rows, r.n_row_bytes) if r.rows and r.n_row_bytes else b""
        self.row_cuts = arr(r.row_cuts, r.n_row_cuts, ROW_CUT_DTYPE) if r.row_cuts else np.zeros(0, dtype=ROW_CUT_DTYPE)
        self.n_ok_sites = r.n_ok_sites
        self.rendered = bool(r.rows) or (not r.sites and r.n_row_cuts > 0) or bool(r.n_ok_sites)
        if r.sites:
            # the packed form of a file without samples (bvcf_params.packed_sites): expanded here into the arrays of the full
            # form, so that line i is lines[i] and its first record alleles[i] whichever way the batch came back
            self.sites = arr(r.sites, r.n_lines, SITE_DTYPE)
            self.full_lines = arr(r.lines, r.n_full_lines, LINE_DTYPE)
            raw = arr(r.alleles, r.n_alleles, ALLELE_DTYPE)
            self.lines, self.alleles = self._expand(self.sites, self.full_lines, raw, r.n_lines)
        elif r.row_cuts:
            # rendered rows (bvcf_params.render_sites): only the lines left to the host have records -- lines[cut.slot]
            self.full_lines = arr(r.lines, r.n_full_lines, LINE_DTYPE)
            self.lines = self.full_lines
            self.alleles = arr(r.alleles, r.n_alleles, ALLELE_DTYPE)
        else:
            self.lines = arr(r.lines, r.n_lines, LINE_DTYPE)
            self.alleles = arr(r.alleles, r.n_alleles, ALLELE_DTYPE)
        self.n_lines = r.n_lines
        self.errs = arr(r.errs, r.n_errs, ERR_DTYPE)
        self.cmap = arr(r.cmap, r.n_cmap_bytes, np.dtype("u1"))
        # want_dosage: one int8 row per alleles[] slot (rows of slots without a record hold garbage)
        # bvcf_submit_bgzf: the batch's inflated text (lines[].off point into it)
        self.text = C.string_at(r.text, r.n_text_bytes) if r.text and r.n_text_bytes else b""
        # ... with samples only the line heads (CHROM..INFO), packed: line i's bytes start at text[head_off[i]]
        self.head_off = arr(r.head_off, r.n_lines, np.dtype("<u4")) if r.head_off else None
        # want_name_lists: (off[3], len[3]) per alleles[] slot into the text arena `names`
        self.name_lists = arr(r.name_lists, r.n_alleles, NAMES_DTYPE) if r.name_lists else None
        self.names = C.string_at(r.names, r.n_name_bytes) if r.names and r.n_name_bytes else b""
        self.dosage = None
        if r.dosage and r.n_alleles:
            self.dosage = arr(r.dosage, r.n_alleles * r.dosage_stride, np.dtype("i1")).reshape(r.n_alleles, r.dosage_stride)

    @staticmethod
    def _expand(sites, full_lines, raw_alleles, n):
        lines = np.zeros(n, dtype=LINE_DTYPE)
        # the batch's alleles[]: the n_full first records, the further alleles right behind them (rec_first points there);
        # expanded: slot i for line i, the further alleles behind the n lines
        n_full = len(full_lines)
        extras = raw_alleles[n_full:]
        alleles = np.concatenate([np.zeros(n, dtype=ALLELE_DTYPE), extras])
        full = (sites["status"] & SITE_FULL) != 0
        fi = sites["full_idx"][full]
        assert len(set(fi.tolist())) == len(fi) and (fi < n_full).all(), "full_idx: distinct slots below n_full_lines"
        moved = full_lines.copy()
        multi = moved["n_rec"] > 1
        assert (moved["rec_first"][multi] >= n_full).all() and (moved["rec_first"][multi] + moved["n_rec"][multi] - 1 <= len(raw_alleles)).all()
        moved["rec_first"][multi] = moved["rec_first"][multi] - n_full + n
        lines[full] = moved[fi]
        assert (full_lines[fi]["gt_task"] == np.nonzero(full)[0]).all(), "a full record names its line"
        firsts = raw_alleles[fi].copy()
        # packed lines: a SNP with its position taken verbatim (or a line that failed the gate: no record)
        pk = ~full
        lines["off"][pk] = sites["off"][pk]
        lines["len"][pk] = sites["len"][pk]
        fe = sites["fend"].astype(np.uint32)
        fe9 = np.concatenate([fe, np.full((n, 1), 0xFF, dtype=np.uint32)], axis=1)
        fe9 = np.where(fe9 == 0xFF, sites["len"][:, None], fe9)
        lines["fend"][pk] = fe9[pk]
        lines["n_rec"][pk] = (sites["status"][pk] == LINE_OK).astype(np.uint32)
        lines["n_fields"][pk] = sites["n_fields"][pk]
        lines["gt_task"][pk] = np.nonzero(pk)[0]
        lines["status"][pk] = sites["status"][pk]
        idx = np.nonzero(pk)[0]
        alleles["line"][idx] = idx
        alleles["alt_len"][idx] = 1
        alleles["cmap_off"][idx] = NO_CMAP
        alleles["ref"][idx] = sites["ref"][pk]
        alleles["alt_base"][idx] = sites["alt_base"][pk]
        alleles["trtv"][idx] = sites["trtv"][pk]
        alleles["flags"][idx] = 1
        alleles["gt_task"][idx] = 0xFFFFFFFF
        alleles[np.nonzero(full)[0]] = firsts
        return lines, alleles

    def records(self, i):
        """the output alleles of line i, in order"""
        L = self.lines[i]
        n = int(L["n_rec"])
        if n == 0:
            return self.alleles[:0]
        idx = [i] + [int(L["rec_first"]) + j - 1 for j in range(1, n)]
        return self.alleles[idx]

    def record_slots(self, i):
        """indices into alleles[] (and dosage[]) of line i's output alleles, in order"""
        L = self.lines[i]
        n = int(L["n_rec"])
        return [] if n == 0 else [i] + [int(L["rec_first"]) + j - 1 for j in range(1, n)]

    def line_head(self, i):
        """the bytes of line i that came back with a bvcf_submit_bgzf batch: the whole line (no samples) or its head up
        to the end of the INFO column"""
        L = self.lines[i]
        if self.head_off is None:
            return self.text[int(L["off"]):int(L["off"]) + int(L["len"])]
        n = min(int(L["fend"][7]), int(L["len"]))
        return self.text[int(self.head_off[i]):int(self.head_off[i]) + n]

    def name_list(self, slot, q):
        """list q (0 het, 1 hom, 2 missing) of alleles[slot] as the device rendered it"""
        nl = self.name_lists[slot]
        return self.names[int(nl["off"][q]):int(nl["off"][q]) + int(nl["len"][q])]

    def classes(self, allele_row):
        """per-sample class codes (0 none, 1 het, 2 hom, 3 missing) of one allele record"""
        off = int(allele_row["cmap_off"])
        ns = self.n_samples
        if int(allele_row["flags"]) & 2:  # BVCF_ALLELE_CMAP_SPARSE: count, then (byte index << 8 | byte) entries
            words = self.cmap[off:off + 64].view("<u4")
            m = np.zeros((ns + 3) // 4, dtype=np.uint8)
            for e in words[1:1 + int(words[0])]:
                m[int(e) >> 8] = int(e) & 0xFF
        else:
            m = self.cmap[off:off + (ns + 3) // 4]
        return ((m[:, None] >> np.array([0, 2, 4, 6], dtype=np.uint8)) & 3).reshape(-1)[:ns]


class Ctx:
    """one bvcf_ctx (one GPU)"""

    def __init__(self, n_header_fields, allow="PASS,.", exclude="", device=0, eol_chars=1, eol_byte=b"\n",
                 max_batch_bytes=0, max_lines=0, max_alleles=0, cmap_bytes=0, n_slots=0, want_class_maps=True,
                 path=0, want_dosage=False, sample_names=None, delimiter=";", packed_sites=False, render_sites=False,
                 empty_field="!", keep_pos=False, keep_id=False, keep_info=False):
        p = Params()
        p.abi_version = ABI_VERSION
        p.device = device
        p.n_header_fields = n_header_fields
        p.eol_chars = eol_chars
        p.eol_byte = eol_byte[0]
        p.want_class_maps = int(want_class_maps)
        p.want_dosage = int(want_dosage)
        self._keep = [allow.encode(), exclude.encode()]
        p.allow_filter, p.exclude_filter = self._keep
        p.max_batch_bytes = max_batch_bytes
        p.max_lines = max_lines
        p.max_alleles = max_alleles
        p.cmap_bytes = cmap_bytes
        p.n_slots = n_slots
        p.path = path
        p.packed_sites = int(packed_sites or render_sites)
        p.render_sites = int(render_sites)
        p.want_name_lists = int(sample_names is not None)
        self.h = C.c_void_p()
        rc = lib.bvcf_create(C.byref(self.h), C.byref(p))
        if rc:
            raise BvcfError(rc, lib.bvcf_last_error(None).decode())
        if render_sites:
            lib.bvcf_set_row_format.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int]
            self._check(lib.bvcf_set_row_format(self.h, empty_field.encode(), int(keep_pos), int(keep_id), int(keep_info)))
        if sample_names is not None:  # device-side rendering of the het / hom / missing name lists
            enc = [x.encode() if isinstance(x, str) else x for x in sample_names]
            ptrs = (C.c_char_p * max(len(enc), 1))(*enc)
            lens = (C.c_uint32 * max(len(enc), 1))(*[len(x) for x in enc])
            self._check(lib.bvcf_set_sample_names(self.h, ptrs, lens, len(enc), delimiter.encode()))

    def close(self):
        if self.h:
            lib.bvcf_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise BvcfError(rc, lib.bvcf_last_error(self.h).decode())

    def submit(self, block, seq=0):
        self._blk = block  # keep alive until collect
        self._check(lib.bvcf_submit(self.h, block, len(block), seq))

    def submit_bgzf(self, comp, n_own, skip_first_line, first_off=0, seq=0):
        """whole BGZF blocks: n_own bytes of own blocks, then look-ahead blocks (see include/bvcf.h)"""
        self._blk = comp
        self._check(lib.bvcf_submit_bgzf(self.h, comp, len(comp), n_own, int(skip_first_line), first_off, seq))

    def submit_device(self, dptr, nbytes, seq=0):
        self._check(lib.bvcf_submit_device(self.h, dptr, nbytes, seq))

    def reserve(self, lines, alleles, cmap_bytes):
        self._check(lib.bvcf_reserve(self.h, lines, alleles, cmap_bytes))

    def collect(self):
        r = Result()
        rc = lib.bvcf_collect(self.h, C.byref(r))
        if rc == E_CAPACITY:
            raise BvcfError(rc, "capacity: need lines=%d alleles=%d cmap=%d" % (r.need_lines, r.need_alleles, r.need_cmap_bytes))
        self._check(rc)
        return Batch(r)

    def process(self, block):
        self.submit(block)
        return self.collect()

    def bench_device(self, dptrs, nbytes, iters, slots=0):
        """kernel chain `iters` times over resident blocks (rotating); results stay on the device.  Batch i runs on
        slot i % slots (0 = all of the ctx's slots; 1 = strictly one batch after the other).
        -> (chain ms per batch, genotype-scan ms per batch, [lines, alleles, errs, cmap bytes, tasks])"""
        n = len(dptrs)
        ptrs = (C.c_void_p * n)(*dptrs)
        sizes = (C.c_size_t * n)(*nbytes)
        chain = (C.c_float * iters)()
        scan = (C.c_float * iters)()
        counts = (C.c_uint64 * 5)()
        self._check(lib.bvcf_bench_device_slots(self.h, ptrs, sizes, n, iters, slots, chain, scan, counts))
        return list(chain), list(scan), list(counts)

    def path(self):
        """1 = census path, 2 = streaming path"""
        return lib.bvcf_path(self.h)

    def stream_kernel(self):
        """the streaming kernel the next batch goes through: "k_stream", "k_stream_gen", or None off the streaming path"""
        lib.bvcf_bench_stream_kernel.argtypes = [C.c_void_p]
        lib.bvcf_bench_stream_kernel.restype = C.c_int
        k = lib.bvcf_bench_stream_kernel(self.h)
        return None if k < 0 else ("k_stream_gen" if k else "k_stream")

    def counters(self):
        out = (C.c_uint64 * 8)()
        self._check(lib.bvcf_counters(self.h, out))
        return list(out)
