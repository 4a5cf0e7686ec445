"""ctypes binding of the CPU oracle (oracle/liborc.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


class OrcConfig(C.Structure):
    _fields_ = [
        ("empty_field", C.c_char_p),
        ("field_delimiter", C.c_char_p),
        ("keep_id", C.c_int),
        ("keep_info", C.c_int),
        ("keep_pos", C.c_int),
        ("allow_filter", C.c_char_p),
        ("exclude_filter", C.c_char_p),
        ("n_threads", C.c_int),
        ("normalize_header", C.c_int),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "liborc.so")
        src = os.path.join(ORACLE_DIR, "bvcf_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liborc.so"])
        L = C.CDLL(so)
        L.orc_alt_is_valid.argtypes = [C.c_char_p, C.c_size_t]
        L.orc_alt_is_valid.restype = C.c_int
        L.orc_get_alleles_flat.argtypes = [C.c_char_p] * 4 + [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.orc_get_alleles_flat.restype = C.c_size_t
        L.orc_make_het_hom_flat.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_char_p, C.c_void_p, C.c_void_p,
                                            C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_make_het_hom_flat.restype = C.c_int
        L.orc_run.argtypes = [C.POINTER(OrcConfig), C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p),
                              C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                              C.POINTER(C.c_uint64)]
        L.orc_run.restype = C.c_int
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_get_trtv.argtypes = [C.c_char, C.c_char_p, C.c_size_t]
        L.orc_get_trtv.restype = C.c_char
        _lib = L
    return _lib


def make_config(cfg=None, n_threads=1):
    """cfg uses the key names of tests/golden/known_answers.json (emptyField, keepId, allow, ...)."""
    cfg = cfg or {}
    c = OrcConfig()
    c.empty_field = cfg.get("emptyField", "!").encode()
    c.field_delimiter = cfg.get("fieldDelimiter", ";").encode()
    c.keep_id = int(cfg.get("keepId", False))
    c.keep_info = int(cfg.get("keepInfo", False))
    c.keep_pos = int(cfg.get("keepPos", False))
    c.allow_filter = cfg.get("allow", "PASS,.").encode()
    c.exclude_filter = cfg.get("exclude", "").encode()
    c.n_threads = n_threads
    c.normalize_header = int(cfg.get("normalizeHeader", True))
    return c


def alt_is_valid(alt):
    b = alt.encode()
    return bool(lib().orc_alt_is_valid(b, len(b)))


def get_alleles(chrom, pos, ref, alt):
    """-> (type, [[pos, ref, alt, idx], ...], log text)"""
    out = C.create_string_buffer(1 << 16)
    log = C.create_string_buffer(1 << 14)
    n = lib().orc_get_alleles_flat(chrom.encode(), pos.encode(), ref.encode(), alt.encode(), out, len(out), log, len(log))
    rows = out.raw[:n].decode().split("\n")
    alleles = []
    for r in rows[1:]:
        if r:
            p, rf, a, i = r.split("\t")
            alleles.append([p, rf, a, int(i)])
    return rows[0], alleles, log.value.decode()


def make_het_hom(line, n_header, allele):
    """-> (classes list, dosages list, ac, an); classes: 0 none 1 het 2 hom 3 missing"""
    b = line.encode() if isinstance(line, str) else line
    ns = max(n_header - 9, 0)
    cls = (C.c_uint8 * max(ns, 1))()
    dos = (C.c_int8 * max(ns, 1))()
    ac, an = C.c_int(0), C.c_int(0)
    rv = lib().orc_make_het_hom_flat(b, len(b), n_header, allele.encode(), cls, dos, C.byref(ac), C.byref(an))
    assert rv == 0
    return list(cls)[:ns], list(dos)[:ns], ac.value, an.value


def run(vcf_bytes, cfg=None, n_threads=1):
    """readVcf on an in-memory file -> (rc, output bytes without header, stderr text, n data lines)"""
    c = make_config(cfg, n_threads)
    out, err = C.c_void_p(), C.c_void_p()
    n_out, n_err, n_rows = C.c_size_t(), C.c_size_t(), C.c_uint64()
    rc = lib().orc_run(C.byref(c), vcf_bytes, len(vcf_bytes), C.byref(out), C.byref(n_out), C.byref(err),
                       C.byref(n_err), C.byref(n_rows))
    o = C.string_at(out, n_out.value)
    e = C.string_at(err, n_err.value).decode(errors="replace")
    lib().orc_free(out)
    lib().orc_free(err)
    return rc, o, e, n_rows.value


def run_dosage(vcf_bytes, cfg=None):
    """the rows of --dosageOutput (main.go:576-584) -> list of (locus, [int8 per sample]) in input order"""
    L = lib()
    L.orc_run_dosage.argtypes = [C.POINTER(OrcConfig), C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p),
                                 C.POINTER(C.c_size_t)]
    L.orc_run_dosage.restype = C.c_int
    c = make_config(cfg, 1)
    out, n_out = C.c_void_p(), C.c_size_t()
    rc = L.orc_run_dosage(C.byref(c), vcf_bytes, len(vcf_bytes), C.byref(out), C.byref(n_out))
    text = C.string_at(out, n_out.value).decode()
    L.orc_free(out)
    assert rc == 0
    rows = []
    for ln in text.splitlines():
        locus, _, d = ln.partition("\t")
        rows.append((locus, [int(x) for x in d.split(",")] if d else []))
    return rows
