"""ctypes binding of benchgen/libbvcf_synth.so: deterministic synthetic VCF rows on host and device."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbvcf_synth.so")

PROFILES = {
    # BASELINE.json configs[1..3]
    "c2": dict(n_samples=0, p_multi=0, p_indel=0, p_bad=0),
    "c3": dict(n_samples=2504, p_multi=0, p_indel=0, p_bad=0),
    "c4": dict(n_samples=2504, p_multi=2000, p_indel=1500, p_bad=100),
    # not a BASELINE config: GATK-style sample fields "x|y:DP:GQ" (9-10 B, no fixed stride) — the
    # general (ballot / prefix-sum) scan path
    "c5": dict(n_samples=2504, p_multi=0, p_indel=0, p_bad=0, fmt_extra=1),
    # ... with one sample in twenty haploid ("x:DP:GQ": the males of a chrX file)
    "c5h": dict(n_samples=2504, p_multi=0, p_indel=0, p_bad=0, fmt_extra=1, haploid=1),
    # not a BASELINE config: every row a common variant (AF ~ 0.3, ~1 300 sample names per output row, ~11 KB of TSV
    # per row) -- the worst case for the host formatter
    "c3d": dict(n_samples=2504, p_multi=0, p_indel=0, p_bad=0, dense=1),
    # not BASELINE configs: small cohorts (a trio-sized and a panel-sized GATK-style file; census path, k_gt's general scan)
    "g10": dict(n_samples=10, p_multi=500, p_indel=1500, p_bad=0, fmt_extra=1),
    "g100": dict(n_samples=100, p_multi=500, p_indel=1500, p_bad=0, fmt_extra=1),
}
SEED = 20130502


class SynthCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_samples", C.c_uint32), ("p_multi", C.c_uint32), ("p_indel", C.c_uint32),
                ("p_bad", C.c_uint32), ("pos0", C.c_uint32), ("reserved", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        try:
            import torch  # noqa: F401  one HIP runtime per process (see bystro-vcf_amd/__init__.py)
        except ImportError:
            pass
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        L = C.CDLL(_SO)
        L.synth_header.argtypes = [C.POINTER(SynthCfg), C.c_char_p, C.c_size_t]
        L.synth_header.restype = C.c_size_t
        L.synth_rows_bytes_host.argtypes = [C.POINTER(SynthCfg), C.c_uint64, C.c_uint64]
        L.synth_rows_bytes_host.restype = C.c_uint64
        L.synth_fill_host.argtypes = [C.POINTER(SynthCfg), C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64]
        L.synth_fill_host.restype = C.c_uint64
        L.synth_lengths_dev.argtypes = [C.POINTER(SynthCfg), C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
        L.synth_fill_dev.argtypes = [C.POINTER(SynthCfg), C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def make_cfg(profile="c3", seed=SEED, **over):
    p = dict(PROFILES[profile])
    p.update(over)
    c = SynthCfg()
    c.seed = seed
    c.n_samples = p["n_samples"]
    c.p_multi, c.p_indel, c.p_bad = p["p_multi"], p["p_indel"], p["p_bad"]
    c.pos0 = 10177
    c.reserved = int(p.get("align16", 0)) | (2 if p.get("fmt_extra", 0) else 0) | (4 if p.get("dense", 0) else 0) | (8 if p.get("haploid", 0) else 0)
    return c


def header(cfg):
    n = lib().synth_header(C.byref(cfg), None, 0)
    buf = C.create_string_buffer(n + 1)
    lib().synth_header(C.byref(cfg), buf, n + 1)
    return buf.raw[:n]


def n_header_fields(cfg):
    return 8 + (1 + cfg.n_samples if cfg.n_samples else 0)


def rows_host(cfg, first, n):
    """bytes of rows [first, first+n)"""
    total = lib().synth_rows_bytes_host(C.byref(cfg), first, n)
    buf = (C.c_uint8 * total)()
    got = lib().synth_fill_host(C.byref(cfg), first, n, buf, total)
    assert got == total
    return bytes(buf)


def rows_device(cfg, first, n, pad=64):
    """torch.uint8 CUDA tensor holding rows [first, first+n) plus `pad` bytes; returns (tensor, nbytes)"""
    import torch
    lens = torch.empty(n, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    rc = lib().synth_lengths_dev(C.byref(cfg), first, n, lens.data_ptr(), st)
    assert rc == 0
    ends = torch.cumsum(lens, 0)
    nbytes = int(ends[-1].item())
    offs = ends - lens
    out = torch.full((nbytes + pad,), 10, dtype=torch.uint8, device="cuda")
    rc = lib().synth_fill_dev(C.byref(cfg), first, n, offs.data_ptr(), out.data_ptr(), st)
    assert rc == 0
    torch.cuda.synchronize()
    return out, nbytes
