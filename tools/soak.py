#!/usr/bin/env python3
"""Randomised differential soak (not part of the test suite): many random VCF shapes through the HIP path and the
oracle, both device paths (the streaming one with k_stream, with k_stream_gen, and choosing per batch) and the split
scans of very wide lines, TSV + log + dosage rows compared.  usage: python tools/soak.py [n_cases] [seed0]"""
import os
import random
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bystro_vcf_amd as bv  # noqa: E402
import oracle_lib as orc  # noqa: E402
import vcfgen  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
bad = 0
for i in range(n_cases):
    rng = random.Random(seed0 + i)
    ns = rng.choice([0, 1, 3, 17, 63, 64, 65, 255, 256, 257, 300, 511, 512, 700, 1023, 1024, 1500, 2047, 2048, 2504,
                     2559, 2560, 2561, 3000, rng.randint(1, 3500)])
    if os.environ.get("SOAK_NS"):  # e.g. SOAK_NS=0: sites-only files only (k_sites2 and its census), many more lines
        ns = int(os.environ["SOAK_NS"])
    n_lines = rng.randint(20, 160 if ns > 1000 else 400) if ns else rng.randint(20, 6000)
    weird = rng.choice([0.0, 0.0, 0.001, 0.01, 0.05, 0.3])
    fmt_extra = rng.random() < 0.4
    eol = "\r\n" if rng.random() < float(os.environ.get("SOAK_CRLF", "0.1")) else "\n"
    vcf = vcfgen.gen_vcf(seed0 + i, n_lines, ns, fmt_extra, weird, eol)
    cfg = rng.choice([{"allow": ""}, {}, {"keepId": True, "keepInfo": True, "keepPos": True, "exclude": "q10"}])
    if rng.random() < 0.3:  # (round 5: the tail of a rendered sites-only row is made of it)
        cfg = dict(cfg, emptyField=rng.choice(["NA", ".", "sixteen_bytes_xx", "seventeen_bytes_x", ""]))
    rc_o, out_o, log_o, n_o = orc.run(vcf, cfg)
    want_dos = orc.run_dosage(vcf, cfg) if ns else []
    for path in ("1", "2", "2g", "2a", "wide"):
        # "wide": the census path with the scans of one line split over waves, as for cohorts of >= 32 768 samples;
        # "2g": the streaming path with k_stream_gen pinned; "2a": the ctx picks k_stream / k_stream_gen per batch
        os.environ["BVCF_PATH"] = "1" if path == "wide" else path[0]
        os.environ.pop("BVCF_GEN_STREAM", None)
        if path in ("2", "2g"):
            os.environ["BVCF_GEN_STREAM"] = "1" if path == "2g" else "0"
        os.environ.pop("BVCF_WIDE", None)
        os.environ.pop("BVCF_WIDE_WIN", None)
        os.environ["BVCF_DEVICE_NAMES"] = rng.choice(["0", "1"])  # host join / device-rendered name lists
        os.environ["BVCF_RENDER_SITES"] = rng.choice(["0", "1", "1"])  # sites-only files: rows by the host / on the device
        os.environ["BVCF_PACKED_SITES"] = rng.choice(["0", "1", "1", "1"])
        if path == "wide":
            os.environ["BVCF_WIDE"] = "1"
            os.environ["BVCF_WIDE_WIN"] = str(rng.choice([64, 100, 777, 1024, 4096, 65536]))
        with tempfile.TemporaryDirectory() as td:
            c = dict(cfg)
            if ns:
                c["dosageOutput"] = os.path.join(td, "d.arrow")
            rc_g, out_g, log_g, n_g = bv.run_buffer(vcf, c, max_batch_bytes=rng.choice([0, 1 << 20, 1 << 22]))
            ok = (rc_g != 0) == (rc_o != 0) and out_g == out_o and log_g == log_o and n_g == n_o
            if ok and ns and rc_g == 0:
                import pyarrow.ipc as ipc
                t = ipc.open_file(c["dosageOutput"]).read_all()
                cols = [t.column(k).to_pylist() for k in range(1, t.num_columns)]
                got = [(loc, [col[r] for col in cols]) for r, loc in enumerate(t.column(0).to_pylist())]
                ok = got == want_dos
        if not ok:
            bad += 1
            print("MISMATCH case %d seed %d path %s ns %d lines %d weird %g fmt_extra %s eol %r cfg %s" % (
                i, seed0 + i, path, ns, n_lines, weird, fmt_extra, eol, cfg), flush=True)
    if i % 20 == 19:
        print("%d cases done, %d mismatches" % (i + 1, bad), flush=True)
print("soak: %d cases x 5 modes, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
