"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/bvcf.h declares, its
struct layouts match the header, and without a GPU it fails loudly instead of falling back."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bv():
    import bystro_vcf_amd as b
    return b


def test_exports_every_declared_symbol(bv):
    hdr = open(os.path.join(ROOT, "include", "bvcf.h")).read()
    declared = set(re.findall(r"\b(bvcf_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"bvcf_ctx"}
    assert declared == set(bv.EXPORTS), declared ^ set(bv.EXPORTS)
    for name in declared:
        assert hasattr(bv.lib, name), name
    # the measurement hooks live in their own header, outside the drop-in ABI
    bench = open(os.path.join(ROOT, "include", "bvcf_bench.h")).read()
    hooks = set(re.findall(r"\b(bvcf_[a-z_0-9]+)\s*\(", bench))
    assert hooks == set(bv.BENCH_EXPORTS) and not (hooks & declared)
    for name in hooks:
        assert hasattr(bv.lib, name), name
    # ... and so does the partition logic exported for host-only tests
    plan = open(os.path.join(ROOT, "include", "bvcf_plan.h")).read()
    entries = set(re.findall(r"\b(bvcf_[a-z_0-9]+)\s*\(", plan))
    assert entries == set(bv.PLAN_EXPORTS) and not (entries & declared)
    for name in entries:
        assert hasattr(bv.lib, name), name


def test_struct_layouts_match_header(bv, tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "bvcf.h"\n#include "bvcf_plan.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(bvcf_line),sizeof(bvcf_allele),sizeof(bvcf_err),sizeof(bvcf_params),"
                   "sizeof(bvcf_result),sizeof(bvcf_config),sizeof(bvcf_range_plan),sizeof(bvcf_text_cut),"
                   "sizeof(bvcf_thread_budget),sizeof(bvcf_plan_block),sizeof(bvcf_site),sizeof(bvcf_row_cut));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes == [bv.LINE_DTYPE.itemsize, bv.ALLELE_DTYPE.itemsize, bv.ERR_DTYPE.itemsize,
                     C.sizeof(bv.Params), C.sizeof(bv.Result), C.sizeof(bv.Config), C.sizeof(bv.RangePlan),
                     C.sizeof(bv.TextCut), C.sizeof(bv.ThreadBudget), C.sizeof(bv.PlanBlock), bv.SITE_DTYPE.itemsize,
                     bv.ROW_CUT_DTYPE.itemsize]


def test_header_known_answers(bv, known_answers):
    for case in known_answers["header"]:
        got = bv.string_header({"keepPos": case["keepPos"], "keepId": case["keepId"], "keepInfo": case["keepInfo"]})
        assert got.split("\t") == case["expected"], case["cite"]


def test_no_cpu_fallback(bv):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(bv.BvcfError) as ei:
        bv.Ctx(8)
    assert ei.value.rc == bv.E_NODEV
    rc, out, log, _ = bv.run_buffer(b"##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
                                    b"1\t5\t.\tA\tG\t.\tPASS\t.\n")
    assert rc == bv.E_NODEV and out == b"" and "no CPU fallback" in log


def test_fatal_paths_precede_device_use(bv):
    # main.go:262-264, 292-294: decided on the host before any ctx exists
    rc, _, log, _ = bv.run_buffer(b"#CHROM\tPOS\n1\t2\n")
    assert rc == bv.E_FATAL and "Not a VCF file" in log
    rc, _, log, _ = bv.run_buffer(b"##fileformat=VCFv4.2\n##x\n")
    assert rc == bv.E_FATAL and "No header found" in log


def test_cli_flag_surface():
    exe = os.path.join(ROOT, "bystro-vcf_amd", "bystro-vcf")
    if not os.path.exists(exe):
        pytest.skip("CLI not built")
    p = subprocess.run([exe, "--nope"], input=b"", capture_output=True)
    assert p.returncode == 2 and b"flag provided but not defined: -nope" in p.stderr
    p = subprocess.run([exe, "--noOut"], input=b"", capture_output=True)
    assert p.returncode == 1 and b"When specifying --noOut, must specify --dosageOutput" in p.stderr
    p = subprocess.run([exe, "--noOut", "--out", "x"], input=b"", capture_output=True)
    assert p.returncode == 1 and b"Cannot specify --noOut and --out" in p.stderr
