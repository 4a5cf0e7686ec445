"""the dosage-matrix file writer (bvcf_arrow_*, host-only) read back with pyarrow"""
import ctypes as C
import os
import random

import numpy as np
import pytest

pa = pytest.importorskip("pyarrow")
import pyarrow.ipc as ipc  # noqa: E402

import bystro_vcf_amd as bv  # noqa: E402

lib = bv.lib
lib.bvcf_arrow_open.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32),
                                C.c_uint32, C.c_uint32, C.c_int]
lib.bvcf_arrow_append.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_void_p]
lib.bvcf_arrow_close.argtypes = [C.c_void_p]


def write(path, names, rows, rows_per_batch=0, level=0):
    w = C.c_void_p()
    arr = (C.c_char_p * max(len(names), 1))(*[n.encode() for n in names])
    lens = (C.c_uint32 * max(len(names), 1))(*[len(n.encode()) for n in names])
    assert lib.bvcf_arrow_open(C.byref(w), str(path).encode(), arr, lens, len(names), rows_per_batch, level) == 0
    for locus, d in rows:
        a = np.asarray(d, dtype=np.int8)
        assert lib.bvcf_arrow_append(w, locus.encode(), len(locus.encode()), a.ctypes.data) == 0
    assert lib.bvcf_arrow_close(w) == 0


def check(path, names, rows, n_batches):
    r = ipc.open_file(str(path))
    assert r.schema.names == ["locus"] + names
    assert r.schema.field(0).type == pa.string()
    assert all(r.schema.field(i + 1).type == pa.int8() for i in range(len(names)))
    assert r.num_record_batches == n_batches
    t = r.read_all()
    t.validate(full=True)
    assert t.num_rows == len(rows)
    assert t.column(0).to_pylist() == [locus for locus, _ in rows]
    for s in range(len(names)):
        assert t.column(s + 1).to_pylist() == [int(d[s]) for _, d in rows]


def test_reference_table_shape(tmp_path):
    """the table of TestGenotypeMatrix (main_test.go:2911-2977)"""
    rows = [("chr1:1000:A:T", [2, 1, 0]), ("chr2:200:C:G", [1, 0, 2]), ("chr22:300:G:T", [-1, -1, 2])]
    write(tmp_path / "m.feather", ["S1", "S2", "S3"], rows)
    check(tmp_path / "m.feather", ["S1", "S2", "S3"], rows, 1)


@pytest.mark.parametrize("n_samples,n_rows,per_batch,level", [
    (1, 1, 0, 0), (3, 0, 0, 0), (7, 23, 5, 0), (2504, 37, 16, 0), (300, 5000, 0, 1), (64, 5001, 0, 0),
    (5, 12, 4, -1), (1000, 3, 2, -1),
])
def test_round_trip(tmp_path, n_samples, n_rows, per_batch, level):
    rng = random.Random(n_samples * 7919 + n_rows)
    names = ["HG%05d" % i if i % 3 else "sé%d.x" % i for i in range(n_samples)]
    rows = []
    for r in range(n_rows):
        kind = rng.random()
        d = [0] * n_samples if kind < 0.5 else [rng.choice([-1, 0, 0, 0, 1, 2, 3, 127]) for _ in range(n_samples)]
        rows.append(("chr%d:%d:%s:%s" % (rng.randint(1, 22), rng.randint(1, 10**9), "ACGT"[r % 4],
                                         rng.choice(["A", "+ACGT", "-12"])), d))
    p = tmp_path / "m.arrow"
    write(p, names, rows, per_batch, level)
    per = per_batch or 5000
    check(p, names, rows, (n_rows + per - 1) // per)
    if level >= 0 and n_rows * n_samples > 100000:
        assert os.path.getsize(p) < n_rows * n_samples  # zstd did something
