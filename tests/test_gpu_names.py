"""Device-side rendering of the het / hom / missing sample-name lists (main.go:612-656): the text the
k_name_* kernels write must be strings.Join(names of the class, fieldDelimiter) in header order, for dense maps and
sparse lists, any name width and delimiter, and through arena growth; the TSV is the same bytes with the host join
(BVCF_DEVICE_NAMES=0)."""
import random

import numpy as np
import pytest

import oracle_lib as orc
import vcfgen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bv():
    import bystro_vcf_amd as b
    return b


def _vcf(seed, ns, n_lines, names, dense=0.0):
    rng = random.Random(seed)
    rows, pos = [], 50
    for li in range(n_lines):
        pos += rng.randint(1, 30)
        p_alt = dense if rng.random() < 0.5 else rng.choice([0.0005, 0.002, 0.01])
        alts = "T" if rng.random() < 0.8 else "T,G"
        gts = []
        for _ in range(ns):
            u = rng.random()
            if u < p_alt:
                gts.append(rng.choice(["0|1", "1|0", "1|1", "1|2" if "," in alts else "1|1"]))
            elif u < p_alt + 0.002:
                gts.append(rng.choice([".|.", "0|.", ".|1"]))
            else:
                gts.append("0|0")
        rows.append("\t".join(["3", str(pos), ".", "C", alts, ".", "PASS", ".", "GT"] + gts))
    return (vcfgen.header(ns, names=names) + "\n".join(rows) + "\n").encode()


@pytest.mark.parametrize("ns,name_len,delim,dense,path", [
    (2504, 7, ";", 0.3, 2), (2504, 7, ";", 0.3, 1), (300, 3, "|", 0.2, 0), (1030, 24, ";;", 0.4, 2), (70, 40, ",", 0.5, 1),
    (513, 1, "<-16 bytes long->", 0.3, 2),
])
def test_device_lists_equal_join_of_names(bv, ns, name_len, delim, dense, path):
    rng = random.Random(ns)
    names = ["".join(rng.choice("ABCxyz019_") for _ in range(rng.randint(1, name_len))) + "%d" % i for i in range(ns)]
    vcf = _vcf(ns, ns, 120, names, dense)
    body = vcf[vcf.index(b"\n", vcf.index(b"#CHROM")) + 1:]
    if len(delim) > 16:
        with pytest.raises(bv.BvcfError):
            bv.Ctx(9 + ns, allow="", path=path, sample_names=names, delimiter=delim)
        return
    ctx = bv.Ctx(9 + ns, allow="", path=path, sample_names=names, delimiter=delim)
    b = ctx.process(body)
    ctx.close()
    assert b.name_lists is not None
    n_checked = 0
    for i in range(len(b.lines)):
        for slot in b.record_slots(i):
            r = b.alleles[slot]
            if r["ac"] == 0:
                continue
            cls = b.classes(r)
            for q, code in enumerate((1, 2, 3)):
                want = delim.join(names[s] for s in np.flatnonzero(cls == code)).encode()
                assert b.name_list(slot, q) == want, (i, int(r["alt_idx"]), q)
                n_checked += 1
    assert n_checked > 200


@pytest.mark.parametrize("device_names", ["1", "0"])
@pytest.mark.parametrize("ns,dense,batch", [(2504, 0.3, 0), (2504, 0.9, 1 << 20), (260, 0.6, 1 << 20), (17, 0.5, 0)])
def test_tsv_identical_with_device_and_host_join(bv, monkeypatch, device_names, ns, dense, batch):
    """the same bytes as the oracle whether the lists come off the device or from the host's join; the 0.9-dense case
    in 1 MiB blocks needs more arena than the first reservation (names are twice the input text): it grows"""
    monkeypatch.setenv("BVCF_DEVICE_NAMES", device_names)
    rng = random.Random(ns)
    names = ["S%s" % ("x" * rng.randint(0, 12)) + str(i) for i in range(ns)]
    vcf = _vcf(77 + ns, ns, 500 if ns > 1000 else 900, names, dense)
    for cfg in ({"allow": ""}, {"allow": "", "fieldDelimiter": "&&", "emptyField": "NA", "keepInfo": True}):
        rc_o, out_o, log_o, _ = orc.run(vcf, cfg)
        rc_g, out_g, log_g, _ = bv.run_buffer(vcf, cfg, max_batch_bytes=batch)
        assert rc_o == 0 and rc_g == 0
        assert out_g == out_o and log_g == log_o


def test_cli_dense_rows(bv):
    """the CLI over a dense file in small blocks, both ways"""
    import os
    import subprocess
    import benchgen as bg
    cfg = bg.make_cfg("c3d")
    vcf = bg.header(cfg) + bg.rows_host(cfg, 100, 1200)
    want = (bv.string_header() + "\n").encode() + orc.run(vcf, None, n_threads=8)[1]
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bystro-vcf_amd", "bystro-vcf")
    for dn in ("1", "0"):
        p = subprocess.run([exe, "--batchMB", "2", "--devices", "0,0"], input=vcf, capture_output=True, timeout=300,
                           env=dict(os.environ, BVCF_DEVICE_NAMES=dn))
        assert p.returncode == 0, p.stderr[-300:]
        assert p.stdout == want, dn
