"""Pins the CPU oracle (oracle/bvcf_oracle.c) against the reference's own known answers
(main_test.go tables, transcribed in tests/golden/known_answers.json) and its 1000-Genomes
regression pair (previous_out_check/)."""
import oracle_lib as orc


def test_get_alleles_known_answers(known_answers):
    for case in known_answers["get_alleles"]:
        typ, alleles, _ = orc.get_alleles(case["chrom"], case["pos"], case["ref"], case["alt"])
        assert typ == case["type"], case["cite"]
        assert alleles == case["alleles"], case["cite"]


def test_alt_is_valid_known_answers(known_answers):
    for alt, expect in known_answers["alt_is_valid"]:
        assert orc.alt_is_valid(alt) == expect, alt


def test_make_het_hom_known_answers(known_answers):
    for case in known_answers["make_het_hom"]:
        cls, _, ac, an = orc.make_het_hom(case["line"], case["n_header"], case["allele"])
        got = (cls.count(2), cls.count(1), cls.count(3), ac, an)
        want = (case["n_hom"], case["n_het"], case["n_missing"], case["ac"], case["an"])
        assert got == want, case["cite"]


def test_dosage_known_answers(known_answers):
    for case in known_answers["dosage"]:
        _, dos, _, _ = orc.make_het_hom(case["line"], case["n_header"], case["allele"])
        assert dos == case["dosages"], case["cite"]


def test_line_passes_known_answers(known_answers):
    # linePasses is reached through readVcf: a passing record yields >= 1 row
    for case in known_answers["line_passes"]:
        vcf = "##fileformat=VCFv4.x\n" + "\t".join(case["header"]) + "\n" + "\t".join(case["record"]) + "\n"
        rc, out, _, _ = orc.run(vcf.encode(), {"allow": case["allow"], "exclude": case["exclude"]})
        assert rc == 0
        assert (len(out) > 0) == case["expect"], case["cite"]


def check_rows(case, out):
    rows = [r.split("\t") for r in out.decode().split("\n") if r]
    assert len(rows) == case["n_rows"], case["cite"]
    for r in rows:
        assert len(r) == case["n_cols"], case["cite"]
    for row, col, text in case["asserts"]:
        assert rows[row][col] == text, (case["cite"], row, col, rows[row])


def test_end_to_end_known_answers(known_answers):
    for case in known_answers["end_to_end"]:
        rc, out, _, _ = orc.run(case["vcf"].encode(), case["config"])
        assert rc == 0, case["cite"]
        check_rows(case, out)


def test_fatal_paths():
    rc, _, err, _ = orc.run(b"#CHROM\tPOS\n1\t2\n")
    assert rc == 1 and "Not a VCF file" in err  # main.go:262-264
    rc, _, err, _ = orc.run(b"##fileformat=VCFv4.2\n##x\n")
    assert rc == 1 and "No header found" in err  # main.go:292-294


def test_unterminated_last_line_is_dropped():
    # main.go:354-358: io.EOF before the terminator discards the partial line
    h = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    rc, out, _, n = orc.run((h + "1\t5\t.\tA\tG\t.\tPASS\t.\n1\t6\t.\tA\tC\t.\tPASS\t.").encode())
    assert rc == 0 and n == 1
    assert out.decode().split("\t")[:5] == ["chr1", "5", "SNP", "A", "G"]


def test_chrom_prefix_rule():
    # main.go:570-574
    h = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    body = "".join("%s\t5\t.\tA\tG\t.\tPASS\t.\n" % c for c in ["1", "chr1", "chr", "contig1", "cX", "chrom"])
    _, out, _, _ = orc.run((h + body).encode())
    got = [r.split("\t")[0] for r in out.decode().split("\n") if r]
    assert got == ["chr1", "chr1", "chrchr", "contig1", "chrcX", "chrom"]


def test_golden_1kg_sorted_identity(golden_1kg):
    """previous_out_check/README.md:5-9: sort both sides, diff.  19 821 rows."""
    vcf, want_sorted, want_header = golden_1kg
    for nt in (1, 4):
        rc, out, err, n = orc.run(vcf, n_threads=nt)
        assert rc == 0 and n == 19747
        rows = out.split(b"\n")
        assert rows[-1] == b""
        assert len(rows) - 1 == 19821
        assert sorted(rows[:-1]) == want_sorted
        assert err.count("ALT not ACTG") == 17
    assert want_header.decode().split("\t") == [
        "chrom", "pos", "type", "ref", "alt", "trTv", "heterozygotes", "heterozygosity", "homozygotes",
        "homozygosity", "missingGenos", "missingness", "ac", "an", "sampleMaf"]


def test_threads_preserve_input_order(golden_1kg):
    vcf = golden_1kg[0]
    a = orc.run(vcf, n_threads=1)[1]
    b = orc.run(vcf, n_threads=8)[1]
    assert a == b


def test_dosage_rows_reference_table():
    """TestGenotypeMatrix, main_test.go:2911-2977: the rows the reference hands to its Arrow writer"""
    hdr = "##fileformat=VCFv4.x\n" + "\t".join(["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT",
                                                "S1", "S2", "S3"]) + "\n"
    rows = [["1", "1000", "rs1", "A", "T", ".", "PASS", "DP=100", "GT", "1|1", "0|1", "0|0"],
            ["2", "200", "rs2", "C", "G", ".", "PASS", "DP=100", "GT", "0|1", "0|0", "1|1"],
            ["22", "300", "rs2", "G", "T", ".", "PASS", "DP=100", "GT", "0|.", "0|.", "1|1"]]
    vcf = (hdr + "".join("\t".join(r) + "\n" for r in rows)).encode()
    assert orc.run_dosage(vcf) == [("chr1:1000:A:T", [2, 1, 0]), ("chr2:200:C:G", [1, 0, 2]),
                                   ("chr22:300:G:T", [-1, -1, 2])]
