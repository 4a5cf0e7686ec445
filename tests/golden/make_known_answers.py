#!/usr/bin/env python3
"""Writes tests/golden/known_answers.json.

Every entry is a hand transcription of an input/expected pair asserted by the
reference's own unit tests (/root/reference/main_test.go @ 2024_10_08); `cite`
gives the lines.  Nothing here is computed by the oracle or by the HIP path:
these are the known answers both are checked against.

Float expectations are the literal text of strconv.FormatFloat(x,'G',3,64) for
the simple fractions the reference tests use (1/3 -> 0.333, 1/6 -> 0.167, ...).
"""
import json
import os

T = "\t".join
VERSION = "##fileformat=VCFv4.x"


def vcf(header, *records):
    return VERSION + "\n" + T(header) + "\n" + "".join(T(r) + "\n" for r in records)


H8 = ["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO"]
H4S = H8 + ["FORMAT", "Sample1", "Sample2", "Sample3", "Sample4"]
H5S = H4S + ["Sample5"]

# ---------------------------------------------------------------- getAlleles
# main_test.go:295-522 (TestUpdateFieldsWithAlt); alleles = [pos, ref, alt, altIdx]
get_alleles = [
    dict(cite="main_test.go:296-311", chrom="chr1", pos="100", ref="T", alt="C", type="SNP",
         alleles=[["100", "T", "C", 0]]),
    dict(cite="main_test.go:313-321", chrom="chr1", pos="100", ref="TCCT", alt="TCCA", type="SNP",
         alleles=[["103", "T", "A", 0]]),
    dict(cite="main_test.go:323-331", chrom="chr1", pos="100", ref="TGCT", alt="TGAT", type="SNP",
         alleles=[["102", "C", "A", 0]]),
    dict(cite="main_test.go:333-341", chrom="chr1", pos="100", ref="TGCT", alt="AGCT", type="SNP",
         alleles=[["100", "T", "A", 0]]),
    dict(cite="main_test.go:343-372", chrom="chr1", pos="100", ref="TCGT", alt="GTAA", type="MNP",
         alleles=[["100", "T", "G", 0], ["101", "C", "T", 0], ["102", "G", "A", 0], ["103", "T", "A", 0]]),
    dict(cite="main_test.go:374-405", chrom="chr1", pos="100", ref="TCGT", alt="TAGC", type="MNP",
         alleles=[["101", "C", "A", 0], ["103", "T", "C", 0]]),
    dict(cite="main_test.go:407-437", chrom="chr1", pos="100", ref="TCGT", alt="TCGC", type="SNP",
         alleles=[["103", "T", "C", 0]]),
    dict(cite="main_test.go:439-448", chrom="chr1", pos="100", ref="TC", alt="T", type="DEL",
         alleles=[["101", "C", "-1", 0]]),
    dict(cite="main_test.go:450-458", chrom="chr1", pos="100", ref="TAGCGT", alt="T", type="DEL",
         alleles=[["101", "A", "-5", 0]]),
    dict(cite="main_test.go:460-469", chrom="chr1", pos="100", ref="TAGCTT", alt="TA", type="DEL",
         alleles=[["102", "G", "-4", 0]]),
    dict(cite="main_test.go:471-480", chrom="chr1", pos="100", ref="TAGCTT", alt="TAC", type="",
         alleles=[]),
    dict(cite="main_test.go:482-497", chrom="chr1", pos="100", ref="TAGCTT", alt="TAT", type="DEL",
         alleles=[["102", "G", "-3", 0]]),
    dict(cite="main_test.go:499-509", chrom="chr1", pos="100", ref="T", alt="TAGCTT", type="INS",
         alleles=[["100", "T", "+AGCTT", 0]]),
    dict(cite="main_test.go:511-521", chrom="chr1", pos="100", ref="TT", alt="TAGCTT", type="INS",
         alleles=[["100", "T", "+AGCT", 0]]),
    # end-to-end tests that pin getAlleles rows (columns 1,3,4 of the output)
    dict(cite="main_test.go:2348,2391,2452", chrom="20", pos="4", ref="GCACG", alt="G,GTCACACG",
         type="MULTIALLELIC", alleles=[["5", "C", "-4", 0], ["4", "G", "+TCA", 1]]),
    dict(cite="main_test.go:2526-2527,2575,2583", chrom="16", pos="84034434", ref="GAGGGAGACAGAGGGAAGT",
         alt="G,GGGGAGACAGAGGGAAGT", type="MULTIALLELIC",
         alleles=[["84034435", "A", "-18", 0], ["84034435", "A", "-1", 1]]),
    dict(cite="main_test.go:2604-2606,2634-2658", chrom="1", pos="874816",
         ref="CCCCCTCATCACCTCCCCAGCCACGGTGAGGACCCACCCTGGCATGATCT",
         alt="CCCCCTCATCACCTCCCCAGCCACGGTGAGGACCCACCCTGGCATGATCTCCCCTCATCACCTCCCCAGCCACGGTGAGGACCCACCCTGGCATGATCT,"
             "GCCCCTCATCACCTCCCCAGCCACGGTGAGGACCCACCCTGGCATGATCT,C,"
             "CTCCCCTCATCACCTCCCCAGCCACGGTGAGGACCCACCCTGGCATGATCT",
         type="MULTIALLELIC",
         alleles=[["874816", "C", "+CCCCTCATCACCTCCCCAGCCACGGTGAGGACCCACCCTGGCATGATCT", 0],
                  ["874816", "C", "G", 1], ["874817", "C", "-49", 2], ["874816", "C", "+T", 3]]),
    dict(cite="main_test.go:2679-2680,2708,2716", chrom="1", pos="1265061", ref="CGT", alt="TGT,C",
         type="MULTIALLELIC", alleles=[["1265061", "C", "T", 0], ["1265062", "G", "-2", 1]]),
    dict(cite="main_test.go:2737-2738,2765", chrom="1", pos="1265062", ref="CGT", alt="CGA", type="SNP",
         alleles=[["1265064", "T", "A", 0]]),
    dict(cite="main_test.go:2785-2786,2815-2839", chrom="1", pos="1000", ref="ACGT", alt="GATC", type="MNP",
         alleles=[["1000", "A", "G", 0], ["1001", "C", "A", 0], ["1002", "G", "T", 0], ["1003", "T", "C", 0]]),
    dict(cite="main_test.go:2857-2859", chrom="1", pos="1000", ref="A", alt="AA,AC,AG,AT,C,G,T,ATA,ATC,ATG,ATT",
         type="MULTIALLELIC",
         alleles=[["1000", "A", "+A", 0], ["1000", "A", "+C", 1], ["1000", "A", "+G", 2], ["1000", "A", "+T", 3],
                  ["1000", "A", "C", 4], ["1000", "A", "G", 5], ["1000", "A", "T", 6], ["1000", "A", "+TA", 7],
                  ["1000", "A", "+TC", 8], ["1000", "A", "+TG", 9], ["1000", "A", "+TT", 10]]),
    dict(cite="main_test.go:1204 (CTT->CT, only vcfPos asserted; row derived from main.go:971-998)",
         chrom="10", pos="1000", ref="CTT", alt="CT", type="DEL", alleles=[["1001", "T", "-1", 0]]),
]

# ---------------------------------------------------------------- altIsValid, main_test.go:571-650
alt_is_valid = [
    ["ACTG", True], [".", False], ["]13 : 123456]T", False], ["C[2 : 321682[", False],
    [".A", False], ["G.", False], ["<DUP>", False], ["A,C", False],
]

# ---------------------------------------------------------------- linePasses, main_test.go:524-569
_R = ["20", "4", ".", "GCG", "G,GCGCG", ".", None, "DP=100"]
line_passes = [
    dict(cite="main_test.go:529-538", header=H8, record=[x or "PASS" for x in _R], allow="PASS,.", exclude="", expect=True),
    dict(cite="main_test.go:540-548", header=H8, record=[x or "." for x in _R], allow="PASS,.", exclude="", expect=True),
    dict(cite="main_test.go:550-558", header=H8, record=[x or "blah" for x in _R], allow="", exclude="", expect=True),
    dict(cite="main_test.go:560-568", header=H8, record=[x or "blah" for x in _R], allow="", exclude="blah", expect=False),
]

# ---------------------------------------------------------------- makeHetHomozygotes
# main_test.go:652-951; header is 9 fixed + S1..S4
_SH = ["10", "1000", "rs#", "C", "T", "100", "PASS", "AC=1"]
_GL = ":-0.03,-1.12,-5.00"


def mhh(cite, fmt, samples, allele, n_hom, n_het, n_missing, ac, an, extra=None):
    fields = _SH + [fmt] + samples + (extra or [])
    return dict(cite=cite, n_header=13, line=T(fields), allele=allele, n_hom=n_hom, n_het=n_het,
                n_missing=n_missing, ac=ac, an=an)


make_het_hom = [
    mhh("main_test.go:660-677", "GT", ["0|0", "0|0", "0|0", "0|0"], "1", 0, 0, 0, 0, 8),
    mhh("main_test.go:679-696", "GT", ["0|1", "0|1", "0|1", "0|1"], "1", 0, 4, 0, 4, 8),
    mhh("main_test.go:699-716", "GT", [".|.", ".|.", ".|1", "1|."], "1", 0, 0, 4, 0, 0),
    mhh("main_test.go:718-737", "GT", [".|1", "0|1", "0|1", "0|1"], "1", 0, 3, 1, 3, 6, extra=["0"]),
    mhh("main_test.go:739-756", "GT", ["1|.", "0|1", "0|1", "0|1"], "1", 0, 3, 1, 3, 6, extra=["0.5"]),
    mhh("main_test.go:758-775", "GT", ["1|1", "1|1", "0|1", "0|1"], "1", 2, 2, 0, 6, 8, extra=["0.5"]),
    mhh("main_test.go:777-797", "GT", ["1|2", "1|1", "0|1", "0|1"], "1", 1, 3, 0, 5, 8),
    mhh("main_test.go:799-815", "GT", ["1|2", "1|1", "0|1", "0|1"], "2", 0, 1, 0, 1, 8),
    mhh("main_test.go:817-833", "GT:DS:GL", ["1|2" + _GL, "1|1" + _GL, "0|1" + _GL, "0|1" + _GL], "2", 0, 1, 0, 1, 8),
    mhh("main_test.go:835-851", "GT:DS:GL", ["1|2|1" + _GL, "1|1" + _GL, "0|1" + _GL, "0|1" + _GL], "2", 0, 1, 0, 1, 9),
    mhh("main_test.go:853-869", "GT", ["1|2|1", "1|1", "0|1", "0|1"], "2", 0, 1, 0, 1, 9),
    mhh("main_test.go:871-887", "GT:DS:GL", ["2|2|2" + _GL, "1|1" + _GL, "0|1" + _GL, "0|1" + _GL], "2", 1, 0, 0, 3, 9),
    mhh("main_test.go:889-905", "GT", ["2|2|2", "1|1", "0|1", "0|1"], "2", 1, 0, 0, 3, 9),
    mhh("main_test.go:916-932", "GT", ["0", ".", "1", "0"], "1", 1, 0, 1, 1, 3),
    mhh("main_test.go:934-950", "GT:DS:GL", ["0:1", ".:1", "1:1", "0:1"], "1", 1, 0, 1, 1, 3),
]

# ---------------------------------------------------------------- header(), main_test.go:74-169
BASE = ["chrom", "pos", "type", "ref", "alt", "trTv", "heterozygotes", "heterozygosity", "homozygotes",
        "homozygosity", "missingGenos", "missingness", "ac", "an", "sampleMaf"]
header = [
    dict(cite="main_test.go:75-86", keepPos=False, keepId=False, keepInfo=False, expected=BASE),
    dict(cite="main_test.go:88-99", keepPos=True, keepId=False, keepInfo=False, expected=BASE + ["vcfPos"]),
    dict(cite="main_test.go:101-112", keepPos=True, keepId=True, keepInfo=False, expected=BASE + ["vcfPos", "id"]),
    dict(cite="main_test.go:114-126", keepPos=False, keepId=False, keepInfo=True, expected=BASE + ["alleleIdx", "info"]),
    dict(cite="main_test.go:128-140", keepPos=False, keepId=True, keepInfo=True, expected=BASE + ["id", "alleleIdx", "info"]),
    dict(cite="main_test.go:142-154", keepPos=True, keepId=True, keepInfo=True,
         expected=BASE + ["vcfPos", "id", "alleleIdx", "info"]),
    dict(cite="main_test.go:156-168", keepPos=True, keepId=False, keepInfo=True,
         expected=BASE + ["vcfPos", "alleleIdx", "info"]),
]

# ---------------------------------------------------------------- flag parsing, main_test.go:19-57
flags = dict(
    cite="main_test.go:19-57",
    args=["--keepInfo", "--keepId", "--keepPos", "--in", "/path/to/file", "--err", "/path/to/err",
          "--cpuProfile", "/path/to/profile", "--emptyField", ".", "--out", "/path/to/out",
          "--fieldDelimiter", "&", "--allowFilter", "PASS,., somethingElse ",
          "--excludeFilter", "unwanted_one, unwanted_two "],
    expect=dict(keepInfo=True, keepId=True, keepPos=True, inPath="/path/to/file", errPath="/path/to/err",
                cpuProfile="/path/to/profile", outPath="/path/to/out", emptyField=".", fieldDelimiter="&",
                allowedFilters=["PASS", ".", "somethingElse"], excludedFilters=["unwanted_one", "unwanted_two"]),
)

# ---------------------------------------------------------------- end to end (readVcf string -> TSV)
# config keys mirror main.go:63-80; allow "" == nil map (tests build Config{} directly).
# asserts: [row, col, text]; col < 0 counts from the end of the row.
CFG0 = dict(emptyField="!", fieldDelimiter=";", keepId=False, keepInfo=False, keepPos=False, allow="", exclude="")


def cfg(**kw):
    c = dict(CFG0)
    c.update(kw)
    return c


def multi_asserts(nrows, chrom, vcfpos, rsid, info, miss, per_row):
    """common assert block of TestOutputsSamplesVcfPosIdAndInfo's multiallelic sub-cases"""
    a = []
    for r in range(nrows):
        a += [[r, 0, chrom], [r, -4, vcfpos], [r, -3, rsid], [r, -2, str(r)], [r, -1, info], [r, 11, miss]]
        for col, val in per_row[r].items():
            a.append([r, int(col), val])
    return a


_PR_4S = [{"7": "0", "9": "0.333", "12": "2", "13": "6", "14": "0.333"},
          {"7": "0.333", "9": "0", "12": "1", "13": "6", "14": "0.167"}]
_PR_5S = [{"7": "0.4", "9": "0", "12": "2", "13": "10", "14": "0.2"},
          {"7": "0.2", "9": "0.2", "12": "3", "13": "10", "14": "0.3"}]
_ALL = cfg(keepPos=True, keepId=True, keepInfo=True)
_PASSDOT = cfg(allow="PASS,.")

end_to_end = [
    dict(cite="main_test.go:959-1001 TestHandlesAllMissing", config=CFG0,
         vcf=vcf(H4S, ["10", "1000", "rs123", "A", "T", "100", "PASS", "AC=1", "GT", "./.", "./1", "1/.", "./0"],
                 ["10", "1000", "rs124", "A", "C", "100", "PASS", "AC=1", "GT", ".|.", "1|.", "1|.", ".|0"]),
         n_rows=0, n_cols=15, asserts=[]),
    dict(cite="main_test.go:230-270 TestWriteSampleListWhenNoSamples (9-field records under an 8-field header are dropped)",
         config=CFG0,
         vcf=vcf(H8, ["10", "1000", "rs123", "A", "T", "100", "PASS", "AC=1", "GT"],
                 ["10", "1000", "rs124", "A", "C", "100", "PASS", "AC=1", "GT"]),
         n_rows=0, n_cols=15, asserts=[]),
    dict(cite="main_test.go:1003-1056 TestOutputsInfo #1", config=cfg(keepInfo=True),
         vcf=vcf(H8, ["10", "1000", "rs#", "C", "T", "100", "PASS", "AC=1"]),
         n_rows=1, n_cols=17, asserts=[[0, 0, "chr10"], [0, 4, "T"], [0, 5, "1"], [0, -2, "0"], [0, -1, "AC=1"]]),
    dict(cite="main_test.go:1058-1104 TestOutputsInfo #2", config=cfg(keepInfo=True),
         vcf=vcf(H8, ["10", "1000", "rs#", "C", "T,G", "100", "PASS", "AC=1"]),
         n_rows=2, n_cols=17,
         asserts=[[0, 0, "chr10"], [0, 5, "0"], [0, -2, "0"], [0, -1, "AC=1"],
                  [1, 0, "chr10"], [1, 5, "0"], [1, -2, "1"], [1, -1, "AC=1"]]),
    dict(cite="main_test.go:1107-1155 TestOutputsId #1", config=cfg(keepId=True),
         vcf=vcf(H8, ["10", "1000", "rs123", "C", "T", "100", "PASS", "AC=1"]),
         n_rows=1, n_cols=16, asserts=[[0, 0, "chr10"], [0, 5, "1"], [0, -1, "rs123"]]),
    dict(cite="main_test.go:1157-1196 TestOutputsId #2", config=cfg(keepId=True),
         vcf=vcf(H8, ["10", "1000", "rs456", "C", "T,G", "100", "PASS", "AC=1"]),
         n_rows=2, n_cols=16,
         asserts=[[0, 0, "chr10"], [0, 5, "0"], [0, -1, "rs456"], [1, 0, "chr10"], [1, 5, "0"], [1, -1, "rs456"]]),
    dict(cite="main_test.go:1199-1240 TestOutputsVcfPos #1", config=cfg(keepPos=True),
         vcf=vcf(H8, ["10", "1000", "rs#", "CTT", "CT", "100", "PASS", "AC=1"]),
         n_rows=1, n_cols=16, asserts=[[0, -1, "1000"]]),
    dict(cite="main_test.go:1242-1273 TestOutputsVcfPos #2", config=cfg(keepPos=True),
         vcf=vcf(H8, ["10", "1003", "rs#", "C", "T,G", "100", "PASS", "AC=1"]),
         n_rows=2, n_cols=16, asserts=[[0, -1, "1003"], [1, -1, "1003"]]),
    dict(cite="main_test.go:1276-1331 TestOutputsVcfPosIdAndInfo", config=_ALL,
         vcf=vcf(H8, ["10", "1000", "rs1", "CTT", "CT", "100", "PASS", "AC=1"]),
         n_rows=1, n_cols=19, asserts=[[0, -4, "1000"], [0, -3, "rs1"], [0, -2, "0"], [0, -1, "AC=1"]]),
    dict(cite="main_test.go:1333-1444 TestOutputsSamplesVcfPosIdAndInfo #1", config=_ALL,
         vcf=vcf(H4S, ["10", "1000", "rs123", "A", "T", "100", "PASS", "AC=1", "GT", "0/0", "0/1", "1/1", "./."]),
         n_rows=1, n_cols=19,
         asserts=[[0, 0, "chr10"], [0, 5, "2"], [0, -4, "1000"], [0, -3, "rs123"], [0, -2, "0"], [0, -1, "AC=1"],
                  [0, 6, "Sample2"], [0, 7, "0.333"], [0, 8, "Sample3"], [0, 9, "0.333"], [0, 10, "Sample4"],
                  [0, 11, "0.25"], [0, 12, "3"], [0, 13, "6"], [0, 14, "0.5"]]),
    dict(cite="main_test.go:1446-1568 #2 (T,G with |)", config=_ALL,
         vcf=vcf(H4S, ["10", "1000", "rs456", "C", "T,G", "100", "PASS", "AC=1", "GT", "1|1", "0|0", "0|2", ".|."]),
         n_rows=2, n_cols=19, asserts=multi_asserts(2, "chr10", "1000", "rs456", "AC=1", "0.25", _PR_4S)),
    dict(cite="main_test.go:1570-1693 #3 (T,G with /)", config=_ALL,
         vcf=vcf(H4S, ["10", "1000", "rs456", "C", "T,G", "100", "PASS", "AC=1", "GT", "1/1", "0/0", "0/2", "./."]),
         n_rows=2, n_cols=19, asserts=multi_asserts(2, "chr10", "1000", "rs456", "AC=1", "0.25", _PR_4S)),
    dict(cite="main_test.go:1697-1821 #4 (GT:GQ with |)", config=_ALL,
         vcf=vcf(H4S, ["10", "1000", "rs456", "C", "T,G", "100", "PASS", "AC=1", "GT:GQ",
                       "1|1:1,2,3", "0|0:4,5,6", "0|2:1,3,5", ".|.:0,0,0"]),
         n_rows=2, n_cols=19, asserts=multi_asserts(2, "chr10", "1000", "rs456", "AC=1", "0.25", _PR_4S)),
    dict(cite="main_test.go:1823-1955 #5 (GT:GQ with /, bare ./.)", config=_ALL,
         vcf=vcf(H4S, ["10", "1000", "rs456", "C", "T,G", "100", "PASS", "AC=1", "GT:GQ",
                       "1/1:1,2,3", "0/0:4,5,6", "0/2:1,3,5", "./."]),
         n_rows=2, n_cols=19, asserts=multi_asserts(2, "chr10", "1000", "rs456", "AC=1", "0.25", _PR_4S)),
    dict(cite="main_test.go:1957-2083 #6 (5 samples, no missing, |)", config=_ALL,
         vcf=vcf(H5S, ["15", "1001", "rs457", "C", "T,G", "100", "PASS", "AC=1", "GT", "0|1", "2|0", "2|2", "0|0", "1|0"]),
         n_rows=2, n_cols=19, asserts=multi_asserts(2, "chr15", "1001", "rs457", "AC=1", "0", _PR_5S)),
    dict(cite="main_test.go:2085-2211 #7 (5 samples, /)", config=_ALL,
         vcf=vcf(H5S, ["15", "1002", "rs457", "C", "T,G", "100", "PASS", "AC=2", "GT", "0/1", "2/0", "2/2", "0/0", "1/0"]),
         n_rows=2, n_cols=19, asserts=multi_asserts(2, "chr15", "1002", "rs457", "AC=2", "0", _PR_5S)),
    dict(cite="main_test.go:2213-2339 #8 (5 samples, / and GT:GL)", config=_ALL,
         vcf=vcf(H5S, ["15", "1001", "rs457", "C", "T,G", "100", "PASS", "AC=2", "GT:GL",
                       "0/1:4,5,6", "2/0:7,8,9", "2/2:1,2,3", "0/0:.,.,.", "1/0:1,2,5"]),
         n_rows=2, n_cols=19, asserts=multi_asserts(2, "chr15", "1001", "rs457", "AC=2", "0", _PR_5S)),
    dict(cite="main_test.go:2342-2518 TestOutputMultiallelic", config=_PASSDOT,
         vcf=vcf(H8 + ["Format", "Sample1", "Sample2", "Sample3", "Sample4"],
                 ["20", "4", ".", "GCACG", "G,GTCACACG", ".", "PASS", "DP=100", "GT", "0|0", "0|1", "2|2", ".|."]),
         n_rows=2, n_cols=15,
         asserts=[[0, 0, "chr20"], [0, 5, "0"], [0, 4, "-4"], [0, 1, "5"], [0, 3, "C"], [0, 6, "Sample2"], [0, 8, "!"],
                  [0, 10, "Sample4"], [0, 7, "0.333"], [0, 9, "0"], [0, 11, "0.25"], [0, 12, "1"], [0, 13, "6"],
                  [0, 14, "0.167"],
                  [1, 0, "chr20"], [1, 5, "0"], [1, 4, "+TCA"], [1, 1, "4"], [1, 3, "G"], [1, 6, "!"], [1, 8, "Sample3"],
                  [1, 10, "Sample4"], [1, 7, "0"], [1, 9, "0.333"], [1, 11, "0.25"], [1, 12, "2"], [1, 13, "6"],
                  [1, 14, "0.333"]]),
    dict(cite="main_test.go:2520-2596 TestOutputComplexMultiDel", config=_PASSDOT,
         vcf=vcf(H8, ["16", "84034434", "rs141446650", "GAGGGAGACAGAGGGAAGT", "G,GGGGAGACAGAGGGAAGT", ".", "PASS", "DP=100"]),
         n_rows=2, n_cols=15,
         asserts=[[0, 0, "chr16"], [0, 9, "0"], [0, 11, "0"], [0, 3, "A"], [0, 1, "84034435"], [0, 4, "-18"],
                  [1, 0, "chr16"], [1, 9, "0"], [1, 11, "0"], [1, 3, "A"], [1, 1, "84034435"], [1, 4, "-1"]]),
    dict(cite="main_test.go:2598-2671 TestOutputComplexDel", config=_PASSDOT,
         vcf=vcf(H8, ["1", "874816", "rs200996316", "CCCCCTCATCACCTCCCCAGCCACGGTGAGGACCCACCCTGGCATGATCT",
                      "CCCCCTCATCACCTCCCCAGCCACGGTGAGGACCCACCCTGGCATGATCTCCCCTCATCACCTCCCCAGCCACGGTGAGGACCCACCCTGGCATGATCT,"
                      "GCCCCTCATCACCTCCCCAGCCACGGTGAGGACCCACCCTGGCATGATCT,C,"
                      "CTCCCCTCATCACCTCCCCAGCCACGGTGAGGACCCACCCTGGCATGATCT", ".", "PASS", "DP=100"]),
         n_rows=4, n_cols=15,
         asserts=[[0, 0, "chr1"], [0, 3, "C"], [0, 1, "874816"],
                  [0, 4, "+CCCCTCATCACCTCCCCAGCCACGGTGAGGACCCACCCTGGCATGATCT"],
                  [1, 0, "chr1"], [1, 3, "C"], [1, 1, "874816"], [1, 4, "G"],
                  [2, 0, "chr1"], [2, 3, "C"], [2, 1, "874817"], [2, 4, "-49"],
                  [3, 0, "chr1"], [3, 3, "C"], [3, 1, "874816"], [3, 4, "+T"]]),
    dict(cite="main_test.go:2673-2729 TestOutputMultiallelicSnp", config=_PASSDOT,
         vcf=vcf(H8, ["1", "1265061", "rs138351882;rs563042459", "CGT", "TGT,C", ".", "PASS", "DP=100"]),
         n_rows=2, n_cols=15,
         asserts=[[0, 0, "chr1"], [0, 3, "C"], [0, 1, "1265061"], [0, 4, "T"],
                  [1, 0, "chr1"], [1, 3, "G"], [1, 1, "1265062"], [1, 4, "-2"]]),
    dict(cite="main_test.go:2731-2778 TestComplexSnp", config=_PASSDOT,
         vcf=vcf(H8, ["1", "1265062", "rs138351882;rs563042459", "CGT", "CGA", ".", "PASS", "DP=100"]),
         n_rows=1, n_cols=15, asserts=[[0, 0, "chr1"], [0, 3, "T"], [0, 1, "1265064"], [0, 4, "A"]]),
    dict(cite="main_test.go:2780-2852 TestMNP", config=_PASSDOT,
         vcf=vcf(H8, ["1", "1000", "rs138351882;rs563042459", "ACGT", "GATC", ".", "PASS", "DP=100"]),
         n_rows=4, n_cols=15,
         asserts=[[0, 0, "chr1"], [0, 3, "A"], [0, 1, "1000"], [0, 4, "G"], [1, 3, "C"], [1, 1, "1001"], [1, 4, "A"],
                  [2, 3, "G"], [2, 1, "1002"], [2, 4, "T"], [3, 3, "T"], [3, 1, "1003"], [3, 4, "C"]]),
]

# TestManyAlleles, main_test.go:2854-2908 (11 rows; the 12th expectedAlleles entry is unused)
_many_alts = ["+A", "+C", "+G", "+T", "C", "G", "T", "+TA", "+TC", "+TG", "+TT"]
_many_ac = ["4", "2", "2", "2", "2", "2", "2", "2", "2", "2", "3"]
_many_hom = ["S1;S1_2", "S2", "S3", "S4", "S5", "S6", "S7", "S8", "S9", "S10", "S11;S11_HAPLOID"]
_a = []
for r in range(11):
    _a += [[r, 8, _many_hom[r]], [r, 12, _many_ac[r]], [r, 13, "25"], [r, 4, _many_alts[r]]]
end_to_end.append(dict(
    cite="main_test.go:2854-2908 TestManyAlleles", config=_PASSDOT,
    vcf=vcf(H8 + ["FORMAT", "S1", "S1_2", "S2", "S3", "S4", "S5", "S6", "S7", "S8", "S9", "S10", "S11", "S11_HAPLOID"],
            ["1", "1000", "rs1", "A", "AA,AC,AG,AT,C,G,T,ATA,ATC,ATG,ATT", ".", "PASS", "DP=100", "GT",
             "1|1", "1|1", "2|2", "3|3", "4|4", "5|5", "6|6", "7|7", "8|8", "9|9", "10|10", "11|11", "11"]),
    n_rows=11, n_cols=15, asserts=_a))

# dosage known answers (N2, not yet on the HIP path): main_test.go:2911-2977
dosage = [
    dict(cite="main_test.go:2916,2962", line=T(["1", "1000", "rs1", "A", "T", ".", "PASS", "DP=100", "GT", "1|1", "0|1", "0|0"]),
         n_header=12, allele="1", dosages=[2, 1, 0]),
    dict(cite="main_test.go:2917,2966", line=T(["2", "200", "rs2", "C", "G", ".", "PASS", "DP=100", "GT", "0|1", "0|0", "1|1"]),
         n_header=12, allele="1", dosages=[1, 0, 2]),
    dict(cite="main_test.go:2918,2970", line=T(["22", "300", "rs2", "G", "T", ".", "PASS", "DP=100", "GT", "0|.", "0|.", "1|1"]),
         n_header=12, allele="1", dosages=[-1, -1, 2]),
]

out = dict(get_alleles=get_alleles, alt_is_valid=alt_is_valid, line_passes=line_passes,
           make_het_hom=make_het_hom, header=header, flags=flags, end_to_end=end_to_end, dosage=dosage)

if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "known_answers.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("wrote", path, {k: (len(v) if isinstance(v, list) else 1) for k, v in out.items()})
