"""Seeded synthetic VCF text for parity tests (pure Python; small sizes only)."""
import random

FIXED = ["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"]


def header(n_samples, with_format=True, names=None):
    cols = FIXED[:8]
    if n_samples or with_format:
        cols = FIXED[:9]
    names = names or ["S%05d" % i for i in range(n_samples)]
    return "##fileformat=VCFv4.2\n##source=vcfgen\n" + "\t".join(cols + names) + "\n"


def rand_bases(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def gen_ref_alt(rng, weird=0.05):
    """REF, ALT covering SNP, MNP, INS, DEL, padded/intercalated indels, multiallelics and junk"""
    kind = rng.random()
    ref = rng.choice("ACGT")
    if kind < 0.45:  # SNP (sometimes REF == ALT)
        alt = rng.choice("ACGT") if rng.random() < 0.1 else rng.choice([b for b in "ACGT" if b != ref])
    elif kind < 0.55:  # simple insertion
        alt = ref + rand_bases(rng, rng.randint(1, 6))
    elif kind < 0.65:  # simple deletion
        ref = ref + rand_bases(rng, rng.randint(1, 6))
        alt = ref[0]
    elif kind < 0.72:  # MNP / padded SNP
        n = rng.randint(2, 5)
        ref = rand_bases(rng, n)
        alt = "".join(rng.choice("ACGT") if rng.random() < 0.5 else r for r in ref)
    elif kind < 0.80:  # padded indel sharing a suffix
        core = rand_bases(rng, rng.randint(1, 4))
        suf = rand_bases(rng, rng.randint(0, 3))
        ins = rand_bases(rng, rng.randint(1, 5))
        if rng.random() < 0.5:
            ref, alt = core + suf, core + ins + suf
        else:
            ref, alt = core + ins + suf, core + suf
    elif kind < 0.95:  # multiallelic
        n = rng.randint(2, 4) if rng.random() < 0.9 else rng.randint(9, 13)
        if rng.random() < 0.5:
            ref = rand_bases(rng, rng.randint(1, 4))
        alts = []
        for _ in range(n):
            r = rng.random()
            if r < 0.4:
                alts.append(rng.choice("ACGT"))
            elif r < 0.6:
                alts.append(ref + rand_bases(rng, rng.randint(1, 4)))
            elif r < 0.75 and len(ref) > 1:
                alts.append(ref[: rng.randint(1, len(ref) - 1)])
            elif r < 0.85:
                alts.append(rand_bases(rng, len(ref)))
            elif r < 0.92:
                alts.append(rng.choice(["<CN0>", "<INS:ME:ALU>", "*", ".", "N", ""]))
            else:
                alts.append(rand_bases(rng, rng.randint(1, 7)))
        alt = ",".join(alts)
    else:  # junk
        alt = rng.choice(["<DEL>", ".", "N", "a", "]13:123456]T", "A,", ",A", "AN"])
    if rng.random() < weird:
        ref = rng.choice(["N", "a", "AN", ref])
    return ref, alt


GT_COMMON = ["0|0"] * 20 + ["0|1", "1|0", "1|1", "0/0", "0/1", "1/1", ".|.", "./.", "0|.", ".|1"]
GT_WEIRD = ["1", "0", ".", "2|1", "1|2", "2|2", "0|2", "3|1", "10|1", "1|10", "11|11", "0/1/1", "1|1|1", "1|.|1",
            "", "1|", "|1", "01|1", "1/2|1", ":|.", ".|:", "./|", "0", "12", "0|0|0", "1|0:", "x|y", "1||1"]


def gen_gt(rng, n_alts, weird):
    r = rng.random()
    if r < weird:
        return rng.choice(GT_WEIRD)
    if n_alts > 1 and r < weird + 0.3:
        a = rng.randint(0, n_alts)
        b = rng.randint(0, n_alts)
        return "%d%s%d" % (a, rng.choice("|/"), b)
    return rng.choice(GT_COMMON)


def gen_line(rng, n_samples, pos, fmt_extra=False, weird=0.03, filters=("PASS", ".", "q10", "LowQual;s50")):
    ref, alt = gen_ref_alt(rng, weird)
    chrom = rng.choice(["1", "chr1", "X", "chrM", "22", "c", "chr", "contig_1"])
    pos_s = str(pos) if rng.random() > weird else rng.choice(["abc", "", "-5", "+7", "1e3", "0", "99999999999999999999"])
    info = "AC=%d;AF=%.3f;DP=%d" % (rng.randint(1, 50), rng.random(), rng.randint(1, 5000))
    if rng.random() < 0.02:
        info += ";CSQ=" + rand_bases(rng, rng.randint(300, 2500), "ACGT|,")
    cols = [chrom, pos_s, "rs%d" % rng.randint(1, 10**6) if rng.random() < 0.7 else ".", ref, alt,
            "%d" % rng.randint(1, 999), rng.choice(filters), info]
    n_alts = alt.count(",") + 1
    if n_samples:
        fmt = "GT:DP:GQ" if fmt_extra else "GT"
        cols.append(fmt)
        for _ in range(n_samples):
            g = gen_gt(rng, n_alts, weird)
            if fmt_extra and rng.random() < 0.9:
                g += ":%d:%d" % (rng.randint(0, 99), rng.randint(0, 99))
            cols.append(g)
    r = rng.random()
    if r < weird / 2:
        cols = cols[:-1]  # too few fields
    elif r < weird:
        cols.append("0|0")  # too many fields
    return "\t".join(cols) + "\n"


def gen_vcf(seed, n_lines, n_samples, fmt_extra=False, weird=0.03, eol="\n"):
    rng = random.Random(seed)
    out = [header(n_samples)]
    pos = 10000
    for _ in range(n_lines):
        pos += rng.randint(1, 300)
        out.append(gen_line(rng, n_samples, pos, fmt_extra, weird))
        if rng.random() < weird / 4:
            out.append(rng.choice(["\n", "#comment\tline\n", "\t\t\t\n"]))
    s = "".join(out)
    if eol != "\n":
        s = s.replace("\n", eol)
    return s.encode()
