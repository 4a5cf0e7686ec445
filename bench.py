#!/usr/bin/env python3
"""bench.py — variants/sec of the HIP per-line variant pipeline on 1KG-chr1-shaped synthetic VCF.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus 2 --cpu-dry      (no GPU: the N > 1 plumbing over gloo, the CPU oracle per rank; a test)

With N > 1 and no launcher around it (WORLD_SIZE unset) the parent -- before any torch / GPU call -- starts
torch.distributed.run as a child process, relays its stdout and exits with its code.

One process per GPU.  A STEP = one pass of the whole kernel chain (line index, head/getAlleles, genotype scan,
finish) over the rank's whole resident set: `--blocks` blocks of `--rows` synthetic rows each (BASELINE.json
configs[2]: 2 504 samples, biallelic SNPs, ~10 166 B/row; by default 8 x 311 296 rows = 2.49 M rows = 25.3 GB of text
per step, so the 20 steps of the driver's run visit 49.8 M rows -- eight times the 6.2 M rows of configs[2] -- in
about 0.1 s).  A block is what one bvcf_submit takes (< 4 GiB: offsets are 32-bit); the blocks are generated on the
device before timing and visited in order, 25 GB between two visits of the same byte (the Infinity Cache holds
256 MiB), so every launch streams its text from HBM.  Blocks are dealt to `--slots` result slots (default 3 =
the library's own default since round 5: bvcf_params.n_slots 0; two are even on configs[2] and 5 % slower on configs[3]), each with its
own HIP stream, exactly as bvcf_submit deals them: the short latency-bound kernels that end one block's chain overlap
the following blocks' scans.  Records are independent: rank r owns its own rows (weak
scaling), no collective in the data path; the per-rank variant counts are summed over RCCL at the end.

Rank 0 writes the WHOLE record to bench_full.json (next to this file, and under gpurun_out/ when that exists) and
prints ONE compact JSON line (<= 3 KB: headline, config, roofline, cpu_baseline, one flat
{wall_s, variants_per_min, steady_variants_per_s, sha256_equal} per end-to-end leg) as the LAST line of stdout.
N > 1 adds ranks_seen (RCCL all-reduce of ones), per_rank_variants_per_s and e2e_all_devices (one
`bystro-vcf --devices 0,..,N-1` process over configs[2]-shaped text and BGZF, whole output hashed against the oracle's).
The whole record holds:
  roofline      the dominant kernel (k_stream on the streaming path, k_gt on the census path, k_sites on sites-only
                input).  achieved / frac: algorithmic text bytes per launch / its mean HIP-event duration with ONE
                block at a time (the kernel alone on the GPU; a pass right after the timed region).  chain_frac:
                the same bytes / the timed region's time per block (ms_per_step / blocks) / peak -- what the whole
                chain sustains with `--slots` blocks in flight.  in_timed_region_*: the kernel's HIP-event duration inside
                the timed region, where it shares the CUs with the previous block's tail kernels (longer than the
                time per block: launches overlap; informational).
  e2e           (rank 0, N == 1) the CLI end to end: `bystro-vcf --in <file in /dev/shm> > /dev/null` over
                configs[2]'s own 6.2 M rows (the rank's stream, written to /dev/shm once), every run kept with its
                wall clock and its BVCF_TIMING stage split (`runs`; `cold` = the first exec of the CLI on the box,
                `warm` = the best later run), the md5 of its output against the oracle CLI's on a prefix of the file,
                and `full_output_check`: sha256 of the WHOLE output of one more CLI run against the sha256 of what
                the oracle run of the cpu_baseline leg printed for the same file.  PCIe-inclusive: never `value`.
  e2e_bgzf      the reference's published shape (README.md:10,49: `pigz -d -c in.vcf.gz | bystro-vcf`) without the
                host decompressor: the same number of rows as a BGZF file (one block of the stream compressed at
                bgzip's level 6 and its member stream repeated -- BGZF blocks are independent) -> `--in x.vcf.gz`,
                inflated on the device; prefix md5 against the oracle; whole-output sha256 against the oracle's
                rows for that block, repeated.
  e2e_stdin     `cat file | bystro-vcf` through a real pipe, text and BGZF: the drop-in stdin surface.
  e2e_eight_workers_one_gpu  `--devices d,d,d,d,d,d,d,d`: the product's multi-device partition (eight ctxs, per-worker
                readers and formatter pools, one ordered writer) over the same files on the one GPU there is; whole
                output hashed.
  e2e_c2        the same for configs[1]'s sites-only rows (20 M of them: the packed form end to end, rows rendered on the
                device), default flags; e2e_c2_bgzf: the same rows as a BGZF file (what dbSNP / gnomAD sites files are).
  e2e_c5        the same for GATK-style rows (GT:DP:GQ sample fields, 24 KB per row: k_stream_gen), default flags.
  e2e_c4        configs[3]'s rows (20 % multiallelic + 15 % indels) with --keepId --keepInfo through the CLI, the
                whole output hashed against the oracle CLI's with the same flags.
  cpu_baseline  (rank 0, N == 1) the CPU oracle -- the C restatement of the reference algorithm, kind "port" -- over
                the same file with all host cores and with 4 threads (the README's box has 4 cores).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
CLI = os.path.join(ROOT, "bystro-vcf_amd", "bystro-vcf")
ORACLE = os.path.join(ROOT, "oracle", "bvcf_oracle")

# per profile: rows per block, resident blocks (a step visits all of them)
SHAPES = {"c2": (1_000_000, 8), "c3": (311_296, 8), "c4": (262_144, 8), "c5": (98_304, 8), "c5h": (98_304, 8)}
WORKLOADS = {"c2": "BASELINE configs[1]: sites-only, biallelic SNPs, 0 samples",
             "c3": "BASELINE configs[2]: 1KG-Phase3 chr1-shaped, 2504 samples, biallelic SNPs",
             "c4": "BASELINE configs[3]: 2504 samples, 20% multiallelic + 15% indels",
             "c5": "not a BASELINE config: 2504 samples with GT:DP:GQ fields (k_stream_gen once the ctx has seen the shape)",
             "c5h": "not a BASELINE config: as c5, one sample in twenty with haploid calls (chrX-like)"}


def rank_blocks(rank, n_blocks, rows):
    """first row of each resident block of `rank`: rank r owns rows [r*B*R, (r+1)*B*R) — disjoint
    shards, no data-path collective (records are independent, main.go:534-698)"""
    return [(rank * n_blocks + b) * rows for b in range(n_blocks)]


def reduce_over_ranks(elapsed, n_variants, device, world):
    """slowest rank's time and the job's total variant count (the only collective: RCCL on GPUs)"""
    import torch
    import torch.distributed as dist
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=device)
    n_var = torch.tensor([float(n_variants)], dtype=torch.float64, device=device)
    if dist.is_initialized():
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        dist.all_reduce(n_var, op=dist.ReduceOp.SUM)
    return float(t_el.item()), float(n_var.item())


def _hash_stdout(cmd, stdin_path=None, algo="md5", env=None, timeout_s=900):
    """(return code, hex digest of stdout, bytes of stdout, stderr text, wall seconds)"""
    h = hashlib.new(algo)
    n = 0
    t0 = time.perf_counter()
    with open(stdin_path or os.devnull, "rb") as f, tempfile.TemporaryFile() as errf:
        p = subprocess.Popen(cmd, stdin=f, stdout=subprocess.PIPE, stderr=errf, env=env)
        try:
            for chunk in iter(lambda: p.stdout.read(1 << 24), b""):
                h.update(chunk)
                n += len(chunk)
                if time.perf_counter() - t0 > timeout_s:
                    p.kill()
                    break
            p.wait()
        finally:
            if p.poll() is None:
                p.kill()
        errf.seek(0, 2)
        errf.seek(max(0, errf.tell() - 4096))
        err = errf.read().decode(errors="replace")
    return p.returncode, h.hexdigest(), n, err, time.perf_counter() - t0


def _md5_stdout(cmd, stdin_path=None):
    rc, hx, _, _, _ = _hash_stdout(cmd, stdin_path)
    return rc, hx


def _stages_of(stderr_text):
    for ln in reversed(stderr_text.splitlines()):
        if ln.startswith("[bvcf timing-json] "):
            return json.loads(ln[len("[bvcf timing-json] "):])
    return None


_SPLICE_PRODUCER = ("import fcntl, os, sys\n"
                    "fd = os.open(sys.argv[1], os.O_RDONLY)\n"
                    "fcntl.fcntl(1, 1031, 1 << 20)  # F_SETPIPE_SZ\n"
                    "while os.splice(fd, 1, 1 << 20):\n"
                    "    pass\n")


def _run_cli(args, cat_path=None, timeout_s=300, splice=False):
    """one timed run of the CLI with stdout -> /dev/null.  cat_path: `cat <path> | bystro-vcf` through a real pipe
    (stdin is then the pipe's read end, as in the reference's `pigz -d -c in.vcf.gz | bystro-vcf`, README.md:10).
    splice: the producer is not cat (read() + write(): two copies in one thread) but splice(2) of the file's page-cache
    pages into the pipe (no copy on the producer's side): what the CLI's own pipe reader can take.
    -> {"wall_s", "stages"} or {"error"}"""
    env = dict(os.environ, BVCF_TIMING="json")
    t0 = time.perf_counter()
    cat = None
    try:
        with open(os.devnull, "wb") as out, tempfile.TemporaryFile() as errf:
            if cat_path:
                cat = subprocess.Popen([sys.executable, "-c", _SPLICE_PRODUCER, cat_path] if splice else ["cat", cat_path], stdout=subprocess.PIPE)
                p = subprocess.Popen([CLI] + args, stdin=cat.stdout, stdout=out, stderr=errf, env=env)
                cat.stdout.close()  # the CLI holds the read end now
            else:
                p = subprocess.Popen([CLI] + args, stdin=subprocess.DEVNULL, stdout=out, stderr=errf, env=env)
            try:
                p.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
                return {"error": "the CLI did not finish within %d s" % timeout_s}
            wall = time.perf_counter() - t0
            errf.seek(0, 2)
            errf.seek(max(0, errf.tell() - 16384))
            err = errf.read().decode(errors="replace")
    finally:
        if cat is not None:
            if cat.poll() is None:
                cat.kill()
            cat.wait()
    if p.returncode != 0:
        return {"error": "CLI rc %d: %s" % (p.returncode, err[-300:])}
    return {"wall_s": wall, "stages": _stages_of(err)}


def e2e_leg(what, args, rows, text_bytes, runs=2, cat_path=None, timeout_s=300, file_bytes=None, splice=False):
    """`runs` runs of the CLI, every one kept (wall clock around the process, BVCF_TIMING=json stage split).  A run that
    does not come back within timeout_s is reported, not waited for: the bench line must not depend on this leg."""
    out = {"input": what, "argv": " ".join((["splice(FILE -> pipe)" if splice else "cat FILE", "|"] if cat_path else []) + ["bystro-vcf"] + args),
           "rows": rows, "text_bytes": text_bytes, "runs": []}
    if file_bytes is not None:
        out["file_bytes"] = file_bytes
    for _ in range(runs):
        r = _run_cli(args, cat_path, timeout_s, splice)
        if "error" in r:
            out["error"] = r["error"]
            break
        st = r["stages"] or {}
        r["variants_per_s"] = rows / r["wall_s"]
        if st.get("steady_s"):
            r["steady_variants_per_s"] = rows / st["steady_s"]
        if st and st.get("lines_in") != rows:
            out["error"] = "the CLI saw %s lines, the input has %d" % (st.get("lines_in"), rows)
        out["runs"].append(r)
    if not out["runs"]:
        return out
    best = min(out["runs"], key=lambda r: r["wall_s"])
    out.update({"wall_s": best["wall_s"], "variants_per_s": rows / best["wall_s"], "variants_per_min": rows / best["wall_s"] * 60,
                "text_GBps": text_bytes / best["wall_s"] / 1e9, "stages": best["stages"],
                "meets_50M_variants_per_min": rows / best["wall_s"] * 60 >= 50e6})
    st = best["stages"] or {}
    if st.get("steady_s"):
        # from the first block's submit to the last byte written: the run without process / HIP start-up and teardown
        out["steady_variants_per_s"] = rows / st["steady_s"]
        out["steady_variants_per_min"] = rows / st["steady_s"] * 60
        out["steady_text_GBps"] = text_bytes / st["steady_s"] / 1e9
    return out


def cold_warm(leg):
    """the first exec of the CLI on the box against the best later one, with where the difference went (every stage
    whose time differs by more than 50 ms between the two runs)"""
    rs = leg.get("runs") or []
    if len(rs) < 2:
        return None
    cold, warm = rs[0], min(rs[1:], key=lambda r: r["wall_s"])
    out = {"cold_wall_s": cold["wall_s"], "warm_wall_s": warm["wall_s"], "ratio": cold["wall_s"] / warm["wall_s"]}
    cs, ws = cold.get("stages") or {}, warm.get("stages") or {}
    diff = {}
    for k, v in cs.items():
        if isinstance(v, (int, float)) and isinstance(ws.get(k), (int, float)) and k.endswith("_s") or k in (
                "major_faults", "minor_faults", "in_blocks", "vol_ctx_switches", "invol_ctx_switches"):
            if isinstance(v, (int, float)) and isinstance(ws.get(k), (int, float)):
                d = v - ws[k]
                if (k.endswith("_s") and abs(d) > 0.05) or (not k.endswith("_s") and d):
                    diff[k] = {"cold": v, "warm": ws[k]}
    out["stages_that_differ"] = diff
    out["outside_the_run_s"] = {"cold": cold["wall_s"] - cs.get("total_s", 0.0), "warm": warm["wall_s"] - ws.get("total_s", 0.0),
                                "what": "wall clock around the process minus bvcf_run_fd's own total: exec, dynamic loading of "
                                        "the HIP runtime's libraries, exit"}
    return out


def prefix_check(hip_args, hip_in, oracle_in, oracle_args, rows):
    """parity on a prefix of the same stream: md5 of the CLI's stdout == md5 of the oracle CLI's"""
    rc_g, m_g = _md5_stdout([CLI, "--in", hip_in] + hip_args)
    rc_o, m_o = _md5_stdout([ORACLE, "--in", oracle_in, "--threads", str(min(usable_cpus(), 64))] + oracle_args)
    return {"rows": rows, "hip": m_g, "oracle": m_o, "equal": rc_g == 0 and rc_o == 0 and m_g == m_o}


def bgzf_of(data_mv, level=6, threads=16, block=0xFF00):
    """BGZF member stream of a buffer (no EOF marker), bgzip's framing and level; zlib releases the GIL, so threads do"""
    import bgzf
    from concurrent.futures import ThreadPoolExecutor
    n = len(data_mv)
    span = block * 64
    def part(off):
        return b"".join(bgzf.bgzf_block(bytes(data_mv[o:min(o + block, off + span, n)]), level)
                        for o in range(off, min(off + span, n), block))
    with ThreadPoolExecutor(max(1, threads)) as ex:
        return b"".join(ex.map(part, range(0, n, span)))


def write_e2e_file(header, blocks, sizes, want_rows, rows_per_block, make_block):
    """header + `want_rows` rows of the rank's stream (whole blocks: the resident ones first, the rest generated on the
    device as they are written) into /dev/shm, or /tmp when that is too small.  The row count is cut to what half of the
    free space holds.  -> (path, rows, bytes, where)"""
    import torch
    per_block = max(sizes)
    base, n_blk = "/tmp", 1
    for cand in ("/dev/shm", "/tmp"):
        try:
            st = os.statvfs(cand)
        except OSError:
            continue
        fit = int(st.f_bavail * st.f_frsize * 0.5 // per_block)
        if fit >= 1:
            base, n_blk = cand, max(1, min(-(-want_rows // rows_per_block), fit))
            break
    path = os.path.join(base, "bvcf_bench_e2e_%d.vcf" % os.getpid())
    stage = torch.empty(1 << 28, dtype=torch.uint8, pin_memory=True)
    total = len(header)
    with open(path, "wb") as f:
        f.write(header)
        for b in range(n_blk):
            t, nb = (blocks[b], sizes[b]) if b < len(blocks) else make_block(b)
            for off in range(0, nb, stage.numel()):
                n = min(stage.numel(), nb - off)
                stage[:n].copy_(t[off:off + n])
                f.write(memoryview(stage[:n].numpy()))
            total += nb
            del t
    return path, n_blk * rows_per_block, total, base


def real_data_leg(bv, bg, cfg, device, args, kernel, synthetic_GBps):
    """the same kernel over REAL 1000-Genomes lines (the reference's regression input, tests/golden/: 19 747 rows x 2 504
    samples, replicated to one block), one block at a time: the synthetic row model lets the scan skip 53 % of its
    chunks as all-reference -- this shows what the real allele-count spectrum and real line heads do to the rate"""
    import gzip
    import torch
    with gzip.open(os.path.join(ROOT, "tests", "golden", "1kg_chr1_20klines.vcf.gz"), "rb") as f:
        raw = f.read()
    body = raw[raw.index(b"\n", raw.index(b"#CHROM")) + 1:]
    n_body = body.count(b"\n")
    reps = max(1, args.rows // n_body)
    rows = reps * n_body
    dev = torch.frombuffer(bytearray(body), dtype=torch.uint8).cuda()
    t = torch.full((len(body) * reps + bv.DEVICE_PAD,), 10, dtype=torch.uint8, device="cuda")
    for r in range(reps):
        t[r * len(body):(r + 1) * len(body)] = dev
    nbytes = len(body) * reps
    ns = cfg.n_samples
    stride = ((ns + 3) // 4 + 15) & ~15
    n_alt_cap = rows * 4 + 1024
    ctx = bv.Ctx(bg.n_header_fields(cfg), device=device, max_batch_bytes=nbytes, n_slots=1, max_lines=rows + 16, max_alleles=n_alt_cap,
                 cmap_bytes=min((n_alt_cap + nbytes // (4 * ns + 8) + 16 * 8192) * stride + 4096, 0xFFFFFF00), path=args.path)
    try:
        ctx.bench_device([t.data_ptr()], [nbytes], 2, slots=1)
        chain, scan, counts = ctx.bench_device([t.data_ptr()], [nbytes], 8, slots=1)
    finally:
        ctx.close()
    ms = sum(scan) / len(scan)
    gbps = nbytes / (ms * 1e-3) / 1e9
    return {
        "input": "tests/golden/1kg_chr1_20klines.vcf.gz: %d real rows x %d = %d rows, %.2f GB per block, resident in HBM" % (n_body, reps, rows, nbytes / 1e9),
        "kernel": kernel, "mean_launch_ms": ms, "GBps": gbps, "frac": gbps / HBM_PEAK_GBPS,
        "chain_ms_one_block_at_a_time": sum(chain) / len(chain),
        "variants_per_s_one_block_at_a_time": rows / (sum(chain) / len(chain) * 1e-3),
        "ratio_to_synthetic_kernel_rate": gbps / synthetic_GBps if synthetic_GBps else None,
    }


def usable_cpus():
    """CPUs this process can actually use: hardware threads, affinity mask and the cgroup CPU quota, whichever is least
    (the GPU boxes show 256 hardware threads to a container that gets 16 cores' worth of time)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    rel = ""
    try:
        for ln in open("/proc/self/cgroup"):
            if ln.startswith("0::"):
                rel = ln[3:].strip().rstrip("/")
    except OSError:
        pass
    while True:
        try:
            q, per = open("/sys/fs/cgroup" + rel + "/cpu.max").read().split()[:2]
            if q != "max" and int(per) > 0:
                n = min(n, max(1, -(-int(q) // int(per))))
        except (OSError, ValueError):
            pass
        if not rel:
            break
        rel = rel[:rel.rfind("/")] if "/" in rel else ""
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            n = min(n, max(1, -(-q // per)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(path, rows, profile, oracle_args=(), four=True):
    """oracle/bvcf_oracle (the CLI of the C restatement: N workers over 64-line batches, split-then-scan, per-allele
    rescan) over the e2e file (mapped, not copied), as in README.md:49: all the cores this process may use -- that run's
    output goes through sha256 for the whole-output check of the e2e legs; it is printed after the timed readVcf, so the
    rate does not see the pipe -- and 4 threads with the output to /dev/null"""
    cores = usable_cpus()

    def run(threads, hashed):
        cmd = [ORACLE, "--in", path, "--threads", str(threads), "--timing"] + list(oracle_args)
        if hashed:
            rc, hx, n_out, err, wall = _hash_stdout(cmd, algo="sha256")
        else:
            t0 = time.perf_counter()
            with open(os.devnull, "wb") as out:
                p = subprocess.run(cmd, stdout=out, stderr=subprocess.PIPE, timeout=900)
            wall, rc, err, hx, n_out = time.perf_counter() - t0, p.returncode, p.stderr.decode(errors="replace"), None, None
        m = re.search(r"\[oracle timing\] rows (\d+) threads (\d+) read ([\d.]+) process ([\d.]+) write ([\d.]+)", err)
        assert rc == 0 and m and int(m.group(1)) == rows, err[-300:]
        return {"threads": threads, "process_s": float(m.group(4)), "read_s": float(m.group(3)), "write_s": float(m.group(5)),
                "wall_s": wall, "variants_per_s": rows / float(m.group(4)), "output_sha256": hx, "output_bytes": n_out}

    full = run(cores, True)
    out = {
        "value": full["variants_per_s"], "unit": "variants/s", "cores": cores, "kind": "port",
        "sample": "%d rows of the same synthetic %s stream (the e2e file), oracle/bvcf_oracle%s with %d worker threads over 64-line "
                  "batches (the host shows %d hardware threads, the CPU quota of this process is %d cores); rate = "
                  "rows / readVcf time (%.2f s), input mapped into memory beforehand (%.2f s), output hashed afterwards"
                  % (rows, profile, (" " + " ".join(oracle_args)) if oracle_args else "", cores, os.cpu_count() or 1, cores,
                     full["process_s"], full["read_s"]),
        "all_cores": full,
        "published_reference": "README.md:44-52: 6.2 M variants in 2 m 45 s on a 4-core i3.2xlarge = 37.6 k variants/s (Go, pigz-bound)",
    }
    if four:
        f4 = run(4, False)
        out["threads_4"] = {"value": f4["variants_per_s"], "unit": "variants/s", "cores": 4, "process_s": f4["process_s"],
                            "wall_s": f4["wall_s"]}
    return out


def full_output_check(hip_args, oracle_run, what):
    """one more (untimed) CLI run whose WHOLE output is hashed, against the hash of what the oracle printed for the same
    file in the cpu_baseline leg"""
    rc, hx, n_out, err, wall = _hash_stdout([CLI] + hip_args, algo="sha256")
    return {"what": what, "hip_sha256": hx, "oracle_sha256": oracle_run["output_sha256"], "output_bytes": n_out,
            "oracle_output_bytes": oracle_run["output_bytes"], "hip_rc": rc,
            "equal": rc == 0 and hx == oracle_run["output_sha256"] and n_out == oracle_run["output_bytes"]}


def host_legs(line, args, cfg, bg, bv, blocks, sizes, rank, local_rank, release_device):
    """everything that runs the CLI (rank 0, N == 1).  The files live in /dev/shm and are removed in any case."""
    import torch
    ns = cfg.n_samples
    dev = str(local_rank)
    files = []

    def tmp(name):
        q = "%s.%s" % (files[0], name)
        files.append(q)
        return q

    try:
        hdr = bg.header(cfg)
        first0 = rank_blocks(rank, args.blocks, args.rows)[0]
        path, f_rows, f_bytes, where = write_e2e_file(
            hdr, blocks, sizes, args.e2e_rows, args.rows,
            lambda b: bg.rows_device(cfg, first0 + b * args.rows, args.rows, pad=bv.DEVICE_PAD))
        files.append(path)
        n_blk = f_rows // args.rows
        if not args.no_e2e:
            # prefix of the same stream for the md5 checks (rows [first, first + n) of block 0), as text and as BGZF; and
            # block 0 alone as BGZF members, which the compressed file repeats
            p_rows = min(args.rows, 65_536 if ns else 1_000_000)
            pt, pn = bg.rows_device(cfg, first0, p_rows, pad=bv.DEVICE_PAD)
            prefix, prefix_gz, block0, gz = tmp("prefix"), tmp("prefix.gz"), tmp("block0"), tmp("gz")
            p_host = pt[:pn].cpu().numpy()
            with open(prefix, "wb") as f:
                f.write(hdr)
                f.write(memoryview(p_host))
            import bgzf as _bg
            eof = _bg.bgzf_block(b"")
            hdr_gz = bgzf_of(memoryview(hdr))
            with open(prefix_gz, "wb") as f:
                f.write(hdr_gz + bgzf_of(memoryview(p_host), threads=usable_cpus()) + eof)
            del pt, p_host
            t0 = time.perf_counter()
            b0 = blocks[0][:sizes[0]].cpu().numpy()
            with open(block0, "wb") as f:
                f.write(hdr)
                f.write(memoryview(b0))
            members = bgzf_of(memoryview(b0), threads=usable_cpus())
            del b0
            with open(gz, "wb") as f:
                f.write(hdr_gz)
                for _ in range(n_blk):
                    f.write(members)
                f.write(eof)
            gz_bytes, gz_text = os.path.getsize(gz), len(hdr) + n_blk * sizes[0]
            t_gz = time.perf_counter() - t0
            del members
        # the device is the CLI's from here on: the resident blocks and the bench ctx go
        release_device()
        blocks.clear()
        torch.cuda.empty_cache()

        base = None
        if not args.no_cpu_baseline:
            base = line["cpu_baseline"] = cpu_baseline(path, f_rows, args.profile)
        if not args.no_e2e:
            src = "%d rows, %.2f GB of the same synthetic stream in %s" % (f_rows, f_bytes / 1e9, where)
            e = e2e_leg(src + " -> bystro-vcf --in (HIP) -> /dev/null", ["--in", path, "--devices", dev], f_rows, f_bytes, runs=3)
            e["devices"] = dev
            e["cold_vs_warm"] = cold_warm(e)
            e["md5_check"] = prefix_check(["--devices", dev], prefix, prefix, [], p_rows)
            if base:
                e["full_output_check"] = full_output_check(["--in", path, "--devices", dev], base["all_cores"],
                                                           "sha256 of all %d rows' output, CLI vs the oracle run of cpu_baseline" % f_rows)
            line["e2e"] = e

            g = e2e_leg("%d rows as BGZF (level 6, %.3f GB for %.2f GB of text, made in %.1f s: block 0 of the stream compressed once, "
                        "its member stream %d times) in %s -> bystro-vcf --in x.vcf.gz, inflated on the device -> /dev/null"
                        % (f_rows, gz_bytes / 1e9, gz_text / 1e9, t_gz, n_blk, where),
                        ["--in", gz, "--devices", dev], f_rows, gz_text, runs=3, file_bytes=gz_bytes)
            g["md5_check"] = prefix_check(["--devices", dev], prefix_gz, prefix, [], p_rows)
            # the whole output: the oracle's rows for header + block 0, n_blk times behind one header line
            po = subprocess.run([ORACLE, "--in", block0, "--threads", str(usable_cpus())], stdout=subprocess.PIPE,
                                stderr=subprocess.DEVNULL)
            rc_o, ob = po.returncode, po.stdout
            cut = ob.index(b"\n") + 1
            h = hashlib.sha256(ob[:cut])
            for _ in range(n_blk):
                h.update(memoryview(ob)[cut:])
            want_n = cut + n_blk * (len(ob) - cut)
            rc, hx, n_out, _, _ = _hash_stdout([CLI, "--in", gz, "--devices", dev], algo="sha256")
            g["full_output_check"] = {"what": "sha256 of all %d rows' output vs the oracle's rows for block 0, %d times" % (f_rows, n_blk),
                                      "hip_sha256": hx, "oracle_sha256": h.hexdigest(), "output_bytes": n_out,
                                      "equal": rc == 0 and rc_o == 0 and hx == h.hexdigest() and n_out == want_n}
            del ob
            line["e2e_bgzf"] = g

            line["e2e_stdin"] = {
                "text": e2e_leg("cat <the e2e file: " + src + "> | bystro-vcf (stdin is a pipe) -> /dev/null", ["--devices", dev],
                                f_rows, f_bytes, runs=1, cat_path=path, timeout_s=240),
                "text_spliced": e2e_leg("the same file through a pipe whose producer does not copy (splice(2) of the page cache "
                                        "into the pipe, 1 MiB at a time) | bystro-vcf -> /dev/null: the CLI's pipe reader, not cat's speed",
                                        ["--devices", dev], f_rows, f_bytes, runs=1, cat_path=path, timeout_s=240, splice=True),
                "bgzf": e2e_leg("cat <the BGZF file of e2e_bgzf> | bystro-vcf (stdin is a pipe) -> /dev/null", ["--devices", dev],
                                f_rows, gz_text, runs=2, cat_path=gz, file_bytes=gz_bytes),
                "note": "README.md:10,49 runs the reference as `pigz -d -c in.vcf.gz | bystro-vcf`; the BGZF pipe is that run "
                        "without pigz (the device inflates), the text pipe is what a decompressor in front would have to deliver",
            }
            # the product's multi-device partition at configs[2]'s size on the one GPU there is: eight workers --
            # eight ctxs, each with its own byte ranges of the file, readers and formatter pool, one ordered writer -- over
            # the same 63 GB, text and BGZF; the whole text output hashed like the one-worker run's
            eight = ",".join([dev] * 8)
            w8 = {"text": e2e_leg(src + " -> bystro-vcf --in --devices " + eight + " (eight workers on one GPU)",
                                  ["--in", path, "--devices", eight], f_rows, f_bytes, runs=1, timeout_s=180),
                  "bgzf": e2e_leg("the BGZF file of e2e_bgzf -> bystro-vcf --in --devices " + eight, ["--in", gz, "--devices", eight],
                                  f_rows, gz_text, runs=1, timeout_s=180, file_bytes=gz_bytes)}
            if base and "output_sha256" in base.get("all_cores", {}):
                w8["full_output_check"] = full_output_check(
                    ["--in", path, "--devices", eight], base["all_cores"],
                    "sha256 of all %d rows' output with eight workers, CLI vs the oracle run of cpu_baseline" % f_rows)
            line["e2e_eight_workers_one_gpu"] = w8
            n_vis = torch.cuda.device_count()
            if n_vis > 1:
                # (not part of `value`, which is this rank's GPU alone) the same files dealt range by range to every visible
                # device by the one CLI process
                all_dev = ",".join(str(d) for d in range(n_vis))
                line["e2e_all_devices"] = {
                    "text": e2e_leg(src + " -> bystro-vcf --in --devices " + all_dev, ["--in", path, "--devices", all_dev],
                                    f_rows, f_bytes, runs=2, timeout_s=180),
                    "bgzf": e2e_leg("the BGZF file of e2e_bgzf -> bystro-vcf --in --devices " + all_dev, ["--in", gz, "--devices", all_dev],
                                    f_rows, gz_text, runs=2, timeout_s=180, file_bytes=gz_bytes)}
        for q in files:
            if os.path.exists(q):
                os.unlink(q)
        del files[:]

        # ---- two more files end to end, whole output against the oracle: configs[3] with --keepId --keepInfo, and GATK-style
        # sample fields (GT:DP:GQ: the file shape most cohort VCFs have; k_stream_gen)
        if not args.no_e2e and not args.no_cpu_baseline and args.profile == "c3":
            for key, prof, want, flags, what in (
                    ("e2e_c4", "c4", args.e2e_c4_rows, ["--keepId", "--keepInfo"], "BASELINE configs[3]'s synthetic stream (20% multiallelic + 15% indels)"),
                    ("e2e_c5", "c5", args.e2e_c5_rows, [], "GATK-style rows (2 504 samples, GT:DP:GQ, 24 KB per row: not a BASELINE config)"),
                    ("e2e_c2", "c2", args.e2e_c2_rows, [], "BASELINE configs[1]'s synthetic stream (sites-only, biallelic: the packed form of the results)")):
                if want <= 0:
                    continue
                cfg_x = bg.make_cfg(prof)
                r_x = SHAPES[prof][0]
                n_x = max(1, -(-want // r_x))
                per_row = 25_000 if prof == "c5" else (160 if prof == "c2" else 4 * cfg_x.n_samples + 400)
                path_x, rows_x, bytes_x, where_x = write_e2e_file(
                    bg.header(cfg_x), [], [r_x * per_row], n_x * r_x, r_x,
                    lambda b, cfg_x=cfg_x, r_x=r_x: bg.rows_device(cfg_x, b * r_x, r_x, pad=bv.DEVICE_PAD))
                files.append(path_x)
                torch.cuda.empty_cache()
                base_x = cpu_baseline(path_x, rows_x, prof, oracle_args=flags, four=False)
                leg = e2e_leg("%d rows, %.2f GB of %s in %s -> bystro-vcf --in %s-> /dev/null" % (
                                  rows_x, bytes_x / 1e9, what, where_x, " ".join(flags) + (" " if flags else "")),
                              ["--in", path_x, "--devices", dev] + flags, rows_x, bytes_x, runs=2)
                leg["full_output_check"] = full_output_check(
                    ["--in", path_x, "--devices", dev] + flags, base_x["all_cores"],
                    "sha256 of all %d rows' output%s, CLI vs oracle CLI" % (rows_x, " with " + " ".join(flags) if flags else ""))
                leg["cpu_baseline"] = {k: base_x[k] for k in ("value", "unit", "cores", "kind", "sample")}
                line[key] = leg
                os.unlink(path_x)
                if prof == "c2":
                    # the shape such files come in (dbSNP / gnomAD sites: .vcf.gz): the same number of rows as BGZF -- block
                    # 0's member stream n_x times -- inflated on the device, rows rendered there, only the lines left to
                    # the host come back as text; whole output against the oracle's rows of block 0, n_x times
                    import bgzf as _bg
                    t0, nb0 = bg.rows_device(cfg_x, 0, r_x, pad=bv.DEVICE_PAD)
                    b0 = t0[:nb0].cpu().numpy()
                    del t0
                    hdr_x = bg.header(cfg_x)
                    blk0 = path_x + ".block0"
                    files.append(blk0)
                    with open(blk0, "wb") as f:
                        f.write(hdr_x)
                        f.write(memoryview(b0))
                    members = bgzf_of(memoryview(b0), threads=usable_cpus())
                    del b0
                    gz_x = path_x + ".gz"
                    files.append(gz_x)
                    with open(gz_x, "wb") as f:
                        f.write(bgzf_of(memoryview(hdr_x)))
                        for _ in range(n_x):
                            f.write(members)
                        f.write(_bg.bgzf_block(b""))
                    del members
                    gz_bytes_x, gz_text_x = os.path.getsize(gz_x), len(hdr_x) + n_x * nb0
                    gleg = e2e_leg("%d sites-only rows as BGZF (level 6, %.3f GB for %.2f GB of text: block 0's members %d times) -> "
                                   "bystro-vcf --in x.vcf.gz -> /dev/null" % (n_x * r_x, gz_bytes_x / 1e9, gz_text_x / 1e9, n_x),
                                   ["--in", gz_x, "--devices", dev], n_x * r_x, gz_text_x, runs=2, file_bytes=gz_bytes_x)
                    po = subprocess.run([ORACLE, "--in", blk0, "--threads", str(usable_cpus())], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
                    ob = po.stdout
                    cut = ob.index(b"\n") + 1
                    h = hashlib.sha256(ob[:cut])
                    for _ in range(n_x):
                        h.update(memoryview(ob)[cut:])
                    rc_g, hx_g, n_g, _, _ = _hash_stdout([CLI, "--in", gz_x, "--devices", dev], algo="sha256")
                    gleg["full_output_check"] = {"what": "sha256 of all %d rows' output vs the oracle's rows for block 0, %d times" % (n_x * r_x, n_x),
                                                 "hip_sha256": hx_g, "oracle_sha256": h.hexdigest(), "output_bytes": n_g,
                                                 "equal": rc_g == 0 and po.returncode == 0 and hx_g == h.hexdigest() and n_g == cut + n_x * (len(ob) - cut)}
                    del ob
                    line["e2e_c2_bgzf"] = gleg
                    for q in (blk0, gz_x):
                        os.unlink(q)
    except Exception as exc:  # the host legs inform; the measured line above stands without them
        import traceback
        line.setdefault("host_legs_error", (repr(exc) + " @ " + traceback.format_exc().splitlines()[-3].strip())[:500])
    finally:
        for q in files:
            if q and os.path.exists(q):
                os.unlink(q)


def all_devices_leg(line, args, cfg, bg, bv, blocks, sizes, world, release_device):
    """N > 1, rank 0, after the timed region: the product's own multi-device run -- ONE `bystro-vcf --devices 0,..,N-1`
    process, contiguous byte ranges of the file dealt to N workers (one ctx per GPU), one ordered writer -- over
    configs[2]-shaped rows as text and as BGZF; the whole text output hashed against the oracle CLI's (the checker; no
    cpu_baseline at N > 1)."""
    import torch
    files = []
    try:
        hdr = bg.header(cfg)
        first0 = rank_blocks(0, args.blocks, args.rows)[0]
        path, f_rows, f_bytes, where = write_e2e_file(
            hdr, blocks, sizes, args.all_devices_rows, args.rows,
            lambda b: bg.rows_device(cfg, first0 + b * args.rows, args.rows, pad=bv.DEVICE_PAD))
        files.append(path)
        n_blk = f_rows // args.rows
        import bgzf as _bg
        b0 = blocks[0][:sizes[0]].cpu().numpy()
        members = bgzf_of(memoryview(b0), threads=usable_cpus())
        del b0
        gz = path + ".gz"
        files.append(gz)
        with open(gz, "wb") as f:
            f.write(bgzf_of(memoryview(hdr)))
            for _ in range(n_blk):
                f.write(members)
            f.write(_bg.bgzf_block(b""))
        gz_bytes, gz_text = os.path.getsize(gz), len(hdr) + n_blk * sizes[0]
        del members
        release_device()
        blocks.clear()
        torch.cuda.empty_cache()
        devs = ",".join(str(d) for d in range(world))
        src = "%d rows, %.2f GB of the same synthetic stream in %s" % (f_rows, f_bytes / 1e9, where)
        out = {"devices": devs,
               "text": e2e_leg(src + " -> bystro-vcf --in --devices " + devs, ["--in", path, "--devices", devs], f_rows, f_bytes,
                               runs=2, timeout_s=240),
               "bgzf": e2e_leg("the same rows as BGZF (block 0 of the stream %d times, %.2f GB) -> bystro-vcf --in x.vcf.gz --devices %s"
                               % (n_blk, gz_bytes / 1e9, devs), ["--in", gz, "--devices", devs], f_rows, gz_text, runs=2,
                               timeout_s=240, file_bytes=gz_bytes)}
        rc_o, hx_o, n_o, _, _ = _hash_stdout([ORACLE, "--in", path, "--threads", str(min(usable_cpus(), 64))], algo="sha256")
        chk = full_output_check(["--in", path, "--devices", devs], {"output_sha256": hx_o, "output_bytes": n_o},
                                "sha256 of all %d rows' output over %d devices, CLI vs the oracle CLI" % (f_rows, world))
        chk["equal"] = chk["equal"] and rc_o == 0
        out["full_output_check"] = chk
        line["e2e_all_devices"] = out
    except Exception as exc:  # informs; the measured line stands without it
        line["host_legs_error"] = repr(exc)[:300]
    finally:
        for q in files:
            if os.path.exists(q):
                os.unlink(q)


LIBRARY_DEFAULT_SLOTS = 3  # bvcf_params.n_slots == 0 (include/bvcf.h; 2 until round 5)
LINE_LIMIT = 3072  # bytes: the driver keeps 8 KB of stdout; the verdict asks for <= 3 KB


def _sig(x, n=5):
    """floats to n significant digits (the line is a record, not a data file)"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (n, x))


def _leg_summary(leg):
    """ONE flat summary of an end-to-end leg: {wall_s, variants_per_min, steady_variants_per_s, sha256_equal}"""
    if not isinstance(leg, dict):
        return None
    if "wall_s" not in leg:
        return {"error": str(leg.get("error", "no run"))[:80]}
    chk = leg.get("full_output_check")
    out = {"wall_s": _sig(leg["wall_s"], 4), "variants_per_min": _sig(leg.get("variants_per_min"), 4),
           "steady_variants_per_s": _sig(leg.get("steady_variants_per_s"), 4),
           "sha256_equal": (bool(chk["equal"]) if isinstance(chk, dict) and "equal" in chk else None)}
    if "error" in leg:
        out["error"] = str(leg["error"])[:80]
    return out


def compact_line(full):
    """the ONE line the driver reads (last line of stdout, <= LINE_LIMIT bytes): headline, config, roofline,
    cpu_baseline and one flat summary per end-to-end leg.  Everything else stays in bench_full.json."""
    line = {k: _sig(full.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                              "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = full.get("config") or {}
    line["config"] = {k: _sig(cfg[k]) for k in ("workload", "rows_per_step_per_gpu", "rows_per_block", "resident_blocks_per_gpu",
                                                 "bytes_per_row", "n_samples", "input", "blocks_in_flight", "flags") if k in cfg}
    rf = full.get("roofline")
    if isinstance(rf, dict):
        line["roofline"] = {k: _sig(rf.get(k)) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                                           "algorithmic_bytes_per_launch", "mean_launch_ms", "chain_frac", "chain_ms_one_block_at_a_time")}
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = {k: _sig(cb.get(k)) for k in ("value", "unit", "cores", "kind")}
        line["cpu_baseline"]["sample"] = str(cb.get("sample_short") or cb.get("sample", ""))[:120]
    for k in ("variants_per_min", "text_GBps", "ranks_seen", "value_at_library_default_slots"):
        if k in full:
            line[k] = _sig(full[k])
    if "per_rank_variants_per_s" in full:
        line["per_rank_variants_per_s"] = [_sig(v, 4) for v in full["per_rank_variants_per_s"]]
    if isinstance(full.get("real_data"), dict) and "frac" in full["real_data"]:
        line["real_data_frac"] = _sig(full["real_data"]["frac"], 4)
    for key in ("e2e", "e2e_bgzf", "e2e_c2", "e2e_c2_bgzf", "e2e_c4", "e2e_c5"):
        if key in full:
            line[key] = _leg_summary(full[key])
    for key, subs in (("e2e_stdin", ("text", "text_spliced", "bgzf")), ("e2e_eight_workers_one_gpu", ("text", "bgzf")),
                      ("e2e_all_devices", ("text", "bgzf"))):
        grp = full.get(key)
        if isinstance(grp, dict):
            for sub in subs:
                if sub in grp:
                    sm = _leg_summary(grp[sub])
                    if sub == "text" and isinstance(grp.get("full_output_check"), dict):
                        sm["sha256_equal"] = bool(grp["full_output_check"].get("equal"))
                    line["%s_%s" % (key, sub)] = sm
    if "host_legs_error" in full:
        line["host_legs_error"] = str(full["host_legs_error"])[:160]
    line["full"] = "bench_full.json"
    txt = json.dumps(line, separators=(",", ":"))
    # never let the record outgrow what the driver reads: drop the least important keys first
    for k in ("real_data_frac", "e2e_eight_workers_one_gpu_bgzf", "e2e_stdin_text_spliced", "e2e_all_devices_bgzf", "e2e_c2_bgzf",
              "e2e_eight_workers_one_gpu_text", "e2e_stdin_bgzf", "e2e_stdin_text", "per_rank_variants_per_s"):
        if len(txt) <= LINE_LIMIT:
            break
        line.pop(k, None)
        txt = json.dumps(line, separators=(",", ":"))
    assert len(txt) <= LINE_LIMIT, len(txt)
    return txt


def emit(full):
    """bench_full.json next to bench.py (and under gpurun_out/ when it exists), then the compact line as the LAST line
    of stdout"""
    blob = json.dumps(full, indent=1)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_full.json"), "w") as f:
                    f.write(blob)
            except OSError:
                pass
    sys.stdout.flush()
    print(compact_line(full), flush=True)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(argv, n):
    """`python bench.py --gpus N` without a launcher: this parent -- which has made no GPU call and has imported neither
    torch nor the library -- starts one rank per GPU as a CHILD process (torch.distributed.run), relays its stdout and
    exits with its code; rank 0's compact line is repeated at the end if anything was printed after it."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "4")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)
    last_json, last = None, None
    for raw in p.stdout:
        ln = raw.decode(errors="replace").rstrip("\n")
        print(ln, flush=True)
        last = ln
        if ln.startswith('{"metric"'):
            last_json = ln
    rc = p.wait()
    if last_json is not None and last is not last_json:
        print(last_json, flush=True)
    return rc


def cpu_dry(args):
    """--cpu-dry: the N > 1 plumbing of this file -- launch, rank shards, barrier, max-over-ranks clock, the count
    reduction, rank 0's line -- over gloo with NO GPU: every rank runs the CPU oracle (the checker, as tests/test_dist_cpu.py
    does) over its own small shard.  It measures nothing about the product and says so in `metric`."""
    import torch
    import torch.distributed as dist
    import benchgen as bg
    import oracle_lib as orc
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    rows, n_blocks = min(args.rows, 400), min(args.blocks, 2)
    cfg = bg.make_cfg(args.profile, **({"n_samples": args.samples} if args.samples else {}))
    hdr = bg.header(cfg)
    texts = [hdr + bg.rows_host(cfg, first, rows) for first in rank_blocks(rank, n_blocks, rows)]

    def step():
        n = 0
        for t in texts:
            rc, _, _, k = orc.run(t)
            assert rc == 0
            n += k
        return n
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    n_var = sum(step() for _ in range(args.steps))
    if world > 1:
        dist.barrier()
    mine = time.perf_counter() - t0
    elapsed, total = reduce_over_ranks(mine, n_var, "cpu", world)
    ones, rates = rank_census(n_var / mine, "cpu", world)
    if rank == 0:
        emit({"metric": "cpu-dry plumbing check (oracle per rank over gloo; NOT the product, NOT variants/sec of anything shipped)",
              "value": total / elapsed, "unit": "variants/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
              "dtype": "u8", "data": "synthetic",
              "config": {"workload": "cpu-dry: " + WORKLOADS[args.profile], "rows_per_step_per_gpu": rows * n_blocks, "rows_per_block": rows,
                         "resident_blocks_per_gpu": n_blocks, "n_samples": cfg.n_samples, "input": "host memory (no GPU)"},
              "ranks_seen": ones, "per_rank_variants_per_s": rates, "rows_in_timed_region": int(total)})
    if world > 1:
        dist.destroy_process_group()


def rank_census(my_rate, device, world):
    """(ranks seen = all-reduce of ones, every rank's own variants/s): the N > 1 line shows that every rank took part"""
    import torch
    import torch.distributed as dist
    one = torch.ones(1, dtype=torch.float64, device=device)
    rates = torch.zeros(world, dtype=torch.float64, device=device)
    rates[int(os.environ.get("RANK", "0")) if world > 1 else 0] = my_rate
    if dist.is_initialized():
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        dist.all_reduce(rates, op=dist.ReduceOp.SUM)
    return int(one.item()), [float(v) for v in rates.tolist()]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=0, help="rows per block per GPU (0 = the profile's default)")
    ap.add_argument("--blocks", type=int, default=0, help="resident blocks per GPU; a step visits all of them (0 = default)")
    ap.add_argument("--profile", default="c3", choices=["c2", "c3", "c4", "c5", "c5h"])
    ap.add_argument("--samples", type=int, default=0, help="experiment: another sample count for the profile (e.g. 100000 with --rows 640)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-real-data", action="store_true", help="skip the cross-check on real 1000-Genomes lines")
    ap.add_argument("--e2e-rows", type=int, default=6_200_000,
                    help="rows of the end-to-end / cpu_baseline file (rounded up to whole blocks; BASELINE configs[2] is 6.2 M rows = "
                         "63 GB, written to /dev/shm -- fewer when it does not hold them)")
    ap.add_argument("--e2e-c4-rows", type=int, default=1_048_576,
                    help="rows of the configs[3] end-to-end leg (--keepId --keepInfo, whole output hashed against the oracle); 0 = skip")
    ap.add_argument("--e2e-c2-rows", type=int, default=20_000_000,
                    help="rows of the sites-only (configs[1]) end-to-end leg, whole output hashed against the oracle; 0 = skip")
    ap.add_argument("--e2e-c5-rows", type=int, default=2_064_384,
                    help="rows of the GATK-style (GT:DP:GQ) end-to-end leg (50 GB of text: a run of 1.2 s, so that the CLI's 0.2 s of "
                         "start-up does not decide whether it meets 50 M variants/min), whole output hashed against the oracle; 0 = skip")
    ap.add_argument("--align16", action="store_true", help="experiment: 16-byte aligned sample regions")
    ap.add_argument("--no-class-maps", action="store_true", help="experiment: counts only, no 2-bit class maps")
    ap.add_argument("--no-packed-sites", action="store_true",
                    help="sites-only input (profile c2): the full form of the results (128 bytes per line) instead of the packed one "
                         "(ABI 6, 32 bytes per line, what the CLI asks for)")
    ap.add_argument("--path", type=int, default=0, help="0 choose, 1 census path, 2 streaming path")
    ap.add_argument("--slots", type=int, default=3,
                    help="blocks in flight per GPU: block i runs on slot i %% slots, each slot on its own HIP stream, as "
                         "bvcf_submit deals them (bvcf_params.n_slots; the library's default); 1 = strictly one block after "
                         "the other")
    ap.add_argument("--golden", action="store_true",
                    help="experiment: real 1000-Genomes lines (tests/golden/1kg_chr1_20klines.vcf.gz, 19 747 rows "
                         "replicated to --rows) instead of the synthetic model")
    ap.add_argument("--cpu-dry", action="store_true",
                    help="no GPU: the launch / shard / reduce plumbing over gloo with the CPU oracle per rank (a test of this "
                         "file's N > 1 path, not a measurement)")
    ap.add_argument("--over", default="", help="experiment: row-model knobs of the profile changed, e.g. p_multi=0,p_indel=0")
    ap.add_argument("--all-devices-rows", type=int, default=6_200_000,
                    help="N > 1: rows of rank 0's `bystro-vcf --devices 0,..,N-1` leg after the timed region; 0 = skip")
    args = ap.parse_args()
    d_rows, d_blocks = SHAPES[args.profile]
    args.rows = args.rows or d_rows
    args.blocks = args.blocks or d_blocks

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: be the launcher (before anything here touches torch or the GPU)
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.cpu_dry:
        return cpu_dry(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    host_group = None
    # (BVCF_BENCH_FORCE_MULTI=1, with the launcher's environment for ONE rank: the N > 1 code -- RCCL group, gloo group,
    # reductions, rank 0's all-devices leg, the host-side wait -- on a one-GPU box; tests/test_gpu_bench_legs.py)
    multi = world > 1 or os.environ.get("BVCF_BENCH_FORCE_MULTI") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        # a host-side group for the wait at the very end (rank 0 runs the CLI over every device then: the other ranks
        # must not sit in a spinning RCCL kernel on theirs)
        import datetime
        host_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(minutes=30))

    import benchgen as bg
    import bystro_vcf_amd as bv

    over = {k: int(v) for k, v in (kv.split("=") for kv in args.over.split(",") if kv)}
    cfg = bg.make_cfg(args.profile, align16=int(args.align16), **({"n_samples": args.samples} if args.samples else {}), **over)
    ns = cfg.n_samples
    # ---- synthetic blocks, generated on this rank's GPU; rank r owns rows [r*B*R, (r+1)*B*R)
    blocks, sizes = [], []
    if args.golden:
        import gzip
        with gzip.open(os.path.join(ROOT, "tests", "golden", "1kg_chr1_20klines.vcf.gz"), "rb") as f:
            raw = f.read()
        body = raw[raw.index(b"\n", raw.index(b"#CHROM")) + 1:]
        n_body = body.count(b"\n")
        reps = max(1, args.rows // n_body)
        args.rows = reps * n_body
        host = torch.frombuffer(bytearray(body), dtype=torch.uint8)
        for b in range(args.blocks):
            t = torch.full((len(body) * reps + bv.DEVICE_PAD,), 10, dtype=torch.uint8, device="cuda")
            dev = host.cuda()
            for r in range(reps):
                t[r * len(body):(r + 1) * len(body)] = dev
            blocks.append(t)
            sizes.append(len(body) * reps)
        torch.cuda.synchronize()
    else:
        for first in rank_blocks(rank, args.blocks, args.rows):
            t, nbytes = bg.rows_device(cfg, first, args.rows, pad=bv.DEVICE_PAD)
            blocks.append(t)
            sizes.append(nbytes)
    max_bytes = max(sizes)
    stride = ((ns + 3) // 4 + 15) & ~15
    # (c5: the first launches go through k_stream, which leaves every line of this shape to a k_gt task of its own)
    n_alt_cap = args.rows * (4 if args.profile == "c4" or args.golden else 2 if args.profile.startswith("c5") else 1) + 1024
    ctx = bv.Ctx(bg.n_header_fields(cfg), device=local_rank, max_batch_bytes=max_bytes, n_slots=max(1, args.slots),
                 max_lines=args.rows + 16, max_alleles=n_alt_cap,
                 # (the streaming path hands every wave a range of map slots sized by bytes, one per 4 ns + 8 of them,
                 # plus two of slack)
                 cmap_bytes=min((n_alt_cap + (max_bytes // (4 * ns + 8) if ns else 0) + 16 * 8192) * stride + 4096, 0xFFFFFF00),
                 path=args.path,
                 want_class_maps=not args.no_class_maps, packed_sites=ns == 0 and not args.no_packed_sites)
    ptrs = [t.data_ptr() for t in blocks]

    def barrier():
        if multi:
            dist.barrier()

    # ---- warm-up, then exactly K timed steps between barrier + synchronize on both sides; a step = every block once
    n_launch = args.steps * args.blocks
    if args.warmup:
        ctx.bench_device(ptrs, sizes, args.warmup * args.blocks)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    chain_ms, gt_ms, counts = ctx.bench_device(ptrs, sizes, n_launch)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0

    assert counts[0] == args.rows, counts
    # the final count gather over RCCL/xGMI (and the slowest rank's clock)
    my_rate = args.rows * n_launch / elapsed
    elapsed, total_variants = reduce_over_ranks(elapsed, args.rows * n_launch, "cuda", world)
    ranks_seen, per_rank = rank_census(my_rate, "cuda", world)
    # outside the timed region (rank 0): the same chain strictly one block after the other, for the dominant
    # kernel's duration when it has the GPU to itself
    alone_ms, alone_chain_ms, default_slots_rate = None, None, None
    if rank == 0:
        a_chain, alone, _ = ctx.bench_device(ptrs, sizes, max(args.blocks, 8), slots=1)
        alone_ms = sum(alone) / len(alone)
        alone_chain_ms = sum(a_chain) / len(a_chain)
        if not multi and args.slots != LIBRARY_DEFAULT_SLOTS:
            # the same timed region with the library's default number of blocks in flight (bvcf_params.n_slots = 0)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ctx.bench_device(ptrs, sizes, n_launch, slots=LIBRARY_DEFAULT_SLOTS)
            torch.cuda.synchronize()
            default_slots_rate = args.rows * n_launch / (time.perf_counter() - t1)

    if rank == 0:
        mean_bytes = sum(sizes) / len(sizes)
        gt_mean_ms = sum(gt_ms) / len(gt_ms)
        chain_mean_ms = sum(chain_ms) / len(chain_ms)
        ms_per_block = elapsed / n_launch * 1e3
        # algorithmic bytes of one launch of the dominant kernel.  Census path with samples, k_gt: the GT text of
        # every row (4 bytes per sample).  Streaming path, k_stream, and sites-only input, k_sites: every byte of
        # every row (it is also the pass that finds the lines).
        streaming = ctx.path() == 2
        sites_kernel = {"0": "k_head", "1": "k_sites", "3": "k_sites1"}.get(os.environ.get("BVCF_SITES", ""),
                                                                              "k_sites2" if args.no_packed_sites else "k_sites2p")
        kernel = ctx.stream_kernel() if streaming else ("k_gt" if ns else sites_kernel)
        alg_bytes = int(mean_bytes) if (streaming or not ns) else args.rows * 4 * ns
        achieved = alg_bytes / (alone_ms * 1e-3) / 1e9 if alone_ms else None
        pmc = None
        try:
            with open(os.path.join(ROOT, "profiles", "k_gt_hbm_traffic.json")) as f:
                pmc = json.load(f).get(args.profile, {}).get("k_stream_traffic_bytes_per_launch_per_row" if streaming else "traffic_bytes_per_launch_per_row")
        except OSError:
            pass
        line = {
            "metric": "variants/sec",
            "value": total_variants / elapsed,
            "unit": "variants/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": WORKLOADS[args.profile] + (" with " + args.over if args.over else ""),
                "rows_per_step_per_gpu": args.rows * args.blocks, "rows_per_block": args.rows,
                "resident_blocks_per_gpu": args.blocks, "rows_in_timed_region": int(total_variants),
                "bytes_per_row": mean_bytes / args.rows, "bytes_per_step_per_gpu": sum(sizes), "n_samples": ns,
                "flags": "default (--allowFilter PASS,.), class maps on", "input": "resident in HBM",
                "blocks_in_flight": args.slots,
            },
            "roofline": {
                "bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBPS) if achieved else None,
                "traffic": (pmc * args.rows) if pmc else None,
                # NOT measured in this run (PMC counters need rocprofv3 around the process): the per-row figure of the
                # committed counter passes, times this run's rows per launch
                "traffic_source": "profiles/k_gt_hbm_traffic.json (rocprofv3 --pmc passes, tools/derive_traffic.py); not measured in this run" if pmc else None,
                "algorithmic_bytes_per_launch": alg_bytes,
                "mean_launch_ms": alone_ms,
                "how": "HIP events around the kernel on its launch stream, one block at a time (the kernel alone on the GPU), "
                       "%d launches right after the timed region" % max(args.blocks, 8),
                # what the whole chain sustains inside the timed region: bytes per block / wall time per block
                "chain_frac": mean_bytes / (ms_per_block * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "chain_GBps": mean_bytes / (ms_per_block * 1e-3) / 1e9,
                "ms_per_block": ms_per_block,
                "chain_ms_one_block_at_a_time": alone_chain_ms,
                # informational: the kernel's duration inside the timed region, where it shares the CUs with the
                # previous blocks' tail kernels -- launches overlap, so this is longer than ms_per_block
                "in_timed_region_mean_launch_ms": gt_mean_ms,
                "in_timed_region_frac": alg_bytes / (gt_mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if gt_mean_ms else None,
                "in_timed_region_chain_latency_ms": chain_mean_ms,
            },
            "text_GBps": sum(sizes) * world / (elapsed / args.steps) / 1e9,
            "variants_per_min": total_variants / elapsed * 60,
        }
        if default_slots_rate:
            line["value_at_library_default_slots"] = default_slots_rate
            line["config"]["library_default_slots"] = LIBRARY_DEFAULT_SLOTS
        if multi:
            line["ranks_seen"] = ranks_seen
            line["per_rank_variants_per_s"] = per_rank
        if not multi and streaming and args.profile == "c3" and not args.golden and not args.no_real_data:
            try:
                line["real_data"] = real_data_leg(bv, bg, cfg, local_rank, args, kernel, achieved)
            except Exception as exc:
                line["real_data"] = {"error": repr(exc)[:300]}
        want_host_legs = not multi and not args.golden and not (args.no_e2e and args.no_cpu_baseline)
        if want_host_legs:
            host_legs(line, args, cfg, bg, bv, blocks, sizes, rank, local_rank, ctx.close)
    if multi:
        # every rank lets go of its device (the CLI of rank 0 is about to use all of them), then waits on the host
        if rank == 0 and args.all_devices_rows > 0 and not args.no_e2e and not args.golden:
            all_devices_leg(line, args, cfg, bg, bv, blocks, sizes, world, ctx.close)
        ctx.close()
        blocks.clear()
        torch.cuda.empty_cache()
        dist.barrier(group=host_group)
    if rank == 0:
        emit(line)
    ctx.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
