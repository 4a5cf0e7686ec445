#!/usr/bin/env python3
"""bench.py — variants/sec of the HIP per-line variant pipeline on 1KG-chr1-shaped synthetic VCF.

    python bench.py --gpus N --steps K --warmup W            (N == 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  A step = one pass of the whole kernel chain (line index, head/getAlleles,
genotype scan, finish) over one resident batch of `--rows` synthetic rows (BASELINE.json configs[2]:
2 504 samples, biallelic SNPs, ~10 164 B/row).  `--blocks` distinct batches are generated on the
device before timing and visited round-robin, each far larger than the 256 MiB Infinity Cache, so
every step streams its text from HBM.  Batches are dealt to `--slots` result slots (default 2, the
library's default), each with its own HIP stream, exactly as bvcf_submit deals them: the short
latency-bound kernels that end one batch's chain overlap the next batch's scan.  Records are independent: rank r owns its own rows (weak
scaling), no collective in the data path; the per-rank variant counts are summed over RCCL at the end.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (k_stream on the streaming
path, k_gt on the census path): achieved = algorithmic text bytes per launch / its mean HIP-event
duration inside the timed region.  `cpu_baseline` (rank 0, N == 1 only) times the CPU oracle — the C restatement of the
reference algorithm, kind "port" — on a bounded sample of the same row model.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(profile, rows_total=400_000, chunk=25_000):
    """the oracle (oracle/bvcf_oracle.c), all host cores, on a bounded sample of the same rows"""
    import benchgen as bg
    import oracle_lib as orc

    cfg = bg.make_cfg(profile)
    hdr = bg.header(cfg)
    cores = os.cpu_count() or 1
    elapsed, rows, out_rows = 0.0, 0, 0
    for first in range(0, rows_total, chunk):
        vcf = hdr + bg.rows_host(cfg, 10_000_000 + first, chunk)
        t0 = time.perf_counter()
        rc, out, _, n = orc.run(vcf, None, n_threads=cores)
        elapsed += time.perf_counter() - t0
        assert rc == 0 and n == chunk
        rows += n
        out_rows += out.count(b"\n")
    return {
        "value": rows / elapsed, "unit": "variants/s", "cores": cores, "kind": "port",
        "sample": "%d rows of the same synthetic %s stream (%d-row chunks), oracle/bvcf_oracle.c with %d worker "
                  "threads over 64-line batches, output discarded; %.1f s wall" % (rows, profile, chunk, cores, elapsed),
    }


def rank_blocks(rank, n_blocks, rows):
    """first row of each resident batch of `rank`: rank r owns rows [r*B*R, (r+1)*B*R) — disjoint
    shards, no data-path collective (records are independent, main.go:534-698)"""
    return [(rank * n_blocks + b) * rows for b in range(n_blocks)]


def reduce_over_ranks(elapsed, n_variants, device, world):
    """slowest rank's time and the job's total variant count (the only collective: RCCL on GPUs)"""
    import torch
    import torch.distributed as dist
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=device)
    n_var = torch.tensor([float(n_variants)], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        dist.all_reduce(n_var, op=dist.ReduceOp.SUM)
    return float(t_el.item()), float(n_var.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=131072, help="rows per step per GPU")
    ap.add_argument("--blocks", type=int, default=4, help="distinct resident batches per GPU")
    ap.add_argument("--profile", default="c3", choices=["c2", "c3", "c4", "c5"])
    ap.add_argument("--samples", type=int, default=0, help="experiment: another sample count for the profile (e.g. 100000 with --rows 640)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--align16", action="store_true", help="experiment: 16-byte aligned sample regions")
    ap.add_argument("--no-class-maps", action="store_true", help="experiment: counts only, no 2-bit class maps")
    ap.add_argument("--path", type=int, default=0, help="0 choose, 1 census path, 2 streaming path")
    ap.add_argument("--slots", type=int, default=2,
                    help="batches in flight per GPU: batch i runs on slot i %% slots, each slot on its own HIP stream, as "
                         "bvcf_submit deals them (bvcf_params.n_slots, library default 2); 1 = strictly one batch after "
                         "the other")
    ap.add_argument("--golden", action="store_true",
                    help="experiment: real 1000-Genomes lines (tests/golden/1kg_chr1_20klines.vcf.gz, 19 747 rows "
                         "replicated to --rows) instead of the synthetic model")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import benchgen as bg
    import bystro_vcf_amd as bv

    if args.profile == "c5" and args.path == 0:
        args.path = 1  # FORMAT is not plain GT: the host driver (choose_path) sends such files down the census path
    cfg = bg.make_cfg(args.profile, align16=int(args.align16), **({"n_samples": args.samples} if args.samples else {}))
    ns = cfg.n_samples
    # ---- synthetic batches, generated on this rank's GPU; rank r owns rows [r*B*R, (r+1)*B*R)
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
    n_alt_cap = args.rows * (4 if args.profile == "c4" or args.golden else 1) + 1024
    ctx = bv.Ctx(bg.n_header_fields(cfg), device=local_rank, max_batch_bytes=max_bytes, n_slots=max(1, args.slots),
                 max_lines=args.rows + 16, max_alleles=n_alt_cap,
                 cmap_bytes=(n_alt_cap + 16 * 8192) * stride + 4096,  # + the slack of the streaming path's per-wave slot ranges
                 path=args.path,
                 want_class_maps=not args.no_class_maps)
    ptrs = [t.data_ptr() for t in blocks]

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- warm-up, then exactly K timed steps between barrier + synchronize on both sides
    if args.warmup:
        ctx.bench_device(ptrs, sizes, args.warmup)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    chain_ms, gt_ms, counts = ctx.bench_device(ptrs, sizes, args.steps)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0

    assert counts[0] == args.rows, counts
    # the final count gather over RCCL/xGMI (and the slowest rank's clock)
    elapsed, total_variants = reduce_over_ranks(elapsed, args.rows * args.steps, "cuda", world)
    # outside the timed region (rank 0): the same chain strictly one batch after the other, for the dominant
    # kernel's duration when it has the GPU to itself
    alone_ms = None
    if rank == 0 and args.slots > 1:
        _, alone, _ = ctx.bench_device(ptrs, sizes, max(4, min(args.steps, 8)), slots=1)
        alone_ms = sum(alone) / len(alone)

    if rank == 0:
        mean_bytes = sum(sizes[i % args.blocks] for i in range(args.steps)) / args.steps
        gt_mean_ms = sum(gt_ms) / len(gt_ms)
        chain_mean_ms = sum(chain_ms) / len(chain_ms)
        # algorithmic bytes of one launch of the dominant kernel.  Census path, k_gt: the GT text of
        # every row (4 bytes per sample).  Streaming path, k_stream: every byte of every row (it is
        # also the pass that finds the lines).
        streaming = ctx.path() == 2
        kernel = "k_stream" if streaming else "k_gt"
        gt_bytes = int(mean_bytes) if streaming else args.rows * 4 * ns
        achieved = gt_bytes / (gt_mean_ms * 1e-3) / 1e9 if ns else None
        pmc = None
        try:
            with open(os.path.join(ROOT, "profiles", "k_gt_hbm_traffic.json")) as f:
                pmc = json.load(f).get(args.profile, {}).get("k_stream_traffic_bytes_per_launch_per_row" if ctx.path() == 2 else "traffic_bytes_per_launch_per_row")
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
                "workload": {"c2": "BASELINE configs[1]: sites-only, 1M biallelic SNPs, 0 samples",
                             "c3": "BASELINE configs[2]: 1KG-Phase3 chr1-shaped, 2504 samples, biallelic SNPs",
                             "c4": "BASELINE configs[3]: 2504 samples, 20% multiallelic + 15% indels",
                             "c5": "not a BASELINE config: 2504 samples with GT:DP:GQ fields (general scan path)"}[args.profile],
                "rows_per_step_per_gpu": args.rows, "resident_batches_per_gpu": args.blocks,
                "bytes_per_row": mean_bytes / args.rows, "n_samples": ns,
                "flags": "default (--allowFilter PASS,.), class maps on", "input": "resident in HBM",
                "batches_in_flight": args.slots,
            },
            "roofline": {
                "bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBPS) if achieved else None,
                "traffic": (pmc * args.rows) if pmc else None,
                "algorithmic_bytes_per_launch": gt_bytes, "mean_launch_ms": gt_mean_ms,
                # (informational, measured after the timed region) the same kernel with one batch at a time: in the
                # timed region the end of the previous batch's chain shares the GPU with it
                "mean_launch_ms_one_batch_at_a_time": alone_ms,
                "frac_one_batch_at_a_time": (gt_bytes / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if alone_ms and ns else None,
                "note": ("with --slots > 1 a launch shares the GPU with the kernels that end the previous batch's chain, so "
                         "its duration (mean_launch_ms, achieved, frac) is longer than the step time would suggest; "
                         "*_one_batch_at_a_time is the same kernel with the GPU to itself") if args.slots > 1 else None,
            },
            # one batch's kernel chain from its first to its last kernel (HIP events): a latency -- with more than one
            # batch in flight consecutive chains overlap, and the step time is ms_per_step
            "chain": {"mean_ms": chain_mean_ms},
            "text_GBps": mean_bytes * world / (elapsed / args.steps) / 1e9,
            "variants_per_min": total_variants / elapsed * 60,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.profile)
        print(json.dumps(line))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
