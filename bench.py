#!/usr/bin/env python3
"""bench.py -- BGZF-BAM full scan -> Arrow on MI355X (BASELINE.json config 2).

One "step" = one complete pass of the hot path over one synthetic coordinate-sorted BGZF-BAM:
BGZF inflate of every member -> record boundary scan -> field extract -> Arrow column buffers
in HBM, for `SELECT *` (the 12 core columns).  The compressed file is resident in HBM when the
timed region starts; nothing is cached between steps (every step re-inflates and re-extracts).

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--blocks B]
N > 1: one rank per GPU.  Started by a launcher (torch.distributed.run sets WORLD_SIZE) the process is one rank; started
bare (`python bench.py --gpus N`) it spawns the N ranks as child processes itself, before any GPU call, and relays
rank 0's line.  Default workload at N > 1 is config 5: ONE file of N x --blocks members whose BAI plan is
sharded in order over the ranks, every rank uploading and inflating only its run (weak scaling, no collective on the
data path); `--mode shards` gives every rank its own independent file instead.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DEFAULT_BLOCKS = 655360  # config 2: ~10 GiB compressed BGZF-BAM
HBM_PEAK_GBS = 8000.0    # MI355X HBM3E peak (MI355X_MICROARCH.md)
K1_NAME = "k_bgzf_inflate_v4"


COLL_DEVICE = "cuda"  # where the bench's own collectives (barrier, MAX of times, SUM of counts) live


def bench_fastq(args, pkg, rank, local_rank, world, torch, dist):
    """BGZF-FASTQ + GZI full scan (4 Utf8 columns), one partition per rank over its own file."""
    synth = os.path.join(ROOT, "tools", "_build", "synth_fastq")
    if not os.path.exists(synth):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools")])
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    path = os.path.join(shm, f"bioscan_synth_{os.getpid()}_r{rank}.fastq.bgz")
    ncpu = os.cpu_count() or 1
    t0 = time.time()
    meta = json.loads(subprocess.check_output([synth, path, str(args.blocks), str(42 + rank),
                                               str(max(1, min(16, ncpu // max(1, world))))]).decode())
    t_gen = time.time() - t0
    prov = pkg.FastqTableProvider(path, None, device_id=local_rank)
    plan = prov.scan(target_partitions=1)
    assert plan.num_partitions() == 1
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU baseline: the C restatement (oracle/bioscan_oracle.c::oracle_fastq_scan_mem) on a bounded sample of the same file
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import c_oracle
        nb_s = min(int(meta["n_blocks"]), 65536)
        need = int(meta["compressed_bytes"] * 1.1 * (nb_s + 1) / max(1, int(meta["n_blocks"]))) + (1 << 20)
        with open(path, "rb") as f:
            head = f.read(need)
        thr = max(1, min(16, ncpu))
        r = c_oracle.fastq_scan(head, threads=thr, max_blocks=nb_s)
        cpu = {"value": round(r["n_rows"] / r["seconds_total"] / 1e6, 3), "unit": "Mrec/s", "cores": thr, "kind": "port",
               "sample": f"first {r['n_blocks']} BGZF members of the same file ({r['inflated_bytes'] / 1e9:.2f} GB text, {r['n_rows']} reads), "
                         f"all four columns, {'libdeflate' if r['used_libdeflate'] else 'zlib'} inflate, {thr} threads (oracle/bioscan_oracle.c)",
               "decoded_GB_s": round(r["inflated_bytes"] / r["seconds_total"] / 1e9, 3), "seconds": round(r["seconds_total"], 3),
               "seconds_inflate": round(r["seconds_inflate"], 3)}
        del head
    for p in (path, path + ".gzi"):
        if not args.keep_file:
            try:
                os.unlink(p)
            except OSError:
                pass

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    st = None
    for _ in range(args.warmup):
        st = plan.execute_device(0, args.batch_size)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = plan.execute_device(0, args.batch_size)
    sync()
    elapsed = time.perf_counter() - t0
    rows, ubytes = float(st["n_rows"]), float(st["inflated_bytes"])
    if int(st["n_rows"]) != int(meta["n_records"]):  # full-size property: every read the generator wrote comes back
        raise SystemExit(f"rank {rank}: the scan returned {st['n_rows']} rows, the generator wrote {meta['n_records']} reads")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=COLL_DEVICE)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([rows, ubytes], dtype=torch.float64, device=COLL_DEVICE)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        rows, ubytes = [float(x) for x in c.tolist()]
    e2e = None
    if rank == 0 and world == 1 and not args.no_end_to_end:
        # file resident in HBM -> Arrow buffers in host memory through the chunk pipeline (FastqExecState); never part of `value`
        runs = [plan.execute_drain(0, args.batch_size) for _ in range(2)]
        r = runs[-1]
        if int(r["n_rows"]) != int(meta["n_records"]):
            raise SystemExit(f"end-to-end stream returned {r['n_rows']} rows, the generator wrote {meta['n_records']} reads")
        e2e = {"Mrec_s": round(r["n_rows"] / r["seconds"] / 1e6, 3), "seconds": round(r["seconds"], 3),
               "ms_to_first_batch": round(r["seconds_to_first_batch"] * 1e3, 2), "n_batches": r["n_batches"],
               "first_run_seconds": round(runs[0]["seconds"], 3),
               "what": "bioscan_execute + bioscan_next until end of stream, every batch released at once; chunks of 2048 doubling to "
                       "%d BGZF members, HBM and host footprint O(chunk)" % int(os.environ.get("BIOSCAN_CHUNK_MEMBERS", 16384))}
    if rank == 0:
        per_step = elapsed / args.steps
        cu = float(st["compressed_bytes"]) + float(st["inflated_bytes"])
        achieved = cu / (st["ms_inflate"] * 1e-3) / 1e9
        print(json.dumps({
            "end_to_end": e2e,
            "metric": "bgzf_fastq_full_scan_records_per_sec", "value": round(rows / per_step / 1e6, 3), "unit": "Mrec/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(per_step * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "BGZF-FASTQ + GZI full scan, synthetic 101bp reads", "n_blocks_per_gpu": meta["n_blocks"],
                       "records_per_gpu": meta["n_records"], "compressed_bytes_per_gpu": meta["compressed_bytes"],
                       "inflated_bytes_per_gpu": meta["inflated_bytes"], "batch_size": args.batch_size},
            "decoded_GB_s": round(ubytes / per_step / 1e9, 3),
            "stage_ms": {"inflate": round(st["ms_inflate"], 3), "crc32": round(st["ms_crc"], 3),
                         "newline_index": round(st["ms_chain"], 3), "extract": round(st["ms_extract"], 3)},
            "roofline": {"bound": "hbm", "kernel": K1_NAME, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None},
            "cpu_baseline": cpu,
            "reference_published": {"source": "openspec/changes/refactor-single-thread-partition-reads/design.md:29-36",
                                    "Mrec_s": {"1 thread": 1.49, "8 threads": 11.2, "8 partitions reader+consumer threads": 17.4},
                                    "note": "different file (523 MB, 26.5 M reads) and unstated developer machine"},
            "setup_s": {"generate": round(t_gen, 1)}}))
    if world > 1:
        dist.destroy_process_group()


def bench_vcf(args, pkg, rank, local_rank, world, torch, dist):
    """BASELINE.json configs 3 and 4.
    vcf-sites:   sites-only VCF.bgz + TBI, predicate chrom='chr1' -> TBI region pruning, then scan of every column.
    vcf-samples: N-sample VCF.bgz, full scan of every column incl. `genotypes: Struct<GT,GQ,DP: List<..>>`, then the
                 list UDFs of benchmarks/vcf/vcf_multisample_bench.rs:46-59 restricted to list_avg / list_gte / list_lte."""
    synth = os.path.join(ROOT, "tools", "_build", "synth_vcf")
    if not os.path.exists(synth):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools")])
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    path = os.path.join(shm, f"bioscan_synth_{os.getpid()}_r{rank}.vcf.gz")
    ncpu = os.cpu_count() or 1
    thr = str(max(1, min(16, ncpu // max(1, world))))
    sites = args.format == "vcf-sites"
    n_lines = args.lines if args.lines else (60_000_000 if sites else 200_000)
    t0 = time.time()
    cmd = [synth, "sites", path, str(n_lines), str(42 + rank), thr] if sites else \
          [synth, "samples", path, str(n_lines), str(args.samples), str(42 + rank), thr]
    meta = json.loads(subprocess.check_output(cmd).decode())
    t_gen = time.time() - t0
    prov = pkg.VcfTableProvider(path, device_id=local_rank)
    prov.make_resident()
    filters = [("chrom", "=", "chr1")] if sites else []
    plan = prov.scan(filters=filters, target_partitions=1)
    assert plan.num_partitions() == 1
    cpu_in = None
    if rank == 0 and not args.no_cpu_baseline:
        # inputs of the CPU baseline (C oracle on the same bytes): file image + the member / text range of the scan
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import struct
        import vcf_oracle
        data = open(path, "rb").read()
        b0 = b1 = x0 = x1 = 0
        coffs, uoffs, o, uo = [], [], 0, 0
        while o + 28 <= len(data):
            bs = struct.unpack_from("<H", data, o + 16)[0] + 1
            coffs.append(o)
            uoffs.append(uo)
            uo += struct.unpack_from("<I", data, o + bs - 4)[0]
            o += bs
        coffs.append(o)
        uoffs.append(uo)
        if sites:
            tbi = vcf_oracle.parse_tbi(open(path + ".tbi", "rb").read())
            ch = vcf_oracle.tbi_query_chunks(tbi, tbi.names.index("chr1"), None, None)
            import bisect
            b0 = bisect.bisect_left(coffs, ch[0][0] >> 16)
            be = bisect.bisect_left(coffs, ch[-1][1] >> 16)
            b1 = min(be + 2, len(coffs) - 1)
            x0 = ch[0][0] & 0xFFFF
            x1 = uoffs[be] - uoffs[b0] + (ch[-1][1] & 0xFFFF)
        else:
            # header length: decode the leading members until the #CHROM line is complete
            import zlib
            txt, k = b"", 0
            while True:
                xl = struct.unpack_from("<H", data, coffs[k] + 10)[0]
                txt += zlib.decompress(data[coffs[k] + 12 + xl:coffs[k + 1] - 8], -15)
                k += 1
                i = txt.find(b"\n#CHROM")
                j = txt.find(b"\n", i + 1) if i >= 0 else -1
                if j >= 0:
                    x0 = j + 1
                    break
        cpu_in = (data, b0, b1, x0, x1)
    for p in (path, path + ".tbi"):
        if not args.keep_file:
            try:
                os.unlink(p)
            except OSError:
                pass
    udfs = [("GQ", "list_avg", 0.0), ("DP", "list_avg", 0.0), ("GQ", "list_gte", 10), ("DP", "list_gte", 10), ("DP", "list_lte", 200)]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        if sites:
            return plan.execute_device(0, args.batch_size), []
        ds = plan.execute_device_stream(0, args.batch_size)
        us = [ds.list_udf(f, u, t) for f, u, t in udfs]
        st = ds.stats
        ds.close()
        return st, us
    st, us = None, []
    for _ in range(args.warmup):
        st, us = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st, us = step()
    sync()
    elapsed = time.perf_counter() - t0
    rows, ubytes = float(st["n_rows"]), float(st["inflated_bytes"])
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=COLL_DEVICE)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([rows, ubytes], dtype=torch.float64, device=COLL_DEVICE)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        rows, ubytes = [float(x) for x in c.tolist()]
    cpu = None
    if cpu_in is not None:
        import c_oracle
        data, b0, b1, x0, x1 = cpu_in
        cores = min(16, ncpu)
        info = [("AC", "list_int"), ("AN", "int"), ("AF", "list_float"), ("DP", "int")]
        if sites:
            info += [("DB", "flag"), ("SEGDUP", "flag"), ("LCR", "flag"), ("VT", "string"), ("RSRC", "list_string"),
                     ("CULPRIT", "string"), ("VQSLOD", "float")]
        r = c_oracle.vcf_scan(data, info, 0 if sites else args.samples, True, cores, b0, b1, x0, x1)
        del data, cpu_in
        if int(r["n_rows"]) != int(st["n_rows"]):
            raise SystemExit(f"CPU baseline decoded {r['n_rows']} rows, GPU {st['n_rows']}")
        cpu = {"value": round(r["n_rows"] / r["seconds_total"] / 1e6, 4), "unit": "Mrows/s", "cores": cores, "kind": "port",
               "sample": ("the same scan: the members of the chr1 tabix chunks" if sites else "the same scan: every member of the file")
                         + f" ({r['n_blocks']} BGZF members, {r['inflated_bytes'] / 1e9:.2f} GB text, {r['n_rows']} rows), every column"
                         + ("" if sites else " + the five list UDF results")
                         + f", {'libdeflate' if r['used_libdeflate'] else 'zlib'} inflate, {cores} threads (oracle/bioscan_oracle.c)",
               "seconds": round(r["seconds_total"], 3), "seconds_inflate": round(r["seconds_inflate"], 3)}
        if not sites:
            # the UDF checksums of the two implementations must agree
            assert int(r["avg_gq_valid"]) == us[0]["count_a"] and int(r["gq_gte_true"]) == us[2]["count_a"], (r, us)
            assert int(r["dp_gte_true"]) == us[3]["count_a"] and int(r["dp_lte_true"]) == us[4]["count_a"], (r, us)
    e2e = None
    if rank == 0 and world == 1 and not args.no_end_to_end:
        # file resident in HBM -> Arrow buffers in host memory through the chunk pipeline (VcfChunkStream); never part of `value`
        runs = [plan.execute_drain(0, args.batch_size) for _ in range(2)]
        r = runs[-1]
        if int(r["n_rows"]) != int(st["n_rows"]):
            raise SystemExit(f"end-to-end stream returned {r['n_rows']} rows, the device-resident scan {st['n_rows']}")
        e2e = {"Mrows_s": round(r["n_rows"] / r["seconds"] / 1e6, 4), "seconds": round(r["seconds"], 3),
               "ms_to_first_batch": round(r["seconds_to_first_batch"] * 1e3, 2), "n_batches": r["n_batches"],
               "first_run_seconds": round(runs[0]["seconds"], 3),
               "what": "bioscan_execute + bioscan_next until end of stream, every batch released at once; chunks of 2048 doubling to "
                       "%d BGZF members, HBM and host footprint O(chunk)" % int(os.environ.get("BIOSCAN_CHUNK_MEMBERS", 16384))}
    if rank == 0:
        per_step = elapsed / args.steps
        # algorithmic bytes of the text stage: decoded text read once by the delimiter index + once by the
        # extract kernels, Arrow buffers written once
        text_bytes = float(st["inflated_bytes"])
        alg = float(st["compressed_bytes"]) + 3 * text_bytes + float(st["arrow_bytes"])
        gpu_ms = st["ms_total_gpu"] + sum(u["ms_kernel"] for u in us)
        achieved = alg / (gpu_ms * 1e-3) / 1e9 if gpu_ms else 0.0
        out = {
            "metric": "vcf_bgz_tbi_region_scan_rows_per_sec" if sites else "vcf_multisample_genotypes_scan_rows_per_sec",
            "value": round(rows / per_step / 1e6, 4), "unit": "Mrows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(per_step * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": ("VCF.bgz + TBI region-predicate scan (chrom='chr1'), sites-only, gnomAD-like INFO" if sites else
                                    f"{args.samples}-sample VCF.bgz genotypes Struct<List> scan + list_avg/list_gte/list_lte"),
                       "n_lines_per_gpu": meta["n_lines"], "n_lines_chr1": meta["n_lines_chr1"], "n_samples": meta["n_samples"],
                       "n_blocks_file": meta["n_blocks"], "n_blocks_decoded": st["n_blocks"],
                       "compressed_bytes_per_gpu": meta["compressed_bytes"], "inflated_bytes_file": meta["inflated_bytes"],
                       "batch_size": args.batch_size},
            "decoded_GB_s": round(ubytes / per_step / 1e9, 3),
            "cells_per_sec_G": None if sites else round(rows * meta["n_samples"] * 3 / per_step / 1e9, 3),
            "stage_ms": {"inflate": round(st["ms_inflate"], 3), "crc32": round(st["ms_crc"], 3),
                         "delimiter_index+keys": round(st["ms_chain"], 3), "select": round(st["ms_select"], 3),
                         "extract": round(st["ms_extract"], 3), "udfs": [round(u["ms_kernel"], 3) for u in us]},
            "arrow_bytes": st["arrow_bytes"], "scan_wall_ms": round(st["ms_wall"], 3), "gpu_ms": round(gpu_ms, 3),
            "roofline": {"bound": "hbm", "kernel": "VCF pipeline (inflate + text kernels)", "achieved": round(achieved, 3),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None},
            "cpu_baseline": cpu,
            "end_to_end": e2e,
            "setup_s": {"generate": round(t_gen, 1)}}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def spawn_ranks(n, argv):
    """Start `n` ranks of this script under torch.distributed.run as a child process tree and return its exit status.
    Called before torch is imported: the parent never initialises the GPU."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=int(os.environ.get("BIOSCAN_BENCH_BLOCKS", DEFAULT_BLOCKS)))
    ap.add_argument("--batch-size", type=int, default=8192)
    ap.add_argument("--projection", default="*", help="'*' (12 core columns), 'chrom,start' or 'count'")
    ap.add_argument("--cpu-sample-blocks", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-stream (D2H-inclusive) leg")
    ap.add_argument("--keep-file", action="store_true")
    ap.add_argument("--lines", type=int, default=0, help="vcf-*: number of VCF lines (default 60 M sites / 200 k multi-sample)")
    ap.add_argument("--samples", type=int, default=1000, help="vcf-samples: number of samples")
    ap.add_argument("--format", default="bam", choices=["bam", "fastq", "vcf-sites", "vcf-samples"],
                    help="bam (default, BASELINE.json config 2) or fastq (BGZF-FASTQ + GZI full scan: the only scan the "
                         "reference publishes numbers for, openspec/.../design.md:29-36)")
    ap.add_argument("--partition-threads", type=int, default=0,
                    help="indexed mode: partitions of a rank executed concurrently from this many threads, the way DataFusion drives "
                         "execute(partition) from its worker pool; 1 = one after another, which keeps stage_ms and the roofline's launch "
                         "time free of overlap -- with more threads they are sums of overlapping stream times.  0 (default) = 1 at "
                         "N = 1, min(8, partitions of the rank) at N > 1")
    ap.add_argument("--mode", default=None, choices=["sequential", "indexed", "shards"],
                    help="sequential (default at N = 1): the rank scans its file as one partition. "
                         "indexed (default at N > 1; config 5 / SURVEY 8e): every rank opens the SAME file of --blocks x N members, the "
                         "BAI plan (target_partitions = 8 x ranks) is sharded in order across ranks "
                         "(bio-format-core/src/range_planning.rs:147-195) and each rank uploads and inflates only its partitions' "
                         "members -- per-GPU work stays that of config 2, so the scaling is weak. "
                         "shards: N > 1 with one independent file of --blocks members per rank (seed 42 + rank), each one partition")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (one per GPU, RCCL over xGMI for
        # the bench's own barrier / MAX / SUM) before this process has imported torch or made any HIP call -- a process that
        # has touched the GPU must never exec or fork into another GPU program -- relay their output (rank 0 prints the JSON
        # line) and exit with their status.
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if world > 1 and args.mode is None:
        args.mode = "indexed"   # ONE file, its BAI plan sharded in order over the ranks
    if args.mode is None:
        args.mode = "sequential"
    if args.partition_threads <= 0:
        args.partition_threads = 8 if (world > 1 and args.mode == "indexed") else 1

    import torch
    import torch.distributed as dist
    global COLL_DEVICE
    # BIOSCAN_BENCH_REHEARSE=1: walk the N > 1 control flow on a box with fewer GPUs than ranks (every rank on cuda:0,
    # gloo instead of RCCL for the bench's own barrier / MAX / SUM).  Not a measurement mode.
    rehearse = os.environ.get("BIOSCAN_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
        COLL_DEVICE = "cpu"
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as ge
    pkg = ge._load_pkg()
    pkg.load_library()
    if args.format == "fastq":
        return bench_fastq(args, pkg, rank, local_rank, world, torch, dist)
    if args.format.startswith("vcf"):
        return bench_vcf(args, pkg, rank, local_rank, world, torch, dist)

    # ---- synthetic input (deterministic in (blocks, seed)) ----
    synth = os.path.join(ROOT, "tools", "_build", "synth_bam")
    if not os.path.exists(synth):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools")])
    # indexed mode at N > 1 (config 5): ONE file of --blocks x N members, opened by every rank and COMPRESSED BY EVERY RANK: rank
    # k generates the k-th of N contiguous runs of the file's tiles (tools/synth_bam.c `part K N`, the file is a pure
    # function of (members, seed) tile by tile) with its share of the box's cores, then copies its run to its place in the
    # file (`place K N`; rank 0 adds header, EOF member and the merged BAI) -- r03 had rank 0 compress all N shares alone
    # (~36 s per share on a 16-core grant, seven ranks waiting in a broadcast).  Otherwise every rank writes its own file.
    # The file (~26 KB per member) goes to /dev/shm when it fits there with room to spare, else to /tmp.  If neither has the
    # space: at N = 1 the member count is reduced and reported as such; at N > 1 the bench REFUSES (a per-GPU size that
    # differs from the N = 1 line would make the scaling curve meaningless).
    shared = world > 1 and args.mode == "indexed"
    ncpu = os.cpu_count() or 1
    try:
        ncpu = min(ncpu, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    path, meta, t_gen = None, None, 0.0
    need = int(args.blocks * 27000 * 1.15) * max(1, world)
    shm = None
    if rank == 0 or not shared:
        for cand in ("/dev/shm", "/tmp"):
            try:
                if os.path.isdir(cand) and os.access(cand, os.W_OK):
                    vfs = os.statvfs(cand)
                    if vfs.f_bavail * vfs.f_frsize > need:
                        shm = cand
                        break
            except OSError:
                pass
        if shm is None:
            if world > 1:
                avail = {c: os.statvfs(c).f_bavail * os.statvfs(c).f_frsize for c in ("/dev/shm", "/tmp") if os.path.isdir(c)}
                raise SystemExit(f"bench.py --gpus {world}: the input needs {need / 1e9:.0f} GB of scratch space "
                                 f"({args.blocks} members per GPU) and neither /dev/shm nor /tmp has it ({avail}); "
                                 f"run with a smaller --blocks on every N of the curve instead of letting N > 1 shrink alone")
            shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
            vfs = os.statvfs(shm)
            fit = int(vfs.f_bavail * vfs.f_frsize * 0.8 / (27000 * 1.15))
            args.blocks = max(4096, min(args.blocks, fit))
    t0 = time.time()
    if shared:
        box = [os.path.join(shm, f"bioscan_synth_{os.getpid()}_shared.bam") if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, device=torch.device(COLL_DEVICE if COLL_DEVICE == "cpu" else f"cuda:{local_rank}"))
        path = box[0]
        gen_threads = max(1, ncpu // world)
        common = [synth, path, str(args.blocks * world), "42", str(gen_threads), "6"]
        subprocess.check_output(common + ["part", str(rank), str(world)])
        dist.barrier()          # every part and its sidecar exist
        out = subprocess.check_output(common + ["place", str(rank), str(world)]).decode()
        dist.barrier()          # the file is whole
        box = [json.loads(out) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, device=torch.device(COLL_DEVICE if COLL_DEVICE == "cpu" else f"cuda:{local_rank}"))
        meta = box[0]
        if rank == 0:
            for k in range(world):
                try:
                    os.unlink(f"{path}.part{k}.idx")
                except OSError:
                    pass
    else:
        path = os.path.join(shm, f"bioscan_synth_{os.getpid()}_r{rank}.bam")
        gen_threads = max(1, min(16, ncpu // max(1, world)))
        seed = 42 if args.mode == "indexed" else 42 + rank
        meta = json.loads(subprocess.check_output([synth, path, str(args.blocks), str(seed), str(gen_threads)]).decode())
    t_gen = time.time() - t0

    # ---- CPU baseline sample (rank 0, N == 1 only): read now, timed after the GPU steps ----
    cpu = None
    cpu_sample = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            nproc_box = len(os.sched_getaffinity(0))
        except AttributeError:
            nproc_box = ncpu
        # the all-cores leg wants >= 1024 members per thread
        sample_blocks = min(max(2 * args.cpu_sample_blocks, 1024 * nproc_box), meta["n_blocks"])
        nbytes = int(meta["compressed_bytes"] * min(1.0, (sample_blocks + 64) / meta["n_blocks"])) + (1 << 20)
        with open(path, "rb") as f:
            sample_bytes = f.read(nbytes)
        with open(path + ".bai", "rb") as f:
            cpu_sample = (sample_bytes, sample_blocks, f.read())
        del sample_bytes

    # ---- provider: load + make the compressed bytes resident in HBM (outside the timed region) ----
    t0 = time.time()
    prov = pkg.BamTableProvider(path, None, True, None, index_path=None if args.mode == "indexed" else "",
                                device_id=local_rank)
    if args.mode != "indexed":
        prov.make_resident()  # (indexed mode: each rank uploads only what its partitions inflate, below)
    t_load = time.time() - t0

    def unlink_input():
        if not args.keep_file:
            for p in (path, path + ".bai"):
                try:
                    os.unlink(p)
                except OSError:
                    pass
    if not shared:
        unlink_input()
    if args.projection == "*":
        projection = None
    elif args.projection == "count":
        projection = []
    else:
        names = prov.schema().names
        projection = [names.index(c) for c in args.projection.split(",")]
    if args.mode == "indexed":
        plan = prov.scan(projection=projection, target_partitions=8 * world)
        weights = [plan.partition_estimated_bytes(p) for p in range(plan.num_partitions())]
        # contiguous runs in plan order whose heaviest run is as light as possible (sharding.py: the reference's in-order rule lets a
        # one-byte difference in the estimates decide between 8 | 8 and 9 | 7 equal partitions, and the slowest rank is the bench's time)
        my_parts = pkg.shard_partitions_balanced(weights, world)[rank]
        t0 = time.time()
        plan.make_resident(my_parts)   # only the compressed byte range this rank's partitions cover
        t_load += time.time() - t0
        res_lo, res_hi = prov.resident_range(local_rank)
        if shared:
            dist.barrier()             # every rank has mapped the file and uploaded its range
            if rank == 0:
                unlink_input()
    else:
        plan = prov.scan(projection=projection, target_partitions=1)
        assert plan.num_partitions() == 1
        my_parts = [0]

    # indexed mode: the rank's partitions are executed the way DataFusion drives an ExecutionPlan -- execute(partition) from
    # several worker threads at once (the reference: one OS thread per partition, bio-format-core/src/sync_stream.rs:19-29);
    # every execute owns its stream and scratch, so the partitions' kernels interleave on the device
    pool = None
    if args.partition_threads > 1 and len(my_parts) > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=min(args.partition_threads, len(my_parts)))

    def run_step():
        tot = None
        if pool is not None:
            results = list(pool.map(lambda p: plan.execute_device(p, args.batch_size), my_parts))
        else:
            results = [plan.execute_device(p, args.batch_size) for p in my_parts]
        for st in results:
            if tot is None:
                tot = dict(st)
            else:
                for k in ("n_blocks", "compressed_bytes", "inflated_bytes", "arrow_bytes", "n_records", "n_rows", "ms_inflate",
                          "ms_chain", "ms_extract", "ms_crc", "ms_keys", "ms_select", "ms_wall"):
                    tot[k] += st[k]
                tot["chain_iterations"] = max(tot["chain_iterations"], st["chain_iterations"])
        return tot

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    stats = None
    for _ in range(args.warmup):
        stats = run_step()
    sync()
    t0 = time.perf_counter()
    infl_ms, chain_ms, extract_ms = [], [], []
    for _ in range(args.steps):
        stats = run_step()
        infl_ms.append(stats["ms_inflate"])
        chain_ms.append(stats["ms_chain"])
        extract_ms.append(stats["ms_extract"])
    sync()
    elapsed = time.perf_counter() - t0
    # full-size property: every record the generator wrote comes back (and K2 has checked the CRC32 of every member)
    if not shared and int(stats["n_rows"]) != int(meta["n_records"]):
        raise SystemExit(f"rank {rank}: the scan returned {stats['n_rows']} rows, the generator wrote {meta['n_records']} records")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=COLL_DEVICE)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([float(stats["n_rows"]), float(stats["inflated_bytes"]), float(stats["compressed_bytes"]),
                            float(stats["arrow_bytes"])], dtype=torch.float64, device=COLL_DEVICE)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        tot_rows, tot_u, tot_c, tot_a = [float(x) for x in cnt.tolist()]
        if shared and int(tot_rows) != int(meta["n_records"]):   # the ranks' runs of the plan together return every record once
            raise SystemExit(f"the {world} ranks returned {int(tot_rows)} rows, the generator wrote {meta['n_records']} records")
    else:
        tot_rows, tot_u, tot_c, tot_a = (float(stats["n_rows"]), float(stats["inflated_bytes"]),
                                         float(stats["compressed_bytes"]), float(stats["arrow_bytes"]))

    def k1_source_hash():
        import hashlib
        with open(os.path.join(ROOT, "datafusion-bio-formats_amd", "csrc", "inflate_v4.hip"), "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]

    def pmc_traffic_per_member():
        """HBM bytes per BGZF member of K1 from the committed rocprofv3 PMC passes (separate FETCH_SIZE / WRITE_SIZE runs
        of this bench at 65 536 members, values in KiB as reported; tools/profile_k1.sh).  The passes are offline: they
        are only used when profiles/rNN/k1_pmc_source.sha256 names the K1 source this library was built from, so the
        figure can never describe an older kernel."""
        import csv
        import glob
        stamps = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "k1_pmc_source.sha256")))
        if not stamps or open(stamps[-1]).read().split()[0] != k1_source_hash():
            return None
        d = os.path.dirname(stamps[-1])
        tot = 0.0
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            f = os.path.join(d, f"k1_pmc_{c}_65536blocks.csv")
            if not os.path.exists(f):
                return None
            rows = [r for r in csv.DictReader(open(f)) if K1_NAME in r["Kernel_Name"] and int(r["Grid_Size"]) > 6400]
            if not rows:
                return None
            tot += sum(float(r["Counter_Value"]) for r in rows) / len(rows) * 1024.0
        return tot / 65536.0

    # ---- end to end (rank 0 at N = 1): file resident in HBM -> Arrow buffers in host memory, through the chunk pipeline
    #      (D2H of chunk k overlapped with the kernels of chunk k + 1); never part of `value` ----
    e2e = None
    if rank == 0 and world == 1 and args.mode == "sequential" and not args.no_end_to_end:
        runs = [plan.execute_drain(0, args.batch_size) for _ in range(2)]  # the first run also pins the host blocks
        r = runs[-1]
        if int(r["n_rows"]) != int(meta["n_records"]):
            raise SystemExit(f"end-to-end stream returned {r['n_rows']} rows, the generator wrote {meta['n_records']} records")
        e2e = {"Mrec_s": round(r["n_rows"] / r["seconds"] / 1e6, 3), "seconds": round(r["seconds"], 3),
               "link_GB_s": round(float(stats["arrow_bytes"]) / r["seconds"] / 1e9, 3),
               "ms_to_first_batch": round(r["seconds_to_first_batch"] * 1e3, 2), "n_batches": r["n_batches"],
               "first_run_seconds": round(runs[0]["seconds"], 3),
               "projections": {},
               "what": "bioscan_execute + bioscan_next until end of stream, every batch released at once; compressed file resident in "
                       "HBM, chunks of 2048 doubling to %d BGZF members, Arrow buffers copied D2H into recycled pinned blocks "
                       "(BIOSCAN_HOST_POOL_GB=%s)" % (int(os.environ.get("BIOSCAN_CHUNK_MEMBERS", 16384)), os.environ.get("BIOSCAN_HOST_POOL_GB", "8")),
               "host_pool_gb": float(os.environ.get("BIOSCAN_HOST_POOL_GB", "8"))}
        # the same stream with room for every result block of the scan (r03's default cap): what a host with memory to spare gets
        if "BIOSCAN_HOST_POOL_GB" not in os.environ:
            os.environ["BIOSCAN_HOST_POOL_GB"] = "64"   # (the library reads the cap at every release)
            try:
                rb = [plan.execute_drain(0, args.batch_size) for _ in range(2)][-1]
                e2e["host_pool_64gb"] = {"Mrec_s": round(rb["n_rows"] / rb["seconds"] / 1e6, 3), "seconds": round(rb["seconds"], 3),
                                         "link_GB_s": round(float(stats["arrow_bytes"]) / rb["seconds"] / 1e9, 3)}
            finally:
                del os.environ["BIOSCAN_HOST_POOL_GB"]

    if e2e is not None and args.projection == "*":
        # the same stream for the narrow projections (what `SELECT chrom, start` and `COUNT(*)` consumers get)
        names = prov.schema().names
        for label, proj in (("chrom,start", [names.index("chrom"), names.index("start")]), ("count", [])):
            pl2 = prov.scan(projection=proj, target_partitions=1)
            r2 = [pl2.execute_drain(0, args.batch_size) for _ in range(2)][-1]
            if int(r2["n_rows"]) != int(meta["n_records"]):
                raise SystemExit(f"end-to-end stream ({label}) returned {r2['n_rows']} rows, the generator wrote {meta['n_records']} records")
            e2e["projections"][label] = {"Mrec_s": round(r2["n_rows"] / r2["seconds"] / 1e6, 3), "seconds": round(r2["seconds"], 3),
                                         "ms_to_first_batch": round(r2["seconds_to_first_batch"] * 1e3, 2)}

    if cpu_sample is not None:
        # The C oracle's STREAMING scan (oracle/bioscan_oracle.c: oracle_bam_scan_stream) on a bounded sample of the same file,
        # host cores of this box: one thread per partition, each inflating one member at a time and appending to its own
        # per-batch column builders -- the reference's executor shape (sync_stream.rs:19-29, physical_exec.rs:408-573).
        # Partition starts come from the file's BAI (record-aligned virtual offsets), as the reference's do.
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import c_oracle
        data, sample_blocks, bai = cpu_sample
        try:
            nproc = len(os.sched_getaffinity(0))  # the cores this process may use (what `nproc` prints)
        except AttributeError:
            nproc = ncpu
        cores = min(16, nproc)  # a one-GPU box's CPU share; the all-cores run of SURVEY 8(d) is threads_sweep[str(nproc)]
        # the CPU time the box actually grants (cgroup v2 cpu.max / v1 cfs quota): `nproc` may list cores the quota never schedules
        quota = None
        for qf, pf in (("/sys/fs/cgroup/cpu.max", None), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
            try:
                if pf is None:
                    a, b = open(qf).read().split()[:2]
                    quota = None if a == "max" else round(int(a) / int(b), 2)
                else:
                    a, b = int(open(qf).read()), int(open(pf).read())
                    quota = None if a <= 0 else round(a / b, 2)
                break
            except (OSError, ValueError):
                continue

        def leg(threads, blocks):
            blocks = min(blocks, sample_blocks)
            plan = c_oracle.stream_plan_bai(data, bai, threads, blocks)
            r = c_oracle.stream_scan(data, plan, True, args.batch_size)
            return r

        base = min(args.cpu_sample_blocks, sample_blocks)
        one = leg(1, max(1024, base // 4))
        main_leg = leg(cores, min(2 * base, sample_blocks))
        rate1 = one["n_rows"] / one["seconds_total"]

        def brief(r):
            rate = r["n_rows"] / r["seconds_total"]
            return {"Mrec_s": round(rate / 1e6, 3), "blocks": int(r["n_blocks"]), "seconds": round(r["seconds_total"], 3),
                    "seconds_inflate": round(r["seconds_inflate_avg"], 3), "seconds_build": round(r["seconds_build_avg"], 3),
                    "parallel_efficiency": round(rate / r["threads"] / rate1, 3)}
        sweep = {"1": brief(one)}
        for thr, blocks in ((8, base), (nproc, sample_blocks)):
            if thr not in (1, cores) and thr <= nproc:
                sweep[str(thr)] = brief(leg(thr, blocks))
        del data, cpu_sample
        st = main_leg
        cpu = {
            "value": round(st["n_rows"] / st["seconds_total"] / 1e6, 3), "unit": "Mrec/s", "cores": cores, "kind": "port",
            "nproc": nproc, "os_cpu_count": ncpu, "cgroup_cpu_quota_cores": quota,
            "sample": f"first {st['n_blocks']} BGZF blocks of the same file ({st['inflated_bytes'] / 1e9:.2f} GB inflated, "
                      f"{st['n_rows']} records), SELECT * core columns in batches of {args.batch_size}, "
                      f"{'libdeflate' if st['used_libdeflate'] else 'zlib'} inflate + CRC32, {cores} threads = {cores} BAI partitions, each streaming "
                      f"member by member into its own builders (oracle_bam_scan_stream; nproc of the box: {nproc}, its all-cores run is threads_sweep['{nproc}'])",
            "decoded_GB_s": round(st["inflated_bytes"] / st["seconds_total"] / 1e9, 3),
            "seconds": round(st["seconds_total"], 3),
            "seconds_inflate": round(st["seconds_inflate_avg"], 3), "seconds_build": round(st["seconds_build_avg"], 3),
            "seconds_note": "per-thread averages: inflate + CRC32 of the members / record decode into the batch builders; there is no serial phase",
            "parallel_efficiency": round(st["n_rows"] / st["seconds_total"] / cores / rate1, 3),
            "parallel_efficiency_note": "rate per thread relative to the 1-thread leg; a leg with more threads than cgroup_cpu_quota_cores "
                                        "(or than the physical cores behind nproc) time-slices and cannot scale -- per-thread seconds grow with the thread count there",
            "threads_sweep": sweep,
        }

    if rank == 0:
        per_step = elapsed / args.steps
        gb = meta["compressed_bytes"] / 1e9
        if world == 1:
            workload = "BGZF-BAM full-table scan (config 2), synthetic 150bp paired reads, seed 42" + \
                       (", as the BAI plan of 8 partitions" if args.mode == "indexed" else "")
        elif shared:
            workload = (f"config 5: ONE BGZF-BAM of {meta['n_blocks']} members ({gb:.1f} GB compressed; config-2 generator, seed 42, "
                        f"{args.blocks} members per GPU) scanned by {world} GPUs: the BAI plan of {8 * world} partitions sharded in "
                        "order over the ranks (range_planning.rs:147-195), every rank uploads and inflates only its partitions' members")
        else:
            workload = (f"independent shards: {world} GPUs x {meta['n_blocks']} BGZF members ({gb:.1f} GB compressed each, "
                        f"{gb * world:.1f} GB in total), config-2 generator with seed 42 + rank, one block-range shard per GPU")
        avg_infl = sum(infl_ms) / len(infl_ms)
        c, u = float(stats["compressed_bytes"]), float(stats["inflated_bytes"])
        achieved = (c + u) / (avg_infl * 1e-3) / 1e9  # GB/s, algorithmic bytes of K1 = C read + U written
        out = {
            "metric": "bgzf_bam_full_scan_records_per_sec", "value": round(tot_rows / per_step / 1e6, 3), "unit": "Mrec/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(per_step * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload,
                       "n_blocks_per_gpu": meta["n_blocks"] // (world if shared else 1),
                       "compressed_bytes_per_gpu": meta["compressed_bytes"] // (world if shared else 1),
                       "inflated_bytes_per_gpu": meta["inflated_bytes"] // (world if shared else 1),
                       "records_per_gpu": meta["n_records"] // (world if shared else 1),
                       **({"file_blocks": meta["n_blocks"], "file_compressed_bytes": meta["compressed_bytes"],
                           "file_records": meta["n_records"], "plan_partitions": 8 * world,
                           "rank0_partitions": len(my_parts), "rank0_resident_bytes": int(res_hi - res_lo)} if shared else {}),
                       "projection": args.projection, "batch_size": args.batch_size, "mode": args.mode, "deflate_level": meta["level"],
                       "deflater": meta["deflate"], "parallelism": (f"{world} ranks, contiguous runs of one BAI plan, no collective on the data path" if shared
                                       else f"{world} independent block-range shard(s), no collective"),
                       **({"partition_threads": args.partition_threads,
                           "stage_ms_note": "partitions run concurrently: stage_ms and roofline.avg_launch_ms are sums of overlapping stream times"}
                          if pool is not None else {})},
            "decoded_GB_s": round(tot_u / per_step / 1e9, 3),
            "pipeline_algorithmic_GB_s": round((tot_c + 2 * tot_u + tot_a) / per_step / 1e9, 3),
            "pipeline_hbm_frac": round((tot_c + 2 * tot_u + tot_a) / per_step / 1e9 / (HBM_PEAK_GBS * world), 5),
            "stage_ms": {"inflate": round(avg_infl, 3), "record_chain": round(sum(chain_ms) / len(chain_ms), 3),
                         "extract": round(sum(extract_ms) / len(extract_ms), 3), "crc32": round(stats["ms_crc"], 3),
                         "keys": round(stats["ms_keys"], 3), "select": round(stats["ms_select"], 3),
                         "wall_last_step": round(stats["ms_wall"], 3), "chain_iterations": stats["chain_iterations"]},
            "roofline": {"bound": "hbm", "kernel": K1_NAME, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": (int(pmc_traffic_per_member() * stats["n_blocks"]) if pmc_traffic_per_member() else None),
                         "traffic_source": "offline PMC: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, profiles/rNN/k1_pmc_*, "
                                           "stamped with the sha256 of the K1 source they were collected on; null when that is not the source of this build), "
                                           "per-member average x members of this launch; FETCH_SIZE as reported (K1 issues dword / 8-byte / unaligned 16-byte loads, "
                                           "widths the guide calls uncalibrated; K2 in the same profile confirms the x2 rule for 16 B/lane reads)",
                         "k1_source_sha256_16": k1_source_hash(),
                         "algorithmic_bytes_per_launch": int(c + u), "avg_launch_ms": round(avg_infl, 3)},
            "cpu_baseline": cpu,
            "end_to_end": e2e,
            "setup_s": {"generate": round(t_gen, 1), "load_and_h2d": round(t_load, 1)},
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
