#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X sDTW alignment stage on the BASELINE.json headline configuration.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1, driver's form)

One process per GPU.  Without WORLD_SIZE/RANK in the environment `--gpus N` (N > 1) makes this process a supervisor
that starts N fresh child ranks (sigfish_amd/launch.py) BEFORE anything touches the GPU and exits with their worst
code; a mismatch between --gpus and the ranks that actually exist is an error, never a silently smaller run.

Workload (config.workload = "ncov_r9_dna_q250", BASELINE.json configs[2]): synthetic R9 DNA reads (250 events,
5 % shorter) against the 29 903 b nCoV-2019 reference, both strands, -q 250.  A "step" is one pass of the hot
path (sfa_align_batch_device: plan + fill kernel + finalize + trace kernel + finalize) over one batch of --reads reads
per GPU whose query events are already resident in HBM; for N > 1 the rows of all timed steps are gathered to rank 0
over RCCL once, after the last step and inside the timed region.  Reads shard
across ranks with no data-path collective; the reference event model is broadcast once, before the timed region.

Scaling (the line's `scaling` key; `scaling_definitions` spells both out):
  weak    (default)          every rank gets --reads reads per step: total work grows with N
  strong  (--total-reads T)  T reads per step are split over the ranks by contiguous ranges (dist.shard_range), total work is
                             fixed: BASELINE.json configs[3] is `--workload r10_dna_1mb_q250 --total-reads 1000000 --gpus 8`
After the timed region, for N > 1, rank 0 checks that the rows it received from every rank are the rows that rank computed
(`gather_verified`; --no-check-gather skips it).

The JSON line also carries
  roofline      of the dominant kernel (sdtw_fill_kernel).  `bound` is "valu": the kernel never materialises the cost matrix,
                so VALU issue binds, not HBM.  `achieved`/`peak`/`frac` are lane-operations per second: DP
                cells/s x VALU instructions per cell (counted from the shipped ISA) against 256 CUs x 4 SIMDs x 32 lanes/clk x
                2.4 GHz.  The HBM view north_star asks for sits beside it as `roofline.hbm` (algorithmic bytes per launch /
                mean kernel duration, measured with HIP events on the kernel's stream, against 8 TB/s), with the counter
                traffic and `traffic_over_algorithmic`.
  cpu_baseline  sigfish's own CPU alignment stage (the reference sources compiled into oracle/_ref; kind "reference")
                timed on this box's host cores on a bounded sample of the same workload; falls back to our CPU
                restatement (kind "port") when that build is absent.
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_LANE_OPS = 256 * 4 * 32 * 2.4e9  # 256 CUs x 4 SIMD-32 x 2.4 GHz = 7.86e13 lane-ops/s


def host_cores():
    """CPU share of this process: cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def quote_profile(profiles_dir, workload, reads, opts, build_id):
    """(traffic bytes per fill launch | None, where that came from / why not, VALU issue figures | None) from the committed PMC
    passes: profiles/<tag>_meta.json names build id, workload, reads per step and options of the profile <tag>; only a profile
    of THIS build on THIS workload, batch size and options is quoted.  Tags are compared in natural order (r03_v10 after
    r03_v9, r10 after r9), never by modification time -- a checkout does not keep it."""
    import glob
    traffic, traffic_src, issue = None, None, None

    def natural(path):
        return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(path))]
    metas = []
    for mp in sorted(glob.glob(os.path.join(profiles_dir, "r*_meta.json")), key=natural):
        try:
            m = json.load(open(mp))
        except ValueError:
            continue
        if m.get("workload") == workload and int(m.get("reads", -1)) == reads and sorted(m.get("opts", [])) == sorted(opts):
            metas.append((mp, m))
    same = [(mp, m) for mp, m in metas if m.get("build_id") == build_id]
    if not metas:
        return None, f"not profiled: no profiles/*_meta.json for workload {workload}, {reads} reads, options {sorted(opts)}", None
    if not same:
        mp, m = metas[-1]
        return None, f"stale: {os.path.basename(mp)} was taken on build {m.get('build_id')}, this library is build {build_id}", None
    mp, m = same[-1]
    tag = os.path.basename(mp)[:-len("_meta.json")]
    pmc = os.path.join(profiles_dir, tag + "_pmc_summary.csv")
    if not os.path.exists(pmc):
        return None, f"profiles/{tag}_meta.json has no PMC summary next to it", None
    vals, durs = {}, {}
    for line in open(pmc).read().splitlines()[1:]:
        kname, counter, _, mean, dur = line.rsplit(",", 4)
        if "sdtw_fill_kernel" in kname:
            vals[counter] = vals.get(counter, 0.0) + float(mean)
            durs[counter] = float(dur)
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        # FETCH_SIZE x2 on gfx950 + WRITE_SIZE, KB -> bytes (MI355X_MICROARCH.md section HBM)
        traffic = round((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
        traffic_src = f"profiles/{tag}_pmc_summary.csv (separate rocprofv3 --pmc passes of build {m['build_id']}, same workload, batch size and options)"
    if "SQ_INSTS_VALU" in vals and "GRBM_GUI_ACTIVE" in vals:
        # wave-instructions per cycle and SIMD over the profiled launch (GRBM_GUI_ACTIVE is summed over the 8 XCDs), against
        # what this opcode mix reaches in isolation (tools/valu_ceiling.hip: 0.455 -- v_min3 is half rate), and the clock
        # the chip sustained meanwhile (power-limited; nominal 2.4 GHz)
        cycles = vals["GRBM_GUI_ACTIVE"] / 8
        rate = vals["SQ_INSTS_VALU"] / 1024 / cycles
        issue = {"issue_rate": round(rate, 4), "issue_ceiling": 0.455, "attainable_frac": round(rate / 0.455, 4),
                 "sustained_clock_ghz": round(cycles / (durs["GRBM_GUI_ACTIVE"] * 1e-3) / 1e9, 3),
                 "source": f"profiles/{tag}_pmc_summary.csv (SQ_INSTS_VALU, GRBM_GUI_ACTIVE); ceiling: tools/valu_ceiling.hip"}
    return traffic, traffic_src, issue


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU per step (default 100 000 at -q 250, same cell count otherwise)")
    ap.add_argument("--workload", default="ncov_r9_dna_q250")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--cpu-reads", type=int, default=None,
                    help="fixed size of the CPU baseline's sample instead of a time target (BASELINE.md section 3: 2048 / 64 / 1024 reads for configs 3 / 4 / 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-buffers", action="store_true",
                    help="also time sfa_align_batch with HOST query/result buffers (PCIe-inclusive; never `value`)")
    ap.add_argument("--opt", action="append", default=[], help="sfa_set_option key=value (tuning experiments)")
    ap.add_argument("--total-reads", type=int, default=None,
                    help="strong scaling: this many reads per step in total, split over the ranks by contiguous ranges (default: weak, --reads per rank)")
    ap.add_argument("--check-gather", action="store_true", help="(default for N > 1; kept for old command lines)")
    ap.add_argument("--no-check-gather", action="store_true",
                    help="skip the check after the timed region that rank 0 received exactly the rows every rank computed (checksums)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end leg (raw BLOW5 -> PAF through the command line)")
    ap.add_argument("--e2e-reads", type=int, default=400_000, help="reads in the generated BLOW5 files of the end-to-end leg")
    args = ap.parse_args()

    # ---- ranks: the launcher's, or our own (never a re-exec of a process that has touched the GPU) ----------------
    from sigfish_amd import launch  # pure Python, loads nothing native
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if not launch.launched_by_a_launcher() and args.gpus > 1:
        sys.exit(launch.spawn_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    launch_test = os.environ.get("SFA_BENCH_LAUNCH_TEST") == "1"  # tests/test_bench_launch.py: rendezvous only, gloo, no GPU

    import torch
    import torch.distributed as dist

    import sigfish_amd as S
    from sigfish_amd import dist as D
    from sigfish_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but {world} rank(s) exist (WORLD_SIZE={os.environ.get('WORLD_SIZE')}): refusing to "
                         f"report a {world}-rank number under an {args.gpus}-GPU label")
    if launch_test:  # the launch path on CPU: N ranks rendezvous over gloo and count themselves
        dist.init_process_group("gloo")
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        line = {"launch_test": True, "n_gpus": int(t.item()), "world": dist.get_world_size()}
        if args.total_reads is not None:  # strong scaling: every rank reports the range it would align
            mine = torch.tensor(D.shard_range(args.total_reads, rank, world), dtype=torch.int64)
            got = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(got, mine)
            line.update(scaling="strong", read_ranges=[[int(a), int(b)] for a, b in got])
        if rank == 0:
            print(json.dumps(line), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if torch.cuda.device_count() < (local_rank + 1 if world > 1 else 1):  # (device_count does not initialise the GPU here)
        raise SystemExit(f"rank {rank}: local GPU {local_rank} does not exist ({torch.cuda.device_count()} visible)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    torch.zeros(1, device=dev)
    # PyTorch's wheel bundles its own HIP runtime next to the system one the library links; they coexist in one process when
    # torch's is initialised FIRST.  Do not rely on import order: check it.
    assert torch.cuda.is_initialized(), "torch.cuda must be initialised before libsigfish_amd.so makes its first HIP call"
    force_dist = os.environ.get("SFA_DIST_FORCE") == "1"  # rehearse the RCCL path with one rank
    if force_dist and not launch.launched_by_a_launcher():  # a bare `python bench.py`: be our own one-rank launcher
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(launch.free_port()))
    if world > 1 or force_dist:
        dist.init_process_group("nccl", device_id=dev)

    # ---- reference event model: built on rank 0, broadcast over RCCL/xGMI, resident in every rank's HBM ----
    ref = flag = None
    if rank == 0:
        ref, flag, _, _, _ = synth.workload(args.workload, n_reads=8, seed=0)
    ref, flag = D.broadcast_ref(ref, flag, device=dev)
    al = S.Aligner(ref, flag, device=local_rank)
    for kv in args.opt:
        k, v = kv.split("=")
        al.set_option(k, int(v))

    # ---- this rank's shard of reads: synthetic, generated here, uploaded to HBM before the timed region ----
    qlen = int(re.search(r"_q(\d+)$", args.workload).group(1))  # every workload name ends in its -q value
    if args.total_reads is not None and args.reads is not None:
        raise SystemExit("--reads (per rank, weak scaling) and --total-reads (all ranks together, strong scaling) exclude each other")
    if args.total_reads is not None and args.total_reads < world:
        raise SystemExit(f"--total-reads {args.total_reads} leaves ranks without reads at {world} ranks")
    strong = args.total_reads is not None
    if strong:  # this rank's contiguous range of the job's reads
        lo, hi = D.shard_range(args.total_reads, rank, world)
        args.reads = hi - lo
    elif args.reads is None:
        args.reads = 100_000 * 250 // qlen
    q, q_off, _ = synth.make_reads(ref, args.reads, qlen=qlen, seed=1000 + rank)
    n = args.reads
    lens = q_off[1:] - q_off[:-1]
    strands = 1 if ref.reverse is None else 2
    cols = int(ref.ref_lengths.sum()) * strands
    cells = int(lens.sum()) * cols
    alg_bytes = int((4 * lens + 4 * cols + 32).sum())
    d_q = torch.from_numpy(q).to(dev)
    # rows of every timed step stay in HBM; ONE gather to rank 0 after the last step (the path has no other exchange)
    row_bytes = n * S.RESULT_DTYPE.itemsize
    n_slots = max(args.steps, 1)
    d_out = torch.zeros(n_slots * row_bytes, dtype=torch.uint8, device=dev)
    shard_reads = [n] * world if not strong else [D.shard_range(args.total_reads, r, world)[1] - D.shard_range(args.total_reads, r, world)[0] for r in range(world)]
    counts = [c * n_slots for c in shard_reads]
    torch.cuda.synchronize()

    fill_ms, trace_ms, launches = [], [], 0

    def step(i, record):
        nonlocal launches
        al.align_db_device(d_q.data_ptr(), q_off, n, d_out.data_ptr() + (i % n_slots) * row_bytes, sync=True)
        if record:
            p = al.profile()
            fill_ms.append(p["fill_ms"])
            trace_ms.append(p["trace_ms"])
            launches += p["fill_launches"]

    for i in range(args.warmup):
        step(i, False)
    if world > 1 or force_dist:
        D.gather_rows(d_out, counts)  # warm the gather path (connections are set up on first use)
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, True)
    gathered = None
    if world > 1 or force_dist:  # final gather of the result rows (24 B/read/step) to rank 0, in read order
        gathered = D.gather_rows(d_out, counts)
    torch.cuda.synchronize()
    if world > 1 or force_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1 or force_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    gather_verified = None
    # outside the timed region: rank r's slice on rank 0 against what rank r holds, and (strong scaling) the work every rank did
    if (world > 1 or force_dist or args.check_gather) and not args.no_check_gather:
        import zlib
        mine = torch.tensor([zlib.crc32(d_out.cpu().numpy().tobytes())], dtype=torch.int64, device=dev)
        sums = [mine]
        if world > 1 or force_dist:
            sums = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(sums, mine)
        if rank == 0:
            got = gathered.view(np.uint8) if gathered is not None else d_out.cpu().numpy()
            item = S.RESULT_DTYPE.itemsize
            cuts = np.concatenate([[0], np.cumsum(counts)]) * item
            gather_verified = [zlib.crc32(got[cuts[r]:cuts[r + 1]].tobytes()) for r in range(world)] == [int(t.item()) for t in sums]
    # cells and algorithmic bytes of the WHOLE job (ranks differ by a read under strong scaling, and by their reads' lengths)
    job = torch.tensor([cells, alg_bytes, n], dtype=torch.int64, device=dev)
    if world > 1 or force_dist:
        dist.all_reduce(job)
    job_cells, job_bytes, job_reads = (int(v) for v in job.tolist())

    # every rank leaves the process group here, rank 0 included: nothing below is collective, and rank 0's end-to-end leg runs
    # other processes for minutes -- no communicator should be waiting for peers that have exited meanwhile
    rccl_world = dist.get_world_size() if dist.is_initialized() else 1
    if world > 1 or force_dist:
        dist.destroy_process_group()
    if rank != 0:
        al.close()
        return  # (this rank's GPU is free from here on: rank 0's end-to-end leg below starts one process per GPU)

    total_reads = job_reads * args.steps
    value = total_reads / elapsed
    kern_s = (sum(fill_ms) / max(len(fill_ms), 1)) / 1e3  # mean fill time per step (all fill launches of a step)
    launches_per_step = max(launches // max(args.steps, 1), 1)
    achieved = alg_bytes / kern_s / 1e9
    cells_per_s_kernel = cells / kern_s
    # VALU instructions per DP cell: counted from the steady-state loop of the kernel this workload runs, in the ISA of the
    # library that is loaded (tools/isa_check.py; the Makefile leaves the figures next to the .so)
    std = bool(flag & S.DTW)
    isa_key = "std_fill" if std else ("fill32" if qlen > 256 else "headline_fill")  # (queries beyond 256 events: the 32-row kernels)
    ops_per_cell, ops_src = 49 / 16, "constant 49/16 (no ISA statistics next to the library)"
    try:
        isa = json.load(open(os.path.join(ROOT, "sigfish_amd", "lib", "libsigfish_amd.isa.json")))
        ops_per_cell = float(isa[isa_key]["valu_per_cell"])
        ops_src = f"sigfish_amd/lib/libsigfish_amd.isa.json: median steady-state loop of {isa_key} ({isa[isa_key]['kernel']})"
    except (OSError, KeyError, TypeError, ValueError):
        pass
    # HBM bytes per fill launch and the measured VALU issue rate: counters cannot be collected inside this run, so they are
    # quoted from the committed PMC passes of THIS build on this workload, batch size and options (quote_profile); otherwise
    # `traffic` is null and `traffic_source` says why.
    traffic, traffic_src, issue = quote_profile(os.path.join(ROOT, "profiles"), args.workload, n, args.opt, S.build_id())
    out = {
        "metric": ("reads/s (sDTW alignment stage: nCoV-2019 R9 DNA, -q 250, both strands)" if args.workload == "ncov_r9_dna_q250"
                   else f"reads/s (sDTW alignment stage: {args.workload})"),
        "value": round(value, 1),
        "unit": "reads/s",
        "n_gpus": world,
        "rccl_world_size": rccl_world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "scaling_definitions": {"weak": "every rank aligns --reads reads per step: total work grows with n_gpus (default)",
                                "strong": "--total-reads reads per step are split over the ranks by contiguous ranges: total work is fixed"},
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "library_build": S.build_id(),
        **({"gather_verified": gather_verified} if gather_verified is not None else {}),
        "config": {"workload": args.workload, "reads_per_gpu": n, **({"total_reads": args.total_reads} if strong else {}),
                   "reads_per_rank": shard_reads, "query_events": qlen, "ref_kmers": int(ref.ref_lengths.sum()),
                   "strands": strands, "sharding": f"reads x{world}", "cells_per_read_full": qlen * cols},
        "dp_cells_per_s": round(job_cells * args.steps / elapsed, 1),
        # per launch of the dominant kernel ON RANK 0 (n reads): the kernel is VALU-issue bound, the HBM view sits beside it
        "roofline": {
            "bound": "valu", "kernel": "sdtw_fill_kernel",
            "achieved": round(cells_per_s_kernel * ops_per_cell / 1e12, 4), "peak": round(VALU_LANE_OPS / 1e12, 4), "unit": "Tlane-op/s",
            "frac": round(cells_per_s_kernel * ops_per_cell / VALU_LANE_OPS, 4),
            "what": "DP cells/s of the kernel x VALU instructions per cell (shipped ISA) against 256 CUs x 4 SIMDs x 32 lanes/clk x 2.4 GHz; "
                    "`attainable_frac` is the measured issue rate against what this opcode mix reaches in isolation (v_min3 is half rate)",
            "cells_per_s": round(cells_per_s_kernel, 1), "ops_per_cell": round(ops_per_cell, 4), "ops_per_cell_source": ops_src,
            "peak_cells_per_s": round(VALU_LANE_OPS / ops_per_cell, 1),
            **(issue or {"attainable_frac": None}),
            "traffic": traffic, "traffic_source": traffic_src,
            "traffic_over_algorithmic": None if traffic is None else round(traffic * launches_per_step / alg_bytes, 4),
            "algorithmic_bytes_per_step": alg_bytes, "fill_launches_per_step": launches_per_step,
            "kernel_ms_per_step": round(kern_s * 1e3, 3),
            "trace_kernel_ms_per_step": round(sum(trace_ms) / max(len(trace_ms), 1), 3),
            "hbm": {"achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 6),
                    "what": "algorithmic bytes per launch (SURVEY.md 8d: 4 B per query event + 4 B per reference level and strand + 32 B per read) "
                            "/ mean kernel duration"},
        },
    }

    if args.host_buffers and world == 1:
        al.align_db(q, q_off)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            al.align_db(q, q_off)
        out["pcie_inclusive_reads_per_s"] = round(n * args.steps / (time.perf_counter() - t1), 1)

    # ---- CPU baseline on a bounded sample of this rank's reads, all host cores of this box ---------------------
    # kind "reference": the reference's OWN align_db (dtw_single over work_db), compiled from its sources into
    # oracle/_ref (travels with the repo); kind "port": our CPU restatement, when that build is not there.
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O  # checker / baseline only
        cores = host_cores()
        oref = O.RefSynth(ref.names, ref.seq_lengths, ref.ref_lengths, ref.st_offset, ref.forward, ref.reverse)
        use_ref = os.path.exists(O.REF_BENCH)

        def cpu_run(m):
            if use_ref:
                return O.reference_align_batch(q, q_off[:m + 1], oref, flag, threads=cores)
            t1 = time.perf_counter()
            rows = O.align_batch(q, q_off[:m + 1], oref, flag, threads=cores)
            return rows, time.perf_counter() - t1

        if args.cpu_reads is not None:
            sample = max(1, min(n, args.cpu_reads))
        else:
            pilot = min(n, cores * 4)
            _, pdt = cpu_run(pilot)
            sample = int(max(pilot, min(n, pilot / pdt * args.cpu_seconds)))
        want, dt = cpu_run(sample)
        last = ((args.steps - 1) % n_slots) * row_bytes
        got = np.frombuffer(d_out[last:last + row_bytes].cpu().numpy().tobytes(), dtype=S.RESULT_DTYPE)[:sample]
        out["cpu_baseline"] = {"value": round(sample / dt, 2), "unit": "reads/s", "cores": cores,
                               "kind": "reference" if use_ref else "port",
                               "sample": f"first {sample} reads of the same batch, {dt:.1f} s wall in the alignment stage, "
                                         f"{int(lens[:sample].sum()) * cols / dt:.3e} cells/s",
                               "parity_on_sample": bool(got.tobytes() == want.tobytes())}
        out["speedup_vs_cpu_baseline"] = round(value / (sample / dt), 1)
    al.close()
    # ---- end to end (SURVEY.md 8d "plus end-to-end wall"): raw BLOW5 -> PAF through `sigfish-amd dtw`, whole process, on files
    # generated here; extra keys, never `value`.  After al.close(): the command line wants the GPU to itself.
    if world == 1 and not args.no_e2e and args.workload == "ncov_r9_dna_q250":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import e2e_bench
        try:
            out["end_to_end"] = e2e_bench.measure(reads=args.e2e_reads, threads=min(host_cores(), 16))
        except Exception as e:  # the headline line must not depend on scratch space for multi-GB files
            out["end_to_end"] = {"error": str(e)[:300]}
    # ... and for N > 1 the same command line as it is meant to run on a node: one process per GPU (`--ranks N --device 0..N-1`),
    # against one process on one GPU, same file, outputs compared.  The other ranks of this bench have released their GPUs.
    if world > 1 and not args.no_e2e and args.workload == "ncov_r9_dna_q250":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import e2e_bench
        try:
            time.sleep(2.0)  # the other ranks are leaving
            # this process still holds a context on GPU 0: beyond four GPUs the command line gets the OTHER GPUs, so that no more
            # processes use the node's GPUs at once than this bench itself did (rank 0 + one per remaining GPU)
            devs = list(range(world)) if world <= 4 else list(range(1, world))
            out["end_to_end"] = e2e_bench.measure_sharded(len(devs), devs, threads=min(host_cores(), 16 * len(devs)), reads=4 * args.e2e_reads)
        except Exception as e:
            out["end_to_end"] = {"error": str(e)[:300]}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
