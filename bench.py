#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X KMP packet-payload matcher.

Workload (BASELINE.json configs[1], per GPU): 1 000 000 synthetic UDP payloads of 1500 B
(bytes uniform over a..z, NUL-free), one 16-byte pattern planted in ~10 % of the packets;
generator: include/kmp_synth.h, seed 1234.  A "step" is one full pass of the hot path over that
arena (scan kernel + partial-count reduce, then the cross-GPU count sum), inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- HBM roofline of the scan kernel: algorithmic bytes (sum of payload lengths of
                  one launch) / average launch duration measured with HIP events on the launch
                  stream, against 8.0 TB/s.
  cpu_baseline -- the CPU restatement of openmp_data.c:126-178 (oracle/, "port") timed on the
                  host cores on the same arena (rank 0, N == 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
NEEDLE = b"NEEDLE_16B_PATRN"
PAYLOAD_LEN = 1500
SEED = 1234


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def host_cores() -> int:
    """CPU threads this process may really use: the scheduler affinity capped by the cgroup CPU
    quota (a GPU box exposes all 256 logical CPUs but grants a 16-CPU share)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, -(-int(txt[0]) // int(txt[1]))))
            else:
                q = int(txt[0])
                p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, -(-q // p)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle", type=int, default=400, help="untimed passes before the warm-up so that the GPU's clocks have "
                    "settled (a 250 us kernel needs ~100 ms of load to leave the DVFS transient; profiles/r01_sustained.txt)")
    ap.add_argument("--packets-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--depth", type=int, default=0, help="chunk loads in flight per wavefront (0 = library default)")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reps", type=int, default=3)
    ap.add_argument("--force-dist", action="store_true", help="rehearsal: create the process group and run the all-reduce path "
                    "even with one rank (checks the RCCL plumbing on a 1-GPU box)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank code path on a box with fewer GPUs than ranks)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import multithreading_string_matching_amd as K
    from multithreading_string_matching_amd import dist as kd
    from multithreading_string_matching_amd.matcher import OPT_BLOCKS_PER_CU, OPT_DEPTH, GpuMatcher

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(ndev, 1)
    if dev_index >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    n = args.packets_per_gpu
    first_id = rank * n                      # every rank scans its own shard of the global stream
    sp = K.SynthParams.make(seed=SEED, needle=NEEDLE, plant_permille=100)

    # Everything runs on one explicit (non-default) torch stream: the scan kernels are enqueued on it
    # through the C-ABI, and torch.distributed orders its collectives after the CURRENT stream -- so
    # the all-reduce of step i waits for step i's counts and overlaps step i+1's scan.
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    m = GpuMatcher(dev_index)
    m.set_stream(stream.cuda_stream)
    if args.depth:
        m.set_option(OPT_DEPTH, args.depth)
    if args.blocks_per_cu:
        m.set_option(OPT_BLOCKS_PER_CU, args.blocks_per_cu)

    # ---- synthetic arena, generated on the device ------------------------------------------
    stride = (PAYLOAD_LEN + 15) // 16 * 16
    nbytes = n * stride + 64
    d_arena = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    d_arena[n * stride:].zero_()
    d_off = torch.empty(n, dtype=torch.int64, device=dev)
    d_len = torch.empty(n, dtype=torch.int32, device=dev)
    m.fixed_index(d_off, d_len, PAYLOAD_LEN, 16)
    m.synth_fill(d_arena, d_off, d_len, sp, first_pkt_id=first_id)
    m.set_patterns([NEEDLE])
    m.attach_arena(d_arena, d_off, d_len)
    n_chk, payload_bytes = m.arena_info()
    assert (n_chk, payload_bytes) == (n, n * PAYLOAD_LEN)

    # ---- correctness before timing ------------------------------------------------------------
    planted = K.synth_count_planted(sp, n, PAYLOAD_LEN, first_pkt_id=first_id)
    got, _ = m.scan()
    if int(got[0]) != planted:
        raise SystemExit(f"rank {rank}: GPU count {int(got[0])} != planted {planted}")

    # ---- timed region -----------------------------------------------------------------------------
    # The only exchange of the path is the SUM of the per-pattern counters (mpi_dumping.c:202).  Step i leaves
    # its counter in slot i % G of one of two device buffers; a full buffer is all-reduced in ONE collective
    # (G x 8 bytes) while the next G steps fill the other buffer: fewer, larger collectives, none of them on
    # the scan stream's critical path.
    G = 4
    bufs = [torch.zeros(G, dtype=torch.int64, device=dev) for _ in range(2)]
    works = [None, None]

    def step(i):
        b, j = (i // G) % 2, i % G
        if j == 0 and works[b] is not None:
            works[b].wait()                    # this buffer's previous all-reduce has finished (stream-side wait)
            works[b] = None
        m.scan_enqueue(bufs[b][j:j + 1])       # scan kernel + partial reduce -> the slot (device memory)
        if use_dist and j == G - 1:
            works[b] = dist.all_reduce(bufs[b], op=dist.ReduceOp.SUM, async_op=True)

    def drain(n_steps):
        """Reduce the buffer the last steps left partly filled, then wait for everything."""
        if use_dist and n_steps % G:
            b = (n_steps // G) % 2
            works[b] = dist.all_reduce(bufs[b], op=dist.ReduceOp.SUM, async_op=True)
        for b, w in enumerate(works):
            if w is not None:
                w.wait()
                works[b] = None

    for i in range(args.settle):               # device settle-in, not part of W or K
        m.scan_enqueue(bufs[0][0:1])
    torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    drain(args.warmup)
    kd.barrier()
    torch.cuda.synchronize()
    m.profile_begin(args.steps)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain(args.steps)
    torch.cuda.synchronize()
    kd.barrier()
    t1 = time.perf_counter()
    launch_ms = m.profile_end(args.steps)
    elapsed = kd.max_over_ranks(t1 - t0, device=dev)
    avg_launch_ms = kd.max_over_ranks(float(launch_ms.mean()) if len(launch_ms) else 0.0, device=dev)

    last = args.steps - 1
    total = int(bufs[(last // G) % 2][last % G].item())
    planted_all = planted
    if use_dist:
        t = torch.tensor([planted], dtype=torch.int64, device=dev)
        kd.reduce_counts(t)
        planted_all = int(t.item())
    if total != planted_all:
        raise SystemExit(f"rank {rank}: reduced count {total} != planted {planted_all}")

    # ---- CPU baseline + sample parity (rank 0, N == 1) -------------------------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        o = oracle.load()
        host = d_arena.cpu().numpy()
        off = d_off.cpu().numpy().astype(np.uint64)
        ln = d_len.cpu().numpy().astype(np.uint32)
        cores = host_cores()
        best = None
        for _ in range(max(1, args.cpu_reps)):
            counts, dt = o.count(host, off, ln, [NEEDLE], threads=cores)
            if int(counts[0]) != planted:
                raise SystemExit(f"oracle count {int(counts[0])} != GPU/planted {planted}")
            best = dt if best is None else min(best, dt)
        # one-thread figure (the serial.c loop, serial.c:153-155) on the first 100 000 packets
        n1 = min(n, 100_000)
        t1 = time.perf_counter()
        c1, _ = o.count(host, off[:n1], ln[:n1], [NEEDLE], threads=0)
        t1 = time.perf_counter() - t1
        if int(c1[0]) != K.synth_count_planted(sp, n1, PAYLOAD_LEN, first_pkt_id=first_id):
            raise SystemExit("oracle serial count differs from the planted count")
        cpu = {
            "value": round(payload_bytes / best / 1e9, 3), "unit": "GB/s", "cores": cores, "kind": "port",
            "serial_1thread_GBps": round(n1 * PAYLOAD_LEN / t1 / 1e9, 3),
            "sample": f"the full per-GPU workload ({n} x {PAYLOAD_LEN} B, 1 pattern), best of {max(1, args.cpu_reps)} passes, "
                      f"openmp_data.c:126-178 bracket, {best:.3f} s per pass",
            "matches_per_s": round(planted / best, 1),
        }
        # Where the reference's own object code travelled with the repo (oracle/_ref, built from /root/reference in the
        # build container): its kmp_matcher (serial.c:190-215), one call per payload, calls spread over the same threads.
        # That is the baseline proper ("reference"); the port's figure stays beside it.
        try:
            ref = oracle.load_ref()
        except Exception:
            ref = None
        if ref is not None and getattr(ref, "has_driver", False):
            rbest = None
            for _ in range(max(1, args.cpu_reps)):
                rc, rdt = ref.count_arena(host, off, ln, NEEDLE, cores)
                if rc != planted:
                    raise SystemExit(f"reference kmp_matcher count {rc} != GPU/planted {planted}")
                rbest = rdt if rbest is None else min(rbest, rdt)
            cpu["port_GBps"] = cpu["value"]
            cpu.update({"value": round(payload_bytes / rbest / 1e9, 3), "kind": "reference",
                        "matches_per_s": round(planted / rbest, 1),
                        "sample": f"the full per-GPU workload ({n} x {PAYLOAD_LEN} B, 1 pattern), best of {max(1, args.cpu_reps)} passes of "
                                  f"the reference's own kmp_matcher object code (serial.c:190-215) over the arena, one call per payload, "
                                  f"{cores} OpenMP threads (guided, as openmp_data.c:157-175), {rbest:.3f} s per pass; the port "
                                  f"(oracle/kmp_oracle.c, openmp_data.c:126-178 bracket) takes {best:.3f} s"})
        del host

    if rank == 0:
        bytes_all = payload_bytes * world * args.steps
        value = bytes_all / elapsed / 1e9
        achieved = payload_bytes / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        # HBM bytes per launch from the PMC counters (FETCH_SIZE x 2 on gfx950, own rocprofv3 pass,
        # profiles/roofline_traffic.json); only valid for the workload it was collected on
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.isfile(tpath) and n == 1_000_000:
            with open(tpath) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        out = {
            "metric": "payload GB/s scanned + matches/s, 1M x 1500B pkts, 1 pattern",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {
                "workload": f"{n} synthetic {PAYLOAD_LEN} B UDP payloads per GPU (a..z, NUL-free, seed {SEED}), "
                            f"one 16-byte pattern planted in ~10% of packets; BASELINE configs[1]",
                "packets_per_gpu": n, "payload_len": PAYLOAD_LEN, "patterns": 1, "pattern_len": len(NEEDLE),
                "parallelism": f"packets sharded over {world} GPU(s), all-reduce(SUM) of counts",
            },
            "matches_per_s": round(total * args.steps / elapsed, 1),
            "matches_per_pass": total,
            "pct_hbm_peak_per_gpu": round(100.0 * value / world / HBM_PEAK_GBS, 2),
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": "kmp_scan_flat_kernel", "launch_ms_avg": round(avg_launch_ms, 5),
                "algorithmic_bytes_per_launch": payload_bytes,
            },
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)

    m.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
