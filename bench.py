#!/usr/bin/env python3
"""bench.py -- pose-energy evaluations per second of the DFIRE hot path on 1k4c (MI355X).

A "step" is one pass of the hot path over one batch: ONE launch of the batched pose-energy
kernel (plus its tiny tail kernel) over --batch poses that already live in HBM.  The workload
is BASELINE.json's metric configuration: the 1k4c membrane system (3413 receptor atoms incl.
453 membrane beads x 3268 ligand atoms, 11 153 684 atom pairs per pose), DFIRE scoring with the
synthetic DCparams (the real table is not in the reference mount), poses = the example's 200
starting poses replicated with seeded 0.25 A translation jitter.

    python bench.py --gpus 1 --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Rank 0 prints ONE JSON line.  `value` = poses evaluated by all ranks / max-over-ranks time.
`roofline` prices the pair kernel against the HBM roof with ALGORITHMIC bytes:
bytes/pose = 26*(N_rec+N_lig) + 8*P_cut + 64, P_cut counted on the GPU for the actual batch;
kernel time from HIP events on the launch stream.  `cpu_baseline` times the CPU oracle
(oracle/, a loop-for-loop C port of the Rust reference, which cannot be built here) on this
box's host cores over a bounded sample of the same poses.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def load_case(pkg, workload):
    g = os.path.join(ROOT, "tests", "golden", workload)
    if workload == "1k4c":
        return dict(method="dfire", rec=os.path.join(g, "lightdock_receptor_membrane.pdb"),
                    lig=os.path.join(g, "lightdock_ligand.pdb"), kw={}, pos=os.path.join(g, "initial_positions_0.dat"))
    if workload == "1ppe":
        return dict(method="dfire", rec=os.path.join(g, "lightdock_1ppe_e.pdb"), lig=os.path.join(g, "lightdock_1ppe_i.pdb"),
                    kw=dict(rec_active=["E.ILE.16"]), pos=os.path.join(g, "initial_positions_0.dat"))
    raise SystemExit("unknown workload " + workload)


def read_positions(path):
    return np.array([[float(v) for v in line.split(" ")] for line in open(path).read().splitlines()])[:, :7]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(case, table, poses, budget_s, threads):
    """Time the CPU oracle (test infrastructure used ONLY as the reported baseline) over a bounded
    sample of the bench poses: (a) one thread, the stand-in for the single-threaded Rust path;
    (b) `threads` host threads, one pose stream each, the stand-in for `ant_thony.py --cores N`."""
    orc = ge.oracle()
    scorer = orc.Scorer(case["method"], case["rec"], case["lig"], potential=table, **case["kw"])
    scorer.energy_row(poses[0])                                   # warm
    n1 = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < min(2.0, budget_s / 8) and n1 < len(poses):
        scorer.energy_row(poses[n1])
        n1 += 1
    single = n1 / (time.perf_counter() - t0)
    n = int(max(threads, min(len(poses), (budget_s * 7 / 8) * single)))
    n -= n % threads
    sample = poses[:n]
    out = np.zeros(n)

    def work(k):
        for i in range(k, n, threads):
            out[i] = scorer.energy_row(sample[i])       # ctypes releases the GIL

    ths = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "evals/s", "cores": threads, "kind": "port", "single_thread_value": single,
            "cpu_model": cpu_model(),
            "sample": "%d of the bench poses on %d threads in %.1f s wall (+ %d poses on 1 thread); C oracle -O2, f64, "
                      "no SIMD intrinsics" % (n, threads, dt, n1)}, out


def measured_traffic(args, info):
    """HBM bytes per pair-kernel launch from the rocprofv3 PMC passes committed under profiles/
    (FETCH_SIZE / WRITE_SIZE collected in separate runs of this same command; see
    profiles/README.md for the unit and gfx950 corrections).  None when no profile matches the
    workload, batch and kernel of this run."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(path))
    except (OSError, ValueError):
        return None
    key = "%s:%d:%s" % (args.workload, args.batch, info["pair_kernel_name"])
    return t.get(key, {}).get("hbm_bytes_per_launch")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8192, help="poses per GPU per step")
    ap.add_argument("--workload", default="1k4c")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="CPU-baseline budget in core-seconds (0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for dry runs)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("LD_BENCH_FORCE_DEVICE") is not None:      # dry runs of the N > 1 path on a 1-GPU box
        local = int(os.environ["LD_BENCH_FORCE_DEVICE"])
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the pose-energy path has no CPU fallback")
    if local >= torch.cuda.device_count():      # ranks that each see only their own GPU (HIP_VISIBLE_DEVICES per rank)
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)

    if rank == 0:
        ge.ensure_built()
    if dist is not None:
        dist.barrier()
    pkg = ge.package()
    pkg.init(local)
    case = load_case(pkg, args.workload)
    table = pkg.synth.dcparams()
    scorer = pkg.Scorer.from_pdb(case["method"], case["rec"], case["lig"], potential=table, **case["kw"])
    info = scorer.kernel_info()
    base = read_positions(case["pos"])
    # swarms shard across ranks with no exchange: every rank gets its own, differently seeded batch
    poses = pkg.synth.jitter(base, args.batch, seed=1000 + rank)

    dev = torch.device("cuda", local)
    d_poses = torch.from_numpy(poses).to(dev)
    d_out = torch.empty(args.batch, dtype=torch.float64, device=dev)
    d_cnt = torch.zeros(args.batch, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    scorer.set_stream(stream.cuda_stream)

    def step(counts=False):
        scorer.energy_batch_device(args.batch, d_poses.data_ptr(), poses.shape[1], d_out.data_ptr(), None,
                                   d_cnt.data_ptr() if counts else None)

    # P_cut of this batch (counting variant of the kernel, outside the timed region)
    step(counts=True)
    torch.cuda.synchronize()
    p_cut = d_cnt.cpu().numpy().astype(np.int64)
    try:
        blocks = float(scorer.last_block_counts(args.batch).mean())     # 8x8 blocks the box culling let through
    except pkg.LightdockError:
        blocks = None
    algo_bytes_launch = float(info["stream_bytes_per_pose"] * args.batch + 8 * p_cut.sum())

    multi = pkg.multi
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    scorer.enable_timing(True)
    scorer.pair_kernel_time()          # reset

    def timed():
        for _ in range(args.steps):
            step()

    # barrier + synchronize on both sides, MAX over ranks (multi.timed_region)
    elapsed = multi.timed_region(timed, dist, sync=torch.cuda.synchronize)
    kern_ms, launches = scorer.pair_kernel_time()
    scorer.enable_timing(False)
    energies = d_out.cpu().numpy()

    if rank == 0:
        total = args.batch * args.steps * world
        kern_s = kern_ms / 1e3 / max(launches, 1)
        achieved = algo_bytes_launch / kern_s / 1e9
        out = {
            "metric": "pose-energy evals/sec (DFIRE, %s)" % args.workload,
            "value": total / elapsed, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s DFIRE pose-energy batch, %d poses/GPU/step, %d x %d atoms, synthetic DCparams"
                                   % (args.workload, args.batch, scorer.num_atoms(0), scorer.num_atoms(1)),
                       "poses_per_step_per_gpu": args.batch, "pair_tests_per_pose": info["pair_tests_per_pose"],
                       "mean_pairs_in_cutoff": float(p_cut.mean()), "mean_8x8_blocks_evaluated": blocks, "parallelism": "swarm-sharded x%d, no collectives" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args, info),
                         "kernel": info["pair_kernel_name"], "kernel_ms": 1e3 * kern_s,
                         "algorithmic_bytes_per_launch": algo_bytes_launch,
                         "pair_tests_per_s": info["pair_tests_per_pose"] * args.batch / kern_s},
        }
        if args.cpu_seconds > 0 and world == 1:      # reported baseline: rank 0 at N = 1 only
            threads = min(len(os.sched_getaffinity(0)), 64)
            cb, cpu_e = cpu_baseline(case, table, poses, args.cpu_seconds, threads)
            out["cpu_baseline"] = cb
            n = len(cpu_e)
            rel = float(np.max(np.abs(energies[:n] - cpu_e) / np.maximum(np.abs(cpu_e), 1e-9)))
            out["parity_max_rel_err_vs_cpu_sample"] = rel
            if rel > 1e-4:
                raise SystemExit("parity violated: %g" % rel)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
