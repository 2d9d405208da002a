#!/usr/bin/env python3
"""bench.py -- pose-energy evaluations per second of the GSO + DFIRE/DNA hot path (MI355X).

    python bench.py --gpus 1 --steps K --warmup W [--workload W]     (N > 1: launched by torch.distributed.run)

Rank 0 prints ONE JSON line.  `value` = pose evaluations by all ranks / max-over-ranks wall time of
exactly K steps between barrier + synchronize on both sides.  Workloads (a "step" = one pass of
the hot path over one batch that already lives in HBM):

  1k4c (default)  BASELINE.json's metric configuration: ONE launch of the batched DFIRE pose-energy
                  kernel (+ its tail kernel) over --batch poses of the 1k4c membrane system (3413 x
                  3268 atoms, 11 153 684 atom pairs per pose); poses = the example's 200 starting
                  poses replicated with seeded 0.25 A translation jitter; every rank its own batch
                  (weak scaling).
  1ppe            the same for the 1ppe system (config 2).
  1azp-dna        DNA scoring + receptor/ligand ANM on the 1azp system (config 4), same shape.
  2uuy            DFIRE + receptor/ligand ANM (10 + 10 modes) on the 2uuy system: the block-major path's ANM form
                  (`dfire_bm_pairs<., true>`; LIGHTDOCK_BM_ANM=0 runs the pose-major kernel instead).
  gso-1ppe        config 5 as written: --swarms (default 1024) x 200 glowworms of 1ppe DFIRE SHARDED
                  over the ranks, a step = one GSO step of every swarm (flag memset, K1 over the
                  glowworms that moved, tail, K2); total work fixed (strong scaling).
  gso-1k4c        the headline system inside the GSO loop, --swarms (default 64) per rank (weak).

DFIRE uses the synthetic DCparams (the real table is not in the reference mount).
`roofline` reports what BINDS the pair kernel: these kernels are lookup / reduction loops whose table reads never leave
the CU (LDS) or the L2, so the headline roof is the vector issue rate -- wave-level vector instructions of a launch
(rocprofv3 SQ_INSTS_VALU of this same command, committed under profiles/ and tied to the kernel sources by a hash) times
the 4.5 cycles most of them take on gfx950, over the kernel time measured live with HIP events on the launch stream --
with the LDS pipe's busy share beside it (`roofline.lds`).  `roofline.hbm_model` keeps SURVEY 8d's byte model
(DFIRE 26*(N_rec+N_lig) + 8*P_cut + 64 per pose with P_cut counted on the GPU for the actual batch; DNA
48*(N_rec+N_lig) + 240 per ANM-deformed atom + 64) against the 8 TB/s HBM roof: a bookkeeping figure, because the
8*P_cut table bytes are served from LDS; it is flagged `model_exceeded` when it passes 1.  `roofline.traffic` is the HBM
traffic the counters saw.  `cpu_baseline` times the CPU oracle (oracle/, a loop-for-loop C port of the Rust reference,
which cannot be built here) on this box's host cores, pthreads inside the library, over a bounded sample of the same poses.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0                      # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
VALU_LANE_OPS_PER_S = 256 * 4 * 16 * 2.4e9  # 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz: one vector op per lane per cycle
L2_PEAK_GBS = 34500.0                      # aggregate L2 bandwidth, same guide ("L2 (per XCD)": ~34.5 TB/s)
L2_REQUEST_BYTES = 128                     # one TCP_TCC_READ_REQ = one 128-byte line (tools/microbench/l2_request_size.py)
L2_GATHER_CEILING_GBS = 33300.0            # what the L2s deliver to a dense gather of random 128-byte lines of an L2-resident table;
                                           # 25 300 when the lines are spread evenly over 5.5 MB (profiles/r02_l2_gather_ceiling.txt)
GOLD = os.path.join(ROOT, "tests", "golden")


def valu_issue_cost():
    """Time one wave-level vector instruction of the kind these kernels run (packed f32, conversions, 64-bit integer adds)
    occupies a SIMD: measured by tools/microbench/valu_rate.hip (8 waves per SIMD, independent copies of v_cvt_u32_f32) and
    committed as profiles/valu_issue.json together with the shader clock the SIMDs HELD during that loop (s_memtime over
    s_memrealtime), i.e. the figure in cycles is a measurement, not 'ns x a nominal 2.4 GHz'.  The roof is priced in time."""
    try:
        v = json.load(open(os.path.join(ROOT, "profiles", "valu_issue.json")))
        return float(v["ns_per_inst"]), v
    except (OSError, ValueError, KeyError):
        return 1.86, {"ns_per_inst": 1.86, "source": "profiles/r02_valu_issue_rates.txt (no clock stamp: 4.5 cycles only if the SIMDs hold 2.4 GHz)"}


def load_case(name):
    g = os.path.join(GOLD, name)
    if name == "1k4c":
        return dict(method="dfire", rec=os.path.join(g, "lightdock_receptor_membrane.pdb"), lig=os.path.join(g, "lightdock_ligand.pdb"),
                    kw={}, pos=os.path.join(g, "initial_positions_0.dat"), cols=7)
    if name == "1ppe":
        return dict(method="dfire", rec=os.path.join(g, "lightdock_1ppe_e.pdb"), lig=os.path.join(g, "lightdock_1ppe_i.pdb"),
                    kw=dict(rec_active=["E.ILE.16"]), pos=os.path.join(g, "initial_positions_0.dat"), cols=7)
    if name == "1azp":
        return dict(method="dna", rec=os.path.join(g, "lightdock_protein.pdb"), lig=os.path.join(g, "lightdock_dna.pdb"),
                    kw=dict(rec_active=["A.TRP.24", "A.VAL.26", "A.ARG.42"], lig_active=["B.DT.13"],
                            rec_nmodes=np.load(os.path.join(g, "rec_nm.npy")), rec_num_anm=10,
                            lig_nmodes=np.load(os.path.join(g, "lig_nm.npy")), lig_num_anm=10, use_anm=True),
                    pos=os.path.join(g, "initial_positions_0.dat"), cols=27)
    if name == "2uuy":
        return dict(method="dfire", rec=os.path.join(g, "lightdock_2UUY_rec.pdb"), lig=os.path.join(g, "lightdock_2UUY_lig.pdb"),
                    kw=dict(rec_nmodes=np.load(os.path.join(g, "rec_nm.npy")), rec_num_anm=10,
                            lig_nmodes=np.load(os.path.join(g, "lig_nm.npy")), lig_num_anm=10, use_anm=True),
                    pos=os.path.join(g, "initial_positions_0.dat"), cols=27)
    raise SystemExit("unknown system " + name)


WORKLOADS = {  # name -> (system, kind, default batch / swarms)
    "1k4c": ("1k4c", "k1", 8192), "1ppe": ("1ppe", "k1", 65536), "1azp-dna": ("1azp", "k1", 16384),
    "gso-1ppe": ("1ppe", "gso", 1024), "gso-1k4c": ("1k4c", "gso", 64), "2uuy": ("2uuy", "k1", 16384),
}


def read_positions(path, cols):
    return np.array([[float(v) for v in line.split(" ")] for line in open(path).read().splitlines()])[:, :cols]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cores():
    """Cores this process may really use: the affinity mask, cut by a cgroup CPU quota if there is one
    (a container with 64 visible CPUs and a quota of 16 scales to 16, whatever the thread count)."""
    n = len(os.sched_getaffinity(0))
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            text = open(path).read().split()
            if path.endswith("cpu.max"):
                if text[0] != "max":
                    quota = float(text[0]) / float(text[1])
            else:
                q = float(text[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    return n, quota


def cpu_baseline(case, table, poses, budget_s, threads):
    """Time the CPU oracle (test infrastructure used ONLY as the reported baseline) over a bounded
    sample of the bench poses: (a) one thread, the stand-in for the single-threaded Rust path;
    (b) `threads` pthreads inside the oracle library, every thread its own stream of poses, the
    stand-in for `ant_thony.py --cores N` (example/1czy/execution.sh:24)."""
    orc = ge.oracle()
    kw = dict(case["kw"])
    if case["method"] == "dfire":
        kw["potential"] = table
    scorer = orc.Scorer(case["method"], case["rec"], case["lig"], **kw)
    scorer.energy_row(poses[0])                                   # warm
    n1 = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < min(2.0, budget_s / 8) and n1 < len(poses):
        scorer.energy_row(poses[n1])
        n1 += 1
    single = n1 / (time.perf_counter() - t0)
    wall = max(3.0, budget_s * 7 / 8 / threads)                   # at least 3 s of wall clock on all threads
    n = int(max(threads, min(len(poses), wall * single * threads)))
    n -= n % threads
    t0 = time.perf_counter()
    out = scorer.energy_rows_mt(poses[:n], threads)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "evals/s", "cores": threads, "kind": "port", "single_thread_value": single,
            "cpu_model": cpu_model(),
            "sample": "%d of the bench poses on %d pthreads in %.1f s wall (+ %d poses on 1 thread); C oracle -O2, f64, "
                      "no SIMD intrinsics" % (n, threads, dt, n1)}, out


KERNEL_SOURCES = ("lightdock-rust_amd/csrc/kernels/dfire_bm.hip", "lightdock-rust_amd/csrc/kernels/dfire_bm.hpp",
                  "lightdock-rust_amd/csrc/kernels/dfire_device.hpp", "lightdock-rust_amd/csrc/kernels/dfire_packed.hip",
                  "lightdock-rust_amd/csrc/kernels/dfire_packed.hpp", "lightdock-rust_amd/csrc/kernels/dfire_tiled.hip",
                  "lightdock-rust_amd/csrc/kernels/dfire_tiled.hpp", "lightdock-rust_amd/csrc/kernels/pose_energy.hip",
                  "lightdock-rust_amd/csrc/kernels/pose_energy.hpp", "lightdock-rust_amd/csrc/kernels/gso_step.hip",
                  "lightdock-rust_amd/csrc/scorer.cpp", "lightdock-rust_amd/csrc/host/spatial_order.cpp")


def kernel_source_hash():
    """What ties a committed counter profile to the build it was taken from: a hash of the kernel sources and of
    scorer.cpp (launch shapes, LUTs, layouts) and spatial_order.cpp (the atom order: how many blocks a pose has), comments and whitespace left out.  tools/update_traffic.py stamps it on every entry of profiles/traffic.json."""
    import hashlib
    import re
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        text = open(os.path.join(ROOT, rel), "r", encoding="utf-8", errors="replace").read()
        # the CODE: comments and layout do not change what a profile measured
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        text = re.sub(r"//[^\n]*", " ", text)
        h.update(rel.encode())
        h.update(" ".join(text.split()).encode())
    return h.hexdigest()[:16]


def profile_entry(workload, units, kernel, flags=""):
    """Per-launch counter values of the pair kernel from the rocprofv3 PMC passes committed under
    profiles/ (collected in separate --pmc runs of this same command; profiles/README.md gives the
    unit and gfx950 corrections).  Returns (entry, stale): {} when no profile matches workload, units per launch,
    flags and kernel; stale = the entry was taken from another build of the kernels (its counters are NOT reported)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return {}, False
    e = t.get("%s:%d:%s%s" % (workload, units, kernel, flags), {})
    if e and e.get("source_hash") != kernel_source_hash():
        return {}, True
    return e, False


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the layout
    torch.distributed.run gives), BEFORE anything initialises a GPU in this process.  Rank 0 prints the JSON line."""
    import socket
    import subprocess
    forced = os.environ.get("LD_BENCH_FORCE_DEVICE")
    if forced is None and "--backend" in argv and argv[argv.index("--backend") + 1] == "gloo" and os.environ.get("LD_BENCH_NO_GPU_CHECK"):
        visible = n                                   # CPU dry run of the launcher (tests/test_multi_cpu.py)
    else:
        import torch                                  # counting devices does not initialise the GPU
        visible = torch.cuda.device_count()
    if forced is None and n > visible:
        raise SystemExit("--gpus %d but %d device(s) visible (set LD_BENCH_FORCE_DEVICE=<id> for a dry run of the N-rank path on one GPU)" % (n, visible))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    # Poll: when one rank dies the others would sit in their rendezvous or a collective until the backend's timeout
    # (10 to 30 minutes) -- end them and report.
    import time as _time
    codes = [None] * n
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            deadline = _time.time() + 5.0          # the others may be failing for the same reason: let them say so
            while _time.time() < deadline and any(p.poll() is None for p in procs):
                _time.sleep(0.05)
            ended = []
            for r, p in enumerate(procs):
                if p.poll() is None:
                    p.terminate()
                    try:
                        p.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        p.wait()
                    ended.append(r)
            failed = [(r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0 and r not in ended]
            raise SystemExit("rank(s) failed: %s%s" % (", ".join("%d (exit %d)" % rc for rc in failed),
                                                       "; ended rank(s) %s that were still waiting" % ended if ended else ""))
        _time.sleep(0.05)
    raise SystemExit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="1k4c", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="poses per GPU per step (pose-energy workloads; 0 = the workload's default)")
    ap.add_argument("--swarms", type=int, default=0, help="GSO workloads: swarms (gso-1ppe: in total, sharded; gso-1k4c: per GPU)")
    ap.add_argument("--cpu-seconds", type=float, default=200.0, help="CPU-baseline budget in core-seconds (0 = skip)")
    ap.add_argument("--zero-last-bin", action="store_true",
                    help="DFIRE: zero bin 19 (14..15 A) of the synthetic table, as DFIRE's reference state does by construction; "
                         "NOT the headline configuration, reported separately in DESIGN.md")
    ap.add_argument("--zero-bead-rows", action="store_true",
                    help="DFIRE: zero the rows of receptor type 167 (lightdock's membrane beads, src/dfire.rs:40,77) in the synthetic table -- "
                         "what a DCparams without statistics for the beads would hold; their subtiles are then listed within the interface "
                         "distance only (VERDICT r05 item 8).  NOT the headline configuration, reported separately in DESIGN.md")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for dry runs)")
    ap.add_argument("--no-stats", action="store_true",
                    help="skip the counting launch (in-cutoff pairs, blocks) in front of the timed region: for profiler runs, whose per-kernel "
                         "means then cover one population of launches; the byte model is not reported")
    args = ap.parse_args()
    system, kind, default_size = WORKLOADS[args.workload]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus, sys.argv[1:])          # does not return

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the pose-energy path has no CPU fallback")
    local = ge.package().multi.device_of_rank(local, torch.cuda.device_count(), os.environ.get("LD_BENCH_FORCE_DEVICE"))
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
    multi = pkg.multi
    case = load_case(system)
    table = pkg.synth.dcparams() if case["method"] == "dfire" else None
    if table is not None and args.zero_last_bin:
        table = table.copy()
        table.reshape(169, 169, 20)[:, :, 19] = 0.0
    if table is not None and args.zero_bead_rows:
        table = table.copy()
        table.reshape(169, 169, 20)[167, :, :] = 0.0
    kw = dict(case["kw"])
    if table is not None:
        kw["potential"] = table
    scorer = pkg.Scorer.from_pdb(case["method"], case["rec"], case["lig"], **kw)
    info = scorer.kernel_info()
    base = read_positions(case["pos"], case["cols"])
    dev = torch.device("cuda", local)
    stream = torch.cuda.current_stream()
    n_rec, n_lig = scorer.num_atoms(0), scorer.num_atoms(1)
    extra = {}

    if kind == "k1":
        batch = args.batch or default_size
        # swarms shard across ranks with no exchange: every rank gets its own, differently seeded batch
        poses = pkg.synth.jitter(base, batch, seed=1000 + rank)
        d_poses = torch.from_numpy(poses).to(dev)
        d_out = torch.empty(batch, dtype=torch.float64, device=dev)
        d_cnt = torch.zeros(batch, dtype=torch.int32, device=dev)
        # a stream of torch's own (not the NULL stream: ld_scorer_set_stream(NULL) means "the handle's own stream", which torch events cannot see)
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        scorer.set_stream(stream.cuda_stream)

        def step(counts=False):
            scorer.energy_batch_device(batch, d_poses.data_ptr(), poses.shape[1], d_out.data_ptr(), None,
                                       d_cnt.data_ptr() if counts else None)

        # pairs inside the outer cutoff of this batch (a counting launch, outside the timed region)
        blocks = None
        if args.no_stats:
            p_cut = np.zeros(batch, dtype=np.int64)
        else:
            step(counts=True)
            torch.cuda.synchronize()
            p_cut = d_cnt.cpu().numpy().astype(np.int64)
            try:
                blocks = float(scorer.last_block_counts(batch).mean())     # 8x8 blocks the box culling let through
            except pkg.LightdockError:
                blocks = None
        gather_bytes = 8 * int(p_cut.sum()) if case["method"] == "dfire" else 0
        algo_bytes_launch = float(info["stream_bytes_per_pose"] * batch + gather_bytes)
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        scorer.enable_timing(True)
        scorer.pair_kernel_time()          # reset

        # the spread of the timed steps: one event per step boundary on the launch stream (the scorer runs on torch's current
        # stream here, so torch events see it); recorded inside the timed region, read after it
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]

        def timed():
            marks[0].record(stream)
            for k in range(args.steps):
                step()
                marks[k + 1].record(stream)

        elapsed = multi.timed_region(timed, dist, sync=torch.cuda.synchronize)     # barrier + synchronize both sides, MAX over ranks
        step_ms = np.array([marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps)])
        kern_ms, launches = scorer.pair_kernel_time()
        scorer.enable_timing(False)
        energies = d_out.cpu().numpy()
        total_evals = batch * args.steps * world
        scaling = "weak"
        shape = "%s %s pose-energy batch, %d poses/GPU/step, %d x %d atoms%s" % (
            system, case["method"].upper(), batch, n_rec, n_lig, ((", synthetic DCparams" + (" with bin 19 zeroed" if args.zero_last_bin else "") + (" with the membrane beads' rows (type 167) zeroed" if args.zero_bead_rows else "")) if table is not None else "") + (", 10 + 10 ANM modes" if case["kw"].get("use_anm") else ""))
        extra = {"poses_per_step_per_gpu": batch, "nominal_pair_tests_per_pose": info["pair_tests_per_pose"],
                 "mean_pairs_in_cutoff": float(p_cut.mean()), "mean_8x8_blocks_evaluated": blocks}
        if case["method"] == "dfire" and scorer.bm_quiet_subtiles():
            # (receptor subtiles whose rows of the potential are zero: listed within the interface distance only; the block
            # count above is the counting launch's, which tests every box against the full cutoff)
            extra["quiet_receptor_subtiles"] = scorer.bm_quiet_subtiles()
        units_per_launch = batch
        cpu_poses = poses
        pairs_for_err = p_cut if not args.no_stats else np.full(batch, float(n_rec) * n_lig)
    else:
        swarms_total = args.swarms or default_size
        if args.workload == "gso-1ppe":      # config 5: a fixed set of swarms, sharded
            mine = list(multi.shard(swarms_total, rank, world))
            scaling = "strong"
        else:
            mine = list(range(rank * swarms_total, (rank + 1) * swarms_total))
            scaling = "weak"
        if not mine:
            raise SystemExit("rank %d has no swarm: --swarms %d over %d ranks" % (rank, swarms_total, world))
        if system == "1ppe":                 # swarm 0 = the example's, the others SURVEY 8d's synthetic recipe
            pos = np.stack([base if s == 0 else pkg.synth.swarm(200, seed=s) for s in mine])
        else:
            pos = np.stack([base if s == 0 else pkg.synth.jitter(base, 200, seed=s) for s in mine])
        gso = pkg.GSO(scorer, pos)
        gso.run(max(args.warmup, 6))         # the share of glowworms that move settles after about 6 steps
        e0 = gso.num_evals                    # synchronises

        def timed():
            gso.run(args.steps)

        elapsed = multi.timed_region(timed, dist, sync=torch.cuda.synchronize)
        step_ms = None                            # (a GSO run is one asynchronous call: no per-step marks)
        evals = gso.num_evals - e0
        total_evals = int(multi.sum_over_ranks(evals, dist))
        # K1 / K2 split of this rank: a few more steps with events around the pair kernel
        scorer.enable_timing(True)
        scorer.pair_kernel_time()
        e1 = gso.num_evals
        t0 = time.perf_counter()
        gso.run(10)
        e2 = gso.num_evals
        dt10 = time.perf_counter() - t0
        kern_ms, launches = scorer.pair_kernel_time()
        scorer.enable_timing(False)
        # algorithmic bytes of the K1 launches: P_cut of a pose taken as the mean over the swarms' start poses
        d_poses = torch.from_numpy(pos.reshape(-1, pos.shape[-1])[:4096].copy()).to(dev)
        nb = d_poses.shape[0]
        d_out = torch.empty(nb, dtype=torch.float64, device=dev)
        d_cnt = torch.zeros(nb, dtype=torch.int32, device=dev)
        scorer.energy_batch_device(nb, d_poses.data_ptr(), pos.shape[-1], d_out.data_ptr(), None, d_cnt.data_ptr())
        torch.cuda.synchronize()
        mean_cut = float(d_cnt.cpu().numpy().mean())
        pairs_for_err = d_cnt.cpu().numpy().astype(np.float64)
        energies = d_out.cpu().numpy()
        evals_per_launch = (e2 - e1) / max(launches, 1)
        algo_bytes_launch = float((info["stream_bytes_per_pose"] + 8 * mean_cut) * evals_per_launch)
        shape = "%s DFIRE GSO, %d swarms x 200 glowworms%s, %d x %d atoms, synthetic DCparams%s" % (
            system, swarms_total, " sharded over the ranks" if scaling == "strong" else " per GPU", n_rec, n_lig,
            (" with bin 19 zeroed" if args.zero_last_bin else "") + (" with the membrane beads' rows (type 167) zeroed" if args.zero_bead_rows else ""))
        extra = {"swarms_this_rank": len(mine), "glowworms": 200, "mean_pairs_in_cutoff_of_start_poses": mean_cut,
                 "gso_steps_per_s": args.steps / elapsed,
                 "k1_k2_split": {"pair_kernel_ms_per_step": kern_ms / max(launches, 1), "whole_step_ms": 1e3 * dt10 / 10,
                                 "moved_fraction": evals_per_launch / (len(mine) * 200.0),
                                 "note": "10 extra steps with HIP events around K1; "
                                         "the rest of a step is the tail kernel, K2 and launch gaps"}}
        units_per_launch = evals_per_launch
        blocks = None
        cpu_poses = pos.reshape(-1, pos.shape[-1])

    if rank == 0:
        kern_s = kern_ms / 1e3 / max(launches, 1)
        achieved = algo_bytes_launch / kern_s / 1e9
        flags = ("" if not args.zero_last_bin else ":zero-last-bin") + ("" if not args.zero_bead_rows else ":zero-bead-rows")
        prof, stale = profile_entry(args.workload, int(units_per_launch) if kind == "k1" else len(mine), info["pair_kernel_name"], flags)
        # The vector units issue most of what these kernels run (packed f32, f64, conversions, integer max) at one wave
        # instruction per ~4.5 cycles per SIMD (profiles/r02_valu_issue_rates.txt); only plain f32 add / mul / fma and
        # bitwise operations go faster.  The instruction-count floor of a launch is what that rate allows.
        binding = None
        issue_ns, issue_src = valu_issue_cost()
        if prof.get("valu_insts_per_launch"):
            floor_ms = 1e-6 * prof["valu_insts_per_launch"] * issue_ns / 1024
            binding = {"what": prof.get("binding", "valu-issue"), "valu_floor_ms": floor_ms, "frac_of_kernel_time": floor_ms / (1e3 * kern_s),
                       "how": "SQ_INSTS_VALU (profiles/traffic.json) x %.3f ns per wave instruction per SIMD (profiles/valu_issue.json) / 1024 SIMDs" % issue_ns,
                       "issue_cost": issue_src}
            if prof.get("vmem_insts_per_launch"):
                binding["tcp_floor_ms"] = 1e3 * prof["vmem_insts_per_launch"] * 17 / (256 * 2.4e9)
        compute = None
        if prof.get("valu_insts_per_launch"):
            lane_ops = 64.0 * prof["valu_insts_per_launch"] / kern_s
            compute = {"bound": "vector issue (f64/f32 VALU, one op per lane per cycle)", "achieved": lane_ops / 1e12, "peak": VALU_LANE_OPS_PER_S / 1e12,
                       "unit": "T lane-ops/s", "frac": lane_ops / VALU_LANE_OPS_PER_S,
                       "valu_insts_per_launch": prof["valu_insts_per_launch"], "source": prof.get("valu_source")}
        l1 = None
        if prof.get("tcp_cache_accesses_per_launch"):     # one tag lookup per cycle per CU
            rate = prof["tcp_cache_accesses_per_launch"] / kern_s
            l1 = {"bound": "L1 (TCP) tag lookups, one per cycle per CU", "achieved": rate / 1e9, "peak": 256 * 2.4, "unit": "G lookups/s",
                  "frac": rate / (256 * 2.4e9), "l2_read_requests_per_launch": prof.get("l2_read_requests_per_launch")}
        l2 = None
        if prof.get("l2_read_requests_per_launch"):      # what binds the DFIRE gather: lines filled into the L1s
            rate = L2_REQUEST_BYTES * prof["l2_read_requests_per_launch"] / kern_s / 1e9
            l2 = {"bound": "L2 -> L1 fills (128-byte lines)", "achieved": rate, "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": rate / L2_PEAK_GBS,
                  "l2_read_requests_per_launch": prof["l2_read_requests_per_launch"],
                  "random_line_gather_ceiling": L2_GATHER_CEILING_GBS, "frac_of_gather_ceiling": rate / L2_GATHER_CEILING_GBS,
                  "source": "TCP_TCC_READ_REQ_sum per launch of the pair kernel (profiles/), 128 bytes each; ceiling measured by "
                            "tools/microbench/l2_gather.hip (profiles/r02_l2_gather_ceiling.txt)"}
        hbm_frac = achieved / HBM_PEAK_GBS
        hbm_model = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac,
                     "model_exceeded": bool(hbm_frac > 1.0), "algorithmic_bytes_per_launch": algo_bytes_launch,
                     "note": "SURVEY 8d's byte model over the kernel time: 8 bytes per in-cutoff pair are table reads that the block-major "
                             "kernels serve from LDS (the pose-major ones from L2), so this is bookkeeping, not HBM utilisation; the "
                             "measured HBM traffic is `traffic`"}
        if args.no_stats and kind == "k1":
            hbm_model = None      # (no counting launch ran: P_cut unknown)
        lds = None
        if prof.get("lds_busy_cycles_per_launch"):
            busy = prof["lds_busy_cycles_per_launch"] / (256 * 2.4e9 * kern_s)
            lds = {"bound": "LDS pipe busy (SQ_LDS_IDX_ACTIVE per CU)", "frac": busy, "bank_conflict_share": prof.get("lds_bank_conflict_share"),
                   "note": "of the whole sequence's time; the pair kernel alone keeps it busier"}
        if binding:      # what binds: vector issue
            roof = {"bound": "valu-issue", "achieved": prof["valu_insts_per_launch"] / kern_s / 1e9, "peak": 1024 / issue_ns,
                    "unit": "G wave-instructions/s", "frac": binding["frac_of_kernel_time"]}
        else:            # no counter profile of this build: the byte model, flagged as such
            roof = {"bound": "hbm", "achieved": achieved if hbm_model else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": hbm_frac if hbm_model else None, "model_exceeded": bool(hbm_frac > 1.0)}
        # the two figures SURVEY 8(d) / the tier's rule define, at the top level beside the binding roof: the byte MODEL over the
        # kernel time, and the HBM traffic the counters saw over the kernel time, both against the 8 TB/s peak
        roof["frac_survey_8d"] = hbm_frac if (hbm_model is not None) else None
        roof["hbm_traffic_frac"] = (prof["hbm_bytes_per_launch"] / kern_s / 1e9 / HBM_PEAK_GBS) if prof.get("hbm_bytes_per_launch") else None
        roof.update({"traffic": prof.get("hbm_bytes_per_launch"), "kernel": info["pair_kernel_name"], "kernel_ms": 1e3 * kern_s,
                     "numerator_source": "`achieved` (valu-issue) = SQ_INSTS_VALU per launch READ FROM profiles/traffic.json (the committed rocprofv3 --pmc pass "
                                         "of this command, hash-tied to the kernel sources; not re-measured in this run) over the LIVE kernel time (HIP events); "
                                         "`frac_survey_8d` = SURVEY 8d's algorithmic bytes (P_cut counted live on the GPU) over the live kernel time / 8 TB/s; "
                                         "`hbm_traffic_frac` = `traffic` (FETCH_SIZE + WRITE_SIZE passes, profiles/traffic.json) over the live kernel time / 8 TB/s",
                     "note": ("`kernel_ms` brackets the whole block-major sequence (memset, dfire_bm_pose, _cull, _plan, _census, _order, _pairs, "
                              "_gather) and `traffic`, `compute`, `binding`, `lds` are sums over it; vector instructions from the committed "
                              "rocprofv3 pass of this command (profiles/), most of them 4.5 cycles per wave on a SIMD; "
                              "without a profile of this build (`profile_stale`) the byte model stands in") if info["pair_kernel_name"].startswith("dfire_bm") else
                             ("vector instructions from the committed rocprofv3 pass of this command (profiles/) at 4.5 cycles per wave on a SIMD, over "
                              "the live kernel time; the working set is L2 resident -- see also `l2`, `l1`"),
                     "profile_stale": stale, "hbm_model": hbm_model, "binding": binding, "compute": compute, "lds": lds, "l1": l1, "l2": l2,
                     "nominal_pair_tests_per_s": info["pair_tests_per_pose"] * units_per_launch / kern_s,
                     "evaluated_pair_tests_per_s": (64.0 * blocks * units_per_launch / kern_s) if blocks else None})
        # `dtype` names the type of the reference's arithmetic and of the OUTPUT; what the timed kernels compute in:
        if info["pair_kernel_name"].startswith("dfire_bm"):
            arithmetic = ("f32 filter (affine-map posing, packed-f32 distance form, 1/16-unit cell LUT), i64 fixed-point sums (table values "
                          "rounded once to 2^-40), f64 exact path for every pair whose cell holds a bin step / the cutoff / an interface decision; "
                          "decisions (cutoff, bin, interface flag, pair count) = the reference's f64 decisions, bit for bit")
        elif info["pair_kernel_name"].startswith("dfire_packed"):
            arithmetic = "f32 filter, f64 sums, f64 exact path for every doubtful pair; decisions = the reference's f64 decisions"
        else:
            arithmetic = "f64 throughout, the reference's operation order (one v_rcp_f64 + Newton step in the DNA pair term)"
        out = {
            "metric": "pose-energy evals/sec (%s, %s)" % (case["method"].upper(), args.workload),
            "value": total_evals / elapsed, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "arithmetic": arithmetic, "data": "synthetic",
            "config": dict({"workload": shape, "parallelism": "swarm-sharded x%d, no collectives" % world}, **extra),
            "roofline": roof,
        }
        if step_ms is not None and len(step_ms):
            # how much a single run's mean can be trusted: the spread of the K timed steps (events between steps on the launch stream)
            out["ms_per_step_min"] = float(step_ms.min())
            out["ms_per_step_median"] = float(np.median(step_ms))
            out["ms_per_step_max"] = float(step_ms.max())
            out["value_at_median_step"] = float(units_per_launch * world / (np.median(step_ms) * 1e-3))
        if args.cpu_seconds > 0 and world == 1:      # reported baseline: rank 0 at N = 1 only
            visible, quota = host_cores()
            threads = min(visible, 64)
            if quota:                                # more threads than the quota only time-slice
                threads = max(1, min(threads, int(quota + 0.5)))
            cb, cpu_e = cpu_baseline(case, table, cpu_poses, args.cpu_seconds, threads)
            cb["visible_cpus"] = visible
            cb["cgroup_cpu_quota"] = quota       # None = unlimited; `cores` = the threads run = min(visible, 64, quota)
            out["cpu_baseline"] = cb
            n = min(len(cpu_e), len(energies))
            diff = np.abs(energies[:n] - cpu_e[:n])
            # what was measured, plainly: the largest absolute error, and the largest error relative to max(|energy|, 1e-9)
            out["parity_max_abs_err_vs_cpu_sample"] = float(np.max(diff))
            out["parity_max_rel_err_plain"] = float(np.max(diff / np.maximum(np.abs(cpu_e[:n]), 1e-9)))
            if info["pair_kernel_name"].startswith("dfire_bm"):
                # the block-major path sums table values as 64-bit fixed point, every value rounded once to 2^-(44 - e), 2^e >= the table's
                # largest |value| (2^-40 for the synthetic table; no bench complex needs the count-aware extra bits x of dfire_bm_fix_scale):
                # an ABSOLUTE error model, |err| <= P_cut * 2^-(45 - e) * 0.0157 per pose (src/dfire.rs:347).  The gate is relative on what
                # exceeds THAT allowance (P_cut counted on the GPU for this batch: 8e-10 for a 1k4c pose, 2-7e-12 observed).
                e_bits = int(np.ceil(np.log2(max(1.0, float(np.max(np.abs(table)))))))
                m = min(n, len(pairs_for_err))
                allow = np.full(n, float(np.max(pairs_for_err)) * 2.0 ** -(45 - e_bits) * 0.0157)
                allow[:m] = np.asarray(pairs_for_err[:m], dtype=np.float64) * 2.0 ** -(45 - e_bits) * 0.0157
                out["parity_abs_allowance_model_max"] = float(allow.max())
                diff = np.maximum(diff - allow, 0.0)
            rel = float(np.max(diff / np.maximum(np.abs(cpu_e[:n]), 1e-9)))
            out["parity_max_rel_err_vs_cpu_sample"] = rel    # the GATE's measure (block-major: of what exceeds the fixed point's model allowance)
            if rel > 1e-9:                           # (north_star's tolerance is 1e-4)
                raise SystemExit("parity violated: %g" % rel)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
