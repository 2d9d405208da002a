#!/usr/bin/env python3
"""Secondary measurements (not the driver's bench line): the other BASELINE.json configs.

  * gso:  1ppe DFIRE, S swarms x 200 glowworms, K GSO steps on one GPU (config 5 per GPU share)
  * dna:  1azp DNA + ANM pose-energy batch (config 4)
  * k1:   1ppe DFIRE pose-energy batch (config 2)
Prints one JSON object per measurement.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def positions(path, cols=None):
    rows = np.array([[float(v) for v in line.split(" ")] for line in open(path).read().splitlines()])
    return rows if cols is None else rows[:, :cols]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--swarms", type=int, default=128)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--what", default="gso,dna,k1,anm")
    args = ap.parse_args()
    import torch
    torch.cuda.init()
    pkg = ge.package()
    pkg.init(0)
    table = pkg.synth.dcparams()
    g = os.path.join(ROOT, "tests", "golden")
    what = args.what.split(",")

    if "gso" in what or "k1" in what:
        d = os.path.join(g, "1ppe")
        s = pkg.Scorer.from_pdb("dfire", os.path.join(d, "lightdock_1ppe_e.pdb"), os.path.join(d, "lightdock_1ppe_i.pdb"),
                                rec_active=["E.ILE.16"], potential=table)
    if "gso" in what:
        base = positions(os.path.join(d, "initial_positions_0.dat"), 7)
        pos = np.stack([base] + [pkg.synth.swarm(200, seed=k) for k in range(1, args.swarms)])
        gso = pkg.GSO(s, pos)
        gso.run(6)
        e0 = gso.num_evals
        t0 = time.perf_counter()
        gso.run(args.steps)
        e1 = gso.num_evals                # synchronises
        dt = time.perf_counter() - t0
        print(json.dumps({"what": "gso 1ppe dfire", "swarms": args.swarms, "glowworms": 200, "steps": args.steps,
                          "steps_per_s": args.steps / dt, "evals_per_s": (e1 - e0) / dt,
                          "swarm_steps_per_s": args.steps * args.swarms / dt}))
    if "gso1k4c" in what:       # the headline system inside the GSO loop: every swarm starts from the example's
        d4 = os.path.join(g, "1k4c")   # 200 poses with its own seeded jitter
        s4 = pkg.Scorer.from_pdb("dfire", os.path.join(d4, "lightdock_receptor_membrane.pdb"),
                                 os.path.join(d4, "lightdock_ligand.pdb"), potential=table)
        base4 = positions(os.path.join(d4, "initial_positions_0.dat"), 7)
        pos = np.stack([base4] + [pkg.synth.jitter(base4, 200, seed=k) for k in range(1, args.swarms)])
        gso = pkg.GSO(s4, pos)
        gso.run(4)
        e0 = gso.num_evals
        t0 = time.perf_counter()
        gso.run(args.steps)
        e1 = gso.num_evals
        dt = time.perf_counter() - t0
        print(json.dumps({"what": "gso 1k4c dfire", "swarms": args.swarms, "glowworms": 200, "steps": args.steps,
                          "steps_per_s": args.steps / dt, "evals_per_s": (e1 - e0) / dt,
                          "swarm_steps_per_s": args.steps * args.swarms / dt}))
    if "k1" in what:
        base = positions(os.path.join(d, "initial_positions_0.dat"), 7)
        poses = pkg.synth.jitter(base, args.batch * 4, seed=3)
        s.energy_batch(poses[:1024])
        t0 = time.perf_counter()
        s.energy_batch(poses)
        dt = time.perf_counter() - t0
        print(json.dumps({"what": "k1 1ppe dfire (host buffers, PCIe inclusive)", "poses": len(poses), "evals_per_s": len(poses) / dt}))
    if "anm" in what:
        d2 = os.path.join(g, "2uuy")
        s2 = pkg.Scorer.from_pdb("dfire", os.path.join(d2, "lightdock_2UUY_rec.pdb"), os.path.join(d2, "lightdock_2UUY_lig.pdb"),
                                 rec_nmodes=np.load(os.path.join(d2, "rec_nm.npy")), rec_num_anm=10,
                                 lig_nmodes=np.load(os.path.join(d2, "lig_nm.npy")), lig_num_anm=10, use_anm=True, potential=table)
        base = positions(os.path.join(d2, "initial_positions_0.dat"))
        poses = pkg.synth.jitter(base, args.batch, seed=5)
        dev = torch.device("cuda:0")
        d_poses = torch.from_numpy(poses).to(dev)
        d_out = torch.empty(args.batch, dtype=torch.float64, device=dev)
        for _ in range(2):
            s2.energy_batch_device(args.batch, d_poses.data_ptr(), poses.shape[1], d_out.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            s2.energy_batch_device(args.batch, d_poses.data_ptr(), poses.shape[1], d_out.data_ptr())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"what": "k1 2uuy dfire + receptor/ligand ANM (HBM resident, incl. per-pose receptor image)",
                          "poses": args.batch, "evals_per_s": 5 * args.batch / dt}))
    if "dna" in what:
        d = os.path.join(g, "1azp")
        rec_nm = np.load(os.path.join(d, "rec_nm.npy"))
        lig_nm = np.load(os.path.join(d, "lig_nm.npy"))
        s = pkg.Scorer.from_pdb("dna", os.path.join(d, "lightdock_protein.pdb"), os.path.join(d, "lightdock_dna.pdb"),
                                rec_active=["A.TRP.24", "A.VAL.26", "A.ARG.42"], lig_active=["B.DT.13"], rec_nmodes=rec_nm,
                                rec_num_anm=10, lig_nmodes=lig_nm, lig_num_anm=10, use_anm=True)
        base = positions(os.path.join(d, "initial_positions_0.dat"))
        poses = pkg.synth.jitter(base, args.batch, seed=4)
        dev = torch.device("cuda:0")
        d_poses = torch.from_numpy(poses).to(dev)
        d_out = torch.empty(args.batch, dtype=torch.float64, device=dev)
        s.set_stream(torch.cuda.current_stream().cuda_stream)
        for _ in range(2):
            s.energy_batch_device(args.batch, d_poses.data_ptr(), poses.shape[1], d_out.data_ptr())
        torch.cuda.synchronize()
        s.enable_timing(True)
        s.pair_kernel_time()
        t0 = time.perf_counter()
        for _ in range(5):
            s.energy_batch_device(args.batch, d_poses.data_ptr(), poses.shape[1], d_out.data_ptr())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms, n = s.pair_kernel_time()
        info = s.kernel_info()
        print(json.dumps({"what": "k1 1azp dna+anm (HBM resident)", "poses": args.batch, "evals_per_s": 5 * args.batch / dt,
                          "pair_tests_per_s": info["pair_tests_per_pose"] * args.batch / (ms / n / 1e3),
                          "kernel_ms": ms / n, "algorithmic_GBps": info["stream_bytes_per_pose"] * args.batch / (ms / n / 1e3) / 1e9}))


if __name__ == "__main__":
    main()
