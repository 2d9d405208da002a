#!/usr/bin/env python3
"""GPU box: what a K1 launch costs whose active mask is (almost) empty -- the late stage of a GSO run,
where few glowworms still move.  1ppe DFIRE, 204 800 poses (1024 swarms x 200)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = ge.package(); pkg.init(0)
g = os.path.join(ROOT, "tests", "golden", "1ppe")
s = pkg.Scorer.from_pdb("dfire", os.path.join(g, "lightdock_1ppe_e.pdb"), os.path.join(g, "lightdock_1ppe_i.pdb"),
                        rec_active=["E.ILE.16"], potential=pkg.synth.dcparams())
base = np.array([[float(v) for v in l.split(" ")] for l in open(os.path.join(g, "initial_positions_0.dat")).read().splitlines()])[:, :7]
n = 204800
poses = pkg.synth.jitter(base, n, seed=1)
dev = torch.device("cuda:0")
d_poses = torch.from_numpy(poses).to(dev)
d_out = torch.zeros(n, dtype=torch.float64, device=dev)
s.set_stream(torch.cuda.current_stream().cuda_stream)
for frac in (1.0, 0.5, 0.1, 0.01, 0.0):
    act = (np.random.default_rng(2).random(n) < frac).astype(np.uint8)
    d_act = torch.from_numpy(act).to(dev)
    for _ in range(2):
        s.energy_batch_device(n, d_poses.data_ptr(), 7, d_out.data_ptr(), d_act.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        s.energy_batch_device(n, d_poses.data_ptr(), 7, d_out.data_ptr(), d_act.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("active fraction %.2f: %.3f ms per launch (%d active poses)" % (frac, 1e3 * dt, act.sum()))
