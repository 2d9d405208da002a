#!/usr/bin/env python3
"""GPU box: what a GSO step costs when few glowworms move (the late stage of a run).  1ppe DFIRE, S swarms x 200 glowworms; per
block of 10 steps: wall time per step and the share of glowworms evaluated.  A swarm whose glowworms all sit on one pose has no
neighbours and never moves: `live` = the share of ordinary swarms among them.  Usage: gso_tail.py [swarms] [steps] [live share]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.package(); pkg.init(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
live = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
g = os.path.join(ROOT, "tests", "golden", "1ppe")
s = pkg.Scorer.from_pdb("dfire", os.path.join(g, "lightdock_1ppe_e.pdb"), os.path.join(g, "lightdock_1ppe_i.pdb"),
                        rec_active=["E.ILE.16"], potential=pkg.synth.dcparams())
base = np.array([[float(v) for v in l.split(" ")] for l in open(os.path.join(g, "initial_positions_0.dat")).read().splitlines()])[:, :7]
pos = np.stack([(base if k == 0 else pkg.synth.swarm(200, seed=k)) if k < live * S else np.repeat(base[k % 200][None], 200, axis=0) for k in range(S)])
gso = pkg.GSO(s, pos)
print("kernel:", s.kernel_info()["pair_kernel_name"])
e_prev = gso.num_evals
for blk in range(steps // 10):
    t0 = time.perf_counter()
    gso.run(10)
    e = gso.num_evals          # synchronises
    dt = time.perf_counter() - t0
    print("steps %3d..%3d: %.3f ms per step, %.2f %% of the glowworms evaluated per step" % (10 * blk + 1, 10 * blk + 10, 1e2 * dt, 100.0 * (e - e_prev) / (10.0 * S * 200)))
    e_prev = e
