"""GPU box: 200 GSO steps of 2uuy (DFIRE + ANM, 10 + 10 modes: the block-major path's ANM form) against the oracle -- neighbour
counts and targets identical at every checked step, energies and poses compared.  Usage: python tools/long_run_check_anm.py [steps]"""
import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import __graft_entry__ as ge
import torch; torch.cuda.init()
from conftest import case_kwargs, case_positions
pkg, orc = ge.package(), ge.oracle()
pkg.init(0)
table = pkg.synth.dcparams()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
method, rec, lig, kw = case_kwargs("2uuy", orc, table)
poses = case_positions("2uuy", orc)
hip = pkg.Scorer.from_pdb(method, rec, lig, **kw)
cpu = orc.Scorer(method, rec, lig, **kw)
print("kernel:", hip.kernel_info()["pair_kernel_name"], "pose columns:", poses.shape[1])
a, b = pkg.GSO(hip, poses), orc.GSO(cpu, poses)
worst_e = worst_p = 0.0
for step in range(1, steps + 1):
    a.step(); b.step()
    if step % 10 == 0 or step < 5:
        sa, sb = a.read(0), b.state()
        assert np.array_equal(sa["target"], sb["target"]) and np.array_equal(sa["n_neighbors"], sb["n_neighbors"]), step
        assert np.array_equal(sa["moved"], sb["moved"]), step
        worst_e = max(worst_e, float(np.max(np.maximum(np.abs(sa["scoring"] - sb["scoring"]) - 1e-11, 0.0) / np.maximum(1e-9, np.abs(sb["scoring"])))))
        worst_p = max(worst_p, float(np.max(np.abs(sa["poses"] - sb["poses"]))))
print("%d GSO steps 2uuy DFIRE + ANM: indices identical at every checked step; worst energy error (relative, of what exceeds 1e-11) %.3e, worst pose deviation %.3e; evals %d / %d" % (steps, worst_e, worst_p, a.num_evals, b.num_evals))
