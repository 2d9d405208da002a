import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as ge
import torch; torch.cuda.init()
pkg, orc = ge.package(), ge.oracle()
pkg.init(0)
table = pkg.synth.dcparams()
g = os.path.join(ge.GOLDEN, "1ppe")
rec, lig = os.path.join(g, "lightdock_1ppe_e.pdb"), os.path.join(g, "lightdock_1ppe_i.pdb")
poses = orc.parse_positions(os.path.join(g, "initial_positions_0.dat"))[:, :7]
hip = pkg.Scorer.from_pdb("dfire", rec, lig, rec_active=["E.ILE.16"], potential=table)
cpu = orc.Scorer("dfire", rec, lig, rec_active=["E.ILE.16"], potential=table)
a, b = pkg.GSO(hip, poses), orc.GSO(cpu, poses)
worst = 0.0
for step in range(1, 401):
    a.step(); b.step()
    if step % 20 == 0 or step < 5:
        sa, sb = a.read(0), b.state()
        assert np.array_equal(sa["target"], sb["target"]) and np.array_equal(sa["n_neighbors"], sb["n_neighbors"]), step
        worst = max(worst, float(np.max(np.abs(sa["scoring"] - sb["scoring"]) / np.maximum(1e-9, np.abs(sb["scoring"])))),
                    float(np.max(np.abs(sa["poses"] - sb["poses"]))))
print("400 GSO steps 1ppe DFIRE: indices identical at every checked step; worst rel energy / abs pose deviation %.3e" % worst, "evals", a.num_evals, b.num_evals)
