import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as ge
import torch; torch.cuda.init()
pkg, orc = ge.package(), ge.oracle(); pkg.init(0)
table = pkg.synth.dcparams()
for name, files in (("1ppe", ("lightdock_1ppe_e.pdb", "lightdock_1ppe_i.pdb")), ("1k4c", ("lightdock_receptor_membrane.pdb", "lightdock_ligand.pdb"))):
    g = os.path.join(ge.GOLDEN, name)
    s = pkg.Scorer.from_pdb("dfire", os.path.join(g, files[0]), os.path.join(g, files[1]), potential=table)
    poses = orc.parse_positions(os.path.join(g, "initial_positions_0.dat"))[:, :7]
    for p in poses[:20]: s.energy(p[:3], p[3:7])
    t0 = time.perf_counter()
    for p in poses: s.energy(p[:3], p[3:7])
    one = (time.perf_counter() - t0) / len(poses)
    s.energy_batch(poses)
    t0 = time.perf_counter()
    for _ in range(20): s.energy_batch(poses)
    batch = (time.perf_counter() - t0) / 20
    print("%s: Score::energy-equivalent call %.1f us; one 200-pose batch call %.1f us (%.2f us per pose)" % (name, one * 1e6, batch * 1e6, batch * 1e6 / len(poses)))
