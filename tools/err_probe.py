import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
pkg = ge.package(); orc = ge.oracle(); pkg.init(0)
table = pkg.synth.dcparams()
for name in ("1ppe", "1k4c"):
    g = os.path.join(ge.GOLDEN, name)
    import glob
    pdbs = sorted(glob.glob(os.path.join(g, "lightdock_*.pdb")))
    rec = [p for p in pdbs if p.endswith("_e.pdb") or "rec" in p][0] if name == "1ppe" else None
    if name == "1ppe":
        rec, lig = os.path.join(g, "lightdock_1ppe_e.pdb"), os.path.join(g, "lightdock_1ppe_i.pdb")
    else:
        import json
        st = json.load(open(os.path.join(g, "setup.json")))
        rec, lig = os.path.join(g, "lightdock_" + st["receptor_pdb"]), os.path.join(g, "lightdock_" + st["ligand_pdb"])
    poses = orc.parse_positions(os.path.join(g, "initial_positions_0.dat"))[:64, :7]
    hip = pkg.Scorer.from_pdb("dfire", rec, lig, potential=table)
    cpu = orc.Scorer("dfire", rec, lig, potential=table)
    got = hip.energy_batch(poses); want = cpu.energy_rows(poses)
    d = np.abs(got - want)
    print(name, "max abs %.3e  max rel %.3e  (|E| range %.2f..%.2f)" % (d.max(), (d / np.maximum(1e-12, np.abs(want))).max(), np.abs(want).min(), np.abs(want).max()))
