#!/usr/bin/env python3
"""GPU box: seeded random molecules of random sizes, random poses (overlapping ones included), restraints and
membrane beads -- default DFIRE kernel against the oracle (energies) and against the all-pairs kernel
(in-cutoff pair counts).  With `anm` as third argument the molecules flex (src/dfire.rs:288-320): 0 to 10 random normal
modes a molecule, random amplitudes -- some of them large enough to make the pose WILD for the block-major path's ANM form.
Usage: python tools/fuzz_parity.py [cases] [first seed] [anm]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
import torch  # noqa: E402
from test_gpu_parity import _random_molecule, _write_pdb  # noqa: E402

pkg, orc = ge.package(), ge.oracle()
pkg.init(0)
table = pkg.synth.dcparams()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
anm = len(sys.argv) > 3 and sys.argv[3] == "anm"
dev = torch.device("cuda:0")
worst = 0.0
with tempfile.TemporaryDirectory() as d:
    for seed in range(first, first + cases):
        rng = np.random.default_rng(seed)
        n_rec = int(rng.choice([1, 7, 8, 9, 63, 64, 65, int(rng.integers(1, 2600))]))
        n_lig = int(rng.choice([1, 8, 64, 65, int(rng.integers(1, 900))]))
        rec, lig = os.path.join(d, "rec.pdb"), os.path.join(d, "lig.pdb")
        rec_atoms = _random_molecule(rng, n_rec, float(rng.uniform(10, 45)), "A", with_beads=int(rng.integers(0, 6)) if n_rec >= 64 else 0)
        lig_atoms = _random_molecule(rng, n_lig, float(rng.uniform(6, 30)), "B")
        _write_pdb(rec, rec_atoms)
        _write_pdb(lig, lig_atoms)
        rec_active = ["A.%s.%d" % (rec_atoms[0][1], rec_atoms[0][3])]
        lig_active = ["B.%s.%d" % (lig_atoms[-1][1], lig_atoms[-1][3])]
        n = int(rng.integers(1, 70))
        poses = np.zeros((n, 7))
        poses[:, :3] = rng.uniform(-30, 30, (n, 3))
        poses[: max(1, n // 4), :3] = rng.uniform(-3, 3, (max(1, n // 4), 3))
        q = rng.normal(size=(n, 4))
        poses[:, 3:] = q / np.linalg.norm(q, axis=1, keepdims=True) * rng.uniform(0.5, 2.0, (n, 1))
        kw = dict(rec_active=rec_active, lig_active=lig_active, potential=table)
        stride = 7
        if anm:
            k_rec, k_lig = int(rng.integers(0, 11)), int(rng.integers(0, 11))
            if k_rec + k_lig == 0:
                k_lig = 3
            # mode vectors of ~unit length per atom (like a normalised ANM mode x sqrt(atoms)), amplitudes of a few angstroms
            rec_modes = rng.normal(size=(k_rec, len(rec_atoms), 3)) * rng.uniform(0.05, 0.6)
            lig_modes = rng.normal(size=(k_lig, len(lig_atoms), 3)) * rng.uniform(0.05, 0.6)
            amps = rng.normal(size=(n, k_rec + k_lig)) * rng.uniform(0.2, 3.0)
            amps[rng.random(n) < 0.1] *= 40.0          # a tenth of the poses: absurd amplitudes (wild)
            poses = np.ascontiguousarray(np.concatenate([poses, amps], axis=1))
            stride = poses.shape[1]
            kw.update(use_anm=True, rec_num_anm=k_rec, lig_num_anm=k_lig,
                      rec_nmodes=rec_modes.ravel() if k_rec else None, lig_nmodes=lig_modes.ravel() if k_lig else None)
        cpu = orc.Scorer("dfire", rec, lig, **kw)
        want = cpu.energy_rows(poses)
        hip = pkg.Scorer.from_pdb("dfire", rec, lig, **kw)
        os.environ["LIGHTDOCK_DFIRE_KERNEL"] = "allpairs"
        try:
            ref = pkg.Scorer.from_pdb("dfire", rec, lig, **kw)
        finally:
            os.environ.pop("LIGHTDOCK_DFIRE_KERNEL")
        d_poses = torch.from_numpy(poses).to(dev)
        out = []
        for s in (hip, ref):
            d_out = torch.zeros(n, dtype=torch.float64, device=dev)
            d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
            s.energy_batch_device(n, d_poses.data_ptr(), stride, d_out.data_ptr(), None, d_cnt.data_ptr())
            torch.cuda.synchronize()
            out.append((d_out.cpu().numpy(), d_cnt.cpu().numpy()))
        err = float(np.max(np.abs(hip.energy_batch(poses) - want) / np.maximum(np.abs(want), 1.0)))
        worst = max(worst, err)
        ok = err < 1e-11 and np.array_equal(out[0][1], out[1][1]) and np.max(np.abs(out[0][0] - want) / np.maximum(np.abs(want), 1.0)) < 1e-11
        print("seed %3d  rec %4d lig %3d poses %2d  %s err %.2e  pairs %d  %s" % (seed, n_rec, n_lig, n, hip.kernel_info()["pair_kernel_name"], err, int(out[0][1].sum()), "ok" if ok else "MISMATCH"))
        if not ok:
            sys.exit(1)
print("all %d cases agree; worst error %.2e of max(|E|, 1)" % (cases, worst))
