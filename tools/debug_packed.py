#!/usr/bin/env python3
"""GPU box: the DFIRE kernel variants against each other and the oracle, pose by pose
(energies, in-cutoff pair counts).  Usage: python tools/debug_packed.py [case ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
import torch  # noqa: E402
from conftest import case_kwargs, case_positions  # noqa: E402

pkg, orc = ge.package(), ge.oracle()
pkg.init(0)
table = pkg.synth.dcparams()
VARIANTS = [("default", {}), ("packed", {"LIGHTDOCK_DFIRE_KERNEL": "packed"}), ("packed2", {"LIGHTDOCK_DFIRE_KERNEL": "packed", "LIGHTDOCK_PACKED_CELLS": "2"}), ("eps50", {"LIGHTDOCK_PACKED_EPS_SCALE": "50"}),
            ("tiled", {"LIGHTDOCK_DFIRE_KERNEL": "tiled"}), ("allpairs", {"LIGHTDOCK_DFIRE_KERNEL": "allpairs"})]
for name in sys.argv[1:] or ["1ppe", "1k4c", "2uuy"]:
    method, rec, lig, kw = case_kwargs(name, orc, table)
    cpu = orc.Scorer(method, rec, lig, **kw)
    poses = case_positions(name, orc)[:64]
    want = cpu.energy_rows(poses)
    dev = torch.device("cuda:0")
    d_poses = torch.from_numpy(poses).to(dev)
    ref_cnt = None
    for vn, env in VARIANTS:
        os.environ.update(env)
        s = pkg.Scorer.from_pdb(method, rec, lig, **kw)
        for k in env:
            os.environ.pop(k)
        d_out = torch.zeros(64, dtype=torch.float64, device=dev)
        d_cnt = torch.zeros(64, dtype=torch.int32, device=dev)
        s.energy_batch_device(64, d_poses.data_ptr(), poses.shape[1], d_out.data_ptr(), None, d_cnt.data_ptr())
        torch.cuda.synchronize()
        got, cnt = d_out.cpu().numpy(), d_cnt.cpu().numpy()
        got2 = s.energy_batch(poses)
        rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-9)
        if ref_cnt is None:
            ref_cnt = cnt
        print("%-6s %-9s max rel %.2e (pose %d)  counting==plain %s  counts differ from first variant in %d poses  %s"
              % (name, vn, rel.max(), int(rel.argmax()), np.array_equal(got, got2), int((cnt != ref_cnt).sum()),
                 s.kernel_info()["pair_kernel_name"]))
