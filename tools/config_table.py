#!/usr/bin/env python3
"""Wall-clock of every BASELINE.json config on this box (one GPU), next to the CPU oracle CLI.
The reference publishes only whole-CLI wall-clock times (README.md:27-146, M3 Pro, 1 thread)."""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg, orc = ge.package(), ge.oracle()
orc.lib()
G = os.path.join(ROOT, "tests", "golden")
PUBLISHED = {"1ppe": 4.252, "1k4c": 112.132, "1azp": 14.228, "2uuy": 8.108}   # README.md of the reference, seconds


def run(cli, case, method, steps, with_table):
    d = tempfile.mkdtemp()
    try:
        src = os.path.join(G, case)
        for f in ("rec_nm.npy", "lig_nm.npy"):
            if os.path.exists(os.path.join(src, f)):
                shutil.copy(os.path.join(src, f), d)
        if with_table:
            os.makedirs(os.path.join(d, "data"))
            pkg.synth.write_dcparams(os.path.join(d, "data", "DCparams"))
        t0 = time.perf_counter()
        r = subprocess.run([cli, os.path.join(src, "setup.json"), os.path.join(src, "initial_positions_0.dat"), str(steps), method],
                           cwd=d, capture_output=True, text=True)
        dt = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr[-500:]
        return dt
    finally:
        shutil.rmtree(d, ignore_errors=True)


rows = []
for case, method, cpu_steps in (("1ppe", "dfire", 100), ("1k4c", "dfire", 5), ("2uuy", "dfire", 20), ("1azp", "dna", 20)):
    gpu = run(pkg.CLI_PATH, case, method, 100, method == "dfire")
    gpu2 = run(pkg.CLI_PATH, case, method, 100, method == "dfire")      # second run: page cache warm
    cpu = run(orc.CLI_PATH, case, method, cpu_steps, method == "dfire")
    rows.append({"case": case, "method": method, "gpu_cli_100_steps_s": min(gpu, gpu2), "cpu_oracle_cli_s": cpu,
                 "cpu_oracle_steps": cpu_steps, "reference_readme_100_steps_s_M3": PUBLISHED.get(case)})
    print(json.dumps(rows[-1]), flush=True)
