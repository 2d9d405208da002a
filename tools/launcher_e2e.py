"""End to end on one GPU, files included: S synthetic 1ppe swarms written as
initial_positions_<i>.dat, launch.py over all of them for 100 steps, 11 gso files per swarm."""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
pkg = ge.package()
run = tempfile.mkdtemp(prefix="ld_e2e_")
os.makedirs(os.path.join(run, "data"))
pkg.synth.write_dcparams(os.path.join(run, "data", "DCparams"))
init = os.path.join(run, "init"); os.makedirs(init)
for s in range(S):
    rows = pkg.synth.swarm(200, seed=s)
    with open(os.path.join(init, "initial_positions_%d.dat" % s), "w") as f:
        for r in rows:
            f.write(" ".join("%.9f" % v for v in r) + "\n")
setup = os.path.join(ROOT, "tests", "golden", "1ppe", "setup.json")
t0 = time.perf_counter()
r = subprocess.run([sys.executable, os.path.join(ROOT, "lightdock-rust_amd", "launch.py"), setup, "100", "dfire", "--swarms", "0-%d" % (S - 1),
                    "--init-dir", init], cwd=run, capture_output=True, text=True)
dt = time.perf_counter() - t0
assert r.returncode == 0, r.stderr[-2000:]
files = sum(len(os.listdir(os.path.join(run, "swarm_%d" % s))) for s in range(S))
print("launch.py: %d swarms x 200 glowworms x 100 steps on one GPU, start-up and %d output files included: %.2f s wall (%.1f swarms/s)" % (S, files, dt, S / dt))
