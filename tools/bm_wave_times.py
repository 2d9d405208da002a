#!/usr/bin/env python3
"""GPU box: lifetimes of the persistent waves of dfire_bm_pairs (LIGHTDOCK_BM_DEBUG): how evenly the jobs are spread."""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["LIGHTDOCK_BM_DEBUG"] = "/tmp/bm_debug.txt"
subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-seconds", "0"] + sys.argv[1:], check=True, stdout=subprocess.DEVNULL)
d = np.loadtxt("/tmp/bm_debug.txt")
t0, t1, jobs, batches = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
start = t0.min()
life = (t1 - t0) / 100.0   # us (100 MHz)
end = (t1 - start) / 100.0
print("waves %d; kernel span %.1f us; wave lifetime mean %.1f us, min %.1f, max %.1f; last start %.1f us" % (len(d), end.max(), life.mean(), life.min(), life.max(), (t0.max() - start) / 100.0))
print("jobs per wave mean %.1f (min %d max %d); batches per wave mean %.1f (min %d, max %d); total batches %d" % (jobs.mean(), jobs.min(), jobs.max(), batches.mean(), batches.min(), batches.max(), batches.sum()))
print("us per batch (lifetime / batches): mean %.2f" % (life.sum() / batches.sum()))
h, edges = np.histogram(end, bins=10)
print("wave end-time histogram (us):", " ".join("%d@%.0f" % (c, e) for c, e in zip(h, edges[1:])))
print("time in batches: mean %.1f us per wave (%.2f us per batch); in exact-path drains %.1f us; in block set-up %.1f us; in job set-up %.1f us (%.2f us per job)" % ((d[:, 4] / 100).mean(), d[:, 4].sum() / 100 / batches.sum(), (d[:, 5] / 100).mean(), (d[:, 6] / 100).mean(), (d[:, 7] / 100).mean(), d[:, 7].sum() / 100 / max(jobs.sum(), 1)))
late = end > np.percentile(end, 90)
print("slowest 10%% of waves: batches %.0f, batch time %.0f us, drain time %.0f us, jobs %.1f" % (batches[late].mean(), (d[late, 4] / 100).mean(), (d[late, 5] / 100).mean(), jobs[late].mean()))
per_cu = batches.reshape(-1, 8).sum(1)
print("batches per CU: mean %.0f min %d max %d" % (per_cu.mean(), per_cu.min(), per_cu.max()))
