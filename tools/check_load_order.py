import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as ge
ge.smoke()                       # library first, torch not imported yet
import torch
print("torch after the library: cuda available", torch.cuda.is_available())
x = torch.arange(8, device="cuda", dtype=torch.float64)
print("torch op", float((x * x).sum()))
pkg = ge.package()
s_maps = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l]
print("hip runtimes mapped:", sorted(set(s_maps)))
