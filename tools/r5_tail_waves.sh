cd $GRAFT_REPO_ROOT
out=gpurun_out/r05tail; mkdir -p $out
LIGHTDOCK_BM_DEBUG=$out/waves.txt timeout 300 python3 tools/gso_tail.py 1024 40 0.01 > $out/log_dbg.txt 2>&1
tail -2 $out/log_dbg.txt
python3 - $out/waves.txt <<'PY'
import numpy as np, sys
d=np.loadtxt(sys.argv[1])
t0,t1,jobs,batches=d[:,0],d[:,1],d[:,2],d[:,3]
life=(t1-t0)/100
print(len(d), "span", (t1.max()-t0.min())/100, "life mean",life.mean(), life.min(), life.max(), "last start", (t0.max()-t0.min())/100)
print("jobs mean",jobs.mean(),"max",jobs.max(),"batches mean",batches.mean(), "max", batches.max(), "sum", batches.sum())
print("batch time", (d[:,4]/100).mean(), "drain",(d[:,5]/100).mean(), "max", (d[:,5]/100).max(), "block",(d[:,6]/100).mean(),"job",(d[:,7]/100).mean())
busy = jobs > 0
print("waves with jobs", busy.sum(), "life of those", life[busy].mean(), life[busy].max(), " idle waves life", life[~busy].mean() if (~busy).any() else 0)
i = np.argsort(-life)[:5]
print(d[i][:, 2:], life[i])
PY
