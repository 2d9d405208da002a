import os, sys, subprocess, numpy as np
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
os.environ["LIGHTDOCK_BM_DEBUG"]="/tmp/cull_debug.txt"; os.environ["LIGHTDOCK_HIP_VARIANT"]="culltimes"
subprocess.run([sys.executable, os.path.join(ROOT,"bench.py"),"--steps","2","--warmup","1","--cpu-seconds","0","--no-stats"]+sys.argv[1:],check=True,stdout=subprocess.DEVNULL)
d=np.loadtxt("/tmp/cull_debug.txt")
d=d[d[:,2]>0]
life=(d[:,1]-d[:,0])/100.0
st=(d[:,0]-d[:,0].min())/100.0
wg=(d[:,7]//4).astype(int)
for lo,hi in ((0,256),(256,512),(512,768),(768,1024),(1024,1280)):
    m=(wg>=lo)&(wg<hi)
    if m.any(): print("  workgroups %4d..%4d: start %.1f us after the first, life %.1f us, items %.1f"%(lo,hi-1,st[m].mean(),life[m].mean(),d[m,2].mean()))
print("culling waves sampled %d; life mean %.1f us (min %.1f max %.1f); items per wave %.1f"%(len(d),life.mean(),life.min(),life.max(),d[:,2].mean()))
for k,nm in ((3,"waiting for the drawn ticket"),(4,"item loads + boxes of 8 poses"),(5,"per-pose tile / subtile tests"),(6,"flush (LDS atomics, global atomics, entries)")):
    print("  %-48s %.1f us per wave, %.2f us per item"%(nm,(d[:,k]/100).mean(),(d[:,k].sum()/100)/d[:,2].sum()))
