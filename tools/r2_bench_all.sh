cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bench_all
for w in 1k4c 1ppe 1azp-dna gso-1ppe gso-1k4c; do
  echo "== $w"; timeout 600 python bench.py --workload $w 2>&1 | tail -1 | tee gpurun_out/bench_all/$w.json | cut -c1-400
done
echo "== gso-1ppe on 2 ranks of one GPU (gloo timing collectives)"
LD_BENCH_FORCE_DEVICE=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload gso-1ppe --backend gloo --steps 10 2>&1 | tail -1 | tee gpurun_out/bench_all/gso-1ppe_2ranks.json | cut -c1-300
