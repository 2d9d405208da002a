# GPU box: the default bench.py line of every workload with the committed profiles/traffic.json in place
# (-> gpurun_out/r04_<workload>_bench.json; copied to profiles/ by hand)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in 1k4c 1ppe 1azp-dna gso-1ppe gso-1k4c 2uuy; do
  n=$(echo $w | tr - _)
  if [ $w = 1k4c ]; then timeout 600 python bench.py > gpurun_out/r04_${n}_bench.json 2> gpurun_out/r04_${n}_bench.err
  else timeout 600 python bench.py --workload $w > gpurun_out/r04_${n}_bench.json 2> gpurun_out/r04_${n}_bench.err; fi
  tail -1 gpurun_out/r04_${n}_bench.json | cut -c1-120
done
