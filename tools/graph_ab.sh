# GPU box: GSO steps per second with and without the hipGraph replay, for several swarm counts
cd $GRAFT_REPO_ROOT
for what in gso gso1k4c; do
  for n in 1 8 64 512; do
    if [ $what = gso1k4c ] && [ $n -gt 64 ]; then continue; fi
    for g in 1 0; do
      r=$(LIGHTDOCK_GSO_GRAPH=$g timeout 200 python tools/bench_extra.py --what $what --swarms $n --steps 60 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f steps/s  %.0f evals/s' % (d['steps_per_s'], d['evals_per_s']))")
      echo "$what swarms=$n graph=$g: $r"
    done
  done
done
