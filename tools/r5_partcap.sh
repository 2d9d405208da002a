# round 5: how the pair kernel's time depends on the entries a job may hold (LIGHTDOCK_BM_PART_CAP): usage bash tools/r5_partcap.sh <workload> <caps...>
cd $GRAFT_REPO_ROOT
w=${1:-1k4c}; shift
caps=${@:-1024 768 512 256}
for round in 1 2; do
for cap in $caps; do
  r=$(LIGHTDOCK_BM_PART_CAP=$cap timeout 90 python bench.py --workload $w --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f evals/s kernel %.3f ms' % (d['value'], d['roofline']['kernel_ms']))" 2>&1 | tail -1)
  echo "$w part cap $cap: $r"
done
done
