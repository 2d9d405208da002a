#!/bin/bash
# round 6, seventh GPU call: the whole GPU suite; config 5's share with the memset gone; traces of the 1024- and 128-swarm steps
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_seventh; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
bash tools/ab6.sh 2 > $O/ab_1k4c.txt 2>&1; cat $O/ab_1k4c.txt
for s in 1024 128; do for v in base r5; do
  if [ $v = base ]; then unset LIGHTDOCK_HIP_VARIANT; else export LIGHTDOCK_HIP_VARIANT=$v; fi
  echo "gso-1ppe $s swarms $v: $(timeout 200 python3 bench.py --workload gso-1ppe --swarms $s --cpu-seconds 0 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f evals/s step %.4f ms' % (d['value'], d['ms_per_step']))")"
done; done | tee $O/share.txt
unset LIGHTDOCK_HIP_VARIANT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for s in 1024 128 1; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_gso$s -- python3 tools/bench_extra.py --what gso --swarms $s --steps 40 > $O/gso${s}_traced.json 2> $O/gso${s}_trace.log
  echo "== gso 1ppe $s swarms"; python3 tools/step_timeline.py $O/trace_gso$s | tee $O/gso${s}_timeline.txt
done
