#!/bin/bash
# round 6: kernel traces of a ONE-swarm GSO step (1k4c, 1ppe; block-major and pose-major K1), of a 128-swarm
# step of 1ppe (one GPU's share of config 5 at 8 GPUs) and of single Score::energy-equivalent calls
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_small; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {  # name, env, args...
  name=$1; shift; envs=$1; shift
  env $envs timeout 200 python3 tools/bench_extra.py "$@" > $O/$name.json 2>&1
  ( export $envs; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -- python3 tools/bench_extra.py "$@" > $O/${name}_traced.json 2> $O/${name}_trace.log )
  echo "== $name: $(tail -1 $O/$name.json)"
  python3 tools/step_timeline.py $O/trace_$name | tee $O/${name}_timeline.txt
}
run gso1k4c_1swarm_bm "LD_X=1" --what gso1k4c --swarms 1 --steps 200
run gso1k4c_1swarm_packed "LIGHTDOCK_TILED_LATENCY=1" --what gso1k4c --swarms 1 --steps 200
run gso1ppe_1swarm_bm "LD_X=1" --what gso --swarms 1 --steps 200
run gso1ppe_1swarm_packed "LIGHTDOCK_TILED_LATENCY=1" --what gso --swarms 1 --steps 200
run gso1ppe_128swarms "LD_X=1" --what gso --swarms 128 --steps 60
timeout 200 python3 tools/call_latency.py > $O/call_latency.txt 2>&1; cat $O/call_latency.txt
