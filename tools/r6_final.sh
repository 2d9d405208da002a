#!/bin/bash
# round 6, the final measurement set on ONE box, with the committed profiles/traffic.json of these sources in place:
#   the default bench line of every workload, the pair kernel's wave timers (1k4c, 1ppe, 2uuy), the GSO step when nothing / 1 % /
#   10 % / everything moves, config 5 as one GPU's share at N = 1, 2, 4, 8, the one-swarm and 128-swarm step timelines (kernel
#   traces), single-call latencies, the ANM fuzz, the microbenchmarks.   -> gpurun_out/r06_final/      usage (GPU box): bash tools/r6_final.sh
cd "${GRAFT_REPO_ROOT:?}" || exit 1
out=gpurun_out/r06_final; mkdir -p $out
for w in 1k4c 1ppe 1azp-dna gso-1ppe gso-1k4c 2uuy; do
  n=$(echo $w | tr - _)
  timeout 600 python bench.py --workload $w > $out/${n}_bench.json 2> $out/${n}_bench.err
  tail -1 $out/${n}_bench.json | cut -c1-110
done
for i in 2 3; do timeout 300 python bench.py > $out/1k4c_bench_run$i.json 2>/dev/null; tail -1 $out/1k4c_bench_run$i.json | cut -c1-110; done
# separately labelled lines (never the headline): the synthetic table with bin 19 zeroed / with the membrane beads' rows zeroed; the headline complex at other batch sizes
timeout 300 python bench.py --zero-last-bin --cpu-seconds 0 > $out/1k4c_zero_last_bin.json 2>/dev/null; tail -1 $out/1k4c_zero_last_bin.json | cut -c1-110
timeout 300 python bench.py --zero-bead-rows --cpu-seconds 0 > $out/1k4c_zero_bead_rows.json 2>/dev/null; tail -1 $out/1k4c_zero_bead_rows.json | cut -c1-110
timeout 300 python bench.py --workload gso-1k4c --zero-bead-rows --cpu-seconds 0 > $out/gso_1k4c_zero_bead_rows.json 2>/dev/null; tail -1 $out/gso_1k4c_zero_bead_rows.json | cut -c1-110
for b in 4096 8192 16384 19456 32768 65536; do echo "1k4c, $b poses a launch: $(timeout 200 python3 bench.py --cpu-seconds 0 --batch $b 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f evals/s, step %.4f ms' % (d['value'], d['ms_per_step']))")"; done | tee $out/batch_sweep.txt
for w in 1k4c 1ppe 2uuy; do
  echo "== $w" > $out/bm_wave_times_$w.txt
  timeout 200 python3 tools/bm_wave_times.py --workload $w 2>&1 | grep -v amdgpu.ids >> $out/bm_wave_times_$w.txt
  cat $out/bm_wave_times_$w.txt
done
: > $out/gso_tail.txt
for live in 0 0.01 0.1 1; do
  echo "live share $live" >> $out/gso_tail.txt
  timeout 300 python3 tools/gso_tail.py 1024 60 $live 2>&1 | tail -3 >> $out/gso_tail.txt
done
cat $out/gso_tail.txt
for s in 1024 512 256 128; do
  timeout 400 python bench.py --workload gso-1ppe --swarms $s --steps 40 --warmup 6 --cpu-seconds 0 > $out/gso_1ppe_share_$s.json 2> $out/gso_1ppe_share_$s.err
  python -c "import json,sys; d=json.loads(open('$out/gso_1ppe_share_$s.json').read().strip().splitlines()[-1]); print('swarms %4d: %.2f M evals/s, %.3f ms per step, moved %.2f' % ($s, d['value']/1e6, d['ms_per_step'], d['config']['k1_k2_split']['moved_fraction']))" | tee -a $out/share.txt
done
timeout 900 python3 tools/fuzz_parity.py 200 1 anm > $out/fuzz_anm.txt 2>&1; tail -2 $out/fuzz_anm.txt
( cd tools/microbench && timeout 300 ./valu_rate ) > $out/valu_issue_rates.txt 2>&1
( cd tools/microbench && timeout 300 ./mfma_batch 2000 10 26 ) > $out/mfma_batch.txt 2>&1; cat $out/mfma_batch.txt
timeout 200 python3 tools/call_latency.py 2>&1 | grep -v amdgpu.ids > $out/call_latency.txt; cat $out/call_latency.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {  # name, env, args...
  name=$1; shift; envs=$1; shift
  env $envs timeout 200 python3 tools/bench_extra.py "$@" > $out/single_swarm_$name.json 2>&1
  ( export $envs; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$name -- python3 tools/bench_extra.py "$@" > /dev/null 2> $out/trace_$name.log )
  { echo "== $name ($envs): $(tail -1 $out/single_swarm_$name.json)"; python3 tools/step_timeline.py $out/trace_$name; } | tee $out/single_swarm_${name}_timeline.txt
  rm -rf $out/trace_$name
}
run 1k4c_bm "LD_X=1" --what gso1k4c --swarms 1 --steps 200
run 1k4c_packed "LIGHTDOCK_TILED_LATENCY=1" --what gso1k4c --swarms 1 --steps 200
run 1ppe_bm "LD_X=1" --what gso --swarms 1 --steps 200
run 1ppe_packed "LIGHTDOCK_TILED_LATENCY=1" --what gso --swarms 1 --steps 200
run 1ppe_128swarms "LD_X=1" --what gso --swarms 128 --steps 60
run 1ppe_1024swarms "LD_X=1" --what gso --swarms 1024 --steps 40
