# round 5, first GPU call: the GPU suite, the default bench line, the issue-rate microbenchmark with its clock stamps,
# batch sizes either side of the default, wave times of the pair kernel.  Usage (GPU box): bash tools/r5_first.sh <tag>
tag=${1:-r05a}
cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -5 $out/pytest.log
timeout 300 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.json
timeout 120 tools/microbench/valu_rate > $out/valu_rate.txt 2>&1; tail -3 $out/valu_rate.txt
for b in 4096 16384; do
  timeout 200 python bench.py --batch $b --cpu-seconds 0 --no-stats 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b: %.0f evals/s kernel %.3f ms' % (d['value'], d['roofline']['kernel_ms']))" | tee -a $out/batches.txt
done
timeout 200 python tools/bm_wave_times.py > $out/wave_times.txt 2>&1; cat $out/wave_times.txt
