# round 5: kernel trace of the 2uuy bench (block-major ANM form), and the wave timers of its pair kernel
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r05anm/trace; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python3 bench.py --workload 2uuy --steps 10 --warmup 3 --cpu-seconds 0 --no-stats > $out/bench_traced.json 2> $out/trace.log
f=$(find $out/t -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("  %-60s calls %4s avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
LIGHTDOCK_BM_DEBUG=$out/waves.txt timeout 120 python bench.py --workload 2uuy --steps 2 --warmup 1 --cpu-seconds 0 --no-stats > /dev/null 2>&1
python3 tools/bm_wave_times.py $out/waves.txt 2>&1 | tail -25
