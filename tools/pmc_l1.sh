# L1 (TCP) vs L2 (TCC) request counters of the tiled kernel (own --pmc passes, no tracing)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pmc_l1
B="python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0"
timeout 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d gpurun_out/pmc_l1/p1 -- $B > /dev/null 2>gpurun_out/pmc_l1/e1.log
timeout 200 rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d gpurun_out/pmc_l1/p2 -- $B > /dev/null 2>gpurun_out/pmc_l1/e2.log
python3 - <<'P'
import csv,glob,collections
for p in sorted(glob.glob('gpurun_out/pmc_l1/p*/*/*counter_collection.csv')):
    rows=list(csv.DictReader(open(p)))
    agg=collections.defaultdict(list)
    for r in rows:
        if 'dfire_tiled_pairs<false' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for c,v in agg.items(): print('%-40s n=%d mean=%.5g'%(c,len(v),sum(v)/len(v)))
P
tail -3 gpurun_out/pmc_l1/e*.log | grep -i "error\|invalid" | head
