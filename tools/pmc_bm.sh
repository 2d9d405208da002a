# SQ / LDS counters of the block-major DFIRE kernels (own --pmc passes, no tracing).
# Usage (GPU box): bash tools/pmc_bm.sh <tag> [bench args]   -> gpurun_out/pmc_<tag>/
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/pmc_$tag; mkdir -p $out
B="python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 $*"
i=0
while read -r set; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- $B > /dev/null 2> $out/e$i.log
done <<'SETS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS
SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_FLAT
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
GRBM_GUI_ACTIVE GRBM_COUNT
SETS
python3 - $out <<'P'
import csv,glob,collections,sys
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for p in sorted(glob.glob(out+'/p*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(p)):
        k=r['Kernel_Name']
        for short in ('dfire_bm_pairs<false','dfire_bm_cull<false','dfire_bm_gather','dfire_packed_pairs<false'):
            if short in k: agg[short][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    for c,v in sorted(d.items()): print('%-26s %-34s n=%d mean=%.5g'%(k,c,len(v),sum(v)/len(v)))
P
grep -il "error\|invalid\|not found" $out/e*.log | head
