cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pmc2
B="python3 bench.py --steps 3 --warmup 1 --batch 4096 --cpu-seconds 0"
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc2/p1 -- $B > /dev/null 2>gpurun_out/pmc2/e1.log
timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc2/p2 -- $B > /dev/null 2>gpurun_out/pmc2/e2.log
timeout 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d gpurun_out/pmc2/p3 -- $B > /dev/null 2>gpurun_out/pmc2/e3.log
timeout 240 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d gpurun_out/pmc2/p4 -- $B > /dev/null 2>gpurun_out/pmc2/e4.log
python3 - <<'P'
import csv,glob,collections
for p in sorted(glob.glob('gpurun_out/pmc2/p*/*/*counter_collection.csv')):
    rows=list(csv.DictReader(open(p)))
    agg=collections.defaultdict(list)
    for r in rows:
        if 'dfire_tiled_pairs<false>' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for c,v in agg.items(): print('%-32s n=%d mean=%.4g'%(c,len(v),sum(v)/len(v)))
    r=[r for r in rows if 'dfire_tiled_pairs<false>' in r['Kernel_Name']]
    if r: print('  vgpr',r[0]['VGPR_Count'],'sgpr',r[0]['SGPR_Count'],'lds',r[0]['LDS_Block_Size'],'wg',r[0]['Workgroup_Size'],'grid',r[0]['Grid_Size'])
P
