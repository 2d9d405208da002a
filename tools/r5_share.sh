# round 5, VERDICT r04 item 5: BASELINE config 5 (1ppe DFIRE, 1024 swarms x 200 glowworms over the GPUs of a node) as ONE GPU sees it at
# N = 1, 2, 4, 8: swarms never exchange data, so the N-GPU step time IS the step time of one GPU's share.  Bench lines of
# `--workload gso-1ppe --swarms 1024 / 512 / 256 / 128` on one device -> gpurun_out/r05_share/; tools/share_table.py folds them
# into the table of DESIGN section 7 (predicted, not measured on N devices).   usage (GPU box): bash tools/r5_share.sh
cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_share; mkdir -p $out
for s in 1024 512 256 128; do
  timeout 400 python bench.py --workload gso-1ppe --swarms $s --steps 40 --warmup 6 --cpu-seconds 0 > $out/gso_1ppe_share_$s.json 2> $out/gso_1ppe_share_$s.err
  python -c "import json,sys; d=json.loads(open('$out/gso_1ppe_share_$s.json').read().strip().splitlines()[-1]); print('swarms %4d: %.2f M evals/s, %.3f ms per step, moved %.2f' % ($s, d['value']/1e6, d['ms_per_step'], d['config']['k1_k2_split']['moved_fraction']))"
done
