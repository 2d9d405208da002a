# round 5, the final measurement set on ONE box, with the committed profiles/traffic.json of these sources in place:
#   the default bench line of every workload, the pair kernel's wave timers (1k4c, 1ppe, 2uuy), the GSO step when nothing / 1 % /
#   10 % / everything moves, and config 5 as one GPU's share at N = 1, 2, 4, 8.   -> gpurun_out/r05_final/
# usage (GPU box): bash tools/r5_final.sh
cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_final; mkdir -p $out
for w in 1k4c 1ppe 1azp-dna gso-1ppe gso-1k4c 2uuy; do
  n=$(echo $w | tr - _)
  timeout 600 python bench.py --workload $w > $out/${n}_bench.json 2> $out/${n}_bench.err
  tail -1 $out/${n}_bench.json | cut -c1-110
done
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
