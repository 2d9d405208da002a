# round 5: part-cap sweep with the installed library, then the pair kernel's wave timers at two caps
cd $GRAFT_REPO_ROOT
bash tools/r5_partcap.sh ${1:-1k4c} 1792 1536 1280 1024 768
for cap in 1792 1024; do
  echo "== wave timers, part cap $cap"
  LIGHTDOCK_BM_PART_CAP=$cap python3 tools/bm_wave_times.py --workload ${1:-1k4c} 2>&1 | tail -9
done
