# round 5: A/B of the prebuilt library variants (lightdock-rust_amd/lib/variants/*.so) on one box: the parity subset on the installed
# library first, then bench lines and pair-kernel wave times per variant.  Usage (GPU box): bash tools/r5_ab.sh <tag> [pytest -k expr]
tag=${1:-r05ab}; kexpr=${2:-"pose_energies or kernel_variants or gso_steps or full_size or frame_edges"}
cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$kexpr" > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -3 $out/pytest.log
bash tools/ab.sh 2>&1 | tee $out/ab.txt
bash tools/ab_wave_times.sh 2>&1 | tee $out/wave_times.txt
