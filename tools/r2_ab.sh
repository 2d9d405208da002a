cd $GRAFT_REPO_ROOT
L=lightdock-rust_amd/lib
cp $L/liblightdock_hip.so /tmp/keep.so
for v in base dna1; do
  cp $L/variants/$v.so $L/liblightdock_hip.so
  echo "== $v"; timeout 200 python bench.py --workload 1azp-dna --cpu-seconds 20 --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f evals/s kernel %.3f ms parity %.3e' % (d['value'], d['roofline']['kernel_ms'], d['parity_max_rel_err_vs_cpu_sample']))"
  timeout 300 python -m pytest tests -m gpu -x -q -k "dna or 1azp or pydock" 2>&1 | tail -2
done
cp /tmp/keep.so $L/liblightdock_hip.so
