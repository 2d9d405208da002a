#!/bin/bash
# A/B of prebuilt library variants on ONE box WITHOUT touching the installed library: every run loads
# lightdock-rust_amd/lib/variants/<name>.so through LIGHTDOCK_HIP_VARIANT (lightdock-rust_amd/__init__.py).
# Usage (on the GPU box): bash tools/ab6.sh <rounds> [bench args...]      -- interleaves the variants <rounds> times; every run
# under its own `timeout`.  "base" = the installed library.
set -u
shopt -s nullglob
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
rounds=${1:-2}; shift
L=lightdock-rust_amd/lib
line() { python3 -c "import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f evals/s  step %.4f (min %.4f med %.4f) ms  kernel %.4f ms' % (d['value'], d['ms_per_step'], d.get('ms_per_step_min',0), d.get('ms_per_step_median',0), d['roofline']['kernel_ms']))
except Exception as e: print('FAILED', e)"; }
for round in $(seq 1 "$rounds"); do
  echo "base $(timeout 120 python3 bench.py --cpu-seconds 0 "$@" 2>&1 | line)"
  for v in $L/variants/*.so; do
    n=$(basename "$v" .so)
    echo "$n $(LIGHTDOCK_HIP_VARIANT=$n timeout 120 python3 bench.py --cpu-seconds 0 "$@" 2>&1 | line)"
  done
done
