#!/bin/bash
# Register / LDS / spill figures of the DFIRE block-major kernels as the compiler reports them (no GPU needed).
# usage: bash tools/resource_usage.sh [extra hipcc flags]
cd "$(dirname "$0")/../lightdock-rust_amd" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -I../include -Icsrc "$@" \
  -Rpass-analysis=kernel-resource-usage -c csrc/kernels/dfire_bm.hip -o /tmp/ru.o 2>&1 \
  | grep -E "remark:" | sed -e 's/.*remark: //' -e 's/\[-Rpass-analysis=kernel-resource-usage\]//' | tr '\n' ' ' | sed -e 's/Function Name:/\n/g' | awk '{$1=$1};1'
