#!/bin/bash
# round 6: what binds the culling kernel -- timing build without its returning atomics on the tile pairs' entry counters (wrong lists)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_19; mkdir -p $O
bash tools/trace_variants.sh > $O/trace.txt 2>&1; grep -E "^==|cull<false|pairs<false" $O/trace.txt
bash tools/trace_variants.sh --workload 1ppe > $O/trace_1ppe.txt 2>&1; grep -E "^==|cull<false|pairs<false" $O/trace_1ppe.txt
