#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace of the per-config table (tools/config_table.py),
# summary into profiles/<tag>_configs_kernel_stats.csv + the table itself as JSON.
set -u
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python3 -c 'import __graft_entry__ as g; g.build_hip()'
OUT=gpurun_out/prof_cfg_$TAG
rm -rf "$OUT"; mkdir -p "$OUT" profiles
# the table itself from an unprofiled run (the tracer adds ~25 us to a 20 us kernel), the per-kernel durations from a traced one
python3 tools/config_table.py profiles/${TAG}_config_table.json > $OUT/table.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/trace -- python3 tools/config_table.py > $OUT/table_traced.log 2>&1
T=$(find $OUT/trace -name '*results.db' | head -1)
python3 tools/rocprof_summary.py stats "$T" profiles/${TAG}_configs_kernel_stats.csv
grep '^|' $OUT/table.log > profiles/${TAG}_config_table.md
cp profiles/${TAG}_config* gpurun_out/ 2>/dev/null
rm -rf "$OUT"
cat profiles/${TAG}_config_table.md
head -12 profiles/${TAG}_configs_kernel_stats.csv
