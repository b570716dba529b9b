#!/bin/bash
# rocprofv3 kernel summary of the inference configuration (BASELINE config 5), run through gpurun from the repo root
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_infer
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_infer -- python $R/bench.py --mode infer --batch ${INFER_BATCH:-16} --steps 5 --warmup 2 > $R/gpurun_out/prof_infer.json 2> $R/gpurun_out/prof_infer.err
cd $R
cp gpurun_out/prof_infer/*/*_kernel_stats.csv gpurun_out/prof_infer_kernel_stats.csv
rm -rf gpurun_out/prof_infer
python bench.py --mode infer --batch ${INFER_BATCH:-16} | cut -c1-400
