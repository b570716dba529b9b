#!/bin/bash
# Regenerates the rocprofv3 summaries under gpurun_out/ (copy what should be judged into profiles/): run through gpurun from
# the repo root.  One trace of the default three-stream bench, one of the one-stream mode bench.py's roofline is measured in.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_eager $R/gpurun_out/prof_one
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_eager -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_eager.json 2> $R/gpurun_out/prof_eager.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_one -- python $R/bench.py --steps 5 --warmup 2 --no-overlap --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_one.json 2> $R/gpurun_out/prof_one.err
cd $R
python tools/prof_summary.py gpurun_out/prof_eager 7 > gpurun_out/prof_eager_summary.txt
python tools/prof_summary.py gpurun_out/prof_one 7 > gpurun_out/prof_one_summary.txt
cp gpurun_out/prof_eager/*/*_kernel_stats.csv gpurun_out/prof_eager_kernel_stats.csv
cp gpurun_out/prof_one/*/*_kernel_stats.csv gpurun_out/prof_one_kernel_stats.csv
rm -rf gpurun_out/prof_eager gpurun_out/prof_one          # the raw traces exceed what gpurun merges back
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
cat gpurun_out/bench_final.json
