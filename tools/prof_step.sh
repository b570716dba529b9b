#!/bin/bash
# Kernel traces of bench.py (three streams and one stream) -> summaries, stream timeline and a per-launch listing under
# gpurun_out/$1.  Run through gpurun from the repo root:  gpurun -- bash tools/prof_step.sh r2x
set -e
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/p3 $O/p1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p3 -- python $R/bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-roofline --no-extras > $O/p3.json 2> $O/p3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p1 -- python $R/bench.py --steps 5 --warmup 2 --repeats 1 --no-overlap --plan-shared --no-cpu-baseline --no-roofline --no-extras > $O/p1.json 2> $O/p1.err
cd $R
python tools/prof_summary.py $O/p3 7 > $O/p3_summary.txt
python tools/prof_summary.py $O/p1 7 > $O/p1_summary.txt
python tools/timeline.py $O/p3 > $O/p3_timeline.txt
python tools/step_gantt.py $O/p3 > $O/p3_gantt.txt
python tools/step_gantt.py $O/p1 > $O/p1_gantt.txt
cp $O/p3/*/*_kernel_stats.csv $O/p3_kernel_stats.csv
cp $O/p1/*/*_kernel_stats.csv $O/p1_kernel_stats.csv
rm -rf $O/p3 $O/p1
cat $O/p3_timeline.txt
head -34 $O/p1_summary.txt
