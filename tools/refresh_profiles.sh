#!/bin/bash
# Profile artefacts -> gpurun_out/$TAG (copy what should be judged into profiles/): run through gpurun from the repo
# root:  gpurun -- bash tools/refresh_profiles.sh r2final <commit>
#   p3_* / p1_*      rocprofv3 --kernel-trace --stats of bench.py, three streams (default) / one stream (the mode of the
#                    bench's live roofline measurement): summaries, kernel stats csv, per-launch listing
#   traffic.json     HBM bytes per launch of every MFMA kernel family (two --pmc passes, kernel trace only)
#   bench_line.json  the JSON line of a default `python bench.py` run of the same call (reads traffic.json's copy under
#                    profiles/ if present)
set -e
TAG=${1:-prof}
COMMIT=${2:-?}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
bash $R/tools/prof_step.sh $TAG > $O/prof_step.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_fetch $O/pmc_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 3 --warmup 1 --repeats 1 --no-overlap --plan-shared --no-cpu-baseline --no-roofline --no-extras > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 3 --warmup 1 --repeats 1 --no-overlap --plan-shared --no-cpu-baseline --no-roofline --no-extras > /dev/null 2> $O/pmc_write.err
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/traffic.json train_b16_256_bf16 $COMMIT > $O/traffic.txt
rm -rf $O/pmc_fetch $O/pmc_write
mkdir -p profiles && cp $O/traffic.json profiles/r04_traffic.json      # (in the box's copy: the bench line below reads it)
python bench.py > $O/bench_line.json 2> $O/bench_line.err
python bench.py --mode infer --batch 64 > $O/bench_infer_b64.json 2> $O/bench_infer.err
cat $O/traffic.txt
head -c 600 $O/bench_line.json
