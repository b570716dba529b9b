#!/bin/bash
# HBM bytes per launch of the dominant kernel from two separate rocprofv3 --pmc passes (no tracing domains besides
# --kernel-trace): run through gpurun from the repo root; writes gpurun_out/traffic_gather_ring.json
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-overlap --no-cpu-baseline --no-roofline --no-extras > /dev/null 2> $R/gpurun_out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python $R/bench.py --steps 3 --warmup 1 --no-overlap --no-cpu-baseline --no-roofline --no-extras > /dev/null 2> $R/gpurun_out/pmc_write.err
cd $R
python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/traffic_gather_ring.json
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
