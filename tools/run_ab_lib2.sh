#!/bin/bash
# As run_ab_lib.sh, for the weight-gradient kernels too: kernel tests, per-layer tables (old .so through PFA_LIB, then the new
# build), bench.py A/B on three streams and on one.   usage: tools/run_ab_lib2.sh <tag> <old .so>
tag=$1; old=$2
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "patch2 or tconv or conv_forward or dgrad or wgrad" > $out/pytest.txt 2>&1
echo "pytest rc=$?" >> $out/pytest.txt
tail -3 $out/pytest.txt
grep -q "pytest rc=0" $out/pytest.txt || exit 1
PFA_LIB=$PWD/$old timeout -k 10 200 python tools/wgrad_patch_bench.py > $out/wgp_old.txt 2>&1
timeout -k 10 200 python tools/wgrad_patch_bench.py > $out/wgp_new.txt 2>&1
timeout -k 10 600 python tools/ab.py --rounds 3 "old:PFA_LIB=$PWD/$old" "new:PFA_X=1" > $out/ab3.txt 2>&1
tail -2 $out/ab3.txt
timeout -k 10 600 python tools/ab.py --rounds 2 --args "--no-overlap" "old:PFA_LIB=$PWD/$old" "new:PFA_X=1" > $out/ab1.txt 2>&1
tail -2 $out/ab1.txt
