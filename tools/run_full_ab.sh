#!/bin/bash
# The whole GPU test suite, then the same-box A/B of the current build against an older build of the library (PFA_LIB):
# per-layer tables, bench.py through launch lists (GPU-bound), eager three-stream, one stream.
#   usage: tools/run_full_ab.sh <tag> <old .so>
tag=$1; old=$2
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest.txt 2>&1
echo "pytest rc=$?" >> $out/pytest.txt
tail -3 $out/pytest.txt
grep -q "pytest rc=0" $out/pytest.txt || exit 1
PFA_LIB=$PWD/$old timeout -k 10 200 python tools/patch2_bench.py > $out/patch2_old.txt 2>&1
timeout -k 10 200 python tools/patch2_bench.py > $out/patch2_new.txt 2>&1
PFA_LIB=$PWD/$old timeout -k 10 200 python tools/wgrad_patch_bench.py > $out/wgp_old.txt 2>&1
timeout -k 10 200 python tools/wgrad_patch_bench.py > $out/wgp_new.txt 2>&1
timeout -k 10 600 python tools/ab.py --rounds 3 --args=--replay "old:PFA_LIB=$PWD/$old" "new:PFA_X=1" > $out/ab_replay.txt 2>&1
tail -2 $out/ab_replay.txt
timeout -k 10 600 python tools/ab.py --rounds 3 "old:PFA_LIB=$PWD/$old" "new:PFA_X=1" > $out/ab_eager.txt 2>&1
tail -2 $out/ab_eager.txt
timeout -k 10 600 python tools/ab.py --rounds 2 --args=--no-overlap "old:PFA_LIB=$PWD/$old" "new:PFA_X=1" > $out/ab_one.txt 2>&1
tail -2 $out/ab_one.txt
