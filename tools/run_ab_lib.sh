#!/bin/bash
# Same-box A/B of two BUILDS of the library: the patch kernels' tests, the per-layer table of tools/patch2_bench.py with the
# old library (PFA_LIB) and the new one, then bench.py through tools/ab.py.   usage: tools/run_ab_lib.sh <tag> <old .so> [pytest -k expression]
tag=$1; old=$2; kexpr=${3:-"patch2 or tconv or conv_forward or dgrad"}
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "$kexpr" > $out/pytest.txt 2>&1
echo "pytest rc=$?" >> $out/pytest.txt
tail -3 $out/pytest.txt
grep -q "pytest rc=0" $out/pytest.txt || exit 1
PFA_LIB=$PWD/$old timeout -k 10 200 python tools/patch2_bench.py > $out/patch2_old.txt 2>&1
timeout -k 10 200 python tools/patch2_bench.py > $out/patch2_new.txt 2>&1
timeout -k 10 600 python tools/ab.py --rounds 3 "old:PFA_LIB=$PWD/$old" "new:PFA_X=1" > $out/ab.txt 2>&1
tail -4 $out/ab.txt
