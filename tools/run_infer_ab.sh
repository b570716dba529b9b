#!/bin/bash
# Same-box A/B of two builds of the library on the inference bench (config 5) and the other shapes.  usage: tools/run_infer_ab.sh <tag> <old .so>
tag=$1; old=$2; out=gpurun_out/$tag; mkdir -p $out
for r in 1 2; do
  for arm in old new; do
    if [ $arm = old ]; then export PFA_LIB=$PWD/$old; else unset PFA_LIB; fi
    python bench.py --mode infer --batch 64 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$arm infer b64', d['value'])" | tee -a $out/infer_ab.txt
  done
done
for arm in old new; do
  if [ $arm = old ]; then export PFA_LIB=$PWD/$old; else unset PFA_LIB; fi
  python bench.py --size 512 --batch 8 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$arm 512 b8', d['value'], d['ms_per_step'])" | tee -a $out/infer_ab.txt
  python bench.py --precision fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$arm fp32', d['value'], d['ms_per_step'])" | tee -a $out/infer_ab.txt
  python bench.py --batch 64 --steps 10 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$arm b64', d['value'], d['ms_per_step'])" | tee -a $out/infer_ab.txt
done
