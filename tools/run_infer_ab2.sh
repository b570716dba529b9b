#!/bin/bash
tag=$1; old=$2; out=gpurun_out/$tag; mkdir -p $out
one() { # label, env...
  python bench.py $BARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d.get('ms_per_step'))" | tee -a $out/ab.txt
}
for r in 1 2; do
  BARGS="--mode infer --batch 64 --no-cpu-baseline"
  PFA_LIB=$PWD/$old one "old infer"
  one "new infer"
  PFA_KNOBS=1=4096 one "new infer no-4img"
  BARGS="--precision fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-extras"
  PFA_LIB=$PWD/$old one "old fp32"
  one "new fp32"
done
