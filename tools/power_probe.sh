#!/bin/bash
# Clock and power of the GPU while bench.py runs (three streams, then one stream): is the step power-bound?
# gpurun -- bash tools/power_probe.sh <tag>
TAG=${1:-power}
O=gpurun_out/$TAG; mkdir -p $O
probe() {   # label, bench args
  python bench.py --steps 1500 --warmup 20 --repeats 1 --no-cpu-baseline --no-roofline --no-extras $2 > $O/$1.json 2>/dev/null &
  BP=$!
  sleep 4
  for i in $(seq 1 12); do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|edge)" | tr '\n' ';' ; echo; sleep 0.25; done > $O/$1.smi
  wait $BP
}
probe three ""
probe one "--no-overlap"
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ';' > $O/idle.smi
for f in three one; do echo "== $f: $(python -c "import json;d=json.load(open('$O/$f.json'));print(d['ms_per_step'],'ms/step')")"; head -4 $O/$f.smi | cut -c1-400; done
echo "== idle"; cat $O/idle.smi | cut -c1-300
