#!/bin/bash
# In-situ A/B of two BUILDS of the library (base = the old .so given as $2, through PFA_LIB; alt = the current build):
# -> per-kernel average durations side by side.   gpurun -- bash tools/insitu_ab.sh <tag> <knobs, e.g. 3=6>
set -e
TAG=${1:-ab}; KN=${2:-3=6}; BARGS=${3:-"--steps 5 --warmup 2 --repeats 1 --no-overlap --no-cpu-baseline --no-roofline --no-extras"}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for arm in base alt; do
  rm -rf $O/$arm
  if [ $arm = base ]; then export PFA_LIB=$R/$KN; else unset PFA_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$arm -- python $R/bench.py $BARGS > $O/$arm.json 2> $O/$arm.err
  cp $O/$arm/*/*_kernel_stats.csv $O/${arm}_kernel_stats.csv
  rm -rf $O/$arm
done
cd $R
python - "$O" "$KN" <<'PY'
import csv, sys
O, KN = sys.argv[1], sys.argv[2]
def load(f):
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs'])) for r in csv.DictReader(open(f))}
a, b = load(O + '/base_kernel_stats.csv'), load(O + '/alt_kernel_stats.csv')
print(f"whole run (us), old library {KN} vs the new build")
ta = sum(v[1] for v in a.values()) / 1e3; tb = sum(v[1] for v in b.values()) / 1e3
print(f"all kernels: {ta:8.1f} us   vs {tb:8.1f} us")
for n in sorted(set(a) | set(b), key=lambda n: -(a.get(n, (0, 0))[1] + b.get(n, (0, 0))[1])):
    ca, da = a.get(n, (0, 0)); cb, db = b.get(n, (0, 0))
    if abs(da - db) / 1e3 < 3 and ca == cb: continue
    print(f"  {n[:80]:80s} {ca:5.0f} x {da/max(ca,1)/1e3:7.1f} us = {da/7e3:7.1f}   |  {cb:5.0f} x {db/max(cb,1)/1e3:7.1f} us = {db/7e3:7.1f}")
PY
