#!/bin/bash
# Kernel-trace A/B of the THREE-STREAM step under two environments: which kernels take longer, which shorter, inside the step.
#   gpurun -- bash tools/insitu_ab_env.sh <tag> "<ENV=VAL ... of arm a>" "<... of arm b>" [bench args]
TAG=${1:-abenv}; EA=$2; EB=$3; BARGS=${4:-"--replay --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-roofline --no-extras"}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for arm in a b a2 b2; do
  rm -rf $O/$arm
  case $arm in a|a2) E=$EA;; *) E=$EB;; esac
  ( for kv in $E; do export "$kv"; done
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/$arm -- python $R/bench.py $BARGS > $O/$arm.json 2> $O/$arm.err )
  cp $O/$arm/*/*_kernel_stats.csv $O/${arm}_kernel_stats.csv
  rm -rf $O/$arm
done
cd $R
python - "$O" <<'PY'
import csv, sys, re
O = sys.argv[1]
def load(f):
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs'])) for r in csv.DictReader(open(f))}
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*', '', n)[:58]
A = [load(O + f'/{x}_kernel_stats.csv') for x in ('a', 'a2')]
B = [load(O + f'/{x}_kernel_stats.csv') for x in ('b', 'b2')]
steps = 25.0
def tot(d): return sum(v[1] for v in d.values()) / 1e3 / steps
print(f"kernel time per step (us): a {tot(A[0]):8.1f} {tot(A[1]):8.1f}   b {tot(B[0]):8.1f} {tot(B[1]):8.1f}")
names = set().union(*A, *B)
rows = []
for n in names:
    a = [d.get(n, (0, 0))[1] / 1e3 / steps for d in A]; b = [d.get(n, (0, 0))[1] / 1e3 / steps for d in B]
    rows.append((sum(b) / 2 - sum(a) / 2, n, a, b))
for dlt, n, a, b in sorted(rows, key=lambda r: -abs(r[0]))[:24]:
    print(f"  {short(n):58s} a {a[0]:7.1f} {a[1]:7.1f}   b {b[0]:7.1f} {b[1]:7.1f}   b - a {dlt:+7.1f} us/step")
PY
