#!/bin/bash
# SQ / LDS counters of single launches of the patch kernels -> gpurun_out/$1/pmc_patch2.txt (gpurun -- bash tools/pmc_patch2.sh tag)
set -e
TAG=${1:-pmc}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  rm -rf $O/pass$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pass$i -- python $R/tools/pmc_patch2.py > /dev/null 2> $O/pass$i.err || echo "pass $i failed: $set"
done
cd $R
python - "$O" <<'PY' > $O/pmc_patch2.txt
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.OrderedDict()
for d in sorted(glob.glob(O + '/pass*/')):
    fs = glob.glob(d + '*/*counter_collection.csv')
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        k = (r['Kernel_Name'][:60], r['Dispatch_Id'] if 'Dispatch_Id' in r else r.get('Correlation_Id'))
        name = r['Kernel_Name']
        if 'k_conv' not in name and 'k_wgrad' not in name and 'k_tconv4' not in name: continue
        acc.setdefault(name[:70], collections.OrderedDict()).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k)
    for c, v in cs.items():
        v = sorted(v)
        print(f"    {c:28s} median {v[len(v)//2]:14.0f}  ({len(v)} launches)")
PY
cat $O/pmc_patch2.txt
