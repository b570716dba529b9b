import csv, glob, collections, sys
d = sys.argv[1]; steps = int(sys.argv[2])
f = glob.glob(f'{d}/*/*_kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel ms {tot/1e6:.2f}  per step {tot/1e6/steps:.3f}")
for r in rows[:16]:
    print(f"{r['Name'][:70]:70s} calls/step={int(r['Calls'])/steps:6.1f} ms/step={float(r['TotalDurationNs'])/1e6/steps:7.3f} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={float(r['Percentage']):5.1f}")
f = glob.glob(f'{d}/*/*_kernel_trace.csv')[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    name = r['Kernel_Name']
    short = None
    for key, s in (('gather_gemmIDF16bLi256ELi128','ring256'),('gather_gemmIDF16bLi128ELi128','gather128'),('gather_gemmIDF16bLi128ELi64','gather64'),('gather_gemmIDF16bLi256ELi16','gather16'),('k_wgradIDF16bLi128','wgrad128'),('k_wgradIDF16bLi16','wgrad16'),('splitk_finish','finish')):
        if key in name: short = s
    if not short: continue
    key = (short, int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']))
    agg.setdefault(key, []).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print()
for k,v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:28]:
    print(f"{k[0]:10s} blocks={k[1]:5d} n/step={len(v)/steps:4.1f} avg_us={sum(v)/len(v):7.1f} ms/step={sum(v)/1e3/steps:6.3f}")
