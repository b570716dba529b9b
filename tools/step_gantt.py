"""Every launch of the last full step of a rocprofv3 kernel trace of bench.py: stream, start offset, duration, kernel, grid."""
import csv, glob, re, sys
d = sys.argv[1]
rows = sorted(csv.DictReader(open(glob.glob(f'{d}/*/*_kernel_trace.csv')[0])), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_adam(' in r['Kernel_Name'] and int(r['Grid_Size_X']) > 256 * 1000]
start, end = idx[-3] + 1, idx[-1] + 1
seg = rows[start:end]
t0 = int(seg[0]['Start_Timestamp'])
sid = {}
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    m = re.match(r'_ZN12_GLOBAL__N_1\d+(k_[a-z0-9_]+)I(.*?)EEv', n)
    if m:
        args = re.sub(r'Li(\d+)E', r'\1,', m.group(2)).replace('DF16b', 'bf16,').strip(',')
        return f"{m.group(1)}<{args}>"
    return n.split('(')[0][:40]
prev_end = {}
for r in seg:
    s = sid.setdefault(r['Stream_Id'], len(sid))
    b, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    gap = b - prev_end.get(s, b)
    prev_end[s] = e
    print(f"s{s} {b/1e3:8.1f} +{(e-b)/1e3:6.1f} gap{gap/1e3:6.1f}  {short(r['Kernel_Name']):44s} grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}")
print("step wall us", (int(seg[-1]['End_Timestamp']) - t0) / 1e3)
