"""Every launch of the last full step of a rocprofv3 kernel trace of bench.py: stream, start offset, duration, kernel, grid."""
import csv, glob, re, sys
d = sys.argv[1]
rows = sorted(csv.DictReader(open(glob.glob(f'{d}/*/*_kernel_trace.csv')[0])), key=lambda r: int(r['Start_Timestamp']))
# one step = from one k_l1_tanh_bwd launch (exactly one per step) to the next: a full cycle of the schedule, starting mid-step
idx = [i for i, r in enumerate(rows) if 'l1_tanh_bwd' in r['Kernel_Name']]
start, end = idx[-2], idx[-1]
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
# per-stream busy time and the union of busy intervals (GPU time with at least one kernel running)
iv = sorted((int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0) for r in seg)
busy, cur_s, cur_e = 0, None, None
for b, e in iv:
    if cur_e is None or b > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = b, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
per = {}
for r in seg:
    s_ = sid[r['Stream_Id']]
    per[s_] = per.get(s_, 0) + int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print("launches", len(seg), "sum of durations us", sum(per.values()) / 1e3, "per stream us", {k: round(v / 1e3, 1) for k, v in sorted(per.items())},
      "at least one kernel running us", busy / 1e3)
