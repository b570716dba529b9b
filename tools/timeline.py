"""Stream-level view of a rocprofv3 kernel trace of bench.py: busy time per stream, union busy time, gaps."""
import csv, glob, sys
d = sys.argv[1]
rows = sorted(csv.DictReader(open(glob.glob(f'{d}/*/*_kernel_trace.csv')[0])), key=lambda r: int(r['Start_Timestamp']))
# last step = from behind the last Adam launch of the step before to the last Adam launch of the run (the bench runs
# STEPS steps in all: warm-up + timed; every step has the same number of Adam launches)
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 7
idx = [i for i, r in enumerate(rows) if 'k_adam(' in r['Kernel_Name']]
per = max(1, len(idx) // STEPS)
start, end = idx[-per - 1] + 1, idx[-1] + 1
seg = rows[start:end]
t0, t1 = int(seg[0]['Start_Timestamp']), int(seg[-1]['End_Timestamp'])
print(f"step wall {(t1 - t0) / 1e3:.1f} us, {len(seg)} launches")
streams = {}
for r in seg:
    streams.setdefault(r['Stream_Id'], []).append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
for s, ks in streams.items():
    busy = sum(e - b for b, e, _ in ks)
    print(f"stream {s}: {len(ks):4d} kernels, busy {busy / 1e3:8.1f} us ({100 * busy / (t1 - t0):.0f}% of the step)")
ev = sorted((b, e) for r in seg for b, e in [(int(r['Start_Timestamp']), int(r['End_Timestamp']))])
union = 0; cur_b, cur_e = ev[0]
gaps = []
for b, e in ev[1:]:
    if b > cur_e:
        union += cur_e - cur_b; gaps.append((b - cur_e, cur_e - t0)); cur_b, cur_e = b, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_b
print(f"GPU busy (union) {union / 1e3:.1f} us = {100 * union / (t1 - t0):.1f}% ; idle {(t1 - t0 - union) / 1e3:.1f} us in {len(gaps)} gaps")
print("largest gaps (us @ offset):", [(round(g / 1e3, 1), round(o / 1e3)) for g, o in sorted(gaps, reverse=True)[:10]])
tot = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
print(f"sum of kernel durations {tot / 1e3:.1f} us -> average concurrency {tot / union:.2f}")
