"""Per-launch fabric/HBM traffic (L2 misses) vs duration of the MFMA kernels inside one training step.
usage: step_traffic.py <dir of a `rocprofv3 --kernel-trace --pmc FETCH_SIZE` run> <dir of a `--pmc WRITE_SIZE` run>
(both of `python bench.py --steps 3 --warmup 1 --no-overlap --no-cpu-baseline --no-roofline --no-extras`).
bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 corrections, MI355X_MICROARCH.md)."""
import collections, csv, glob, sys


def load(d, counter):
    f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
    out = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = int(r["Dispatch_Id"])
        if k not in out:
            out[k] = [r["Kernel_Name"], int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1), 0.0]
        out[k][2] += float(r["Counter_Value"])
    return out


F, W = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
tr = glob.glob(f"{sys.argv[1]}/*/*_kernel_trace.csv")[0]
dur = {int(r["Dispatch_Id"]): (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(tr))}
ids, idsw = list(F), list(W)
adam, adamw = [i for i in ids if "k_adam" in F[i][0]], [i for i in idsw if "k_adam" in W[i][0]]
fw = [i for i in ids if adam[-3] < i <= adam[-1]]
ww = [i for i in idsw if adamw[-3] < i <= adamw[-1]]
assert len(fw) == len(ww)
tot, tt = collections.Counter(), collections.Counter()
print("kernel    blocks      us      MB    GB/s")
for a, b in zip(fw, ww):
    name = F[a][0]
    short = ("ring" if "Li256ELi128" in name else "g128" if "Li128ELi128" in name else "g64" if "Li128ELi64" in name else "g16" if "Li256ELi16" in name else
             "wgrad" if "k_wgradI" in name else "scatter" if "scatter" in name else None)
    mb = (2 * F[a][2] + W[b][2]) * 1024 / 1e6
    key = short or name.split("(")[0][-28:]
    tot[key] += mb
    tt[key] += dur[a]
    if short:
        print(f"{short:8s} {F[a][1]:6d} {dur[a]:7.1f} {mb:7.1f} {mb / dur[a] * 1e3:7.0f}")
print()
for k, v in tot.most_common(14):
    print(f"{k:32s} {v:8.1f} MB {tt[k]:8.1f} us  {v / tt[k] * 1e3:7.0f} GB/s")
print("total MB/step", round(sum(tot.values()), 1), "kernel us/step", round(sum(tt.values()), 1))
