"""Per-launch HBM traffic of the dominant kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

gfx950 corrections (MI355X_MICROARCH.md, HBM): the counters are in KiB; FETCH_SIZE reports exactly half of the bytes of
a wide coalesced read stream -> bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, sys

KEY = "gather_gemmIDF16bLi256ELi128ELi4ELi2ELi3"          # k_gather_gemm<bf16,256,128,4,2,3>: the 3-stage ring


def per_launch(d, counter):
    f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
    vals = {}
    for r in csv.DictReader(open(f)):
        if KEY in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals[r["Dispatch_Id"]] = vals.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return list(vals.values())


fetch = per_launch(sys.argv[1], "FETCH_SIZE")
write = per_launch(sys.argv[2], "WRITE_SIZE")
out = dict(kernel="k_gather_gemm<bf16,256,128,4,2,3>", launches=len(fetch), fetch_kib_avg=sum(fetch) / len(fetch),
           write_kib_avg=sum(write) / len(write),
           traffic_bytes_per_launch=(2 * sum(fetch) / len(fetch) + sum(write) / len(write)) * 1024,
           note="bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)",
           command="rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --steps 3 --warmup 1 --no-overlap "
                   "--no-cpu-baseline --no-roofline")
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(out)
