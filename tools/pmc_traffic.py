"""Per-launch HBM traffic of every MFMA kernel family of the step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

gfx950 corrections (MI355X_MICROARCH.md, HBM): the counters are in KiB; FETCH_SIZE reports exactly half of the bytes of
a wide coalesced read stream -> bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> <workload key> [commit]
The output is what bench.py reads as roofline.traffic (profiles/r02_traffic.json): kernels keyed by the names the library
reports (pfa_last_gather_kernel / the timing store)."""
import csv, datetime, glob, json, re, sys


def display(name):
    """rocprofv3 kernel name (mangled or demangled) -> the library's display name, or None for non-MFMA kernels."""
    def ints(pat):
        m = re.search(pat, name)
        return [int(x) for x in re.findall(r"-?\d+", m.group(1))] if m else None
    if "k_conv_patch2" in name:
        a = ints(r"k_conv_patch2(?:I|<)([^>E]*(?:E[^>E]*)*)") or ints(r"k_conv_patch2<([^>]*)>")
        a = [x for x in (re.findall(r"(?:Li|<|, ?)(\d+)", name.split("k_conv_patch2")[1][:40]))]
        a = [int(x) for x in a[:3]]
        return f"k_conv_patch2<{'conv_s2' if a[0] else 'tconv'},{a[1]},{('8x32', '16x16', '4x8x8')[a[2]]}>"
    if "k_conv_s1_patch32ILi64" in name or "k_conv_s1_patch32<64" in name:
        return "k_conv_s1_patch<bf16,8x32,64>"
    if "k_conv_s1_inplaceILi64E" in name or "k_conv_s1_inplace<64>" in name:
        return "k_conv_s1_inplace<bf16,8x32,64>"
    if "k_conv_s1_inplace" in name:
        return "k_conv_s1_inplace<bf16,8x32,128>"
    if "k_conv_s1_patch" in name:                                     # (the earlier forms of the 128-channel stride-1 patch kernel)
        return "k_conv_s1_patch<bf16,8x32,128>"
    if "k_tconv4" in name:
        return "k_tconv4<64,8x32>"
    if "k_wgrad_patch_reduce" in name or "k_wgrad_reduce" in name or "k_wgrad_image_reduce" in name:
        return None
    if "k_wgrad_image" in name:
        a = [int(x) for x in re.findall(r"(?:Li|<|, ?)(\d+)", name.split("k_wgrad_image")[1][:20])[:2]]
        return f"k_wgrad_image<{a[0]},{'2x8ch' if a[1] == 2 else '8ch'}>"
    if "k_wgrad_patch" in name:
        a = [int(x) for x in re.findall(r"(?:Li|<|, ?)(\d+)", name.split("k_wgrad_patch")[1][:30])[:2]]
        return f"k_wgrad_patch<s{a[0]},{'2x32' if a[1] == 32 else '4x16'}>"
    if "k_conv_image" in name:
        a = [int(x) for x in re.findall(r"(?:Li|<|, ?)(\d+)", name.split("k_conv_image")[1][:20])[:1]]
        return "k_conv_image<2x8ch,64>" if a[0] == 2 else "k_conv_image<8ch,64>"
    if "k_wgrad" in name:
        tail = name.split("k_wgrad")[1][:60]
        bf = "DF16b" in tail or "__bf16" in tail or "bf16" in tail
        a = [int(x) for x in re.findall(r"(?:Li|, ?)(\d+)", tail)]
        if not bf or not a:
            return None
        if a[0] == 16:
            return "k_wgrad<bf16,16>"
        if a[0] == 256:
            return "k_wgrad<bf16,256,ring>"
        rows = ",rows" if ("Lb1" in tail or "true" in tail) else ""          # the row-aligned staging (FAST)
        return f"k_wgrad<bf16,128,kg2{rows}>" if len(a) >= 5 and a[4] == 2 else f"k_wgrad<bf16,128{rows}>"
    if "k_gather_gemm" in name:
        tail = name.split("k_gather_gemm")[1][:70]
        a = [int(x) for x in re.findall(r"(?:Li|, ?)(\d+)", tail)][:5]
        if len(a) < 5 or not ("DF16b" in tail or "bf16" in tail):
            return None
        return f"k_gather_gemm<bf16,{a[0]},{a[1]},{a[2]},{a[3]},{a[4]}>"
    return None


def per_kernel(d, counter):
    f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = display(r["Kernel_Name"])
        if k is None:
            continue
        per.setdefault(k, {}).setdefault(r["Dispatch_Id"], 0.0)
        per[k][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: list(v.values()) for k, v in per.items()}


if __name__ == "__main__":
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out_path, key = sys.argv[3], sys.argv[4]
    commit = sys.argv[5] if len(sys.argv) > 5 else "?"
    kernels = {}
    for k in sorted(set(fetch) & set(write)):
        f, w = fetch[k], write[k]
        kernels[k] = dict(launches=len(f), fetch_kib_avg=round(sum(f) / len(f), 1), write_kib_avg=round(sum(w) / len(w), 1),
                          traffic_bytes_per_launch=round((2 * sum(f) / len(f) + sum(w) / len(w)) * 1024))
    try:
        out = json.load(open(out_path))
    except (OSError, ValueError):
        out = {}
    out.update(commit=commit, date=datetime.date.today().isoformat(),
               note="bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request); averages "
                    "over every launch of the kernel in 4 single-stream steps",
               command="rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --steps 3 --warmup 1 --repeats 1 "
                       "--no-overlap --no-cpu-baseline --no-roofline --no-extras")
    out.setdefault("workloads", {})[key] = kernels
    json.dump(out, open(out_path, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k:40s} {v['launches']:4d} launches  {v['traffic_bytes_per_launch'] / 1e6:8.1f} MB / launch")
