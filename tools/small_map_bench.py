#!/usr/bin/env python3
"""The small-map launches of the default generator (M <= 1024 rows: e5-e7, d7-d5 forward and four of their input gradients)
alone, each launch six times, under a kernel trace:

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -- python tools/small_map_bench.py run [warm|cold] [knob1] [knob0] [slices x4]
    python tools/small_map_bench.py rows OUT

run:  warm = the same launch repeated, cold = a 1 GB fill between launches; knob1 = ablation bits (pfa_set_knob(1, .)),
      knob0 = tile plan (pfa_set_knob(0, .): 0 default, 7 no ring tiles, 6 128x64 everywhere); the last argument scales the
      slice count the engine would choose by value / 4 (8 = twice, 2 = half).
rows: per layer, the average duration of the MFMA launch and of its slice-combining pass (the last four of six repeats).
DESIGN.md section 6 quotes its numbers."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(argv):
    sys.path.insert(0, os.path.join(ROOT, "gan-aug-pfa_amd"))
    import torch
    from pfa_gan import kernels as K, _lib
    B, dt = 16, torch.bfloat16
    ws = K.Workspace(torch.device("cuda"))

    def T(*shape):
        return torch.randn(shape, device="cuda", dtype=torch.float32).to(dt)
    # name, mode, source (h, w, channel parts), output channels, position grid, output size
    layers = [
        ("e5 fwd", K.CONV_S2, (16, 16, [512]), 512, (8, 8), (8, 8)),
        ("e6 fwd", K.CONV_S2, (8, 8, [512]), 512, (4, 4), (4, 4)),
        ("e7 fwd", K.CONV_S2, (4, 4, [512]), 512, (2, 2), (2, 2)),
        ("d7 fwd", K.TCONV_S2, (2, 2, [512, 512]), 512, (2, 2), (4, 4)),
        ("d6 fwd", K.TCONV_S2, (4, 4, [512, 512]), 512, (4, 4), (8, 8)),
        ("d5 fwd", K.TCONV_S2, (8, 8, [512, 512]), 512, (8, 8), (16, 16)),
        ("d5 dgrad", K.CONV_S2, (16, 16, [512]), 1024, (8, 8), (8, 8)),
        ("d6 dgrad", K.CONV_S2, (8, 8, [512]), 1024, (4, 4), (4, 4)),
        ("e5 dgrad", K.TCONV_S2, (8, 8, [512]), 512, (8, 8), (16, 16)),
        ("e6 dgrad", K.TCONV_S2, (4, 4, [512]), 512, (4, 4), (8, 8)),
    ]
    mode_name = argv[0] if argv else "warm"
    _lib.lib().pfa_set_knob(1, int(argv[1]) if len(argv) > 1 else 0)
    _lib.lib().pfa_set_knob(0, int(argv[2]) if len(argv) > 2 else 0)
    scale = int(argv[3]) if len(argv) > 3 else 0
    flush = torch.empty(256 << 20, device="cuda", dtype=torch.float32) if mode_name == "cold" else None
    for name, mode, (h, w, cs), n, (hq, wq), (ho, wo) in layers:
        srcs = [T(B, h, w, c) for c in cs]
        wgt = T(n, 16, sum(cs))
        o = torch.empty((B, ho, wo, n), device="cuda", dtype=dt)
        sk = K.choose_splitk(mode, B * hq * wq, n, sum(cs), dt)
        if scale:
            sk = max(1, sk * scale // 4)
        print(name, "slices", sk, flush=True)
        for _ in range(6):
            if flush is not None:
                flush.fill_(1.0)
            K.gather_gemm(mode, srcs, wgt, n, [o], (B, hq, wq), ws, splitk=sk)
        torch.cuda.synchronize()


def rows(argv):
    f = glob.glob(f"{argv[0]}/*/*_kernel_trace.csv")[0]
    trace = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    seq = []
    for r in trace:
        n = r["Kernel_Name"]
        if "gather_gemm" in n or "finish" in n or "conv_patch" in n or "tconv4" in n:
            kind = "finish" if "finish" in n else "mfma"
            seq.append((kind, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]),
                        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    tot = [0.0, 0.0]
    for i in range(0, len(seq), 12):                     # six repeats of (MFMA launch, combining pass)
        pair = seq[i:i + 12]
        g = [x[2] for x in pair if x[0] == "mfma"][2:]
        c = [x[2] for x in pair if x[0] == "finish"][2:]
        ga, ca = sum(g) / max(1, len(g)), sum(c) / max(1, len(c))
        tot[0] += ga
        tot[1] += ca
        print(f"MFMA launch: {pair[0][1]:4d} workgroups {ga:6.1f} us | combining pass: {pair[1][1] if len(pair) > 1 else 0:4d} workgroups {ca:5.1f} us")
    print(f"sum: MFMA launches {tot[0]:.1f} us, combining passes {tot[1]:.1f} us")


if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in ("run", "rows"):
        sys.exit(__doc__)
    (run if sys.argv[1] == "run" else rows)(sys.argv[2:])
