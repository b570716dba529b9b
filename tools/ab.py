#!/usr/bin/env python3
"""Same-box A/B of bench.py under environment switches: every arm is a child process, arms alternate over `--rounds`
rounds (box drift hits all arms alike), the median ms/step per arm is printed next to the first arm's.

    python tools/ab.py [--rounds 3] [--args "--batch 16"] base: fused:PFA_BN_STATS_FUSED=1 knob10:PFA_KNOBS=10=128

An arm is name:ENV=VALUE[;ENV=VALUE...] (empty after the colon = the environment as it is)."""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--args", default="")
    ap.add_argument("arms", nargs="+")
    a = ap.parse_args()
    arms = []
    for spec in a.arms:
        name, _, envs = spec.partition(":")
        env = dict(os.environ)
        for kv in filter(None, envs.split(";")):
            k, _, v = kv.partition("=")
            env[k] = v
        arms.append((name, env))
    res = {n: [] for n, _ in arms}
    for r in range(a.rounds):
        for name, env in arms:
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-roofline", "--no-extras"] + a.args.split()
            p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode or not lines:
                print(f"{name}: FAILED rc={p.returncode}: {p.stderr[-400:]}", flush=True)
                continue
            d = json.loads(lines[-1])
            res[name].append(d["ms_per_step"])
            print(f"round {r} {name:>14}: {d['ms_per_step']:.3f} ms/step  {d['value']:.0f} {d['unit']}", flush=True)
    base = statistics.median(res[arms[0][0]]) if res[arms[0][0]] else None
    for name, _ in arms:
        if res[name]:
            m = statistics.median(res[name])
            rel = f"{100 * (base / m - 1):+.2f} % vs {arms[0][0]}" if base else ""
            print(f"{name:>14}: median {m:.3f} ms/step over {len(res[name])} runs {sorted(res[name])}  {rel}")


if __name__ == "__main__":
    main()
