#!/usr/bin/env python3
"""A/B of two (or more) builds of libsvsx.so on the same workloads, one child process per library (SVSX_LIB):
   python tools/ab_libs.py --libs prod=variants/prod/libsvsx.so,new= --script tools/diag_early.py -- --nq 10000,20000 --windows 128,23
(an empty path = the in-tree library).  The script must print JSON lines with an "ms" field; rows are matched on all other scalar fields.
Each library runs `--repeat` times, alternating, and the table shows the median: clocks drift by a few per cent within one call."""
import argparse, json, os, statistics, subprocess, sys

ap = argparse.ArgumentParser()
ap.add_argument("--libs", required=True); ap.add_argument("--script", default="tools/diag_early.py"); ap.add_argument("--repeat", type=int, default=2)
ap.add_argument("rest", nargs=argparse.REMAINDER)
a = ap.parse_args()
libs = [tuple(x.split("=", 1)) for x in a.libs.split(",")]
rest = a.rest[1:] if a.rest and a.rest[0] == "--" else a.rest
rows = {}
for rep in range(a.repeat):
    for name, path in libs:
        env = dict(os.environ)
        if path:
            env["SVSX_LIB"] = path
        else:
            env.pop("SVSX_LIB", None)
        out = subprocess.run([sys.executable, a.script] + rest, env=env, capture_output=True, text=True)
        if out.returncode != 0:
            print(f"{name}: FAILED\n{out.stderr[-2000:]}")
            continue
        for line in out.stdout.splitlines():
            try:
                j = json.loads(line)
            except ValueError:
                continue
            key = tuple((k, v) for k, v in sorted(j.items()) if k not in ("ms", "qps", "kernel", "ids_identical", "ms_per_batch", "min_ms", "build_s") and not isinstance(v, (list, dict)))
            rows.setdefault(key, {}).setdefault(name, []).append(j["ms"])
base = libs[0][0]
for key, by in rows.items():
    desc = " ".join(f"{k}={v}" for k, v in key)
    cells = []
    for name, _ in libs:
        if name in by:
            m = statistics.median(by[name])
            rel = f" ({100.0 * (statistics.median(by[base]) / m - 1.0):+.1f} %)" if name != base and base in by else ""
            cells.append(f"{name} {m:.4f} ms{rel}")
    print(f"{desc:50s} " + "   ".join(cells))
