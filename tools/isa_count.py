#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc -S listing:  tools/isa_count.py file.s 'search_kernel<0, 0, 0, 96, false, 2, 0>' [first_line last_line]
Counts VALU / SALU / LDS / VMEM / branch / waitcnt instructions (whole kernel, or between two line numbers relative to the kernel start)."""
import re, subprocess, sys
path, want = sys.argv[1], sys.argv[2]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 30
lines = open(path).read().splitlines()
start = end = None
for i, l in enumerate(lines):
    if l.startswith("_Z") and ":" in l:
        name = l.split(":")[0]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if want in dem:
            start = i
    elif start is not None and end is None and l.strip().startswith("s_endpgm"):
        end = i
        break
assert start is not None, "kernel not found"
mix = {}
for l in lines[start + lo:min(end, start + hi)]:
    t = l.strip()
    if not t or t.startswith((";", ".", "_Z")) or t.endswith(":"):
        continue
    op = t.split()[0]
    if op.startswith(("v_",)): k = "VALU"
    elif op.startswith("ds_"): k = "LDS"
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): k = "VMEM"
    elif op.startswith("s_waitcnt"): k = "waitcnt"
    elif op.startswith(("s_cbranch", "s_branch")): k = "branch"
    elif op.startswith("s_nop"): k = "nop"
    elif op.startswith("s_"): k = "SALU"
    else: k = "other"
    mix[k] = mix.get(k, 0) + 1
print(f"lines {start + lo}..{min(end, start + hi)}", mix, "total", sum(mix.values()))
