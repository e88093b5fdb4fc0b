#!/usr/bin/env python3
"""tools/extract_kernel.py file.s mangled-fragment out.s : one kernel's listing out of a hipcc -S file"""
import sys
lines = open(sys.argv[1]).read().splitlines()
start = max(i for i, l in enumerate(lines) if l.startswith("_Z") and sys.argv[2] in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
open(sys.argv[3], "w").write("\n".join(lines[start:end + 1]) + "\n")
print(start, end)
