#!/usr/bin/env python3
"""Register / occupancy table of every search_kernel instantiation (hipcc -Rpass-analysis=kernel-resource-usage).
   tools/kernel_regs.py [extra hipcc flags...]"""
import os, re, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
csrc = os.environ.get("SVSX_CSRC") or os.path.join(here, "..", "scalablevectorsearch_amd", "csrc")
txt = ""
procs = []
for name in ("f32", "f16", "i8", "u8"):  # one translation unit per row type
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
           "-fno-fast-math", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(csrc, f"svsx_search_{name}.hip"), "-o", f"/tmp/_regs_{name}.o"] + sys.argv[1:]
    procs.append(subprocess.Popen(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True))
for pr in procs:
    txt += pr.communicate()[1]
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]
    g = lambda k: int(m.group(1)) if (m := re.search(re.escape(k) + r": (\d+)", b)) else -1
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    m = re.search(r"search_kernel<(.*?)>", dem)
    print(f"{m.group(1) if m else dem:40s} VGPR {g('VGPRs'):4d} SGPR {g('TotalSGPRs'):4d} scratch {g('ScratchSize [bytes/lane]'):4d} waves/SIMD {g('Occupancy [waves/SIMD]')}")
