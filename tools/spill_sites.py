"""Where a kernel spills: python tools/spill_sites.py FILE.hip MANGLED_SUBSTRING  -- compiles FILE to gfx950 assembly and prints, for
the kernels whose mangled name contains the substring, every scratch access with its line offset inside the kernel and the
nearest preceding source line marker (.loc), so that spills can be attributed to a phase of the kernel."""
import re
import subprocess
import sys

src, key = sys.argv[1], sys.argv[2]
asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-gline-tables-only", "-S", "-DIS_LAYER_M1=2", "-DIS_LAYER_GEO=1", "-Wno-unused-function"] + sys.argv[3:] + [
                      "--cuda-device-only", src, "-o", "-"], capture_output=True, text=True).stdout.split("\n")
starts = [i for i, l in enumerate(asm) if re.match(r"_Z\S+:", l)]
for a, b in zip(starts, starts[1:] + [len(asm)]):
    name = asm[a].split(":")[0]
    if key not in name:
        continue
    print(name[:70], "lines", b - a)
    loc = "?"
    for i in range(a, b):
        m = re.search(r"\.loc\s+\d+\s+(\d+)", asm[i])
        if m:
            loc = m.group(1)
        if "scratch_" in asm[i]:
            print(f"   +{i - a:6d}  src line {loc:>5s}  {asm[i].strip()[:70]}")
