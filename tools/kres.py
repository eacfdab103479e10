"""Register / LDS / scratch usage of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), one line each:
    python tools/kres.py immunostruct_amd/csrc/egnn_layer_bwd.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src = sys.argv[1]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[2:],
                     capture_output=True, text=True).stderr
cur = None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name)}
        continue
    for key, pat in (("vgpr", r"\bVGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
            if key == "lds":
                print(f"{cur['name'][:90]:90s} vgpr {cur.get('vgpr'):3d} agpr {cur.get('agpr'):3d} scratch {cur.get('scratch'):4d} "
                      f"spill {cur.get('vspill'):3d} occ {cur.get('occ')} lds {cur['lds']}")
                cur = None
if "error" in out:
    print(out[-3000:])
