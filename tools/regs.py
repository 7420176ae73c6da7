"""Registers, scratch and LDS of every render_kernel instance (hipcc -Rpass-analysis=kernel-resource-usage); rebuilds the library."""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
lib = os.path.join(g.CSRC, "libray_amd.so")
r = subprocess.run([g.HIPCC] + g.HIP_FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-o", lib, os.path.join(g.CSRC, "ray_amd.hip")], capture_output=True, text=True, cwd=g.CSRC)
if r.returncode:
    print(r.stderr[-4000:]); sys.exit(1)
for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
    name = b.split()[0]
    if len(sys.argv) > 1 and sys.argv[1] not in name: continue
    f = lambda k: re.search(k + r": (\d+)", b).group(1)
    dm = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    print(f"{dm[:60]:60s} VGPR {f('VGPRs')} SGPR {f('SGPRs')} scratch {f('ScratchSize .bytes/lane.')} occ {f('Occupancy .waves/SIMD.')} sspill {f('SGPRs Spill')} vspill {f('VGPRs Spill')} LDS {f('LDS Size .bytes/block.')}")
