"""Register and scratch use of every kernel of the library, from the compiler's resource remarks.
usage: make -C raytracer_amd/csrc asm 2>&1 | python tools/kernel_resources.py [FILTER]"""
import re
import subprocess
import sys

want = sys.argv[1] if len(sys.argv) > 1 else ""
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    body = m.group(1).strip()
    if body.startswith("Function Name:"):
        cur = {"name": body.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in body:
        k, v = body.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    if want and want not in name:
        continue
    print(f'{name[:90]:90s} vgpr {r.get("VGPRs")} sgpr {r.get("SGPRs")} spillV {r.get("VGPRs Spill")} spillS {r.get("SGPRs Spill")} '
          f'scratch {r.get("ScratchSize [bytes/lane]")} occ {r.get("Occupancy [waves/SIMD]")}')
