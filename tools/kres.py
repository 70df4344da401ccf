"""Per-kernel register / scratch / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kres.py robust_e_nerf_amd/csrc/ren_mlp_x.hip [filter] [extra hipcc flags...]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)
    if flt and flt not in name:
        continue
    print(f"{name:60s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>4s} scratch {r.get('ScratchSize [bytes/lane]','?'):>5s} "
          f"spill {r.get('VGPRs Spill','?'):>4s} occ {r.get('Occupancy [waves/SIMD]','?'):>2s} lds {r.get('LDS Size [bytes/block]','?')}")
