#!/usr/bin/env python3
"""Register / LDS / spill table of every kernel in the given csrc/*.hip files (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python scripts/kernel_resources.py igemm_dma.hip [more.hip ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "progressive-stable-diffusion_amd", "csrc")
for f in sys.argv[1:]:
    # (the flags of lib.build(): MFMA accumulators stay in VGPRs)
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
                        "-I" + os.path.join(ROOT, "include"),
                        "-c", os.path.join(CSRC, f), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True)
    cur = {}
    rows = []
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
        if not m:
            continue
        txt = m.group(1)
        if txt.startswith("Function Name:"):
            cur = {"name": txt.split(":", 1)[1].strip()}
            rows.append(cur)
        elif ":" in txt:
            k, v = txt.split(":", 1)
            cur[k.strip()] = v.strip()
    print(f"== {f}")
    print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'occ':>4s} {'LDS':>7s} {'scratch':>7s}")
    for c in rows:
        name = subprocess.run(["c++filt", c["name"]], capture_output=True, text=True).stdout.strip()
        name = name.replace("(anonymous namespace)::", "").split("(")[0]
        print(f"{name[:70]:70s} {c.get('VGPRs','?'):>5s} {c.get('AGPRs','?'):>5s} {c.get('TotalSGPRs','?'):>5s} "
              f"{c.get('VGPRs Spill','?'):>6s} {c.get('SGPRs Spill','?'):>6s} {c.get('Occupancy [waves/SIMD]','?'):>4s} "
              f"{c.get('LDS Size [bytes/block]','?'):>7s} {c.get('ScratchSize [bytes/lane]','?'):>7s}")
