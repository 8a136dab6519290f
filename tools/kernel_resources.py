"""Compiles csrc/aie_capi.hip with -Rpass-analysis=kernel-resource-usage and prints one line per kernel:
VGPRs, AGPRs, SGPRs, scratch bytes per lane, occupancy (waves per SIMD), LDS.   python tools/kernel_resources.py"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-comment",
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "ai-economist_amd", "csrc", "aie_capi.hip"),
           "-o", "/tmp/aie_resources.so", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = {}
    rows = []
    for line in out.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
        if not m:
            continue
        txt = m.group(1).strip()
        if txt.startswith("Function Name:"):
            cur = {"name": txt.split(":", 1)[1].strip()}
            rows.append(cur)
        elif ":" in txt:
            k, v = txt.split(":", 1)
            cur[k.strip()] = v.strip()
    print("%-58s %5s %5s %5s %8s %4s %7s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "occ", "LDS"))
    for r in rows:
        print("%-58s %5s %5s %5s %8s %4s %7s" % (r["name"][:58], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"),
                                                 r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"),
                                                 r.get("LDS Size [bytes/block]")))


if __name__ == "__main__":
    main()
