"""Static instruction statistics per kernel: compiles the device side of csrc/aie_capi.hip to assembly
(hipcc --cuda-device-only -S) and counts instructions by class.   python tools/kernel_asm_stats.py [min_instructions]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out = os.path.join(tempfile.gettempdir(), "aie_kernels.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-comment",
                    "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S",
                    os.path.join(ROOT, "ai-economist_amd", "csrc", "aie_capi.hip"), "-o", out], check=True)
    floor = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    cur, counts = None, {}
    for line in open(out):
        m = re.match(r"^([A-Za-z_]\w*):", line)
        if m and not line.startswith(".L"):
            cur = m.group(1)
            counts[cur] = dict(total=0, s_load=0, salu=0, valu=0, lds=0, vmem=0, scratch=0, waitcnt=0)
            continue
        t = line.strip()
        if cur is None or not t or t.startswith((".", ";", "//")) or t.endswith(":"):
            continue
        op, c = t.split()[0], counts[cur]
        c["total"] += 1
        if op.startswith(("s_load", "s_buffer_load")): c["s_load"] += 1
        elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith("scratch_"): c["scratch"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")): c["vmem"] += 1
        elif op.startswith("v_"): c["valu"] += 1
    print("%-44s %7s %7s %7s %7s %6s %6s %8s %8s" % ("kernel", "total", "s_load", "salu", "valu", "lds", "vmem", "scratch", "waitcnt"))
    for k, v in counts.items():
        if v["total"] >= floor:
            print("%-44s %7d %7d %7d %7d %6d %6d %8d %8d" % (k[:44], v["total"], v["s_load"], v["salu"], v["valu"], v["lds"],
                                                              v["vmem"], v["scratch"], v["waitcnt"]))


if __name__ == "__main__":
    main()
