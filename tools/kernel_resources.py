#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS use of every HIP source (device-only compile to assembly, no GPU needed).
Flags scratch (private segment) and spills: a kernel that touches scratch in its hot loop is a perf bug here.
    python tools/kernel_resources.py [file.hip ...]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "jatts_amd", "csrc", "*.hip")))
    for f in files:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only",
                            "-Wno-pass-failed", "-S", "-o", out, f], check=True, stderr=subprocess.DEVNULL)
            name, rec = None, {}
            for line in open(out):
                m = re.match(r"\s+\.amdhsa_kernel\s+(\S+)", line)
                if m:
                    name, rec[name] = m.group(1), {}
                m = re.match(r"\s+\.amdhsa_(private_segment_fixed_size|next_free_vgpr|accum_offset)\s+(\d+)", line)
                if m and name:
                    rec[name][m.group(1)] = int(m.group(2))
                m = re.match(r"\s+\.(vgpr_spill_count|sgpr_spill_count):\s+(\d+)", line)
                if m and name:
                    pass
            print(f"== {os.path.basename(f)}")
            for k, v in rec.items():
                demangled = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
                demangled = demangled.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
                flag = "  <-- SCRATCH" if v.get("private_segment_fixed_size", 0) else ""
                print(f"  {demangled[:70]:70s} regs={v.get('next_free_vgpr', 0):4d} (arch {v.get('accum_offset', 0):3d}) "
                      f"scratch={v.get('private_segment_fixed_size', 0):4d}{flag}")


if __name__ == "__main__":
    main()
