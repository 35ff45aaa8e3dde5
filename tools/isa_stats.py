"""Static instruction mix and register use of the specialised kernel's instances (runs anywhere: hipcc cross-compiles gfx950):
    python tools/isa_stats.py [--kernel SUBSTR] [--keep DIR] [-- extra hipcc flags]
Compiles vaporetto_amd/csrc/kernels_fast.hip with -save-temps and -Rpass-analysis=kernel-resource-usage and prints, per matching
kernel: VGPRs, SGPRs, spilled SGPRs, scratch, and the static counts of VALU / spill moves (v_readlane, v_writelane) / v_mov / quarter-rate
multiplies / SALU / LDS / VMEM / SMEM instructions and the code size.  What the design notes of DESIGN.md 4.2 quote."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="score_tiles_fast_kernelILi3ELi4ELb0ELb0")
    ap.add_argument("--file", default="kernels_fast.hip")
    ap.add_argument("--keep", default="")
    ap.add_argument("flags", nargs="*")
    a = ap.parse_args()
    d = a.keep or tempfile.mkdtemp(prefix="isa_")
    os.makedirs(d, exist_ok=True)
    src = os.path.join(ROOT, "vaporetto_amd", "csrc", a.file)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", os.path.join(d, "k.o"), "-save-temps=obj",
           "-Rpass-analysis=kernel-resource-usage"] + a.flags
    r = subprocess.run(cmd, stderr=subprocess.PIPE, cwd=os.path.dirname(src))
    if r.returncode:
        sys.stderr.write(r.stderr.decode())
        raise SystemExit(1)
    rem = r.stderr.decode()
    asm = next(os.path.join(d, f) for f in os.listdir(d) if f.endswith("gfx950.s"))
    text = open(asm).read()
    for m in re.finditer(r"^(\S*%s\S*):\s*; @" % re.escape(a.kernel), text, re.M):
        name = m.group(1)
        body = text[m.end():text.index(".end_amdhsa_kernel", m.end())]
        c = dict(valu=0, spill=0, vmov=0, mul_lo=0, salu=0, lds=0, vmem=0, smem=0, branch=0)
        for line in body.splitlines():
            s = line.strip()
            if not s or s[0] in ";." or s.endswith(":"):
                continue
            op = s.split()[0]
            if op.startswith(("v_readlane", "v_writelane")):
                c["spill"] += 1
            if op.startswith("v_mov"):
                c["vmov"] += 1
            if op.startswith(("v_mul_lo_u32", "v_mul_hi")):
                c["mul_lo"] += 1
            if op.startswith("v_"):
                c["valu"] += 1
            elif op.startswith(("s_load", "s_buffer")):
                c["smem"] += 1
            elif op.startswith(("s_cbranch", "s_branch")):
                c["branch"] += 1
                c["salu"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
                c["vmem"] += 1
        res = {}
        blk = rem[rem.index("Function Name: " + name):]
        for key in ("TotalSGPRs", "VGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill"):
            mm = re.search(re.escape(key) + r": (\d+)", blk)
            res[key] = int(mm.group(1)) if mm else None
        size = re.search(r"\.size\s+%s, .*\n" % re.escape(name), text)
        print(name)
        print("   ", res)
        print("   ", c)
    if not a.keep:
        print("(temps in %s)" % d)


if __name__ == "__main__":
    main()
