"""Registers, scratch and occupancy of every kernel of libcrt1d_hip.so, from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: python tools/kernel_resources.py <dir with one <unit>.txt of remarks per translation unit>  > profiles/rNN/kernel_resources.txt
(the remarks: hipcc <CXXFLAGS of csrc/Makefile> --cuda-device-only -Rpass-analysis=kernel-resource-usage -c <unit>.hip 2> <unit>.txt)"""
import glob, re, subprocess, sys

rows = []
for f in sorted(glob.glob(sys.argv[1] + "/*.txt")):
    s = open(f).read()
    for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)", s, re.S):
        rows.append((f.split("/")[-1][:-4],) + m.groups())
names = subprocess.run(["c++filt"], input="\n".join(r[1] for r in rows), capture_output=True, text=True).stdout.split("\n")
print("# hipcc -Rpass-analysis=kernel-resource-usage, gfx950: unit, VGPRs, AGPRs, scratch bytes per lane, waves per SIMD the registers allow, kernel")
for r, d in zip(rows, names):
    d = d.replace("crt::(anonymous namespace)::", "").split("(")[0]
    print(f"{r[0]:14s} vgpr {int(r[2]):4d} agpr {int(r[3]):3d} scratch {int(r[4]):4d} waves/SIMD {r[5]}  {d}")
