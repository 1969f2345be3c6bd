#!/bin/bash
# registers / scratch / occupancy of every kernel of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage)
#   tools/kernel_resources.sh <unit: solve_closed|tri_n79_f64|tri_zq_f64|tri_zqpa|...> [name filter]
R=$(cd "$(dirname "$0")/.." && pwd)
UNIT=$1; FILT=${2:-}
C=$R/crt1d_amd/csrc
case $UNIT in
  tri_n79_f64) SRC=tri_inst.hip; DEF="-DTRI_SCHEME=TriN79 -DTRI_TAG=n79 -DTRI_TIO=double -DTRI_TIOTAG=f64";;
  tri_n79_f32) SRC=tri_inst.hip; DEF="-DTRI_SCHEME=TriN79 -DTRI_TAG=n79 -DTRI_TIO=float -DTRI_TIOTAG=f32";;
  tri_zq_f64) SRC=tri_inst.hip; DEF="-DTRI_SCHEME=TriZq -DTRI_TAG=zq -DTRI_TIO=double -DTRI_TIOTAG=f64";;
  tri_zq_f32) SRC=tri_inst.hip; DEF="-DTRI_SCHEME=TriZq -DTRI_TAG=zq -DTRI_TIO=float -DTRI_TIOTAG=f32";;
  *) SRC=$UNIT.hip; DEF="";;
esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -ffp-contract=off $DEF -Rpass-analysis=kernel-resource-usage -c $C/$SRC -o /tmp/kr_$$.o 2>&1 | python3 -c "
import sys, re, subprocess
name = None
rows = []
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m: name = m.group(1); vg = sc = None
    m = re.search(r' VGPRs: (\d+)', l)
    if m: vg = m.group(1)
    m = re.search(r'ScratchSize \[bytes/lane\]: (\d+)', l)
    if m: sc = m.group(1)
    m = re.search(r'Occupancy \[waves/SIMD\]: (\d+)', l)
    if m: rows.append((name, vg, sc, m.group(1)))
names = subprocess.run(['c++filt'], input='\n'.join(r[0] for r in rows), capture_output=True, text=True).stdout.split('\n')
for (n, vg, sc, oc), d in zip(rows, names):
    d = d.replace('crt::(anonymous namespace)::', '').replace('(crt::SolveArgs, ', '(').split('(')[0]
    if '$FILT' in d: print(f'vgpr {vg:>4} scratch {sc:>4} waves/SIMD {oc}  {d}')
"
rm -f /tmp/kr_$$.o
