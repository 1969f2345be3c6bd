#!/bin/bash
# A/B aid: build variants/libcrt1d_hip_<name>.so = the current objects with ONE translation unit recompiled under extra flags.
#   tools/build_variant.sh <name> <unit: solve_closed|tri_n79_f64|tri_zq_f64|tri_zqpa|api|colpre|...> "<extra hipcc flags>"
# run with CRT1D_HIP_LIB=variants/libcrt1d_hip_<name>.so  (crt1d_amd/_lib.py)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; UNIT=$2; FLAGS=$3
C=$R/crt1d_amd/csrc
mkdir -p $R/variants
CXX="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -ffp-contract=off -Wall -Wno-unused-function"
case $UNIT in
  tri_n79_f64) SRC=tri_inst.hip; DEF="-DTRI_SCHEME=TriN79 -DTRI_TAG=n79 -DTRI_TIO=double -DTRI_TIOTAG=f64";;
  tri_n79_f32) SRC=tri_inst.hip; DEF="-DTRI_SCHEME=TriN79 -DTRI_TAG=n79 -DTRI_TIO=float -DTRI_TIOTAG=f32";;
  tri_zq_f64) SRC=tri_inst.hip; DEF="-DTRI_SCHEME=TriZq -DTRI_TAG=zq -DTRI_TIO=double -DTRI_TIOTAG=f64";;
  tri_zq_f32) SRC=tri_inst.hip; DEF="-DTRI_SCHEME=TriZq -DTRI_TAG=zq -DTRI_TIO=float -DTRI_TIOTAG=f32";;
  *) SRC=$UNIT.hip; DEF="";;
esac
/opt/rocm/bin/hipcc $CXX $DEF $FLAGS -c $C/$SRC -o $R/variants/${UNIT}_$NAME.o
OBJS=""
for o in colpre solve_closed solve_tridiag solve_tridiag_tile tri_zqpa api prep buffers tri_n79_f64 tri_n79_f32 tri_zq_f64 tri_zq_f32; do
  if [ $o == $UNIT ]; then OBJS="$OBJS $R/variants/${UNIT}_$NAME.o"; else OBJS="$OBJS $C/$o.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS -o $R/variants/libcrt1d_hip_$NAME.so -Wl,-rpath,/opt/rocm/lib
echo built variants/libcrt1d_hip_$NAME.so
