#!/bin/bash
# Knock-out builds of the four-wave register-fed split tile: the stamped diagnostic library (scripts/diag/x3r_stamps.sh) once per value of
# NG_X3R_KO (bit mask of parts of the K loop LEFT OUT, csrc/igemm_x3r.h: 1 conversion VALU, 2 v_cvt_pk -> v_and, 4 raw-row fetches,
# 8 B fetches + stores, 16 block 3's fragment reads) -> scripts/diag/libnirgan_x3rko_<mask>.so.  Results are wrong by construction; the
# stamps say what the remaining parts cost.  Build here, run on the GPU box:
#     bash scripts/diag/x3r_knockout.sh 0 1 2 4 8 12 16 31
#     gpurun -- 'for k in 0 1 2 4 8 12 16 31; do X3R_LIB=scripts/diag/libnirgan_x3rko_$k.so python3 scripts/diag/x3r_stamps.py; done'
set -e
cd "$(dirname "$0")/../.."
bash scripts/diag/x3r_stamps.sh > /dev/null
O=scripts/diag/x3rstamp_build
for k in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DNG_X3R_STAMP -DNG_X3R_KO=$k -c nir-gan_amd/csrc/igemm_conv.hip -o $O/igemm_conv_ko$k.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/diag/libnirgan_x3rko_$k.so $(ls $O/*.o | grep -v igemm_conv) $O/igemm_conv_ko$k.o && echo built ko $k ) &
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait
rm -f $O/igemm_conv_ko*.o
