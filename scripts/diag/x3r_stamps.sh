#!/bin/bash
# Diagnostic build of the library with in-kernel stamps in the four-wave register-fed split tile (-DNG_X3R_STAMP: s_memtime at the block
# boundaries of a K-tile and around the epilogue, per-wave sums in a device array read back by nirgan_x3r_stamps) ->
# scripts/diag/libnirgan_x3rstamp.so; scripts/diag/x3r_stamps.py prints cycles per segment.  The stamps drain the LDS counter: the
# numbers attribute time, they are not the product kernel's.  Build here (hipcc cross-compiles), run on the GPU box.
set -e
cd "$(dirname "$0")/../.."
O=scripts/diag/x3rstamp_build; mkdir -p $O
for f in nir-gan_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [ "$b" = igemm_conv ] || [ ! -f $O/$b.o ] || [ $f -nt $O/$b.o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DNG_X3R_STAMP $* -c $f -o $O/$b.o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/diag/libnirgan_x3rstamp.so $O/*.o
echo built scripts/diag/libnirgan_x3rstamp.so
