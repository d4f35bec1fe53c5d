#!/bin/bash
# Diagnostic build of the library with the split tile's epilogue switches (-DNG_X3_DIAG: nirgan_wino6_desc.algo bit 0x100 = the tile's
# global stores are skipped, 0x200 = the whole epilogue) -> scripts/diag/libnirgan_x3diag.so; scripts/diag/x3_parts.py times the trunk's
# plane GEMM with each.  Results are WRONG with a switch set: part timings only.  Build here (hipcc cross-compiles), run on the GPU box.
set -e
cd "$(dirname "$0")/../.."
O=scripts/diag/x3diag_build; mkdir -p $O
for f in nir-gan_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [ "$b" = igemm_conv ] || [ "$b" = wino6 ] || [ ! -f $O/$b.o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DNG_X3_DIAG -c $f -o $O/$b.o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/diag/libnirgan_x3diag.so $O/*.o
echo built scripts/diag/libnirgan_x3diag.so
