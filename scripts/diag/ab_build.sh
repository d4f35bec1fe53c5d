#!/bin/bash
# A second build of the library with extra -D switches, for same-process A/B timings of a kernel variant against the shipped library
# (scripts/diag/ab_step.py switches the backend between timing blocks of ONE trainer: same buffers, same box, same clock).
#     bash scripts/diag/ab_build.sh NG_X3_NT_STORES        -> scripts/diag/libnirgan_ab.so
set -e
cd "$(dirname "$0")/../.."
O=scripts/diag/ab_build; rm -rf $O; mkdir -p $O
D=""; for d in "$@"; do D="$D -D$d"; done
for f in nir-gan_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result $D -c $f -o $O/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/diag/libnirgan_ab.so $O/*.o
echo built scripts/diag/libnirgan_ab.so with$D
