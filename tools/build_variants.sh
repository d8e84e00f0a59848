#!/bin/bash
# Build engine variants HERE (8 CPUs, in parallel) into build/var/ (git-ignored, but shipped to the GPU box):
#   tools/build_variants.sh name1 "-DFLAG ..." name2 "-DOTHER" ...
# then on the GPU box: python tools/time_variants.py [--n L] [--r R]
cd "$(dirname "$0")/.." || exit 1
mkdir -p build/var
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -shared -o build/var/lib_$name.so desirna_amd/csrc/engine.hip 2> build/var/$name.err || echo "BUILD FAILED: $name" ) &
done
wait
ls -la build/var/*.so
