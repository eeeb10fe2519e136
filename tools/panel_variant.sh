#!/bin/bash
# A/B build of ONE source file: tools/panel_variant.sh <name> "<-D flags>" [file, default gf_panel] -> alegnn_amd/libgfhip_<name>.so (the other
# objects are the shipped ones; load with GFHIP_EXPERIMENTS=1 GFHIP_LIB=<path>)
cd "$(dirname "$0")/../graph-neural-networks_amd" || exit 1
NAME=$1; EXTRA=$2; F=${3:-gf_panel}; mkdir -p build/$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../include -Icsrc -Wall -Wno-unused-function $EXTRA -c csrc/$F.hip -o build/$NAME/$F.o || exit 1
OBJS=$(ls build/*.o | grep -v "/$F.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o alegnn_amd/libgfhip_$NAME.so $OBJS build/$NAME/$F.o
