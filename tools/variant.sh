#!/bin/bash
# A/B build: recompile ONE csrc source with extra flags and link it with the objects of the last regular build
#   tools/variant.sh <name> <source.hip> [-DFOO ...]   ->  pytorch-kaldi-resnet_amd/variants/libspkhip_<name>.so  (use with SPK_LIB=...)
set -e
cd "$(dirname "$0")/../pytorch-kaldi-resnet_amd"
NAME=$1; SRC=$2; shift 2
mkdir -p variants
OBJ=variants/${NAME}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -x hip -c csrc/$SRC -o $OBJ "$@"
OTHERS=$(ls build/*.o | grep -v "build/${SRC%.*}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libspkhip_${NAME}.so $OBJ $OTHERS
echo variants/libspkhip_${NAME}.so
