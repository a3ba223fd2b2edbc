#!/bin/bash
# A/B build: recompile ONE OR MORE csrc sources (comma-separated) with extra flags and link them with the objects of the last regular build
#   tools/variant.sh <name> <source.hip[,source2.hip...]> [-DFOO ...]   ->  pytorch-kaldi-resnet_amd/variants/libspkhip_<name>.so  (use with SPK_LIB=...)
set -e
cd "$(dirname "$0")/../pytorch-kaldi-resnet_amd"
NAME=$1; SRCS=$2; shift 2
mkdir -p variants
OBJS=""; EXCL=""
for SRC in $(echo $SRCS | tr ',' ' '); do
    OBJ=variants/${NAME}_${SRC%.*}.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -x hip -c csrc/$SRC -o $OBJ "$@" &
    OBJS="$OBJS $OBJ"; EXCL="$EXCL|build/${SRC%.*}.o"
done
wait
OTHERS=$(ls build/*.o | grep -v -E "^(${EXCL#|})$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libspkhip_${NAME}.so $OBJS $OTHERS
echo variants/libspkhip_${NAME}.so
