#!/bin/bash
# build_variant.sh NAME FILE.hip 'sed-expression' [SOURCE] -- libffpic_hip_NAME.so = the shipped objects with ONE translation unit rebuilt from a sed-edited copy
# (A/B runs of two builds in one gpurun call: FFHIP_LIB=libffpic_hip_NAME.so picks the library in ffpic_amd/capi.py)
set -e
HERE=$(cd "$(dirname "$0")/../../ffpic_amd/csrc" && pwd)
NAME=$1; SRC=$2; EXPR=$3; FROM=${4:-$HERE/$SRC}   # SOURCE: another edition of FILE.hip (e.g. `git show HEAD:ffpic_amd/csrc/FILE.hip > /tmp/old.hip`)
make -s -C "$HERE" -j8
sed "$EXPR" "$FROM" > "$HERE/variant_$NAME.hip"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Wall -Wno-unused-function -I"$HERE/../../include" -I"$HERE" -c "$HERE/variant_$NAME.hip" -o "$HERE/variant_$NAME.o"
OBJS=$(ls "$HERE"/ffhip_*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$HERE/../libffpic_hip_$NAME.so" $OBJS "$HERE/variant_$NAME.o" -ldl -lpthread
rm -f "$HERE/variant_$NAME.hip" "$HERE/variant_$NAME.o"
echo "built ffpic_amd/libffpic_hip_$NAME.so"
