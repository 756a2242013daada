#!/bin/bash
# A variant build of libelprep_hip.so into build_ab/ (git-ignored, travels with gpurun): one source file with sed expressions applied
# (constants of an A/B session), everything else as built.  usage: build_var.sh <name> <file-without-.hip> '<sed expr>' ['<sed expr>' ...]
set -e
NAME=$1; F=$2; shift; shift
cd "$(dirname "$0")/../../elprep_amd/csrc"
mkdir -p ../../build_ab
cp $F.hip var_tmp_$NAME.hip
for e in "$@"; do sed -i "$e" var_tmp_$NAME.hip; done
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off"
/opt/rocm/bin/hipcc $FLAGS -c var_tmp_$NAME.hip -o ../../build_ab/$NAME.o
rm -f var_tmp_$NAME.hip
objs=""; for o in *.o; do [ "$o" = "$F.o" ] && objs="$objs ../../build_ab/$NAME.o" || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_ab/lib_$NAME.so $objs -ldl
echo built build_ab/lib_$NAME.so
