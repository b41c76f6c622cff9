#!/bin/bash
# scratch/build_variant.sh NAME "-DFOO=1 ..." [FILE=sift] : FILE.hip rebuilt with extra flags, linked with the current objects into scratch/variants/lib_NAME.so
set -e
F=${3:-sift}
cd "$(dirname "$0")/.."
mkdir -p scratch/variants
B=imagemosaicing_amd/csrc/build
python -c "from imagemosaicing_amd import build as b; b.generate()"      # blur16_asm.inc (generated, lives in $B)
PF=""; [ "$F" = ransac ] && PF="-fno-slp-vectorize"                      # build.py PER_FILE
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result $PF $2 -Iinclude -I$B -x hip -c imagemosaicing_amd/csrc/$F.hip -o scratch/variants/${F}_$1.o
OBJS=$(ls $B/*.o | grep -v $F.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/variants/lib_$1.so scratch/variants/${F}_$1.o $OBJS
rm -f scratch/variants/${F}_$1.o
ls -la scratch/variants/lib_$1.so
