#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra hipcc flags, e.g. -DHJ_BLOCK_STREAM_WORDS=0>" [file.hip ...]
# Builds nvimagecodec_amd/variants/lib_<name>.so: the named .hip files (default gpu_huffman.hip) recompiled with the flags, every
# other object taken from the shipped build.  For A/B runs on one GPU box (tools/ab_entropy.sh); variants/ is git-ignored.
set -e
cd "$(dirname "$0")/../nvimagecodec_amd/csrc"
NAME=$1; FLAGS=$2; shift 2
FILES=${@:-gpu_huffman.hip}
make -s -j8 >/dev/null
mkdir -p build_$NAME ../variants
OBJS=""
for o in build/*.o; do
  b=$(basename $o .o); keep=1
  for f in $FILES; do [ "$b" = "$(basename $f .hip)" ] && keep=0; done
  [ $keep = 1 ] && OBJS="$OBJS $o"
done
for f in $FILES; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wextra -fvisibility=hidden -I../../include --offload-arch=gfx950 $FLAGS -c $f -o build_$NAME/$(basename $f .hip).o
  OBJS="$OBJS build_$NAME/$(basename $f .hip).o"
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../variants/lib_$NAME.so $OBJS -lpthread -ldl
echo "nvimagecodec_amd/variants/lib_$NAME.so"
