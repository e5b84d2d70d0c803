#!/bin/bash
# build tools/variants/libsjmi_<name>.so from the working tree with extra compiler flags for ONE source (VARIANT_SRC, default
# stage1.hip; the other sources are compiled once into /tmp/objs and re-used while they are older than their objects).
# usage: [VARIANT_SRC=coop_walk.hip] tools/build_variant.sh name [-DX=1 ...]
set -e
R=/root/repo; cd $R
name=$1; shift
VS=${VARIANT_SRC:-stage1.hip}
mkdir -p /tmp/objs tools/variants
OBJS=""
for f in stage1.hip strings.hip batch.hip walk.hip coop_walk.hip masks.hip sjmi_api.hip host/simdjson_parser.cpp; do
  [ "$f" = "$VS" ] && continue
  o=/tmp/objs/$(basename $f).o
  newest=$(ls -t simdjson-java_amd/csrc/*.h simdjson-java_amd/csrc/host/*.h include/*.h simdjson-java_amd/csrc/$f | head -1)
  if [ ! -f $o ] || [ $newest -nt $o ]; then hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -I include -c simdjson-java_amd/csrc/$f -o $o; fi
  OBJS="$OBJS $o"
done
# (VARIANT_FILE: compile this file in place of the source -- e.g. `git show HEAD:simdjson-java_amd/csrc/strings.hip` saved beside it)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -I include "$@" -c ${VARIANT_FILE:-simdjson-java_amd/csrc/$VS} -o /tmp/objs/variant_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -pthread /tmp/objs/variant_$name.o $OBJS -o tools/variants/libsjmi_$name.so
echo built tools/variants/libsjmi_$name.so
