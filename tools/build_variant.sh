#!/bin/bash
# build tools/variants/libsjmi_<name>.so from the working tree with extra compiler flags for stage1.hip only (the other sources
# are compiled once into /tmp/objs and re-used while they are older than their objects).  usage: tools/build_variant.sh name [-DX=1 ...]
set -e
R=/root/repo; cd $R
name=$1; shift
mkdir -p /tmp/objs tools/variants
OBJS=""
for f in strings.hip batch.hip walk.hip coop_walk.hip masks.hip sjmi_api.hip host/simdjson_parser.cpp; do
  o=/tmp/objs/$(basename $f).o
  newest=$(ls -t simdjson-java_amd/csrc/*.h simdjson-java_amd/csrc/host/*.h include/*.h simdjson-java_amd/csrc/$f | head -1)
  if [ ! -f $o ] || [ $newest -nt $o ]; then hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -I include -c simdjson-java_amd/csrc/$f -o $o; fi
  OBJS="$OBJS $o"
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -I include "$@" -c simdjson-java_amd/csrc/stage1.hip -o /tmp/objs/stage1_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -pthread /tmp/objs/stage1_$name.o $OBJS -o tools/variants/libsjmi_$name.so
echo built tools/variants/libsjmi_$name.so
