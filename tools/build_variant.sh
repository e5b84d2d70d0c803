#!/bin/bash
# tools/build_variant.sh NAME [-Dflags...] -> tools/variants/libsjmi_NAME.so (A/B experiments; SJMI_LIB=... selects it)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/tools/variants
cd $R/simdjson-java_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread "$@" -I $R/include stage1.hip strings.hip batch.hip walk.hip coop_walk.hip masks.hip sjmi_api.hip host/simdjson_parser.cpp -o $R/tools/variants/libsjmi_$name.so
echo built $name
