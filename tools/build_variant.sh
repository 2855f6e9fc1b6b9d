#!/bin/bash
# tools/build_variant.sh NAME "-DTCMI_F_BLOCK=256 ..."  ->  trueconsense_amd/lib/variants/libtcmi_NAME.so  (A/B builds; select with TCMI_LIB=...)
set -e
cd "$(dirname "$0")/../trueconsense_amd/csrc"
name=$1; shift
out=../lib/variants; mkdir -p $out/obj_$name
for f in tally.hip tally_fast.hip tally_planes.hip call.hip api.cpp readset.cpp consensus_walk.cpp insert_tokens.cpp bam_reader.cpp pipeline.cpp; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c $f -o $out/obj_$name/${f%.*}.o &
done
wait
hipcc --offload-arch=gfx950 -shared -o $out/libtcmi_$name.so $out/obj_$name/*.o -lz -lpthread
echo built $out/libtcmi_$name.so
