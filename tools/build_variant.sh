#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG=.. ..."  ->  _ab_libs/NAME.so (an experiment build of csrc/, not tracked);
# select it with DSW_HIP_LIB=$PWD/_ab_libs/NAME.so (tools/ab_libs.sh)
set -e
name="$1"; flags="$2"
root="$(cd "$(dirname "$0")/.." && pwd)"
src="$root/deepsphere-weather_amd/csrc"
obj="$root/_ab_libs/obj_$name"; mkdir -p "$obj"
for f in "$src"/*.hip; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c "$f" -o "$obj/$(basename "$f" .hip).o" ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/_ab_libs/$name.so" "$obj"/*.o
rm -rf "$obj"
echo "built _ab_libs/$name.so"
