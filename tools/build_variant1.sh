#!/bin/bash
# tools/build_variant1.sh NAME FILE.hip "-DFLAG=.."  ->  _ab_libs/NAME.so: an experiment build of ONE translation unit linked
# with the product objects of the others (csrc/obj/, from the last `python -m dsw_amd.build`); select with DSW_HIP_LIB
set -e
name="$1"; file="$2"; flags="$3"
root="$(cd "$(dirname "$0")/.." && pwd)"
src="$root/deepsphere-weather_amd/csrc"
mkdir -p "$root/_ab_libs"
o="$root/_ab_libs/${name}_$(basename "$file" .hip).o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c "$src/$file" -o "$o"
others=$(ls "$src"/obj/*.o | grep -v "/$(basename "$file" .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/_ab_libs/$name.so" $others "$o"
rm -f "$o"
echo "built _ab_libs/$name.so"
