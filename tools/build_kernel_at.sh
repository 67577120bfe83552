#!/bin/bash
# tools/build_kernel_at.sh NAME COMMIT [EXTRA_FLAGS]: libtagdig_NAME.so = today's host code with the kernel headers
# (kernels.hpp, kernel_fast.hpp, kernel_fast2.hpp) as they were at COMMIT -- baselines for tools/ab_inproc.py.
set -e
name=$1; commit=$2; extra=$3
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p "$tmp/x" "$tmp/include"
cp -r "$root/tagdigger_amd/csrc" "$tmp/x/csrc"          # (csrc includes ../../include)
cp "$root"/include/*.h "$tmp/include/"
d="$tmp/x/csrc"
for f in kernels.hpp kernel_fast.hpp kernel_fast2.hpp; do git -C "$root" show "$commit:tagdigger_amd/csrc/$f" > "$d/$f"; done
rm -f "$d"/*.o
( cd "$d" && make -s -j8 OUT=libtagdig_out.so CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function $extra" )
cp "$d/libtagdig_out.so" "$root/tagdigger_amd/libtagdig_$name.so"
rm -rf "$tmp"
echo "built tagdigger_amd/libtagdig_$name.so (kernels at $commit)"
