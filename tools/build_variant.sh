#!/bin/bash
# tools/bin/libbde_<name>.so = the library built with extra compiler flags (A/B experiments; tools/bin is git-ignored)
#   bash tools/build_variant.sh nomfma "-DBDE_EXP_NOMFMA"
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=/tmp/bde_variant_$name
mkdir -p $out $root/tools/bin
make -s -C $root/beyond_deep_ensembles_amd/csrc OUT=$out CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I../../include $*" -j4
cp $out/libbde_hip.so $root/tools/bin/libbde_$name.so
echo "built tools/bin/libbde_$name.so ($*)"
