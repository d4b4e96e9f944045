#!/bin/bash
# A side-by-side library that differs from the built one in ONE translation unit compiled with extra flags (A/B of kernel
# variants behind -D macros): the other objects are copied, so a variant costs one compile and a link.
# Usage: tools/build_variant.sh <tag> <source.hip> <flags...>   ->  geosss_amd/libgsss_<tag>.so  (load with GSSS_HIP_LIB)
set -eu
TAG=$1; SRC=$2; shift 2
cd "$(dirname "$0")/.."
OBJ=geosss_amd/csrc/_obj_libgsss_$TAG
rm -rf $OBJ; cp -rp geosss_amd/csrc/_obj $OBJ; rm -f $OBJ/${SRC%.hip}.o
GSSS_HIPCC_FLAGS="$*" python -m geosss_amd.build --out geosss_amd/libgsss_$TAG.so 2>&1 | tail -1
