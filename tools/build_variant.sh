#!/bin/bash
# Builds a variant of liblins_ieskf.so for A/B timing into ab/<name>.so.  usage: tools/build_variant.sh name -DLINS_X=1 ...
# VARIANT_MAKE="VAR=value ..." adds make variables (e.g. VARIANT_MAKE="FLAGS_map_kernels=" builds map_kernels.hip WITH the
# SLP vectoriser: the canary of tests/test_gpu_canaries.py)
name=$1; shift
cd "$(dirname "$0")/../lins---lidar-inertial-slam_amd/csrc" && mkdir -p ../../ab/$name &&
  make -j32 BUILD=build_$name OUT=../../ab/$name EXTRA="$*" $VARIANT_MAKE ../../ab/$name/liblins_ieskf.so > /tmp/build_$name.log 2>&1 &&
  mv ../../ab/$name/liblins_ieskf.so ../../ab/$name.so && rmdir ../../ab/$name && rm -rf build_$name && echo "built ab/$name.so ($*)"
