#!/bin/bash
# tools/build_variant.sh TAG "EXTRA FLAGS" [objects to rebuild...]
# A/B builds of the encoder: copies dcdf_amd/csrc/_build to _build_TAG, rebuilds the named objects (default: the
# sidelen-256 int32-row instantiation bench.py times) with the extra flags and links dcdf_amd/libdcdf_k2r_TAG.so.
# Run it with DCDF_K2R_LIB=dcdf_amd/libdcdf_k2r_TAG.so.
set -e
cd "$(dirname "$0")/../dcdf_amd/csrc"
TAG=$1; EXTRA=$2; shift 2 || true
OBJS=${@:-enc_L8_P0_V1.o}
make -s -j8
rm -rf _build_$TAG && cp -rp _build _build_$TAG
for o in $OBJS; do rm -f _build_$TAG/$o; done
make -s -j8 OUT=_build_$TAG LIB=../libdcdf_k2r_$TAG.so EXTRA="$EXTRA"
ls -la ../libdcdf_k2r_$TAG.so
