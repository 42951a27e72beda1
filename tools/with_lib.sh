#!/bin/bash
# Run a command with a variant build of the library in place of the in-tree one (GPU box, scratch copy of the repo):
#   bash tools/with_lib.sh variants/lib_x.so python tools/bvh2_bench.py
lib=$1; shift
cp rayz_amd/csrc/librayz_hip.so /tmp/librayz_hip.keep && cp "$lib" rayz_amd/csrc/librayz_hip.so && "$@"; rc=$?
cp /tmp/librayz_hip.keep rayz_amd/csrc/librayz_hip.so
exit $rc
