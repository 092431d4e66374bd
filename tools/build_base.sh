#!/bin/bash
# Builds the library of a git revision (default HEAD) beside the working tree's, for tools/ab_lib.sh:
#   tools/build_base.sh [rev]   ->  tools/probe/libganrev_base.so (rev) and tools/probe/libganrev_new.so (copy of the working tree's build)
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$ROOT/gan-reverser_amd/csrc/build_base
rm -rf $W && mkdir -p $W/src/gan-reverser_amd/csrc $W/src/include $W/src/gan-reverser_amd/ganrev
git -C $ROOT archive $REV gan-reverser_amd/csrc include | tar -x -C $W/src
make -C $W/src/gan-reverser_amd/csrc -j8 > $W/build.log 2>&1 || { tail -20 $W/build.log; exit 1; }
mkdir -p $ROOT/tools/probe
cp $W/src/gan-reverser_amd/ganrev/libganrev.so $ROOT/tools/probe/libganrev_base.so
cp $ROOT/gan-reverser_amd/ganrev/libganrev.so $ROOT/tools/probe/libganrev_new.so
ls -la $ROOT/tools/probe/libganrev_base.so $ROOT/tools/probe/libganrev_new.so
