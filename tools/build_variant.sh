#!/bin/bash
# A/B builds of the library with one source compiled under extra flags (experiment knobs, timing-only builds):
#   tools/build_variant.sh NAME SOURCE.hip [-DFLAG ...]  ->  gpurun_variants/libadvstep_NAME.so   (select with ADVSTEP_LIB=...)
# The other sources are compiled once into /tmp/objs and re-linked.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; shift 2
CS=$ROOT/audio_deepfake_adversarial_attacks_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -I$ROOT/include"
mkdir -p /tmp/objs $ROOT/gpurun_variants
for f in $CS/*.hip; do
  b=$(basename $f .hip)
  if [ "$b" != "$SRC" ] && { [ ! -f /tmp/objs/$b.o ] || [ $f -nt /tmp/objs/$b.o ]; }; then hipcc $FLAGS -c $f -o /tmp/objs/$b.o 2>/dev/null & fi
done
hipcc $FLAGS "$@" -c $CS/$SRC.hip -o /tmp/objs/${SRC}_$NAME.o 2>/dev/null
wait
OBJS=""
for f in $CS/*.hip; do b=$(basename $f .hip); if [ "$b" = "$SRC" ]; then OBJS="$OBJS /tmp/objs/${SRC}_$NAME.o"; else OBJS="$OBJS /tmp/objs/$b.o"; fi; done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $ROOT/gpurun_variants/libadvstep_$NAME.so
echo built gpurun_variants/libadvstep_$NAME.so
