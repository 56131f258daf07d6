#!/bin/sh
# Build a variant of libpraline_dp.so whose dp_split16 / dp_tb2 translation units are compiled with extra flags
# (A/B and ablation experiments: PRALINE_LIB=<path> selects the build at run time).
#   scripts/build_variant.sh <name> [flags for dp_split16_instance.hip / dp_tb2_instance.hip ...]
# -> variants/libpraline_dp_<name>.so   (git-ignored; travels to the GPU box with the snapshot)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
C=$ROOT/praline_amd/csrc
OUT=$ROOT/variants; mkdir -p $OUT/obj_$NAME
make -s -C $C all
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -Wall -Wno-unused-but-set-variable -fno-fast-math -ffp-contract=off -fno-honor-nans -mllvm -amdgpu-mfma-vgpr-form"
if [ -n "$VARIANT_PIPE_ONLY" ]; then cp $C/build/dp_split16.o $OUT/obj_$NAME/dp_split16.o; else /opt/rocm/bin/hipcc $FLAGS "$@" -c $C/dp_split16_instance.hip -o $OUT/obj_$NAME/dp_split16.o & fi
/opt/rocm/bin/hipcc $FLAGS "$@" -c $C/dp_pipe_instance.hip -o $OUT/obj_$NAME/dp_pipe.o &
if [ -n "$VARIANT_TB" ]; then /opt/rocm/bin/hipcc $FLAGS "$@" -c $C/dp_tb2_instance.hip -o $OUT/obj_$NAME/dp_tb2.o & else cp $C/build/dp_tb2.o $OUT/obj_$NAME/dp_tb2.o; fi
if [ -n "$VARIANT_QUAD" ]; then /opt/rocm/bin/hipcc ${FLAGS% -mllvm -amdgpu-mfma-vgpr-form} "$@" -c $C/dp_quad_instance.hip -o $OUT/obj_$NAME/dp_quad.o & fi
if [ -n "$VARIANT_PK16" ]; then /opt/rocm/bin/hipcc ${FLAGS% -mllvm -amdgpu-mfma-vgpr-form} "$@" -c $C/dp_pk16_instance.hip -o $OUT/obj_$NAME/dp_pk16.o & fi
if [ -n "$VARIANT_RAWB" ]; then /opt/rocm/bin/hipcc ${FLAGS% -mllvm -amdgpu-mfma-vgpr-form} "$@" -c $C/dp_rawb_instance.hip -o $OUT/obj_$NAME/dp_rawb.o & fi
if [ -n "$VARIANT_HOST" ]; then /opt/rocm/bin/hipcc ${FLAGS% -mllvm -amdgpu-mfma-vgpr-form} "$@" -c $C/praline_dp.hip -o $OUT/obj_$NAME/praline_dp.o & fi
if [ -n "$VARIANT_BATCH" ]; then for n in 2 8 10 12 14 16; do /opt/rocm/bin/hipcc ${FLAGS% -mllvm -amdgpu-mfma-vgpr-form} $VARIANT_BATCH_FLAGS "$@" -DPRALINE_NSTEP_INST=$n -c $C/dp_instance.hip -o $OUT/obj_$NAME/dp_instance_$n.o & done; fi
wait
OBJS=$(ls $C/build/*.o | grep -v "dp_split16.o\|dp_tb2.o\|dp_pipe.o")
if [ -n "$VARIANT_BATCH" ]; then OBJS="$(echo "$OBJS" | grep -v "dp_instance_") $(ls $OUT/obj_$NAME/dp_instance_*.o)"; fi
if [ -n "$VARIANT_HOST" ]; then OBJS="$(echo "$OBJS" | grep -v "praline_dp.o") $OUT/obj_$NAME/praline_dp.o"; fi
if [ -n "$VARIANT_QUAD" ]; then OBJS="$(echo "$OBJS" | grep -v "dp_quad.o") $OUT/obj_$NAME/dp_quad.o"; fi
if [ -n "$VARIANT_PK16" ]; then OBJS="$(echo "$OBJS" | grep -v "dp_pk16.o") $OUT/obj_$NAME/dp_pk16.o"; fi
if [ -n "$VARIANT_RAWB" ]; then OBJS="$(echo "$OBJS" | grep -v "dp_rawb.o") $OUT/obj_$NAME/dp_rawb.o"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $OUT/obj_$NAME/dp_split16.o $OUT/obj_$NAME/dp_pipe.o $OUT/obj_$NAME/dp_tb2.o -o $OUT/libpraline_dp_$NAME.so
echo built $OUT/libpraline_dp_$NAME.so
