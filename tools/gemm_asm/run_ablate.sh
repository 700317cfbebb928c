#!/bin/bash
# GPU-box job: where the one-wave-per-SIMD asm loop loses its cycles — filler-rate variants and ablations (DMA loads dropped, fragment
# reads dropped; wrong numerics) on random and zero operands.   gpurun -- bash tools/gemm_asm/run_ablate.sh
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/gemm_asm
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LOG=$OUT/r04_gemm_asm_ablate.log
: > $LOG
M=$((214 * 1376))
GEN=${GEN:-gen_loop.py}
build() {  # name, generator flags
  python3 $REPO/tools/gemm_asm/$GEN $2 > $REPO/tools/gemm_asm/loop_body.inc
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -w $REPO/tools/gemm_asm/gemm_asm_lab.hip -o $OUT/lab_$1 || echo "build failed: $1" | tee -a $LOG
}
run() {  # name, extra args
  echo "== $1 $2" | tee -a $LOG
  timeout 300 $OUT/lab_$1 $2 --secs 1 $M 2048 1024 $M 4096 1024 $M 1024 4096 2>&1 | grep -v "^check" | tee -a $LOG
}
IFS=';' read -ra VARS <<< "${VARIANTS:-base|--rate0 1.0 --rate1 2.0;even|--rate0 0.5 --rate1 1.1;nodma|--ablate dma;noread|--ablate read;neither|--ablate dma,read}"
for v in "${VARS[@]}"; do
  name=${v%%|*}; flags=${v#*|}
  build $name "$flags"
  nc=""; case "$flags" in *ablate*) nc="--nocheck";; esac
  run $name "$nc"
  run $name "$nc --zero"
done
rm -f $OUT/lab_*
echo done
