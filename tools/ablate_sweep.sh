#!/bin/bash
# Stage ablation of k_graph_sweep: builds libmpfmt variants with -DSWEEP_ABL=n into build_abl/ (here, no GPU needed),
# `tools/ablate_sweep.sh run` then times each on the GPU box through MPFMT_LIB_PATH.  Ablated kernels give wrong masks.
set -e
cd "$(dirname "$0")/.."
C=motionplanning.jl_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form=1 -Iinclude"
if [ "$1" = "run" ]; then
  for n in 0 1 2 3 4 5; do
    printf "SWEEP_ABL=%s " $n
    MPFMT_OPT_SWEEP_SORTED=${SORTED:-0} MPFMT_LIB_PATH=$PWD/build_abl/libmpfmt_abl$n.so timeout 200 python bench.py --no-cpu-baseline --no-solve --steps 20 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['submetrics']['kernel_ms']; print(k['sweep_graph'], k['rdisc_sort'], d['ms_per_step'])"
  done
  exit 0
fi
mkdir -p build_abl
OTHERS=$(ls $C/*.o | grep -v kernels_sweep.o)
for n in 0 1 2 3 4 5; do
  /opt/rocm/bin/hipcc $FLAGS -DSWEEP_ABL=$n -c $C/kernels_sweep.hip -o build_abl/sweep_abl$n.o &
done
wait
for n in 0 1 2 3 4 5; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_abl/libmpfmt_abl$n.so $OTHERS build_abl/sweep_abl$n.o -ldl
  rm build_abl/sweep_abl$n.o
done
ls -la build_abl
