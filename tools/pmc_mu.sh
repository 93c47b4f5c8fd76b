#!/bin/bash
# Collect SQ counters of fg_mu_kernel (one rocprofv3 --pmc pass per counter group).
# usage: tools/pmc_mu.sh <tag> [bench args...]   (run on the GPU box, from the repo root)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
G2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH"
G3="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_IFETCH"
i=0
for g in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $g --output-format csv -d gpurun_out/pmc_${tag}_g$i -- python bench.py --no-cpu-baseline "$@" > gpurun_out/pmc_${tag}_g$i.log 2>&1 || exit 1
done
