#!/bin/bash
# usage: tools/pmc_run.sh <outdir> <pmc_kernel args...>   — three counter passes (SQ timing, LDS / instruction mix, HBM bytes)
out=$1; shift
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -o a -- python3 tools/pmc_kernel.py "$@" > $out.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --output-format csv -d $out/p2 -o a -- python3 tools/pmc_kernel.py "$@" >> $out.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/p3 -o a -- python3 tools/pmc_kernel.py "$@" >> $out.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/p4 -o a -- python3 tools/pmc_kernel.py "$@" >> $out.log 2>&1
python3 tools/pmc_summ.py $out conv
