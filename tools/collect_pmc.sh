#!/bin/bash
# Three separate rocprofv3 counter passes (FETCH_SIZE | WRITE_SIZE | SQ+GRBM) of one command,
# as MI355X_MICROARCH.md prescribes; --pmc is never combined with other trace domains.
#   tools/collect_pmc.sh <tag> python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-check
# Output: gpurun_out/pmc_<tag>_{fetch,write,sq}/
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${tag}_fetch -- "$@" > gpurun_out/pmc_${tag}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${tag}_write -- "$@" > gpurun_out/pmc_${tag}_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d gpurun_out/pmc_${tag}_sq -- "$@" > gpurun_out/pmc_${tag}_sq.log 2>&1
echo "pmc passes done: $tag"
