#!/bin/bash
# Kernel statistics + counter passes of the bench command (memory traffic, MFMA busy, the issue / wait split of the attention
# kernel), and kernel statistics + memory-traffic passes of the pooling kernel at a chip-filling size.  Counter passes carry
# --pmc only (no trace options).  Run on the GPU box from the repo root:  bash scripts/collect_profiles.sh r03
# Raw output goes to gpurun_out/prof, the summaries to gpurun_out/prof_summary (copy those into profiles/).
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-reference-points --no-other-configs > $OUT/stats.log 2>&1 || exit 1
echo stats done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pool -o pool -- python3 $ROOT/scripts/pool_saturated.py > $OUT/pool.log 2>&1 || exit 1
echo pool done
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-reference-points --no-other-configs > $OUT/fetch.log 2>&1 || exit 1
echo fetch done
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-reference-points --no-other-configs > $OUT/write.log 2>&1 || exit 1
echo write done
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-reference-points --no-other-configs > $OUT/mfma.log 2>&1 || exit 1
echo mfma done
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/issue -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-reference-points --no-other-configs > $OUT/issue.log 2>&1 || exit 1
echo issue done
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/poolfetch -o pool -- python3 $ROOT/scripts/pool_saturated.py > $OUT/poolfetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/poolwrite -o pool -- python3 $ROOT/scripts/pool_saturated.py > $OUT/poolwrite.log 2>&1 || exit 1
echo pool counters done
python3 $ROOT/scripts/build_profiles.py $OUT $ROOT/gpurun_out/prof_summary $TAG
