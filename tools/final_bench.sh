#!/bin/bash
# Round-end measurement: default bench (JSON line), the same command under rocprofv3 kernel stats, and the
# compressible variant.  usage: tools/final_bench.sh <tag>   -> gpurun_out/<tag>_*
tag=${1:-r01x}
mkdir -p gpurun_out
python bench.py > gpurun_out/${tag}_bench64g_default.json 2> gpurun_out/${tag}_bench64g_default.err
python bench.py --kind mixed --no-cpu-baseline > gpurun_out/${tag}_bench64g_mixed.json 2> gpurun_out/${tag}_bench64g_mixed.err
python bench.py --tree mixed-sizes --no-cpu-baseline > gpurun_out/${tag}_bench64g_mixed_sizes.json 2> gpurun_out/${tag}_bench64g_mixed_sizes.err
python bench.py --codec zstd --kind mixed --no-cpu-baseline > gpurun_out/${tag}_bench64g_zstd_mixed.json 2> gpurun_out/${tag}_bench64g_zstd_mixed.err
python bench.py --codec zstd --file-mib 16384 --no-cpu-baseline > gpurun_out/${tag}_bench64g_zstd_4x16g.json 2> gpurun_out/${tag}_bench64g_zstd_4x16g.err
here=$(pwd); cd /tmp && export TMPDIR=/tmp; cd $here
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_prof.log 2>&1
cp gpurun_out/${tag}_prof/p_kernel_stats.csv gpurun_out/${tag}_bench64g_kernel_stats.csv 2>/dev/null
rm -f gpurun_out/${tag}_prof/p_kernel_trace.csv
tail -c 600 gpurun_out/${tag}_bench64g_default.json; head -12 gpurun_out/${tag}_bench64g_kernel_stats.csv
