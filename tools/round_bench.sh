#!/bin/bash
# Round measurement set.  usage: tools/round_bench.sh <tag>   -> gpurun_out/<tag>_*  (copy what should be judged into profiles/)
#   <tag>_bench64g_default.json      the driver's command (headline + secondary + cpu_baseline)
#   <tag>_bench64g_kernel_stats.csv  rocprofv3 --kernel-trace --stats of the headline configuration
#   <tag>_bench64g_zstd_*.json       the zstd configurations (compressible; BASELINE.json configs[4] shape)
#   <tag>_pmc_traffic_8g.json        HBM traffic per kernel (FETCH_SIZE / WRITE_SIZE in separate passes, guide's correction)
tag=${1:-r02x}
mkdir -p gpurun_out
python bench.py > gpurun_out/${tag}_bench64g_default.json 2> gpurun_out/${tag}_bench64g_default.err
python bench.py --codec zstd --kind mixed --no-cpu-baseline > gpurun_out/${tag}_bench64g_zstd_mixed.json 2> gpurun_out/${tag}_bench64g_zstd_mixed.err
python bench.py --codec zstd --file-mib 16384 --no-cpu-baseline > gpurun_out/${tag}_bench64g_zstd_4x16g.json 2> gpurun_out/${tag}_bench64g_zstd_4x16g.err
python bench.py --codec zstd --file-mib 16384 --kind mixed --no-cpu-baseline > gpurun_out/${tag}_bench64g_zstd_4x16g_mixed.json 2> gpurun_out/${tag}_bench64g_zstd_4x16g_mixed.err
here=$(pwd); cd /tmp && export TMPDIR=/tmp; cd $here
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/${tag}_prof.log 2>&1
cp gpurun_out/${tag}_prof/p_kernel_stats.csv gpurun_out/${tag}_bench64g_kernel_stats.csv 2>/dev/null
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_mixed -o p -- python bench.py --steps 2 --warmup 1 --kind mixed --no-cpu-baseline > gpurun_out/${tag}_prof_mixed.log 2>&1
cp gpurun_out/${tag}_prof_mixed/p_kernel_stats.csv gpurun_out/${tag}_bench64g_mixed_kernel_stats.csv 2>/dev/null
rm -f gpurun_out/${tag}_prof*/p_kernel_trace.csv
bash tools/pmc_exact_traffic.sh gpurun_out/${tag}_xtraffic_8g.json $((8 << 30)) --gib 8 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > gpurun_out/${tag}_pmc_traffic.log 2>&1; bash tools/pmc_exact_traffic.sh gpurun_out/${tag}_xtraffic_8g_mixed.json $((8 << 30)) --gib 8 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --kind mixed >> gpurun_out/${tag}_pmc_traffic.log 2>&1

tail -c 400 gpurun_out/${tag}_bench64g_default.json; echo; head -8 gpurun_out/${tag}_bench64g_kernel_stats.csv; tail -12 gpurun_out/${tag}_pmc_traffic.log
