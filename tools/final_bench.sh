#!/bin/bash
# Round-end measurement: the driver's command (JSON line: headline + every secondary workload with the reference's CPU path, its codec
# ratio and the drop-in measurement beside it), the single-workload lines WITH their cpu_baseline (a bounded sample of that tree), and --
# each in a run of its own, WITHOUT the secondary workloads, so that a kernel's average is over one workload -- rocprofv3 kernel stats
# of the headline, the compressible tree and the zstd tree, and the exact memory-side traffic (L2 request-size counters) of the three.
# usage: tools/final_bench.sh <tag>   -> gpurun_out/<tag>_*   (copy what is to be judged into profiles/)
tag=${1:-r06x}
mkdir -p gpurun_out
python bench.py > gpurun_out/${tag}_bench64g_default.json 2> gpurun_out/${tag}_bench64g_default.err
python bench.py --kind mixed --cpu-gib 4 --no-secondary > gpurun_out/${tag}_bench64g_mixed.json 2> gpurun_out/${tag}_bench64g_mixed.err
python bench.py --kind mixed --dups --cpu-gib 4 --no-secondary > gpurun_out/${tag}_bench64g_dedup.json 2> gpurun_out/${tag}_bench64g_dedup.err
python bench.py --codec zstd --kind mixed --cpu-gib 4 --no-secondary > gpurun_out/${tag}_bench64g_zstd_mixed.json 2> gpurun_out/${tag}_bench64g_zstd_mixed.err
python bench.py --codec zstd --zstd-settings 4 --kind mixed --no-cpu-baseline --no-secondary > gpurun_out/${tag}_bench64g_zstd4_mixed.json 2> gpurun_out/${tag}_bench64g_zstd4_mixed.err
# BASELINE.json configs[4]'s shape (4 x 16 GiB PAK-style files, ZStd) on data the codec can compress, and on random bytes
python bench.py --codec zstd --file-mib 16384 --kind mixed --cpu-gib 4 --no-secondary > gpurun_out/${tag}_bench64g_zstd_4x16g_mixed.json 2> gpurun_out/${tag}_bench64g_zstd_4x16g_mixed.err
python bench.py --codec zstd --file-mib 16384 --cpu-gib 4 --no-secondary > gpurun_out/${tag}_bench64g_zstd_4x16g.json 2> gpurun_out/${tag}_bench64g_zstd_4x16g.err
here=$(pwd); cd /tmp && export TMPDIR=/tmp; cd $here
for leg in "default:" "mixed:--kind mixed" "zstd_mixed:--codec zstd --kind mixed"; do
  name=${leg%%:*}; args=${leg#*:}
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_$name -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary $args > gpurun_out/${tag}_prof_$name.log 2>&1
  cp gpurun_out/${tag}_prof_$name/p_kernel_stats.csv gpurun_out/${tag}_bench64g_${name}_kernel_stats.csv 2>/dev/null
  rm -rf gpurun_out/${tag}_prof_$name
done
tools/pmc_exact_traffic.sh gpurun_out/${tag}_xtraffic_8g.json $((8<<30)) --gib 8 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > gpurun_out/${tag}_xtraffic_8g.log 2>&1
tools/pmc_exact_traffic.sh gpurun_out/${tag}_xtraffic_8g_mixed.json $((8<<30)) --gib 8 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --kind mixed > gpurun_out/${tag}_xtraffic_8g_mixed.log 2>&1
tools/pmc_exact_traffic.sh gpurun_out/${tag}_xtraffic_8g_zstd_mixed.json $((8<<30)) --gib 8 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --kind mixed --codec zstd > gpurun_out/${tag}_xtraffic_8g_zstd_mixed.log 2>&1
tail -c 400 gpurun_out/${tag}_bench64g_default.json; head -8 gpurun_out/${tag}_bench64g_mixed_kernel_stats.csv | cut -c1-200
