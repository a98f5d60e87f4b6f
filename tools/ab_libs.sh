#!/bin/bash
# same-box A/B of two builds of the library: usage tools/ab_libs.sh <libA> <libB> <rounds> -- <bench args...>
# prints value / ms_per_step / ratio / match-finder ms of every run, alternating A and B
A=$1; B=$2; R=$3; shift 4
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); k=d['kernels']; print(d['value'], 'GB/s', d['ms_per_step'], 'ms ratio', d['result']['ratio'], 'match finder', k['lz4_segments']['ms_per_step'], 'stitch', k.get('lz4_stitch',{}).get('ms_per_step'), 'zstd_encode', k.get('zstd_encode',{}).get('ms_per_step'))"; }
for r in $(seq $R); do for L in "$A" "$B"; do echo -n "$(basename $(dirname $L)) [$*]: "; LTHIP_LIB_PATH=$(pwd)/$L python bench.py "$@" --no-cpu-baseline --no-secondary --no-live-traffic 2>/dev/null | line; done; done
