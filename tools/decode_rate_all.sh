#!/bin/bash
# The restore-side table of DESIGN.md §6: tools/decode_rate.py over the data kinds and 512 / 64 / 1 blocks of 8 MiB in flight,
# then the serial paths and per-kernel times of the 4 GiB mixed case.  usage: tools/decode_rate_all.sh > profiles/<tag>_decode_rate.txt
# (the LTHIP_* switches used here exist in the ablation build only: `make ablations`)
export LTHIP_LIB_PATH=${LTHIP_LIB_PATH:-$(cd "$(dirname "$0")/.." && pwd)/build/ablations/liblongtail_hip.so}
for k in mixed records tokens lines random; do
  for g in 4 0.5 0.0078125; do
    echo "== $k, $g GiB"
    timeout 300 python tools/decode_rate.py $g $k 2>&1 | grep "^lz4:\|^zstd:"
  done
done
echo "== mixed, 4 GiB, serial paths (LTHIP_LZ4_SERIAL_DECODER=1; LTHIP_ZSTD_DBG=1: one wave per payload)"
LTHIP_LZ4_SERIAL_DECODER=1 LTHIP_ZSTD_DBG=1 timeout 600 python tools/decode_rate.py 4 mixed 2>&1 | grep "^lz4:\|^zstd:"
echo "== mixed, 4 GiB, one block per piece (LTHIP_ZSTD_SUB=0: k_zstd_prepare + k_zstd_execute<false>)"
LTHIP_ZSTD_SUB=0 timeout 300 python tools/decode_rate.py 4 mixed 2>&1 | grep "^zstd:"
echo "== payloads made by the REFERENCE encoders (sliding-window LZ4 blocks; zstd frames whose blocks depend on each other), mixed"
for n in 512 64 1; do
  timeout 900 python tools/decode_rate_ref.py $n mixed 2>&1 | grep "^lz4\|^zstd"
done
echo "== the same zstd frames, one wave per payload (LTHIP_ZSTD_DBG=8: no block-parallel path for other encoders' frames)"
LTHIP_ZSTD_DBG=8 timeout 900 python tools/decode_rate_ref.py 512 mixed 2>&1 | grep "^zstd"
echo "== per-kernel times, mixed, 4 GiB (rocprofv3 --kernel-trace --stats; two decodes of each codec in the run)"
bash tools/prof_decode.sh 4 mixed dra 2>&1 | grep -v "Opened result\|^lz4:\|^zstd:" | grep "k_lz4_pd\|k_zstd_sub\|k_zstd_exec\|k_zstd_prep\|k_zstd_plain\|k_zstd_split\|k_zstd_decode"
echo "== per-kernel times, 512 reference-made payloads (two decodes of each codec in the run)"
bash tools/prof_ref.sh 512 mixed drb 2>&1 | grep "k_lz4_pd\|k_zstd_blk\|k_zstd_exec\|k_zstd_split\|k_zstd_decode"
