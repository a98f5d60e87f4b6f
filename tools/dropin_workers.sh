for a in "--cpu-gib 8" "--kind mixed --cpu-gib 4" "--kind mixed --codec zstd --cpu-gib 4"; do python bench.py --gib 8 --steps 1 --warmup 0 --no-secondary --no-live-traffic $a 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); d=j['secondary']['drop_in']
print(d.get('hip_plugins_by_workers'), 'cpu', d['upsync_GBps']['cpu_plugins'], d.get('error'))"; done
