cp longtail_amd/liblongtail_hip.so build/cur.so
cp build/prof/liblongtail_hip_prof.so longtail_amd/liblongtail_hip.so
for kind in mixed tokens records lines; do
python - $kind <<'PY'
import sys, os
kind = sys.argv[1]
sys.argv = ["k5_probe.py", "2", "0", kind, "lz4", "12345"]
sys.path.insert(0, "tools")
import longtail_amd.lib as L
exec(open("tools/k5_probe.py").read())
L.load().dll.lthip_k5_prof_dump(1)
PY
done
cp build/cur.so longtail_amd/liblongtail_hip.so
