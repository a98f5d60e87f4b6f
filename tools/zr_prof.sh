cp longtail_amd/liblongtail_hip.so build/cur.so
cp build/prof/liblongtail_hip_prof.so longtail_amd/liblongtail_hip.so
python - "$@" <<'PY'
import sys, runpy
sys.argv = ["tools/decode_rate_ref.py"] + sys.argv[1:]
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import longtail_amd.lib as L
g = runpy.run_path("tools/decode_rate_ref.py")
L.load().dll.lthip_zb_prof_dump()
PY
cp build/cur.so longtail_amd/liblongtail_hip.so
