"""Imported FIRST by the tools that set LTHIP_* ablation switches: those exist in the ablation build only (`make ablations`:
build/ablations/liblongtail_hip.so, -DLTHIP_ABLATIONS), so the process is pointed at that library before longtail_amd.lib loads."""
import os
from pathlib import Path

os.environ.setdefault("LTHIP_LIB_PATH", str(Path(__file__).resolve().parent.parent / "build" / "ablations" / "liblongtail_hip.so"))
