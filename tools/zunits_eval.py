#!/usr/bin/env python
"""Entropy-stage experiments without a GPU: runs the host model of the zstd piece encoder (oracle/zstd_model.c = zstd_block_core.h with one
lane) over match-finder units dumped on the GPU box by tools/zunits_dump.py, for several ZbInput.flags, and decodes every frame it builds
with the REFERENCE decoder (oracle/_ref).  usage: tools/zunits_eval.py gpurun_out/zunits_q0.npz [flags,flags,...]"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from tests._libs import oracle as get_oracle, ref as get_ref

o = get_oracle()
try:
    r = get_ref()
except Exception:  # no reference build here: sizes only
    r = None
d = o.dll
d.ltz_model_encode_block_src.restype = C.c_uint32
d.ltz_model_encode_block_src.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
d.ltz_model_last_sub.restype = C.POINTER(C.c_uint16)
d.ltz_model_sub_blocks(1)
z = np.load(sys.argv[1])
flag_list = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,2").split(",")]
kinds = sorted({k.rsplit("_", 1)[0] for k in z.files})
print(f"{'kind':8s} " + " ".join(f"{'flags=' + str(f):>16s}" for f in flag_list) + "   (ratio; frame bytes of the kernel at dump time)")
for kind in ("mixed", "records", "tokens", "lines", "text"):
    if kind not in kinds:
        continue
    raw, meta, lits, recs = z[kind + "_raw"], z[kind + "_meta"], z[kind + "_lits"], z[kind + "_recs"]
    n = len(raw)
    cols = []
    for flags in flag_list:
        d.ltz_model_flags(flags)
        frame = bytearray(b"\x28\xb5\x2f\xfd\xe0" + int(n).to_bytes(8, "little"))
        for i in range((n + 131071) // 131072):
            size = min(131072, n - i * 131072)
            nu = (size + 4095) // 4096
            last = (i + 1) * 131072 >= n
            m, l, rc = (np.ascontiguousarray(a[i * 32 : i * 32 + nu]) for a in (meta, lits, recs))
            piece = np.ascontiguousarray(raw[i * 131072 : i * 131072 + size])
            out = np.zeros(140000, np.uint8)
            cs = d.ltz_model_encode_block_src(m.ctypes.data, l.ctypes.data, rc.ctypes.data, nu, size, piece.ctypes.data, out.ctypes.data)
            if cs:
                if last:
                    sub = d.ltz_model_last_sub()
                    out[cs - 3 - (sub[nu - 1] & 0x7FFF)] |= 1
                frame += out[:cs].tobytes()
            else:
                h = (1 if last else 0) | (size << 3)
                frame += h.to_bytes(3, "little") + piece.tobytes()
        if r is not None:
            err, back = r.decompress(1, np.frombuffer(bytes(frame), np.uint8).copy(), n)
            assert err == 0 and (back == raw).all(), (kind, flags, err)
        cols.append(n / len(frame))
    print(f"{kind:8s} " + " ".join(f"{c:16.4f}" for c in cols) + f"   ({n / int(z[kind + '_frame'][0]):.4f})")
