"""box calibration (vt_calibrate) a few times in a row: f16 MFMA TFLOP/s, sustained shader MHz, L2 -> register delivery TB/s.  usage: calib.py [reps=5]"""
import sys, ctypes as C; sys.path.insert(0, '/root/repo')
import torch
from vistracker_amd import _lib as L
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
work = torch.empty(L.lib().vt_calibrate_workspace_bytes(), dtype=torch.uint8, device="cuda")
out = (C.c_double * 8)()
for i in range(reps):
    L.check(L.lib().vt_calibrate(work.data_ptr(), out, L.stream_ptr()))
    print(f"calibration {i}: mfma_f16 {out[0]:.1f} TFLOP/s ({out[0] / 2516.6:.3f} of 2516.6), shader clock {out[1]:.0f} MHz, L2->register delivery {out[2]:.2f} TB/s, "
          f"kernels {out[3]:.2f} + {out[4]:.2f} ms")
