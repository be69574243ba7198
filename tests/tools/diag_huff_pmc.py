#!/usr/bin/env python3
"""One configuration of the device entropy decoder (16 x 4K files, one restart interval per MCU row) for
rocprofv3 --pmc: instructions per symbol step = SQ_INSTS_* / (waves x steps)."""
import io, os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from PIL import Image
from ffpic_amd import capi, ops
L = capi.require_device(0)
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:2160, 0:3840]
img = np.stack([128 + 100 * np.sin(xx / 37.0) * np.cos(yy / 23.0), 128 + 90 * np.cos(xx / 11.0 + yy / 53.0), (xx * 255 / 3839 + yy * 255 / 2159) / 2], axis=2)
img = np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)
bio = io.BytesIO(); Image.fromarray(img).save(bio, "JPEG", quality=85, subsampling=2, restart_marker_rows=1); data = bio.getvalue()
g, cy, cu, cv, q = ops.jpeg_entropy_batch_gpu([data] * 16, n_threads=8)
nz = int(np.count_nonzero(cy) + np.count_nonzero(cu) + np.count_nonzero(cv)) // 16
blocks = g.y_blocks + 2 * g.c_blocks
print("per picture: nonzero coefficients", nz, "blocks", blocks, "-> symbols about", nz + blocks, "; per interval", (nz + blocks) // g.mcu_rows)
