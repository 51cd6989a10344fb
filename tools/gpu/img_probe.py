"""Image-stage kernels alone at a given resolution (no tracking running beside them): 60 pyramid builds + ORB planes + corner response."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import larvio_amd
from larvio_amd import ops
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
rng = np.random.default_rng(1)
y, x = np.mgrid[0:h, 0:w].astype(np.float32)
img = np.clip(110 + 70 * np.sin(x / 37.0) * np.cos(y / 23.0) + rng.normal(0, 12, (h, w)), 0, 255).astype(np.uint8)
ctx = larvio_amd.Context(0)
p = ops.Pyramid(ctx, w, h, 21, 3)
for _ in range(60):
    p.build(img, clahe=True); p.orb_prepare(); p.min_eigen_map()
ctx.sync()
