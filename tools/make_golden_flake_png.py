"""Condense the reference's own render examples/flake.png (README.md:276: `render --width 1280 --height 720
--anti-aliasing 3 flake.yml`, default seeds / factor / gamma) into a small soft golden: 40x40-pixel block
means of the 8-bit image.  Run in the build container, where /root/reference exists:
python tools/make_golden_flake_png.py"""
import numpy as np
from PIL import Image

ref = np.asarray(Image.open("/root/reference/examples/flake.png").convert("RGB")).astype(np.float64)
H, W, b = 720, 1280, 40
blocks = ref.reshape(H // b, b, W // b, b, 3).mean((1, 3)).astype(np.float32)
np.savez_compressed("tests/golden/flake_png_soft.npz", block40=blocks, mean_rgb=ref.mean((0, 1)))
print(blocks.shape, ref.mean((0, 1)))
