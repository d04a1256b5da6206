"""Condense the reference's own render examples/demo.png (README.md:192-195: `demo --width 1920
--height 1080 --anti-aliasing 3`, seeds 42/54, factor 1, gamma 1) into a small soft golden:
40x40-pixel block means of the 8-bit image and the colour of its (noise-free) sky rows.
Run in the build container, where /root/reference exists:  python tools/make_golden_demo_png.py"""
import numpy as np
from PIL import Image

ref = np.asarray(Image.open("/root/reference/examples/demo.png").convert("RGB")).astype(np.float64)
H, W, b = 1080, 1920, 40
blocks = ref.reshape(H // b, b, W // b, b, 3).mean((1, 3)).astype(np.float32)
sky = ref[:400].reshape(-1, 3)
assert (sky == sky[0]).all()
np.savez_compressed("tests/golden/demo_png_soft.npz", block40=blocks, sky_rgb=sky[0].astype(np.uint8), sky_rows=np.int32(400))
print(blocks.shape, sky[0])
