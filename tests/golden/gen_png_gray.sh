#!/bin/bash
# Regenerates tests/golden/png_color_*.png and png_gray_*.npy: colour PNGs written by PIL (seeded noise + gradients,
# RGB and RGBA) and their grey decode by the real libpng of this image (gen_png_gray.c).  Container-only.
set -e
cd "$(dirname "$0")"
gcc gen_png_gray.c -I/opt/conda/include -L/opt/conda/lib -lpng16 -lz -o /tmp/gen_png_gray
python - <<'PY'
import numpy as np
from PIL import Image
rng = np.random.RandomState(7)
a = rng.randint(0, 256, size=(24, 40, 3)).astype(np.uint8)
a[:8] = np.stack(np.meshgrid(np.arange(40) * 6, np.arange(8) * 30), -1).sum(-1)[..., None] % 256   # grey ramp rows
a[8:12, :, 0], a[8:12, :, 1], a[8:12, :, 2] = 255, np.arange(40) * 6, 0                            # saturated rows
Image.fromarray(a, "RGB").save("png_color_rgb.png")
b = np.concatenate([rng.randint(0, 256, size=(16, 24, 3)), rng.randint(0, 256, size=(16, 24, 1))], -1).astype(np.uint8)
Image.fromarray(b, "RGBA").save("png_color_rgba.png")
PY
for n in rgb rgba; do
  LD_LIBRARY_PATH=/opt/conda/lib /tmp/gen_png_gray png_color_$n.png /tmp/gray_$n.raw
done
python - <<'PY'
import numpy as np
for n, shape in (("rgb", (24, 40)), ("rgba", (16, 24))):
    np.save("png_gray_%s.npy" % n, np.fromfile("/tmp/gray_%s.raw" % n, dtype=np.uint8).reshape(shape))
PY
ls -la png_color_*.png png_gray_*.npy
