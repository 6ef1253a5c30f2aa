"""Drop-in for /root/reference/src/util.py (same function names and file formats), without OpenCV.

Only `normal` is on the timed hot path (bilateral_filter uses it, process_functional.py:428); the file I/O is what
match.py needs around the timed region.  Image decode / PGM encode live in OpenCV + libpng in the reference
(match.py:118-119, util.py:52); neither is in this image, so parity of those two calls is unpinned:
  * read_gray() decodes with PIL and converts colour to grey with libpng's rgb_to_gray arithmetic (what
    cv2.imread(..., IMREAD_GRAYSCALE) asks of libpng for a PNG) - pinned bit for bit against the real libpng 1.6.37
    by fixtures (tests/golden/gen_png_gray.c); OpenCV itself is not in this image;
  * saveDisparity() writes the saturate-cast (round-half-even, clamp 0..255) uint8 map as binary PGM, which is what
    cv2.imwrite does with a float32 matrix.
"""
import argparse
import os
import struct

import numpy as np


def readPfm(filename):
    """util.py:6-25 - greyscale 'Pf', bottom-up rows, scale sign gives endianness."""
    with open(filename, 'rb') as f:
        line = f.readline().decode('latin-1')
        assert line.strip() == "Pf"  # one sample per pixel
        items = f.readline().decode('latin-1').strip().split()
        width = int(items[0])
        height = int(items[1])
        scale = float(f.readline().decode('latin-1').strip())
        dtype = "<f4" if scale < 0 else ">f4"
        data = np.frombuffer(f.read(4 * width * height), dtype=dtype).reshape(height, width)
    return np.ascontiguousarray(data[::-1]).astype(np.float32)


def parseCalib(filename):
    """util.py:27-43 - Middlebury calib.txt: lines 4, 5, 6 hold width=, height=, ndisp=."""
    with open(filename, "r") as f:
        lines = f.readlines()

    def value(line):
        line = line.strip()
        return int(line[line.find('=') + 1:])

    width = value(lines[4])
    height = value(lines[5])
    ndisp = value(lines[6])
    return height, width, ndisp


def normal(mean, std_dev):
    """util.py:45-48 - Gaussian pdf closure, evaluated in float64."""
    constant1 = 1. / (np.sqrt(2 * np.pi) * std_dev)
    constant2 = -1. / (2 * std_dev * std_dev)
    return lambda x: constant1 * np.exp(constant2 * ((x - mean) ** 2))


def saveDisparity(disparity_map, filename):
    """util.py:50-52 - cv2.imwrite of a float32 map: saturate-cast to uint8; .pgm -> binary P5."""
    assert len(disparity_map.shape) == 2
    with np.errstate(invalid="ignore"):
        u8 = np.clip(np.rint(np.nan_to_num(np.asarray(disparity_map, np.float64), nan=0.0)), 0, 255).astype(np.uint8)
    height, width = u8.shape
    ext = os.path.splitext(filename)[1].lower()
    if ext in (".pgm", ""):
        with open(filename, "wb") as o:
            o.write(("P5\n%d %d\n255\n" % (width, height)).encode())
            o.write(u8.tobytes())
    else:
        from PIL import Image
        Image.fromarray(u8).save(filename)


def writePfm(disparity_map, filename):
    """util.py:54-70 - 'Pf', '<width> <height>', '-1.0', then little-endian float32 rows bottom-up."""
    assert len(disparity_map.shape) == 2
    height, width = disparity_map.shape
    disparity_map = disparity_map.astype(np.float32)
    with open(filename, "wb") as o:
        o.write(b"Pf\n")
        o.write(("{} {}\n".format(width, height)).encode())
        o.write(b"-1.0\n")
        o.write(np.ascontiguousarray(disparity_map[::-1]).astype("<f4").tobytes())


def saveTimeFile(times, path):
    """util.py:72-75."""
    with open(path, "w") as o:
        o.write("{}".format(times))


def testMk(dirName):
    """util.py:77-79, made safe for several ranks creating the same directory at once."""
    os.makedirs(dirName, exist_ok=True)


def recurMk(path):
    """util.py:81-86 - mkdir -p."""
    os.makedirs(path if os.path.isabs(path) else os.path.join("/", path), exist_ok=True)


def read_gray(path):
    """Stand-in for cv2.imread(path, cv2.IMREAD_GRAYSCALE) (match.py:118-119): uint8 [H,W]."""
    from PIL import Image
    im = Image.open(path)
    if im.mode in ("L", "P", "1", "I;16", "I"):
        if im.mode != "L":
            im = im.convert("L")
        return np.asarray(im, dtype=np.uint8)
    rgb = np.asarray(im.convert("RGB"), dtype=np.uint32)
    # What OpenCV's PNG reader asks of libpng for IMREAD_GRAYSCALE: png_set_rgb_to_gray(png, 1, 0.299, 0.587).  libpng
    # turns the two weights into 15-bit integers by truncation (29900 * 32768 / 100000 = 9797, 58700 * 32768 / 100000
    # = 19234, blue = 32768 - 9797 - 19234 = 3737) and its 8-bit path truncates the weighted sum as well
    # (pngrtran.c, png_set_rgb_to_gray_fixed / png_do_rgb_to_gray: "(rc*red + gc*green + bc*blue) >> 15").
    # Pinned against the real libpng (1.6.37, decoding as OpenCV's PngDecoder sets it up) by the fixtures
    # tests/golden/png_color_*.png / png_gray_*.npy (generator: tests/golden/gen_png_gray.c); cv2 itself is not in
    # this image, so its own share (the call above) stays unpinned.
    gray = (rgb[:, :, 0] * 9797 + rgb[:, :, 1] * 19234 + rgb[:, :, 2] * 3737) >> 15
    return gray.astype(np.uint8)


class ExplicitStore(argparse.Action):
    """argparse action for `-g/--gpu`: stores the value like the default action and records that the flag was GIVEN
    (`<dest>_explicit`), whatever its spelling (-g 1, -g1, --gpu=1, an abbreviation such as --gp 1).  match.py and
    train.py pin the process to that card through HIP_VISIBLE_DEVICES only then - or when the environment is empty."""

    def __call__(self, parser, namespace, values, option_string=None):
        setattr(namespace, self.dest, values)
        setattr(namespace, self.dest + "_explicit", True)


def pin_gpu(args, world):
    """The reference pins the process to the requested card through the environment (match.py:59, train.py:57); ROCm
    reads HIP_VISIBLE_DEVICES, torch also honours CUDA_VISIBLE_DEVICES.  An explicit -g always wins; without one a
    scheduler's own HIP_VISIBLE_DEVICES stands (the default "0" only fills an empty environment); under torchrun
    (world > 1: one rank per GPU) -g is ignored."""
    if world == 1 and (getattr(args, "gpu_explicit", False) or "HIP_VISIBLE_DEVICES" not in os.environ):
        os.environ["HIP_VISIBLE_DEVICES"] = args.gpu
        os.environ["CUDA_VISIBLE_DEVICES"] = args.gpu
