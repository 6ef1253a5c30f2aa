"""Patch sampler for training the matching network - drop-in for /root/reference/src/datagenerator.py
(ImageDataGenerator, :13-240): same constructor arguments, attributes and methods, NumPy only.

A mini-batch is drawn from ONE stereo pair (datagenerator.py:137-216): `batch_size` pixels with a finite, non-occluded
ground-truth disparity; for each one the left patch centred on it, a positive right patch centred within
`dataset_pos` pixels of the true match and a negative one displaced by `dataset_neg_low .. dataset_neg_high` pixels to
either side.  Patches are cut from images zero-padded by (patch-1)/2, so border pixels are legal centres.

Differences a maintainer should know:
  * images are decoded by util.read_gray (PIL + libpng's grey conversion, pinned in tests) instead of cv2;
  * the per-pixel Python loops are replaced by vectorised draws with the same distributions and the same rejection
    rules (int() truncation of the displaced column, redraw until it lies inside the image); the stream of random
    numbers is therefore not the reference's - pass `rng` (a numpy Generator) for reproducible batches.
"""
import numpy as np

from util import read_gray, readPfm


class ImageDataGenerator(object):

    def __init__(self, left_image_list_file, shuffle=False, patch_size=(11, 11), in_left_suffix='im0.png',
                 in_right_suffix='im1.png', gt_suffix='disp0GT.pfm', dataset_neg_low=1.5, dataset_neg_high=6,
                 dataset_pos=0.5, rng=None):
        self.shuffle = shuffle
        self.patch_size = patch_size
        self.in_left_suffix = in_left_suffix
        self.in_right_suffix = in_right_suffix
        self.gt_suffix = gt_suffix
        self.dataset_neg_low = dataset_neg_low
        self.dataset_neg_high = dataset_neg_high
        self.dataset_pos = dataset_pos
        self.rng = rng if rng is not None else np.random.default_rng()
        self.pointer = 0          # which pair the next batch comes from (datagenerator.py:44-46)
        self.read_image_list(left_image_list_file)
        self.prefetch()
        if self.shuffle:
            self.shuffle_data()

    def read_image_list(self, image_list):
        """Right-view and ground-truth paths follow from the left path by suffix replacement (:54-70)."""
        with open(image_list) as f:
            self.left_paths = [line.strip() for line in f if line.strip()]
        self.right_paths = [p.replace(self.in_left_suffix, self.in_right_suffix) for p in self.left_paths]
        self.gt_paths = [p.replace(self.in_left_suffix, self.gt_suffix) for p in self.left_paths]
        self.data_size = len(self.left_paths)

    @staticmethod
    def _standardise(gray_u8):
        img = gray_u8.astype(np.float32) / 255.                       # :85 (training divides by 255, matching does not;
        return (img - np.mean(img, axis=(0, 1))) / np.std(img, axis=(0, 1))   # standardisation removes the factor)

    def prefetch(self):
        """All pairs are kept in memory (:73-97): stereo training sets are a few dozen images."""
        self.left_images = [self._standardise(read_gray(p)) for p in self.left_paths]
        self.right_images = [self._standardise(read_gray(p)) for p in self.right_paths]
        self.gt_images = [np.asarray(readPfm(p), dtype=np.float32) for p in self.gt_paths]

    def shuffle_data(self):
        order = self.rng.permutation(self.data_size)
        for name in ("left_paths", "right_paths", "gt_paths", "left_images", "right_images", "gt_images"):
            setattr(self, name, [getattr(self, name)[i] for i in order])

    def reset_pointer(self):
        self.pointer = 0
        if self.shuffle:
            self.shuffle_data()

    def _padded(self, image):
        ph, pw = self.patch_size
        out = np.zeros((image.shape[0] + ph - 1, image.shape[1] + pw - 1), dtype=np.float32)
        out[(ph - 1) // 2:(ph - 1) // 2 + image.shape[0], (pw - 1) // 2:(pw - 1) // 2 + image.shape[1]] = image
        return out

    def _cut(self, padded, rows, cols):
        """[B, ph, pw, 1] patches whose top-left corners in the padded image are (rows, cols) = centred on the pixel."""
        ph, pw = self.patch_size
        rr = rows[:, None, None] + np.arange(ph)[None, :, None]
        cc = cols[:, None, None] + np.arange(pw)[None, None, :]
        return padded[rr, cc][..., None].astype(np.float32)

    def _displaced(self, right_col, width, draw):
        """int(right_col + deviation), redrawn per sample until it lies inside the image (:199-212)."""
        col = np.full(right_col.shape, -1, dtype=np.int64)
        todo = np.ones(right_col.shape, dtype=bool)
        while todo.any():
            dev = draw(int(todo.sum()))
            col[todo] = np.trunc(right_col[todo] + dev).astype(np.int64)     # Python int(): towards zero
            todo = (col < 0) | (col >= width)
        return col

    def next_batch(self, batch_size):
        left_image = self.left_images[self.pointer]
        right_image = self.right_images[self.pointer]
        gt_image = self.gt_images[self.pointer]
        assert left_image.shape == right_image.shape
        assert left_image.shape[0:2] == gt_image.shape
        height, width = left_image.shape[0:2]
        rng = self.rng

        # distinct rows and distinct columns first (:161-162), then redraw the samples that land on an unknown
        # (inf) or occluded (match left of the image) pixel anywhere in the image (:165-170)
        rows = rng.permutation(height)[:batch_size].astype(np.int64)
        cols = rng.permutation(width)[:batch_size].astype(np.int64)
        assert len(rows) == batch_size and len(cols) == batch_size, "batch_size exceeds the image height or width"

        def invalid(r, c):
            g = gt_image[r, c]
            bad = np.isinf(g)
            bad |= np.where(bad, 0, np.trunc(np.where(bad, 0, g))) > c
            return bad

        bad = invalid(rows, cols)
        while bad.any():
            n = int(bad.sum())
            rows[bad] = rng.integers(0, height, size=n)
            cols[bad] = rng.integers(0, width, size=n)
            bad = invalid(rows, cols)

        pl, pr = self._padded(left_image), self._padded(right_image)
        patches_left = self._cut(pl, rows, cols)
        right_col = cols - np.trunc(gt_image[rows, cols]).astype(np.int64)
        pos_col = self._displaced(right_col, width,
                                  lambda n: rng.uniform(-1 * self.dataset_pos, self.dataset_pos, size=n))

        def neg_dev(n):
            dev = rng.uniform(self.dataset_neg_low, self.dataset_neg_high, size=n)
            return np.where(rng.integers(-1, 1, size=n) == -1, -dev, dev)

        neg_col = self._displaced(right_col, width, neg_dev)
        patches_right_pos = self._cut(pr, rows, pos_col)
        patches_right_neg = self._cut(pr, rows, neg_col)
        self.pointer += 1
        return patches_left, patches_right_pos, patches_right_neg

    def next_pair(self):
        i = self.pointer
        assert self.left_images[i].shape == self.right_images[i].shape
        assert self.left_images[i].shape[0:2] == self.gt_images[i].shape
        self.pointer += 1
        return self.left_images[i], self.right_images[i], self.gt_images[i]
