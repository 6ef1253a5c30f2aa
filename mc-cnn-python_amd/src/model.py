"""Feature head of the "fast" MC-CNN - drop-in for /root/reference/src/model.py (NET, :9-65; conv, :90-125).

Same constructor arguments and attributes (`.X`, `.conv1 ... .convN`, `.features`); the TensorFlow graph is
replaced by eager PyTorch-ROCm convolutions (MIOpen), which is where north_star keeps the small conv stack.
Topology (model.py:51-64): num_conv_layers x (3x3 VALID conv, 64 maps), ReLU after all but the last, then
channel-wise L2 normalisation x * rsqrt(max(sum x^2, 1e-12)).  Fully convolutional: 11x11 patches give
[B,1,1,64]; an image zero-padded once by (patch-1)/2 (process_functional.py:20-25) gives [1,H,W,64].

Differences a maintainer should know:
  * `NET(x, ...)` evaluates immediately when `x` is a tensor / array (there is no session); `NET(None, ...)` builds
    the weights only, and `net(x)` / `net.features_hwc(image)` evaluate later.
  * weights come from `restore(checkpoint)` - a TF bundle prefix (the reference's --resume) or an .npz.
  * whole-image features use the HIP epilogue mccnn_l2norm_chw_to_hwc (NCHW conv output -> NHWC unit vectors).
"""
import math
import os

# MIOpen times every applicable solver the first time it meets a convolution shape; its reference "naive" direct
# FORWARD convolutions take ~150 ms per call at image size and never win - leaving them out of the search saves ~6 s
# per process.  Only the forward solver is touched (the backward / weight-gradient fallbacks of training stay
# available), and the user's own setting, if any, is kept.
os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "0")

import numpy as np
import torch
import torch.nn.functional as F

import tf_checkpoint


class NET(object):

    def __init__(self, x=None, weights_path='DEFAULT',
                 input_patch_size=11, num_conv_layers=5, num_conv_feature_maps=64,
                 conv_kernel_size=3, batch_size=128, device=None, seed=0):
        self.X = x
        self.batch_size = batch_size
        self.input_patch_size = input_patch_size
        self.num_conv_layers = num_conv_layers
        self.num_conv_feature_maps = num_conv_feature_maps
        self.conv_kernel_size = conv_kernel_size
        self.WEIGHTS_PATH = 'pretrain.npy' if weights_path == 'DEFAULT' else weights_path
        if device is None:
            if torch.is_tensor(x):
                device = x.device
            else:
                device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)
        self._init_variables(seed)
        if x is not None:
            self.create()

    # -- variables: conv{k}/weights [k,k,Cin,Cout] (HWIO) and conv{k}/biases [Cout], model.py:98-101 -----------------
    def _init_variables(self, seed):
        g = torch.Generator().manual_seed(seed)
        k, nf = self.conv_kernel_size, self.num_conv_feature_maps
        self.weights = []  # torch layout [Cout, Cin, k, k]
        self.biases = []
        cin = 1
        for _ in range(self.num_conv_layers):
            # tf.get_variable's default initializer is glorot_uniform
            limit = math.sqrt(6.0 / (k * k * cin + k * k * nf))
            w = (torch.rand((nf, cin, k, k), generator=g) * 2 - 1) * limit
            blim = math.sqrt(6.0 / (nf + nf))
            b = (torch.rand((nf,), generator=g) * 2 - 1) * blim
            self.weights.append(w.to(self.device))
            self.biases.append(b.to(self.device))
            cin = nf

    def set_layers(self, layers):
        """layers: list of (w_hwio float32 [k,k,Cin,Cout], bias [Cout]) as stored by TensorFlow."""
        assert len(layers) == self.num_conv_layers, "checkpoint has %d conv layers, NET was built with %d" % (
            len(layers), self.num_conv_layers)
        self.weights = [torch.from_numpy(np.ascontiguousarray(np.transpose(w, (3, 2, 0, 1)))).to(self.device)
                        for w, _ in layers]
        self.biases = [torch.from_numpy(np.ascontiguousarray(b)).to(self.device) for _, b in layers]
        return self

    def get_layers(self):
        return [(np.ascontiguousarray(w.detach().permute(2, 3, 1, 0).cpu().numpy()), b.detach().cpu().numpy())
                for w, b in zip(self.weights, self.biases)]

    def restore(self, checkpoint):
        """The reference's `saver.restore(sess, checkpoint)` (process_functional.py:43)."""
        return self.set_layers(tf_checkpoint.load_fast_net_weights(checkpoint))

    # -- the two unused .npy helpers of the reference (model.py:67-85), kept for API completeness ---------------------
    def load_initial_weights(self, session=None):
        weights_dict = np.load(self.WEIGHTS_PATH, encoding='bytes', allow_pickle=True).item()
        layers = self.get_layers()
        for name, value in weights_dict.items():
            name = name.decode() if isinstance(name, bytes) else name
            scope, var = name.split(":")[0].split("/")
            idx = int(scope.replace("conv", "")) - 1
            w, b = layers[idx]
            layers[idx] = (np.asarray(value, np.float32), b) if var == "weights" else (w, np.asarray(value, np.float32))
        self.set_layers(layers)

    def save_weights(self, session=None, file_name='pretrain.npy'):
        weights_dict = {}
        for k, (w, b) in enumerate(self.get_layers(), start=1):
            weights_dict["conv%d/weights:0" % k] = w
            weights_dict["conv%d/biases:0" % k] = b
        np.save(file_name, weights_dict)

    # -- forward ------------------------------------------------------------------------------------------------------
    def _convs_nchw(self, x):
        """x: [B,1,h,w] -> list of conv outputs, NCHW (ReLU on all but the last, model.py:51-61)."""
        outs = []
        nl = self.num_conv_layers
        for k in range(nl):
            x = F.conv2d(x, self.weights[k], self.biases[k])
            if k < nl - 1:
                x = F.relu(x)
            outs.append(x)
        return outs

    def _convs_device(self, images, pad):
        """The whole-image stack on the GPU.  images: [B,H,W] standardised views.  Layer 1 (1 -> 64 maps: 9
        multiply-adds per output) runs fused with the zero padding, its bias and ReLU in one HIP launch
        (mccnn_conv1_pad_bias_relu); layers 2..n are MIOpen convolutions without bias, each followed by ONE in-place
        HIP pass for bias + ReLU (mccnn_bias_act); the last layer's bias is left to the normalisation epilogue.
        Returns [B,64,H+2pad-2n,W+2pad-2n] (contiguous)."""
        import stereo_device
        nl = self.num_conv_layers
        if self.conv_kernel_size == 3 and nl >= 2:
            x = stereo_device.conv1_pad_bias_relu(images.contiguous(), self.weights[0], self.biases[0], pad)
            first = 1
        else:
            x = F.pad(images[:, None], (pad, pad, pad, pad))  # zero-pad ONCE; every conv is VALID
            first = 0
        for k in range(first, nl):
            x = F.conv2d(x, self.weights[k], None)
            if k < nl - 1:
                x = stereo_device.bias_act_(x.contiguous(), self.biases[k], True)
        return x.contiguous()

    @staticmethod
    def _l2_normalize_last(x):
        # tf.nn.l2_normalize(x, dim=-1): x * rsqrt(max(sum(x^2), 1e-12))  (model.py:64)
        s = torch.clamp((x * x).sum(dim=-1, keepdim=True), min=1e-12)
        return x * torch.rsqrt(s)

    def create(self):
        """Evaluates the graph on self.X (NHWC [B,h,w,1]) and exposes conv1..convN and features (NHWC)."""
        self(self.X)

    def __call__(self, x):
        x = torch.as_tensor(x, dtype=torch.float32).to(self.device)
        assert x.dim() == 4 and x.shape[-1] == 1, "NET expects NHWC input with one channel (model.py:36-40)"
        self.X = x
        outs = self._convs_nchw(x.permute(0, 3, 1, 2).contiguous())
        for k, o in enumerate(outs, start=1):
            setattr(self, "conv%d" % k, o.permute(0, 2, 3, 1))
        self.features = self._l2_normalize_last(outs[-1].permute(0, 2, 3, 1))
        return self.features

    def features_hwc(self, image_hw):
        """Whole-image path of compute_features (process_functional.py:20-34, 62-67) on the GPU.
        image_hw: standardised float32 device tensor [H,W] -> [H,W,64] unit feature vectors."""
        import stereo_device
        pad = (self.input_patch_size - 1) // 2
        assert pad == self.num_conv_layers * (self.conv_kernel_size - 1) // 2, \
            "patch size must equal the receptive field so that features keep the image size"
        out = self._convs_device(image_hw[None], pad)[0]      # [64,H,W] without the last bias
        return stereo_device.l2norm_chw_to_hwc(out, self.biases[-1])

    def supports_split_features(self):
        """The hand-written matrix-core feature kernels are built for 3x3 convolutions, 64 maps, >= 2 layers."""
        return self.conv_kernel_size == 3 and self.num_conv_feature_maps == 64 and self.num_conv_layers >= 2

    def _split_weights(self):
        """Packed f16 hi/lo weights of layers 2..n for the split-operand kernels, rebuilt when a weight tensor changes."""
        import stereo_device
        key = tuple((w.data_ptr(), w._version) for w in self.weights[1:])
        cache = getattr(self, "_split_cache", None)
        if cache is None or cache[0] != key:
            cache = (key, [stereo_device.conv3x3_split_pack(w) for w in self.weights[1:]])
            self._split_cache = cache
        return cache[1]

    def _split_flag(self):
        """Device int the split kernels set when an activation leaves the f16 range of the stored records
        (|x| >= 65504 / act_scale = 255.9; the trained checkpoint peaks near 6 on standardised images).  Round 3
        guessed the range from a probe image; the kernels now report what actually happened."""
        f = getattr(self, "_split_sat", None)
        if f is None or f.device != self.weights[0].device:
            f = torch.zeros((1,), dtype=torch.int32, device=self.weights[0].device)
            self._split_sat = f
        return f

    def split_saturated(self, reset=True):
        """True when a split-operand feature launch since the last reset clamped an activation (one .item() sync)."""
        f = getattr(self, "_split_sat", None)
        if f is None:
            return False
        hit = bool(int(f.item()))
        if hit and reset:
            f.zero_()
        return hit

    def features_pair_hwc_split(self, left_hw, right_hw):
        """features_pair_hwc on the matrix cores (csrc/conv_mfma.hip): every float32 operand as two f16 numbers, three
        MFMA products per multiply, float32 accumulation (main and cross terms in accumulators of their own) - as
        close to a float64 evaluation as the library path, not bit-identical to it.  The default feature path of
        StereoMatcher / match.py / process_functional.compute_features.  Needs the 64-map 3x3 topology with at least
        two layers; activations must stay below 65504 / 256 in magnitude - split_saturated() tells if one did not."""
        import warnings
        import stereo_device
        pad = (self.input_patch_size - 1) // 2
        assert pad == self.num_conv_layers * (self.conv_kernel_size - 1) // 2
        if not self.supports_split_features():
            raise ValueError("the split-operand feature kernels are built for >= 2 layers of 64 maps, 3x3")
        packed = self._split_weights()
        flag = self._split_flag()
        H, W = left_hw.shape
        # the kernels address their records with 32-bit byte offsets: 256 B per padded pixel, both views in one batch
        limit = 0x7ffffff0 // 256
        padded = (H + 2 * pad - 2) * (W + 2 * pad - 2)
        if padded > limit:
            rows = max(64, (limit // (W + 2 * pad)) // 2)
            warnings.warn("split-operand features: a %dx%d view exceeds the kernels' 32-bit record offsets; using the "
                          "float32 library convolutions in bands of %d rows instead" % (W, H, rows))
            return self.features_pair_hwc(left_hw, right_hw, tile_rows=rows)
        batches = [torch.stack((left_hw, right_hw)).contiguous()] if 2 * padded <= limit else \
            [left_hw[None].contiguous(), right_hw[None].contiguous()]          # too big as a pair: view by view
        outs = []
        nl = self.num_conv_layers
        for views in batches:
            x = stereo_device.conv1_split(views, self.weights[0].detach().contiguous(), self.biases[0].detach(), pad,
                                          sat_flag=flag)
            for k in range(1, nl):
                pk, ws = packed[k - 1]
                x = stereo_device.conv3x3_split(x, pk, ws, self.biases[k].detach(), last=(k == nl - 1), sat_flag=flag)
            outs += [x[i] for i in range(x.shape[0])]
        return outs[0], outs[1]

    def features_pair_hwc(self, left_hw, right_hw, tile_rows=None):
        """Both views through the shared-weight stack as one batch of two (the Siamese towers are the same weights,
        model.py:98 AUTO_REUSE / train.py:76-78): half the launches, twice the work per launch.

        tile_rows: evaluate the stack on horizontal bands of that many output rows (each band sees its (patch-1)/2
        halo rows of the once-padded image, so every output pixel gets exactly the receptive field it has in the
        untiled run) - the working version of the 4-quadrant scheme the reference left commented out
        (process_functional.py:46-60) for images whose activations do not fit; peak activation memory scales with the
        band instead of the image.  None (default): the whole image at once."""
        import stereo_device
        pad = (self.input_patch_size - 1) // 2
        assert pad == self.num_conv_layers * (self.conv_kernel_size - 1) // 2
        pair = torch.stack((left_hw, right_hw))
        if tile_rows is None or tile_rows >= pair.shape[1]:
            out = self._convs_device(pair, pad)                                       # [2,64,H,W], last bias pending
            return (stereo_device.l2norm_chw_to_hwc(out[0], self.biases[-1]),
                    stereo_device.l2norm_chw_to_hwc(out[1], self.biases[-1]))
        H, W = pair.shape[1], pair.shape[2]
        x = F.pad(pair[:, None], (pad, pad, pad, pad))                                # zero-pad ONCE (pf:20-25)
        feats = [torch.empty((H, W, self.num_conv_feature_maps), dtype=torch.float32, device=pair.device)
                 for _ in range(2)]
        for y0 in range(0, H, int(tile_rows)):
            y1 = min(y0 + int(tile_rows), H)
            t = x[:, :, y0:y1 + 2 * pad, :]
            for k in range(self.num_conv_layers):
                t = F.conv2d(t, self.weights[k], None)
                if k < self.num_conv_layers - 1:
                    t = stereo_device.bias_act_(t.contiguous(), self.biases[k], True)
            t = t.contiguous()
            for v in range(2):   # the band's unit vectors go straight into the [H,W,64] result
                stereo_device.l2norm_chw_to_hwc(t[v], self.biases[-1], out=feats[v][y0:y1])
        return feats[0], feats[1]


if __name__ == "__main__":
    net = NET(torch.zeros([128, 11, 11, 1]))
    print(net.features.shape)
