"""Drop-in for /root/reference/src/match.py: same command line, same output files, MI355X hot path.

    python match.py -g 0 --list_file LIST --resume CKPT_PREFIX --data_dir DATA --save_dir OUT -t TAG -s 0 -e 14

For every left image `.../im0.png` in the list (index window [start, end] inclusive, match.py:83-91) it reads
im1.png + calib.txt beside it, standardises both views, runs the timed region (features -> cost volume -> CBCA x2
-> SGM -> CBCA x16 -> WTA -> interpolation -> sub-pixel -> median -> bilateral) on one GPU and writes
    <save_dir>/submit_<tag>/<rel>/disp0MCCNN.pfm, timeMCCNN.txt   and   <save_dir>/submit_<tag>_imgs/<rel>/disp0MCCNN.pgm
exactly where the reference does (match.py:99-110, 182-184).

By default every stage after the conv features is bit-identical to the reference's NumPy code on the same inputs.
Addition: --fast computes the cost volume on the matrix cores instead of in NumPy's summation order (<= 2e-6 per score;
near-ties in the WTA can then resolve differently) and leaves every other stage as it is - a few per cent faster;
--fast --separable_cbca adds the aggregation through float64 prefix sums on plane-major volumes (<= 1e-6 per
iteration), the fast variant of rounds 2-3, which the bit-exact aggregation has overtaken since.
Multi-GPU: launch one process per GPU with different -g / -s / -e, as the reference intends (match.py:17, 26-28),
or use `torchrun --nproc-per-node N match.py ...`: rank r then takes the pairs i = r (mod N) of the window.
"""
import argparse
import contextlib
import os
import time
from datetime import datetime

import numpy as np

import util

# Flag names, types and defaults are the reference's command line (match.py:15-43) - that is the drop-in contract;
# the help texts are this project's.
parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                                 description="MC-CNN stereo matching of a list of Middlebury-style pairs on one MI355X")
parser.add_argument("-g", "--gpu", type=str, default="0", action=util.ExplicitStore,
                    help="index of the GPU this process uses (when not given, a HIP_VISIBLE_DEVICES already in the "
                         "environment stands; ignored under torchrun: one rank per GPU)")
parser.add_argument("-ps", "--patch_size", type=int, default=11,
                    help="receptive field of the matching network (11 = five 3x3 layers)")
parser.add_argument("--list_file", type=str, required=True, help="text file with one left-image path (.../im0.png) per line")
parser.add_argument("--resume", type=str, default=None,
                    help="network weights: TensorFlow checkpoint prefix as written by the reference's train.py, or an "
                         ".npz of conv<k>/weights and conv<k>/biases")
parser.add_argument("--data_dir", type=str, required=True, help="root of the input tree (prefix of the listed paths)")
parser.add_argument("--save_dir", type=str, required=True, help="root under which submit_<tag>/ and submit_<tag>_imgs/ are written")
parser.add_argument("-t", "--tag", type=str, required=True, help="name of this run (part of the output directory names)")
parser.add_argument("-s", "--start", type=int, required=True,
                    help="first list index to match (inclusive) - split a list over several processes with -s/-e")
parser.add_argument("-e", "--end", type=int, required=True, help="last list index to match (inclusive)")

# hyper-parameters, defaults from the MC-CNN paper as in match.py:31-43 (the three CBCA values are declared float there
# but only work as the ints they default to - they are coerced with int() below)
parser.add_argument("--cbca_intensity", type=float, default=0.02, help="CBCA: largest |I(q) - I(p)| along a support arm")
parser.add_argument("--cbca_distance", type=float, default=14, help="CBCA: arm length limit (arms are shorter than this)")
parser.add_argument("--cbca_num_iterations1", type=float, default=2, help="CBCA iterations before SGM")
parser.add_argument("--cbca_num_iterations2", type=float, default=16, help="CBCA iterations after SGM")
parser.add_argument("--sgm_P1", type=float, default=2.3, help="SGM: penalty for a disparity change of one")
parser.add_argument("--sgm_P2", type=float, default=55.9, help="SGM: penalty for larger disparity changes")
parser.add_argument("--sgm_Q1", type=float, default=4, help="SGM: penalties divided by this where one view has an intensity edge")
parser.add_argument("--sgm_Q2", type=float, default=8, help="SGM: penalties divided by this where both views have one")
parser.add_argument("--sgm_D", type=float, default=0.08, help="SGM: intensity difference that counts as an edge")
parser.add_argument("--sgm_V", type=float, default=1.5, help="SGM: P1 is divided by this in the vertical directions")
parser.add_argument("--blur_sigma", type=float, default=6, help="bilateral filter: spatial sigma")
parser.add_argument("--blur_threshold", type=float, default=2, help="bilateral filter: intensity gate")
# additions of this implementation
parser.add_argument("--fast", action="store_true",
                    help="the cost volume on the matrix cores (split-f16 operands, <= 2e-6 per score) in front of the "
                         "bit-exact stages: a few per cent faster, final map within the stated tolerance "
                         "(src/tolerances.py).  Without it every stage after the conv features is bit-identical to the "
                         "reference's NumPy code")
parser.add_argument("--separable_cbca", action="store_true",
                    help="with --fast: also the separable float64-prefix aggregation on plane-major volumes (<= 1e-6 per "
                         "iteration; the fast variant of rounds 2-3, slower than the default since round 4)")
parser.add_argument("--exact", action="store_true", help="(default; kept for compatibility) the bit-exact variants")
parser.add_argument("--features", choices=("library", "split_f16"), default=None,
                    help="conv feature stack: 'split_f16' (default where the network is 3x3 / 64 maps: the hand-written "
                         "matrix-core kernels, float32 operands as two f16 parts, as close to a float64 evaluation as the "
                         "library) or 'library' (float32 convolutions of MIOpen, 1.5 ms slower per 750x500 pair).  A pair "
                         "whose activations leave the kernels' range is repeated with the library automatically")
parser.add_argument("--pairs_in_flight", type=int, default=1,
                    help="stereo pairs matched concurrently on this GPU, each on its own HIP stream with its own "
                         "workspace.  The kernels of a KITTI-sized or smaller pair do not fill 256 CUs (a 256x256x64 "
                         "pair reaches a third of the throughput of a 750x500x256 one); 2 overlaps the launch ramps of "
                         "one pair with the other.  Results are identical; timeMCCNN.txt then holds each pair's own "
                         "wall time, overlap included")
# opt-in departures from the reference's results: what the MC-CNN paper does and the reference names but leaves out
parser.add_argument("--paper_support_regions", action="store_true",
                    help="CBCA support regions intersected with the other view's at every disparity (paper sec. 4.1; "
                         "the reference skips this as impractical, process_functional.py:122-144)")
parser.add_argument("--paper_interpolation", action="store_true",
                    help="fill mismatches from 16 rays and occlusions from the left (paper sec. 4.4; "
                         "process_functional.py:318, :361 note the reference uses 4 directions and the right)")
parser.add_argument("--numpy1_promotion", action="store_true",
                    help="evaluate the sub-pixel formula as NumPy < 2 promotes its scalars (float64, rounded once), "
                         "i.e. as the reference's own Python 2.7 environment does; differs by <= 2.5e-5 px")

# different file names
left_image_suffix = "im0.png"
left_gt_suffix = "disp0GT.pfm"
right_image_suffix = "im1.png"
right_gt_suffix = "disp1GT.pfm"
calib_suffix = "calib.txt"

out_file = "disp0MCCNN.pfm"
out_img_file = "disp0MCCNN.pgm"
out_time_file = "timeMCCNN.txt"


def hyper_parameters(args):
    return dict(cbca_intensity=args.cbca_intensity, cbca_distance=int(args.cbca_distance),
                cbca_num_iterations1=int(args.cbca_num_iterations1),
                cbca_num_iterations2=int(args.cbca_num_iterations2),
                sgm_P1=args.sgm_P1, sgm_P2=args.sgm_P2, sgm_Q1=args.sgm_Q1, sgm_Q2=args.sgm_Q2, sgm_D=args.sgm_D,
                sgm_V=args.sgm_V, blur_sigma=args.blur_sigma, blur_threshold=args.blur_threshold)


def main(argv=None):
    args = parser.parse_args(argv)
    if args.fast and args.exact:
        parser.error("--fast and --exact exclude each other")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    util.pin_gpu(args, world)     # before torch initialises HIP

    import torch
    import _hipabi as hip
    import stereo_device as sd
    from distributed import shard_indices
    from model import NET

    hip.require_device()
    # one rank per GPU under torchrun; MCCNN_SHARED_GPU=1 (tests) lets several ranks share the visible GPUs
    shared = os.environ.get("MCCNN_SHARED_GPU") == "1"
    torch.cuda.set_device((local_rank % torch.cuda.device_count() if shared else local_rank) if world > 1 else 0)

    result_root = os.path.join(args.save_dir, "submit_{}".format(args.tag))        # match.py:69-72
    image_root = os.path.join(args.save_dir, "submit_{}_imgs".format(args.tag))
    util.recurMk(os.path.abspath(result_root))
    util.recurMk(os.path.abspath(image_root))

    with open(args.list_file, "r") as f:
        left_paths = [line.strip() for line in f.readlines()]

    net = NET(None, input_patch_size=args.patch_size, num_conv_layers=(args.patch_size - 1) // 2, batch_size=1,
              device="cuda")
    net.restore(args.resume)  # loaded once and kept resident (the reference re-restores per pair)
    in_flight = max(1, int(args.pairs_in_flight))

    def make_matcher(features):
        return sd.StereoMatcher(
            net, hyper_parameters(args),
            cv_mode=hip.MCCNN_CV_MFMA if args.fast else hip.MCCNN_CV_EXACT,
            cbca_order=hip.MCCNN_CBCA_SEPARABLE if (args.fast and args.separable_cbca) else hip.MCCNN_CBCA_REFERENCE_ORDER,
            features=features,
            on_saturation="ignore",      # several pairs may be in flight: finish() polls the flag and repeats them
            extras=dict(both_view_support=args.paper_support_regions,
                        interpolation_directions=16 if args.paper_interpolation else 4,
                        occlusion_from_left=args.paper_interpolation, numpy1_promotion=args.numpy1_promotion))

    matchers = [make_matcher("miopen" if args.features == "library" else "auto") for _ in range(in_flight)]
    streams = [torch.cuda.Stream() for _ in range(in_flight)] if in_flight > 1 else [None]
    pending = []          # pairs launched and not yet written: (device map, done event, start time, output paths)
    launched = 0

    redo = {"left": 0, "matcher": None}

    def finish(entry):
        disparity, done, stTime, out_path, out_time_path, out_img_path, images = entry
        done.synchronize()
        # the hand-written feature kernels report an activation beyond the range of their stored records (never seen
        # on standardised images with the trained weights): that pair - and, with several in flight, the ones that
        # shared the flag with it - is matched again with the float32 library convolutions
        if matchers[0].features == "split_f16" and matchers[0].features_saturated():
            redo["left"] = in_flight
        if redo["left"] > 0:
            redo["left"] -= 1
            if redo["matcher"] is None:
                redo["matcher"] = make_matcher("miopen")
            print("[{}] activations left the matrix-core feature kernels' range: {} repeated with the float32 library "
                  "convolutions".format(rank, out_path))
            disparity = redo["matcher"].match(images[0], images[1], images[2])
            torch.cuda.synchronize()
        left_disparity_map = disparity.cpu().numpy()
        endTime = time.time()
        util.saveDisparity(left_disparity_map, out_img_path)
        util.writePfm(left_disparity_map, out_path)
        util.saveTimeFile(endTime - stTime, out_time_path)
        print("[{}] {}: {:.3f} s -> {}".format(rank, datetime.now(), endTime - stTime, out_path))

    for index in shard_indices(args.start, args.end, len(left_paths), rank, world):
        left_path = left_paths[index]
        right_path = left_path.replace(left_image_suffix, right_image_suffix)
        calib_path = left_path.replace(left_image_suffix, calib_suffix)
        # outputs mirror the input tree under the two roots (match.py:99-110): <root>/<dir of im0.png relative to data_dir>
        pair_dir = os.path.dirname(left_path)
        res_dir = pair_dir.replace(args.data_dir, result_root)
        img_dir = pair_dir.replace(args.data_dir, image_root)
        util.recurMk(os.path.abspath(res_dir))
        util.recurMk(os.path.abspath(img_dir))
        out_path = os.path.join(res_dir, out_file)
        out_time_path = os.path.join(res_dir, out_time_file)
        out_img_path = os.path.join(img_dir, out_img_file)

        height, width, ndisp = util.parseCalib(calib_path)
        print("[{}] pair {}: {} | {}  ({}x{}, ndisp {})".format(rank, index, left_path, right_path, width, height, ndisp))

        # decode + standardise (match.py:118-125): population std, no /255
        views = []
        for path in (left_path, right_path):
            g = util.read_gray(path).astype(np.float32)
            g = (g - np.mean(g, axis=(0, 1))) / np.std(g, axis=(0, 1))
            views.append(np.expand_dims(g, axis=2))
        left_image, right_image = views
        assert left_image.shape == (height, width, 1)
        assert right_image.shape == (height, width, 1)

        # timed region (match.py:129-179): host arrays in, host array out, device-synchronised.  With several pairs
        # in flight the pair is launched on its slot's stream and collected when the slot comes round again.
        slot = launched % in_flight
        launched += 1
        if len(pending) == in_flight:
            finish(pending.pop(0))        # the oldest pair is the one that used this slot
        stTime = time.time()
        ctx = torch.cuda.stream(streams[slot]) if in_flight > 1 else contextlib.nullcontext()
        with ctx:
            dev_l = torch.from_numpy(left_image).cuda()
            dev_r = torch.from_numpy(right_image).cuda()
            disparity = matchers[slot].match(dev_l, dev_r, ndisp)
            done = torch.cuda.Event()
            done.record()
        pending.append((disparity, done, stTime, out_path, out_time_path, out_img_path, (dev_l, dev_r, ndisp)))
        if in_flight == 1:
            finish(pending.pop(0))
    while pending:
        finish(pending.pop(0))


if __name__ == "__main__":
    main()
