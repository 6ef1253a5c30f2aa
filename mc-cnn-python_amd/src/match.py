"""Drop-in for /root/reference/src/match.py: same command line, same output files, MI355X hot path.

    python match.py -g 0 --list_file LIST --resume CKPT_PREFIX --data_dir DATA --save_dir OUT -t TAG -s 0 -e 14

For every left image `.../im0.png` in the list (index window [start, end] inclusive, match.py:83-91) it reads
im1.png + calib.txt beside it, standardises both views, runs the timed region (features -> cost volume -> CBCA x2
-> SGM -> CBCA x16 -> WTA -> interpolation -> sub-pixel -> median -> bilateral) on one GPU and writes
    <save_dir>/submit_<tag>/<rel>/disp0MCCNN.pfm, timeMCCNN.txt   and   <save_dir>/submit_<tag>_imgs/<rel>/disp0MCCNN.pgm
exactly where the reference does (match.py:99-110, 182-184).

Additions (all optional): --exact selects the bit-exact variants of the two stages that have a faster, tolerance-
bounded form (cost volume on NumPy's summation order instead of MFMA; CBCA in the reference's list order).
Multi-GPU: launch one process per GPU with different -g / -s / -e, as the reference intends (match.py:17, 26-28),
or use `torchrun --nproc-per-node N match.py ...`: rank r then takes the pairs i = r (mod N) of the window.
"""
import argparse
import os
import time
from datetime import datetime

import numpy as np

parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                                 description="stereo matching based on trained model and post-processing")
parser.add_argument("-g", "--gpu", type=str, default="0", help="gpu id to use, \
                    multiple ids should be separated by commons(e.g. 0,1,2,3)")
parser.add_argument("-ps", "--patch_size", type=int, default=11, help="length for height/width of square patch")
parser.add_argument("--list_file", type=str, required=True, help="path to file containing left image list")
parser.add_argument("--resume", type=str, default=None, help="path to checkpoint to resume from. \
                    (TensorFlow bundle prefix as in the reference, or an .npz of conv<k>/weights|biases)")
parser.add_argument("--data_dir", type=str, required=True, help="path to root dir to data.")
parser.add_argument("--save_dir", type=str, required=True, help="path to root dir to save results")
parser.add_argument("-t", "--tag", type=str, required=True, help="tag used to indicate one run")
parser.add_argument("-s", "--start", type=int, required=True, help="index of first image to do matching,\
                                                                    this is used for parallel matching of different images")
parser.add_argument("-e", "--end", type=int, required=True, help="index of last image to do matching")

# hyperparemeters, use suggested value from origin paper as default (match.py:31-43; the three CBCA counts are declared
# float there but only work as the ints they default to - they are coerced with int() here)
parser.add_argument("--cbca_intensity", type=float, default=0.02, help="intensity threshold for cross-based cost aggregation")
parser.add_argument("--cbca_distance", type=float, default=14, help="distance threshold for cross-based cost aggregation")
parser.add_argument("--cbca_num_iterations1", type=float, default=2, help="cross-based cost aggregation rounds before SGM")
parser.add_argument("--cbca_num_iterations2", type=float, default=16, help="cross-based cost aggregation rounds after SGM")
parser.add_argument("--sgm_P1", type=float, default=2.3, help="hyperparemeter used in semi-global matching")
parser.add_argument("--sgm_P2", type=float, default=55.9, help="hyperparemeter used in semi-global matching")
parser.add_argument("--sgm_Q1", type=float, default=4, help="hyperparemeter used in semi-global matching")
parser.add_argument("--sgm_Q2", type=float, default=8, help="hyperparemeter used in semi-global matching")
parser.add_argument("--sgm_D", type=float, default=0.08, help="hyperparemeter used in semi-global matching")
parser.add_argument("--sgm_V", type=float, default=1.5, help="hyperparemeter used in semi-global matching")
parser.add_argument("--blur_sigma", type=float, default=6, help="hyperparemeter used in bilateral filter")
parser.add_argument("--blur_threshold", type=float, default=2, help="hyperparemeter used in bilateral filter")
parser.add_argument("--exact", action="store_true", help="bit-exact stage variants (NumPy-order cost volume, "
                    "reference-order CBCA) instead of the fast tolerance-bounded ones")

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

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        # the reference pins the process to the requested card through the environment (match.py:59)
        os.environ.setdefault("HIP_VISIBLE_DEVICES", args.gpu)
        os.environ["CUDA_VISIBLE_DEVICES"] = args.gpu

    import torch
    import _hipabi as hip
    import stereo_device as sd
    import util
    from model import NET

    hip.require_device()
    torch.cuda.set_device(local_rank if world > 1 else 0)

    patch_height = args.patch_size
    save_dir = args.save_dir
    data_dir = args.data_dir
    save_res_dir = os.path.join(save_dir, "submit_{}".format(args.tag))
    save_img_dir = os.path.join(save_dir, "submit_{}_imgs".format(args.tag))
    util.recurMk(os.path.abspath(save_res_dir))
    util.recurMk(os.path.abspath(save_img_dir))

    with open(args.list_file, "r") as i:
        img_paths = i.readlines()

    net = NET(None, input_patch_size=patch_height, num_conv_layers=(patch_height - 1) // 2, batch_size=1,
              device="cuda")
    net.restore(args.resume)  # loaded once and kept resident (the reference re-restores per pair)
    matcher = sd.StereoMatcher(
        net, hyper_parameters(args),
        cv_mode=hip.MCCNN_CV_EXACT if args.exact else hip.MCCNN_CV_MFMA,
        cbca_order=hip.MCCNN_CBCA_REFERENCE_ORDER if args.exact else hip.MCCNN_CBCA_SEPARABLE)

    from distributed import shard_indices
    for index in shard_indices(args.start, args.end, len(img_paths), rank, world):
        left_path = img_paths[index].strip()
        print("index: {}".format(index))
        right_path = left_path.replace(left_image_suffix, right_image_suffix)
        calib_path = left_path.replace(left_image_suffix, calib_suffix)

        res_dir = left_path.replace(data_dir, save_res_dir)
        img_dir = left_path.replace(data_dir, save_img_dir)
        res_dir = res_dir[:res_dir.rfind(left_image_suffix) - 1]
        img_dir = img_dir[:img_dir.rfind(left_image_suffix) - 1]
        util.recurMk(os.path.abspath(res_dir))
        util.recurMk(os.path.abspath(img_dir))

        out_path = os.path.join(res_dir, out_file)
        out_time_path = os.path.join(res_dir, out_time_file)
        out_img_path = os.path.join(img_dir, out_img_file)

        height, width, ndisp = util.parseCalib(calib_path)
        print("left_image: {}\nright_image: {}".format(left_path, right_path))
        print("height: {}, width: {}, ndisp: {}".format(height, width, ndisp))
        print("out_path: {}\nout_time_path: {}\nout_img_path: {}".format(out_path, out_time_path, out_img_path))

        # reading images (match.py:118-125)
        left_image = util.read_gray(left_path).astype(np.float32)
        right_image = util.read_gray(right_path).astype(np.float32)
        left_image = (left_image - np.mean(left_image, axis=(0, 1))) / np.std(left_image, axis=(0, 1))
        right_image = (right_image - np.mean(right_image, axis=(0, 1))) / np.std(right_image, axis=(0, 1))
        left_image = np.expand_dims(left_image, axis=2)
        right_image = np.expand_dims(right_image, axis=2)
        assert left_image.shape == (height, width, 1)
        assert right_image.shape == (height, width, 1)
        print("{}: images read".format(datetime.now()))

        # timed region (match.py:129-179): host arrays in, host array out, device-synchronised
        stTime = time.time()
        dev_l = torch.from_numpy(left_image).cuda()
        dev_r = torch.from_numpy(right_image).cuda()
        disparity = matcher.match(dev_l, dev_r, ndisp)
        left_disparity_map = disparity.cpu().numpy()
        endTime = time.time()
        print("{}: refined".format(datetime.now()))

        util.saveDisparity(left_disparity_map, out_img_path)
        util.writePfm(left_disparity_map, out_path)
        util.saveTimeFile(endTime - stTime, out_time_path)
        print("{}: saved".format(datetime.now()))


if __name__ == "__main__":
    main()
