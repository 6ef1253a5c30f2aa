"""Training of the matching network - drop-in for /root/reference/src/train.py (same command line), on PyTorch-ROCm.

    python train.py --list_dir LISTS --tensorboard_dir LOG --checkpoint_dir CKPT [-bs 128 -mr 0.2 -lr 0.002 -bt 0.9 ...]

What the reference's graph does (train.py:71-106) and what happens here:
  * three weight-shared towers (left, right+, right-) of model.NET on 11x11 patches (:76-78)   -> ONE NET applied to
    the three batches stacked into one [3B,11,11,1] batch (the towers ARE the same variables, AUTO_REUSE);
  * similarity = dot product of the L2-normalised 64-vectors (:85-87); loss = mean(max(0, margin - s+ + s-)) (:90-93);
  * tf.train.MomentumOptimizer(lr, beta) (:105-106): accum = beta*accum + grad; var -= lr*accum
    -> torch.optim.SGD(momentum=beta, dampening=0, nesterov=False), the same recurrence;
  * one mini-batch per training image per epoch, drawn by datagenerator.ImageDataGenerator (:159-164);
  * validation loss over val.txt every --val_freq epochs (:182-197); a checkpoint every --save_freq epochs (:176-180).
Checkpoints are `.npz` files (conv<k>/weights HWIO, conv<k>/biases, plus the momentum slots) named
`model_epoch<N>.ckpt.npz`; match.py's --resume and NET.restore read them (and the reference's TensorFlow bundles).
Differences from the reference a maintainer should know: checkpoints are written as `.npz` only (match.py / train.py of
THIS package read them and the reference's TensorFlow bundles, incl. the Momentum slots on --resume; the reference
cannot read the .npz - one-way compatibility); the logged hinge_loss is the loss of the batch BEFORE its update (the
reference re-evaluates the graph after the step, train.py:169-173); the patch sampler's RNG state is not checkpointed.
TensorBoard is not in this image: the two scalars the reference logs (hinge_loss, val_hinge_loss) go to
`<tensorboard_dir>/scalars.jsonl`, one JSON object per point with the reference's step numbering.
Multi-GPU (not in the reference): under torchrun every rank draws its own batches and the gradients are averaged with
all_reduce (RCCL) before the step - synchronous data parallelism; rank 0 writes logs and checkpoints.
"""
import argparse
import json
import os
from datetime import datetime

import numpy as np

import util

parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                                 description="training of the MC-CNN matching network (fast architecture)")
parser.add_argument("-g", "--gpu", type=str, default="0", action=util.ExplicitStore,
                    help="index of the GPU to train on (when not given, a HIP_VISIBLE_DEVICES already in the environment "
                         "stands; ignored under torchrun)")
parser.add_argument("-ps", "--patch_size", type=int, default=11, help="side of the square training patches")
parser.add_argument("-bs", "--batch_size", type=int, default=128, help="patch triplets per mini-batch")
parser.add_argument("-mr", "--margin", type=float, default=0.2, help="margin of the hinge loss")
parser.add_argument("-lr", "--learning_rate", type=float, default=0.002, help="step size")
parser.add_argument("-bt", "--beta", type=float, default=0.9, help="momentum (declared int in the reference, which "
                    "only works at its default 0.9)")
parser.add_argument("--list_dir", type=str, required=True, help="directory holding train.txt and val.txt (left-image lists)")
parser.add_argument("--tensorboard_dir", type=str, required=True, help="directory for the scalar log")
parser.add_argument("--checkpoint_dir", type=str, required=True, help="directory for the checkpoints")
parser.add_argument("--resume", type=str, default=None, help="checkpoint to start from (.npz of this script, or a "
                    "TensorFlow bundle prefix of the reference); default: fresh glorot-uniform weights")
parser.add_argument("--start_epoch", type=int, default=0, help="first epoch (inclusive)")
parser.add_argument("--end_epoch", type=int, default=14, help="last epoch (exclusive)")
parser.add_argument("--print_freq", type=int, default=10, help="log the training loss every this many batches")
parser.add_argument("--save_freq", type=int, default=1, help="write a checkpoint every this many epochs")
parser.add_argument("--val_freq", type=int, default=1, help="validate every this many epochs")
parser.add_argument("--seed", type=int, default=0, help="seed of the weight initialisation and the patch sampler")


def hinge_loss(features, batch_size, margin):
    """features: [3B,1,1,64] unit vectors of the stacked (left, right+, right-) batch -> mean hinge loss (train.py:80-93)."""
    f = features.reshape(3, batch_size, -1)
    cosine_pos = (f[0] * f[1]).sum(dim=-1)
    cosine_neg = (f[0] * f[2]).sum(dim=-1)
    return (margin - cosine_pos + cosine_neg).clamp(min=0.0).mean()


class Trainer(object):
    """The trainable twin of model.NET: the same conv{k}/weights + biases as torch Parameters, forward through
    NET's own evaluation code, momentum SGD, (optional) gradient averaging across ranks."""

    def __init__(self, net, learning_rate, beta, margin):
        import torch
        self.net = net
        self.params = []
        for k in range(net.num_conv_layers):
            net.weights[k] = torch.nn.Parameter(net.weights[k].clone())
            net.biases[k] = torch.nn.Parameter(net.biases[k].clone())
            self.params += [net.weights[k], net.biases[k]]
        self.opt = torch.optim.SGD(self.params, lr=learning_rate, momentum=beta, dampening=0.0, nesterov=False)
        self.margin = margin

    def loss(self, batch_left, batch_right_pos, batch_right_neg):
        import torch
        x = torch.from_numpy(np.concatenate([batch_left, batch_right_pos, batch_right_neg], axis=0)).to(self.net.device)
        return hinge_loss(self.net(x), batch_left.shape[0], self.margin)

    def step(self, batch_left, batch_right_pos, batch_right_neg):
        import torch.distributed as dist
        self.opt.zero_grad(set_to_none=True)
        loss = self.loss(batch_left, batch_right_pos, batch_right_neg)
        loss.backward()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            for p in self.params:            # synchronous data parallelism: average the gradients, then the same step
                dist.all_reduce(p.grad)
                p.grad /= dist.get_world_size()
        self.opt.step()
        return float(loss.detach())

    def state(self):
        """{name: array}: variables in TensorFlow's naming/layout plus the optimizer's momentum slots."""
        out = {}
        for k, (w, b) in enumerate(self.net.get_layers(), start=1):
            out["conv%d/weights" % k] = w
            out["conv%d/biases" % k] = b
        for k in range(self.net.num_conv_layers):
            for name, p in (("weights", self.net.weights[k]), ("biases", self.net.biases[k])):
                buf = self.opt.state.get(p, {}).get("momentum_buffer")
                if buf is not None:
                    a = buf.detach().cpu().numpy()
                    out["conv%d/%s/Momentum" % (k + 1, name)] = np.transpose(a, (2, 3, 1, 0)) if a.ndim == 4 else a
        return out

    def load_state(self, path):
        """Weights from an .npz of this script or a TensorFlow bundle; momentum slots when the .npz has them."""
        import torch
        import tf_checkpoint
        layers = tf_checkpoint.load_fast_net_weights(path)
        with torch.no_grad():
            for k, (w, b) in enumerate(layers):
                self.net.weights[k].copy_(torch.from_numpy(np.ascontiguousarray(np.transpose(w, (3, 2, 0, 1)))))
                self.net.biases[k].copy_(torch.from_numpy(np.ascontiguousarray(b)))
        # the optimizer's momentum slots, as saver.restore brings them back in the reference (train.py:141-143):
        # from this script's .npz, or from the "<var>/Momentum" tensors of a TensorFlow bundle
        if os.path.isfile(path) and path.endswith(".npz"):
            z = np.load(path)
            slots = {k: z[k] for k in z.files if k.endswith("/Momentum")}
        else:
            slots = {k: v for k, v in tf_checkpoint.load_checkpoint(path, skip_slots=False).items()
                     if k.endswith("/Momentum")}
        for k in range(self.net.num_conv_layers):
            for name, p in (("weights", self.net.weights[k]), ("biases", self.net.biases[k])):
                a = slots.get("conv%d/%s/Momentum" % (k + 1, name))
                if a is not None:
                    a = np.transpose(a, (3, 2, 0, 1)) if a.ndim == 4 else a
                    self.opt.state[p]["momentum_buffer"] = torch.from_numpy(np.ascontiguousarray(a)).to(p.device)


def main(argv=None):
    args = parser.parse_args(argv)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    util.pin_gpu(args, world)     # an explicit -g pins the card (train.py:57), before torch initialises HIP

    import torch
    import distributed as mgpu
    from datagenerator import ImageDataGenerator
    from model import NET

    on_gpu = torch.cuda.is_available()
    if on_gpu:
        torch.cuda.set_device(local_rank if world > 1 else 0)
    device = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    mgpu.init("nccl" if on_gpu else "gloo", device if on_gpu else None)

    os.makedirs(args.tensorboard_dir, exist_ok=True)
    os.makedirs(args.checkpoint_dir, exist_ok=True)
    ps = (args.patch_size, args.patch_size)
    train_generator = ImageDataGenerator(os.path.join(args.list_dir, "train.txt"), shuffle=True, patch_size=ps,
                                         rng=np.random.default_rng(args.seed + 1000 * rank))
    val_generator = ImageDataGenerator(os.path.join(args.list_dir, "val.txt"), shuffle=False, patch_size=ps,
                                       rng=np.random.default_rng(args.seed + 7))
    train_batches_per_epoch = train_generator.data_size
    val_batches_per_epoch = val_generator.data_size

    net = NET(None, input_patch_size=args.patch_size, num_conv_layers=(args.patch_size - 1) // 2,
              batch_size=args.batch_size, device=device, seed=args.seed)
    trainer = Trainer(net, args.learning_rate, args.beta, args.margin)
    if args.resume is not None:
        trainer.load_state(args.resume)
    log = open(os.path.join(args.tensorboard_dir, "scalars.jsonl"), "a") if rank == 0 else None

    def scalar(tag, value, step):
        if log is not None:
            log.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")
            log.flush()

    print("[{}] {}: {} training pairs, {} validation pairs, {} rank(s) on {}".format(
        rank, datetime.now(), train_batches_per_epoch, val_batches_per_epoch, world, device))
    for epoch in range(args.start_epoch, args.end_epoch):
        for batch in range(train_batches_per_epoch):
            loss = trainer.step(*train_generator.next_batch(args.batch_size))
            if (batch + 1) % args.print_freq == 0:
                scalar("hinge_loss", loss, epoch * train_batches_per_epoch + batch)        # train.py:169-173
        if (epoch + 1) % args.save_freq == 0 and rank == 0:
            name = os.path.join(args.checkpoint_dir, "model_epoch" + str(epoch + 1) + ".ckpt.npz")
            np.savez(name, **trainer.state())
            print("[{}] {}: epoch {} saved to {}".format(rank, datetime.now(), epoch + 1, name))
        if (epoch + 1) % args.val_freq == 0:
            with torch.no_grad():
                val_ls = sum(float(trainer.loss(*val_generator.next_batch(args.batch_size)))
                             for _ in range(val_batches_per_epoch)) / (1. * max(val_batches_per_epoch, 1))
            print("[{}] {}: epoch {} validation loss: {}".format(rank, datetime.now(), epoch + 1, val_ls))
            scalar("val_hinge_loss", val_ls, train_batches_per_epoch * (epoch + 1))        # train.py:196-197
        val_generator.reset_pointer()
        train_generator.reset_pointer()
    if log is not None:
        log.close()
    mgpu.finalize()


if __name__ == "__main__":
    main()
