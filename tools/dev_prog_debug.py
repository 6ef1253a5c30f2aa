#!/usr/bin/env python3
"""Dev (GPU box): one small case through the assembled kernel, the simulator and cbca_hwd; prints where they differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mc-cnn-python_amd/src", "mc-cnn-python_amd/csrc/asm", "tests/asmtools", "tests", "oracle", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import stereo_device as sd, synthetic, cbca_prog_gen as gen, cbca_prog_ref as ref, asm_sim
import dev_prog_check as dc

def main():
    H, W, D = 12, 17, 8
    for debug in (1, 0):
        P = gen.Params(vpl=4, W=12, debug=debug)
        g = gen.Gen(P).build(); L = g.layout()
        base = "/tmp/dbg%d" % debug
        open(base + ".s", "w").write(g.render())
        os.system("%s/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c %s.s -o %s.o && %s/ld.lld -shared %s.o -o %s.hsaco" % (dc.LLVM, base, base, dc.LLVM, base, base))
        mod = dc.Module(base + ".hsaco", P.name())
        Li = synthetic.make_pair(H, W, 8, seed=0)[0]
        img = torch.from_numpy(Li[:, :, 0]).cuda()
        sup = sd.cross_arms(img, 0.02, 14)
        Dp = sd.hwd_pitch(D)
        a = (torch.arange(H * W * Dp, device="cuda", dtype=torch.float32).reshape(H, W, Dp) % 1000) * 0.001 + 1.0
        progs, meta = dc.programs_for(sup, H, W, L)
        out = torch.full_like(a, float("nan"))
        mod.launch((8 * meta["band_groups"], meta["ngroups"], 1), dc.kargs([a, a], [out, out], [progs, progs], [sup, sup], Dp, H, W, 1, meta))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        # simulator on the same data
        mem = asm_sim.Memory()
        sup0 = sup.cpu().numpy().view(np.uint32).reshape(H, W)
        a_in = mem.alloc(a.cpu().numpy()); a_out = mem.alloc(np.full((H, W, Dp), np.nan, np.float32))
        a_prog = mem.alloc(progs.cpu().numpy()); a_sup = mem.alloc(np.concatenate([sup0.reshape(-1), np.zeros(64, np.uint32)]))
        karg = np.frombuffer(dc.kargs([a, a], [out, out], [progs, progs], [sup, sup], Dp, H, W, 1, meta), np.uint32).copy()
        for i, v in ((0, a_in), (2, a_in), (4, a_out), (6, a_out), (8, a_prog), (10, a_prog), (12, a_sup), (14, a_sup)):
            karg[i], karg[i + 1] = v & 0xffffffff, v >> 32
        a_k = mem.alloc(karg)
        wave = asm_sim.Wave(g, mem)
        for bx in range(8 * meta["band_groups"]):
            for by in range(meta["ngroups"]):
                wave.run({0: a_k & 0xffffffff, 1: a_k >> 32, 2: bx, 3: by, 4: 0}, np.arange(64, dtype=np.uint32), P.nvgpr)
        simo = mem.get(a_out, np.float32, H * W * Dp).reshape(H, W, Dp)
        same = (got == simo) | (np.isnan(got) & np.isnan(simo))
        print("debug=%d: gpu == simulator on %d of %d values" % (debug, same.sum(), same.size))
        np.set_printoptions(linewidth=250, precision=4, suppress=True)
        print("pixel map of agreement (all d):"); print(same.all(axis=2).astype(int))
        print("gpu   d=0:"); print(got[:, :, 0])
        print("sim   d=0:"); print(simo[:, :, 0])
        cnt = (sup0 >> 20)
        print("region sizes:"); print(cnt)

main()
