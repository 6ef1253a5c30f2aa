#!/usr/bin/env python3
"""CPU-baseline worker (test infrastructure, like everything under oracle/): runs the C restatement of the reference's
timed region (match.py:129-179) on one HxW window at D disparities and prints the seconds it took.  bench.py's
`cpu_baseline` leg starts one of these per host core it wants to load; nothing on the product path imports this.

    python oracle/cpu_window.py H W D SEED WEIGHTS.npz
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "mc-cnn-python_amd", "src"))   # synthetic.py, tf_checkpoint.py (pure NumPy)


def main():
    H, W, D, seed = (int(x) for x in sys.argv[1:5])
    import oracle as o
    import synthetic
    import tf_checkpoint
    layers = tf_checkpoint.load_fast_net_weights(sys.argv[5])
    L, R, _, _, _ = synthetic.make_pair(H, W, D, seed=seed)
    o.lib()
    t0 = time.perf_counter()
    o.match_pair(L, R, D, layers)
    print("%.6f" % (time.perf_counter() - t0))


if __name__ == "__main__":
    main()
