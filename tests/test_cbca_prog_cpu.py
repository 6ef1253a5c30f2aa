"""CPU checks of the program-driven reference-order aggregation (csrc/asm/cbca_prog_gen.py):
  * the generator's own size model equals what the assembler emits (label offsets = op encodings),
  * the plain-Python program builder + op-level interpreter reproduce the oracle's pf:149-163 bit for bit,
  * the generated kernel, executed instruction by instruction by tests/asmtools/asm_sim.py on the same programs,
    reproduces the oracle bit for bit (control flow, relative window addressing, waitcnt discipline, epilogue).
No GPU, no compute through the product library."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mc-cnn-python_amd", "csrc", "asm"))
sys.path.insert(0, os.path.join(ROOT, "tests", "asmtools"))
import cbca_prog_gen as gen          # noqa: E402
import cbca_prog_ref as ref          # noqa: E402
import asm_sim                       # noqa: E402
import oracle as o                   # noqa: E402
import synthetic                     # noqa: E402
from helpers import assert_bits_strict  # noqa: E402

LLVM = "/opt/rocm/lib/llvm/bin"


def support_words(img, tau=0.02, Lmax=14):
    arms, cnt = o.cross_arms(img, tau, Lmax)
    a = arms.astype(np.uint32)
    return (a[..., 0] | (a[..., 1] << 5) | (a[..., 2] << 10) | (a[..., 3] << 15) | (cnt.astype(np.uint32) << 20)).astype(np.uint32)


def make_case(H, W, D, seed, flat=False):
    rng = np.random.default_rng(seed)
    if flat:
        img = np.zeros((H, W, 1), np.float32)
    else:
        img = synthetic.make_pair(H, W, min(D, max(W - 2, 1)), seed=seed)[0]
    vol = (rng.random((D, H, W), dtype=np.float32) * 3 - 2).astype(np.float32)
    return img, vol


def oracle_cbca(img, vol):
    out, _ = o.cost_volume_aggregation(img, img, vol, vol, 0.02, 14, 1)
    return out


@pytest.mark.parametrize("vpl,w,nb,k", [(4, 12, 1, 2), (3, 12, 1, 2), (2, 12, 1, 2), (4, 8, 1, 2), (4, 10, 2, 2), (4, 20, 1, 4),
                                         (2, 16, 1, 4)])
def test_size_model_matches_the_assembler(tmp_path, vpl, w, nb, k):
    if not os.path.exists(os.path.join(LLVM, "clang")):
        pytest.skip("no ROCm assembler")
    g = gen.Gen(gen.Params(vpl=vpl, K=k, W=w, NB=nb)).build()
    s = tmp_path / "k.s"
    s.write_text(g.render())
    obj = tmp_path / "k.o"
    subprocess.check_call([os.path.join(LLVM, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa",
                           "-mcpu=gfx950", "-c", str(s), "-o", str(obj)])
    dis = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", str(obj)]).decode()
    off = g.offsets()
    addr = {}
    for line in dis.splitlines():
        if "//" in line and ":" in line.split("//")[1]:
            a = int(line.split("//")[1].split(":")[0].strip(), 16)
            addr.setdefault(line.split("//")[0].strip().split()[0], []).append(a)
    assert addr["s_set_gpr_idx_off"] == [off["end"]]
    assert addr["s_endpgm"] == [off["done"]]
    assert addr["s_getpc_b64"][0] + 4 == off["after_getpc"]


@pytest.mark.parametrize("H,W,D,seed,flat", [(12, 17, 8, 0, False), (30, 41, 5, 1, False), (9, 33, 4, 2, True),
                                               (40, 48, 6, 3, False)])
@pytest.mark.parametrize("w,nb,k,pipe", [(12, 1, 2, 0), (8, 1, 2, 0), (10, 2, 2, 0), (6, 2, 2, 0), (20, 1, 4, 0), (12, 1, 4, 0),
                                          (16, 1, 1, 0), (42, 1, 4, 21), (42, 1, 4, 42), (24, 1, 4, 7), (30, 1, 2, 12)])
def test_programs_reproduce_the_oracle_op_level(H, W, D, seed, flat, w, nb, k, pipe):
    L = gen.Gen(gen.Params(vpl=4, K=k, W=w, NB=nb, pipe=pipe)).build().layout()
    L["pix"] = 4 * D
    img, vol = make_case(H, W, D, seed, flat)
    sup0 = support_words(img)
    want = oracle_cbca(img, vol)
    hwd = np.ascontiguousarray(vol.transpose(1, 2, 0))
    got = np.full((H, W, D), np.nan, np.float32)
    for y0 in range(0, H, L["K"]):
        for x0 in range(0, W, L["G"]):
            prog = ref.build_program(sup0, H, W, y0, x0, L)
            assert len(prog) <= ref.prog_stride_dwords(L)
            for (y, x), q in ref.run_program(prog, hwd, H, W, y0, x0, L, sup0).items():
                got[y, x] = q
    assert np.array_equal(got.transpose(2, 0, 1), want)


@pytest.mark.parametrize("vpl,k,w", [(4, 4, 20), (4, 2, 12), (3, 4, 20), (2, 4, 20), (4, 4, 9), (4, 1, 16)])
def test_cxx_builder_equals_the_python_statement(tmp_path, vpl, k, w):
    """csrc/cbca_prog_build.h (what the device kernel runs per patch), compiled for the host, against
    tests/asmtools/cbca_prog_ref.py: the same ops, word for word, on textured, flat and tiny images."""
    import ctypes
    P = gen.Params(vpl=vpl, K=k, W=w)
    g = gen.Gen(P).build()
    L = g.layout()
    (tmp_path / "layout.h").write_text(gen.header(L, P) + "#define TEST_LAYOUT CBCA_PROG_V%d_LAYOUT\n" % vpl)
    so = tmp_path / "libprog.so"
    subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", "-std=c++17",
                           "-I" + os.path.join(ROOT, "mc-cnn-python_amd", "csrc"), "-I" + str(tmp_path),
                           os.path.join(ROOT, "tests", "asmtools", "prog_build_host.cpp"), "-o", str(so)])
    lib = ctypes.CDLL(str(so))
    assert lib.prog_stride_dwords() == ref.prog_stride_dwords(L)
    cap = lib.prog_stride_dwords()
    u32p = ctypes.POINTER(ctypes.c_uint32)
    for (H, W, seed, flat) in ((12, 17, 0, False), (30, 41, 1, False), (9, 33, 2, True), (40, 48, 3, False), (3, 4, 4, False),
                               (31, 64, 5, True)):
        img, _ = make_case(H, W, 4, seed, flat)
        sup0 = np.ascontiguousarray(support_words(img))
        padded = np.concatenate([sup0.reshape(-1), np.zeros(16, np.uint32)])   # the builder reads whole groups of G words
        assert lib.prog_band_rows(H) == ref.band_rows_of(H, k)
        for y0 in range(0, H, k):
            for x0 in range(0, W, L["G"]):
                # the full programs and the ones that leave out the anchors whose region is the pixel itself
                for skip, fn in ((False, lib.prog_build_patch), (True, lib.prog_build_patch_skip)):
                    want = ref.build_program(sup0, H, W, y0, x0, L, skip_unit=skip)
                    got = np.zeros(cap, np.uint32)
                    n = fn(padded.ctypes.data_as(u32p), H, W, y0, x0, got.ctypes.data_as(u32p), cap)
                    assert n == len(want) <= cap, (H, W, y0, x0, skip, n, len(want))
                    assert np.array_equal(got[:n], want), (H, W, y0, x0, skip)


def simulate(g, L, img, vol, code_addr=0x7e00fffff000, out_init=None, return_in=False):
    """Runs every workgroup of one single-volume launch of the generated kernel in the simulator.  The skip kernel
    (g.P.skip) runs the skip programs; out_init [D, H, W] is then what the output buffer holds beforehand."""
    P = g.P
    D, H, W = vol.shape
    VPL = P.VPL
    Dp = -(-D // 4) * 4 if VPL != 3 else -(-D // 3) * 3
    sup0 = support_words(img)
    L = dict(L, pix=4 * Dp)
    progs, meta = ref.build_all(sup0, H, W, L, skip_unit=P.skip)
    hwd = np.zeros((H, W, Dp), np.float32)
    hwd[:, :, :D] = vol.transpose(1, 2, 0)
    mem = asm_sim.Memory()
    a_in = mem.alloc(hwd)
    out0 = np.full((H, W, Dp), np.nan, np.float32)
    if out_init is not None:
        out0[:, :, :D] = out_init.transpose(1, 2, 0)
    a_out = mem.alloc(out0)
    a_prog = mem.alloc(progs)
    a_sup = mem.alloc(np.concatenate([sup0.reshape(-1), np.zeros(64, np.uint32)]))
    nchunks = -(-Dp // (64 * VPL))
    karg = np.zeros(0x60 // 4, np.uint32)

    def put64(i, v):
        karg[i], karg[i + 1] = v & 0xffffffff, v >> 32
    put64(0, a_in), put64(2, a_in), put64(4, a_out), put64(6, a_out), put64(8, a_prog), put64(10, a_prog)
    put64(12, a_sup), put64(14, a_sup)
    karg[16:24] = [Dp, H, W, nchunks, meta["band_rows"], meta["band_groups"], meta["stride"] * 4, meta["ngroups"]]
    a_k = mem.alloc(karg)
    wave = asm_sim.Wave(g, mem, code_addr=code_addr)
    tot = dict(ins=0, valu=0, salu=0, vmem=0, setpc=0)
    for bx in range(8 * meta["band_groups"]):
        for by in range(meta["ngroups"]):
            for bz in range(nchunks):
                st = wave.run({0: a_k & 0xffffffff, 1: a_k >> 32, 2: bx, 3: by, 4: bz}, np.arange(64, dtype=np.uint32),
                              P.nvgpr)
                for k in tot:
                    tot[k] += st[k]
    out = mem.get(a_out, np.float32, H * W * Dp).reshape(H, W, Dp)[:, :, :D]
    if return_in:
        vin = mem.get(a_in, np.float32, H * W * Dp).reshape(H, W, Dp)[:, :, :D]
        return out.transpose(2, 0, 1), tot, vin.transpose(2, 0, 1)
    return out.transpose(2, 0, 1), tot


def simulate_tile(g, L, img, vol, out_init=None):
    """Every workgroup of one single-volume launch of a TILE kernel (cbca_prog_gen.py, tile = NW waves per workgroup that
    share their region rows through LDS) in the simulator: NW waves stepped from barrier to barrier over one LDS."""
    P = g.P
    D, H, W = vol.shape
    Dp = -(-D // 4) * 4 if P.VPL != 3 else -(-D // 3) * 3
    sup0 = support_words(img)
    L = dict(L, pix=4 * Dp)
    progs, meta = ref.build_all_tiles(sup0, H, W, L, skip_unit=P.skip)
    hwd = np.zeros((H, W, Dp), np.float32)
    hwd[:, :, :D] = vol.transpose(1, 2, 0)
    mem = asm_sim.Memory()
    a_in = mem.alloc(hwd)
    out0 = np.full((H, W, Dp), np.nan, np.float32)
    if out_init is not None:
        out0[:, :, :D] = out_init.transpose(1, 2, 0)
    a_out = mem.alloc(out0)
    a_prog = mem.alloc(progs)
    a_sup = mem.alloc(np.concatenate([sup0.reshape(-1), np.zeros(64, np.uint32)]))
    nchunks = -(-Dp // (64 * P.VPL))
    karg = np.zeros(0x60 // 4, np.uint32)

    def put64(i, v):
        karg[i], karg[i + 1] = v & 0xffffffff, v >> 32
    put64(0, a_in), put64(2, a_in), put64(4, a_out), put64(6, a_out), put64(8, a_prog), put64(10, a_prog)
    put64(12, a_sup), put64(14, a_sup)
    karg[16:24] = [Dp, H, W, nchunks, meta["band_rows"], meta["band_groups"], meta["stride"] * 4, meta["ngroups"]]
    a_k = mem.alloc(karg)
    tot = {}
    for bx in range(8 * meta["band_groups"]):
        for by in range(meta["ntx"]):
            for bz in range(nchunks):
                st = asm_sim.run_workgroup(g, mem, P.tile, {0: a_k & 0xffffffff, 1: a_k >> 32, 2: bx, 3: by, 4: bz}, P.nvgpr,
                                           P.LDS_BYTES)
                for k, v in st.items():
                    tot[k] = tot.get(k, 0) + v
    out = mem.get(a_out, np.float32, H * W * Dp).reshape(H, W, Dp)[:, :, :D]
    return out.transpose(2, 0, 1), tot, meta


@pytest.mark.parametrize("vpl,nw,H,W,D,seed,flat", [(4, 4, 12, 17, 8, 0, False), (4, 4, 9, 33, 4, 2, True), (4, 4, 24, 45, 6, 3, False),
                                                     (4, 2, 14, 23, 5, 4, False), (2, 4, 11, 26, 6, 5, False),
                                                     (4, 4, 7, 12, 300, 7, False), (4, 4, 16, 44, 4, 8, True)])
def test_tile_kernel_reproduces_the_oracle_in_the_simulator(vpl, nw, H, W, D, seed, flat):
    """The kernel whose region rows travel through LDS once per TILE of nw patches (a workgroup of nw waves in lock
    step: STEP = barrier, each wave's share of the row by buffer_load ... lds, wait, barrier; LOADL = a patch's window
    out of LDS; the unchanged ADD lines): the oracle's bits, with the simulator checking that no wave reads LDS bytes
    whose request has not been retired by its issuer and published by a barrier."""
    g = gen.Gen(gen.Params(vpl=vpl, K=4, W=20, tile=nw)).build()
    L = g.layout()
    img, vol = make_case(H, W, D, seed, flat)
    got, st, meta = simulate_tile(g, L, img, vol)
    assert_bits_strict(got, oracle_cbca(img, vol), "tile kernel in the simulator against the oracle")
    assert meta["steps"] > 0 and meta["slots"] >= meta["steps"]


def test_tile_skip_kernel_equals_the_oracle_in_the_simulator():
    g = gen.Gen(gen.Params(vpl=4, K=4, W=20, tile=4, skip=True)).build()
    L = g.layout()
    img, vol0 = make_case(24, 45, 6, 3)
    v1 = oracle_cbca(img, vol0)
    v2 = oracle_cbca(img, v1)
    v3 = oracle_cbca(img, v2)
    unit = (support_words(img) & 0xfffff) == 0
    assert unit.any() and not unit.all()
    got, _, _ = simulate_tile(g, L, img, v2, out_init=v1)
    assert_bits_strict(got, v3, "tile skip kernel against the oracle's third iteration")
    untouched, _, _ = simulate_tile(g, L, img, v2)
    assert np.isnan(untouched[:, unit]).all() and np.array_equal(untouched[:, ~unit], v3[:, ~unit])


@pytest.mark.parametrize("vpl,w,nb,H,W,D,seed,flat", [(4, 12, 1, 12, 17, 8, 0, False), (4, 12, 1, 9, 33, 4, 2, True),
                                                       (3, 12, 1, 14, 23, 6, 4, False), (2, 12, 1, 11, 16, 6, 5, False),
                                                       (4, 8, 1, 13, 21, 5, 6, True), (4, 12, 1, 7, 12, 300, 7, False),
                                                       (4, 10, 2, 12, 17, 8, 0, False), (4, 6, 2, 9, 33, 4, 2, True),
                                                       (3, 10, 2, 14, 23, 6, 4, False),
                                                       # nb = 0: the program-managed ring of 42 slots, widest unit w
                                                       (4, 21, 0, 12, 17, 8, 0, False), (4, 42, 0, 9, 33, 4, 2, True),
                                                       (3, 21, 0, 14, 23, 6, 4, False), (2, 21, 0, 11, 16, 6, 5, False),
                                                       (4, 9, 0, 13, 21, 5, 6, True), (4, 21, 0, 7, 12, 300, 7, False),
                                                       (4, 21, 0, 40, 48, 4, 3, False)])
@pytest.mark.parametrize("k", [2, 4])
def test_generated_kernel_reproduces_the_oracle_in_the_simulator(vpl, w, nb, H, W, D, seed, flat, k):
    g = gen.Gen(gen.Params(vpl=vpl, K=k, W=w if nb else 42, NB=nb or 1, pipe=0 if nb else w)).build()
    L = g.layout()
    img, vol = make_case(H, W, D, seed, flat)
    got, st = simulate(g, L, img, vol)
    want = oracle_cbca(img, vol)
    assert_bits_strict(got, want, "simulated kernel against the oracle")


@pytest.mark.parametrize("vpl,w,k,H,W,D,seed,flat", [(4, 20, 4, 12, 17, 8, 0, False), (4, 12, 2, 14, 23, 5, 1, False),
                                                     (3, 12, 4, 14, 23, 6, 4, False), (2, 12, 4, 11, 16, 6, 5, False),
                                                     (4, 20, 4, 9, 33, 4, 2, True), (4, 20, 4, 24, 32, 8, 3, False)])
def test_skip_kernel_leaves_unit_regions_alone_and_equals_the_oracle_in_the_simulator(vpl, w, k, H, W, D, seed, flat):
    """The skip programs + the skip kernel (third and later iterations of a ping-pong pair): pixels whose support region
    is the pixel itself are neither read for their own sake nor written - the output buffer must already hold their
    value (here: the oracle's second iteration) - and every other pixel gets the oracle's bits.  Unit-region pixels that
    were NOT pre-filled stay NaN: the kernel really does not touch them."""
    g = gen.Gen(gen.Params(vpl=vpl, K=k, W=w, skip=True)).build()
    L = g.layout()
    img, vol0 = make_case(H, W, D, seed, flat)
    v1 = oracle_cbca(img, vol0)
    v2 = oracle_cbca(img, v1)
    v3 = oracle_cbca(img, v2)
    unit = (support_words(img) & 0xfffff) == 0
    assert flat or unit.any()
    assert np.array_equal(v3[:, unit], v2[:, unit]) and np.array_equal(v2[:, unit], v1[:, unit])     # the fixed point
    got, _ = simulate(g, L, img, v2, out_init=v1)          # iteration 3 writes into the buffer iteration 1 wrote
    assert_bits_strict(got, v3, "skip kernel in the simulator against the oracle's third iteration")
    untouched, _ = simulate(g, L, img, v2)                 # nothing pre-filled: unit regions stay NaN
    assert np.isnan(untouched[:, unit]).all() and np.array_equal(untouched[:, ~unit], v3[:, ~unit])


def test_unit_regions_are_fixed_points_after_one_iteration_including_negative_zero():
    """What the skip variant rests on (pf:156-161 with aver_num = 1): v1 = (0 + v0) / 1 turns -0.0 into +0.0 and is
    otherwise v0; v2 = v1 bit for bit, also for inf, NaN and subnormals."""
    img, vol = make_case(24, 32, 8, 3)
    unit = (support_words(img) & 0xfffff) == 0
    assert unit.any()
    vol[:, unit] = np.resize(np.array([-0.0, 0.0, np.inf, -np.inf, np.nan, 1e-42, -1e-42, 1.5], np.float32),
                             vol[:, unit].shape)
    v1 = oracle_cbca(img, vol)
    v2 = oracle_cbca(img, v1)
    a, b = v1[:, unit].view(np.uint32), v2[:, unit].view(np.uint32)
    nan = np.isnan(v1[:, unit])
    assert np.array_equal(a[~nan], b[~nan]) and np.isnan(v2[:, unit][nan]).all()
    assert not np.signbit(v1[:, unit][vol[:, unit] == 0]).any()          # -0.0 became +0.0 in the first iteration
    assert np.array_equal(v1[:, unit][~nan & (vol[:, unit] != 0)], vol[:, unit][~nan & (vol[:, unit] != 0)])


def _skip_schedule():
    """stereo_device.skip_schedule without importing torch: the function is pure Python."""
    src = open(os.path.join(ROOT, "mc-cnn-python_amd", "src", "stereo_device.py")).read()
    ns = {}
    exec(src[src.index("def skip_schedule"):src.index("def cbca_prog_pair")], ns)
    return ns["skip_schedule"]


def _canon(a):
    u = a.view(np.uint32).copy()
    u[np.isnan(a)] = 0x7fc00000
    return u


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 16])
def test_refresh_schedule_equals_the_oracle(n, fused):
    """stereo_device.skip_schedule (round 6) restated on the CPU checker: "refresh" = a full iteration that also writes
    v1 = (0 + v0) / 1 of every unit-region pixel back into its INPUT buffer (the kernel's store races with neighbours that
    read the pixel: modelled here by running the iteration on the input with v1 ALREADY in place, and on the input as
    it was - both must give the same bits), "skip" leaves the unit-region pixels of its destination alone, "wta" / "full"
    write everything.  With -0.0 / inf / NaN / subnormals on such pixels the final buffer equals n iterations of
    pf:149-163 bit for bit whichever buffer the last iteration writes."""
    kinds = _skip_schedule()(n, fused)
    assert len(kinds) == n and ("refresh" not in kinds[1:]) and (kinds[-1] == "wta") == fused
    img, vol = make_case(24, 32, 6, 3)
    unit = (support_words(img) & 0xfffff) == 0
    assert unit.any() and not unit.all()
    vol[:, unit] = np.resize(np.array([-0.0, 0.0, np.inf, -np.inf, np.nan, 1e-42, -1e-42, 1.5, -0.0], np.float32),
                             vol[:, unit].shape)
    want = vol
    for _ in range(n):
        want = oracle_cbca(img, want)
    bufs = [vol.copy(), np.full_like(vol, 12345.0)]
    for it, kind in enumerate(kinds):
        src, dst = bufs[it % 2], bufs[1 - it % 2]
        res = oracle_cbca(img, src)
        if kind == "refresh":
            refreshed = src.copy()
            refreshed[:, unit] = res[:, unit]
            again = oracle_cbca(img, refreshed)           # what a wave computes that reads the already rewritten pixels
            assert np.array_equal(_canon(again), _canon(res))
            bufs[it % 2] = refreshed
        if kind == "skip":
            res[:, unit] = dst[:, unit]
        bufs[1 - it % 2] = res
    got = bufs[n % 2]
    assert np.array_equal(_canon(got), _canon(want))
    if n >= 2 and n % 2 == 0 and not fused:               # the case the refresh launch exists for
        assert kinds[0] == "refresh" and set(kinds[1:]) == {"skip"}


def test_refresh_kernel_in_the_simulator():
    """The generated refresh kernel: the oracle's bits in `out`, and in `in` v1 at every unit-region pixel (-0.0 became
    +0.0 there) with every other input voxel untouched."""
    g = gen.Gen(gen.Params(vpl=4, K=4, W=20, refresh=True)).build()
    L = g.layout()
    img, vol = make_case(24, 32, 6, 3)
    unit = (support_words(img) & 0xfffff) == 0
    vol[:, unit] = np.resize(np.array([-0.0, 0.0, np.inf, -np.inf, np.nan, 1e-42, -1e-42, 1.5, -0.0], np.float32),
                             vol[:, unit].shape)
    want = oracle_cbca(img, vol)
    got, _, vin = simulate(g, L, img, vol, return_in=True)
    assert_bits_strict(got, want, "refresh kernel, output")
    assert_bits_strict(vin[:, unit], want[:, unit], "refresh kernel, unit-region pixels of the input")
    assert np.array_equal(vin[:, ~unit].view(np.uint32), vol[:, ~unit].view(np.uint32))
    assert (np.signbit(vol[:, unit]) & (vol[:, unit] == 0)).any() and not (np.signbit(vin[:, unit]) & (vin[:, unit] == 0)).any()


@pytest.mark.parametrize("n", [2, 3, 4, 5, 16])
def test_skipping_from_the_second_iteration_on_equals_the_oracle(n):
    """Round 5's schedule (skip_schedule(refresh_first=False)), restated on the CPU checker: iteration 1 full (in -> out), every
    later iteration leaves the unit-region pixels of its destination as they are - v1 in `out`, the ORIGINAL v0 in `in` -
    and an even count ends with a full iteration.  With -0.0 / inf / NaN / subnormals on such pixels the final buffer
    equals n iterations of pf:149-163 bit for bit: as an operand of another pixel's sum v0 and v1 are interchangeable."""
    img, vol = make_case(24, 32, 6, 3)
    unit = (support_words(img) & 0xfffff) == 0
    assert unit.any() and not unit.all()
    vol[:, unit] = np.resize(np.array([-0.0, 0.0, np.inf, -np.inf, np.nan, 1e-42, -1e-42, 1.5, -0.0], np.float32),
                             vol[:, unit].shape)
    want = vol
    for _ in range(n):
        want = oracle_cbca(img, want)
    bufs = [vol.copy(), np.full_like(vol, 12345.0)]
    for it in range(n):
        src, dst = bufs[it % 2], bufs[1 - it % 2]
        res = oracle_cbca(img, src)
        skip = it >= 1 and not (n % 2 == 0 and it == n - 1)
        if skip:
            res[:, unit] = dst[:, unit]
        bufs[1 - it % 2] = res
    got = bufs[n % 2]
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_unit_region_fixed_point_for_every_kind_of_float32():
    """The arithmetic fact itself, on two million random bit patterns plus the special encodings: with aver_num = 1 the
    reference computes y = (0 + x) / 1 (pf:156-161, float32); y differs from x only for -0.0 and signalling NaNs, and
    (0 + y) / 1 == y bit for bit - so a pixel whose region is the pixel itself stops changing after one iteration."""
    rng = np.random.default_rng(5)
    bits = np.concatenate([rng.integers(0, 2 ** 32, 2_000_000, dtype=np.uint64).astype(np.uint32),
                           np.array([0x00000000, 0x80000000, 0x7f800000, 0xff800000, 0x7fc00000, 0xffc00000, 0x7f800001,
                                     0xff812345, 0x00000001, 0x80000001, 0x007fffff, 0x7f7fffff, 0xff7fffff], np.uint32)])
    x = bits.view(np.float32)
    with np.errstate(invalid="ignore"):
        y = ((np.float32(0) + x) / np.float32(1)).astype(np.float32)
        z = ((np.float32(0) + y) / np.float32(1)).astype(np.float32)
    assert np.array_equal(y.view(np.uint32), z.view(np.uint32))
    changed = y.view(np.uint32) != bits
    quiet = bits | np.uint32(0x00400000)
    assert np.all((bits[changed] == 0x80000000) | (np.isnan(x[changed]) & (y.view(np.uint32)[changed] == quiet[changed])))


@pytest.mark.parametrize("skip", [False, True])
def test_generated_kernel_with_early_first_op_in_the_simulator(skip):
    """Params(early=True) (experimental, measured slower, off by default): the program's first op from a scalar load, the
    first LOAD on its own copy of the line - and an empty skip program ends on that one word."""
    g = gen.Gen(gen.Params(vpl=4, K=4, W=20, early=True, skip=skip)).build()
    L = g.layout()
    img, vol0 = make_case(24, 32, 8, 3)
    if not skip:
        got, _ = simulate(g, L, img, vol0)
        assert np.array_equal(got, oracle_cbca(img, vol0))
    else:
        v1 = oracle_cbca(img, vol0)
        v2 = oracle_cbca(img, v1)
        got, _ = simulate(g, L, img, v2, out_init=v1)
        assert_bits_strict(got, oracle_cbca(img, v2), "early-dispatch skip kernel against the oracle")


def test_generated_kernel_with_scalar_prefetch_in_the_simulator():
    """The L2 warm-up loads never leave the volume (the simulator faults on any access outside an allocation)."""
    for (H, W, D, pf) in ((12, 17, 8, 5), (9, 33, 4, 20), (11, 16, 6, 10)):
        g = gen.Gen(gen.Params(vpl=4, W=12, PF=pf)).build()
        img, vol = make_case(H, W, D, 3)
        got, _ = simulate(g, g.layout(), img, vol)
        assert np.array_equal(got, oracle_cbca(img, vol))


def test_generated_wta_kernel_in_the_simulator(tmp_path):
    """The last-iteration kernel (aggregation + a7's first strict minimum): its aggregation part runs in the simulator
    like the plain kernel's, its reduction tail is evaluated as pf:245-254 states it; the assembler accepts the tail and
    the size model still holds (the tail's DPP instructions are 8 bytes)."""
    g = gen.Gen(gen.Params(vpl=4, K=4, W=20, wta=True)).build()
    s = tmp_path / "k.s"
    s.write_text(g.render())
    if os.path.exists(os.path.join(LLVM, "clang")):
        obj = tmp_path / "k.o"
        subprocess.check_call([os.path.join(LLVM, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa",
                               "-mcpu=gfx950", "-c", str(s), "-o", str(obj)])
        dis = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", str(obj)]).decode()
        ends = [int(ln.split("//")[1].split(":")[0].strip(), 16) for ln in dis.splitlines() if "s_endpgm" in ln]
        assert ends == [g.offsets()["done"]]
    L = g.layout()
    plain = gen.Gen(gen.Params(vpl=4, K=4, W=20)).build().layout()
    assert all(L[k] == plain[k] for k in ("add", "load", "wait", "refill", "end")), "one set of programs serves both kernels"
    img, vol = make_case(11, 14, 7, 3)
    D, H, W = vol.shape
    sup0 = support_words(img)
    progs, meta = ref.build_all(sup0, H, W, L)
    Dp = 8
    hwd = np.zeros((H, W, Dp), np.float32)
    hwd[:, :, :D] = vol.transpose(1, 2, 0)
    mem = asm_sim.Memory()
    a_in, a_out = mem.alloc(hwd), mem.alloc(np.full((H, W, Dp), np.nan, np.float32))
    a_prog = mem.alloc(progs)
    a_sup = mem.alloc(np.concatenate([sup0.reshape(-1), np.zeros(64, np.uint32)]))
    a_disp = mem.alloc(np.full((H, W), -7.0, np.float32))
    karg = np.zeros(0x80 // 4, np.uint32)
    for i, v in ((0, a_in), (2, a_in), (4, a_out), (6, a_out), (8, a_prog), (10, a_prog), (12, a_sup), (14, a_sup),
                 (24, a_disp), (26, a_disp)):
        karg[i], karg[i + 1] = v & 0xffffffff, v >> 32
    karg[16:24] = [Dp, H, W, 1, meta["band_rows"], meta["band_groups"], meta["stride"] * 4, meta["ngroups"]]
    karg[28], karg[29] = D, 1
    a_k = mem.alloc(karg)
    wave = asm_sim.Wave(g, mem)
    for bx in range(8 * meta["band_groups"]):
        for by in range(meta["ngroups"]):
            wave.run({0: a_k & 0xffffffff, 1: a_k >> 32, 2: bx, 3: by, 4: 0}, np.arange(64, dtype=np.uint32), g.P.nvgpr)
    want = oracle_cbca(img, vol)
    got = mem.get(a_out, np.float32, H * W * Dp).reshape(H, W, Dp)[:, :, :D].transpose(2, 0, 1)
    assert np.array_equal(got, want)
    disp = mem.get(a_disp, np.float32, H * W).reshape(H, W)
    assert np.array_equal(disp, np.argmin(want, axis=0).astype(np.float32))


def test_simulated_kernel_special_values_and_code_address_carry():
    """inf / NaN / signed zeros travel through the add chains like through the reference's running sum; the code base
    sits right below a 4 GiB boundary so that the dispatcher's 64-bit address arithmetic carries."""
    g = gen.Gen(gen.Params(vpl=4, W=12)).build()
    L = g.layout()
    img, vol = make_case(10, 14, 8, 11)
    vol[1, 3, 4] = np.inf
    vol[2, 5, 6] = -np.inf
    vol[3, 7, 8] = np.nan
    vol[4, :, :] = -0.0
    got, _ = simulate(g, L, img, vol, code_addr=0x7e00ffffff00)
    want = oracle_cbca(img, vol)
    assert np.array_equal(got.view(np.uint32) & 0x7fffffff if False else np.isnan(got), np.isnan(want))
    m = ~np.isnan(want)
    assert np.array_equal(got[m].view(np.uint32), want[m].view(np.uint32))
