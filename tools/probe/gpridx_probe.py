#!/usr/bin/env python3
"""Probe (GPU box): what VGPR index mode does on gfx950 - which operands M0[15:12] selects, whether a plain SALU write
of M0 changes index and enables while the mode is on, and whether v_readlane / VOP1 are touched by SRC1 mode.
Each variant is a few instructions between a common prologue (v10..v41 = 1000 + register number as float, v0..v7 = 0)
and epilogue (v0..v7 -> out[r * 64 + lane]).   python tools/probe/gpridx_probe.py"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

LLVM = "/opt/rocm/lib/llvm/bin"
OUT = "/tmp/gpridx"

VARIANTS = {
    "on SRC1 idx=3: v_add_f32 v0, v1, v10": """
  s_mov_b32 s8, 3
  s_set_gpr_idx_on s8, gpr_idx(SRC1)
  v_add_f32 v0, v1, v10
  s_set_gpr_idx_off
""",
    "on SRC0 idx=3: v_add_f32 v0, v10, v1": """
  s_mov_b32 s8, 3
  s_set_gpr_idx_on s8, gpr_idx(SRC0)
  v_add_f32 v0, v10, v1
  s_set_gpr_idx_off
""",
    "on SRC1 idx=0 then s_mov m0 = 0x2005: v_add_f32 v0, v1, v10": """
  s_mov_b32 s8, 0
  s_set_gpr_idx_on s8, gpr_idx(SRC1)
  s_mov_b32 m0, 0x2005
  s_nop 4
  v_add_f32 v0, v1, v10
  s_set_gpr_idx_off
""",
    "on SRC1 idx=0 then s_lshr m0 (0x20070000 >> 16), 3 salu, v_add v0, v1, v10 ; m0 readback in v2": """
  s_mov_b32 s8, 0
  s_set_gpr_idx_on s8, gpr_idx(SRC1)
  s_mov_b32 s9, 0x20070000
  s_lshr_b32 m0, s9, 16
  s_add_u32 s10, s10, 1
  s_addc_u32 s11, s11, 0
  s_nop 0
  v_add_f32 v0, v1, v10
  s_set_gpr_idx_off
  s_mov_b32 s12, m0
  v_mov_b32 v2, s12
""",
    "on SRC1 idx=4: 4 back-to-back v_add (v0..v3) += v10..v13": """
  s_mov_b32 s8, 4
  s_set_gpr_idx_on s8, gpr_idx(SRC1)
  v_add_f32 v0, v0, v10
  v_add_f32 v1, v1, v11
  v_add_f32 v2, v2, v12
  v_add_f32 v3, v3, v13
  s_set_gpr_idx_off
""",
    "on SRC1 idx=6: v_readlane s9 <- v10 lane s10=5 ; v_mov v3 <- v10 (VOP1)": """
  s_mov_b32 s8, 6
  s_mov_b32 s10, 5
  s_set_gpr_idx_on s8, gpr_idx(SRC1)
  v_readlane_b32 s9, v10, s10
  v_mov_b32 v3, v10
  s_set_gpr_idx_off
  v_mov_b32 v0, s9
""",
    "m0 after s_set_gpr_idx_on s8=3 SRC1 (v0 bits)": """
  s_mov_b32 m0, 0
  s_mov_b32 s8, 3
  s_set_gpr_idx_on s8, gpr_idx(SRC1)
  s_mov_b32 s9, m0
  s_set_gpr_idx_off
  v_mov_b32 v0, s9
""",
    "setpc then indexed add (idx=2 via s_lshr m0), like the dispatcher": """
  s_mov_b32 s8, 0
  s_set_gpr_idx_on s8, gpr_idx(SRC1)
  s_mov_b32 s9, 0x20020000
  s_getpc_b64 s[12:13]
  s_lshr_b32 m0, s9, 16
  s_add_u32 s12, s12, 20
  s_addc_u32 s13, s13, 0
  s_setpc_b64 s[12:13]
  s_nop 0
  v_add_f32 v0, v0, v10
  v_add_f32 v1, v1, v11
  s_set_gpr_idx_off
""",
}


def source(name, body):
    pro = ["  s_load_dwordx2 s[4:5], s[0:1], 0x0", "  v_lshlrev_b32 v9, 2, v0"]
    for r in range(10, 42):
        pro.append("  v_mov_b32 v%d, %s" % (r, hex(np.float32(1000 + r).view(np.uint32))))
    for r in range(0, 8):
        pro.append("  v_mov_b32 v%d, 0" % r)
    pro.append("  s_waitcnt lgkmcnt(0)")
    epi = ["  s_and_b32 s5, s5, 0xffff", "  s_mov_b32 s6, 0x10000", "  s_mov_b32 s7, 0x00020000"]
    for r in range(8):
        epi.append("  buffer_store_dword v%d, v9, s[4:7], 0 offen offset:%d" % (r, 256 * r))
    epi.append("  s_endpgm")
    return "\n".join(['.amdgcn_target "amdgcn-amd-amdhsa--gfx950"', ".text", ".globl %s" % name, ".p2align 8",
                      ".type %s,@function" % name, "%s:" % name] + pro + [body] + epi + [
        ".rodata", ".p2align 6", ".amdhsa_kernel %s" % name, "  .amdhsa_kernarg_size 8", "  .amdhsa_user_sgpr_count 2",
        "  .amdhsa_user_sgpr_kernarg_segment_ptr 1", "  .amdhsa_system_vgpr_workitem_id 0", "  .amdhsa_next_free_vgpr 48",
        "  .amdhsa_next_free_sgpr 32", "  .amdhsa_accum_offset 48", "  .amdhsa_ieee_mode 1", "  .amdhsa_dx10_clamp 1",
        "  .amdhsa_float_denorm_mode_32 3", ".end_amdhsa_kernel", ".amdgpu_metadata", "---", "amdhsa.version:", "  - 1",
        "  - 2", "amdhsa.kernels:", "  - .name: %s" % name, "    .symbol: %s.kd" % name, "    .kernarg_segment_size: 8",
        "    .kernarg_segment_align: 8", "    .group_segment_fixed_size: 0", "    .private_segment_fixed_size: 0",
        "    .wavefront_size: 64", "    .sgpr_count: 40", "    .vgpr_count: 48", "    .agpr_count: 0",
        "    .max_flat_workgroup_size: 64", "    .args:", "      - .offset: 0", "        .size: 8",
        "        .value_kind: by_value", "amdhsa.target: amdgcn-amd-amdhsa--gfx950", "...", ".end_amdgpu_metadata", ""])


def main():
    os.makedirs(OUT, exist_ok=True)
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    torch.cuda.init()
    for n, (title, body) in enumerate(VARIANTS.items()):
        name = "probe%d" % n
        base = os.path.join(OUT, name)
        open(base + ".s", "w").write(source(name, body))
        subprocess.check_call([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c",
                               base + ".s", "-o", base + ".o"])
        subprocess.check_call([LLVM + "/ld.lld", "-shared", base + ".o", "-o", base + ".hsaco"])
        mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipModuleLoad(ctypes.byref(mod), (base + ".hsaco").encode()) == 0
        assert hip.hipModuleGetFunction(ctypes.byref(fn), mod, name.encode()) == 0
        out = torch.zeros((8, 64), dtype=torch.float32, device="cuda")
        karg = np.array([out.data_ptr()], np.uint64).tobytes()
        buf = ctypes.create_string_buffer(karg, len(karg))
        size = ctypes.c_size_t(len(karg))
        extra = (ctypes.c_void_p * 5)(1, ctypes.cast(buf, ctypes.c_void_p).value, 2,
                                       ctypes.cast(ctypes.pointer(size), ctypes.c_void_p).value, 3)
        rc = hip.hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, None, None, extra)
        assert rc == 0, rc
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        print("%s\n    v0..v7 lane0 = %s\n    as bits  = %s" % (title, [float(x) for x in o[:, 0]],
                                                               [hex(int(x)) for x in o[:, 0].view(np.uint32)]), flush=True)


if __name__ == "__main__":
    main()
