#!/usr/bin/env python3
"""Probe (GPU box): does VGPR index mode cost VALU throughput?  Waves run N x 32 dependent-by-4 v_add_f32 with the
second operand addressed relatively (SRC1, M0 = 0x2004) or plainly; 1, 2, 4 waves per SIMD.
    python tools/probe/gpridx_rate.py"""
import ctypes, os, subprocess
import numpy as np, torch
LLVM = "/opt/rocm/lib/llvm/bin"; OUT = "/tmp/gpridx_rate"

def source(name, indexed, pk):
    body = ["  s_load_dwordx2 s[4:5], s[0:1], 0x0", "  s_load_dword s6, s[0:1], 0x8", "  v_lshlrev_b32 v60, 2, v0"]
    for r in range(0, 40):
        body.append("  v_mov_b32 v%d, 1.0" % r)
    body += ["  s_waitcnt lgkmcnt(0)", "  s_mov_b32 s8, 4"]
    if indexed:
        body.append("  s_set_gpr_idx_on s8, gpr_idx(SRC1)")
    body.append("loop_%s:" % name)
    for rep in range(8):
        for c in range(4):
            if pk:
                if c < 2:
                    body.append("  v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (2 * c, 2 * c + 1, 2 * c, 2 * c + 1, 8 + 4 * (rep % 4) + 2 * c, 9 + 4 * (rep % 4) + 2 * c))
            else:
                body.append("  v_add_f32 v%d, v%d, v%d" % (c, c, 8 + 4 * (rep % 4) + c))
    body += ["  s_sub_u32 s6, s6, 1", "  s_cmp_lg_u32 s6, 0", "  s_cbranch_scc1 loop_%s" % name]
    if indexed:
        body.append("  s_set_gpr_idx_off")
    body += ["  s_and_b32 s5, s5, 0xffff", "  s_mov_b32 s6, 0x10000000", "  s_mov_b32 s7, 0x00020000",
             "  v_add_f32 v0, v0, v1", "  v_add_f32 v0, v0, v2", "  v_add_f32 v0, v0, v3",
             "  v_cmp_eq_f32 vcc, 0x4b189680, v0", "  s_cbranch_vccz done_%s" % name,
             "  buffer_store_dword v0, v60, s[4:7], 0 offen", "done_%s:" % name, "  s_endpgm"]
    return "\n".join(['.amdgcn_target "amdgcn-amd-amdhsa--gfx950"', ".text", ".globl %s" % name, ".p2align 8",
                      ".type %s,@function" % name, "%s:" % name] + body + [
        ".rodata", ".p2align 6", ".amdhsa_kernel %s" % name, "  .amdhsa_kernarg_size 16", "  .amdhsa_user_sgpr_count 2",
        "  .amdhsa_user_sgpr_kernarg_segment_ptr 1", "  .amdhsa_system_vgpr_workitem_id 0", "  .amdhsa_next_free_vgpr 64",
        "  .amdhsa_next_free_sgpr 32", "  .amdhsa_accum_offset 64", "  .amdhsa_ieee_mode 1", "  .amdhsa_dx10_clamp 1",
        "  .amdhsa_float_denorm_mode_32 3", ".end_amdhsa_kernel", ".amdgpu_metadata", "---", "amdhsa.version:", "  - 1",
        "  - 2", "amdhsa.kernels:", "  - .name: %s" % name, "    .symbol: %s.kd" % name, "    .kernarg_segment_size: 16",
        "    .kernarg_segment_align: 8", "    .group_segment_fixed_size: 0", "    .private_segment_fixed_size: 0",
        "    .wavefront_size: 64", "    .sgpr_count: 40", "    .vgpr_count: 64", "    .agpr_count: 0",
        "    .max_flat_workgroup_size: 64", "    .args:", "      - .offset: 0", "        .size: 8",
        "        .value_kind: by_value", "      - .offset: 8", "        .size: 8", "        .value_kind: by_value",
        "amdhsa.target: amdgcn-amd-amdhsa--gfx950", "...", ".end_amdgpu_metadata", ""])

def main():
    os.makedirs(OUT, exist_ok=True)
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    torch.cuda.init()
    out = torch.zeros((1 << 20,), dtype=torch.float32, device="cuda")
    for name, indexed, pk in (("plain", 0, 0), ("indexed", 1, 0), ("plain_pk", 0, 1), ("indexed_pk", 1, 1)):
        base = os.path.join(OUT, name)
        open(base + ".s", "w").write(source(name, indexed, pk))
        subprocess.check_call([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", base + ".s", "-o", base + ".o"])
        subprocess.check_call([LLVM + "/ld.lld", "-shared", base + ".o", "-o", base + ".hsaco"])
        mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipModuleLoad(ctypes.byref(mod), (base + ".hsaco").encode()) == 0
        assert hip.hipModuleGetFunction(ctypes.byref(fn), mod, name.encode()) == 0
        for wps in (1, 2, 4, 8):
            iters = 20000
            karg = np.array([out.data_ptr(), iters], np.uint64).tobytes()
            buf = ctypes.create_string_buffer(karg, len(karg)); size = ctypes.c_size_t(len(karg))
            extra = (ctypes.c_void_p * 5)(1, ctypes.cast(buf, ctypes.c_void_p).value, 2, ctypes.cast(ctypes.pointer(size), ctypes.c_void_p).value, 3)
            lds = (160 * 1024 // (4 * wps)) & ~255
            def go():
                assert hip.hipModuleLaunchKernel(fn, 256 * 4 * wps, 1, 1, 64, 1, 1, lds, None, None, extra) == 0
            go(); torch.cuda.synchronize()
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); go(); b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b)
            nins = iters * (16 if pk else 32)
            print("%-11s waves/SIMD %d: %.3f ms -> %.2f clk per instruction per SIMD @2.4GHz" % (name, wps, ms, ms * 1e-3 * 2.4e9 / (nins * wps)), flush=True)

if __name__ == "__main__":
    main()
