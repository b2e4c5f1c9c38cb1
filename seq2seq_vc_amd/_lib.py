"""ctypes binding of libs2svc_hip.so (the C ABI declared in include/s2svc_hip.h).

There is deliberately NO fallback: if the shared object is missing or fails to load, importing
any compute op raises.  `build_library()` (used by `__graft_entry__.build()`) cross-compiles the
HIP sources for gfx950 with hipcc; it needs no GPU.
"""
import ctypes
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# S2SVC_LIB: load another build of the SAME library (A/B timing of two kernel variants on one GPU box)
LIB_PATH = os.environ.get("S2SVC_LIB") or os.path.join(CSRC, "libs2svc_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


DEBUG_EXEC_LIB_PATH = os.path.join(CSRC, "libs2svc_hip_dbgexec.so")


def build_library(force=False, verbose=True, debug_exec=False):
    """Compile csrc/*.hip -> csrc/libs2svc_hip.so for gfx950 (in-tree, so it travels to the GPU box).
    debug_exec: the same sources with -DS2SVC_DEBUG_EXEC (common.h: the DPP / v_permlane swap reductions trap when EXEC is not all
    ones) -> csrc/libs2svc_hip_dbgexec.so, loaded through S2SVC_LIB by the GPU case `debug_exec_build_runs_clean` only."""
    lib_path = DEBUG_EXEC_LIB_PATH if debug_exec else LIB_PATH
    if not debug_exec and os.environ.get("S2SVC_LIB"):
        lib_path = os.path.join(CSRC, "libs2svc_hip.so")        # never build over an alternative library named by S2SVC_LIB
    srcs = [os.path.join(CSRC, f) for f in sources()]
    deps = srcs + [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_common.h"),
                   os.path.join(_HERE, "..", "include", "s2svc_hip.h")]
    if not force and os.path.exists(lib_path):
        newest = max(os.path.getmtime(p) for p in deps)
        if os.path.getmtime(lib_path) >= newest:
            return lib_path
    objdir = os.path.join(CSRC, "build", "dbgexec") if debug_exec else os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        hdr_time = max(os.path.getmtime(p) for p in deps[len(srcs):])
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src)
                and os.path.getmtime(obj) >= hdr_time):
            return obj
        # -fno-slp-vectorize: no compiler-formed packed-fp32 (v_pk_fma_f32 / v_pk_mul_f32 with op_sel operand swizzles).
        # With them the spline-gradient kernel of the duration predictor returned, a few times per thousand launches and
        # only while another stream kept the chip busy, a wrong element in the last partially active 16-lane row of a wave
        # (same inputs, different output); without them 0 of 800 full AAS-VC steps differ (DESIGN.md section 5 "Hazard"; profiles/AB_LOG.md).
        # The step times are unchanged (the arithmetic that matters is MFMA and explicit 16-byte memory operations).
        # -fno-vectorize: the loop vectoriser forms the same packed operations in a few element-wise kernels.
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-fno-slp-vectorize",
               "-fno-vectorize", "-c", src, "-o", obj]
        if debug_exec:
            cmd.insert(2, "-DS2SVC_DEBUG_EXEC")
        if verbose:
            print("[s2svc build]", " ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib_path] + objs
    if verbose:
        print("[s2svc build]", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return lib_path


TORCH_OPS_LIB_PATH = os.path.join(CSRC, "libs2svc_torch_ops.so")


def build_torch_ops(force=False, verbose=True):
    """csrc/torch_ops.cpp -> csrc/libs2svc_torch_ops.so: the TORCH_LIBRARY registration of the kernels (torch.ops.s2svc.*).  Host-only
    C++ against the torch headers, linked to libs2svc_hip.so (found at run time beside it: rpath $ORIGIN); g++, ~20 s."""
    import torch
    from torch.utils import cpp_extension as E
    src = os.path.join(CSRC, "torch_ops.cpp")
    deps = [src, os.path.join(_HERE, "..", "include", "s2svc_hip.h")]
    if not force and os.path.exists(TORCH_OPS_LIB_PATH) and os.path.getmtime(TORCH_OPS_LIB_PATH) >= max(os.path.getmtime(p) for p in deps):
        return TORCH_OPS_LIB_PATH
    tlib = E.library_paths()[0]
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    cmd += [f"-I{p}" for p in E.include_paths()] + ["-I/opt/rocm/include", src, "-o", TORCH_OPS_LIB_PATH,
                                                    f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip",
                                                    f"-L{CSRC}", "-ls2svc_hip", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
    if verbose:
        print("[s2svc build]", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return TORCH_OPS_LIB_PATH


c_i32, c_i64, c_f32, c_u64, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_uint64, ctypes.c_void_p


class Operand(ctypes.Structure):
    _fields_ = [("ptr", c_vp), ("ld", c_i64), ("layout", c_i32), ("mode", c_i32), ("C", c_i32), ("T", c_i32),
                ("pad", c_i32), ("T1", c_i32), ("F1", c_i32), ("T2", c_i32), ("F2", c_i32), ("bs0", c_i64),
                ("bs1", c_i64), ("zero_padded", c_i32), ("reserved_", c_i32)]


class GemmDesc(ctypes.Structure):
    _fields_ = [("A", Operand), ("B", Operand), ("C", c_vp), ("ldc", c_i64), ("cbs0", c_i64), ("cbs1", c_i64),
                ("c_dtype", c_i32), ("bias", c_vp), ("res", c_vp), ("ldr", c_i64), ("rbs0", c_i64), ("rbs1", c_i64),
                ("M", c_i32), ("N", c_i32), ("K", c_i32), ("nb0", c_i32), ("nb1", c_i32), ("act", c_i32),
                ("alpha", c_f32), ("dtype", c_i32), ("accumulate", c_i32), ("splitk", c_i32), ("ws", c_vp),
                ("a_rowsum", c_vp), ("a_rowsum_ws", c_vp), ("a_rowsum_accumulate", c_i32), ("tile_hint", c_i32),
                ("emask", c_vp), ("ldm", c_i64), ("drop_p", c_f32), ("emask_mode", c_i32), ("seed_base", c_vp),
                ("seed_off", c_u64), ("c_map", c_i32), ("cm_T1", c_i32), ("cm_F1", c_i32), ("cm_Tc", c_i32), ("cm_Fc", c_i32),
                ("cm_pt", c_i32), ("cm_pf", c_i32), ("reserved3_", c_i32), ("c_pre", c_vp)]


class ScalarTerms(ctypes.Structure):
    _fields_ = [("x", c_vp * 8), ("n", c_i32 * 8), ("w", c_f32 * 8), ("k", c_i32), ("reserved_", c_i32)]


class Gather3Job(ctypes.Structure):
    _fields_ = [("in_", c_vp), ("out", c_vp), ("s0", c_i64), ("s1", c_i64), ("s2", c_i64), ("off", c_i64), ("n0", c_i32),
                ("n1", c_i32), ("n2", c_i32), ("out_dtype", c_i32)]


class ColreduceItem(ctypes.Structure):
    _fields_ = [("dy", c_vp), ("x", c_vp), ("mean", c_vp), ("rstd", c_vp), ("out_sum", c_vp), ("out_dot", c_vp), ("ws", c_vp),
                ("dtype", c_i32), ("rows", c_i32), ("D", c_i32), ("mode", c_i32), ("accumulate", c_i32), ("ws_chunks", c_i32),
                ("scale", c_f32), ("reserved_", c_i32)]


_SIGS = {
    "s2svc_gemm": [ctypes.POINTER(GemmDesc), c_vp],
    "s2svc_gemm_grouped_ok": [c_vp],
    "s2svc_tconv2d_weights": [c_i32, c_i32, c_vp, c_vp, c_vp],
    "s2svc_gemm_grouped": [c_vp, c_i32, c_i32, c_vp],
    "s2svc_gemm_grouped_batched": [c_vp, c_i32, c_vp],
    "s2svc_gemm_wgrad_ok": [c_vp],
    "s2svc_gemm_wgrad_grouped": [c_vp, c_i32, c_vp, c_vp],
    "s2svc_gemm_wgrad_grouped_bg": [c_vp, c_i32, c_vp, c_vp, c_i32],
    "s2svc_gemm_set_w8": [c_i32, c_i32],
    "s2svc_gemm_set_8ph": [c_i32],
    "s2svc_launch_floor": [c_i32, c_i32, c_vp, c_vp],
    "s2svc_event_create": [c_vp],
    "s2svc_event_destroy": [c_vp],
    "s2svc_event_record": [c_vp, c_vp],
    "s2svc_stream_wait_event": [c_vp, c_vp],
    "s2svc_decode_ln_linear_supported": [c_i32, c_i32, c_i32],
    "s2svc_decode_ln_linear": [ctypes.POINTER(GemmDesc), c_vp, c_vp, c_f32, c_vp, c_i64, c_vp],
    "s2svc_length_regulate_index": [c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_length_regulate_fwd": [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_f32, c_vp, c_vp],
    "s2svc_length_regulate_bwd": [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_attn_durations": [c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_layernorm_fwd": [c_i32, c_i32, c_i32, c_vp, c_vp, c_f32, c_f32, c_vp, c_u64, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_layernorm_bwd": [c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_vp, c_u64, c_vp, c_vp, c_vp],
    "s2svc_layernorm_bwd_pg_chunks": [c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_layernorm_bwd_pg": [c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_vp, c_u64, c_vp, c_vp, c_vp, c_vp],
    "s2svc_colreduce_grouped": [c_vp, c_i32, c_vp],
    "s2svc_colreduce": [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp],
    "s2svc_bn_finalize": [c_i32, c_i32, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp],
    "s2svc_rstd_from_var": [c_i32, c_f32, c_vp, c_vp, c_vp],
    "s2svc_bn_apply": [c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_f32, c_vp, c_u64, c_vp, c_vp, c_i32, c_vp, c_vp],
    "s2svc_bn_bwd": [c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp],
    "s2svc_bn_stats": [c_i32, c_i32, c_i32, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp],
    "s2svc_attn_softmax_fwd": [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_vp, c_i32, c_f32,
                               c_vp, c_u64, c_vp, c_vp, c_vp],
    "s2svc_attn_softmax_bwd": [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_f32, c_f32, c_vp, c_u64, c_vp, c_vp, c_i32,
                               c_i32, c_i32, c_vp],
    "s2svc_act_dropout_fwd": [c_i32, c_i64, c_vp, c_i32, c_f32, c_vp, c_u64, c_vp, c_vp],
    "s2svc_act_dropout_bwd": [c_i32, c_i64, c_vp, c_vp, c_i32, c_f32, c_vp, c_u64, c_vp, c_vp],
    "s2svc_posenc_fwd": [c_i32, c_i64, c_i32, c_i32, c_vp, c_f32, c_vp, c_vp, c_f32, c_vp, c_u64, c_vp, c_vp],
    "s2svc_posenc_bwd": [c_i32, c_i64, c_i32, c_i32, c_vp, c_f32, c_vp, c_f32, c_vp, c_u64, c_vp, c_vp, c_vp, c_vp],
    "s2svc_axpby": [c_i32, c_i64, c_f32, c_vp, c_f32, c_vp, c_vp, c_vp],
    "s2svc_add_n": [c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_weighted_sum": [ctypes.POINTER(ScalarTerms), c_vp, c_vp],
    "s2svc_weighted_sum_bwd": [ctypes.POINTER(ScalarTerms), c_vp, c_vp],
    "s2svc_scalars_axpy": [ctypes.POINTER(ScalarTerms), c_f32, c_vp, c_vp],
    "s2svc_fill_zero": [c_vp, c_i64, c_vp],
    "s2svc_seed_advance": [c_vp, c_u64, c_vp],
    "s2svc_pad_cols": [c_i32, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp],
    "s2svc_decoder_input": [c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp, c_vp, c_vp],
    "s2svc_stop_labels": [c_i32, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp],
    "s2svc_append_eos": [c_i32, c_i32, c_i64, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp],
    "s2svc_copy_rows": [c_i32, c_i64, c_i32, c_i64, c_vp, c_vp, c_vp],
    "s2svc_add_head_bias": [c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_add_head_bias_ld": [c_i32, c_i64, c_i32, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_add_rows": [c_i32, c_i64, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp],
    "s2svc_glu_fwd": [c_i32, c_i64, c_i32, c_vp, c_vp, c_vp],
    "s2svc_glu_bwd": [c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp],
    "s2svc_cast": [c_i32, c_i32, c_i64, c_vp, c_vp, c_vp],
    "s2svc_gather3_grouped": [c_vp, c_i32, c_vp],
    "s2svc_permute_inner": [c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp],
    "s2svc_gather3": [c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp],
    "s2svc_mas": [c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_mas_binloss_bwd": [c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_seq_loss_fwd": [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp],
    "s2svc_seq_loss_bwd": [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp,
                           c_vp, c_vp, c_vp],
    "s2svc_guided_attn_loss_fwd": [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp],
    "s2svc_guided_attn_loss_bwd": [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp],
    "s2svc_adam_step": [c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_f32, c_f32, c_f32, c_vp, c_vp, c_vp],
    "s2svc_transpose_tiles": [c_i64, c_vp, c_vp, c_vp, c_vp],
    "s2svc_col2im_s2": [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp],
    "s2svc_interp_nearest": [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_interp_nearest_bwd": [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_dwconv": [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp],
    "s2svc_dwconv_add": [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp],
    "s2svc_dwconv_wgrad": [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp],
    "s2svc_convmod_supported": [c_i32, c_i32],
    "s2svc_convmod_fwd": [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_bn_swish_apply": [c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp],
    "s2svc_convmod_bwd": [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                          c_vp, c_vp, c_vp],
    "s2svc_convmod_wgrad_final": [c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp],
    "s2svc_bn_stats_vec": [c_i32, c_i32, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp],
    "s2svc_bn_act_apply_vec": [c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_f32, c_vp, c_u64, c_vp, c_vp, c_i32, c_vp, c_vp],
    "s2svc_bn_act_bwd_vec": [c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_f32, c_vp, c_u64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                             c_i32, c_vp, c_vp],
    "s2svc_pairwise_l2_logsoftmax": [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_pairwise_l2_bwd_g": [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_rowscale": [c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp],
    "s2svc_gauss_upsample_probs": [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp],
    "s2svc_forward_sum": [c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp],
    "s2svc_betabinom_prior": [c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp],
    "s2svc_conv_in1_fwd": [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_conv_in1_wgrad": [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp],
    "s2svc_attn_fused_supported": [c_i32, c_i32, c_i32, c_i32],
    "s2svc_attn_fused_fwd": [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i32,
                             c_f32, c_f32, c_vp, c_u64, c_vp, c_i32, c_vp, c_i64, c_i64, c_vp],
    "s2svc_attn_fused_bwd": [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64,
                             c_i64, c_vp, c_vp, c_i32, c_f32, c_f32, c_vp, c_u64, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64,
                             c_i64, c_vp],
    "s2svc_relattn_supported": [c_i32, c_i32, c_i32, c_i32],
    "s2svc_relattn_fwd": [c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_f32,
                          c_f32, c_vp, c_u64, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp],
    "s2svc_attn_map_supported": [c_i32, c_i32, c_i32, c_i32],
    "s2svc_attn_map_product_supported": [c_i32],
    "s2svc_attn_map_fwd": [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i32, c_f32, c_f32, c_vp, c_u64,
                           c_vp, c_vp, c_i32, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp],
    "s2svc_attn_map_bwd": [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp, c_f32, c_f32, c_vp, c_u64,
                           c_vp, c_i32, c_vp, c_i32, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp],
    "s2svc_duration_loss_fwd": [c_i32, c_i32, c_vp, c_vp, c_vp, c_f32, c_i32, c_vp, c_vp, c_vp],
    "s2svc_duration_loss_bwd": [c_i32, c_i32, c_vp, c_vp, c_vp, c_f32, c_i32, c_vp, c_vp, c_vp, c_vp],
    "s2svc_embedding_fwd": [c_i32, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp],
    "s2svc_embedding_bwd": [c_i32, c_i64, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp, c_i32, c_vp],
    "s2svc_mask_rows": [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp],
    "s2svc_expand_fwd": [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_expand_bwd": [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_ln_act_fwd": [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_f32, c_i32, c_vp, c_vp, c_f32, c_vp, c_u64, c_vp, c_vp,
                         c_vp, c_vp],
    "s2svc_dw_ln_act_fwd": [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_ln_act_bwd": [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_f32, c_vp, c_u64, c_vp,
                         c_vp, c_vp, c_vp],
    "s2svc_rq_spline_fwd": [c_i32, c_i32, c_i32, c_vp, c_vp, c_f32, c_f32, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp],
    "s2svc_rq_spline_bwd": [c_i32, c_i32, c_i32, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_sdp_head_fwd": [c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_sdp_head_bwd": [c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_sdp_mid_fwd": [c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_sdp_mid_bwd": [c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_sdp_tail_fwd": [c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp],
    "s2svc_sdp_tail_bwd": [c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp],
    "s2svc_sdp_inverse_out": [c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_decode_posenc": [c_i32, c_i32, c_i32, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp],
    "s2svc_decode_attn": [c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_i32,
                          c_f32, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp],
    "s2svc_decode_emit": [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp,
                          c_vp],
    "s2svc_decode_advance": [c_vp, c_vp, c_u64, c_vp],
    "s2svc_decode_emit_advance": [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i64, c_f32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp,
                                  c_vp, c_vp, c_u64, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp],
    "s2svc_reflect_pad": [c_i64, c_i32, c_vp, c_vp, c_vp],
    "s2svc_magnitude": [c_i64, c_i32, c_vp, c_vp, c_vp],
    "s2svc_log_clamp": [c_i64, c_i32, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp],
    "s2svc_reflect_pad_batch": [c_i32, c_i64, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp],
    "s2svc_mel_log_batch": [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp],
    "s2svc_stft_logmel_fft_supported": [c_i32, c_i32, c_i32],
    "s2svc_stft_logmel_fft8": [c_i32, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp,
                               c_vp],
    "s2svc_stft_logmel_fft": [c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32,
                              c_i32, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp],
    "s2svc_ragged_to_padded": [c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
}
_RET64 = {"s2svc_gemm_wgrad_ws_floats": [c_vp, c_i32], "s2svc_mas_ws_bytes": [c_i32, c_i32, c_i32], "s2svc_forward_sum_ws_bytes": [c_i32, c_i32, c_i32]}

_lib = None


def exported_symbols():
    """Every symbol include/s2svc_hip.h declares (checked by the CPU-side ABI test)."""
    return sorted(list(_SIGS) + list(_RET64) + ["s2svc_last_error", "s2svc_abi_version"])


def lib():
    """Load the shared object (once).  Raises if it is absent -- there is no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must be imported FIRST: it carries its own libamdhip64; loading ours afterwards makes the
    # dynamic linker bind this library to the SAME HIP runtime instance (one device context, shared streams).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the seq2seq-vc HIP kernels are not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc, no GPU). "
            "There is no CPU fallback for the product path.")
    L = ctypes.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = ctypes.c_int
    for name, args in _RET64.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = ctypes.c_int64
    L.s2svc_last_error.restype = ctypes.c_char_p
    L.s2svc_abi_version.restype = ctypes.c_int
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        msg = lib().s2svc_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")
