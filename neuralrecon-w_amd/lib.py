"""ctypes binding of libneuconw_hip.so (C ABI declared in include/neuconw_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent this
module raises -- a GPU box must run the hand-written HIP kernels or nothing.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# NEUCONW_HIP_LIB: another build of the SAME library (experiment variants from build.py NCW_BUILD_TAG)
LIB_PATH = os.environ.get("NEUCONW_HIP_LIB") or os.path.join(HERE, "libneuconw_hip.so")

PREC_F32 = 0
PREC_BF16 = 1
PREC_F16 = 2  # fp16 operands, f32 accumulate; backward needs the loss scale (renderer.grad_scale)
MAX_LAYERS = 12
MAX_SEGS = 4
ABI_VERSION = 18  # 18: NcwSdfStash.s = residuals of h (adj_mode 2), NcwSdfNet.adj_mode, NcwColorNet.act_split;  17: ncw_ray_voxel_trace (all ray / voxel intersections: kaolin's unbatched_raytrace contract);  16: NcwSdfNet.wt_lo (adjoint sweep with hi + lo weights), NcwNerfNet.w_*_lo;  15: forward-only render form of ncw_sdf_fwd / ncw_color_fwd / ncw_nerf_fwd (NULL stash members);  14: NcwColorNet.w_*_lo (split colour weights, forward);  13: NcwNerfStash.aux_bias;  12: ncw_aux_ray_bias, NcwColorStash.aux_bias, ncw_source_hash;  11: marching cubes (ncw_mc_count / ncw_mc_emit replace the marching-tetrahedra entry points);  10: split-precision SDF value path (NcwSdfNet.w_lo, NcwPackDesc.residual);  9: device-resident optimiser state (ncw_adam_step_dev, NcwAdamState), dynamic loss scale (grad_scale_dev / grad_mul_dev);  8: NcwPoints mode 4 (idx / count), NcwWgradDesc.n_points_dev, ncw_bg_select;  7: fp16 (prec 2), grad_scale / grad_mul;  6: ray prologue / inv_s / loss launches, NcwCompositeOut.weights_max;  5: ncw_scatter_add_rows;  4: ncw_batch_assemble;  3: ordered fp32 wgrad, d_a_rows / ncw_ray_sum_rows, per-ray d_inv_s;  2: 2: NcwWgradDesc.ksplit/n_points, NcwCompositeIn.cos_anneal_dev, ray tail / mesh / optimiser entry points


class NcwSeg(C.Structure):
    _fields_ = [("col0", C.c_int32), ("ncols", C.c_int32), ("dcol0", C.c_int32), ("_pad", C.c_int32)]


class NcwPackDesc(C.Structure):
    _fields_ = [
        ("src", C.c_void_p), ("g", C.c_void_p), ("bias", C.c_void_p), ("dst_w", C.c_void_p), ("dst_b", C.c_void_p),
        ("ld", C.c_int32), ("row0", C.c_int32), ("nrows", C.c_int32), ("drow0", C.c_int32),
        ("rb_out", C.c_int32), ("rb_in", C.c_int32), ("transpose", C.c_int32), ("prec", C.c_int32),
        ("scale", C.c_float), ("nseg", C.c_int32), ("seg", NcwSeg * MAX_SEGS), ("residual", C.c_int32), ("_pad", C.c_int32),
    ]


class NcwUnpackDesc(C.Structure):
    _fields_ = [
        ("dw", C.c_void_p), ("db", C.c_void_p), ("src", C.c_void_p), ("g", C.c_void_p),
        ("d_src", C.c_void_p), ("d_g", C.c_void_p), ("d_bias", C.c_void_p),
        ("ld", C.c_int32), ("ldw", C.c_int32), ("row0", C.c_int32), ("nrows", C.c_int32), ("drow0", C.c_int32),
        ("scale", C.c_float), ("grad_mul", C.c_float), ("accumulate", C.c_int32), ("nseg", C.c_int32),
        ("seg", NcwSeg * MAX_SEGS), ("grad_mul_dev", C.c_void_p),
    ]


class NcwAdamState(C.Structure):
    _fields_ = [("step", C.c_int32), ("good", C.c_int32), ("skipped", C.c_int32), ("skip_now", C.c_int32),
                ("coef", C.c_float), ("step_size", C.c_float), ("bc2_sqrt", C.c_float), ("last_norm", C.c_float)]


class NcwSdfNet(C.Structure):
    _fields_ = [
        ("w", C.c_void_p * MAX_LAYERS), ("b", C.c_void_p * MAX_LAYERS), ("wt", C.c_void_p * MAX_LAYERS),
        ("w_feat", C.c_void_p), ("b_feat", C.c_void_p), ("wt_feat", C.c_void_p),
        ("n_layers", C.c_int32), ("skip_layer", C.c_int32), ("rb", C.c_int32), ("multires", C.c_int32),
        ("scale", C.c_float), ("adj_mode", C.c_int32), ("w_lo", C.c_void_p * MAX_LAYERS),
        ("wt_lo", C.c_void_p * MAX_LAYERS),
    ]


class NcwPoints(C.Structure):
    _fields_ = [("x", C.c_void_p), ("rays_o", C.c_void_p), ("rays_d", C.c_void_p), ("z", C.c_void_p),
                ("sample_dist", C.c_void_p), ("per_ray", C.c_int32), ("mode", C.c_int32),
                ("gmin", C.c_float * 3), ("gmax", C.c_float * 3), ("gorigin", C.c_float * 3), ("gradius", C.c_float),
                ("gdim", C.c_int32), ("_gpad", C.c_int32), ("gstart", C.c_int64),
                ("idx", C.c_void_p), ("count", C.c_void_p)]


class NcwSdfStash(C.Structure):
    _fields_ = [("gamma", C.c_void_p), ("h", C.c_void_p * MAX_LAYERS), ("s", C.c_void_p * MAX_LAYERS),
                ("t", C.c_void_p * MAX_LAYERS), ("feat", C.c_void_p), ("dfeat", C.c_void_p),
                ("qbar", C.c_void_p * MAX_LAYERS), ("zbar", C.c_void_p * MAX_LAYERS), ("zsdf", C.c_void_p),
                ("one", C.c_void_p)]


class NcwWgradDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("dense", C.c_void_p), ("dbias", C.c_void_p),
                ("rbx", C.c_int32), ("rby", C.c_int32), ("ld", C.c_int32), ("ksplit", C.c_int32),
                ("n_points", C.c_int64), ("n_points_dev", C.c_void_p)]


class NcwColorNet(C.Structure):
    _fields_ = [("w_f", C.c_void_p), ("wt_f", C.c_void_p), ("b_f", C.c_void_p),
                ("w_e", C.c_void_p * 4), ("wt_e", C.c_void_p * 4), ("b_e", C.c_void_p * 4),
                ("w_l", C.c_void_p * 8), ("wt_l", C.c_void_p * 8), ("b_l", C.c_void_p * 8),
                ("n_head", C.c_int32), ("n_lin", C.c_int32), ("rbf", C.c_int32), ("rbh", C.c_int32),
                ("rbc", C.c_int32), ("n_a", C.c_int32),
                ("w_f_lo", C.c_void_p), ("w_e_lo", C.c_void_p * 4), ("w_l_lo", C.c_void_p * 8),
                ("act_split", C.c_int32), ("_pad", C.c_int32)]


class NcwColorStash(C.Structure):
    _fields_ = [("aux1", C.c_void_p), ("aux2", C.c_void_p), ("f", C.c_void_p), ("e", C.c_void_p * 4),
                ("x", C.c_void_p * 8), ("zf", C.c_void_p), ("ze", C.c_void_p * 4), ("zx", C.c_void_p * 8),
                ("zo", C.c_void_p), ("aux_bias", C.c_void_p)]


class NcwNerfNet(C.Structure):
    _fields_ = [("w_p", C.c_void_p * 8), ("wt_p", C.c_void_p * 8), ("b_p", C.c_void_p * 8),
                ("w_alpha", C.c_void_p), ("wt_alpha", C.c_void_p), ("b_alpha", C.c_void_p),
                ("w_feat", C.c_void_p), ("wt_feat", C.c_void_p), ("b_feat", C.c_void_p),
                ("w_a", C.c_void_p * 4), ("wt_a", C.c_void_p * 4), ("b_a", C.c_void_p * 4),
                ("w_rgb", C.c_void_p), ("wt_rgb", C.c_void_p), ("b_rgb", C.c_void_p),
                ("D", C.c_int32), ("skip", C.c_int32), ("rbn", C.c_int32), ("rbh", C.c_int32),
                ("n_head", C.c_int32), ("n_a", C.c_int32),
                ("w_p_lo", C.c_void_p * 8), ("w_alpha_lo", C.c_void_p), ("w_feat_lo", C.c_void_p), ("w_a_lo", C.c_void_p * 4),
                ("w_rgb_lo", C.c_void_p)]


class NcwNerfStash(C.Structure):
    _fields_ = [("gp", C.c_void_p), ("aux1", C.c_void_p), ("h", C.c_void_p * 9), ("featn", C.c_void_p),
                ("e", C.c_void_p * 4), ("zp", C.c_void_p * 8), ("zalpha", C.c_void_p), ("zfeat", C.c_void_p),
                ("ze", C.c_void_p * 4), ("zrgb", C.c_void_p), ("aux_bias", C.c_void_p)]


def _ptr_struct(name, fields_ptr, fields_other=()):
    return type(name, (C.Structure,), {"_fields_": [(f, C.c_void_p) for f in fields_ptr] + list(fields_other)})


NcwCompositeIn = _ptr_struct(
    "NcwCompositeIn",
    ["rays_o", "rays_d", "z", "z_feed", "sample_dist", "sdf", "grad", "rgb", "density", "bg_rgb", "inv_s",
     "background_rgb", "cos_anneal_dev"],
    [("cos_anneal", C.c_float), ("R", C.c_int32), ("S", C.c_int32), ("O", C.c_int32), ("has_bg", C.c_int32),
     ("trim_sphere", C.c_int32)],
)
NcwCompositeOut = _ptr_struct(
    "NcwCompositeOut",
    ["color", "color_sphere", "color_bg", "weights", "weights_sum", "cdf", "inside", "depth", "normals", "eik",
     "mid_z", "dists", "bg_alpha", "weights_max"],
)
NcwCompositeGrad = _ptr_struct(
    "NcwCompositeGrad",
    ["d_color", "d_weights_sum", "d_depth", "d_eik_num", "d_sdf", "d_grad", "d_rgb", "d_density", "d_bg_rgb",
     "d_inv_s"],
    [("grad_scale", C.c_float), ("grad_scale_dev", C.c_void_p)],
)

_VP = C.c_void_p
_PROTOS = {
    "ncw_mc_count": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ncw_mc_emit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ncw_ray_tail_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ncw_ray_tail_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "ncw_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_float, C.c_void_p]),
    "ncw_adam_step_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                    C.c_float, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int,
                                    C.c_float, C.c_float, C.c_void_p]),
    "ncw_wgrad_tiled": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "ncw_wgrad_tiled_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "ncw_sdf_infer_points": (C.c_int, [C.POINTER(NcwSdfNet), C.c_int, C.POINTER(NcwPoints), C.c_int64, C.c_void_p,
                                       C.c_void_p]),
    "ncw_voxel_build": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ncw_ray_voxel_near_far": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_float, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ncw_ray_voxel_trace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ncw_color_fwd": (C.c_int, [C.POINTER(NcwColorNet), C.c_int, C.POINTER(NcwPoints), C.c_int64, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(NcwColorStash), C.c_void_p]),
    "ncw_color_bwd": (C.c_int, [C.POINTER(NcwColorNet), C.c_int, C.POINTER(NcwPoints), C.c_int64, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(NcwColorStash),
                                C.c_void_p]),
    "ncw_batch_assemble": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ncw_ray_prologue": (C.c_int, [_VP, C.c_int, C.c_int64, C.POINTER(C.c_float), C.c_float, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "ncw_aux_ray_bias": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, _VP, _VP, C.c_int, C.c_int64, _VP, C.c_int, _VP]),
    "ncw_inv_s_fwd": (C.c_int, [_VP, _VP, _VP, _VP]),
    "ncw_inv_s_bwd": (C.c_int, [_VP, C.c_int64, _VP, _VP, _VP]),
    "ncw_loss_fwd": (C.c_int, [_VP, _VP, C.c_int64, _VP, _VP, C.c_int64, _VP, C.c_int64, C.c_float, C.c_float, C.c_float,
                               C.c_float, _VP, _VP]),
    "ncw_loss_bwd": (C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                               _VP, _VP, _VP, _VP, _VP]),
    "ncw_scatter_add_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "ncw_ray_sum_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "ncw_nerf_fwd": (C.c_int, [C.POINTER(NcwNerfNet), C.c_int, C.POINTER(NcwPoints), C.c_void_p, C.c_int64,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(NcwNerfStash), C.c_void_p]),
    "ncw_nerf_bwd": (C.c_int, [C.POINTER(NcwNerfNet), C.c_int, C.POINTER(NcwPoints), C.c_int64, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(NcwNerfStash), C.c_void_p]),
    "ncw_nerf_refine": (C.c_int, [C.POINTER(NcwNerfNet), C.c_int, C.POINTER(NcwPoints), C.c_int64, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "ncw_sdf_fwd": (C.c_int, [C.POINTER(NcwSdfNet), C.c_int, C.POINTER(NcwPoints), C.c_int64, C.c_void_p, C.c_void_p,
                              C.POINTER(NcwSdfStash), C.c_void_p]),
    "ncw_sdf_bwd": (C.c_int, [C.POINTER(NcwSdfNet), C.c_int, C.POINTER(NcwPoints), C.c_int64, C.c_void_p, C.c_void_p,
                              C.POINTER(NcwSdfStash), C.c_void_p]),
    "ncw_wgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "ncw_wgrad_ordered": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p,
                                    C.c_void_p]),
    "ncw_wgrad_ordered_scratch_floats": (C.c_int64, [C.c_int]),
    "ncw_stash_from_rows": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ncw_stash_to_rows": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ncw_sdf_infer_rays": (C.c_int, [C.POINTER(NcwSdfNet), C.c_int, _VP, _VP, _VP, C.c_int, C.c_int, _VP, _VP]),
    "ncw_sample_coarse": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int, C.c_int, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP]),
    "ncw_upsample": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int, C.c_int, C.c_float, C.c_int, _VP, _VP]),
    "ncw_sort_merge": (C.c_int, [_VP, C.c_int, _VP, C.c_int, _VP, _VP, C.c_int, _VP, _VP, _VP]),
    "ncw_boundary": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, C.c_int, _VP, _VP]),
    "ncw_bg_select": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int, C.c_int, C.c_int, _VP, _VP, _VP, _VP]),
    "ncw_composite_fwd": (C.c_int, [C.POINTER(NcwCompositeIn), C.POINTER(NcwCompositeOut), _VP]),
    "ncw_composite_bwd": (C.c_int, [C.POINTER(NcwCompositeIn), C.POINTER(NcwCompositeGrad), _VP]),
    "ncw_abi_version": (C.c_int, []),
    "ncw_source_hash": (C.c_char_p, []),
    "ncw_device_info": (C.c_int, [C.c_char_p, C.c_int]),
    "ncw_grad_norm_scratch_floats": (C.c_int64, []),
    "ncw_grad_norm": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ncw_pack_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ncw_unpack_grads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ncw_sdf_infer": (C.c_int, [C.POINTER(NcwSdfNet), C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
}

_lib = None

# Optional per-entry-point HIP-event timing (bench.py's roofline leg): when PROFILE is a dict, every
# C-ABI call is bracketed by events recorded on torch's current stream (the stream the kernel is
# launched on) and appended to PROFILE[name].
PROFILE = None


def _wrap_timed(name, fn):
    def call(*args):
        if PROFILE is None:
            return fn(*args)
        import torch

        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*args)
        e1.record()
        PROFILE.setdefault(name, []).append((e0, e1))
        return r

    return call



class NeuconwHipError(RuntimeError):
    pass


def exported_symbols():
    """Names every entry point include/neuconw_hip.h declares (used by the CPU-side ABI test)."""
    return sorted(_PROTOS)


def get_lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise NeuconwHipError(
            "libneuconw_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'`. "
            "There is no CPU / PyTorch fallback for the hot path." % LIB_PATH
        )
    lib = C.CDLL(LIB_PATH)
    ns = type("NcwLib", (), {})()
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale: fail loudly
        fn.restype = res
        fn.argtypes = args
        setattr(ns, name, _wrap_timed(name, fn))
    v = lib.ncw_abi_version()
    if v != ABI_VERSION:
        raise NeuconwHipError("libneuconw_hip.so ABI %d != binding ABI %d: rebuild" % (v, ABI_VERSION))
    # the library must have been built from the sources in this tree (build.source_hash): a stale prebuilt .so -- e.g. one
    # shipped to a GPU box after csrc/ changed -- is refused, not silently tested.  Probe variants (NEUCONW_HIP_LIB) are exempt.
    if not os.environ.get("NEUCONW_HIP_LIB") and os.path.isdir(os.path.join(HERE, "csrc")):
        from . import build as _b

        built, now = lib.ncw_source_hash().decode(), _b.source_hash()
        if built != now:
            raise NeuconwHipError("libneuconw_hip.so was built from other sources (%s.. != %s..): rebuild with "
                                  "`python -c 'import __graft_entry__ as g; g.build()'`" % (built[:12], now[:12]))
    ns._cdll = lib
    _lib = ns
    return ns


def check(code, what):
    if code != 0:
        raise NeuconwHipError("%s failed with code %d" % (what, code))


def stream_ptr(device=None):
    import torch

    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    """Device pointer of a contiguous torch tensor (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    assert t.is_contiguous(), "non-contiguous tensor handed to the C ABI"
    return C.c_void_p(t.data_ptr())
