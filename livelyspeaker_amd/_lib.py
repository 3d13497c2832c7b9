"""ctypes binding of ``include/ls_hip.h`` (the C-ABI of the gfx950 engine).

There is deliberately NO fallback: if ``libls_hip.so`` is missing or fails to load, importing the
engine raises; nothing in the product path ever routes through the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_i64p = C.POINTER(C.c_int64)
c_i32p = C.POINTER(C.c_int32)

LS_SAMPLER_DDPM, LS_SAMPLER_DDIM = 0, 1
LS_NOISE_TAPE, LS_NOISE_PHILOX = 0, 1
LS_PRECISION_FP32, LS_PRECISION_BF16X3 = 0, 1


class LsConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("njoints", "nfeats", "nframes", "n_prefix_tokens", "n_pre_seq",
                                         "latent_dim", "layers", "audio_len", "n_speakers", "n_emotions",
                                         "device", "reserved")]


class LsSchedule(C.Structure):
    _fields_ = [("n_steps", C.c_int32), ("reserved", C.c_int32), ("timestep_map", c_i64p)] + [
        (n, c_f64p) for n in ("sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "posterior_mean_coef1",
                              "posterior_mean_coef2", "posterior_log_variance_clipped", "alphas_cumprod",
                              "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod")]


class LsCond(C.Structure):
    _fields_ = [("batch", C.c_int32), ("on_device", C.c_int32), ("audio_input", C.c_void_p),
                ("origin_x", C.c_void_p), ("vid_indices", C.c_void_p), ("emo", C.c_void_p), ("scale", C.c_void_p)]


class LsSampleArgs(C.Structure):
    _fields_ = [("sampler", C.c_int32), ("noise_mode", C.c_int32), ("skip_timesteps", C.c_int32),
                ("const_noise", C.c_int32), ("on_device", C.c_int32), ("use_graph", C.c_int32),
                ("clip_denoised", C.c_int32), ("two_pass_always", C.c_int32), ("eta", C.c_float), ("n_dump", C.c_int32), ("dump_steps", c_i32p), ("dump_out", C.c_void_p),
                ("x_init", C.c_void_p), ("init_image", C.c_void_p), ("eps_tape", C.c_void_p),
                ("noise_tape", C.c_void_p), ("seed", C.c_uint64), ("sample_offset", C.c_uint64),
                ("out", C.c_void_p), ("seg_begin", C.c_int32), ("seg_count", C.c_int32), ("inpaint_mask", C.c_void_p),
                ("inpainted_motion", C.c_void_p), ("inpaint_noise", C.c_void_p), ("inpaint_noised", C.c_int32), ("reserved2", C.c_int32)]


class LsForwardArgs(C.Structure):
    _fields_ = [("on_device", C.c_int32), ("no_sync", C.c_int32), ("x", C.c_void_p), ("timesteps", C.c_void_p),
                ("eps_cond", C.c_void_p), ("eps_uncond", C.c_void_p), ("out_cond", C.c_void_p),
                ("out_uncond", C.c_void_p), ("out_cfg", C.c_void_p), ("trace", C.c_void_p)]


class LsStepArgs(C.Structure):
    _fields_ = [("sampler", C.c_int32), ("index", C.c_int32), ("on_device", C.c_int32), ("eta", C.c_float),
                ("clip_denoised", C.c_int32), ("two_pass_always", C.c_int32), ("x", C.c_void_p), ("eps_cond", C.c_void_p), ("eps_uncond", C.c_void_p), ("noise", C.c_void_p),
                ("sample", C.c_void_p), ("pred_xstart", C.c_void_p), ("indices", C.c_void_p), ("no_sync", C.c_int32),
                ("indices_on_device", C.c_int32), ("inpaint_mask", C.c_void_p), ("inpainted_motion", C.c_void_p), ("inpaint_noise", C.c_void_p)]


class LsSagConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("njoints", "nfeats", "nframes", "latent_dim", "ff_size", "num_layers",
                                         "num_heads", "n_pre_poses", "device", "reserved")]


class LsPostConfig(C.Structure):
    _fields_ = [("njoints", C.c_int32), ("n_pairs", C.c_int32), ("n_pose_joints", C.c_int32), ("thres", C.c_float),
                ("pair_a", C.c_int32 * 8), ("pair_b", C.c_int32 * 8), ("change_angle", C.c_float * 8),
                ("bone_parent", C.c_int32 * 16), ("bone_child", C.c_int32 * 16), ("bone_len", C.c_float * 16),
                ("mean_dir_vec", C.c_float * 48)]


class LsTiming(C.Structure):
    _fields_ = [("prepare_ms", C.c_float), ("loop_ms", C.c_float), ("total_ms", C.c_float),
                ("n_step_launches", C.c_int32), ("graph_replayed", C.c_int32), ("single_pass", C.c_int32),
                ("tape_upload_ms", C.c_float), ("n_segments", C.c_int32), ("step_path", C.c_int32),
                ("tail_samples", C.c_int32), ("tail_path", C.c_int32), ("tail2_samples", C.c_int32), ("tail2_path", C.c_int32), ("n_cus", C.c_int32), ("coop_slices", C.c_int32)]


class LsTrainConfig(C.Structure):
    _fields_ = [("model", LsConfig), ("lambda_vel", C.c_float), ("kld_weight", C.c_float), ("diffusion_steps", C.c_int32),
                ("reserved", C.c_int32)]


class LsTrainBatch(C.Structure):
    _fields_ = [("batch", C.c_int32), ("on_device", C.c_int32), ("x_start", C.c_void_p), ("t", C.c_void_p), ("noise", C.c_void_p),
                ("drop", C.c_void_p), ("eps", C.c_void_p), ("audio_input", C.c_void_p), ("origin_x", C.c_void_p),
                ("vid_indices", C.c_void_p), ("emo", C.c_void_p)]


class LsTrainTerms(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("rot_mse", "vel_mse", "kld", "loss", "total", "fwd_ms", "bwd_ms", "reserved")]


class LsEvalConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("pose_dim", "n_frames", "base", "hidden1", "hidden2", "device")]


EXPORTS = ("ls_abi_version", "ls_create", "ls_destroy", "ls_last_error", "ls_set_weight", "ls_commit_weights",
           "ls_set_schedule", "ls_prepare", "ls_prepare_async", "ls_sample", "ls_forward", "ls_step", "ls_q_sample", "ls_read",
           "ls_get_timing", "ls_synchronize", "ls_stream_order", "ls_stream", "ls_sag_stream", "ls_train_stream", "ls_eval_stream", "ls_philox_x_init", "ls_shard_range", "ls_set_precision", "ls_set_path", "ls_plan_query", "ls_plan_coop_slices", "ls_trng_randn", "ls_trng_fill_steps", "ls_trng_stats", "ls_trng_set_jump", "ls_trng_jump_check", "ls_trng_pairs_debug", "ls_sag_create", "ls_sag_destroy", "ls_sag_last_error",
           "ls_sag_set_weight", "ls_sag_commit_weights", "ls_sag_decode", "ls_sag_decode_async", "ls_sag_last_decode_ms", "ls_ted_post", "ls_beat_post",
           "ls_train_create", "ls_train_destroy", "ls_train_last_error", "ls_train_set_schedule", "ls_train_param_count",
           "ls_train_flat_size", "ls_train_param_info", "ls_train_set_weight", "ls_train_get_weight", "ls_train_forward_backward",
           "ls_train_adamw", "ls_train_read", "ls_train_get_moment", "ls_train_set_moment", "ls_train_get_step", "ls_train_set_step",
           "ls_eval_create", "ls_eval_destroy", "ls_eval_last_error", "ls_eval_set_weight", "ls_eval_commit_weights", "ls_eval_features")

_lib = None


class EngineError(RuntimeError):
    pass


_lib_override = None


def use_library(path: str) -> None:
    """Developer tooling only (tools/ab_variants.py, tools/phase_profile.py): bind an alternative build of the library.
    Must be called, in code, before the first engine is created; no environment variable selects the binary."""
    global _lib_override
    if _lib is not None:
        raise EngineError("use_library() must be called before the library is loaded")
    _lib_override = os.path.abspath(path)


def library_path() -> str:
    return _lib_override or _build.LIB


def load_library(build_if_missing: bool = True):
    """dlopen libls_hip.so (building it in-tree first if it is missing or stale and hipcc is present)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: torch bundles its own libamdhip64; if libls_hip.so pulled in /opt/rocm's copy
    # first, torch's later initialisation reports "No HIP GPUs are available".  Importing torch first makes the
    # loader resolve our NEEDED libamdhip64.so.N to the copy torch already mapped (same SONAME).
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    path = library_path()
    if build_if_missing and _lib_override is None and _build.is_stale():
        try:
            _build.build_library()
        except Exception as e:      # no hipcc on this box: fall through to whatever .so travelled here
            if not os.path.exists(path):
                raise EngineError(f"libls_hip.so is missing and could not be built: {e}") from e
    if not os.path.exists(path):
        raise EngineError(f"{path} not found: run `python -m livelyspeaker_amd.build` (no CPU fallback exists)")
    lib = C.CDLL(path)
    lib.ls_abi_version.restype = C.c_int
    lib.ls_create.argtypes = [C.POINTER(LsConfig), C.POINTER(C.c_void_p)]
    lib.ls_destroy.argtypes = [C.c_void_p]
    lib.ls_destroy.restype = None
    lib.ls_last_error.argtypes = [C.c_void_p]
    lib.ls_last_error.restype = C.c_char_p
    lib.ls_set_weight.argtypes = [C.c_void_p, C.c_char_p, c_f32p, C.c_size_t]
    lib.ls_commit_weights.argtypes = [C.c_void_p]
    lib.ls_set_schedule.argtypes = [C.c_void_p, C.POINTER(LsSchedule)]
    lib.ls_prepare.argtypes = [C.c_void_p, C.POINTER(LsCond)]
    lib.ls_prepare_async.argtypes = [C.c_void_p, C.POINTER(LsCond)]
    lib.ls_sample.argtypes = [C.c_void_p, C.POINTER(LsSampleArgs)]
    lib.ls_forward.argtypes = [C.c_void_p, C.POINTER(LsForwardArgs)]
    lib.ls_step.argtypes = [C.c_void_p, C.POINTER(LsStepArgs)]
    lib.ls_q_sample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ls_read.argtypes = [C.c_void_p, C.c_char_p, c_f32p, C.c_size_t]
    lib.ls_read.restype = C.c_longlong
    lib.ls_get_timing.argtypes = [C.c_void_p, C.POINTER(LsTiming)]
    lib.ls_synchronize.argtypes = [C.c_void_p]
    lib.ls_stream_order.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    for fn in ("ls_stream", "ls_sag_stream", "ls_train_stream", "ls_eval_stream"):
        getattr(lib, fn).argtypes = [C.c_void_p]
        getattr(lib, fn).restype = C.c_void_p
    lib.ls_philox_x_init.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
    lib.ls_set_precision.argtypes = [C.c_void_p, C.c_int]
    lib.ls_set_path.argtypes = [C.c_void_p, C.c_int]
    lib.ls_plan_query.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float)]
    lib.ls_plan_coop_slices.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.ls_trng_randn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
    lib.ls_trng_fill_steps.argtypes = [C.c_void_p, C.c_size_t] + [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.ls_trng_stats.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.ls_trng_pairs_debug.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7
    lib.ls_shard_range.argtypes = [C.c_int64, C.c_int32, C.c_int32, c_i64p, c_i64p]
    lib.ls_sag_create.argtypes = [C.POINTER(LsSagConfig), C.POINTER(C.c_void_p)]
    lib.ls_sag_destroy.argtypes = [C.c_void_p]
    lib.ls_sag_destroy.restype = None
    lib.ls_sag_last_error.argtypes = [C.c_void_p]
    lib.ls_sag_last_error.restype = C.c_char_p
    lib.ls_sag_set_weight.argtypes = [C.c_void_p, C.c_char_p, c_f32p, C.c_size_t]
    lib.ls_sag_commit_weights.argtypes = [C.c_void_p]
    lib.ls_sag_decode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ls_sag_decode_async.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ls_sag_last_decode_ms.argtypes = [C.c_void_p]
    lib.ls_sag_last_decode_ms.restype = C.c_float
    lib.ls_ted_post.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(LsPostConfig), C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p]
    lib.ls_beat_post.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ls_train_create.argtypes = [C.POINTER(LsTrainConfig), C.POINTER(C.c_void_p)]
    lib.ls_train_destroy.argtypes = [C.c_void_p]
    lib.ls_train_destroy.restype = None
    lib.ls_train_last_error.argtypes = [C.c_void_p]
    lib.ls_train_last_error.restype = C.c_char_p
    lib.ls_train_set_schedule.argtypes = [C.c_void_p, c_f64p, c_f64p, c_i64p]
    lib.ls_train_param_count.argtypes = [C.c_void_p]
    lib.ls_train_flat_size.argtypes = [C.c_void_p]
    lib.ls_train_flat_size.restype = C.c_int64
    lib.ls_train_param_info.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, c_i64p, c_i64p]
    lib.ls_train_set_weight.argtypes = [C.c_void_p, C.c_char_p, c_f32p, C.c_size_t]
    lib.ls_train_get_weight.argtypes = [C.c_void_p, C.c_char_p, c_f32p, C.c_size_t]
    lib.ls_train_forward_backward.argtypes = [C.c_void_p, C.POINTER(LsTrainBatch), C.c_void_p, C.POINTER(LsTrainTerms)]
    lib.ls_train_adamw.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]
    lib.ls_train_read.argtypes = [C.c_void_p, C.c_char_p, c_f32p, C.c_size_t]
    lib.ls_train_get_moment.argtypes = [C.c_void_p, C.c_int, C.c_char_p, c_f32p, C.c_size_t]
    lib.ls_train_set_moment.argtypes = [C.c_void_p, C.c_int, C.c_char_p, c_f32p, C.c_size_t]
    lib.ls_train_get_step.argtypes = [C.c_void_p]
    lib.ls_train_get_step.restype = C.c_int64
    lib.ls_train_set_step.argtypes = [C.c_void_p, C.c_int64]
    lib.ls_eval_create.argtypes = [C.POINTER(LsEvalConfig), C.POINTER(C.c_void_p)]
    lib.ls_eval_destroy.argtypes = [C.c_void_p]
    lib.ls_eval_destroy.restype = None
    lib.ls_eval_last_error.argtypes = [C.c_void_p]
    lib.ls_eval_last_error.restype = C.c_char_p
    lib.ls_eval_set_weight.argtypes = [C.c_void_p, C.c_char_p, c_f32p, C.c_size_t]
    lib.ls_eval_commit_weights.argtypes = [C.c_void_p]
    lib.ls_eval_features.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    if lib.ls_abi_version() != 5:
        raise EngineError("libls_hip.so ABI version mismatch")
    _lib = lib
    return lib


def _is_cuda(a) -> bool:
    return hasattr(a, "is_cuda") and bool(a.is_cuda)


class _Marshal:
    """Turns a group of arrays (numpy / torch CPU / torch CUDA) into raw pointers for one ABI call.
    If ANY member is a CUDA tensor the whole group is passed as device pointers (on_device=1) and outputs
    are torch CUDA tensors; otherwise everything is host numpy.  Keeps the converted buffers alive."""

    def __init__(self, device_index: int, *members, stream=None):
        self.on_device = any(_is_cuda(m) for m in members if m is not None)
        self.device_index = device_index
        self.stream = stream        # the handle's HIP stream (ls_stream & co.): inputs are ordered in front of it, not host-synchronised
        self.keep = []
        if self.on_device:
            import torch
            self.torch = torch
            self.dev = torch.device("cuda", device_index)

    def f32(self, a, shape=None):
        return self._conv(a, np.float32, shape)

    def i64(self, a, shape=None):
        return self._conv(a, np.int64, shape)

    def u8(self, a, shape=None):
        """bool / byte mask -> bytes"""
        if a is None:
            return None
        if self.on_device:
            t = self.torch.as_tensor(a).to(device=self.dev, dtype=self.torch.uint8).contiguous()
            if shape is not None:
                assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape))
            self.keep.append(t)
            return C.c_void_p(t.data_ptr())
        if hasattr(a, "detach"):
            a = a.detach().cpu().numpy()
        n = np.ascontiguousarray(a, dtype=np.uint8)
        if shape is not None:
            assert tuple(n.shape) == tuple(shape), (n.shape, tuple(shape))
        self.keep.append(n)
        return n.ctypes.data_as(C.c_void_p)

    def _conv(self, a, dtype, shape):
        if a is None:
            return None
        if self.on_device:
            t = self.torch.as_tensor(a)
            t = t.to(device=self.dev, dtype=self.torch.float32 if dtype == np.float32 else self.torch.int64).contiguous()
            if shape is not None:
                assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape))
            self.keep.append(t)
            return C.c_void_p(t.data_ptr())
        if hasattr(a, "detach"):
            a = a.detach().cpu().numpy()
        n = np.ascontiguousarray(a, dtype=dtype)
        if shape is not None:
            assert tuple(n.shape) == tuple(shape), (n.shape, tuple(shape))
        self.keep.append(n)
        return n.ctypes.data_as(C.c_void_p)

    def ready(self):
        """Call right before the ABI call.  The engine's streams are non-blocking (nothing orders them against torch's), and the
        inputs -- as well as the dtype / contiguity conversions `_conv` may have just enqueued (e.g. the strided `emo[:, 0]` of
        the BEAT callers) and the allocation of the outputs -- belong to torch's current stream: the handle's stream is made to
        wait for that stream's work so far (an event, ls_stream_order) instead of the host waiting for it."""
        if self.on_device:
            ts = self.torch.cuda.current_stream(self.dev).cuda_stream
            if self.stream is None or load_library().ls_stream_order(self.device_index, C.c_void_p(ts), C.c_void_p(self.stream)) != 0:
                self.torch.cuda.current_stream(self.dev).synchronize()

    def done_async(self):
        """After a no_sync ABI call: torch's current stream waits for the handle's work (outputs, and inputs it still reads), so
        consumers on that stream and the allocator's reuse of the marshalled temporaries stay ordered without a host wait."""
        if self.on_device:
            ts = self.torch.cuda.current_stream(self.dev).cuda_stream
            if self.stream is None or load_library().ls_stream_order(self.device_index, C.c_void_p(self.stream), C.c_void_p(ts)) != 0:
                raise EngineError("ls_stream_order failed")

    def out(self, shape):
        """(object, pointer) for an fp32 output of this call."""
        if self.on_device:
            t = self.torch.empty(tuple(shape), dtype=self.torch.float32, device=self.dev)
            return t, C.c_void_p(t.data_ptr())
        n = np.empty(tuple(shape), np.float32)
        return n, n.ctypes.data_as(C.c_void_p)


def stream_order(device_index: int, first, then) -> None:
    """Work enqueued on stream ``then`` from now on waits for what stream ``first`` holds so far (an event; no host wait).  Streams are
    raw handles: ``Engine._stream`` / ``SagEngine._stream`` or ``torch.cuda.current_stream().cuda_stream``."""
    if load_library().ls_stream_order(int(device_index), C.c_void_p(first), C.c_void_p(then)) != 0:
        raise EngineError("ls_stream_order failed")


def _order_after_torch(device_index: int, stream) -> None:
    """Make a handle's stream wait for the work enqueued so far on torch's current stream of that device."""
    import torch
    ts = torch.cuda.current_stream(torch.device("cuda", device_index))
    if stream is None or load_library().ls_stream_order(device_index, C.c_void_p(ts.cuda_stream), C.c_void_p(stream)) != 0:
        ts.synchronize()


def _np32(a) -> np.ndarray:
    """Host fp32 C-contiguous view/copy of a numpy array or CPU torch tensor."""
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def plan_query(batch, dataset="ted", single_pass=False, precision="fp32", n_cus=256):
    """The step plan `auto` makes for ``batch`` clips (no GPU needed): ([(family, first, count), ...], model ms per step); family as in
    ``Engine.timing()['step_path']`` (0 fused, 1 batch-level, 2 sample-split, 3 one-pass-per-workgroup)."""
    lib = load_library()
    out = (C.c_int * 10)()
    ms = C.c_float()
    rc = lib.ls_plan_query(int(dataset != "ted"), int(batch), int(bool(single_pass)), {"fp32": 0, "bf16x3": 1}.get(precision, precision), int(n_cus), out, C.byref(ms))
    if rc != 0:
        raise EngineError(f"ls_plan_query failed ({rc})")
    return [(out[1 + 3 * i], out[2 + 3 * i], out[3 + 3 * i]) for i in range(out[0])], float(ms.value)


def plan_coop_slices(groups, dataset="ted", n_cus=256):
    """Slice workgroups per (sample, CFG pass) the sample-split kernel uses for ``groups`` (sample, pass) groups: 8, 4 or 2."""
    lib = load_library()
    rc = lib.ls_plan_coop_slices(int(dataset != "ted"), int(groups), int(n_cus))
    if rc < 0:
        raise EngineError(f"ls_plan_coop_slices failed ({rc})")
    return rc


class Engine:
    """One handle = one GPU. Thin, typed wrapper over the C-ABI; all arrays in/out are host numpy
    (torch CUDA tensors on the same device may be passed to ``sample``/``prepare`` via ``*_device``)."""

    #: which kernels the steps of a 34-frame model run on when the caller does not say: "auto" (the sample-split kernel for small
    #: batches, one workgroup per sample or per (sample, pass) otherwise), "fused", "batch", "coop", "pass" (ls_set_path)
    default_path = "auto"

    def __init__(self, njoints, nfeats, n_prefix_tokens, audio_len, n_emotions=0, nframes=34, n_pre_seq=4,
                 latent_dim=512, layers=8, n_speakers=1400, device=0, path=None):
        self.lib = load_library()
        self.cfg = LsConfig(njoints, nfeats, nframes, n_prefix_tokens, n_pre_seq, latent_dim, layers, audio_len,
                            n_speakers, n_emotions, device, 0)
        self.h = C.c_void_p()
        rc = self.lib.ls_create(C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            raise EngineError(f"ls_create failed ({rc}): {self.lib.ls_last_error(None).decode()}")
        self._stream = self.lib.ls_stream(self.h)
        self.path = path or Engine.default_path
        if nframes != 34:
            self.path = "batch"                       # other frame counts have only the batch-level kernels (ls_set_path refuses the rest)
        elif self.path != "auto":
            self.set_path(self.path)
        self.J, self.F, self.T, self.D = njoints, nfeats, nframes, latent_dim
        self.S = nframes + n_prefix_tokens
        self.layers = layers
        self.device = device
        self.batch = 0
        self.n_steps = 0
        self._keep = []

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.ls_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise EngineError(f"{what} failed ({rc}): {self.lib.ls_last_error(self.h).decode()}")
        return rc

    def set_precision(self, mode):
        """'fp32' (exact, default) or 'bf16x3' (split-precision channel mixing on the bf16 matrix cores)."""
        code = {"fp32": 0, "bf16x3": 1}.get(mode, mode)
        self._check(self.lib.ls_set_precision(self.h, int(code)), "ls_set_precision")
        self.precision = mode

    def set_path(self, mode):
        """'auto' (default: the sample-split kernel for small batches, one workgroup per sample otherwise), 'fused', 'batch'
        (batch-level kernels, 21 launches per step), 'coop' (sample-split kernel), 'pass' (one workgroup per (sample, CFG pass), two per
        CU), 'pass4' (its 4-wave form at every grid size), 'coop8' / 'coop4' / 'coop2' (the sample-split kernel with that many channel
        slices per (sample, CFG pass)); applies from the next prepare().  None = 'auto'.  On a long-sequence model (145-160 tokens) 'auto'
        runs the sampling loop on the one-launch mixer where every launch is at least 7/8 full, 'batch' forces the batch-level kernels and
        'coop' the mixer at every batch size; the other modes raise (ls_set_path in include/ls_hip.h)."""
        mode = "auto" if mode is None else mode
        code = {"auto": 0, "fused": 1, "batch": 2, "coop": 3, "pass": 4, "pass4": 5, "coop4": 6, "coop2": 7, "coop8": 8}.get(mode, mode)
        self._check(self.lib.ls_set_path(self.h, int(code)), "ls_set_path")
        self.path = mode

    # ---- weights / schedule --------------------------------------------------------------------
    def load_state_dict(self, sd: dict):
        for k, v in sd.items():
            a = _np32(v)
            self._check(self.lib.ls_set_weight(self.h, k.encode(), a.ctypes.data_as(c_f32p), a.size), f"ls_set_weight({k})")
        self._check(self.lib.ls_commit_weights(self.h), "ls_commit_weights")

    def set_schedule(self, sched):
        """sched: object with the GaussianDiffusion table attributes + timestep_map."""
        tabs = {}
        s = LsSchedule()
        s.n_steps = int(sched.num_timesteps)
        tmap = np.ascontiguousarray(np.asarray(sched.timestep_map), dtype=np.int64)
        s.timestep_map = tmap.ctypes.data_as(c_i64p)
        for name, _ in LsSchedule._fields_[3:]:
            tabs[name] = np.ascontiguousarray(getattr(sched, name), dtype=np.float64)
            assert tabs[name].shape == (s.n_steps,), name
            setattr(s, name, tabs[name].ctypes.data_as(c_f64p))
        self._check(self.lib.ls_set_schedule(self.h, C.byref(s)), "ls_set_schedule")
        self.n_steps = s.n_steps

    # ---- per call --------------------------------------------------------------------------------
    def prepare(self, y: dict, wait: bool = True):
        """Once-per-call stage.  ``wait=False`` (ls_prepare_async) only enqueues it on the engine's stream: the next sample / forward /
        step is ordered behind it, and work on other streams (the SAG decode) overlaps it.  The marshalled inputs are kept alive on
        this object until the next prepare, by which time a synchronising call has long consumed them."""
        emo = y.get("emo") if self.cfg.n_prefix_tokens == 2 else None
        if emo is not None:
            # scripts_beat/model/RAG.py:125 reads y['emo'][:, 0] only, whatever the width: a bare [B] vector, [B, 1] or any padded
            # [B, W] is accepted and presented to the ABI in its [B, T] form (column 0 repeated)
            if getattr(emo, "ndim", 2) not in (1, 2):
                raise EngineError(f"y['emo'] must be [B] or [B, W], got shape {tuple(emo.shape)}")
            col = emo if emo.ndim == 1 else emo[:, 0]
            if hasattr(col, "expand") and not isinstance(col, np.ndarray):
                emo = col[:, None].expand(col.shape[0], self.T)
            else:
                emo = np.broadcast_to(np.asarray(col)[:, None], (len(col), self.T))
        m = _Marshal(self.device, y["audio_input"], y["origin_x"], y["vid_indices"], y["scale"], emo, stream=self._stream)
        B = int(y["audio_input"].shape[0])
        c = LsCond(B, int(m.on_device), m.f32(y["audio_input"], (B, self.cfg.audio_len)),
                   m.f32(y["origin_x"], (B, self.J, self.F, self.T)), m.i64(y["vid_indices"], (B,)),
                   m.i64(emo, (B, self.T)), m.f32(y["scale"], (B,)))
        m.ready()
        if wait:
            self._check(self.lib.ls_prepare(self.h, C.byref(c)), "ls_prepare")
            self._prepare_inputs = None
        else:
            self._check(self.lib.ls_prepare_async(self.h, C.byref(c)), "ls_prepare_async")
            self._prepare_inputs = (m, y)       # device buffers (and any converted copies) stay valid while the copies are in flight
            if m.on_device:
                # conv1 reads a device-resident waveform IN PLACE (no ingest copy): whatever the caller enqueues next on torch's stream --
                # e.g. refilling the same batch buffer -- has to wait for the engine's stream, exactly like the outputs of a no_sync call
                m.done_async()
        self.batch = B

    def _xshape(self):
        return (self.batch, self.J, self.F, self.T)

    def forward(self, x, t, eps_c, eps_u, trace=False):
        m = _Marshal(self.device, x, t, eps_c, eps_u, stream=self._stream)
        B, D = self.batch, self.D
        eps_c = eps_c.reshape(B, D)
        eps_u = eps_u.reshape(B, D)
        oc, poc = m.out(self._xshape())
        ou, pou = m.out(self._xshape())
        og, pog = m.out(self._xshape())
        tr, ptr = m.out((B, self.layers + 1, 2 * self.S, D)) if trace else (None, None)
        a = LsForwardArgs(int(m.on_device), 0, m.f32(x, self._xshape()), m.i64(t, (B,)), m.f32(eps_c, (B, D)),
                          m.f32(eps_u, (B, D)), poc, pou, pog, ptr)
        m.ready()
        self._check(self.lib.ls_forward(self.h, C.byref(a)), "ls_forward")
        return (oc, ou, og, tr) if trace else (oc, ou, og)

    def step(self, sampler, index, x, eps_c, eps_u, noise, eta=0.0, clip_denoised=False, two_pass_always=False, indices=None,
             no_sync=False, inpaint=None):
        """One p_sample / ddim_sample step.  ``indices``: one schedule index per sample ([B] int64; numpy / CPU tensor = validated
        on the host, CUDA tensor = never read by the host) instead of the uniform ``index``.  ``no_sync`` (device tensors only):
        return without waiting for the GPU; the outputs are ordered behind the step on torch's current stream."""
        inp = inpaint or (None, None, None)          # (mask, motion, q_sample noise or None): p_mean_variance's inpainting branch
        m = _Marshal(self.device, x, eps_c, eps_u, noise, inp[1], stream=self._stream)
        B, D = self.batch, self.D
        out, pout = m.out(self._xshape())
        x0, px0 = m.out(self._xshape())
        pidx, idx_dev = None, 0
        if indices is not None:
            if _is_cuda(indices):
                if not m.on_device:
                    raise EngineError("device `indices` need device tensors for the other arguments")
                t = indices.to(dtype=m.torch.int64).contiguous()
                if tuple(t.shape) != (B,):
                    raise EngineError(f"indices must be [{B}], got {tuple(t.shape)}")
                m.keep.append(t)
                pidx, idx_dev = C.c_void_p(t.data_ptr()), 1
            else:
                n = np.ascontiguousarray(indices.detach().cpu().numpy() if hasattr(indices, "detach") else indices, dtype=np.int64)
                if n.shape != (B,):
                    raise EngineError(f"indices must be [{B}], got {n.shape}")
                m.keep.append(n)
                pidx = n.ctypes.data_as(C.c_void_p)
        nosync = int(bool(no_sync) and m.on_device)
        a = LsStepArgs(sampler, int(index), int(m.on_device), eta, int(clip_denoised), int(two_pass_always), m.f32(x, self._xshape()),
                       m.f32(eps_c.reshape(B, D)), m.f32(eps_u.reshape(B, D)), m.f32(noise, self._xshape()), pout, px0, pidx, nosync, idx_dev,
                       m.u8(inp[0], self._xshape()), m.f32(inp[1], self._xshape()), m.f32(inp[2], self._xshape()))
        m.ready()
        self._check(self.lib.ls_step(self.h, C.byref(a)), "ls_step")
        if nosync:
            m.done_async()
            self._inflight = m          # marshalled copies stay referenced until the next call replaces them
        return out, x0

    def sample(self, sampler=LS_SAMPLER_DDPM, x_init=None, eps_tape=None, noise_tape=None, init_image=None,
               skip_timesteps=0, eta=0.0, const_noise=False, dump_steps=None, philox_seed=None, sample_offset=0,
               use_graph=True, clip_denoised=False, device_out=False, two_pass_always=False, segment=None, inpaint=None):
        """Run the whole loop. TAPE mode when tapes are given, PHILOX mode when ``philox_seed`` is.
        Outputs are torch CUDA tensors if any input is one (or ``device_out``), else numpy."""
        inp = inpaint or (None, None, None, False)     # (mask, motion, q_sample noise tape or None, re-noise?): the inpainting branch
        members = [x_init, eps_tape, noise_tape, init_image, inp[1]]
        if device_out:
            import torch
            members.append(torch.empty(1, device=torch.device("cuda", self.device)))
        m = _Marshal(self.device, *members, stream=self._stream)
        a = LsSampleArgs()
        a.sampler, a.skip_timesteps, a.const_noise, a.on_device = sampler, skip_timesteps, int(const_noise), int(m.on_device)
        a.use_graph, a.eta, a.clip_denoised = int(use_graph), eta, int(clip_denoised)
        a.two_pass_always = int(two_pass_always)
        n_exec = self.n_steps - skip_timesteps
        if segment is not None:         # (first executed-step counter, count): one piece of a TAPE-mode loop, tapes hold these steps only
            a.seg_begin, a.seg_count = int(segment[0]), int(segment[1])
            n_tape, last = a.seg_count, a.seg_begin + a.seg_count == n_exec
        else:
            n_tape, last = n_exec, True
        if philox_seed is None:
            a.noise_mode = LS_NOISE_TAPE
            a.eps_tape = m.f32(eps_tape, (n_tape, 2, self.batch, self.D))
            a.noise_tape = m.f32(noise_tape, (n_tape,) + self._xshape())
        else:
            a.noise_mode = LS_NOISE_PHILOX
            a.seed, a.sample_offset = int(philox_seed), int(sample_offset)
        a.x_init = m.f32(x_init, self._xshape())
        a.init_image = m.f32(init_image, self._xshape())
        if inp[0] is not None:
            a.inpaint_mask = m.u8(inp[0], self._xshape())
            a.inpainted_motion = m.f32(inp[1], self._xshape())
            a.inpaint_noised = int(bool(inp[3]))
            if inp[2] is not None:
                a.inpaint_noise = m.f32(inp[2], (n_tape,) + self._xshape())
        out, a.out = m.out(self._xshape())
        dumps = None
        if dump_steps:
            ds = np.ascontiguousarray(dump_steps, dtype=np.int32)
            dumps, a.dump_out = m.out((len(ds),) + self._xshape())
            a.n_dump, a.dump_steps = len(ds), ds.ctypes.data_as(c_i32p)
            m.keep.append(ds)
        m.ready()
        self._check(self.lib.ls_sample(self.h, C.byref(a)), "ls_sample")
        if not last:
            self._segment_inputs = m     # page-locked host tapes of this segment: referenced until the next segment call has returned
            return None
        self._segment_inputs = None
        return (out, dumps) if dump_steps else out

    def q_sample(self, index, x_start, noise):
        m = _Marshal(self.device, x_start, noise, stream=self._stream)
        out, pout = m.out(tuple(x_start.shape))
        n = int(np.prod(x_start.shape))
        m.ready()
        self._check(self.lib.ls_q_sample(self.h, index, int(m.on_device), n, m.f32(x_start), m.f32(noise), pout), "ls_q_sample")
        return out

    def philox_x_init(self, batch, seed, sample_offset=0) -> np.ndarray:
        out = np.empty((batch, self.J, self.F, self.T), np.float32)
        self._check(self.lib.ls_philox_x_init(self.h, batch, int(seed), int(sample_offset), 0,
                                              out.ctypes.data_as(C.c_void_p)), "ls_philox_x_init")
        return out

    def read(self, name: str) -> np.ndarray:
        shapes = {"audio_feat": (self.batch, self.T, 256), "static_c": (self.batch, self.T, self.D),
                  "static_u": (self.batch, self.T, self.D), "z_mu": (self.batch, self.D),
                  "z_logvar": (self.batch, self.D), "z_std": (self.batch, self.D), "temb": (self.n_steps, self.D)}
        out = np.empty(shapes[name], np.float32)
        n = self._check(self.lib.ls_read(self.h, name.encode(), out.ctypes.data_as(c_f32p), out.size), f"ls_read({name})")
        assert n == out.size
        return out

    def synchronize(self):
        """Wait for everything enqueued on the engine's stream (an asynchronous prepare included)."""
        self._check(self.lib.ls_synchronize(self.h), "ls_synchronize")

    def timing(self) -> dict:
        t = LsTiming()
        self.lib.ls_get_timing(self.h, C.byref(t))
        return {k: getattr(t, k) for k, _ in LsTiming._fields_}


class SagEngine:
    """ctypes wrapper of the SAG decoder handle (ls_sag_*): Decoder_TRANSFORMER.forward on the GPU."""

    def __init__(self, njoints=9, nfeats=3, nframes=34, latent_dim=512, ff_size=1024, num_layers=3, num_heads=4,
                 n_pre_poses=4, device=0):
        self.lib = load_library()
        self.cfg = LsSagConfig(njoints, nfeats, nframes, latent_dim, ff_size, num_layers, num_heads, n_pre_poses, device, 0)
        self.h = C.c_void_p()
        rc = self.lib.ls_sag_create(C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            raise EngineError(f"ls_sag_create failed ({rc}): {self.lib.ls_sag_last_error(None).decode()}")
        self.J, self.F, self.T, self.D, self.device = njoints, nfeats, nframes, latent_dim, device
        self._stream = self.lib.ls_sag_stream(self.h)

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.ls_sag_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise EngineError(f"{what} failed ({rc}): {self.lib.ls_sag_last_error(self.h).decode()}")

    def load_state_dict(self, sd: dict):
        for k, v in sd.items():
            a = _np32(v)
            self._check(self.lib.ls_sag_set_weight(self.h, k.encode(), a.ctypes.data_as(c_f32p), a.size), f"ls_sag_set_weight({k})")
        self._check(self.lib.ls_sag_commit_weights(self.h), "ls_sag_commit_weights")

    def last_decode_ms(self) -> float:
        return float(self.lib.ls_sag_last_decode_ms(self.h))

    def decode(self, x, z, mask=None, wait=True):
        """``wait=False`` (device tensors only): enqueue on the decoder's stream and return at once; the output tensor is complete once
        that stream has reached this point -- ``order_after(self, other_engine)`` / ``stream_order`` puts a consumer behind it."""
        m = _Marshal(self.device, x, z, mask, stream=self._stream)
        B = int(x.shape[0])
        out, pout = m.out((B, self.J, self.F, self.T))
        pmask = None
        if mask is not None:
            if m.on_device:
                t = m.torch.as_tensor(mask).to(device=m.dev, dtype=m.torch.uint8).contiguous()
                m.keep.append(t)
                pmask = C.c_void_p(t.data_ptr())
            else:
                a = mask.detach().cpu().numpy() if hasattr(mask, "detach") else mask
                a = np.ascontiguousarray(a, dtype=np.uint8)
                m.keep.append(a)
                pmask = a.ctypes.data_as(C.c_void_p)
        m.ready()
        if not wait:
            if not m.on_device:
                raise EngineError("decode(wait=False) needs device tensors")
            self._check(self.lib.ls_sag_decode_async(self.h, B, m.f32(x, (B, self.J, self.F, self.T)), m.f32(z, (B, self.D)), pmask, pout),
                        "ls_sag_decode_async")
            self._async_inputs = m          # the marshalled inputs stay referenced until the next decode on this (in-order) stream
            return out
        self._check(self.lib.ls_sag_decode(self.h, B, int(m.on_device), m.f32(x, (B, self.J, self.F, self.T)),
                                           m.f32(z, (B, self.D)), pmask, pout), "ls_sag_decode")
        self._async_inputs = None
        return out


class Trainer:
    """ctypes wrapper of the training-step handle (ls_train_*).  Master parameters, gradients and Adam moments are flat
    device arrays; ``grad`` is a torch CUDA tensor owned by this object so a data-parallel caller can all-reduce it
    (RCCL) between ``forward_backward`` and ``adamw``."""

    def __init__(self, njoints, nfeats, n_prefix_tokens, audio_len, n_emotions=0, nframes=34, n_pre_seq=4, layers=8,
                 n_speakers=1400, device=0, diffusion_steps=1000, lambda_vel=1.0, kld_weight=0.01):
        import torch
        self.lib = load_library()
        self.cfg = LsTrainConfig(LsConfig(njoints, nfeats, nframes, n_prefix_tokens, n_pre_seq, 512, layers, audio_len, n_speakers,
                                          n_emotions, device, 0), lambda_vel, kld_weight, diffusion_steps, 0)
        self.h = C.c_void_p()
        rc = self.lib.ls_train_create(C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            raise EngineError(f"ls_train_create failed ({rc}): {self.lib.ls_train_last_error(None).decode()}")
        self.J, self.F, self.T, self.device, self.n_prefix = njoints, nfeats, nframes, device, n_prefix_tokens
        self._stream = self.lib.ls_train_stream(self.h)
        self.params = {}
        key = C.create_string_buffer(256)
        off, num = C.c_int64(), C.c_int64()
        for i in range(self.lib.ls_train_param_count(self.h)):
            self._check(self.lib.ls_train_param_info(self.h, i, key, 256, C.byref(off), C.byref(num)), "ls_train_param_info")
            self.params[key.value.decode()] = (int(off.value), int(num.value))
        self.flat_size = int(self.lib.ls_train_flat_size(self.h))
        self.grad = torch.zeros(self.flat_size, dtype=torch.float32, device=torch.device("cuda", device))
        self.shapes = {}

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.ls_train_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise EngineError(f"{what} failed ({rc}): {self.lib.ls_train_last_error(self.h).decode()}")

    def load_state_dict(self, sd: dict):
        for k, v in sd.items():
            a = _np32(v)
            self.shapes[k] = tuple(a.shape)
            self._check(self.lib.ls_train_set_weight(self.h, k.encode(), a.ctypes.data_as(c_f32p), a.size), f"ls_train_set_weight({k})")

    def state_dict(self) -> dict:
        out = {}
        for k, (_, n) in self.params.items():
            a = np.empty(n, np.float32)
            self._check(self.lib.ls_train_get_weight(self.h, k.encode(), a.ctypes.data_as(c_f32p), n), f"ls_train_get_weight({k})")
            out[k] = a.reshape(self.shapes.get(k, (n,)))
        return out

    def set_schedule(self, sched):
        """sched: anything with .sqrt_alphas_cumprod, .sqrt_one_minus_alphas_cumprod (fp64) and .timestep_map."""
        a = np.ascontiguousarray(sched.sqrt_alphas_cumprod, dtype=np.float64)
        b = np.ascontiguousarray(sched.sqrt_one_minus_alphas_cumprod, dtype=np.float64)
        m = np.ascontiguousarray(sched.timestep_map, dtype=np.int64)
        if not (len(a) == len(b) == len(m) == self.cfg.diffusion_steps):
            raise EngineError("schedule length does not match diffusion_steps")
        self._check(self.lib.ls_train_set_schedule(self.h, a.ctypes.data_as(c_f64p), b.ctypes.data_as(c_f64p), m.ctypes.data_as(c_i64p)),
                    "ls_train_set_schedule")

    def forward_backward(self, x_start, t, noise, y: dict, drop, eps) -> dict:
        """One forward + loss + backward; gradients land in ``self.grad`` (flat, device). Returns the loss terms."""
        import torch
        m = _Marshal(self.device, x_start, noise, drop, eps, y["audio_input"], y["origin_x"], y["vid_indices"], y.get("emo"), stream=self._stream)
        B = int(x_start.shape[0])
        xs = (B, self.J, self.F, self.T)
        tt = np.ascontiguousarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.int64)
        assert tt.shape == (B,)
        tb = LsTrainBatch(B, int(m.on_device), m.f32(x_start, xs), tt.ctypes.data_as(C.c_void_p), m.f32(noise, xs), m.f32(drop, (B,)),
                          m.f32(np.asarray(eps).reshape(B, 512) if not hasattr(eps, "detach") else eps.reshape(B, 512), (B, 512)),
                          m.f32(y["audio_input"]), m.f32(y["origin_x"], xs), m.i64(y["vid_indices"], (B,)),
                          m.i64(y["emo"], (B, self.T)) if self.n_prefix == 2 else None)
        terms = LsTrainTerms()
        m.ready()
        _order_after_torch(self.device, self._stream)     # whatever last touched self.grad on torch's stream (zero_, an all-reduce)
        self._check(self.lib.ls_train_forward_backward(self.h, C.byref(tb), C.c_void_p(self.grad.data_ptr()), C.byref(terms)),
                    "ls_train_forward_backward")
        return {n: float(getattr(terms, n)) for n in ("rot_mse", "vel_mse", "kld", "loss", "total", "fwd_ms", "bwd_ms")}

    def grads(self) -> dict:
        g = self.grad.detach().cpu().numpy()
        return {k: g[o:o + n].reshape(self.shapes.get(k, (n,))).copy() for k, (o, n) in self.params.items()}

    def adamw(self, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        _order_after_torch(self.device, self._stream)     # an all-reduce of self.grad may still be in flight: ordered, not waited for
        self._check(self.lib.ls_train_adamw(self.h, C.c_void_p(self.grad.data_ptr()), lr, betas[0], betas[1], eps, weight_decay),
                    "ls_train_adamw")

    def optimizer_state(self) -> dict:
        """{'step': int, 'exp_avg': {key: array}, 'exp_avg_sq': {key: array}} (what opt%09d.pt holds, train_loop.py:222-227)."""
        out = {"step": int(self.lib.ls_train_get_step(self.h)), "exp_avg": {}, "exp_avg_sq": {}}
        for which, name in ((1, "exp_avg"), (2, "exp_avg_sq")):
            for k, (_, n) in self.params.items():
                a = np.empty(n, np.float32)
                self._check(self.lib.ls_train_get_moment(self.h, which, k.encode(), a.ctypes.data_as(c_f32p), n), "ls_train_get_moment")
                out[name][k] = a.reshape(self.shapes.get(k, (n,)))
        return out

    def load_optimizer_state(self, st: dict):
        for which, name in ((1, "exp_avg"), (2, "exp_avg_sq")):
            for k, v in st[name].items():
                a = _np32(v)
                self._check(self.lib.ls_train_set_moment(self.h, which, k.encode(), a.ctypes.data_as(c_f32p), a.size), "ls_train_set_moment")
        self._check(self.lib.ls_train_set_step(self.h, int(st["step"])), "ls_train_set_step")

    def read(self, name: str, shape) -> np.ndarray:
        a = np.empty(tuple(shape), np.float32)
        self._check(self.lib.ls_train_read(self.h, name.encode(), a.ctypes.data_as(c_f32p), a.size), f"ls_train_read({name})")
        return a


class EvalEngine:
    """ctypes wrapper of the FGD feature extractor (ls_eval_*): PoseEncoderConv in eval mode on the GPU."""

    def __init__(self, pose_dim=27, n_frames=34, base=32, hidden=(256, 128), device=0):
        self.lib = load_library()
        self.cfg = LsEvalConfig(pose_dim, n_frames, base, hidden[0], hidden[1], device)
        self.h = C.c_void_p()
        rc = self.lib.ls_eval_create(C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            raise EngineError(f"ls_eval_create failed ({rc}): {self.lib.ls_eval_last_error(None).decode()}")
        self.pose_dim, self.T, self.base, self.device = pose_dim, n_frames, base, device
        self._stream = self.lib.ls_eval_stream(self.h)

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.ls_eval_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise EngineError(f"{what} failed ({rc}): {self.lib.ls_eval_last_error(self.h).decode()}")

    def load_state_dict(self, sd: dict):
        for k, v in sd.items():
            a = _np32(v).ravel()
            self._check(self.lib.ls_eval_set_weight(self.h, k.encode(), a.ctypes.data_as(c_f32p), a.size), f"ls_eval_set_weight({k})")
        self._check(self.lib.ls_eval_commit_weights(self.h), "ls_eval_commit_weights")

    def features(self, poses):
        m = _Marshal(self.device, poses, stream=self._stream)
        B = int(poses.shape[0])
        out, pout = m.out((B, self.base))
        m.ready()
        self._check(self.lib.ls_eval_features(self.h, B, int(m.on_device), m.f32(poses, (B, self.T, self.pose_dim)), pout), "ls_eval_features")
        return out
