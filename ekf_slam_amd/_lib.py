"""ctypes binding of libekfslam.so -- exactly the symbols include/ekfslam.h declares.

There is no CPU or PyTorch fallback: if the shared library is missing this raises, and on a machine
without a HIP device ``ekf_create`` returns EKF_ERR_NO_DEVICE (surfaced as ``EkfError``).
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EKF_LIB_PATH") or os.path.join(_HERE, "libekfslam.so")   # override: A/B builds

EKF_OK = 0
EKF_ERR_INVALID_ARG, EKF_ERR_NO_DEVICE, EKF_ERR_HIP, EKF_ERR_CAPACITY = 1, 2, 3, 4
EKF_ERR_INDEX, EKF_ERR_LOOKUP, EKF_ERR_STATE, EKF_ERR_COMM = 5, 6, 7, 8
EKF_MODE_KNOWN, EKF_MODE_UC = 0, 1
EKF_COMM_ID_BYTES = 128
EKF_STORE_F64, EKF_STORE_F32 = 0, 1
EKF_ARITH_F64, EKF_ARITH_F32, EKF_ARITH_SPLIT3 = 0, 1, 2
(EKF_KERNEL_DOWNDATE, EKF_KERNEL_GATHER, EKF_KERNEL_PREDICT, EKF_KERNEL_ASSOCIATE, EKF_KERNEL_APPEND,
 EKF_KERNEL_ROWPANEL, EKF_KERNEL_EXCHANGE, EKF_KERNEL_COUNT) = range(8)

_d = ctypes.c_double
_dp = ctypes.POINTER(ctypes.c_double)
_i32 = ctypes.c_int32
_i64 = ctypes.c_int64
_vp = ctypes.c_void_p


class EkfConfig(ctypes.Structure):
    """struct ekf_config (include/ekfslam.h)."""
    _fields_ = [("C", _d), ("Rc", _d * 2), ("s_cost", _d), ("s_thresh", _d), ("w_pos", _d),
                ("capacity_landmarks", _i64), ("mode", _i32), ("storage", _i32), ("device", _i32),
                ("tile", _i32), ("rank", _i32), ("world", _i32), ("batch", _i32), ("async_flush", _i32), ("device_assoc", _i32),
                ("pass_direction", _i32), ("force_sharded", _i32), ("pass_arith", _i32), ("reserved", _i32 * 2)]


# name -> (restype, argtypes); every symbol of include/ekfslam.h
SIGNATURES = {
    "ekf_abi_version": (_i32, []),
    "ekf_status_string": (ctypes.c_char_p, [_i32]),
    "ekf_config_default": (_i32, [ctypes.POINTER(EkfConfig), _i32]),
    "ekf_create": (_i32, [ctypes.POINTER(EkfConfig), ctypes.POINTER(_vp)]),
    "ekf_destroy": (_i32, [_vp]),
    "ekf_last_error": (ctypes.c_char_p, [_vp]),
    "ekf_set_stream": (_i32, [_vp, _vp]),
    "ekf_sync": (_i32, [_vp]),
    "ekf_flush": (_i32, [_vp]),
    "ekf_pending": (_i32, [_vp, ctypes.POINTER(_i32)]),
    "ekf_set_params": (_i32, [_vp, _d, _dp, _d, _d, _d]),
    "ekf_predict": (_i32, [_vp, _dp]),
    "ekf_motion_model": (_i32, [_dp, _i64, _dp, _dp, _dp]),
    "ekf_append": (_i32, [_vp, _dp, _dp, _dp, _d]),
    "ekf_correct": (_i32, [_vp, _dp, _dp, _i64]),
    "ekf_associate": (_i32, [_vp, _dp, _dp, ctypes.POINTER(_i32), ctypes.POINTER(_i64), _dp, _dp]),
    "ekf_associate_begin": (_i32, [_vp, _dp, _dp, _i32]),
    "ekf_associate_finish": (_i32, [_vp, ctypes.POINTER(_i32), ctypes.POINTER(_i64), _dp, _dp]),
    "ekf_measure": (_i32, [_vp, _dp, _i64, _dp, _dp, _dp, _i64]),
    "ekf_comm_unique_id": (_i32, [ctypes.c_char_p]),
    "ekf_comm_init": (_i32, [_vp, ctypes.c_char_p]),
    "ekf_hint_next": (_i32, [_vp, _i64]),
    "ekf_correct_begin": (_i32, [_vp, _dp, _dp, _i64]),
    "ekf_correct_finish": (_i32, [_vp]),
    "ekf_prefetch_rows": (_i32, [_vp, ctypes.POINTER(_i64), _i32]),
    "ekf_prefetch_next": (_i32, [_vp, ctypes.POINTER(_i64), _i32]),
    "ekf_prefetch_begin": (_i32, [_vp, ctypes.POINTER(_i64), _i32]),
    "ekf_prefetch_finish": (_i32, [_vp]),
    "ekf_exchange_info": (_i32, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_i64),
                                 ctypes.POINTER(_i64)]),
    "ekf_exchange_set_buffers": (_i32, [_vp, _vp, _vp]),
    "ekf_exchange_local": (_i32, [ctypes.POINTER(_vp), _i32]),
    "ekf_exchange_set_hook": (_i32, [_vp, _vp, _vp]),
    "ekf_shard_owner": (_i32, [_i32, _i64, _i64]),
    "ekf_shard_slot": (_i64, [_i32, _i64, _i64]),
    "ekf_shard_panel_source": (_i32, [_i32, _i64, _i64, ctypes.POINTER(_i32), ctypes.POINTER(_i64)]),
    "ekf_num_landmarks": (_i32, [_vp, ctypes.POINTER(_i64)]),
    "ekf_get_x": (_i32, [_vp, _dp]),
    "ekf_set_x": (_i32, [_vp, _dp, _i64]),
    "ekf_get_s": (_i32, [_vp, _dp]),
    "ekf_set_s": (_i32, [_vp, _dp, _i64]),
    "ekf_diag_poke_device_signature": (_i32, [_vp, _i64, _d]),
    "ekf_get_P": (_i32, [_vp, _dp]),
    "ekf_set_P": (_i32, [_vp, _dp, _i64]),
    "ekf_get_P_block": (_i32, [_vp, _i64, _i64, _i64, _i64, _dp]),
    "ekf_get_P_diag_blocks": (_i32, [_vp, _dp]),
    "ekf_get_Q": (_i32, [_vp, _dp]),
    "ekf_load_lowrank_state": (_i32, [_vp, _i64, _dp, _dp, _dp, _dp, _i64]),
    "ekf_checkpoint_save": (_i32, [_vp, ctypes.c_char_p]),
    "ekf_checkpoint_load": (_i32, [_vp, ctypes.c_char_p]),
    "ekf_P_digest": (_i32, [_vp, _dp]),
    "ekf_device_bytes": (_i32, [_vp, ctypes.POINTER(_i64)]),
    "ekf_kernel_timing_enable": (_i32, [_vp, _i32, _i32]),
    "ekf_kernel_timing_read": (_i32, [_vp, _i32, ctypes.POINTER(_i64), _dp]),
    "ekf_downdate_kernel_name": (ctypes.c_char_p, [_vp, ctypes.POINTER(_i32)]),
    "ekf_downdate_algorithmic_bytes": (_i32, [_vp, ctypes.POINTER(_i64)]),
}

_LIB = None


class EkfError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("libekfslam status %d: %s" % (status, message))
        self.status = status


def build():
    """Compile libekfslam.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")])


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libekfslam.so is not built (%s); run __graft_entry__.build() or "
                              "`make -C ekf_slam_amd/csrc` -- there is no fallback path" % LIB_PATH)
        # One HIP runtime per process: the PyTorch-ROCm wheel bundles its own libamdhip64 / librccl (sonames
        # libamdhip64.so.7 / librccl.so.1).  If torch is going to be used in this process (streams,
        # torch.distributed) it must be loaded FIRST so that libekfslam's NEEDED libamdhip64.so.7 and its
        # dlopen("librccl.so.1") bind to the copies torch already mapped; loading /opt/rocm's copy first and
        # torch's second leaves two runtimes fighting over the device ("No HIP GPUs are available").
        if os.environ.get("EKF_NO_TORCH", "0") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)       # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB

EXCHANGE_HOOK = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p)      # int32_t (*)(void *ctx): ekf_exchange_set_hook
