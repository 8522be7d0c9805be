"""ctypes binding of libsage355.so (C ABI: include/sage355.h).

PyTorch is plumbing here: it owns device memory and streams; every argument
that crosses the boundary is a raw device pointer, a size or a stream handle.
There is NO fallback: if the library is missing or a call fails, this raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_int32, c_int64, c_size_t, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SAGE355_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libsage355.so")   # env: A/B another build
CSRC_DIR = os.path.join(os.path.dirname(_HERE), "csrc")

ACT_RELU, ACT_SIGMOID, ACT_NONE = 0, 1, 2
TAG_INNER, TAG_OUTER, TAG_INNER_SELF = 1, 2, 3
MAX_FANOUT = 64
ABI_VERSION = 6
EINVAL, EUNSUPPORTED, ELAUNCH, ENOSPACE = -1, -2, -3, -4      # include/sage355.h

# every symbol include/sage355.h declares (tests check the library exports each one)
SYMBOLS = [
    "sage_abi_version", "sage_last_error", "sage_build_arch", "sage_frontier_reset", "sage_sample_neighbors",
    "sage_frontier_insert", "sage_gather_mean", "sage_linear_act", "sage_layer_forward", "sage_layer_forward_supported",
    "sage_forward2_layout", "sage_forward2_init", "sage_forward2", "sage_forward2_profiled",
    "sage_linear_act_backward", "sage_gather_mean_backward",
    "sage_linear_act_backward_workspace_bytes", "sage_linear_act_backward_ws",
    "sage_gather_mean_backward_workspace_bytes", "sage_gather_mean_backward_ws",
    "sage_row_order_workspace_bytes", "sage_row_order",
    "sage_two_hop_grad_w1_workspace_bytes", "sage_two_hop_grad_w1",
    "sage_prepared_weight_bytes", "sage_prepare_weights",
    "sage_pipe_create", "sage_pipe_destroy", "sage_pipe_update_weights", "sage_pipe_submit", "sage_pipe_submit_profiled", "sage_pipe_submit_many",
    "sage_pipe_join", "sage_pipe_fork", "sage_pipe_reset", "sage_pipe_set_threads", "sage_pipe_flush",
    "sage_pipe_express_count",
]
PIPE_MAX_DEPTH = 8


class SageError(RuntimeError):
    pass


class Frontier(Structure):
    _fields_ = [("keys", c_void_p), ("rows", c_void_p), ("capacity", c_int32), ("nodes", c_void_p),
                ("count", c_void_p), ("max_nodes", c_int32)]


class Model(Structure):
    _fields_ = [("rowptr1", c_void_p), ("col1", c_void_p), ("rowptr2", c_void_p), ("col2", c_void_p),
                ("num_nodes", c_int64), ("table", c_void_p), ("table_ld", c_int64), ("d0", c_int32),
                ("w1", c_void_p), ("h1", c_int32), ("w2", c_void_p), ("h2", c_int32),
                ("k1", c_int32), ("k2", c_int32), ("concat", c_int32), ("agg_self_loop", c_int32),
                ("act1", c_int32), ("act2", c_int32), ("nan_empty", c_int32), ("fused", c_int32), ("ws_batch", c_int32),
                ("queue", c_void_p), ("queue_len", c_int32), ("queue_cursor", c_void_p), ("w1_prepared", c_void_p), ("seed_map", c_void_p),
                ("table_sliced", c_void_p), ("table_slice_floats", c_int32), ("w1_is_identity", c_int32)]


class Batch(Structure):      # sage_batch_t, lives in device memory (16 bytes)
    _fields_ = [("seeds", c_void_p), ("seed", c_uint64)]


class WsLayout(Structure):
    _fields_ = [("total_bytes", c_size_t), ("counters", c_size_t),
                ("hash_keys", c_size_t), ("hash_rows", c_size_t), ("hash_capacity", c_int32),
                ("s1_nodes", c_size_t), ("max_s1", c_int32),
                ("nbr2", c_size_t), ("slot2", c_size_t), ("cnt2", c_size_t), ("self_slot2", c_size_t),
                ("row2", c_size_t), ("self_row2", c_size_t),
                ("nbr1", c_size_t), ("cnt1", c_size_t),
                ("agg1", c_size_t), ("h1", c_size_t), ("agg2", c_size_t), ("layer1_split", c_int32)]


_lib = None


def build(verbose=False):
    """Compile libsage355.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    import subprocess
    res = subprocess.run(["make", "-C", CSRC_DIR, "-j8"], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise SageError("building libsage355.so failed (see output above)")
    return LIB_PATH


def lib():
    """The loaded library.  Missing library = hard error (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SageError(f"{LIB_PATH} not found: build it with `make -C {CSRC_DIR}` "
                        "(or __graft_entry__.build()); sage355 has no non-HIP path")
    # torch ships its own libamdhip64.so (same SONAME as /opt/rocm's).  It must be in the process
    # BEFORE libsage355 is mapped, so that the loader resolves our NEEDED libamdhip64.so.7 to that
    # one runtime: two HIP runtimes in a process do not share streams or allocations.
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    L.sage_abi_version.restype = c_int32
    L.sage_last_error.restype = c_char_p
    L.sage_build_arch.restype = c_char_p
    if L.sage_abi_version() != ABI_VERSION:
        raise SageError(f"libsage355 ABI {L.sage_abi_version()} != binding ABI {ABI_VERSION}")
    P, I32, I64 = c_void_p, c_int32, c_int64
    L.sage_frontier_reset.argtypes = [POINTER(Frontier), I32, P]
    L.sage_sample_neighbors.argtypes = [P, P, I64, P, I32, P, I32, c_uint64, c_uint32, P, P, P,
                                        POINTER(Frontier), I32, P, P, P]
    L.sage_frontier_insert.argtypes = [P, P, I32, P, I32, P, POINTER(Frontier), P, P, P]
    L.sage_gather_mean.argtypes = [P, I64, I64, I32, P, P, I32, I32, P, P, P, P, P, I64, P]
    L.sage_linear_act.argtypes = [P, I64, P, P, I64, I32, P, I64, I32, I32, I32, P, P, I64, P]
    L.sage_layer_forward.argtypes = [P, I64, I64, I32, P, P, I32, I32, P, P, P, P, I32, P, P, I64, I32, I32, P, I64, P]
    L.sage_layer_forward_supported.argtypes = [I32, I32, I32]
    L.sage_forward2_layout.argtypes = [POINTER(Model), I32, POINTER(WsLayout)]
    L.sage_forward2_init.argtypes = [POINTER(Model), P, c_size_t, I32, P]
    L.sage_forward2.argtypes = [POINTER(Model), P, c_size_t, P, I32, c_uint64, P, I64, P]
    L.sage_forward2_profiled.argtypes = [POINTER(Model), P, c_size_t, P, I32, c_uint64, P, I64, P, POINTER(c_void_p)]
    L.sage_linear_act_backward.argtypes = [P, I64, P, P, I64, I32, P, I64, I32, I32, P, I64, P, I64, I32, P,
                                           P, I64, P, I64, P]
    L.sage_gather_mean_backward.argtypes = [P, I64, I32, P, P, I32, I32, P, P, P, P, I64, I64, P]
    L.sage_linear_act_backward_workspace_bytes.argtypes = [I32, I32, I32, I32]
    L.sage_linear_act_backward_ws.argtypes = [P, I64, P, P, I64, I32, P, I64, I32, I32, P, I64, P, I64, I32, P,
                                              P, I64, P, I64, P, P, c_size_t, P]
    L.sage_two_hop_grad_w1_workspace_bytes.argtypes = [I32, I32, I32, I32, I32]
    L.sage_two_hop_grad_w1.argtypes = [P, I64, P, P, I32, P, I32, P, I64, I32, I32, P, I64, I32, I32, P, I64, P, P, I64, P, c_size_t, P]
    L.sage_row_order_workspace_bytes.argtypes = [I32]
    L.sage_row_order.argtypes = [P, I32, P, I32, P, P, c_size_t, P]
    L.sage_gather_mean_backward_workspace_bytes.argtypes = [I32, I32, I64]
    L.sage_gather_mean_backward_ws.argtypes = [P, I64, I32, P, P, I32, I32, P, P, P, P, I64, P, I64, P, c_size_t, P]
    L.sage_prepared_weight_bytes.argtypes = [I32, I32, I32]
    L.sage_prepare_weights.argtypes = [P, I64, I32, I32, I32, P, c_size_t, P]
    L.sage_pipe_create.argtypes = [POINTER(Model), I32, I32, POINTER(c_void_p), c_size_t, POINTER(c_void_p), POINTER(c_void_p)]
    L.sage_pipe_destroy.argtypes = [P]
    L.sage_pipe_update_weights.argtypes = [P, P, P, P]
    L.sage_pipe_submit.argtypes = [P, P, c_uint64, P, I64]
    L.sage_pipe_submit_profiled.argtypes = [P, P, c_uint64, P, I64, POINTER(c_void_p)]
    L.sage_pipe_submit_many.argtypes = [P, P, I64, POINTER(c_uint64), I32, P, I64, I64, I32, I32]
    L.sage_pipe_join.argtypes = [P, P]
    L.sage_pipe_fork.argtypes = [P, P]
    L.sage_pipe_reset.argtypes = [P]
    L.sage_pipe_set_threads.argtypes = [P, I32, I32]
    L.sage_pipe_flush.argtypes = [P]
    L.sage_pipe_express_count.argtypes = [P]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name == "sage_prepared_weight_bytes" or name.endswith("_workspace_bytes"):
            fn.restype = c_size_t
        elif name == "sage_pipe_express_count":
            fn.restype = c_int64
        elif name not in ("sage_last_error", "sage_build_arch"):
            fn.restype = c_int32
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        msg = lib().sage_last_error().decode("utf-8", "replace")
        raise SageError(f"{what or 'libsage355'} failed (rc={rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def stream_handle():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
