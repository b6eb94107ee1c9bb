"""utils/training.py:151-177 + train.py:61-64 -- optimiser objects and the gradient step.

``compute_gradients(optimizer, loss_model, store)`` = global-norm clip + apply, on the flat
buffer of a ParamStore, with ONE all-reduce of that buffer when torch.distributed is initialised
(backend "nccl" is RCCL over xGMI on ROCm; "gloo" in CPU tests).
"""
import os

import torch
import torch.distributed as dist

from . import ops


class AdamOptimizer:
    """tf.train.AdamOptimizer(lr, epsilon=1e-4) (train.py:64): epsilon outside the bias correction."""

    def __init__(self, learning_rate=0.01, beta1=0.9, beta2=0.999, epsilon=1e-4):
        self.lr, self.beta1, self.beta2, self.epsilon = learning_rate, beta1, beta2, epsilon
        self.sgd = False


class GradientDescentOptimizer:
    """tf.train.GradientDescentOptimizer (train.py:61-62)."""

    def __init__(self, learning_rate=0.01):
        self.lr, self.beta1, self.beta2, self.epsilon = learning_rate, 0.9, 0.999, 1e-4
        self.sgd = True


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def dp_active():
    """True when the step must take the data-parallel path: more than one rank, or MULTINN_DP_REHEARSAL=1 with an initialised
    process group (a 1-rank RCCL rehearsal of exactly the N>1 code path on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("MULTINN_DP_REHEARSAL") == "1"


class CabiComm:
    """The step's collective through the C ABI (mnn_comm_init / mnn_allreduce_flat: RCCL over xGMI), for hosts without torch.distributed's
    collectives -- and the path MULTINN_COMM=capi takes here.  The 128-byte unique id is created by rank 0 and handed to the other ranks by
    `exchange` (default: a torch.distributed object broadcast, which only uses the store / any backend; a host may pass its own function
    rank-0-bytes -> everybody's-bytes)."""

    def __init__(self, rank, world, exchange=None):
        import ctypes as C
        from . import _lib
        self._lib, self.rank, self.world = _lib, rank, world
        buf = C.create_string_buffer(128)
        if rank == 0:
            _lib.call("mnn_comm_unique_id", buf)
        uid = bytes(buf.raw)
        if world > 1:
            if exchange is None:
                box = [uid]
                dist.broadcast_object_list(box, src=0)
                uid = box[0]
            else:
                uid = exchange(uid)
        self._h = C.c_void_p()
        _lib.call("mnn_comm_init", C.byref(self._h), int(rank), int(world), C.create_string_buffer(uid, 128))

    def all_reduce(self, grad):
        import ctypes as C
        assert grad.is_cuda and grad.dtype == torch.float32 and grad.is_contiguous()
        self._lib.call("mnn_allreduce_flat", self._h, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(grad.data_ptr()), grad.numel())
        return grad

    def close(self):
        if self._h:
            self._lib.call("mnn_comm_destroy", self._h)
            self._h = None


_CABI_COMM = None


def setup_cabi_comm():
    """Create the C-ABI communicator NOW (MULTINN_COMM=capi with an initialised process group): ncclCommInitRank and the id broadcast must
    not run inside a hipGraph capture, which is where the first all-reduce of a graphed step would otherwise create it.  Idempotent; the
    communicator is destroyed at interpreter exit."""
    global _CABI_COMM
    if _CABI_COMM is None and dp_active() and os.environ.get("MULTINN_COMM") == "capi" and torch.cuda.is_available():
        import atexit
        _CABI_COMM = CabiComm(dist.get_rank(), dist.get_world_size())
        atexit.register(lambda: _CABI_COMM is not None and _CABI_COMM.close())
    return _CABI_COMM


def allreduce_flat(grad):
    """The single data-parallel exchange of the step: sum the flat gradient over ranks.  torch.distributed (backend "nccl" = RCCL) by
    default; MULTINN_COMM=capi issues the same RCCL all-reduce through the library's own entry points (CabiComm)."""
    global _CABI_COMM
    if dp_active():
        if os.environ.get("MULTINN_COMM") == "capi" and grad.is_cuda:
            if _CABI_COMM is None:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("MULTINN_COMM=capi: call training.setup_cabi_comm() before capturing a step (the communicator cannot be created inside a capture)")
                setup_cabi_comm()
            _CABI_COMM.all_reduce(grad)
        else:
            dist.all_reduce(grad, op=dist.ReduceOp.SUM)
    return grad


def compute_gradients(optimizer, store, clip_norm=5.0, lr=None, reduce=True):
    """clip_by_global_norm(5.0) (hard-coded in the reference, training.py:166; R9 honours the
    argument) + optimizer.apply_gradients.  The gradient must already be in store.grad.
    Returns the device scalar holding sum(grad^2) (global norm squared, after the all-reduce)."""
    if reduce:
        allreduce_flat(store.grad)
    sumsq = torch.zeros(1, device=store.grad.device)
    ops.sumsq(store.grad, sumsq)
    store.step += 1
    # the step count is read from store.step_dev ON DEVICE (= store.step - 1 here), so a captured graph stays valid
    ops.clip_adam_step(store.theta, store.grad, store.m, store.v, sumsq, clip_norm, optimizer.lr if lr is None else lr,
                       optimizer.beta1, optimizer.beta2, optimizer.epsilon, store.step, optimizer.sgd, store.step_dev, store.skipped)
    ops.step_increment(store.step_dev, sumsq, clip_norm, store.ls_dyn, store.ls_good, store.LS_GROW_AFTER)       # a skipped (non-finite) step does not count; the f16 loss scale follows
    return sumsq


def compute_gradients_multi(optimizer, stores, clip_norm=5.0, lr=None, reduce=True):
    """The optimiser step of a mode that trains SEVERAL generators on one loss (multinn_jamming.py:235-241: `compute_gradients(optimizer,
    mean track loss, all generators' + feedback variables)`): ONE global norm over every store's gradient (training.py:166 clips the
    whole variable list together), then the clipped TF-Adam step per store.  Each store's flat gradient is summed over the ranks first
    (data parallel).  Returns the device scalar holding the global sum of squares."""
    if reduce:
        for st in stores:
            allreduce_flat(st.grad)
    sumsq = torch.zeros(1, device=stores[0].grad.device)
    for st in stores:
        ops.sumsq(st.grad, sumsq)                       # accumulates (f32 atomic add into the one word)
    for st in stores:
        st.step += 1
        ops.clip_adam_step(st.theta, st.grad, st.m, st.v, sumsq, clip_norm, optimizer.lr if lr is None else lr,
                           optimizer.beta1, optimizer.beta2, optimizer.epsilon, st.step, optimizer.sgd, st.step_dev, st.skipped)
        ops.step_increment(st.step_dev, sumsq, clip_norm, st.ls_dyn, st.ls_good, st.LS_GROW_AFTER)
    return sumsq
