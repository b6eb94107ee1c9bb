"""Host-side mirror of /root/reference/multinn/models/common: Model, RNN, NADE, RBM, DBN.

Same class names, constructor arguments and method names as the reference, with eager ROCm
tensors in place of TF1 symbolic tensors: a method that built graph ops in the reference runs
the corresponding HIP kernels here (through the C ABI in ``multinn_amd.ops``).

Parameters of a trainable module live in ONE flat float32 buffer (``ParamStore``) together with
its gradient and Adam slots; that flat gradient buffer is also the data-parallel all-reduce
buffer.
"""
import abc
import contextlib
import gc
import math
import os

import torch

from . import ops
from ._lib import MnnError

STREAM_DROPOUT, STREAM_NADE, STREAM_RBM_H, STREAM_RBM_V, STREAM_DBN_ENC, STREAM_DBN_DEC = range(6)


def default_device():
    if not torch.cuda.is_available():
        raise MnnError("multinn_amd needs a ROCm device: there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


# --------------------------------------------------------------------------------------------
@contextlib.contextmanager
def graph_capture(g, **kw):
    """`with torch.cuda.graph(g, **kw)` with the cyclic garbage collector switched off for the duration of the capture.  torch collects once
    when the capture begins; a collection that starts DURING it runs finalisers of whatever cyclic garbage the captured Python code has left
    behind (dead generators with their own graphs, streams, pools) in the middle of the capture -- one full GPU test run of round 6 died that
    way ("Fatal Python error: Aborted", "Garbage-collecting", inside a captured sampling scan).  Reference-counted frees are unaffected."""
    with torch.cuda.graph(g, **kw):
        was = gc.isenabled()
        gc.disable()
        try:
            yield
        finally:
            if was:
                gc.enable()


class ScanGraphs:
    """hipGraph cache for whole sampling scans (rnn_estimator.py:271-323, multinn_feedback.py:120-218).

    A scan is `num_steps` x {sample, LSTM step, Dense}: a dozen small launches per generated step, ~0.5 ms of host time
    against ~0.1 ms of device time when launched eagerly.  The whole scan -- intro pass, weight packing and every
    step, each with its own RNG counter baked into its node -- is captured once per (shape, num_steps, seed) and
    replayed; inputs are copied into the captured buffer, the result is a copy of the captured output.
    MULTINN_GENERATE_GRAPH=0 keeps the eager loop."""

    def __init__(self, max_entries=4):
        import collections
        self._cache = collections.OrderedDict()
        self._max = max_entries

    @staticmethod
    def enabled(x):
        return x.is_cuda and os.environ.get("MULTINN_GENERATE_GRAPH", "1") != "0" and not torch.cuda.is_current_stream_capturing()

    def run(self, key, x, scan, warm, after_capture):
        """scan(static_x) -> output tensor (captured); warm(static_x): a short eager run that creates parameters and
        workspaces before the capture; after_capture(): drop host-side caches that now point into the graph's pool."""
        ent = self._cache.get(key)
        if ent is None:
            static_x = x.clone()
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                warm(static_x)
            cur.wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with graph_capture(g, capture_error_mode="thread_local"):
                out = scan(static_x)
            after_capture()
            ent = (g, static_x, out)
            self._cache[key] = ent
            while len(self._cache) > self._max:
                self._cache.popitem(last=False)
        else:
            self._cache.move_to_end(key)
        g, static_x, out = ent
        static_x.copy_(x)
        g.replay()
        return out.clone()


# --------------------------------------------------------------------------------------------
class ParamStore:
    """Flat f32 parameter / gradient / Adam-slot buffers with named views."""

    def __init__(self, device=None):
        self.device = device
        self._specs = []          # (name, shape, init_fn)
        self.theta = self.grad = self.m = self.v = None
        self.views, self.gviews = {}, {}
        self.step = 0

    def declare(self, name, shape, init):
        assert self.theta is None, "declare() after materialize()"
        self._specs.append((name, tuple(shape), init))

    def materialize(self):
        if self.theta is not None:
            return
        dev = self.device or default_device()
        n = sum(math.prod(s) for _, s, _ in self._specs)
        self.theta = torch.zeros(n, device=dev)
        self.grad = torch.zeros(n, device=dev)
        self.m = torch.zeros(n, device=dev)
        self.v = torch.zeros(n, device=dev)
        self.step_dev = torch.zeros(1, device=dev, dtype=torch.int32)     # device copy of `step` (read by kernels under hipGraph replay)
        # optimiser steps the device skipped because the gradient norm was not finite (mnn_clip_adam_step; f16 loss-scale overflow)
        self.skipped = torch.zeros(1, device=dev, dtype=torch.int32)
        # the dynamic part of the f16 loss scale, [m, 1 / m] (+ applied steps since its last change): a skipped step halves m, LS_GROW_AFTER
        # applied steps in a row double it back up to 1 (mnn_step_increment); the backward passes of precision "fp16" multiply their gradient
        # seed by m and the finished gradient by 1 / m, both read on the device
        self.ls_dyn = torch.ones(2, device=dev)
        self.ls_good = torch.zeros(1, device=dev, dtype=torch.int32)
        off = 0
        for name, shape, init in self._specs:
            k = math.prod(shape)
            self.views[name] = self.theta[off:off + k].view(shape)
            self.gviews[name] = self.grad[off:off + k].view(shape)
            if init is not None:
                self.views[name].copy_(init(shape).to(dev))
            off += k

    def __getitem__(self, name):
        return self.views[name]

    def offset(self, name):
        """First word of variable `name` in the flat buffers (declaration order)."""
        off = 0
        for n, shape, _ in self._specs:
            if n == name:
                return off
            off += math.prod(shape)
        raise KeyError(name)

    def names(self):
        return [n for n, _, _ in self._specs]

    LS_GROW_AFTER = 200

    def check(self, tolerate_overflow=False):
        """Raise FloatingPointError if an optimiser step has been skipped on the device since the last check (non-finite gradient norm:
        the f16 backward pass overflowed, or a NaN reached the gradient).  Synchronises; clears the counter.
        tolerate_overflow (a training loop in precision "fp16"): skipped steps are what the dynamic loss scale feeds on -- warn, and raise only
        when the scale has been halved down to its floor without finding a finite gradient (that is a NaN, not an overflow)."""
        if self.theta is None:
            return
        n = int(self.skipped.item())
        if n:
            self.skipped.zero_()
            self.step = int(self.step_dev.item())       # the host mirror counts attempts; the device counter only APPLIED steps
            m = float(self.ls_dyn[0]) if getattr(self, "ls_dyn", None) is not None else 1.0
            if tolerate_overflow and m > 2.0 ** -19:
                import warnings
                warnings.warn(f"{n} optimiser step(s) skipped (non-finite gradient norm: f16 overflow); the loss scale multiplier is now {m:g}")
                return
            raise FloatingPointError(f"{n} optimiser step(s) skipped: the gradient norm was not finite (f16 loss-scale overflow or NaN "
                                     f"gradient; dynamic loss-scale multiplier {m:g}); lower LstmStack.loss_scale_rows or use precision='bf16'")

    def state_dict(self):
        return dict(theta=self.theta.cpu(), m=self.m.cpu(), v=self.v.cpu(), step=int(self.step_dev.item()), names=self.names(),
                    shapes=[s for _, s, _ in self._specs], ls_dyn=self.ls_dyn.cpu(), ls_good=int(self.ls_good.item()))

    def load_state_dict(self, sd):
        if list(sd["names"]) != self.names() or [tuple(s) for s in sd["shapes"]] != [s for _, s, _ in self._specs]:
            raise ValueError("checkpoint does not match the model's variables")
        self.theta.copy_(sd["theta"]); self.m.copy_(sd["m"]); self.v.copy_(sd["v"])
        self.step = int(sd["step"])
        self.step_dev.fill_(self.step)
        if "ls_dyn" in sd:                               # (checkpoints of earlier builds: the dynamic loss scale restarts at 1)
            self.ls_dyn.copy_(sd["ls_dyn"])
            self.ls_good.fill_(int(sd.get("ls_good", 0)))


def glorot_uniform(gen, fan_in, fan_out):
    """tf.contrib.layers.xavier_initializer() / glorot_uniform (rbm.py:36, rnn_nade.py:56)."""
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return lambda shape: (torch.rand(shape, generator=gen) * 2 - 1) * lim


def truncated_normal(gen, std):
    """tf.truncated_normal_initializer(stddev) (nade.py:49-50): resample beyond two sigma."""
    def f(shape):
        x = torch.randn(shape, generator=gen)
        bad = x.abs() > 2
        while bool(bad.any()):
            x[bad] = torch.randn(int(bad.sum()), generator=gen)
            bad = x.abs() > 2
        return x * std
    return f


def zeros_init(shape):
    return torch.zeros(shape)


# --------------------------------------------------------------------------------------------
class Model(abc.ABC):
    """models/common/model.py:9-234."""

    # store.step the 16-bit packed copies were made at; -1 = stale.  Every "stale" mark also starts a new pack epoch: the f32 repacks of the
    # deterministic sampling steps (LstmStack.det_job) are redone once per epoch -- i.e. once per scan, inside its captured graph
    @property
    def _packed_step(self):
        return self.__dict__.get("_packed_step_v", -1)

    @_packed_step.setter
    def _packed_step(self, v):
        self.__dict__["_packed_step_v"] = v
        if v == -1:
            self.__dict__["_pack_epoch"] = self.__dict__.get("_pack_epoch", 0) + 1

    def __init__(self, name="model"):
        self._name = name
        self._is_built = False
        self._variables = {}
        self._trainable_variables = []
        self._placeholders = {}
        self._metrics, self._metrics_upd = None, None
        self._summaries = {"weights": None, "metrics": None, "gradients": None}
        self.store = None

    name = property(lambda self: self._name)
    is_built = property(lambda self: self._is_built)
    variables = property(lambda self: self._variables)
    trainable_variables = property(lambda self: self._trainable_variables)
    placeholders = property(lambda self: self._placeholders)
    metrics = property(lambda self: self._metrics)
    metrics_upd = property(lambda self: self._metrics_upd)
    summaries = property(lambda self: self._summaries)

    @property
    def weight_summary(self):
        """model.py:104-107 (TensorBoard histograms) -> per-variable (mean, std) on the host."""
        return {k: (float(v.mean()), float(v.std())) for k, v in self._flat_named().items()}

    def _flat_named(self):
        return dict(self.store.views) if self.store is not None else {}

    def build(self, x=None, y=None, lengths=None, is_train=None, mode="eval"):
        if mode not in ("train", "eval", "generate"):        # model.py:146-149
            raise ValueError("`mode` must be one of: 'train', 'eval', 'generate'.")

    @abc.abstractmethod
    def build_metrics(self, targets, predictions, cond_probs=None, log_probs=None):
        ...

    def save(self, sess=None, ckpt_dir=None, global_step=None, write_meta_graph=False):
        """model.py:180-207.  torch.save of the flat state (optimizer slots included, R12)."""
        os.makedirs(ckpt_dir, exist_ok=True)
        path = os.path.join(ckpt_dir, f"{self.name}.pt")
        torch.save(self.store.state_dict(), path)
        return path

    def load(self, sess=None, ckpt_dir=None):
        """model.py:209-234: returns False when no checkpoint exists."""
        path = os.path.join(ckpt_dir, f"{self.name}.pt")
        if not os.path.exists(path):
            return False
        self.store.load_state_dict(torch.load(path))
        return True


# --------------------------------------------------------------------------------------------
class RNN(Model):
    """models/common/rnn.py: MultiRNNCell of DropoutWrapper(CudnnCompatibleLSTMCell)."""

    def __init__(self, num_units=128, keep_prob=1.0, attn_length=0, learn_zero_state=False, name="rnn"):
        super().__init__(name=name)
        if isinstance(num_units, int):
            num_units = [num_units]
        if attn_length or learn_zero_state:
            raise NotImplementedError("attn_length / learn_zero_state are not on the hot path (rnn.py:127-145)")
        for u in num_units:
            if u % 32:
                raise ValueError("LSTM units must be a multiple of 32 (gate-interleaved MFMA layout)")
        self._num_units = list(num_units)
        self._keep_prob = keep_prob
        self._is_train = False

    num_units = property(lambda self: self._num_units)
    num_layers = property(lambda self: len(self._num_units))
    keep_prob = property(lambda self: self._keep_prob)

    def declare(self, store, n_in, gen, prefix="rnn"):
        """Kernel [in+u, 4u] glorot-uniform, bias zeros (LSTMBlockCell defaults)."""
        self.n_in = n_in
        self.prefix = prefix
        for l, u in enumerate(self._num_units):
            store.declare(f"{prefix}/cell_{l}/kernel", (n_in + u, 4 * u), glorot_uniform(gen, n_in + u, 4 * u))
            store.declare(f"{prefix}/cell_{l}/bias", (4 * u,), zeros_init)
            n_in = u
        self.store = store

    def layer_inputs(self):
        return [self.n_in] + self._num_units[:-1]

    def build_cell(self, is_train):
        self._is_train = bool(is_train) if is_train is not None else True
        self._is_built = True

    def effective_keep_prob(self):
        return self._keep_prob if self._is_train else 1.0      # rnn.py:117-120

    def build_metrics(self, targets, predictions, cond_probs=None, log_probs=None):
        return [], [], None

    def zero_state(self, batch_size, dtype=torch.float32):
        """rnn.py:155-176: tuple of (c, h) per layer."""
        dev = self.store.theta.device
        return tuple((torch.zeros((batch_size, u), device=dev), torch.zeros((batch_size, u), device=dev, dtype=dtype))
                     for u in self._num_units)

    def __call__(self, inputs, state, *args, **kwargs):
        if not self._is_built:
            raise RuntimeError("RNN cell is not built yet, build it with `cell.build_cell()` before calling")
        raise NotImplementedError("single-step calls go through RnnEstimator.single_step")


# --------------------------------------------------------------------------------------------
class NADE(Model):
    """models/common/nade.py.  w_enc/w_dec are stored [D, Hn] (reference [D,1,Hn] / [D,Hn,1])."""

    def __init__(self, num_dims, num_hidden=128, internal_bias=False, name="nade"):
        super().__init__(name=name)
        if internal_bias:
            raise NotImplementedError("a standalone NADE with internal biases is not built: RnnNade / RnnMultiNADE(internal_bias=True) fold "
                                      "them into the Dense bias (generators.py), which is the only place the reference uses them (rnn_nade.py:245-251)")
        if num_hidden > 256:
            raise ValueError("NADE hidden units > 256 are not supported by the HIP scan kernels")
        self._num_dims, self._num_hidden, self._internal_bias = num_dims, num_hidden, internal_bias
        self._is_built = True

    num_dims = property(lambda self: self._num_dims)
    num_hidden = property(lambda self: self._num_hidden)
    internal_bias = property(lambda self: self._internal_bias)

    def declare(self, store, gen, prefix="nade"):
        std = 1.0 / math.sqrt(self._num_dims)
        store.declare(f"{prefix}/w_enc", (self._num_dims, self._num_hidden), truncated_normal(gen, std))
        store.declare(f"{prefix}/w_dec", (self._num_dims, self._num_hidden), truncated_normal(gen, std))
        self.store, self.prefix = store, prefix

    _w_enc_t = _w_dec_t = None      # set when the weights are views of a generator's stacked store

    @property
    def w_enc(self):
        return self._w_enc_t if self._w_enc_t is not None else self.store[f"{self.prefix}/w_enc"]

    @property
    def w_dec(self):
        return self._w_dec_t if self._w_dec_t is not None else self.store[f"{self.prefix}/w_dec"]

    def build_metrics(self, targets, predictions, cond_probs=None, log_probs=None):
        from .metrics import base_metrics
        return base_metrics(log_probs, targets, predictions, log_probs)

    def log_prob(self, x, b_enc=None, b_dec=None):
        """nade.py:155-229.  x u8/float [N,D]; returns (nll [N], cond_p [N,D])."""
        if b_enc is None or b_dec is None:
            raise ValueError("Bias values should be provided when `internal_bias` is `False`")
        N = x.shape[0]
        if b_enc.shape[0] == 1 != N:
            b_enc = b_enc.expand(N, -1)
        if b_dec.shape[0] == 1 != N:
            b_dec = b_dec.expand(N, -1)
        bias = torch.cat([b_enc, b_dec], 1).contiguous()
        nll = torch.empty(N, device=bias.device)
        cp = torch.empty((N, self._num_dims), device=bias.device)
        ops.nade_logprob_fwd(x.to(torch.uint8).contiguous(), bias, self.w_enc, self.w_dec, 1, self._num_dims, self._num_hidden,
                             None, nll, cp)
        return nll, cp

    def sample(self, b_enc=None, b_dec=None, n=None, temperature=None, seed=0, row0=0, sub=0):
        """nade.py:231-308.  Returns (samples u8 [N,D], nll [N])."""
        if b_enc is None or b_dec is None:
            raise ValueError("Bias values should be provided when `internal_bias` is `False`")
        N = n or b_enc.shape[0]
        bias = torch.cat([b_enc.expand(N, -1), b_dec.expand(N, -1)], 1).contiguous()
        out = torch.empty((N, self._num_dims), device=bias.device, dtype=torch.uint8)
        nll = torch.empty(N, device=bias.device)
        ops.nade_sample(bias, self.w_enc, self.w_dec, 1, self._num_dims, self._num_hidden, temperature, seed, row0, sub, out, nll=nll)
        return out, nll


# --------------------------------------------------------------------------------------------
class RBM(Model):
    """models/common/rbm.py."""

    def __init__(self, num_dims, num_hidden=128, k=10, name="rbm"):
        super().__init__(name=name)
        self._num_dims, self._num_hidden, self._k = num_dims, num_hidden, k
        self._is_built = True
        self.seed = 0

    num_dims = property(lambda self: self._num_dims)
    num_hidden = property(lambda self: self._num_hidden)
    k = property(lambda self: self._k)

    def declare(self, store, gen, prefix="rbm"):
        store.declare(f"{prefix}/W", (self._num_dims, self._num_hidden), glorot_uniform(gen, self._num_dims, self._num_hidden))
        store.declare(f"{prefix}/bv", (1, self._num_dims), zeros_init)      # rbm.py:62: [W, bv, bh]
        store.declare(f"{prefix}/bh", (1, self._num_hidden), zeros_init)
        self.store, self.prefix = store, prefix

    W = property(lambda self: self.store[f"{self.prefix}/W"])
    bh = property(lambda self: self.store[f"{self.prefix}/bh"])
    bv = property(lambda self: self.store[f"{self.prefix}/bv"])

    def build_metrics(self, targets, predictions, cond_probs=None, log_probs=None):
        """rbm.py:96-146 with per-row free energy (R4); mean([N,N]) == mean_n, so the scalars match."""
        from .metrics import base_metrics
        cost, free_energy = self.free_energy_cost(targets, predictions)
        if log_probs is None and cond_probs is not None:
            from .encoders import reconstruction_cost
            log_probs = reconstruction_cost(targets, cond_probs)   # tf.losses.log_loss summed over the visibles
        else:
            raise ValueError("Incorrect arguments. Either `cond_probs`, or `log_probs` should be provided on `rbm.build_metrics()` function call")
        metrics, upd, summ = base_metrics(cost, targets, predictions, log_probs)
        metrics["free_energy"] = free_energy.mean()
        return metrics, upd, summ

    def forward(self, v, bh=None, seed=None, row0=0, sub=0, stream=STREAM_RBM_H):
        """rbm.py:148-167 -> (p_h f32 [N,Hn], h u8 [N,Hn])."""
        bh = bh if bh is not None else self.bh
        N = v.shape[0]
        p = torch.empty((N, self._num_hidden), device=v.device)
        h = torch.empty((N, self._num_hidden), device=v.device, dtype=torch.uint8)
        vv = v if v.dtype in (torch.uint8, torch.float32) else v.float()
        ops.rbm_hidden(vv.contiguous(), self.W, bh, stream, self.seed if seed is None else seed, row0, sub, p, h)
        return p, h

    def reconstruct(self, h, bv=None, seed=None, row0=0, sub=0, stream=STREAM_RBM_V):
        """rbm.py:169-190 -> (p_v f32 [N,D], v u8 [N,D])."""
        bv = bv if bv is not None else self.bv
        N = h.shape[0]
        p = torch.empty((N, self._num_dims), device=h.device)
        v = torch.empty((N, self._num_dims), device=h.device, dtype=torch.uint8)
        hh = h if h.dtype in (torch.uint8, torch.float32) else h.float()
        ops.rbm_visible(hh.contiguous(), self.W, bv, stream, self.seed if seed is None else seed, row0, sub, p, v)
        return p, v

    def sample(self, v, bh=None, bv=None, k=None, seed=None, row0=0, row_ids=None, sub0=0):
        """rbm.py:192-231; k=None -> self.k (R1).  Returns (p_v, v_sample u8)."""
        k = self._k if k is None else k
        bh = bh if bh is not None else self.bh
        bv = bv if bv is not None else self.bv
        N, D = v.shape
        p_v = torch.empty((N, D), device=v.device)
        v_s = torch.empty((N, D), device=v.device, dtype=torch.uint8)
        ops.rbm_gibbs(v.to(torch.uint8).contiguous(), self.W, bh, bv, k, self.seed if seed is None else seed, row0, row_ids, sub0, p_v, v_s)
        return p_v, v_s

    def free_energy(self, v, bh=None, bv=None):
        bh = bh if bh is not None else self.bh
        bv = bv if bv is not None else self.bv
        F = torch.empty(v.shape[0], device=v.device)
        return ops.rbm_free_energy(v.to(torch.uint8).contiguous(), self.W, bh, bv, F)

    def free_energy_cost(self, v, v_sample, bh=None, bv=None):
        """rbm.py:233-263 (per row): returns (cost [N], free_energy [N])."""
        Fv = self.free_energy(v, bh, bv)
        return Fv - self.free_energy(v_sample, bh, bv), Fv

    def visible_bias_init_ops(self, v):
        """rbm.py:286-297: returns the (not yet executed) init ops -- call each to assign bv = log(1e-6 + p/(1-p)), p the mean
        activation of every visible unit over the batch (over ALL ranks' batches under data parallelism)."""
        from .training import dp_active
        import torch.distributed as dist

        def assign_bv():
            N, D = v.shape
            vf = torch.empty((N, D), device=v.device)
            ops.convert2d(v if v.dtype in (torch.uint8, torch.float32) else v.float(), vf)
            stat = torch.zeros(D + 1, device=v.device)         # [column sums | row count]: one buffer, one all-reduce
            ops.bias_grad(vf, stat[:D], accumulate=True)
            n_tot = float(N)
            if dp_active():
                ops.fill(stat[D:], float(N))
                dist.all_reduce(stat)
                n_tot = float(stat[D])
            ops.rbm_visible_bias_init(stat[:D], n_tot, self.bv.view(-1))
        return [assign_bv]

    def _flat_delta(self):
        """[W | bv | bh] are consecutive in the store (declare()): their flat slice and a delta buffer of the same layout."""
        W, bh = self.W, self.bh
        off = W.storage_offset() - self.store.theta.storage_offset()
        n = self._num_dims * self._num_hidden + self._num_dims + self._num_hidden
        assert bh.storage_offset() - W.storage_offset() == self._num_dims * self._num_hidden + self._num_dims
        return self.store.theta[off:off + n]

    def _cd_update(self, v, lr, seed=None, row0=0, sub0=0):
        """rbm.py:299-335: returns (update_ops, [dW, dbv, dbh]) and applies the deltas (assign_add).  Under data parallelism the flat
        delta [dW | dbv | dbh] is summed over the ranks (ONE all-reduce) and N is the global row count, so every rank
        applies the full-batch update."""
        from .training import dp_active
        import torch.distributed as dist
        seed = self.seed if seed is None else seed
        N = v.shape[0]
        D, Hn = self._num_dims, self._num_hidden
        vu = v.to(torch.uint8).contiguous()
        p_v_s, v_s = self.sample(vu, self.bh, self.bv, self._k, seed, row0, None, sub0)
        _, h = self.forward(vu, None, seed, row0, sub0 + self._k)
        p_h_s, _ = self.forward(v_s, None, seed, row0, sub0 + self._k + 1)
        n_tot = N
        if dp_active():
            cnt = torch.full((1,), float(N), device=v.device)
            dist.all_reduce(cnt)
            n_tot = float(cnt)
        lrn = lr / n_tot
        # outer-product sums on MFMA: [D,N] x [N,Hn]
        Np = ops.round_up(N, 4)
        def t_(x, rows):
            return ops.transpose(x, torch.zeros((rows, Np), device=v.device))
        vT, hT = t_(vu, D), t_(h, Hn)
        pvT, phT = t_(p_v_s, D), t_(p_h_s, Hn)
        delta = torch.zeros(D * Hn + D + Hn, device=v.device)
        dW, dbv, dbh = delta[:D * Hn].view(D, Hn), delta[D * Hn:D * Hn + D].view(1, D), delta[D * Hn + D:].view(1, Hn)
        neg = torch.empty((D, Hn), device=v.device)
        ops.gemm_tn(vT, hT, dW)
        ops.gemm_tn(pvT, phT, neg)
        ops.axpby(lrn, dW.view(-1), -lrn, neg.view(-1), dW.view(-1))
        ops.rbm_cd_bias_delta(vu, p_v_s, h, p_h_s, lrn, dbv.view(-1), dbh.view(-1))
        if dp_active():
            dist.all_reduce(delta)
        theta = self._flat_delta()
        ops.axpby(1.0, theta, 1.0, delta, theta)            # assign_add of W, bv, bh (rbm.py:329-333)
        return [], [dW, dbv, dbh]

    def train(self, v, lr, **kw):
        """rbm.py:265-284."""
        init_ops = self.visible_bias_init_ops(v)
        update_ops, gradients = self._cd_update(v, lr, **kw)
        return init_ops, update_ops, gradients


# --------------------------------------------------------------------------------------------
class DBN(Model):
    """models/common/dbn.py: stacked RBMs; forward feeds SAMPLED codes upward (dbn.py:136-157)."""

    def __init__(self, num_dims, num_hidden, k=10, name="dbn", seed=23, device=None):
        super().__init__(name=name)
        if isinstance(num_hidden, int):
            num_hidden = [num_hidden]
        self._num_dims, self._num_hidden = num_dims, list(num_hidden)
        gen = torch.Generator().manual_seed(seed)
        self.store = ParamStore(device)
        self._rbms = []
        n_in = num_dims
        for i, nh in enumerate(self._num_hidden):
            r = RBM(n_in, nh, k=k, name=f"{name}/rbm_{i}")
            r.declare(self.store, gen, prefix=f"{name}/rbm_{i}")
            self._rbms.append(r)
            n_in = nh
        self.store.materialize()
        self._is_built = True

    num_dims = property(lambda self: self._num_dims)
    num_hidden = property(lambda self: self._num_hidden)
    num_layers = property(lambda self: len(self._num_hidden))
    rbms = property(lambda self: self._rbms)
    rbm_layers = property(lambda self: self._rbms)          # the reference's name (dbn.py:60)

    def build_metrics(self, targets, predictions, cond_probs=None, log_probs=None):
        return self._rbms[0].build_metrics(targets, predictions, cond_probs, log_probs)

    def forward(self, x, seed=0, row0=0, sub=0):
        """dbn.py:136-157 -> (p_h of the top layer, sampled h u8)."""
        h, p = x, None
        for i, r in enumerate(self._rbms):
            p, h = r.forward(h, None, seed, row0, (sub << 4) | i, STREAM_DBN_ENC)
        return p, h

    def reconstruct(self, h, seed=0, row0=0, sub=0):
        """dbn.py:159-180 -> (p_v of the bottom layer, sampled v u8)."""
        v, p = h, None
        for i in range(len(self._rbms) - 1, -1, -1):
            p, v = self._rbms[i].reconstruct(v, None, seed, row0, (sub << 4) | i, STREAM_DBN_DEC)
        return p, v
