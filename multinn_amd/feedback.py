"""Feedback-RNN sampling scan: multinn_feedback.py:120-218 with the recurrent feedback
module of multinn_feedback_rnn.py:41-79.  Composition of the per-track RnnNade generators and one more
LSTM stack, all running through the same C-ABI kernels."""
import weakref

import torch

from . import ops
from .common import Model, RNN, ParamStore
from .generators import LstmStack, RnnEstimator, _compute_dtype, det_steps


class FeedbackRnn(Model):
    """The Feedback module: RNN(num_units) over the stacked per-track codes (multinn_feedback_rnn.py:30-79)."""

    def __init__(self, num_inputs, num_units, keep_prob=1.0, precision="bf16", seed=23, device=None, name="feedback"):
        super().__init__(name=name)
        self.dtype = _compute_dtype(precision)
        self._rnn = RNN(num_units=num_units, keep_prob=keep_prob)
        self.store = ParamStore(device)
        self._rnn.declare(self.store, num_inputs, torch.Generator().manual_seed(seed), prefix="feedback/rnn")
        self.store.materialize()
        self._stack = LstmStack(self._rnn, self.store, self.dtype)
        self._stack.owner = weakref.ref(self)            # (its pack epoch dates the stack's repacked sampling weights: LstmStack.det_job; weak: no cycle)
        self._rnn.build_cell(False)
        self.num_inputs, self.num_units = num_inputs, list(self._rnn.num_units)
        self.seed, self.row0, self._ctx = seed, 0, None
        self._is_built = True

    det_sampling = RnnEstimator.det_sampling            # the sampling scans' deterministic f32 arithmetic (see RnnEstimator)

    def build_metrics(self, targets, predictions, cond_probs=None, log_probs=None):
        return [], [], None

    def _tm(self, x):
        B, T, Din = x.shape
        out = torch.zeros((T, B, self._stack.ld0), device=x.device, dtype=self.dtype)
        out[:, :, :Din] = x.transpose(0, 1).to(self.dtype)
        return out

    def run(self, x, initial_state=None, train=False):
        """_apply_feedback(single_step=False): x [B,T,Din] -> (outputs [B,T,F] f32, final_state).  train=True keeps what backward() needs
        and applies the module's output dropout (multinn_feedback_rnn.py:56-57: built with is_train)."""
        if not train and self.det_sampling:
            # sampling: deterministic f32 steps on the master weights (generators.RnnEstimator.det_sampling), one per time step
            xs = x if x.dtype in (torch.uint8, torch.float32) else x.float()
            st = initial_state
            outs = []
            for t in range(xs.shape[1]):
                h, st = self._stack.det_step(xs[:, t], st)
                outs.append(h)
            return torch.stack(outs, 1), st
        self._stack.pack()
        self._rnn.build_cell(bool(train))
        kp = self._rnn.effective_keep_prob() if train else 1.0
        y, ctx, final = self._stack.forward(self._tm(x), kp, self.seed, self.row0, save=bool(train), state0=initial_state,
                                            step_dev=self.store.step_dev)
        self._ctx = dict(lstm=ctx, kp=kp) if train else None
        return y.transpose(0, 1).float(), [(c.clone(), h.clone()) for c, h in final]

    def backward(self, d_out, n_valid=None):
        """d_out f32 [B,T,F]: gradient wrt the feedback vectors -> the module's kernel / bias gradients (its inputs are codes: no gradient).
        n_valid: valid rows the generators' mean-over-rows losses divide by (all ranks; default B*T) -- sizes the f16 loss scale."""
        if self._ctx is None:
            raise RuntimeError("FeedbackRnn.backward: run(..., train=True) first")
        self.store.grad.zero_()
        dy = d_out.transpose(0, 1).contiguous()
        # f16 operands: the pass runs on loss-scaled values (LstmStack.loss_scale); d_out comes from mean-over-rows losses, ~1/(B T) per element
        # (times the dynamic multiplier of this store's f16 loss scale, a device word: ParamStore.ls_dyn)
        ls = self._stack.loss_scale(n_valid if n_valid else dy.shape[0] * dy.shape[1])
        if ls != 1.0:
            dy = dy * (self.store.ls_dyn[0:1] * ls)
        self._stack.backward(dy, self._ctx["lstm"], self._ctx["kp"], self.seed, self.row0, step_dev=self.store.step_dev)
        if ls != 1.0:
            self.store.grad.mul_(self.store.ls_dyn[1:2] * (1.0 / ls))

    def single(self, x, state):
        """_apply_feedback(single_step=True): x [B,Din] -> (output [B,F] f32, new_state)."""
        if self.det_sampling:
            return self._stack.det_step(x if x.dtype in (torch.uint8, torch.float32) else x.float(), state)
        xin = torch.zeros((x.shape[0], self._stack.ld0), device=x.device, dtype=self.dtype)
        ops.convert2d(x.contiguous(), xin[:, :x.shape[1]])
        h, new = self._stack.single_step(xin, state)
        return h.float(), [(c.clone(), hh.clone()) for c, hh in new]


class DNN(Model):
    """models/common/dnn.py:16-138: a stack of Dense layers with one activation (sigmoid in the reference's only use,
    multinn_feedback.py:48-52), Xavier-uniform kernels, zero biases.  Each layer = sigmoid(x . W + b) runs as
    `mnn_rbm_hidden` (f32, deterministic summation order: the same kernel as the RBM hidden pass)."""

    def __init__(self, num_units=128, num_inputs=None, seed=23, device=None, name="dnn"):
        super().__init__(name=name)
        self._num_units = [num_units] if isinstance(num_units, int) else list(num_units)
        self.store = ParamStore(device)
        self._seed = seed
        self._is_built = False
        if num_inputs is not None:
            self._materialize(num_inputs)

    num_units = property(lambda self: self._num_units)
    num_layers = property(lambda self: len(self._num_units))

    def _materialize(self, num_inputs):
        if self.store.theta is not None:
            return
        from .common import glorot_uniform, zeros_init
        gen = torch.Generator().manual_seed(self._seed)
        n_in = num_inputs
        for l, u in enumerate(self._num_units):
            self.store.declare(f"{self.name}/dense_{l}/kernel", (n_in, u), glorot_uniform(gen, n_in, u))
            self.store.declare(f"{self.name}/dense_{l}/bias", (u,), zeros_init)
            n_in = u
        self.store.materialize()
        self._is_built = True

    def build(self, x=None, y=None, lengths=None, is_train=None, mode="eval"):
        if x is not None:
            self._materialize(x.shape[-1])
        self._is_built = True

    def build_metrics(self, targets, predictions, cond_probs=None, log_probs=None):
        return [], [], None                                  # dnn.py:96-110: a base block without metrics of its own

    def __call__(self, x, save=False):
        """x [..., n_in] (any float / u8 tensor) -> [..., units[-1]] f32.  save=True keeps every layer's input and output for backward()."""
        self._materialize(x.shape[-1])
        h = x.reshape(-1, x.shape[-1]).float().contiguous()
        acts = [h]
        for l, u in enumerate(self._num_units):
            out = torch.empty((h.shape[0], u), device=h.device)
            ops.rbm_hidden(h, self.store[f"{self.name}/dense_{l}/kernel"], self.store[f"{self.name}/dense_{l}/bias"].view(1, u), 0, 0, 0, 0, p_h=out)
            h = out
            acts.append(h)
        self._acts = acts if save else None
        return h.reshape(x.shape[:-1] + (self._num_units[-1],))

    def backward(self, d_out):
        """d_out f32 [..., units[-1]]: gradient wrt the outputs of the last call(save=True) -> kernel / bias gradients of every layer
        (dW = x^T dz, db = sum dz, dz = dy y (1 - y), dy_below = dz W^T: dnn.py:60-76 by hand).  No gradient wrt the module's inputs (codes)."""
        if getattr(self, "_acts", None) is None:
            raise RuntimeError("DNN.backward: call the module with save=True first")
        g = self.store.gviews
        self.store.grad.zero_()
        dy = d_out.reshape(-1, d_out.shape[-1]).float().contiguous()
        N = dy.shape[0]
        Np = ops.round_up(N, 4)

        def tr(xm):
            o = torch.zeros((xm.shape[1], Np), device=xm.device)
            return ops.transpose(xm, o)
        for l in range(len(self._num_units) - 1, -1, -1):
            x_in, y = self._acts[l], self._acts[l + 1]
            dz = ops.sigmoid_grad(dy, y, torch.empty_like(y))
            sk = int(max(1, min(256, Np // 256)))
            ops.gemm_tn(tr(x_in), tr(dz), g[f"{self.name}/dense_{l}/kernel"], accumulate=True, split_k=sk)      # [n_in, N] . [N, u]
            ops.bias_grad(dz, g[f"{self.name}/dense_{l}/bias"], accumulate=True)
            if l > 0:
                W = self.store[f"{self.name}/dense_{l}/kernel"]                 # [n_in, u]: K = u contiguous, the B operand as stored
                dy = torch.empty((N, W.shape[0]), device=dz.device)
                ops.gemm_tn(dz, W, dy)


class FeedbackDnn(Model):
    """The Dense Feedback module of the Feedback MultINN (multinn_feedback.py:46-52, 103-123): a DNN over the stacked per-track
    codes, applied to every time step independently; it carries no state (`_apply_feedback` returns zeros(1))."""

    def __init__(self, num_inputs, num_units, seed=23, device=None, name="feedback"):
        super().__init__(name=name)
        self._dnn = DNN(num_units, num_inputs, seed=seed, device=device, name=f"{name}/dnn")
        self.store = self._dnn.store
        self.num_inputs, self.num_units = num_inputs, list(self._dnn.num_units)
        self._is_built = True

    def build_metrics(self, targets, predictions, cond_probs=None, log_probs=None):
        return [], [], None

    def run(self, x, initial_state=None, train=False):
        """_apply_feedback(single_step=False): x [B,T,Din] -> (outputs [B,T,F] f32, state placeholder)."""
        return self._dnn(x, save=bool(train)), None

    def backward(self, d_out, n_valid=None):
        """d_out f32 [B,T,F]: gradient wrt the feedback vectors -> the Dense layers' gradients (f32 arithmetic: n_valid is unused)."""
        self._dnn.backward(d_out)

    def single(self, x, state):
        """_apply_feedback(single_step=True): x [B,Din] -> (output [B,F] f32, state placeholder)."""
        return self._dnn(x), None


class FeedbackRnnSampler:
    """MultINNFeedback.generate (multinn_feedback.py:120-173) with PassEncoders: M per-track generators whose
    inputs are concat(track code, feedback vector)."""

    def __init__(self, generators, feedback):
        self.generators, self.feedback = generators, feedback
        self.num_tracks = len(generators)
        self.concurrent = True       # the M generators of a step run on M streams (parallel branches of the captured scan)

    def generate(self, x_u8, num_steps):
        """x_u8 [B,Ti,P,M] intro piano-rolls -> samples u8 [B,num_steps,P,M].  One hipGraph replay per call on the
        device (common.ScanGraphs): the M generators and the feedback module step inside the same captured scan."""
        from .common import ScanGraphs
        if not ScanGraphs.enabled(x_u8):
            return self._generate_scan(x_u8, num_steps)
        if getattr(self, "_scan_graphs", None) is None:
            self._scan_graphs = ScanGraphs()
        key = (tuple(x_u8.shape), int(num_steps), tuple((g.seed, g.row0) for g in self.generators))

        def stale():
            for g in self.generators:
                g._packed_step = -1
            if hasattr(self.feedback, "_packed_step"):
                self.feedback._packed_step = -1

        def scan(sx):
            stale()                                    # pack inside the graph: a replay always sees the current weights
            return self._generate_scan(sx, num_steps)

        return self._scan_graphs.run(key, x_u8, scan, lambda sx: self._generate_scan(sx, min(int(num_steps), 2)), stale)

    def _group_dense(self, hs, rnn_states):
        """The M generators' Dense layers on their top outputs in one launch -> their RnnEstimatorStateTuples."""
        jobs, outs = zip(*[g._det_dense_job(h) for g, h in zip(self.generators, hs)])
        ops.dense_det(list(jobs))
        return [g._state_from_dense(o, tuple(st)) for g, o, st in zip(self.generators, outs, rnn_states)]

    def _generate_scan(self, x_u8, num_steps):
        B, Ti, P, M = x_u8.shape
        assert M == self.num_tracks
        enc = torch.cat([torch.zeros((B, 1, P, M), device=x_u8.device, dtype=torch.uint8), x_u8], 1)      # multi_encoder_nn.py:73-76
        return self.generate_encoded([enc[..., i] for i in range(M)], num_steps)

    def generate_encoded(self, enc_tracks, num_steps):
        """The scan on per-track ENCODED inputs (multinn_feedback.py:120-173 between the encoders): enc_tracks = M x u8
        [B, Ti+1, E] (zero first step included) -> sampled codes u8 [B, num_steps, E, M]; the caller decodes them through
        its encoders (identity for PassEncoder)."""
        M = self.num_tracks
        enc_tracks = [e.contiguous() for e in enc_tracks]            # (views of a [B, T, P, M] roll have inner stride M)
        B, _, P = enc_tracks[0].shape
        dev = enc_tracks[0].device
        x_u8 = enc_tracks[0]
        stack = torch.stack(enc_tracks, 3).reshape(B, -1, P * M)                                    # feature p*M+m (stack axis 3 + reshape)
        x_fb, fb_state = self.feedback.run(stack)
        det = all(getattr(g, "det_sampling", False) for g in self.generators)
        # deterministic arithmetic: the M generators' LSTM steps of a time step run as ONE launch per layer, their Dense layers as one more
        # (ops.lstm_step_det / dense_det take up to 8 jobs) -- the tracks are independent inside a step
        group = det and all(hasattr(g, "_det_dense_job") for g in self.generators) and \
            len({len(g._rnn.num_units) for g in self.generators}) == 1 and M <= 8
        states = []
        for i, g in enumerate(self.generators):
            g._materialize(P + x_fb.shape[-1])
            g._rnn.build_cell(False)
            if not det:
                g._ensure_packed()
        if group:
            sts = [None] * M
            for t in range(stack.shape[1]):
                res = det_steps([g._stack for g in self.generators], [e[:, t] for e in enc_tracks], sts, [x_fb[:, t]] * M)
                sts = [r[1] for r in res]
            states = self._group_dense([r[0] for r in res], sts)
        else:
            for i, g in enumerate(self.generators):
                states.append(g.steps(torch.cat([enc_tracks[i].float(), x_fb], -1)))                # multinn_feedback.py:143-149
        out = torch.empty((B, num_steps, P, M), device=dev, dtype=torch.uint8)
        # Inside a step the tracks are independent: generator i's {sample | LSTM step, Dense} run on stream i, joined on
        # the main stream around the feedback step.  Their single steps take the launch-per-step LSTM kernels: persistent launches
        # spin on their own workgroups and must not share the device with one another (LstmStack.persist_single_step).
        # all-NADE generators of one shape: their M sampling scans of a step are ONE launch too (ops.nade_sample_multi), writing straight into
        # track m of out[:, s] -- a generated step is then 6 launches on one stream: samples | feedback LSTM x layers | generators' LSTM x
        # layers | generators' Dense (35 graph nodes per step in round 3)
        gs = self.generators
        if group and all(type(g).__name__ == "RnnNade" and g.num_tracks == 1 and g.num_dims == P and g.num_hidden[-1] == gs[0].num_hidden[-1]
                         and g.row0 == gs[0].row0 for g in gs):
            Hn = gs[0].num_hidden[-1]
            fb_strided = getattr(self.feedback, "det_sampling", False) and hasattr(self.feedback, "_stack")
            for s in range(num_steps):                                                              # _feedback_recurrence (175-218)
                views = [out[:, s, :, i] for i in range(M)]                                          # u8 [B, P] views of track i (element stride M)
                ops.nade_sample_multi([dict(bias=states[i].dense, w_enc=g.store["nade/w_enc"][0], w_dec=g.store["nade/w_dec"][0], seed=g.seed,
                                            samples=views[i]) for i, g in enumerate(gs)], P, Hn, 1.0, gs[0].row0, s)
                st = out[:, s].reshape(B, P * M)                                                     # [B, P*M] view, feature p*M+m
                fb, fb_state = self.feedback.single(st if fb_strided else st.contiguous(), fb_state)
                res = det_steps([g._stack for g in gs], views, [list(s_.rnn_state) for s_ in states], [fb] * M)
                states = self._group_dense([r[0] for r in res], [r[1] for r in res])
            return out
        par = self.concurrent and x_u8.is_cuda and M > 1
        main = torch.cuda.current_stream() if x_u8.is_cuda else None
        if par:
            if getattr(self, "_lanes", None) is None or len(self._lanes) != M:
                self._lanes = [torch.cuda.Stream() for _ in range(M)]
            lanes = self._lanes
            for g in self.generators:
                g._stack.persist_single_step = False
        import contextlib
        on = (lambda i: torch.cuda.stream(lanes[i])) if par else (lambda i: contextlib.nullcontext())
        try:
            for s in range(num_steps):                                                              # _feedback_recurrence (175-218)
                samples = []
                for i, g in enumerate(self.generators):
                    if par and (s == 0 or group):           # grouped steps run on the main stream: every lane waits for their Dense outputs
                        lanes[i].wait_stream(main)
                    with on(i):
                        g._gen_step = s
                        g._last_dense = states[i].dense
                        smp, _ = g.sample_single(None, states[i])
                    samples.append(smp)
                if par:
                    for ln in lanes:
                        main.wait_stream(ln)
                st = torch.stack(samples, -1)                                                       # [B,P,M]
                out[:, s] = st
                fb, fb_state = self.feedback.single(st.reshape(B, P * M), fb_state)
                if group:                               # on the main stream: two grouped LSTM launches + one grouped Dense for all M tracks
                    res = det_steps([g._stack for g in self.generators], samples, [list(s_.rnn_state) for s_ in states], [fb] * M)
                    states = self._group_dense([r[0] for r in res], [r[1] for r in res])
                    continue
                for i, g in enumerate(self.generators):
                    if par:
                        lanes[i].wait_stream(main)
                    with on(i):
                        states[i] = g.single_step(samples[i], states[i], x2=fb) if det else \
                            g.single_step(torch.cat([samples[i].float(), fb], 1), states[i])
            if par:
                for ln in lanes:
                    main.wait_stream(ln)
        finally:
            if par:
                for g in self.generators:
                    g._stack.persist_single_step = True
        return out


FeedbackSampler = FeedbackRnnSampler          # the scan is the same for both feedback modules (multinn_feedback.py:120-218)
