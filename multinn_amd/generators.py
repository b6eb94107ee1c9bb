"""Host-side mirror of /root/reference/multinn/models/generators: Generator, RnnEstimator,
RnnNade, RnnMultiNADE, RnnRBM -- same constructor arguments, method names and return arity.

Eager semantics: ``build(x, y, lengths, is_train, mode)`` RUNS the forward pass (the reference
builds graph ops that a later ``sess.run`` executes), ``train(optimizer, lr)`` runs the backward
pass, the data-parallel all-reduce and the clipped optimiser step.  All compute goes through
the C ABI (multinn_amd.ops); internal tensors are time-major ([T,B,...], row n = t*B + b) and
rows past ``lengths`` are masked by a zero row weight instead of being gathered away
(utils/sequences.py:6-37 defines only the ORDER of the API-level flat outputs, reproduced by
``flat_index``).
"""
import abc
import collections
import math
import os

import weakref

import torch

from . import ops
from .common import Model, RNN, NADE, RBM, ParamStore, ScanGraphs, glorot_uniform, zeros_init, default_device, graph_capture
from .training import compute_gradients, world, dp_active, AdamOptimizer

_RnnEstimatorStateTuple = collections.namedtuple("RnnEstimatorStateTuple", ("b_enc", "b_dec", "rnn_state"))


class RnnEstimatorStateTuple(_RnnEstimatorStateTuple):
    """rnn_estimator.py:12-36.  `dense` (not in the reference) keeps the [b_enc | b_dec] matrix the biases are
    views of, so the sampling kernel can take it without a copy."""
    dense = None

    @property
    def dtype(self):
        return self.b_enc.dtype


def _compute_dtype(precision):
    if precision in ("bf16", torch.bfloat16):
        return torch.bfloat16
    if precision in ("fp16", "f16", torch.float16):
        return torch.float16
    if precision in ("fp32", "f32", torch.float32):
        return torch.float32
    raise ValueError("precision must be 'fp16', 'bf16' or 'fp32'")


def flat_index(lengths, B, T, device):
    """Time-major row ids (t*B+b) of the valid rows in the reference's flat order: b-major,
    then t (utils/sequences.py:30-31)."""
    t = torch.arange(T, device=device)[None, :].expand(B, T)
    b = torch.arange(B, device=device)[:, None].expand(B, T)
    if lengths is None:
        return (t * B + b).reshape(-1)
    m = t < lengths.to(device)[:, None]
    return (t * B + b)[m]


# ------------------------------------------------------------------------------------------------
# Lockstep execution of several generators (the M per-track generators of the jamming mode, multinn_jamming.py:40-68,213-221).  The parts of a
# generator that launch an LSTM recurrence are written as Python generator functions (`*_co`) that YIELD the launch instead of issuing it:
#     ("resident_fwd" | "resident_bwd" | "cluster_fwd" | "cluster_bwd" | "rowpar_fwd" | "rowpar_bwd", T, B, descriptor, keep_prob, workspace)
# `drive` runs one of them alone (every request becomes its own launch: the ordinary path).  `drive_group` advances M of them side by side and
# turns the M requests of a rendezvous into ONE launch where the library has a multi-job form (ops.lstm_recurrence_multi: the CU-resident and
# cluster recurrences own their rows for the whole sequence, so independent layers simply share a grid); everything between two rendezvous
# (GEMMs, Gibbs chains, ...) is issued per generator, in generator order, on the same stream.
_SINGLE = {"resident_fwd": lambda T, B, d, kp, ws: ops.lstm_resident_fwd(T, B, d, kp),
           "resident_bwd": lambda T, B, d, kp, ws: ops.lstm_resident_bwd(T, B, d, kp),
           "cluster_fwd": lambda T, B, d, kp, ws: ops.lstm_cluster_fwd(T, B, d, kp, ws),
           "cluster_bwd": lambda T, B, d, kp, ws: ops.lstm_cluster_bwd(T, B, d, kp, ws),
           "rowpar_fwd": lambda T, B, d, kp, ws: ops.lstm_rowpar_fwd(T, B, d, kp, ws),
           "rowpar_bwd": lambda T, B, d, kp, ws: ops.lstm_rowpar_bwd(T, B, d, kp, ws)}


def drive(co):
    """Run one `*_co` generator function to its end, issuing every recurrence it asks for as a launch of its own; returns its return value."""
    try:
        req = next(co)
        while True:
            _SINGLE[req[0]](*req[1:])
            req = co.send(None)
    except StopIteration as e:
        return e.value


def drive_group(cos):
    """Run M `*_co` generator functions in lockstep (see above); returns the list of their return values.  They must ask for the same
    sequence of recurrences (same kinds and shapes: the caller checks that the generators are alike before grouping them)."""
    n = len(cos)
    results, reqs, alive = [None] * n, [None] * n, [True] * n

    def advance(i, first):
        try:
            reqs[i] = next(cos[i]) if first else cos[i].send(None)
        except StopIteration as e:
            results[i], reqs[i], alive[i] = e.value, None, False

    for i in range(n):
        advance(i, True)
    while any(alive):
        if not all(alive):
            raise RuntimeError("drive_group: the grouped generators did not ask for the same sequence of recurrences")
        kind, T, B, _, kp, _ = reqs[0]
        same = all(r[0] == kind and r[1] == T and r[2] == B and r[4] == kp and r[3].units == reqs[0][3].units for r in reqs)
        multi = same and n > 1 and kind.split("_")[0] in ("resident", "cluster")
        if multi and kind == "cluster_bwd" and not ops.lstm_cluster_bwd_multi_ok(B, reqs[0][3].units, n):
            multi = False
        if multi and kind.startswith("cluster") and (n * (B // 32)) % 8 != 0:
            multi = False
        if multi:
            ops.lstm_recurrence_multi(kind, T, B, [r[3] for r in reqs], kp, [r[5] for r in reqs] if kind.startswith("cluster") else None)
        else:
            for r in reqs:
                _SINGLE[r[0]](*r[1:])
        for i in range(n):
            advance(i, False)
    return results


# ------------------------------------------------------------------------------------------------
class LstmStack:
    """Executes an RNN (multi-layer LSTM) on packed, gate-interleaved weights."""

    def __init__(self, rnn, store, dtype):
        self.rnn, self.store, self.dtype = rnn, store, dtype
        self.al = 8 if dtype in ops.H16 else 4
        self.h16 = dtype in ops.H16                     # 16-bit operands (bf16 or IEEE half): the persistent / matrix-core forms apply
        self.ld0 = ops.round_up(rnn.n_in, 64)          # K of the input projection: multiple of 64 selects the LDS-DMA GEMM
        self.packed = None

    # IEEE half has 5 exponent bits: the backward pass of a mean-over-rows loss (seeds of 1/N ~ 4e-6 at the bench shape) would sit in its
    # subnormals.  The owner of a backward pass (RnnNade / RnnRBM / FeedbackRnn) multiplies its gradient seed by loss_scale(N) -- a power of
    # two, so every f32 result is the unscaled one times 2^k exactly -- and multiplies store.grad (and d loss / d inputs) by 1/scale at its
    # end: callers always see unscaled gradients.  The seed of a row is 1 / n_valid (VALID rows of all ranks, not B*T: on a ragged window the
    # two differ by up to max_len), so 256 * n_valid keeps |d logits| <= 256 and typical dz around 1..100 (f16: 6e-5 .. 65504) for ragged
    # batches too.  An overflow that happens anyway leaves a non-finite gradient norm: mnn_clip_adam_step skips that update on the device
    # and Generator.check() raises.
    loss_scale_rows = 256.0

    def loss_scale(self, n_valid):
        """n_valid: number of VALID rows the mean-over-rows loss divides by (summed over all ranks)."""
        if self.dtype != torch.float16:
            return 1.0
        if n_valid is None:
            raise RuntimeError("loss_scale: the valid-row count lives on the device (ragged_on_device): use the generator's device-side scale")
        return float(2.0 ** round(math.log2(self.loss_scale_rows * max(int(n_valid), 1))))

    def pack(self):
        dev = self.store.theta.device
        self.packed = []
        for l, (n_in, u) in enumerate(zip(self.rnn.layer_inputs(), self.rnn.num_units)):
            ld = self.ld0 if l == 0 else n_in
            p = dict(wx_t=torch.empty((4 * u, ld), device=dev, dtype=self.dtype), wh_t=torch.empty((4 * u, u), device=dev, dtype=self.dtype),
                     wh_p=torch.empty((u, 4 * u), device=dev, dtype=self.dtype),
                     wx_p=torch.empty((n_in, 4 * u), device=dev, dtype=self.dtype) if l > 0 else None,
                     bias_p=torch.empty(4 * u, device=dev), n_in=n_in, u=u, ld=ld)
            ops.lstm_pack_weights(self.store[f"{self.rnn.prefix}/cell_{l}/kernel"], self.store[f"{self.rnn.prefix}/cell_{l}/bias"], n_in, u,
                                  p["wx_t"], p["wh_t"], p["wh_p"], p["wx_p"], p["bias_p"])
            if self.h16 and (l == 0 or self.rowpar):
                # the persistent recurrences read xproj gate-minor: the projection GEMM gets the rows in that order (layer 1 of the two-layer
                # form; every layer of the row-parallel form, whose layers each have their own projection GEMM)
                p["wx_gm"], p["bias_gm"] = torch.empty_like(p["wx_t"]), torch.empty_like(p["bias_p"])
                ops.lstm_rows_gate_minor(p["wx_t"], p["bias_p"], p["wx_gm"], p["bias_gm"])
            self.packed.append(p)

    chunk = 16        # timesteps per wavefront chunk
    # Measured on MI355X / ROCm 7.2 (profiles/round1_c_wavefront_note.md): hipGraph replay runs the per-layer branches
    # one after the other and the eager multi-stream form is host-bound, so the wavefront is OFF by default.
    pipelined = False

    def _lanes(self, n, dev):
        """n streams: the current one plus cached side streams (one per LSTM layer)."""
        if not hasattr(self, "_side") or len(self._side) < n - 1:
            self._side = [torch.cuda.Stream(device=dev) for _ in range(n - 1)]
        return [torch.cuda.current_stream()] + self._side[:n - 1]

    fused_layers = True   # run two-layer stacks as a wavefront inside single launches (mnn_lstm2_seq_*)

    def _fused2(self, B=0):
        """Two layers per launch pay off while one layer's step does not fill the chip (measured: C2 B=256 7.1 -> 6.2
        ms/step; TGT B=1024, 768 blocks per launch, 35.7 -> 36.4 ms/step)."""
        if not (self.fused_layers and len(self.packed) == 2 and self.dtype == torch.bfloat16
                and all(ops.lstm_fused_outputs(self.dtype, p["u"]) for p in self.packed)):
            return False
        blocks = sum(p["u"] // 32 for p in self.packed) * -(-max(B, 1) // 32)
        return blocks <= 512

    persistent = os.environ.get("MULTINN_PERSIST", "1") != "0"   # one launch for all T steps (lstm_persist.hip) when the grid fits the device

    # A persistent launch spins on its own workgroups and needs ALL of them resident: two such launches must never share the device.
    # A caller that runs several stacks on concurrent streams (the feedback sampling scan) clears this for its single steps, which then
    # take the launch-per-step kernels (T = 1: nothing to keep resident anyway).
    persist_single_step = True

    def _persist(self, B, T=2):
        if not (self.persistent and len(self.packed) == 2 and self.h16):
            return False
        if T == 1 and not self.persist_single_step:
            return False
        return ops.lstm2_persist_ok(B, self.packed[0]["u"], self.packed[1]["u"])

    # Row-parallel persistent form (lstm_rowpar.hip): one launch per LAYER for all T steps, weights in LDS, a wave per 32-row tile.  For
    # large batches, where the two-layer form's fixed cost per 32-row item (K split over the waves, LDS reduction, workgroup barriers) is
    # paid several times per timestep: TGT [1024,256,88,5] forward 15.6 us per timestep there.
    rowpar = os.environ.get("MULTINN_ROWPAR", "1") != "0"
    rowpar_min_batch = int(os.environ.get("MULTINN_ROWPAR_MIN_BATCH", "512"))
    # input projections the row-parallel form reads (bias included): stored in the 16-bit compute type by default (half the bytes of the
    # step's largest tensor), f32 with MULTINN_ROWPAR_XPROJ=f32
    rowpar_xproj_f32 = os.environ.get("MULTINN_ROWPAR_XPROJ", "16") == "f32"

    # CU-resident form of a 256-unit layer inside the row-parallel path (lstm_resident.hip): the layer's whole recurrent matrix sits on every CU
    # and a workgroup owns four batch rows, so a timestep has no hand-off between workgroups (MULTINN_RESIDENT=0: row-parallel kernels only)
    resident = os.environ.get("MULTINN_RESIDENT", "1") != "0"

    def _resident(self, l, B, T):
        # (the kernels address their tensors through 2 GB buffer descriptors: the saved gates [T, B, 4u] in 16 bits are the largest)
        return (self.resident and not self.rowpar_xproj_f32 and ops.lstm_resident_ok(B, self.packed[l]["u"])
                and T * B * self.packed[l]["u"] * 8 < 2 ** 31)

    # Cluster form of the same idea for a 512-unit layer (lstm_cluster.hip): eight CUs share 32 rows, each keeps 64 units' recurrent weights in
    # its registers, h[t] / dz[t] are exchanged through the XCD's L2 (MULTINN_CLUSTER=0: row-parallel kernels for that layer)
    cluster = os.environ.get("MULTINN_CLUSTER", "1") != "0"

    def _cluster(self, l, B, T):
        return (self.cluster and not self.rowpar_xproj_f32 and ops.lstm_cluster_ok(B, self.packed[l]["u"])
                and T * B * self.packed[l]["u"] * 8 < 2 ** 31)

    def _cluster_bwd(self, l, B, T):
        """The cluster BACKWARD needs every cluster's eight workgroups on one XCD (its two-deep exchange area lives in that XCD's L2; the forward has
        a write-through fall-back).  The library asks the placement once per device and batch size on the host; when it says no -- a repartitioned
        device, MNN_PERSIST_NO_LOCAL -- this layer's backward takes the row-parallel kernels (same descriptors, same saved activations), said once."""
        if not (self._cluster(l, B, T) and T >= 4):
            return False
        if ops.lstm_cluster_bwd_ok(B, self.packed[l]["u"]):
            return True
        if not getattr(LstmStack, "_cluster_bwd_warned", False):
            LstmStack._cluster_bwd_warned = True
            import warnings
            warnings.warn("multinn_amd: the clusters of the 512-unit recurrence are not dealt onto single XCDs on this device; its backward runs on "
                          "the row-parallel kernels (lstm_rowpar_bwd) instead of lstm_cluster_bwd")
        return False

    @property
    def rowpar_xproj_dtype(self):
        return torch.float32 if self.rowpar_xproj_f32 else self.dtype

    # set by a mode that runs several stacks in lockstep (drive_group): the row-parallel path -- whose CU-resident / cluster recurrences have a
    # multi-job launch -- also below rowpar_min_batch, where one stack alone is faster on the two-layer persistent form
    group_rowpar = False

    def _rowpar(self, B, T=2, state0=None):
        if not (self.rowpar and self.h16 and state0 is None and T > 1 and (B >= self.rowpar_min_batch or self.group_rowpar) and B % 32 == 0):
            return False
        return all("wx_gm" in p and ops.lstm_rowpar_ok(B, p["u"]) for p in self.packed)

    def _rp_workspace(self, l, T, B, dev):
        if not hasattr(self, "_rpws"):
            self._rpws = {}
        key = (l, T, B)
        if key not in self._rpws:
            self._rpws[key] = ops.lstm_rowpar_workspace(T, B, self.packed[l]["u"], dev)
        return self._rpws[key]

    # One weight-gradient GEMM per layer over the concatenated operand [x^T ; h_prev^T] (row-parallel form): dz^T is streamed once for dWx and
    # dWh, and 2 x 4 column tiles of a K slice share every dz^T panel on an XCD's L2 instead of 2 x 2 (layer 1 at [1024,256,88,5]: 1.33 ->
    # 1.03 ms for the pair, scratch/gemm_merge_probe.py).  The layer's forward writes h^T -- and the layer below its y^T -- into views of it.
    merge_wgrads = os.environ.get("MULTINN_MERGE_WGRADS", "1") != "0"

    def _cat_shape(self, l, Np):
        p = self.packed[l]
        return (p["ld"] + p["u"], Np)

    def input_T(self, T, B, dev):
        """Buffer for the caller's transposed copy of the stack's input (x^T [ld0, Np], layer 1's weight-gradient operand): a view of layer
        1's concatenated operand when the row-parallel form with merged weight-gradient GEMMs will run, else None (caller allocates)."""
        if not (self.merge_wgrads and self._rowpar(B, T)):
            return None
        Np = ops.round_up(T * B, 64)
        self._cat0 = (torch.zeros if Np != T * B else torch.empty)(self._cat_shape(0, Np), device=dev, dtype=self.dtype)
        return self._cat0[:self.packed[0]["ld"]]

    def _forward_rowpar(self, x_tm, keep_prob, seed, row0, save, step_dev):
        return drive(self._forward_rowpar_co(x_tm, keep_prob, seed, row0, save, step_dev))

    def _forward_rowpar_co(self, x_tm, keep_prob, seed, row0, save, step_dev):
        """Layer by layer: gate-minor input projection (one GEMM over all T*B rows), then the layer's whole recurrence in one launch -- YIELDED
        to the driver (drive / drive_group above), which issues it alone or together with the same layer of other stacks."""
        T, B, _ = x_tm.shape
        dev, N = x_tm.device, T * B
        Np = ops.round_up(N, 64)
        zalloc = torch.zeros if Np != N else torch.empty
        inp, ctx, final = x_tm, [], []
        cats = [None] * len(self.packed)
        if save and self.merge_wgrads:
            for l, p in enumerate(self.packed):
                if l > 0 and p["ld"] != self.packed[l - 1]["u"]:
                    continue                            # a padded input pitch: the layer below's y^T is not this layer's x^T row for row
                c0 = getattr(self, "_cat0", None) if l == 0 else None
                if c0 is not None and tuple(c0.shape) == self._cat_shape(0, Np) and c0.device == dev:
                    cats[l], self._cat0 = c0, None
                else:
                    cats[l] = zalloc(self._cat_shape(l, Np), device=dev, dtype=self.dtype)
        for l, p in enumerate(self.packed):
            u = p["u"]
            # the input projection is the step's largest tensor (TGT layer 1: 2.1 GB in f32): written and read once, in bf16 by default
            xproj = torch.empty((T, B, 4 * u), device=dev, dtype=self.rowpar_xproj_dtype)
            ops.gemm_tn(inp.view(N, -1), p["wx_gm"], xproj.view(N, -1), bias=p["bias_gm"])
            h = torch.empty((T, B, u), device=dev, dtype=self.dtype)
            mask = y = None
            if keep_prob < 1.0:
                mask = torch.empty((T, B, u), device=dev, dtype=torch.uint8)
                ops.dropout_mask(mask, keep_prob, seed, row0, l, step_dev)
                y = torch.empty_like(h)
            gates = torch.empty((T, B, 4 * u), device=dev, dtype=self.dtype) if save else None         # this form saves its activations in 16 bits
            c = torch.empty((T, B, u), device=dev)
            hT = yT = None
            if save:
                hT = cats[l][p["ld"]:] if cats[l] is not None else torch.empty((u, Np), device=dev, dtype=self.dtype)
                hT[:, :B].zero_()                      # h_{-1} = 0; columns [B, T*B) are written by the launch
                if Np != N:
                    hT[:, N:].zero_()
                nxt = cats[l + 1] if l + 1 < len(self.packed) else None
                yT = nxt[:u] if nxt is not None else zalloc((u, Np), device=dev, dtype=self.dtype)
            d = ops.lstm2_fwd_layer(xproj, p["wh_t"], None, None, gates, c, h, hT, y, mask, yT=yT, gates_dtype=self.dtype,
                                    xproj_dtype=self.rowpar_xproj_dtype)
            if self._resident(l, B, T):
                yield ("resident_fwd", T, B, d, keep_prob, None)
            elif self._cluster(l, B, T):
                yield ("cluster_fwd", T, B, d, keep_prob, self._rp_workspace(l, T, B, dev))
            else:
                yield ("rowpar_fwd", T, B, d, keep_prob, self._rp_workspace(l, T, B, dev))
            out = y if y is not None else h
            if save:
                ctx.append(dict(inp=inp, gates=gates, c=c, h=h, c0=None, h0=None, hT=hT, mask=mask, yT=yT,
                                inT=ctx[l - 1]["yT"] if l > 0 else None, persist=True, rowpar=True, catT=cats[l]))
            final.append((c[-1], h[-1]))
            inp = out
        return inp, ctx, final

    def _backward_rowpar(self, dy, ctx, keep_prob, need_dx=False):
        return drive(self._backward_rowpar_co(dy, ctx, keep_prob, need_dx))

    def _backward_rowpar_co(self, dy, ctx, keep_prob, need_dx=False):
        """Top layer first: the layer's whole backward recurrence in one launch (dropout backward of its output folded in; yielded to the
        driver like the forward's), then the gradient wrt its input as one GEMM (dz row-major x Wx), which is the next layer's dh_ext."""
        T, B, _ = dy.shape
        dev, N = dy.device, T * B
        Np = ops.round_up(N, 64)
        zalloc = torch.zeros if Np != N else torch.empty
        dh = dy.contiguous()
        st = [None] * len(self.packed)
        for l in range(len(self.packed) - 1, -1, -1):
            p, cx = self.packed[l], ctx[l]
            u = p["u"]
            # dz^T in the K-BLOCKED layout [N/32, 4u, 32] (MNN_GEMM_A_KBLOCK32): a wave's 128 gate columns x 32 rows are one contiguous
            # 8 KB slab (one kilobyte per store instruction) instead of 128 runs of 64 bytes 512 KB apart -- 0.7 us less per timestep on the
            # backward chain, and the GEMM only changes its LDS-DMA source addresses
            kb = self.kblock_wgrads and cx.get("catT") is not None and Np == N and N % 64 == 0
            dzT = torch.empty((N // 32, 4 * u, 32), device=dev, dtype=self.dtype) if kb else zalloc((4 * u, Np), device=dev, dtype=self.dtype)
            dzc = torch.empty((T, B, 4 * u), device=dev, dtype=self.dtype) if (l > 0 or (kb and need_dx)) else None
            db_p = self._accum(l, dev)[2]
            e = ops.lstm2_bwd_layer(dh.view(T, B, u), p["wh_p"], cx["gates"], cx["c"], None, dzc, ops.lstm_seq_bwd_workspace(B, u, dev), dzT, db_p,
                                    cx["mask"] if keep_prob < 1.0 else None, gates_dtype=self.dtype)
            if self._resident(l, B, T):
                yield ("resident_bwd", T, B, e, keep_prob, None)
            elif self._cluster_bwd(l, B, T):
                yield ("cluster_bwd", T, B, e, keep_prob, self._rp_workspace(l, T, B, dev))
            else:
                yield ("rowpar_bwd", T, B, e, keep_prob, self._rp_workspace(l, T, B, dev))
            st[l] = dict(dzT=dzT, dzc=dzc, db_p=db_p)
            if l > 0:
                dh = torch.empty((N, p["n_in"]), device=dev)
                ops.gemm_tn(dzc.view(N, 4 * u), p["wx_p"], dh)
        if getattr(self, "keep_debug", False):
            self._dbg_dzT = [s_["dzT"] for s_ in st]
        keep = [self._weight_grads(l, ctx[l], st[l]["dzT"], st[l]["db_p"], T, B, dz=st[l]["dzc"]) for l in range(len(self.packed) - 1, -1, -1)]
        return self._input_grad(st[0]["dzT"], T, B, dz=st[0]["dzc"]) if need_dx else None

    def _workspace(self, T, B, dev):
        """Flags + exchange area of the persistent launches, one per (T, B) (kept alive: captured graphs point at it)."""
        if not hasattr(self, "_pws"):
            self._pws = {}
        if (T, B) not in self._pws:
            self._pws[(T, B)] = ops.lstm2_persist_workspace(T, B, self.packed[0]["u"], self.packed[1]["u"], dev)
        return self._pws[(T, B)]

    def check(self):
        """Raise if a persistent launch ever gave up waiting (synchronises the device)."""
        for (T, B), ws in getattr(self, "_pws", {}).items():
            ops.lstm2_persist_check(ws, B, self.packed[0]["u"], self.packed[1]["u"])
        for ws in getattr(self, "_rpws", {}).values():
            ops.lstm_rowpar_check(ws)

    @staticmethod
    def _chunks(T, step):
        return [(t0, min(T, t0 + step)) for t0 in range(0, T, step)]

    def forward_co(self, x_tm, keep_prob=1.0, seed=0, row0=0, save=True, state0=None, step_dev=None):
        """`forward` as a generator function for drive / drive_group: the row-parallel path yields its recurrence launches, every other path
        runs at once."""
        T, B, _ = x_tm.shape
        if self._rowpar(B, T, state0):
            return (yield from self._forward_rowpar_co(x_tm, keep_prob, seed, row0, save, step_dev))
        return self.forward(x_tm, keep_prob, seed, row0, save, state0, step_dev)

    def backward_co(self, dy, ctx, keep_prob=1.0, seed=0, row0=0, need_dx=False, step_dev=None):
        if ctx and ctx[0].get("rowpar"):
            return (yield from self._backward_rowpar_co(dy, ctx, keep_prob, need_dx))
        return self.backward(dy, ctx, keep_prob, seed, row0, need_dx, step_dev)

    def forward(self, x_tm, keep_prob=1.0, seed=0, row0=0, save=True, state0=None, step_dev=None):
        """x_tm [T,B,ld0] compute dtype.  Returns (y [T,B,u_last], ctx, final_state[(c,h)...]).

        The T-step recurrences are latency-bound chains, so the layers run as a WAVEFRONT: layer l works on
        time chunk c on its own HIP stream while layer l-1 is already on chunk c+1 (events order the chunks)."""
        T, B, _ = x_tm.shape
        dev = x_tm.device
        L = len(self.packed)
        if self._rowpar(B, T, state0):
            return self._forward_rowpar(x_tm, keep_prob, seed, row0, save, step_dev)
        persist = self._persist(B, T)
        bufs = []
        for l, p in enumerate(self.packed):
            u = p["u"]
            h = torch.empty((T, B, u), device=dev, dtype=self.dtype)
            hT = None
            if save:                        # transposed previous-state operand of dWh, written by the step kernels
                Np = ops.round_up(T * B, 64)            # columns [0,B) = h_{-1} = 0 (or h0), [B, T*B) written by the step kernels
                hT = torch.empty((u, Np), device=dev, dtype=self.dtype)
                hT[:, :B].zero_()
                if Np != T * B:
                    hT[:, T * B:].zero_()
                if state0 is not None:
                    ops.transpose(state0[l][1].to(self.dtype).contiguous(), hT[:, :B])
            bufs.append(dict(xproj=None if (persist and l == 1) else torch.empty((T, B, 4 * u), device=dev), gates=torch.empty((T, B, 4 * u), device=dev) if save else None,
                             c=torch.empty((T, B, u), device=dev), h=h, y=torch.empty_like(h) if keep_prob < 1.0 else h, hT=hT,
                             c0=state0[l][0] if state0 is not None else None,
                             h0=state0[l][1].to(self.dtype) if state0 is not None else None))
        piped = self.pipelined and L > 1 and T > self.chunk
        chunks = self._chunks(T, self.chunk if piped else T)
        lanes = self._lanes(L if piped else 1, dev)
        main = lanes[0]
        for s in lanes[1:]:
            s.wait_stream(main)
        # layer 0's input projection has no dependency: one big GEMM
        p0, b0 = self.packed[0], bufs[0]
        if persist:
            ops.gemm_tn(x_tm.view(T * B, -1), p0["wx_gm"], b0["xproj"].view(T * B, -1), bias=p0["bias_gm"])       # gate-minor columns
        else:
            ops.gemm_tn(x_tm.view(T * B, -1), p0["wx_t"], b0["xproj"].view(T * B, -1), bias=p0["bias_p"])
        if persist or self._fused2(B):
            # persist: ONE launch for the whole recurrence of both layers; else ONE launch per timestep for the whole
            # stack: layer-0 step s | layer-1 projection s-1 | layer-1 step s-2
            p1, b1 = self.packed[1], bufs[1]
            masks = [None, None]
            if keep_prob < 1.0:
                for l, bf in enumerate(bufs):
                    masks[l] = torch.empty(bf["h"].shape, device=dev, dtype=torch.uint8)
                    ops.dropout_mask(masks[l], keep_prob, seed, row0, l, step_dev)
            yT = [None, None]
            if persist and save:            # the persistent launch also emits y^T of both layers: no transposes before the weight gradients
                Np = ops.round_up(T * B, 64)
                yT = [(torch.zeros if Np != T * B else torch.empty)((p["u"], Np), device=dev, dtype=self.dtype) for p in self.packed]
            d0 = ops.lstm2_fwd_layer(b0["xproj"], p0["wh_t"], b0["h0"], b0["c0"], b0["gates"], b0["c"], b0["h"], b0["hT"],
                                     b0["y"] if masks[0] is not None else None, masks[0], yT=yT[0])
            d1 = ops.lstm2_fwd_layer(b1["xproj"], p1["wh_t"], b1["h0"], b1["c0"], b1["gates"], b1["c"], b1["h"], b1["hT"],
                                     b1["y"] if masks[1] is not None else None, masks[1], p1["wx_t"], p1["bias_p"], yT=yT[1])
            for bf, yt in zip(bufs, yT):
                bf["yT"] = yt
            if persist:
                ops.lstm2_persist_fwd(T, B, d0, d1, keep_prob, self._workspace(T, B, dev))
            else:
                ops.lstm2_seq_fwd(T, B, d0, d1, keep_prob)
            for bf, mk in zip(bufs, masks):
                bf["mask"] = mk
            chunks = []
        done = [[None] * len(chunks) for _ in range(L)]
        for ci, (t0, t1) in enumerate(chunks):
            for l, (p, bf) in enumerate(zip(self.packed, bufs)):
                lane = lanes[l] if piped else main
                with torch.cuda.stream(lane):
                    if l > 0:
                        if piped:
                            lane.wait_event(done[l - 1][ci])
                        inp = bufs[l - 1]["y"]
                        ops.gemm_tn(inp[t0:t1].view((t1 - t0) * B, -1), p["wx_t"], bf["xproj"][t0:t1].view((t1 - t0) * B, -1), bias=p["bias_p"])
                    ops.lstm_seq_fwd(bf["xproj"], p["wh_t"], bf["h0"], bf["c0"], bf["gates"], bf["c"], bf["h"], t0, t1, bf["hT"])
                    if keep_prob < 1.0:
                        ops.dropout_fwd(bf["h"][t0:t1], bf["y"][t0:t1], keep_prob, seed, row0, l, step_dev, t0)
                    if piped and l < L - 1:
                        done[l][ci] = torch.cuda.Event()
                        done[l][ci].record(lane)
        for s in lanes[1:]:
            main.wait_stream(s)
        ctx = []
        if save:
            for l, bf in enumerate(bufs):
                ctx.append(dict(inp=x_tm if l == 0 else bufs[l - 1]["y"], gates=bf["gates"], c=bf["c"], h=bf["h"], c0=bf["c0"], h0=bf["h0"],
                                hT=bf["hT"], mask=bf.get("mask"), yT=bf.get("yT"), inT=bufs[l - 1].get("yT") if l > 0 else None,
                                persist=persist))        # persist: the saved gates are gate-minor -- only the persistent backward reads them
        final = [(bf["c"][-1], bf["h"][-1]) for bf in bufs]
        return bufs[-1]["y"], ctx, final

    @staticmethod
    def _split_k(rows_out, cols_out, K):
        # about 512 workgroups of the 128 x 128 tile (two per CU): every slice adds its tile with f32 atomics, and those run at one
        # chip-wide rate (~1.3 TB/s) -- 16 slices of dWh1 were 67 MB of adds, half of that GEMM's time (profiles/tools/gemm_sweep.py)
        # (at K >= 64 k the adds are a small share again and more slices win: 1024 workgroups)
        # a power of two from 8 up: equal K slices and the XCD-local mapping of id % split_k (the Dense gradient [256 x 704], K = 262144: 64
        # slices 159 us, 85 slices 213 us)
        tiles = -(-rows_out // 128) * -(-cols_out // 128)
        target = 512 if K < 65536 else 1024
        sk = int(max(1, min(target // max(tiles, 1), K // 1024)))
        return 1 << (sk.bit_length() - 1) if sk >= 8 else sk

    # MULTINN_KBLOCK_WGRADS=0: dz^T as a plain [4u, N] matrix instead of the K-blocked layout.  (A third form -- no dz^T at all, the GEMM reading
    # dz row-major through transposing LDS loads -- was measured net-neutral in round 3, profiles/round3_e_kmajor.md, and removed in round 4.)
    kblock_wgrads = os.environ.get("MULTINN_KBLOCK_WGRADS", "1") != "0"

    def _weight_grads(self, l, cx, dzT, db_p, T, B, dz=None):
        """dWx^T[4u,ld] = dz^T . inp ; dWh^T[4u,u] = dz^T . h_prev  (reduction over the N rows); dzT [4u,Np] and
        h_prev^T come straight from the step kernels, db_p from their epilogue."""
        p = self.packed[l]
        u, ld, n_in = p["u"], p["ld"], p["n_in"]
        N = T * B
        Np = N if dzT.dim() == 3 else dzT.shape[1]
        dev = db_p.device
        inT = cx.get("inT")                 # the producer's own transposed copy (persistent forward: y^T of the layer below)
        cat = cx.get("catT")
        if cat is not None:                 # [x^T ; h_prev^T] in one buffer (h^T is a view of it): one GEMM for both gradients
            if inT is None:
                ops.transpose(cx["inp"].view(N, ld), cat[:ld])
            elif inT.data_ptr() != cat.data_ptr():
                cat[:ld].copy_(inT)
            if not hasattr(self, "_acc_cat"):
                self._acc_cat = {}
            if l not in self._acc_cat:
                self._acc_cat[l] = torch.zeros((4 * u, ld + u), device=dev)
            dw_cat = self._acc_cat[l]
            # one resident round of 256 x 256 tiles (one per CU), the slice count a multiple of 4: measured at [1024 x 768], K = 262144:
            # split 20 400 us, 16 461, 21 669, 24 613, 32 494 (scratch/gemm_merge_probe2.py)
            tiles = -(-4 * u // 256) * -(-(ld + u) // 256)
            sk = max(1, min(256 // tiles // 4 * 4 if 256 // tiles >= 4 else 256 // tiles, Np // 1024))
            if dzT.dim() == 3:              # K-blocked dz^T
                ops.gemm_tn(dzT, cat, dw_cat, accumulate=True, split_k=sk, a_kblock=True)
            else:
                ops.gemm_tn(dzT, cat, dw_cat, accumulate=True, split_k=sk)
            ops.lstm_unpack_grads_cat(dw_cat, db_p, n_in, u, ld, self.store.gviews[f"{self.rnn.prefix}/cell_{l}/kernel"],
                                      self.store.gviews[f"{self.rnn.prefix}/cell_{l}/bias"])
            return (cat, dw_cat)
        if inT is None:
            inT = (torch.zeros if Np != N else torch.empty)((ld, Np), device=dev, dtype=self.dtype)
            ops.transpose(cx["inp"].view(N, ld), inT)
        # dwx_t / dwh_t / db_p are persistent accumulators (zero between steps: the unpack below clears what it reads), so the
        # split-K slices add into them without a zero-fill launch in front of every GEMM
        dwx_t, dwh_t, _ = self._accum(l, dev)
        ops.gemm_tn(dzT, inT, dwx_t, accumulate=True, split_k=self._split_k(4 * u, ld, Np))
        ops.gemm_tn(dzT, cx["hT"], dwh_t, accumulate=True, split_k=self._split_k(4 * u, u, Np))
        ops.lstm_unpack_grads(dwx_t, dwh_t, db_p, n_in, u, self.store.gviews[f"{self.rnn.prefix}/cell_{l}/kernel"],
                              self.store.gviews[f"{self.rnn.prefix}/cell_{l}/bias"], consume=True)
        return (inT, dwx_t, dwh_t)      # kept alive until the streams are joined

    def _accum(self, l, dev):
        """Packed weight-gradient accumulators of layer l: (dWx^T [4u, ld], dWh^T [4u, u], db [4u]) f32, allocated zeroed once."""
        if not hasattr(self, "_acc"):
            self._acc = {}
        if l not in self._acc:
            p = self.packed[l]
            self._acc[l] = (torch.zeros((4 * p["u"], p["ld"]), device=dev), torch.zeros((4 * p["u"], p["u"]), device=dev),
                            torch.zeros(4 * p["u"], device=dev))
        return self._acc[l]

    def backward(self, dy, ctx, keep_prob=1.0, seed=0, row0=0, need_dx=False, step_dev=None):
        """dy f32 [T,B,u_last]: gradient wrt the (dropped) top output.  Accumulates the kernel / bias gradients
        into the store's flat gradient buffer.  Same wavefront as forward, top layer first, chunks descending;
        each layer's weight-gradient GEMMs then run on that layer's stream."""
        if ctx and ctx[0].get("rowpar"):
            return self._backward_rowpar(dy, ctx, keep_prob, need_dx)
        T, B, _ = dy.shape
        dev = dy.device
        L = len(self.packed)
        piped = self.pipelined and L > 1 and T > self.chunk
        chunks = self._chunks(T, self.chunk if piped else T)
        lanes = self._lanes(L if piped else 1, dev)
        main = lanes[0]
        for s in lanes[1:]:
            s.wait_stream(main)
        dyl = [None] * L
        dyl[L - 1] = dy.view(T, B, -1)
        persist = bool(ctx[0].get("persist"))       # the layout of the saved gates is the forward's choice
        st = []
        Np = ops.round_up(T * B, 64)
        for l, p in enumerate(self.packed):
            u = p["u"]
            fused = ops.lstm_fused_outputs(self.dtype, u)          # bf16 step kernels emit dz^T and sum(dz) themselves
            dz = None if fused else torch.empty((T, B, 4 * u), device=dev)
            st.append(dict(dz=dz, dzc=None if persist else (dz if self.dtype == torch.float32 else torch.empty((T, B, 4 * u), device=dev, dtype=self.dtype)),
                           dzT=(torch.zeros if Np != T * B else torch.empty)((4 * u, Np), device=dev, dtype=self.dtype),
                           db_p=self._accum(l, dev)[2],
                           dh=torch.empty((T, B, u), device=dev) if (keep_prob < 1.0 and not (persist and ctx[l].get("mask") is not None)) else None,
                           ws=ops.lstm_seq_bwd_workspace(B, u, dev)))
            if l < L - 1 and not persist:
                dyl[l] = torch.empty((T, B, u), device=dev)
        lane_of = lambda l: lanes[L - 1 - l] if piped else main          # the top layer leads, on the current stream
        done = [[None] * len(chunks) for _ in range(L)]
        if (persist or (self._fused2(B) and ctx[0]["h0"] is None)) and (keep_prob >= 1.0 or ctx[0].get("mask") is not None):
            p0, p1 = self.packed
            fold = persist and keep_prob < 1.0 and ctx[1].get("mask") is not None    # the persistent backward applies layer 2's mask itself
            if keep_prob < 1.0 and not fold:
                ops.dropout_bwd(dyl[1], st[1]["dh"], keep_prob, seed, row0, 1, False, step_dev, 0)
            dh1 = st[1]["dh"] if (keep_prob < 1.0 and not fold) else dyl[1]
            dh0 = None if persist else dyl[0]              # written by the fused launches (stage Q), dropout already applied
            e0 = ops.lstm2_bwd_layer(dh0, p0["wh_p"], ctx[0]["gates"], ctx[0]["c"], ctx[0]["c0"], st[0]["dzc"], st[0]["ws"], st[0]["dzT"], st[0]["db_p"],
                                     ctx[0].get("mask"))
            e1 = ops.lstm2_bwd_layer(dh1, p1["wh_p"], ctx[1]["gates"], ctx[1]["c"], ctx[1]["c0"], st[1]["dzc"], st[1]["ws"], st[1]["dzT"], st[1]["db_p"],
                                     ctx[1]["mask"] if fold else None, p1["wx_p"])
            if persist:
                ops.lstm2_persist_bwd(T, B, e0, e1, keep_prob, self._workspace(T, B, dev))
            else:
                ops.lstm2_seq_bwd(T, B, e0, e1, keep_prob)
            chunks = []
        for ci in range(len(chunks) - 1, -1, -1):
            t0, t1 = chunks[ci]
            for l in range(L - 1, -1, -1):
                p, cx, s_ = self.packed[l], ctx[l], st[l]
                lane = lane_of(l)
                with torch.cuda.stream(lane):
                    if piped and l < L - 1:
                        lane.wait_event(done[l + 1][ci])
                    if keep_prob < 1.0:
                        ops.dropout_bwd(dyl[l][t0:t1], s_["dh"][t0:t1], keep_prob, seed, row0, l, False, step_dev, t0)
                        dh = s_["dh"]
                    else:
                        dh = dyl[l]
                    ops.lstm_seq_bwd(dh, p["wh_p"], cx["gates"], cx["c"], cx["c0"], s_["dz"], s_["dzc"], None, None, t0, t1, s_["ws"],
                                     s_["dzT"], s_["db_p"])
                    if l > 0:
                        ops.gemm_tn(s_["dzc"][t0:t1].view((t1 - t0) * B, -1), p["wx_p"], dyl[l - 1][t0:t1].view((t1 - t0) * B, -1))
                        if piped:
                            done[l][ci] = torch.cuda.Event()
                            done[l][ci].record(lane)
        if getattr(self, "keep_debug", False):      # tests: the recurrence's own (atomic-free, hence run-to-run bit-stable) outputs
            self._dbg_dzT = [s_["dzT"] for s_ in st]
        keep = []
        for l in range(L - 1, -1, -1):
            with torch.cuda.stream(lane_of(l)):
                keep.append(self._weight_grads(l, ctx[l], st[l]["dzT"], st[l]["db_p"], T, B))
        for s in lanes[1:]:
            main.wait_stream(s)
        return self._input_grad(st[0]["dzT"], T, B) if need_dx else None

    def _input_grad(self, dzT0, T, B, dz=None):
        """Gradient wrt the stack's inputs, f32 [T,B,n_in] = dz_0 . Wx_0^T (only the feedback modes consume it: the feedback vector is part of
        every per-track generator's input, multinn_feedback.py:85-91).  Every form of the recurrence leaves layer 0's dz as dz^T [4u, N]
        (the weight-gradient operand): one transpose pass makes the K-contiguous A operand."""
        p = self.packed[0]
        N = T * B
        if p.get("wx_p0") is None:          # [ld0, 4u]: the packed (gate-interleaved) input weights with K = 4u contiguous, once per pack
            p["wx_p0"] = torch.empty((p["ld"], 4 * p["u"]), device=p["wx_t"].device, dtype=self.dtype)
            ops.transpose(p["wx_t"], p["wx_p0"])
        if dzT0.dim() == 3:
            dz = dz.view(N, 4 * p["u"])
        else:
            dz = torch.empty((N, 4 * p["u"]), device=dzT0.device, dtype=self.dtype)
            ops.transpose(dzT0[:, :N], dz)
        dx = torch.empty((N, p["n_in"]), device=dz.device)
        ops.gemm_tn(dz, p["wx_p0"][:p["n_in"]], dx)
        return dx.view(T, B, p["n_in"])

    def single_step(self, x, state):
        """One time step (rnn_nade.py:268): x [B,ld0] compute dtype, state [(c,h)...] -> (h_top, new_state)."""
        y, _, final = self.forward(x.view(1, *x.shape), 1.0, save=False, state0=state)
        return y[0], [(c, h) for c, h in final]

    # -- deterministic f32 single steps (csrc/det_step.hip): the arithmetic of every sampling scan -----------------------------------
    owner = None          # weakref to the estimator this stack belongs to (its pack epoch dates the repacked weights below)
    _det_pack, _det_pack_key = None, None

    def det_job(self, l, x, n_x, x2, st):
        """Descriptor of layer l's deterministic step (ops.lstm_step_det) with fresh f32 outputs.  The master weights are read through their
        repacked copy (ops.det_lstm_pack: the same numbers in the kernel's load order), remade when the weights may have changed: a new
        store.step or a new pack epoch of the owner -- once per sampling scan, inside its graph."""
        u = self.rnn.num_units[l]
        ref = x if x is not None else x2
        c = torch.empty((ref.shape[0], u), device=ref.device)
        h = torch.empty_like(c)
        pre = self.rnn.prefix
        key = (self.store.step, getattr(self.owner() if self.owner is not None else None, "_pack_epoch", 0))
        if self._det_pack is None or self._det_pack_key != key:
            self._det_pack = [ops.det_lstm_pack(self.store[f"{pre}/cell_{k}/kernel"], self.rnn.num_units[k]) for k in range(len(self.rnn.num_units))]
            self._det_pack_key = key
        return dict(x=x, n_x=n_x, x2=x2, h_prev=None if st is None else st[1], c_prev=None if st is None else st[0],
                    W=self.store[f"{pre}/cell_{l}/kernel"], Wp=self._det_pack[l], bias=self.store[f"{pre}/cell_{l}/bias"], c_out=c, h_out=h)

    def det_step(self, x, state, x2=None):
        """One deterministic f32 step of the stack: x u8 | f32 [B, n_x] (unit inner stride), optional x2 f32 [B, n_x2] concatenated behind
        it; state [(c, h)...] f32 or None (zero state) -> (h_top f32 [B, u_last], new_state)."""
        return det_steps([self], [x], [state], [x2])[0]


def _det_f32(t):
    return t if t is None or t.dtype == torch.float32 else t.float()


def det_steps(stacks, xs, states, x2s=None):
    """One deterministic step of several LSTM stacks of equal depth (the M per-track generators of a feedback-scan step): layer by layer,
    the stacks' jobs of a layer run as ONE launch.  Returns [(h_top, new_state)] per stack."""
    n = len(stacks)
    x2s = x2s if x2s is not None else [None] * n
    L = len(stacks[0].rnn.num_units)
    assert all(len(s.rnn.num_units) == L for s in stacks)
    inp, inp2 = list(xs), list(x2s)
    new = [[] for _ in range(n)]
    for l in range(L):
        jobs = []
        for i, s in enumerate(stacks):
            x = inp[i]
            if x is not None and x.dtype not in (torch.uint8, torch.float32):
                x = x.float()
            st = None if states[i] is None else (_det_f32(states[i][l][0]).contiguous(), _det_f32(states[i][l][1]).contiguous())
            n_x = s.rnn.layer_inputs()[l] - (inp2[i].shape[1] if inp2[i] is not None else 0)
            jobs.append(s.det_job(l, x, n_x, inp2[i], st))
        ops.lstm_step_det(jobs)
        for i, j in enumerate(jobs):
            new[i].append((j["c_out"], j["h_out"]))
            inp[i], inp2[i] = j["h_out"], None
    return [(inp[i], new[i]) for i in range(n)]



# ------------------------------------------------------------------------------------------------
class Generator(Model):
    """models/generators/generator.py:9-205."""

    def __init__(self, num_dims, num_hidden, num_hidden_rnn, keep_prob=1.0, internal_bias=False, name="generator", track_name="all"):
        super().__init__(name=name)
        self._track_name = track_name
        self._num_dims = num_dims
        self._num_hidden = [num_hidden] if isinstance(num_hidden, int) else list(num_hidden)
        self._num_hidden_rnn = [num_hidden_rnn] if isinstance(num_hidden_rnn, int) else list(num_hidden_rnn)
        self._keep_prob, self._internal_bias = keep_prob, internal_bias
        self._lengths = self._inputs = None

    num_dims = property(lambda self: self._num_dims)
    num_hidden = property(lambda self: self._num_hidden)
    num_hidden_rnn = property(lambda self: self._num_hidden_rnn)
    track_name = property(lambda self: self._track_name)
    keep_prob = property(lambda self: self._keep_prob)
    internal_bias = property(lambda self: self._internal_bias)

    def build(self, x=None, y=None, lengths=None, is_train=None, mode="eval"):
        super().build(mode=mode)
        self._inputs, self._lengths = x, lengths

    @abc.abstractmethod
    def zero_state(self, batch_size):
        ...

    def forward(self):
        return self._outputs

    @abc.abstractmethod
    def generate(self, x, num_steps):
        ...

    def pretrain(self, optimizer, lr, run_optimizer=True):
        return [], [], self.metrics, self.metrics_upd, self.summaries

    def train(self, optimizer, lr, run_optimizer=True):
        """generator.py:176-205: backward of metrics['batch/loss'] + clipped optimiser step."""
        self.backward()                     # leaves self._dx (f32 [T,B,n_in], time-major) when self.need_dx is set by the mode
        summaries = dict(self.summaries)
        if run_optimizer:
            self._grad_sumsq = compute_gradients(optimizer, self.store, self.clip_norm, lr)
            self._packed_step = -1          # weights changed: re-pack before the next forward
        return [], [], self.metrics, self.metrics_upd, summaries


class RnnEstimator(Generator):
    """models/generators/rnn_estimator.py:39-323."""

    def __init__(self, num_dims, num_hidden, num_hidden_rnn, keep_prob=1.0, internal_bias=True, name="rnn-rbm", track_name="all",
                 num_inputs=None, precision="bf16", seed=23, device=None, clip_norm=5.0):
        super().__init__(num_dims, num_hidden, num_hidden_rnn, keep_prob, internal_bias, name, track_name)
        self.dtype = _compute_dtype(precision)
        self.seed, self.clip_norm = seed, clip_norm
        self.row0 = 0                     # global index of this rank's first sequence (data parallel)
        # weight of this generator's loss in the optimised objective: a mode that trains M per-track generators on the MEAN track loss
        # with one global-norm clip over all of them (multinn_jamming.py:235-241) sets 1/M; the generator's own metrics stay unscaled
        self.grad_scale = 1.0
        self.need_dx = False              # a feedback mode sets it: backward() then also leaves d loss / d inputs in self._dx (f32 [T,B,n_in])
        self._dx = None
        self.store = ParamStore(device)
        self._gen = torch.Generator().manual_seed(seed)
        self._num_inputs = num_inputs
        self._packed_step = -1
        self._init_rnn()
        self._init_estimator()

    # -- construction ---------------------------------------------------------------------------
    def _init_rnn(self):
        self._rnn = RNN(num_units=self.num_hidden_rnn, keep_prob=self.keep_prob)

    @abc.abstractmethod
    def _init_estimator(self):
        ...

    def _materialize(self, num_inputs):
        if self.store.theta is not None:
            return
        self._num_inputs = self._num_inputs or num_inputs
        self._declare(self._num_inputs)
        self.store.materialize()
        self._stack = LstmStack(self._rnn, self.store, self.dtype)
        self._stack.owner = weakref.ref(self)            # (weak: a cycle would leave dead generators -- and their captured graphs -- to the garbage
                                                         #  collector, which may then run in the middle of another capture and abort the process)
        self._trainable_variables = [self.store[n] for n in self.store.names()]
        self._variables = dict(self.store.views)

    def _get_rnn_zero_state(self, batch_size):
        return self._rnn.zero_state(batch_size, self.dtype)

    def _ensure_packed(self):
        if self._packed_step != self.store.step or self._stack.packed is None:
            self._stack.pack()
            self._pack_estimator()
            self._packed_step = self.store.step

    # -- layout helpers -------------------------------------------------------------------------
    def _to_time_major_inputs(self, x):
        """x [B,T,Din] (u8/float) -> [T,B,ld0] compute dtype, zero padded."""
        B, T, Din = x.shape
        out = torch.zeros((T, B, self._stack.ld0), device=x.device, dtype=self.dtype)
        out[:, :, :Din] = x.transpose(0, 1).to(self.dtype)
        return out

    # set by a caller that captures ragged steps (the mode classes' graphed_train_step): row weights, valid-row count and f16 loss scale of a
    # ragged window are then derived on the device, without a host read
    ragged_on_device = False
    _ls_dev = None

    def _row_weight(self, lengths, B, T, device):
        """1/N_valid on valid rows (N_valid summed over ALL ranks), 0 on padding.  Leaves the host copy of N_valid in self._n_valid (the
        f16 loss scale is derived from it)."""
        if lengths is None:
            # full-length batches: every rank holds B*T valid rows, the total is known on the host -- no copy, no collective (this
            # path runs inside captured steps)
            n_ranks = torch.distributed.get_world_size() if (dp_active() and torch.distributed.is_initialized()) else 1
            self._n_valid = B * T * n_ranks
            return torch.full((T * B,), 1.0 / float(B * T * n_ranks), device=device)
        mask = (torch.arange(T, device=device)[:, None] < lengths.to(device)[None, :]).float()
        n_tot = mask.sum()
        if dp_active() and not torch.cuda.is_current_stream_capturing():
            torch.distributed.all_reduce(n_tot)
        if self.ragged_on_device:
            # nothing is read on the host (a captured ragged step of a mode class: MultINNCore.graphed_train_step(lengths=...)): the valid-row
            # count stays a device scalar, and so does the f16 loss scale derived from it (LstmStack.loss_scale's rule, on the device)
            n_tot = n_tot.clamp_min(1.0)
            self._n_valid = None
            self._ls_dev = torch.exp2(torch.round(torch.log2(self._stack.loss_scale_rows * n_tot))).reshape(1) if self.dtype == torch.float16 else None
            return (mask / n_tot).reshape(-1).contiguous()
        self._n_valid = max(int(n_tot), 1)              # ragged windows run eagerly: a host read is allowed here
        return (mask / n_tot).reshape(-1).contiguous()

    # The sampling scans -- generate(), and the feedback modes' scans through steps() / single_step() -- run in DETERMINISTIC f32 arithmetic
    # on the master weights whatever the training precision is (csrc/det_step.hip; the reference samples in f32 too): every LSTM step, Dense
    # and conditional is a fixed sequence of IEEE operations that the tests' C checker restates, so a whole scan is checked bit for bit
    # (BASELINE.json: "bit-exact for Bernoulli sampling indices under a fixed RNG").  MULTINN_DET_SAMPLING=0 restores the throughput
    # kernels (hardware exp2 / rcp activations, packed 16-bit weights), whose scans can only be checked to a tolerance.
    det_sampling = os.environ.get("MULTINN_DET_SAMPLING", "1") != "0"

    def steps(self, inputs, initial_state=None):
        """rnn_estimator.py:237-252: run the RNN over `inputs` [B,T,Din] and return the state after the last step."""
        if self.det_sampling:
            self._materialize(inputs.shape[-1])
            if inputs.dim() == 2:
                inputs = inputs[:, None, :]
            x = inputs if inputs.dtype in (torch.uint8, torch.float32) else inputs.float()
            st = None if initial_state is None else [(c, h) for c, h in initial_state.rnn_state]
            h = None
            for t in range(x.shape[1]):
                h, st = self._stack.det_step(x[:, t], st)
            return self._det_state(h, st)
        return self._get_state(inputs, initial_state=initial_state, last_outputs=True)

    def _scan_in_one_call(self, x, num_steps):
        return None                         # estimators without a one-call scan (RnnRBM) step through sample_single / single_step

    def _det_single_step(self, inputs, initial_state, x2=None):
        """single_step in the deterministic arithmetic: inputs u8 | f32 [B, n_x]; x2 (optional, f32 [B, F]) is concatenated behind it -- the
        feedback vector of multinn_feedback.py:85-91, read in place instead of through a torch.cat."""
        h, new = self._stack.det_step(inputs, [(c, hh) for c, hh in initial_state.rnn_state], x2=x2)
        return self._det_state(h, new)

    def check(self, tolerate_overflow=False):
        """Raise if a persistent recurrence launch of this generator ever gave up on a bounded spin (LstmStack.check)."""
        if getattr(self, "_stack", None) is not None:
            self._stack.check()
        # an optimiser step the device skipped (non-finite gradient norm) raises here -- unless the caller is a training loop in precision "fp16",
        # whose dynamic loss scale has already answered the overflow (ParamStore.check)
        self.store.check(tolerate_overflow and self.dtype == torch.float16)

    def _unscale(self, ls):
        """End of a loss-scaled backward pass (LstmStack.loss_scale): gradients and d loss / d inputs back to their true scale."""
        if torch.is_tensor(ls):                         # compacted ragged window: ls is the device word 1 / scale (ops.ragged_index)
            self.store.grad.mul_(ls)
            if self._dx is not None:
                self._dx.mul_(ls)
        elif ls != 1.0:
            ops.axpby(1.0 / ls, self.store.grad, 0.0, None, self.store.grad)
            if self._dx is not None:
                flat = self._dx.view(-1)
                ops.axpby(1.0 / ls, flat, 0.0, None, flat)

    def graphed_build_train(self, x, y, optimizer, lr=None, warmup=2):
        """The generic captured optimiser step: build(x, y, None, True, 'train') + train(optimizer, lr) as hipGraph replays, for the
        generators that train on encoder outputs rather than on a raw piano-roll batch (RnnRBM: jamming mode, RnnMultiNADE: composer
        mode; RnnNade.graphed_train_step is the fused piano-roll form).  Eagerly such a step is host-bound (RnnRBM at [256,128,88]:
        kernels 4 ms, wall 14 ms).  Returns run(x=None, y=None) -> loss.  Step-dependent values (dropout seed, Gibbs seed, Adam step)
        are read from store.step_dev on the device; under data parallelism the gradient all-reduce stays an eager call between two
        graphs.  Full-length batches only."""
        from .training import allreduce_flat
        sx, sy = x.clone(), y.clone()
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.build(sx, sy, None, True, "train")
                self.train(optimizer, lr)
        cur.wait_stream(side)
        multi = dp_active()
        g_fb, g_opt = torch.cuda.CUDAGraph(), None
        with graph_capture(g_fb, capture_error_mode="thread_local"):
            self.build(sx, sy, None, True, "train")
            if multi:
                self.backward()
            else:
                self.train(optimizer, lr)
            loss = self._loss
        if multi:
            g_opt = torch.cuda.CUDAGraph()
            with graph_capture(g_opt, pool=g_fb.pool(), capture_error_mode="thread_local"):
                self._grad_sumsq = compute_gradients(optimizer, self.store, self.clip_norm, lr, reduce=False)
        self._packed_step = -1
        self.store.step -= 1            # the captured step has not executed (host mirror of store.step_dev)

        def run(x=None, y=None):
            if x is not None:
                sx.copy_(x)
            if y is not None:
                sy.copy_(y)
            g_fb.replay()
            if g_opt is not None:
                allreduce_flat(self.store.grad)
                g_opt.replay()
            self.store.step += 1
            return loss
        run.graph = g_fb
        return run

    def _step_input(self, B, device):
        """[B, ld0] staging row block of single_step: the zero padding beyond the input width is written once, every step converts its
        input into the prefix (the step's GEMM has read the previous contents by then: same stream)."""
        buf = getattr(self, "_xstep", None)
        if buf is None or buf.shape[0] != B or buf.device != torch.device(device) or buf.shape[1] != self._stack.ld0 \
                or torch.cuda.is_current_stream_capturing() != getattr(self, "_xstep_captured", False):
            buf = self._xstep = torch.zeros((B, self._stack.ld0), device=device, dtype=self.dtype)
            self._xstep_captured = torch.cuda.is_current_stream_capturing()
        return buf

    # -- sampling -------------------------------------------------------------------------------
    def generate(self, x, num_steps):
        """rnn_estimator.py:271-298: intro pass, then num_steps x {sample_single, single_step}.
        x [B,Ti,Din]; returns samples u8 [B,num_steps,num_output].  On the device the whole scan is ONE hipGraph
        replay (captured per shape / num_steps / seed, see common.ScanGraphs); same kernels, same RNG counters, same bits."""
        if not ScanGraphs.enabled(x):
            return self._generate_scan(x, num_steps)
        if getattr(self, "_scan_graphs", None) is None:
            self._scan_graphs = ScanGraphs()
        key = (tuple(x.shape), x.dtype, int(num_steps), self.seed, self.row0)

        def scan(sx):
            self._packed_step = -1                     # pack inside the graph: a replay always sees the current weights
            return self._generate_scan(sx, num_steps)

        def after():
            self._packed_step = -1                     # the packed copies now live in the graph's pool

        return self._scan_graphs.run(key, x, scan, lambda sx: self._generate_scan(sx, min(int(num_steps), 2)), after)

    def _generate_scan(self, x, num_steps):
        self._materialize(x.shape[-1])
        self._rnn.build_cell(False)
        if self.det_sampling:
            whole = self._scan_in_one_call(x, num_steps)        # LSTM-(Multi)NADE on a byte piano-roll: mnn_generate_scan runs the whole scan
            if whole is not None:
                return whole
            state = self.steps(x)
        else:
            self._ensure_packed()
            state = self._get_state(x, lengths=None, last_outputs=True)
        intro = x[:, -1, :]
        out = []
        for s in range(num_steps):
            self._gen_step = s
            samples, _ = self.sample_single(intro, state)
            state = self.single_step(samples, state)
            intro = samples
            out.append(samples)
        return torch.stack(out, 1)


# ------------------------------------------------------------------------------------------------
class RnnNade(RnnEstimator):
    """models/generators/rnn_nade.py: LSTM -> Dense -> NADE."""

    def __init__(self, num_dims, num_hidden, num_hidden_rnn, keep_prob=1.0, internal_bias=False, name="rnn-nade", track_name="all", **kw):
        self._tracks = getattr(self, "_tracks", ["all"])
        super().__init__(num_dims, num_hidden, num_hidden_rnn, keep_prob, internal_bias, name, track_name, **kw)
        self._num_output = self.num_tracks * self.num_dims

    tracks = property(lambda self: self._tracks)
    num_tracks = property(lambda self: len(self._tracks))

    def _init_estimator(self):
        # internal_bias=True (nade.py:69-87, rnn_nade.py:245-251): the NADE's own b_enc / b_dec are ADDED to the Dense outputs.  A broadcast
        # add in front of the scan is the same as adding them to the Dense layer's bias vector, so the kernels never see them: the forward GEMM
        # takes dense/bias + [b_enc | b_dec] (one axpby over n_out words), and their gradient is the Dense bias gradient (same column sums).
        self._nades = [NADE(self.num_dims, self.num_hidden[-1], internal_bias=False, name=f"nade_{m}") for m in range(self.num_tracks)]
        self._nade = self._nades[0]

    def _declare(self, num_inputs):
        """Variable order rnn, nade(s), dense (rnn_nade.py:117-120)."""
        M, D, Hn, R = self.num_tracks, self.num_dims, self.num_hidden[-1], self.num_hidden_rnn[-1]
        self._rnn.declare(self.store, num_inputs, self._gen)
        # all tracks' NADE weights are contiguous so that the kernels see [tracks, D, Hn]
        from .common import truncated_normal
        std = 1.0 / (D ** 0.5)
        self.store.declare("nade/w_enc", (M, D, Hn), truncated_normal(self._gen, std))
        self.store.declare("nade/w_dec", (M, D, Hn), truncated_normal(self._gen, std))
        if self.internal_bias:                          # adjacent in the flat buffer, in the Dense output's column order [tracks x Hn | tracks x D]
            self.store.declare("nade/b_enc", (M, Hn), truncated_normal(self._gen, std))
            self.store.declare("nade/b_dec", (M, D), truncated_normal(self._gen, std))
        n_out = M * (D + Hn)
        self.store.declare("dense/kernel", (R, n_out), glorot_uniform(self._gen, R, n_out))
        self.store.declare("dense/bias", (n_out,), zeros_init)
        self.n_out = n_out
        self.ldo = ops.round_up(n_out, 64)

    def _materialize(self, num_inputs):
        first = self.store.theta is None
        super()._materialize(num_inputs)
        if first:
            for m, nd in enumerate(self._nades):            # per-track views of the stacked NADE weights
                nd._w_enc_t, nd._w_dec_t = self.store["nade/w_enc"][m], self.store["nade/w_dec"][m]

    def _pack_estimator(self):
        dev = self.store.theta.device
        R = self.num_hidden_rnn[-1]
        self._fc_t = torch.empty((self.n_out, R), device=dev, dtype=self.dtype)        # [n_out, R]: forward B operand
        ops.transpose(self.store["dense/kernel"], self._fc_t)
        self._fc_p = torch.zeros((R, self.ldo), device=dev, dtype=self.dtype)           # [R, n_out]: dgrad B operand
        ops.convert2d(self.store["dense/kernel"], self._fc_p[:, :self.n_out])
        if self.internal_bias:
            self._fc_bias = torch.empty(self.n_out, device=dev)
            ops.axpby(1.0, self.store["dense/bias"], 1.0, self._internal_flat(self.store.theta), self._fc_bias)
        else:
            self._fc_bias = self.store["dense/bias"]
        if self._nade_mfma():                           # 16-bit copy of the decoder weights for the matrix-core NADE kernels
            M, D, Hn = self.num_tracks, self.num_dims, self.num_hidden[-1]
            if self._nade_exact():                      # fp16 mode: the weights as f16 hi | lo pairs for the split-operand MFMA form
                self._wdec_bf = torch.empty((M, D, Hn), device=dev, dtype=torch.float32)
                ops.nade_f32_pack(self.store["nade/w_dec"].view(M * D, Hn), self._wdec_bf.view(M * D, Hn))
            else:
                self._wdec_bf = torch.empty((M, D, Hn), device=dev, dtype=torch.bfloat16)
                ops.convert2d(self.store["nade/w_dec"].view(M * D, Hn), self._wdec_bf.view(M * D, Hn))

    def _internal_flat(self, flat):
        """[b_enc | b_dec] of all tracks as one n_out-long slice of a flat parameter-shaped buffer (theta or its gradient)."""
        o = self.store.offset("nade/b_enc")
        return flat[o:o + self.n_out]

    nade_mfma = os.environ.get("MULTINN_NADE_MFMA", "1") != "0"

    nade_dense_above = float(os.environ.get("MULTINN_NADE_DENSE_ABOVE", "0.07"))   # density above which the f32 scan replaces the matrix-core form

    def _nade_fwd(self, v, out, rw, nll, cond_p, d_out, a_fin, n_rows_dev=None, unsafe=None):
        M, D, Hn = self.num_tracks, self.num_dims, self.num_hidden[-1]
        exact = self._nade_exact()
        if self.nade_dense_above >= 1.0:                 # gate off: always the matrix-core form
            return ops.nade_logprob_fwd_auto(v, out, self.store["nade/w_enc"], self.store["nade/w_dec"], self._wdec_bf, M, D, Hn, None, None,
                                             1.0, rw, nll, cond_p, d_out, a_fin, exact=exact, n_rows_dev=n_rows_dev)
        if getattr(self, "_gate", None) is None or self._gate.device != out.device:
            self._gate = torch.zeros(1 + ops.DENSITY_SLOTS, device=out.device, dtype=torch.int32)            # [gate | partial counts]
        counted = bool(getattr(self, "_v_counted", False)) and rw is not None      # (the on-demand conditionals pass of a train build runs later: not counted)
        return ops.nade_logprob_fwd_auto(v, out, self.store["nade/w_enc"], self.store["nade/w_dec"], self._wdec_bf, M, D, Hn, self._gate[:1],
                                         self._gate[1:], self.nade_dense_above, rw, nll, cond_p, d_out, a_fin, exact=exact, counted=counted,
                                         n_rows_dev=n_rows_dev, unsafe=unsafe)

    # fp16 mode: the split-operand matrix-core scan (nade_mfma.hip, SPLIT: f16 hi + lo pairs, three 16-bit MFMA products) or the f32 vector scan
    nade_exact = os.environ.get("MULTINN_NADE_EXACT_MFMA", "1") != "0"

    def _nade_mfma(self):
        """The matrix-core NADE forward is in use: bf16 mode (16-bit operands) or fp16 mode (hi + lo operand pairs, 22 bits), and a hidden width it covers;
        otherwise the f32 VALU kernels."""
        if not (self.nade_mfma and ops.nade_mfma_ok(self.num_hidden[-1])):
            return False
        return self.dtype == torch.bfloat16 or (self.dtype == torch.float16 and self.nade_exact)

    def _nade_exact(self):
        """fp16 mode: BASELINE.json's 1e-4 on every conditional needs more than the 8 / 11 bits of a single bf16 / f16 operand in the decoder
        dot products (the LSTM / Dense operands do not): the matrix-core scan carries them as f16 hi + lo pairs there."""
        return self.dtype == torch.float16

    # -- forward --------------------------------------------------------------------------------
    def build(self, x=None, y=None, lengths=None, is_train=None, mode="eval"):
        """rnn_nade.py:64-124.  x inputs [B,T,Din], y targets [B,T,tracks*D] (u8 or float)."""
        return drive(self._build_co(x, y, lengths, is_train, mode))

    def _build_co(self, x=None, y=None, lengths=None, is_train=None, mode="eval"):
        """`build` as a generator function (drive / drive_group: the LSTM recurrences are yielded to the driver)."""
        Generator.build(self, x, y, lengths, is_train, mode)
        self._materialize(x.shape[-1] if x is not None else self._num_inputs)
        self._rnn.build_cell(is_train)
        if mode in ("train", "eval"):
            B, T, _ = x.shape
            M, D = self.num_tracks, self.num_dims
            x_tm = self._to_time_major_inputs(x)
            # rnn_multinade.py:97-101: reshape(flat,[-1,D,M]) unstacked on the last axis (track-minor)
            v = y.to(torch.uint8).transpose(0, 1).reshape(T, B, D, M).permute(3, 0, 1, 2).contiguous() if M > 1 \
                else y.to(torch.uint8).transpose(0, 1).contiguous().view(1, T, B, D)
            compact = None
            if lengths is not None and mode == "train" and self.dtype in ops.H16 and self.ragged_compact and x.is_cuda:
                # ragged window of a generator that trains on encoder outputs (composer: RnnMultiNADE; jamming / feedback with NADE generators):
                # Dense + NADE on the valid rows only, as build_pianoroll does for the joint mode (see there); the targets and row weights are
                # brought into compact order here (the piano-roll pass does it for the joint path)
                dev = x.device
                len_dev = lengths.to(device=dev, dtype=torch.int32).contiguous()
                n_total_dev = None
                if dp_active():
                    n_total_dev = len_dev.clamp(0, T).sum().float().reshape(1)
                    torch.distributed.all_reduce(n_total_dev)
                idx, inv, hdr = ops.ragged_index(len_dev, B, T, n_total_dev, self._stack.loss_scale_rows if self.dtype == torch.float16 else 0.0)
                compact = dict(idx=idx, inv=inv, hdr=hdr, hdr_f=hdr.view(torch.float32))
                v = v.view(M, T * B, D).index_select(1, idx.long()).view(M, T, B, D)          # every row of the permutation: padding rows behind the valid ones
                k = torch.arange(T * B, device=dev, dtype=torch.int32)
                rw = torch.where(k < hdr[0], compact["hdr_f"][1], torch.zeros((), device=dev))
                lengths = len_dev
                self._n_valid = None
            else:
                rw = self._row_weight(lengths, B, T, x.device)
            yield from self._forward_tm_co(x_tm, v, rw, lengths, B, T, train=(mode == "train"), compact=compact)
        self._is_built = True

    # MULTINN_RAGGED_COMPACT=0: ragged windows keep their padding rows in the Dense + NADE part (weight 0), as before round 6
    ragged_compact = os.environ.get("MULTINN_RAGGED_COMPACT", "1") != "0"
    ragged_gemm_rows = os.environ.get("MULTINN_RAGGED_GEMM_ROWS", "1") != "0"     # the Dense GEMMs of a compacted window skip their padding too (ops.gemm_tn m_rows / k_rows)

    def build_pianoroll(self, x_u8, lengths=None, is_train=True, mode="train", n_total_dev=None):
        """Fast joint path: x_u8 [B,T,P,M] piano-roll batch; fuses multinn_joint.py:83-89,132-139
        (zero first step, inputs = enc[:, :-1], targets = enc[:, 1:]) into one kernel.
        n_total_dev (f32 [1], optional): valid rows of ALL ranks of a ragged window, for callers that must not run a collective here
        (a captured step under data parallelism)."""
        if self.num_tracks != 1:
            raise ValueError("build_pianoroll is the joint (single NADE) path")
        B, T, P, Mtr = x_u8.shape
        D = P * Mtr
        self._materialize(D)
        Generator.build(self, None, None, lengths, is_train, mode)
        self._rnn.build_cell(is_train)
        dev = x_u8.device
        x_tm = torch.empty((T, B, self._stack.ld0), device=dev, dtype=self.dtype)
        v = torch.empty((T, B, D), device=dev, dtype=torch.uint8)
        rw = torch.empty(T * B, device=dev)
        # Ragged window, 16-bit train step: Dense + NADE run on the VALID rows only (the reference drops padded rows before the NADE:
        # utils/sequences.py:6-37, rnn_nade.py:91-92,225; the LSTM still steps them: impute_finished=False).  The compaction index, the row
        # count, 1 / n_valid and the f16 loss scale all live on the device (ops.ragged_index): nothing of the step depends on a host-side
        # count, so graphed_train_step captures it once for any lengths.
        compact = None
        if lengths is not None and mode == "train" and self.dtype in ops.H16 and self.ragged_compact:
            len_dev = lengths.to(device=dev, dtype=torch.int32).contiguous()
            if n_total_dev is None and dp_active():
                n_total_dev = len_dev.clamp(0, T).sum().float().reshape(1)
                torch.distributed.all_reduce(n_total_dev)
            idx, inv, hdr = ops.ragged_index(len_dev, B, T, n_total_dev, self._stack.loss_scale_rows if self.dtype == torch.float16 else 0.0)
            compact = dict(idx=idx, inv=inv, hdr=hdr, hdr_f=hdr.view(torch.float32))
            lengths = len_dev
            n_valid = 0
            self._n_valid = None                    # device-side only (compact["hdr"])
        elif lengths is not None:
            n_tot = lengths.sum().to(dev).float()
            if dp_active():
                torch.distributed.all_reduce(n_tot)
            n_valid = int(n_tot)
            self._n_valid = max(n_valid, 1)
        else:
            n_valid = B * T * world()[1]
            self._n_valid = max(n_valid, 1)
        x_tmT = None
        if mode == "train" and self.dtype in ops.H16:              # the same pass also writes x^T, layer 1's weight-gradient operand
            self._ensure_packed()
            x_tmT = self._stack.input_T(T, B, dev)                  # (a view of the layer's concatenated operand where that form runs)
            if x_tmT is None:
                Np = ops.round_up(T * B, 64)
                x_tmT = (torch.zeros if Np != T * B else torch.empty)((self._stack.ld0, Np), device=dev, dtype=self.dtype)
        # the NADE forward's density gate needs the number of set target cells: counted by this pass while it writes them (no second pass over v)
        cnt = None
        if x_tmT is not None and self._nade_mfma() and self.nade_dense_above < 1.0:
            if getattr(self, "_gate", None) is None or self._gate.device != dev:
                self._gate = torch.zeros(1 + ops.DENSITY_SLOTS, device=dev, dtype=torch.int32)            # [gate | partial counts]
            cnt = self._gate[1:]
        self._v_counted = ops.pianoroll_shift_timemajor(x_u8.view(B, T, D), lengths, x_tm, v, rw, n_valid, inputs_t=x_tmT, count=cnt,
                                                        compact=(compact["inv"], compact["hdr"]) if compact else None)
        self._forward_tm(x_tm, v.view(1, T, B, D), rw, lengths, B, T, train=(mode == "train"), x_tmT=x_tmT, compact=compact)
        self._v_counted = False
        self._is_built = True

    def _forward_tm(self, x_tm, v, rw, lengths, B, T, train, x_tmT=None, compact=None):
        return drive(self._forward_tm_co(x_tm, v, rw, lengths, B, T, train, x_tmT, compact))

    def _forward_tm_co(self, x_tm, v, rw, lengths, B, T, train, x_tmT=None, compact=None):
        """compact (build_pianoroll, ragged windows): v and rw arrive in COMPACT row order; the LSTM output is gathered into it (rows behind the
        valid ones zeroed) and everything from the Dense layer on -- out, nll, d_out, a_fin -- lives in compact order too."""
        M, D, Hn = self.num_tracks, self.num_dims, self.num_hidden[-1]
        N, dev = T * B, x_tm.device
        self._ensure_packed()
        kp = self._rnn.effective_keep_prob()
        y, ctx, _ = yield from self._stack.forward_co(x_tm, kp, self.seed, self.row0, save=train, step_dev=self.store.step_dev)
        if ctx and x_tmT is not None:
            ctx[0]["inT"] = x_tmT
        nrows = None
        if compact is not None:
            R = y.shape[-1]
            Np = ops.round_up(N, 64)
            y_c = torch.empty((N, R), device=dev, dtype=self.dtype)
            y_cT = (torch.zeros if Np != N else torch.empty)((R, Np), device=dev, dtype=self.dtype) if train else None
            ops.rows_gather16(y.view(N, R), compact["idx"], compact["hdr"], y_c, y_cT)
            compact["y_cT"] = y_cT
            y_dense, nrows = y_c, compact["hdr"]
        else:
            y_dense = y.view(N, -1)
        out = torch.empty((N, self.ldo), device=dev)      # columns >= n_out are padding of the row pitch: never read
        ops.gemm_tn(y_dense, self._fc_t, out[:, :self.n_out], bias=self._fc_bias, m_rows=nrows if self.ragged_gemm_rows else None)      # (compact: row tiles of padding are skipped)
        nll = torch.empty((M, N), device=dev)
        cond_p = None if train else torch.empty((M, N, D), device=dev)     # the train step needs the loss only: 4 N D bytes less to write per
        rw_m = rw / M if M > 1 else rw                                       # step; `cond_probs` fills it on demand (see the property)
        d_out = None
        if train:
            d_out = torch.empty((N, self.ldo), device=dev)
            if self.ldo != self.n_out and self.dtype == torch.float32:
                d_out[:, self.n_out:].zero_()       # fp32: d_out itself is the dgrad operand; bf16: grad_rows_fanout writes the zero padding
        a_fin = torch.empty((M, N, Hn), device=dev) if train else None
        # gradient seed only (the reported loss stays unscaled): the mode's weight of this generator's loss, and the f16 loss scale
        # (f16: times the DYNAMIC multiplier m of the store, a device word a skipped step halves -- ParamStore.ls_dyn; _unscale gets 1 / (scale m))
        dyn = self.store.ls_dyn if (train and self.dtype == torch.float16) else None
        if compact is not None:                         # the scale is a device word (hdr[2]); its inverse is applied by _unscale
            ls = compact["hdr_f"][2:3] if (train and self.dtype == torch.float16) else 1.0
            if torch.is_tensor(ls):
                rw_g = rw_m * (ls * dyn[0:1] * self.grad_scale)
                ls = compact["hdr_f"][3:4] * dyn[1:2]    # what _unscale multiplies by
            else:
                rw_g = rw_m if self.grad_scale == 1.0 else rw_m * self.grad_scale
        elif train and self._n_valid is None and self.dtype == torch.float16:     # ragged_on_device without compaction: the device-side scale
            rw_g = rw_m * (self._ls_dev * dyn[0:1] * self.grad_scale)
            ls = dyn[1:2] / self._ls_dev
        else:
            ls = self._stack.loss_scale(self._n_valid) if (train and self._n_valid is not None) else 1.0
            gs = self.grad_scale * ls
            if dyn is not None:
                rw_g = rw_m * (dyn[0:1] * gs)
                ls = dyn[1:2] * (1.0 / ls)               # a tensor: _unscale multiplies by it
            else:
                rw_g = rw_m if gs == 1.0 else rw_m * gs
        if self._nade_mfma():
            # bf16 compute mode: the decoder dot products run as a block-sparse bf16 GEMM over each row's hidden states while the batch is
            # piano-roll-sparse; a dense batch takes the f32 vector form (decided on the device, per launch: ops.nade_logprob_fwd_auto)
            # (train, density gate in use: the dense forward also counts the waves that left |a| <= 40 -- none = the dense backward's licence, see ops.nade_logprob_bwd)
            unsafe = torch.empty(1, device=dev, dtype=torch.int32) if (train and self.nade_dense_above < 1.0) else None
            self._nade_fwd(v.view(M, N, D), out, rw_g if train else None, nll, cond_p, d_out, a_fin, n_rows_dev=nrows, unsafe=unsafe)
        else:
            assert compact is None or Hn <= 256, "compacted rows: the f32 scan covers Hn <= 256"
            ops.nade_logprob_fwd(v.view(M, N, D), out, self.store["nade/w_enc"], self.store["nade/w_dec"], M, D, Hn,
                                 rw_g if train else None, nll, cond_p, d_out, a_fin, n_rows_dev=nrows)
        loss = torch.zeros(1, device=dev)
        ops.weighted_sum(nll.view(-1), rw_m.repeat(M) if M > 1 else rw_m, loss)      # statistical.py:34 / rnn_multinade.py:202-203
        self._ctx = dict(x_tm=x_tm, v=v, rw=rw_m, y=y, lstm=ctx, out=out, d_out=d_out, a_fin=a_fin, kp=kp, seed=self.seed, B=B, T=T, ls=ls,
                         compact=compact, unsafe=unsafe if self._nade_mfma() else None)
        self._nll_tm, self._cond_tm, self._loss = nll, cond_p, loss
        self._flat_idx = None
        self._lengths = lengths
        self._metrics = {"batch/loss": loss, "log_likelihood": loss}
        self._metrics_upd = []

    # -- API-order views of the flat outputs (b-major then t, sequences.py:30-31) ---------------
    def _idx(self):
        """Positions of the API-order rows (b-major, then t: sequences.py:30-31) in the arrays the kernels wrote: time-major rows, or -- for a
        compacted ragged window -- compact rows (through the window's inverse permutation)."""
        if self._flat_idx is None:
            fi = flat_index(self._lengths, self._ctx["B"], self._ctx["T"], self._loss.device)
            cp = self._ctx.get("compact")
            self._flat_idx = cp["inv"].long()[fi] if cp is not None else fi
        return self._flat_idx

    @property
    def log_probs(self):
        r = [self._nll_tm[m][self._idx()] for m in range(self.num_tracks)]
        return r[0] if self.num_tracks == 1 else r

    @property
    def cond_probs(self):
        if self._cond_tm is None:           # built in train mode: one more decoder pass over the saved Dense output, conditionals only
            cx = self._ctx
            M, D, Hn = self.num_tracks, self.num_dims, self.num_hidden[-1]
            N = cx["B"] * cx["T"]
            cp = torch.empty((M, N, D), device=cx["out"].device)
            if self._nade_mfma():
                self._nade_fwd(cx["v"].view(M, N, D), cx["out"], None, None, cp, None, None,
                               n_rows_dev=cx["compact"]["hdr"] if cx.get("compact") else None)
            else:
                ops.nade_logprob_fwd(cx["v"].view(M, N, D), cx["out"], self.store["nade/w_enc"], self.store["nade/w_dec"], M, D, Hn, None, None, cp, None,
                                     None)
            self._cond_tm = cp
        r = [self._cond_tm[m][self._idx()] for m in range(self.num_tracks)]
        return r[0] if self.num_tracks == 1 else r

    @property
    def _outputs(self):
        """rnn_nade.py:100: cast(cond_probs >= .5)."""
        cp = self.cond_probs
        return (cp >= 0.5).float() if self.num_tracks == 1 else [(c >= 0.5).float() for c in cp]

    def build_metrics(self, targets, predictions, cond_probs=None, log_probs=None):
        return self._nade.build_metrics(targets, predictions, cond_probs, log_probs)

    # -- backward -------------------------------------------------------------------------------
    def backward(self):
        """Gradient of metrics['batch/loss'] wrt rnn + nade + dense variables into store.grad."""
        return drive(self._backward_co())

    def _backward_co(self):
        cx = self._ctx
        if cx["d_out"] is None:
            raise RuntimeError("build(..., mode='train') must run before train()")
        M, D, Hn, R = self.num_tracks, self.num_dims, self.num_hidden[-1], self.num_hidden_rnn[-1]
        B, T = cx["B"], cx["T"]
        N, dev = B * T, cx["out"].device
        g = self.store.gviews
        self.store.grad.zero_()
        d_out = cx["d_out"]
        compact = cx.get("compact")
        ops.nade_logprob_bwd(cx["v"].view(M, N, D), cx["out"], self.store["nade/w_enc"], self.store["nade/w_dec"], M, D, Hn, cx["a_fin"],
                             d_out, g["nade/w_enc"], g["nade/w_dec"], n_rows_dev=compact["hdr"] if compact else None,
                             unsafe=cx.get("unsafe"))
        # dense: dK[R,n_out] = y^T d_out ; db = sum d_out ; dy = d_out K^T
        Np = ops.round_up(N, 64)
        zalloc = torch.zeros if Np != N else torch.empty
        yT = cx["lstm"][-1].get("yT") if cx["lstm"] else None      # emitted by the persistent recurrence
        if compact is not None:
            yT = compact["y_cT"]                                    # compact row order, like d_out (rows behind the valid ones are zero in both)
        if yT is None:
            yT = zalloc((R, Np), device=dev, dtype=self.dtype)
            ops.transpose(cx["y"].view(N, R), yT)
        doT = zalloc((self.n_out, Np), device=dev, dtype=self.dtype)
        if self.dtype == torch.float32:
            ops.transpose(d_out[:, :self.n_out], doT)
            ops.bias_grad(d_out[:, :self.n_out], g["dense/bias"], accumulate=True)
            do_c = d_out
        else:                               # one pass over d_out: bf16 copy, bf16 transpose, bias gradient
            do_c = torch.empty((N, self.ldo), device=dev, dtype=self.dtype)
            ops.grad_rows_fanout(d_out, self.n_out, do_c, doT, g["dense/bias"])
        if self.internal_bias:              # d b_enc / d b_dec = the Dense bias gradient (same column sums of d_out)
            gi = self._internal_flat(self.store.grad)
            ops.axpby(1.0, gi, 1.0, g["dense/bias"], gi)
        nrows = compact["hdr"] if (compact is not None and self.ragged_gemm_rows) else None      # compact: the row sums stop at the valid rows, padding row tiles are skipped
        ops.gemm_tn(yT, doT, g["dense/kernel"], accumulate=True, split_k=LstmStack._split_k(R, self.n_out, Np), k_rows=nrows)
        del yT, doT
        dy = torch.empty((N, R), device=dev)
        ops.gemm_tn(do_c, self._fc_p, dy, m_rows=nrows)
        if compact is not None:                                     # back into time-major order for the LSTM backward (padding rows: 0)
            dy_c, dy = dy, torch.empty((N, R), device=dev)
            ops.rows_scatter_f32(dy_c, compact["inv"], compact["hdr"], dy)
        self._dx = yield from self._stack.backward_co(dy.view(T, B, R), cx["lstm"], cx["kp"], cx["seed"], self.row0, need_dx=self.need_dx,
                                                      step_dev=self.store.step_dev)
        self._unscale(cx.get("ls", 1.0))

    def train_step(self, x_u8, lengths, optimizer, lr=None):
        """One optimiser step on a piano-roll batch (train.py:178-189's sess.run)."""
        self.build_pianoroll(x_u8, lengths, is_train=True, mode="train")
        self.train(optimizer, lr)
        return self._loss

    def graphed_train_step(self, x_u8, optimizer, lr=None, warmup=2, lengths=None):
        """Capture one whole optimiser step (plumbing, packing, forward, backward, clip, Adam: ~300 launches) into
        hipGraphs and return ``run(x=None, lengths=None) -> loss``: the T-step recurrences are launch-bound on the host otherwise.
        Step-dependent values (dropout seed, Adam step) are read from store.step_dev on the device.  Under data
        parallelism the ONE gradient all-reduce stays an eager torch.distributed call between two graphs
        (forward+backward | clip+Adam), so nothing of RCCL is captured.

        lengths (int32 [B], optional): capture the RAGGED step.  The captured graph holds a static copy of the lengths; the compaction index,
        the valid-row count, 1 / n_valid and the f16 loss scale are computed from it ON THE DEVICE inside the graph (ops.ragged_index), so one
        capture serves every later ``run(x, lengths)``.  16-bit modes only (the compacted path); under data parallelism the total row count of
        all ranks is all-reduced eagerly in front of the replay."""
        from .training import allreduce_flat, setup_cabi_comm
        setup_cabi_comm()               # MULTINN_COMM=capi: the communicator exists before anything is captured
        static_x = x_u8.clone()
        ragged = lengths is not None
        if ragged and not (self.dtype in ops.H16 and self.ragged_compact):
            raise ValueError("graphed_train_step(lengths=...) needs the compacted ragged path (16-bit precision, MULTINN_RAGGED_COMPACT unset)")
        dev = x_u8.device
        T = x_u8.shape[1]
        static_len = lengths.to(device=dev, dtype=torch.int32).clone() if ragged else None
        multi = dp_active()
        static_ntot = torch.zeros(1, device=dev) if (ragged and multi) else None

        def set_total():
            if static_ntot is not None:
                static_ntot.copy_(static_len.clamp(0, T).sum().float().reshape(1))
                torch.distributed.all_reduce(static_ntot)

        set_total()
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.build_pianoroll(static_x, static_len, is_train=True, mode="train", n_total_dev=static_ntot)
                self.train(optimizer, lr)
        cur.wait_stream(side)
        self._packed_step = -1          # the captured step packs the weights itself, whatever ran before (warmup = 0: a caller's own steps)
        g_fb, g_opt = torch.cuda.CUDAGraph(), None
        if not multi:
            with graph_capture(g_fb):
                self.build_pianoroll(static_x, static_len, is_train=True, mode="train")
                self.train(optimizer, lr)
                loss = self._loss
        else:
            # thread_local: the process group's watchdog thread may touch the runtime while this thread captures
            with graph_capture(g_fb, capture_error_mode="thread_local"):
                self.build_pianoroll(static_x, static_len, is_train=True, mode="train", n_total_dev=static_ntot)
                self.backward()
                loss = self._loss
            g_opt = torch.cuda.CUDAGraph()
            with graph_capture(g_opt, pool=g_fb.pool(), capture_error_mode="thread_local"):
                self._grad_sumsq = compute_gradients(optimizer, self.store, self.clip_norm, lr, reduce=False)
            self._packed_step = -1
        self.store.step -= 1            # the captured step has not executed (host mirror of store.step_dev)

        def run(x=None, lengths=None):
            if x is not None:
                static_x.copy_(x)
            if lengths is not None:
                if not ragged:
                    raise ValueError("this step was captured for full-length windows: capture it with lengths= to feed ragged ones")
                static_len.copy_(lengths.to(device=dev, dtype=torch.int32))
                set_total()
            g_fb.replay()
            if g_opt is not None:
                allreduce_flat(self.store.grad)
                g_opt.replay()
            self.store.step += 1
            return loss
        run.graph = g_fb
        run.ragged = ragged
        return run

    # -- state / sampling -----------------------------------------------------------------------
    def zero_state(self, batch_size):
        """rnn_nade.py:158-171 (RnnMultiNADE: per-track lists, R8)."""
        self._materialize(self._num_inputs)
        dev = self.store.theta.device
        z = lambda n: torch.zeros((batch_size, n), device=dev)
        be = z(self.num_hidden[-1]) if self.num_tracks == 1 else [z(self.num_hidden[-1]) for _ in self._tracks]
        bd = z(self.num_dims) if self.num_tracks == 1 else [z(self.num_dims) for _ in self._tracks]
        return RnnEstimatorStateTuple(be, bd, self._get_rnn_zero_state(batch_size))

    def _build_biases(self, outputs):
        """rnn_nade.py:234-251 / rnn_multinade.py:231-256: b_enc block(s) first, then b_dec block(s)."""
        M, D, Hn = self.num_tracks, self.num_dims, self.num_hidden[-1]
        be = [outputs[:, m * Hn:(m + 1) * Hn] for m in range(M)]
        bd = [outputs[:, M * Hn + m * D:M * Hn + (m + 1) * D] for m in range(M)]
        return (be[0], bd[0]) if M == 1 else (be, bd)

    def _dense(self, h):
        out = torch.empty((h.shape[0], self.ldo), device=h.device)        # columns [n_out, ldo) are alignment only: no kernel reads them
        ops.gemm_tn(h, self._fc_t, out[:, :self.n_out], bias=self._fc_bias)
        return out

    def _state_from_dense(self, out, rnn_state):
        be, bd = self._build_biases(out)
        st = RnnEstimatorStateTuple(be, bd, rnn_state)
        st.dense = out
        self._last_dense = out
        return st

    def _get_state(self, inputs, lengths=None, initial_state=None, last_outputs=False):
        """rnn_nade.py:173-232."""
        self._materialize(inputs.shape[-1])
        self._ensure_packed()
        if inputs.dim() == 2:
            inputs = inputs[:, None, :]
        B, T, _ = inputs.shape
        x_tm = self._to_time_major_inputs(inputs)
        st0 = [(c, h) for c, h in initial_state.rnn_state] if initial_state is not None else None
        y, _, final = self._stack.forward(x_tm, self._rnn.effective_keep_prob(), self.seed, self.row0, save=False, state0=st0)
        if last_outputs:
            out = self._dense(y[-1].contiguous())
        else:
            out = self._dense(y.view(T * B, -1))[flat_index(lengths, B, T, inputs.device)]
        return self._state_from_dense(out, tuple((c.clone(), h.clone()) for c, h in final))

    def _scan_in_one_call(self, x, num_steps):
        """rnn_estimator.py:271-298 through ONE C-ABI call (mnn_generate_scan: intro pass + num_steps x {NADE sample, LSTM step, Dense} enqueued
        by the library's own host loop) when the inputs are the byte piano-roll itself; the same kernels and bits as the step-by-step path."""
        if x.dtype != torch.uint8 or x.shape[-1] != self.num_tracks * self.num_dims or int(num_steps) < 1:
            return None
        pre = self._rnn.prefix
        layers = [(self.store[f"{pre}/cell_{l}/kernel"], self.store[f"{pre}/cell_{l}/bias"]) for l in range(len(self._rnn.num_units))]
        return ops.generate_scan(x.contiguous(), num_steps, layers, self.store["dense/kernel"], self._det_fc_bias(), self.num_tracks, self.num_dims,
                                 self.num_hidden[-1], self.store["nade/w_enc"], self.store["nade/w_dec"], 1.0, self.seed, self.row0)

    def _det_fc_bias(self):
        if not self.internal_bias:
            return self.store["dense/bias"]
        b = torch.empty(self.n_out, device=self.store.theta.device)          # dense/bias + [b_enc | b_dec] (see _init_estimator)
        ops.axpby(1.0, self.store["dense/bias"], 1.0, self._internal_flat(self.store.theta), b)
        return b

    def _det_dense_job(self, h):
        """(job, out): the Dense layer on h f32 [B, R] in the deterministic arithmetic (ops.dense_det), master weights in place."""
        out = torch.empty((h.shape[0], self.ldo), device=h.device)            # columns [n_out, ldo) are alignment only: no kernel reads them
        return dict(x=h, W=self.store["dense/kernel"], bias=self._det_fc_bias(), out=out[:, :self.n_out]), out

    def _det_state(self, h, rnn_state):
        job, out = self._det_dense_job(h)
        ops.dense_det([job])
        return self._state_from_dense(out, tuple(rnn_state))

    def single_step(self, inputs, initial_state, x2=None):
        """rnn_nade.py:253-277."""
        if self.det_sampling:
            return self._det_single_step(inputs, initial_state, x2)
        if x2 is not None:
            inputs = torch.cat([inputs.float(), x2], 1)
        x = self._step_input(inputs.shape[0], inputs.device)
        ops.convert2d(inputs.contiguous() if inputs.dtype in (torch.uint8, torch.float32, torch.bfloat16, torch.float16) else inputs.float(),
                      x[:, :inputs.shape[1]])
        h, new = self._stack.single_step(x, [(c, hh) for c, hh in initial_state.rnn_state])
        return self._state_from_dense(self._dense(h.contiguous()), tuple(new))       # views of buffers this step allocated: no copies

    def log_prob(self, inputs, targets_flat, lengths=None):
        """rnn_nade.py:279-302 (API-order outputs)."""
        state = self._get_state(inputs, lengths=lengths)
        return self._nade.log_prob(targets_flat, state.b_enc, state.b_dec)

    def sample_single(self, inputs, state, temperature=1.0):
        """rnn_nade.py:304-318 / rnn_multinade.py:295-317: returns (sample u8 [B,num_output], nll)."""
        M, D, Hn = self.num_tracks, self.num_dims, self.num_hidden[-1]
        out = state.dense if getattr(state, "dense", None) is not None else self._last_dense
        Bn = out.shape[0]
        smp = torch.empty((Bn, M * D), device=out.device, dtype=torch.uint8)
        nll = torch.empty((M, Bn), device=out.device)
        ops.nade_sample(out, self.store["nade/w_enc"], self.store["nade/w_dec"], M, D, Hn, temperature, self.seed, self.row0,
                        getattr(self, "_gen_step", 0), smp, track_minor=(M > 1), nll=nll)
        return smp, (nll[0] if M == 1 else [nll[m] for m in range(M)])


class RnnMultiNADE(RnnNade):
    """models/generators/rnn_multinade.py: one LSTM, one Dense, ``len(tracks)`` NADEs."""

    def __init__(self, num_dims, num_hidden, num_hidden_rnn, tracks, keep_prob=1.0, internal_bias=False, name="rnn-multinade", **kw):
        self._tracks = list(tracks)
        super().__init__(num_dims, num_hidden, num_hidden_rnn, keep_prob, internal_bias, name, track_name="all", **kw)


# ------------------------------------------------------------------------------------------------
class RnnRBM(RnnEstimator):
    """models/generators/rnn_rbm.py: LSTM -> (Wuh, Wuv) -> RBM with CD-k Gibbs sampling.

    Reference defects R1-R4 are resolved as recorded there: k = rbm.k in
    sample_single, lengths forwarded to _get_state, per-row free energy, and
    ``bias_mode='conditional'`` (Boulanger-Lewandowski) as the trained loss; 'internal'
    reproduces the as-written metric."""

    def __init__(self, num_dims, num_hidden, num_hidden_rnn, keep_prob=1.0, internal_bias=True, k=10, name="rnn-rbm", track_name="all",
                 bias_mode="conditional", **kw):
        self._k = k
        self.bias_mode = bias_mode
        super().__init__(num_dims, num_hidden, num_hidden_rnn, keep_prob, internal_bias, name, track_name, **kw)
        self._num_output = self.num_dims

    k = property(lambda self: self._k)

    def _init_estimator(self):
        self._rbm = RBM(self.num_dims, self.num_hidden[-1], k=self._k)

    def _declare(self, num_inputs):
        """Variable order rbm [W,bv,bh], rnn, [Wuh, Wuv] (rnn_rbm.py:135-138)."""
        D, Hn, R = self.num_dims, self.num_hidden[-1], self.num_hidden_rnn[-1]
        self._rbm.declare(self.store, self._gen)
        self._rnn.declare(self.store, num_inputs, self._gen)
        self.store.declare("Wuh", (R, Hn), glorot_uniform(self._gen, R, Hn))
        self.store.declare("Wuv", (R, D), glorot_uniform(self._gen, R, D))
        self.n_out = Hn + D
        self.ldo = ops.round_up(self.n_out, 64)

    def _pack_estimator(self):
        dev = self.store.theta.device
        R, Hn, D = self.num_hidden_rnn[-1], self.num_hidden[-1], self.num_dims
        wu = torch.cat([self.store["Wuh"], self.store["Wuv"]], 1).contiguous()          # [R, Hn+D]
        self._wu_t = torch.empty((self.n_out, R), device=dev, dtype=self.dtype)
        ops.transpose(wu, self._wu_t)
        self._wu_p = torch.zeros((R, self.ldo), device=dev, dtype=self.dtype)
        ops.convert2d(wu, self._wu_p[:, :self.n_out])
        self._bias_cat = torch.cat([self._rbm.bh.view(-1), self._rbm.bv.view(-1)]).contiguous() if self.internal_bias \
            else torch.zeros(self.n_out, device=dev)

    def _biases(self, h):
        """rnn_rbm.py:240-259: bh_t = rbm.bh + o.Wuh ; bv_t = rbm.bv + o.Wuv as ONE GEMM."""
        out = torch.empty((h.shape[0], self.ldo), device=h.device)
        if self.ldo != self.n_out:
            out[:, self.n_out:].zero_()
        ops.gemm_tn(h, self._wu_t, out[:, :self.n_out], bias=self._bias_cat)
        return out

    def build(self, x=None, y=None, lengths=None, is_train=None, mode="eval"):
        """rnn_rbm.py:71-143."""
        return drive(self._build_co(x, y, lengths, is_train, mode))

    def _build_co(self, x=None, y=None, lengths=None, is_train=None, mode="eval"):
        """`build` as a generator function (drive / drive_group: the LSTM recurrences are yielded to the driver)."""
        Generator.build(self, x, y, lengths, is_train, mode)
        self._materialize(x.shape[-1] if x is not None else self._num_inputs)
        self._rnn.build_cell(is_train)
        if mode in ("train", "eval"):
            B, T, _ = x.shape
            D, Hn = self.num_dims, self.num_hidden[-1]
            N, dev = B * T, x.device
            self._ensure_packed()
            x_tm = self._to_time_major_inputs(x)
            v0 = x.to(torch.uint8).transpose(0, 1).contiguous().view(N, -1)       # chain starts from inputs_flat (rnn_rbm.py:112)
            tgt = y.to(torch.uint8).transpose(0, 1).contiguous().view(N, D)
            rw = self._row_weight(lengths, B, T, dev)
            kp = self._rnn.effective_keep_prob()
            seed = self.seed + self.store.step
            yy, ctx, _ = yield from self._stack.forward_co(x_tm, kp, self.seed, self.row0, save=(mode == "train"), step_dev=self.store.step_dev)
            out = self._biases(yy.view(N, -1))
            bh_t, bv_t = out[:, :Hn], out[:, Hn:Hn + D]
            # global flat row ids keep the Gibbs uniforms independent of the data-parallel split
            rows = (torch.arange(T, device=dev)[:, None] * 65536 + (self.row0 + torch.arange(B, device=dev))[None, :]).reshape(-1).int()
            p_v = torch.empty((N, D), device=dev)
            v_s = torch.empty((N, D), device=dev, dtype=torch.uint8)
            # seed + step, the step read on the device (store.step_dev == store.step here): the same draws as passing seed + store.step,
            # and a captured step (graphed_build_train) draws anew at every replay
            ops.rbm_gibbs(v0[:, :D].contiguous(), self._rbm.W, bh_t, bv_t, self._k, self.seed, 0, rows, 0, p_v, v_s, seed_step=self.store.step_dev)
            if self.bias_mode == "conditional":
                bh_u, bv_u = bh_t, bv_t
            else:
                bh_u, bv_u = self._rbm.bh, self._rbm.bv
            Fv = torch.empty(N, device=dev); Fs = torch.empty(N, device=dev)
            # train mode: the same pass leaves sigmoid(z) of both chains' ends, the hidden activations of the free-energy gradient (backward)
            sv = torch.empty((N, Hn), device=dev) if mode == "train" else None
            ss = torch.empty((N, Hn), device=dev) if mode == "train" else None
            ops.rbm_free_energy(tgt, self._rbm.W, bh_u, bv_u, Fv, p_h=sv)
            ops.rbm_free_energy(v_s, self._rbm.W, bh_u, bv_u, Fs, p_h=ss)
            cost = Fv - Fs
            loss = torch.zeros(1, device=dev)
            ops.weighted_sum(cost, rw, loss)
            self._ctx = dict(y=yy, lstm=ctx, out=out, tgt=tgt, v_s=v_s, rw=rw, kp=kp, seed=seed, B=B, T=T, bh_u=bh_u, bv_u=bv_u,
                             n_valid=self._n_valid, ls_dev=self._ls_dev if self._n_valid is None else None, sv=sv, ss=ss)
            self._cost_tm, self._F_tm, self._pv_tm, self._vs_tm, self._loss = cost, Fv, p_v, v_s, loss
            self._lengths, self._flat_idx = lengths, None
            self._recon_tm = torch.empty(N, device=dev)
            ops.log_loss_rows(tgt, p_v, self._recon_tm)                                                       # rbm.py:124-129
            fe, ll = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
            ops.weighted_sum(Fv, rw, fe)
            ops.weighted_sum(self._recon_tm, rw, ll)
            self._metrics = {"batch/loss": loss, "free_energy": fe, "log_likelihood": ll}
            self._metrics_upd = []
        self._is_built = True

    def _idx(self):
        """Positions of the API-order rows (b-major, then t: sequences.py:30-31) in the arrays the kernels wrote: time-major rows, or -- for a
        compacted ragged window -- compact rows (through the window's inverse permutation)."""
        if self._flat_idx is None:
            fi = flat_index(self._lengths, self._ctx["B"], self._ctx["T"], self._loss.device)
            cp = self._ctx.get("compact")
            self._flat_idx = cp["inv"].long()[fi] if cp is not None else fi
        return self._flat_idx

    cond_probs = property(lambda self: self._pv_tm[self._idx()])
    _outputs = property(lambda self: self._vs_tm[self._idx()].float())
    free_energy = property(lambda self: self._F_tm[self._idx()])
    cost = property(lambda self: self._cost_tm[self._idx()])
    reconstruction_cost = property(lambda self: self._recon_tm[self._idx()])

    def build_metrics(self, targets, predictions, cond_probs=None, log_probs=None):
        return self._rbm.build_metrics(targets, predictions, cond_probs, log_probs)

    def backward(self):
        return drive(self._backward_co())

    def _backward_co(self):
        cx = self._ctx
        D, Hn, R = self.num_dims, self.num_hidden[-1], self.num_hidden_rnn[-1]
        B, T = cx["B"], cx["T"]
        N, dev = B * T, cx["out"].device
        g = self.store.gviews
        self.store.grad.zero_()
        sv, ss = cx.get("sv"), cx.get("ss")           # sigmoid(z(v)), sigmoid(z(v_s)): left by the forward's free-energy passes
        if sv is None:
            raise RuntimeError("build(..., mode='train') must run before train()")
        # dF/dbh = -sigmoid(z), dF/dbv = -v, dF/dW = -v^T sigmoid(z); cost = F(v) - F(v_s), v_s constant (rbm.py:229).  One pass writes the
        # Dense-output-shaped gradient block and the two scaled hidden blocks of d cost / d W = v_s^T (w ss) - v^T (w sv)
        d_out = torch.empty((N, self.ldo), device=dev)
        pos = torch.empty((N, Hn), device=dev); neg = torch.empty((N, Hn), device=dev)
        dyn = self.store.ls_dyn if self.dtype == torch.float16 else None          # the dynamic multiplier of the f16 loss scale (ParamStore.ls_dyn)
        if cx["n_valid"] is None:                    # ragged_on_device: the scale is a device scalar (folded into the row weights); _unscale gets its inverse
            lsd = cx.get("ls_dev")
            if lsd is not None and dyn is not None:
                lsd = lsd * dyn[0:1]
            ops.rbm_cd_rows(cx["tgt"], cx["v_s"], sv, ss, cx["rw"] if lsd is None else cx["rw"] * lsd, self.grad_scale, d_out, pos, neg)
            ls = 1.0 if lsd is None else 1.0 / lsd
        else:
            ls = self._stack.loss_scale(cx["n_valid"])
            if dyn is not None:
                ops.rbm_cd_rows(cx["tgt"], cx["v_s"], sv, ss, cx["rw"] * dyn[0:1], self.grad_scale * ls, d_out, pos, neg)
                ls = dyn[1:2] * (1.0 / ls)
            else:
                ops.rbm_cd_rows(cx["tgt"], cx["v_s"], sv, ss, cx["rw"], self.grad_scale * ls, d_out, pos, neg)
        # d cost / d W = v_s^T pos + v^T neg, [D, N] . [N, Hn] with K = N rows.  16-bit modes: the operands in the compute type (v, v_s are 0 / 1:
        # exact; pos / neg are loss-scaled products of a weight and a sigmoid) on the LDS-DMA GEMM -- as f32 products on v_mfma_f32_32x32x2_f32
        # (1/16 of the 16-bit rate) the two GEMMs were 0.53 ms per track of the 3.6 ms jamming step (round 4 profile); fp32 mode keeps f32.
        h16 = self._stack.h16
        Np = ops.round_up(N, 64 if h16 else 4)
        def tr(xm, rows):                                # (the zero fill is for the padding columns only: four fills per track and step at C3 otherwise)
            o = (torch.zeros if Np != N else torch.empty)((rows, Np), device=dev, dtype=self.dtype if h16 else torch.float32)
            return ops.transpose(xm, o)
        # two output tiles and K = N rows -- without split-K two workgroups walk the whole batch (6.7 ms of a 19.6 ms step at N = 32 768);
        # slices of >= 256 rows, up to one workgroup per CU
        sk = int(max(1, min(256 // (-(-D // 128) * -(-Hn // 128)), Np // 256)))
        gW = g[f"{self._rbm.prefix}/W"]
        ops.gemm_tn(tr(cx["v_s"], D), tr(pos, Hn), gW, accumulate=True, split_k=sk)
        ops.gemm_tn(tr(cx["tgt"], D), tr(neg, Hn), gW, accumulate=True, split_k=sk)
        if self.bias_mode != "conditional":
            ops.bias_grad(d_out[:, :Hn], g[f"{self._rbm.prefix}/bh"].view(-1), accumulate=True)
            ops.bias_grad(d_out[:, Hn:Hn + D], g[f"{self._rbm.prefix}/bv"].view(-1), accumulate=True)
            self._dx = None
            self._unscale(ls)
            return                                   # as written: no gradient reaches the LSTM / Wuh / Wuv (R3)
        Np8 = ops.round_up(N, 64)
        zalloc = torch.zeros if Np8 != N else torch.empty
        yT = cx["lstm"][-1].get("yT") if cx["lstm"] else None      # emitted by the persistent / row-parallel recurrences
        if yT is None:
            yT = zalloc((R, Np8), device=dev, dtype=self.dtype)
            ops.transpose(cx["y"].view(N, R), yT)
        doT = zalloc((self.n_out, Np8), device=dev, dtype=self.dtype)
        if self.dtype == torch.float32:
            if self.internal_bias:
                ops.bias_grad(d_out[:, :Hn], g[f"{self._rbm.prefix}/bh"].view(-1), accumulate=True)
                ops.bias_grad(d_out[:, Hn:Hn + D], g[f"{self._rbm.prefix}/bv"].view(-1), accumulate=True)
            ops.transpose(d_out[:, :self.n_out], doT)
            do_c = d_out
        else:
            # one pass over d_out: 16-bit copy (the input-gradient operand), 16-bit transpose (the Wuh / Wuv gradient operand) and the column
            # sums, which ARE the gradients of rbm.bh | rbm.bv (internal_bias: rnn_rbm.py:240-259 adds them to the Dense outputs) -- instead of
            # two bias_grad passes, a transpose and a convert2d
            do_c = torch.empty((N, self.ldo), device=dev, dtype=self.dtype)
            colsum = torch.zeros(self.n_out, device=dev)
            ops.grad_rows_fanout(d_out, self.n_out, do_c, doT, colsum)
            if self.internal_bias:
                gh, gv = g[f"{self._rbm.prefix}/bh"].view(-1), g[f"{self._rbm.prefix}/bv"].view(-1)
                ops.axpby(1.0, colsum[:Hn], 1.0, gh, gh)
                ops.axpby(1.0, colsum[Hn:Hn + D], 1.0, gv, gv)
        # Wuh [R,Hn] and Wuv [R,D] are separate variables: one accumulating product per block of the transposed gradient
        ops.gemm_tn(yT, doT[:Hn], g["Wuh"], accumulate=True, split_k=LstmStack._split_k(R, Hn, Np8))
        ops.gemm_tn(yT, doT[Hn:Hn + D], g["Wuv"], accumulate=True, split_k=LstmStack._split_k(R, D, Np8))
        dy = torch.empty((N, R), device=dev)
        ops.gemm_tn(do_c, self._wu_p, dy)
        self._dx = yield from self._stack.backward_co(dy.view(T, B, R), cx["lstm"], cx["kp"], self.seed, self.row0, need_dx=self.need_dx,
                                                      step_dev=self.store.step_dev)
        self._unscale(ls)

    def zero_state(self, batch_size):
        self._materialize(self._num_inputs)
        dev = self.store.theta.device
        return RnnEstimatorStateTuple(torch.zeros((batch_size, self.num_hidden[-1]), device=dev),
                                      torch.zeros((batch_size, self.num_dims), device=dev), self._get_rnn_zero_state(batch_size))

    def _get_state(self, inputs, lengths=None, initial_state=None, last_outputs=False):
        """rnn_rbm.py:184-238 (dynamic_rnn; identical to the decode loop on valid rows)."""
        self._materialize(inputs.shape[-1])
        self._ensure_packed()
        if inputs.dim() == 2:
            inputs = inputs[:, None, :]
        B, T, _ = inputs.shape
        x_tm = self._to_time_major_inputs(inputs)
        st0 = [(c, h) for c, h in initial_state.rnn_state] if initial_state is not None else None
        y, _, final = self._stack.forward(x_tm, self._rnn.effective_keep_prob(), self.seed, self.row0, save=False, state0=st0)
        out = self._biases(y[-1].contiguous()) if last_outputs else self._biases(y.view(T * B, -1))[flat_index(lengths, B, T, inputs.device)]
        Hn, D = self.num_hidden[-1], self.num_dims
        return RnnEstimatorStateTuple(out[:, :Hn], out[:, Hn:Hn + D], tuple((c.clone(), h.clone()) for c, h in final))

    def _det_state(self, h, rnn_state):
        """rnn_rbm.py:240-259 in the deterministic arithmetic: bh_t = rbm.bh + h . Wuh, bv_t = rbm.bv + h . Wuv (two jobs, one launch)."""
        Hn, D = self.num_hidden[-1], self.num_dims
        out = torch.empty((h.shape[0], self.ldo), device=h.device)
        ib = self.internal_bias
        ops.dense_det([dict(x=h, W=self.store["Wuh"], bias=self._rbm.bh.view(-1) if ib else None, out=out[:, :Hn]),
                       dict(x=h, W=self.store["Wuv"], bias=self._rbm.bv.view(-1) if ib else None, out=out[:, Hn:Hn + D])])
        return RnnEstimatorStateTuple(out[:, :Hn], out[:, Hn:Hn + D], tuple(rnn_state))

    def single_step(self, inputs, initial_state, x2=None):
        """rnn_rbm.py:261-281."""
        if self.det_sampling:
            return self._det_single_step(inputs, initial_state, x2)
        if x2 is not None:
            inputs = torch.cat([inputs.float(), x2], 1)
        x = self._step_input(inputs.shape[0], inputs.device)
        ops.convert2d(inputs.contiguous(), x[:, :inputs.shape[1]])
        h, new = self._stack.single_step(x, [(c, hh) for c, hh in initial_state.rnn_state])
        out = self._biases(h.contiguous())
        Hn, D = self.num_hidden[-1], self.num_dims
        return RnnEstimatorStateTuple(out[:, :Hn], out[:, Hn:Hn + D], tuple(new))

    def sample_single(self, inputs, state):
        """rnn_rbm.py:283-297 with k = rbm.k (R1): returns (sample u8, cond_prob)."""
        p_v, v = self._rbm.sample(inputs[:, :self.num_dims], state.b_enc, state.b_dec, self._k, self.seed, self.row0, None,
                                  getattr(self, "_gen_step", 0) * max(self._k, 1))
        return v, p_v

    def pretrain(self, optimizer, lr, run_optimizer=True):
        """rnn_rbm.py:299-322: one CD-k update of the RBM module on the flattened inputs (`rbm.train`, no optimiser, no clipping).
        Returns the documented 5-tuple (the reference returns `rbm.train`'s 3-tuple although its docstring and every caller --
        multinn_jamming.py:219-221 -- expect five values)."""
        x = self._inputs
        B, T, _ = x.shape
        flat = x.to(torch.uint8)[:, :, :self.num_dims].reshape(B * T, self.num_dims)
        if self._lengths is not None:
            m = torch.arange(T, device=x.device)[None, :] < self._lengths.to(x.device)[:, None]
            flat = flat[m.reshape(-1)]
        self._materialize(x.shape[-1])
        if not run_optimizer:
            # the only caller that passes False (multinn_jamming.py:219-241 without separate losses) discards the CD update ops in favour of
            # its joint gradient step and keeps the init ops: nothing is applied here
            return self._rbm.visible_bias_init_ops(flat.contiguous()), [], self.metrics, self.metrics_upd, self.summaries
        # eager semantics: the CD-k update is APPLIED by this call and update_ops comes back empty, while init_ops are
        # returned unexecuted (the reference runs them once, before the first update: train_encoders.py:150-153) -- run them BEFORE the first
        # pretrain() call, not after it, or the first update's visible-bias delta is overwritten
        init_ops, update_ops, self._cd_gradients = self._rbm.train(flat.contiguous(), lr, seed=self.seed + self.store.step, row0=self.row0 * T)
        self._packed_step = -1                          # rbm.bh / rbm.bv feed the packed bias row
        return init_ops, update_ops, self.metrics, self.metrics_upd, self.summaries
