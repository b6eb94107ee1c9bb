"""The tensor plumbing of the five MultINN operation modes around the Encoder / Generator plugin classes
(/root/reference/multinn/models/multinn): which slice of the `[batch, time, pitch, track]` piano-roll each encoder
sees, how the encodings are stacked for the generators (inputs = enc[:, :-1], targets = enc[:, 1:]), how generator
outputs are decoded back, how the per-track losses are combined and optimised, and how samples are put back together.

    MultINNJoint        multinn_joint.py:41-215       one encoder over P*M features, one generator            (C1, C2, TGT)
    MultINNJamming      multinn_jamming.py:31-250     M per-track encoders, M independent generators           (C3)
    MultINNComposer     multinn_composer.py:33-204    M per-track encoders, ONE RnnMultiNADE over the stack    (C4)
    MultINNFeedback     multinn_feedback.py:46-218    jamming + Dense feedback module
    MultINNFeedbackRnn  multinn_feedback_rnn.py:30-79 jamming + recurrent feedback module                      (C5)
    MultINN             multinn.py:24-53              facade choosing one of them by `params['mode']`

Same constructor arguments (`config`, `params` dicts with the YAML keys of configs/*.yaml), method names and return
arity as the reference.  The reference builds a TF1 graph once and feeds placeholders at every `sess.run`; here
`build(x, lengths=..., is_train=..., mode=...)` takes the fed values and RUNS the forward pass on the device,
`train_generators(optimizer, lr)` runs backward + the clipped optimiser step, `generate(num_steps)` runs the sampling
scan.  The encoder-level ("global") metrics the reference adds to the graph but never evaluates in a training
`sess.run` are computed on first access of `.metrics`.  Everything numeric goes through the Encoder / Generator classes
(HIP kernels behind the C ABI); this file only slices, stacks and reshapes device tensors.
"""
import abc
import os

import torch

from .common import Model, graph_capture
from .encoders import PassEncoder, DBNEncoder
from .generators import RnnNade, RnnRBM, RnnMultiNADE
from . import ops
from .training import compute_gradients_multi, world


def flatten_maybe_padded_sequences(x, lengths=None):
    """utils/sequences.py:6-37: `[B,T,...]` -> `[N,...]`, rows (b,t) with t < lengths[b], b-major then t."""
    B, T = x.shape[0], x.shape[1]
    flat = x.reshape(B * T, *x.shape[2:])
    if lengths is None:
        return flat
    m = torch.arange(T, device=x.device)[None, :] < lengths.to(x.device)[:, None]
    return flat[m.reshape(-1)]


class _LazyMetrics(dict):
    """`metrics['global']` of train_generators (multinn_joint.py:286): the encoder-level metrics, evaluated when first read."""

    def __init__(self, fn):
        super().__init__()
        self._fn, self._done = fn, False

    def _fill(self):
        if not self._done:
            self._done = True
            super().update(self._fn())

    def __getitem__(self, k):
        self._fill()
        return super().__getitem__(k)

    def __contains__(self, k):
        self._fill()
        return super().__contains__(k)

    def keys(self):
        self._fill()
        return super().keys()

    def items(self):
        self._fill()
        return super().items()

    def __iter__(self):
        self._fill()
        return super().__iter__()

    def __len__(self):
        self._fill()
        return super().__len__()


class MultINNCore(Model):
    """core/multinn_core.py:17-448 + core/multinn_interface.py."""

    def __init__(self, config, params, name="MultINN", precision="bf16", seed=None, device=None):
        super().__init__(name=name)
        self._mode = "core"
        self._config, self._params = config, params
        self._encoder_type = params["encoder"]["type"]
        self._generator_type = params["generator"]["type"]
        if self._encoder_type == "Pass":
            encoder_class = PassEncoder
        elif self._encoder_type in ("RBM", "DBN"):
            encoder_class = DBNEncoder
        else:
            raise ValueError("Incorrect encoder type, supported types are `Pass`, `RBM`, and `DBN`")
        if self._generator_type == "RBM":
            generator_class = RnnRBM
        elif self._generator_type == "NADE":
            generator_class = RnnNade
        else:
            raise ValueError("Incorrect generator type, supported types are `RBM`, and `NADE`")
        num_dims = config["data"]["pitch_range"]["highest"] - config["data"]["pitch_range"]["lowest"]
        self._num_dims = num_dims * config["training"]["num_pixels"]
        self._tracks = list(config["data"]["instruments"])
        self._feedback_module = False
        self._keep_prob = params["keep_prob"]
        # tune_encoder only removes a tf.stop_gradient on the encoded inputs (multinn_joint.py:117-122, multi_encoder_nn.py:110-113).  No
        # optimiser in the reference ever applies that gradient: Generator.train differentiates wrt the GENERATOR's trainable_variables
        # only (generator.py:201), the encoders train by contrastive divergence (dbn_encoder.py:192-240), PassEncoder has no variables and
        # the DBN codes are floor(p + u) samples (zero gradient).  Both settings therefore give the same updates; the flag is kept and reported.
        self._tune_encoder = bool(params["tune_encoder"])
        self.precision, self.device = precision, device
        self.seed = config["training"].get("random_seed", 23) if seed is None else seed
        self.clip_norm = 5.0                               # utils/training.py:166 (hard-coded in the reference, R9)
        self.separate_losses = False                       # jamming / composer default (multinn_jamming.py:156)
        self._row0 = 0
        self._x = self._lengths = self._is_train = None
        self._placeholders = {"x": None, "lengths": None, "is_train": None}
        self._encoders = self._init_encoders(encoder_class)
        self._generators = self._init_generators(generator_class)
        self._x_encoded = self._x_hidden = self._outputs_probs = self._outputs = self._inputs = None

    # -- construction ---------------------------------------------------------------------------
    def _encoder_kwargs(self, i=0):
        """Build-only arguments next to the reference's (`num_dims`, `num_hidden`, `track_name`)."""
        return {} if self._encoder_type == "Pass" else dict(seed=self.seed + 1000 + i, device=self.device)

    def _generator_kwargs(self, i=0):
        return dict(precision=self.precision, seed=self.seed + i, device=self.device, clip_norm=self.clip_norm)

    @abc.abstractmethod
    def _init_encoders(self, encoder_class):
        ...

    @abc.abstractmethod
    def _init_generators(self, generator_class):
        ...

    mode = property(lambda self: self._mode)
    num_dims = property(lambda self: self._num_dims)
    tracks = property(lambda self: self._tracks)
    num_tracks = property(lambda self: len(self._tracks))
    encoder_type = property(lambda self: self._encoder_type)
    generator_type = property(lambda self: self._generator_type)
    encoders = property(lambda self: self._encoders)
    generators = property(lambda self: self._generators)
    feedback_module = property(lambda self: self._feedback_module)
    keep_prob = property(lambda self: self._keep_prob)
    tune_encoder = property(lambda self: self._tune_encoder)
    trainable_feedback_variables = property(lambda self: [])

    @property
    def loss(self):
        return self.metrics["batch/loss"]

    @property
    def row0(self):
        return self._row0

    @row0.setter
    def row0(self, v):
        """Global index of this rank's first sequence: every RNG counter is keyed by GLOBAL rows (data parallel)."""
        self._row0 = int(v)
        for g in self._generators:
            g.row0 = self._row0

    @property
    def trainable_encoder_variables(self):
        return [v for e in self._encoders for v in e.trainable_variables]

    @property
    def trainable_generator_variables(self):
        return [v for g in self._generators for v in g.trainable_variables]

    # -- build ----------------------------------------------------------------------------------
    def build(self, x=None, y=None, lengths=None, is_train=None, mode="eval"):
        """multinn_core.py:178-244.  x: u8 (or float 0/1) `[B,T,P,M]` piano-roll batch -- the value fed to the reference's `x`
        placeholder; lengths int32 `[B]` or None; is_train bool."""
        Model.build(self, mode=mode)
        if x is None:
            raise ValueError("build() needs the piano-roll batch x [B,T,P,M] (the reference feeds it through a placeholder)")
        if x.dim() != 4 or x.shape[2] != self._num_dims or x.shape[3] != self.num_tracks:
            raise ValueError(f"x must be [batch, time, {self._num_dims}, {self.num_tracks}], got {tuple(x.shape)}")
        x = x if x.dtype == torch.uint8 else (x != 0).to(torch.uint8)
        self._x, self._lengths, self._is_train, self._build_mode = x.contiguous(), lengths, bool(is_train), mode
        self._placeholders = {"x": self._x, "lengths": lengths, "is_train": self._is_train}
        T1 = x.shape[1] + 1
        for e in self._encoders:                       # encoder draws: flat (b, t) rows of the padded batch, global ids
            if hasattr(e, "row0"):
                e.row0 = self._row0 * T1
        self._inputs = self._x_encoded = None
        self._metrics = self._metrics_upd = None
        self._x_hidden = self._outputs_probs = self._outputs = None
        self._build_all(mode)
        self._variables = {"encoders": [e.variables for e in self._encoders], "generators": [g.variables for g in self._generators],
                           "feedback": self.trainable_feedback_variables}
        self._trainable_variables = self.trainable_encoder_variables + self.trainable_generator_variables + self.trainable_feedback_variables
        self._is_built = True

    def _build_all(self, mode):
        self._inputs = self._build_inputs()
        self._build_encoders("eval")
        self._x_encoded = self._encode_inputs()
        self._build_generators(mode)

    def _ensure_global_metrics(self):
        """multinn_core.py:226-241 (outputs, decodings, targets, encoder-level metrics), evaluated on demand."""
        if self._metrics is None and self._is_built and getattr(self, "_build_mode", None) in ("train", "eval"):
            self._x_hidden = self._build_generator_outputs()
            self._outputs_probs, self._outputs = self._decode_generator_outputs()
            targets = self._build_targets()
            self._metrics, self._metrics_upd, self._summaries["metrics"] = self.build_metrics(
                targets=targets, predictions=self._outputs, cond_probs=self._outputs_probs)
        return self._metrics

    metrics = property(lambda self: self._ensure_global_metrics())
    metrics_upd = property(lambda self: (self._ensure_global_metrics(), self._metrics_upd)[1])

    # -- sampling -------------------------------------------------------------------------------
    def sampler(self, num_beats):
        """multinn_core.py:324-341: number of model time steps in `num_beats` beats, then generate()."""
        d = self._config["data"]
        pitch_span = d["pitch_range"]["highest"] - d["pitch_range"]["lowest"]
        return self.generate(num_beats * d["beat_resolution"] * pitch_span // self._num_dims)

    def evaluator(self):
        """multinn_core.py:343-362: musical metrics of the fed batch reshaped into bars `[B, bars, 4*beat_resolution, pitch_span, M]`."""
        from . import metrics as MM
        d = self._config["data"]
        pitch_span = d["pitch_range"]["highest"] - d["pitch_range"]["lowest"]
        bars = self._x.reshape(self._x.shape[0], -1, 4 * d["beat_resolution"], pitch_span, self.num_tracks)
        return MM.compute_sample_metrics(bars)

    def _combine_track_metrics(self, track_metrics, track_metrics_upd, track_summaries, global_scope=None):
        """multinn_core.py:364-413: per-key lists over the tracks, averaged when a global scope is given."""
        assert len(track_metrics) == self.num_tracks
        metrics, metrics_upd = {}, []
        for i in range(self.num_tracks):
            for k, m in track_metrics[i].items():
                metrics.setdefault(k, []).append(m)
            metrics_upd += list(track_metrics_upd[i] or [])
        if global_scope is not None:
            for k, v in metrics.items():
                if all(torch.is_tensor(t) for t in v):
                    metrics[k] = torch.stack([t.reshape(()) for t in v]).mean()
                else:
                    metrics[k] = sum(float(t) for t in v) / len(v)
        return metrics, metrics_upd, {"metrics": None, "weights": None, "gradients": None}

    def load_encoders(self, sess=None, ckpt_dir=None):
        """multinn_core.py:425-448: restore the ENCODER variables only, from the checkpoint `save()` wrote into ckpt_dir (the reference flow:
        train_encoders.py `model.save(encoders_dir)`, then train.py:126 `model.load_encoders(encoders_dir)`).  The encoder stores are read
        from the mode's one checkpoint file; a directory that only holds the older per-encoder `{encoder.name}.pt` files is still read (with a
        warning), and a directory with neither returns False."""
        if self._encoder_type == "Pass":
            return True
        path = self._ckpt_path(ckpt_dir)
        if not os.path.exists(path):
            # the reference restores whatever checkpoint sits in the directory, by variable name: encoders pre-trained through ANOTHER mode
            # class (other default name, e.g. MultINN-jamming) are found by content -- any mode checkpoint with these tracks and encoders
            import glob
            import warnings
            # newest first (the reference restores the MOST RECENT checkpoint: tf.train.get_checkpoint_state, multinn_core.py:440-444); a
            # candidate must carry these tracks, this many encoders AND the same variable names per encoder -- a file that does not is skipped,
            # not an error; map_location='cpu': a directory of large generator checkpoints is scanned without touching device memory
            hits = []
            cands = sorted(glob.glob(os.path.join(ckpt_dir, "*.pt")), key=lambda f: os.path.getmtime(f), reverse=True)
            for cand in cands:
                try:
                    b = torch.load(cand, map_location="cpu")
                except Exception:
                    continue
                if not (isinstance(b, dict) and list(b.get("tracks", [])) == list(self._tracks) and len(b.get("encoders", [])) == len(self._encoders)
                        and all(sd is not None for sd in b["encoders"])):
                    continue
                names_ok = True
                for e, sd in zip(self._encoders, b["encoders"]):
                    if e.store.theta is None:
                        e.store.materialize()
                    names_ok = names_ok and isinstance(sd, dict) and list(sd.get("names", [])) == e.store.names()
                if names_ok:
                    hits.append(cand)
            if hits:
                if len(hits) > 1:
                    warnings.warn(f"{ckpt_dir}: several mode checkpoints hold encoders for these tracks ({[os.path.basename(h) for h in hits]}); "
                                  f"loading the most recent, {os.path.basename(hits[0])}")
                path = hits[0]
        if os.path.exists(path):
            blob = torch.load(path, map_location="cpu")
            if list(blob.get("tracks", [])) != list(self._tracks) or len(blob.get("encoders", [])) != len(self._encoders):
                raise ValueError(f"checkpoint {path} does not match this model's tracks / encoders")
            if any(sd is None for sd in blob["encoders"]):
                return False                            # written before the encoders had variables: nothing to restore
            for e, sd in zip(self._encoders, blob["encoders"]):
                if e.store.theta is None:
                    e.store.materialize()               # load-before-first-build, as load() does
                # the reference's Saver(trainable_encoder_variables) restores the WEIGHTS only (multinn_core.py:431-436): the Adam slots and
                # the step count of the pre-training run are not carried into the generator training
                if list(sd["names"]) != e.store.names():
                    raise ValueError(f"checkpoint {path}: encoder variables {sd['names']} do not match {e.store.names()}")
                e.store.theta.copy_(sd["theta"])
            return True
        old = [os.path.join(ckpt_dir, f"{e.name}.pt") for e in self._encoders]
        if all(os.path.exists(p) for p in old):
            import warnings
            warnings.warn(f"{ckpt_dir} holds per-encoder checkpoint files of the older format and no {os.path.basename(path)}: loading those")
            return all(e.load(None, ckpt_dir) for e in self._encoders)
        return False

    # -- checkpoints (model.py:180-234: ONE tf Saver over uniquely scoped variables) ------------------------------------------------
    # One file per mode, `{name}.pt`: every generator / encoder / feedback store under its own index (the M per-track generators of the
    # jamming and feedback modes all carry the default name 'rnn-nade' / 'rnn-rbm': per-model files keyed by name overwrote one another),
    # parameters AND Adam slots and step counts (R12).  Stores that are only created at the first build (a generator's variables need its
    # input width, the feedback module is built from the stacked codes) are materialised from the checkpoint's own shapes before loading.
    def _ckpt_path(self, ckpt_dir):
        return os.path.join(ckpt_dir, f"{self.name}.pt")

    def _feedback_state(self):
        fl = getattr(self, "_feedback_layer", None)
        return None if fl is None or fl.store.theta is None else fl.store.state_dict()

    def save(self, sess=None, ckpt_dir=None, global_step=None, write_meta_graph=False):
        os.makedirs(ckpt_dir, exist_ok=True)
        sd = lambda m: None if m.store is None or m.store.theta is None else m.store.state_dict()
        blob = dict(mode=self._mode, tracks=list(self._tracks), generators=[sd(g) for g in self._generators],
                    encoders=[sd(e) for e in self._encoders], feedback=self._feedback_state(), global_step=global_step)
        path = self._ckpt_path(ckpt_dir)
        torch.save(blob, path)
        return path

    @staticmethod
    def _first_kernel_rows(sd, suffix="cell_0/kernel"):
        for n, shp in zip(sd["names"], sd["shapes"]):
            if n.endswith(suffix):
                return int(shp[0]), int(shp[1])
        raise ValueError("checkpoint holds no variable ending in " + suffix)

    def load(self, sess=None, ckpt_dir=None):
        path = self._ckpt_path(ckpt_dir)
        if not os.path.exists(path):
            if os.path.isdir(ckpt_dir) and any(f.endswith(".pt") for f in os.listdir(ckpt_dir)):
                import warnings
                warnings.warn(f"{ckpt_dir} holds .pt files but no {os.path.basename(path)} (per-model files of an older format are not "
                              "read by load()): starting from the initial weights")
            return False
        blob = torch.load(path)
        if blob.get("mode") != self._mode or list(blob.get("tracks", [])) != list(self._tracks):
            raise ValueError(f"checkpoint {path} was written by mode {blob.get('mode')!r} / tracks {blob.get('tracks')}, "
                             f"this model is {self._mode!r} / {self._tracks}")
        if len(blob["generators"]) != len(self._generators) or len(blob["encoders"]) != len(self._encoders):
            raise ValueError("checkpoint does not match the model's generators / encoders")
        for g, sd in zip(self._generators, blob["generators"]):
            if sd is None:
                continue
            if g.store.theta is None:                   # variables are declared at the first build: take the input width from the checkpoint
                rows, cols = self._first_kernel_rows(sd)
                g._materialize(rows - cols // 4)
            g.store.load_state_dict(sd)
            g._packed_step = -1
        for e, sd in zip(self._encoders, blob["encoders"]):
            if sd is not None:
                e.store.load_state_dict(sd)
        fb = blob.get("feedback")
        if fb is not None:
            if not self._feedback_module:
                raise ValueError("checkpoint holds a feedback module, this mode has none")
            if self._feedback_layer is None:
                rows, cols = self._first_kernel_rows(fb) if any(n.endswith("cell_0/kernel") for n in fb["names"]) \
                    else self._first_kernel_rows(fb, "dense_0/kernel")
                n_in = rows - cols // 4 if any(n.endswith("cell_0/kernel") for n in fb["names"]) else rows
                self._feedback_layer = self._init_feedback(n_in)
            self._feedback_layer.store.load_state_dict(fb)
            if hasattr(self._feedback_layer, "_packed_step"):
                self._feedback_layer._packed_step = -1
        return True

    def check(self, tolerate_overflow=False):
        """Raise if a persistent recurrence launch of any generator (or of the feedback module) ever gave up on a bounded spin: its outputs
        were garbage (LstmStack.check; synchronises the device).  The driver calls it before every validation pass and checkpoint.
        tolerate_overflow (the training loop): optimiser steps skipped in precision "fp16" are the dynamic loss scale at work -- a warning."""
        for g in self._generators:
            if getattr(g, "_stack", None) is not None:
                g._stack.check()
            if getattr(g, "store", None) is not None:
                g.store.check(tolerate_overflow and getattr(g, "dtype", None) == torch.float16)      # optimiser steps skipped on the device (non-finite gradient norm)
        fl = getattr(self, "_feedback_layer", None)
        if fl is not None and getattr(fl, "_stack", None) is not None:
            fl._stack.check()
        if fl is not None and getattr(fl, "store", None) is not None:
            fl.store.check(tolerate_overflow and getattr(fl, "dtype", None) == torch.float16)
        for e in getattr(self, "_encoders", None) or []:      # the encoders' stores take optimiser steps too (pretrain_encoders, compute_gradients)
            if getattr(e, "store", None) is not None:
                e.store.check()

    # -- train.py:178-189: one `sess.run([update_ops, loss], feed_dict)` -----------------------------
    def generator_loss(self):
        """Mean over the generators of their `batch/loss` (the `loss` / `loglik` train.py:74-75 logs and validates on)."""
        ls = [g.metrics["batch/loss"].reshape(()) for g in self._generators]
        return ls[0] if len(ls) == 1 else torch.stack(ls).mean()

    def train_step(self, x, lengths, optimizer, lr=None):
        """build(x, lengths, is_train=True, mode='train') + train_generators(optimizer, lr): one optimiser step from a zero RNN state."""
        self.build(x, lengths=lengths, is_train=True, mode="train")
        self.train_generators(optimizer, lr)
        return self.generator_loss()

    def _all_stores(self):
        out = [g.store for g in self._generators]
        fl = getattr(self, "_feedback_layer", None)
        return out + ([fl.store] if fl is not None else [])

    def graphed_train_step(self, x, optimizer, lr=None, warmup=2, lengths=None):
        """`train_step` as ONE hipGraph replay (single rank): encoders, every generator's build and backward, the feedback module and the joint
        clipped step -- eagerly a mode's step is host-bound (five generators: ~200 launches and the Python between them).  Step-dependent
        values (dropout / Gibbs seeds, Adam's step) are read from each store's device-side counter, so every replay is the next step.
        lengths (int32 [B], optional): capture the RAGGED step; the graph holds a static copy of the lengths and everything derived from them
        (row weights, valid-row counts, f16 loss scales, the compaction of the NADE generators' rows) is computed on the device inside the graph
        (`RnnEstimator.ragged_on_device`), so one capture serves every later `run(x, lengths)`.  Returns run(x=None, lengths=None) -> loss."""
        from .training import dp_active
        if dp_active():
            raise NotImplementedError("graphed_train_step of a mode is single-rank (the generators' own captured steps handle data parallelism)")
        sx = x.clone()
        ragged = lengths is not None
        sl = lengths.to(device=x.device, dtype=torch.int32).clone() if ragged else None
        if ragged:
            for g in self._generators:
                g.ragged_on_device = True
        cur, side = torch.cuda.current_stream(), torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            # warm-up steps are REAL optimiser steps: variables, persistent-kernel attributes and workspaces must exist before the capture.
            # warmup = 0 is for a caller that has just run this very step eagerly (driver._captured_step: the capture then executes nothing)
            for _ in range(int(warmup)):
                self.train_step(sx, sl, optimizer, lr)
        cur.wait_stream(side)
        for g in self._generators:
            g._packed_step = -1
        graph = torch.cuda.CUDAGraph()
        with graph_capture(graph):
            loss = self.train_step(sx, sl, optimizer, lr)
        stores = self._all_stores()
        for st in stores:
            st.step -= 1                                # the captured step has not executed (host mirror of step_dev)

        def run(x=None, lengths=None):
            if x is not None:
                sx.copy_(x)
            if lengths is not None:
                if not ragged:
                    raise ValueError("this step was captured for full-length windows: capture it with lengths= to feed ragged ones")
                sl.copy_(lengths.to(device=sx.device, dtype=torch.int32))
            graph.replay()
            for st in stores:
                st.step += 1
            return loss
        run.graph = graph
        run.ragged = ragged
        return run

    def build_pianoroll(self, x, lengths=None, is_train=False, mode="eval"):
        """The driver's evaluation entry (same name as RnnNade.build_pianoroll)."""
        self.build(x, lengths=lengths, is_train=is_train, mode=mode)

    # -- shared pieces of the train step ----------------------------------------------------------
    def _with_global(self, out):
        init_ops, update_ops, metrics, metrics_upd, summaries = out
        metrics = dict(metrics)
        metrics["global"] = _LazyMetrics(lambda: self.metrics)          # multinn_joint.py:284-286
        return init_ops, update_ops, metrics, metrics_upd, summaries


# ================================================================================================
class MultINNJoint(MultINNCore):
    """multinn_joint.py: one Encoder + one Generator over the stacked tracks (feature index p*M+m)."""

    def __init__(self, config, params, name="MultINN-joint", **kw):
        super().__init__(config, params, name=name, **kw)
        self._mode = "joint"

    def _init_encoders(self, encoder_class):
        num_dims = self.num_dims * self.num_tracks                       # multinn_joint.py:41-52
        encoders = [encoder_class(num_dims=num_dims, num_hidden=self._params["encoder"]["num_hidden"], track_name="all",
                                  **self._encoder_kwargs())]
        self._encoder = encoders[0]
        self._num_dims_generator = self._encoder.num_hidden[-1]
        return encoders

    def _init_generators(self, generator_class):
        generators = [generator_class(num_dims=self._num_dims_generator, num_hidden=self._params["generator"]["num_hidden"],
                                      num_hidden_rnn=self._params["generator"]["num_hidden_rnn"], keep_prob=self.keep_prob,
                                      **self._generator_kwargs())]          # multinn_joint.py:65-74
        self._generator = generators[0]
        return generators

    def _fused(self):
        """PassEncoder + RnnNade: zero-pad, shift, inputs / targets slicing fused into the generator's one plumbing kernel
        (RnnNade.build_pianoroll: multinn_joint.py:83-89,132-139 in one pass over the uint8 batch)."""
        return self._encoder_type == "Pass" and isinstance(self._generator, RnnNade)

    def _build_all(self, mode):
        if self._fused() and mode in ("train", "eval"):
            self._generator.build_pianoroll(self._x, self._lengths, is_train=self._is_train, mode=mode)
            return
        super()._build_all(mode)

    def _build_inputs(self):
        B, T, P, M = self._x.shape
        inputs = self._x.reshape(B, T, P * M)                             # multinn_joint.py:83-89
        return torch.cat([torch.zeros((B, 1, P * M), device=inputs.device, dtype=inputs.dtype), inputs], 1)

    def _build_targets(self):
        return flatten_maybe_padded_sequences(self._x, self._lengths).reshape(-1, self.num_dims * self.num_tracks)

    def _build_encoders(self, mode="eval"):
        self._encoder.build(self._inputs, lengths=self._lengths, mode=mode)

    def _encode_inputs(self):
        _, x_encoded = self._encoder.encode()
        return x_encoded                                                  # stop_gradient: nothing differentiates through it

    def _build_generators(self, mode="eval"):
        self._generator.build(x=self._x_encoded[:, :-1], y=self._x_encoded[:, 1:], lengths=self._lengths, is_train=self._is_train,
                              mode=mode)                                  # multinn_joint.py:132-139

    def _build_generator_outputs(self):
        return self._generator.forward()

    def _decode_generator_outputs(self):
        return self._encoder.decode(self._x_hidden)

    def build_metrics(self, targets, predictions, cond_probs=None, log_probs=None):
        """multinn_joint.py:159-186: encoder-level metrics, loss / NLL / perplexity divided by the number of tracks."""
        metrics, metrics_upd, summaries = self._encoder.build_metrics(targets=targets, predictions=predictions, cond_probs=cond_probs)
        for k in ("batch/loss", "log_likelihood", "perplexity"):
            metrics[k] = metrics[k] / self.num_tracks
        return metrics, metrics_upd, summaries

    def generate(self, num_steps):
        """multinn_joint.py:188-215 -> u8 `[B, num_steps, P, M]`."""
        if self._x_encoded is None:
            MultINNCore._build_all(self, "generate")
        samples_h = self._generator.generate(self._x_encoded, num_steps)
        _, samples = self._encoder.decode(samples_h)
        return samples.reshape(-1, num_steps, self.num_dims, self.num_tracks).to(torch.uint8)

    def train_encoders(self, optimizer, lr, layer=0):
        if self._inputs is None:
            self._inputs = self._build_inputs()
            self._build_encoders("eval")
        return self._encoder.train(optimizer, lr, layer=layer)

    def pretrain_generators(self, optimizer, lr, separate_losses=True):
        return self._with_global(self._generator.pretrain(optimizer, lr))

    def train_generators(self, optimizer, lr, separate_losses=True):
        return self._with_global(self._generator.train(optimizer, lr))


# ================================================================================================
class MultIEncoderNN(MultINNCore):
    """core/multi_encoder_nn.py: one Encoder per track."""

    def _init_encoders(self, encoder_class):
        encoders = [encoder_class(num_dims=self.num_dims, num_hidden=self._params["encoder"]["num_hidden"], track_name=self.tracks[i],
                                  **self._encoder_kwargs(i)) for i in range(self.num_tracks)]           # multi_encoder_nn.py:41-50
        self._num_dims_generator = encoders[0].num_hidden[-1]
        return encoders

    def _build_inputs(self):
        """multi_encoder_nn.py:66-76: zero first step, then one `[B,T+1,P]` sequence per track."""
        B, T, P, M = self._x.shape
        padded = torch.cat([torch.zeros((B, 1, P, M), device=self._x.device, dtype=self._x.dtype), self._x], 1)
        return [t.contiguous() for t in padded.unbind(-1)]

    def _build_targets(self):
        return list(flatten_maybe_padded_sequences(self._x, self._lengths).unbind(-1))        # multi_encoder_nn.py:78-87

    def _build_encoders(self, mode="eval"):
        for i in range(self.num_tracks):
            self.encoders[i].build(self._inputs[i], lengths=self._lengths, mode=mode)

    def _encode_inputs(self):
        return [self.encoders[i].encode()[1] for i in range(self.num_tracks)]

    def _stack_encoded(self):
        """multinn_composer.py:73-80 / multinn_feedback.py:85-91: stack on axis 3, reshape to `[B,T+1,E*M]` (feature e*M+m)."""
        st = torch.stack(self._x_encoded, dim=3)
        return st.reshape(st.shape[0], st.shape[1], self._num_dims_generator * self.num_tracks)

    def build_metrics(self, targets, predictions, cond_probs=None, log_probs=None):
        """multi_encoder_nn.py:117-153: per-track encoder metrics, averaged over the tracks."""
        tm, tu, ts = [], [], []
        for i in range(self.num_tracks):
            m, u, s = self.encoders[i].build_metrics(targets=targets[i], predictions=predictions[i], cond_probs=cond_probs[i])
            tm.append(m); tu.append(u); ts.append({"metrics": s})
        return self._combine_track_metrics(tm, tu, ts, global_scope=f"metrics/{self.encoders[0].name}/global/")

    def train_encoders(self, optimizer, lr, layer=0):
        """multi_encoder_nn.py:155-195."""
        if self._inputs is None:
            self._inputs = self._build_inputs()
            self._build_encoders("eval")
        init_ops, update_ops, tm, tu, ts = [], [], [], [], []
        for i in range(self.num_tracks):
            io, uo, m, mu, s = self.encoders[i].train(optimizer, lr, layer=layer)
            init_ops += io; update_ops += uo
            tm.append(m or {}); tu.append(mu); ts.append(s or {})
        metrics, metrics_upd, summaries = self._combine_track_metrics(tm, tu, ts, global_scope="metrics/global/")
        return init_ops, update_ops, metrics, metrics_upd, summaries

    def _decode_tracks(self, hidden):
        probs, outs = [], []
        for i in range(self.num_tracks):
            p, d = self.encoders[i].decode(hidden[i])
            probs.append(p); outs.append(d)
        return probs, outs


class MultINNJamming(MultIEncoderNN):
    """multinn_jamming.py: M independent per-track generators."""

    def __init__(self, config, params, name="MultINN-jamming", **kw):
        super().__init__(config, params, name=name, **kw)
        self._mode = "jamming"

    def _init_generators(self, generator_class):
        return [generator_class(num_dims=self._num_dims_generator, num_hidden=self._params["generator"]["num_hidden"],
                                num_hidden_rnn=self._params["generator"]["num_hidden_rnn"], keep_prob=self.keep_prob,
                                track_name=self.tracks[i], **self._generator_kwargs(i)) for i in range(self.num_tracks)]    # :40-48

    def _generator_io(self, i):
        return self._x_encoded[i][:, :-1], self._x_encoded[i][:, 1:]                            # multinn_jamming.py:61-65

    # The M generators are independent (each has its own LSTM, multinn_jamming.py:40-48) and are trained on one loss (:213-221): their build and
    # backward passes run in LOCKSTEP (generators.drive_group), so that the same layer's recurrence of all tracks is ONE launch of the
    # CU-resident / cluster kernels (five tracks x 256 rows fill the chip; one track alone keeps 64 CUs busy).  MULTINN_JAMMING_GROUP=0: one
    # generator after the other, each on the two-layer persistent form.
    group_generators = os.environ.get("MULTINN_JAMMING_GROUP", "1") != "0"

    def _grouped(self, mode):
        """True when the generators' recurrences can share launches: a train / eval build of >= 2 like generators in a 16-bit mode whose stacks
        (two layers, 512 and 256 units) the cluster / CU-resident kernels cover at this batch size."""
        from . import ops as _ops
        from .generators import RnnEstimator
        gens = self.generators
        if not (self.group_generators and mode in ("train", "eval") and 2 <= len(gens) <= 8 and type(self) is MultINNJamming):
            return False
        if not all(isinstance(g, RnnEstimator) and type(g) is type(gens[0]) and g.dtype in _ops.H16 for g in gens):
            return False
        x0 = self._generator_io(0)[0]
        B, T = x0.shape[0], x0.shape[1]
        units = list(gens[0].num_hidden_rnn)
        if not (T >= 4 and B % 256 == 0 and units == [512, 256] and all(list(g.num_hidden_rnn) == units for g in gens)):
            return False
        return bool(_ops.lstm_cluster_ok(B, 512) and _ops.lstm_resident_ok(B, 256) and (len(gens) * (B // 32)) % 8 == 0)

    def _build_generators(self, mode="eval"):
        from .generators import drive_group
        scale = 1.0 if self.separate_losses else 1.0 / self.num_tracks
        grouped = self._grouped(mode)
        cos = []
        for i in range(self.num_tracks):
            gi, gt = self._generator_io(i)
            g = self.generators[i]
            g.grad_scale = scale                            # mean track loss, one clip over all generators (multinn_jamming.py:235-241)
            if grouped:
                g._materialize(gi.shape[-1])
                g._stack.group_rowpar = True
                cos.append(g._build_co(x=gi, y=gt, lengths=self._lengths, is_train=self._is_train, mode=mode))
            else:
                if getattr(g, "_stack", None) is not None:
                    g._stack.group_rowpar = False
                g.build(x=gi, y=gt, lengths=self._lengths, is_train=self._is_train, mode=mode)
        if grouped:
            drive_group(cos)
        self._built_grouped = grouped

    def _build_generator_outputs(self):
        return [self.generators[i].forward() for i in range(self.num_tracks)]

    def _decode_generator_outputs(self):
        return self._decode_tracks(self._x_hidden)

    def generate(self, num_steps):
        """multinn_jamming.py:101-133 -> u8 `[B, num_steps, P, M]`."""
        if self._x_encoded is None:
            MultINNCore._build_all(self, "generate")
        music = []
        for i in range(self.num_tracks):
            samples_h = self.generators[i].generate(self._x_encoded[i], num_steps)
            music.append(self.encoders[i].decode(samples_h)[1].to(torch.uint8))
        return torch.stack(music, dim=3)

    def pretrain_generators(self, optimizer, lr, separate_losses=False):
        return self._train_generators(optimizer, lr, pretrain=True, separate_losses=separate_losses)

    def train_generators(self, optimizer, lr, separate_losses=False):
        return self._with_global(self._train_generators(optimizer, lr, pretrain=False, separate_losses=separate_losses))

    def _extra_stores(self):
        return []

    def _backward_extra(self):
        """Hook between the generators' backward passes and the joint optimiser step (the feedback modes back-propagate into their module)."""

    def _train_generators(self, optimizer, lr, pretrain=False, separate_losses=False):
        """multinn_jamming.py:186-245.  Per generator: `pretrain` / `train` with run_optimizer = separate_losses; then, whenever NOT
        separate_losses -- pre-training included (:235: the reference replaces the collected update ops by the joint gradient step, so a
        pre-training call without separate losses is the visible-bias init ops plus an ordinary step on the mean track loss) -- ONE clipped
        step on the mean track loss over all generators' (and the feedback module's) variables.  The per-track weight of that mean (1/M) is
        applied to the gradient seed at build time from `self.separate_losses`; an argument that differs from the attribute is honoured by
        rescaling the finished gradients (everything downstream of the seed is linear in it)."""
        M = self.num_tracks
        built_scale = 1.0 if self.separate_losses else 1.0 / M
        want_scale = 1.0 if separate_losses else 1.0 / M
        init_ops, update_ops, tm, tu, ts = [], [], [], [], []
        lockstep = bool(getattr(self, "_built_grouped", False)) and not pretrain and type(self) is MultINNJamming
        if lockstep:                                    # the M backward passes side by side: one launch per layer's backward recurrence
            from .generators import drive_group
            drive_group([g._backward_co() for g in self.generators])
        for i, g in enumerate(self.generators):
            if lockstep:
                io, uo, m, mu, s = [], [], g.metrics, g.metrics_upd, dict(g.summaries)      # what g.train(run_optimizer=False) returns
            elif pretrain:
                io, uo, m, mu, s = g.pretrain(optimizer, lr, run_optimizer=separate_losses)
                if not separate_losses:
                    g.backward()                    # the joint step below differentiates the same batch/loss
            else:
                io, uo, m, mu, s = g.train(optimizer, lr, run_optimizer=False)           # backward only
            if (not pretrain or not separate_losses) and want_scale != built_scale:
                r = want_scale / built_scale
                ops.axpby(r, g.store.grad, 0.0, None, g.store.grad)
                if g._dx is not None:
                    ops.axpby(r, g._dx.view(-1), 0.0, None, g._dx.view(-1))
            if separate_losses and not pretrain:        # generator.py:199-205: each generator's own clipped step
                from .training import compute_gradients
                g._grad_sumsq = compute_gradients(optimizer, g.store, g.clip_norm, lr)
                g._packed_step = -1
            init_ops += io; update_ops += uo
            tm.append(m); tu.append(mu); ts.append(s or {})
        metrics, metrics_upd, summaries = self._combine_track_metrics(tm, tu, ts, global_scope=f"metrics/{self.generators[0].name}/global/")
        if not separate_losses:
            self._backward_extra()
            stores = [g.store for g in self.generators] + self._extra_stores()
            self._grad_sumsq = compute_gradients_multi(optimizer, stores, self.clip_norm, lr)
            for g in self.generators:
                g._packed_step = -1
            update_ops = []                         # multinn_jamming.py:237: the joint step REPLACES the collected update ops (it has run)
        return init_ops, update_ops, metrics, metrics_upd, summaries


class MultINNComposer(MultIEncoderNN):
    """multinn_composer.py: per-track encoders, ONE generator with a shared LSTM and one NADE per track."""

    def __init__(self, config, params, name="MultINN-composer", **kw):
        super().__init__(config, params, name=name, **kw)
        self._mode = "composer"

    def _init_generators(self, generator_class):
        if self.generator_type == "RBM":
            raise NotImplementedError("MultiRNNRBM is not implemented yet :(")                 # multinn_composer.py:44-45
        generators = [RnnMultiNADE(num_dims=self._num_dims_generator, num_hidden=self._params["generator"]["num_hidden"],
                                   num_hidden_rnn=self._params["generator"]["num_hidden_rnn"], tracks=self.tracks,
                                   keep_prob=self.keep_prob, **self._generator_kwargs())]       # multinn_composer.py:49-57
        self._generator = generators[0]
        return generators

    def _build_generators(self, mode="eval"):
        self._x_encoded_stack = self._stack_encoded()                                            # multinn_composer.py:73-87
        self._generator.build(x=self._x_encoded_stack[:, :-1], y=self._x_encoded_stack[:, 1:], lengths=self._lengths,
                              is_train=self._is_train, mode=mode)

    def _build_generator_outputs(self):
        return self._generator.forward()

    def _decode_generator_outputs(self):
        return self._decode_tracks(self._x_hidden)

    def generate(self, num_steps):
        """multinn_composer.py:114-151 -> u8 `[B, num_steps, P, M]`."""
        if self._x_encoded is None:
            MultINNCore._build_all(self, "generate")
        samples_h = self._generator.generate(self._x_encoded_stack, num_steps)
        samples_h = samples_h.reshape(samples_h.shape[0], num_steps, self._num_dims_generator, self.num_tracks).unbind(-1)
        music = [self.encoders[i].decode(samples_h[i].contiguous())[1].to(torch.uint8) for i in range(self.num_tracks)]
        return torch.stack(music, dim=3)

    def pretrain_generators(self, optimizer, lr, separate_losses=False):
        return self._generator.pretrain(optimizer, lr)

    def train_generators(self, optimizer, lr, separate_losses=False):
        return self._with_global(self._generator.train(optimizer, lr))


class MultINNFeedback(MultINNJamming):
    """multinn_feedback.py: jamming + a Dense feedback module over the stacked encodings of the step."""

    def __init__(self, config, params, name="MultINN-feedback", **kw):
        super().__init__(config, params, name=name, **kw)
        self._mode = "feedback"
        self._feedback_module = True
        self._feedback_layer = None
        self._x_encoded_stack = self._x_feedback = self._feedback_final_state = None

    def _init_feedback(self, num_inputs):
        from .feedback import FeedbackDnn
        return FeedbackDnn(num_inputs, self._params["generator"]["feedback"], seed=self.seed + 500, device=self.device)   # :46-52

    def _apply_feedback(self, inputs, initial_state=None, single_step=False, train=False):
        if single_step:
            return self._feedback_layer.single(inputs, initial_state)
        return self._feedback_layer.run(inputs, initial_state, train=train)

    def _build_generators(self, mode="eval"):
        """multinn_feedback.py:54-101.  In train mode the feedback module keeps what its backward needs and every generator is asked for the
        gradient wrt its inputs: the feedback vector is columns [E, E + F) of each generator's input (:85-91), and the mean track loss is
        minimised over the generators' AND the module's variables (multinn_jamming.py:235-241 with trainable_feedback_variables, :97)."""
        train = mode == "train"
        self._x_encoded_stack = self._stack_encoded()
        if self._feedback_layer is None:
            self._feedback_layer = self._init_feedback(self._x_encoded_stack.shape[-1])
        self._feedback_layer.row0 = self._row0
        self._x_feedback, self._feedback_final_state = self._apply_feedback(self._x_encoded_stack, single_step=False, train=train)
        for g in self.generators:
            g.need_dx = train and not self.separate_losses
        super()._build_generators(mode)

    def _extra_stores(self):
        return [] if self._feedback_layer is None else [self._feedback_layer.store]

    def _backward_extra(self):
        """d mean-track-loss / d feedback vectors = the feedback columns of every generator's input gradient, summed over the tracks (each
        generator's backward already carries the 1/M of the mean), zero for the last step (inputs are [:, :-1]); then the module's backward."""
        E, F = self._num_dims_generator, self._x_feedback.shape[-1]
        B, T1, _ = self._x_feedback.shape
        T = T1 - 1
        dev = self._x_feedback.device
        acc = torch.zeros((T1, B, F), device=dev)                    # time-major like the generators' dx
        tmp = torch.empty((T * B, F), device=dev)
        for g in self.generators:
            dx = g._dx                                               # f32 [T, B, E + F]
            ops.convert2d(dx.view(T * B, E + F)[:, E:], tmp)
            flat = acc.view(-1)[:T * B * F]
            ops.axpby(1.0, flat, 1.0, tmp.view(-1), flat)
        nv = getattr(self.generators[0], "_n_valid", None)
        self._feedback_layer.backward(acc.transpose(0, 1).contiguous(), n_valid=nv)   # [B, T+1, F], the row order of the forward call

    def _generator_io(self, i):
        inputs = torch.cat([self._x_encoded[i].float(), self._x_feedback], dim=-1)               # multinn_feedback.py:85-91
        return inputs[:, :-1], self._x_encoded[i][:, 1:]

    def generate(self, num_steps):
        """multinn_feedback.py:120-173 -> u8 `[B, num_steps, P, M]`: one joint scan over the M generators and the feedback module."""
        from .feedback import FeedbackRnnSampler
        if self._x_encoded is None:
            self._inputs = self._build_inputs()
            self._build_encoders("eval")
            self._x_encoded = self._encode_inputs()
        if self._feedback_layer is None:
            self._feedback_layer = self._init_feedback(self._num_dims_generator * self.num_tracks)
        if getattr(self, "_sampler", None) is None:
            self._sampler = FeedbackRnnSampler(self.generators, self._feedback_layer)
        if self._encoder_type == "Pass":
            samples_h = self._sampler.generate(self._x, num_steps)                                # whole scan = one hipGraph replay
        else:
            samples_h = self._sampler.generate_encoded([e.to(torch.uint8) for e in self._x_encoded], num_steps)
        music = [self.encoders[i].decode(samples_h[..., i].contiguous())[1].to(torch.uint8) for i in range(self.num_tracks)]
        return torch.stack(music, dim=3)

    @property
    def trainable_feedback_variables(self):
        fl = self._feedback_layer
        return [] if fl is None else [fl.store[n] for n in fl.store.names()]


class MultINNFeedbackRnn(MultINNFeedback):
    """multinn_feedback_rnn.py: the feedback module is an RNN over the stacked encodings (keeps history)."""

    def __init__(self, config, params, name="MultINN-feedback-rnn", **kw):
        super().__init__(config, params, name=name, **kw)
        self._mode = "feedback-rnn"

    def _init_feedback(self, num_inputs):
        from .feedback import FeedbackRnn
        return FeedbackRnn(num_inputs, self._params["generator"]["feedback"], keep_prob=self.keep_prob, precision=self.precision,
                           seed=self.seed + 500, device=self.device)                              # multinn_feedback_rnn.py:30-39


# ================================================================================================
class MultINN:
    """multinn.py:9-299: facade over the mode classes; every attribute and method is the chosen model's."""

    _MODES = {"joint": MultINNJoint, "composer": MultINNComposer, "jamming": MultINNJamming, "feedback": MultINNFeedback,
              "feedback-rnn": MultINNFeedbackRnn}

    def __init__(self, config, params, mode="feedback-rnn", name="MultINN", **kw):
        if mode not in self._MODES:
            raise ValueError("Incorrect operation mode, choose from `joint`, `composer`, `jamming`, `feedback`, and `feddback-rnn`.")
        self._model = self._MODES[mode](config, params, name=name, **kw)

    def __getattr__(self, k):
        return getattr(self.__dict__["_model"], k)

    def __setattr__(self, k, v):
        if k == "_model":
            self.__dict__[k] = v
        else:
            setattr(self._model, k, v)
