"""Host-side mirror of /root/reference/multinn/models/encoders: Encoder, PassEncoder, DBNEncoder."""
import abc

import torch

from .common import Model, DBN, STREAM_DBN_ENC, STREAM_DBN_DEC


class Encoder(Model):
    """models/encoders/encoder.py:8-166."""

    def __init__(self, num_dims, num_hidden, name="encoder", track_name="all"):
        super().__init__(name=name)
        self._num_dims = num_dims
        self._num_hidden = [num_hidden] if isinstance(num_hidden, int) else list(num_hidden)
        self._track_name = track_name
        self._lengths = None
        self._enc_probs, self._encodings, self._dec_probs, self._decodings = [], [], [], []     # per layer (encoder.py:33-36)

    num_dims = property(lambda self: self._num_dims)
    num_hidden = property(lambda self: self._num_hidden)
    num_layers = property(lambda self: len(self._num_hidden))
    track_name = property(lambda self: self._track_name)
    encodings = property(lambda self: self._encodings)
    decodings = property(lambda self: self._decodings)
    enc_probs = property(lambda self: self._enc_probs)
    dec_probs = property(lambda self: self._dec_probs)

    @abc.abstractmethod
    def encode(self, x=None):
        ...

    @abc.abstractmethod
    def decode(self, h=None):
        ...

    def train(self, optimizer, lr, layer=0):
        return [], [], self.metrics, self.metrics_upd, self.summaries


class PassEncoder(Encoder):
    """models/encoders/pass_encoder.py: identity encoder."""

    def __init__(self, num_dims, num_hidden=None, name="pass-encoder", track_name="all"):
        """pass_encoder.py:12-29: `num_hidden` is accepted and unused (the encoder's hidden width IS `num_dims`)."""
        super().__init__(num_dims, num_dims, name=name, track_name=track_name)

    def build(self, x=None, y=None, lengths=None, is_train=None, mode="eval"):
        super().build(x, y, lengths, is_train, mode)
        self._inputs = x
        self._enc_probs, self._encodings = [x], [x]         # pass_encoder.py:52-53: per-layer lists of one entry
        self._dec_probs, self._decodings = [x], [x]
        self._is_built = True

    def encode(self, x=None):
        """pass_encoder.py:99-112."""
        if x is None:
            return self._enc_probs[-1], self._encodings[-1]
        return x, x

    def decode(self, h=None):
        """pass_encoder.py:114-127."""
        if h is None:
            return self._dec_probs[-1], self._decodings[-1]
        return h, h

    def train(self, optimizer, lr, layer=0):
        """pass_encoder.py:129-136: nothing to train."""
        return [], [], {}, [], {}

    def build_metrics(self, targets, predictions, cond_probs=None, log_probs=None, layer=None):
        """pass_encoder.py:56-97: 'reconstruction cost' = sum_d tf.losses.log_loss(targets, cond_probs)."""
        from .metrics import base_metrics
        if log_probs is None and cond_probs is not None:
            log_probs = reconstruction_cost(targets, cond_probs)
        else:                                               # pass_encoder.py:86-88 (raised also when log_probs IS given, as written)
            raise ValueError("Incorrect arguments. Either `cond_probs`, or `log_probs` should be provided on `encoder.build_metrics()` "
                             "function call")
        return base_metrics(log_probs, targets, predictions, log_probs)


def reconstruction_cost(targets, cond_probs):
    """sum_d tf.losses.log_loss(targets, cond_probs) per row (epsilon 1e-7): mnn_log_loss_rows on the device."""
    from . import ops
    t2 = targets.reshape(-1, targets.shape[-1])
    p2 = cond_probs.reshape(-1, cond_probs.shape[-1]).float().contiguous()
    out = torch.empty(t2.shape[0], device=p2.device)
    ops.log_loss_rows((t2 != 0).to(torch.uint8).contiguous(), p2, out)
    return out.reshape(targets.shape[:-1])


class DBNEncoder(Encoder):
    """models/encoders/dbn_encoder.py: DBN encode/decode with SAMPLED binary codes (52-106, 136-190).

    `encodings` / `enc_probs` / `decodings` / `dec_probs` are per-layer lists as in the reference (encoder.py:33-36,
    dbn_encoder.py:83-97): entry i belongs to RBM layer i, `encodings[-1]` is the code the generators consume and
    `decodings[0]` the reconstruction in input space."""

    def __init__(self, num_dims, num_hidden, k=2, name="dbn-encoder", track_name="all", seed=23, device=None):
        super().__init__(num_dims, num_hidden, name=name, track_name=track_name)
        self._dbn = DBN(num_dims, self._num_hidden, k=k, name=f"{name}/{track_name}", seed=seed, device=device)
        self.store = self._dbn.store
        self.seed, self.row0, self._sub = seed, 0, 0
        self._variables = dict(self.store.views)
        self._trainable_variables = [self.store[n] for n in self.store.names()]
        self._lengths = None
        self._dec_pending = None

    dbn = property(lambda self: self._dbn)

    def build(self, x=None, y=None, lengths=None, is_train=None, mode="eval"):
        """dbn_encoder.py:52-106: forward pass through every layer (sampled codes upward), then the reconstruction
        pass back down, every layer's probabilities and samples kept.  The reference BUILDS both passes into its graph and executes the
        reconstruction only when something fetches it (a generator's train step never does); here the downward pass runs ON DEMAND -- at the
        first read of `decodings` / `dec_probs` / `decode()` / `metrics` -- with the RNG counters of this build, so its draws are the ones an
        eager pass would have made (the composer's train step spent 1.4 of its 13.9 ms on reconstructions nothing read)."""
        super().build(x, y, lengths, is_train, mode)
        self._inputs, self._lengths = x, lengths
        self._enc_probs, self._encodings, self._dec_probs, self._decodings = [], [], [], []
        self._dec_pending = None
        if x is not None and mode in ("train", "eval"):
            f, lead = self._flat(x.to(torch.uint8) if x.dtype != torch.uint8 else x)
            h = f
            for i, r in enumerate(self._dbn.rbms):                      # same counters as DBN.forward: encode(x) == encodings[-1]
                p, h = r.forward(h, None, self.seed, self.row0, (self._sub << 4) | i, STREAM_DBN_ENC)
                self._enc_probs.append(p.view(*lead, -1)); self._encodings.append(h.view(*lead, -1))
            self._dec_pending = (h, lead, self.seed, self.row0, self._sub)
        self._metrics = self._metrics_upd = None             # dbn_encoder.py:98-104: built on demand (see `metrics`)
        self._is_built = True

    def _ensure_decoded(self):
        """The reconstruction pass of the last build (see `build`), run once, with that build's seed / row / sub-stream counters."""
        if self._dec_pending is not None:
            v, lead, seed, row0, sub = self._dec_pending
            self._dec_pending = None
            for i in range(len(self._dbn.rbms) - 1, -1, -1):
                p, v = self._dbn.rbms[i].reconstruct(v, None, seed, row0, (sub << 4) | i, STREAM_DBN_DEC)
                self._dec_probs.insert(0, p.view(*lead, -1)); self._decodings.insert(0, v.view(*lead, -1))

    decodings = property(lambda self: (self._ensure_decoded(), self._decodings)[1])
    dec_probs = property(lambda self: (self._ensure_decoded(), self._dec_probs)[1])

    def _ensure_metrics(self):
        if self._metrics is None and self._inputs is not None:
            self._ensure_decoded()
            if self._decodings:
                self._metrics, self._metrics_upd, self._summaries["metrics"] = self.build_metrics(
                    targets=self._inputs, predictions=self._decodings[0], cond_probs=self._dec_probs[0])
        return self._metrics

    metrics = property(lambda self: self._ensure_metrics())
    metrics_upd = property(lambda self: (self._ensure_metrics(), self._metrics_upd)[1])

    def _flat(self, x):
        return x.reshape(-1, x.shape[-1]).contiguous(), x.shape[:-1]

    def encode(self, x=None):
        """dbn_encoder.py:136-162: [B,T,P] -> (p_h, h) with h sampled; the zero-padded step is encoded too."""
        if x is None:
            return self._enc_probs[-1], self._encodings[-1]
        f, lead = self._flat(x.to(torch.uint8) if x.dtype != torch.uint8 else x)
        p, h = self._dbn.forward(f, self.seed, self.row0, self._sub)
        return p.view(*lead, -1), h.view(*lead, -1)

    def decode(self, h=None):
        """dbn_encoder.py:164-190."""
        if h is None:
            self._ensure_decoded()
            return self._dec_probs[0], self._decodings[0]
        f, lead = self._flat(h.to(torch.uint8) if h.dtype != torch.uint8 else h)
        p, v = self._dbn.reconstruct(f, self.seed, self.row0, self._sub)
        return p.view(*lead, -1), v.view(*lead, -1)

    def build_metrics(self, targets, predictions, cond_probs=None, log_probs=None, layer=None):
        return self._dbn.build_metrics(targets.reshape(-1, targets.shape[-1]), predictions.reshape(-1, predictions.shape[-1]),
                                       cond_probs.reshape(-1, cond_probs.shape[-1]) if cond_probs is not None else None, log_probs)

    def train(self, optimizer, lr, layer=0):
        """dbn_encoder.py:192-240: one CD-k update of RBM `layer` on its inputs (the batch for layer 0, the sampled codes of
        layer-1 otherwise), flattened by `lengths`; metrics are that RBM's own reconstruction of its inputs (R7: the
        reference reads a `track_name` the RBM lacks -- nothing here depends on it)."""
        assert 0 <= layer < self.num_layers                 # dbn_encoder.py:208
        x = self._inputs if layer == 0 else self._encodings[layer - 1]
        f, _ = self._flat(x.to(torch.uint8) if x.dtype != torch.uint8 else x)
        if self._lengths is not None:                        # flatten_maybe_padded_sequences: b-major rows with t < lengths[b]
            B, T = x.shape[0], x.shape[1]
            m = torch.arange(T, device=x.device)[None, :] < self._lengths.to(x.device)[:, None]
            f = f[m.reshape(-1)].contiguous()
        rbm = self._dbn.rbms[layer]
        rbm.seed = self.seed
        self._sub += 1
        sub0 = self._sub * 64
        _, enc = rbm.forward(f, None, self.seed, self.row0, sub0 + 62)
        dec_p, dec = rbm.reconstruct(enc, None, self.seed, self.row0, sub0 + 62)
        metrics, metrics_upd, _ = rbm.build_metrics(targets=f, predictions=dec, cond_probs=dec_p)
        init_ops, update_ops, grads = rbm.train(f, lr, row0=self.row0, sub0=sub0)
        return init_ops, update_ops, metrics, metrics_upd, self.summaries
