"""Host-side mirror of /root/reference/multinn/models/encoders: Encoder, PassEncoder, DBNEncoder."""
import abc

import torch

from .common import Model, DBN


class Encoder(Model):
    """models/encoders/encoder.py:8-166."""

    def __init__(self, num_dims, num_hidden, name="encoder", track_name="all"):
        super().__init__(name=name)
        self._num_dims = num_dims
        self._num_hidden = [num_hidden] if isinstance(num_hidden, int) else list(num_hidden)
        self._track_name = track_name
        self._encodings = self._decodings = self._enc_probs = self._dec_probs = None

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

    def __init__(self, num_dims, name="pass-encoder", track_name="all"):
        super().__init__(num_dims, num_dims, name=name, track_name=track_name)

    def build(self, x=None, y=None, lengths=None, is_train=None, mode="eval"):
        super().build(x, y, lengths, is_train, mode)
        self._inputs = x
        if x is not None:
            self._enc_probs, self._encodings = self.encode(x)
            self._dec_probs, self._decodings = self.decode(self._encodings)
        self._is_built = True

    def encode(self, x=None):
        x = self._inputs if x is None else x
        return x, x

    def decode(self, h=None):
        h = self._encodings if h is None else h
        return h, h

    def build_metrics(self, targets, predictions, cond_probs=None, log_probs=None):
        """pass_encoder.py:56-97: 'reconstruction cost' = sum_d tf.losses.log_loss(targets, cond_probs)."""
        from .metrics import base_metrics
        lp = reconstruction_cost(targets, cond_probs)
        return base_metrics(lp, targets, predictions, lp)


def reconstruction_cost(targets, cond_probs):
    """sum_d tf.losses.log_loss(targets, cond_probs) per row (epsilon 1e-7): mnn_log_loss_rows on the device."""
    from . import ops
    t2 = targets.reshape(-1, targets.shape[-1])
    p2 = cond_probs.reshape(-1, cond_probs.shape[-1]).float().contiguous()
    out = torch.empty(t2.shape[0], device=p2.device)
    ops.log_loss_rows((t2 != 0).to(torch.uint8).contiguous(), p2, out)
    return out.reshape(targets.shape[:-1])


class DBNEncoder(Encoder):
    """models/encoders/dbn_encoder.py: DBN encode/decode with SAMPLED binary codes (52-106, 136-190)."""

    def __init__(self, num_dims, num_hidden, k=10, name="dbn-encoder", track_name="all", seed=23, device=None):
        super().__init__(num_dims, num_hidden, name=name, track_name=track_name)
        self._dbn = DBN(num_dims, self._num_hidden, k=k, name=f"{name}/{track_name}", seed=seed, device=device)
        self.store = self._dbn.store
        self.seed, self.row0, self._sub = seed, 0, 0

    dbn = property(lambda self: self._dbn)

    def build(self, x=None, y=None, lengths=None, is_train=None, mode="eval"):
        super().build(x, y, lengths, is_train, mode)
        self._inputs = x
        if x is not None:
            self._enc_probs, self._encodings = self.encode(x)
            self._dec_probs, self._decodings = self.decode(self._encodings)
        self._metrics = self._metrics_upd = None             # dbn_encoder.py:98-104: built on demand (see `metrics`)
        self._is_built = True

    def _ensure_metrics(self):
        if self._metrics is None and self._inputs is not None:
            self._metrics, self._metrics_upd, self._summaries["metrics"] = self.build_metrics(
                targets=self._inputs, predictions=self._decodings, cond_probs=self._dec_probs)
        return self._metrics

    metrics = property(lambda self: self._ensure_metrics())
    metrics_upd = property(lambda self: (self._ensure_metrics(), self._metrics_upd)[1])

    def _flat(self, x):
        return x.reshape(-1, x.shape[-1]).contiguous(), x.shape[:-1]

    def encode(self, x=None):
        """dbn_encoder.py:136-162: [B,T,P] -> (p_h, h) with h sampled; the zero-padded step is encoded too."""
        x = self._inputs if x is None else x
        f, lead = self._flat(x.to(torch.uint8) if x.dtype != torch.uint8 else x)
        p, h = self._dbn.forward(f, self.seed, self.row0, self._sub)
        return p.view(*lead, -1), h.view(*lead, -1)

    def decode(self, h=None):
        """dbn_encoder.py:164-190."""
        h = self._encodings if h is None else h
        f, lead = self._flat(h.to(torch.uint8) if h.dtype != torch.uint8 else h)
        p, v = self._dbn.reconstruct(f, self.seed, self.row0, self._sub)
        return p.view(*lead, -1), v.view(*lead, -1)

    def build_metrics(self, targets, predictions, cond_probs=None, log_probs=None):
        return self._dbn.build_metrics(targets.reshape(-1, targets.shape[-1]), predictions.reshape(-1, predictions.shape[-1]),
                                       cond_probs.reshape(-1, cond_probs.shape[-1]) if cond_probs is not None else None, log_probs)

    def train(self, optimizer, lr, layer=0):
        """dbn_encoder.py:192-240: CD-k on RBM ``layer`` fed with sampled codes of the layers below (R7 n/a)."""
        assert 0 <= layer < self.num_layers                 # dbn_encoder.py:208
        f, _ = self._flat(self._inputs.to(torch.uint8))
        for i in range(layer):
            _, f = self._dbn.rbms[i].forward(f, None, self.seed, self.row0, (self._sub << 4) | i, 4)
        rbm = self._dbn.rbms[layer]
        rbm.seed = self.seed
        self._sub += 1
        init_ops, update_ops, grads = rbm.train(f, lr, row0=self.row0, sub0=self._sub * 64)
        return init_ops, update_ops, self.metrics, self.metrics_upd, self.summaries
