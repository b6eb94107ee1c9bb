"""Second, independent restatement: torch-CPU float32 autograd, op-for-op with the TF graph.

TEST INFRASTRUCTURE ONLY.  Same reference files as oracle/generators.py, but written
against torch autograd so that (a) the hand-derived backward formulas of the NumPy
oracle are cross-checked and (b) ``bench.py``'s ``cpu_baseline`` leg has a CPU "port"
of the reference formulation to time: per-timestep
LSTMBlockCell steps, Dense, the per-visible NADE loop (nade.py:225-226) in row chunks,
autograd backward, global-norm clip 5.0 and TF-style Adam.
"""
import math

import torch


def lstm_cell(x, c, h, W, b):
    """tf LSTMBlockCell, gate order i,ci,f,o, forget_bias 0 (rnn.py:124)."""
    z = torch.cat([x, h], 1) @ W + b
    i, g, f, o = z.chunk(4, 1)
    c2 = torch.tanh(g) * torch.sigmoid(i) + c * torch.sigmoid(f)
    return torch.tanh(c2) * torch.sigmoid(o), c2


def lstm_seq(x, layers, keep_prob=1.0, drop_u=None):
    """x[B,T,in]; returns y[B,T,u_last] (dropped outputs) and final state."""
    B, T, _ = x.shape
    state = [(x.new_zeros(B, W.shape[1] // 4), x.new_zeros(B, W.shape[1] // 4)) for W, _ in layers]
    ys = []
    for t in range(T):
        inp = x[:, t]
        for l, (W, b) in enumerate(layers):
            c, h = state[l]
            h2, c2 = lstm_cell(inp, c, h, W, b)
            state[l] = (c2, h2)
            if keep_prob < 1.0:
                keep = torch.floor(torch.tensor(keep_prob, dtype=torch.float32) + drop_u[l][:, t]).to(h2.dtype)
                inp = h2 / keep_prob * keep
            else:
                inp = h2
        ys.append(inp)
    return torch.stack(ys, 1), state


def nade_log_prob(v, b_enc, b_dec, w_enc, w_dec):
    """nade.py:155-229 as a Python loop over the visible order."""
    a = b_enc
    logp = v.new_zeros(v.shape[0])
    cond = []
    for i in range(v.shape[1]):
        h = torch.sigmoid(a)
        p = torch.sigmoid(b_dec[:, i] + h @ w_dec[i])
        vi = v[:, i]
        logp = logp + vi * torch.log(1e-6 + p) + (1 - vi) * torch.log(1e-6 + 1 - p)
        cond.append(p)
        a = a + vi[:, None] * w_enc[i][None, :]
    return -logp, torch.stack(cond, 1)


def flatten(t, lengths):
    if lengths is None:
        return t.reshape((-1,) + tuple(t.shape[2:]))
    m = torch.arange(t.shape[1])[None, :] < torch.as_tensor(lengths)[:, None]
    return t[m]


def rnn_nade_loss(inputs, targets, lengths, P, keep_prob=1.0, drop_u=None, tracks=1, row_chunk=None):
    """P: dict of torch tensors mirroring oracle.generators.init_rnn_nade."""
    y, _ = lstm_seq(inputs, P['lstm'], keep_prob, drop_u)
    yf = flatten(y, lengths)
    out = yf @ P['fc_k'] + P['fc_b']
    Hn, D = P['w_enc'][0].shape[1], P['w_enc'][0].shape[0]
    tf_ = flatten(targets, lengths)
    losses, nlls, conds = [], [], []
    for m in range(tracks):
        b_enc = out[:, m * Hn:(m + 1) * Hn]
        b_dec = out[:, tracks * Hn + m * D:tracks * Hn + (m + 1) * D]
        tgt = tf_ if tracks == 1 else tf_.reshape(-1, D, tracks)[..., m]
        if row_chunk is None:
            nll, cond = nade_log_prob(tgt, b_enc, b_dec, P['w_enc'][m], P['w_dec'][m])
        else:
            parts = [nade_log_prob(tgt[s:s + row_chunk], b_enc[s:s + row_chunk], b_dec[s:s + row_chunk],
                                   P['w_enc'][m], P['w_dec'][m]) for s in range(0, tgt.shape[0], row_chunk)]
            nll = torch.cat([a for a, _ in parts])
            cond = torch.cat([c for _, c in parts])
        losses.append(nll.mean())
        nlls.append(nll)
        conds.append(cond)
    return torch.stack(losses).mean(), nlls, conds


def to_torch(p, requires_grad=True, dtype=torch.float32):
    def cv(a):
        return torch.tensor(a, dtype=dtype, requires_grad=requires_grad)
    P = dict(lstm=[(cv(W), cv(b)) for W, b in p['lstm']])
    for k, v in p.items():
        if k == 'lstm':
            continue
        P[k] = [cv(a) for a in v] if isinstance(v, list) else cv(v)
    return P


def flat_params(P):
    out = []
    for W, b in P['lstm']:
        out += [W, b]
    for m in range(len(P['w_enc'])):
        out += [P['w_enc'][m], P['w_dec'][m]]
    return out + [P['fc_k'], P['fc_b']]


class TFAdam:
    """tf.train.AdamOptimizer(lr, eps=1e-4) with clip_by_global_norm(5.0) in front
    (train.py:64, utils/training.py:163-175)."""

    def __init__(self, params, lr=0.01, b1=0.9, b2=0.999, eps=1e-4, clip=5.0):
        self.params, self.lr, self.b1, self.b2, self.eps, self.clip = params, lr, b1, b2, eps, clip
        self.t = 0
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]

    @torch.no_grad()
    def step(self):
        gn = math.sqrt(sum(float((p.grad.double() ** 2).sum()) for p in self.params))
        scale = self.clip * min(1.0 / gn, 1.0 / self.clip) if gn > 0 else 1.0
        self.t += 1
        lr_t = self.lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        for p, m, v in zip(self.params, self.m, self.v):
            g = p.grad * scale
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            p.sub_(lr_t * m / (v.sqrt() + self.eps))
            p.grad = None
        return gn


def rbm_free_energy(v, W, bh, bv):
    return -torch.nn.functional.softplus(v @ W + bh).sum(1) - (v * bv).sum(1)


def rnn_rbm_loss(inputs, targets, v_sample, P):
    """Conditional-bias free-energy cost with the Gibbs sample supplied (constant)."""
    y, _ = lstm_seq(inputs, P['lstm'])
    yf = flatten(y, None)
    bh_t = P['bh'] + yf @ P['Wuh']
    bv_t = P['bv'] + yf @ P['Wuv']
    tgt = flatten(targets, None)
    cost = rbm_free_energy(tgt, P['W'], bh_t, bv_t) - rbm_free_energy(v_sample, P['W'], bh_t, bv_t)
    return cost.mean()
